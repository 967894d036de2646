// Does v_mfma_f32_32x32x16_f16 honour f16 DENORMAL inputs on gfx950, or flush them to zero?
// A = one denormal value d in every slot, B = 1.0 in every slot: every output = 16 * d if honoured, 0 if flushed.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/mfma_denorm_probe.hip -o tools/bin/mfma_denorm_probe && tools/bin/mfma_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const uint16_t* abits, const uint16_t* bbits, float* out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(_Float16, abits[0]);
    b[i] = __builtin_bit_cast(_Float16, bbits[0]);
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  uint16_t *da, *db; float* dout;
  hipMalloc(&da, 2); hipMalloc(&db, 2); hipMalloc(&dout, 4);
  struct { uint16_t a, b; const char* what; double expect; } cases[] = {
    {0x0001, 0x3c00, "A = 2^-24 (smallest denormal), B = 1", 16 * 5.9604644775390625e-08},
    {0x0200, 0x3c00, "A = 2^-15 (denormal), B = 1", 16 * 3.0517578125e-05},
    {0x3c00, 0x0200, "A = 1, B = 2^-15 (denormal)", 16 * 3.0517578125e-05},
    {0x0200, 0x6400, "A = 2^-15 (denormal), B = 1024", 16 * 3.0517578125e-05 * 1024},
    {0x0400, 0x3c00, "A = 2^-14 (smallest normal), B = 1", 16 * 6.103515625e-05},
  };
  int bad = 0;
  for (auto& c : cases) {
    hipMemcpy(da, &c.a, 2, hipMemcpyHostToDevice);
    hipMemcpy(db, &c.b, 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    float r; hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
    printf("%-40s -> %.9g (honoured: %.9g)%s\n", c.what, r, c.expect, r == (float)c.expect ? "" : "   <-- differs");
    bad += r != (float)c.expect;
  }
  printf(bad ? "f16 denormal inputs are NOT all honoured\n" : "f16 denormal inputs are honoured (no flush)\n");
  return 0;
}
