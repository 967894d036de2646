// Probe (round 5): issue rate of v_mfma_f32_32x32x16_f16 on ONE wave per SIMD as a function of how many
// independent accumulators are interleaved (K = 1, 2, 3, 4, 6, 8) and of what sits between the MFMAs
// (nothing, one s_nop 1 per K MFMAs, one ds_read_b128 per MFMA).  Prints cycles per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF "v_mfma_f32_32x32x16_f16 "
template <int K, int MODE>
__global__ __launch_bounds__(256, 1) void k(const f32x4* in, float* out, unsigned long long* t, int iters) {
  __shared__ f32x4 lds[2048];
  for (int i = 0; i < 8; ++i) lds[threadIdx.x + 256 * i] = in[threadIdx.x];
  __syncthreads();
  f32x4 a = in[threadIdx.x], b = in[256 + threadIdx.x];
  f32x16 c[8];
  for (int i = 0; i < 8; ++i) c[i] = f32x16{0};
  const f32x4* lp = lds + (threadIdx.x & 63);
  float v[4] = {1.f, 2.f, 3.f, 4.f}, w = in[512 + threadIdx.x][0];
  f32x4 pf[8];
  for (int i = 0; i < 8; ++i) pf[i] = lp[64 * i];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 24 / K; ++r) {
      if (MODE == 1) asm volatile("s_nop 1");
#pragma unroll
      for (int i = 0; i < K; ++i) {
        if (MODE == 2) { f32x4 x = lp[64 * ((r * K + i) & 7)]; asm volatile("" : "+v"(x)); a = x; }
        if (MODE >= 30) asm volatile(MF "%0, %1, %2, %0" : "+a"(c[i]) : "v"(pf[(r * K + i + 4) & 7]), "v"(b));
        else asm volatile(MF "%0, %1, %2, %0" : "+a"(c[i]) : "v"(a), "v"(b));
        if (MODE >= 10 && MODE < 30) {          // MODE - 10 independent plain VALU after every MFMA (asm: order pinned)
#pragma unroll
          for (int f = 0; f < (MODE >= 20 ? MODE - 20 : MODE - 10); ++f) {
            if (MODE >= 20) v[f & 3] = __builtin_fmaf(v[f & 3], 1.0001f, 0.5f);   // compiler-scheduled VALU
            else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[f & 3]) : "v"(w));
          }
        }
        if (MODE >= 30) {                          // prefetched LDS reads: one ds_read_b128 per MFMA, consumed K MFMAs later
          pf[(r * K + i) & 7] = lp[64 * ((r * K + i) & 7) + 512 * (it & 1)];
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c[i][0];
  s += v[0] + v[1] + v[2] + v[3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int K, int MODE>
void run(const f32x4* in, float* out, unsigned long long* t, int nwg) {
  const int iters = 200;
  k<K, MODE><<<nwg, 256>>>(in, out, t, iters);
  hipDeviceSynchronize();
  unsigned long long h;
  hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("K=%d mode=%d nwg=%d : %.1f cycles/MFMA\n", K, MODE, nwg, (double)h / (iters * (24 / K) * K));
}
int main() {
  f32x4* in; float* out; unsigned long long* t;
  hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 22); hipMalloc(&t, 1 << 16);
  hipMemset(in, 0, 1 << 20);
  for (int nwg : {1, 256}) {
    run<4, 0>(in, out, t, nwg);
    run<4, 11>(in, out, t, nwg); run<4, 12>(in, out, t, nwg); run<4, 13>(in, out, t, nwg); run<4, 14>(in, out, t, nwg);
    run<4, 15>(in, out, t, nwg); run<4, 16>(in, out, t, nwg); run<4, 18>(in, out, t, nwg);
    run<4, 22>(in, out, t, nwg); run<4, 24>(in, out, t, nwg); run<4, 26>(in, out, t, nwg); run<4, 28>(in, out, t, nwg);
    run<4, 30>(in, out, t, nwg); run<8, 30>(in, out, t, nwg);
  }
  return 0;
}
