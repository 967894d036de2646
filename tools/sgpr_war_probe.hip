// Probe (gfx950, ROCm 7.2): may an SALU instruction overwrite an SGPR that a VALU instruction issued
// just before it reads as a scalar operand?  (WAR on SGPRs: hipcc inserts nothing, and for plain
// VALU nothing is needed - the operand is read at issue.  The question is the PACKED fp32 ops,
// v_pk_fma_f32 / v_pk_mul_f32, which take more than one pass per wave on this SIMD.)
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/sgpr_war_probe.hip && /tmp/probe
//
// Shape taken from the tail of the split-f16 KV state of k_encoder64:
//     8 x v_pk_fma_f32 v[..], v[..], s[2:3], v[..] op_sel_hi:[1,0,1]   (kv = main + cross * 2^-11)
//     ds_bpermute_b32 ... ; s_lshl_b64 s[2:3], s[88:89], 15             (address arithmetic reuses s[2:3])
// A victim wave (s_setprio 3) runs   s2 <- 1.0 ; DEPTH x v_pk_fma (x * s2 + 0) ; GAP ; s2 <- 0.0
// and every result must equal x = 3.0; a lane that reads the overwritten scalar returns 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define PK(i, j) "v_pk_fma_f32 v[" #i ":" #j "], v[230:231], s[40:41], v[232:233] op_sel_hi:[1,0,1]\n"
#define FMA(i) "v_fma_f32 v" #i ", v230, s40, v232\n"
#define G0 ""
#define G1 "s_nop 0\n"
#define G2 "s_nop 1\n"
#define G4 "s_nop 3\n"
#define GDS "ds_bpermute_b32 v240, v241, v242\n"

#define VICTIM(BODY, GAP, CLOBBER)                                                                  \
  asm volatile(                                                                                     \
      "s_setprio 3\n s_mov_b32 s42, %[iters]\n"                                                     \
      "v_mov_b32 v230, 3.0\n v_mov_b32 v231, 3.0\n v_mov_b32 v232, 0\n v_mov_b32 v233, 0\n"           \
      "v_mov_b32 v241, 0\n v_mov_b32 v242, 0\n v_mov_b32 v250, 0\n"                                  \
      "1:\n"                                                                                        \
      "s_mov_b32 s40, 1.0\n s_mov_b32 s41, 1.0\n s_nop 3\n"                                          \
      BODY GAP CLOBBER                                                                              \
      "s_nop 7\n"                                                                                   \
      /* count the results that are not 3.0 (v200..v215 as written by BODY; unwritten ones hold 3.0) */ \
      "v_cmp_neq_f32 vcc, 3.0, v200\n v_addc_co_u32 v250, vcc, 0, v250, vcc\n"                       \
      "v_cmp_neq_f32 vcc, 3.0, v201\n v_addc_co_u32 v250, vcc, 0, v250, vcc\n"                       \
      "v_cmp_neq_f32 vcc, 3.0, v214\n v_addc_co_u32 v250, vcc, 0, v250, vcc\n"                       \
      "v_cmp_neq_f32 vcc, 3.0, v215\n v_addc_co_u32 v250, vcc, 0, v250, vcc\n"                       \
      "s_sub_u32 s42, s42, 1\n s_cmp_lg_u32 s42, 0\n s_cbranch_scc1 1b\n"                            \
      "s_setprio 0\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %[o], v250\n"                                  \
      : [o] "=v"(out)                                                                               \
      : [iters] "s"(iters)                                                                          \
      : "s40", "s41", "s42", "s43", "scc", "vcc", "v200", "v201", "v202", "v203", "v204", "v205", "v206",   \
        "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v230", "v231", "v232",    \
        "v233", "v240", "v241", "v242", "v250", "memory")

#define INIT "v_mov_b32 v200, 3.0\n v_mov_b32 v201, 3.0\n v_mov_b32 v214, 3.0\n v_mov_b32 v215, 3.0\n"
#define PK8 INIT PK(200, 201) PK(202, 203) PK(204, 205) PK(206, 207) PK(208, 209) PK(210, 211) PK(212, 213) PK(214, 215)
#define PK1 INIT PK(214, 215)
#define FMA8 INIT FMA(200) FMA(201) FMA(202) FMA(203) FMA(204) FMA(205) FMA(214) FMA(215)
#define CLOB_MOV "s_mov_b32 s40, 0\n"
#define CLOB_SHL "s_lshl_b64 s[40:41], s[42:43], 15\n s_mov_b32 s40, 0\n"

__global__ __launch_bounds__(512) void k_probe(int variant, int sibling, int iters, int* res) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int out = -1;
  if (wave >= 4) {
    if (sibling == 1) {   // the other wave of the SIMD streams packed VALU work too
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 a = {1.f, 2.f}, b = {1.0001f, 0.9999f};
      for (int i = 0; i < iters * 16; ++i) a = __builtin_elementwise_fma(a, b, b);
      if (a[0] == 123.f) res[0] = 1;
    }
    return;
  }
  switch (variant) {
    case 0: VICTIM(PK8, G0, CLOB_MOV); break;
    case 1: VICTIM(PK8, G1, CLOB_MOV); break;
    case 2: VICTIM(PK8, G2, CLOB_MOV); break;
    case 3: VICTIM(PK8, G4, CLOB_MOV); break;
    case 4: VICTIM(PK8, GDS, CLOB_SHL); break;     // the kernel's own sequence
    case 5: VICTIM(PK1, G0, CLOB_MOV); break;
    case 6: VICTIM(FMA8, G0, CLOB_MOV); break;     // control: plain fp32 FMA with the same scalar operand
    case 7: VICTIM(PK8, G0, ""); break;            // control: no overwrite
  }
  res[(blockIdx.x * 4 + wave) * 64 + lane] = out;
}

int main() {
  const int iters = 20000, blocks = 256;
  int* d;
  (void)hipMalloc(&d, blocks * 256 * sizeof(int));
  std::vector<int> h(blocks * 256);
  const char* what[8] = {"8 x v_pk_fma, s_mov right behind            ", "8 x v_pk_fma, 1 state, s_mov                ",
                         "8 x v_pk_fma, 2 states, s_mov               ", "8 x v_pk_fma, 4 states, s_mov               ",
                         "8 x v_pk_fma, ds_bpermute, s_lshl_b64 (kernel)", "1 x v_pk_fma, s_mov right behind            ",
                         "8 x v_fma_f32 (control), s_mov right behind ", "8 x v_pk_fma, no overwrite (control)        "};
  for (int sib = 0; sib < 2; ++sib)
    for (int v = 0; v < 8; ++v) {
      (void)hipMemset(d, 0, blocks * 256 * sizeof(int));
      hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(512), 0, 0, v, sib, iters, d);
      (void)hipMemcpy(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost);
      long bad = 0, q[4] = {0, 0, 0, 0}, tot = 0;
      for (size_t i = 0; i < h.size(); ++i)
        if (h[i] != 0) { ++bad; ++q[(i & 63) >> 4]; tot += h[i]; }
      printf("%s sibling %s : %6ld of %zu lanes saw the overwritten scalar (lanes 0-15 %ld, 16-31 %ld, 32-47 %ld, 48-63 %ld), %ld of %.0f results\n",
             what[v], sib ? "packed VALU" : "idle       ", bad, h.size(), q[0], q[1], q[2], q[3], tot, 4.0 * h.size() * iters);
    }
  (void)hipFree(d);
  return 0;
}
