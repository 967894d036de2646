// Reproducer (gfx950, ROCm 7.2): how many WAIT STATES does a VALU read of a
// v_mfma_f32_32x32x16_f16 result need - and do branch instructions provide them?
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/mfma_branch_hazard_probe.hip && /tmp/probe
//
// Background: hipcc's hazard recognizer keeps a VALU read of an 8-pass XDL MFMA
// result 11 wait states behind the MFMA and counts EVERY instruction in between as one state,
// s_cbranch_* / s_branch included.  The split-f16 KV state of k_encoder64 ended in
//     v_mfma ... ; v_add ; v_max3 ; s_cbranch_execz (not taken) ; s_branch (taken) ;
//     v_mbcnt ; s_mov ; v_mbcnt ; s_nop 4 ; v_pk_fma <- reads the MFMA's accumulator
// (12 states by that count) in the build that had a run-time branch between a masked and an
// unmasked row-tile path, and returned timing-dependent states (2-800 of 25 000 forwards; 23 % of
// them once the wave ran at s_setprio 3).  The single-path builds (the pad is one s_nop) never failed.
//
// One workgroup of 8 waves per CU.  Waves 0-3 (one per SIMD) are victims at s_setprio 3, waves 4-7
// share their SIMDs and are idle (0) or stream MFMAs (1).  A victim repeats
//     acc <- 0 ; MFMA acc += ones x ones (every element becomes 16) ; GAP ; snap += acc[R]
// so snap must end at exactly 16 * iters in every lane; a read that overtakes the MFMA's write of
// register R sees the 0.  GAP = k wait states made of: N  k x s_nop;  B  (k-2) x s_nop +
// s_cbranch_execz (not taken) + s_branch (taken);  X  (k-1) x s_nop + s_cbranch_execz;
// T  (k-1) x s_nop + s_branch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define NOP0 ""
#define NOP1 "s_nop 0\n"
#define NOP2 "s_nop 1\n"
#define NOP3 "s_nop 2\n"
#define NOP4 "s_nop 3\n"
#define NOP5 "s_nop 4\n"
#define NOP6 "s_nop 5\n"
#define NOP7 "s_nop 6\n"
#define NOP8 "s_nop 7\n"
#define NOP9 "s_nop 8\n"
#define NOP10 "s_nop 9\n"
#define NOP11 "s_nop 10\n"
#define NOP12 "s_nop 11\n"
#define NOP13 "s_nop 12\n"
#define NOP14 "s_nop 13\n"
// k wait states made of k VALU (V) or k SALU (S) instructions instead of s_nop
#define V1 "v_mov_b32 v241, v241\n"
#define V2 V1 V1
#define V4 V2 V2
#define V8 V4 V4
#define V16 V8 V8
#define S1 "s_mov_b32 s41, s41\n"
#define S2 S1 S1
#define S4 S2 S2
#define S8 S4 S4
#define S16 S8 S8
// the tail of the failing k_encoder64 build, instruction for instruction (12 states by hipcc's count)
#define KTAIL "v_add_f32 v241, v241, v241\n v_max3_f32 v242, v242, v242, v242\n s_cbranch_execz 2f\n s_branch 3f\n 2:\n s_nop 0\n 3:\n" \
              "v_mbcnt_lo_u32_b32 v243, -1, 0\n s_mov_b32 s41, 0x3a000000\n v_mbcnt_hi_u32_b32 v243, -1, v243\n s_nop 4\n"
// ... and the same without the two branch instructions but two more s_nop states
#define KTAIL_NB "v_add_f32 v241, v241, v241\n v_max3_f32 v242, v242, v242, v242\n s_nop 1\n" \
              "v_mbcnt_lo_u32_b32 v243, -1, 0\n s_mov_b32 s41, 0x3a000000\n v_mbcnt_hi_u32_b32 v243, -1, v243\n s_nop 4\n"
#define BR_B "s_cbranch_execz 2f\n s_branch 3f\n 2:\n s_nop 0\n 3:\n"
#define BR_X "s_cbranch_execz 3f\n 3:\n"
#define BR_T "s_branch 3f\n s_nop 0\n 3:\n"

#define VICTIM(GAP, REG)                                                                              \
  asm volatile(                                                                                       \
      "s_setprio 3\n s_mov_b32 s40, %[iters]\n v_mov_b32 v240, 0\n"                                   \
      "v_mov_b32 v200, %[one]\n v_mov_b32 v201, %[one]\n v_mov_b32 v202, %[one]\n v_mov_b32 v203, %[one]\n" \
      "v_mov_b32 v204, %[one]\n v_mov_b32 v205, %[one]\n v_mov_b32 v206, %[one]\n v_mov_b32 v207, %[one]\n" \
      "1:\n"                                                                                          \
      "v_mov_b32 v208, 0\n v_mov_b32 v209, 0\n v_mov_b32 v210, 0\n v_mov_b32 v211, 0\n"               \
      "v_mov_b32 v212, 0\n v_mov_b32 v213, 0\n v_mov_b32 v214, 0\n v_mov_b32 v215, 0\n"               \
      "v_mov_b32 v216, 0\n v_mov_b32 v217, 0\n v_mov_b32 v218, 0\n v_mov_b32 v219, 0\n"               \
      "v_mov_b32 v220, 0\n v_mov_b32 v221, 0\n v_mov_b32 v222, 0\n v_mov_b32 v223, 0\n"               \
      "s_nop 7\n"                                                                                     \
      "v_mfma_f32_32x32x16_f16 v[208:223], v[204:207], v[200:203], v[208:223]\n"                      \
      GAP                                                                                             \
      "v_add_f32 v240, v240, " REG "\n"                                                               \
      "s_nop 15\n s_nop 15\n s_nop 15\n"                                                              \
      "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"                              \
      "s_setprio 0\n v_mov_b32 %[o], v240\n"                                                          \
      : [o] "=v"(out)                                                                                 \
      : [iters] "s"(iters), [one] "s"(one)                                                            \
      : "s40", "scc", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209",  \
        "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", \
        "v222", "v223", "v240", "v241", "v242", "v243", "s41")

// gap kinds x state counts, for the FIRST and the LAST accumulator register
#define KINDS(K, KM1, KM2, ID)                                           \
  case 8 * ID + 0: VICTIM(NOP##K, "v208"); break;                       \
  case 8 * ID + 1: VICTIM(NOP##KM2 BR_B, "v208"); break;                \
  case 8 * ID + 2: VICTIM(NOP##KM1 BR_X, "v208"); break;                \
  case 8 * ID + 3: VICTIM(NOP##KM1 BR_T, "v208"); break;                \
  case 8 * ID + 4: VICTIM(NOP##K, "v223"); break;                       \
  case 8 * ID + 5: VICTIM(NOP##KM2 BR_B, "v223"); break;                \
  case 8 * ID + 6: VICTIM(NOP##KM1 BR_X, "v223"); break;                \
  case 8 * ID + 7: VICTIM(NOP##KM1 BR_T, "v223"); break;
static const int STATES[] = {2, 4, 6, 8, 9, 10, 11, 12, 13, 14};
// second table: acc[15] behind k VALU / k SALU instructions, and the kernel's literal tail
#define VS(ID, VSTR, SSTR)                                  \
  case 100 + 2 * ID: VICTIM(VSTR, "v223"); break;            \
  case 101 + 2 * ID: VICTIM(SSTR, "v223"); break;
static const int VS_STATES[] = {8, 10, 11, 12, 13, 14, 16, 20, 24, 32};

__global__ __launch_bounds__(512) void k_probe(int variant, int sibling, int iters, float* res) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t one = 0x3c003c00u;   // f16 (1, 1)
  float out = -1.f;
  if (wave >= 4) {   // the sibling wave of each victim's SIMD
    if (sibling == 1) {
      typedef float f16acc __attribute__((ext_vector_type(16)));
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      f16acc a = {0}, b = {0};
      h8 x = {1, 1, 1, 1, 1, 1, 1, 1};
      for (int i = 0; i < iters * 6; ++i) {
        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a, 0, 0, 0);
        b = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, b, 0, 0, 0);
      }
      if (a[0] + b[0] == 123.f) res[0] = 1.f;
    }
    return;
  }
  switch (variant) {
    KINDS(2, 1, 0, 0) KINDS(4, 3, 2, 1) KINDS(6, 5, 4, 2) KINDS(8, 7, 6, 3) KINDS(9, 8, 7, 4)
    KINDS(10, 9, 8, 5) KINDS(11, 10, 9, 6) KINDS(12, 11, 10, 7) KINDS(13, 12, 11, 8) KINDS(14, 13, 12, 9)
    VS(0, V8, S8) VS(1, V8 V2, S8 S2) VS(2, V8 V2 V1, S8 S2 S1) VS(3, V8 V4, S8 S4) VS(4, V8 V4 V1, S8 S4 S1)
    VS(5, V8 V4 V2, S8 S4 S2) VS(6, V16, S16) VS(7, V16 V4, S16 S4) VS(8, V16 V8, S16 S8) VS(9, V16 V16, S16 S16)
    case 200: VICTIM(KTAIL, "v223"); break;
    case 201: VICTIM(KTAIL_NB, "v223"); break;
  }
  res[(blockIdx.x * 4 + wave) * 64 + lane] = out;
}

int main() {
  const int iters = 4000, blocks = 256;
  float* d;
  hipMalloc(&d, blocks * 256 * sizeof(float));
  std::vector<float> h(blocks * 256);
  const char* kind[4] = {"N  k nops                        ", "B  k-2 nops + execz(nt) + branch ", "X  k-1 nops + execz (not taken)  ",
                         "T  k-1 nops + s_branch (taken)   "};
  for (int sib = 0; sib < 2; ++sib)
    for (int id = 0; id < 10; ++id)
      for (int v = 0; v < 8; ++v) {
        hipMemset(d, 0, blocks * 256 * sizeof(float));
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(512), 0, 0, 8 * id + v, sib, iters, d);
        hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
        long bad = 0, q[4] = {0, 0, 0, 0};
        double lost = 0;
        for (size_t i = 0; i < h.size(); ++i)
          if (h[i] != 16.f * iters) { ++bad; ++q[(i & 63) >> 4]; lost += 16.f * iters - h[i]; }
        printf("states %2d  %s reg %s  sibling %s : %7ld of %zu lanes short (lanes 0-15 %ld, 16-31 %ld, 32-47 %ld, 48-63 %ld), "
               "%.0f stale reads\n", STATES[id], kind[v & 3], v < 4 ? "acc[0] " : "acc[15]", sib ? "MFMA" : "idle", bad, h.size(),
               q[0], q[1], q[2], q[3], lost / 16.0);
      }
  auto run = [&](int variant, int sib, const char* what) {
    hipMemset(d, 0, blocks * 256 * sizeof(float));
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(512), 0, 0, variant, sib, iters, d);
    hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
    long bad = 0, q[4] = {0, 0, 0, 0};
    double lost = 0;
    for (size_t i = 0; i < h.size(); ++i)
      if (h[i] != 16.f * iters) { ++bad; ++q[(i & 63) >> 4]; lost += 16.f * iters - h[i]; }
    printf("%s reg acc[15]  sibling %s : %7ld of %zu lanes short (lanes 0-15 %ld, 16-31 %ld, 32-47 %ld, 48-63 %ld), %.0f of %.0f reads stale\n",
           what, sib ? "MFMA" : "idle", bad, h.size(), q[0], q[1], q[2], q[3], lost / 16.0, (double)h.size() * iters);
  };
  char buf[96];
  for (int sib = 0; sib < 2; ++sib) {
    for (int id = 0; id < 10; ++id) {
      snprintf(buf, sizeof buf, "%2d VALU instructions (v_mov)      ", VS_STATES[id]);
      run(100 + 2 * id, sib, buf);
      snprintf(buf, sizeof buf, "%2d SALU instructions (s_mov)      ", VS_STATES[id]);
      run(101 + 2 * id, sib, buf);
    }
    run(200, sib, "k_encoder64 tail as built (12 by hipcc's count, 2 of them branches)");
    run(201, sib, "the same, s_nop 1 in place of the two branches                    ");
  }
  hipFree(d);
  return 0;
}
