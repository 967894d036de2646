// Probe (gfx950): how close to a v_mfma_f32_32x32x16_f16 may a VALU instruction write the
// MFMA's B operand registers - before it (RAW) and after it (WAR) - when the OTHER wave of the
// SIMD keeps the matrix pipe busy?   Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/mfma_hazard_probe.hip
//
// One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) are the victims, waves 4-7 share
// their SIMDs and issue MFMAs back to back (1), 1-KB global loads (2) or both (3).  A victim repeats
//     B <- 1.0 (4 x v_mov)   s_nop PRE   D += ones x B (one MFMA)   s_nop POST   B <- 2.0 (4 x v_mov)
// so D must end at exactly 16 * iters in every lane; any read of a 2.0 (an operand not yet
// written, or already overwritten) shows as a larger value in that lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define PROD_MOV "v_mov_b32 v200, %[one]\n v_mov_b32 v201, %[one]\n v_mov_b32 v202, %[one]\n v_mov_b32 v203, %[one]\n"
#define PROD_CVTPK "v_cvt_pk_f16_f32 v200, v230, v230\n v_cvt_pk_f16_f32 v201, v230, v230\n v_cvt_pk_f16_f32 v202, v230, v230\n v_cvt_pk_f16_f32 v203, v230, v230\n"
#define PROD_PKRTZ "v_cvt_pkrtz_f16_f32 v200, v230, v230\n v_cvt_pkrtz_f16_f32 v201, v230, v230\n v_cvt_pkrtz_f16_f32 v202, v230, v230\n v_cvt_pkrtz_f16_f32 v203, v230, v230\n"
#define PROD_PKMUL "v_pk_mul_f32 v[200:201], v[232:233], v[234:235]\n v_pk_mul_f32 v[202:203], v[232:233], v[234:235]\n"
#define PROD_MIX "v_fma_mixlo_f16 v200, v230, v234, v236\n v_fma_mixhi_f16 v200, v230, v234, v236\n v_fma_mixlo_f16 v201, v230, v234, v236\n v_fma_mixhi_f16 v201, v230, v234, v236\n" \
                 "v_fma_mixlo_f16 v202, v230, v234, v236\n v_fma_mixhi_f16 v202, v230, v234, v236\n v_fma_mixlo_f16 v203, v230, v234, v236\n v_fma_mixhi_f16 v203, v230, v234, v236\n"
#define VICTIM(PROD, PRE, POST)                                                                       \
  asm volatile(                                                                                   \
      "s_mov_b32 s40, %[iters]\n"                                                                 \
      "v_mov_b32 v208, 0\n v_mov_b32 v209, 0\n v_mov_b32 v210, 0\n v_mov_b32 v211, 0\n"           \
      "v_mov_b32 v212, 0\n v_mov_b32 v213, 0\n v_mov_b32 v214, 0\n v_mov_b32 v215, 0\n"           \
      "v_mov_b32 v216, 0\n v_mov_b32 v217, 0\n v_mov_b32 v218, 0\n v_mov_b32 v219, 0\n"           \
      "v_mov_b32 v220, 0\n v_mov_b32 v221, 0\n v_mov_b32 v222, 0\n v_mov_b32 v223, 0\n"           \
      "v_mov_b32 v204, %[one]\n v_mov_b32 v205, %[one]\n v_mov_b32 v206, %[one]\n v_mov_b32 v207, %[one]\n" \
      "v_mov_b32 v200, %[two]\n v_mov_b32 v201, %[two]\n v_mov_b32 v202, %[two]\n v_mov_b32 v203, %[two]\n" \
      "v_mov_b32 v230, 1.0\n v_mov_b32 v231, 1.0\n v_mov_b32 v232, %[one]\n v_mov_b32 v233, %[one]\n"  \
      "v_mov_b32 v234, 1.0\n v_mov_b32 v235, 1.0\n v_mov_b32 v236, 0\n"                          \
      "s_nop 15\n"                                                                                \
      "1:\n"                                                                                      \
      PROD                                                                                        \
      PRE                                                                                         \
      "v_mfma_f32_32x32x16_f16 v[208:223], v[204:207], v[200:203], v[208:223]\n"                  \
      POST                                                                                        \
      "v_mov_b32 v200, %[two]\n v_mov_b32 v201, %[two]\n v_mov_b32 v202, %[two]\n v_mov_b32 v203, %[two]\n" \
      "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"                          \
      "s_nop 15\n s_nop 15\n"                                                                     \
      "v_mov_b32 %[o0], v208\n v_mov_b32 %[o1], v212\n v_mov_b32 %[o2], v216\n v_mov_b32 %[o3], v223\n" \
      : [o0] "=v"(o[0]), [o1] "=v"(o[1]), [o2] "=v"(o[2]), [o3] "=v"(o[3])                         \
      : [iters] "s"(iters), [one] "s"(one), [two] "s"(two)                                        \
      : "s40", "scc", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210",  \
        "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223",   \
        "v230", "v231", "v232", "v233", "v234", "v235", "v236")

// The mma16_split3 pattern: c += A.B1 ; m += A.B2 ; (VALU writes A2) ; c += A2.B2 - the third MFMA
// depends on the first through its accumulator - then EVERY source register is overwritten with
// 2.0 in the slots right after (the next iteration's conversions, as hipcc schedules them).
// c must end at 32 * iters, m at 16 * iters.  GAP: wait states between the third MFMA and the
// overwrites.
#define TRIPLE(GAP)                                                                               \
  asm volatile(                                                                                   \
      "s_mov_b32 s40, %[iters]\n"                                                                 \
      "v_mov_b32 v208, 0\n v_mov_b32 v209, 0\n v_mov_b32 v210, 0\n v_mov_b32 v211, 0\n"           \
      "v_mov_b32 v212, 0\n v_mov_b32 v213, 0\n v_mov_b32 v214, 0\n v_mov_b32 v215, 0\n"           \
      "v_mov_b32 v216, 0\n v_mov_b32 v217, 0\n v_mov_b32 v218, 0\n v_mov_b32 v219, 0\n"           \
      "v_mov_b32 v220, 0\n v_mov_b32 v221, 0\n v_mov_b32 v222, 0\n v_mov_b32 v223, 0\n"           \
      "v_mov_b32 v224, 0\n v_mov_b32 v225, 0\n v_mov_b32 v226, 0\n v_mov_b32 v227, 0\n"           \
      "v_mov_b32 v228, 0\n v_mov_b32 v229, 0\n v_mov_b32 v230, 0\n v_mov_b32 v231, 0\n"           \
      "v_mov_b32 v232, 0\n v_mov_b32 v233, 0\n v_mov_b32 v234, 0\n v_mov_b32 v235, 0\n"           \
      "v_mov_b32 v236, 0\n v_mov_b32 v237, 0\n v_mov_b32 v238, 0\n v_mov_b32 v239, 0\n"           \
      "v_mov_b32 v180, 1.0\n"                                                                     \
      "v_mov_b32 v204, %[one]\n v_mov_b32 v205, %[one]\n v_mov_b32 v206, %[one]\n v_mov_b32 v207, %[one]\n" \
      "s_nop 15\n"                                                                                \
      "1:\n"                                                                                      \
      "v_cvt_pk_f16_f32 v200, v180, v180\n v_cvt_pk_f16_f32 v201, v180, v180\n v_cvt_pk_f16_f32 v202, v180, v180\n v_cvt_pk_f16_f32 v203, v180, v180\n" \
      "v_cvt_pkrtz_f16_f32 v196, v180, v180\n v_cvt_pkrtz_f16_f32 v197, v180, v180\n v_cvt_pkrtz_f16_f32 v198, v180, v180\n v_cvt_pkrtz_f16_f32 v199, v180, v180\n" \
      "s_nop 0\n"                                                                                 \
      "v_mfma_f32_32x32x16_f16 v[208:223], v[204:207], v[200:203], v[208:223]\n"                  \
      "v_cvt_pk_f16_f32 v192, v180, v180\n v_cvt_pk_f16_f32 v193, v180, v180\n"                   \
      "v_mfma_f32_32x32x16_f16 v[224:239], v[204:207], v[196:199], v[224:239]\n"                  \
      "v_cvt_pk_f16_f32 v194, v180, v180\n v_cvt_pk_f16_f32 v195, v180, v180\n"                   \
      "v_mov_b32 v200, %[two]\n v_mov_b32 v201, %[two]\n"                                         \
      "s_nop 0\n"                                                                                 \
      "v_mfma_f32_32x32x16_f16 v[208:223], v[192:195], v[196:199], v[208:223]\n"                  \
      GAP                                                                                         \
      "v_mov_b32 v196, %[two]\n v_mov_b32 v197, %[two]\n v_mov_b32 v198, %[two]\n v_mov_b32 v199, %[two]\n" \
      "v_mov_b32 v192, %[two]\n v_mov_b32 v193, %[two]\n v_mov_b32 v194, %[two]\n v_mov_b32 v195, %[two]\n" \
      "v_mov_b32 v202, %[two]\n v_mov_b32 v203, %[two]\n"                                         \
      "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"                          \
      "s_nop 15\n s_nop 15\n"                                                                     \
      "v_mov_b32 %[o0], v208\n v_mov_b32 %[o1], v223\n v_mov_b32 %[o2], v224\n v_mov_b32 %[o3], v239\n" \
      : [o0] "=v"(o[0]), [o1] "=v"(o[1]), [o2] "=v"(o[2]), [o3] "=v"(o[3])                         \
      : [iters] "s"(iters), [one] "s"(one), [two] "s"(two)                                        \
      : "s40", "scc", "v180", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", \
        "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210",  \
        "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223",   \
        "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236",   \
        "v237", "v238", "v239")

#define AGGRESSOR()                                                                               \
  asm volatile(                                                                                   \
      "s_mov_b32 s40, %[iters]\n"                                                                 \
      "1:\n"                                                                                      \
      "v_mfma_f32_32x32x16_f16 v[100:115], v[92:95], v[96:99], v[100:115]\n"                      \
      "v_mfma_f32_32x32x16_f16 v[116:131], v[92:95], v[96:99], v[116:131]\n"                      \
      "v_mfma_f32_32x32x16_f16 v[132:147], v[92:95], v[96:99], v[132:147]\n"                      \
      "v_mfma_f32_32x32x16_f16 v[148:163], v[92:95], v[96:99], v[148:163]\n"                      \
      "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"                          \
      "s_nop 15\n s_nop 15\n v_mov_b32 %[o0], v100\n"                                             \
      : [o0] "=v"(o[0])                                                                           \
      : [iters] "s"(iters)                                                                        \
      : "s40", "scc", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103",  \
        "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", \
        "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", \
        "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", \
        "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", \
        "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163")

// Sibling waves that keep the VECTOR-MEMORY return path busy instead (2), or both pipes (3): a
// stream of 1-KB global_load_dwordx4 (L2-resident buffer) landing in the sibling wave's VGPRs
// while the victim's MFMA fetches operands a VALU instruction has just written.
#define AGGRESSOR_LD(MFMAS)                                                                       \
  asm volatile(                                                                                   \
      "s_mov_b32 s40, %[iters]\n"                                                                 \
      "v_mov_b32 v91, %[voff]\n"                                                                  \
      "1:\n"                                                                                      \
      "global_load_dwordx4 v[100:103], v91, %[base]\n"                                            \
      "global_load_dwordx4 v[104:107], v91, %[base] offset:1024\n"                                \
      "global_load_dwordx4 v[108:111], v91, %[base] offset:2048\n"                                \
      "global_load_dwordx4 v[112:115], v91, %[base] offset:3072\n"                                \
      MFMAS                                                                                       \
      "v_add_u32 v91, 0x2000, v91\n v_and_b32 v91, 0x3fffff, v91\n"                               \
      "s_waitcnt vmcnt(2)\n"                                                                      \
      "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"                          \
      "s_waitcnt vmcnt(0)\n s_nop 15\n v_mov_b32 %[o0], v100\n"                                   \
      : [o0] "=v"(o[0])                                                                           \
      : [iters] "s"(iters), [voff] "v"((threadIdx.x & 63) * 16), [base] "s"(lbuf)                 \
      : "s40", "scc", "memory", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",  \
        "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", \
        "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", \
        "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", \
        "v141", "v142", "v143", "v144", "v145", "v146", "v147")
#define LD_MFMAS "v_mfma_f32_32x32x16_f16 v[116:131], v[92:95], v[96:99], v[116:131]\n v_mfma_f32_32x32x16_f16 v[132:147], v[92:95], v[96:99], v[132:147]\n"

#define KERNEL(NAME, PROD, PRE, POST)                                                                  \
  __global__ __launch_bounds__(512) void NAME(float* out, int iters_, int aggr, const float* lbuf) {                 \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                            \
    const int iters = __builtin_amdgcn_readfirstlane(wave < 4 ? iters_ : 2 * iters_);             \
    const unsigned one = 0x3c003c00u, two = 0x40004000u;                                          \
    float o[4] = {0.f, 0.f, 0.f, 0.f};                                                            \
    if (wave < 4) {                                                                               \
      VICTIM(PROD, PRE, POST);                                                                        \
      float* dst = out + ((size_t)blockIdx.x * 4 + wave) * 256 + (threadIdx.x & 63) * 4;          \
      dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];                                 \
    } else if (aggr) {                                                                            \
      if (aggr == 1) { AGGRESSOR(); } else if (aggr == 2) { AGGRESSOR_LD(""); } else { AGGRESSOR_LD(LD_MFMAS); }  \
      if (o[0] == 12345.f) out[0] = o[0];                                                         \
    }                                                                                             \
  }

#define KERNEL3(NAME, GAP)                                                                        \
  __global__ __launch_bounds__(512) void NAME(float* out, int iters_, int aggr, const float* lbuf) {                 \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                            \
    const int iters = __builtin_amdgcn_readfirstlane(wave < 4 ? iters_ : 3 * iters_);             \
    const unsigned one = 0x3c003c00u, two = 0x40004000u;                                          \
    float o[4] = {0.f, 0.f, 0.f, 0.f};                                                            \
    if (wave < 4) {                                                                               \
      TRIPLE(GAP);                                                                                \
      float* dst = out + ((size_t)blockIdx.x * 4 + wave) * 256 + (threadIdx.x & 63) * 4;          \
      dst[0] = o[0] * 0.5f; dst[1] = o[1] * 0.5f; dst[2] = o[2]; dst[3] = o[3];                   \
    } else if (aggr) {                                                                            \
      if (aggr == 1) { AGGRESSOR(); } else if (aggr == 2) { AGGRESSOR_LD(""); } else { AGGRESSOR_LD(LD_MFMAS); }  \
      if (o[0] == 12345.f) out[0] = o[0];                                                         \
    }                                                                                             \
  }

#define N0 ""
#define N1 "s_nop 0\n"
#define N2 "s_nop 1\n"
#define N4 "s_nop 3\n"
#define N8 "s_nop 7\n"
#define N16 "s_nop 15\n"
#define N64 "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
#define N3 "s_nop 2\n"
#define FAMILY(P, PROD) KERNEL(k_##P##0, PROD, N0, N64) KERNEL(k_##P##1, PROD, N1, N64) KERNEL(k_##P##2, PROD, N2, N64) \
                        KERNEL(k_##P##3, PROD, N3, N64) KERNEL(k_##P##4, PROD, N4, N64)
FAMILY(mov, PROD_MOV)
FAMILY(cvtpk, PROD_CVTPK)
FAMILY(pkrtz, PROD_PKRTZ)
FAMILY(pkmul, PROD_PKMUL)
FAMILY(mix, PROD_MIX)
KERNEL(k_war0, PROD_MOV, N64, N0)
KERNEL(k_war1, PROD_MOV, N64, N1)

KERNEL3(k_triple0, N0)
KERNEL3(k_triple1, N1)
KERNEL3(k_triple8, N8)

typedef void (*kern_t)(float*, int, int, const float*);
struct Case { const char* name; kern_t k; };

int main() {
  const int blocks = 256, iters = 2000;
  float* d;
  hipMalloc(&d, (size_t)blocks * 4 * 256 * sizeof(float));
  std::vector<float> h((size_t)blocks * 4 * 256);
  float* lbuf;
  hipMalloc(&lbuf, (size_t)8 << 20);
  hipMemset(lbuf, 0, (size_t)8 << 20);
#define ROWS(P, T) {T " -> MFMA, 0 states between", k_##P##0}, {T " 1 state", k_##P##1}, {T " 2 states", k_##P##2}, \
                   {T " 3 states", k_##P##3}, {T " 4 states", k_##P##4}
  Case cases[] = {ROWS(mov, "v_mov_b32"), ROWS(cvtpk, "v_cvt_pk_f16_f32"), ROWS(pkrtz, "v_cvt_pkrtz_f16_f32"),
                  ROWS(pkmul, "v_pk_mul_f32"), ROWS(mix, "v_fma_mixlo/hi_f16"),
                  {"MFMA -> v_mov_b32 of its B operand, 0 states", k_war0}, {"MFMA -> v_mov_b32 1 state", k_war1},
                  {"split3 triple, sources rewritten 0 states after", k_triple0}, {"split3 triple, 1 state", k_triple1},
                  {"split3 triple, 8 states", k_triple8}};
  for (int aggr = 0; aggr < 4; ++aggr)
    for (auto& c : cases) {
      long bad = 0, quarter[4] = {0, 0, 0, 0};
      float worst = 0.f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(d, 0, h.size() * sizeof(float));
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(512), 0, 0, d, iters, aggr, lbuf);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
        const float want = 16.0f * iters;
        for (size_t i = 0; i < h.size(); ++i)
          if (h[i] != want) {
            ++bad; ++quarter[((i / 4) % 64) / 16];
            if (h[i] - want > worst) worst = h[i] - want;
          }
      }
      printf("%-48s sibling stream %d (0 none, 1 MFMA, 2 loads, 3 loads+MFMA): %8ld wrong values of %zu  (lanes 0-15: %ld, 16-31: %ld, 32-47: %ld, 48-63: %ld; worst excess %.0f)\n",
             c.name, aggr, bad, 3 * h.size(), quarter[0], quarter[1], quarter[2], quarter[3], worst);
    }
  return 0;
}
