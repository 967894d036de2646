// Shared definitions for the OETR gfx950 kernels (device helpers + host-side
// launch declarations).  CDNA4 only: wave = 64 lanes, f32 MFMA 32x32x2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace oetr {

constexpr int C = 256;        // d_model
constexpr int NH = 8;         // heads
constexpr int HD = 32;        // head dim  (= one 32-wide MFMA n-tile)
constexpr int FF = 512;       // MLP hidden
constexpr int TM = 32;        // token rows per workgroup tile (= MFMA M)
constexpr int LDA = C + 4;    // padded row stride (floats) of a [TM][256] LDS tile
constexpr int LDH = FF + 4;   // padded row stride of the [TM][512] hidden tile
constexpr int NTHREADS = 256; // 4 waves, one per SIMD
constexpr int KV_FLOATS = NH * HD * HD;  // 8192: one KV state (all heads)
constexpr float LN_EPS = 1e-5f;
constexpr float ATTN_EPS = 1e-6f;
constexpr int MAX_TOKENS = 10000;  // NECK.MAX_SHAPE 100x100 (reference default.py:25-28)

// Ablation switches for timing attribution (tools/ablate.sh builds a second
// library with -DOETR_ABLATE; the shipped build compiles them out).
#ifdef OETR_ABLATE
#define ABL(flags, bit) (((flags) & (bit)) != 0)
#else
#define ABL(flags, bit) false
#endif
// Phase timing (tools/phase_timing.py): wave 0 of every workgroup stamps
// s_memtime at phase boundaries.  Compiled out of the shipped build.
// Cumulative timing (tools/ablate_run.py): with OETR_ABLATE = (n+1) << 16 every workgroup
// returns when it reaches phase boundary n, so kernel time vs n is the cost of the
// phases up to n under real conditions (launch, loads and barriers included).
#if defined(OETR_ABLATE)
#define PHASE_STAMP(p, idx) do { if (((p).dbg >> 16) == (idx) + 1) return; } while (0)
#elif defined(OETR_PHASE_TIMING) && OETR_PHASE_TIMING == 2   // per WAVE, workgroups 0..15 (tools/wave_skew64.py)
#define PHASE_STAMP(p, idx)                                                              \
  do {                                                                                   \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 16 && (p).tbuf)                          \
      (p).tbuf[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (idx)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#elif defined(OETR_PHASE_TIMING) && OETR_PHASE_TIMING == 3   // REAL time (s_memrealtime, 100 MHz): launch boundaries, clock (tools/launch_boundary.py)
#define PHASE_STAMP(p, idx)                                                              \
  do {                                                                                   \
    if (threadIdx.x == 0 && (p).tbuf) (p).tbuf[blockIdx.x * 16 + (idx)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#elif defined(OETR_PHASE_TIMING)
#define PHASE_STAMP(p, idx)                                                              \
  do {                                                                                   \
    if (threadIdx.x == 0 && (p).tbuf) (p).tbuf[blockIdx.x * 16 + (idx)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define PHASE_STAMP(p, idx) do {} while (0)
#endif
#if defined(OETR_ABLATE) && defined(OETR_PHASE_TIMING)
#error "OETR_ABLATE and OETR_PHASE_TIMING are separate builds"
#endif
enum { ABL_KVREDUCE = 1, ABL_GELU = 2, ABL_ELU = 4, ABL_LN = 8, ABL_GEMM = 16, ABL_STORE = 32,
       ABL_XLOAD = 64, ABL_WLOAD = 128, ABL_ATTN = 256, ABL_KVSTATE = 512 };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Write-through 16-byte store (round 4): `buffer_store_dwordx4 ... sc1` from a wave-uniform base.
// A plain store leaves its line dirty in the XCD's L2 until the end-of-kernel release writes every
// dirty line back at once - the launch boundary then costs its fixed ~1.5 us PLUS dirty bytes /
// ~6 TB/s (MI355X_MICROARCH.md, price-list row "boundary": ~20 MB of residual rows, phi(Q)
// fragments and partial states per encoder launch at 8 pairs = ~3 us exposed at each of the eight
// boundaries).  An sc1 store leaves L2 as it is issued, under the kernel's own compute, and the
// boundary is down to its fixed part; the price is that the line is dropped from L2, so the next
// launch reads it from the Infinity Cache instead.  OETR_WT is the set of encoder outputs stored
// this way (bit 0: residual rows x, 1: phi(Q) fragments, 2: partial linear-attention states);
// same values to the same addresses either way - results are bit-identical.  Measured
// (profiles/r4_wt_stores.txt, one-process A/B, serial step): all three -1.8 % at 32 pairs @1024x1024,
// -1.4 % at 8 pairs 640x640 vs 1280x1280, -0.5 % at 8 pairs @640x640, +0.7 % at one pair; overlapped
// throughput within the noise (+0.8 % at 8 pairs @640x640) - i.e. the flush is a small part of what a
// boundary costs here; shipped because it is free and never worse beyond the noise.
#ifndef OETR_WT
#define OETR_WT 7
#endif
template <bool WT>
__device__ __forceinline__ void store16(float* base_uniform, unsigned byte_off, const f32x4& v) {
  if constexpr (WT) {
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base_uniform, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, v), rs, (int)byte_off, 0, 16 /* sc1 */);
  } else {
    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base_uniform) + byte_off) = v;
  }
}

// Arithmetic of the GEMM-shaped stages (values == oetr_dtype in include/oetr_hip.h).
// Everything that is not a GEMM operand (LayerNorm, phi, normalisers, softmax,
// residual stream, accumulators) is fp32 in every mode.
//   GM_F32   exact fp32 products on v_mfma_f32_32x32x2_f32
//   GM_SPLIT fp32-class: a = ah + al/2^11 in f16, 3 v_mfma_f32_32x32x16_f16 per product
//   GM_F16   one v_mfma_f32_32x32x16_f16 per product (operands rounded to f16, RNE)
//   GM_BF16  one v_mfma_f32_32x32x16_bf16 per product (operands rounded to bf16, RNE)
enum { GM_F32 = 0, GM_SPLIT = 1, GM_F16 = 2, GM_BF16 = 3 };
constexpr bool gm_half(int m) { return m != GM_F32; }          // operands are 16-bit planes
constexpr int gm_planes(int m) { return m == GM_SPLIT ? 2 : 1; }  // planes per operand

// Per-GEMM-site arithmetic of the two-plane (split) mode - the precision policy
// (DESIGN.md 3.10).  A product a.b with a = ah + al/2^11, b = bh + bl/2^11 keeps
//   SITE_FULL   ah.bh + (ah.bl + al.bh)/2^11      3 MFMAs, fp32-class
//   SITE_ACT_HI ah.bh + ah.bl/2^11                2 MFMAs, activations rounded to f16
//   SITE_W_HI   ah.bh + al.bh/2^11                2 MFMAs, weights rounded to f16
//   SITE_HI     ah.bh                             1 MFMA,  both rounded to f16 (RNE: the hi
//                                                 planes ARE round-to-nearest f16 values)
// Fragments a site does not use are neither fetched (weight lo plane) nor read (activation
// lo plane).  Ignored by the single-plane modes.
enum { SITE_FULL = 0, SITE_ACT_HI = 1, SITE_W_HI = 2, SITE_HI = 3 };
constexpr bool site_w_lo(int s) { return s == SITE_FULL || s == SITE_ACT_HI; }    // uses bl
constexpr bool site_act_lo(int s) { return s == SITE_FULL || s == SITE_W_HI; }    // uses al
constexpr int site_mfmas(int s) { return s == SITE_FULL ? 3 : s == SITE_HI ? 1 : 2; }
// Policies (oetr_dtype -> policy id, api.hip).  The assignment is what survives the north_star
// bar - boxes within 1e-3 IoU of the fp32 reference on every golden, sharpened heads included -
// when ONE site at a time, then the combination, is reduced (tools/site_drift.py on the CPU
// oracle; tests/test_gpu_precision.py + profiles/r3_site_drift.json on the GPU):
//   Q, K and the decoder's K projection tolerate f16 operands (phi(Q) enters numerator and
//   normaliser alike, K only through sums over all source tokens); V, merge, both MLP GEMMs and
//   the attention contractions do not (each alone: 1 - IoU = 3e-3 .. 3e-2).
// Policy 0 = every site fp32-class.  The OETR_SITE_* macros exist for the per-site drift study
// (variant builds of the library), not for shipping.
#ifndef OETR_SITE_Q
#define OETR_SITE_Q SITE_FULL
#endif
#ifndef OETR_SITE_K
#define OETR_SITE_K SITE_FULL
#endif
#ifndef OETR_SITE_V
#define OETR_SITE_V SITE_FULL
#endif
#ifndef OETR_SITE_MERGE
#define OETR_SITE_MERGE SITE_FULL
#endif
#ifndef OETR_SITE_MLP1
#define OETR_SITE_MLP1 SITE_FULL
#endif
#ifndef OETR_SITE_MLP2
#define OETR_SITE_MLP2 SITE_FULL
#endif
#ifndef OETR_SITE_DEC_K
#define OETR_SITE_DEC_K SITE_FULL
#endif
#ifndef OETR_SITE_DEC_V
#define OETR_SITE_DEC_V SITE_FULL
#endif
template <int POL> struct SitePolicy;
template <> struct SitePolicy<0> {
  static constexpr int Q = OETR_SITE_Q, K = OETR_SITE_K, V = OETR_SITE_V, MERGE = OETR_SITE_MERGE,
                       MLP1 = OETR_SITE_MLP1, MLP2 = OETR_SITE_MLP2, DEC_K = OETR_SITE_DEC_K,
                       DEC_V = OETR_SITE_DEC_V;
};
template <> struct SitePolicy<1> {   // OETR_DTYPE_F32_SPLIT_QK16
  static constexpr int Q = SITE_HI, K = SITE_HI, V = SITE_FULL, MERGE = SITE_FULL, MLP1 = SITE_FULL,
                       MLP2 = SITE_FULL, DEC_K = SITE_HI, DEC_V = SITE_FULL;
};
constexpr int N_POLICIES = 2;

// "f32 via split f16" GEMM mode: a = ah + al/2^11 with ah = f16(a),
// al = f16((a - ah) * 2^11); a*b ~= ah*bh + (ah*bl + al*bh)/2^11 on
// v_mfma_f32_32x32x16_f16 (3 MFMAs, 16x the f32 MFMA rate each) with f32
// accumulation.  Dropped al*bl term and the rounding of al are ~2^-22 relative:
// fp32-class results (measured against fp64: same error as torch fp32, see
// DESIGN.md §3.6).  Inputs must stay below the f16 range (65504).
constexpr float SPLIT_SCALE = 2048.0f;
constexpr float SPLIT_INV = 1.0f / 2048.0f;
constexpr int LDAH = C + 8;    // row stride (halves) of a [TM][256] f16 plane: 528 B = conflict-free b128
constexpr int LDHH = FF + 8;   // row stride (halves) of a [TM][512] f16 plane

// Token geometry of one forward call. "side" 0/1 = image1/image2 batch.
struct Geom {
  int N;         // pairs
  int L[2];      // tokens per image
  int hf[2], wf[2];
  int nt[2];     // TM-row tiles per image
  int row0[2];   // first row of the side in token-major [rows][256] buffers
  int prow0[2];  // first row of the side in the position table buffer
  int tile0[2];  // first tile slot of the side
  int ntiles;    // N * (nt[0] + nt[1])
  int rows;      // N * (L[0] + L[1])
};

// ---------------------------------------------------------------- device
#if defined(__HIPCC__)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Row of accumulator register r for a 32x32 MFMA C/D tile (col = lane & 31).
__device__ __forceinline__ int crow(int r, int half) {
  return (r & 3) + 8 * (r >> 2) + 4 * half;
}

// exp(x) for x <= 0 on the hardware exp2 unit with a compensated argument:
// x*log2(e) is formed as t + r (t = fl(x*L2E_HI), r = the rounding residue +
// x*L2E_LO), exp(x) = 2^t * (1 + r ln2).  Error ~1 ulp of v_exp_f32
// (tools: 3.6e-13 from the argument handling alone); 6 instructions, no
// branches, vs ~15 for ocml expf.
__device__ __forceinline__ float exp_neg(float x) {
  const float L2E_HI = 1.4426950216293335f, L2E_LO = 1.9259629911266175e-08f;
  const float t = x * L2E_HI;
  float r = fmaf(x, L2E_HI, -t);
  r = fmaf(x, L2E_LO, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.69314718055994531f, e);
}
// phi(x) = elu(x) + 1 = x + 1 (x > 0) | e^x (x <= 0).  torch evaluates expm1(x) + 1 for
// x <= 0, which is exp(x) to within one rounding of the final add (<= 6e-8 absolute).
// Branch-free as a MEDIAN: e^x >= x + 1 everywhere, and 1 lies between them on the other side
// (x > 0: 1 < x + 1 < e^x;  x < 0: x + 1 < e^x < 1), so phi(x) = med3(x + 1, e^x, 1) - add, mul,
// v_exp_f32, v_med3_f32 (round 2: max + min + the compensated exp_neg + add = 9; an overflowed
// e^x = inf for large x is never the median).  e^x = exp2(x log2 e) with the product rounded
// once: relative error <= |x| 4e-8 + one ulp of v_exp_f32, i.e. <= 8e-8 absolute since
// |x| e^x <= 1/e - the size of torch's own final rounding.
__device__ __forceinline__ float elu1(float x) {
  return __builtin_amdgcn_fmed3f(x + 1.0f, __builtin_amdgcn_exp2f(x * 1.4426950408889634f), 1.0f);
}

// DPP lane exchange inside a row of 16 lanes (no LDS traffic, unlike
// __shfl_xor which lowers to ds_bpermute).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// Sum over the 8 lanes sharing lane>>3: quad_perm [1,0,3,2], [2,3,0,1], then
// row_half_mirror (lane i <-> 7-i, i.e. the other quad).
__device__ __forceinline__ float sum8(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  return v;
}
// Sum over all 64 lanes (result in every lane): DPP inside rows of 16, then two
// cross-row exchanges (the only steps that need the LDS permute network).
__device__ __forceinline__ float wave_sum(float v) {
  v = sum8(v);
  v += dpp_mov<0x140>(v);  // row_mirror
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// Exact-erf GELU (torch nn.GELU default): 0.5 v (1 + erf(v / sqrt 2)).
//
// GELU needs erf only to ABSOLUTE accuracy (it is added to 1), so one formula serves
// every v:  gelu(v) = max(v, 0) - |v| (0.5 erfc(|v| / sqrt 2)),  0.5 erfc = 2^(w Q(w) - 1),
// w = min(|v|, 5.5), Q = degree-5 polynomial for log2(erfc(w / sqrt 2)) / w on [0, 5.5], fitted
// minimax on the GELU error itself (weight 0.5 w^2 erfc ln2 = d gelu / d Q: the exponent needs
// accuracy only where |v| erfc is not small) - tools/fit_erf.py: max abs error 3.1e-7 over
// |v| <= 12 (half an ulp of the result at |v| ~ 4 is 2.4e-7), 1.6e-7 for |v| < 2, every operation
// rounded to fp32.  Round 2's uniform degree-9 fit of Q measured 2.4e-7 / 1.2e-7 with four more
// FMAs per value; round 1's two-branch erf form 6.8e-7.
constexpr float GELU_T = 5.5f;
constexpr int GELU_DEG = 5;
constexpr float GELU_Q[GELU_DEG + 1] = {-1.151147082e+00f, -4.589156863e-01f, -5.323818670e-02f,
                                        7.977462822e-03f,  -7.398742635e-04f, 2.992419385e-05f};
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x), w = fminf(ax, GELU_T);
  float q = GELU_Q[GELU_DEG];
#pragma unroll
  for (int i = GELU_DEG - 1; i >= 0; --i) q = fmaf(q, w, GELU_Q[i]);
  const float e = __builtin_amdgcn_exp2f(fmaf(w, q, -1.0f));   // 0.5 erfc: the 0.5 rides in the exponent
  return fmaf(-ax, e, fmaxf(x, 0.f));                            // (-|x| is a source modifier, not an instruction)
}

// Two GELUs per instruction stream: the polynomial chain runs as v_pk_fma_f32
// (packed fp32, 2 lanes of work per VALU slot); same operations in the same order
// as gelu_erf (results agree to the last rounding of the packed/scalar code paths).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float c) { return f32x2{c, c}; }
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
  const f32x2 ax = __builtin_elementwise_abs(v);
  const f32x2 w = f32x2{fminf(ax[0], GELU_T), fminf(ax[1], GELU_T)};
  f32x2 q = splat2(GELU_Q[GELU_DEG]);
#pragma unroll
  for (int i = GELU_DEG - 1; i >= 0; --i) q = __builtin_elementwise_fma(q, w, splat2(GELU_Q[i]));
  const f32x2 u = __builtin_elementwise_fma(w, q, splat2(-1.0f));
  f32x2 g;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    g[i] = fmaf(-fabsf(v[i]), __builtin_amdgcn_exp2f(u[i]), fmaxf(v[i], 0.f));
  return g;
}

// Four GELUs as TWO interleaved packed chains: the polynomial is a serial chain of dependent
// FMAs (one chain alone issues an instruction every ~8 cycles beside MFMAs); two independent
// chains stepping together fill each other's latency.  Same operations per value as gelu_erf2.
__device__ __forceinline__ f32x4 gelu_erf4(const f32x4& v) {
  const f32x2 v0 = f32x2{v[0], v[1]}, v1 = f32x2{v[2], v[3]};
  const f32x2 a0 = __builtin_elementwise_abs(v0), a1 = __builtin_elementwise_abs(v1);
  const f32x2 w0 = f32x2{fminf(a0[0], GELU_T), fminf(a0[1], GELU_T)};
  const f32x2 w1 = f32x2{fminf(a1[0], GELU_T), fminf(a1[1], GELU_T)};
  f32x2 q0 = splat2(GELU_Q[GELU_DEG]), q1 = splat2(GELU_Q[GELU_DEG]);
#pragma unroll
  for (int i = GELU_DEG - 1; i >= 0; --i) {
    q0 = __builtin_elementwise_fma(q0, w0, splat2(GELU_Q[i]));
    q1 = __builtin_elementwise_fma(q1, w1, splat2(GELU_Q[i]));
  }
  const f32x2 u0 = __builtin_elementwise_fma(w0, q0, splat2(-1.0f));
  const f32x2 u1 = __builtin_elementwise_fma(w1, q1, splat2(-1.0f));
  f32x4 g;
  g[0] = fmaf(-fabsf(v[0]), __builtin_amdgcn_exp2f(u0[0]), fmaxf(v[0], 0.f));
  g[2] = fmaf(-fabsf(v[2]), __builtin_amdgcn_exp2f(u1[0]), fmaxf(v[2], 0.f));
  g[1] = fmaf(-fabsf(v[1]), __builtin_amdgcn_exp2f(u0[1]), fmaxf(v[1], 0.f));
  g[3] = fmaf(-fabsf(v[3]), __builtin_amdgcn_exp2f(u1[1]), fmaxf(v[3], 0.f));
  return g;
}

// XCD-aware bijective remap: hardware block b runs on XCD b % 8; give each
// XCD a contiguous run of logical tiles so tiles of one image pair share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// acc[t] += A[32 x K] * W_tile(nt0 + t), t < NT.
//   A : LDS, row-major, row stride lda floats (16-B aligned rows).
//   Wp: weights repacked in MFMA B-fragment order:
//       float4 index ((ntile * K/8 + ks) * 64 + lane) holds, for output column
//       n = 32*ntile + (lane & 31), inputs k = 8*ks + 4*(lane >> 5) + {0..3}.
//   Each wave streams its own weight fragments straight into registers with
//   fully coalesced 1-KiB loads; activations are one ds_read_b128 per k-step.
//   Both operands are double-buffered in registers one chunk (U k-steps =
//   8*U*NT MFMAs) ahead.  The sched_barriers pin "issue the next chunk's loads,
//   THEN run this chunk's MFMAs": without them hipcc sinks the prefetch loads
//   to the end of the chunk and waits on them at once (vmcnt(0) per chunk).
template <int NT, int U>
struct GemmRegs {
  f32x4 b[U][NT];
  f32x4 a[U];
};

template <int NT, int U>
__device__ __forceinline__ void gemm_fetch(GemmRegs<NT, U>& r, const float* a_ptr,
                                           const f32x4* const (&w_ptr)[NT], int chunk, unsigned ln) {
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t) r.b[u][t] = w_ptr[t][(chunk * U + u) * 64 + ln];
#pragma unroll
  for (int u = 0; u < U; ++u)
    r.a[u] = *reinterpret_cast<const f32x4*>(a_ptr + (chunk * U + u) * 8);
}

template <int NT, int U>
__device__ __forceinline__ void gemm_mma(const GemmRegs<NT, U>& r, f32x16 (&acc)[NT]) {
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r.a[u][j], r.b[u][t][j], acc[t], 0, 0, 0);
}

#ifndef OETR_GEMM_U
#define OETR_GEMM_U 4
#endif
// (Fetching a GEMM's first weight chunk one phase early was measured and is a
// loss: +7 us per encoder launch - tools/variants A/B - so every GEMM starts cold.)
template <int K, int NT, int U = OETR_GEMM_U>
__device__ __forceinline__ void gemm_rows32(const float* __restrict__ A, int lda,
                                            const f32x4* __restrict__ Wp, int nt0,
                                            int lane, f32x16 (&acc)[NT], int dbg = 0) {
  if (ABL(dbg, ABL_GEMM)) return;
  constexpr int KS = K / 8;
  constexpr int NCH = KS / U;
  static_assert(KS % (2 * U) == 0, "K must be a multiple of 16*U");
  const float* a_ptr = A + (lane & 31) * lda + 4 * (lane >> 5);
  const f32x4* w_ptr[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) w_ptr[t] = Wp + (size_t)(nt0 + t) * KS * 64;   // (uniform when nt0 is: lane comes last)

  // K is summed in NB blocks, each in an accumulator of its own that starts from zero and is added to `acc` when
  // the block is done (round 6): v_mfma_f32_32x32x2_f32 adds ONE product per accumulator step,
  // so a single accumulator is a chain of K = 256 .. 512 dependent fp32 additions per output - the exact-fp32 build
  // was the LEAST accurate one (cxy 2.1e-2 px on the sharpened 1024 / 1280-px goldens where the f16-MFMA builds,
  // 16 .. 32 steps per output, stay at 4e-3).  Four blocks: chains of 64 .. 128, then four adds - the blocked
  // summation every CPU GEMM (the reference's included) does anyway.  32 VALU adds per block and wave beside
  // 64 .. 128 MFMAs of 64 cycles.
  constexpr int NB = 4, CPB = NCH / NB;
  static_assert(NCH % (2 * NB) == 0, "K must be a multiple of 16*U*NB");
  GemmRegs<NT, U> r0, r1;
  gemm_fetch<NT, U>(r0, a_ptr, w_ptr, 0, (unsigned)lane);
  for (int b = 0; b < NB; ++b) {
    f32x16 part[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) part[t] = f32x16{0};
    for (int c = b * CPB; c < (b + 1) * CPB; c += 2) {
      gemm_fetch<NT, U>(r1, a_ptr, w_ptr, c + 1, (unsigned)lane);
      __builtin_amdgcn_sched_barrier(0);
      gemm_mma<NT, U>(r0, part);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 < NCH) gemm_fetch<NT, U>(r0, a_ptr, w_ptr, c + 2, (unsigned)lane);
      __builtin_amdgcn_sched_barrier(0);
      gemm_mma<NT, U>(r1, part);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] += part[t];
  }
}

// ---- 16-bit-plane GEMM core (GM_SPLIT / GM_F16 / GM_BF16) -----------------
// hi/lo halves of two floats: (hi0,hi1) and (lo0,lo1) packed as f16x2.
__device__ __forceinline__ void split2(float a, float b, f16x2& hi, f16x2& lo) {
  // hi = RNE f16 (v_cvt_pk_f16_f32): the hi plane alone is the f16 rounding of the value
  // (SITE_HI), and an unrepresentable value becomes inf, which the range guard reads off the
  // bits.  lo = (a - hi) * 2^11, truncated (v_cvt_pkrtz: <= 2^-21 relative, inside the budget -
  // tests: fp32-class vs fp64), formed as a * 2^11 - hi * 2^11: both products and their
  // difference are exact (a - hi is representable), and the f16 operand goes straight into
  // v_fma_mix_f32 - 6 VALU per pair instead of 8 (no v_cvt_f32_f16).
  // (An inline-asm v_fma_mixlo_f16 / v_fma_mixhi_f16 pair - no pack instruction - was tried and
  //  removed: same speed in a one-process A/B, and hipcc pads no hazards around asm: a
  //  v_fma_mixhi_f16 result consumed by an MFMA one issue slot later gave timing-dependent
  //  linear-attention states, 116 of 46 944 forwards.)
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f ab = v2f{a, b};
  hi = __builtin_convertvector(ab, f16x2);
  const v2f sc = ab * v2f{SPLIT_SCALE, SPLIT_SCALE};   // (one v_pk_mul_f32)
  lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(
                                     __builtin_fmaf((float)hi[0], -SPLIT_SCALE, sc[0]),
                                     __builtin_fmaf((float)hi[1], -SPLIT_SCALE, sc[1])));
}
// Range guard of the f16-based modes (GM_SPLIT, GM_F16): every activation that is converted
// into a GEMM operand leaves its f16 bit pattern in a running maximum (Range, two VALU per
// two values); a kernel ends with range_report(), which sets FLAG_F16_RANGE in the
// workspace's status word (one atomicOr, only when violated) if any operand could not be
// represented (|x| >= 65520 rounds to inf; NaN counts too).  The host reads the word with
// oetr_query_flags() / oetr_read_flags_async().  Weights are checked at create.
constexpr uint32_t FLAG_F16_RANGE = 1u;   // == OETR_FLAG_F16_RANGE
constexpr uint32_t FLAG_EXCHANGE = 2u;    // == OETR_FLAG_EXCHANGE (decoder.hip: exchange_sum)
constexpr uint32_t FLAG_PUBLISHED = 0x80000000u;   // == OETR_FLAG_PUBLISHED: set in every word stored into a flag slot
// Status block of a workspace (include/oetr_hip.h: OETR_WORKSPACE_STATUS_BYTES, zeroed by
// oetr_workspace_init): word 0 = flags; words 16..31 = per-image call counters of the split
// decoder; from byte 256 its exchange granules [16 images][5][4][256] x 8 bytes.
constexpr int DEC_SPLIT_K = 4, DEC_SPLIT_MAX_IMAGES = 16, DEC_SPLIT_EXCHANGES = 5;
constexpr int DEC_DBG_FAULT = 0x5a;   // DecLaunch::dbg: oetr_debug_decoder_fault (decoder.hip: exchange_sum)
constexpr size_t STATUS_EPOCH_WORD = 16, STATUS_XCH_OFFSET = 256,
                 STATUS_XCH_BYTES = (size_t)DEC_SPLIT_MAX_IMAGES * DEC_SPLIT_EXCHANGES * DEC_SPLIT_K * 256 * 8,
                 STATUS_BYTES = STATUS_XCH_OFFSET + STATUS_XCH_BYTES;
constexpr float F16_MAX = 65504.0f;
constexpr bool gm_f16_range(int m) { return m == GM_SPLIT || m == GM_F16; }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
struct Range {
  // Running maximum of the f16 BIT PATTERNS (sign cleared) of every converted pair, per lane:
  // v_and + v_pk_max_u16 per pair, no scalar work.  The conversions round to nearest, so a
  // value the format cannot hold arrives as inf (0x7c00; NaN is larger) and the ordering of
  // f16 magnitudes is the ordering of their bit patterns.  (Round 2 kept a wave-uniform mask
  // in SGPRs - v_max + v_cmp + s_or per pair and ~450 SGPR spills per kernel.)
  //
  // Round 3: where the fp32 SOURCE values are at hand (cvt_planes2 - every conversion of the
  // encoder / heads / neck GEMM operands) the guard is one v_max3_f32 per pair on |a|, |b|
  // (source modifiers) instead of two instructions on the converted bits: a value rounds to
  // f16 inf exactly when |x| >= 65520.  NaN SOURCES are not seen by this form (max drops
  // them); a NaN can only arise downstream of an inf, which is seen - or come in with the
  // caller's features, in which case the boxes are NaN and no precision helps.
  uint32_t mx = 0;
  float fm = 0.f;
  __device__ __forceinline__ void see2(float a, float b) {
    fm = __builtin_fmaxf(__builtin_fmaxf(fm, __builtin_fabsf(a)), __builtin_fabsf(b));
  }
  __device__ __forceinline__ void see_hi(uint32_t hi_pair) {
    mx = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, mx),
                                                                __builtin_bit_cast(u16x2, hi_pair & 0x7fff7fffu)));
  }
  __device__ __forceinline__ bool bad() const {
    return (mx & 0xffffu) >= 0x7c00u || (mx >> 16) >= 0x7c00u || !(fm < 65520.0f);
  }
};
template <int M>
__device__ __forceinline__ void range_report(const Range& rg, uint32_t* flags) {
  if constexpr (gm_f16_range(M)) {
    if (__builtin_amdgcn_ballot_w64(rg.bad()) != 0 && (threadIdx.x & 63) == 0) atomicOr(flags, FLAG_F16_RANGE);
  }
}
// Two floats -> the mode's operand representation: `hi` = the (only, or high) plane's two
// 16-bit values, `lo` = the low plane's (GM_SPLIT only).  Single-plane modes round to
// nearest even (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32).
template <int M>
__device__ __forceinline__ void cvt_planes2(float a, float b, uint32_t& hi, uint32_t& lo, Range& rg) {
  if constexpr (M == GM_SPLIT) {
    f16x2 h, l;
    split2(a, b, h, l);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
  } else if constexpr (M == GM_F16) {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
    lo = 0;
  } else {
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
    lo = 0;
  }
  if constexpr (gm_f16_range(M)) rg.see2(a, b);
}
// One 32x32x16 MFMA on raw 16-byte fragments in the mode's element type.
template <int M>
__device__ __forceinline__ f32x16 mma16(const f32x4& a, const f32x4& b, const f32x16& c) {
  if constexpr (M == GM_BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// 8 floats -> one 16-byte MFMA fragment per plane of the fp32-class split
// (a = hi + lo/2^11): the operands of the small attention blocks are formed in
// registers, straight from accumulators / f32 LDS tiles.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8(const f32x4& a0, const f32x4& a1, f32x4& hi, f32x4& lo,
                                       Range& rg) {
  uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
  cvt_planes2<GM_SPLIT>(a0[0], a0[1], h0, l0, rg);
  cvt_planes2<GM_SPLIT>(a0[2], a0[3], h1, l1, rg);
  cvt_planes2<GM_SPLIT>(a1[0], a1[1], h2, l2, rg);
  cvt_planes2<GM_SPLIT>(a1[2], a1[3], h3, l3, rg);
  hi = __builtin_bit_cast(f32x4, u32x4{h0, h1, h2, h3});
  lo = __builtin_bit_cast(f32x4, u32x4{l0, l1, l2, l3});
}
constexpr float FA_SH = 4.0f;   // all-pairs attention: log2 of the common factor P, O and l carry (attention.hip: m_run)
// Two floats -> (hi, lo) f16 pairs with an UNSCALED lo plane: hi = RNE f16(x), lo = RNE f16(x - hi) (the difference
// is exact in f32; v_fma_mix_f32 reads the f16 half directly; v_cvt_pk_f16_f32 for both planes).  x = hi + lo to
// <= 2^-23 relative wherever lo is a normal f16 number; below that (|lo| < 2^-14, i.e. |x| < ~0.25) lo is a DENORMAL
// with absolute error <= 2^-25 - v_mfma_f32_32x32x16_f16 honours f16 denormal inputs on gfx950
// (tools/mfma_denorm_probe.hip, run on the box: profiles/r6_full_attention_steps.txt).  Users: the all-pairs
// attention (attention.hip: k_full_attention_split; encoder.hip: full_attention_tile).
// A product against such a plane has the scale of the hi . hi product and accumulates into the SAME accumulator.
// `neg1` = -1.0f held in an SGPR the compiler cannot see through (with the literal hipcc rewrites the fma into
// v_cvt_f32_f16 + v_sub_f32: two instructions per element instead of one).
__device__ __forceinline__ void split2u(float a, float b, float neg1, uint32_t& hi, uint32_t& lo, Range& rg) {
  const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{__builtin_fmaf((float)h[0], neg1, a),
                                                                     __builtin_fmaf((float)h[1], neg1, b)}, f16x2));
  rg.see2(a, b);
}
// main += ah.bh ; cross += ah.bl + al.bh   (one k16 step of a split product; ONE cross
// accumulator: these blocks are 2-4 steps long and short of registers, the dependent
// cross MFMAs cost a few stalled cycles)
//
// The operands of these MFMAs come straight out of VALU conversions (split2), not from LDS or a
// load.  Left to hipcc's scheduler the conversions of `al` sit between the MFMAs, an operand
// register is written one or two issue slots before the MFMA that reads it and rewritten right
// after - legal by hipcc's hazard tables and by the hardware probes (tools/mfma_hazard_probe.hip:
// every VALU producer needs ONE wait state before the MFMA, overwriting a source right after it
// is safe, with or without a sibling MFMA stream; tools/mfma_branch_hazard_probe.hip: a VALU read
// of the result needs 12 states for the last accumulator register, 6 for the first, and hipcc
// provides them).  Round 3 fenced these triples (every operand complete OETR_SPLIT3_PAD + 1
// states before the first MFMA, the three MFMAs back to back) when the split-f16 KV state
// returned timing-dependent results; round 4 showed the fences were NOT what removed the
// failures (the fenced two-path state still failed 7 of 37 000 at s_setprio 3; what every
// failing build shares is a run-time branch between two forms of the state code, see
// encoder.hip).  The attention apply - the only user left - has one code path and 0 differing
// of 858 000 forwards with the fences; they stay as they were measured (0.3 us per launch).
#ifndef OETR_SPLIT3_PAD
#define OETR_SPLIT3_PAD 7
#endif
#define OETR_STR2(x) #x
#define OETR_STR(x) OETR_STR2(x)
template <bool FENCE = true>
__device__ __forceinline__ void mma16_split3(const f32x4& ah, const f32x4& al, const f32x4& bh,
                                             const f32x4& bl, f32x16& main, f32x16& cross) {
  if constexpr (FENCE) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop " OETR_STR(OETR_SPLIT3_PAD));
    __builtin_amdgcn_sched_barrier(0);
  }
  cross = mma16<GM_SPLIT>(ah, bl, cross);
  main = mma16<GM_SPLIT>(ah, bh, main);
  cross = mma16<GM_SPLIT>(al, bh, cross);
  if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
}

// Store 4 consecutive floats of a row as 4 16-bit values per plane (8-byte stores).
// Planes are typed _Float16* for storage only (bf16 bit patterns in GM_BF16).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int M, bool LO = true>
__device__ __forceinline__ void store_planes4(_Float16* hi_row, _Float16* lo_row, int c,
                                              const f32x4& v, Range& rg) {
  uint32_t h0, l0, h1, l1;
  cvt_planes2<M>(v[0], v[1], h0, l0, rg);
  cvt_planes2<M>(v[2], v[3], h1, l1, rg);
  *reinterpret_cast<u32x2*>(hi_row + c) = u32x2{h0, h1};
  if constexpr (gm_planes(M) == 2 && LO) *reinterpret_cast<u32x2*>(lo_row + c) = u32x2{l0, l1};
}
__device__ __forceinline__ void store_split4(_Float16* hi_row, _Float16* lo_row, int c,
                                             const f32x4& v, Range& rg) {
  store_planes4<GM_SPLIT>(hi_row, lo_row, c, v, rg);
}

template <int M, int NT, int U>
struct GemmRegsH {
  f32x4 bh[U][NT], bl[U][gm_planes(M) == 2 ? NT : 1];  // 8 halves each (raw 16-byte fragments)
  f32x4 ah[U], al[gm_planes(M) == 2 ? U : 1];
};

// acc[t] += A[32 x K] * W_tile(nt0 + t), operands in the mode's 16-bit planes.
//   Ahi/Alo: LDS planes, row stride lda halves (Alo: GM_SPLIT only).
//   Whi/Wlo: weights in 16-bit fragment order: 16-byte unit ((ntile*K/16 + ks)*64 + lane)
//            holds, for output column n = 32*ntile + (lane&31), inputs
//            k = 16*ks + 8*(lane>>5) + {0..7}.
//   GM_SPLIT: acc = main + cross/2^11 is formed at the end; `acc` enters as the initial
//   main part.
#ifndef OETR_SPLIT_DEPTH
#define OETR_SPLIT_DEPTH 3   // register buffers in the weight/activation prefetch ring
#endif
#ifndef OETR_SPLIT_U1
#define OETR_SPLIT_U1 2   // chunk depth (k16 steps) when a wave owns one n-tile (8-wave shape)
#endif
template <int M, int K, int NT, int U = (NT == 1 ? OETR_SPLIT_U1 : 4)>
__device__ __forceinline__ void gemm_rows32_h(const _Float16* __restrict__ Ahi,
                                              const _Float16* __restrict__ Alo, int lda,
                                              const f32x4* __restrict__ Whi,
                                              const f32x4* __restrict__ Wlo, int nt0, int lane,
                                              f32x16 (&acc)[NT]) {
  constexpr bool TWO = gm_planes(M) == 2;
  constexpr int KS = K / 16;
  constexpr int NCH = KS / U;
  static_assert(KS % (2 * U) == 0, "K must be a multiple of 32*U");
  const int a_off = (lane & 31) * lda + 8 * (lane >> 5);
  const _Float16* ah_ptr = Ahi + a_off;
  const _Float16* al_ptr = Alo + a_off;
  const f32x4* wh_ptr[NT];
  const f32x4* wl_ptr[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    wh_ptr[t] = Whi + (size_t)(nt0 + t) * KS * 64;   // (uniform when nt0 is: lane comes last)
    wl_ptr[t] = Wlo + (size_t)(nt0 + t) * KS * 64;
  }
  // two cross accumulators per tile: three independent MFMA chains per k-step
  // (a single one makes every other MFMA wait on its predecessor's result)
  f32x16 cross[TWO ? NT : 1], cross2[TWO ? NT : 1];
  if constexpr (TWO) {
#pragma unroll
    for (int t = 0; t < NT; ++t) { cross[t] = f32x16{0}; cross2[t] = f32x16{0}; }
  }

  const unsigned ln = (unsigned)lane;
  auto fetch = [&](GemmRegsH<M, NT, U>& r, int chunk) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        r.bh[u][t] = wh_ptr[t][(chunk * U + u) * 64 + ln];
        if constexpr (TWO) r.bl[u][t] = wl_ptr[t][(chunk * U + u) * 64 + ln];
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      r.ah[u] = *reinterpret_cast<const f32x4*>(ah_ptr + (chunk * U + u) * 16);
      if constexpr (TWO) r.al[u] = *reinterpret_cast<const f32x4*>(al_ptr + (chunk * U + u) * 16);
    }
  };
  auto mma = [&](const GemmRegsH<M, NT, U>& r) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = mma16<M>(r.ah[u], r.bh[u][t], acc[t]);
        if constexpr (TWO) {
          cross[t] = mma16<M>(r.ah[u], r.bl[u][t], cross[t]);
          cross2[t] = mma16<M>(r.al[u], r.bh[u][t], cross2[t]);
        }
      }
    }
  };

#if OETR_SPLIT_DEPTH == 3
  // ring of three register buffers: two chunks of fragments in flight while the
  // third is consumed (per-CU weight streaming is latency-bound: ~2000 cycles per
  // L2 round trip under the all-workgroups-read-the-same-lines load)
  GemmRegsH<M, NT, U> r0, r1, r2;
  fetch(r0, 0);
  fetch(r1, 1);
  for (int c = 0; c < NCH; c += 3) {
    if (c + 2 < NCH) fetch(r2, c + 2);
    __builtin_amdgcn_sched_barrier(0);
    mma(r0);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < NCH) {
      if (c + 3 < NCH) fetch(r0, c + 3);
      __builtin_amdgcn_sched_barrier(0);
      mma(r1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c + 2 < NCH) {
      if (c + 4 < NCH) fetch(r1, c + 4);
      __builtin_amdgcn_sched_barrier(0);
      mma(r2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#else
  GemmRegsH<M, NT, U> r0, r1;
  fetch(r0, 0);
  for (int c = 0; c < NCH; c += 2) {
    fetch(r1, c + 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(r0);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 2 < NCH) fetch(r0, c + 2);
    __builtin_amdgcn_sched_barrier(0);
    mma(r1);
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
  if constexpr (TWO) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(cross[t][r] + cross2[t][r], SPLIT_INV, acc[t][r]);
  }
}

// Accumulator tiles -> 16-bit planes (columns col0 + 32*t + lane&31).
template <int M, int NT>
__device__ __forceinline__ void acc_to_lds_planes(_Float16* Shi, _Float16* Slo, int lds, int col0,
                                                  int lane, const f32x16 (&acc)[NT], Range& rg) {
  const int half = lane >> 5, c = col0 + (lane & 31);
  uint16_t* H = reinterpret_cast<uint16_t*>(Shi);
  uint16_t* Lo = reinterpret_cast<uint16_t*>(Slo);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      uint32_t hi, lo;
      cvt_planes2<M>(acc[t][r], acc[t][r + 1], hi, lo, rg);
      const int o0 = crow(r, half) * lds + c + 32 * t, o1 = crow(r + 1, half) * lds + c + 32 * t;
      H[o0] = (uint16_t)hi; H[o1] = (uint16_t)(hi >> 16);
      if constexpr (gm_planes(M) == 2) { Lo[o0] = (uint16_t)lo; Lo[o1] = (uint16_t)(lo >> 16); }
    }
}

// Write a wave's two 32x32 accumulator tiles (columns col0 + 32*t + lane&31)
// into a row-major LDS tile.
template <int NT>
__device__ __forceinline__ void acc_to_lds(float* S, int lds, int col0, int lane,
                                           const f32x16 (&acc)[NT]) {
  const int half = lane >> 5, c = col0 + (lane & 31);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) S[crow(r, half) * lds + c + 32 * t] = acc[t][r];
}

// LDS region sizes (floats) that fit either representation of an A tile.
constexpr int TILE_FLOATS = (2 * TM * LDAH * 2 + 3) / 4;    // 8448 >= TM*LDA = 8320
constexpr int HID_FLOATS = (2 * TM * LDHH * 2 + 3) / 4;     // 16640 >= TM*LDH = 16512
static_assert(TILE_FLOATS >= TM * LDA && HID_FLOATS >= TM * LDH, "region sizes");

// A GEMM A-operand tile in LDS, in the representation of mode M: an f32 tile
// (GM_F32) or one / two 16-bit planes.  (Single-plane modes keep the two-plane
// footprint: the regions are shared with the f32 staging tiles anyway.)
template <int M>
struct ATile {
  static constexpr bool HALF = gm_half(M);
  float* f;        // f32 mode
  _Float16 *h, *l; // 16-bit planes (l: GM_SPLIT only)
  int ldf, ldh;
  Range* rg;       // the thread's range guard (see Range)
  __device__ __forceinline__ ATile(float* base, int ldf_, int ldh_, Range* rg_)
      : f(base), h(reinterpret_cast<_Float16*>(base)),
        l(reinterpret_cast<_Float16*>(base) + TM * ldh_), ldf(ldf_), ldh(ldh_), rg(rg_) {}
  // 4 consecutive values of one row (LO = false: the consumer site reads the hi plane only)
  template <bool LO = true>
  __device__ __forceinline__ void put4(int row, int c, const f32x4& v) const {
    if constexpr (HALF) store_planes4<M, LO>(h + row * ldh, l + row * ldh, c, v, *rg);
    else *reinterpret_cast<f32x4*>(f + row * ldf + c) = v;
  }
  template <int NT>
  __device__ __forceinline__ void put_acc(int col0, int lane, const f32x16 (&acc)[NT]) const {
    if constexpr (HALF) acc_to_lds_planes<M, NT>(h, l, ldh, col0, lane, acc, *rg);
    else acc_to_lds<NT>(f, ldf, col0, lane, acc);
  }
  template <int K, int NT>
  __device__ __forceinline__ void gemm(const f32x4* w, const f32x4* w_lo, int nt0, int lane,
                                       f32x16 (&acc)[NT], int dbg) const {
    if constexpr (HALF) gemm_rows32_h<M, K, NT>(h, l, ldh, w, w_lo, nt0, lane, acc);
    else gemm_rows32<K, NT>(f, ldf, w, nt0, lane, acc, dbg);
  }
};

// Sum over the TPR (8 or 16) consecutive lanes that share a row.
template <int TPR>
__device__ __forceinline__ float row_sum(float v) {
  v = sum8(v);
  if (TPR == 16) v += dpp_mov<0x140>(v);  // row_mirror: lane i <-> 15 - i
  return v;
}

// LayerNorm over a [TM][256] LDS tile with TPR threads per row: thread tid owns
// row tid/TPR and the float4 columns i*TPR + tid%TPR (the threads of a row read
// contiguous 16-byte pieces per step).  Row sums need only 3-4 DPP exchanges.
// Returns (x - mean) * rstd in registers; the caller applies its affine(s).
template <int TPR, int F4>
__device__ __forceinline__ void ln_rows(const float* S, int tid, f32x4 (&xn)[F4], int dbg) {
  const f32x4* src = reinterpret_cast<const f32x4*>(S + (tid / TPR) * LDA) + (tid % TPR);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < F4; ++i) {
    xn[i] = src[i * TPR];
    s += (xn[i][0] + xn[i][1]) + (xn[i][2] + xn[i][3]);
  }
  if (ABL(dbg, ABL_LN)) return;
  const float mean = row_sum<TPR>(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < F4; ++i) {
    xn[i] -= mean;
    q += (xn[i][0] * xn[i][0] + xn[i][1] * xn[i][1]) + (xn[i][2] * xn[i][2] + xn[i][3] * xn[i][3]);
  }
  const float rstd = 1.0f / sqrtf(row_sum<TPR>(q) * (1.0f / C) + LN_EPS);
#pragma unroll
  for (int i = 0; i < F4; ++i) xn[i] *= rstd;
}

// ---------------------------------------------------------------------------
// Weight stream of one wave across the GEMMs of a kernel.
//
// A GEMM that fetches its first B fragments when it is called starts with an
// exposed L2 round trip (~2000 cycles under the all-workgroups-read-the-same-
// lines load) and every isolated call pays it: 7 calls per B;A launch.  The
// stream keeps a ring of D B-fragment chunks in registers that runs ACROSS
// calls: while a GEMM consumes its last chunks it already fetches the first
// PRE = D-1 chunks of the next GEMM's weights, so those are in flight during
// the epilogue / LayerNorm / barrier between the two.  A fragments come from
// LDS one chunk ahead.  P is the ring slot of a GEMM's chunk 0 (compile time);
// adv(P, K) is the slot the following GEMM starts at.
//
// Generic form (exact-f32 mode, or a wave owning several n-tiles): no run-
// ahead, plain calls.
// ---------------------------------------------------------------------------
#ifndef OETR_WSTREAM
#define OETR_WSTREAM 1
#endif
// Ring geometry (tools/variants A/B on MI355X, k_encoder<B,A> at 8 pairs @640x640, split
// mode): chunks of 2 k16-steps, ring of 3: 46.1 us; ring of 4: 50.7 (registers); chunks
// of ONE step, ring of 6 (5 steps = 10 KB per wave in flight): 42.3; ring of 8: 48.4.
// The single-plane modes stream at ~47 B/clk/CU whatever the depth (L1-rate bound).
#ifndef OETR_RING
#define OETR_RING 6     // ring depth (chunks) of the two-plane (split) weight stream
#endif
#ifndef OETR_RING1
#define OETR_RING1 6    // ring depth of the single-plane (f16 / bf16) weight stream
#endif
#ifndef OETR_WS_U
#define OETR_WS_U 1     // k16 steps per chunk
#endif
template <int M, int NT, bool STREAMED = (gm_half(M) && NT == 1)>
struct WStream {
  static constexpr int adv(int, int) { return 0; }
  __device__ __forceinline__ void set_lane(int) {}
  template <int K, int P, int SITE = SITE_FULL>
  __device__ __forceinline__ void prime(const f32x4*, const f32x4*, int, int) {}
  template <int K, int P, int NK, int SITE = SITE_FULL, int NSITE = SITE_FULL, class AT>
  __device__ __forceinline__ void gemm(const AT& A, const f32x4* W, const f32x4* Wl, int nt0,
                                       int lane, f32x16 (&acc)[NT], const f32x4*, const f32x4*,
                                       int, int dbg) {
    static_assert(SITE == SITE_FULL, "reduced sites need the streamed one-n-tile-per-wave shape");
    A.template gemm<K, NT>(W, Wl, nt0, lane, acc, dbg);
  }
};

#if OETR_WSTREAM
template <int M>
struct WStream<M, 1, true> {
  static constexpr bool TWO = gm_planes(M) == 2;
  static constexpr int U = OETR_WS_U, D = TWO ? OETR_RING : OETR_RING1, PRE = D - 1;
  struct BChunk { f32x4 bh[U], bl[TWO ? U : 1]; };
  struct AChunk { f32x4 ah[U], al[TWO ? U : 1]; };
  BChunk ring[D];
  int dbg = 0;  // ablation flags (OETR_ABLATE builds only)
  // lane as the LAST, 32-bit index of every fragment address (see WStream2T::ln): with a scalar
  // wave index the slab pointers are uniform and the loads take SGPR base + lane offset
  unsigned ln = 0;
  __device__ __forceinline__ void set_lane(int lane) { ln = (unsigned)lane; }

  static constexpr int adv(int P, int K) { return (P + K / 16 / U) % D; }

  template <int SLOT, int SITE>
  __device__ __forceinline__ void fetch(const f32x4* wh, const f32x4* wl, int chunk) {
    if (ABL(dbg, ABL_WLOAD)) return;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ring[SLOT].bh[u] = wh[(chunk * U + u) * 64 + ln];
      if constexpr (TWO && site_w_lo(SITE)) ring[SLOT].bl[u] = wl[(chunk * U + u) * 64 + ln];
    }
  }
  template <int P, int J, int SITE>
  __device__ __forceinline__ void fetch_first(const f32x4* wh, const f32x4* wl) {
    if constexpr (J < PRE) {
      fetch<(P + J) % D, SITE>(wh, wl, J);
      fetch_first<P, J + 1, SITE>(wh, wl);
    }
  }
  // Issue the first PRE chunks of a GEMM's weights (K = its reduction length, SITE = its
  // arithmetic: a site that does not use the weights' lo plane does not fetch it).
  template <int K, int P, int SITE = SITE_FULL>
  __device__ __forceinline__ void prime(const f32x4* W, const f32x4* Wl, int nt0, int lane) {
    const size_t off = (size_t)nt0 * (K / 16) * 64;
    (void)lane;
    fetch_first<P, 0, SITE>(W + off, Wl + off);
    __builtin_amdgcn_sched_barrier(0);
  }

  template <int K, int P, int NK, int CI, int SITE, int NSITE>
  __device__ __forceinline__ void step(const _Float16* ah_ptr, const _Float16* al_ptr,
                                       const f32x4* wh, const f32x4* wl, const f32x4* nwh,
                                       const f32x4* nwl, AChunk (&a)[2], f32x16& acc,
                                       f32x16& cross, f32x16& cross2) {
    constexpr int NCH = K / 16 / U;
    if constexpr (CI < NCH) {
      constexpr int PF = CI + PRE;
      if constexpr (PF < NCH) fetch<(P + PF) % D, SITE>(wh, wl, PF);
      else if constexpr (NK != 0) fetch<(P + PF) % D, NSITE>(nwh, nwl, PF - NCH);
      if constexpr (CI + 1 < NCH) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          a[(CI + 1) & 1].ah[u] = *reinterpret_cast<const f32x4*>(ah_ptr + ((CI + 1) * U + u) * 16);
          if constexpr (TWO && site_act_lo(SITE))
            a[(CI + 1) & 1].al[u] = *reinterpret_cast<const f32x4*>(al_ptr + ((CI + 1) * U + u) * 16);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const BChunk& b = ring[(P + CI) % D];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc = mma16<M>(a[CI & 1].ah[u], b.bh[u], acc);
        if constexpr (TWO && site_w_lo(SITE)) cross = mma16<M>(a[CI & 1].ah[u], b.bl[u], cross);
        if constexpr (TWO && site_act_lo(SITE)) cross2 = mma16<M>(a[CI & 1].al[u], b.bh[u], cross2);
      }
      __builtin_amdgcn_sched_barrier(0);
      step<K, P, NK, CI + 1, SITE, NSITE>(ah_ptr, al_ptr, wh, wl, nwh, nwl, a, acc, cross, cross2);
    }
  }

  // acc += A . W^T for this wave's n-tile nt0; chunks 0..PRE-1 of W are already in
  // the ring (prime<K,P> or the previous gemm's NK).  NK != 0: the next GEMM has
  // reduction length NK, weights (nW, nWl, nnt0) and arithmetic NSITE; its first chunks are
  // fetched while this one finishes.  SITE: this GEMM's arithmetic (SITE_*).
  template <int K, int P, int NK, int SITE = SITE_FULL, int NSITE = SITE_FULL, class AT>
  __device__ __forceinline__ void gemm(const AT& A, const f32x4* W, const f32x4* Wl, int nt0,
                                       int lane, f32x16 (&acc)[1], const f32x4* nW,
                                       const f32x4* nWl, int nnt0, int) {
    if (ABL(dbg, ABL_GEMM)) return;
    static_assert(PRE * U * 16 <= K && (NK == 0 || PRE * U * 16 <= NK), "ring deeper than a GEMM");
    const size_t off = (size_t)nt0 * (K / 16) * 64;
    const size_t noff = (size_t)nnt0 * ((NK ? NK : 16) / 16) * 64;
    const int a_off = (lane & 31) * A.ldh + 8 * (lane >> 5);
    const _Float16* ah_ptr = A.h + a_off;
    const _Float16* al_ptr = A.l + a_off;
    AChunk a[2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[0].ah[u] = *reinterpret_cast<const f32x4*>(ah_ptr + u * 16);
      if constexpr (TWO && site_act_lo(SITE)) a[0].al[u] = *reinterpret_cast<const f32x4*>(al_ptr + u * 16);
    }
    f32x16 cross = {0}, cross2 = {0};
    step<K, P, NK, 0, SITE, NSITE>(ah_ptr, al_ptr, W + off, Wl + off, NK ? nW + noff : nullptr,
                                   NK ? nWl + noff : nullptr, a, acc[0], cross, cross2);
    if constexpr (TWO && SITE != SITE_HI) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] = fmaf(cross[r] + cross2[r], SPLIT_INV, acc[0][r]);
    }
  }
};
#endif  // OETR_WSTREAM

// ---- 64-token workgroup shape (k_encoder64, conv_p_body64) ----
constexpr int RT = 64;                                   // token rows per workgroup
constexpr int R_FLOATS = (2 * RT * LDAH * 2 + 3) / 4;    // 16896 >= RT * LDA = 16640
static_assert(R_FLOATS >= RT * LDA, "region holds the f32 tile too");
template <int M, int ROWS_P = RT>
struct PlanesT {  // 16-bit planes [ROWS_P][LDAH] (hi, and lo*2^11 in GM_SPLIT) in one region; ROWS_P = 64 or 32 token rows
  _Float16 *h, *l;
  Range* rg;
  __device__ __forceinline__ PlanesT(float* base, Range* rg_)
      : h(reinterpret_cast<_Float16*>(base)), l(reinterpret_cast<_Float16*>(base) + ROWS_P * LDAH), rg(rg_) {}
  // a window of a larger plane pair (row stride LDAH): first row of the hi plane, first row of the lo plane
  __device__ __forceinline__ PlanesT(_Float16* hi_row0, _Float16* lo_row0, Range* rg_) : h(hi_row0), l(lo_row0), rg(rg_) {}
  // LO = false: the consumer site reads the hi plane only (common.h: SITE_*), the lo plane is
  // not written
  template <bool LO = true>
  __device__ __forceinline__ void put4(int row, int c, const f32x4& v) const {
    store_planes4<M, LO>(h + row * LDAH, l + row * LDAH, c, v, *rg);
  }
  // accumulator registers r, r + 1 (rows crow(r), crow(r + 1) of row tile mt, column col0 + lane & 31)
  template <int R, bool LO = true>
  __device__ __forceinline__ void put_pair(int mt, int col0, int lane, float v0, float v1) const {
    const int half = lane >> 5, c = col0 + (lane & 31);
    uint32_t hi, lo;
    cvt_planes2<M>(v0, v1, hi, lo, *rg);
    uint16_t* H = reinterpret_cast<uint16_t*>(h) + mt * 32 * LDAH;
    uint16_t* Lo = reinterpret_cast<uint16_t*>(l) + mt * 32 * LDAH;
    const int o0 = crow(R, half) * LDAH + c, o1 = crow(R + 1, half) * LDAH + c;
    H[o0] = (uint16_t)hi; H[o1] = (uint16_t)(hi >> 16);
    if constexpr (gm_planes(M) == 2 && LO) { Lo[o0] = (uint16_t)lo; Lo[o1] = (uint16_t)(lo >> 16); }
  }
  __device__ __forceinline__ void put_acc(int mt, int col0, int lane, const f32x16& acc) const {
    acc_to_lds_planes<M, 1>(h + mt * 32 * LDAH, l + mt * 32 * LDAH, LDAH, col0, lane,
                            *reinterpret_cast<const f32x16(*)[1]>(&acc), *rg);
  }
};
typedef PlanesT<GM_SPLIT> Planes2;

#ifndef OETR_RING2
#define OETR_RING2 4   // k16 steps of B fragments in the ring (one being consumed), split mode
#endif
#ifndef OETR_RING2_1P
#define OETR_RING2_1P 4   // the same for the single-plane modes
#endif
#ifndef OETR_RING2_32
#define OETR_RING2_32 6   // ... and for the 32-row form of the body (NMT = 1)
#endif
struct NoEpi { template <class T> __device__ __forceinline__ void operator()(T) const {} };

// ROWS: how many of the tile's two 32-row MFMA tiles run.  2 / 1: fixed at compile time - the
// GEMM steps are branch-free, which is what lets hipcc interleave a step's MFMAs with the
// epilogue slice issued beside them (EPI below); the encoder picks the body per workgroup.
// 0: decided at run time from the number of valid rows (a wave-uniform branch around every
// second-tile MFMA: conv-P work items, single-plane modes - measured better there: their steps
// are two MFMAs long).  With one row tile the accumulators of the second keep their (finite)
// initial values, which every consumer masks by row validity.
// NMT: MFMA row tiles of the workgroup's token tile - 2 (64 token rows, the shape all of the above
// describes) or 1 (32 token rows: round 4's form of the 32-row encoder - the same body with one
// row tile, so a fragment feeds 3 MFMAs instead of 6 and the GEMM phases are bound by the weight
// stream again; ROWS is then irrelevant).
template <int M, int ROWS = 0, int NMT = 2>
struct WStream2T {
  static constexpr bool TWO = gm_planes(M) == 2;
  // (one row tile: the GEMM phases are bound by the weight stream, which wants more fragments in flight -
  //  and the registers of the second row tile are free for them)
  static constexpr int D = NMT == 1 ? OETR_RING2_32 : TWO ? OETR_RING2 : OETR_RING2_1P, PRE = D - 1, NS = C / 16;  // every GEMM here has K = 256: 16 steps
  struct BStep { f32x4 bh, bl; };
  struct AStep { f32x4 ah[NMT], al[NMT]; };
  BStep ring[D];
  bool two_rt = true;
  // lane as the LAST, 32-bit index of every fragment address: the slab pointers stay uniform
  // (the wave index is a scalar), so the loads take an SGPR base + lane offset + immediate and
  // the per-step address arithmetic is SALU work
  unsigned ln = 0;
  __device__ __forceinline__ void set_lane(int lane) { ln = (unsigned)lane; }
  __device__ __forceinline__ bool two() const { return NMT == 2 && (ROWS == 2 || (ROWS == 0 && two_rt)); }
  __device__ __forceinline__ void set_rows(int nvalid) {
    if constexpr (ROWS == 0 && NMT == 2) two_rt = __builtin_amdgcn_readfirstlane(nvalid > 32 ? 1 : 0) != 0;
  }
  static constexpr int adv(int P) { return (P + NS) % D; }

  template <int SLOT, int SITE>
  __device__ __forceinline__ void fetch(const f32x4* wh, const f32x4* wl, int step) {
    ring[SLOT].bh = wh[step * 64 + ln];
    if constexpr (TWO && site_w_lo(SITE)) ring[SLOT].bl = wl[step * 64 + ln];
  }
  template <int P, int J, int SITE>
  __device__ __forceinline__ void fetch_first(const f32x4* wh, const f32x4* wl) {
    if constexpr (J < PRE) {
      fetch<(P + J) % D, SITE>(wh, wl, J);
      fetch_first<P, J + 1, SITE>(wh, wl);
    }
  }
  // First PRE steps of the weight slab (n-tile nt0, k16-steps from ks0) of a matrix
  // packed with KTOT/16 steps per n-tile.
  template <int KTOT, int P, int SITE = SITE_FULL>
  __device__ __forceinline__ void prime(const f32x4* W, const f32x4* Wl, int nt0, int ks0, int lane) {
    const size_t off = ((size_t)nt0 * (KTOT / 16) + ks0) * 64;
    (void)lane;
    fetch_first<P, 0, SITE>(W + off, Wl + off);
    __builtin_amdgcn_sched_barrier(0);
  }
  template <int SITE>
  __device__ __forceinline__ static void load_a(AStep& a, const _Float16* ah_ptr, const _Float16* al_ptr,
                                                int step) {
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      a.ah[mt] = *reinterpret_cast<const f32x4*>(ah_ptr + mt * 32 * LDAH + step * 16);
      if constexpr (TWO && site_act_lo(SITE))
        a.al[mt] = *reinterpret_cast<const f32x4*>(al_ptr + mt * 32 * LDAH + step * 16);
    }
  }
  // EPI: work of ANOTHER GEMM's epilogue (or any VALU / LDS / store work independent of this
  // GEMM) issued in 16 slices, one per k16 step - epi(std::integral_constant<int, CI>{}) for
  // CI = 0..15 - inside the step's scheduling region and BEFORE its MFMAs in program order:
  // hipcc then spreads the slice over the gaps between the MFMAs (placed after them it issues
  // all MFMAs first and the slice behind them - both waves of a SIMD then alternate in
  // lockstep between an MFMA-only and a VALU-only stretch and nothing overlaps; measured).
  // TR: the TRANSPOSED product - weights as the MFMA A operand, activations as B - so that the
  // accumulator of lane (l & 31, half) holds token (l & 31) x channels crow(r, half) of the
  // n-tile (instead of channel (l & 31) x tokens crow(r, half)).  Same fragments, same LDS
  // reads, same MFMA count.
  template <int P, bool HAS_NEXT, int CI, class EPI, bool TR, int SITE, int NSITE>
  __device__ __forceinline__ void step(const _Float16* ah_ptr, const _Float16* al_ptr,
                                       const f32x4* wh, const f32x4* wl, const f32x4* nwh,
                                       const f32x4* nwl, AStep (&a)[2], f32x16 (&acc)[NMT],
                                       f32x16 (&cross)[NMT], EPI& epi) {
    if constexpr (CI < NS) {
      constexpr int PF = CI + PRE;
      if constexpr (PF < NS) fetch<(P + PF) % D, SITE>(wh, wl, PF);
      else if constexpr (HAS_NEXT) fetch<(P + PF) % D, NSITE>(nwh, nwl, PF - NS);
      if constexpr (CI + 1 < NS) load_a<SITE>(a[(CI + 1) & 1], ah_ptr, al_ptr, CI + 1);
      __builtin_amdgcn_sched_barrier(0);
#if defined(OETR_SOAK_AMP) && (OETR_SOAK_AMP & 1)   // soak builds only: the waves of a workgroup drift apart behind their loads
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      const BStep& b = ring[(P + CI) % D];
      const AStep& ac = a[CI & 1];
      epi(std::integral_constant<int, CI>{});
      auto mm = [](const f32x4& act, const f32x4& wgt, const f32x16& c) {
        if constexpr (TR) return mma16<M>(wgt, act, c);
        else return mma16<M>(act, wgt, c);
      };
      acc[0] = mm(ac.ah[0], b.bh, acc[0]);
      if constexpr (NMT == 2) { if (two()) acc[1] = mm(ac.ah[1], b.bh, acc[1]); }
      if constexpr (TWO && site_w_lo(SITE)) {
        cross[0] = mm(ac.ah[0], b.bl, cross[0]);
        if constexpr (NMT == 2) { if (two()) cross[1] = mm(ac.ah[1], b.bl, cross[1]); }
      }
      if constexpr (TWO && site_act_lo(SITE)) {
        cross[0] = mm(ac.al[0], b.bh, cross[0]);
        if constexpr (NMT == 2) { if (two()) cross[1] = mm(ac.al[1], b.bh, cross[1]); }
      }
      __builtin_amdgcn_sched_barrier(0);
      step<P, HAS_NEXT, CI + 1, EPI, TR, SITE, NSITE>(ah_ptr, al_ptr, wh, wl, nwh, nwl, a, acc, cross, epi);
    }
  }
  // acc[mt] += A[32*mt .. 32*mt+31][0..255] . Wslab^T  for this wave's n-tile.  The first
  // PRE steps of the slab are already in the ring; HAS_NEXT: the next GEMM's slab
  // (nW, nWl, nnt0, nks0 of a matrix with NKTOT/16 steps per n-tile, arithmetic NSITE) is
  // primed meanwhile.
  template <int KTOT, int P, bool HAS_NEXT, int NKTOT, int SITE = SITE_FULL, int NSITE = SITE_FULL,
            bool TR = false, int RP>
  __device__ __forceinline__ void gemm(const PlanesT<M, RP>& A, const f32x4* W, const f32x4* Wl, int nt0,
                                       int ks0, int lane, f32x16 (&acc)[NMT], const f32x4* nW,
                                       const f32x4* nWl, int nnt0, int nks0) {
    NoEpi none;
    gemm_epi<KTOT, P, HAS_NEXT, NKTOT, SITE, NSITE, TR>(A, W, Wl, nt0, ks0, lane, acc, nW, nWl, nnt0,
                                                        nks0, none);
  }
  template <int KTOT, int P, bool HAS_NEXT, int NKTOT, int SITE = SITE_FULL, int NSITE = SITE_FULL,
            bool TR = false, class EPI, int RP>
  __device__ __forceinline__ void gemm_epi(const PlanesT<M, RP>& A, const f32x4* W, const f32x4* Wl, int nt0,
                                           int ks0, int lane, f32x16 (&acc)[NMT], const f32x4* nW,
                                           const f32x4* nWl, int nnt0, int nks0, EPI& epi) {
    const size_t off = ((size_t)nt0 * (KTOT / 16) + ks0) * 64;
    const size_t noff = ((size_t)nnt0 * (NKTOT / 16) + nks0) * 64;
    const int a_off = (lane & 31) * LDAH + 8 * (lane >> 5);
    const _Float16* ah_ptr = A.h + a_off;
    const _Float16* al_ptr = A.l + a_off;
    AStep a[2];
    load_a<SITE>(a[0], ah_ptr, al_ptr, 0);
    f32x16 cross[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) cross[mt] = f32x16{0};
    step<P, HAS_NEXT, 0, EPI, TR, SITE, NSITE>(ah_ptr, al_ptr, W + off, Wl + off,
                                               HAS_NEXT ? nW + noff : nullptr,
                                               HAS_NEXT ? nWl + noff : nullptr, a, acc, cross, epi);
    if constexpr (TWO && SITE != SITE_HI) {
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = fmaf(cross[mt][r], SPLIT_INV, acc[mt][r]);
    }
  }
};
typedef WStream2T<GM_SPLIT> WStream2;

#endif  // __HIPCC__

// ------------------------------------------------------------------ host
// sets the calling thread's oetr_last_error() text; returns `status` (api.hip)
int set_last_error(int status, const char* msg);
// Repacked weights of one encoder layer (device pointers).
struct EncLayerDev {
  const f32x4 *wq, *wk, *wv, *wmerge, *w1, *w2;  // fragment-packed (f32 mode: values;
                                                 //  split mode: f16 hi plane)
  const f32x4 *wq_l, *wk_l, *wv_l, *wmerge_l, *w1_l, *w2_l;  // split mode: f16 lo plane
  const float *lnq_w, *lnq_b, *lnkv_w, *lnkv_b, *ln2_w, *ln2_b;
};
// Decoder cross-attention K/V projections applied to the encoder memory.
struct DecKVDev {
  const f32x4 *wk[2], *wv[2];
  const f32x4 *wk_l[2], *wv_l[2];  // split mode lo planes
  const float *bk[2], *bv[2];
};

struct EncLaunch {
  Geom g;
  float* x;              // [rows][256] token-major activations (in/out, in place)
  float* qp;             // phi(Q) of the layer being finished / next: [rows][256] token-major (32-row
                         // kernel) or [tiles][64][256] tile-major (64-row kernel)
  const float* pos;      // [L0+L1][256] token-major position table
  const float* kv_in;    // partial KV states of the layer being finished
  const float* ks_in;    //   [ntiles][8192] / [ntiles][256]
  float* kv_out;         // partial KV states for the next layer
  float* ks_out;
  // TAIL==1 (decoder preparation) outputs, per token tile:
  float* att0_out;       //  [ntiles][256] partial message of decoder layer 0's cross-attn
  float* z0_out;         //  [ntiles][8]   partial normaliser  (query is a create-time constant)
  float* dkv1_out;       //  [ntiles][8192] partial KV state for decoder layer 1
  float* dks1_out;       //  [ntiles][256]
  const float* dec_q0;   //  [2][256] phi(Q) of decoder layer 0's cross-attn, per side
  EncLayerDev b;         // layer being finished (phase B), if any
  EncLayerDev a;         // next layer (phase A), TAIL==0
  DecKVDev d;            // TAIL==1
  // attention == full (reference EncoderLayer(attention='full'), transformer.py:86-89):
  // phase A stores raw Q (in qp), K row-major and V transposed instead of phi(Q) and the
  // per-tile KV states; phase B runs flash attention over the source image's tokens
  int attn_full;
  // (ping-pong by layer parity like the partial states: a fast tile already writes layer
  //  l+1's K/V while a slow one still reads layer l's)
  const float* kbuf_in;  // [rows][256] K projections (no phi) of the layer being finished
  const float* vt_in;    // per image [256][lpad] V projections, token index minor
  float* kbuf_out;       // ... of the next layer (phase A)
  float* vt_out;
  int lpad[2];           // tokens per image rounded up to TM, per side
  size_t vt_off[2];      // first float of the side's images in vt
  int tile_rows;         // token rows per workgroup: 32 (TM) or 64 (split mode, k_encoder64);
                         // g.nt / g.tile0 / g.ntiles are in units of this tile
  int b_cross;           // phase-B layer is a cross layer
  int kv_reduced;        // kv_in / ks_in hold ONE reduced state per image ([2N][8192] / [2N][256], side 0
                         // first) instead of per-tile partials: k_kv_reduce ran between the launches
  int policy;            // precision policy id (SitePolicy<>): 0 = every site fp32-class
  int dbg;               // ablation flags (OETR_ABLATE builds only)
  long long* tbuf;       // per-phase cycle stamps (OETR_PHASE_TIMING builds only)
  uint32_t* flags;       // the handle's status word (FLAG_F16_RANGE), see Range
  // First launch only (has_b == false): the inputs as the reference hands them over - NCHW
  // features [N][256][L] and position windows [256][L] per side.  The launch then does the
  // `flatten(2).permute(0, 2, 1)` of transformer.py:338-345 in its tile load (and writes x /
  // the position table back token-major for the later launches) instead of a launch of its
  // own.  NULL: x and pos already hold the token-major data (oetr_forward_tokens).
  const float* feat_nchw[2];
  const float* pos_nchw[2];
  float* pos_out;        // = pos, writable (the token-major table the n == 0 tiles fill in)
  // forward_dummy's optional masks (reference src/model.py:229): per side [N][L] floats, a token's
  // value multiplies its phi(Q) row where it is a query and its phi(K) / V rows where it is a source
  // (linear_attention.py:37-41).  Both NULL = no masks; otherwise both set (GM_SPLIT, policy 0).
  const float* mask[2];
};

// has_b: run phase B (finish a layer); tail: 0 = phase A of next encoder layer,
// 1 = decoder K/V preparation, 2 = nothing.
hipError_t launch_encoder(const EncLaunch& p, bool has_b, int tail, int mode, hipStream_t s);
// Sum the per-tile partial linear-attention states of every image (fixed tile order) once,
// between the launch that wrote them and the launch whose every workgroup would otherwise
// re-reduce them: g = the ENCODER tile geometry; kvr [2N][8192], ksr [2N][256].
hipError_t launch_kv_reduce(const Geom& g, const float* kvp, const float* ksp, float* kvr, float* ksr,
                            hipStream_t s);

struct MhaDev {
  const float *wq_t, *wk_t, *wv_t, *wm_t;  // transposed [in][out]
  const float *bq, *bk, *bv;
};
struct DecLayerDev {
  MhaDev self_attn, cross;
  const float *w1_t, *w2_t;  // [256][512], [512][256] transposed
  const float *n1w, *n1b, *n2w, *n2b, *n3w, *n3b;
};
struct DecLaunch {
  Geom g;
  DecLayerDev layer[2];
  const float* qe[2];      // query embeddings per side [256]
  const float* tgt1;       // [2][256]   create-time constants (k_decoder_consts)
  const float* qkv1;       // [2][768]
  const float* att0_part;  // [ntiles][256] from the encoder tail
  const float* z0_part;    // [ntiles][8]
  const float* dkv1;       // [ntiles][8192]
  const float* dks1;       // [ntiles][256]
  float* hs;               // [2N][256]
  int convp_units;         // beside the conv-P GEMMs: (tile, tap) units per conv work item (api.hip: dec_launch)
  int ksplit;              // workgroups per image: 1, or OETR_DEC_SPLIT_K with the three fields below
  unsigned long long* xch; // [2N][5 exchanges][4][256] {tag, value} granules (workspace status block)
  unsigned* xch_epoch;     // [2N] call counters = tags (workspace status block)
  uint32_t* flags;         // the workspace's status word (OETR_FLAG_EXCHANGE)
  long long* tbuf;         // OETR_PHASE_TIMING builds only
  int dbg;                 // OETR_ABLATE builds; DEC_DBG_FAULT (oetr_debug_decoder_fault)
};
struct DecConstLaunch {
  DecLayerDev layer[2];
  const float* qe[2];
  float *tgt1, *q0, *qkv1;
};
hipError_t launch_decoder_consts(const DecConstLaunch& p, hipStream_t s);
hipError_t launch_decoder(const DecLaunch& p, hipStream_t s);

struct HeadsDev {
  const f32x4* conv_w;     // 9 taps x fragment-packed [256 out][256 in]
  const f32x4* conv_w_l;   // split mode lo plane
  const float *conv_b, *gn_w, *gn_b, *out_w, *out_b;
  const float *tlbr0_t;    // [256 in][256 out] transposed
  const float *tlbr2_w, *tlbr2_b;  // [4][256], [4]
};
struct HeatLaunch {
  Geom g;
  HeadsDev w;
  const float* mem[2];     // per side memory [N][L][256]
  const float* hs[2];      // per side hs [N][256]
  float* conv_out;         // [rows][256]
  float* gn_part;          // [ntiles][32][2]  (mean, M2) per tile & group
  float* sm_part;          // [ntiles][4]  (max, sum e, sum e x, sum e y) softmax partials per tile
  float* logits;           // [rows]
  float* cxy[2];           // [N][2] per side
  int img_h[2];
  // fused tail (forward path): size regression + boxes in the same launch
  float* tlbr[2];          // [N][4] per side, or NULL
  float* box[2];           // [N][4] per side, or NULL
  int img_w[2];
  uint32_t* flags;         // the handle's status word (FLAG_F16_RANGE)
  uint32_t* publish;       // ABI 6 (oetr_forward*_flagslot): mapped host word k_heat_final moves the status word into, or NULL
  int force_staged_conv;   // direct form: k_heat_conv64 (per-tap staging) even where the halo-resident form fits
  int convp_units;         // (tile, tap) units per conv-P work item, 3 .. 9 (conv_p.h), set by launch_decoder_convp from DecLaunch
  const float* mask[2];    // forward_dummy's masks per side [N][L] or NULL: logits of tokens with mask == 0
                           // are filled with -1e9 before the softmax (reference src/model.py:166-171)
};
hipError_t launch_heat_conv(const HeatLaunch& p, int mode, hipStream_t s);
// 64 token rows per workgroup, two-plane mode only: the direct conv of the forward path for large batches
hipError_t launch_heat_conv64(const HeatLaunch& p, int mode, hipStream_t s);
// Forward path: decoder (2N workgroups) and the hs-independent part of the heat-map
// conv, P_tap = W_tap . memory (one workgroup per token tile), in ONE launch so that
// the 16-CU decoder runs beside the conv GEMMs; then the cheap combine
// conv_out[l] = b + sum_tap att[l+tap] * P_tap[l+tap].
hipError_t launch_decoder_convp(const DecLaunch& d, const HeatLaunch& h, float* P, int mode,
                                hipStream_t s);
hipError_t launch_heat_combine(const HeatLaunch& h, const float* P, hipStream_t s);
hipError_t launch_heat_final(const HeatLaunch& p, hipStream_t s);
hipError_t launch_size_regression(const HeadsDev& w, const float* hs1, const float* hs2,
                                  int n, float* tlbr1, float* tlbr2, hipStream_t s);
hipError_t launch_boxes(const float* cxy, const float* tlbr, int n, int max_h, int max_w,
                        float* box, hipStream_t s);
// ---- neck (input_proj -> PatchMerging -> input_proj2), neck.hip ----
constexpr int BBC = 1024;        // backbone (ResNet layer3) channels
constexpr int NECK_MT = 256;     // output positions per conv workgroup (largest shape; 192 / 128 too)
constexpr int NECK_PIX = 16;     // kernel pixels per conv workgroup (its K slice = 16 * 256)
constexpr int NECK_XROW = 2 * C;    // halves per pixel of the neck's X buffer: 8 chunks of [32 hi | 32 lo]
constexpr int NECK_RW_MIN_WO = 16;  // narrowest output map the row-window conv kernel takes
struct NeckGeom {
  int n_img, hb, wb, ho, wo;
  int HW;        // hb * wb
  int rows_in;   // n_img * HW  (row rows_in of the X planes is all zero: padding source)
  int M;         // n_img * ho * wo output positions
};
struct NeckProjLaunch {
  NeckGeom g;
  const float* bb;              // [n_img][1024][HW]
  const f32x4 *wh[2], *wl[2];   // input_proj weight, K halves, f16 fragment planes
  const float *bias, *ln_w, *ln_b;
  _Float16 *xh, *xl;            // [rows_in + 1][256] LayerNorm'ed projection, split planes
  uint32_t* flags;              // the neck handle's status word (FLAG_F16_RANGE)
};
struct NeckConvDesc {
  const f32x4 *wh, *wl;   // [split][nhalf][4 n-tiles][256 k16-steps][64 lanes] 16-byte units
  const f32x4 *wh_rw, *wl_rw;   // the same values in k_neck_conv_rw's step order
  float* part;            // [nsplit][M][ncols] partial sums
  int log2ks, pad, nsplit, nhalf, ncols;
  int item0;              // first work item of this conv within a tile's items
};
struct NeckConvLaunch {
  NeckGeom g;
  const _Float16 *xh, *xl;
  NeckConvDesc conv[3];
  int items_per_mt;       // K slices x column halves of all three convs (22)
  int mt_rows;            // output positions per workgroup: 256, 192 or 128 (neck_conv_rows)
  int row_window;         // 1: k_neck_conv_rw (wo >= NECK_RW_MIN_WO), 0: k_neck_conv
  int nblocks;            // items_per_mt * number of mt_rows-position tiles
};
struct NeckOutLaunch {
  NeckGeom g;
  const float* part[3];
  int nsplit[3];
  const float* bias[3];
  const f32x4 *wh, *wl;   // input_proj2 weight [256][512]
  const float* bias2;
  float* feat;            // [n_img][256][ho*wo]  (NCHW), or
  float* tokens;          // [n_img * ho*wo][256] (token-major: the hot path's x rows); one of the two
  uint32_t* flags;
};
hipError_t launch_neck_proj(const NeckProjLaunch& p, hipStream_t s);
hipError_t launch_neck_conv(const NeckConvLaunch& p, hipStream_t s);
int neck_conv_rows(int M, int items_per_mt, int num_cus, int max_rows);
hipError_t launch_neck_out(const NeckOutLaunch& p, hipStream_t s);

// calib.hip (oetr_debug_mfma_rate): dense f16 MFMA rate the chip sustains, TFLOP/s; synchronises `s`
hipError_t measure_mfma_rate(int num_cus, double seconds, double* tflops, hipStream_t s);

hipError_t launch_linear_attention(const float* q, const float* k, const float* v, const float* q_mask,
                                   const float* kv_mask, int n, int L, int S, float* out, float* state,
                                   hipStream_t s);
hipError_t launch_full_attention(const float* q, const float* k, const float* v, int n,
                                 int L, int S, float* out, hipStream_t s);
hipError_t launch_full_attention_split(const float* q, const float* k, const float* v, int n,
                                       int L, int S, float* out, uint32_t* flags, hipStream_t s);

}  // namespace oetr
