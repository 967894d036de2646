// Shared definitions for the OETR gfx950 kernels (device helpers + host-side
// launch declarations).  CDNA4 only: wave = 64 lanes, f32 MFMA 32x32x2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
enum { ABL_KVREDUCE = 1, ABL_GELU = 2, ABL_ELU = 4, ABL_LN = 8, ABL_GEMM = 16, ABL_STORE = 32 };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Row of accumulator register r for a 32x32 MFMA C/D tile (col = lane & 31).
__device__ __forceinline__ int crow(int r, int half) {
  return (r & 3) + 8 * (r >> 2) + 4 * half;
}

// phi(x) = elu(x) + 1.  torch evaluates expm1(x) + 1 for x <= 0, which is
// exp(x) to within one rounding of the final add (<= 6e-8 absolute); exp is
// used directly (branch-free, ~1/3 of the instructions of expm1).
__device__ __forceinline__ float elu1(float x) {
  return x > 0.f ? x + 1.0f : expf(x);
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
// Branch-free fp32 erf (coefficients and error analysis: tools/fit_erf.py;
// max 1.3 ulp / 7.7e-8 abs vs double-precision erf).  Both branches are
// evaluated and selected, so a wave never diverges; ocml's erff costs ~3x the
// issue slots in the GELU epilogue because of its divergent range split.
//   |x| <= 0.92 : x + x*P(x^2)
//   |x| >  0.92 : sign(x) * (1 - 2^(t*R(t))),  t = min(|x|, 4)
__device__ __forceinline__ float erf_f32(float x) {
  const float ax = fabsf(x);
  const float t = fminf(ax, 4.0f);
  const float s = x * x;
  float p = 8.404849575e-05f;
  p = fmaf(p, s, -8.151340561e-04f);
  p = fmaf(p, s, 5.201837672e-03f);
  p = fmaf(p, s, -2.685974483e-02f);
  p = fmaf(p, s, 1.128370225e-01f);
  p = fmaf(p, s, -3.761263422e-01f);
  p = fmaf(p, s, 1.283791667e-01f);
  const float small_v = fmaf(x, p, x);
  float r = 4.358980029e-07f;
  r = fmaf(r, t, -7.196586003e-06f);
  r = fmaf(r, t, 2.439359676e-05f);
  r = fmaf(r, t, 3.708157171e-04f);
  r = fmaf(r, t, -5.214545892e-03f);
  r = fmaf(r, t, 3.451753623e-02f);
  r = fmaf(r, t, -1.537478043e-01f);
  r = fmaf(r, t, -9.159608808e-01f);
  r = fmaf(r, t, -1.628399401e+00f);
  const float large_v = copysignf(1.0f - __builtin_amdgcn_exp2f(t * r), x);
  return ax <= 0.92f ? small_v : large_v;
}
// Exact-erf GELU (torch nn.GELU default): 0.5 x (1 + erf(x / sqrt 2)).
__device__ __forceinline__ float gelu_erf(float x) {
#ifdef OETR_OCML_ERF
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#else
  return 0.5f * x * (1.0f + erf_f32(x * 0.70710678118654752440f));
#endif
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
                                           const f32x4* const (&w_ptr)[NT], int chunk) {
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t) r.b[u][t] = w_ptr[t][(chunk * U + u) * 64];
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
  for (int t = 0; t < NT; ++t) w_ptr[t] = Wp + (size_t)(nt0 + t) * KS * 64 + lane;

  GemmRegs<NT, U> r0, r1;
  gemm_fetch<NT, U>(r0, a_ptr, w_ptr, 0);
  for (int c = 0; c < NCH; c += 2) {
    gemm_fetch<NT, U>(r1, a_ptr, w_ptr, c + 1);
    __builtin_amdgcn_sched_barrier(0);
    gemm_mma<NT, U>(r0, acc);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 2 < NCH) gemm_fetch<NT, U>(r0, a_ptr, w_ptr, c + 2);
    __builtin_amdgcn_sched_barrier(0);
    gemm_mma<NT, U>(r1, acc);
    __builtin_amdgcn_sched_barrier(0);
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

#endif  // __HIPCC__

// ------------------------------------------------------------------ host
// Repacked weights of one encoder layer (device pointers).
struct EncLayerDev {
  const f32x4 *wq, *wk, *wv, *wmerge, *w1, *w2;  // fragment-packed
  const float *lnq_w, *lnq_b, *lnkv_w, *lnkv_b, *ln2_w, *ln2_b;
};
// Decoder cross-attention K/V projections applied to the encoder memory.
struct DecKVDev {
  const f32x4 *wk[2], *wv[2];
  const float *bk[2], *bv[2];
};

struct EncLaunch {
  Geom g;
  float* x;              // [rows][256] token-major activations (in/out, in place)
  float* qp;             // [rows][256] phi(Q) of the layer being finished / next
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
  int b_cross;           // phase-B layer is a cross layer
  int dbg;               // ablation flags (OETR_ABLATE builds only)
};

// has_b: run phase B (finish a layer); tail: 0 = phase A of next encoder layer,
// 1 = decoder K/V preparation, 2 = nothing.
hipError_t launch_prep_tokens(const Geom& g, const float* feat1, const float* feat2,
                              const float* pos1, const float* pos2, float* x,
                              float* pos_tok, hipStream_t s);
hipError_t launch_encoder(const EncLaunch& p, bool has_b, int tail, hipStream_t s);

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
  float* logits;           // [rows]
  float* cxy[2];           // [N][2] per side
  int img_h[2];
  // fused tail (forward path): size regression + boxes in the same launch
  float* tlbr[2];          // [N][4] per side, or NULL
  float* box[2];           // [N][4] per side, or NULL
  int img_w[2];
};
hipError_t launch_heat_conv(const HeatLaunch& p, hipStream_t s);
hipError_t launch_heat_final(const HeatLaunch& p, hipStream_t s);
hipError_t launch_size_regression(const HeadsDev& w, const float* hs1, const float* hs2,
                                  int n, float* tlbr1, float* tlbr2, hipStream_t s);
hipError_t launch_boxes(const float* cxy, const float* tlbr, int n, int max_h, int max_w,
                        float* box, hipStream_t s);
hipError_t launch_linear_attention(const float* q, const float* k, const float* v, int n,
                                   int L, int S, float* out, hipStream_t s);
hipError_t launch_full_attention(const float* q, const float* k, const float* v, int n,
                                 int L, int S, float* out, hipStream_t s);

}  // namespace oetr
