// Fused encoder kernels of the OETR feature-correlation transformer (gfx950).
//
// Reference semantics: EncoderLayer.forward (src/models/transformer.py:104-142)
// with LinearAttention (src/models/linear_attention.py:22-50), layer order
// self,cross x4 as in QueryTransformer.forward (transformer.py:349-358).
//
// Linear attention needs one all-to-all per layer (the per-head 32x32 state
// KV = sum_s phi(K)^T V/S and sum_s phi(K) over every token of the source
// image).  That seam is where the launches are cut:
//
//   phase A(l): LN_q/LN_kv(+pos) -> Q,K,V projections -> phi(Q) to HBM,
//               per-tile partial KV / Ksum states to HBM
//   ------------------------- kernel boundary -------------------------
//   phase B(l): reduce the source image's partial states, phi(Q).KV.Z*S,
//               merge, residual, LN2, MLP(GELU), residual -> x
//
// and one launch runs  B(l) ; A(l+1)  on the same 32-token tile, so the
// activations stay in LDS/registers across the layer boundary.  9 launches
// cover the 8 layers; the last one (TAIL=1) computes the decoder's
// cross-attention K/V states from the finished memory instead of A(l+1).
//
// Work decomposition: one workgroup = 32 tokens x 256 channels, 4 waves (one
// per SIMD).  Wave w owns output columns [64w, 64w+64) of every GEMM = heads
// 2w, 2w+1, so a head's 32x32 MFMA tile never leaves its wave: phi(K)^T V is
// computed straight from the K and V accumulators, and the reduced KV state
// is consumed as an MFMA B operand in the register layout it was produced in.
#include <type_traits>

#include "common.h"

// The two small attention blocks of the f16-based modes - the per-tile KV state phi(K)^T V and
// the attention apply phi(Q).KV - take their MFMA operands straight from VALU conversions.
//  * APPLY runs as fp32-class split f16 MFMAs (common.h: mma16_split3, fenced).
//  * STATE runs as split f16 MFMAs too (kv_state_64, OETR_SPLIT_STATE = 1: 6 f16 MFMAs per row tile
//    instead of 16 f32 ones, 1.8 us per 64-row launch), in ONE code path.  History, because the argument
//    for shipping it is empirical: round 2's form of this state returned timing-dependent results (2 - 800
//    of 25 000 forwards) and was replaced by exact f32 MFMAs in round 3.  Round 4's hazard study
//    (profiles/r4_hazard_study.txt, HISTORY.md) found what every failing build shares - BOTH row-tile code
//    paths of that form in the kernel, a workgroup-uniform run-time branch between a path with the row masks
//    (ragged tiles) and one without (full tiles): builds with either path alone, 0 differing of 157 000
//    forwards under both amplifiers (vmcnt(0) before every GEMM step, s_setprio 3 around the state);
//    two-path builds 14 .. 8 002 of 37 000, fenced or not.  Every hazard distance in the failing instruction
//    stream was checked against hardware probes (tools/mfma_branch_hazard_probe.hip, tools/sgpr_war_probe.hip,
//    tools/mfma_hazard_probe.hip) and holds: THE MECHANISM IS NOT NAMED.  Round 5 rebuilt the state with the
//    masks applied unconditionally (one path) and soaked it with the amplifiers as compile options: 0
//    differing of 1 559 115 forwards (profiles/r5_determinism_soak.txt).  That is evidence, not an
//    explanation, so the tripwire runs where the driver runs it (round 6): `make` also builds
//    liboetr_hip_soak.so (-DOETR_SOAK_AMP=3, both amplifiers), and
//    tests/test_gpu_determinism.py::test_amplified_soak_build_is_deterministic runs the interleaved-shape
//    loop against it for 20 s in every GPU test run.  If that test ever reports a differing forward, set
//    OETR_SPLIT_STATE 0 (the f32-MFMA state is still in the source) before anything else.

#ifndef OETR_SOAK_AMP
#define OETR_SOAK_AMP 0   // determinism-soak builds (tools/r5_soak.sh): 1 = vmcnt(0) before every GEMM step, 2 = s_setprio 3 around the state
#endif
#ifndef OETR_SPLIT_STATE
#define OETR_SPLIT_STATE 1   // linear-attention state of the two-plane encoder kernels on f16 MFMAs (kv_state_64)
#endif

namespace oetr {

// ---------------------------------------------------------------------------
// Fused  B(l) ; A(l+1)  kernel.
// ---------------------------------------------------------------------------
// LDS regions (floats).  S1/S2/H hold a GEMM A operand: an f32 tile [32][260]
// (resp. [32][516]) in f32 mode, or two f16 planes (hi, lo) [32][264] (resp.
// [32][520]) in split mode - nearly the same bytes.
constexpr int S0_OFF = 0;                                   // f32 tile (LN input, phi(Q))
constexpr int S1_OFF = S0_OFF + TM * LDA;
constexpr int H_OFF = S1_OFF + TILE_FLOATS;  // hidden tile; S2 aliases its start
constexpr int KSUM_OFF = H_OFF + HID_FLOATS;
constexpr int Z_OFF = KSUM_OFF + C;
constexpr int LNP_OFF = Z_OFF + TM * NH;  // LayerNorm affines: ln2 w,b | lnq w,b | lnkv w,b
constexpr int SMEM_FLOATS = LNP_OFF + 6 * C;

// Workgroup shape: NW waves.  NW = 4 (one wave per SIMD) for the exact-f32
// mode, where the MFMA pipe is the bound; NW = 8 (two per SIMD) for the split
// mode, where it is not: twice the weight bytes in flight per CU and one wave's
// VALU / LDS phases run under the other's MFMAs.  Wave w owns NT = 8/NW
// 32-column tiles (= heads) of every 256-wide GEMM.
template <int NW>
struct EncCfg {
  static constexpr int NT = 8 / NW;
  static constexpr int THREADS = 64 * NW;
  static constexpr int WC = 32 * NT;        // output columns per wave
  static constexpr int TPR = THREADS / TM;  // threads per row in row-wise phases (8 / 16)
  static constexpr int F4 = 64 / TPR;       // float4 per thread per row
};

// phi(K)^T (V/S) (32x32, one head) and the per-lane partial of sum phi(K) for one 32-row
// tile, straight from the K and V accumulators.  phi branch-free (max(x,0) + exp(min(x,0))
// == elu(x)+1 bit for bit) on scalars, four at a time between sched_barriers: as a select
// hipcc branches per element (with whole-tuple copies when done in place), unfenced it
// schedules all exps at once and spills.
// MASKED: see kv_state_64 - msk_s[row] = the token's kv_mask value, 0 past the last valid row.
template <int MODE, bool MASKED = false>
__device__ __forceinline__ void kv_state_32(const f32x16& accK, const f32x16& accV, bool skip_phi,
                                            int S_len, int nvalid, int half, f32x16& kv,
                                            float& ksum, Range& rg, const float* msk_s = nullptr) {
  const float inv_len = 1.0f / (float)S_len;
  kv = f32x16{0};
  ksum = 0.f;
  {
    // One f32 MFMA (64 matrix-pipe cycles) per accumulator register: the operands of register
    // r + 1 - phi, row mask, 1/S - are computed while the MFMA of register r runs (a fence per
    // step keeps hipcc from batching the exps; same operations in the same order: same bits).
    auto operands = [&](int r, float& k, float& v) {
      float m;
      if constexpr (MASKED) m = msk_s[crow(r, half)];
      else m = crow(r, half) < nvalid ? 1.0f : 0.0f;
      const float x = accK[r];
      k = (skip_phi ? x : elu1(x)) * m;
      v = accV[r] * (inv_len * m);
      ksum += k;
    };
    float kc, vc;
    operands(0, kc, vc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float kn = 0.f, vn = 0.f;
      if (r + 1 < 16) operands(r + 1, kn, vn);
      kv = __builtin_amdgcn_mfma_f32_32x32x2f32(kc, vc, kv, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      kc = kn; vc = vn;
    }
  }
}

// phi(K)^T (V/S) for this wave's heads straight from the K and V accumulators,
// plus sum_s phi(K); stores the per-tile partial states.
template <int MODE, int NT, bool MASKED = false>
__device__ __forceinline__ void kv_state_store(f32x16 (&accK)[NT], f32x16 (&accV)[NT],
                                               bool skip_phi, int S_len, int nvalid, int lane,
                                               int wave, float* __restrict__ kv_out,
                                               float* __restrict__ ks_out, int slot, Range& rg,
                                               const float* msk_s = nullptr) {
  const int half = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float ksum;
    f32x16 kv;
    kv_state_32<MODE, MASKED>(accK[t], accV[t], skip_phi, S_len, nvalid, half, kv, ksum, rg, msk_s);
    const int h = NT * wave + t;
    f32x4* dst = reinterpret_cast<f32x4*>(kv_out) + ((size_t)slot * NH + h) * 4 * 64 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = {kv[4 * q], kv[4 * q + 1], kv[4 * q + 2], kv[4 * q + 3]};
      dst[q * 64] = o;
    }
    ksum += __shfl_xor(ksum, 32, 64);
    if (half == 0) ks_out[(size_t)slot * C + h * HD + lane] = ksum;
  }
}

// Coalesced [TM][256] tile load HBM -> LDS.  Rows past the image end re-read
// the last valid row: a guarded load would cost a branch plus a full vmcnt(0)
// round trip per element, and those rows are never stored anyway.
template <int THREADS>
__device__ __forceinline__ void load_tile(float* S, const float* __restrict__ src, int nvalid,
                                          int tid) {
#pragma unroll
  for (int i = 0; i < (TM * C / 4) / THREADS; ++i) {
    const int idx = tid + THREADS * i;
    const int r = idx >> 6, c4 = idx & 63;
    *reinterpret_cast<f32x4*>(S + r * LDA + 4 * c4) =
        reinterpret_cast<const f32x4*>(src + (size_t)min(r, nvalid - 1) * C)[c4];
  }
}

// The same tile from the reference's layout: src -> element (channel 0, token l0) of an
// image's [256][L] map; S[r][c] = src[c * L + r].  A half-wave reads 32 consecutive tokens
// of one channel (128 B); the transposed LDS writes are 2-way conflicted (row stride 260).
template <int THREADS, int ROWS>
__device__ __forceinline__ void load_tile_nchw(float* S, const float* __restrict__ src, int L,
                                               int nvalid, int tid) {
  constexpr int CG = THREADS / ROWS;
  const int r = tid % ROWS, cg = tid / ROWS;
  const float* s = src + min(r, nvalid - 1);
#pragma unroll
  for (int i = 0; i < C / CG; ++i) {
    const int c = cg + CG * i;
    S[r * LDA + c] = s[(size_t)c * L];
  }
}
// ... and back out token-major (rows of 1 KB): dst -> row l0 of a [L][256] table
template <int THREADS, int ROWS>
__device__ __forceinline__ void store_tile_tokens(float* __restrict__ dst, const float* S, int nvalid,
                                                  int tid) {
#pragma unroll
  for (int i = 0; i < (ROWS * C / 4) / THREADS; ++i) {
    const int idx = tid + THREADS * i;
    const int r = idx >> 6, c4 = idx & 63;
    if (r < nvalid)
      reinterpret_cast<f32x4*>(dst + (size_t)r * C)[c4] = *reinterpret_cast<const f32x4*>(S + r * LDA + 4 * c4);
  }
}

// ---------------------------------------------------------------------------
// attention == 'full' (reference FullAttention, src/models/linear_attention.py:53-87,
// selected by EncoderLayer(attention='full'), transformer.py:86-89): softmax(Q K^T /
// sqrt(D)) V over ALL tokens of the source image, for this tile's 32 queries and the
// wave's head.  Flash style with the fp32-class f16 split (attention.hip has the
// stand-alone version): S^T = K Q^T with Q as the B operand held in registers, the
// online softmax lane-local (a lane owns one query column), O^T += V^T P^T with P taken
// straight from the S^T accumulator registers.  K rows and V^T rows come from the
// buffers phase A wrote; nothing L x S ever exists in memory.  Result -> message tile S1.
template <int MODE>
__device__ __forceinline__ void full_attention_tile(const EncLaunch& p, int n, int ss, size_t row_base,
                                                    int nvalid, int lane, int head,
                                                    const ATile<MODE>& S1, Range& rg) {
  const Geom& g = p.g;
  const int half = lane >> 5, col = lane & 31;
  const int S = g.L[ss];
  const float* kb = p.kbuf_in + ((size_t)g.row0[ss] + (size_t)n * S) * C + head * HD + 8 * half;
  const float* vr = p.vt_in + p.vt_off[ss] + ((size_t)n * C + head * HD + col) * p.lpad[ss] + 4 * half;
  constexpr float NEG = -1.0e30f;   // finite "minus infinity": exp_neg(NEG - m) == 0, no NaN
  const float temp = 0.17677669529663687f;   // 1 / sqrt(32)
  if constexpr (MODE == GM_F32) {
    // Exact-fp32 products (v_mfma_f32_32x32x2_f32; the build an f16-based handle falls back to
    // when an operand leaves the f16 range).  Same register geometry as below: S^T rows = keys
    // (crow(r, half)), columns = queries; the k-slot of step j is channel 16 half + j for
    // S^T = K Q^T and key crow(j, half) for O^T = V^T P - so P feeds the second product from
    // the accumulator registers as it stands.
    f32x4 qf[4];
    {
      const float* qp = p.qp + (row_base + min(col, nvalid - 1)) * C + head * HD + 16 * half;
#pragma unroll
      for (int i = 0; i < 4; ++i) qf[i] = *reinterpret_cast<const f32x4*>(qp + 4 * i);
    }
    const float* kb32 = p.kbuf_in + ((size_t)g.row0[ss] + (size_t)n * S) * C + head * HD + 16 * half;
    auto load32 = [&](int k0, f32x4 (&kk)[4], f32x4 (&vv)[4]) {
      const float* kr = kb32 + (size_t)min(k0 + col, S - 1) * C;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kk[i] = *reinterpret_cast<const f32x4*>(kr + 4 * i);
        vv[i] = *reinterpret_cast<const f32x4*>(vr + k0 + 8 * i);   // keys k0 + 8 i + 4 half + 0..3 = crow(4 i + .., half)
      }
    };
    f32x16 o = {0};
    float m_run = NEG, l_run = 0.f;
    f32x4 kk[4], vv[4];
    load32(0, kk, vv);
    for (int k0 = 0; k0 < S; k0 += 32) {
      f32x4 kc[4], vc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { kc[i] = kk[i]; vc[i] = vv[i]; }
      if (k0 + 32 < S) load32(k0 + 32, kk, vv);
      f32x16 st = {0};
#pragma unroll
      for (int j = 0; j < 16; ++j) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[j >> 2][j & 3], qf[j >> 2][j & 3], st, 0, 0, 0);
      float mt = NEG;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        st[r] = (k0 + crow(r, half) < S) ? st[r] * temp : NEG;
        mt = fmaxf(mt, st[r]);
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      const float m_new = fmaxf(m_run, mt);
      const float alpha = exp_neg(fminf(m_run - m_new, 0.f));
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = exp_neg(fminf(st[r] - m_new, 0.f)); ps += st[r]; }
      ps += __shfl_xor(ps, 32, 64);
      l_run = l_run * alpha + ps;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
      for (int j = 0; j < 16; ++j) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[j >> 2][j & 3], st[j], o, 0, 0, 0);
    }
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      S1.put4(col, head * HD + 8 * g4 + 4 * half,
              f32x4{o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv});
    return;
  } else {
  // Round 6: the formulation of attention.hip's k_full_attention_split, both passes - 1/sqrt(D) log2(e)
  // folded into Q before its split (scores in log2 units: v_exp_f32 directly); K's, Q's and P's lo planes UNSCALED
  // (split2u: the MFMA honours f16 denormals), so the three products of a score go into ONE accumulator - chain from
  // zero, small products first (the MFMA's aligned sum cuts small addends beside large ones), -m as an MFMA product
  // (A = ones in two k-slots, B = -m as an f16 pair, m kept to 21 bits) between the two big steps; m raised LAZILY
  // (only when some lane's tile maximum exceeds it by more than 2^FA_LAZY: P' = 2^FA_SH P <= 2^12, inside the f16 range
  // of its hi plane; the final 1/l removes the common factor); the P.V accumulator of V's lo plane folded once at the
  // end; keys past S masked in a compile-time variant of the LAST tile only; l a per-lane partial.
  // ~125 VALU instructions per 32-key tile (first pass: ~190; round 2: ~330).
  float neg1 = -1.0f;               // (opaque to the compiler: see split2u)
  asm volatile("" : "+s"(neg1));
  auto split8u = [&](const f32x4& a0, const f32x4& a1, f32x4& hi, f32x4& lo, Range& r) {
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    split2u(a0[0], a0[1], neg1, h0, l0, r);
    split2u(a0[2], a0[3], neg1, h1, l1, r);
    split2u(a1[0], a1[1], neg1, h2, l2, r);
    split2u(a1[2], a1[3], neg1, h3, l3, r);
    hi = __builtin_bit_cast(f32x4, u32x4{h0, h1, h2, h3});
    lo = __builtin_bit_cast(f32x4, u32x4{l0, l1, l2, l3});
  };
  f32x4 qh[2], ql[2];
  {
    const float qs = temp * 1.4426950408889634f;
    const float* qp = p.qp + (row_base + min(col, nvalid - 1)) * C + head * HD + 8 * half;
#pragma unroll
    for (int s = 0; s < 2; ++s)
      split8u(*reinterpret_cast<const f32x4*>(qp + 16 * s) * qs, *reinterpret_cast<const f32x4*>(qp + 16 * s + 4) * qs,
              qh[s], ql[s], rg);
  }
  auto load = [&](int k0, f32x4 (&kk)[4], f32x4 (&vv)[4]) {
    const float* kr = kb + (size_t)min(k0 + col, S - 1) * C;   // rows past S: clamped, masked below
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      kk[2 * s] = *reinterpret_cast<const f32x4*>(kr + 16 * s);
      kk[2 * s + 1] = *reinterpret_cast<const f32x4*>(kr + 16 * s + 4);
      // k-slot 8*half + i of step s <-> key k0 + 16s + 8(i>>2) + 4*half + (i&3)
      vv[2 * s] = *reinterpret_cast<const f32x4*>(vr + k0 + 16 * s);
      vv[2 * s + 1] = *reinterpret_cast<const f32x4*>(vr + k0 + 16 * s + 8);
    }
  };
  constexpr float FA_LAZY = 8.0f;
  f32x16 o = {0}, oc = {0};   // O^T and the part that meets V's (2^11-scaled) lo plane: rows = d (crow(r, half)), col = query
  float m_run = -FA_SH, l_run = 0.f;   // m_run = (reference maximum) - FA_SH; -m_run as an f16 pair in k-slots 0, 1:
  const f32x4 afrag = __builtin_bit_cast(f32x4, u32x4{lane < 32 ? 0x3c003c00u : 0u, 0u, 0u, 0u});
  f32x4 mfrag = __builtin_bit_cast(f32x4, u32x4{lane < 32 ? 0x00004400u : 0u, 0u, 0u, 0u});   // (4.0h, 0)
  static_assert(FA_SH == 4.0f, "initial mfrag encodes 4.0 as f16 0x4400");
  f32x4 kk[4], vv[4];
  load(0, kk, vv);
  auto tile = [&](int k0, auto tail_c) {
    constexpr bool tail = decltype(tail_c)::value;
    f32x4 kh[2], kl[2], vh[2], vl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      split8u(kk[2 * s], kk[2 * s + 1], kh[s], kl[s], rg);
      split8(vv[2 * s], vv[2 * s + 1], vh[s], vl[s], rg);
    }
    if (!tail && k0 + 32 < S) load(k0 + 32, kk, vv);   // next tile's rows under this tile's math
    f32x16 st = {0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      st = mma16<GM_SPLIT>(kh[s], ql[s], st);
      st = mma16<GM_SPLIT>(kl[s], qh[s], st);
    }
    st = mma16<GM_SPLIT>(kh[0], qh[0], st);
    st = mma16<GM_SPLIT>(afrag, mfrag, st);     // - m_run
    st = mma16<GM_SPLIT>(kh[1], qh[1], st);
    if constexpr (tail) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (k0 + crow(r, half) >= S) st[r] = NEG;
    }
    float mt = NEG;
#pragma unroll
    for (int r = 0; r < 16; r += 2) mt = __builtin_fmaxf(__builtin_fmaxf(mt, st[r]), st[r + 1]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const bool first = k0 == 0;
    if (first || __builtin_amdgcn_ballot_w64(mt > FA_SH + FA_LAZY) != 0) {
      // the new reference, rounded to 21 significant bits and a multiple of 2^-24: exactly the sum of two f16 values
      const float nm = -(m_run + (first ? mt - FA_SH : fmaxf(mt - FA_SH, 0.f)));
      float nq = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, nm) + 4u) & ~7u);
      if (__builtin_fabsf(nm) < 0.0625f) nq = __builtin_rintf(nm * 16777216.0f) * (1.0f / 16777216.0f);
      const float adj = -nq - m_run;
      if (!first) {
        const float alpha = __builtin_amdgcn_exp2f(-adj);
        l_run *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[r] *= alpha; oc[r] *= alpha; }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] -= adj;
      m_run = -nq;
      const _Float16 mh = (_Float16)nq, ml = (_Float16)(nq - (float)mh);
      const uint32_t mb = (uint32_t)__builtin_bit_cast(unsigned short, mh) | ((uint32_t)__builtin_bit_cast(unsigned short, ml) << 16);
      mfrag[0] = __builtin_bit_cast(float, lane < 32 ? mb : 0u);
      rg.see2(nq, 0.f);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = __builtin_amdgcn_exp2f(st[r]);
      l_run += st[r];
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 ph, pl;
      Range none;   // (P' <= 2^(FA_SH + FA_LAZY): nothing to guard)
      split8u(f32x4{st[8 * s], st[8 * s + 1], st[8 * s + 2], st[8 * s + 3]},
              f32x4{st[8 * s + 4], st[8 * s + 5], st[8 * s + 6], st[8 * s + 7]}, ph, pl, none);
      o = mma16<GM_SPLIT>(vh[s], ph, o);
      oc = mma16<GM_SPLIT>(vl[s], ph, oc);
      o = mma16<GM_SPLIT>(vh[s], pl, o);
    }
  };
  {
    int k0 = 0;
    for (; k0 + 32 <= S; k0 += 32) tile(k0, std::false_type{});
    if (k0 < S) tile(k0, std::true_type{});     // the ragged last tile: the only one with keys past S
  }
  const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {   // registers 4*g4.. = d rows 8*g4 + 4*half + 0..3 of query `col`
    f32x4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = fmaf(oc[4 * g4 + i], SPLIT_INV, o[4 * g4 + i]) * inv;
    S1.put4(col, head * HD + 8 * g4 + 4 * half, y);
  }
  }   // f16-based modes
}

// LayerNorm affines -> LDS, six rows of 256 [ln2 w, b | lnq w, b | lnkv w, b].  Every source
// pointer is a compile-time choice per 256-thread group: with a run-time row index hipcc built
// a pointer table in scratch and fetched through it with dependent flat loads - three
// serialized round trips (scratch -> flat -> LDS) at the top of every launch.
template <int THREADS, bool HAS_B, int TAIL>
__device__ __forceinline__ void stage_ln_params(float* lnp_s, const EncLaunch& p, int tid) {
  static_assert(THREADS == 256 || THREADS == 512, "one or two 256-thread groups");
  const int c = tid & (C - 1);
  if constexpr (THREADS == 512) {
    const bool hi = __builtin_amdgcn_readfirstlane(tid >> 8) != 0;   // waves 4..7: the bias rows
    if (HAS_B) lnp_s[tid] = (hi ? p.b.ln2_b : p.b.ln2_w)[c];
    if (TAIL == 0) {
      lnp_s[2 * C + tid] = (hi ? p.a.lnq_b : p.a.lnq_w)[c];
      lnp_s[4 * C + tid] = (hi ? p.a.lnkv_b : p.a.lnkv_w)[c];
    }
  } else {
    if (HAS_B) { lnp_s[c] = p.b.ln2_w[c]; lnp_s[C + c] = p.b.ln2_b[c]; }
    if (TAIL == 0) {
      lnp_s[2 * C + c] = p.a.lnq_w[c]; lnp_s[3 * C + c] = p.a.lnq_b[c];
      lnp_s[4 * C + c] = p.a.lnkv_w[c]; lnp_s[5 * C + c] = p.a.lnkv_b[c];
    }
  }
}

// MASKED (forward_dummy's masks, see encoder64_body): instantiated for the exact-fp32 build - the
// route a masked batch is re-run on when the default arithmetic reports an operand out of range.
template <bool HAS_B, int TAIL, int MODE, int NW, bool FULL = false, int POL = 0, bool MASKED = false>
__global__ __launch_bounds__(64 * NW) void k_encoder(EncLaunch p) {
  static_assert(!MASKED || !FULL, "masks: linear attention only");
  constexpr bool SPLIT = gm_half(MODE);  // GEMM operands live in 16-bit LDS planes
  using SP = SitePolicy<POL>;            // arithmetic per GEMM site (two-plane mode, NW = 8)
  static_assert(POL == 0 || (gm_planes(MODE) == 2 && NW == 8 && !FULL), "policies: split mode, one n-tile per wave");
  static_assert(!FULL || ((gm_f16_range(MODE) || MODE == GM_F32) && NW == 8),
                "full attention: f16-based modes or exact fp32, one head per wave");
  using Cfg = EncCfg<NW>;
  constexpr int NT = Cfg::NT, THREADS = Cfg::THREADS, WC = Cfg::WC, TPR = Cfg::TPR, F4 = Cfg::F4;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  float* S0 = smem + S0_OFF;
  Range rg;  // f16 range guard of this thread's operand conversions (common.h)
  const ATile<MODE> S1(smem + S1_OFF, LDA, LDAH, &rg);
  const ATile<MODE> S2(smem + H_OFF, LDA, LDAH, &rg);
  const ATile<MODE> Hh(smem + H_OFF, LDH, LDHH, &rg);
  float* ksum_s = smem + KSUM_OFF;
  float* z_s = smem + Z_OFF;
  float* lnp_s = smem + LNP_OFF;
  using WS = WStream<MODE, NT>;
  WS ws;  // this wave's weight stream (runs ahead across the GEMMs below)
#ifdef OETR_ABLATE
  if constexpr (SPLIT && NT == 1) ws.dbg = p.dbg;
#endif
  // ring slot of each GEMM's first chunk
  constexpr int P_MERGE = 0;
  constexpr int P_W1A = WS::adv(P_MERGE, C);
  constexpr int P_W1B = WS::adv(P_W1A, C);
  constexpr int P_W2 = WS::adv(P_W1B, C);
  constexpr int P_T0 = HAS_B ? WS::adv(P_W2, FF) : 0;  // first GEMM of the tail
  constexpr int P_T1 = WS::adv(P_T0, C);
  constexpr int P_T2 = WS::adv(P_T1, C);
  constexpr int P_T3 = WS::adv(P_T2, C);

  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar)
  const int half = lane >> 5, col = lane & 31;
  const int wcol = WC * wave;               // first of this wave's output columns
  ws.set_lane(lane);
  const int lrow = tid / TPR, lpart = tid % TPR;  // (row, float4 slot) in row-wise phases

  // ---- tile identity (pair-major logical order: n, side, tile) ----
  const int logical = xcd_remap(blockIdx.x, g.ntiles);
  const int per = g.nt[0] + g.nt[1];
  const int n = logical / per;
  const int rem = logical - n * per;
  const int side = rem >= g.nt[0];
  const int t_idx = side ? rem - g.nt[0] : rem;
  const int L = g.L[side];
  const int l0 = t_idx * TM;
  const int nvalid = min(TM, L - l0);
  const size_t row_base = (size_t)g.row0[side] + (size_t)n * L + l0;
  const int slot = g.tile0[side] + n * g.nt[side] + t_idx;
  const bool nchw = !HAS_B && p.feat_nchw[0] != nullptr;   // first launch on NCHW inputs (launch-uniform)

  // LayerNorm affines -> LDS once (the row-wise phases then issue no global loads
  // that would queue behind the weight stream's run-ahead fetches)
  stage_ln_params<THREADS, HAS_B, TAIL>(lnp_s, p, tid);

  f32x16 xacc[NT];  // residual stream of this wave's columns (C layout)
  PHASE_STAMP(p, 0);

  if (HAS_B) {
    // ================= phase B: finish layer l =================
    const int ss = p.b_cross ? 1 - side : side;
    const int S_len = g.L[ss];
    // (kv_reduced: one pre-reduced state per image - k_kv_reduce - instead of the tiles' partials)
    const int nts = (p.kv_reduced || ABL(p.dbg, ABL_KVREDUCE)) ? 1 : g.nt[ss];
    const int src_slot0 = p.kv_reduced ? ss * g.N + n : g.tile0[ss] + n * g.nt[ss];

    if (!FULL && !ABL(p.dbg, ABL_XLOAD)) load_tile<THREADS>(S0, p.qp + row_base * C, nvalid, tid);  // phi(Q) tile
    // residual x in accumulator layout
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(crow(r, half), nvalid - 1);
        // (uniform base + 32-bit per-lane offset: an SGPR-base load, no 64-bit VALU address)
        xacc[t][r] = ABL(p.dbg, ABL_XLOAD) ? 0.5f : (p.x + row_base * C + wcol + 32 * t)[(unsigned)(row * C + col)];
      }
    if constexpr (FULL) {
      full_attention_tile<MODE>(p, n, ss, row_base, nvalid, lane, wave, S1, rg);
      ws.template prime<C, P_MERGE>(p.b.wmerge, p.b.wmerge_l, NT * wave, lane);
      PHASE_STAMP(p, 1);
    } else {
    // reduce the source image's partial KV states (fixed order -> deterministic);
    // the result is already in B-operand register order.  Several tiles are in
    // flight per round trip (32 float4 loads per lane).
    f32x4 kvB[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) kvB[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int KVR = NT == 1 ? 7 : 4;  // tiles per round trip (28 / 32 float4 per lane in flight)
      const f32x4* kvp = reinterpret_cast<const f32x4*>(p.kv_in) +
                         ((size_t)src_slot0 * NH + NT * wave) * 256 + lane;
      const float* ksp = p.ks_in + (size_t)src_slot0 * C + (tid & (C - 1));
      float ks = 0.f;
      if (nts == 1) {   // pre-reduced (or a single tile): one state, no redundant clamped loads
#pragma unroll
        for (int e = 0; e < 4 * NT; ++e) kvB[e >> 2][e & 3] = kvp[e * 64];
        ks = ksp[0];
      } else
      for (int ti0 = 0; ti0 < nts; ti0 += KVR) {
        f32x4 tmp[KVR][4 * NT];
        float kt[KVR];
#pragma unroll
        for (int u = 0; u < KVR; ++u) {
          const int ti = min(ti0 + u, nts - 1);
#pragma unroll
          for (int e = 0; e < 4 * NT; ++e) tmp[u][e] = kvp[(size_t)ti * (NH * 256) + e * 64];
          kt[u] = ksp[(size_t)ti * C];
        }
#pragma unroll
        for (int u = 0; u < KVR; ++u)
          if (ti0 + u < nts) {
#pragma unroll
            for (int e = 0; e < 4 * NT; ++e) kvB[e >> 2][e & 3] += tmp[u][e];
            ks += kt[u];
          }
      }
      if (THREADS == C || tid < C) ksum_s[tid] = ks;
    }
    ws.template prime<C, P_MERGE, SP::MERGE>(p.b.wmerge, p.b.wmerge_l, NT * wave, lane);
    __syncthreads();
    PHASE_STAMP(p, 1);

    // Z[row][h] = 1 / (phi(Q)[row,h,:] . Ksum[h,:] + eps)
    if (!ABL(p.dbg, ABL_ATTN) && (THREADS == TM * NH || tid < TM * NH)) {
      const int r = tid >> 3, h = tid & 7;
      const f32x4* qrow = reinterpret_cast<const f32x4*>(S0 + r * LDA + h * HD);
      const f32x4* kk = reinterpret_cast<const f32x4*>(ksum_s + h * HD);
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 a = qrow[i], b = kk[i];
        dot += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
      }
      z_s[r * NH + h] = 1.0f / (dot + ATTN_EPS);
    }
    __syncthreads();

    // message = (phi(Q) . KV) * Z * S  for this wave's heads -> S1
    if (!ABL(p.dbg, ABL_ATTN)) {
      float zr[NT][16];  // read before any S1 store: LDS stores would otherwise
#pragma unroll           // serialise these reads one by one (may-alias)
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) zr[t][r] = z_s[crow(r, half) * NH + NT * wave + t];
      f32x16 macc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) macc[t] = f32x16{0};
      if constexpr (gm_f16_range(MODE)) {
        // phi(Q).KV as 2 k16 steps of the fp32-class split per head (same reads of the
        // phi(Q) tile; k-slot 8*half + i of step s <-> d = 16s + 8*(i>>2) + 4*half + (i&3)
        // in BOTH operands)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f32x16 c1 = {0};
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const float* qrow = S0 + col * LDA + (NT * wave + t) * HD + 4 * half + 16 * s2;
            f32x4 ah, al, bh, bl;
            split8(*reinterpret_cast<const f32x4*>(qrow), *reinterpret_cast<const f32x4*>(qrow + 8),
                   ah, al, rg);
            split8(kvB[t][2 * s2], kvB[t][2 * s2 + 1], bh, bl, rg);
            mma16_split3(ah, al, bh, bl, macc[t], c1);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) macc[t][r] = fmaf(c1[r], SPLIT_INV, macc[t][r]);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(S0 + col * LDA + (NT * wave + t) * HD +
                                                            4 * half + ks * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              macc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], kvB[t][ks][j], macc[t], 0, 0, 0);
          }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) macc[t][r] = macc[t][r] * zr[t][r] * (float)S_len;
      S1.template put_acc<NT>(wcol, lane, macc);
    }
    }  // !FULL
    __syncthreads();
    PHASE_STAMP(p, 2);

    // x1 = x + message . Wmerge^T
    ws.template gemm<C, P_MERGE, C, SP::MERGE, SP::MLP1>(S1, p.b.wmerge, p.b.wmerge_l, NT * wave, lane, xacc,
                                                         p.b.w1, p.b.w1_l, 2 * NT * wave, p.dbg);
    acc_to_lds<NT>(S0, LDA, wcol, lane, xacc);
    __syncthreads();
    PHASE_STAMP(p, 3);

    // LN2(x1) -> S1
    {
      f32x4 xn[F4];
      ln_rows<TPR, F4>(S0, tid, xn, p.dbg);
      const f32x4* gw = reinterpret_cast<const f32x4*>(lnp_s) + lpart;
      const f32x4* gb = reinterpret_cast<const f32x4*>(lnp_s + C) + lpart;
#pragma unroll
      for (int i = 0; i < F4; ++i)
        S1.put4(lrow, 4 * (i * TPR + lpart), xn[i] * gw[i * TPR] + gb[i * TPR]);
    }
    __syncthreads();
    PHASE_STAMP(p, 4);

    // hidden = gelu(LN2(x1) . W1^T) -> Hh   (wave w: hidden columns [2*WC*w, 2*WC*(w+1)))
    {
      f32x16 hacc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) hacc[t] = f32x16{0};
      ws.template gemm<C, P_W1A, C, SP::MLP1, SP::MLP1>(S1, p.b.w1, p.b.w1_l, 2 * NT * wave, lane, hacc, p.b.w1,
                                                        p.b.w1_l, 2 * NT * wave + NT, p.dbg);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          if (ABL(p.dbg, ABL_GELU)) continue;
          const f32x2 ge = gelu_erf2(f32x2{hacc[t][r], hacc[t][r + 1]});
          hacc[t][r] = ge[0];
          hacc[t][r + 1] = ge[1];
        }
      Hh.template put_acc<NT>(2 * wcol, lane, hacc);
#pragma unroll
      for (int t = 0; t < NT; ++t) hacc[t] = f32x16{0};
      ws.template gemm<C, P_W1B, FF, SP::MLP1, SP::MLP2>(S1, p.b.w1, p.b.w1_l, 2 * NT * wave + NT, lane, hacc,
                                                         p.b.w2, p.b.w2_l, NT * wave, p.dbg);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          if (ABL(p.dbg, ABL_GELU)) continue;
          const f32x2 ge = gelu_erf2(f32x2{hacc[t][r], hacc[t][r + 1]});
          hacc[t][r] = ge[0];
          hacc[t][r + 1] = ge[1];
        }
      Hh.template put_acc<NT>(2 * wcol + WC, lane, hacc);
    }
    __syncthreads();
    PHASE_STAMP(p, 5);

    // x2 = x1 + hidden . W2^T ; write back
    ws.template gemm<FF, P_W2, (TAIL == 2 ? 0 : C), SP::MLP2, (TAIL == 0 ? SP::Q : SP::DEC_K)>(
        Hh, p.b.w2, p.b.w2_l, NT * wave, lane, xacc, TAIL == 0 ? p.a.wq : p.d.wk[0],
        TAIL == 0 ? p.a.wq_l : p.d.wk_l[0], NT * wave, p.dbg);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = crow(r, half);
        if (row < nvalid && !ABL(p.dbg, ABL_STORE)) (p.x + row_base * C + wcol + 32 * t)[(unsigned)(row * C + col)] = xacc[t][r];
      }
    if (TAIL != 2) acc_to_lds<NT>(S0, LDA, wcol, lane, xacc);
    __syncthreads();  // also: every wave is done reading Hh before S2 (alias) is written
    PHASE_STAMP(p, 6);
  } else {
    if (nchw) {  // first launch, reference layout: transpose on the way in (S1 is free until phase A writes it)
      load_tile_nchw<THREADS, TM>(S0, p.feat_nchw[side] + (size_t)n * C * L + l0, L, nvalid, tid);
      load_tile_nchw<THREADS, TM>(smem + S1_OFF, p.pos_nchw[side] + l0, L, nvalid, tid);
    } else {
      load_tile<THREADS>(S0, p.x + row_base * C, nvalid, tid);  // first launch: x from HBM
    }
    if (TAIL == 0) ws.template prime<C, P_T0, SP::Q>(p.a.wq, p.a.wq_l, NT * wave, lane);
    if (TAIL == 1) ws.template prime<C, P_T0, SP::DEC_K>(p.d.wk[0], p.d.wk_l[0], NT * wave, lane);
    __syncthreads();
    if (nchw) {  // token-major copies for the later launches (residual reads, position rows)
      store_tile_tokens<THREADS, TM>(p.x + row_base * C, S0, nvalid, tid);
      if (n == 0) store_tile_tokens<THREADS, TM>(p.pos_out + (size_t)(g.prow0[side] + l0) * C, smem + S1_OFF, nvalid, tid);
    }
  }

  const f32x4* pos = reinterpret_cast<const f32x4*>(
                         p.pos + (size_t)(g.prow0[side] + l0 + min(lrow, nvalid - 1)) * C) + lpart;
  // (MASKED: the tile's mask values in the normaliser buffer, which phase B is done with; a
  //  barrier lies between this store and every read)
  float* msk_s = z_s;
  if constexpr (MASKED && TAIL != 2) {
    if (tid < TM) msk_s[tid] = tid < nvalid ? p.mask[side][(size_t)n * L + l0 + tid] : 0.f;
  }
  if (TAIL == 0) {
    // ================= phase A: start layer l+1 =================
    // q_in = LN_q(x)+pos -> S1 ; kv_in = LN_kv(x)+pos -> S2 (one set of row stats)
    {
      f32x4 xn[F4], ps[F4];
      if (!HAS_B && nchw) {   // the position tile sits (f32, transposed on load) where S1 goes
        const f32x4* pl = reinterpret_cast<const f32x4*>(smem + S1_OFF + lrow * LDA) + lpart;
#pragma unroll
        for (int i = 0; i < F4; ++i) ps[i] = pl[i * TPR];
      } else {
#pragma unroll
        for (int i = 0; i < F4; ++i) ps[i] = pos[i * TPR];
      }
      ln_rows<TPR, F4>(S0, tid, xn, p.dbg);
      if (!HAS_B && nchw) __syncthreads();   // every thread holds its position values: S1 may be written
      const f32x4* qw = reinterpret_cast<const f32x4*>(lnp_s + 2 * C) + lpart;
      const f32x4* qb = reinterpret_cast<const f32x4*>(lnp_s + 3 * C) + lpart;
      const f32x4* kw = reinterpret_cast<const f32x4*>(lnp_s + 4 * C) + lpart;
      const f32x4* kb = reinterpret_cast<const f32x4*>(lnp_s + 5 * C) + lpart;
#pragma unroll
      for (int i = 0; i < F4; ++i) {
        S1.template put4<site_act_lo(SP::Q)>(lrow, 4 * (i * TPR + lpart), (xn[i] * qw[i * TPR] + qb[i * TPR]) + ps[i]);
        S2.template put4<site_act_lo(SP::K) || site_act_lo(SP::V)>(lrow, 4 * (i * TPR + lpart),
                                                                   (xn[i] * kw[i * TPR] + kb[i * TPR]) + ps[i]);
      }
    }
    __syncthreads();
    PHASE_STAMP(p, 7);

    {  // phi(Q) -> HBM
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
      ws.template gemm<C, P_T0, C, SP::Q, SP::K>(S1, p.a.wq, p.a.wq_l, NT * wave, lane, acc, p.a.wk, p.a.wk_l,
                                                 NT * wave, p.dbg);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = crow(r, half);
          if (row < nvalid && !ABL(p.dbg, ABL_STORE))
            (p.qp + row_base * C + wcol + 32 * t)[(unsigned)(row * C + col)] =
                (FULL || ABL(p.dbg, ABL_ELU)) ? acc[t][r] : MASKED ? elu1(acc[t][r]) * msk_s[row] : elu1(acc[t][r]);
        }
    }
    PHASE_STAMP(p, 8);
    f32x16 accK[NT], accV[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { accK[t] = f32x16{0}; accV[t] = f32x16{0}; }
    ws.template gemm<C, P_T1, C, SP::K, SP::V>(S2, p.a.wk, p.a.wk_l, NT * wave, lane, accK, p.a.wv, p.a.wv_l,
                                               NT * wave, p.dbg);
    ws.template gemm<C, P_T2, 0, SP::V>(S2, p.a.wv, p.a.wv_l, NT * wave, lane, accV, nullptr, nullptr, 0,
                                        p.dbg);
    PHASE_STAMP(p, 9);
    if constexpr (FULL) {
      // K row-major, V transposed ([channel][token], padded to whole tiles: every row of
      // the tile is written - rows past the image end hold the duplicated last row, finite)
      float* vt = p.vt_out + p.vt_off[side] + ((size_t)n * C + wcol + col) * p.lpad[side] + l0 + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = crow(r, half);
        if (row < nvalid) p.kbuf_out[(row_base + row) * C + wcol + col] = accK[0][r];
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<f32x4*>(vt + 8 * g4) =
            f32x4{accV[0][4 * g4], accV[0][4 * g4 + 1], accV[0][4 * g4 + 2], accV[0][4 * g4 + 3]};
    } else if (!ABL(p.dbg, ABL_KVSTATE))
    kv_state_store<MODE, NT, MASKED>(accK, accV, ABL(p.dbg, ABL_ELU), L, nvalid, lane, wave, p.kv_out, p.ks_out, slot, rg, msk_s);
    PHASE_STAMP(p, 10);
    // (lnp_s[0] - a LayerNorm-2 weight of the layer just finished - is dead by now)
  } else if (TAIL == 1) {
    // ============ decoder preparation (transformer.py:240-246) ============
    // k = (memory + pos) Wk^T + bk ; v = memory Wv^T + bv  (no norm, no pos on v)
    {
      const f32x4* src = reinterpret_cast<const f32x4*>(S0 + lrow * LDA) + lpart;
#pragma unroll
      for (int i = 0; i < F4; ++i) {
        const f32x4 xv = src[i * TPR];
        S1.template put4<site_act_lo(SP::DEC_K)>(lrow, 4 * (i * TPR + lpart), xv + pos[i * TPR]);  // k input: memory + pos
        if (SPLIT) S2.template put4<site_act_lo(SP::DEC_V)>(lrow, 4 * (i * TPR + lpart), xv);      // v input (f32 mode reads S0)
      }
    }
    __syncthreads();
    // biases first: a load issued after the stream's run-ahead fetches would wait for them
    float bias_k[2][NT], bias_v[2][NT];
#pragma unroll
    for (int dl = 0; dl < 2; ++dl)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        bias_k[dl][t] = p.d.bk[dl][wcol + 32 * t + col];
        bias_v[dl][t] = p.d.bv[dl][wcol + 32 * t + col];
      }
    const ATile<MODE> S0t(S0, LDA, LDAH, &rg);
    const ATile<MODE>& Vin = SPLIT ? S2 : S0t;  // f32 mode reads the memory tile itself
    auto dec_layer = [&](auto DL) {
      constexpr int dl = decltype(DL)::value;
      constexpr int PK = dl == 0 ? P_T0 : P_T2, PV = dl == 0 ? P_T1 : P_T3;
      f32x16 accK[NT], accV[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accK[t][r] = bias_k[dl][t]; accV[t][r] = bias_v[dl][t]; }
      ws.template gemm<C, PK, C, SP::DEC_K, SP::DEC_V>(S1, p.d.wk[dl], p.d.wk_l[dl], NT * wave, lane, accK,
                                                       p.d.wv[dl], p.d.wv_l[dl], NT * wave, p.dbg);
      ws.template gemm<C, PV, (dl == 0 ? C : 0), SP::DEC_V, SP::DEC_K>(Vin, p.d.wv[dl], p.d.wv_l[dl], NT * wave,
                                                                       lane, accV, p.d.wk[1], p.d.wk_l[1],
                                                                       NT * wave, p.dbg);
      if constexpr (dl == 1) {
        kv_state_store<MODE, NT, MASKED>(accK, accV, ABL(p.dbg, ABL_ELU), L, nvalid, lane, wave, p.dkv1_out,
                                         p.dks1_out, slot, rg, msk_s);
      } else {
        // Decoder layer 0's query is a create-time constant q0 (decoder.hip), and
        // its cross-attention is linear in the state, so this tile contributes
        //   att0[h,v] += sum_d q0[h,d] * KV_tile[h][d][v],  z0[h] += q0[h,:].Ksum_tile[h,:]
        // instead of a full 8192-float state.
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int h = NT * wave + t;
          float ksum;
          f32x16 kv;
          kv_state_32<MODE, MASKED>(accK[t], accV[t], false, L, nvalid, half, kv, ksum, rg, msk_s);
          const float* q0 = p.dec_q0 + side * C + h * HD;
          float a = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) a += q0[crow(r, half)] * kv[r];
          a += __shfl_xor(a, 32, 64);
          ksum += __shfl_xor(ksum, 32, 64);
          float z = q0[col] * ksum;  // lanes col and col+32 hold the same value
          z += __shfl_xor(z, 1, 64);
          z += __shfl_xor(z, 2, 64);
          z += __shfl_xor(z, 4, 64);
          z += __shfl_xor(z, 8, 64);
          z += __shfl_xor(z, 16, 64);
          if (half == 0) p.att0_out[(size_t)slot * C + h * HD + col] = a;
          if (lane == 0) p.z0_out[(size_t)slot * NH + h] = z;
        }
      }
    };
    dec_layer(std::integral_constant<int, 0>{});
    dec_layer(std::integral_constant<int, 1>{});
  }
  range_report<MODE>(rg, p.flags);
}

// ===========================================================================
// 64-token variant of the fused  B(l) ; A(l+1)  kernel (split mode, 8 waves).
//
// With 32-token tiles a workgroup streams 2 MB of weight fragments per launch for
// 32 rows, and that stream (per-CU L2->register rate) bounds the GEMM phases.
// Here every fragment feeds TWO 32-row MFMA tiles, so the same stream serves 64
// tokens and the GEMM phases become MFMA/stream balanced.  LDS (160 KB) does not
// hold the 32-token layout twice, so:
//   * the residual stream lives in registers only (xacc[2]), in the TRANSPOSED accumulator
//     layout from the attention apply to MLP2 (merge, MLP1 and MLP2 run with the weights as
//     the A operand): a lane holds 16 channels of ONE token per row tile, in quads of 4
//     consecutive channels.  The x tile and phi(Q) arrive as full rows through LDS, LN2 is
//     computed from the registers (per-lane mean / M2, one 9-KB exchange, Chan's update),
//     every plane store is 8 bytes, x leaves through an f32 tile in full 1-KB rows;
//   * the MLP runs in two hidden halves of 256: MLP1a, MLP1b (+ GELU(a) -> R2), MLP2a (K = 256,
//     + GELU(b) -> R1), MLP2b, so the hidden tile needs one region per half.
//   R1: f32 [64][260]  or  planes hi|lo [64][264]      (LN staging / GEMM A operand)
//   R2: f32 [64][260] (phi(Q))  or  planes (kv input, hidden half)
// The per-wave weight stream (WStream2) is a ring of 6 k16-steps of B fragments
// (5 in flight) running ahead across GEMM calls like WStream.
// ===========================================================================
constexpr int LNX_LD = 36;                               // (padded: 16 lanes' b128 reads hit 16 bank groups)
// LDS layout (floats) of a workgroup of RTW = 64 or 32 token rows: 148 KB / 77.5 KB
template <int RTW>
struct E2 {
  static constexpr int RFL = (2 * RTW * LDAH * 2 + 3) / 4;   // one region: planes hi | lo, or the f32 tile
  static_assert(RFL >= RTW * LDA, "region holds the f32 tile too");
  static constexpr int R1 = 0, R2 = RFL, KSUM = 2 * RFL, LNP = KSUM + C;
  static constexpr int LNX = LNP + 6 * C;                    // LayerNorm partial statistics [tokens][16][mean, M2]
  static constexpr int SMEM = LNX + RTW * LNX_LD;
};

// phi(K)^T (V/S) and sum phi(K) of a 64-row tile (two MFMA row tiles) for head = wave.
// MASKED (forward_dummy's masks, linear_attention.py:37-41): msk_s[row] = the token's kv_mask value,
// 0 past the tile's last valid row - it multiplies phi(K) and V where the unmasked form has the
// row-validity 1 / 0 (an instantiation of its own: the default path's instruction stream is untouched).
template <int MODE, int NMT = 2, bool MASKED = false>
__device__ __forceinline__ void kv_state_64(const f32x16 (&accK)[NMT], const f32x16 (&accV)[NMT],
                                            int S_len, int nvalid, int half, bool two, f32x16& kv,
                                            float& ksum, Range& rg, const float* msk_s = nullptr) {
  // Branch-free phi (common.h: elu1, a median) on scalars, a few at a time: unfenced, hipcc
  // schedules all 32 exps at once and spills.  ONE code path with the row masks for every tile
  // (see the file header: no run-time choice between a masked and an unmasked form).
  const float inv_len = 1.0f / (float)S_len;
  kv = f32x16{0};
  ksum = 0.f;
  // (laundered: the row-validity tests are otherwise CSE'd with the residual loads' row
  //  clamps at the top of the kernel and 32 values live - spilled - until here)
  int nv2 = nvalid - 4 * half;
  asm volatile("" : "+v"(nv2));
  if constexpr (MODE == GM_SPLIT && OETR_SPLIT_STATE) {
    // Round 5: the contraction over a row tile's 32 tokens as 2 k16 steps of the fp32-class split
    // (6 f16 MFMAs of 32 cycles instead of 16 f32 MFMAs of 64: the largest single residual of the
    // encoder launch, profiles/r4_ablation_bounds.txt).  k-slot i of step s <-> token
    // crow(8 s + i, half) - the same bijection for A = phi(K)^T and B = V/S.  ONE code path: the row
    // masks on every tile, the MFMA triples fenced (mma16_split3) - the form of round 4's hazard
    // study that never produced a differing forward (single-path builds: 0 of 157 000 under both
    // amplifiers; the failures needed a run-time choice between two forms of this block), re-soaked
    // in round 5 (profiles/r5_determinism_soak.txt).
    f32x16 c1 = {0};
#if OETR_SOAK_AMP & 2   // soak builds only: the amplifier of round 4's study (the wave outruns its SIMD sibling)
    __builtin_amdgcn_s_setprio(3);
#endif
    auto row_tile = [&](auto MT_) {
      constexpr int mt = decltype(MT_)::value;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        f32x4 k0, k1, v0, v1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = 8 * st + i;
          float m;
          if constexpr (MASKED) m = msk_s[32 * mt + crow(r, half)];
          else m = 32 * mt + crow(r, 0) < nv2 ? 1.0f : 0.0f;
          const float kk = elu1(accK[mt][r]) * m;
          const float vv = accV[mt][r] * (inv_len * m);
          ksum += kk;
          if (i < 4) { k0[i] = kk; v0[i] = vv; } else { k1[i - 4] = kk; v1[i - 4] = vv; }
        }
        f32x4 ah, al, bh, bl;
        split8(k0, k1, ah, al, rg);
        split8(v0, v1, bh, bl, rg);
        mma16_split3(ah, al, bh, bl, kv, c1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    row_tile(std::integral_constant<int, 0>{});
    if constexpr (NMT == 2) {
      if (two) row_tile(std::integral_constant<int, 1>{});   // (else: no valid row in the second row tile)
    }
#if OETR_SOAK_AMP & 2
    __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) kv[r] = fmaf(c1[r], SPLIT_INV, kv[r]);
  } else {
    // (see kv_state_32: the operands of the next register under the current f32 MFMA)
    auto row_tile = [&](auto MT_) {
      constexpr int mt = decltype(MT_)::value;
      auto operands = [&](int r, float& k, float& v) {
        float m;
        if constexpr (MASKED) m = msk_s[32 * mt + crow(r, half)];
        else m = 32 * mt + crow(r, 0) < nv2 ? 1.0f : 0.0f;
        const float x = accK[mt][r];
        k = (elu1(x)) * m;
        ksum += k;
        v = accV[mt][r] * (inv_len * m);
      };
      float kc, vc;
      operands(0, kc, vc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float kn = 0.f, vn = 0.f;
        if (r + 1 < 16) operands(r + 1, kn, vn);
        kv = __builtin_amdgcn_mfma_f32_32x32x2f32(kc, vc, kv, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        kc = kn; vc = vn;
      }
    };
    row_tile(std::integral_constant<int, 0>{});
    if constexpr (NMT == 2) {
      if (two) row_tile(std::integral_constant<int, 1>{});   // (else: no valid row in the second row tile)
    }
  }
  ksum += __shfl_xor(ksum, 32, 64);
}
__device__ __forceinline__ void kv_state_write(const f32x16& kv, float ksum, int lane, int wave,
                                               float* __restrict__ kv_out,
                                               float* __restrict__ ks_out, int slot) {
  f32x4* dst = reinterpret_cast<f32x4*>(kv_out) + ((size_t)slot * NH + wave) * 4 * 64;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 o = {kv[4 * q], kv[4 * q + 1], kv[4 * q + 2], kv[4 * q + 3]};
    store16<(OETR_WT & 4) != 0>(reinterpret_cast<float*>(dst), (q * 64 + (unsigned)lane) * 16u, o);
  }
  if (lane < 32) (ks_out + (size_t)slot * C + wave * HD)[(unsigned)lane] = ksum;
}

// Body of k_encoder64 for one workgroup.  ROWS (WStream2T): 2 = both 32-row MFMA tiles hold valid
// rows, 1 = only the first does (a ragged last tile of an image: every piece of work on the
// second row tile is compiled out), 0 = decided at run time (single-plane modes).
// RTW: token rows of the workgroup, 64 (two MFMA row tiles) or 32 (one: round 4's 32-row encoder,
// the same body - transposed residual stream, fragment-major phi(Q), epilogues beside MFMAs -
// with a weight fragment feeding 3 MFMAs instead of 6).
// MASKED: p.mask[side] holds forward_dummy's mask of the side's images, [N][L] floats (src/model.py:229;
// x_mask / source_mask of every encoder layer, transformer.py:349-358, memory_mask of the decoder,
// :361-381): phi(Q) rows are multiplied by the token's mask (q_mask), phi(K) and V rows likewise
// (kv_mask, kv_state_64) - in phase A of every layer and in the decoder preparation.
template <bool HAS_B, int TAIL, int MODE, int POL, int ROWS, int RTW = RT, bool MASKED = false>
__device__ __forceinline__ void encoder64_body(const EncLaunch& p, float* smem) {
  constexpr int NMT = RTW / 32, THREADS = 512, TPR = THREADS / RTW, F4 = 64 / TPR;
  using L2 = E2<RTW>;
  using SP = SitePolicy<POL>;   // arithmetic per GEMM site (two-plane mode only)
  float* R1f = smem + L2::R1;
  float* R2f = smem + L2::R2;
  Range rg;
  const PlanesT<MODE, RTW> P1(R1f, &rg), P2(R2f, &rg);
  // f32 staging tile of the residual stream between phase B and the tail (LayerNorm input):
  // R2 after phase B (R1 then holds the second hidden half until MLP2b has read it)
  float* Xf = HAS_B ? R2f : R1f;
  float* ksum_s = smem + L2::KSUM;
  float* lnp_s = smem + L2::LNP;
  float* lnx_s = smem + L2::LNX;
  using WS = WStream2T<MODE, ROWS, NMT>;
  WS ws;
  constexpr int P_MERGE = 0;
  // (in EXECUTION order: merge, MLP1a, MLP1b, MLP2a, MLP2b - round 3 chained them as 1a, 2a, 1b, 2b, which
  //  only a ring depth dividing 16 forgives)
  constexpr int P_1A = WS::adv(P_MERGE), P_1B = WS::adv(P_1A), P_2A = WS::adv(P_1B), P_2B = WS::adv(P_2A);
  constexpr int P_T0 = HAS_B ? WS::adv(P_2B) : 0;
  constexpr int P_T1 = WS::adv(P_T0), P_T2 = WS::adv(P_T1), P_T3 = WS::adv(P_T2);

  const Geom& g = p.g;
  // (the wave index as a SCALAR: everything derived from it - weight slab pointers, row indices
  //  of the row-per-wave copies - is then SALU work and the loads use SGPR bases)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int wcol = 32 * wave;
  const int lrow = tid / TPR, lpart = tid % TPR;

  const int logical = xcd_remap(blockIdx.x, g.ntiles);
  const int per = g.nt[0] + g.nt[1];
  const int n = logical / per;
  const int rem = logical - n * per;
  const int side = rem >= g.nt[0];
  const int t_idx = side ? rem - g.nt[0] : rem;
  const int L = g.L[side];
  const int l0 = t_idx * RTW;
  const int nvalid = min(RTW, L - l0);
  const size_t row_base = (size_t)g.row0[side] + (size_t)n * L + l0;
  const int slot = g.tile0[side] + n * g.nt[side] + t_idx;
  const size_t qrow_base = (size_t)slot * RTW;   // this tile's rows of the TILE-major phi(Q) buffer
  const bool nchw = !HAS_B && p.feat_nchw[0] != nullptr;   // first launch on NCHW inputs (launch-uniform)
  ws.set_rows(nvalid);
  ws.set_lane(lane);

  stage_ln_params<THREADS, HAS_B, TAIL>(lnp_s, p, tid);
  // position rows of this tile for the tail (issued early: nothing waits on them yet)
  const f32x4* pos = reinterpret_cast<const f32x4*>(
                         p.pos + (size_t)(g.prow0[side] + l0 + min(lrow, nvalid - 1)) * C) + lpart;

  f32x16 xacc[NMT];  // residual stream of this wave's 32 channels, both row tiles (transposed C layout)
  PHASE_STAMP(p, 0);
  if (!HAS_B) PHASE_STAMP(p, 13);   // (first launch: slots 13 / 14 survive the <B,..> launches behind it - tools/phase_timing_a.py)

  if (HAS_B) {
    const int ss = p.b_cross ? 1 - side : side;
    const int S_len = g.L[ss];
    // (kv_reduced: one pre-reduced state per image - k_kv_reduce - instead of the tiles' partials)
    const int nts = p.kv_reduced ? 1 : g.nt[ss];
    const int src_slot0 = p.kv_reduced ? ss * g.N + n : g.tile0[ss] + n * g.nt[ss];

    // ---- round 3: the residual stream lives in the TRANSPOSED accumulator layout (lane = token,
    //      register quads = 4 consecutive channels), merge / MLP2 run transposed like MLP1 ----
    // reduce the source image's partial KV states (fixed order): the registers are this head's
    // state in MFMA fragment order (A operand of the transposed apply)
    f32x4 kvB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) kvB[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int KVR = 7;
      const unsigned ln = lane, kl = tid & (C - 1);
      const f32x4* kvp = reinterpret_cast<const f32x4*>(p.kv_in) + ((size_t)src_slot0 * NH + wave) * 256;
      const float* ksp = p.ks_in + (size_t)src_slot0 * C;
      float ks = 0.f;
      if (nts == 1) {   // pre-reduced (or a single tile): one state, no redundant clamped loads
#pragma unroll
        for (int e = 0; e < 4; ++e) kvB[e] = kvp[e * 64 + ln];
        ks = ksp[kl];
      } else
      for (int ti0 = 0; ti0 < nts; ti0 += KVR) {
        f32x4 tmp[KVR][4];
        float kt[KVR];
#pragma unroll
        for (int u = 0; u < KVR; ++u) {
          const int ti = min(ti0 + u, nts - 1);
#pragma unroll
          for (int e = 0; e < 4; ++e) tmp[u][e] = kvp[(size_t)ti * (NH * 256) + e * 64 + ln];
          kt[u] = ksp[(size_t)ti * C + kl];
        }
#pragma unroll
        for (int u = 0; u < KVR; ++u)
          if (ti0 + u < nts) {
#pragma unroll
            for (int e = 0; e < 4; ++e) kvB[e] += tmp[u][e];
            ks += kt[u];
          }
      }
      if (tid < C) ksum_s[tid] = ks;
    }
    ws.template prime<C, P_MERGE, SP::MERGE>(p.b.wmerge, p.b.wmerge_l, wave, 0, lane);
    // phi(Q) of this wave's head for the lane's token, both row tiles: the B fragments of the
    // transposed apply exactly as phase A left them (round 4: FRAGMENT-major phi(Q) buffer,
    // [slot][head][row tile][4][64 lanes] 16-byte units = {step 0 hi, step 0 lo, step 1 hi, step 1 lo}
    // f16 planes - or four f32x4 k-groups in the bf16 build, whose apply runs on f32 MFMAs).  Eight
    // coalesced 1-KB loads per wave straight into registers: no LDS staging tile, no conversion here.
    f32x4 qfrag[NMT][4];
    {
      const f32x4* qf = reinterpret_cast<const f32x4*>(p.qp + qrow_base * C) + (size_t)wave * NMT * 4 * 64;
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int u = 0; u < 4; ++u) qfrag[mt][u] = qf[(mt * 4 + u) * 64 + (unsigned)lane];
    }
    // residual x of token (32 mt + col), channels wcol + 8 g + 4 half + 0..3: straight into the
    // merge GEMM's accumulators, issued LAST (the loads return in order - nothing waits on these
    // before the merge GEMM, the state reduction and the apply run under them)
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      const float* xr = p.x + (row_base + min(32 * mt + col, nvalid - 1)) * C + wcol + 4 * half;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 8 * g4);
#pragma unroll
        for (int i = 0; i < 4; ++i) xacc[mt][4 * g4 + i] = v[i];
      }
    }
    __syncthreads();
    PHASE_STAMP(p, 1);

    // message^T[v][t] = sum_d KV^T[v][d] phi(Q)^T[d][t], and the normaliser's dot product
    // phi(Q)[t,:] . Ksum as one more row product of the same B fragments (every accumulator row
    // of `zacc` is that dot): Z needs no LDS pass of its own.  Lane = token t, registers = v.
    {
      f32x4 kvh[2], kvl[2], ksh[2], ksl[2];
      const float inv_S = 1.0f / (float)S_len;
      (void)inv_S;
      if constexpr (gm_f16_range(MODE)) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          split8(kvB[2 * s2], kvB[2 * s2 + 1], kvh[s2], kvl[s2], rg);
          // (Ksum / S - the mean of phi(K), inside the f16 range whenever phi(K) is - in the
          //  k-slot order of the fragments)
          const float* kr = ksum_s + wave * HD + 4 * half + 16 * s2;
          split8(*reinterpret_cast<const f32x4*>(kr) * inv_S, *reinterpret_cast<const f32x4*>(kr + 8) * inv_S,
                 ksh[s2], ksl[s2], rg);
        }
      }
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        if (mt == 1 && !ws.two()) break;  // ragged tile: rows 32.. are never stored
        f32x16 macc = {0}, zacc = {0};
        if constexpr (gm_f16_range(MODE)) {
          f32x16 c1 = {0}, cz = {0};
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const f32x4 qh = qfrag[mt][2 * s2], ql = qfrag[mt][2 * s2 + 1];
            mma16_split3(kvh[s2], kvl[s2], qh, ql, macc, c1);
            mma16_split3(ksh[s2], ksl[s2], qh, ql, zacc, cz);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) macc[r] = fmaf(c1[r], SPLIT_INV, macc[r]);
          zacc[0] = fmaf(cz[0], SPLIT_INV, zacc[0]);
        } else {
#pragma unroll
          for (int ks4 = 0; ks4 < 4; ++ks4) {
            const f32x4 q = qfrag[mt][ks4];
            const f32x4 kk = *reinterpret_cast<const f32x4*>(ksum_s + wave * HD + 4 * half + ks4 * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              macc = __builtin_amdgcn_mfma_f32_32x32x2f32(kvB[ks4][j], q[j], macc, 0, 0, 0);
              zacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[j], q[j], zacc, 0, 0, 0);
            }
          }
        }
        const float zdot = gm_f16_range(MODE) ? zacc[0] * (float)S_len : zacc[0];
        const float zs = (float)S_len / (zdot + ATTN_EPS);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          P1.put4(32 * mt + col, wcol + 8 * g4 + 4 * half,
                  f32x4{macc[4 * g4] * zs, macc[4 * g4 + 1] * zs, macc[4 * g4 + 2] * zs, macc[4 * g4 + 3] * zs});
      }
    }
    __syncthreads();
    PHASE_STAMP(p, 2);

    // x1 = x + message . Wmerge^T   (transposed product)
    ws.template gemm<C, P_MERGE, true, C, SP::MERGE, SP::MLP1, true>(P1, p.b.wmerge, p.b.wmerge_l, wave, 0, lane,
                                                                     xacc, p.b.w1, p.b.w1_l, wave, 0);
    PHASE_STAMP(p, 3);
    // LN2(x1) from the registers: per lane the statistics of its 16 channels of a token (mean,
    // M2), the 16 partials of a token (8 waves x 2 half-waves) combined after ONE exchange
    // through LDS (Chan's parallel update: as accurate as the two-pass form)
    {
      float pm[NMT], pq[NMT];
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += xacc[mt][r];
        pm[mt] = sum * (1.0f / 16.0f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = xacc[mt][r] - pm[mt]; q = fmaf(d, d, q); }
        pq[mt] = q;
        *reinterpret_cast<f32x2*>(lnx_s + (32 * mt + col) * LNX_LD + 2 * (2 * wave + half)) = f32x2{pm[mt], pq[mt]};
      }
      __syncthreads();   // partials visible; every wave is done reading the message planes
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        if (mt == 1 && !ws.two()) break;
        const f32x4* pp = reinterpret_cast<const f32x4*>(lnx_s + (32 * mt + col) * LNX_LD);
        f32x4 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = pp[i];   // (mean, M2) x 16 partials
        float msum = 0.f, qsum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { msum += t[i][0] + t[i][2]; qsum += t[i][1] + t[i][3]; }
        const float mean = msum * (1.0f / 16.0f);
        float dev = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d0 = t[i][0] - mean, d1 = t[i][2] - mean;
          dev = fmaf(d0, d0, dev);
          dev = fmaf(d1, d1, dev);
        }
        const float rstd = 1.0f / sqrtf((qsum + 16.0f * dev) * (1.0f / C) + LN_EPS);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int c0 = wcol + 8 * g4 + 4 * half;
          const f32x4 gw = *reinterpret_cast<const f32x4*>(lnp_s + c0), gb = *reinterpret_cast<const f32x4*>(lnp_s + C + c0);
          f32x4 y;
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = (xacc[mt][4 * g4 + i] - mean) * rstd * gw[i] + gb[i];
          P1.put4(32 * mt + col, c0, y);
        }
      }
    }
    __syncthreads();

    PHASE_STAMP(p, 4);
    // MLP in two hidden halves, the GELU epilogue of each half issued under the NEXT GEMM's MFMAs:
    //   MLP1a | MLP1b + gelu(a) -> R2 | barrier | MLP2a(R2) + gelu(b) -> R1 | barrier | MLP2b(R1)
    // (R1 = the LN2 planes, free once every wave has finished MLP1b: the first barrier)
    {
      constexpr int SNEXT = TAIL == 0 ? SP::Q : SP::DEC_K;   // first GEMM of the tail
      constexpr bool XTR = true;                            // residual stream in the transposed layout
      f32x16 haccA[NMT] = {}, haccB[NMT] = {};
      // MLP1 runs TRANSPOSED (WStream2T: TR): a lane then holds, per register quad, FOUR
      // CONSECUTIVE hidden channels of one token - the GELU epilogue writes its planes with
      // 8-byte LDS stores (2 per quad) instead of 2-byte ones (8 per quad), one quad = two
      // interleaved packed-GELU chains every other k16 step
      ws.template gemm<C, P_1A, true, C, SP::MLP1, SP::MLP1, true>(P1, p.b.w1, p.b.w1_l, wave, 0, lane, haccA,
                                                                   p.b.w1, p.b.w1_l, 8 + wave, 0);
      const bool two = ws.two();
      auto gelu_to = [&](const PlanesT<MODE, RTW>& dst, const f32x16 (&h)[NMT]) {
        return [&, two](auto CI_) {
          // 4 register quads per row tile over the 16 k16 steps: every 2nd step (two row tiles) / every 4th (one)
          constexpr int CI = decltype(CI_)::value, mt = NMT == 2 ? CI / 8 : 0, g4 = NMT == 2 ? (CI % 8) / 2 : CI / 4;
          if constexpr (CI % (NMT == 2 ? 2 : 4) == 0) {
            if (mt == 1 && !two) return;   // (hidden rows 32.. stay unwritten: never consumed)
            const f32x4 ge = gelu_erf4(f32x4{h[mt][4 * g4], h[mt][4 * g4 + 1], h[mt][4 * g4 + 2], h[mt][4 * g4 + 3]});
            dst.template put4<site_act_lo(SP::MLP2)>(32 * mt + col, wcol + 8 * g4 + 4 * half, ge);
          }
        };
      };
      auto epiA = gelu_to(P2, haccA);
      ws.template gemm_epi<C, P_1B, true, FF, SP::MLP1, SP::MLP2, true>(P1, p.b.w1, p.b.w1_l, 8 + wave, 0, lane,
                                                                        haccB, p.b.w2, p.b.w2_l, wave, 0, epiA);
      __syncthreads();   // hidden half a complete; every wave is done reading the LN2 planes
      PHASE_STAMP(p, 5);
      auto epiB = gelu_to(P1, haccB);
      ws.template gemm_epi<FF, P_2A, true, FF, SP::MLP2, SP::MLP2, XTR>(P2, p.b.w2, p.b.w2_l, wave, 0, lane, xacc,
                                                                        p.b.w2, p.b.w2_l, wave, 16, epiB);
      __syncthreads();   // hidden half b complete
      PHASE_STAMP(p, 6);
      PHASE_STAMP(p, 7);
      ws.template gemm<FF, P_2B, (TAIL != 2), C, SP::MLP2, SNEXT, XTR>(P1, p.b.w2, p.b.w2_l, wave, 16, lane, xacc,
                                                                       TAIL == 0 ? p.a.wq : p.d.wk[0],
                                                                       TAIL == 0 ? p.a.wq_l : p.d.wk_l[0], wave, 0);
    }
    // x1 tile -> the f32 staging region (one 16-byte LDS store per register quad), then HBM in
    // full 1-KB rows.  (The region's planes were last read by a GEMM every wave finished before
    // a barrier above: R2 by MLP2a; R1 holds the second hidden half, which MLP2b reads.)
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<f32x4*>(Xf + (32 * mt + col) * LDA + wcol + 8 * g4 + 4 * half) =
            f32x4{xacc[mt][4 * g4], xacc[mt][4 * g4 + 1], xacc[mt][4 * g4 + 2], xacc[mt][4 * g4 + 3]};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RTW / 8; ++i) {
      const int r = wave + 8 * i;   // (scalar: a wave-uniform branch, an SGPR row address)
      if (r < nvalid)
        store16<(OETR_WT & 1) != 0>(p.x + (row_base + r) * C, 16u * (unsigned)lane,
                                    *reinterpret_cast<const f32x4*>(Xf + r * LDA + 4 * lane));
    }
    PHASE_STAMP(p, 8);
    if (TAIL == 2) { range_report<MODE>(rg, p.flags); return; }
  } else {
    if (nchw) {  // first launch, reference layout: transpose on the way in (R2 is free until phase A writes P2)
      load_tile_nchw<THREADS, RTW>(R1f, p.feat_nchw[side] + (size_t)n * C * L + l0, L, nvalid, tid);
      load_tile_nchw<THREADS, RTW>(R2f, p.pos_nchw[side] + l0, L, nvalid, tid);
    } else {
#pragma unroll
      for (int i = 0; i < (RTW * C / 4) / THREADS; ++i) {  // first launch: x from HBM
        const int idx = tid + THREADS * i;
        const int r = idx >> 6, c4 = idx & 63;
        *reinterpret_cast<f32x4*>(R1f + r * LDA + 4 * c4) =
            reinterpret_cast<const f32x4*>(p.x + (row_base + min(r, nvalid - 1)) * C)[c4];
      }
    }
    if (TAIL == 0) ws.template prime<C, P_T0, SP::Q>(p.a.wq, p.a.wq_l, wave, 0, lane);
    if (TAIL == 1) ws.template prime<C, P_T0, SP::DEC_K>(p.d.wk[0], p.d.wk_l[0], wave, 0, lane);
    __syncthreads();
    if (nchw) {  // token-major copies for the later launches (residual reads, position rows)
      store_tile_tokens<THREADS, RTW>(p.x + row_base * C, R1f, nvalid, tid);
      if (n == 0) store_tile_tokens<THREADS, RTW>(p.pos_out + (size_t)(g.prow0[side] + l0) * C, R2f, nvalid, tid);
    }
    PHASE_STAMP(p, 14);   // (first launch: input tile in LDS - transposed from NCHW - and stored token-major)
  }

  // (MASKED: the tile's mask values go where the LayerNorm exchange buffer was - dead after phase B;
  //  two barriers lie between this store and the first read)
  if constexpr (MASKED && TAIL != 2) {
    if (tid < RTW) lnx_s[tid] = tid < nvalid ? p.mask[side][(size_t)n * L + l0 + tid] : 0.f;
  }
  if (TAIL == 0) {
    // ================= phase A: start layer l+1 =================
    {
      f32x4 xn[F4], ps[F4];
      if (!HAS_B && nchw) {   // the position tile sits (f32, transposed on load) where P2 goes
        const f32x4* pl = reinterpret_cast<const f32x4*>(R2f + lrow * LDA) + lpart;
#pragma unroll
        for (int i = 0; i < F4; ++i) ps[i] = pl[i * TPR];
      } else {
#pragma unroll
        for (int i = 0; i < F4; ++i) ps[i] = pos[i * TPR];
      }
      ln_rows<TPR, F4>(Xf, tid, xn, 0);
      __syncthreads();
      const f32x4* qw = reinterpret_cast<const f32x4*>(lnp_s + 2 * C) + lpart;
      const f32x4* qb = reinterpret_cast<const f32x4*>(lnp_s + 3 * C) + lpart;
      const f32x4* kw = reinterpret_cast<const f32x4*>(lnp_s + 4 * C) + lpart;
      const f32x4* kb = reinterpret_cast<const f32x4*>(lnp_s + 5 * C) + lpart;
#pragma unroll
      for (int i = 0; i < F4; ++i) {
        P1.template put4<site_act_lo(SP::Q)>(lrow, 4 * (i * TPR + lpart), (xn[i] * qw[i * TPR] + qb[i * TPR]) + ps[i]);
        P2.template put4<site_act_lo(SP::K) || site_act_lo(SP::V)>(lrow, 4 * (i * TPR + lpart),
                                                                   (xn[i] * kw[i * TPR] + kb[i * TPR]) + ps[i]);
      }
    }
    __syncthreads();
    PHASE_STAMP(p, 9);
    // phi(Q) -> HBM, issued under the K GEMM's MFMAs (two accumulator values per k16 step).  The
    // phi(Q) buffer is TILE-major ([slot][64][256], rows past an image's end are padding), so
    // every store is unconditional: an SGPR row address, one per-lane offset, no select
    f32x16 accQ[NMT] = {};   // (value-initialised: a zeroing loop here cost the first launch's kernel 33 VGPRs and 28 spills)
    // (issuing the x store under this GEMM, or phi(K) under the V GEMM below, was measured
    //  neutral to slightly slower - one-process A/B, 51.6 vs 51.9 us; only the GELU epilogues
    //  and the phi(Q) store pay for the interleave)
    ws.template gemm<C, P_T0, true, C, SP::Q, SP::K, true>(P1, p.a.wq, p.a.wq_l, wave, 0, lane, accQ, p.a.wk,
                                                           p.a.wk_l, wave, 0);
    PHASE_STAMP(p, 10);
    f32x16 accK[NMT] = {}, accV[NMT] = {};
    {
      // phi(Q) -> HBM under the K GEMM's MFMAs, one register pair per k16 step.  The Q GEMM ran
      // TRANSPOSED: lane (token 32 mt + col, half) holds channels 8 g + 4 half + i of its head in
      // register 4 g + i - which is, for k16 step s2 of the apply, exactly the B fragment of that
      // lane (k-slot 8 half + i' <-> d = 16 s2 + 8 (i' >> 2) + 4 half + (i' & 3)): registers
      // 8 s2 .. 8 s2 + 7.  So the wave stores the consumer's register image, split into the f16
      // planes here (6 VALU per pair beside MFMAs instead of in the apply), 16 bytes per lane and
      // 1 KB per instruction: 8 stores per lane instead of 32.
      f32x4* qf = reinterpret_cast<f32x4*>(p.qp + qrow_base * C) + (size_t)wave * NMT * 4 * 64;   // (scalar)
      const bool two = ws.two();
      u32x4 qhi = {0, 0, 0, 0}, qlo = {0, 0, 0, 0};
      auto qepi = [&](auto CI_) {
        // 8 register pairs per row tile over the 16 k16 steps: one per step (two row tiles) / every 2nd step (one)
        constexpr int CI = decltype(CI_)::value, mt = NMT == 2 ? CI / 8 : 0, pr = NMT == 2 ? CI % 8 : CI / 2;
        constexpr int s2 = pr / 4, j = pr % 4;
        if constexpr (NMT == 1 && CI % 2 == 1) return;
        if (mt == 1 && !two) return;   // (rows 32.. of a one-row-tile workgroup are never read)
        // (MASKED: q_mask of the lane's token, from the tile's mask values in LDS)
        const float qmask = MASKED ? lnx_s[32 * mt + col] : 1.0f;
        const float a = MASKED ? elu1(accQ[mt][2 * pr]) * qmask : elu1(accQ[mt][2 * pr]);
        const float b = MASKED ? elu1(accQ[mt][2 * pr + 1]) * qmask : elu1(accQ[mt][2 * pr + 1]);
        if constexpr (MODE == GM_BF16) {   // the bf16 build's apply takes f32 fragments: k-group = register quad
          if constexpr (pr % 2 == 0) { qhi[0] = __builtin_bit_cast(uint32_t, a); qhi[1] = __builtin_bit_cast(uint32_t, b); }
          else {
            qhi[2] = __builtin_bit_cast(uint32_t, a); qhi[3] = __builtin_bit_cast(uint32_t, b);
            store16<(OETR_WT & 2) != 0>(reinterpret_cast<float*>(qf), ((mt * 4 + pr / 2) * 64 + (unsigned)lane) * 16u,
                                        __builtin_bit_cast(f32x4, qhi));
          }
        } else {
          uint32_t h, l;
          cvt_planes2<GM_SPLIT>(a, b, h, l, rg);
          qhi[j] = h; qlo[j] = l;
          if constexpr (j == 3) {
            store16<(OETR_WT & 2) != 0>(reinterpret_cast<float*>(qf), ((mt * 4 + 2 * s2) * 64 + (unsigned)lane) * 16u,
                                        __builtin_bit_cast(f32x4, qhi));
            store16<(OETR_WT & 2) != 0>(reinterpret_cast<float*>(qf), ((mt * 4 + 2 * s2 + 1) * 64 + (unsigned)lane) * 16u,
                                        __builtin_bit_cast(f32x4, qlo));
          }
        }
      };
      ws.template gemm_epi<C, P_T1, true, C, SP::K, SP::V>(P2, p.a.wk, p.a.wk_l, wave, 0, lane, accK, p.a.wv,
                                                           p.a.wv_l, wave, 0, qepi);
    }
    ws.template gemm<C, P_T2, false, C, SP::V>(P2, p.a.wv, p.a.wv_l, wave, 0, lane, accV, nullptr, nullptr,
                                               0, 0);
    PHASE_STAMP(p, 11);
    f32x16 kv;
    float ksum;
    kv_state_64<MODE, NMT, MASKED>(accK, accV, L, nvalid, half, ws.two(), kv, ksum, rg, lnx_s);
    kv_state_write(kv, ksum, lane, wave, p.kv_out, p.ks_out, slot);
    PHASE_STAMP(p, 12);
    // (the LayerNorm exchange buffer is dead after phase B)
  } else if (TAIL == 1) {
    // ============ decoder preparation (transformer.py:240-246) ============
    float bias_k[2], bias_v[2];
#pragma unroll
    for (int dl = 0; dl < 2; ++dl) {
      bias_k[dl] = p.d.bk[dl][wcol + col];
      bias_v[dl] = p.d.bv[dl][wcol + col];
    }
    {
      f32x4 xv[F4], ps[F4];
      const f32x4* src = reinterpret_cast<const f32x4*>(Xf + lrow * LDA) + lpart;
#pragma unroll
      for (int i = 0; i < F4; ++i) { xv[i] = src[i * TPR]; ps[i] = pos[i * TPR]; }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < F4; ++i) {
        P1.template put4<site_act_lo(SP::DEC_K)>(lrow, 4 * (i * TPR + lpart), xv[i] + ps[i]);  // k input: memory + pos
        P2.template put4<site_act_lo(SP::DEC_V)>(lrow, 4 * (i * TPR + lpart), xv[i]);          // v input: memory
      }
    }
    __syncthreads();
    auto dec_layer = [&](auto DL) {
      constexpr int dl = decltype(DL)::value;
      constexpr int PK = dl == 0 ? P_T0 : P_T2, PV = dl == 0 ? P_T1 : P_T3;
      f32x16 accK[NMT], accV[NMT];
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { accK[mt][r] = bias_k[dl]; accV[mt][r] = bias_v[dl]; }
      ws.template gemm<C, PK, true, C, SP::DEC_K, SP::DEC_V>(P1, p.d.wk[dl], p.d.wk_l[dl], wave, 0, lane, accK,
                                                             p.d.wv[dl], p.d.wv_l[dl], wave, 0);
      // (no run-ahead across the state epilogue below: K, V, the state and the exp
      //  temporaries already fill the register file; layer 1's K slab is primed after it)
      ws.template gemm<C, PV, false, C, SP::DEC_V>(P2, p.d.wv[dl], p.d.wv_l[dl], wave, 0, lane, accV, nullptr,
                                                   nullptr, 0, 0);
      f32x16 kv;
      float ksum;
      kv_state_64<MODE, NMT, MASKED>(accK, accV, L, nvalid, half, ws.two(), kv, ksum, rg, lnx_s);
      if constexpr (dl == 1) {
        kv_state_write(kv, ksum, lane, wave, p.dkv1_out, p.dks1_out, slot);
      } else {
        // partial message / normaliser of decoder layer 0's cross-attention (see k_encoder)
        const float* q0 = p.dec_q0 + side * C + wave * HD;
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += q0[crow(r, half)] * kv[r];
        a += __shfl_xor(a, 32, 64);
        float z = q0[col] * ksum;
        z += __shfl_xor(z, 1, 64);
        z += __shfl_xor(z, 2, 64);
        z += __shfl_xor(z, 4, 64);
        z += __shfl_xor(z, 8, 64);
        z += __shfl_xor(z, 16, 64);
        if (half == 0) p.att0_out[(size_t)slot * C + wave * HD + col] = a;
        if (lane == 0) p.z0_out[(size_t)slot * NH + wave] = z;
        ws.template prime<C, P_T2, SP::DEC_K>(p.d.wk[1], p.d.wk_l[1], wave, 0, lane);
      }
    };
    dec_layer(std::integral_constant<int, 0>{});
    dec_layer(std::integral_constant<int, 1>{});
  }
  range_report<MODE>(rg, p.flags);
}

template <bool HAS_B, int TAIL, int MODE, int POL = 0, bool MASKED = false>
__global__ __launch_bounds__(512) void k_encoder64(EncLaunch p) {
  __shared__ __attribute__((aligned(16))) float smem[E2<RT>::SMEM];
  if constexpr (gm_planes(MODE) == 2) {
    // split mode: the row-tile count is a compile-time property of the body (branch-free GEMM
    // steps, which is what lets hipcc interleave the epilogue slices with the MFMAs), chosen
    // per workgroup - a workgroup-uniform branch at the top instead of one around every MFMA
    const Geom& g = p.g;
    const int logical = xcd_remap(blockIdx.x, g.ntiles);
    const int per = g.nt[0] + g.nt[1];
    const int rem = logical - (logical / per) * per;
    const int side = rem >= g.nt[0];
    const int t_idx = side ? rem - g.nt[0] : rem;
    const int nvalid = min(RT, g.L[side] - t_idx * RT);
    if (__builtin_amdgcn_readfirstlane(nvalid > 32 ? 1 : 0)) encoder64_body<HAS_B, TAIL, MODE, POL, 2, RT, MASKED>(p, smem);
    else encoder64_body<HAS_B, TAIL, MODE, POL, 1, RT, MASKED>(p, smem);
  } else {
    encoder64_body<HAS_B, TAIL, MODE, POL, 0, RT, MASKED>(p, smem);
  }
}

// 32 token rows per workgroup on the same body (round 4; two-plane mode): one MFMA row tile, 77.5 KB
// of LDS.  Replaces k_encoder<..., 8> for the linear-attention split builds - that kernel stays for
// the exact-fp32 mode (4 waves), the single-plane modes and attention = 'full'.
template <bool HAS_B, int TAIL, int MODE, int POL = 0, bool MASKED = false>
__global__ __launch_bounds__(512) void k_encoder32m(EncLaunch p) {
  __shared__ __attribute__((aligned(16))) float smem[E2<TM>::SMEM];
  encoder64_body<HAS_B, TAIL, MODE, POL, 2, TM, MASKED>(p, smem);
}

#ifndef OETR_SPLIT_WAVES
#define OETR_SPLIT_WAVES 8
#endif
#ifndef OETR_F32_WAVES
#define OETR_F32_WAVES 4
#endif

template <int MODE>
static hipError_t launch_encoder_mode(const EncLaunch& p, bool has_b, int tail, hipStream_t s) {
  const dim3 grid(p.g.ntiles);
  if (p.mask[0] || p.mask[1]) {
    // forward_dummy's masks: built for the two-plane arithmetic (every site fp32-class, or the
    // precision policy), linear attention, both tile sizes
    if constexpr (MODE == GM_SPLIT) {
      if (!p.mask[0] || !p.mask[1] || (p.policy != 0 && p.policy != 1) || p.attn_full) return hipErrorInvalidValue;
#define OETR_LAUNCHM(B, T)                                                                                    \
  do {                                                                                                        \
    if (p.policy == 1) {                                                                                      \
      if (p.tile_rows == RT) hipLaunchKernelGGL((k_encoder64<B, T, MODE, 1, true>), grid, dim3(512), 0, s, p);  \
      else hipLaunchKernelGGL((k_encoder32m<B, T, MODE, 1, true>), grid, dim3(512), 0, s, p);                   \
    } else {                                                                                                  \
      if (p.tile_rows == RT) hipLaunchKernelGGL((k_encoder64<B, T, MODE, 0, true>), grid, dim3(512), 0, s, p);  \
      else hipLaunchKernelGGL((k_encoder32m<B, T, MODE, 0, true>), grid, dim3(512), 0, s, p);                   \
    }                                                                                                         \
  } while (0)
      if (has_b) {
        if (tail == 0) OETR_LAUNCHM(true, 0);
        else if (tail == 1) OETR_LAUNCHM(true, 1);
        else OETR_LAUNCHM(true, 2);
      } else {
        if (tail == 0) OETR_LAUNCHM(false, 0);
        else return hipErrorInvalidValue;   // (no B phase + a tail = zero encoder layers: api.hip rejects enc_layers < 1 before any launch)
      }
#undef OETR_LAUNCHM
      return hipGetLastError();
    } else if constexpr (MODE == GM_F32) {
      // exact fp32: round 1-3's 32-row kernel, 4 waves (the re-run route of a masked batch)
      if (!p.mask[0] || !p.mask[1] || p.policy != 0 || p.attn_full) return hipErrorInvalidValue;
#define OETR_LAUNCHMF(B, T) \
  hipLaunchKernelGGL((k_encoder<B, T, MODE, OETR_F32_WAVES, false, 0, true>), grid, dim3(64 * OETR_F32_WAVES), 0, s, p)
      if (has_b) {
        if (tail == 0) OETR_LAUNCHMF(true, 0);
        else if (tail == 1) OETR_LAUNCHMF(true, 1);
        else OETR_LAUNCHMF(true, 2);
      } else {
        if (tail == 0) OETR_LAUNCHMF(false, 0);
        else return hipErrorInvalidValue;
      }
#undef OETR_LAUNCHMF
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if constexpr (gm_half(MODE)) {
    if (p.tile_rows == RT) {
#define OETR_LAUNCH64(B, T) hipLaunchKernelGGL((k_encoder64<B, T, MODE>), grid, dim3(512), 0, s, p)
      if constexpr (MODE == GM_SPLIT) {   // precision policies exist for the two-plane mode
        if (p.policy == 1) {
#define OETR_LAUNCH64P(B, T) hipLaunchKernelGGL((k_encoder64<B, T, MODE, 1>), grid, dim3(512), 0, s, p)
          if (has_b) {
            if (tail == 0) OETR_LAUNCH64P(true, 0);
            else if (tail == 1) OETR_LAUNCH64P(true, 1);
            else OETR_LAUNCH64P(true, 2);
          } else {
            if (tail == 0) OETR_LAUNCH64P(false, 0);
            else if (tail == 1) OETR_LAUNCH64P(false, 1);
            else return hipErrorInvalidValue;
          }
#undef OETR_LAUNCH64P
          return hipGetLastError();
        }
      }
      if (p.policy != 0) return hipErrorInvalidValue;
      if (has_b) {
        if (tail == 0) OETR_LAUNCH64(true, 0);
        else if (tail == 1) OETR_LAUNCH64(true, 1);
        else OETR_LAUNCH64(true, 2);
      } else {
        if (tail == 0) OETR_LAUNCH64(false, 0);
        else if (tail == 1) OETR_LAUNCH64(false, 1);
        else return hipErrorInvalidValue;
      }
#undef OETR_LAUNCH64
      return hipGetLastError();
    }
  }
  if constexpr (MODE == GM_SPLIT) {
    if (!p.attn_full) {   // 32 token rows, two-plane mode, linear attention: the 64-row kernel's body on one row tile
      if (p.policy != 0 && p.policy != 1) return hipErrorInvalidValue;
#define OETR_LAUNCH32M(B, T)                                                                    \
  do {                                                                                          \
    if (p.policy == 1) hipLaunchKernelGGL((k_encoder32m<B, T, MODE, 1>), grid, dim3(512), 0, s, p); \
    else hipLaunchKernelGGL((k_encoder32m<B, T, MODE, 0>), grid, dim3(512), 0, s, p);            \
  } while (0)
      if (has_b) {
        if (tail == 0) OETR_LAUNCH32M(true, 0);
        else if (tail == 1) OETR_LAUNCH32M(true, 1);
        else OETR_LAUNCH32M(true, 2);
      } else {
        if (tail == 0) OETR_LAUNCH32M(false, 0);
        else if (tail == 1) OETR_LAUNCH32M(false, 1);
        else return hipErrorInvalidValue;
      }
#undef OETR_LAUNCH32M
      return hipGetLastError();
    }
  }
  // round 1-3's 32-row kernel: exact fp32 (4 waves), the single-plane modes, attention = 'full'
  constexpr int NW = gm_half(MODE) ? OETR_SPLIT_WAVES : OETR_F32_WAVES;
  if (p.policy != 0) return hipErrorInvalidValue;   // (policies exist in the two-plane mode only: handled above)
  if (p.attn_full) {
    // (the exact-fp32 build runs 8 waves here too: one head per wave in the attention core)
    if constexpr ((gm_f16_range(MODE) && NW == 8) || MODE == GM_F32) {
#define OETR_LAUNCHF(B, T) hipLaunchKernelGGL((k_encoder<B, T, MODE, 8, true>), grid, dim3(64 * 8), 0, s, p)
      if (has_b) {
        if (tail == 0) OETR_LAUNCHF(true, 0);
        else if (tail == 1) OETR_LAUNCHF(true, 1);
        else OETR_LAUNCHF(true, 2);
      } else {
        if (tail == 0) OETR_LAUNCHF(false, 0);
        else return hipErrorInvalidValue;
      }
#undef OETR_LAUNCHF
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if constexpr (MODE != GM_SPLIT) {
#define OETR_LAUNCH(B, T) hipLaunchKernelGGL((k_encoder<B, T, MODE, NW>), grid, dim3(64 * NW), 0, s, p)
    if (has_b) {
      if (tail == 0) OETR_LAUNCH(true, 0);
      else if (tail == 1) OETR_LAUNCH(true, 1);
      else OETR_LAUNCH(true, 2);
    } else {
      if (tail == 0) OETR_LAUNCH(false, 0);
      else if (tail == 1) OETR_LAUNCH(false, 1);
      else OETR_LAUNCH(false, 2);
    }
#undef OETR_LAUNCH
    return hipGetLastError();
  }
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// Pre-reduction of the partial linear-attention states, once per image and head, between the
// launch that wrote them (phase A) and the launch that consumes them (phase B): otherwise
// EVERY workgroup of an image re-reads and re-sums all of the source image's partials
// (7 x 33 KB per 64-token workgroup at 400 tokens, 13 x 33 KB per 32-token one - the largest
// item of its prologue).  Same summation order as the in-kernel loop (tile 0, 1, ...):
// bit-identical results.  One wave per (image, head).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kv_reduce(Geom g, const float* __restrict__ kvp,
                                                   const float* __restrict__ ksp, float* __restrict__ kvr,
                                                   float* __restrict__ ksr) {
  // one workgroup per (image, head), wave e sums the e-th 1-KB quarter of the head's 4-KB state
  // over the image's tiles (round 4: four waves instead of one per state - 9.5 -> see profiles)
  const int img = blockIdx.x >> 3, head = blockIdx.x & 7, lane = threadIdx.x & 63;
  const int e = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int side = img >= g.N, n = img - side * g.N;
  const int nt = g.nt[side];
  const int slot0 = g.tile0[side] + n * nt;
  const f32x4* src = reinterpret_cast<const f32x4*>(kvp) + ((size_t)slot0 * NH + head) * 256 + e * 64 + lane;
  const float* ks = ksp + (size_t)slot0 * C + head * HD + (lane & 31);
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  float kacc = 0.f;
  constexpr int U = 16;   // tiles in flight per round trip
  for (int t0 = 0; t0 < nt; t0 += U) {
    f32x4 tmp[U];
    float kt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = min(t0 + u, nt - 1);
      tmp[u] = src[(size_t)t * (NH * 256)];
      kt[u] = ks[(size_t)t * C];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (t0 + u < nt) {
        acc += tmp[u];
        kacc += kt[u];
      }
  }
  reinterpret_cast<f32x4*>(kvr)[((size_t)img * NH + head) * 256 + e * 64 + lane] = acc;
  if (e == 0 && lane < 32) ksr[(size_t)img * C + head * HD + lane] = kacc;
}

hipError_t launch_kv_reduce(const Geom& g, const float* kvp, const float* ksp, float* kvr, float* ksr,
                            hipStream_t s) {
  hipLaunchKernelGGL(k_kv_reduce, dim3(2 * g.N * NH), dim3(256), 0, s, g, kvp, ksp, kvr, ksr);
  return hipGetLastError();
}

hipError_t launch_encoder(const EncLaunch& p, bool has_b, int tail, int mode, hipStream_t s) {
  switch (mode) {
    case GM_F32: return launch_encoder_mode<GM_F32>(p, has_b, tail, s);
    case GM_SPLIT: return launch_encoder_mode<GM_SPLIT>(p, has_b, tail, s);
    case GM_F16: return launch_encoder_mode<GM_F16>(p, has_b, tail, s);
    case GM_BF16: return launch_encoder_mode<GM_BF16>(p, has_b, tail, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace oetr
