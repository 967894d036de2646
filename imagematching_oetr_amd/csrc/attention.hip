// Stand-alone attention cores (gfx950), exported for parity tests against the
// reference classes and for the optional "all-pairs" variant.
//
//  linear: LinearAttention.forward (src/models/linear_attention.py:22-50).
//          Inside the model this math is fused into encoder.hip; here it is a
//          two-launch form (state reduce over S, then apply over L).
//  full  : FullAttention.forward (src/models/linear_attention.py:53-87):
//          softmax(QK^T/sqrt(D))V, flash-style - the L x S score volume lives
//          only in MFMA accumulators, 32 queries x 32 keys at a time.
//          S^T = K Q^T is computed "swapped" so every lane owns one query
//          column: row max / row sum are in-lane over 16 registers + one
//          cross-half shuffle, and exp(S^T) feeds the P.V MFMA as the A
//          operand straight from the accumulator registers (no LDS at all).
#include "common.h"

namespace oetr {

// ------------------------------------------------------------------ linear
// one block per (n, h): KV[d][v] = sum_s phi(K[s,d]) * V[s,v]/S ; Ksum[d]
// kv_mask [n][S] / q_mask [n][L] (linear_attention.py:37-41) or NULL: a token's value multiplies its
// phi(K) and V rows / its phi(Q) row (x 1.0f without a mask: the same bits as before)
__global__ __launch_bounds__(256) void k_lin_state(const float* __restrict__ k,
                                                   const float* __restrict__ v,
                                                   const float* __restrict__ kv_mask, int S,
                                                   float* __restrict__ state) {
  __shared__ float ks[64][HD + 1], vs[64][HD + 1];
  const int tid = threadIdx.x, nh = blockIdx.x, n = nh / NH, h = nh % NH;
  const int d = tid >> 3, v0 = (tid & 7) * 4;
  float kv[4] = {0.f, 0.f, 0.f, 0.f}, ksum = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    for (int i = tid; i < 64 * HD; i += 256) {
      const int sl = i >> 5, c = i & 31, s = s0 + sl;
      const size_t off = (((size_t)n * S + s) * NH + h) * HD + c;
      const float mk = (kv_mask != nullptr && s < S) ? kv_mask[(size_t)n * S + s] : 1.0f;
      ks[sl][c] = s < S ? elu1(k[off]) * mk : 0.f;
      vs[sl][c] = s < S ? (v[off] * mk) / (float)S : 0.f;
    }
    __syncthreads();
    for (int sl = 0; sl < 64; ++sl) {
      const float kk = ks[sl][d];
      ksum += kk;
#pragma unroll
      for (int j = 0; j < 4; ++j) kv[j] += kk * vs[sl][v0 + j];
    }
    __syncthreads();
  }
  float* st = state + (size_t)nh * (HD * HD + HD);
#pragma unroll
  for (int j = 0; j < 4; ++j) st[d * HD + v0 + j] = kv[j];
  if ((tid & 7) == 0) st[HD * HD + d] = ksum;
}

// one block per (n, h, 64 queries)
__global__ __launch_bounds__(256) void k_lin_apply(const float* __restrict__ q,
                                                   const float* __restrict__ state,
                                                   const float* __restrict__ q_mask, int L, int S,
                                                   float* __restrict__ out) {
  __shared__ float kv[HD][HD + 1], ksum[HD];
  const int tid = threadIdx.x, nh = blockIdx.y, n = nh / NH, h = nh % NH;
  const float* st = state + (size_t)nh * (HD * HD + HD);
  for (int i = tid; i < HD * HD; i += 256) kv[i >> 5][i & 31] = st[i];
  if (tid < HD) ksum[tid] = st[HD * HD + tid];
  __syncthreads();
  const int l = blockIdx.x * 64 + (tid >> 2), v0 = (tid & 3) * 8;
  if (l >= L) return;
  const size_t off = (((size_t)n * L + l) * NH + h) * HD;
  float fq[HD], z = 0.f;
  const float mq = q_mask != nullptr ? q_mask[(size_t)n * L + l] : 1.0f;
#pragma unroll
  for (int dd = 0; dd < HD; ++dd) { fq[dd] = elu1(q[off + dd]) * mq; z += fq[dd] * ksum[dd]; }
  z = 1.0f / (z + ATTN_EPS);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float s = 0.f;
#pragma unroll
    for (int dd = 0; dd < HD; ++dd) s += fq[dd] * kv[dd][v0 + j];
    out[off + v0 + j] = s * z * (float)S;
  }
}

// `state`: caller-provided n*8 states of 1056 floats between the two kernels
hipError_t launch_linear_attention(const float* q, const float* k, const float* v, const float* q_mask,
                                   const float* kv_mask, int n, int L, int S, float* out, float* state,
                                   hipStream_t s) {
  hipLaunchKernelGGL(k_lin_state, dim3(n * NH), dim3(256), 0, s, k, v, kv_mask, S, state);
  hipLaunchKernelGGL(k_lin_apply, dim3((L + 63) / 64, n * NH), dim3(256), 0, s, q, state, q_mask, L, S, out);
  return hipGetLastError();
}

// -------------------------------------------------------------------- full
// one wave per (n, h, 32 queries); 4 waves per block.
__global__ __launch_bounds__(256) void k_full_attention(const float* __restrict__ q,
                                                        const float* __restrict__ k,
                                                        const float* __restrict__ v, int n_img,
                                                        int L, int S, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5;
  const int col = lane & 31;
  const int qtiles = (L + 31) / 32;
  const long wid = (long)blockIdx.x * 4 + wave;
  if (wid >= (long)n_img * NH * qtiles) return;
  const int qt = (int)(wid % qtiles);
  const int h = (int)((wid / qtiles) % NH);
  const int n = (int)(wid / ((long)qtiles * NH));
  const int q0 = qt * 32;
  const float temp = 1.0f / sqrtf((float)HD);

  // Q fragment (B operand of S^T = K Q^T): Q[query = col][d = 8ks + 4half + j]
  f32x4 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q0 + col < L)
      qf[ks] = *reinterpret_cast<const f32x4*>(
          q + (((size_t)n * L + q0 + col) * NH + h) * HD + 8 * ks + 4 * half);
  }
  f32x16 o = {0};
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < S; k0 += 32) {
    // S^T tile: rows = keys, cols = queries
    f32x16 st = {0};
    const bool krow_ok = k0 + col < S;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f32x4 kf = {0.f, 0.f, 0.f, 0.f};
      if (krow_ok)
        kf = *reinterpret_cast<const f32x4*>(
            k + (((size_t)n * S + k0 + col) * NH + h) * HD + 8 * ks + 4 * half);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[ks][j], st, 0, 0, 0);
    }
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = (k0 + crow(r, half) < S) ? st[r] * temp : -INFINITY;
      mt = fmaxf(mt, st[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = expf(m_run - m_new);  // 0 on the first tile
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = expf(st[r] - m_new); ps += st[r]; }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    // rescale O rows (row = query crow(r, half)) by that query's alpha
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= __shfl(alpha, crow(r, half), 64);
    // O += P V : A = P[query = col][key = crow(r, half)], B = V[key][d = col]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + crow(r, half);
      const float vv = key < S ? v[(((size_t)n * S + key) * NH + h) * HD + col] : 0.f;
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[r], vv, o, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qi = q0 + crow(r, half);
    const float lq = __shfl(l_run, crow(r, half), 64);
    if (qi < L) out[(((size_t)n * L + qi) * NH + h) * HD + col] = o[r] / lq;
  }
}

hipError_t launch_full_attention(const float* q, const float* k, const float* v, int n, int L,
                                 int S, float* out, hipStream_t s) {
  const long waves = (long)n * NH * ((L + 31) / 32);
  hipLaunchKernelGGL(k_full_attention, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, q, k,
                     v, n, L, S, out);
  return hipGetLastError();
}

// -------------------------------------------------------------- full, split f16
// The same math on the f16 matrix pipe with the fp32-class operand split of common.h
// (a = hi + lo, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation): 13 MFMAs
// of 32 cycles per 32-key x 32-query tile (6 + 1 for the scores, 6 for P.V) instead of 32
// f32 MFMAs of 64 cycles.
//
// One block = (image, head, 256 queries), 8 waves x 32 queries.  Per 64-key tile the block
// converts K and V once into split planes in LDS (K row-major, V transposed with the key
// order the MFMA k-slots want; a ring of three buffers: tile t + 2 is loaded under step t's math
// and written behind it, ONE barrier per tile), then every wave computes, with no lane exchange
// beyond one cross-half maximum per tile:
//   S^T = K . Q^T      A = K fragments (LDS),  B = Q (registers, split once)
//   P   = 2^(S^T - m)  in the lane that owns the query column
//   O^T += V^T . P^T   A = V^T fragments (LDS), B = P straight from the S^T accumulator
//                      registers (k-slot 8*half + i of step s <-> key crow(8s+i, half))
// O^T keeps the query in the lane, so the online-softmax rescale and the final 1/l are
// per-lane scalars.  The L x S score volume exists only in accumulators.
//
// Round 6 (0.19 of the split roof at L = S = 4096; the kernel was VALU-bound,
// ~20 VALU issue slots per score element beside 0.75 MFMA): what a score element costs now -
//   * 1/sqrt(D) and log2(e) are folded into Q before its split, scores arrive in log2 units: P = v_exp_f32(x)
//     (was: multiply, select, the 6-instruction compensated exp and two clamps);
//   * keys past S are masked in the LAST tile only (a compile-time variant of the step);
//   * the reference maximum m follows the true maximum lazily: O, its cross accumulator and l are rescaled only in
//     tiles where some lane's maximum moved (wave-uniform ballot; after the first few tiles it rarely does);
//   * the cross accumulator of P.V (V's lo plane) runs across tiles and is folded into O once at the end;
//   * the row sum l stays a per-lane partial (its two half-lanes are added once at the end).
// Later the same round (`scores`, `m_run`, `probs` below; steps and numbers: profiles/r6_full_attention_steps.txt):
//   * the score side needs NO VALU instruction per element any more: K's and Q's lo planes are UNSCALED (split2u), so
//     all three products of the split go into ONE accumulator (no cross accumulator, no fold), and -m enters as one
//     more MFMA (A = ones in two k-slots, B = -m as an f16 pair) instead of a per-tile fill of the accumulators;
//   * P carries a factor 2^FA_SH (the final 1/l removes it) and its lo plane is unscaled too: no multiply in its split.
//   6.5 -> 4.5 VALU instructions per score element (exp, row sum, half a max3, convert, residual, convert).
// P is still split into two planes like every other operand: a single f16 plane for P (VERDICT r5's suggestion) was
// measured first - 3e-5 .. 8e-5 against the fp64 oracle where the goldens allow 5e-6, for 13 % of launch time
// (profiles/r6_full_attention_p1plane.txt; the -DOETR_FA_P1PLANE build keeps the experiment reproducible).
constexpr int FA_KT = 64;            // keys per tile
constexpr int FA_KP = 40;            // halves per K row (32 d + 8): conflict-free ds_read_b128
constexpr int FA_VP = FA_KT + 8;     // halves per V^T row (64 key slots + 8): 9 x 16 B, conflict-free
#ifndef OETR_FA_WAVES
#define OETR_FA_WAVES 8
#endif
constexpr int FA_WAVES = OETR_FA_WAVES;   // waves per workgroup, 32 queries each
struct FaTile {
  _Float16 Kh[FA_KT * FA_KP], Kl[FA_KT * FA_KP];   // K: hi, UNSCALED lo (split2u)
  _Float16 Vh[HD * FA_VP], Vl[HD * FA_VP];   // [d][key slot]
};

__global__ __launch_bounds__(64 * FA_WAVES) void k_full_attention_split(const float* __restrict__ q,
                                                                        const float* __restrict__ k,
                                                                        const float* __restrict__ v,
                                                                        int L, int S, float* __restrict__ out,
                                                                        uint32_t* flags) {
  __shared__ __attribute__((aligned(16))) FaTile ring[3];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, col = lane & 31;
  constexpr int QB = 32 * FA_WAVES;
  const int qchunks = (L + QB - 1) / QB;
  const int qc = blockIdx.x % qchunks, nh = blockIdx.x / qchunks, n = nh / NH, h = nh % NH;
  const int q0 = qc * QB + wave * 32;
  const bool active = q0 < L;      // (wave-uniform: a wave past the image end only helps staging)
  Range rg;
  float neg1 = -1.0f;               // (opaque to the compiler: see split2u)
  asm volatile("" : "+s"(neg1));

  // Q as the B operand of S^T = K Q^T: lane (query = col) holds d = 16s + 8*half + 0..7, already
  // multiplied by 1/sqrt(D) * log2(e) - the scores come out of the MFMAs in log2 units
  const float qs = 0.17677669529663687f * 1.4426950408889634f;
  f32x4 qh[2], ql[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    if (q0 + col < L) {
      const float* qp = q + (((size_t)n * L + q0 + col) * NH + h) * HD + 16 * s + 8 * half;
      a0 = *reinterpret_cast<const f32x4*>(qp) * qs;
      a1 = *reinterpret_cast<const f32x4*>(qp + 4) * qs;
    }
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;   // (unscaled lo plane, like K's: `scores`)
    split2u(a0[0], a0[1], neg1, h0, l0, rg);
    split2u(a0[2], a0[3], neg1, h1, l1, rg);
    split2u(a1[0], a1[1], neg1, h2, l2, rg);
    split2u(a1[2], a1[3], neg1, h3, l3, rg);
    qh[s] = __builtin_bit_cast(f32x4, u32x4{h0, h1, h2, h3});
    ql[s] = __builtin_bit_cast(f32x4, u32x4{l0, l1, l2, l3});
  }
  f32x16 o = {0}, oc = {0};   // O^T and the part that meets V's (2^11-scaled) lo plane: rows = d (crow(r, half)), col = query
  // (masked scores and the initial maximum are a large FINITE negative: no inf - inf anywhere)
  constexpr float NEG = -1.0e30f;
  // Everything behind the exponential carries a factor 2^FA_SH (P' = 2^FA_SH . P; O, its cross accumulator and l with
  // it - the final 1/l removes it exactly like 2^-m): P's lo plane is then f16(P' - hi) with NO 2^11 scaling, its product
  // with V_hi has the scale of hi . V_hi and goes into the SAME accumulator - one VALU multiply per score element fewer.
  // FA_SH = 4: the lo plane of a weight P >= 2^-6 is a normal f16 number, below that a denormal with absolute error
  // 2^-25 = 2^-29 of the row's largest weight (at S = 4096 equal weights: 1e-7 of the output after averaging).  The
  // factor costs accuracy where the scores are tiny: x = s - m + FA_SH is formed at magnitude FA_SH (2^-23 absolute at
  // 4; with 11 - the lo plane normal down to 2^-13 - the S = 2 fuzz cases sat at 4-5x torch fp32's own drift).
  // m_run = (reference maximum) - FA_SH.  It enters the scores THROUGH THE MATRIX PIPE: one more MFMA per 32-key half
  // whose A fragment is 1 in k-slots 0 and 1 (every key row) and whose B fragment holds, in the lane's query column,
  // -m_run as two f16 values (hi + lo) in those slots - the product is -m_run in every row of the column.  m_run is
  // kept to 21 significant bits (and a multiple of 2^-24), so hi + lo IS m_run: whatever its value, it is the SAME
  // number in every tile between two adjustments - the common factor the final 1/l removes - and the rescale factor of
  // an adjustment is computed from the same two numbers.  The reference only has to be NEAR the maximum; this near
  // (x_max = FA_SH to 2^-21 |m|) the row's largest weight is P' = 2^FA_SH (1 + eps), which the two planes hold almost
  // exactly - an integer-valued m_run (exact power-of-two rescales) left P'_max anywhere within a factor sqrt(2), with the
  // generic 2^-23 representation error that l, summed from the unsplit P', does not share: 1.8e-7 on a ONE-key row.
  // |m_run| beyond the f16 range (scores beyond 6e4 in log2 units) is reported through the range guard like any other
  // operand.  2 MFMAs per tile on a pipe that is 40 % busy, against a per-element subtraction (or, before, the fold
  // and the per-tile fill of the accumulators with -m).
  // (Tried on the way: a 16-register tuple holding -m_run as the C operand of every chain's first MFMA - hipcc copies
  //  the tuple at the join behind the lazy branch, more moves than the fill; and see `scores` for why the chain cannot
  //  simply START at -m_run.)
  float m_run = -FA_SH, l_run = 0.f;
  const uint32_t onebits = lane < 32 ? 0x3c003c00u : 0u;           // (1.0h, 1.0h) in k-slots 0, 1 of the half-0 lanes
  const f32x4 afrag = __builtin_bit_cast(f32x4, u32x4{onebits, 0u, 0u, 0u});
  f32x4 mfrag = __builtin_bit_cast(f32x4, u32x4{lane < 32 ? 0x00004400u : 0u, 0u, 0u, 0u});   // (4.0h, 0): -m_run = FA_SH
  static_assert(FA_SH == 4.0f, "initial mfrag encodes 4.0 as f16 0x4400");

  // staging roles.  K: thread -> (key row = tid >> 3, 4 consecutive d = 4 * (tid & 7)), one 16-byte load.
  // V: thread -> (d = tid & 31, keys 4g .. 4g + 3, g = tid >> 5): four 4-byte loads (a half-wave reads the 128
  // contiguous bytes of one key), ONE 8-byte store per plane - the four keys are consecutive k-slots of the
  // P.V contraction: key 16u + 8a + 4b + c sits in slot 16u + 8b + 4a + c (bits 2 and 3 of the key swapped)
  // (512 such items of each kind per tile: FA_ITEMS per thread)
  constexpr int FA_ITEMS = 512 / (64 * FA_WAVES);
  f32x4 kreg[FA_ITEMS], vreg[FA_ITEMS];
  auto load_tile = [&](int k0) {
    const size_t base = ((size_t)n * S) * NH * HD + h * HD;
#pragma unroll
    for (int it = 0; it < FA_ITEMS; ++it) {
      const int item = tid + 64 * FA_WAVES * it;
      const int kr = item >> 3, kd = 4 * (item & 7), vd = item & 31, vg = item >> 5;
      kreg[it] = f32x4{0.f, 0.f, 0.f, 0.f};
      vreg[it] = kreg[it];
      if (k0 + kr < S) kreg[it] = *reinterpret_cast<const f32x4*>(k + base + (size_t)(k0 + kr) * (NH * HD) + kd);
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (k0 + 4 * vg + b < S) vreg[it][b] = v[base + (size_t)(k0 + 4 * vg + b) * (NH * HD) + vd];
    }
  };
  auto write_tile = [&](FaTile& B) {
#pragma unroll
    for (int it = 0; it < FA_ITEMS; ++it) {
      const int item = tid + 64 * FA_WAVES * it;
      const int kr = item >> 3, kd = 4 * (item & 7), vd = item & 31, vg = item >> 5;
      const int vslot = 16 * (vg >> 2) + 8 * (vg & 1) + 4 * ((vg >> 1) & 1);
      uint32_t h0, l0, h1, l1;
      split2u(kreg[it][0], kreg[it][1], neg1, h0, l0, rg);
      split2u(kreg[it][2], kreg[it][3], neg1, h1, l1, rg);
      *reinterpret_cast<u32x2*>(B.Kh + kr * FA_KP + kd) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(B.Kl + kr * FA_KP + kd) = u32x2{l0, l1};
      cvt_planes2<GM_SPLIT>(vreg[it][0], vreg[it][1], h0, l0, rg);
      cvt_planes2<GM_SPLIT>(vreg[it][2], vreg[it][3], h1, l1, rg);
      *reinterpret_cast<u32x2*>(B.Vh + vd * FA_VP + vslot) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(B.Vl + vd * FA_VP + vslot) = u32x2{l0, l1};
    }
  };
  // ---- the per-tile pieces -------------------------------------------------------------------------
  // S^T tile of the keys in B: rows = keys (two 32-key halves), cols = queries; 14 MFMAs, operands from LDS and
  // registers that no VALU result of the step feeds - free to run beside the softmax of the tile before.
  // ONE accumulator per 32-key half; the three products of the split all have the scale of K_hi . Q_hi:
  //   K_hi . Q_lo  +  K_lo . Q_hi  +  (-m_run)  +  K_hi . Q_hi          (both lo planes unscaled: split2u)
  // - no cross accumulator (32 registers) and no fold.  ORDER MATTERS: the chain starts from zero with the SMALL
  // products (2^-12 of the scores) and takes the big ones last.  The MFMA adds its 16 products and C in one aligned
  // sum, and small addends beside a large one lose their low bits there: with the accumulator started at -m and the big
  // products first, the error against the fp64 oracle was 2-3x the two-accumulator kernel's (6.8x torch fp32's own
  // drift on the worst fuzz case, 4.76e-6 on the sharpened probe where the goldens allow 5e-6).  With the small terms
  // summed among themselves first, their SUM is cut once when the first big step comes in - the fold's single rounding.
  // (Measured and dropped: Q's lo plane kept 2^11-scaled against a third K plane K_hi x 2^-11 - Q' = Q x 0.255 is
  //  small, its unscaled lo plane mostly denormal - bought nothing on the probes or the fuzz, for 3 % of launch time.)
  // The two 32-key halves are independent chains; their MFMAs alternate.
  auto scores = [&](const FaTile& B, f32x16 (&sa)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) sa[j] = f32x16{0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 al[2], ah[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        al[j] = *reinterpret_cast<const f32x4*>(B.Kl + (32 * j + col) * FA_KP + 16 * s + 8 * half);
        ah[j] = *reinterpret_cast<const f32x4*>(B.Kh + (32 * j + col) * FA_KP + 16 * s + 8 * half);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) sa[j] = mma16<GM_SPLIT>(ah[j], ql[s], sa[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) sa[j] = mma16<GM_SPLIT>(al[j], qh[s], sa[j]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 ah[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        ah[j] = *reinterpret_cast<const f32x4*>(B.Kh + (32 * j + col) * FA_KP + 16 * s + 8 * half);
#pragma unroll
      for (int j = 0; j < 2; ++j) sa[j] = mma16<GM_SPLIT>(ah[j], qh[s], sa[j]);
      if (s == 0) {   // - m_run BETWEEN the two big steps: every intermediate sum is about half the size of m
#pragma unroll
        for (int j = 0; j < 2; ++j) sa[j] = mma16<GM_SPLIT>(afrag, mfrag, sa[j]);
      }
    }
  };
  // x = s - m, m = the lane's query's reference maximum.  m follows the true maximum LAZILY: it is
  // raised - and O, its cross accumulator, l and this tile's x rescaled - only when some lane's tile maximum
  // exceeds the reference by more than 2^FA_LAZY (wave-uniform ballot; rare after the first tile, which always
  // sets the reference).  In between P = 2^x may exceed 1 - by at most 2^FA_LAZY = 256: P' = 2^FA_SH . P <= 2^12 stays
  // far inside the f16 range of its hi plane; the final 1/l removes the common factor exactly as it removes 2^-m.
  constexpr float FA_LAZY = 8.0f;
  auto maximum = [&](f32x16 (&sa)[2], int k0, auto tail_c, bool first) {
    constexpr bool tail = decltype(tail_c)::value;   // (compile time: as a run-time flag hipcc if-converted the 96 compare / select / add
                                                     //  instructions of the mask into EVERY tile's step - a third of its VALU work)
    if constexpr (tail) {   // keys past S: only the last tile has any
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + 32 * j + crow(r, half) >= S) sa[j][r] = NEG;
    }
    float mt = NEG;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mt = __builtin_fmaxf(__builtin_fmaxf(mt, sa[j][r]), sa[j][r + 1]);   // v_max3_f32
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    if (first || __builtin_amdgcn_ballot_w64(mt > FA_SH + FA_LAZY) != 0) {
      // the new reference: the lane's maximum (first tile: nothing accumulated yet, any direction; later: raised only),
      // rounded to 21 significant bits and to a multiple of 2^-24 - exactly the sum of two f16 values (see m_run)
      const float nm = -(m_run + (first ? mt - FA_SH : fmaxf(mt - FA_SH, 0.f)));
      float nq = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, nm) + 4u) & ~7u);
      if (__builtin_fabsf(nm) < 0.0625f) nq = __builtin_rintf(nm * 16777216.0f) * (1.0f / 16777216.0f);
      const float adj = -nq - m_run;   // (0 in a lane whose reference stays: its -m_run is already of that form)
      if (!first) {
        const float alpha = __builtin_amdgcn_exp2f(-adj);
        l_run *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[r] *= alpha; oc[r] *= alpha; }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[j][r] -= adj;
      m_run = -nq;
      const _Float16 mh = (_Float16)nq, ml = (_Float16)(nq - (float)mh);
      const uint32_t mb = (uint32_t)__builtin_bit_cast(unsigned short, mh) | ((uint32_t)__builtin_bit_cast(unsigned short, ml) << 16);
      mfrag[0] = __builtin_bit_cast(float, lane < 32 ? mb : 0u);
      rg.see2(nq, 0.f);
    }
  };
  // P' = 2^x in place, its row-sum share, and its split planes: B operands of the P.V steps.
  // hi = RNE f16(P'), lo = RNE f16(P' - hi): the difference is exact (v_fma_mix_f32 reads the f16 half directly),
  // UNSCALED - see the note at m_run; four VALU instructions per pair.  Scalar f32 operations only: packed f32 VALU
  // (v_pk_mul_f32 / v_pk_fma_f32, which hipcc's SLP pass forms from adjacent scalar ops - this file is built with
  // -fno-slp-vectorize) costs ~22 cycles per instruction beside MFMAs on gfx950 (MI355X_MICROARCH.md), and here every
  // one of them sits beside MFMAs.  (P' <= 2^(FA_SH + FA_LAZY) = 2^12: nothing for the range guard.)
  auto probs = [&](f32x16 (&sa)[2], f32x4 (&ph)[4], f32x4 (&pl)[4]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sa[j][r] = __builtin_amdgcn_exp2f(sa[j][r]);       // (x = s - m already: `maximum`)
        l_run += sa[j][r];
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = sa[j][8 * s + 2 * i], b = sa[j][8 * s + 2 * i + 1];
          const f16x2 h = __builtin_convertvector(f32x2{a, b}, f16x2);
          hi[i] = __builtin_bit_cast(uint32_t, h);
          lo[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{__builtin_fmaf((float)h[0], neg1, a),
                                                                              __builtin_fmaf((float)h[1], neg1, b)}, f16x2));
        }
        ph[2 * j + s] = __builtin_bit_cast(f32x4, u32x4{hi[0], hi[1], hi[2], hi[3]});
        pl[2 * j + s] = __builtin_bit_cast(f32x4, u32x4{lo[0], lo[1], lo[2], lo[3]});
      }
  };
  // O^T += V^T P^T
  auto apply = [&](const FaTile& B, const f32x4 (&ph)[4], const f32x4 (&pl)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 ah = *reinterpret_cast<const f32x4*>(B.Vh + col * FA_VP + 16 * ks + 8 * half);
      const f32x4 al = *reinterpret_cast<const f32x4*>(B.Vl + col * FA_VP + 16 * ks + 8 * half);
#ifdef OETR_FA_P1PLANE   // experiment (profiles/r6_full_attention_p1plane.txt): P as ONE f16 plane, V split
      o = mma16<GM_SPLIT>(ah, ph[ks], o);
      oc = mma16<GM_SPLIT>(al, ph[ks], oc);
#else
      o = mma16<GM_SPLIT>(ah, ph[ks], o);     // V_hi . P'_hi
      oc = mma16<GM_SPLIT>(al, ph[ks], oc);   // V_lo (x 2^11) . P'_hi
      o = mma16<GM_SPLIT>(ah, pl[ks], o);     // V_hi . P'_lo (unscaled lo plane: the same scale)
#endif
    }
  };

  // ---- software pipeline over the key tiles -----------------------------------------------------------
  // Step t: [S^T of tile t + 1: 12 MFMAs] beside [P of tile t: exp / sum / split - VALU], then [P.V of tile t:
  // 12 MFMAs] beside [cross fold and maximum of tile t + 1 - VALU]: the matrix pipe and the VALU of ONE wave
  // overlap (before: every wave of the workgroup ran the same phase at the same time behind the tile's
  // barrier, and the two waves of a SIMD queued for the same pipe).  Three LDS buffers: step t reads K of
  // tile t + 1 and V of tile t while tile t + 2 is written; one barrier per step.  (The same pipeline at
  // 32-key granularity - half the live accumulators - measured 8 % slower: 561 vs 518 us at L = S = 4096;
  // forced to 128 VGPRs for two workgroups per CU it spills 205 registers; the UNPIPELINED 32-key loop under that
  // cap - four waves per SIMD as each other's cover - spills 37 and takes 886 us: profiles/r6_full_attention_p1plane.txt.)
  const int T = (S + FA_KT - 1) / FA_KT;
  const bool ragged = (S % FA_KT) != 0;
  load_tile(0);
  write_tile(ring[0]);
  if (T > 1) {
    load_tile(FA_KT);
    write_tile(ring[1]);
  }
  __syncthreads();
  f32x16 sA[2] = {}, sB[2] = {};
  if (active) {
    scores(ring[0], sA);
    if (T == 1 && ragged) maximum(sA, 0, std::true_type{}, true);
    else maximum(sA, 0, std::false_type{}, true);
  }
  auto step = [&](int t, f32x16 (&cur)[2], f32x16 (&nxt)[2], auto tail_c) {   // t + 1 < T; tail_c: tile t + 1 is the ragged last one
    if (t + 2 < T) load_tile((t + 2) * FA_KT);
    if (active) {
      const FaTile& BK = ring[(t + 1) % 3];
      const FaTile& BV = ring[t % 3];
      f32x4 ph[4], pl[4];
      scores(BK, nxt);
      probs(cur, ph, pl);
      apply(BV, ph, pl);
      maximum(nxt, (t + 1) * FA_KT, tail_c, false);
    }
    if (t + 2 < T) write_tile(ring[(t + 2) % 3]);
    __syncthreads();
  };
  auto last = [&](int t, f32x16 (&cur)[2]) {
    if (active) {
      f32x4 ph[4], pl[4];
      probs(cur, ph, pl);
      apply(ring[t % 3], ph, pl);
    }
  };
  {
    int t = 0;
    for (; t + 3 < T; t += 2) {       // (neither step reaches the last tile)
      step(t, sA, sB, std::false_type{});
      step(t + 1, sB, sA, std::false_type{});
    }
    // what is left: one step (t + 2 == T), two (t + 3 == T), or none (t + 1 == T); the final step meets the last tile
    auto final_step = [&](int tt, f32x16 (&cur)[2], f32x16 (&nxt)[2]) {
      if (ragged) step(tt, cur, nxt, std::true_type{});
      else step(tt, cur, nxt, std::false_type{});
    };
    if (t + 3 == T) {
      step(t, sA, sB, std::false_type{});
      final_step(t + 1, sB, sA);
      last(t + 2, sA);
    } else if (t + 2 == T) {
      final_step(t, sA, sB);
      last(t + 1, sB);
    } else {
      last(t, sA);
    }
  }
  if (q0 + col < L) {
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    float* dst = out + (((size_t)n * L + q0 + col) * NH + h) * HD + 4 * half;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {   // registers 4*g4 .. 4*g4+3 = d rows 8*g4 + 4*half + 0..3
      f32x4 y;
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = fmaf(oc[4 * g4 + i], SPLIT_INV, o[4 * g4 + i]) * inv;
      *reinterpret_cast<f32x4*>(dst + 8 * g4) = y;
    }
  }
  if (flags) range_report<GM_SPLIT>(rg, flags);
}

hipError_t launch_full_attention_split(const float* q, const float* k, const float* v, int n, int L,
                                       int S, float* out, uint32_t* flags, hipStream_t s) {
  const int qchunks = (L + 32 * FA_WAVES - 1) / (32 * FA_WAVES);
  hipLaunchKernelGGL(k_full_attention_split, dim3((unsigned)(n * NH * qchunks)), dim3(64 * FA_WAVES), 0, s, q,
                     k, v, L, S, out, flags);
  return hipGetLastError();
}

}  // namespace oetr
