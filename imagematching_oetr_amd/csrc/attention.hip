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
// (a = hi + lo/2^11, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation): 12
// MFMAs of 32 cycles per 32x32 score tile instead of 32 f32 MFMAs of 64 cycles.
//
// One block = (image, head, 128 queries), 4 waves x 32 queries.  Per 32-key tile the
// block converts K and V once into split planes in LDS (K row-major, V transposed with
// the key order the MFMA k-slots want), then every wave computes, with NO lane
// exchange at all:
//   S^T = K . Q^T      A = K fragments (LDS),  B = Q (registers, split once)
//   P   = exp(S^T/sqrt(D) - m)  in the lane that owns the query column
//   O^T += V^T . P^T   A = V^T fragments (LDS), B = P straight from the S^T accumulator
//                      registers (k-slot 8*half + i of step s <-> key crow(8s+i, half))
// O^T keeps the query in the lane, so the online-softmax rescale and the final 1/l are
// per-lane scalars.  The L x S score volume exists only in accumulators.
constexpr int FA_PITCH = 40;   // halves per LDS row (32 + 8): conflict-free ds_read_b128

__global__ __launch_bounds__(256) void k_full_attention_split(const float* __restrict__ q,
                                                              const float* __restrict__ k,
                                                              const float* __restrict__ v,
                                                              int L, int S, float* __restrict__ out,
                                                              uint32_t* flags) {
  __shared__ __attribute__((aligned(16))) _Float16 Kh[32 * FA_PITCH], Kl[32 * FA_PITCH];
  __shared__ __attribute__((aligned(16))) _Float16 Vh[32 * FA_PITCH], Vl[32 * FA_PITCH];  // [d][slot]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, col = lane & 31;
  const int qchunks = (L + 127) / 128;
  const int qc = blockIdx.x % qchunks, nh = blockIdx.x / qchunks, n = nh / NH, h = nh % NH;
  const int q0 = qc * 128 + wave * 32;
  const float temp = 1.0f / sqrtf((float)HD);
  Range rg;

  // Q as the B operand of S^T = K Q^T: lane (query = col) holds d = 16s + 8*half + 0..7
  f32x4 qh[2], ql[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    if (q0 + col < L) {
      const float* qp = q + (((size_t)n * L + q0 + col) * NH + h) * HD + 16 * s + 8 * half;
      a0 = *reinterpret_cast<const f32x4*>(qp);
      a1 = *reinterpret_cast<const f32x4*>(qp + 4);
    }
    split8(a0, a1, qh[s], ql[s], rg);
  }
  f32x16 o = {0};   // O^T: rows = d (crow(r, half)), col = query
  // (masked scores and the initial maximum are a large FINITE negative: exp_neg - the
  //  6-instruction exp of common.h, arguments <= 0 - would turn -inf into NaN)
  constexpr float NEG = -1.0e30f;
  float m_run = NEG, l_run = 0.f;

  // staging role: thread -> (key row = tid >> 3, 4 consecutive d = 4 * (tid & 7))
  const int srow = tid >> 3, sd = 4 * (tid & 7);
  // position of key `srow` in the k-slot order of the P.V contraction: swap bits 2 and 3
  const int spos = (srow & ~12) | ((srow & 4) << 1) | ((srow & 8) >> 1);
  auto load_kv = [&](int k0, f32x4& kk, f32x4& vv) {
    kk = f32x4{0.f, 0.f, 0.f, 0.f};
    vv = kk;
    if (k0 + srow < S) {
      const size_t off = (((size_t)n * S + k0 + srow) * NH + h) * HD + sd;
      kk = *reinterpret_cast<const f32x4*>(k + off);
      vv = *reinterpret_cast<const f32x4*>(v + off);
    }
  };
  f32x4 kreg, vreg;
  load_kv(0, kreg, vreg);
  for (int k0 = 0; k0 < S; k0 += 32) {
    __syncthreads();   // every wave is done with the previous tile's planes
    store_planes4<GM_SPLIT>(Kh + srow * FA_PITCH, Kl + srow * FA_PITCH, sd, kreg, rg);
    {
      uint32_t h0, l0, h1, l1;
      cvt_planes2<GM_SPLIT>(vreg[0], vreg[1], h0, l0, rg);
      cvt_planes2<GM_SPLIT>(vreg[2], vreg[3], h1, l1, rg);
      uint16_t* vh = reinterpret_cast<uint16_t*>(Vh);
      uint16_t* vl = reinterpret_cast<uint16_t*>(Vl);
      vh[(sd + 0) * FA_PITCH + spos] = (uint16_t)h0; vh[(sd + 1) * FA_PITCH + spos] = (uint16_t)(h0 >> 16);
      vh[(sd + 2) * FA_PITCH + spos] = (uint16_t)h1; vh[(sd + 3) * FA_PITCH + spos] = (uint16_t)(h1 >> 16);
      vl[(sd + 0) * FA_PITCH + spos] = (uint16_t)l0; vl[(sd + 1) * FA_PITCH + spos] = (uint16_t)(l0 >> 16);
      vl[(sd + 2) * FA_PITCH + spos] = (uint16_t)l1; vl[(sd + 3) * FA_PITCH + spos] = (uint16_t)(l1 >> 16);
    }
    if (k0 + 32 < S) load_kv(k0 + 32, kreg, vreg);   // next tile's rows under this tile's math
    __syncthreads();

    // S^T tile: rows = keys, cols = queries
    f32x16 st = {0}, cr = {0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f32x4 ah = *reinterpret_cast<const f32x4*>(Kh + col * FA_PITCH + 16 * s + 8 * half);
      const f32x4 al = *reinterpret_cast<const f32x4*>(Kl + col * FA_PITCH + 16 * s + 8 * half);
      mma16_split3(ah, al, qh[s], ql[s], st, cr);
    }
    float mt = NEG;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sv = fmaf(cr[r], SPLIT_INV, st[r]);
      st[r] = (k0 + crow(r, half) < S) ? sv * temp : NEG;
      mt = fmaxf(mt, st[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = exp_neg(fminf(m_run - m_new, 0.f));  // 0 on the first tile
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = exp_neg(fminf(st[r] - m_new, 0.f)); ps += st[r]; }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;   // this lane's query, all of its d rows
    // O^T += V^T P^T
    f32x16 oc = {0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 ph, pl;
      split8(f32x4{st[8 * s], st[8 * s + 1], st[8 * s + 2], st[8 * s + 3]},
             f32x4{st[8 * s + 4], st[8 * s + 5], st[8 * s + 6], st[8 * s + 7]}, ph, pl, rg);
      const f32x4 ah = *reinterpret_cast<const f32x4*>(Vh + col * FA_PITCH + 16 * s + 8 * half);
      const f32x4 al = *reinterpret_cast<const f32x4*>(Vl + col * FA_PITCH + 16 * s + 8 * half);
      mma16_split3(ah, al, ph, pl, o, oc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = fmaf(oc[r], SPLIT_INV, o[r]);
  }
  if (q0 + col < L) {
    const float inv = 1.0f / l_run;
    float* dst = out + (((size_t)n * L + q0 + col) * NH + h) * HD + 4 * half;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)   // registers 4*g4 .. 4*g4+3 = d rows 8*g4 + 4*half + 0..3
      *reinterpret_cast<f32x4*>(dst + 8 * g4) =
          f32x4{o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv};
  }
  if (flags) range_report<GM_SPLIT>(rg, flags);
}

hipError_t launch_full_attention_split(const float* q, const float* k, const float* v, int n, int L,
                                       int S, float* out, uint32_t* flags, hipStream_t s) {
  const int qchunks = (L + 127) / 128;
  hipLaunchKernelGGL(k_full_attention_split, dim3((unsigned)(n * NH * qchunks)), dim3(256), 0, s, q,
                     k, v, L, S, out, flags);
  return hipGetLastError();
}

}  // namespace oetr
