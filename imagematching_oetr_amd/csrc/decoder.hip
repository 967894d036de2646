// Single-query decoder of the OETR QueryTransformer (gfx950).
//
// Reference: TransformerDecoder / DecoderLayer.forward
// (src/models/transformer.py:224-284) with MultiHeadAttention (:55-72) and
// LinearAttention (src/models/linear_attention.py:22-50); tgt starts at zero,
// tgt_pos = query embedding, memory = encoder output (transformer.py:361-381).
//
// There is ONE query token per image, so every projection is a 256-wide GEMV
// and the chain is latency bound.  What keeps it short:
//  * everything that does not depend on the images is folded at create time by
//    k_decoder_consts: layer 0's self-attention block acts on tgt = 0, so its
//    output, the following LN2 and the phi(Q) of layer 0's cross-attention are
//    per-side constants; layer 1's self-attention q/k projections see
//    LN1(tgt)+query_embed, so W.query_embed + b is a constant bias and q, k, v
//    come from ONE fused [256 -> 768] pass over LN1(tgt);
//  * layer 0's cross-attention is linear in the memory state and its query is
//    a constant, so the encoder's tail launch already reduced each token tile
//    to a 256-float partial message (+8 partial normalisers) instead of an
//    8192-float state;
//  * one workgroup of 16 waves per image; each GEMV puts a whole matrix in
//    flight (weights stored transposed [in][out]: every wave-load is 1 KiB
//    contiguous, 16 independent float4 loads per thread).
#include "common.h"

namespace oetr {

constexpr int DEC_THREADS = 1024;

// part[kc][NOUT] = sum over this thread's k-chunk of x[k] * Wt[k][NOUT]; the
// caller syncs and then sums the KCH partials.
template <int K, int NOUT>
__device__ __forceinline__ void gemv_partial(const float* __restrict__ Wt, const float* x_s,
                                             float* part_s, int tid) {
  constexpr int NO4 = NOUT / 4;
  constexpr int KCH = DEC_THREADS / NO4;
  constexpr int KPER = K / KCH;
  constexpr int BATCH = KPER < 16 ? KPER : 16;  // float4 loads in flight per thread
  const int kc = tid / NO4, o4 = tid % NO4;
  const f32x4* w = reinterpret_cast<const f32x4*>(Wt) + (size_t)kc * KPER * NO4 + o4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i0 = 0; i0 < KPER; i0 += BATCH) {
    f32x4 wv[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) wv[i] = w[(size_t)(i0 + i) * NO4];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) acc += wv[i] * x_s[kc * KPER + i0 + i];
  }
  *reinterpret_cast<f32x4*>(part_s + kc * NOUT + 4 * o4) = acc;
}
template <int K, int NOUT>
__device__ __forceinline__ float gemv_collect(const float* part_s, int o) {
  constexpr int KCH = DEC_THREADS / (NOUT / 4);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < KCH; ++c) s += part_s[c * NOUT + o];
  return s;
}
// out[NOUT] = act(x . Wt + bias); two barriers.
template <int K, int NOUT>
__device__ __forceinline__ void gemv(const float* __restrict__ Wt, const float* x_s,
                                     const float* __restrict__ bias, float* out_s,
                                     float* part_s, int tid, bool relu = false) {
  gemv_partial<K, NOUT>(Wt, x_s, part_s, tid);
  __syncthreads();
  for (int o = tid; o < NOUT; o += DEC_THREADS) {
    float s = gemv_collect<K, NOUT>(part_s, o);
    if (bias) s += bias[o];
    out_s[o] = relu ? fmaxf(s, 0.f) : s;
  }
  __syncthreads();
}

// LayerNorm of one 256-vector (wave 0); optional second output y + add.
__device__ __forceinline__ void ln_vec(const float* in_s, const float* __restrict__ w,
                                       const float* __restrict__ b, float* out_s,
                                       const float* add_s, float* out_add_s, int tid) {
  if (tid < 64) {
    const f32x4 v = reinterpret_cast<const f32x4*>(in_s)[tid];
    const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / C);
    const f32x4 d = v - mean;
    const float var =
        wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / C);
    const float rstd = 1.0f / sqrtf(var + LN_EPS);
    const f32x4 y = d * rstd * reinterpret_cast<const f32x4*>(w)[tid] +
                    reinterpret_cast<const f32x4*>(b)[tid];
    reinterpret_cast<f32x4*>(out_s)[tid] = y;
    if (out_add_s)
      reinterpret_cast<f32x4*>(out_add_s)[tid] = y + reinterpret_cast<const f32x4*>(add_s)[tid];
  }
  __syncthreads();
}

// Linear attention of one query against one key/value token (L = S = 1):
// the decoder's self-attention (values / v_length with v_length = 1).
__device__ __forceinline__ float self_attn_1x1(const float* q_s, const float* k_s, float v,
                                               int tid) {
  const int h = tid >> 5;
  float z = 0.f, s = 0.f;
  const float vval = v / 1.0f;
#pragma unroll 4
  for (int d = 0; d < HD; ++d) {
    const float fq = elu1(q_s[h * HD + d]), fk = elu1(k_s[h * HD + d]);
    z += fq * fk;
    s += fq * (fk * vval);
  }
  return s * (1.0f / (z + ATTN_EPS)) * 1.0f;
}

// ---------------------------------------------------------------------------
// Create-time constants, one block per side (launched once from oetr_create).
//   tgt1[side]  : tgt after layer 0's self-attention block (tgt0 = 0)
//   q0[side]    : phi(Wq (LN2(tgt1) + qe) + bq) of layer 0's cross-attention
//   qkv1[side]  : [Wq.qe + bq | Wk.qe + bk | bv] of layer 1's self-attention
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(DEC_THREADS) void k_decoder_consts(DecConstLaunch p) {
  __shared__ __attribute__((aligned(16))) float part_s[4096];
  __shared__ __attribute__((aligned(16))) float vec_s[8][C];
  float *tgt = vec_s[0], *t2 = vec_s[1], *qk = vec_s[2], *vq = vec_s[3], *vk = vec_s[4],
        *vv = vec_s[5], *att = vec_s[6], *qe = vec_s[7];
  const int tid = threadIdx.x, side = blockIdx.x;
  const DecLayerDev& w0 = p.layer[0];
  const DecLayerDev& w1 = p.layer[1];
  if (tid < C) { tgt[tid] = 0.f; qe[tid] = p.qe[side][tid]; }
  __syncthreads();
  ln_vec(tgt, w0.n1w, w0.n1b, t2, qe, qk, tid);
  gemv<C, C>(w0.self_attn.wq_t, qk, w0.self_attn.bq, vq, part_s, tid);
  gemv<C, C>(w0.self_attn.wk_t, qk, w0.self_attn.bk, vk, part_s, tid);
  gemv<C, C>(w0.self_attn.wv_t, t2, w0.self_attn.bv, vv, part_s, tid);
  if (tid < C) att[tid] = self_attn_1x1(vq, vk, vv[tid], tid);
  __syncthreads();
  gemv<C, C>(w0.self_attn.wm_t, att, nullptr, tgt, part_s, tid);  // tgt1 = 0 + msg
  if (tid < C) p.tgt1[side * C + tid] = tgt[tid];
  ln_vec(tgt, w0.n2w, w0.n2b, t2, qe, qk, tid);
  gemv<C, C>(w0.cross.wq_t, qk, w0.cross.bq, vq, part_s, tid);
  if (tid < C) p.q0[side * C + tid] = elu1(vq[tid]);
  // layer 1 self-attention bias constants
  gemv<C, C>(w1.self_attn.wq_t, qe, w1.self_attn.bq, vq, part_s, tid);
  gemv<C, C>(w1.self_attn.wk_t, qe, w1.self_attn.bk, vk, part_s, tid);
  if (tid < C) {
    p.qkv1[side * 3 * C + tid] = vq[tid];
    p.qkv1[side * 3 * C + C + tid] = vk[tid];
    p.qkv1[side * 3 * C + 2 * C + tid] = w1.self_attn.bv[tid];
  }
}

hipError_t launch_decoder_consts(const DecConstLaunch& p, hipStream_t s) {
  hipLaunchKernelGGL(k_decoder_consts, dim3(2), dim3(DEC_THREADS), 0, s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Software-pipelined GEMV chain: a stage's weights are consumed in batches of 16
// float4 per thread (64 VGPRs); as soon as a batch has been multiplied in, the
// NEXT batch (of this or the following stage - weights do not depend on data) is
// issued into the same registers, so its L2 round trip runs under the current
// stage's partial-sum exchange, barriers and LayerNorm.
template <int K, int NOUT>
struct GemvShape {
  static constexpr int NO4 = NOUT / 4;
  static constexpr int KCH = DEC_THREADS / NO4;  // k-chunks (threads per output float4)
  static constexpr int KPER = K / KCH;           // k rows per thread
  static constexpr int NB = KPER / 16;           // batches of 16 rows
  static_assert(KPER % 16 == 0, "batching");
};
template <int K, int NOUT, int B>
__device__ __forceinline__ void gemv_issue(const float* __restrict__ Wt, int tid, f32x4 (&wv)[16]) {
  using S = GemvShape<K, NOUT>;
  const int kc = tid / S::NO4, o4 = tid % S::NO4;
  const f32x4* w = reinterpret_cast<const f32x4*>(Wt) + (size_t)(kc * S::KPER + B * 16) * S::NO4 + o4;
#pragma unroll
  for (int i = 0; i < 16; ++i) wv[i] = w[(size_t)i * S::NO4];
  __builtin_amdgcn_sched_barrier(0);  // keep the loads here
}
template <int K, int NOUT, int B>
__device__ __forceinline__ void gemv_fma(const f32x4 (&wv)[16], const float* x_s, int tid, f32x4& acc) {
  using S = GemvShape<K, NOUT>;
  const int kc = tid / S::NO4;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += wv[i] * x_s[kc * S::KPER + B * 16 + i];
}
template <int K, int NOUT>
__device__ __forceinline__ void gemv_put(const f32x4& acc, float* part_s, int tid) {
  using S = GemvShape<K, NOUT>;
  *reinterpret_cast<f32x4*>(part_s + (tid / S::NO4) * NOUT + 4 * (tid % S::NO4)) = acc;
}

__global__ __launch_bounds__(DEC_THREADS) void k_decoder(DecLaunch p) {
  __shared__ __attribute__((aligned(16))) float kv_s[KV_FLOATS];  // [h][d][v], layer 1
  __shared__ __attribute__((aligned(16))) float part_s[3 * 4096];
  __shared__ __attribute__((aligned(16))) float vec_s[8][C];
  __shared__ __attribute__((aligned(16))) float qkv_s[3 * C];
  __shared__ __attribute__((aligned(16))) float hdn_s[FF];
  float *tgt = vec_s[0], *t2 = vec_s[1], *qk = vec_s[2], *vq = vec_s[3], *att = vec_s[4],
        *qe = vec_s[6], *ksum = vec_s[7];

  const Geom& g = p.g;
  const int tid = threadIdx.x;
  const int img = blockIdx.x;
  const int side = img >= g.N, n = side ? img - g.N : img;
  const int L = g.L[side], nts = g.nt[side];
  const int slot0 = g.tile0[side] + n * nts;
  const DecLayerDev& w0 = p.layer[0];
  const DecLayerDev& w1 = p.layer[1];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 wv[16];
  f32x4 acc;
  gemv_issue<C, C, 0>(w0.cross.wm_t, tid, wv);  // stage 1 weights, under the state reduction
  PHASE_STAMP(p, 0);

  // ---- reduce the tile partials of this image in one pass (4 tiles in flight per
  // round trip): layer 1 memory state -> LDS, layer 0 partial messages -> att
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(p.dkv1) + (size_t)slot0 * (KV_FLOATS / 4) + tid;
    const int c = tid & (C - 1), hh = c >> 5;
    f32x4 s0 = zero4, s1 = zero4;
    float ks = 0.f, a0 = 0.f, z0 = 0.f;
    for (int ti0 = 0; ti0 < nts; ti0 += 4) {
      f32x4 a[4], b[4];
      float kt[4], av[4], zv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t ti = min(ti0 + u, nts - 1);
        a[u] = src[ti * (KV_FLOATS / 4)];
        b[u] = src[ti * (KV_FLOATS / 4) + DEC_THREADS];
        kt[u] = p.dks1[(slot0 + ti) * C + c];
        av[u] = p.att0_part[(slot0 + ti) * C + c];
        zv[u] = p.z0_part[(slot0 + ti) * NH + hh];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ti0 + u < nts) { s0 += a[u]; s1 += b[u]; ks += kt[u]; a0 += av[u]; z0 += zv[u]; }
    }
#pragma unroll
    for (int e2 = 0; e2 < 2; ++e2) {
      const int e = tid + DEC_THREADS * e2;  // ((h*4+q)*64 + lane)
      const f32x4 s = e2 ? s1 : s0;
      const int ln = e & 63, q = (e >> 6) & 3, h = e >> 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) kv_s[(h * HD + (j + 8 * q + 4 * (ln >> 5))) * HD + (ln & 31)] = s[j];
    }
    if (tid < C) {
      ksum[tid] = ks;
      att[tid] = a0 * (1.0f / (z0 + ATTN_EPS)) * (float)L;
      tgt[tid] = p.tgt1[side * C + tid];
      qe[tid] = p.qe[side][tid];
    }
  }
  __syncthreads();

  PHASE_STAMP(p, 1);
  // S1: tgt += Wm_c0 . att0
  acc = zero4;
  gemv_fma<C, C, 0>(wv, att, tid, acc);
  gemv_issue<C, FF, 0>(w0.w1_t, tid, wv);
  gemv_put<C, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) tgt[tid] += gemv_collect<C, C>(part_s, tid);
  __syncthreads();
  PHASE_STAMP(p, 2);
  // S2: hdn = relu(W1_0 . LN3(tgt))
  ln_vec(tgt, w0.n3w, w0.n3b, t2, nullptr, nullptr, tid);
  acc = zero4;
  gemv_fma<C, FF, 0>(wv, t2, tid, acc);
  gemv_issue<C, FF, 1>(w0.w1_t, tid, wv);
  gemv_fma<C, FF, 1>(wv, t2, tid, acc);
  gemv_issue<FF, C, 0>(w0.w2_t, tid, wv);
  gemv_put<C, FF>(acc, part_s, tid);
  __syncthreads();
  if (tid < FF) hdn_s[tid] = fmaxf(gemv_collect<C, FF>(part_s, tid), 0.f);
  __syncthreads();
  PHASE_STAMP(p, 3);
  // S3: tgt += W2_0 . hdn
  acc = zero4;
  gemv_fma<FF, C, 0>(wv, hdn_s, tid, acc);
  gemv_issue<FF, C, 1>(w0.w2_t, tid, wv);
  gemv_fma<FF, C, 1>(wv, hdn_s, tid, acc);
  gemv_issue<C, C, 0>(w1.self_attn.wq_t, tid, wv);
  gemv_put<FF, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) tgt[tid] += gemv_collect<FF, C>(part_s, tid);
  __syncthreads();

  PHASE_STAMP(p, 4);
  // S4: layer 1 self-attention: fused q|k|v from LN1(tgt)
  ln_vec(tgt, w1.n1w, w1.n1b, t2, nullptr, nullptr, tid);
  acc = zero4;
  gemv_fma<C, C, 0>(wv, t2, tid, acc);
  gemv_issue<C, C, 0>(w1.self_attn.wk_t, tid, wv);
  gemv_put<C, C>(acc, part_s, tid);
  acc = zero4;
  gemv_fma<C, C, 0>(wv, t2, tid, acc);
  gemv_issue<C, C, 0>(w1.self_attn.wv_t, tid, wv);
  gemv_put<C, C>(acc, part_s + 4096, tid);
  acc = zero4;
  gemv_fma<C, C, 0>(wv, t2, tid, acc);
  gemv_issue<C, C, 0>(w1.self_attn.wm_t, tid, wv);
  gemv_put<C, C>(acc, part_s + 8192, tid);
  __syncthreads();
  if (tid < 3 * C) {
    const int m = tid >> 8, o = tid & (C - 1);
    qkv_s[tid] = gemv_collect<C, C>(part_s + 4096 * m, o) + p.qkv1[side * 3 * C + tid];
  }
  __syncthreads();
  if (tid < 2 * C) qkv_s[tid] = elu1(qkv_s[tid]);  // phi(q), phi(k) once per element
  __syncthreads();
  if (tid < C) {
    // L = S = 1 linear attention (values / v_length with v_length = 1)
    const int h = tid >> 5;
    const float vval = qkv_s[2 * C + tid] / 1.0f;
    float z = 0.f, sacc = 0.f;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) {
      const float fq = qkv_s[h * HD + d], fk = qkv_s[C + h * HD + d];
      z += fq * fk;
      sacc += fq * (fk * vval);
    }
    att[tid] = sacc * (1.0f / (z + ATTN_EPS)) * 1.0f;
  }
  __syncthreads();
  PHASE_STAMP(p, 5);
  // S5: tgt += Wm_s1 . att
  acc = zero4;
  gemv_fma<C, C, 0>(wv, att, tid, acc);
  gemv_issue<C, C, 0>(w1.cross.wq_t, tid, wv);
  gemv_put<C, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) tgt[tid] += gemv_collect<C, C>(part_s, tid);
  __syncthreads();
  PHASE_STAMP(p, 6);
  // S6: layer 1 cross-attention query
  ln_vec(tgt, w1.n2w, w1.n2b, t2, qe, qk, tid);
  acc = zero4;
  gemv_fma<C, C, 0>(wv, qk, tid, acc);
  gemv_issue<C, C, 0>(w1.cross.wm_t, tid, wv);
  gemv_put<C, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) vq[tid] = elu1(gemv_collect<C, C>(part_s, tid) + w1.cross.bq[tid]);
  __syncthreads();
  if (tid < C) {
    const int h = tid >> 5, v = tid & 31;
    float z = 0.f, s = 0.f;
#pragma unroll 4
    for (int d = 0; d < HD; ++d) {
      const float fq = vq[h * HD + d];
      z += fq * ksum[h * HD + d];
      s += fq * kv_s[(h * HD + d) * HD + v];
    }
    att[tid] = s * (1.0f / (z + ATTN_EPS)) * (float)L;
  }
  __syncthreads();
  PHASE_STAMP(p, 7);
  // S7: tgt += Wm_c1 . att
  acc = zero4;
  gemv_fma<C, C, 0>(wv, att, tid, acc);
  gemv_issue<C, FF, 0>(w1.w1_t, tid, wv);
  gemv_put<C, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) tgt[tid] += gemv_collect<C, C>(part_s, tid);
  __syncthreads();
  PHASE_STAMP(p, 8);
  // S8/S9: ReLU MLP
  ln_vec(tgt, w1.n3w, w1.n3b, t2, nullptr, nullptr, tid);
  acc = zero4;
  gemv_fma<C, FF, 0>(wv, t2, tid, acc);
  gemv_issue<C, FF, 1>(w1.w1_t, tid, wv);
  gemv_fma<C, FF, 1>(wv, t2, tid, acc);
  gemv_issue<FF, C, 0>(w1.w2_t, tid, wv);
  gemv_put<C, FF>(acc, part_s, tid);
  __syncthreads();
  if (tid < FF) hdn_s[tid] = fmaxf(gemv_collect<C, FF>(part_s, tid), 0.f);
  __syncthreads();
  acc = zero4;
  gemv_fma<FF, C, 0>(wv, hdn_s, tid, acc);
  gemv_issue<FF, C, 1>(w1.w2_t, tid, wv);
  gemv_fma<FF, C, 1>(wv, hdn_s, tid, acc);
  gemv_put<FF, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) p.hs[(size_t)img * C + tid] = tgt[tid] + gemv_collect<FF, C>(part_s, tid);
  PHASE_STAMP(p, 9);
}

hipError_t launch_decoder(const DecLaunch& p, hipStream_t s) {
  hipLaunchKernelGGL(k_decoder, dim3(2 * p.g.N), dim3(DEC_THREADS), 0, s, p);
  return hipGetLastError();
}

}  // namespace oetr
