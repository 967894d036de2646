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
#include "conv_p.h"

namespace oetr {

constexpr int DEC_THREADS = 1024;  // create-time constants kernel

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
// Software-pipelined GEMV chain, T threads per image.  A stage's weights are
// consumed in batches of 16 float4 per thread (64 VGPRs); as soon as a batch has
// been multiplied in, the NEXT batch (of this or the following stage - weights
// do not depend on data) is issued into the same registers, so its L2 round trip
// runs under the current stage's partial-sum exchange, barriers and LayerNorm.
// (Two register sets / two batches in flight were tried in round 2: 48.9 vs 49.1 us for
//  the chain alone - it runs at 45-50 B/clk, the fill rate of ONE CU's L1, for its 3.9 MB
//  of fp32 weights per image; only more CUs per image would shorten it.)
template <int T, int K, int NOUT>
struct GemvShape {
  static constexpr int NO4 = NOUT / 4;
  static constexpr int KCH = T / NO4;   // k-chunks (threads per output float4)
  static constexpr int KPER = K / KCH;  // k rows per thread
  static constexpr int NB = KPER / 16;  // batches of 16 rows
  static_assert(KPER % 16 == 0 && KCH * NOUT == 4 * T, "batching");
};
template <int T, int K, int NOUT, int B>
__device__ __forceinline__ void gemv_issue(const float* __restrict__ Wt, int tid, f32x4 (&wv)[16]) {
  using S = GemvShape<T, K, NOUT>;
  const int kc = tid / S::NO4, o4 = tid % S::NO4;
  const f32x4* w = reinterpret_cast<const f32x4*>(Wt) + (size_t)(kc * S::KPER + B * 16) * S::NO4 + o4;
#pragma unroll
  for (int i = 0; i < 16; ++i) wv[i] = w[(size_t)i * S::NO4];
  __builtin_amdgcn_sched_barrier(0);  // keep the loads here
}
// All batches of one stage: FMA batch B, then issue batch B+1 - or, after the last
// one, whatever `next` issues (the first batch of the following stage).
template <int T, int K, int NOUT, int B = 0, class Next>
__device__ __forceinline__ void gemv_stage(const float* __restrict__ Wt, const float* x_s, int tid,
                                           f32x4 (&wv)[16], f32x4& acc, Next next) {
  using S = GemvShape<T, K, NOUT>;
  const int kc = tid / S::NO4;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += wv[i] * x_s[kc * S::KPER + B * 16 + i];
  if constexpr (B + 1 < S::NB) {
    gemv_issue<T, K, NOUT, B + 1>(Wt, tid, wv);
    gemv_stage<T, K, NOUT, B + 1>(Wt, x_s, tid, wv, acc, next);
  } else {
    next();
  }
}
// partial sums -> LDS [KCH][NOUT]; collect = sum over the KCH chunks
template <int T, int K, int NOUT>
__device__ __forceinline__ void gemv_put(const f32x4& acc, float* part_s, int tid) {
  using S = GemvShape<T, K, NOUT>;
  *reinterpret_cast<f32x4*>(part_s + (tid / S::NO4) * NOUT + 4 * (tid % S::NO4)) = acc;
}
template <int T, int K, int NOUT>
__device__ __forceinline__ float gemv_get(const float* part_s, int o) {
  using S = GemvShape<T, K, NOUT>;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < S::KCH; ++c) s += part_s[c * NOUT + o];
  return s;
}

// LDS of one decoder workgroup (floats)
template <int T>
struct DecSmem {
  static constexpr int PART = 4 * T;  // one partial-sum buffer
  static constexpr int KV = 0, PARTS = KV + KV_FLOATS, VEC = PARTS + 3 * PART,
                       QKV = VEC + 8 * C, HDN = QKV + 3 * C, TOTAL = HDN + FF;
};

// The whole decoder for image `img` (both layers); see the file header.
template <int T>
__device__ __forceinline__ void decoder_body(const DecLaunch& p, int img, float* smem) {
  using M = DecSmem<T>;
  float* kv_s = smem + M::KV;  // [h][d][v], layer 1
  float* part_s = smem + M::PARTS;
  float* vec = smem + M::VEC;
  float* qkv_s = smem + M::QKV;
  float* hdn_s = smem + M::HDN;
  float *tgt = vec, *t2 = vec + C, *qk = vec + 2 * C, *vq = vec + 3 * C, *att = vec + 4 * C,
        *qe = vec + 6 * C, *ksum = vec + 7 * C;
  constexpr int PART = M::PART;

  const Geom& g = p.g;
  const int tid = threadIdx.x;
  const int side = img >= g.N, n = side ? img - g.N : img;
  const int L = g.L[side], nts = g.nt[side];
  const int slot0 = g.tile0[side] + n * nts;
  const DecLayerDev& w0 = p.layer[0];
  const DecLayerDev& w1 = p.layer[1];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 wv[16];
  f32x4 acc;
  gemv_issue<T, C, C, 0>(w0.cross.wm_t, tid, wv);  // stage 1 weights, under the state reduction
  PHASE_STAMP(p, 0);

  // ---- reduce the tile partials of this image in one pass (several tiles in flight
  // per round trip): layer 1 memory state -> LDS, layer 0 partial messages -> att
  {
    constexpr int NE = (KV_FLOATS / 4) / T;  // float4 of the state per thread
    const f32x4* src = reinterpret_cast<const f32x4*>(p.dkv1) + (size_t)slot0 * (KV_FLOATS / 4) + tid;
    const int c = tid & (C - 1), hh = c >> 5;
    f32x4 sacc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) sacc[e] = zero4;
    float ks = 0.f, a0 = 0.f, z0 = 0.f;
    constexpr int R = T == 512 ? 7 : 4;  // tiles per round trip (R*NE float4 in flight per thread)
    for (int ti0 = 0; ti0 < nts; ti0 += R) {
      f32x4 a[R][NE];
      float kt[R], av[R], zv[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const size_t ti = min(ti0 + u, nts - 1);
#pragma unroll
        for (int e = 0; e < NE; ++e) a[u][e] = src[ti * (KV_FLOATS / 4) + e * T];
        kt[u] = p.dks1[(slot0 + ti) * C + c];
        av[u] = p.att0_part[(slot0 + ti) * C + c];
        zv[u] = p.z0_part[(slot0 + ti) * NH + hh];
      }
#pragma unroll
      for (int u = 0; u < R; ++u)
        if (ti0 + u < nts) {
#pragma unroll
          for (int e = 0; e < NE; ++e) sacc[e] += a[u][e];
          ks += kt[u]; a0 += av[u]; z0 += zv[u];
        }
    }
#pragma unroll
    for (int e2 = 0; e2 < NE; ++e2) {
      const int e = tid + T * e2;  // ((h*4+q)*64 + lane)
      const int ln = e & 63, q = (e >> 6) & 3, h = e >> 8;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        kv_s[(h * HD + (j + 8 * q + 4 * (ln >> 5))) * HD + (ln & 31)] = sacc[e2][j];
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
  gemv_stage<T, C, C>(w0.cross.wm_t, att, tid, wv, acc, [&] { gemv_issue<T, C, FF, 0>(w0.w1_t, tid, wv); });
  gemv_put<T, C, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) tgt[tid] += gemv_get<T, C, C>(part_s, tid);
  __syncthreads();
  PHASE_STAMP(p, 2);
  // S2: hdn = relu(W1_0 . LN3(tgt))
  ln_vec(tgt, w0.n3w, w0.n3b, t2, nullptr, nullptr, tid);
  acc = zero4;
  gemv_stage<T, C, FF>(w0.w1_t, t2, tid, wv, acc, [&] { gemv_issue<T, FF, C, 0>(w0.w2_t, tid, wv); });
  gemv_put<T, C, FF>(acc, part_s, tid);
  __syncthreads();
  for (int o = tid; o < FF; o += T) hdn_s[o] = fmaxf(gemv_get<T, C, FF>(part_s, o), 0.f);
  __syncthreads();
  PHASE_STAMP(p, 3);
  // S3: tgt += W2_0 . hdn
  acc = zero4;
  gemv_stage<T, FF, C>(w0.w2_t, hdn_s, tid, wv, acc,
                       [&] { gemv_issue<T, C, C, 0>(w1.self_attn.wq_t, tid, wv); });
  gemv_put<T, FF, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) tgt[tid] += gemv_get<T, FF, C>(part_s, tid);
  __syncthreads();

  PHASE_STAMP(p, 4);
  // S4: layer 1 self-attention: fused q|k|v from LN1(tgt)
  ln_vec(tgt, w1.n1w, w1.n1b, t2, nullptr, nullptr, tid);
  acc = zero4;
  gemv_stage<T, C, C>(w1.self_attn.wq_t, t2, tid, wv, acc,
                      [&] { gemv_issue<T, C, C, 0>(w1.self_attn.wk_t, tid, wv); });
  gemv_put<T, C, C>(acc, part_s, tid);
  acc = zero4;
  gemv_stage<T, C, C>(w1.self_attn.wk_t, t2, tid, wv, acc,
                      [&] { gemv_issue<T, C, C, 0>(w1.self_attn.wv_t, tid, wv); });
  gemv_put<T, C, C>(acc, part_s + PART, tid);
  acc = zero4;
  gemv_stage<T, C, C>(w1.self_attn.wv_t, t2, tid, wv, acc,
                      [&] { gemv_issue<T, C, C, 0>(w1.self_attn.wm_t, tid, wv); });
  gemv_put<T, C, C>(acc, part_s + 2 * PART, tid);
  __syncthreads();
  for (int i = tid; i < 3 * C; i += T) {
    const int m = i >> 8, o = i & (C - 1);
    const float v = gemv_get<T, C, C>(part_s + PART * m, o) + p.qkv1[side * 3 * C + i];
    qkv_s[i] = m < 2 ? elu1(v) : v;  // phi(q), phi(k) once per element; v as is
  }
  __syncthreads();
  if (tid < C) {
    // L = S = 1 linear attention (values / v_length with v_length = 1)
    const int h = tid >> 5;
    const float vval = qkv_s[2 * C + tid] / 1.0f;
    float z = 0.f, sa = 0.f;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) {
      const float fq = qkv_s[h * HD + d], fk = qkv_s[C + h * HD + d];
      z += fq * fk;
      sa += fq * (fk * vval);
    }
    att[tid] = sa * (1.0f / (z + ATTN_EPS)) * 1.0f;
  }
  __syncthreads();
  PHASE_STAMP(p, 5);
  // S5: tgt += Wm_s1 . att
  acc = zero4;
  gemv_stage<T, C, C>(w1.self_attn.wm_t, att, tid, wv, acc,
                      [&] { gemv_issue<T, C, C, 0>(w1.cross.wq_t, tid, wv); });
  gemv_put<T, C, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) tgt[tid] += gemv_get<T, C, C>(part_s, tid);
  __syncthreads();
  PHASE_STAMP(p, 6);
  // S6: layer 1 cross-attention query
  ln_vec(tgt, w1.n2w, w1.n2b, t2, qe, qk, tid);
  acc = zero4;
  gemv_stage<T, C, C>(w1.cross.wq_t, qk, tid, wv, acc,
                      [&] { gemv_issue<T, C, C, 0>(w1.cross.wm_t, tid, wv); });
  gemv_put<T, C, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) vq[tid] = elu1(gemv_get<T, C, C>(part_s, tid) + w1.cross.bq[tid]);
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
  gemv_stage<T, C, C>(w1.cross.wm_t, att, tid, wv, acc, [&] { gemv_issue<T, C, FF, 0>(w1.w1_t, tid, wv); });
  gemv_put<T, C, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) tgt[tid] += gemv_get<T, C, C>(part_s, tid);
  __syncthreads();
  PHASE_STAMP(p, 8);
  // S8/S9: ReLU MLP
  ln_vec(tgt, w1.n3w, w1.n3b, t2, nullptr, nullptr, tid);
  acc = zero4;
  gemv_stage<T, C, FF>(w1.w1_t, t2, tid, wv, acc, [&] { gemv_issue<T, FF, C, 0>(w1.w2_t, tid, wv); });
  gemv_put<T, C, FF>(acc, part_s, tid);
  __syncthreads();
  for (int o = tid; o < FF; o += T) hdn_s[o] = fmaxf(gemv_get<T, C, FF>(part_s, o), 0.f);
  __syncthreads();
  acc = zero4;
  gemv_stage<T, FF, C>(w1.w2_t, hdn_s, tid, wv, acc, [] {});
  gemv_put<T, FF, C>(acc, part_s, tid);
  __syncthreads();
  if (tid < C) p.hs[(size_t)img * C + tid] = tgt[tid] + gemv_get<T, FF, C>(part_s, tid);
  PHASE_STAMP(p, 9);
}

// ---------------------------------------------------------------------------
// The same chain on DEC_K = 4 workgroups per image (round 4).  One CU streams its
// image's 3.9 MB of fp32 weights at the 45-50 B/clk its L1 fills at = 37-41 us of the
// 50-us chain; four CUs stream a quarter each.  The stages alternate between
//   column split: every workgroup holds the full input vector and produces its own
//                 quarter of the outputs (a head pair / an FFN slab) - nothing to exchange;
//   row split:    the next stage takes that quarter as its input rows and produces a
//                 partial of all 256 outputs - one all-reduce of 256 floats per image,
// so of the nine GEMV stages five end in an exchange (S1 S3 S5 S7 S9) and the attention
// pieces in between run per head pair: the layer-1 memory state is reduced and kept by
// quarters too.  Every workgroup ends each exchange with the same tgt (the four partials
// are added in workgroup order by everyone), LayerNorms are computed redundantly.
//
// Exchange = MI355X_MICROARCH.md "handoff-1to1" / cdna_hip_programming.md Guideline 16 R2:
// the data is the flag - 8-byte {tag, value} granules written by ONE agent-scope (sc1)
// store each and polled with agent-scope loads; no fence, no flag word.  The tag is the
// image's call counter, kept in the workspace's status block (zeroed by
// oetr_workspace_init together with the granule region): read by every workgroup of the
// image at its start and incremented by workgroup 0 after the last exchange - by then
// all four have read it - so graph replays, which repeat the kernel arguments, still see a
// fresh tag per call, and no per-call memset sits on the serial path.
// Residency: the host picks this form only while 2N * DEC_K <= a quarter of the CUs
// (api.hip: decoder_split), the decoder workgroups lead the grid, and a poll that
// outlasts DEC_SPIN_LIMIT of wall clock raises OETR_FLAG_EXCHANGE in the status word and
// moves on (the call's outputs are then invalid, like an operand-range overflow).
constexpr int DEC_K = DEC_SPLIT_K;
constexpr int XCH_STAGES = DEC_SPLIT_EXCHANGES;
constexpr long long DEC_SPIN_LIMIT = 200000;   // wall_clock64 ticks (100 MHz): 2 ms

typedef __attribute__((address_space(1))) unsigned long long gu64;

struct DecSmem4 {
  static constexpr int KV = 0, PARTS = KV + 2 * HD * HD, VEC = PARTS + 3 * 2048, SLAB = VEC + 4 * C,
                       TOTAL = SLAB + 64 * 3 + 192 + 128;
};

// weight fetches of one stage: KPER float4 per thread into w[B0 ..]
//  rows<KS>: rows [j*KS, (j+1)*KS) of Wt[in][256]; thread = (k-chunk tid/64 of KS/8 rows, output float4 tid%64)
template <int KS, int B0>
__device__ __forceinline__ void issue_rows(const float* __restrict__ Wt, int j, int tid, f32x4 (&w)[16]) {
  constexpr int KPER = KS / 8;
  const f32x4* src = reinterpret_cast<const f32x4*>(Wt) + (size_t)(j * KS + (tid >> 6) * KPER) * (C / 4) + (tid & 63);
#pragma unroll
  for (int i = 0; i < KPER; ++i) w[B0 + i] = src[i * (C / 4)];
  __builtin_amdgcn_sched_barrier(0);
}
//  cols<NS, LD>: columns [j*NS, (j+1)*NS) of Wt[256][LD]; thread = (k-chunk tid/(NS/4), output float4 tid%(NS/4))
template <int NS, int LD, int B0>
__device__ __forceinline__ void issue_cols(const float* __restrict__ Wt, int j, int tid, f32x4 (&w)[16]) {
  constexpr int NO4 = NS / 4, KCH = 512 / NO4, KPER = C / KCH;
  const f32x4* src = reinterpret_cast<const f32x4*>(Wt) + (size_t)((tid / NO4) * KPER) * (LD / 4) + j * NO4 + tid % NO4;
#pragma unroll
  for (int i = 0; i < KPER; ++i) w[B0 + i] = src[i * (LD / 4)];
  __builtin_amdgcn_sched_barrier(0);
}
// partial products: part[kc][256] (rows) / part[kc][NS] (cols)
template <int KS, int B0>
__device__ __forceinline__ void fma_rows(const f32x4 (&w)[16], const float* x_s, float* part_s, int tid) {
  constexpr int KPER = KS / 8;
  const int kc = tid >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KPER; ++i) acc += w[B0 + i] * x_s[kc * KPER + i];
  *reinterpret_cast<f32x4*>(part_s + kc * C + 4 * (tid & 63)) = acc;
}
template <int NS, int B0>
__device__ __forceinline__ void fma_cols(const f32x4 (&w)[16], const float* x_s, float* part_s, int tid) {
  constexpr int NO4 = NS / 4, KCH = 512 / NO4, KPER = C / KCH;
  const int kc = tid / NO4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KPER; ++i) acc += w[B0 + i] * x_s[kc * KPER + i];
  *reinterpret_cast<f32x4*>(part_s + kc * NS + 4 * (tid % NO4)) = acc;
}
template <int KCH, int NS>
__device__ __forceinline__ float collect(const float* part_s, int o) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < KCH; ++c) s += part_s[c * NS + o];
  return s;
}

// Sum of `mine` over the image's DEC_K workgroups, element tid of 256 (threads 0..255, whole
// waves); the same value - same order of additions - in every workgroup.  LEADER_ONLY: the
// last exchange, only workgroup 0 needs the sum.
// `dead` (per wave): an earlier exchange of this call timed out - the call's outputs are invalid
// already (OETR_FLAG_EXCHANGE is set), so the remaining exchanges publish and move on without
// waiting: a failed call costs one spin limit, not five.
// OETR_DEBUG_DECODER_FAULT (oetr_debug_decoder_fault, tests only): workgroup 1 of image 0 withholds
// its stage-0 granules, so that the time-out path - flag, early exit of late workgroups, the host's
// status-block reset - can be exercised on an idle device.
template <bool LEADER_ONLY>
__device__ __forceinline__ float exchange_sum(const DecLaunch& p, int img, int stage, int j, unsigned tag,
                                              float mine, int tid, bool& dead) {
  gu64* slot = (gu64*)p.xch + ((size_t)(img * XCH_STAGES + stage) * DEC_K) * C + tid;
  if (!(p.dbg == DEC_DBG_FAULT && img == 0 && j == 1 && stage == 0))
    __hip_atomic_store(slot + j * C, ((unsigned long long)tag << 32) | __float_as_uint(mine), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  if (LEADER_ONLY && j != 0) return 0.f;
  float v[DEC_K];
  const long long t0 = wall_clock64();
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < DEC_K; ++k) {
      const unsigned long long x = __hip_atomic_load(slot + k * C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v[k] = k == j ? mine : __uint_as_float((unsigned)x);
      ok &= k == j || (unsigned)(x >> 32) == tag;
    }
    if (__all(ok) || dead) break;
    __builtin_amdgcn_s_sleep(1);
    if ((spins & 63) == 63 && wall_clock64() - t0 > DEC_SPIN_LIMIT) {
      if ((tid & 63) == 0) atomicOr(p.flags, FLAG_EXCHANGE);
      dead = true;
      break;
    }
  }
  static_assert(DEC_K == 4, "fixed summation order below");
  return (v[0] + v[1]) + (v[2] + v[3]);
}

__device__ __forceinline__ void decoder_body4(const DecLaunch& p, int img, int j, float* smem) {
  using M = DecSmem4;
  float* kv_s = smem + M::KV;      // [2 heads][d][v], layer 1, heads 2j and 2j + 1
  float* part_s = smem + M::PARTS;
  float *tgt = smem + M::VEC, *t2 = tgt + C, *qk = tgt + 2 * C, *qe = tgt + 3 * C;
  float *att = smem + M::SLAB, *ksum = att + 64, *vq = att + 128, *qkv_s = att + 192, *hdn_s = qkv_s + 192;
  constexpr int T = 512;

  const Geom& g = p.g;
  const int tid = threadIdx.x;
  const int side = img >= g.N, n = side ? img - g.N : img;
  const int L = g.L[side], nts = g.nt[side];
  const int slot0 = g.tile0[side] + n * nts;
  const DecLayerDev& w0 = p.layer[0];
  const DecLayerDev& w1 = p.layer[1];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // A workgroup that becomes resident only after its peers have given up on it (OETR_FLAG_EXCHANGE
  // is set: by this call, or by an earlier one whose status the host has not settled yet) must not
  // publish anything: workgroup 0 may have advanced the image's call counter by then, and granules
  // tagged counter + 1 would be taken for the NEXT call's.  The call's outputs are invalid either
  // way; the host re-zeroes the status block before it submits again (hip_engine.py: settle_exchange).
  // (ONE lane reads the word and the workgroup decides together: per-thread loads could straddle a
  //  peer's atomicOr - part of the workgroup gone, the rest publishing under the tag.  smem[M::VEC]
  //  is free here; the barrier behind the read also fences it against the first stage's writes.)
  if (threadIdx.x == 0)
    smem[M::VEC] = __builtin_bit_cast(float, __hip_atomic_load(p.flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __syncthreads();
  const bool late = (__builtin_bit_cast(uint32_t, smem[M::VEC]) & FLAG_EXCHANGE) != 0;
  __syncthreads();
  if (late) return;
  const unsigned tag = __hip_atomic_load(p.xch_epoch + img, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;   // never 0; workgroup 0 stores it back at the end
  bool dead = false;

  f32x4 wa[16], wb[16];   // two stages of weights in flight
  issue_rows<64, 0>(w0.cross.wm_t, j, tid, wa);      // S1
  issue_cols<128, FF, 0>(w0.w1_t, j, tid, wb);       // S2
  PHASE_STAMP(p, 0);

  // ---- this quarter of the image's tile partials: two heads of the layer-1 memory state
  // -> LDS, 64 columns of the layer-0 partial messages -> att
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(p.dkv1) + (size_t)slot0 * (KV_FLOATS / 4) + j * T + tid;
    const int c = 64 * j + (tid & 63), hh = c >> 5;
    f32x4 sacc = zero4;
    float ks = 0.f, a0 = 0.f, z0 = 0.f;
    constexpr int R = 8;   // tiles per round trip
    for (int ti0 = 0; ti0 < nts; ti0 += R) {
      f32x4 a[R];
      float kt[R], av[R], zv[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const size_t ti = min(ti0 + u, nts - 1);
        a[u] = src[ti * (KV_FLOATS / 4)];
        if (tid < 64) {
          kt[u] = p.dks1[(slot0 + ti) * C + c];
          av[u] = p.att0_part[(slot0 + ti) * C + c];
          zv[u] = p.z0_part[(slot0 + ti) * NH + hh];
        }
      }
#pragma unroll
      for (int u = 0; u < R; ++u)
        if (ti0 + u < nts) {
          sacc += a[u];
          if (tid < 64) { ks += kt[u]; a0 += av[u]; z0 += zv[u]; }
        }
    }
    {
      const int ln = tid & 63, q = (tid >> 6) & 3, hl = tid >> 8;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) kv_s[(hl * HD + (jj + 8 * q + 4 * (ln >> 5))) * HD + (ln & 31)] = sacc[jj];
    }
    if (tid < 64) {
      ksum[tid] = ks;
      att[tid] = a0 * (1.0f / (z0 + ATTN_EPS)) * (float)L;
    }
    if (tid < C) {
      tgt[tid] = p.tgt1[side * C + tid];
      qe[tid] = p.qe[side][tid];
    }
  }
  __syncthreads();

  PHASE_STAMP(p, 1);
  // S1: tgt += Wm_c0 . att0                                   (rows: this head pair's att0)
  fma_rows<64, 0>(wa, att, part_s, tid);
  issue_rows<128, 0>(w0.w2_t, j, tid, wa);            // S3
  __syncthreads();
  if (tid < C) tgt[tid] += exchange_sum<false>(p, img, 0, j, tag, collect<8, C>(part_s, tid), tid, dead);
  __syncthreads();
  PHASE_STAMP(p, 2);
  // S2: hdn = relu(W1_0 . LN3(tgt))                           (columns: this FFN slab)
  ln_vec(tgt, w0.n3w, w0.n3b, t2, nullptr, nullptr, tid);
  fma_cols<128, 0>(wb, t2, part_s, tid);
  issue_cols<64, C, 0>(w1.self_attn.wq_t, j, tid, wb);   // S4 q, k
  issue_cols<64, C, 8>(w1.self_attn.wk_t, j, tid, wb);
  __syncthreads();
  if (tid < 128) hdn_s[tid] = fmaxf(collect<16, 128>(part_s, tid), 0.f);
  __syncthreads();
  PHASE_STAMP(p, 3);
  // S3: tgt += W2_0 . hdn                                     (rows: this FFN slab)
  fma_rows<128, 0>(wa, hdn_s, part_s, tid);
  issue_cols<64, C, 0>(w1.self_attn.wv_t, j, tid, wa);   // S4 v
  issue_rows<64, 8>(w1.self_attn.wm_t, j, tid, wa);      // S5
  __syncthreads();
  if (tid < C) tgt[tid] += exchange_sum<false>(p, img, 1, j, tag, collect<8, C>(part_s, tid), tid, dead);
  __syncthreads();

  PHASE_STAMP(p, 4);
  // S4: layer 1 self-attention, this head pair: q | k | v from LN1(tgt)
  ln_vec(tgt, w1.n1w, w1.n1b, t2, nullptr, nullptr, tid);
  fma_cols<64, 0>(wb, t2, part_s, tid);
  fma_cols<64, 8>(wb, t2, part_s + 2048, tid);
  issue_cols<64, C, 0>(w1.cross.wq_t, j, tid, wb);       // S6
  issue_rows<64, 8>(w1.cross.wm_t, j, tid, wb);          // S7
  fma_cols<64, 0>(wa, t2, part_s + 4096, tid);
  __syncthreads();
  if (tid < 192) {
    const int m = tid >> 6, o = tid & 63;
    const float v = collect<32, 64>(part_s + 2048 * m, o) + p.qkv1[side * 3 * C + m * C + 64 * j + o];
    qkv_s[tid] = m < 2 ? elu1(v) : v;   // phi(q), phi(k) once per element; v as is
  }
  __syncthreads();
  if (tid < 64) {
    // L = S = 1 linear attention (values / v_length with v_length = 1)
    const int hl = tid >> 5;
    const float vval = qkv_s[128 + tid] / 1.0f;
    float z = 0.f, sa = 0.f;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) {
      const float fq = qkv_s[hl * HD + d], fk = qkv_s[64 + hl * HD + d];
      z += fq * fk;
      sa += fq * (fk * vval);
    }
    att[tid] = sa * (1.0f / (z + ATTN_EPS)) * 1.0f;
  }
  __syncthreads();
  PHASE_STAMP(p, 5);
  // S5: tgt += Wm_s1 . att
  fma_rows<64, 8>(wa, att, part_s, tid);
  issue_cols<128, FF, 0>(w1.w1_t, j, tid, wa);           // S8
  __syncthreads();
  if (tid < C) tgt[tid] += exchange_sum<false>(p, img, 2, j, tag, collect<8, C>(part_s, tid), tid, dead);
  __syncthreads();
  PHASE_STAMP(p, 6);
  // S6: layer 1 cross-attention, this head pair
  ln_vec(tgt, w1.n2w, w1.n2b, t2, qe, qk, tid);
  fma_cols<64, 0>(wb, qk, part_s, tid);
  __syncthreads();
  if (tid < 64) vq[tid] = elu1(collect<32, 64>(part_s, tid) + w1.cross.bq[64 * j + tid]);
  __syncthreads();
  if (tid < 64) {
    const int hl = tid >> 5, v = tid & 31;
    float z = 0.f, sacc = 0.f;
#pragma unroll 4
    for (int d = 0; d < HD; ++d) {
      const float fq = vq[hl * HD + d];
      z += fq * ksum[hl * HD + d];
      sacc += fq * kv_s[(hl * HD + d) * HD + v];
    }
    att[tid] = sacc * (1.0f / (z + ATTN_EPS)) * (float)L;
  }
  __syncthreads();
  PHASE_STAMP(p, 7);
  // S7: tgt += Wm_c1 . att
  fma_rows<64, 8>(wb, att, part_s, tid);
  issue_rows<128, 0>(w1.w2_t, j, tid, wb);               // S9
  __syncthreads();
  if (tid < C) tgt[tid] += exchange_sum<false>(p, img, 3, j, tag, collect<8, C>(part_s, tid), tid, dead);
  __syncthreads();
  PHASE_STAMP(p, 8);
  // S8/S9: ReLU MLP
  ln_vec(tgt, w1.n3w, w1.n3b, t2, nullptr, nullptr, tid);
  fma_cols<128, 0>(wa, t2, part_s, tid);
  __syncthreads();
  if (tid < 128) hdn_s[tid] = fmaxf(collect<16, 128>(part_s, tid), 0.f);
  __syncthreads();
  fma_rows<128, 0>(wb, hdn_s, part_s, tid);
  __syncthreads();
  if (tid < C) {
    const float s = exchange_sum<true>(p, img, 4, j, tag, collect<8, C>(part_s, tid), tid, dead);
    if (j == 0) {
      p.hs[(size_t)img * C + tid] = tgt[tid] + s;
      if (tid == 0) __hip_atomic_store(p.xch_epoch + img, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  PHASE_STAMP(p, 9);
}

#ifndef OETR_DEC_THREADS
#define OETR_DEC_THREADS 512   // measured equal to 1024 (the chain is L1-rate bound)
#endif
template <int T>
__global__ __launch_bounds__(T) void k_decoder(DecLaunch p) {
  __shared__ __attribute__((aligned(16))) float smem[DecSmem<T>::TOTAL];
  decoder_body<T>(p, blockIdx.x, smem);
}
__global__ __launch_bounds__(512) void k_decoder4(DecLaunch p) {
  __shared__ __attribute__((aligned(16))) float smem[DecSmem4::TOTAL];
  decoder_body4(p, blockIdx.x / DEC_K, blockIdx.x % DEC_K, smem);
}

hipError_t launch_decoder(const DecLaunch& p, hipStream_t s) {
  if (p.ksplit == DEC_K)
    hipLaunchKernelGGL(k_decoder4, dim3(2 * p.g.N * DEC_K), dim3(512), 0, s, p);
  else
    hipLaunchKernelGGL(k_decoder<OETR_DEC_THREADS>, dim3(2 * p.g.N), dim3(OETR_DEC_THREADS), 0, s, p);
  return hipGetLastError();
}

// Decoder (blocks [0, 2N * ksplit)) and conv P tiles (the blocks behind them) in one
// launch: the decoder occupies 2N (x 4) CUs while the P GEMMs fill the rest of the chip;
// decoder blocks come first in the grid so they are dispatched first.
template <int MODE, bool T64>
__global__ __launch_bounds__(512) void k_decoder_convp(DecLaunch d, HeatLaunch h, float* P) {
  constexpr int TILE = T64 ? R_FLOATS : TILE_FLOATS;
  static_assert(DecSmem4::TOTAL <= DecSmem<512>::TOTAL, "decoder LDS");
  constexpr int LDS_FLOATS = DecSmem<512>::TOTAL > TILE ? DecSmem<512>::TOTAL : TILE;
  __shared__ __attribute__((aligned(16))) float smem[LDS_FLOATS];
  const int nd = 2 * d.g.N * d.ksplit;
#ifdef OETR_ROLE_ABL   // timing experiments only: 1 = decoder workgroups only, 2 = conv-P only
  if ((OETR_ROLE_ABL == 1) != ((int)blockIdx.x < nd)) return;
#endif
  if ((int)blockIdx.x < nd) {
    if (d.ksplit == DEC_K) decoder_body4(d, blockIdx.x / DEC_K, blockIdx.x % DEC_K, smem);
    else decoder_body<512>(d, blockIdx.x, smem);
  } else if constexpr (T64) conv_p_body64<MODE>(h, P, blockIdx.x - nd, smem);
  else conv_p_body<MODE>(h, P, blockIdx.x - nd, smem);
}

template <int MODE>
static hipError_t launch_decoder_convp_mode(const DecLaunch& d, const HeatLaunch& h0, float* P,
                                            hipStream_t s) {
  // conv-P work items: in the 16-bit-plane modes 64-token tiles (x convp_split tap groups)
  // whatever tile the encoder runs - P is indexed by row; fewer, longer workgroups leave the
  // one-workgroup decoder chain, then this launch's critical path, more of the L2 (52.0 vs
  // 53.5 us) - else TM-token tiles (h.g).  Beside the split decoder: items of d.convp_units (tile, tap) units (conv_p.h).
  // (round 1's rule - 64-token conv tiles only when the encoder ran 64-token tiles - and its 32-row conv body
  //  for the 16-bit-plane modes are gone: that instantiation was never launched and was the one conv kernel
  //  that spilled, 20 B of scratch per lane)
  HeatLaunch h = h0;
  constexpr bool t64 = gm_half(MODE);
  h.convp_units = t64 && d.ksplit == DEC_K ? d.convp_units : 9;
  const int units = 9 * h.g.N * ((h.g.L[0] + RT - 1) / RT + (h.g.L[1] + RT - 1) / RT);
  const int ptiles = t64 ? (units + h.convp_units - 1) / h.convp_units : h.g.ntiles;
  const dim3 grid(2 * d.g.N * d.ksplit + ptiles);
  hipLaunchKernelGGL((k_decoder_convp<MODE, t64>), grid, dim3(512), 0, s, d, h, P);
  return hipGetLastError();
}

hipError_t launch_decoder_convp(const DecLaunch& d, const HeatLaunch& h, float* P, int mode,
                                hipStream_t s) {
  switch (mode) {
    case GM_F32: return launch_decoder_convp_mode<GM_F32>(d, h, P, s);
    case GM_SPLIT: return launch_decoder_convp_mode<GM_SPLIT>(d, h, P, s);
    case GM_F16: return launch_decoder_convp_mode<GM_F16>(d, h, P, s);
    case GM_BF16: return launch_decoder_convp_mode<GM_BF16>(d, h, P, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace oetr
