// Single-query decoder of the OETR QueryTransformer (gfx950).
//
// Reference: TransformerDecoder / DecoderLayer.forward
// (src/models/transformer.py:224-284) with MultiHeadAttention (:55-72) and
// LinearAttention (src/models/linear_attention.py:22-50); tgt starts at zero,
// tgt_pos = query embedding, memory = encoder output (transformer.py:361-381).
//
// There is ONE query token per image, so every projection is a 256-wide
// GEMV; the only token-parallel work (K/V projections of the memory and the
// per-head phi(K)^T V states) was already done by the encoder's tail launch.
// One workgroup of 16 waves handles one image: the chain of GEMVs is latency
// bound, each GEMV puts a whole 256x256 matrix in flight at once (16 k-chunks
// x 64 float4 columns), weights are stored transposed [in][out] so every
// wave-load is a contiguous 1 KiB.
#include "common.h"

namespace oetr {

constexpr int DEC_THREADS = 1024;

// out[NOUT] = x[K] . Wt[K][NOUT] (+ bias), all 1024 threads.
template <int K, int NOUT>
__device__ __forceinline__ void gemv(const float* __restrict__ Wt, const float* x_s,
                                     const float* __restrict__ bias, float* out_s,
                                     float* part_s, int tid, bool relu = false) {
  constexpr int NO4 = NOUT / 4;
  constexpr int KCH = DEC_THREADS / NO4;
  constexpr int KPER = K / KCH;
  const int kc = tid / NO4, o4 = tid % NO4;
  const f32x4* w = reinterpret_cast<const f32x4*>(Wt) + (size_t)kc * KPER * NO4 + o4;
  constexpr int BATCH = KPER < 8 ? KPER : 8;  // float4 loads in flight per thread
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
  __syncthreads();
  for (int o = tid; o < NOUT; o += DEC_THREADS) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < KCH; ++c) s += part_s[c * NOUT + o];
    if (bias) s += bias[o];
    out_s[o] = relu ? fmaxf(s, 0.f) : s;
  }
  __syncthreads();
}

// LayerNorm of one 256-vector (wave 0), optional second output y + add.
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

__global__ __launch_bounds__(DEC_THREADS) void k_decoder(DecLaunch p) {
  __shared__ __attribute__((aligned(16))) float kv_s[KV_FLOATS];  // [h][d][v]
  __shared__ __attribute__((aligned(16))) float part_s[4096];
  __shared__ __attribute__((aligned(16))) float vec_s[10][C];
  __shared__ __attribute__((aligned(16))) float hdn_s[FF];
  float* tgt = vec_s[0];
  float* t2 = vec_s[1];
  float* qk = vec_s[2];
  float* vq = vec_s[3];
  float* vk = vec_s[4];
  float* vv = vec_s[5];
  float* att = vec_s[6];
  float* msg = vec_s[7];
  float* qe = vec_s[8];
  float* ksum = vec_s[9];

  const Geom& g = p.g;
  const int tid = threadIdx.x;
  const int img = blockIdx.x;
  const int side = img >= g.N, n = side ? img - g.N : img;
  const int L = g.L[side], nts = g.nt[side];
  const int slot0 = g.tile0[side] + n * nts;

  if (tid < C) { tgt[tid] = 0.f; qe[tid] = p.qe[side][tid]; }
  __syncthreads();

  for (int dl = 0; dl < 2; ++dl) {
    const DecLayerDev& w = p.layer[dl];
    // reduce this image's cross-attention states for layer dl into LDS
    {
      const f32x4* src = reinterpret_cast<const f32x4*>(p.dkv[dl]) + (size_t)slot0 * (KV_FLOATS / 4);
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int e = tid + DEC_THREADS * e2;  // ((h*4+q)*64 + lane)
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int ti = 0; ti < nts; ++ti) s += src[(size_t)ti * (KV_FLOATS / 4) + e];
        const int ln = e & 63, q = (e >> 6) & 3, h = e >> 8;
        const int v = ln & 31, hf = ln >> 5;
#pragma unroll
        for (int j = 0; j < 4; ++j) kv_s[(h * HD + (j + 8 * q + 4 * hf)) * HD + v] = s[j];
      }
      if (tid < C) {
        float s = 0.f;
        for (int ti = 0; ti < nts; ++ti) s += p.dks[dl][(size_t)(slot0 + ti) * C + tid];
        ksum[tid] = s;
      }
    }
    // ---- self attention on the single query (L = S = 1) ----
    ln_vec(tgt, w.n1w, w.n1b, t2, qe, qk, tid);
    gemv<C, C>(w.self_attn.wq_t, qk, w.self_attn.bq, vq, part_s, tid);
    gemv<C, C>(w.self_attn.wk_t, qk, w.self_attn.bk, vk, part_s, tid);
    gemv<C, C>(w.self_attn.wv_t, t2, w.self_attn.bv, vv, part_s, tid);
    if (tid < C) {
      const int h = tid >> 5;
      float z = 0.f, s = 0.f;
      const float vval = vv[tid] / 1.0f;  // values / v_length, v_length = 1
#pragma unroll 4
      for (int d = 0; d < HD; ++d) {
        const float fq = elu1(vq[h * HD + d]), fk = elu1(vk[h * HD + d]);
        z += fq * fk;
        s += fq * (fk * vval);
      }
      att[tid] = s * (1.0f / (z + ATTN_EPS)) * 1.0f;
    }
    __syncthreads();
    gemv<C, C>(w.self_attn.wm_t, att, nullptr, msg, part_s, tid);
    if (tid < C) tgt[tid] += msg[tid];
    __syncthreads();
    // ---- cross attention against the memory states ----
    ln_vec(tgt, w.n2w, w.n2b, t2, qe, qk, tid);
    gemv<C, C>(w.cross.wq_t, qk, w.cross.bq, vq, part_s, tid);
    if (tid < C) vq[tid] = elu1(vq[tid]);
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
    gemv<C, C>(w.cross.wm_t, att, nullptr, msg, part_s, tid);
    if (tid < C) tgt[tid] += msg[tid];
    __syncthreads();
    // ---- ReLU MLP ----
    ln_vec(tgt, w.n3w, w.n3b, t2, nullptr, nullptr, tid);
    gemv<C, FF>(w.w1_t, t2, nullptr, hdn_s, part_s, tid, true);
    gemv<FF, C>(w.w2_t, hdn_s, nullptr, msg, part_s, tid);
    if (tid < C) tgt[tid] += msg[tid];
    __syncthreads();
  }
  if (tid < C) p.hs[(size_t)img * C + tid] = tgt[tid];
}

hipError_t launch_decoder(const DecLaunch& p, hipStream_t s) {
  hipLaunchKernelGGL(k_decoder, dim3(2 * p.g.N), dim3(DEC_THREADS), 0, s, p);
  return hipGetLastError();
}

}  // namespace oetr
