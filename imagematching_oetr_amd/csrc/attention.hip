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
__global__ __launch_bounds__(256) void k_lin_state(const float* __restrict__ k,
                                                   const float* __restrict__ v, int S,
                                                   float* __restrict__ state) {
  __shared__ float ks[64][HD + 1], vs[64][HD + 1];
  const int tid = threadIdx.x, nh = blockIdx.x, n = nh / NH, h = nh % NH;
  const int d = tid >> 3, v0 = (tid & 7) * 4;
  float kv[4] = {0.f, 0.f, 0.f, 0.f}, ksum = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    for (int i = tid; i < 64 * HD; i += 256) {
      const int sl = i >> 5, c = i & 31, s = s0 + sl;
      const size_t off = (((size_t)n * S + s) * NH + h) * HD + c;
      ks[sl][c] = s < S ? elu1(k[off]) : 0.f;
      vs[sl][c] = s < S ? v[off] / (float)S : 0.f;
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
                                                   const float* __restrict__ state, int L, int S,
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
#pragma unroll
  for (int dd = 0; dd < HD; ++dd) { fq[dd] = elu1(q[off + dd]); z += fq[dd] * ksum[dd]; }
  z = 1.0f / (z + ATTN_EPS);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float s = 0.f;
#pragma unroll
    for (int dd = 0; dd < HD; ++dd) s += fq[dd] * kv[dd][v0 + j];
    out[off + v0 + j] = s * z * (float)S;
  }
}

hipError_t launch_linear_attention(const float* q, const float* k, const float* v, int n, int L,
                                   int S, float* out, hipStream_t s) {
  // The stand-alone test entry needs n*8 states of 1056 floats between its two kernels:
  // a stream-ordered temporary (no global state, nothing outlives the call).
  float* state = nullptr;
  const size_t need = (size_t)n * NH * (HD * HD + HD);
  hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&state), need * sizeof(float), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_lin_state, dim3(n * NH), dim3(256), 0, s, k, v, S, state);
  hipLaunchKernelGGL(k_lin_apply, dim3((L + 63) / 64, n * NH), dim3(256), 0, s, q, state, L, S, out);
  e = hipGetLastError();
  const hipError_t e2 = hipFreeAsync(state, s);
  return e != hipSuccess ? e : e2;
}

// -------------------------------------------------------------------- full
// one wave per (n, h, 32 queries); 4 waves per block.
__global__ __launch_bounds__(256) void k_full_attention(const float* __restrict__ q,
                                                        const float* __restrict__ k,
                                                        const float* __restrict__ v, int n_img,
                                                        int L, int S, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
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

}  // namespace oetr
