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
#include "common.h"

namespace oetr {

// ---------------------------------------------------------------------------
// NCHW -> token-major transpose of the feature maps and position tables.
// grid.x = (2N + 2) images * ceil(L/64) * 4 channel chunks.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep_tokens(Geom g, const float* __restrict__ feat1,
                                                     const float* __restrict__ feat2,
                                                     const float* __restrict__ pos1,
                                                     const float* __restrict__ pos2,
                                                     float* __restrict__ x,
                                                     float* __restrict__ pos_tok) {
  __shared__ float tile[64][65];
  // decode block -> (image, l-chunk, c-chunk)
  const int lch0 = (g.L[0] + 63) / 64, lch1 = (g.L[1] + 63) / 64;
  const int per0 = lch0 * 4, per1 = lch1 * 4;
  int b = blockIdx.x;
  int side, img;  // img: 0..N-1 features, N = position table
  const int side0_blocks = (g.N + 1) * per0;
  if (b < side0_blocks) { side = 0; img = b / per0; b -= img * per0; }
  else { b -= side0_blocks; side = 1; img = b / per1; b -= img * per1; }
  const int L = g.L[side];
  const int lc = b >> 2, cc = b & 3;
  const float* src;
  float* dst;
  if (img < g.N) {
    src = (side ? feat2 : feat1) + (size_t)img * C * L;
    dst = x + (size_t)(g.row0[side] + img * L) * C;
  } else {
    src = side ? pos2 : pos1;
    dst = pos_tok + (size_t)g.prow0[side] * C;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int l = lc * 64 + tx;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = cc * 64 + ty * 16 + i;
    tile[ty * 16 + i][tx] = (l < L) ? src[(size_t)c * L + l] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ll = lc * 64 + ty * 16 + i;
    if (ll < L) dst[(size_t)ll * C + cc * 64 + tx] = tile[tx][ty * 16 + i];
  }
}

hipError_t launch_prep_tokens(const Geom& g, const float* feat1, const float* feat2,
                              const float* pos1, const float* pos2, float* x,
                              float* pos_tok, hipStream_t s) {
  const int lch0 = (g.L[0] + 63) / 64, lch1 = (g.L[1] + 63) / 64;
  const int blocks = (g.N + 1) * 4 * (lch0 + lch1);
  hipLaunchKernelGGL(k_prep_tokens, dim3(blocks), dim3(256), 0, s, g, feat1, feat2, pos1,
                     pos2, x, pos_tok);
  return hipGetLastError();
}

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
constexpr int SMEM_FLOATS = Z_OFF + TM * NH;

// LayerNorm over a [TM][256] LDS tile with 8 threads per row: thread tid owns
// row tid>>3 and the float4 columns i*8 + (tid&7), i < 8 (so the 8 threads of a
// row read 128 contiguous bytes per step).  Row sums need only three DPP
// exchanges inside an 8-lane group.  Returns the normalised values
// (x - mean) * rstd in registers; the caller applies its affine(s).
__device__ __forceinline__ void ln_rows8(const float* S, int tid, f32x4 (&xn)[8], int dbg) {
  const f32x4* src = reinterpret_cast<const f32x4*>(S + (tid >> 3) * LDA) + (tid & 7);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    xn[i] = src[i * 8];
    s += (xn[i][0] + xn[i][1]) + (xn[i][2] + xn[i][3]);
  }
  if (ABL(dbg, ABL_LN)) return;
  const float mean = sum8(s) * (1.0f / C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    xn[i] -= mean;
    q += (xn[i][0] * xn[i][0] + xn[i][1] * xn[i][1]) + (xn[i][2] * xn[i][2] + xn[i][3] * xn[i][3]);
  }
  const float rstd = 1.0f / sqrtf(sum8(q) * (1.0f / C) + LN_EPS);
#pragma unroll
  for (int i = 0; i < 8; ++i) xn[i] *= rstd;
}

// phi(K)^T (V/S) for this wave's two heads from the K and V accumulators, plus
// sum_s phi(K); stores the per-tile partial states.
__device__ __forceinline__ void kv_state_store(f32x16 (&accK)[2], f32x16 (&accV)[2],
                                               float inv_len_is_div /* ablation: skip phi */, int S_len, int nvalid,
                                               int lane, int wave, float* __restrict__ kv_out,
                                               float* __restrict__ ks_out, int slot) {
  const int half = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float ksum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = crow(r, half) < nvalid;
      const float kf = ok ? (inv_len_is_div != 0.f ? accK[t][r] : elu1(accK[t][r])) : 0.f;
      const float vf = ok ? accV[t][r] / (float)S_len : 0.f;
      accK[t][r] = kf;
      accV[t][r] = vf;
      ksum += kf;
    }
    f32x16 kv = {0};
#pragma unroll
    for (int r = 0; r < 16; ++r)
      kv = __builtin_amdgcn_mfma_f32_32x32x2f32(accK[t][r], accV[t][r], kv, 0, 0, 0);
    const int h = 2 * wave + t;
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

template <bool HAS_B, int TAIL, bool SPLIT>
__global__ __launch_bounds__(NTHREADS) void k_encoder(EncLaunch p) {
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  float* S0 = smem + S0_OFF;
  const ATile<SPLIT> S1(smem + S1_OFF, LDA, LDAH);
  const ATile<SPLIT> S2(smem + H_OFF, LDA, LDAH);
  const ATile<SPLIT> Hh(smem + H_OFF, LDH, LDHH);
  float* ksum_s = smem + KSUM_OFF;
  float* z_s = smem + Z_OFF;

  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int col = lane & 31;

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

  f32x16 xacc[2];  // residual stream of this wave's 64 columns (C layout)

  if (HAS_B) {
    // ================= phase B: finish layer l =================
    const int ss = p.b_cross ? 1 - side : side;
    const int S_len = g.L[ss];
    const int nts = ABL(p.dbg, ABL_KVREDUCE) ? 1 : g.nt[ss];
    const int src_slot0 = g.tile0[ss] + n * g.nt[ss];

    // Loads below never branch on row validity: rows past the end of the image
    // re-read the last valid row (finite values that are never stored), because a
    // guarded load costs a branch plus a full vmcnt(0) round trip per element.
    // phi(Q) tile -> S0 (coalesced float4 rows)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + NTHREADS * i;
      const int r = idx >> 6, c4 = idx & 63;
      const f32x4 v =
          reinterpret_cast<const f32x4*>(p.qp + (row_base + min(r, nvalid - 1)) * C)[c4];
      *reinterpret_cast<f32x4*>(S0 + r * LDA + 4 * c4) = v;
    }
    // residual x in accumulator layout
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(crow(r, half), nvalid - 1);
        xacc[t][r] = p.x[(row_base + row) * C + 64 * wave + 32 * t + col];
      }
    // reduce the source image's partial KV states (fixed order -> deterministic);
    // the result is already in B-operand register order.  Four tiles (32 float4
    // loads per lane) are in flight per round trip.
    f32x4 kvB[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) kvB[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      const f32x4* kvp =
          reinterpret_cast<const f32x4*>(p.kv_in) + ((size_t)src_slot0 * NH + 2 * wave) * 256 + lane;
      const float* ksp = p.ks_in + (size_t)src_slot0 * C + tid;
      float ks = 0.f;
      for (int ti0 = 0; ti0 < nts; ti0 += 4) {
        f32x4 tmp[4][8];
        float kt[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ti = min(ti0 + u, nts - 1);
#pragma unroll
          for (int e = 0; e < 8; ++e) tmp[u][e] = kvp[(size_t)ti * (NH * 256) + e * 64];
          kt[u] = ksp[(size_t)ti * C];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ti0 + u < nts) {
#pragma unroll
            for (int e = 0; e < 8; ++e) kvB[e >> 2][e & 3] += tmp[u][e];
            ks += kt[u];
          }
      }
      ksum_s[tid] = ks;
    }
    __syncthreads();

    // Z[row][h] = 1 / (phi(Q)[row,h,:] . Ksum[h,:] + eps)
    {
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

    // message = (phi(Q) . KV) * Z * S  for this wave's two heads -> S1
    {
      float zr[2][16];  // read before any S1 store: LDS stores would otherwise
#pragma unroll          // serialise these reads one by one (may-alias)
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) zr[t][r] = z_s[crow(r, half) * NH + 2 * wave + t];
      f32x16 macc[2] = {{0}, {0}};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(S0 + col * LDA + (2 * wave + t) * HD +
                                                          4 * half + ks * 8);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            macc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], kvB[t][ks][j], macc[t], 0, 0, 0);
        }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) macc[t][r] = macc[t][r] * zr[t][r] * (float)S_len;
      S1.template put_acc<2>(64 * wave, lane, macc);
    }
    __syncthreads();

    // x1 = x + message . Wmerge^T
    S1.template gemm<C, 2>(p.b.wmerge, p.b.wmerge_l, 2 * wave, lane, xacc, p.dbg);
    acc_to_lds<2>(S0, LDA, 64 * wave, lane, xacc);
    __syncthreads();

    // LN2(x1) -> S1
    {
      f32x4 xn[8];
      ln_rows8(S0, tid, xn, p.dbg);
      const f32x4* gw = reinterpret_cast<const f32x4*>(p.b.ln2_w) + (tid & 7);
      const f32x4* gb = reinterpret_cast<const f32x4*>(p.b.ln2_b) + (tid & 7);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        S1.put4(tid >> 3, 4 * (i * 8 + (tid & 7)), xn[i] * gw[i * 8] + gb[i * 8]);
    }
    __syncthreads();

    // hidden = gelu(LN2(x1) . W1^T) -> Hh   (wave w: hidden columns [128w, 128w+128))
#pragma unroll
    for (int cpart = 0; cpart < 2; ++cpart) {
      f32x16 hacc[2] = {{0}, {0}};
      S1.template gemm<C, 2>(p.b.w1, p.b.w1_l, 4 * wave + 2 * cpart, lane, hacc, p.dbg);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[t][r] = ABL(p.dbg, ABL_GELU) ? hacc[t][r] : gelu_erf(hacc[t][r]);
      Hh.template put_acc<2>(128 * wave + 64 * cpart, lane, hacc);
    }
    __syncthreads();

    // x2 = x1 + hidden . W2^T ; write back
    Hh.template gemm<FF, 2>(p.b.w2, p.b.w2_l, 2 * wave, lane, xacc, p.dbg);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = crow(r, half);
        if (row < nvalid) p.x[(row_base + row) * C + 64 * wave + 32 * t + col] = xacc[t][r];
      }
    if (TAIL != 2) acc_to_lds<2>(S0, LDA, 64 * wave, lane, xacc);
    __syncthreads();  // also: every wave is done reading Hh before S2 (alias) is written
  } else {
    // first launch: x tile straight from HBM
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + NTHREADS * i;
      const int r = idx >> 6, c4 = idx & 63;
      const f32x4 v = reinterpret_cast<const f32x4*>(p.x + (row_base + min(r, nvalid - 1)) * C)[c4];
      *reinterpret_cast<f32x4*>(S0 + r * LDA + 4 * c4) = v;
    }
    __syncthreads();
  }

  if (TAIL == 0) {
    // ================= phase A: start layer l+1 =================
    // q_in = LN_q(x)+pos -> S1 ; kv_in = LN_kv(x)+pos -> S2 (one set of row stats)
    {
      f32x4 xn[8];
      ln_rows8(S0, tid, xn, p.dbg);
      const int r = tid >> 3, part = tid & 7;
      const f32x4* pos = reinterpret_cast<const f32x4*>(
                             p.pos + (size_t)(g.prow0[side] + l0 + min(r, nvalid - 1)) * C) + part;
      const f32x4* qw = reinterpret_cast<const f32x4*>(p.a.lnq_w) + part;
      const f32x4* qb = reinterpret_cast<const f32x4*>(p.a.lnq_b) + part;
      const f32x4* kw = reinterpret_cast<const f32x4*>(p.a.lnkv_w) + part;
      const f32x4* kb = reinterpret_cast<const f32x4*>(p.a.lnkv_b) + part;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 ps = pos[i * 8];
        S1.put4(r, 4 * (i * 8 + part), (xn[i] * qw[i * 8] + qb[i * 8]) + ps);
        S2.put4(r, 4 * (i * 8 + part), (xn[i] * kw[i * 8] + kb[i * 8]) + ps);
      }
    }
    __syncthreads();

    {  // phi(Q) -> HBM
      f32x16 acc[2] = {{0}, {0}};
      S1.template gemm<C, 2>(p.a.wq, p.a.wq_l, 2 * wave, lane, acc, p.dbg);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = crow(r, half);
          if (row < nvalid)
            p.qp[(row_base + row) * C + 64 * wave + 32 * t + col] = ABL(p.dbg, ABL_ELU) ? acc[t][r] : elu1(acc[t][r]);
        }
    }
    f32x16 accK[2] = {{0}, {0}}, accV[2] = {{0}, {0}};
    S2.template gemm<C, 2>(p.a.wk, p.a.wk_l, 2 * wave, lane, accK, p.dbg);
    S2.template gemm<C, 2>(p.a.wv, p.a.wv_l, 2 * wave, lane, accV, p.dbg);
    kv_state_store(accK, accV, ABL(p.dbg, ABL_ELU) ? 1.f : 0.f, L, nvalid, lane, wave, p.kv_out, p.ks_out, slot);
  } else if (TAIL == 1) {
    // ============ decoder preparation (transformer.py:240-246) ============
    // k = (memory + pos) Wk^T + bk ; v = memory Wv^T + bv  (no norm, no pos on v)
    {
      const int r = tid >> 3, part = tid & 7;
      const f32x4* pos = reinterpret_cast<const f32x4*>(
                             p.pos + (size_t)(g.prow0[side] + l0 + min(r, nvalid - 1)) * C) + part;
      const f32x4* src = reinterpret_cast<const f32x4*>(S0 + r * LDA) + part;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 xv = src[i * 8];
        S1.put4(r, 4 * (i * 8 + part), xv + pos[i * 8]);   // k input: memory + pos
        if (SPLIT) S2.put4(r, 4 * (i * 8 + part), xv);     // v input: memory (f32 mode reads S0)
      }
    }
    __syncthreads();
#pragma unroll
    for (int dl = 0; dl < 2; ++dl) {
      f32x16 accK[2], accV[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float bk = p.d.bk[dl][64 * wave + 32 * t + col];
        const float bv = p.d.bv[dl][64 * wave + 32 * t + col];
#pragma unroll
        for (int r = 0; r < 16; ++r) { accK[t][r] = bk; accV[t][r] = bv; }
      }
      S1.template gemm<C, 2>(p.d.wk[dl], p.d.wk_l[dl], 2 * wave, lane, accK, p.dbg);
      if (SPLIT) S2.template gemm<C, 2>(p.d.wv[dl], p.d.wv_l[dl], 2 * wave, lane, accV, p.dbg);
      else gemm_rows32<C, 2>(S0, LDA, p.d.wv[dl], 2 * wave, lane, accV, p.dbg);
      if (dl == 1) {
        kv_state_store(accK, accV, ABL(p.dbg, ABL_ELU) ? 1.f : 0.f, L, nvalid, lane, wave,
                       p.dkv1_out, p.dks1_out, slot);
      } else {
        // Decoder layer 0's query is a create-time constant q0 (decoder.hip), and
        // its cross-attention is linear in the state, so this tile contributes
        //   att0[h,v] += sum_d q0[h,d] * KV_tile[h][d][v],  z0[h] += q0[h,:].Ksum_tile[h,:]
        // instead of a full 8192-float state.
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int h = 2 * wave + t;
          float ksum = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = crow(r, half) < nvalid;
            accK[t][r] = ok ? elu1(accK[t][r]) : 0.f;
            accV[t][r] = ok ? accV[t][r] / (float)L : 0.f;
            ksum += accK[t][r];
          }
          f32x16 kv = {0};
#pragma unroll
          for (int r = 0; r < 16; ++r)
            kv = __builtin_amdgcn_mfma_f32_32x32x2f32(accK[t][r], accV[t][r], kv, 0, 0, 0);
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
    }
  }
}

hipError_t launch_encoder(const EncLaunch& p, bool has_b, int tail, bool split, hipStream_t s) {
  const dim3 grid(p.g.ntiles), block(NTHREADS);
#define OETR_LAUNCH(B, T)                                                           \
  do {                                                                              \
    if (split) hipLaunchKernelGGL((k_encoder<B, T, true>), grid, block, 0, s, p);   \
    else hipLaunchKernelGGL((k_encoder<B, T, false>), grid, block, 0, s, p);        \
  } while (0)
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

}  // namespace oetr
