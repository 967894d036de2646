// P_tap = W_tap . memory for one 32-token tile (device body, 512 threads): the
// hs-independent part of the heat-map 3x3 conv (reference src/model.py:65-77,
// :152-164).  conv(memory * att)[l] = b + sum_tap att[l+tap] * P_tap[l+tap]
// because att is a per-token scalar; the combine is k_heat_combine (heads.hip).
#pragma once
#include "common.h"

namespace oetr {

template <int MODE>
__device__ __forceinline__ void conv_p_body(const HeatLaunch& p, float* __restrict__ P, int tile,
                                            float* smem) {
  constexpr int NW = 8, NT = 1, THREADS = 64 * NW, WC = 32 * NT;
  constexpr int TPR = THREADS / TM, F4 = 64 / TPR;
  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
  const int col = lane & 31;
  const int logical = xcd_remap(tile, g.ntiles);
  const int per = g.nt[0] + g.nt[1];
  const int n = logical / per;
  const int rem = logical - n * per;
  const int side = rem >= g.nt[0];
  const int t_idx = side ? rem - g.nt[0] : rem;
  const int L = g.L[side];
  const int l0 = t_idx * TM;
  const int nvalid = min(TM, L - l0);
  const float* mem = p.mem[side] + ((size_t)n * L + l0) * C;
  const size_t row_base = (size_t)g.row0[side] + (size_t)n * L + l0;

  // the tile's own rows, once (rows past the image end re-read the last valid row)
  Range rg;
  const ATile<MODE> A(smem, LDA, LDAH, &rg);
  {
    const int r = tid / TPR, part = tid % TPR;
    const f32x4* mp = reinterpret_cast<const f32x4*>(mem + (size_t)min(r, nvalid - 1) * C) + part;
#pragma unroll
    for (int i = 0; i < F4; ++i) A.put4(r, 4 * (i * TPR + part), mp[i * TPR]);
  }
  __syncthreads();
  constexpr size_t TAP_UNITS = gm_half(MODE) ? (size_t)C * C / 8 : (size_t)C * C / 4;
  for (int tap = 0; tap < 9; ++tap) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
    A.template gemm<C, NT>(p.w.conv_w + tap * TAP_UNITS, p.w.conv_w_l + tap * TAP_UNITS, NT * wave,
                           lane, acc, 0);
    float* dst = P + ((size_t)tap * g.rows + row_base) * C + WC * wave + col;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = crow(r, half);
        if (row < nvalid) dst[(size_t)row * C + 32 * t] = acc[t][r];
      }
  }
  range_report<MODE>(rg, p.flags);
}

// 64-token variant (split mode): every weight fragment of the 9 taps (2.3 MB per
// workgroup) feeds two MFMA row tiles, and the weight stream runs ahead from tap to
// tap.  Tiles are 64 consecutive tokens of one image: tile index over
// N x (ceil(L0/64) + ceil(L1/64)), pair-major like the encoder's.
// The nine taps of a tile are shared by p.convp_split = 1 or 3 workgroups (adjacent in the grid:
// the tile's rows are staged by each, from L2).  3 = three times as many, shorter work items: at
// 8 pairs 336 instead of 112.  Measured on MI355X (round 4): beside the ONE-workgroup-per-image
// decoder, whose 50-us chain is that launch's critical path, they only slow it down (53.5 -> 57.4
// us); beside the four-workgroup decoder (25 us) the conv items ARE the critical path, and three
// per tile take the launch from 49.6 to 41.6 us at 8 pairs @640x640, 47 -> 27.5 at 4 pairs, 45.5 ->
// 25.7 at one (nine per tile: 48.7 / 36.8 / 26.6 - the staging of the tile per item).
template <int MODE>
__device__ __forceinline__ void conv_p_body64(const HeatLaunch& p, float* __restrict__ P, int item,
                                              float* smem) {
  static_assert(16 % WStream2T<MODE>::D == 0, "tap loop below assumes a ring phase of 0 after every GEMM");
  const int CONVP_SPLIT = p.convp_split, CONVP_TAPS = 9 / CONVP_SPLIT;
  constexpr int THREADS = 512, TPR = THREADS / RT, F4 = 64 / TPR;
  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int nt0 = (g.L[0] + RT - 1) / RT, nt1 = (g.L[1] + RT - 1) / RT;
  const int tile = item / CONVP_SPLIT, tap0 = (item - tile * CONVP_SPLIT) * CONVP_TAPS;
  const int logical = xcd_remap(tile, g.N * (nt0 + nt1));
  const int per = nt0 + nt1;
  const int n = logical / per;
  const int rem = logical - n * per;
  const int side = rem >= nt0;
  const int t_idx = side ? rem - nt0 : rem;
  const int L = g.L[side];
  const int l0 = t_idx * RT;
  const int nvalid = min(RT, L - l0);
  const float* mem = p.mem[side] + ((size_t)n * L + l0) * C;
  const size_t row_base = (size_t)g.row0[side] + (size_t)n * L + l0;

  Range rg;
  const PlanesT<MODE> A(smem, &rg);
  {
    const int r = tid / TPR, part = tid % TPR;
    const f32x4* mp = reinterpret_cast<const f32x4*>(mem + (size_t)min(r, nvalid - 1) * C) + part;
#pragma unroll
    for (int i = 0; i < F4; ++i) A.put4(r, 4 * (i * TPR + part), mp[i * TPR]);
  }
  WStream2T<MODE> ws;
  ws.set_rows(nvalid);
  ws.set_lane(lane);
  constexpr size_t TAP_UNITS = (size_t)C * C / 8;
  ws.template prime<C, 0>(p.w.conv_w + tap0 * TAP_UNITS, p.w.conv_w_l + tap0 * TAP_UNITS, wave, 0, lane);
  __syncthreads();
  float* dst0 = P + row_base * C + 32 * wave + col + (size_t)4 * half * C;
  const int nv2 = nvalid - 4 * half;
  auto store = [&](int tap, const f32x16 (&acc)[2]) {
    float* dst = dst0 + (size_t)tap * g.rows * C;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * mt + crow(r, 0);
        if (row < nv2) dst[(size_t)row * C] = acc[mt][r];
      }
  };
#pragma unroll 1
  for (int tap = tap0; tap < tap0 + CONVP_TAPS - 1; ++tap) {
    f32x16 acc[2] = {f32x16{0}, f32x16{0}};
    const f32x4* w = p.w.conv_w + tap * TAP_UNITS;
    const f32x4* wl = p.w.conv_w_l + tap * TAP_UNITS;
    ws.template gemm<C, 0, true, C>(A, w, wl, wave, 0, lane, acc, w + TAP_UNITS, wl + TAP_UNITS, wave, 0);
    store(tap, acc);
  }
  {
    constexpr int dummy = 0; (void)dummy;
    const int tap = tap0 + CONVP_TAPS - 1;
    f32x16 acc[2] = {f32x16{0}, f32x16{0}};
    ws.template gemm<C, 0, false, C>(A, p.w.conv_w + tap * TAP_UNITS, p.w.conv_w_l + tap * TAP_UNITS, wave,
                                     0, lane, acc, nullptr, nullptr, 0, 0);
    store(tap, acc);
  }
  range_report<MODE>(rg, p.flags);
}

}  // namespace oetr
