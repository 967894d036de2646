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
// Work items (round 5): the launch's work is the list of (tile, tap) UNITS, tile-major - 9 per 64-token
// tile - cut into items of p.convp_units consecutive units; an item that crosses a tile boundary
// stages the next tile's rows and goes on (the weight stream runs ahead from unit to unit either way).
// 9 = one item per tile (beside the one-workgroup-per-image decoder, whose ~50-us chain is that
// launch's critical path: more, shorter items only slow it down, 53.5 -> 57.4 us); 3 = round 4's three
// items per tile; in between: the host sizes the items so that decoder workgroups + items fill the
// chip in ONE round (api.hip: dec_launch) - at 8 pairs @640x640 1 008 units on 192 free CUs = 6 per item,
// where three per tile took two rounds (336 items): k_decoder_convp 41.6 -> see profiles/r5_tail_items.txt.
template <int MODE>
__device__ __forceinline__ void conv_p_body64(const HeatLaunch& p, float* __restrict__ P, int item,
                                              float* smem) {
  static_assert(16 % WStream2T<MODE>::D == 0, "tap loop below assumes a ring phase of 0 after every GEMM");
  constexpr int THREADS = 512, TPR = THREADS / RT, F4 = 64 / TPR;
  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int nt0 = (g.L[0] + RT - 1) / RT, nt1 = (g.L[1] + RT - 1) / RT;
  const int per = nt0 + nt1, ntiles = g.N * per;
  const int u0 = item * p.convp_units, u1 = min(u0 + p.convp_units, 9 * ntiles);
  if (u0 >= u1) return;

  Range rg;
  const PlanesT<MODE> A(smem, &rg);
  WStream2T<MODE> ws;
  ws.set_lane(lane);
  constexpr size_t TAP_UNITS = (size_t)C * C / 8;
  {
    const int tap0 = u0 % 9;
    ws.template prime<C, 0>(p.w.conv_w + tap0 * TAP_UNITS, p.w.conv_w_l + tap0 * TAP_UNITS, wave, 0, lane);
  }
  int cur_tile = -1, nv2 = 0;
  float* dst0 = nullptr;
#pragma unroll 1
  for (int u = u0; u < u1; ++u) {
    const int tile = u / 9, tap = u - 9 * tile;
    if (tile != cur_tile) {   // (workgroup-uniform) stage this tile's rows: the A operand of its taps
      cur_tile = tile;
      const int logical = xcd_remap(tile, ntiles);
      const int n = logical / per;
      const int rem = logical - n * per;
      const int side = rem >= nt0;
      const int t_idx = side ? rem - nt0 : rem;
      const int L = g.L[side];
      const int l0 = t_idx * RT;
      const int nvalid = min(RT, L - l0);
      const float* mem = p.mem[side] + ((size_t)n * L + l0) * C;
      const size_t row_base = (size_t)g.row0[side] + (size_t)n * L + l0;
      __syncthreads();          // every wave has left the previous tile's planes
      {
        const int r = tid / TPR, part = tid % TPR;
        const f32x4* mp = reinterpret_cast<const f32x4*>(mem + (size_t)min(r, nvalid - 1) * C) + part;
#pragma unroll
        for (int i = 0; i < F4; ++i) A.put4(r, 4 * (i * TPR + part), mp[i * TPR]);
      }
      ws.set_rows(nvalid);
      dst0 = P + row_base * C + 32 * wave + col + (size_t)4 * half * C;
      nv2 = nvalid - 4 * half;
      __syncthreads();
    }
    f32x16 acc[2] = {f32x16{0}, f32x16{0}};
    const f32x4* w = p.w.conv_w + tap * TAP_UNITS;
    const f32x4* wl = p.w.conv_w_l + tap * TAP_UNITS;
    if (u + 1 < u1) {           // the next unit's weights (tap + 1, or tap 0 of the next tile) are primed meanwhile
      const int ntap = tap == 8 ? 0 : tap + 1;
      ws.template gemm<C, 0, true, C>(A, w, wl, wave, 0, lane, acc, p.w.conv_w + ntap * TAP_UNITS,
                                      p.w.conv_w_l + ntap * TAP_UNITS, wave, 0);
    } else {
      ws.template gemm<C, 0, false, C>(A, w, wl, wave, 0, lane, acc, nullptr, nullptr, 0, 0);
    }
    float* dst = dst0 + (size_t)tap * g.rows * C;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * mt + crow(r, 0);
        if (row < nv2) dst[(size_t)row * C] = acc[mt][r];
      }
  }
  range_report<MODE>(rg, p.flags);
}

}  // namespace oetr
