// Overlap-regression heads of OETR (gfx950).
//
// Reference: OETR.center_estimation (src/model.py:145-186) with heatmap_conv
// (:65-77) and generate_mesh_grid (:103-107); OETR.size_regression (:188-191,
// tlbr_reg :59-63); box_tlbr_to_xyxy (src/models/utils.py:16-28).
//
//  k_heat_conv : att = memory.hs, 3x3 conv of (memory*att) as an implicit GEMM
//                (9 taps x [32 tok x 256] . [256 x 256] on f32 MFMA, the tap
//                tile gathered+scaled on the fly, double-buffered in LDS),
//                + bias, per-tile GroupNorm partial moments.
//  k_heat_logits: one workgroup per 32-token tile: combine the image's GroupNorm
//                moments (Chan), normalise + ReLU + 1x1 conv -> logits.
//  k_heat_final: one workgroup per image: softmax over the image's tokens,
//                soft-argmax -> centre (x, y); forward path: + box from tlbr.
//  size_reg_body / k_size_reg: sigmoid(W2 relu(W1 hs) + b2) - on the forward path as
//                extra workgroups of the launch behind the decoder.
//  k_boxes     : centre -+ extents, clamped to the image.
#include "common.h"

namespace oetr {

constexpr int GN_GROUPS = 32;
constexpr float GN_EPS = 1e-5f;

// ---------------------------------------------------------------------------
// NW waves per workgroup (4 for the exact-f32 mode, 8 for the split mode: see
// encoder.hip); wave w owns NT = 8/NW 32-column tiles of the 256 outputs.
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void k_heat_conv(HeatLaunch p) {
  constexpr int NT = 8 / NW, THREADS = 64 * NW, WC = 32 * NT;
  constexpr int TPR = THREADS / TM, F4 = 64 / TPR;  // threads / float4s per row when staging
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE_FLOATS];
  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5;
  const int col = lane & 31;

  const int logical = xcd_remap(blockIdx.x, g.ntiles);
  const int per = g.nt[0] + g.nt[1];
  const int n = logical / per;
  const int rem = logical - n * per;
  const int side = rem >= g.nt[0];
  const int t_idx = side ? rem - g.nt[0] : rem;
  const int L = g.L[side], hf = g.hf[side], wf = g.wf[side];
  const int l0 = t_idx * TM;
  const int nvalid = min(TM, L - l0);
  const int slot = g.tile0[side] + n * g.nt[side] + t_idx;
  const float* mem = p.mem[side] + (size_t)n * L * C;
  Range rg;
  // att[l'] = memory[l'] . hs for the halo rows l0-wf-1 .. l0+TM+wf of this tile,
  // once (not per tap): TPR threads per row, DPP row sums.
  __shared__ float att_s[TM + 2 * (100 + 1) + 6];
  const int halo0 = l0 - wf - 1, nhalo = TM + 2 * (wf + 1);
  const int hrow = tid / TPR, hpart = tid % TPR;
  {
    const f32x4* hsp = reinterpret_cast<const f32x4*>(p.hs[side] + (size_t)n * C) + hpart;
    f32x4 hv[F4];
#pragma unroll
    for (int i = 0; i < F4; ++i) hv[i] = hsp[i * TPR];
    for (int r0 = 0; r0 < nhalo; r0 += TM) {
      const int hr = r0 + hrow;
      const int l = min(max(halo0 + hr, 0), L - 1);  // clamped: out-of-image rows are masked below
      const f32x4* mp = reinterpret_cast<const f32x4*>(mem + (size_t)l * C) + hpart;
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < F4; ++i) {
        const f32x4 v = mp[i * TPR];
        d += (v[0] * hv[i][0] + v[1] * hv[i][1]) + (v[2] * hv[i][2] + v[3] * hv[i][3]);
      }
      d = sum8(d);
      if (TPR == 16) d += dpp_mov<0x140>(d);
      if (hpart == 0 && hr < nhalo) att_s[hr] = d;
    }
  }
  __syncthreads();
  // The conv input memory*att is the largest-magnitude tensor of the path (att is a
  // 256-term dot product).  conv is linear, so this tile's att weights are divided by a
  // power of two >= max |att| over its halo (exact) and the GEMM result is multiplied
  // back: the operands the f16-based modes convert are then bounded by |memory|.
  __shared__ float attmax_s[THREADS / 64];
  {
    float m = 0.f;
    for (int i = tid; i < nhalo; i += THREADS) m = fmaxf(m, fabsf(att_s[i]));
    m = wave_max(m);
    if (lane == 0) attmax_s[wave] = m;
  }
  __syncthreads();
  float att_scale = 1.0f, att_unscale = 1.0f;
  {
    float m = attmax_s[0];
#pragma unroll
    for (int i = 1; i < THREADS / 64; ++i) m = fmaxf(m, attmax_s[i]);
    if (m > 0.f && m < INFINITY) {
      const int e = ilogbf(m) + 1;          // 2^e > m
      att_scale = ldexpf(1.0f, -e);
      att_unscale = ldexpf(1.0f, e);
    }
  }

  // gather + scale one tap tile (TPR threads per row, float4 columns i*TPR + part)
  auto stage = [&](int tap, const ATile<MODE>& S) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int l = l0 + hrow;
    const int y = l / wf, x = l - y * wf;
    const int yy = y + dy, xx = x + dx;
    const bool ok = (l < L) && (yy >= 0) && (yy < hf) && (xx >= 0) && (xx < wf);
    // unconditional load from a clamped row + select: a guarded load would cost
    // a branch and a vmcnt(0) round trip per row
    const int src_row = ok ? yy * wf + xx : 0;
    const float att = ok ? att_s[src_row - halo0] * att_scale : 0.f;
    const f32x4* mp = reinterpret_cast<const f32x4*>(mem + (size_t)src_row * C) + hpart;
#pragma unroll
    for (int i = 0; i < F4; ++i) S.put4(hrow, 4 * (i * TPR + hpart), mp[i * TPR] * att);
  };

  f32x16 acc[NT];
  float bias[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bias[t] = p.w.conv_b[WC * wave + 32 * t + col];
    acc[t] = f32x16{0};
  }
  // (the two staging tiles are addressed by arithmetic: an array of tile descriptors indexed by
  //  tap & 1 lived in scratch, 112 bytes per lane)
  auto tile = [&](int i) { return ATile<MODE>(smem + (i & 1) * TILE_FLOATS, LDA, LDAH, &rg); };
  // f32 mode: tap t = [256 out][256 in] of 4-byte values -> C*C/4 float4 units;
  // split mode: f16 values -> C*C/8 16-byte units per plane.
  constexpr size_t TAP_UNITS = gm_half(MODE) ? (size_t)C * C / 8 : (size_t)C * C / 4;
  stage(0, tile(0));
  __syncthreads();
  for (int tap = 0; tap < 9; ++tap) {
    if (tap + 1 < 9) stage(tap + 1, tile(tap + 1));
    tile(tap).template gemm<C, NT>(p.w.conv_w + tap * TAP_UNITS, p.w.conv_w_l + tap * TAP_UNITS,
                                      NT * wave, lane, acc, 0);
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(acc[t][r], att_unscale, bias[t]);
  range_report<MODE>(rg, p.flags);

  // conv output + per-(tile, group) moments for GroupNorm
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = WC * wave + 32 * t + col;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = crow(r, half);
      if (row < nvalid) {
        p.conv_out[((size_t)g.row0[side] + (size_t)n * L + l0 + row) * C + c] = acc[t][r];
        s += acc[t][r];
      }
    }
    // 8 channels of a group = lanes with equal (lane&31)>>3, both halves
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)(nvalid * 8);
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (crow(r, half) < nvalid) { const float d = acc[t][r] - mean; m2 += d * d; }
    m2 += __shfl_xor(m2, 1, 64);
    m2 += __shfl_xor(m2, 2, 64);
    m2 += __shfl_xor(m2, 4, 64);
    m2 += __shfl_xor(m2, 32, 64);
    if ((lane & 39) == 0) {  // lane&7 == 0 and half == 0
      float* dst = p.gn_part + ((size_t)slot * GN_GROUPS + (c >> 3)) * 2;
      dst[0] = mean;
      dst[1] = m2;
    }
  }
}

hipError_t launch_heat_conv(const HeatLaunch& p, int mode, hipStream_t s) {
  const dim3 grid(p.g.ntiles);
  switch (mode) {
    case GM_F32: hipLaunchKernelGGL((k_heat_conv<GM_F32, 4>), grid, dim3(256), 0, s, p); break;
    case GM_SPLIT: hipLaunchKernelGGL((k_heat_conv<GM_SPLIT, 8>), grid, dim3(512), 0, s, p); break;
    case GM_F16: hipLaunchKernelGGL((k_heat_conv<GM_F16, 8>), grid, dim3(512), 0, s, p); break;
    case GM_BF16: hipLaunchKernelGGL((k_heat_conv<GM_BF16, 8>), grid, dim3(512), 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// The same conv for 64 token rows per workgroup (two-plane mode; round 4): the DIRECT form of the
// forward path for large batches.  The P form (k_decoder_convp + k_heat_combine) hides the decoder
// chain behind the hs-independent products W_tap.memory, at the price of a 9 x rows x 1 KB buffer
// written and read once: 1.2 GB per step at 32 pairs @1024x1024, where k_heat_combine alone (HBM-bound)
// costs 160 us and the tap stores make k_decoder_convp 285.  Here the decoder runs first (its 50 us on
// 2N CUs are a few percent of such a step), att = memory.hs is known, and the nine taps accumulate in
// registers: per tap the gathered, att-scaled A tile of the NEXT tap is staged (loads issued before
// the GEMM, conversions and plane stores as the GEMM's epilogue slices beside its MFMAs) into the
// other of two plane regions while this tap's GEMM runs; every weight fragment feeds two MFMA row
// tiles (WStream2T), the weight stream runs from tap to tap.  One conv_out tile and the GroupNorm
// moments of its two 32-row slots leave the workgroup - no P.
// Size regression of one image (reference src/model.py:188-191: sigmoid(W2 relu(W1 hs) + b2)), 512
// threads: the forward path runs it as 2N extra workgroups of the launch that follows the decoder
// (k_heat_combine / k_heat_conv64 / k_heat_conv64h) - it needs hs only, and inside k_heat_final it
// was 4.6 us of that one-workgroup-per-image kernel's serial chain.  tlbr -> p.tlbr[side][4 n ..].
__device__ __forceinline__ void size_reg_body(const HeatLaunch& p, int img, float* smem) {
  float *h_s = smem, *hid_part = smem + C, *hid_s = smem + 3 * C;
  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int side = img >= g.N, n = side ? img - g.N : img;
  const int kc = tid >> 8, o = tid & (C - 1);
  const float* wsrc = p.w.tlbr0_t + (size_t)(kc * 128) * C + o;
  float wv[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) wv[k] = wsrc[(size_t)k * C];
  if (tid < C) h_s[tid] = p.hs[side][(size_t)n * C + tid];
  __syncthreads();
  float a = 0.f;
#pragma unroll
  for (int k0 = 0; k0 < 128; k0 += 32) {
    float wn[32];
    if (k0 + 32 < 128) {
#pragma unroll
      for (int k = 0; k < 32; ++k) wn[k] = wsrc[(size_t)(k0 + 32 + k) * C];
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) a += wv[k] * h_s[kc * 128 + k0 + k];
    if (k0 + 32 < 128) {
#pragma unroll
      for (int k = 0; k < 32; ++k) wv[k] = wn[k];
    }
  }
  hid_part[kc * C + o] = a;
  __syncthreads();
  if (tid < C) hid_s[tid] = fmaxf(hid_part[tid] + hid_part[C + tid], 0.f);
  __syncthreads();
  if (wave < 4) {
    const f32x4 w2 = reinterpret_cast<const f32x4*>(p.w.tlbr2_w + wave * C)[lane];
    const f32x4 hv = reinterpret_cast<const f32x4*>(hid_s)[lane];
    const float d = wave_sum((w2[0] * hv[0] + w2[1] * hv[1]) + (w2[2] * hv[2] + w2[3] * hv[3]));
    if (lane == 0) p.tlbr[side][4 * n + wave] = 1.0f / (1.0f + expf(-(d + p.w.tlbr2_b[wave])));
  }
}

// conv output + per-(32-row slot, group) moments for GroupNorm of a 64-row tile's accumulators
__device__ __forceinline__ void heat_conv64_finish(const HeatLaunch& p, const f32x16 (&acc)[2], const Geom& g, int side, int n,
                                                   int t_idx, int nvalid, size_t row_base, int lane, int wave, int half, int col) {
  // conv output + per-(32-row slot, group) moments for GroupNorm (the slots k_heat_final folds)
  const int c = 32 * wave + col;
  const int slot0 = g.tile0[side] + n * g.nt[side] + 2 * t_idx;   // g = the heads' TM-row geometry
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int nv = min(TM, nvalid - 32 * mt);      // valid rows of this 32-row slot (<= 0: none)
    if (nv <= 0) break;                            // (workgroup-uniform)
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = crow(r, half);
      if (row < nv) {
        p.conv_out[(row_base + 32 * mt + row) * C + c] = acc[mt][r];
        sum += acc[mt][r];
      }
    }
    // 8 channels of a group = lanes with equal (lane&31)>>3, both halves
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)(nv * 8);
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (crow(r, half) < nv) { const float d = acc[mt][r] - mean; m2 += d * d; }
    m2 += __shfl_xor(m2, 1, 64);
    m2 += __shfl_xor(m2, 2, 64);
    m2 += __shfl_xor(m2, 4, 64);
    m2 += __shfl_xor(m2, 32, 64);
    if ((lane & 39) == 0) {  // lane&7 == 0 and half == 0
      float* dst = p.gn_part + ((size_t)(slot0 + mt) * GN_GROUPS + (c >> 3)) * 2;
      dst[0] = mean;
      dst[1] = m2;
    }
  }
}

// ROWS (WStream2T): 2 = both 32-row MFMA tiles hold valid rows, 1 = only the first does - fixed at compile
// time so that the GEMM steps are branch-free and the staging slices interleave with their MFMAs (the
// run-time form, a wave-uniform branch around every second-tile MFMA, measured 250 us instead of 238 at
// 32 pairs @1024x1024); the kernel picks the body per workgroup.
template <int MODE, int ROWS>
__device__ __forceinline__ void heat_conv64_body(const HeatLaunch& p, float* smem, float* att_s, float* attmax_s) {
  static_assert(gm_planes(MODE) == 2, "two-plane mode");
  static_assert(16 % WStream2T<MODE, ROWS>::D == 0, "tap loop below assumes a ring phase of 0 after every GEMM");
  constexpr int THREADS = 512, TPR = THREADS / RT, F4 = 64 / TPR;   // 8 threads per row, 8 float4 each
  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int nt0 = (g.L[0] + RT - 1) / RT, nt1 = (g.L[1] + RT - 1) / RT;
  const int logical = xcd_remap(blockIdx.x, g.N * (nt0 + nt1));
  const int per = nt0 + nt1;
  const int n = logical / per;
  const int rem = logical - n * per;
  const int side = rem >= nt0;
  const int t_idx = side ? rem - nt0 : rem;
  const int L = g.L[side], hf = g.hf[side], wf = g.wf[side];
  const int l0 = t_idx * RT;
  const int nvalid = min(RT, L - l0);
  const float* mem = p.mem[side] + (size_t)n * L * C;
  const size_t row_base = (size_t)g.row0[side] + (size_t)n * L + l0;
  Range rg;
  // att[l'] = memory[l'] . hs for the halo rows l0-wf-1 .. l0+RT+wf of this tile, once
  const int halo0 = l0 - wf - 1, nhalo = RT + 2 * (wf + 1);
  const int hrow = tid / TPR, hpart = tid % TPR;
  {
    const f32x4* hsp = reinterpret_cast<const f32x4*>(p.hs[side] + (size_t)n * C) + hpart;
    f32x4 hv[F4];
#pragma unroll
    for (int i = 0; i < F4; ++i) hv[i] = hsp[i * TPR];
    for (int r0 = 0; r0 < nhalo; r0 += RT) {
      const int hr = r0 + hrow;
      const int l = min(max(halo0 + hr, 0), L - 1);  // clamped: out-of-image rows are masked below
      const f32x4* mp = reinterpret_cast<const f32x4*>(mem + (size_t)l * C) + hpart;
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < F4; ++i) {
        const f32x4 v = mp[i * TPR];
        d += (v[0] * hv[i][0] + v[1] * hv[i][1]) + (v[2] * hv[i][2] + v[3] * hv[i][3]);
      }
      d = sum8(d);
      if (hpart == 0 && hr < nhalo) att_s[hr] = d;
    }
  }
  __syncthreads();
  // att weights of the tile divided by a power of two >= max |att| over its halo (exact), the GEMM
  // result multiplied back: the converted operands are bounded by |memory| (see k_heat_conv)
  {
    float m = 0.f;
    for (int i = tid; i < nhalo; i += THREADS) m = fmaxf(m, fabsf(att_s[i]));
    m = wave_max(m);
    if (lane == 0) attmax_s[wave] = m;
  }
  __syncthreads();
  float att_scale = 1.0f, att_unscale = 1.0f;
  {
    float m = attmax_s[0];
#pragma unroll
    for (int i = 1; i < THREADS / 64; ++i) m = fmaxf(m, attmax_s[i]);
    if (m > 0.f && m < INFINITY) {
      const int e = ilogbf(m) + 1;          // 2^e > m
      att_scale = ldexpf(1.0f, -e);
      att_unscale = ldexpf(1.0f, e);
    }
  }
  // two plane regions, tap t in region t & 1 (addresses computed, not looked up: a run-time index into an
  // array of plane descriptors would put the array in scratch)
  auto region = [&](int t) { return PlanesT<MODE>(smem + (t & 1) * R_FLOATS, &rg); };
  // this thread's row of the tile: source row and att weight of tap `tap`, the row's 8 float4 pieces
  const int lrow = l0 + hrow;
  const int y = lrow / wf, x = lrow - y * wf;
  auto tap_loads = [&](int tap, f32x4 (&v)[F4], float& att) {
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int yy = y + dy, xx = x + dx;
    const bool ok = (lrow < L) && (yy >= 0) && (yy < hf) && (xx >= 0) && (xx < wf);
    const int src_row = ok ? yy * wf + xx : 0;   // unconditional loads from a clamped row + select
    att = ok ? att_s[src_row - halo0] * att_scale : 0.f;
    const f32x4* mp = reinterpret_cast<const f32x4*>(mem + (size_t)src_row * C) + hpart;
#pragma unroll
    for (int i = 0; i < F4; ++i) v[i] = mp[i * TPR];
  };
  WStream2T<MODE, ROWS> ws;
  ws.set_rows(nvalid);
  ws.set_lane(lane);
  constexpr size_t TAP_UNITS = (size_t)C * C / 8;
  {
    f32x4 v[F4];
    float att;
    tap_loads(0, v, att);
    ws.template prime<C, 0>(p.w.conv_w, p.w.conv_w_l, wave, 0, lane);
    const PlanesT<MODE> first = region(0);
#pragma unroll
    for (int i = 0; i < F4; ++i) first.put4(hrow, 4 * (i * TPR + hpart), v[i] * att);
  }
  __syncthreads();
  f32x16 acc[2] = {f32x16{0}, f32x16{0}};
#pragma unroll 1
  for (int tap = 0; tap < 8; ++tap) {
    f32x4 v[F4];
    float att;
    tap_loads(tap + 1, v, att);                  // next tap's rows: in flight under this tap's GEMM
    const PlanesT<MODE> cur = region(tap), nxt = region(tap + 1);
    auto epi = [&](auto CI_) {                   // ... converted and stored beside its MFMAs, one piece per 2 k16 steps
      constexpr int CI = decltype(CI_)::value;
      if constexpr (CI % 2 == 1) nxt.put4(hrow, 4 * ((CI / 2) * TPR + hpart), v[CI / 2] * att);
    };
    const f32x4* w = p.w.conv_w + tap * TAP_UNITS;
    const f32x4* wl = p.w.conv_w_l + tap * TAP_UNITS;
    ws.template gemm_epi<C, 0, true, C>(cur, w, wl, wave, 0, lane, acc, w + TAP_UNITS, wl + TAP_UNITS,
                                        wave, 0, epi);
    __syncthreads();   // next tap's planes complete; every wave is done reading this tap's
  }
  ws.template gemm<C, 0, false, C>(region(8), p.w.conv_w + 8 * TAP_UNITS, p.w.conv_w_l + 8 * TAP_UNITS, wave, 0, lane,
                                   acc, nullptr, nullptr, 0, 0);
  const float bias = p.w.conv_b[32 * wave + col];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = fmaf(acc[mt][r], att_unscale, bias);
  range_report<MODE>(rg, p.flags);

  heat_conv64_finish(p, acc, g, side, n, t_idx, nvalid, row_base, lane, wave, half, col);
}

// The direct conv with the tile's HALO resident in LDS (round 4, second form; token grids up to 40 wide).
// The A operand of tap (dy, dx) is the tile's own rows shifted by dy * wf + dx tokens, so the rows
// l0 - wf - 1 .. l0 + 64 + wf are staged ONCE - att-scaled, as split planes, zeros outside the image -
// and every tap's GEMM reads the same planes from another first row: no per-tap gather, conversion or
// barrier; nine GEMM units back to back on one weight stream.  What a shifted window cannot express is
// the x border (a neighbour across the row end is another row's pixel, not a zero), and that mask
// depends on the OUTPUT row and dx only: the three dx = -1 taps are accumulated first and their sum is
// masked (rows with x = 0), the dx = 0 taps accumulate on top, the dx = +1 taps go to a second
// accumulator that is masked (x = wf - 1) and added at the end.  LDS: (64 + 2 wf + 2) rows x 1056 B:
// 137 KB at 32 x 32 tokens, 154 KB at 40 x 40 (HALO_MAX_ROWS); wider grids take k_heat_conv64.
constexpr int HALO_MAX_WF = 40;
constexpr int HALO_MAX_ROWS = RT + 2 * (HALO_MAX_WF + 1);   // 146
template <int MODE, int ROWS>
__device__ __forceinline__ void heat_conv64h_body(const HeatLaunch& p, _Float16* planes, float* att_s, float* mask_s,
                                                  float* attmax_s) {
  static_assert(gm_planes(MODE) == 2, "two-plane mode");
  static_assert(16 % WStream2T<MODE, ROWS>::D == 0, "the tap sequence below assumes a ring phase of 0 after every GEMM");
  constexpr int THREADS = 512, TPR = THREADS / RT, F4 = 64 / TPR;   // 8 threads per row, 8 float4 each
  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, col = lane & 31;
  const int nt0 = (g.L[0] + RT - 1) / RT, nt1 = (g.L[1] + RT - 1) / RT;
  const int logical = xcd_remap(blockIdx.x, g.N * (nt0 + nt1));
  const int per = nt0 + nt1;
  const int n = logical / per;
  const int rem = logical - n * per;
  const int side = rem >= nt0;
  const int t_idx = side ? rem - nt0 : rem;
  const int L = g.L[side], wf = g.wf[side];
  const int l0 = t_idx * RT;
  const int nvalid = min(RT, L - l0);
  const float* mem = p.mem[side] + (size_t)n * L * C;
  const size_t row_base = (size_t)g.row0[side] + (size_t)n * L + l0;
  Range rg;
  const int halo0 = l0 - wf - 1, nhalo = RT + 2 * (wf + 1);
  const int hrow = tid / TPR, hpart = tid % TPR;
  // att[l'] = memory[l'] . hs for the halo rows; the rows' pieces stay in registers for the staging pass
  // when the halo fits three passes of 64 rows (it does up to wf = 63)
  constexpr int NPASS = (HALO_MAX_ROWS + RT - 1) / RT;   // 3
  f32x4 hv[F4];
  {
    const f32x4* hsp = reinterpret_cast<const f32x4*>(p.hs[side] + (size_t)n * C) + hpart;
#pragma unroll
    for (int i = 0; i < F4; ++i) hv[i] = hsp[i * TPR];
  }
#pragma unroll 1
  for (int ps = 0; ps < NPASS; ++ps) {
    const int hr = ps * RT + hrow;
    const int l = min(max(halo0 + hr, 0), L - 1);  // clamped: out-of-image rows are zeroed in the staging pass
    const f32x4* mp = reinterpret_cast<const f32x4*>(mem + (size_t)l * C) + hpart;
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < F4; ++i) {
      const f32x4 v = mp[i * TPR];
      d += (v[0] * hv[i][0] + v[1] * hv[i][1]) + (v[2] * hv[i][2] + v[3] * hv[i][3]);
    }
    d = sum8(d);
    if (hpart == 0 && hr < nhalo) att_s[hr] = d;
  }
  if (tid < RT) {   // x-border masks of the tile's output rows
    const int l = l0 + tid, x = l - (l / wf) * wf;
    mask_s[tid] = x != 0 ? 1.0f : 0.0f;             // dx = -1 taps
    mask_s[RT + tid] = x != wf - 1 ? 1.0f : 0.0f;   // dx = +1 taps
  }
  __syncthreads();
  {
    float m = 0.f;
    for (int i = tid; i < nhalo; i += THREADS) m = fmaxf(m, fabsf(att_s[i]));
    m = wave_max(m);
    if (lane == 0) attmax_s[wave] = m;
  }
  __syncthreads();
  float att_scale = 1.0f, att_unscale = 1.0f;
  {
    float m = attmax_s[0];
#pragma unroll
    for (int i = 1; i < THREADS / 64; ++i) m = fmaxf(m, attmax_s[i]);
    if (m > 0.f && m < INFINITY) {
      const int e = ilogbf(m) + 1;          // 2^e > m
      att_scale = ldexpf(1.0f, -e);
      att_unscale = ldexpf(1.0f, e);
    }
  }
  _Float16* const hi0 = planes;
  _Float16* const lo0 = planes + HALO_MAX_ROWS * LDAH;
  WStream2T<MODE, ROWS> ws;
  ws.set_rows(nvalid);
  ws.set_lane(lane);
  constexpr size_t TAP_UNITS = (size_t)C * C / 8;
  // tap sequence: dx = -1 (dy = -1, 0, 1), dx = 0, dx = +1;  tap id = 3 (dy + 1) + (dx + 1)
  auto tap_of = [](int k) { return 3 * (k % 3) + k / 3; };
  ws.template prime<C, 0>(p.w.conv_w + tap_of(0) * TAP_UNITS, p.w.conv_w_l + tap_of(0) * TAP_UNITS, wave, 0, lane);
  {  // the halo, once: att-scaled split planes, zero rows outside the image
    const PlanesT<MODE> H(hi0, lo0, &rg);
#pragma unroll 1
    for (int ps = 0; ps < NPASS; ++ps) {
      const int hr = ps * RT + hrow;
      if (hr < nhalo) {
        const int l = halo0 + hr;
        const bool in = l >= 0 && l < L;
        const float att = in ? att_s[hr] * att_scale : 0.f;
        const f32x4* mp = reinterpret_cast<const f32x4*>(mem + (size_t)min(max(l, 0), L - 1) * C) + hpart;
#pragma unroll
        for (int i = 0; i < F4; ++i) H.put4(hr, 4 * (i * TPR + hpart), mp[i * TPR] * att);
      }
    }
  }
  __syncthreads();
  f32x16 acc[2] = {f32x16{0}, f32x16{0}}, accR[2] = {f32x16{0}, f32x16{0}};
  auto window = [&](int k) {   // the planes as tap k sees them: first row = output row 0 shifted by dy wf + dx
    const int dy = k % 3 - 1, dx = k / 3 - 1;
    const int shift = (wf + 1) + dy * wf + dx;
    return PlanesT<MODE>(hi0 + shift * LDAH, lo0 + shift * LDAH, &rg);
  };
  auto run = [&](int k, f32x16 (&a)[2], auto HAS_NEXT_) {
    constexpr bool HAS_NEXT = decltype(HAS_NEXT_)::value;
    const int t = tap_of(k), tn = tap_of(HAS_NEXT ? k + 1 : k);
    ws.template gemm<C, 0, HAS_NEXT, C>(window(k), p.w.conv_w + t * TAP_UNITS, p.w.conv_w_l + t * TAP_UNITS, wave, 0, lane,
                                        a, p.w.conv_w + tn * TAP_UNITS, p.w.conv_w_l + tn * TAP_UNITS, wave, 0);
  };
#pragma unroll 1
  for (int k = 0; k < 3; ++k) run(k, acc, std::true_type{});          // dx = -1
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] *= mask_s[32 * mt + crow(r, half)];
#pragma unroll 1
  for (int k = 3; k < 6; ++k) run(k, acc, std::true_type{});          // dx = 0
#pragma unroll 1
  for (int k = 6; k < 8; ++k) run(k, accR, std::true_type{});         // dx = +1
  run(8, accR, std::false_type{});
  const float bias = p.w.conv_b[32 * wave + col];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      acc[mt][r] = fmaf(fmaf(accR[mt][r], mask_s[RT + 32 * mt + crow(r, half)], acc[mt][r]), att_unscale, bias);
  range_report<MODE>(rg, p.flags);
  heat_conv64_finish(p, acc, g, side, n, t_idx, nvalid, row_base, lane, wave, half, col);
}

template <int MODE>
__global__ __launch_bounds__(512) void k_heat_conv64(HeatLaunch p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * R_FLOATS];
  __shared__ float att_s[RT + 2 * (100 + 1) + 6];
  __shared__ float attmax_s[512 / 64];
  const Geom& g = p.g;
  const int nt0 = (g.L[0] + RT - 1) / RT, nt1 = (g.L[1] + RT - 1) / RT;
  if ((int)blockIdx.x >= g.N * (nt0 + nt1)) return size_reg_body(p, blockIdx.x - g.N * (nt0 + nt1), smem);
  const int logical = xcd_remap(blockIdx.x, g.N * (nt0 + nt1));
  const int rem = logical - (logical / (nt0 + nt1)) * (nt0 + nt1);
  const int side = rem >= nt0;
  const int t_idx = side ? rem - nt0 : rem;
  const int nvalid = min(RT, g.L[side] - t_idx * RT);
  if (__builtin_amdgcn_readfirstlane(nvalid > 32 ? 1 : 0)) heat_conv64_body<MODE, 2>(p, smem, att_s, attmax_s);
  else heat_conv64_body<MODE, 1>(p, smem, att_s, attmax_s);
}

template <int MODE>
__global__ __launch_bounds__(512) void k_heat_conv64h(HeatLaunch p) {
  __shared__ __attribute__((aligned(16))) _Float16 planes[2 * HALO_MAX_ROWS * LDAH];   // 154 176 B
  __shared__ float att_s[HALO_MAX_ROWS + 6];
  __shared__ float mask_s[2 * RT];
  __shared__ float attmax_s[512 / 64];
  const Geom& g = p.g;
  const int nt0 = (g.L[0] + RT - 1) / RT, nt1 = (g.L[1] + RT - 1) / RT;
  if ((int)blockIdx.x >= g.N * (nt0 + nt1))
    return size_reg_body(p, blockIdx.x - g.N * (nt0 + nt1), reinterpret_cast<float*>(planes));
  const int logical = xcd_remap(blockIdx.x, g.N * (nt0 + nt1));
  const int rem = logical - (logical / (nt0 + nt1)) * (nt0 + nt1);
  const int side = rem >= nt0;
  const int t_idx = side ? rem - nt0 : rem;
  const int nvalid = min(RT, g.L[side] - t_idx * RT);
  if (__builtin_amdgcn_readfirstlane(nvalid > 32 ? 1 : 0)) heat_conv64h_body<MODE, 2>(p, planes, att_s, mask_s, attmax_s);
  else heat_conv64h_body<MODE, 1>(p, planes, att_s, mask_s, attmax_s);
}

hipError_t launch_heat_conv64(const HeatLaunch& p, int mode, hipStream_t s) {
  if (mode != GM_SPLIT) return hipErrorInvalidValue;
  const int tiles = p.g.N * ((p.g.L[0] + RT - 1) / RT + (p.g.L[1] + RT - 1) / RT);
  // halo-resident form while both token grids are at most HALO_MAX_WF wide (its LDS holds 64 + 2 wf + 2 rows)
  // + one size-regression workgroup per image behind the tiles (size_reg_body)
  if (p.g.wf[0] <= HALO_MAX_WF && p.g.wf[1] <= HALO_MAX_WF && !p.force_staged_conv)
    hipLaunchKernelGGL((k_heat_conv64h<GM_SPLIT>), dim3(tiles + 2 * p.g.N), dim3(512), 0, s, p);
  else
    hipLaunchKernelGGL((k_heat_conv64<GM_SPLIT>), dim3(tiles + 2 * p.g.N), dim3(512), 0, s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// conv_out[l] = b + sum_tap att[l+tap] * P_tap[l+tap]  (P from k_decoder_convp),
// plus the per-tile GroupNorm moments.  512 threads, 16 per token row.
__global__ __launch_bounds__(512) void k_heat_combine(HeatLaunch p, const float* __restrict__ P) {
  constexpr int TPR = 16, F4 = 4;
  __shared__ __attribute__((aligned(16))) float out_s[TM * LDA];
  __shared__ float att_s[TM + 2 * (100 + 1) + 6];
  const Geom& g = p.g;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= g.ntiles) return size_reg_body(p, blockIdx.x - g.ntiles, out_s);
  const int logical = xcd_remap(blockIdx.x, g.ntiles);
  const int per = g.nt[0] + g.nt[1];
  const int n = logical / per;
  const int rem = logical - n * per;
  const int side = rem >= g.nt[0];
  const int t_idx = side ? rem - g.nt[0] : rem;
  const int L = g.L[side], hf = g.hf[side], wf = g.wf[side];
  const int l0 = t_idx * TM;
  const int nvalid = min(TM, L - l0);
  const int slot = g.tile0[side] + n * g.nt[side] + t_idx;
  const float* mem = p.mem[side] + (size_t)n * L * C;
  const size_t img_row0 = (size_t)g.row0[side] + (size_t)n * L;
  const int halo0 = l0 - wf - 1, nhalo = TM + 2 * (wf + 1);
  const int hrow = tid / TPR, hpart = tid % TPR;
  {  // att for the halo rows
    const f32x4* hsp = reinterpret_cast<const f32x4*>(p.hs[side] + (size_t)n * C) + hpart;
    f32x4 hv[F4];
#pragma unroll
    for (int i = 0; i < F4; ++i) hv[i] = hsp[i * TPR];
    for (int r0 = 0; r0 < nhalo; r0 += TM) {
      const int hr = r0 + hrow;
      const int l = min(max(halo0 + hr, 0), L - 1);
      const f32x4* mp = reinterpret_cast<const f32x4*>(mem + (size_t)l * C) + hpart;
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < F4; ++i) {
        const f32x4 v = mp[i * TPR];
        d += (v[0] * hv[i][0] + v[1] * hv[i][1]) + (v[2] * hv[i][2] + v[3] * hv[i][3]);
      }
      d = sum8(d);
      d += dpp_mov<0x140>(d);
      if (hpart == 0 && hr < nhalo) att_s[hr] = d;
    }
  }
  __syncthreads();
  {
    const int l = l0 + hrow;
    const int y = l / wf, x = l - y * wf;
    f32x4 acc[F4];
#pragma unroll
    for (int i = 0; i < F4; ++i) acc[i] = reinterpret_cast<const f32x4*>(p.w.conv_b)[i * TPR + hpart];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const int yy = y + dy, xx = x + dx;
      const bool ok = (l < L) && (yy >= 0) && (yy < hf) && (xx >= 0) && (xx < wf);
      const int src_row = ok ? yy * wf + xx : 0;
      const float att = ok ? att_s[src_row - halo0] : 0.f;
      const f32x4* pp = reinterpret_cast<const f32x4*>(
                            P + ((size_t)tap * g.rows + img_row0 + src_row) * C) + hpart;
#pragma unroll
      for (int i = 0; i < F4; ++i) acc[i] += pp[i * TPR] * att;
    }
    f32x4* go = reinterpret_cast<f32x4*>(p.conv_out + (img_row0 + min(l, L - 1)) * C) + hpart;
#pragma unroll
    for (int i = 0; i < F4; ++i) {
      if (l < L) go[i * TPR] = acc[i];
      *reinterpret_cast<f32x4*>(out_s + hrow * LDA + 4 * (i * TPR + hpart)) = acc[i];
    }
  }
  __syncthreads();
  if (tid < 256) {  // GroupNorm moments: group = tid >> 3, 8 threads x 4 rows each
    const int grp = tid >> 3, rs = tid & 7;
    float s = 0.f;
    f32x4 v[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rs * 4 + r;
      v[r][0] = *reinterpret_cast<const f32x4*>(out_s + row * LDA + 8 * grp);
      v[r][1] = *reinterpret_cast<const f32x4*>(out_s + row * LDA + 8 * grp + 4);
      if (row < nvalid)
        s += ((v[r][0][0] + v[r][0][1]) + (v[r][0][2] + v[r][0][3])) +
             ((v[r][1][0] + v[r][1][1]) + (v[r][1][2] + v[r][1][3]));
    }
    const float mean = sum8(s) / (float)(nvalid * 8);
    float m2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (rs * 4 + r < nvalid)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float d = v[r][k][j] - mean; m2 += d * d; }
    m2 = sum8(m2);
    if (rs == 0) {
      float* dst = p.gn_part + ((size_t)slot * GN_GROUPS + grp) * 2;
      dst[0] = mean;
      dst[1] = m2;
    }
  }
}

hipError_t launch_heat_combine(const HeatLaunch& h, const float* P, hipStream_t s) {
  hipLaunchKernelGGL(k_heat_combine, dim3(h.g.ntiles + 2 * h.g.N), dim3(512), 0, s, h, P);   // + size regression per image
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// GroupNorm + ReLU + 1x1 conv -> logits of one 32-token tile (reference src/model.py:65-77, :165-168).
// One workgroup per TILE (round 4; rounds 1-3 ran this inside k_heat_final, one workgroup per IMAGE:
// 400 KB of conv_out through one CU in seven dependent passes = 9 of that kernel's 18.8 us at 400
// tokens, 16 of 26.8 at 1024).  Every workgroup folds its image's per-tile moments itself (Chan et al.,
// 16 interleaved chunks of tiles per group, then the 16 partial results in chunk order - the same
// arithmetic in every workgroup of the image); its own rows are in flight meanwhile.
__global__ __launch_bounds__(512, 8) void k_heat_logits(HeatLaunch p) {
  __shared__ float fold_s[16 * GN_GROUPS * 3];
  __shared__ float gmean_s[GN_GROUPS], grstd_s[GN_GROUPS];
  const Geom& g = p.g;
  const int tid = threadIdx.x;
  const int logical = xcd_remap(blockIdx.x, g.ntiles);
  const int per = g.nt[0] + g.nt[1];
  const int n = logical / per;
  const int rem = logical - n * per;
  const int side = rem >= g.nt[0];
  const int t_idx = side ? rem - g.nt[0] : rem;
  const int L = g.L[side], nts = g.nt[side];
  const int slot0 = g.tile0[side] + n * nts;
  const size_t row0 = (size_t)g.row0[side] + (size_t)n * L;
  const int part = tid & 15, l = t_idx * TM + (tid >> 4);
  const f32x4* row = reinterpret_cast<const f32x4*>(p.conv_out + (row0 + min(l, L - 1)) * C) + part;
  f32x4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = row[i * 16];
  // GroupNorm affine + 1x1 conv weights: through LDS (held in registers across the fold they cost the
  // kernel a workgroup per CU: 96 VGPRs -> 2 workgroups of 8 waves; now 4)
  __shared__ __attribute__((aligned(16))) float aff_s[3 * C];
  if (tid < 3 * C / 4) {
    const float* srcp = tid < C / 4 ? p.w.gn_w : tid < C / 2 ? p.w.gn_b : p.w.out_w;
    reinterpret_cast<f32x4*>(aff_s)[tid] = reinterpret_cast<const f32x4*>(srcp)[tid & (C / 4 - 1)];
  }
  {
    const int grp = tid & 31, ch = tid >> 5;
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2* src = reinterpret_cast<const f32x2*>(p.gn_part) + (size_t)slot0 * GN_GROUPS + grp;
    for (int t0 = ch; t0 < nts; t0 += 64) {   // four of this chunk's tiles per round trip
      f32x2 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = src[(size_t)min(t0 + 16 * u, nts - 1) * GN_GROUPS];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ti = t0 + 16 * u;
        if (ti < nts) {
          const float nb = 8.f * (float)min(TM, L - ti * TM);
          const float tot = cnt + nb, delta = q[u][0] - mean;
          mean += delta * (nb / tot);
          m2 += q[u][1] + delta * delta * (cnt * nb / tot);
          cnt = tot;
        }
      }
    }
    float* dst = fold_s + (ch * GN_GROUPS + grp) * 3;
    dst[0] = cnt; dst[1] = mean; dst[2] = m2;
  }
  __syncthreads();
  if (tid < GN_GROUPS) {
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    for (int ch = 0; ch < 16; ++ch) {
      const float* src = fold_s + (ch * GN_GROUPS + tid) * 3;
      const float nb = src[0];
      if (nb > 0.f) {
        const float tot = cnt + nb, delta = src[1] - mean;
        mean += delta * (nb / tot);
        m2 += src[2] + delta * delta * (cnt * nb / tot);
        cnt = tot;
      }
    }
    gmean_s[tid] = mean;
    grstd_s[tid] = 1.0f / sqrtf(m2 / cnt + GN_EPS);
  }
  __syncthreads();
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f4 = i * 16 + part, grp = f4 >> 1;  // float4 index -> channels 4*f4 .. 4*f4+3, group f4 >> 1
    const f32x4 gw = reinterpret_cast<const f32x4*>(aff_s)[f4], gb = reinterpret_cast<const f32x4*>(aff_s + C)[f4],
                ow = reinterpret_cast<const f32x4*>(aff_s + 2 * C)[f4];
    const f32x4 y = (v[i] - gmean_s[grp]) * grstd_s[grp] * gw + gb;
#pragma unroll
    for (int j = 0; j < 4; ++j) d += fmaxf(y[j], 0.f) * ow[j];
  }
  d = sum8(d);
  d += dpp_mov<0x140>(d);  // row_mirror: the other 8 lanes of the 16
  d += p.w.out_b[0];
  // forward_dummy's masks: heatmap.masked_fill_(~mask.bool(), -INF) with INF = 1e9 (model.py:22, :166-171)
  if (p.mask[side] != nullptr && p.mask[side][(size_t)n * L + min(l, L - 1)] == 0.f) d = -1.0e9f;
  __shared__ float logit_s[TM];
  if (part == 0) {
    logit_s[tid >> 4] = l < L ? d : -INFINITY;
    if (l < L) p.logits[row0 + l] = d;
  }
  __syncthreads();
  // this tile's share of the image's softmax / soft-argmax (model.py:173-184), relative to the tile's
  // own maximum: (max, sum e, sum e x, sum e y) - k_heat_final merges the tiles of an image
  if (tid < 64) {
    const int lt = t_idx * TM + (tid & 31);
    const float v = tid < TM ? logit_s[tid] : -INFINITY;
    const float m = wave_max(v);
    const float e = v > -INFINITY ? expf(v - m) : 0.f;
    const int wf = g.wf[side];
    const int y = lt / wf, x = lt - y * wf;
    const float stride = (float)(p.img_h[side] / g.hf[side]);
    const float se = wave_sum(e);
    const float sx = wave_sum(e * (((float)x + 0.5f) * stride));
    const float sy = wave_sum(e * (((float)y + 0.5f) * stride));
    if (tid == 0) {
      float* dst = p.sm_part + (size_t)(slot0 + t_idx) * 4;
      dst[0] = m; dst[1] = se; dst[2] = sx; dst[3] = sy;
    }
  }
}

// Softmax over an image's tokens + soft-argmax -> centre (reference src/model.py:173-184) from the
// per-tile partials of k_heat_logits (one wave per image: lanes merge their tiles in tile order,
// then across the wave); on the forward path also the box from the size regression's tlbr
// (model.py:188-191, models/utils.py:16-28).
__global__ __launch_bounds__(64) void k_heat_final(HeatLaunch p) {
  const Geom& g = p.g;
  const int lane = threadIdx.x;
  const int img = blockIdx.x;
  const int side = img >= g.N, n = side ? img - g.N : img;
  const int nts = g.nt[side];
  const float* part = p.sm_part + (size_t)(g.tile0[side] + n * nts) * 4;
  // Deferred check without runtime dispatches (include/oetr_hip.h, ABI 6): every kernel of this forward
  // that can set a status bit has completed (same stream, this kernel sets none), so one lane moves the
  // word into the caller's mapped host slot - system-scope store, visible to the host once an event
  // behind this launch has completed - and leaves 0 behind for the next call on the workspace.
  if (p.publish && img == 0 && lane == 0) {
    const uint32_t word = __hip_atomic_exchange(p.flags, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p.publish, word | FLAG_PUBLISHED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  float tl = 0.f;
  if (p.box[side] && lane < 4) tl = p.tlbr[side][4 * n + lane];
  float m = -INFINITY, se = 0.f, sx = 0.f, sy = 0.f;
  for (int ti = lane; ti < nts; ti += 64) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(part + 4 * ti);
    const float mn = fmaxf(m, q[0]);
    const float a = expf(m - mn), b = expf(q[0] - mn);   // (m = -inf before the first tile: a = 0)
    se = se * a + q[1] * b;
    sx = sx * a + q[2] * b;
    sy = sy * a + q[3] * b;
    m = mn;
  }
  const float mx = wave_max(m);
  const float sc = expf(m - mx);                          // lanes without a tile: exp(-inf) = 0
  se = wave_sum(se * sc);
  sx = wave_sum(sx * sc) / se;
  sy = wave_sum(sy * sc) / se;
  if (lane == 0) {
    p.cxy[side][2 * n] = sx;
    p.cxy[side][2 * n + 1] = sy;
  }
  if (!p.box[side]) return;
  const float t0 = __shfl(tl, 0, 64), t1 = __shfl(tl, 1, 64), t2 = __shfl(tl, 2, 64), t3 = __shfl(tl, 3, 64);
  if (lane == 0) {
    const float mh = (float)p.img_h[side], mw = (float)p.img_w[side];
    const float t = t0 * mh, l = t1 * mw, b = t2 * mh, r = t3 * mw;
    float* box = p.box[side] + 4 * n;
    box[0] = fminf(fmaxf(sx - l, 0.f), mw);
    box[1] = fminf(fmaxf(sy - t, 0.f), mh);
    box[2] = fminf(fmaxf(sx + r, 0.f), mw);
    box[3] = fminf(fmaxf(sy + b, 0.f), mh);
  }
}

hipError_t launch_heat_final(const HeatLaunch& p, hipStream_t s) {
  hipLaunchKernelGGL(k_heat_logits, dim3(p.g.ntiles), dim3(512), 0, s, p);
  hipLaunchKernelGGL(k_heat_final, dim3(2 * p.g.N), dim3(64), 0, s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_size_reg(HeadsDev w, const float* __restrict__ hs1,
                                                  const float* __restrict__ hs2, int n,
                                                  float* __restrict__ tlbr1,
                                                  float* __restrict__ tlbr2) {
  __shared__ float h_s[C], hid_s[C];
  const int tid = threadIdx.x, img = blockIdx.x;
  const int side = img >= n, i = side ? img - n : img;
  const float* hs = (side ? hs2 : hs1) + (size_t)i * C;
  h_s[tid] = hs[tid];
  __syncthreads();
  float a = 0.f;
#pragma unroll 8
  for (int k = 0; k < C; ++k) a += w.tlbr0_t[k * C + tid] * h_s[k];
  hid_s[tid] = fmaxf(a, 0.f);
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // wave j -> output j
  const f32x4 wv = reinterpret_cast<const f32x4*>(w.tlbr2_w + wave * C)[lane];
  const f32x4 hv = reinterpret_cast<const f32x4*>(hid_s)[lane];
  float d = wave_sum((wv[0] * hv[0] + wv[1] * hv[1]) + (wv[2] * hv[2] + wv[3] * hv[3]));
  if (lane == 0) {
    d += w.tlbr2_b[wave];
    (side ? tlbr2 : tlbr1)[4 * i + wave] = 1.0f / (1.0f + expf(-d));
  }
}

hipError_t launch_size_regression(const HeadsDev& w, const float* hs1, const float* hs2,
                                  int n, float* tlbr1, float* tlbr2, hipStream_t s) {
  hipLaunchKernelGGL(k_size_reg, dim3(2 * n), dim3(256), 0, s, w, hs1, hs2, n, tlbr1, tlbr2);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ void k_boxes(const float* __restrict__ cxy, const float* __restrict__ tlbr, int n,
                        float max_h, float max_w, float* __restrict__ box) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = cxy[2 * i], y = cxy[2 * i + 1];
  const float t = tlbr[4 * i] * max_h, l = tlbr[4 * i + 1] * max_w;
  const float b = tlbr[4 * i + 2] * max_h, r = tlbr[4 * i + 3] * max_w;
  box[4 * i + 0] = fminf(fmaxf(x - l, 0.f), max_w);
  box[4 * i + 1] = fminf(fmaxf(y - t, 0.f), max_h);
  box[4 * i + 2] = fminf(fmaxf(x + r, 0.f), max_w);
  box[4 * i + 3] = fminf(fmaxf(y + b, 0.f), max_h);
}

hipError_t launch_boxes(const float* cxy, const float* tlbr, int n, int max_h, int max_w,
                        float* box, hipStream_t s) {
  hipLaunchKernelGGL(k_boxes, dim3((n + 63) / 64), dim3(64), 0, s, cxy, tlbr, n, (float)max_h,
                     (float)max_w, box);
  return hipGetLastError();
}

}  // namespace oetr
