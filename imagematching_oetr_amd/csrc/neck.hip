// Neck of the OETR feature extractor on gfx950:
//
//   feat = input_proj2( PatchMerging( input_proj( backbone_out ) ) )
//
// reference src/model.py:113-118 and src/models/backbone.py:53-67: a 1x1 conv
// 1024 -> 256, LayerNorm over channels at every position, three stride-2 convs
// (kernel 4 / 8 / 16, padding (k-2)/2, 256 -> 256 / 128 / 128 channels)
// concatenated, and a 1x1 conv 512 -> 256.  9.2 of its 10.2 GFLOP per image are
// the three convs: implicit GEMMs with K = k*k*256 up to 65536.
//
// All products use the split-f16 arithmetic of common.h (a = ah + al*2^-11,
// three f16 MFMAs, fp32 accumulation): fp32-class results at f16 MFMA rate.
//
//   k_neck_proj  32 positions per workgroup: NCHW gather -> LDS split planes,
//                1x1 conv (K = 1024 in two halves, weights streamed as B
//                fragments), LayerNorm, result stored token-major as f16 hi / lo
//                halves in X: per pixel eight 128-byte chunks [32 ch hi | 32 ch lo]
//                (+ one all-zero row used as padding source).
//   k_neck_conv_rw  (default) the three convs in ONE grid.  A workgroup owns 192
//                output positions x 128 output channels x 16 kernel pixels
//                (K = 4096, split-K over the kernel window: 16 / 4 / 1 slices for
//                k = 16 / 8 / 4).  Row-window staging: per (kernel row, x parity,
//                32-channel chunk) the pixels of the tile's output rows go once
//                into a double-buffered LDS tile and tap tau reads it shifted by
//                tau entries; weights straight from L2 into B-fragment registers
//                four k16 steps ahead; one wave per SIMD, every weight fragment
//                feeds six 32-row MFMA tiles (accumulators in AGPRs).  Partials
//                -> HBM.
//   k_neck_conv  the round-1 form (one gathered pixel per output position per
//                kernel pixel, 64-channel stages, two waves per SIMD): kept for
//                output maps narrower than 16 and as the A/B reference.
//   k_neck_out   32 positions per workgroup: fixed-order sum of the partials +
//                biases -> the 512-channel concat as split planes in LDS, 1x1
//                conv 512 -> 256, transposed store to NCHW - or token-major
//                straight into the hot path's workspace (oetr_neck_forward_tokens).
#include <type_traits>
#include "common.h"

namespace oetr {

// ---------------------------------------------------------------------------
// k_neck_proj
// ---------------------------------------------------------------------------
constexpr int NP_S0_OFF = HID_FLOATS;            // A planes [32][520] x2 first
constexpr int NP_PAR_OFF = NP_S0_OFF + TM * LDA; // f32 tile [32][260]
constexpr int NP_SMEM = NP_PAR_OFF + 2 * C;      // LayerNorm affine

// Stage channels [kh*512, kh*512+512) of 32 positions: NCHW rows (one channel,
// 32 consecutive positions = 128 B per half-wave) -> LDS [pos][channel] planes.
__device__ __forceinline__ void proj_stage(const ATile<GM_SPLIT>& A, const float* src, int HW,
                                           int tid) {
  const int pos = tid & 31, cg = tid >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int cb = cg + 16 * i;  // block of 8 channels
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(cb * 8 + j) * HW];
    f16x2 h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      split2(v[2 * j], v[2 * j + 1], h[j], l[j]);
      A.rg->see_hi(__builtin_bit_cast(uint32_t, h[j]));
    }
    const f16x8 hv = {h[0][0], h[0][1], h[1][0], h[1][1], h[2][0], h[2][1], h[3][0], h[3][1]};
    const f16x8 lv = {l[0][0], l[0][1], l[1][0], l[1][1], l[2][0], l[2][1], l[3][0], l[3][1]};
    *reinterpret_cast<f16x8*>(A.h + pos * A.ldh + cb * 8) = hv;
    *reinterpret_cast<f16x8*>(A.l + pos * A.ldh + cb * 8) = lv;
  }
}

__global__ __launch_bounds__(512) void k_neck_proj(NeckProjLaunch p) {
  __shared__ __attribute__((aligned(16))) float smem[NP_SMEM];
  Range rg;
  const ATile<GM_SPLIT> A(smem, LDH, LDHH, &rg);
  float* S0 = smem + NP_S0_OFF;
  float* par = smem + NP_PAR_OFF;
  const NeckGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 31;
  const int tpi = (g.HW + TM - 1) / TM;
  const int img = blockIdx.x / tpi, p0 = (blockIdx.x - img * tpi) * TM;
  const int nvalid = min(TM, g.HW - p0);

  if (tid < 2 * C) par[tid] = tid < C ? p.ln_w[tid] : p.ln_b[tid - C];
  if (blockIdx.x == 0 && tid < 64) {  // the padding row of both planes
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    reinterpret_cast<f32x4*>(p.xh + (size_t)g.rows_in * NECK_XROW)[tid] = z;   // 64 x 16 B = one X row
  }
  f32x16 acc[1];
  {
    const float b = p.bias[32 * wave + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = b;
  }
  using WS = WStream<GM_SPLIT, 1>;
  WS ws;
  ws.set_lane(lane);
  constexpr int P0 = 0, P1 = WS::adv(P0, 512);
  const float* src = p.bb + (size_t)img * BBC * g.HW + p0 + min(tid & 31, nvalid - 1);
  proj_stage(A, src, g.HW, tid);
  ws.template prime<512, P0>(p.wh[0], p.wl[0], wave, lane);
  __syncthreads();
  ws.template gemm<512, P0, 512>(A, p.wh[0], p.wl[0], wave, lane, acc, p.wh[1], p.wl[1], wave, 0);
  __syncthreads();
  proj_stage(A, src + (size_t)512 * g.HW, g.HW, tid);
  __syncthreads();
  ws.template gemm<512, P1, 0>(A, p.wh[1], p.wl[1], wave, lane, acc, nullptr, nullptr, 0, 0);
  acc_to_lds<1>(S0, LDA, 32 * wave, lane, acc);
  __syncthreads();

  // LayerNorm over the 256 channels of each position (backbone.py:59), split, store
  f32x4 xn[4];
  ln_rows<16, 4>(S0, tid, xn, 0);
  const int lrow = tid >> 4, lpart = tid & 15;
  if (lrow < nvalid) {
    const size_t row = (size_t)img * g.HW + p0 + lrow;
    const f32x4* gw = reinterpret_cast<const f32x4*>(par) + lpart;
    const f32x4* gb = reinterpret_cast<const f32x4*>(par + C) + lpart;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {   // channel c -> chunk c >> 5: hi at halves 64*(c>>5) + (c&31), lo 32 further
      const int c = 4 * (i * 16 + lpart);
      _Float16* hrow = p.xh + row * NECK_XROW + (c >> 5) * 32;   // (+ c inside store_split4)
      store_split4(hrow, hrow + 32, c, xn[i] * gw[i * 16] + gb[i * 16], rg);
    }
  }
  range_report<GM_SPLIT>(rg, p.flags);
}

hipError_t launch_neck_proj(const NeckProjLaunch& p, hipStream_t s) {
  const int tpi = (p.g.HW + TM - 1) / TM;
  hipLaunchKernelGGL(k_neck_proj, dim3(p.g.n_img * tpi), dim3(512), 0, s, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_neck_conv
// ---------------------------------------------------------------------------
constexpr int NC_ROWB = 272;                   // staged row: hi 128 B | lo 128 B | pad 16 B
constexpr int NC_STAGES = NECK_PIX * 4;        // 16 pixels x 4 channel quarters
static_assert(NC_ROWB % 16 == 0 && (NC_ROWB / 4) % 64 == 4, "conflict-free b128 row stride");

struct ConvB { f32x4 h, l; };                  // one k16-step of B fragments (hi, lo)
#ifndef NECK_SLICE_MAJOR
#define NECK_SLICE_MAJOR 0   // work-item order: 0 = tile-major (X rows of a tile share an L2), 1 = slice-major
#endif
#ifndef NECK_RW_SLICE_MAJOR
#define NECK_RW_SLICE_MAJOR 0   // the row-window kernel's work-item order (see k_neck_conv_rw)
#endif
#ifndef NECK_RING8
#define NECK_RING8 0   // 8-deep weight ring in the one-wave-per-SIMD shape: measured slower (433 vs 414 us)
#endif
#ifndef NECK_SPREAD
#define NECK_SPREAD 0   // 1: issue the next stage's gather passes between the k16 steps (measured slower: 460 vs 441 us)
#endif
#ifndef NECK_ABL
#define NECK_ABL 0   // timing experiments only: 1 no A gathers, 2 no B loads, 4 no MFMA, 8 no LDS writes,
                     // (row-window kernel) 16 no A re-reads from LDS, 32 no stage barrier
#endif

// RTW = 32-row MFMA tiles per wave: a workgroup covers MT = 64 * RTW output positions
// (two row halves of RTW tiles each) - 256, 192 or 128.  Fewer rows per workgroup mean
// more, smaller work items (better balance over the 256 CUs when the item count is a
// little over a multiple of it) at the price of re-streaming each weight slab more often;
// launch_neck_conv picks the cheapest for the problem size.
// (Capping the 128-row shape at 128 VGPRs so that two workgroups share a CU spills 72 B per
//  lane and runs 1.5x slower: 781 vs 514 us - tools/neck_rows.py.)
template <int RTW>
__global__ __launch_bounds__(512) void k_neck_conv(NeckConvLaunch p) {
  constexpr int MT = 64 * RTW, HALF_ROWS = 32 * RTW, BUF = MT * NC_ROWB;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  __shared__ int2 rowinfo[MT];
  const NeckGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar)
  const int half = lane >> 5, col = lane & 31;
  const unsigned ln = lane;   // last, 32-bit index of the weight-fragment addresses (SGPR base + lane offset)
  const int nt = wave & 3, rh = wave >> 2;  // this wave: n-tile nt, rows [HALF_ROWS*rh, +HALF_ROWS)

  // ---- which output tile / conv / K slice ----
  // Work items are ordered tile-major (all 22 slices of the three convs of one
  // position tile are adjacent) and each XCD takes a contiguous run of them:
  // the gathered X rows of a tile (~1.6 MB) are then shared through ONE L2.
  // (Slice-major - a weight slice stays in L2, the tiles' X rows stream - measured equal:
  //  518.8 vs 525.6 us at 16 maps of 40x40; the kernel is not fabric-bound.)
  const int logical = xcd_remap(blockIdx.x, p.nblocks);
#if NECK_SLICE_MAJOR
  const int mtiles = p.nblocks / p.items_per_mt;
  const int it = logical / mtiles, mt = logical - it * mtiles;
#else
  const int mt = logical / p.items_per_mt, it = logical - mt * p.items_per_mt;
#endif
  const int ci = it >= p.conv[2].item0 ? 2 : (it >= p.conv[1].item0 ? 1 : 0);
  const NeckConvDesc& cd = p.conv[ci];
  const int rem = it - cd.item0;
  const int split = rem / cd.nhalf, nh = rem - split * cd.nhalf;
  const int ksmask = (1 << cd.log2ks) - 1;
  const size_t wbase = ((size_t)(split * cd.nhalf + nh) * 4 + nt) * (NC_STAGES * 4) * 64;
  const f32x4* wh = cd.wh + wbase;
  const f32x4* wl = cd.wl + wbase;

  // ---- staging role: 16 lanes per input row (8 x 16 B of X_hi, 8 of X_lo) ----
  const int slot = tid & 15;
  // piece j = slot & 7 (8 channels) of a 64-channel quarter sits in chunk j >> 2 at 16-byte
  // unit j & 3, its lo half 4 units further (X layout: api.hip neck_carve)
  const f32x4* xplane = reinterpret_cast<const f32x4*>(p.xh) + ((slot & 7) >> 2) * 8 + (slot & 3) + (slot < 8 ? 0 : 4);
  // per output row: {index of its window origin in X (may be negative), (iy0+64)<<16 | ix0+64}
  // kept in LDS - eight rows' worth of loop-invariant registers would not fit
  if (tid < MT) {
    const int hw_o = g.ho * g.wo;
    const int r = mt * MT + tid;
    const int rc = min(r, g.M - 1);
    const int img = rc / hw_o, q = rc - img * hw_o;
    const int oy = q / g.wo, ox = q - oy * g.wo;
    const int iy0 = r < g.M ? 2 * oy - cd.pad : -64, ix0 = 2 * ox - cd.pad;  // iy0 = -64: never valid
    rowinfo[tid] = make_int2(img * g.HW + iy0 * g.wb + ix0, ((iy0 + 64) << 16) | (ix0 + 64));
  }
  __syncthreads();
  auto src_unit = [&](int2 info, int ky, int kx, int cq) -> size_t {
    const int iy = (info.y >> 16) - 64 + ky, ix = (info.y & 0xffff) - 64 + kx;
    const bool ok = (unsigned)iy < (unsigned)g.hb && (unsigned)ix < (unsigned)g.wb;
    const int m = -(int)ok;  // branch-free select (a branch here serialises the row-info reads)
    const int row = ((info.x + ky * g.wb + kx) & m) | (g.rows_in & ~m);
    return (size_t)row * 64 + cq * 16;  // 16-byte units: 64 per pixel, 16 per 64-channel quarter
  };
  f32x4 sreg[RTW] = {};
  auto stage_load = [&](int s, int jh) {  // rows (tid>>4) + 32*(RTW*jh + j), j < RTW, of stage s -> registers
    const int pix = NECK_PIX * split + (s >> 2), cq = s & 3;
    const int ky = pix >> cd.log2ks, kx = pix & ksmask;
    int2 info[RTW];
#pragma unroll
    for (int j = 0; j < RTW; ++j) info[j] = rowinfo[(tid >> 4) + HALF_ROWS * jh + 32 * j];
    if (NECK_ABL & 1) return;
#pragma unroll
    for (int j = 0; j < RTW; ++j) sreg[j] = xplane[src_unit(info[j], ky, kx, cq)];
  };
  auto stage_write = [&](int buf, int jh) {
    char* dst = smem + buf * BUF + ((tid >> 4) + HALF_ROWS * jh) * NC_ROWB + slot * 16;
    if (NECK_ABL & 8) return;
#pragma unroll
    for (int j = 0; j < RTW; ++j) *reinterpret_cast<f32x4*>(dst + 32 * j * NC_ROWB) = sreg[j];
  };

  f32x16 acc[RTW], cross[RTW];
#pragma unroll
  for (int t = 0; t < RTW; ++t) { acc[t] = f32x16{0}; cross[t] = f32x16{0}; }
  ConvB bf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { bf[kk].h = wh[kk * 64 + ln]; bf[kk].l = wl[kk * 64 + ln]; }

  stage_load(0, 0); stage_write(0, 0);
  stage_load(0, 1); stage_write(0, 1);
  __syncthreads();

  // Issue order inside a stage is pinned (sched_barrier): the compiler otherwise sinks
  // the run-ahead loads to their first use and the MFMAs wait a full L2 round trip.
  //   G0 (first row half of stage s+1) | k16 step 0, B(s+1, 0) | step 1, B(s+1, 1) |
  //   W0, G1 (second half) | step 2, B(s+1, 2) | step 3, B(s+1, 3) | W1 | barrier
  const int a_lane_off = (HALF_ROWS * rh + col) * NC_ROWB + 16 * half;
  for (int s = 0; s < NC_STAGES; ++s) {
    const int cur = s & 1;
    const bool more = s + 1 < NC_STAGES;
    const char* abase = smem + cur * BUF + a_lane_off;
    if (more) stage_load(s + 1, 0);
    f32x4 ah[RTW], al[RTW];
#pragma unroll
    for (int t = 0; t < RTW; ++t) {
      ah[t] = *reinterpret_cast<const f32x4*>(abase + t * 32 * NC_ROWB);
      al[t] = *reinterpret_cast<const f32x4*>(abase + t * 32 * NC_ROWB + 128);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f16x8 bh = __builtin_bit_cast(f16x8, bf[kk].h), bl = __builtin_bit_cast(f16x8, bf[kk].l);
#pragma unroll
      for (int tp = 0; tp < RTW; tp += 2) {
        constexpr bool dummy = false; (void)dummy;
        const bool pair = tp + 1 < RTW;   // compile-time after unrolling
        const int t1 = pair ? tp + 1 : tp;
        const f16x8 a0h = __builtin_bit_cast(f16x8, ah[tp]), a0l = __builtin_bit_cast(f16x8, al[tp]);
        const f16x8 a1h = __builtin_bit_cast(f16x8, ah[t1]), a1l = __builtin_bit_cast(f16x8, al[t1]);
        if (NECK_ABL & 4) {
          acc[tp][0] += (float)a0h[0] + (float)a0l[0] + (float)bh[0] + (float)bl[0];
          if (pair) acc[t1][0] += (float)a1h[0] + (float)a1l[0];
        } else {
          // (dependent MFMAs on one accumulator are kept two apart where a pair exists)
          acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh, acc[tp], 0, 0, 0);
          if (pair) acc[t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh, acc[t1], 0, 0, 0);
          cross[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl, cross[tp], 0, 0, 0);
          if (pair) cross[t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl, cross[t1], 0, 0, 0);
          cross[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh, cross[tp], 0, 0, 0);
          if (pair) cross[t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh, cross[t1], 0, 0, 0);
        }
        if (kk < 3) {  // these tiles' fragments for the next k16 step
#pragma unroll
          for (int t = tp; t <= t1; ++t) {
            ah[t] = *reinterpret_cast<const f32x4*>(abase + t * 32 * NC_ROWB + (kk + 1) * 32);
            al[t] = *reinterpret_cast<const f32x4*>(abase + t * 32 * NC_ROWB + (kk + 1) * 32 + 128);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more && !(NECK_ABL & 2)) {  // the same k16 step of the next stage into the slot just consumed
        bf[kk].h = wh[((s + 1) * 4 + kk) * 64 + ln];
        bf[kk].l = wl[((s + 1) * 4 + kk) * 64 + ln];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 1 && more) {
        stage_write(cur ^ 1, 0);
        stage_load(s + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) stage_write(cur ^ 1, 1);
    __syncthreads();
  }

  // ---- partial sums -> HBM ----
  float* out = cd.part + ((size_t)split * g.M) * cd.ncols + nh * 128 + nt * 32 + col;
#pragma unroll
  for (int t = 0; t < RTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * MT + HALF_ROWS * rh + 32 * t + crow(r, half);
      if (row < g.M) out[(size_t)row * cd.ncols] = fmaf(cross[t][r], SPLIT_INV, acc[t][r]);
    }
}

// ---------------------------------------------------------------------------
// k_neck_conv_rw : the same three convs with ROW-WINDOW staging
// ---------------------------------------------------------------------------
// k_neck_conv gathers, for every kernel pixel (ky, kx), one input pixel per output
// position: the stride-2 windows of neighbouring positions overlap ks/2-fold along x, so
// every input pixel of a row is fetched ks/2 times per ky.  Here a stage is
// (ky, x parity, 32-channel chunk): the pixels ix = 2*(ox + c) - pad + parity of the
// touched output rows are staged ONCE as "entries" (row r, column c), and the ks/2 taps
// kx = 2*tau + parity read entry c = ox + tau - i.e. the A operand of tap tau is the same
// LDS tile shifted by tau entries.  Gathered bytes per k16 step drop ks/2-fold
// (8x / 4x / 2x for the 16 / 8 / 4 kernels; the tile grows by (ks/2 - 1) entries per row).
// Work items, partial-sum slabs and the weight VALUES are those of k_neck_conv; only the
// order of the k16 steps inside a slice differs (pack_conv_rw in api.hip).
// Needs wo >= 16 (narrower maps touch too many rows per tile for the LDS / register budget).
constexpr int RW_ENT = 144;   // staged entry: 32 channels hi 64 B | lo 64 B | pad 16 B
static_assert(RW_ENT % 16 == 0 && (RW_ENT / 4) % 32 == 4, "conflict-free b128 entry stride");
// RTW = 32-row MFMA tiles per wave, NW = waves per workgroup: 8 (two per SIMD; wave = n-tile
// x row half) or 4 (one per SIMD with up to 512 registers; wave = n-tile over ALL the
// workgroup's rows - every weight fragment is fetched once per workgroup, not twice).
template <int RTW, int NW> struct RwShape {
  static constexpr int THREADS = 64 * NW;
  static constexpr int MT = 32 * RTW * (NW / 4);
  static constexpr int EPP = THREADS / 8;                             // entries per staging pass
  static constexpr int NE_MAX = MT + 7 * (MT / NECK_RW_MIN_WO + 2);   // 8 taps, wo >= NECK_RW_MIN_WO
  static constexpr int NPASS = (NE_MAX + EPP - 1) / EPP;
  // LDS slots: every output row's run of entries is padded so that the FIRST position of the
  // next row lands 17 slots after the LAST of this one (1 mod 16): the 32 positions of an
  // MFMA row tile then keep distinct slot indices mod 16 across row breaks and their
  // ds_read_b128 stay conflict-free (unpadded: 37 M conflict cycles per launch)
  static constexpr int SLOT_MAX = MT + 16 * (MT / NECK_RW_MIN_WO + 2);
  static constexpr int BUF = SLOT_MAX * RW_ENT;
};

// One body for the three kernel sizes: taps / rows-per-slice are run-time values and
// the k16 steps of a stage run as ks/4 groups of four (the depth of the weight ring).
template <int RTW, int NW>
__global__ __launch_bounds__(64 * NW) void k_neck_conv_rw(NeckConvLaunch p) {
  using SH = RwShape<RTW, NW>;
  constexpr int MT = SH::MT, HALF_ROWS = 32 * RTW, NPASS = SH::NPASS, BUF = SH::BUF, EPP = SH::EPP,
                THREADS = SH::THREADS;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  __shared__ int2 ent[SH::NE_MAX];
  const NeckGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar)
  const int half = lane >> 5, col = lane & 31;
  const unsigned ln = lane;   // last, 32-bit index of the weight-fragment addresses (SGPR base + lane offset)
  const int nt = wave & 3, rh = wave >> 2;

  const int logical = xcd_remap(blockIdx.x, p.nblocks);
#if NECK_RW_SLICE_MAJOR
  const int mtiles = p.nblocks / p.items_per_mt;
  const int it = logical / mtiles, mt = logical - it * mtiles;
#else
  const int mt = logical / p.items_per_mt, it = logical - mt * p.items_per_mt;
#endif
  const int ci = it >= p.conv[2].item0 ? 2 : (it >= p.conv[1].item0 ? 1 : 0);
  const NeckConvDesc& cd = p.conv[ci];
  const int rem = it - cd.item0;
  const int split = rem / cd.nhalf, nh = rem - split * cd.nhalf;
  const int taps = 1 << (cd.log2ks - 1);       // 2 / 4 / 8
  const int nky = NECK_PIX >> cd.log2ks;       // kernel rows in a 16-pixel slice: 4 / 2 / 1
  const int groups = taps >> 1;                // 4-step groups per stage: 1 / 2 / 4
  const int nstg = nky * 16;                   // stages: (ky in slice, x parity, 32-channel chunk)
  const size_t wbase = ((size_t)(split * cd.nhalf + nh) * 4 + nt) * (NC_STAGES * 4) * 64;
  const f32x4* wh = cd.wh_rw + wbase;
  const f32x4* wl = cd.wl_rw + wbase;

  // ---- entry geometry of this tile: positions [p0, plast] cover output rows row0 .. ----
  const int p0 = mt * MT, plast = min(p0 + MT, g.M) - 1;
  const int row0 = p0 / g.wo, ox_p0 = p0 - row0 * g.wo;
  const int per_row = g.wo + taps - 1, n0 = g.wo - ox_p0 + taps - 1;   // entries of row 0 / of a full row
  const int nrows = plast / g.wo - row0 + 1;
  const int ox_plast = plast - (plast / g.wo) * g.wo;
  // row 0 from ox_p0, full middle rows, the last row only up to the tile's last position
  const int NE = nrows == 1 ? ox_plast - ox_p0 + taps : n0 + (nrows - 2) * per_row + ox_plast + taps;
  const int padx = 17 - taps;                       // slot(e) = e + row(e) * padx
  for (int e = tid; e < NE; e += THREADS) {
    int r = 0, c = e;
    if (e >= n0) { r = 1 + (e - n0) / per_row; c = (e - n0) - (r - 1) * per_row; }
    const int grow = row0 + r;                      // output row over all images (< n_img * ho)
    const int img = grow / g.ho, oy = grow - img * g.ho;
    const int ox = (r == 0 ? ox_p0 : 0) + c;        // may run taps-1 past the row: bounds-checked below
    const int iy0 = 2 * oy - cd.pad, ix0 = 2 * ox - cd.pad;
    ent[e] = make_int2(img * g.HW + iy0 * g.wb + ix0, ((iy0 + 64) << 16) | (ix0 + 64));
  }
  int aoff[RTW];   // byte offset of this lane's tap-0 entry (+ its k half) per row tile
#pragma unroll
  for (int t = 0; t < RTW; ++t) {
    const int pos = min(p0 + HALF_ROWS * rh + 32 * t + col, plast);
    const int grow = pos / g.wo, ox = pos - grow * g.wo, r = grow - row0;
    const int e0 = r == 0 ? ox - ox_p0 : n0 + (r - 1) * per_row + ox;
    aoff[t] = (e0 + r * padx) * RW_ENT + 16 * half;
  }
  __syncthreads();

  // ---- staging role: 8 lanes per entry (4 x 16 B of X_hi, 4 of X_lo) ----
  const int piece = tid & 7, eslot = tid >> 3;
  // a stage's 32-channel chunk of a pixel is 128 contiguous bytes [hi | lo]: piece = 16-byte unit
  const char* xplane = reinterpret_cast<const char*>(p.xh) + piece * 16;
  f32x4 sreg[NPASS] = {};
  // Gather addresses depend on (ky, x parity) only - the eight 32-channel chunks of one
  // (ky, parity) are 64 B apart - so they are worked out once per eight stages.
  unsigned src[NPASS];                              // byte offsets of the gathered rows in a plane
  auto stage_addr = [&](int s) {
    const int ky = split * nky + (s >> 4), par = (s >> 3) & 1;
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int2 info = ent[min(q * EPP + eslot, NE - 1)];
      const int iy = (info.y >> 16) - 64 + ky, ix = (info.y & 0xffff) - 64 + par;
      const bool ok = (unsigned)iy < (unsigned)g.hb && (unsigned)ix < (unsigned)g.wb;
      const int m = -(int)ok;
      const int row = ((info.x + ky * g.wb + par) & m) | (g.rows_in & ~m);
      src[q] = (unsigned)row * 1024u;               // (rows_in + 1) * 1024 <= 2^32 (make_neck_geom)
    }
  };
  // (every load in the main loop is unconditional - indices clamped instead: a load behind
  //  a branch makes the wait counts at the join pessimistic, vmcnt(0) in effect)
  auto stage_load = [&](int s) {                    // prologue only: every pass at once
    stage_addr(s);
#pragma unroll
    for (int q = 0; q < NPASS; ++q) sreg[q] = *reinterpret_cast<const f32x4*>(xplane + src[q] + (s & 7) * 128);
  };
  // (unconditional stores too: lanes past the last entry hold its data - the clamped table
  //  read above - and rewrite it; a store under a lane mask is a branch to the wait counter)
  int woff[NPASS];                                  // LDS byte offset of this thread's piece per pass
#pragma unroll
  for (int q = 0; q < NPASS; ++q) {
    const int e = min(q * EPP + eslot, NE - 1);
    const int r = e < n0 ? 0 : 1 + (e - n0) / per_row;
    woff[q] = (e + r * padx) * RW_ENT + piece * 16;
  }
  auto stage_write = [&](int buf) {
    char* dst = smem + buf * BUF;
    if (NECK_ABL & 8) return;
#pragma unroll
    for (int q = 0; q < NPASS; ++q) *reinterpret_cast<f32x4*>(dst + woff[q]) = sreg[q];
  };

  f32x16 acc[RTW], cross[RTW];
#pragma unroll
  for (int t = 0; t < RTW; ++t) { acc[t] = f32x16{0}; cross[t] = f32x16{0}; }
  stage_load(0); stage_write(0);
  __syncthreads();

  // The stage loop exists once per kernel size (GR = k16 steps per stage / 4), chosen by a
  // workgroup-uniform switch: with the step count a run-time value the three paths share
  // their joins, and hipcc's wait-count merge then drains the whole weight ring before
  // every staging store (vmcnt(0) at each stage end - a third of the kernel's time).
  auto stages = [&](auto gr_tag) {
    constexpr int GR = decltype(gr_tag)::value, KSTEPS = 4 * GR;
    // weight ring: RING k16 steps in flight (a divisor of KSTEPS, so slots are static);
    // the one-wave-per-SIMD shape has the registers - and no second wave - for 8
    constexpr int RING = (NW == 4 && GR >= 2 && NECK_RING8) ? 8 : 4;
    ConvB bf[RING];
#pragma unroll
    for (int kk = 0; kk < RING; ++kk) { bf[kk].h = wh[kk * 64 + ln]; bf[kk].l = wl[kk * 64 + ln]; }
    int gstep = RING;                               // next k16 step to fetch into the ring
    for (int s = 0; s < nstg; ++s) {
      const int cur = s & 1;
      const char* abase = smem + cur * BUF;
      // the next stage's gathers go out after the first k16 step (NECK_SPREAD: one pass at
      // a time between the steps instead - slower); the last stage re-stages itself: unused
      const int sn = min(s + 1, nstg - 1), cqn = (sn & 7) * 128;
      if ((sn & 7) == 0) stage_addr(sn);
      f32x4 ah[RTW], al[RTW];
#pragma unroll
      for (int t = 0; t < RTW; ++t) {
        ah[t] = *reinterpret_cast<const f32x4*>(abase + aoff[t]);
        al[t] = *reinterpret_cast<const f32x4*>(abase + aoff[t] + 64);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int st = 0; st < KSTEPS; ++st) {         // tap st >> 1, 16-channel step st & 1
        constexpr int dummy = 0; (void)dummy;
        const int kk = st & (RING - 1);
        const f16x8 bh = __builtin_bit_cast(f16x8, bf[kk].h), bl = __builtin_bit_cast(f16x8, bf[kk].l);
        const int nxt = ((st + 1) >> 1) * RW_ENT + ((st + 1) & 1) * 32;   // the k16 step after this one
#pragma unroll
        for (int tp = 0; tp < RTW; tp += 2) {
          const bool pair = tp + 1 < RTW;
          const int t1 = pair ? tp + 1 : tp;
          const f16x8 a0h = __builtin_bit_cast(f16x8, ah[tp]), a0l = __builtin_bit_cast(f16x8, al[tp]);
          const f16x8 a1h = __builtin_bit_cast(f16x8, ah[t1]), a1l = __builtin_bit_cast(f16x8, al[t1]);
          acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh, acc[tp], 0, 0, 0);
          if (pair) acc[t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh, acc[t1], 0, 0, 0);
          cross[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl, cross[tp], 0, 0, 0);
          if (pair) cross[t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl, cross[t1], 0, 0, 0);
          cross[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh, cross[tp], 0, 0, 0);
          if (pair) cross[t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh, cross[t1], 0, 0, 0);
          if (st + 1 < KSTEPS && !(NECK_ABL & 16)) {
#pragma unroll
            for (int t = tp; t <= t1; ++t) {
              ah[t] = *reinterpret_cast<const f32x4*>(abase + aoff[t] + nxt);
              al[t] = *reinterpret_cast<const f32x4*>(abase + aoff[t] + nxt + 64);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!(NECK_ABL & 1)) {                      // gather passes of the next stage due at this step
#pragma unroll
          for (int q = 0; q < NPASS; ++q)
            if (NECK_SPREAD ? q * (KSTEPS - 4 > 0 ? KSTEPS - 4 : 1) / NPASS == st : st == 0) sreg[q] = *reinterpret_cast<const f32x4*>(xplane + src[q] + cqn);
        }
        const int gi = min(gstep, NC_STAGES * 4 - 1);   // refill the ring slot just consumed
        if (!(NECK_ABL & 2)) {
          bf[kk].h = wh[(size_t)gi * 64 + ln];
          bf[kk].l = wl[(size_t)gi * 64 + ln];
        }
        ++gstep;
        __builtin_amdgcn_sched_barrier(0);
      }
      stage_write(cur ^ 1);
      if (!(NECK_ABL & 32)) __syncthreads();
    }
  };
  if (groups == 4) stages(std::integral_constant<int, 4>{});
  else if (groups == 2) stages(std::integral_constant<int, 2>{});
  else stages(std::integral_constant<int, 1>{});

  float* out = cd.part + ((size_t)split * g.M) * cd.ncols + nh * 128 + nt * 32 + col;
#pragma unroll
  for (int t = 0; t < RTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * MT + HALF_ROWS * rh + 32 * t + crow(r, half);
      if (row < g.M) out[(size_t)row * cd.ncols] = fmaf(cross[t][r], SPLIT_INV, acc[t][r]);
    }
}

// Rows per workgroup: minimise (rounds over the CUs) x (per-item cost ~ rows + a fixed
// part for the 2 MB weight slab every item streams).  16 maps of 40x40 (6400 positions,
// 22 items per tile): 256 rows -> 550 items = 3 rounds; 192 rows -> 748 items = 3 rounds
// of 3/4 the work each.
int neck_conv_rows(int M, int items_per_mt, int num_cus, int max_rows) {
  int best = NECK_MT;
  long best_cost = -1;
  for (int rows : {256, 192, 128}) {
    if (rows > max_rows) continue;
    const long items = (long)items_per_mt * ((M + rows - 1) / rows);
    const long rounds = (items + num_cus - 1) / num_cus;
    const long cost = rounds * (rows + 32);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = rows; }
  }
  return best;
}

hipError_t launch_neck_conv(const NeckConvLaunch& p, hipStream_t s) {
  const dim3 grid(p.nblocks), block(512);
  if (p.row_window) {   // (the 256-row shape of this kernel spills - 107 VGPRs - and is not built)
    if (p.row_window == 2) {   // one wave per SIMD (its 256-row shape needs 256 + 256 registers and spills: not built)
      switch (p.mt_rows) {
        case 192: hipLaunchKernelGGL((k_neck_conv_rw<6, 4>), grid, dim3(256), 0, s, p); break;
        case 128: hipLaunchKernelGGL((k_neck_conv_rw<4, 4>), grid, dim3(256), 0, s, p); break;
        default: return hipErrorInvalidValue;
      }
      return hipGetLastError();
    }
    switch (p.mt_rows) {
      case 192: hipLaunchKernelGGL((k_neck_conv_rw<3, 8>), grid, block, 0, s, p); break;
      case 128: hipLaunchKernelGGL((k_neck_conv_rw<2, 8>), grid, block, 0, s, p); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (p.mt_rows) {
    case 256: hipLaunchKernelGGL((k_neck_conv<4>), grid, block, 0, s, p); break;
    case 192: hipLaunchKernelGGL((k_neck_conv<3>), grid, block, 0, s, p); break;
    case 128: hipLaunchKernelGGL((k_neck_conv<2>), grid, block, 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_neck_out
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_neck_out(NeckOutLaunch p) {
  __shared__ __attribute__((aligned(16))) float smem[NP_PAR_OFF];
  Range rg;
  const ATile<GM_SPLIT> A(smem, LDH, LDHH, &rg);
  float* S0 = smem + NP_S0_OFF;
  const NeckGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), col = lane & 31;
  const int r0 = blockIdx.x * TM;

  using WS = WStream<GM_SPLIT, 1>;
  WS ws;
  ws.set_lane(lane);
  f32x16 acc[1];
  {
    const float b = p.bias2[32 * wave + col];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = b;
  }
  // the 512-channel concat of this tile: bias + partial sums in slice order
  {
    const int lrow = tid >> 4, lpart = tid & 15;
    const int r = r0 + lrow;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = 4 * (i * 16 + lpart);
      const int ci = c4 < 256 ? 0 : (c4 < 384 ? 1 : 2);
      const int cc = c4 - (ci == 0 ? 0 : (ci == 1 ? 256 : 384));
      const int ncols = ci == 0 ? 256 : 128;
      f32x4 v = *reinterpret_cast<const f32x4*>(p.bias[ci] + cc);
      if (r < g.M) {
        const float* src = p.part[ci] + (size_t)r * ncols + cc;
        for (int sidx = 0; sidx < p.nsplit[ci]; ++sidx)
          v += *reinterpret_cast<const f32x4*>(src + (size_t)sidx * g.M * ncols);
      }
      A.put4(lrow, c4, v);
    }
  }
  ws.template prime<512, 0>(p.wh, p.wl, wave, lane);
  __syncthreads();
  ws.template gemm<512, 0, 0>(A, p.wh, p.wl, wave, lane, acc, nullptr, nullptr, 0, 0);
  acc_to_lds<1>(S0, LDA, 32 * wave, lane, acc);
  __syncthreads();
  if (p.tokens) {   // token-major: the rows as they are (256 contiguous floats each)
    const int lrow = tid >> 4, lpart = tid & 15;
    const int r = r0 + lrow;
    if (r < g.M) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c4 = 4 * (i * 16 + lpart);
        *reinterpret_cast<f32x4*>(p.tokens + (size_t)r * C + c4) = *reinterpret_cast<const f32x4*>(&S0[lrow * LDA + c4]);
      }
    }
  } else {
  // transposed store: one channel, 32 consecutive positions per half-wave
  {
    const int pos = tid & 31, cg = tid >> 5;
    const int r = r0 + pos;
    if (r < g.M) {
      const int hw_o = g.ho * g.wo;
      const int img = r / hw_o, q = r - img * hw_o;
      float* dst = p.feat + (size_t)img * C * hw_o + q;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = cg * 16 + i;
        dst[(size_t)c * hw_o] = S0[pos * LDA + c];
      }
    }
  }
  }
  range_report<GM_SPLIT>(rg, p.flags);
}

hipError_t launch_neck_out(const NeckOutLaunch& p, hipStream_t s) {
  hipLaunchKernelGGL(k_neck_out, dim3((p.g.M + TM - 1) / TM), dim3(512), 0, s, p);
  return hipGetLastError();
}

}  // namespace oetr
