// Gate-1 prototype of the token-split encoder decomposition (round 5).
//
// The chain  merge -> +x -> LN2 -> MLP1 -> GELU -> MLP2 -> +x  of one EncoderLayer
// (reference src/models/transformer.py:131-142) as a stand-alone kernel in the NEW shape:
//   * one workgroup = 4 waves (one per SIMD, 512 registers each) = 128 tokens,
//   * a wave owns 32 TOKENS x all 256 channels; every GEMM runs transposed
//     (weights = MFMA A operand, activations = B operand), so the accumulator registers of
//     one GEMM are, after a register-only conversion to split-f16 planes, the B fragments of
//     the next one: activations never touch LDS, LayerNorm is per-lane + one half exchange,
//   * LDS holds ONLY a ring of weight fragments that all four waves read; it is filled by
//     LDS-DMA (global_load_lds_dwordx4) in consumption order, one s_barrier per ring group,
//   * fp32-class split products on ONE accumulator: the accumulator lives in a 2^11-scaled
//     domain (weights carry three f16 planes WH*2^11, WH, WL*2^11; activations two: hi, lo*2^11),
//     acc += (WH 2^11).bh + WH.(bl 2^11) + (WL 2^11).bh = 2^11 (w.b) to fp32 class, all scalings exact.
//
// Build: tools/chain128/build.sh ; run: tools/chain128/run.py (GPU box).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace c128 {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 256, FF = 512, NW = 4, TOK_WG = 128;
constexpr float S11 = 2048.0f, S11_INV = 1.0f / 2048.0f, LN_EPS = 1e-5f;

struct Params {
  const float* msg;      // [T][256] attention message (merge input), token-major
  const float* x;        // [T][256] residual stream
  float* y;              // [T][256]
  const char* wstream;   // weight fragments in consumption order (+ padding groups)
  const float* ln_g;     // [256]
  const float* ln_b;     // [256]
  int T;
  int flags;             // 1: no global input / output traffic (timing attribution only)
  unsigned long long* tbuf;  // optional per-workgroup phase stamps (16 per workgroup)
};

// ---- arithmetic helpers (same forms as csrc/common.h) ----
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo, float& fm) {
  const f32x2 ab = f32x2{a, b};
  const f16x2 h = __builtin_convertvector(ab, f16x2);
  const f32x2 sc = ab * f32x2{S11, S11};
  lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)h[0], -S11, sc[0]),
                                                                __builtin_fmaf((float)h[1], -S11, sc[1])));
  hi = __builtin_bit_cast(uint32_t, h);
  fm = __builtin_fmaxf(__builtin_fmaxf(fm, __builtin_fabsf(a)), __builtin_fabsf(b));   // range guard (v_max3)
}
constexpr float GELU_T = 5.5f;
constexpr float GQ0 = -1.151147082e+00f, GQ1 = -4.589156863e-01f, GQ2 = -5.323818670e-02f,
                GQ3 = 7.977462822e-03f, GQ4 = -7.398742635e-04f, GQ5 = 2.992419385e-05f;
__device__ __forceinline__ float gelu1(float x) {
  const float ax = __builtin_fabsf(x), w = __builtin_fminf(ax, GELU_T);
  float q = GQ5;
  q = __builtin_fmaf(q, w, GQ4); q = __builtin_fmaf(q, w, GQ3); q = __builtin_fmaf(q, w, GQ2);
  q = __builtin_fmaf(q, w, GQ1); q = __builtin_fmaf(q, w, GQ0);
  const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(w, q, -1.0f));
  return __builtin_fmaf(-ax, e, __builtin_fmaxf(x, 0.f));
}
__device__ __forceinline__ f32x2 gelu2pk(f32x2 v) {
  const f32x2 ax = __builtin_elementwise_abs(v);
  const f32x2 w = f32x2{__builtin_fminf(ax[0], GELU_T), __builtin_fminf(ax[1], GELU_T)};
  f32x2 q = f32x2{GQ5, GQ5};
  q = __builtin_elementwise_fma(q, w, f32x2{GQ4, GQ4}); q = __builtin_elementwise_fma(q, w, f32x2{GQ3, GQ3});
  q = __builtin_elementwise_fma(q, w, f32x2{GQ2, GQ2}); q = __builtin_elementwise_fma(q, w, f32x2{GQ1, GQ1});
  q = __builtin_elementwise_fma(q, w, f32x2{GQ0, GQ0});
  const f32x2 u = __builtin_elementwise_fma(w, q, f32x2{-1.f, -1.f});
  f32x2 g;
  g[0] = __builtin_fmaf(-ax[0], __builtin_amdgcn_exp2f(u[0]), __builtin_fmaxf(v[0], 0.f));
  g[1] = __builtin_fmaf(-ax[1], __builtin_amdgcn_exp2f(u[1]), __builtin_fmaxf(v[1], 0.f));
  return g;
}

// ---- MFMAs as inline asm with explicit register files (gfx950: 256 arch VGPRs + 256 AGPRs per lane at one
// wave per SIMD).  hipcc's MFMA builtin takes A / B from VGPRs only, and VALU instructions address VGPRs only -
// with 128 registers of B planes, the weight fragments in flight and the GELU temporaries the 256 VGPRs
// overflow and every GEMM loop is scheduled around spills.  Here the register file of every operand is chosen:
//   residual / output accumulators (128)  AGPR   C/D of merge and MLP2
//   LN2 planes = B of MLP1 (128)          AGPR   written once per tile (v_accvgpr_write), read by MFMAs only
//   hidden accumulators (2 x 32)          VGPR   C/D of MLP1; GELU reads them without v_accvgpr_read
//   hidden planes (32), fragments (48)    VGPR
// hipcc pads no hazards around asm (cdna_hip_programming.md 5.7): the producers of A / B operands are kept at
// least one sched_barrier-separated step away from their MFMA, and accumulators pass through mfma_fence()
// (>= 18 wait states) before any non-MFMA instruction touches them.
#define C128_MFMA "v_mfma_f32_32x32x16_f16 "
// Two MFMAs that share the B operand, on two accumulators, behind two wait states: whatever hipcc
// placed just before the statement (a register copy, a reload, a v_accvgpr_write assembling an
// operand tuple) has then settled - VALU write -> MFMA operand needs them and hipcc pads nothing
// in front of asm.  CF / BF: register file of C/D and of B ('a' AGPR, 'v' VGPR).
#define C128_MMA2(NAME, CF, BF)                                                                          \
  __device__ __forceinline__ void NAME(f32x16& c0, f32x16& c1, const f32x4& a0, const f32x4& a1,        \
                                       const f32x4& b) {                                                 \
    asm volatile("s_nop 1\n\t" C128_MFMA "%0, %2, %4, %0\n\t" C128_MFMA "%1, %3, %4, %1"                 \
                 : "+" CF(c0), "+" CF(c1)                                                                \
                 : "v"(a0), "v"(a1), BF(b));                                                             \
  }
C128_MMA2(mma2_av, "a", "v")   // merge:  C/D AGPR, B VGPR
C128_MMA2(mma2_aa, "a", "a")   // MLP2:   C/D AGPR, B AGPR
C128_MMA2(mma2_vv, "v", "v")   // MLP1:   C/D VGPR, B VGPR (hi plane)
C128_MMA2(mma2_va, "v", "a")   // MLP1:   C/D VGPR, B AGPR (lo plane)
__device__ __forceinline__ void mma2_vv0(f32x16& c0, f32x16& c1, const f32x4& a0, const f32x4& a1, const f32x4& b) {
  asm volatile("s_nop 1\n\t" C128_MFMA "%0, %2, %4, 0\n\t" C128_MFMA "%1, %3, %4, 0"     // chain start: C = 0
               : "=&v"(c0), "=&v"(c1)
               : "v"(a0), "v"(a1), "v"(b));
}
__device__ __forceinline__ void mfma_fence_v(f32x16& x, f32x16& y) {
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ void mfma_fence_a(f32x16 (&acc)[8]) {
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3"
               : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]));
}
__device__ __forceinline__ f32x4 to_agpr(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  f32x4 a;
  asm volatile("v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
               : "=a"(a[0]), "=a"(a[1]), "=a"(a[2]), "=a"(a[3])
               : "v"(w0), "v"(w1), "v"(w2), "v"(w3));
  return a;
}

__device__ __forceinline__ f32x4 pack4(const uint32_t (&w)[4]) {
  return __builtin_bit_cast(f32x4, u32x4{w[0], w[1], w[2], w[3]});
}
// One weight fragment: the A operand (32 output channels x 16 k-slots) of one (m-tile, k16-step):
// planes WH 2^11, WH, WL 2^11.
template <int PL> struct Frag { f32x4 p[PL]; };

// Two fragments that share the B operand, on two accumulators (2^11 domain):
//   acc += (WH 2^11).bh + WH.(bl 2^11) + (WL 2^11).bh
// KIND 0: merge (C/D AGPR, B VGPR);  1: MLP1 (C/D VGPR, bh VGPR, bl AGPR);  2: as 1, first step of a chain
// (C = 0);  3: MLP2 (C/D AGPR, B AGPR)
template <int KIND>
__device__ __forceinline__ void triple2(f32x16& a0, f32x16& a1, const Frag<3>& w0, const Frag<3>& w1,
                                        const f32x4& bh, const f32x4& bl) {
  if constexpr (KIND == 0) {
    mma2_av(a0, a1, w0.p[0], w1.p[0], bh);
    mma2_av(a0, a1, w0.p[1], w1.p[1], bl);
    mma2_av(a0, a1, w0.p[2], w1.p[2], bh);
  } else if constexpr (KIND == 3) {
    mma2_aa(a0, a1, w0.p[0], w1.p[0], bh);
    mma2_aa(a0, a1, w0.p[1], w1.p[1], bl);
    mma2_aa(a0, a1, w0.p[2], w1.p[2], bh);
  } else {
    if constexpr (KIND == 2) mma2_vv0(a0, a1, w0.p[0], w1.p[0], bh);
    else mma2_vv(a0, a1, w0.p[0], w1.p[0], bh);
    mma2_va(a0, a1, w0.p[1], w1.p[1], bl);
    mma2_vv(a0, a1, w0.p[2], w1.p[2], bh);
  }
}

// ---- the weight ring ----
template <int PL, int G, int R, int DBG = 0>
struct Ring {
  static constexpr int FRAG_B = PL * 1024, GROUP_B = G * FRAG_B, P = GROUP_B / (NW * 1024), BYTES = R * GROUP_B;
  static_assert(GROUP_B % (NW * 1024) == 0, "a group splits into 1-KB pieces over the four waves");
  static_assert(R >= 4, "ring depth");
  static constexpr int WAITN = (R - 3) * P;   // DMA pieces of this wave allowed in flight at a sync
  static_assert(WAITN <= 63, "vmcnt field");

  const char* src;        // next group to issue (this wave's share), uniform
  unsigned lds0;          // LDS byte address of the ring
  unsigned issue_slot;    // byte offset of the slot the next issue fills
  unsigned wave_off;      // wave * P * 1024
  unsigned voff;          // lane * 16
  unsigned rd_cur, rd_next;   // byte offsets of the group being read / the next one
  const char* ring_ptr;   // generic pointer to the ring (for ds_read)

  __device__ __forceinline__ void init(const char* wstream, char* ring, int wave, int lane) {
    wave_off = (unsigned)wave * P * 1024;
    src = wstream + wave_off;
    lds0 = (unsigned)(uintptr_t)ring;
    issue_slot = 0;
    voff = (unsigned)lane * 16;
    rd_cur = 0;
    rd_next = GROUP_B;
    ring_ptr = ring;
  }
  // issue this wave's P pieces of the next group: ONE statement (m0 saved / restored once)
  __device__ __forceinline__ void issue() {
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + issue_slot + wave_off);
    unsigned keep;
#define C128_PIECE(n) "s_mov_b32 m0, %[d" #n "]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v" #n "], %[src]\n\t"
    if constexpr (P == 6) {
      asm volatile("s_mov_b32 %[keep], m0\n\t" C128_PIECE(0) C128_PIECE(1) C128_PIECE(2) C128_PIECE(3) C128_PIECE(4) C128_PIECE(5)
                   "s_mov_b32 m0, %[keep]"
                   : [keep] "=&s"(keep)
                   : [src] "s"(src), [d0] "s"(dst), [d1] "s"(dst + 1024), [d2] "s"(dst + 2048), [d3] "s"(dst + 3072),
                     [d4] "s"(dst + 4096), [d5] "s"(dst + 5120), [v0] "v"(voff), [v1] "v"(voff + 1024), [v2] "v"(voff + 2048),
                     [v3] "v"(voff + 3072), [v4] "v"(voff + 4096), [v5] "v"(voff + 5120)
                   : "memory");
    } else if constexpr (P == 4) {
      asm volatile("s_mov_b32 %[keep], m0\n\t" C128_PIECE(0) C128_PIECE(1) C128_PIECE(2) C128_PIECE(3)
                   "s_mov_b32 m0, %[keep]"
                   : [keep] "=&s"(keep)
                   : [src] "s"(src), [d0] "s"(dst), [d1] "s"(dst + 1024), [d2] "s"(dst + 2048), [d3] "s"(dst + 3072),
                     [v0] "v"(voff), [v1] "v"(voff + 1024), [v2] "v"(voff + 2048), [v3] "v"(voff + 3072)
                   : "memory");
    } else if constexpr (P == 3) {
      asm volatile("s_mov_b32 %[keep], m0\n\t" C128_PIECE(0) C128_PIECE(1) C128_PIECE(2)
                   "s_mov_b32 m0, %[keep]"
                   : [keep] "=&s"(keep)
                   : [src] "s"(src), [d0] "s"(dst), [d1] "s"(dst + 1024), [d2] "s"(dst + 2048),
                     [v0] "v"(voff), [v1] "v"(voff + 1024), [v2] "v"(voff + 2048)
                   : "memory");
    } else {
      static_assert(P == 2, "pieces per wave and group");
      asm volatile("s_mov_b32 %[keep], m0\n\t" C128_PIECE(0) C128_PIECE(1)
                   "s_mov_b32 m0, %[keep]"
                   : [keep] "=&s"(keep)
                   : [src] "s"(src), [d0] "s"(dst), [d1] "s"(dst + 1024), [v0] "v"(voff), [v1] "v"(voff + 1024)
                   : "memory");
    }
#undef C128_PIECE
    src += GROUP_B;
    issue_slot = issue_slot + GROUP_B == (unsigned)BYTES ? 0u : issue_slot + GROUP_B;
  }
  // Makes the NEXT group readable (every wave's share landed), then refills the slot of the group
  // before the current one.  Called once per group, before the first read of the next group.
  __device__ __forceinline__ void sync() {
    if constexpr ((DBG & 3) != 0) return;
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(WAITN) : "memory");
    issue();
  }
  __device__ __forceinline__ void prologue() {
#pragma unroll
    for (int q = 0; q < R - 2; ++q) issue();
    sync();   // group 0 readable; issues group R-2
  }
  __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __device__ __forceinline__ void advance_group() {
    rd_cur = rd_next;
    rd_next = rd_next + GROUP_B == (unsigned)BYTES ? 0u : rd_next + GROUP_B;
  }
  // fragment j of the current (NEXT = false) or next (true) group
  template <bool NEXT>
  __device__ __forceinline__ Frag<PL> read(int j) const {
    const f32x4* g = reinterpret_cast<const f32x4*>(ring_ptr + (NEXT ? rd_next : rd_cur) + voff);
    Frag<PL> f;
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) f.p[pl] = g[(j * PL + pl) * 64];
    return f;
  }
};

// Consume NF fragments (a multiple of G, G even) as pairs.  `cur` holds the first pair on entry and the
// first pair of the following segment on exit.  body(jp, fragA, fragB) for jp = 0 .. NF/2-1.
template <int NF, int PL, int G, int R, int DBG, class Body>
__device__ __forceinline__ void segment(Ring<PL, G, R, DBG>& ring, Frag<PL> (&cur)[2], Body&& body) {
  static_assert(NF % G == 0 && G % 2 == 0, "segments cover whole groups");
#pragma unroll
  for (int j = 0; j < NF; j += 2) {
    if (j % G == 0) ring.sync();           // next group readable from here on
    Frag<PL> nxt[2];
    if constexpr ((DBG & 3) == 1) {
      nxt[0] = cur[0];
      nxt[1] = cur[1];
    } else if ((j + 2) % G != 0) {
      nxt[0] = ring.template read<false>((j + 2) % G);
      nxt[1] = ring.template read<false>((j + 3) % G);
    } else {
      nxt[0] = ring.template read<true>(0);
      nxt[1] = ring.template read<true>(1);
      ring.advance_group();
    }
    __builtin_amdgcn_sched_barrier(0);     // next pair's LDS reads are in flight while this pair's MFMAs run
    body(j >> 1, cur[0], cur[1]);
    __builtin_amdgcn_sched_barrier(0);
    cur[0] = nxt[0];
    cur[1] = nxt[1];
  }
}

#define STAMP(i) do { if (p.tbuf && threadIdx.x == 0) p.tbuf[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

// PL: weight planes in the stream (3, or 2 with WH 2^11 formed in registers); G: fragments per ring
// group; R: ring groups; GPK: packed GELU polynomial; NSL1: GELU slices (of 8) issued beside MLP1.
template <int PL, int G, int R, int GPK, int NSL1, int DBG = 0>
__global__ __launch_bounds__(256, 1) void k_chain128(const Params p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  typedef Ring<PL, G, R, DBG> RingT;
  char* ring_mem = smem;
  float* lnp = reinterpret_cast<float*>(smem + RingT::BYTES);   // gamma[256], beta[256]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  const int tok = blockIdx.x * TOK_WG + wave * 32 + (lane & 31);
  const int tokc = (p.flags & 1) ? (lane & 31) : tok < p.T ? tok : p.T - 1;
  STAMP(0);

  lnp[tid] = p.ln_g[tid];
  lnp[256 + tid] = p.ln_b[tid];
  __syncthreads();

  RingT ring;
  ring.init(p.wstream, ring_mem, wave, lane);
  ring.prologue();
  Frag<PL> cur[2];
  cur[0] = ring.template read<false>(0);
  cur[1] = ring.template read<false>(1);

  float fm = 0.f;
  // ---- residual -> accumulators (register 4 j + i of m-tile mt <-> channel 32 mt + 8 j + 4 h + i), 2^11 domain
  f32x16 acc[8];
  {
    const float* xrow = p.x + (size_t)tokc * C + 4 * h;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + 32 * mt + 8 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mt][4 * j + i] = v[i] * S11;
      }
      if (mt & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- merge input -> B planes (k-slot i of step s <-> channel 16 s + 8 (i >> 2) + 4 h + (i & 3)), four k16
  //      steps at a time, one batch ahead of the merge GEMM (stream order [s][mt], pairs (mt, mt + 1))
  {
    const float* mrow = p.msg + (size_t)tokc * C + 4 * h;
    f32x4 raw[8], mh[4], ml[4];
    auto load4 = [&](int s0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        raw[2 * s] = *reinterpret_cast<const f32x4*>(mrow + 16 * (s0 + s));
        raw[2 * s + 1] = *reinterpret_cast<const f32x4*>(mrow + 16 * (s0 + s) + 8);
      }
    };
    auto cvt4 = [&]() {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        uint32_t hi[4], lo[4];
        const f32x4 a0 = raw[2 * s], a1 = raw[2 * s + 1];
        split2(a0[0], a0[1], hi[0], lo[0], fm); split2(a0[2], a0[3], hi[1], lo[1], fm);
        split2(a1[0], a1[1], hi[2], lo[2], fm); split2(a1[2], a1[3], hi[3], lo[3], fm);
        mh[s] = pack4(hi);
        ml[s] = pack4(lo);
      }
    };
    load4(0);
    cvt4();
    STAMP(1);
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
      if (sb < 3) load4(4 * sb + 4);
      __builtin_amdgcn_sched_barrier(0);
      segment<32>(ring, cur, [&](int jp, const Frag<PL>& w0, const Frag<PL>& w1) {
        triple2<0>(acc[(2 * jp) & 7], acc[(2 * jp + 1) & 7], w0, w1, mh[jp >> 2], ml[jp >> 2]);
      });
      if (sb < 3) cvt4();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  mfma_fence_a(acc);
  STAMP(2);

  // ---- LN2 over the 256 channels of a token: 128 registers here + 128 in lane ^ 32; hi planes in VGPRs, lo planes -> AGPRs
  f32x4 lnh[16], lnl[16];
  {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) { s0 += acc[mt][r]; s1 += acc[mt][r + 1]; }
    float sum = s0 + s1;
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / C);    // 2^11 domain
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float d0 = acc[mt][r] - mean, d1 = acc[mt][r + 1] - mean;
        q0 = __builtin_fmaf(d0, d0, q0);
        q1 = __builtin_fmaf(d1, d1, q1);
      }
    float var = q0 + q1;
    var += __shfl_xor(var, 32, 64);
    var *= (1.0f / C) * (S11_INV * S11_INV);     // true variance
    const float rs = S11_INV / __builtin_sqrtf(var + LN_EPS);   // rstd * 2^-11
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      uint32_t hi[2][4], lo[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(lnp + 32 * mt + 8 * j + 4 * h);
        const f32x4 b = *reinterpret_cast<const f32x4*>(lnp + 256 + 32 * mt + 8 * j + 4 * h);
        float n[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) n[i] = __builtin_fmaf((acc[mt][4 * j + i] - mean) * rs, g[i], b[i]);
        split2(n[0], n[1], hi[j >> 1][2 * (j & 1)], lo[j >> 1][2 * (j & 1)], fm);
        split2(n[2], n[3], hi[j >> 1][2 * (j & 1) + 1], lo[j >> 1][2 * (j & 1) + 1], fm);
      }
      lnh[2 * mt] = pack4(hi[0]);
      lnl[2 * mt] = to_agpr(lo[0][0], lo[0][1], lo[0][2], lo[0][3]);
      lnh[2 * mt + 1] = pack4(hi[1]);
      lnl[2 * mt + 1] = to_agpr(lo[1][0], lo[1][1], lo[1][2], lo[1][3]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  STAMP(3);

  // ---- MLP: hidden chunk PAIRS (64 channels = 2 m-tiles of MLP1 = 4 k16 steps of MLP2);
  //      stream M1(0) [M1(cp) M2(cp-1)]_{cp=1..7} M2(7);  M1(cp): [s][m], M2(cp): [ks][mt]
  uint32_t hwh[4], hwl[4];         // words of the hidden-plane step under construction
  f32x4 hph[4], hpl[4];            // hidden planes of one chunk pair (AGPRs): four k16 steps of MLP2
  // registers 2 (i & 7), +1 of tile (i >> 3) -> word (i & 3) of step i >> 2
  auto gelu_slice = [&](const f32x16& t0, const f32x16& t1, int i) {
    const int r = 2 * (i & 7);
    const float v0 = (i >> 3) ? t1[r] : t0[r], v1 = (i >> 3) ? t1[r + 1] : t0[r + 1];
    float g0, g1;
    if constexpr ((DBG & 4) != 0) {
      g0 = v0; g1 = v1;
    } else if constexpr (GPK) {
      const f32x2 g = gelu2pk(f32x2{v0, v1} * f32x2{S11_INV, S11_INV});
      g0 = g[0]; g1 = g[1];
    } else {
      g0 = gelu1(v0 * S11_INV);
      g1 = gelu1(v1 * S11_INV);
    }
    split2(g0, g1, hwh[i & 3], hwl[i & 3], fm);
    if ((i & 3) == 3) {
      hph[i >> 2] = to_agpr(hwh[0], hwh[1], hwh[2], hwh[3]);
      hpl[i >> 2] = to_agpr(hwl[0], hwl[1], hwl[2], hwl[3]);
    }
  };
  // MLP1 of one chunk pair (n0, n1) beside GELU slices 0 .. NSL1-1 of the previous pair (p0, p1)
  auto m1 = [&](auto have_prev, f32x16& n0, f32x16& n1, const f32x16& p0, const f32x16& p1) {
    segment<32>(ring, cur, [&](int jp, const Frag<PL>& w0, const Frag<PL>& w1) {
      if constexpr (decltype(have_prev)::value) {
        if ((jp * NSL1) % 16 < NSL1) gelu_slice(p0, p1, (jp * NSL1) / 16);
      }
      if (jp == 0) triple2<2>(n0, n1, w0, w1, lnh[jp], lnl[jp]);
      else triple2<1>(n0, n1, w0, w1, lnh[jp], lnl[jp]);
    });
    mfma_fence_v(n0, n1);
  };
  // MLP2 of one chunk pair (+ the remaining GELU slices beside its first steps; slice i feeds step i >> 2,
  // which starts at pair-step 4 (i >> 2): slices NSL1 + jp at pair-step jp are always a step ahead for NSL1 >= 8)
  auto m2 = [&](auto have_prev, const f32x16& p0, const f32x16& p1) {
    segment<32>(ring, cur, [&](int jp, const Frag<PL>& w0, const Frag<PL>& w1) {
      if constexpr (decltype(have_prev)::value) {
        if (jp < 16 - NSL1) gelu_slice(p0, p1, NSL1 + jp);
      }
      triple2<3>(acc[(2 * jp) & 7], acc[(2 * jp + 1) & 7], w0, w1, hph[jp >> 2], hpl[jp >> 2]);
    });
  };
  constexpr std::true_type yes{};
  constexpr std::false_type no{};
  f32x16 ha0, ha1, hb0, hb1;
  m1(no, ha0, ha1, ha0, ha1);
  for (int it = 0; it < 3; ++it) {
    m1(yes, hb0, hb1, ha0, ha1);
    m2(yes, ha0, ha1);
    m1(yes, ha0, ha1, hb0, hb1);
    m2(yes, hb0, hb1);
  }
  m1(yes, hb0, hb1, ha0, ha1);
  m2(yes, ha0, ha1);
#pragma unroll
  for (int i = 0; i < 16; ++i) gelu_slice(hb0, hb1, i);
  __builtin_amdgcn_sched_barrier(0);
  m2(no, hb0, hb1);
  mfma_fence_a(acc);
  STAMP(4);
  ring.drain();

  // ---- out = acc 2^-11
  if (tok < p.T && (!(p.flags & 1) || blockIdx.x == 0)) {
    float* yrow = p.y + (size_t)tok * C + 4 * h;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[mt][4 * j + i] * S11_INV;
        *reinterpret_cast<f32x4*>(yrow + 32 * mt + 8 * j) = v;
      }
  }
  if (!(fm < 65520.0f) && p.tbuf) p.tbuf[0] = 0;   // keep the guard alive
  STAMP(5);
}

}  // namespace c128

using namespace c128;

template <int PL, int G, int R, int GPK, int NSL1, int DBG = 0>
static int launch(const Params& p, int nwg, hipStream_t st) {
  typedef Ring<PL, G, R> RingT;
  const int smem = RingT::BYTES + 2048;
  auto k = k_chain128<PL, G, R, GPK, NSL1, DBG>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k, dim3(nwg), dim3(256), smem, st, p);
  return (int)hipGetLastError();
}

extern "C" {
// variant -> (PL, G, R, GPK, NSL1); returns ring geometry for the packer
int chain128_variant_info(int variant, int* PL, int* G, int* R) {
  switch (variant) {
    case 0: *PL = 3; *G = 8; *R = 6; return 0;
    case 1: *PL = 3; *G = 4; *R = 12; return 0;
    case 2: *PL = 3; *G = 8; *R = 6; return 0;   // NSL1 = 12
    case 3: *PL = 3; *G = 8; *R = 6; return 0;   // packed GELU
    case 4: *PL = 3; *G = 8; *R = 6; return 0;   // NSL1 = 8
    case 5: *PL = 3; *G = 4; *R = 12; return 0;  // NSL1 = 8
    case 6: case 7: case 8: case 9: case 10: *PL = 3; *G = 8; *R = 6; return 0;
    default: return -1;
  }
}
int chain128_run(int variant, const float* msg, const float* x, float* y, const void* wstream, const float* ln_g,
                 const float* ln_b, int T, unsigned long long* tbuf, void* stream, int flags) {
  Params p{msg, x, y, reinterpret_cast<const char*>(wstream), ln_g, ln_b, T, flags, tbuf};
  const int nwg = (T + TOK_WG - 1) / TOK_WG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (variant) {
    case 0: return launch<3, 8, 6, 0, 16>(p, nwg, st);
    case 1: return launch<3, 4, 12, 0, 16>(p, nwg, st);
    case 2: return launch<3, 8, 6, 0, 12>(p, nwg, st);
    case 3: return launch<3, 8, 6, 1, 16>(p, nwg, st);
    case 4: return launch<3, 8, 6, 0, 8>(p, nwg, st);
    case 5: return launch<3, 4, 12, 0, 8>(p, nwg, st);
    case 6: return launch<3, 8, 6, 0, 16, 1>(p, nwg, st);   // timing only: no LDS reads, no ring traffic
    case 7: return launch<3, 8, 6, 0, 16, 2>(p, nwg, st);   // timing only: LDS reads, no ring traffic
    case 8: return launch<3, 8, 6, 0, 16, 4>(p, nwg, st);   // timing only: no GELU arithmetic
    case 9: return launch<3, 8, 6, 0, 16, 5>(p, nwg, st);   // timing only: no GELU, no LDS reads, no ring traffic
    case 10: return launch<3, 8, 6, 0, 16, 6>(p, nwg, st);  // timing only: no GELU, no ring traffic
    default: return -1;
  }
}
}
