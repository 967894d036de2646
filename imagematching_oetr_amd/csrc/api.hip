// C ABI of liboetr_hip.so (declared in include/oetr_hip.h): weight repacking,
// workspace layout and launch orchestration.  No torch types, no hidden
// allocation or synchronisation inside forward calls.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/oetr_hip.h"
#include "common.h"

using namespace oetr;

// oetr_set_state_prereduce(h, -1): source tokens per image from which the partial linear-attention
// states are summed by a launch of their own (see forward_impl)
constexpr int OETR_PREREDUCE_MIN_TOKENS = 768;
// oetr_set_tail_mode(h, 0): token rows (N (L1 + L2)) from which the forward path runs the decoder first and the
// heat-map conv in its direct 64-row form (k_heat_conv64) instead of decoder || P = W_tap.memory + combine
constexpr int OETR_DIRECT_TAIL_MIN_ROWS = 16000;   // measured crossover: 12 800 rows P form better by 15 us, 16 000 rows direct by 36 (profiles/r4_tail_forms.txt)

namespace {

thread_local std::string g_err;

oetr_status fail(oetr_status st, const std::string& msg) {
  g_err = msg;
  return st;
}
oetr_status hip_fail(hipError_t e, const char* what) {
  return fail(OETR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e__ = (expr);                            \
    if (e__ != hipSuccess) return hip_fail(e__, #expr); \
  } while (0)

// ---- host-side weight repacking -------------------------------------------
struct Packer {
  std::vector<float> buf;
  size_t reserve_aligned(size_t n) {
    size_t off = (buf.size() + 63) & ~size_t(63);  // 256-B aligned offsets
    buf.resize(off + n);
    return off;
  }
  // W [nout][k] (torch Linear layout) -> MFMA B-fragment order (common.h).
  size_t frag(const float* W, int nout, int k, int ld = -1, int stride_k = 1) {
    if (ld < 0) ld = k;
    const int ks_n = k / 8;
    size_t off = reserve_aligned((size_t)nout * k);
    float* dst = buf.data() + off;
    for (int nt = 0; nt < nout / 32; ++nt)
      for (int ks = 0; ks < ks_n; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const int n = nt * 32 + (lane & 31);
            const int kk = ks * 8 + 4 * (lane >> 5) + j;
            dst[(((size_t)nt * ks_n + ks) * 64 + lane) * 4 + j] =
                W[(size_t)n * ld + (size_t)kk * stride_k];
          }
    return off;
  }
  // float -> bf16 bit pattern, round to nearest even (host side)
  static uint16_t bf16_bits(float w) {
    uint32_t u;
    memcpy(&u, &w, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
  // W [nout][k] -> 16-bit planes in MFMA B-fragment order (common.h: gemm_rows32_h):
  // GM_SPLIT two f16 planes (hi, lo*2^11); GM_F16 / GM_BF16 one plane (RNE), lo_off = hi_off.
  // Returns offsets (in floats) of the planes.
  void frag_h(int mode, const float* W, int nout, int k, size_t* hi_off, size_t* lo_off,
              int ld = -1, int stride_k = 1) {
    if (ld < 0) ld = k;
    const int ks_n = k / 16;
    const size_t plane_floats = (size_t)nout * k / 2;
    *hi_off = reserve_aligned(plane_floats);
    *lo_off = mode == GM_SPLIT ? reserve_aligned(plane_floats) : *hi_off;
    _Float16* hi = reinterpret_cast<_Float16*>(buf.data() + *hi_off);
    _Float16* lo = reinterpret_cast<_Float16*>(buf.data() + *lo_off);
    uint16_t* hb = reinterpret_cast<uint16_t*>(hi);
    for (int nt = 0; nt < nout / 32; ++nt)
      for (int ks = 0; ks < ks_n; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int n = nt * 32 + (lane & 31);
            const int kk = ks * 16 + 8 * (lane >> 5) + j;
            const float w = W[(size_t)n * ld + (size_t)kk * stride_k];
            const size_t idx = (((size_t)nt * ks_n + ks) * 64 + lane) * 8 + j;
            if (mode == GM_BF16) { hb[idx] = bf16_bits(w); continue; }
            const _Float16 h = (_Float16)w;
            hi[idx] = h;
            if (mode == GM_SPLIT) lo[idx] = (_Float16)((w - (float)h) * SPLIT_SCALE);
          }
  }
  // one GEMM weight in the representation of the chosen mode
  void gemm_weight(int mode, const float* W, int nout, int k, size_t* a, size_t* b, int ld = -1,
                   int stride_k = 1) {
    if (gm_half(mode)) frag_h(mode, W, nout, k, a, b, ld, stride_k);
    else { *a = frag(W, nout, k, ld, stride_k); *b = *a; }
  }
  // W [nout][k] -> transposed [k][nout]
  size_t transposed(const float* W, int nout, int k) {
    size_t off = reserve_aligned((size_t)nout * k);
    float* dst = buf.data() + off;
    for (int n = 0; n < nout; ++n)
      for (int kk = 0; kk < k; ++kk) dst[(size_t)kk * nout + n] = W[(size_t)n * k + kk];
    return off;
  }
  size_t copy(const float* v, int n) {
    size_t off = reserve_aligned(n);
    memcpy(buf.data() + off, v, sizeof(float) * n);
    return off;
  }
};

}  // namespace

namespace oetr {
int set_last_error(int status, const char* msg) { return fail((oetr_status)status, msg); }
}  // namespace oetr

enum KernelId { K_ENC_A, K_ENC_BA, K_ENC_BDEC, K_ENC_B, K_DECODER, K_HEAT_CONV,
                K_HEAT_FINAL, K_SIZE_REG, K_BOXES, K_DEC_CONVP, K_HEAT_COMBINE, K_NECK_PROJ,
                K_NECK_CONV, K_NECK_OUT, K_KV_REDUCE, K_COUNT };
static const char* const kKernelNames[K_COUNT] = {
    "k_encoder<A>", "k_encoder<B,A>", "k_encoder<B,dec>", "k_encoder<B>",
    "k_decoder", "k_heat_conv", "k_heat_final", "k_size_reg", "k_boxes", "k_decoder_convp",
    "k_heat_combine", "k_neck_proj", "k_neck_conv", "k_neck_out", "k_kv_reduce"};

struct oetr_trace {
  std::vector<hipEvent_t> ev;  // 2 per launch
  std::vector<int> kid;
  int used = 0;                // launches recorded
  int dropped = 0;
};

struct oetr_ctx {
  oetr_trace* trace = nullptr;
  int device = 0;
  int mode = GM_SPLIT;  // GEMM mode GM_* (common.h) of the oetr_dtype
  int kv_prereduce = -1; // oetr_set_state_prereduce (-1 = auto)
  mutable int dec_fault = 0;   // oetr_debug_decoder_fault: one shot, consumed by the next four-workgroup decoder launch
  int dec_split = 0;     // oetr_set_decoder_split: 0 auto, 1 one workgroup per image, 4 four (decoder.hip: decoder_body4)
  int tail_mode = 0;     // oetr_set_tail_mode: 0 auto, 1 P form (decoder || conv-P, combine), 2 direct (decoder, conv)
  int policy = 0;       // precision policy (SitePolicy<>) of the oetr_dtype: 1 = OETR_DTYPE_F32_SPLIT_QK16
  int enc_tile = 0;    // 0 = auto, 32, 64 (oetr_set_encoder_tile)
  int attn_full = 0;   // OETR_ATTENTION_FULL (oetr_set_attention)
  int num_cus = 256;
  float* dev = nullptr;  // all repacked weights
  size_t dev_floats = 0;
  EncLayerDev enc[OETR_N_ENC];
  DecKVDev dkv;
  DecLayerDev dec[OETR_N_DEC];
  const float* qe[2];
  const float *dec_tgt1, *dec_q0, *dec_qkv1;  // create-time decoder constants
  HeadsDev heads;
};

struct oetr_neck_ctx {
  oetr_trace* trace = nullptr;
  int device = 0;
  int num_cus = 256;
  int conv_rows = 0;     // 0 = auto (neck_conv_rows), else forced 256 / 192 / 128 (A/B timing)
  float* dev = nullptr;  // all repacked weights
  const f32x4 *proj_wh[2], *proj_wl[2];
  const float *proj_b, *ln_w, *ln_b;
  const f32x4 *conv_wh[3], *conv_wl[3];
  const f32x4 *conv_wh_rw[3], *conv_wl_rw[3];   // k_neck_conv_rw's step order
  int conv_kernel = 0;   // 0 = auto (row-window when wo >= NECK_RW_MIN_WO), 1 = gather, 2 = row-window
  const float* conv_b[3];
  const f32x4 *out_wh, *out_wl;
  const float* out_b;
};

namespace {

#ifdef OETR_PHASE_TIMING
long long* g_tbuf = nullptr;
#endif

struct Workspace {
  float *x, *qp, *pos, *kvp[2], *ksp[2], *att0, *z0, *dkv1, *dks1, *conv_out, *gn_part, *sm_part, *hs,
      *logits, *cxy, *tlbr, *convp, *kbuf[2], *vt[2], *kvr, *ksr;
  uint32_t* flags;   // the workspace's status word (first 256 bytes: shape-independent position)
  size_t bytes;
};

bool make_geom(int n, int hf1, int wf1, int hf2, int wf2, Geom* g) {
  if (n <= 0 || hf1 <= 0 || wf1 <= 0 || hf2 <= 0 || wf2 <= 0) return false;
  const long L1 = (long)hf1 * wf1, L2 = (long)hf2 * wf2;
  if (L1 > OETR_MAX_TOKENS || L2 > OETR_MAX_TOKENS) return false;
  if ((long)n * (L1 + L2) > (1L << 30) / C) return false;  // keep 32-bit row indices safe
  g->N = n;
  g->L[0] = (int)L1; g->L[1] = (int)L2;
  g->hf[0] = hf1; g->wf[0] = wf1; g->hf[1] = hf2; g->wf[1] = wf2;
  g->nt[0] = (g->L[0] + TM - 1) / TM; g->nt[1] = (g->L[1] + TM - 1) / TM;
  g->row0[0] = 0; g->row0[1] = n * g->L[0];
  g->prow0[0] = 0; g->prow0[1] = g->L[0];
  g->tile0[0] = 0; g->tile0[1] = n * g->nt[0];
  g->ntiles = n * (g->nt[0] + g->nt[1]);
  g->rows = n * (g->L[0] + g->L[1]);
  return true;
}

// Tile bookkeeping of the encoder / decoder-partial side.  The split mode has a
// 64-token workgroup shape (k_encoder64: every weight fragment feeds two MFMA row
// tiles; 1.2-1.3x fewer CU-microseconds per token, but half as many workgroups).
// auto: 64 once the 32-token grid no longer fits the chip in one wave of
// workgroups; oetr_set_encoder_tile overrides.  The heads keep TM.
int encoder_tile_rows(const oetr_ctx* h, const Geom& g) {
  if (!gm_half(h->mode) || h->attn_full) return TM;
  const int want = h->enc_tile;
  if (want == TM || want == 64) return want;
  return g.ntiles > h->num_cus ? 64 : TM;
}
Geom encoder_geom(const Geom& g, int rows) {
  Geom e = g;
  for (int i = 0; i < 2; ++i) e.nt[i] = (g.L[i] + rows - 1) / rows;
  e.tile0[0] = 0; e.tile0[1] = g.N * e.nt[0];
  e.ntiles = g.N * (e.nt[0] + e.nt[1]);
  return e;
}

Workspace carve(const Geom& g, void* base, bool attn_full = false) {
  Workspace w;
  size_t off = 0;
  auto take = [&](size_t floats) {
    float* p = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr;
    off += (floats * sizeof(float) + 255) & ~size_t(255);
    return p;
  };
  const size_t rows = g.rows, nt = g.ntiles;
  w.flags = reinterpret_cast<uint32_t*>(take(OETR_WORKSPACE_STATUS_BYTES / sizeof(float)));
  w.x = take(rows * C);
  // phi(Q): token-major for the 32-row kernel; the 64-row kernel keeps it TILE-major
  // ([tile][64][256], every image's last tile padded - its stores need no row predicate)
  w.qp = take((rows + (size_t)2 * g.N * 64) * C);
  w.pos = take((size_t)(g.L[0] + g.L[1]) * C);
  for (int i = 0; i < 2; ++i) { w.kvp[i] = take(nt * KV_FLOATS); w.ksp[i] = take(nt * C); }
  w.att0 = take(nt * C); w.z0 = take(nt * NH);
  w.dkv1 = take(nt * KV_FLOATS); w.dks1 = take(nt * C);
  w.conv_out = take(rows * C);
  w.gn_part = take(nt * 32 * 2);
  w.sm_part = take(nt * 4);
  w.hs = take((size_t)2 * g.N * C);
  w.logits = take(rows);
  w.cxy = take((size_t)2 * g.N * 2);
  w.tlbr = take((size_t)2 * g.N * 4);
  w.convp = take((size_t)9 * rows * C);  // P_tap = W_tap . memory (forward path)
  w.kvr = take((size_t)2 * g.N * KV_FLOATS);   // one reduced linear-attention state per image (oetr_set_state_prereduce)
  w.ksr = take((size_t)2 * g.N * C);
  for (int i = 0; i < 2; ++i) {           // attention == full: K rows and V^T, per layer parity
    w.kbuf[i] = attn_full ? take(rows * C) : nullptr;
    w.vt[i] = attn_full ? take((size_t)g.N * C * TM * (g.nt[0] + g.nt[1])) : nullptr;
  }
  w.bytes = off;
  return w;
}

oetr_status check_ws(const Geom& g, void* ws, size_t ws_bytes, Workspace* out, bool attn_full = false) {
  if (!ws) return fail(OETR_ERR_WORKSPACE, "workspace is NULL");
  if (reinterpret_cast<uintptr_t>(ws) & 255)
    return fail(OETR_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  *out = carve(g, ws, attn_full);
  if (out->bytes > ws_bytes)
    return fail(OETR_ERR_WORKSPACE, "workspace too small: need " + std::to_string(out->bytes) +
                                        " bytes, got " + std::to_string(ws_bytes));
  return OETR_OK;
}

HeatLaunch heat_launch(const oetr_ctx* h, const Geom& g, const Workspace& w, const float* mem1,
                       const float* mem2, const float* hs1, const float* hs2, float* cxy1,
                       float* cxy2, int img_h1, int img_h2) {
  HeatLaunch p;
  p.g = g;
  p.w = h->heads;
  p.mem[0] = mem1; p.mem[1] = mem2;
  p.hs[0] = hs1; p.hs[1] = hs2;
  p.conv_out = w.conv_out;
  p.gn_part = w.gn_part;
  p.sm_part = w.sm_part;
  p.logits = w.logits;
  p.cxy[0] = cxy1; p.cxy[1] = cxy2;
  p.img_h[0] = img_h1; p.img_h[1] = img_h2;
  p.tlbr[0] = p.tlbr[1] = nullptr;   // stand-alone centre estimation: no fused tail
  p.box[0] = p.box[1] = nullptr;
  p.img_w[0] = p.img_w[1] = 0;
  p.flags = w.flags;
  p.publish = nullptr;
  p.force_staged_conv = h->tail_mode == 3;
  p.convp_units = 9;
  p.mask[0] = p.mask[1] = nullptr;
  return p;
}

// forward_dummy's masks: both or neither, and only in the arithmetic they are built for
oetr_status check_masks(const oetr_ctx* h, const float* mask1, const float* mask2, bool encoder) {
  if (!mask1 && !mask2) return OETR_OK;
  if (!mask1 || !mask2) return fail(OETR_ERR_BAD_ARG, "masks: pass both mask1 and mask2, or neither");
  if (encoder && ((h->mode != GM_SPLIT && h->mode != GM_F32) || h->attn_full))
    return fail(OETR_ERR_UNSUPPORTED, "masks: built for OETR_DTYPE_F32_SPLIT_F16, OETR_DTYPE_F32_SPLIT_QK16 and OETR_DTYPE_F32 with linear attention "
                                      "(the reference's FullAttention turns a masked query row into NaN, "
                                      "linear_attention.py:74-81)");
  return OETR_OK;
}

// Brackets one kernel launch with two events when a trace is attached.
struct Scoped {
  oetr_trace* t; hipStream_t s; int slot;
  Scoped(oetr_trace* tr, hipStream_t s_, int kid) : t(tr), s(s_), slot(-1) {
    if (!t) return;
    if (2 * (t->used + 1) > (int)t->ev.size()) { t->dropped++; t = nullptr; return; }
    slot = t->used++;
    t->kid[slot] = kid;
    (void)hipEventRecord(t->ev[2 * slot], s);
  }
  ~Scoped() { if (t) (void)hipEventRecord(t->ev[2 * slot + 1], s); }
};
#define TRACED(h, s, kid, expr)            \
  do {                                     \
    Scoped sc__((h)->trace, s, kid);       \
    HIP_TRY(expr);                         \
  } while (0)

// Encoder (+ decoder) shared by forward and feature_correlation.
DecLaunch dec_launch(const oetr_ctx* h, const Geom& g, const Workspace& w, bool beside_convp = false) {
  DecLaunch d;
  d.g = encoder_geom(g, encoder_tile_rows(h, g));  // partials are per ENCODER tile
  for (int i = 0; i < 2; ++i) { d.layer[i] = h->dec[i]; d.qe[i] = h->qe[i]; }
  d.tgt1 = h->dec_tgt1; d.qkv1 = h->dec_qkv1;
  d.att0_part = w.att0; d.z0_part = w.z0; d.dkv1 = w.dkv1; d.dks1 = w.dks1;
  d.hs = w.hs;
  // Four workgroups per image (decoder.hip: decoder_body4) where that shortens the STEP, measured on
  // MI355X (profiles/r4_decoder_split.txt):
  //  * the decoder as a launch of its own (direct tail, feature_correlation): the chain is exposed,
  //    47.6 -> 24.8 us = -3.5 % of a serial step at 8 pairs 640 vs 1280 - taken while the 2N x 4
  //    workgroups are at most a quarter of the CUs: the exchanging workgroups of up to four forwards
  //    in flight on one device (one per stream) are then resident together whatever the dispatch
  //    order - they wait for each other (exchange_sum);
  //  * beside the conv-P GEMMs (P form): taken while the decoder's workgroups and the conv items
  //    (three per 64-token tile then, conv_p.h) fit the chip in ONE round: 1 pair 45.5 -> 25.7 us,
  //    4 pairs 47 -> 27.5, -6...8 % of a serial step.  At 8 pairs @640x640 (400 workgroups) the launch
  //    does get shorter, 52.5 -> 42.6 us, but the eight encoder launches around it each get 1 us
  //    LONGER - the busier launch costs the chip its clock - and the step stays where it was
  //    (358 vs 361 us) while three overlapped streams lose 2.5 %: one workgroup per image there.
  // Round 5: the conv work is cut into items of `convp_units` (tile, tap) units - nine units per 64-token
  // tile - sized so that the decoder's workgroups + the items fill the chip in ONE round: at 8 pairs
  // @640x640 1 008 units on the 192 CUs the 64 decoder workgroups leave = 6 units per item (168 items),
  // where three per tile (336 items) took two rounds; never fewer than 3 units per item (the staging of
  // the tile's rows per item: nine items per tile measured worse than three at every size, round 4).
  const int images = 2 * g.N;
  bool fits = images <= DEC_SPLIT_MAX_IMAGES && images * DEC_SPLIT_K * 4 <= h->num_cus;
  d.convp_units = 9;
  if (fits && beside_convp) {
    if (gm_half(h->mode)) {
      const int units = 9 * g.N * ((g.L[0] + RT - 1) / RT + (g.L[1] + RT - 1) / RT);
      const int avail = h->num_cus - images * DEC_SPLIT_K;
      const int upi = std::max(3, (units + avail - 1) / avail);
      fits = upi <= 9;
      if (fits) d.convp_units = upi;
    } else {
      fits = images * DEC_SPLIT_K + g.ntiles <= h->num_cus;
    }
  }
  if (h->dec_split == DEC_SPLIT_K && d.convp_units == 9 && beside_convp && gm_half(h->mode)) d.convp_units = 3;   // forced split: round 4's items
  d.ksplit = (h->dec_split == DEC_SPLIT_K || (h->dec_split == 0 && fits)) && images <= DEC_SPLIT_MAX_IMAGES
                 ? DEC_SPLIT_K : 1;
  d.flags = w.flags;
  d.xch_epoch = w.flags + STATUS_EPOCH_WORD;
  d.xch = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(w.flags) + STATUS_XCH_OFFSET);
  d.tbuf = nullptr;
  d.dbg = 0;
  if (d.ksplit == DEC_SPLIT_K && h->dec_fault) { d.dbg = DEC_DBG_FAULT; h->dec_fault = 0; }
#ifdef OETR_PHASE_TIMING
  d.tbuf = g_tbuf ? g_tbuf + 16 * 4096 : nullptr;
#endif
  return d;
}

oetr_status run_correlation(oetr_ctx* h, const Geom& g, const Workspace& w, const float* feat1,
                            const float* feat2, const float* pos1, const float* pos2,
                            int enc_layers, hipStream_t s, bool with_decoder = true,
                            bool resident = false, const float* mask1 = nullptr,
                            const float* mask2 = nullptr) {
  // resident: the caller (oetr_forward_tokens) already holds token-major features and
  // position tables in the workspace (oetr_token_buffers) - no transpose launch
  EncLaunch p;
  memset(&p, 0, sizeof(p));
  if (!resident) {   // the first launch transposes its tiles itself (round 1 had a k_prep_tokens launch for it)
    p.feat_nchw[0] = feat1; p.feat_nchw[1] = feat2;
    p.pos_nchw[0] = pos1; p.pos_nchw[1] = pos2;
    p.pos_out = w.pos;
  }
  p.tile_rows = encoder_tile_rows(h, g);
  p.g = encoder_geom(g, p.tile_rows);
#ifdef OETR_ABLATE
  { const char* e = getenv("OETR_ABLATE"); p.dbg = e ? atoi(e) : 0; }
#endif
#ifdef OETR_PHASE_TIMING
  {
    static long long* tb = nullptr;
    if (!tb) (void)hipMalloc(&tb, sizeof(long long) * 16 * 65536);
    p.tbuf = tb;
    g_tbuf = tb;
  }
#endif
  p.x = w.x; p.qp = w.qp; p.pos = w.pos;
  p.mask[0] = mask1; p.mask[1] = mask2;
  p.flags = w.flags;
  p.attn_full = h->attn_full;
  p.policy = h->policy;
  for (int i = 0; i < 2; ++i) p.lpad[i] = g.nt[i] * TM;
  p.vt_off[0] = 0; p.vt_off[1] = (size_t)g.N * C * p.lpad[0];
  p.kbuf_out = w.kbuf[0]; p.vt_out = w.vt[0];
  p.a = h->enc[0];
  p.kv_out = w.kvp[0]; p.ks_out = w.ksp[0];
  // oetr_set_state_prereduce: each image's per-tile partial states are summed ONCE instead of in
  // every consuming workgroup, in a launch of its own between the encoder launches (the consumer
  // launch gets shorter, the extra launch is a 4-5 us latency chain).  -1 (default) = auto: it pays once
  // an image has many partials - measured on MI355X, 64-row tiles (profiles/r4_prereduce_auto.txt):
  // 400 tokens per image (7 partials) +0.9 % step time with it, 1024 tokens (16) -3.9 %
  // (k_encoder64<B,A> 246 -> 225 us), 1600 tokens (25) -4.2 % (120 -> 105 us) - so it is on from
  // OETR_PREREDUCE_MIN_TOKENS source tokens per image.  (Rounds 3-4 also had an in-launch form - the
  // last workgroup of an image to finish reduced its partials behind an agent-scope release / ticket /
  // acquire: bit-identical, and slower at every size; removed in round 5.)
  int prereduce = h->attn_full ? 0 : h->kv_prereduce;
  if (prereduce < 0) prereduce = (g.L[0] >= OETR_PREREDUCE_MIN_TOKENS || g.L[1] >= OETR_PREREDUCE_MIN_TOKENS) ? 1 : 0;
  TRACED(h, s, K_ENC_A, launch_encoder(p, false, 0, h->mode, s));
  p.feat_nchw[0] = p.feat_nchw[1] = p.pos_nchw[0] = p.pos_nchw[1] = nullptr;
  for (int l = 0; l < enc_layers; ++l) {
    p.b = h->enc[l];
    p.b_cross = l & 1;
    p.kv_in = w.kvp[l & 1]; p.ks_in = w.ksp[l & 1];
    p.kv_reduced = 0;
    if (prereduce == 1) {   // sum layer l's per-tile partial states once per image, in a launch of its own
      TRACED(h, s, K_KV_REDUCE, launch_kv_reduce(p.g, w.kvp[l & 1], w.ksp[l & 1], w.kvr, w.ksr, s));
      p.kv_in = w.kvr; p.ks_in = w.ksr;
      p.kv_reduced = 1;
    }
    p.kbuf_in = w.kbuf[l & 1]; p.vt_in = w.vt[l & 1];
    int tail;
    if (l + 1 == OETR_N_ENC) {
      tail = 1;
      p.d = h->dkv;
      p.att0_out = w.att0; p.z0_out = w.z0; p.dkv1_out = w.dkv1; p.dks1_out = w.dks1;
      p.dec_q0 = h->dec_q0;
    } else if (l + 1 < enc_layers) {
      tail = 0;
      p.a = h->enc[l + 1];
      p.kv_out = w.kvp[(l + 1) & 1]; p.ks_out = w.ksp[(l + 1) & 1];
      p.kbuf_out = w.kbuf[(l + 1) & 1]; p.vt_out = w.vt[(l + 1) & 1];
    } else {
      tail = 2;
    }
    TRACED(h, s, tail == 0 ? K_ENC_BA : tail == 1 ? K_ENC_BDEC : K_ENC_B,
           launch_encoder(p, true, tail, h->mode, s));
  }
  if (enc_layers == OETR_N_ENC && with_decoder) {
    DecLaunch d = dec_launch(h, g, w);
    TRACED(h, s, K_DECODER, launch_decoder(d, s));
  }
  return OETR_OK;
}

// Weights must be finite, and representable by the mode's operand type: the f16-based
// modes (split, f16) need |w| < 65504 in every GEMM weight (common.h: Range guards the
// activations at run time; weights are checked once, here).
oetr_status check_weight(const char* name, const float* w, size_t n, bool f16_range) {
  for (size_t i = 0; i < n; ++i) {
    const float a = fabsf(w[i]);
    if (!(a <= 3.4028234e38f))
      return fail(OETR_ERR_UNSUPPORTED, std::string(name) + ": non-finite weight at index " + std::to_string(i));
    if (f16_range && a >= 65504.0f)
      return fail(OETR_ERR_UNSUPPORTED, std::string(name) + ": |w| = " + std::to_string(a) +
                  " at index " + std::to_string(i) + " exceeds the f16 range of this dtype; use "
                  "OETR_DTYPE_BF16 or OETR_DTYPE_F32");
  }
  return OETR_OK;
}
#define CHECK_W(name, ptr, n, f16r)                                      \
  do {                                                                   \
    oetr_status rc__ = check_weight(name, ptr, n, f16r);                 \
    if (rc__) return rc__;                                               \
  } while (0)

oetr_status copy_out(float* dst, const float* src, size_t floats, hipStream_t s) {
  if (!dst) return OETR_OK;
  HIP_TRY(hipMemcpyAsync(dst, src, floats * sizeof(float), hipMemcpyDeviceToDevice, s));
  return OETR_OK;
}

}  // namespace

extern "C" {

const char* oetr_last_error(void) { return g_err.c_str(); }
int oetr_abi_version(void) { return OETR_ABI_VERSION; }

oetr_status oetr_create(const oetr_weights* w, oetr_dtype dtype, int device, oetr_handle* out) {
  if (!w || !out) return fail(OETR_ERR_BAD_ARG, "oetr_create: NULL argument");
  if (w->struct_size != sizeof(oetr_weights) || w->abi_version != OETR_ABI_VERSION)
    return fail(OETR_ERR_BAD_ARG, "oetr_create: oetr_weights size/ABI mismatch");
  static_assert(OETR_WORKSPACE_STATUS_BYTES == STATUS_BYTES && OETR_FLAG_EXCHANGE == FLAG_EXCHANGE, "status block");
  static_assert(OETR_DTYPE_F32 == GM_F32 && OETR_DTYPE_F32_SPLIT_F16 == GM_SPLIT &&
                    OETR_DTYPE_F16 == GM_F16 && OETR_DTYPE_BF16 == GM_BF16, "oetr_dtype == GM_*");
  if (dtype != OETR_DTYPE_F32 && dtype != OETR_DTYPE_F32_SPLIT_F16 && dtype != OETR_DTYPE_F16 &&
      dtype != OETR_DTYPE_BF16 && dtype != OETR_DTYPE_F32_SPLIT_QK16)
    return fail(OETR_ERR_UNSUPPORTED, "dtype must be one of OETR_DTYPE_{F32,F32_SPLIT_F16,F16,BF16,F32_SPLIT_QK16}");
  // GEMM mode (name kept: selects the weight representation); QK16 = the split mode's planes
  // with a per-site precision policy (SitePolicy<1>)
  const int split = dtype == OETR_DTYPE_F32_SPLIT_QK16 ? (int)GM_SPLIT : (int)dtype;
  {  // every pointer must be set
    const float* const* p = reinterpret_cast<const float* const*>(&w->enc[0]);
    const size_t n = (sizeof(oetr_weights) - offsetof(oetr_weights, enc)) / sizeof(float*);
    for (size_t i = 0; i < n; ++i)
      if (!p[i]) return fail(OETR_ERR_BAD_ARG, "oetr_create: weight pointer #" +
                                                   std::to_string(i) + " is NULL");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
    return fail(OETR_ERR_NO_DEVICE, "no HIP device " + std::to_string(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (!strstr(prop.gcnArchName, "gfx950"))
    return fail(OETR_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName +
                                        ", this library is built for gfx950 only");

  {
    const bool f16r = split == GM_SPLIT || split == GM_F16;
    for (int l = 0; l < OETR_N_ENC; ++l) {
      const oetr_encoder_layer_weights& e = w->enc[l];
      const std::string pre = "encoder." + std::to_string(l) + ".";
      CHECK_W((pre + "q_proj").c_str(), e.q_proj, (size_t)C * C, f16r);
      CHECK_W((pre + "k_proj").c_str(), e.k_proj, (size_t)C * C, f16r);
      CHECK_W((pre + "v_proj").c_str(), e.v_proj, (size_t)C * C, f16r);
      CHECK_W((pre + "merge").c_str(), e.merge, (size_t)C * C, f16r);
      CHECK_W((pre + "mlp.0").c_str(), e.mlp0, (size_t)FF * C, f16r);
      CHECK_W((pre + "mlp.2").c_str(), e.mlp2, (size_t)FF * C, f16r);
      const float* vecs[6] = {e.pre_norm_q_w, e.pre_norm_q_b, e.pre_norm_kv_w, e.pre_norm_kv_b,
                              e.norm2_w, e.norm2_b};
      for (int i = 0; i < 6; ++i) CHECK_W((pre + "norm").c_str(), vecs[i], C, false);
    }
    for (int l = 0; l < OETR_N_DEC; ++l) {
      const oetr_decoder_layer_weights& d = w->dec[l];
      const std::string pre = "decoder.layers." + std::to_string(l) + ".";
      const oetr_mha_weights* mh[2] = {&d.self_attn, &d.multihead_attn};
      for (int a = 0; a < 2; ++a) {
        const bool gemm = a == 1;  // the cross-attention k/v projections run as MFMA GEMMs
        CHECK_W((pre + "attn.q_proj").c_str(), mh[a]->q_proj_w, (size_t)C * C, false);
        CHECK_W((pre + "attn.k_proj").c_str(), mh[a]->k_proj_w, (size_t)C * C, gemm && f16r);
        CHECK_W((pre + "attn.v_proj").c_str(), mh[a]->v_proj_w, (size_t)C * C, gemm && f16r);
        CHECK_W((pre + "attn.merge").c_str(), mh[a]->merge, (size_t)C * C, false);
        CHECK_W((pre + "attn.q_bias").c_str(), mh[a]->q_proj_b, C, false);
        CHECK_W((pre + "attn.k_bias").c_str(), mh[a]->k_proj_b, C, false);
        CHECK_W((pre + "attn.v_bias").c_str(), mh[a]->v_proj_b, C, false);
      }
      CHECK_W((pre + "mlp.0").c_str(), d.mlp0, (size_t)FF * C, false);
      CHECK_W((pre + "mlp.2").c_str(), d.mlp2, (size_t)FF * C, false);
      const float* nv[6] = {d.norm1_w, d.norm1_b, d.norm2_w, d.norm2_b, d.norm3_w, d.norm3_b};
      for (int i = 0; i < 6; ++i) CHECK_W((pre + "norm").c_str(), nv[i], C, false);
    }
    CHECK_W("query_embed1", w->query_embed1, C, false);
    CHECK_W("query_embed2", w->query_embed2, C, false);
    CHECK_W("tlbr_reg.0", w->tlbr0_w, (size_t)C * C, false);
    CHECK_W("tlbr_reg.2.weight", w->tlbr2_w, 4 * C, false);
    CHECK_W("tlbr_reg.2.bias", w->tlbr2_b, 4, false);
    CHECK_W("heatmap_conv.0.weight", w->heat_conv_w, (size_t)C * C * 9, f16r);
    CHECK_W("heatmap_conv.0.bias", w->heat_conv_b, C, false);
    CHECK_W("heatmap_conv.1.weight", w->heat_gn_w, C, false);
    CHECK_W("heatmap_conv.1.bias", w->heat_gn_b, C, false);
    CHECK_W("heatmap_conv.3.weight", w->heat_out_w, C, false);
    CHECK_W("heatmap_conv.3.bias", w->heat_out_b, 1, false);
  }

  oetr_ctx* h = new oetr_ctx();
  h->device = device;
  h->mode = split;
  h->policy = dtype == OETR_DTYPE_F32_SPLIT_QK16 ? 1 : 0;
  h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  Packer pk;
  struct EncOff { size_t wq, wk, wv, wm, w1, w2, wq_l, wk_l, wv_l, wm_l, w1_l, w2_l, v[6]; } eo[OETR_N_ENC];
  for (int l = 0; l < OETR_N_ENC; ++l) {
    const oetr_encoder_layer_weights& e = w->enc[l];
    pk.gemm_weight(split, e.q_proj, C, C, &eo[l].wq, &eo[l].wq_l);
    pk.gemm_weight(split, e.k_proj, C, C, &eo[l].wk, &eo[l].wk_l);
    pk.gemm_weight(split, e.v_proj, C, C, &eo[l].wv, &eo[l].wv_l);
    pk.gemm_weight(split, e.merge, C, C, &eo[l].wm, &eo[l].wm_l);
    pk.gemm_weight(split, e.mlp0, FF, C, &eo[l].w1, &eo[l].w1_l);
    pk.gemm_weight(split, e.mlp2, C, FF, &eo[l].w2, &eo[l].w2_l);
    const float* vecs[6] = {e.pre_norm_q_w, e.pre_norm_q_b, e.pre_norm_kv_w,
                            e.pre_norm_kv_b, e.norm2_w, e.norm2_b};
    for (int i = 0; i < 6; ++i) eo[l].v[i] = pk.copy(vecs[i], C);
  }
  struct DecOff { size_t ck, cv, ck_l, cv_l, cbk, cbv, m[2][7], w1, w2, n[6]; } dof[OETR_N_DEC];
  for (int l = 0; l < OETR_N_DEC; ++l) {
    const oetr_decoder_layer_weights& d = w->dec[l];
    pk.gemm_weight(split, d.multihead_attn.k_proj_w, C, C, &dof[l].ck, &dof[l].ck_l);
    pk.gemm_weight(split, d.multihead_attn.v_proj_w, C, C, &dof[l].cv, &dof[l].cv_l);
    dof[l].cbk = pk.copy(d.multihead_attn.k_proj_b, C);
    dof[l].cbv = pk.copy(d.multihead_attn.v_proj_b, C);
    const oetr_mha_weights* mh[2] = {&d.self_attn, &d.multihead_attn};
    for (int a = 0; a < 2; ++a) {
      dof[l].m[a][0] = pk.transposed(mh[a]->q_proj_w, C, C);
      dof[l].m[a][1] = pk.transposed(mh[a]->k_proj_w, C, C);
      dof[l].m[a][2] = pk.transposed(mh[a]->v_proj_w, C, C);
      dof[l].m[a][3] = pk.transposed(mh[a]->merge, C, C);
      dof[l].m[a][4] = pk.copy(mh[a]->q_proj_b, C);
      dof[l].m[a][5] = pk.copy(mh[a]->k_proj_b, C);
      dof[l].m[a][6] = pk.copy(mh[a]->v_proj_b, C);
    }
    dof[l].w1 = pk.transposed(d.mlp0, FF, C);
    dof[l].w2 = pk.transposed(d.mlp2, C, FF);
    const float* nv[6] = {d.norm1_w, d.norm1_b, d.norm2_w, d.norm2_b, d.norm3_w, d.norm3_b};
    for (int i = 0; i < 6; ++i) dof[l].n[i] = pk.copy(nv[i], C);
  }
  const size_t qe1 = pk.copy(w->query_embed1, C), qe2 = pk.copy(w->query_embed2, C);
  // conv weight [o][i][ky][kx] -> 9 fragment-packed [o][i] matrices, tap = ky*3+kx
  // (taps are consecutive: f32 mode one array of 9 matrices; split mode all 9 hi
  //  planes then all 9 lo planes)
  size_t conv_off = 0, conv_off_l = 0;
  if (!gm_half(split)) {
    for (int tap = 0; tap < 9; ++tap) {
      const size_t o = pk.frag(w->heat_conv_w + tap, C, C, /*ld=*/C * 9, /*stride_k=*/9);
      if (tap == 0) conv_off = o;
    }
    conv_off_l = conv_off;
  } else {
    std::vector<size_t> hi(9), lo(9);
    Packer tmp;  // pack per tap, then lay the planes out contiguously per kind
    for (int tap = 0; tap < 9; ++tap) tmp.frag_h(split, w->heat_conv_w + tap, C, C, &hi[tap], &lo[tap], C * 9, 9);
    const size_t plane = (size_t)C * C / 2;
    conv_off = pk.reserve_aligned(9 * plane);
    conv_off_l = split == GM_SPLIT ? pk.reserve_aligned(9 * plane) : conv_off;
    for (int tap = 0; tap < 9; ++tap) {
      memcpy(pk.buf.data() + conv_off + tap * plane, tmp.buf.data() + hi[tap], plane * sizeof(float));
      if (split == GM_SPLIT)
        memcpy(pk.buf.data() + conv_off_l + tap * plane, tmp.buf.data() + lo[tap], plane * sizeof(float));
    }
  }
  const size_t conv_b = pk.copy(w->heat_conv_b, C), gn_w = pk.copy(w->heat_gn_w, C),
               gn_b = pk.copy(w->heat_gn_b, C), out_w = pk.copy(w->heat_out_w, C),
               out_b = pk.copy(w->heat_out_b, 1);
  const size_t dconst = pk.reserve_aligned(2 * C + 2 * C + 2 * 3 * C);  // tgt1 | q0 | qkv1
  const size_t t0 = pk.transposed(w->tlbr0_w, C, C), t2w = pk.copy(w->tlbr2_w, 4 * C),
               t2b = pk.copy(w->tlbr2_b, 4);

  int prev = 0;
  (void)hipGetDevice(&prev);
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc(&h->dev, pk.buf.size() * sizeof(float));
  if (e == hipSuccess)
    e = hipMemcpy(h->dev, pk.buf.data(), pk.buf.size() * sizeof(float), hipMemcpyHostToDevice);
  (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    if (h->dev) (void)hipFree(h->dev);
    delete h;
    return hip_fail(e, "oetr_create: uploading weights");
  }
  h->dev_floats = pk.buf.size();
  const float* B = h->dev;
  auto F4 = [&](size_t off) { return reinterpret_cast<const f32x4*>(B + off); };
  for (int l = 0; l < OETR_N_ENC; ++l) {
    EncLayerDev& d = h->enc[l];
    d.wq = F4(eo[l].wq); d.wk = F4(eo[l].wk); d.wv = F4(eo[l].wv); d.wmerge = F4(eo[l].wm);
    d.w1 = F4(eo[l].w1); d.w2 = F4(eo[l].w2);
    d.wq_l = F4(eo[l].wq_l); d.wk_l = F4(eo[l].wk_l); d.wv_l = F4(eo[l].wv_l);
    d.wmerge_l = F4(eo[l].wm_l); d.w1_l = F4(eo[l].w1_l); d.w2_l = F4(eo[l].w2_l);
    d.lnq_w = B + eo[l].v[0]; d.lnq_b = B + eo[l].v[1];
    d.lnkv_w = B + eo[l].v[2]; d.lnkv_b = B + eo[l].v[3];
    d.ln2_w = B + eo[l].v[4]; d.ln2_b = B + eo[l].v[5];
  }
  for (int l = 0; l < OETR_N_DEC; ++l) {
    h->dkv.wk[l] = F4(dof[l].ck); h->dkv.wv[l] = F4(dof[l].cv);
    h->dkv.wk_l[l] = F4(dof[l].ck_l); h->dkv.wv_l[l] = F4(dof[l].cv_l);
    h->dkv.bk[l] = B + dof[l].cbk; h->dkv.bv[l] = B + dof[l].cbv;
    DecLayerDev& d = h->dec[l];
    MhaDev* mh[2] = {&d.self_attn, &d.cross};
    for (int a = 0; a < 2; ++a) {
      mh[a]->wq_t = B + dof[l].m[a][0]; mh[a]->wk_t = B + dof[l].m[a][1];
      mh[a]->wv_t = B + dof[l].m[a][2]; mh[a]->wm_t = B + dof[l].m[a][3];
      mh[a]->bq = B + dof[l].m[a][4]; mh[a]->bk = B + dof[l].m[a][5];
      mh[a]->bv = B + dof[l].m[a][6];
    }
    d.w1_t = B + dof[l].w1; d.w2_t = B + dof[l].w2;
    d.n1w = B + dof[l].n[0]; d.n1b = B + dof[l].n[1]; d.n2w = B + dof[l].n[2];
    d.n2b = B + dof[l].n[3]; d.n3w = B + dof[l].n[4]; d.n3b = B + dof[l].n[5];
  }
  h->qe[0] = B + qe1; h->qe[1] = B + qe2;
  h->dec_tgt1 = B + dconst; h->dec_q0 = B + dconst + 2 * C; h->dec_qkv1 = B + dconst + 4 * C;
  {  // fold the image-independent part of the decoder once (decoder.hip)
    DecConstLaunch dc;
    for (int i = 0; i < 2; ++i) { dc.layer[i] = h->dec[i]; dc.qe[i] = h->qe[i]; }
    dc.tgt1 = h->dev + dconst; dc.q0 = h->dev + dconst + 2 * C; dc.qkv1 = h->dev + dconst + 4 * C;
    (void)hipSetDevice(device);
    hipError_t ce = launch_decoder_consts(dc, nullptr);
    if (ce == hipSuccess) ce = hipStreamSynchronize(nullptr);
    (void)hipSetDevice(prev);
    if (ce != hipSuccess) {
      (void)hipFree(h->dev);
      delete h;
      return hip_fail(ce, "oetr_create: decoder constant folding");
    }
  }
  h->heads.conv_w = F4(conv_off);
  h->heads.conv_w_l = F4(conv_off_l);
  h->heads.conv_b = B + conv_b; h->heads.gn_w = B + gn_w; h->heads.gn_b = B + gn_b;
  h->heads.out_w = B + out_w; h->heads.out_b = B + out_b;
  h->heads.tlbr0_t = B + t0; h->heads.tlbr2_w = B + t2w; h->heads.tlbr2_b = B + t2b;
  *out = h;
  return OETR_OK;
}

void oetr_destroy(oetr_handle h) {
  if (!h) return;
  if (h->dev) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(h->device);
    (void)hipFree(h->dev);
    (void)hipSetDevice(prev);
  }
  delete h;
}

size_t oetr_workspace_bytes(oetr_handle h, int n_pairs, int hf1, int wf1, int hf2, int wf2) {
  Geom g;
  if (!make_geom(n_pairs, hf1, wf1, hf2, wf2, &g)) {
    g_err = "invalid shape: need N>0 and 1 <= hf*wf <= " + std::to_string(OETR_MAX_TOKENS);
    return 0;
  }
  return carve(g, nullptr, h && h->attn_full).bytes;   // (h may be NULL: shape-only query, linear attention)
}

namespace {
oetr_status forward_impl(oetr_handle h, const float* feat1, const float* feat2,
                         const float* pos1, const float* pos2, int n_pairs, int hf1,
                         int wf1, int hf2, int wf2, int img_h1, int img_w1, int img_h2,
                         int img_w2, void* workspace, size_t workspace_bytes,
                         float* box1, float* box2, const oetr_stage_outputs* st,
                         void* stream, bool resident, const float* mask1 = nullptr,
                         const float* mask2 = nullptr, uint32_t* publish = nullptr) {
  if (!h || (!resident && (!feat1 || !feat2 || !pos1 || !pos2)))
    return fail(OETR_ERR_BAD_ARG, "oetr_forward: NULL handle/input");
  if (oetr_status mrc = check_masks(h, mask1, mask2, true)) return mrc;
  int enc_layers = OETR_N_ENC;
  if (st) {
    if (st->struct_size != sizeof(oetr_stage_outputs))
      return fail(OETR_ERR_BAD_ARG, "oetr_stage_outputs size mismatch");
    enc_layers = st->enc_layers;
    if (enc_layers < 1 || enc_layers > OETR_N_ENC)
      return fail(OETR_ERR_BAD_ARG, "enc_layers must be in 1..8");
  }
  const bool full = enc_layers == OETR_N_ENC;
  if (full && (!box1 || !box2)) return fail(OETR_ERR_BAD_ARG, "oetr_forward: NULL box output");
  Geom g;
  if (!make_geom(n_pairs, hf1, wf1, hf2, wf2, &g))
    return fail(OETR_ERR_BAD_SHAPE, "invalid shape: need N>0 and 1 <= hf*wf <= " +
                                        std::to_string(OETR_MAX_TOKENS));
  if (full && (img_h1 < hf1 || img_h2 < hf2 || img_w1 <= 0 || img_w2 <= 0))
    return fail(OETR_ERR_BAD_SHAPE, "image size smaller than the token grid");
  Workspace w;
  oetr_status rc = check_ws(g, workspace, workspace_bytes, &w, h->attn_full);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = run_correlation(h, g, w, feat1, feat2, pos1, pos2, enc_layers, s, /*with_decoder=*/false, resident,
                       mask1, mask2);
  if (rc) return rc;
  const size_t r1 = (size_t)g.N * g.L[0], r2 = (size_t)g.N * g.L[1];
  if (st) {
    if ((rc = copy_out(st->memory1, w.x, r1 * C, s))) return rc;
    if ((rc = copy_out(st->memory2, w.x + r1 * C, r2 * C, s))) return rc;
  }
  if (!full) return OETR_OK;
  const float* hs1 = w.hs;
  const float* hs2 = w.hs + (size_t)g.N * C;
  float* cxy1 = w.cxy;
  float* cxy2 = w.cxy + 2 * g.N;
  float* tl1 = w.tlbr;
  float* tl2 = w.tlbr + 4 * g.N;
  HeatLaunch hp = heat_launch(h, g, w, w.x, w.x + r1 * C, hs1, hs2, cxy1, cxy2, img_h1, img_h2);
  hp.tlbr[0] = tl1; hp.tlbr[1] = tl2;
  hp.mask[0] = mask1; hp.mask[1] = mask2;
  hp.box[0] = box1; hp.box[1] = box2;
  hp.img_w[0] = img_w1; hp.img_w[1] = img_w2;
  hp.publish = publish;
  // Small batches: decoder || P_tap = W_tap.memory in one launch (the decoder chain, 50 us on 2N CUs, hides
  // behind the conv GEMMs), then the att-weighted combine.  Large batches (two-plane mode): the P buffer's
  // traffic (9 x rows x 1 KB written and read) costs more than the chain - decoder first, then the conv in
  // its direct 64-row form (heads.hip: k_heat_conv64), no P.
  const bool direct = h->mode == GM_SPLIT &&
                      (h->tail_mode >= 2 || (h->tail_mode == 0 && g.rows >= OETR_DIRECT_TAIL_MIN_ROWS));
  if (direct) {
    TRACED(h, s, K_DECODER, launch_decoder(dec_launch(h, g, w), s));
    TRACED(h, s, K_HEAT_CONV, launch_heat_conv64(hp, h->mode, s));
  } else {
    TRACED(h, s, K_DEC_CONVP, launch_decoder_convp(dec_launch(h, g, w, true), hp, w.convp, h->mode, s));
    TRACED(h, s, K_HEAT_COMBINE, launch_heat_combine(hp, w.convp, s));
  }
  TRACED(h, s, K_HEAT_FINAL, launch_heat_final(hp, s));  // + size regression + boxes
  if (st) {
    if ((rc = copy_out(st->hs1, hs1, (size_t)g.N * C, s))) return rc;
    if ((rc = copy_out(st->hs2, hs2, (size_t)g.N * C, s))) return rc;
    if ((rc = copy_out(st->logits1, w.logits, r1, s))) return rc;
    if ((rc = copy_out(st->logits2, w.logits + r1, r2, s))) return rc;
    if ((rc = copy_out(st->cxy1, cxy1, 2 * g.N, s))) return rc;
    if ((rc = copy_out(st->cxy2, cxy2, 2 * g.N, s))) return rc;
    if ((rc = copy_out(st->tlbr1, tl1, 4 * g.N, s))) return rc;
    if ((rc = copy_out(st->tlbr2, tl2, 4 * g.N, s))) return rc;
  }
  return OETR_OK;
}
}  // namespace

oetr_status oetr_forward_stages(oetr_handle h, const float* feat1, const float* feat2,
                                const float* pos1, const float* pos2, int n_pairs, int hf1,
                                int wf1, int hf2, int wf2, int img_h1, int img_w1, int img_h2,
                                int img_w2, void* workspace, size_t workspace_bytes,
                                float* box1, float* box2, const oetr_stage_outputs* st,
                                void* stream) {
  return forward_impl(h, feat1, feat2, pos1, pos2, n_pairs, hf1, wf1, hf2, wf2, img_h1, img_w1,
                      img_h2, img_w2, workspace, workspace_bytes, box1, box2, st, stream, false);
}

oetr_status oetr_forward_masked(oetr_handle h, const float* feat1, const float* feat2,
                                const float* pos1, const float* pos2, const float* mask1,
                                const float* mask2, int n_pairs, int hf1, int wf1, int hf2, int wf2,
                                int img_h1, int img_w1, int img_h2, int img_w2, void* workspace,
                                size_t workspace_bytes, float* box1, float* box2,
                                const oetr_stage_outputs* st, void* stream) {
  return forward_impl(h, feat1, feat2, pos1, pos2, n_pairs, hf1, wf1, hf2, wf2, img_h1, img_w1,
                      img_h2, img_w2, workspace, workspace_bytes, box1, box2, st, stream, false,
                      mask1, mask2);
}

oetr_status oetr_flagslot_device_pointer(void* host_slot, uint32_t** device_slot) {
  if (!host_slot || !device_slot) return fail(OETR_ERR_BAD_ARG, "oetr_flagslot_device_pointer: NULL argument");
  void* d = nullptr;
  const hipError_t e = hipHostGetDevicePointer(&d, host_slot, 0);
  if (e != hipSuccess || !d) {
    (void)hipGetLastError();
    return fail(OETR_ERR_BAD_ARG, "oetr_flagslot_device_pointer: not mapped host memory (hipHostMalloc / "
                                  "hipHostRegister with the mapped attribute)");
  }
  *device_slot = static_cast<uint32_t*>(d);
  return OETR_OK;
}

oetr_status oetr_forward_flagslot(oetr_handle h, const float* feat1, const float* feat2,
                                  const float* pos1, const float* pos2, const float* mask1,
                                  const float* mask2, int n_pairs, int hf1, int wf1, int hf2, int wf2,
                                  int img_h1, int img_w1, int img_h2, int img_w2, void* workspace,
                                  size_t workspace_bytes, float* box1, float* box2,
                                  uint32_t* flag_slot, void* stream) {
  if (!flag_slot) return fail(OETR_ERR_BAD_ARG, "oetr_forward_flagslot: NULL flag_slot");
  return forward_impl(h, feat1, feat2, pos1, pos2, n_pairs, hf1, wf1, hf2, wf2, img_h1, img_w1,
                      img_h2, img_w2, workspace, workspace_bytes, box1, box2, nullptr, stream, false,
                      mask1, mask2, flag_slot);
}

oetr_status oetr_forward_tokens_flagslot(oetr_handle h, int n_pairs, int hf1, int wf1, int hf2, int wf2,
                                         int img_h1, int img_w1, int img_h2, int img_w2, void* workspace,
                                         size_t workspace_bytes, float* box1, float* box2,
                                         uint32_t* flag_slot, void* stream) {
  if (!flag_slot) return fail(OETR_ERR_BAD_ARG, "oetr_forward_tokens_flagslot: NULL flag_slot");
  return forward_impl(h, nullptr, nullptr, nullptr, nullptr, n_pairs, hf1, wf1, hf2, wf2, img_h1,
                      img_w1, img_h2, img_w2, workspace, workspace_bytes, box1, box2, nullptr,
                      stream, true, nullptr, nullptr, flag_slot);
}

oetr_status oetr_token_buffers(oetr_handle h, int n_pairs, int hf1, int wf1, int hf2, int wf2,
                               void* workspace, size_t workspace_bytes, float** tokens1,
                               float** tokens2, float** pos_tokens1, float** pos_tokens2) {
  if (!h || !tokens1 || !tokens2 || !pos_tokens1 || !pos_tokens2)
    return fail(OETR_ERR_BAD_ARG, "oetr_token_buffers: NULL argument");
  Geom g;
  if (!make_geom(n_pairs, hf1, wf1, hf2, wf2, &g))
    return fail(OETR_ERR_BAD_SHAPE, "oetr_token_buffers: invalid shape");
  Workspace w;
  oetr_status rc = check_ws(g, workspace, workspace_bytes, &w, h->attn_full);
  if (rc) return rc;
  *tokens1 = w.x + (size_t)g.row0[0] * C;
  *tokens2 = w.x + (size_t)g.row0[1] * C;
  *pos_tokens1 = w.pos + (size_t)g.prow0[0] * C;
  *pos_tokens2 = w.pos + (size_t)g.prow0[1] * C;
  return OETR_OK;
}

oetr_status oetr_forward_tokens(oetr_handle h, int n_pairs, int hf1, int wf1, int hf2, int wf2,
                                int img_h1, int img_w1, int img_h2, int img_w2, void* workspace,
                                size_t workspace_bytes, float* box1, float* box2, void* stream) {
  return forward_impl(h, nullptr, nullptr, nullptr, nullptr, n_pairs, hf1, wf1, hf2, wf2, img_h1,
                      img_w1, img_h2, img_w2, workspace, workspace_bytes, box1, box2, nullptr,
                      stream, true);
}

oetr_status oetr_forward(oetr_handle h, const float* feat1, const float* feat2,
                         const float* pos1, const float* pos2, int n_pairs, int hf1, int wf1,
                         int hf2, int wf2, int img_h1, int img_w1, int img_h2, int img_w2,
                         void* workspace, size_t workspace_bytes, float* box1, float* box2,
                         void* stream) {
  return oetr_forward_stages(h, feat1, feat2, pos1, pos2, n_pairs, hf1, wf1, hf2, wf2, img_h1,
                             img_w1, img_h2, img_w2, workspace, workspace_bytes, box1, box2,
                             nullptr, stream);
}

oetr_status oetr_feature_correlation(oetr_handle h, const float* feat1, const float* feat2,
                                     const float* pos1, const float* pos2, int n_pairs, int hf1,
                                     int wf1, int hf2, int wf2, void* workspace,
                                     size_t workspace_bytes, float* hs1, float* hs2,
                                     float* memory1, float* memory2, void* stream) {
  return oetr_feature_correlation_masked(h, feat1, feat2, pos1, pos2, nullptr, nullptr, n_pairs, hf1, wf1,
                                         hf2, wf2, workspace, workspace_bytes, hs1, hs2, memory1,
                                         memory2, stream);
}

oetr_status oetr_feature_correlation_masked(oetr_handle h, const float* feat1, const float* feat2,
                                            const float* pos1, const float* pos2, const float* mask1,
                                            const float* mask2, int n_pairs, int hf1, int wf1, int hf2,
                                            int wf2, void* workspace, size_t workspace_bytes,
                                            float* hs1, float* hs2, float* memory1, float* memory2,
                                            void* stream) {
  if (!h || !feat1 || !feat2 || !pos1 || !pos2 || !hs1 || !hs2 || !memory1 || !memory2)
    return fail(OETR_ERR_BAD_ARG, "oetr_feature_correlation: NULL argument");
  if (oetr_status mrc = check_masks(h, mask1, mask2, true)) return mrc;
  Geom g;
  if (!make_geom(n_pairs, hf1, wf1, hf2, wf2, &g))
    return fail(OETR_ERR_BAD_SHAPE, "invalid shape");
  Workspace w;
  oetr_status rc = check_ws(g, workspace, workspace_bytes, &w, h->attn_full);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((rc = run_correlation(h, g, w, feat1, feat2, pos1, pos2, OETR_N_ENC, s, true, false, mask1, mask2))) return rc;
  const size_t r1 = (size_t)g.N * g.L[0], r2 = (size_t)g.N * g.L[1];
  if ((rc = copy_out(memory1, w.x, r1 * C, s))) return rc;
  if ((rc = copy_out(memory2, w.x + r1 * C, r2 * C, s))) return rc;
  if ((rc = copy_out(hs1, w.hs, (size_t)g.N * C, s))) return rc;
  return copy_out(hs2, w.hs + (size_t)g.N * C, (size_t)g.N * C, s);
}

oetr_status oetr_center_estimation(oetr_handle h, const float* hs1, const float* hs2,
                                   const float* memory1, const float* memory2, int n_pairs,
                                   int hf1, int wf1, int hf2, int wf2, int img_h1, int img_h2,
                                   void* workspace, size_t workspace_bytes, float* cxy1,
                                   float* cxy2, void* stream) {
  return oetr_center_estimation_masked(h, hs1, hs2, memory1, memory2, nullptr, nullptr, n_pairs, hf1, wf1,
                                       hf2, wf2, img_h1, img_h2, workspace, workspace_bytes, cxy1, cxy2,
                                       stream);
}

oetr_status oetr_center_estimation_masked(oetr_handle h, const float* hs1, const float* hs2,
                                          const float* memory1, const float* memory2,
                                          const float* mask1, const float* mask2, int n_pairs, int hf1,
                                          int wf1, int hf2, int wf2, int img_h1, int img_h2,
                                          void* workspace, size_t workspace_bytes, float* cxy1,
                                          float* cxy2, void* stream) {
  if (!h || !hs1 || !hs2 || !memory1 || !memory2 || !cxy1 || !cxy2)
    return fail(OETR_ERR_BAD_ARG, "oetr_center_estimation: NULL argument");
  if (oetr_status mrc = check_masks(h, mask1, mask2, false)) return mrc;
  Geom g;
  if (!make_geom(n_pairs, hf1, wf1, hf2, wf2, &g))
    return fail(OETR_ERR_BAD_SHAPE, "invalid shape");
  if (img_h1 < hf1 || img_h2 < hf2)
    return fail(OETR_ERR_BAD_SHAPE, "image height smaller than the token grid");
  Workspace w;
  oetr_status rc = check_ws(g, workspace, workspace_bytes, &w);
  if (rc) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  HeatLaunch hp = heat_launch(h, g, w, memory1, memory2, hs1, hs2, cxy1, cxy2, img_h1, img_h2);
  hp.mask[0] = mask1; hp.mask[1] = mask2;
  TRACED(h, s, K_HEAT_CONV, launch_heat_conv(hp, h->mode, s));
  TRACED(h, s, K_HEAT_FINAL, launch_heat_final(hp, s));
  return OETR_OK;
}

oetr_status oetr_size_regression(oetr_handle h, const float* hs1, const float* hs2, int n_pairs,
                                 float* tlbr1, float* tlbr2, void* stream) {
  if (!h || !hs1 || !hs2 || !tlbr1 || !tlbr2 || n_pairs <= 0)
    return fail(OETR_ERR_BAD_ARG, "oetr_size_regression: bad argument");
  HIP_TRY(launch_size_regression(h->heads, hs1, hs2, n_pairs, tlbr1, tlbr2,
                                 static_cast<hipStream_t>(stream)));
  return OETR_OK;
}

oetr_status oetr_box_tlbr_to_xyxy(const float* cxy, const float* tlbr, int n, int max_h,
                                  int max_w, float* box, void* stream) {
  if (!cxy || !tlbr || !box || n <= 0)
    return fail(OETR_ERR_BAD_ARG, "oetr_box_tlbr_to_xyxy: bad argument");
  HIP_TRY(launch_boxes(cxy, tlbr, n, max_h, max_w, box, static_cast<hipStream_t>(stream)));
  return OETR_OK;
}

size_t oetr_linear_attention_workspace_bytes(int n) {
  return n > 0 ? (size_t)n * NH * (HD * HD + HD) * sizeof(float) : 0;
}

oetr_status oetr_linear_attention(const float* q, const float* k, const float* v, int n, int L,
                                  int S, float* out, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  return oetr_linear_attention_masked(q, k, v, nullptr, nullptr, n, L, S, out, workspace, workspace_bytes,
                                      stream);
}

oetr_status oetr_linear_attention_masked(const float* q, const float* k, const float* v,
                                         const float* q_mask, const float* kv_mask, int n, int L, int S,
                                         float* out, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  if (!q || !k || !v || !out || n <= 0 || L <= 0 || S <= 0)
    return fail(OETR_ERR_BAD_ARG, "oetr_linear_attention: bad argument");
  if (!workspace || workspace_bytes < oetr_linear_attention_workspace_bytes(n))
    return fail(OETR_ERR_WORKSPACE, "oetr_linear_attention: workspace smaller than "
                                    "oetr_linear_attention_workspace_bytes(n)");
  HIP_TRY(launch_linear_attention(q, k, v, q_mask, kv_mask, n, L, S, out, static_cast<float*>(workspace),
                                  static_cast<hipStream_t>(stream)));
  return OETR_OK;
}

oetr_status oetr_full_attention(const float* q, const float* k, const float* v, int n, int L,
                                int S, float* out, void* stream) {
  if (!q || !k || !v || !out || n <= 0 || L <= 0 || S <= 0)
    return fail(OETR_ERR_BAD_ARG, "oetr_full_attention: bad argument");
  HIP_TRY(launch_full_attention(q, k, v, n, L, S, out, static_cast<hipStream_t>(stream)));
  return OETR_OK;
}

oetr_status oetr_full_attention_split(const float* q, const float* k, const float* v, int n, int L,
                                      int S, float* out, uint32_t* flags, void* stream) {
  if (!q || !k || !v || !out || n <= 0 || L <= 0 || S <= 0)
    return fail(OETR_ERR_BAD_ARG, "oetr_full_attention_split: bad argument");
  HIP_TRY(launch_full_attention_split(q, k, v, n, L, S, out, flags, static_cast<hipStream_t>(stream)));
  return OETR_OK;
}

// ---------------------------------------------------------------- neck ----
namespace {

// PatchMerging reductions (reference backbone.py:39-51): kernel, out channels,
// K slices of 16 kernel pixels, 128-column halves.
struct NeckConvShape { int ks, log2ks, cout, nsplit, nhalf; };
constexpr NeckConvShape kNeckConv[3] = {{4, 2, 256, 1, 2}, {8, 3, 128, 4, 1}, {16, 4, 128, 16, 1}};

// W [cout][256][ks][ks] -> per (K slice, column half, n-tile): f16 B fragments of
// the [16 pixels * 256 channels][32] slab, hi and lo*2^11 planes (k_neck_conv).
void pack_conv(Packer& pk, const float* W, const NeckConvShape& cs, size_t* hi_off, size_t* lo_off) {
  const int ks2 = cs.ks * cs.ks, steps = NECK_PIX * C / 16;
  const size_t plane_floats = (size_t)cs.cout * C * ks2 / 2;
  *hi_off = pk.reserve_aligned(plane_floats);
  *lo_off = pk.reserve_aligned(plane_floats);
  _Float16* hi = reinterpret_cast<_Float16*>(pk.buf.data() + *hi_off);
  _Float16* lo = reinterpret_cast<_Float16*>(pk.buf.data() + *lo_off);
  for (int sp = 0; sp < cs.nsplit; ++sp)
    for (int nh = 0; nh < cs.nhalf; ++nh)
      for (int nt = 0; nt < 4; ++nt)
        for (int st = 0; st < steps; ++st)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int n = nh * 128 + nt * 32 + (lane & 31);
              const int k = st * 16 + 8 * (lane >> 5) + j;      // within the slab
              const int pix = NECK_PIX * sp + k / C, c = k % C;  // kernel pixel ky*ks+kx
              const float w = W[((size_t)n * C + c) * ks2 + pix];
              const _Float16 h = (_Float16)w;
              const size_t idx = (((((size_t)sp * cs.nhalf + nh) * 4 + nt) * steps + st) * 64 + lane) * 8 + j;
              hi[idx] = h;
              lo[idx] = (_Float16)((w - (float)h) * SPLIT_SCALE);
            }
}

// The same slabs in k_neck_conv_rw's k16-step order: stage (ky in slice, x parity,
// 32-channel chunk), then tap (kx = 2*tap + parity), then the two 16-channel steps.
void pack_conv_rw(Packer& pk, const float* W, const NeckConvShape& cs, size_t* hi_off, size_t* lo_off) {
  const int ks2 = cs.ks * cs.ks, steps = NECK_PIX * C / 16;
  const int taps = cs.ks / 2, ksteps = 2 * taps, nky = NECK_PIX / cs.ks;
  const size_t plane_floats = (size_t)cs.cout * C * ks2 / 2;
  *hi_off = pk.reserve_aligned(plane_floats);
  *lo_off = pk.reserve_aligned(plane_floats);
  _Float16* hi = reinterpret_cast<_Float16*>(pk.buf.data() + *hi_off);
  _Float16* lo = reinterpret_cast<_Float16*>(pk.buf.data() + *lo_off);
  for (int sp = 0; sp < cs.nsplit; ++sp)
    for (int nh = 0; nh < cs.nhalf; ++nh)
      for (int nt = 0; nt < 4; ++nt)
        for (int st = 0; st < steps; ++st) {
          const int stage = st / ksteps, kk = st % ksteps;
          const int ky = sp * nky + (stage >> 4), par = (stage >> 3) & 1, cq = stage & 7;
          const int kx = 2 * (kk >> 1) + par, c0 = 32 * cq + 16 * (kk & 1);
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int n = nh * 128 + nt * 32 + (lane & 31);
              const int c = c0 + 8 * (lane >> 5) + j;
              const float w = W[((size_t)n * C + c) * ks2 + ky * cs.ks + kx];
              const _Float16 h = (_Float16)w;
              const size_t idx = (((((size_t)sp * cs.nhalf + nh) * 4 + nt) * steps + st) * 64 + lane) * 8 + j;
              hi[idx] = h;
              lo[idx] = (_Float16)((w - (float)h) * SPLIT_SCALE);
            }
        }
}

bool make_neck_geom(int n_img, int hb, int wb, NeckGeom* g) {
  if (n_img <= 0 || hb < 2 || wb < 2 || hb > 400 || wb > 400) return false;
  const long lo = (long)(hb / 2) * (wb / 2);
  if (lo > OETR_MAX_TOKENS) return false;
  // (32-bit byte offsets into X: (rows_in + 1) rows of 1024 B must stay below 2^32)
  if ((long)n_img * hb * wb > (1L << 22) - 2 || n_img >= (1 << 13)) return false;
  g->n_img = n_img; g->hb = hb; g->wb = wb; g->ho = hb / 2; g->wo = wb / 2;
  g->HW = hb * wb;
  g->rows_in = n_img * g->HW;
  g->M = n_img * g->ho * g->wo;
  return true;
}

struct NeckWorkspace {
  uint32_t* flags;   // status word (first 256 bytes)
  _Float16 *xh, *xl;
  float* part[3];
  size_t bytes;
};
NeckWorkspace neck_carve(const NeckGeom& g, void* base) {
  NeckWorkspace w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? static_cast<char*>(base) + off : nullptr;
    off += (bytes + 255) & ~size_t(255);
    return p;
  };
  // X: per input pixel (+ one all-zero padding row) eight 128-byte chunks
  // [32 channels hi | 32 channels lo] - what one conv stage gathers per pixel is one
  // cache line.  xl = xh + 32 halves: "the lo plane" of the same buffer.
  w.flags = reinterpret_cast<uint32_t*>(take(OETR_WORKSPACE_STATUS_BYTES));
  const size_t xbytes = ((size_t)g.rows_in + 1) * NECK_XROW * sizeof(_Float16);
  w.xh = reinterpret_cast<_Float16*>(take(xbytes));
  w.xl = w.xh ? w.xh + 32 : nullptr;
  for (int i = 0; i < 3; ++i)
    w.part[i] = reinterpret_cast<float*>(
        take((size_t)kNeckConv[i].nsplit * g.M * kNeckConv[i].cout * sizeof(float)));
  w.bytes = off;
  return w;
}

}  // namespace

oetr_status oetr_neck_create(const oetr_neck_weights* w, int device, oetr_neck_handle* out) {
  if (!w || !out) return fail(OETR_ERR_BAD_ARG, "oetr_neck_create: NULL argument");
  *out = nullptr;
  if (w->struct_size != sizeof(oetr_neck_weights) || w->abi_version != OETR_ABI_VERSION)
    return fail(OETR_ERR_BAD_ARG, "oetr_neck_create: oetr_neck_weights size/ABI mismatch");
  {
    const float* const* p = reinterpret_cast<const float* const*>(&w->input_proj_w);
    const size_t n = (sizeof(oetr_neck_weights) - offsetof(oetr_neck_weights, input_proj_w)) / sizeof(float*);
    for (size_t i = 0; i < n; ++i)
      if (!p[i]) return fail(OETR_ERR_BAD_ARG, "oetr_neck_create: weight pointer #" +
                                                   std::to_string(i) + " is NULL");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
    return fail(OETR_ERR_NO_DEVICE, "no HIP device " + std::to_string(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (!strstr(prop.gcnArchName, "gfx950"))
    return fail(OETR_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName +
                                        ", this library is built for gfx950 only");
  CHECK_W("input_proj.weight", w->input_proj_w, (size_t)C * BBC, true);
  CHECK_W("input_proj.bias", w->input_proj_b, C, false);
  CHECK_W("patchmerging.norm.weight", w->norm_w, C, false);
  CHECK_W("patchmerging.norm.bias", w->norm_b, C, false);
  for (int i = 0; i < 3; ++i) {
    const NeckConvShape& cs = kNeckConv[i];
    CHECK_W("patchmerging.reductions.weight", w->reduction_w[i], (size_t)cs.cout * C * cs.ks * cs.ks, true);
    CHECK_W("patchmerging.reductions.bias", w->reduction_b[i], cs.cout, false);
  }
  CHECK_W("input_proj2.weight", w->input_proj2_w, (size_t)C * 2 * C, true);
  CHECK_W("input_proj2.bias", w->input_proj2_b, C, false);
  Packer pk;
  size_t pwh[2], pwl[2], cwh[3], cwl[3], rwh[3], rwl[3], cb[3], owh, owl;
  for (int kh = 0; kh < 2; ++kh)  // input_proj, K halves [256][512] of the [256][1024] matrix
    pk.frag_h(GM_SPLIT, w->input_proj_w + kh * 512, C, 512, &pwh[kh], &pwl[kh], BBC);
  const size_t pb = pk.copy(w->input_proj_b, C), lw = pk.copy(w->norm_w, C), lb = pk.copy(w->norm_b, C);
  for (int i = 0; i < 3; ++i) {
    pack_conv(pk, w->reduction_w[i], kNeckConv[i], &cwh[i], &cwl[i]);
    pack_conv_rw(pk, w->reduction_w[i], kNeckConv[i], &rwh[i], &rwl[i]);
    cb[i] = pk.copy(w->reduction_b[i], kNeckConv[i].cout);
  }
  pk.frag_h(GM_SPLIT, w->input_proj2_w, C, 2 * C, &owh, &owl);
  const size_t ob = pk.copy(w->input_proj2_b, C);

  oetr_neck_ctx* h = new oetr_neck_ctx();
  h->device = device;
  h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  int prev = 0;
  (void)hipGetDevice(&prev);
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc(&h->dev, pk.buf.size() * sizeof(float));
  if (e == hipSuccess)
    e = hipMemcpy(h->dev, pk.buf.data(), pk.buf.size() * sizeof(float), hipMemcpyHostToDevice);
  (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    if (h->dev) (void)hipFree(h->dev);
    delete h;
    return hip_fail(e, "oetr_neck_create: uploading weights");
  }
  const float* B = h->dev;
  auto F4 = [&](size_t off) { return reinterpret_cast<const f32x4*>(B + off); };
  for (int kh = 0; kh < 2; ++kh) { h->proj_wh[kh] = F4(pwh[kh]); h->proj_wl[kh] = F4(pwl[kh]); }
  h->proj_b = B + pb; h->ln_w = B + lw; h->ln_b = B + lb;
  for (int i = 0; i < 3; ++i) {
    h->conv_wh[i] = F4(cwh[i]); h->conv_wl[i] = F4(cwl[i]); h->conv_b[i] = B + cb[i];
    h->conv_wh_rw[i] = F4(rwh[i]); h->conv_wl_rw[i] = F4(rwl[i]);
  }
  h->out_wh = F4(owh); h->out_wl = F4(owl); h->out_b = B + ob;
  *out = h;
  return OETR_OK;
}

void oetr_neck_destroy(oetr_neck_handle h) {
  if (!h) return;
  if (h->dev) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(h->device);
    (void)hipFree(h->dev);
    (void)hipSetDevice(prev);
  }
  delete h;
}

size_t oetr_neck_workspace_bytes(oetr_neck_handle h, int n_images, int hb, int wb) {
  NeckGeom g;
  (void)h;  // shape-only: usable before a handle exists
  if (!make_neck_geom(n_images, hb, wb, &g)) return 0;
  return neck_carve(g, nullptr).bytes;
}

namespace {
oetr_status neck_forward_impl(oetr_neck_handle h, const float* backbone_feat, int n_images,
                              int hb, int wb, void* workspace, size_t workspace_bytes,
                              float* feat_out, void* stream, bool token_major,
                              uint32_t* status_word = nullptr) {
  if (!h || !backbone_feat || !feat_out)
    return fail(OETR_ERR_BAD_ARG, "oetr_neck_forward: NULL argument");
  NeckGeom g;
  if (!make_neck_geom(n_images, hb, wb, &g))
    return fail(OETR_ERR_BAD_SHAPE, "oetr_neck_forward: need n_images > 0, 2 <= hb,wb <= 400 and "
                                    "(hb/2)*(wb/2) <= " + std::to_string(OETR_MAX_TOKENS));
  if (!workspace) return fail(OETR_ERR_WORKSPACE, "workspace is NULL");
  if (reinterpret_cast<uintptr_t>(workspace) & 255)
    return fail(OETR_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  const NeckWorkspace w = neck_carve(g, workspace);
  if (w.bytes > workspace_bytes)
    return fail(OETR_ERR_WORKSPACE, "workspace too small: need " + std::to_string(w.bytes) +
                                        " bytes, got " + std::to_string(workspace_bytes));
  hipStream_t s = static_cast<hipStream_t>(stream);

  NeckProjLaunch pp;
  pp.g = g; pp.bb = backbone_feat;
  for (int kh = 0; kh < 2; ++kh) { pp.wh[kh] = h->proj_wh[kh]; pp.wl[kh] = h->proj_wl[kh]; }
  pp.bias = h->proj_b; pp.ln_w = h->ln_w; pp.ln_b = h->ln_b;
  pp.xh = w.xh; pp.xl = w.xl;
  pp.flags = status_word ? status_word : w.flags;
  TRACED(h, s, K_NECK_PROJ, launch_neck_proj(pp, s));

  NeckConvLaunch cp;
  cp.g = g; cp.xh = w.xh; cp.xl = w.xl;
  int items = 0;
  const int order[3] = {2, 1, 0};  // k = 16 (16 slices), k = 8 (4), k = 4 (2 column halves)
  for (int oi = 0; oi < 3; ++oi) {
    const int i = order[oi];
    NeckConvDesc& d = cp.conv[oi];
    const NeckConvShape& cs = kNeckConv[i];
    d.wh = h->conv_wh[i]; d.wl = h->conv_wl[i];
    d.wh_rw = h->conv_wh_rw[i]; d.wl_rw = h->conv_wl_rw[i];
    d.part = w.part[i];
    d.log2ks = cs.log2ks; d.pad = (cs.ks - 2) / 2;
    d.nsplit = cs.nsplit; d.nhalf = cs.nhalf; d.ncols = cs.cout;
    d.item0 = items;
    items += cs.nsplit * cs.nhalf;
  }
  cp.items_per_mt = items;
  // auto: the row-window kernel in its one-wave-per-SIMD shape (fastest at every size tried)
  cp.row_window = h->conv_kernel == 0 ? (g.wo >= NECK_RW_MIN_WO ? 2 : 0)
                                      : (h->conv_kernel == 3 ? 2 : h->conv_kernel == 2);
  if (cp.row_window)   // no 256-row shapes (they spill); 192 rows measured best at every size tried
    cp.mt_rows = h->conv_rows > 0 ? min(h->conv_rows, 192) : (g.M > 128 ? 192 : 128);
  else
    cp.mt_rows = h->conv_rows > 0 ? h->conv_rows : neck_conv_rows(g.M, items, h->num_cus, 256);
  if (cp.row_window && g.wo < NECK_RW_MIN_WO)
    return fail(OETR_ERR_BAD_SHAPE, "oetr_neck_forward: the row-window conv kernel needs an output map >= 16 wide");
  cp.nblocks = items * ((g.M + cp.mt_rows - 1) / cp.mt_rows);
  TRACED(h, s, K_NECK_CONV, launch_neck_conv(cp, s));

  NeckOutLaunch op;
  op.g = g;
  for (int i = 0; i < 3; ++i) { op.part[i] = w.part[i]; op.nsplit[i] = kNeckConv[i].nsplit; op.bias[i] = h->conv_b[i]; }
  op.wh = h->out_wh; op.wl = h->out_wl; op.bias2 = h->out_b;
  op.feat = token_major ? nullptr : feat_out;
  op.tokens = token_major ? feat_out : nullptr;
  op.flags = status_word ? status_word : w.flags;
  TRACED(h, s, K_NECK_OUT, launch_neck_out(op, s));
  return OETR_OK;
}
}  // namespace

oetr_status oetr_neck_forward(oetr_neck_handle h, const float* backbone_feat, int n_images,
                              int hb, int wb, void* workspace, size_t workspace_bytes,
                              float* feat_out, void* stream) {
  return neck_forward_impl(h, backbone_feat, n_images, hb, wb, workspace, workspace_bytes,
                           feat_out, stream, false);
}

oetr_status oetr_neck_forward_tokens(oetr_neck_handle h, const float* backbone_feat, int n_images,
                                     int hb, int wb, void* workspace, size_t workspace_bytes,
                                     float* tokens_out, void* stream) {
  return neck_forward_impl(h, backbone_feat, n_images, hb, wb, workspace, workspace_bytes,
                           tokens_out, stream, true);
}

oetr_status oetr_neck_forward_tokens_status(oetr_neck_handle h, const float* backbone_feat, int n_images,
                                            int hb, int wb, void* workspace, size_t workspace_bytes,
                                            float* tokens_out, uint32_t* status_word, void* stream) {
  if (!status_word) return fail(OETR_ERR_BAD_ARG, "oetr_neck_forward_tokens_status: NULL status_word");
  return neck_forward_impl(h, backbone_feat, n_images, hb, wb, workspace, workspace_bytes,
                           tokens_out, stream, true, status_word);
}

namespace {
// The status word is the first word of a workspace (forward and neck alike): per workspace,
// hence per stream - a query sees and clears only the calls that used THIS workspace.
oetr_status read_flags(int device, void* workspace, void* stream, uint32_t* flags, int clear, bool sync,
                       const char* what) {
  if (!flags || !workspace) return fail(OETR_ERR_BAD_ARG, std::string(what) + ": NULL workspace / output");
  if (reinterpret_cast<uintptr_t>(workspace) & 255)
    return fail(OETR_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  uint32_t* dev_flags = static_cast<uint32_t*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int prev = 0;
  (void)hipGetDevice(&prev);
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMemcpyAsync(flags, dev_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess && clear) e = hipMemsetAsync(dev_flags, 0, sizeof(uint32_t), s);
  if (e == hipSuccess && sync) e = hipStreamSynchronize(s);
  (void)hipSetDevice(prev);
  if (e != hipSuccess) return hip_fail(e, what);
  return OETR_OK;
}
}  // namespace

oetr_status oetr_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
  if (!workspace || workspace_bytes < OETR_WORKSPACE_STATUS_BYTES)
    return fail(OETR_ERR_WORKSPACE, "oetr_workspace_init: workspace is NULL or smaller than its status block");
  if (reinterpret_cast<uintptr_t>(workspace) & 255)
    return fail(OETR_ERR_WORKSPACE, "workspace must be 256-byte aligned");
  HIP_TRY(hipMemsetAsync(workspace, 0, OETR_WORKSPACE_STATUS_BYTES, static_cast<hipStream_t>(stream)));
  return OETR_OK;
}

oetr_status oetr_query_flags(oetr_handle h, void* workspace, void* stream, uint32_t* flags, int clear) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_query_flags: NULL handle");
  return read_flags(h->device, workspace, stream, flags, clear, true, "oetr_query_flags");
}
oetr_status oetr_read_flags_async(oetr_handle h, void* workspace, uint32_t* host_flags, int clear, void* stream) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_read_flags_async: NULL handle");
  return read_flags(h->device, workspace, stream, host_flags, clear, false, "oetr_read_flags_async");
}
oetr_status oetr_neck_query_flags(oetr_neck_handle h, void* workspace, void* stream, uint32_t* flags, int clear) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_neck_query_flags: NULL handle");
  return read_flags(h->device, workspace, stream, flags, clear, true, "oetr_neck_query_flags");
}
oetr_status oetr_neck_read_flags_async(oetr_neck_handle h, void* workspace, uint32_t* host_flags, int clear,
                                       void* stream) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_neck_read_flags_async: NULL handle");
  return read_flags(h->device, workspace, stream, host_flags, clear, false, "oetr_neck_read_flags_async");
}

oetr_status oetr_neck_set_conv_kernel(oetr_neck_handle h, int kind) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_neck_set_conv_kernel: NULL handle");
  if (kind < 0 || kind > 3)
    return fail(OETR_ERR_BAD_ARG, "oetr_neck_set_conv_kernel: kind must be 0 (auto), 1 (gather), 2 (row window) or 3 (row window, one wave per SIMD)");
  h->conv_kernel = kind;
  return OETR_OK;
}

oetr_status oetr_neck_set_conv_rows(oetr_neck_handle h, int rows) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_neck_set_conv_rows: NULL handle");
  if (rows != 0 && rows != 256 && rows != 192 && rows != 128)
    return fail(OETR_ERR_BAD_ARG, "oetr_neck_set_conv_rows: rows must be 0 (auto), 256, 192 or 128");
  h->conv_rows = rows;
  return OETR_OK;
}

oetr_status oetr_neck_set_trace(oetr_neck_handle h, oetr_trace_handle t) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_neck_set_trace: NULL handle");
  h->trace = t;
  return OETR_OK;
}

oetr_status oetr_trace_create(int max_launches, oetr_trace_handle* out) {
  if (!out || max_launches <= 0) return fail(OETR_ERR_BAD_ARG, "oetr_trace_create: bad argument");
  oetr_trace* t = new oetr_trace();
  t->ev.resize(2 * (size_t)max_launches);
  t->kid.resize(max_launches);
  for (auto& e : t->ev) {
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) { delete t; return hip_fail(rc, "hipEventCreate"); }
  }
  *out = t;
  return OETR_OK;
}

void oetr_trace_destroy(oetr_trace_handle t) {
  if (!t) return;
  for (auto& e : t->ev) (void)hipEventDestroy(e);
  delete t;
}

oetr_status oetr_set_encoder_tile(oetr_handle h, int rows) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_set_encoder_tile: NULL handle");
  if (rows != 0 && rows != TM && rows != 64)
    return fail(OETR_ERR_BAD_ARG, "oetr_set_encoder_tile: rows must be 0 (auto), 32 or 64");
  h->enc_tile = rows;
  return OETR_OK;
}

oetr_status oetr_set_tail_mode(oetr_handle h, int mode) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_set_tail_mode: NULL handle");
  if (mode < 0 || mode > 3)
    return fail(OETR_ERR_BAD_ARG, "oetr_set_tail_mode: 0 (auto), 1 (P form), 2 (direct form) or 3 (direct form, per-tap staging)");
  if (mode >= 2 && h->mode != GM_SPLIT)
    return fail(OETR_ERR_UNSUPPORTED, "the direct 64-row heat-map conv is built for the two-plane dtypes (F32_SPLIT_F16, F32_SPLIT_QK16)");
  h->tail_mode = mode;
  return OETR_OK;
}

oetr_status oetr_set_decoder_split(oetr_handle h, int k) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_set_decoder_split: NULL handle");
  if (k != 0 && k != 1 && k != DEC_SPLIT_K)
    return fail(OETR_ERR_BAD_ARG, "oetr_set_decoder_split: 0 (auto), 1 or 4 workgroups per image");
  h->dec_split = k;
  return OETR_OK;
}

oetr_status oetr_debug_decoder_fault(oetr_handle h, int on) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_debug_decoder_fault: NULL handle");
  h->dec_fault = on ? 1 : 0;
  return OETR_OK;
}

oetr_status oetr_debug_mfma_rate(int device, double seconds, double* tflops, void* stream) {
  if (!tflops || !(seconds > 0.0) || seconds > 30.0)
    return fail(OETR_ERR_BAD_ARG, "oetr_debug_mfma_rate: NULL output or seconds outside (0, 30]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
    return fail(OETR_ERR_NO_DEVICE, "no HIP device " + std::to_string(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  int prev = 0;
  (void)hipGetDevice(&prev);
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = measure_mfma_rate(prop.multiProcessorCount, seconds, tflops, static_cast<hipStream_t>(stream));
  (void)hipSetDevice(prev);
  if (e != hipSuccess) return hip_fail(e, "oetr_debug_mfma_rate");
  return OETR_OK;
}

oetr_status oetr_set_state_prereduce(oetr_handle h, int on) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_set_state_prereduce: NULL handle");
  if (on < -1 || on > 1) return fail(OETR_ERR_BAD_ARG, "oetr_set_state_prereduce: -1 (auto), 0 (off) or 1 (a reduction launch between the encoder launches)");
  h->kv_prereduce = on;
  return OETR_OK;
}

oetr_status oetr_set_attention(oetr_handle h, oetr_attention mode) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_set_attention: NULL handle");
  if (mode != OETR_ATTENTION_LINEAR && mode != OETR_ATTENTION_FULL)
    return fail(OETR_ERR_BAD_ARG, "oetr_set_attention: unknown mode");
  if (mode == OETR_ATTENTION_FULL && h->policy != 0)
    return fail(OETR_ERR_UNSUPPORTED, "attention 'full' is not built for OETR_DTYPE_F32_SPLIT_QK16");
  if (mode == OETR_ATTENTION_FULL && !gm_f16_range(h->mode) && h->mode != GM_F32)
    return fail(OETR_ERR_UNSUPPORTED, "attention 'full' is built for OETR_DTYPE_F32_SPLIT_F16, OETR_DTYPE_F16 and OETR_DTYPE_F32");
  h->attn_full = mode == OETR_ATTENTION_FULL;
  return OETR_OK;
}

oetr_status oetr_set_trace(oetr_handle h, oetr_trace_handle t) {
  if (!h) return fail(OETR_ERR_BAD_ARG, "oetr_set_trace: NULL handle");
  h->trace = t;
  return OETR_OK;
}

oetr_status oetr_trace_summary(oetr_trace_handle t, int* n_kernels,
                               const char* names[OETR_TRACE_MAX_KERNELS],
                               int launches[OETR_TRACE_MAX_KERNELS],
                               float total_ms[OETR_TRACE_MAX_KERNELS]) {
  if (!t || !n_kernels || !names || !launches || !total_ms)
    return fail(OETR_ERR_BAD_ARG, "oetr_trace_summary: NULL argument");
  static_assert(K_COUNT <= OETR_TRACE_MAX_KERNELS, "raise OETR_TRACE_MAX_KERNELS");
  for (int i = 0; i < K_COUNT; ++i) { names[i] = kKernelNames[i]; launches[i] = 0; total_ms[i] = 0.f; }
  for (int i = 0; i < t->used; ++i) {
    HIP_TRY(hipEventSynchronize(t->ev[2 * i + 1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, t->ev[2 * i], t->ev[2 * i + 1]));
    launches[t->kid[i]]++;
    total_ms[t->kid[i]] += ms;
  }
  *n_kernels = K_COUNT;
  const int dropped = t->dropped;
  t->used = 0;
  t->dropped = 0;
  if (dropped) return fail(OETR_ERR_BAD_ARG, std::to_string(dropped) + " launches not traced: pool too small");
  return OETR_OK;
}

#ifdef OETR_PHASE_TIMING
// debug: copy the phase stamps of the LAST encoder launch sequence to the host
int oetr_debug_read_tbuf(long long* host, int n_blocks) {
  if (!g_tbuf) return 1;
  return hipMemcpy(host, g_tbuf, sizeof(long long) * 16 * n_blocks, hipMemcpyDeviceToHost) != hipSuccess;
}
int oetr_debug_read_tbuf_decoder(long long* host, int n_blocks) {
  if (!g_tbuf) return 1;
  return hipMemcpy(host, g_tbuf + 16 * 4096, sizeof(long long) * 16 * n_blocks, hipMemcpyDeviceToHost) != hipSuccess;
}
#endif

}  // extern "C"
