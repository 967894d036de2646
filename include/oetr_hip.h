/*
 * liboetr_hip.so - C ABI of the MI355X (gfx950) OETR hot path.
 *
 * The reference (TencentYoutuResearch/ImageMatching-OETR) is pure Python and
 * has no FFI of its own; the seams this library replaces are the Python
 * methods of `OETR` in the reference's src/model.py.  Each entry point below
 * names the reference interface it stands in for (file:line relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds
 * on the reference side.
 *
 * Conventions
 *  - every tensor is fp32, contiguous, DEVICE memory unless marked host;
 *  - caller owns inputs, outputs and the workspace; the library owns only its
 *    repacked copy of the weights (created by oetr_create);
 *  - calls only ENQUEUE work on `stream` (a hipStream_t passed as void*): no
 *    allocation, no synchronisation, so a call sequence is hipGraph-capturable;
 *  - forward calls never modify a handle (the sticky status word of
 *    oetr_query_flags lives in the caller's workspace): concurrent calls with
 *    distinct workspaces on distinct streams are safe.  The explicit setters
 *    (oetr_set_encoder_tile, oetr_set_attention, oetr_set_trace) DO modify it:
 *    call them before the first forward call or while no call is in flight, and
 *    re-query oetr_workspace_bytes after oetr_set_attention;
 *  - no environment variable is read by the library;
 *  - every function returns an oetr_status; oetr_last_error() gives the text
 *    for the calling thread.  Nothing throws.
 *  - `N` is the number of image PAIRS; side 1 has token grid hf1 x wf1
 *    (L1 = hf1*wf1 tokens), side 2 hf2 x wf2.  C = 256 channels, 8 heads.
 */
#ifndef OETR_HIP_H_
#define OETR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: the status word moved from the handle into the workspace (oetr_query_flags takes the
 *    workspace; oetr_workspace_init, oetr_read_flags_async are new), oetr_linear_attention takes
 *    a workspace, OETR_DTYPE_F32_SPLIT_QK16
 * 3: the status block of a workspace grew from 256 bytes to OETR_WORKSPACE_STATUS_BYTES (the split
 *    decoder's call counters and exchange granules live there: oetr_set_decoder_split);
 *    OETR_FLAG_EXCHANGE
 * 4: forward_dummy's optional masks: oetr_forward_masked, oetr_feature_correlation_masked,
 *    oetr_center_estimation_masked, oetr_linear_attention_masked (new exports; nothing else changed)
 * 5: oetr_debug_decoder_fault (new export, tests only); a timed-out split decoder publishes nothing
 *    further and the caller re-initialises the status block (OETR_FLAG_EXCHANGE below)
 * 6: the status word is PUBLISHED BY THE LAST KERNEL of a forward call instead of by two runtime
 *    dispatches behind it: oetr_flagslot_device_pointer, oetr_forward_flagslot,
 *    oetr_forward_tokens_flagslot, oetr_neck_forward_tokens_status; oetr_debug_mfma_rate (measurements only)
 *    (new exports; nothing else changed) */
#define OETR_ABI_VERSION 6
#define OETR_D_MODEL 256
#define OETR_N_HEAD 8
#define OETR_N_ENC 8 /* self,cross x4  - reference src/models/transformer.py:295 */
#define OETR_N_DEC 2 /* reference src/models/transformer.py:304 */
#define OETR_MAX_TOKENS 10000 /* NECK.MAX_SHAPE 100x100, reference src/config/default.py:25-28 */

typedef enum {
  OETR_OK = 0,
  OETR_ERR_BAD_ARG = 1,      /* null pointer, N<=0, bad struct size        */
  OETR_ERR_BAD_SHAPE = 2,    /* token grid empty or > OETR_MAX_TOKENS      */
  OETR_ERR_UNSUPPORTED = 3,  /* dtype / masks / attention mode not built   */
  OETR_ERR_WORKSPACE = 4,    /* workspace too small or misaligned          */
  OETR_ERR_HIP = 5,          /* a HIP runtime call failed                  */
  OETR_ERR_NO_DEVICE = 6     /* no gfx950 device visible                   */
} oetr_status;

/* Arithmetic of the GEMM-shaped stages.  In every mode accumulation, LayerNorm,
 * phi = elu+1, the attention normaliser, softmax, GroupNorm and the residual
 * stream are fp32; the modes differ in how the GEMM OPERANDS are represented:
 *   OETR_DTYPE_F32           exact fp32 products on v_mfma_f32_32x32x2_f32
 *   OETR_DTYPE_F32_SPLIT_F16 fp32-class results from 3 f16 MFMAs per product
 *                            (a = ah + al/2^11 split of both operands; needs
 *                            |GEMM inputs| < 65504, see oetr_status_flags)
 *   OETR_DTYPE_F16           operands rounded to f16 (RNE), one
 *                            v_mfma_f32_32x32x16_f16 per product ("fp16 with fp32
 *                            accumulate", BASELINE configs[4]); same range limit
 *   OETR_DTYPE_BF16          operands rounded to bf16 (RNE), one
 *                            v_mfma_f32_32x32x16_bf16 per product ("bf16 MFMA
 *                            attention", BASELINE configs[2]); fp32 range
 *   OETR_DTYPE_F32_SPLIT_QK16  F32_SPLIT_F16 with a per-GEMM-site precision POLICY: the
 *                            encoder's Q and K projections (reference
 *                            src/models/transformer.py:127-128) and the decoder's K
 *                            projection of the memory (:62) take f16-rounded operands (one
 *                            MFMA per product), every other site keeps the 3-MFMA split.
 *                            Those are the sites that stay inside the north_star bar (boxes
 *                            within 1e-3 IoU of the fp32 reference, sharpened heads included)
 *                            when reduced - phi(Q) enters numerator and normaliser alike, K
 *                            only through sums over all source tokens; V, merge, the MLP and
 *                            the attention contractions themselves do not (per-site table:
 *                            profiles/r3_site_drift.jsonl).  Both encoder workgroup shapes.
 * The reference itself is fp32-only (no autocast anywhere); the two all-rounded 16-bit modes
 * trade parity margin (tests/test_gpu_precision.py records the drift: they MISS the 1e-3 IoU
 * bar) for MFMA rate; QK16 is the reduced mode that meets it. */
typedef enum {
  OETR_DTYPE_F32 = 0,
  OETR_DTYPE_F32_SPLIT_F16 = 1,
  OETR_DTYPE_F16 = 2,
  OETR_DTYPE_BF16 = 3,
  OETR_DTYPE_F32_SPLIT_QK16 = 4
} oetr_dtype;

/* Encoder layer i - reference src/models/transformer.py:83-102.
 * Linear weights are torch layout [out][in], row-major. */
typedef struct {
  const float *q_proj, *k_proj, *v_proj, *merge; /* [256][256], no bias */
  const float *mlp0;                             /* [512][256] */
  const float *mlp2;                             /* [256][512] */
  const float *pre_norm_q_w, *pre_norm_q_b;      /* [256] */
  const float *pre_norm_kv_w, *pre_norm_kv_b;    /* [256] */
  const float *norm2_w, *norm2_b;                /* [256] */
} oetr_encoder_layer_weights;

/* MultiHeadAttention - reference src/models/transformer.py:46-53 */
typedef struct {
  const float *q_proj_w, *q_proj_b; /* [256][256], [256] */
  const float *k_proj_w, *k_proj_b;
  const float *v_proj_w, *v_proj_b;
  const float *merge;               /* [256][256], no bias */
} oetr_mha_weights;

/* DecoderLayer - reference src/models/transformer.py:190-222 (live params) */
typedef struct {
  oetr_mha_weights self_attn, multihead_attn;
  const float *mlp0; /* [512][256] */
  const float *mlp2; /* [256][512] */
  const float *norm1_w, *norm1_b, *norm2_w, *norm2_b, *norm3_w, *norm3_b;
} oetr_decoder_layer_weights;

/* All HOST pointers; copied and repacked by oetr_create. */
typedef struct {
  uint32_t struct_size; /* = sizeof(oetr_weights) */
  uint32_t abi_version; /* = OETR_ABI_VERSION */
  oetr_encoder_layer_weights enc[OETR_N_ENC];
  oetr_decoder_layer_weights dec[OETR_N_DEC];
  const float *query_embed1, *query_embed2;   /* [256]   model.py:80-81 */
  const float *tlbr0_w;                       /* [256][256] model.py:60 */
  const float *tlbr2_w, *tlbr2_b;             /* [4][256],[4] model.py:62 */
  const float *heat_conv_w, *heat_conv_b;     /* [256][256][3][3],[256] model.py:66-73 */
  const float *heat_gn_w, *heat_gn_b;         /* [256] GroupNorm(32) model.py:74 */
  const float *heat_out_w, *heat_out_b;       /* [256] (1x1 conv), [1] model.py:76 */
} oetr_weights;

typedef struct oetr_ctx *oetr_handle;

/* Optional intermediate outputs (device pointers, any may be NULL).
 * Used by parity tests to pin tensors the boxes alone do not constrain. */
typedef struct {
  uint32_t struct_size;     /* = sizeof(oetr_stage_outputs) */
  int32_t enc_layers;       /* encoder layers to run, 1..8; <8 stops after the
                               encoder (only memory1/2 are produced) */
  float *hs1, *hs2;         /* [N][256]        decoder outputs           */
  float *memory1, *memory2; /* [N][L][256]     encoder outputs           */
  float *logits1, *logits2; /* [N][L]          heat-map logits           */
  float *cxy1, *cxy2;       /* [N][2]          soft-argmax centres (x,y) */
  float *tlbr1, *tlbr2;     /* [N][4]          sigmoid extents           */
} oetr_stage_outputs;

/* Text of the last error on this thread ("" if none). */
const char *oetr_last_error(void);

/* ABI version the library was built with. */
int oetr_abi_version(void);

/* Replaces: OETR.__init__ + load_state_dict for the hot-path modules
 * (reference src/model.py:58-84, dloc/core/overlaps/oetr.py:36-42).
 * Copies the weights to `device`, repacked into MFMA fragment order. */
oetr_status oetr_create(const oetr_weights *w, oetr_dtype dtype, int device,
                        oetr_handle *out);
void oetr_destroy(oetr_handle h);

/* Status word of a WORKSPACE: bits set (atomically, sticky) by the kernels of every forward
 * call that used this workspace since it was initialised / last cleared.  The word is the
 * first 4 bytes of the workspace (the first OETR_WORKSPACE_STATUS_BYTES are reserved for it,
 * at a shape-independent position), so callers that overlap batches on several streams - one
 * workspace per stream - see and clear only their own stream's calls.
 *   OETR_FLAG_F16_RANGE  a GEMM operand (activation) reached |x| >= 65520 in an
 *                        f16-based dtype (F32_SPLIT_F16, F32_SPLIT_QK16, F16): the operand
 *                        conversion rounds to nearest even, so 65520 is the first magnitude
 *                        that becomes f16 inf (65504 < |x| < 65520 still rounds to 65504, the
 *                        largest f16, and is handled exactly by the split).  The outputs of
 *                        that call are INVALID.  Re-run
 *                        with a handle created as OETR_DTYPE_F32 or OETR_DTYPE_BF16.
 *   OETR_FLAG_EXCHANGE   the four workgroups of an image's decoder chain (oetr_set_decoder_split)
 *                        did not all become resident within 2 ms - more forwards in flight on
 *                        the device than the automatic rule allows for, or a forced split on a
 *                        busy device.  The outputs of that call are INVALID, and so are those of
 *                        every call that used the workspace while the bit stood (their decoder
 *                        workgroups return at once without publishing).  On seeing it: clear it,
 *                        call oetr_workspace_init again (the per-image call counters and granules
 *                        of the failed call are not trustworthy) and re-run with
 *                        oetr_set_decoder_split(h, 1).
 * oetr_workspace_init   zeroes the status block (enqueued on `stream`); call it once after
 *                       allocating a workspace (or zero the first OETR_WORKSPACE_STATUS_BYTES
 *                       yourself).  Besides the status word the block holds the split decoder's
 *                       per-image call counters (words 16..31) and its exchange granules (from
 *                       byte 256: [16 images][5 exchanges][4 workgroups][256] x 8 bytes): they must
 *                       start from zero and are the library's from then on.
 * oetr_query_flags      copies the word to *flags (host), optionally clears it, and
 *                       SYNCHRONISES `stream` (the one call of this library that does):
 *                       ordered after every forward call enqueued on that stream before.
 * oetr_read_flags_async the same WITHOUT the synchronisation: the copy into `host_flags`
 *                       (pinned host memory for a truly asynchronous copy) and the optional
 *                       clear are enqueued on `stream`; read *host_flags after an event /
 *                       synchronisation of your own.  Enqueue-only, hipGraph-capturable:
 *                       the deferred form of the check (examine batch i's word when batch
 *                       i+1 is submitted).
 * Weights are range-checked by oetr_create (OETR_ERR_UNSUPPORTED).  Nothing like this exists
 * in the fp32 reference; it guards the reduced-range operand formats. */
#define OETR_FLAG_F16_RANGE 1u
#define OETR_FLAG_EXCHANGE 2u
#define OETR_FLAG_PUBLISHED 0x80000000u /* flag slots only (ABI 6): "this slot was written by the call" */
#define OETR_WORKSPACE_STATUS_BYTES (256 + 16 * 5 * 4 * 256 * 8)
oetr_status oetr_workspace_init(void *workspace, size_t workspace_bytes, void *stream);
oetr_status oetr_query_flags(oetr_handle h, void *workspace, void *stream,
                             uint32_t *flags, int clear);
oetr_status oetr_read_flags_async(oetr_handle h, void *workspace,
                                  uint32_t *host_flags, int clear, void *stream);
/* ABI 6 - the deferred check WITHOUT its two runtime dispatches.  oetr_read_flags_async costs one
 * 4-byte copy kernel and one 4-byte fill kernel per batch (rocprofv3: __amd_rocclr_copyBuffer 5.3 us +
 * __amd_rocclr_fillBufferAligned 4.2 us and two launch boundaries in a 343-us step).  The _flagslot
 * forms of the forward calls (below) take a FLAG SLOT instead: one uint32_t of host memory the device can
 * write (hipHostMalloc / hipHostRegister with the mapped attribute; torch's pinned memory is), passed by
 * its DEVICE address.  The last kernel of the call (k_heat_final; every kernel that can set a bit has
 * completed by then - same stream) exchanges the workspace's status word with 0 and stores the old
 * value, with OETR_FLAG_PUBLISHED set, into the slot at system scope.  The slot holds the word of THAT
 * call (and of earlier calls on the workspace that nobody read) once the call has completed: wait for an
 * event recorded behind it, or - cheaper, an event costs the stream a marker packet and ~6 us of idle
 * chip in front of the next kernel - zero the slot before the call and poll it for OETR_FLAG_PUBLISHED
 * (fall back to a stream synchronisation if the poll outlasts your patience).  Nothing else is
 * enqueued.  hipGraph-capturable like every forward call (the slot address is a kernel argument).
 * oetr_flagslot_device_pointer: hipHostGetDevicePointer for callers without HIP bindings - the address
 * the device uses for mapped host memory `host_slot` (OETR_ERR_BAD_ARG when it is not mapped). */
oetr_status oetr_flagslot_device_pointer(void *host_slot, uint32_t **device_slot);

/* Token rows per encoder workgroup: 0 = auto (default), 32 or 64.  64 exists in
 * the 16-bit-operand dtypes only (ignored for OETR_DTYPE_F32): fewer
 * CU-microseconds per token, half as many workgroups - the better shape once the
 * grid exceeds the chip (auto), or when several batches are in flight on
 * different streams.  Results do not depend on it beyond fp32 summation order
 * of the per-tile partial states.  Mutates the handle: call it outside
 * concurrent forward calls. */
oetr_status oetr_set_encoder_tile(oetr_handle h, int rows);

/* Reduce each image's per-tile partial linear-attention states (the all-to-all of
 * LinearAttention, reference src/models/linear_attention.py:45-46) ONCE instead of in every
 * consuming workgroup.  -1 (default): automatic - the reduction launch (1) from 768 source
 * tokens per image, off below (measured on MI355X: +0.9 % step time at 400 tokens per image,
 * -3.9 % at 1024, -4.2 % at 1600).  0: off.  1: in a small launch between the encoder launches.
 * (2, the in-launch last-arriver form of ABI <= 4, measured slower at every size and was removed:
 * OETR_ERR_BAD_ARG.)  Bit-identical results in every setting (same summation order).  Mutates the
 * handle like the other setters. */
oetr_status oetr_set_state_prereduce(oetr_handle h, int on);

/* How the forward path orders its tail (center_estimation, reference src/model.py:145-186; decoder
 * transformer.py:361-381).  1: "P form" - the decoder and the hs-independent products
 * P_tap = W_tap . memory of the 3x3 conv in ONE launch, then the att-weighted combine: the decoder's
 * GEMV chain hides behind the conv GEMMs (best for small batches).  2: "direct form" - decoder first,
 * then the conv of memory * att with the nine taps accumulated in registers, 64 token rows per
 * workgroup, no P buffer (best once the 9 x rows x 1 KB of P traffic outweigh the decoder chain;
 * two-plane dtypes only; token grids up to 40 wide keep the tile's halo resident in LDS as split planes and
 * run the nine taps as shifted windows of it, wider grids stage each tap's gathered tile - 3 forces that
 * second kernel at any width, for measurements).  0 (default): automatic - direct from 16 000 token rows (N (L1 + L2)) in
 * the two-plane dtypes, P form otherwise.  Same arithmetic per product; results agree to fp32
 * summation order (the nine taps are summed in the MFMA accumulator instead of in k_heat_combine).
 * Mutates the handle like the other setters. */
oetr_status oetr_set_tail_mode(oetr_handle h, int mode);

/* Workgroups per image of the single-query decoder chain (TransformerDecoder, reference
 * src/models/transformer.py:224-284: nine 256-wide GEMV stages per image, latency bound: one CU
 * streams an image's 3.9 MB of fp32 weights at its L1 fill rate).  4: four workgroups per image -
 * each streams a quarter of every stage's weights, the stages alternate between a split of the
 * outputs and a split of the inputs, and five 256-float all-reduces per image travel between the
 * four workgroups INSIDE the launch as tagged 8-byte granules in the workspace's status block
 * (no fence, no flag).  The four workgroups wait for each other, so they must be resident together:
 * automatic (0, default) picks 4 only while 2N x 4 workgroups take at most a quarter of the
 * device's CUs (N <= 8 on MI355X: up to four forwards in flight on four streams are then safe
 * whatever the dispatch order) and 1 otherwise; a wait that outlasts 2 ms sets
 * OETR_FLAG_EXCHANGE instead of hanging.  1: one workgroup per image (rounds 1-3).  Results
 * differ between 1 and 4 in fp32 summation order only.  Mutates the handle like the other setters. */
oetr_status oetr_set_decoder_split(oetr_handle h, int workgroups_per_image);

/* Tests only: the NEXT forward call that runs the four-workgroup decoder has workgroup 1 of image
 * 0 withhold its first exchange, so that its peers time out (2 ms) and OETR_FLAG_EXCHANGE is
 * raised on an otherwise idle device - the only way to exercise the caller's recovery path
 * (status-block reset, re-run on one workgroup per image) deterministically.  One shot: the
 * call after that is normal again.  `on` = 0 withdraws a pending fault. */
oetr_status oetr_debug_decoder_fault(oetr_handle h, int on);

/* Measurements only (ABI 6; bench.py): the dense f16 MFMA rate `device` SUSTAINS right now - every CU on
 * back-to-back v_mfma_f32_32x32x16_f16 with pseudo-random operands for about `seconds` (first half
 * settles the power controller, second half is counted), in TFLOP/s.  MI355X reaches its socket power
 * cap on this loop, so the figure is what a power-bound kernel could reach on this box at this moment,
 * as opposed to the nominal 2 500 TFLOP/s.  Allocates nothing; SYNCHRONISES `stream`. */
oetr_status oetr_debug_mfma_rate(int device, double seconds, double *tflops, void *stream);

/* Attention core of the eight encoder layers.  The reference builds
 * QueryTransformer(attention_mode='linear') (src/model.py:82-84; the config knob
 * OETR.NECK.ATTENTION is never read), LINEAR is therefore the default; FULL is
 * EncoderLayer(attention='full') (src/models/transformer.py:86-89): softmax(Q K^T /
 * sqrt(D)) V over all tokens of the source image (FullAttention,
 * src/models/linear_attention.py:53-87) - the all-pairs HW x HW correlation, computed
 * flash style on the f16 matrix pipe (fp32-class operand split), never materialised.
 * Same weights, same state dict.  FULL exists for the f16-based dtypes
 * (F32_SPLIT_F16, F16) and for F32 (exact fp32 MFMA products: the build an out-of-range
 * batch is re-run on), uses 32-token workgroups and a larger workspace (query
 * oetr_workspace_bytes after the call).  Mutates the handle like the other setters. */
typedef enum { OETR_ATTENTION_LINEAR = 0, OETR_ATTENTION_FULL = 1 } oetr_attention;
oetr_status oetr_set_attention(oetr_handle h, oetr_attention mode);

/* Bytes of workspace a forward call needs for this shape (256-B aligned
 * device buffer).  Returns 0 on invalid shape. */
size_t oetr_workspace_bytes(oetr_handle h, int n_pairs, int hf1, int wf1,
                            int hf2, int wf2);

/* Replaces: everything OETR.forward_dummy does after feature_extraction
 * (reference src/model.py:239-252): feature_correlation, center_estimation,
 * size_regression, box_tlbr_to_xyxy.
 *   feat1/feat2 [N][256][hf][wf] (NCHW, what input_proj2 emits)
 *   pos1/pos2   [256][hf][wf]    (batch-broadcast sine table window)
 *   img_h/img_w image sizes in pixels (the reference's mutable self.h1..w2)
 *   box1/box2   [N][4] xyxy pixels (out)                                   */
oetr_status oetr_forward(oetr_handle h, const float *feat1, const float *feat2,
                         const float *pos1, const float *pos2, int n_pairs,
                         int hf1, int wf1, int hf2, int wf2, int img_h1,
                         int img_w1, int img_h2, int img_w2, void *workspace,
                         size_t workspace_bytes, float *box1, float *box2,
                         void *stream);

/* The same call for a caller that produces the features on the device itself (the HIP
 * neck): the `flatten(2).permute(0, 2, 1)` of features and position tables that opens
 * QueryTransformer.forward (reference src/models/transformer.py:338-345) is then the
 * producer's store, not a launch of its own.
 *   oetr_token_buffers   where in `workspace` the hot path keeps its token-major inputs for
 *                        this shape: tokens1 [N*hf1*wf1][256], tokens2 [N*hf2*wf2][256]
 *                        (tokens2 == tokens1 + N*hf1*wf1*256: one 2N-image neck call fills
 *                        both), pos_tokens1/2 [hf*wf][256] (pos[c][l] transposed).  The
 *                        position tables survive forward calls; the tokens do not (the
 *                        encoder works in place).
 *   oetr_forward_tokens  oetr_forward on what those four buffers hold.                    */
oetr_status oetr_token_buffers(oetr_handle h, int n_pairs, int hf1, int wf1,
                               int hf2, int wf2, void *workspace,
                               size_t workspace_bytes, float **tokens1,
                               float **tokens2, float **pos_tokens1,
                               float **pos_tokens2);
oetr_status oetr_forward_tokens(oetr_handle h, int n_pairs, int hf1, int wf1,
                                int hf2, int wf2, int img_h1, int img_w1,
                                int img_h2, int img_w2, void *workspace,
                                size_t workspace_bytes, float *box1,
                                float *box2, void *stream);

/* oetr_forward (mask1 == mask2 == NULL) / oetr_forward_masked without stages, and oetr_forward_tokens,
 * publishing the workspace's status word into `flag_slot` (device address of mapped host memory, see
 * oetr_flagslot_device_pointer above) from their last kernel, which also clears the word.  Same
 * arguments, same results, same error behaviour otherwise; flag_slot must not be NULL. */
oetr_status oetr_forward_flagslot(oetr_handle h, const float *feat1, const float *feat2,
                                  const float *pos1, const float *pos2,
                                  const float *mask1, const float *mask2, int n_pairs,
                                  int hf1, int wf1, int hf2, int wf2, int img_h1,
                                  int img_w1, int img_h2, int img_w2, void *workspace,
                                  size_t workspace_bytes, float *box1, float *box2,
                                  uint32_t *flag_slot, void *stream);
oetr_status oetr_forward_tokens_flagslot(oetr_handle h, int n_pairs, int hf1, int wf1,
                                         int hf2, int wf2, int img_h1, int img_w1,
                                         int img_h2, int img_w2, void *workspace,
                                         size_t workspace_bytes, float *box1,
                                         float *box2, uint32_t *flag_slot, void *stream);

/* Same as oetr_forward, additionally exporting intermediates. box1/box2 may
 * be NULL when stages->enc_layers < 8. */
oetr_status oetr_forward_stages(oetr_handle h, const float *feat1,
                                const float *feat2, const float *pos1,
                                const float *pos2, int n_pairs, int hf1,
                                int wf1, int hf2, int wf2, int img_h1,
                                int img_w1, int img_h2, int img_w2,
                                void *workspace, size_t workspace_bytes,
                                float *box1, float *box2,
                                const oetr_stage_outputs *stages, void *stream);

/* forward_dummy's optional masks (reference src/model.py:229 `forward_dummy(image1, image2,
 * mask1=None, mask2=None)`, the training forward's resize_mask1/2, :256-258).  mask1 [N][hf1*wf1],
 * mask2 [N][hf2*wf2]: device floats at the token grid's resolution (the reference flattens
 * [N,hf,wf], src/models/transformer.py:340-343).  As in the reference a token's value
 *   - multiplies its phi(Q) row where the token is a query and its phi(K) and V rows where it is
 *     a source, in all eight encoder layers (x_mask / source_mask, transformer.py:349-358 ->
 *     LinearAttention q_mask / kv_mask, src/models/linear_attention.py:37-41; V is still divided
 *     by the FULL source length, :43-44) and in the decoder's cross-attention (memory_mask,
 *     transformer.py:361-381; its tgt_mask is None);
 *   - fills the token's heat-map logit with -1e9 before the softmax where it is 0
 *     (src/model.py:22, :166-171) - the `logits` stage output then holds the filled values.
 * Both masks or neither (both NULL = oetr_forward_stages); OETR_ERR_UNSUPPORTED unless the handle
 * is OETR_DTYPE_F32_SPLIT_F16, OETR_DTYPE_F32_SPLIT_QK16 or OETR_DTYPE_F32, with linear attention (the
 * reference's FullAttention turns a masked query's row into NaN, linear_attention.py:74-81).
 * `stages` may be NULL.  Same workspace, same enqueue-only behaviour as oetr_forward. */
oetr_status oetr_forward_masked(oetr_handle h, const float *feat1,
                                const float *feat2, const float *pos1,
                                const float *pos2, const float *mask1,
                                const float *mask2, int n_pairs, int hf1,
                                int wf1, int hf2, int wf2, int img_h1,
                                int img_w1, int img_h2, int img_w2,
                                void *workspace, size_t workspace_bytes,
                                float *box1, float *box2,
                                const oetr_stage_outputs *stages, void *stream);
/* ... and of the two seams: OETR.feature_correlation(feat1, feat2, pos1, pos2, mask1, mask2)
 * (src/model.py:132-143) and OETR.center_estimation(..., mask1, mask2) (:145-186; the masks
 * only fill the logits there, any dtype of handle). */
oetr_status oetr_feature_correlation_masked(
    oetr_handle h, const float *feat1, const float *feat2, const float *pos1,
    const float *pos2, const float *mask1, const float *mask2, int n_pairs,
    int hf1, int wf1, int hf2, int wf2, void *workspace, size_t workspace_bytes,
    float *hs1, float *hs2, float *memory1, float *memory2, void *stream);
oetr_status oetr_center_estimation_masked(
    oetr_handle h, const float *hs1, const float *hs2, const float *memory1,
    const float *memory2, const float *mask1, const float *mask2, int n_pairs,
    int hf1, int wf1, int hf2, int wf2, int img_h1, int img_h2, void *workspace,
    size_t workspace_bytes, float *cxy1, float *cxy2, void *stream);

/* Replaces: OETR.feature_correlation (reference src/model.py:132-143 ->
 * QueryTransformer.forward, src/models/transformer.py:313-383), masks None.
 * Out: hs1,hs2 [N][256]; memory1 [N][L1][256]; memory2 [N][L2][256]. */
oetr_status oetr_feature_correlation(oetr_handle h, const float *feat1,
                                     const float *feat2, const float *pos1,
                                     const float *pos2, int n_pairs, int hf1,
                                     int wf1, int hf2, int wf2, void *workspace,
                                     size_t workspace_bytes, float *hs1,
                                     float *hs2, float *memory1, float *memory2,
                                     void *stream);

/* Replaces: OETR.center_estimation (reference src/model.py:145-186), masks
 * None.  stride = img_h / hf (integer) scales both axes as in :176-181.
 * In: hs [N][256], memory [N][L][256].  Out: cxy1,cxy2 [N][2] = (x,y). */
oetr_status oetr_center_estimation(oetr_handle h, const float *hs1,
                                   const float *hs2, const float *memory1,
                                   const float *memory2, int n_pairs, int hf1,
                                   int wf1, int hf2, int wf2, int img_h1,
                                   int img_h2, void *workspace,
                                   size_t workspace_bytes, float *cxy1,
                                   float *cxy2, void *stream);

/* Replaces: OETR.size_regression (reference src/model.py:188-191).
 * In: hs [N][256].  Out: tlbr [N][4] (top,left,bottom,right fractions). */
oetr_status oetr_size_regression(oetr_handle h, const float *hs1,
                                 const float *hs2, int n_pairs, float *tlbr1,
                                 float *tlbr2, void *stream);

/* Replaces: box_tlbr_to_xyxy (reference src/models/utils.py:16-28).
 * In: cxy [n][2], tlbr [n][4].  Out: box [n][4] xyxy clamped to the image. */
oetr_status oetr_box_tlbr_to_xyxy(const float *cxy, const float *tlbr, int n,
                                  int max_h, int max_w, float *box,
                                  void *stream);

/* Replaces: LinearAttention.forward (reference
 * src/models/linear_attention.py:22-50), masks None.  Stand-alone entry for
 * parity tests of the attention core.  q [N][L][8][32], k,v [N][S][8][32],
 * out [N][L][8][32].  `workspace`: device buffer of
 * oetr_linear_attention_workspace_bytes(n) bytes (the n*8 KV states between the
 * entry's two launches); like every other call: no allocation, enqueue-only. */
size_t oetr_linear_attention_workspace_bytes(int n);
oetr_status oetr_linear_attention(const float *q, const float *k,
                                  const float *v, int n, int L, int S,
                                  float *out, void *workspace,
                                  size_t workspace_bytes, void *stream);

/* ... with LinearAttention.forward's q_mask [N][L] / kv_mask [N][S] (:37-41; device floats, either
 * may be NULL as in the reference): a token's value multiplies its phi(Q) row / its phi(K) and V rows. */
oetr_status oetr_linear_attention_masked(const float *q, const float *k,
                                         const float *v, const float *q_mask,
                                         const float *kv_mask, int n, int L,
                                         int S, float *out, void *workspace,
                                         size_t workspace_bytes, void *stream);

/* Replaces: FullAttention.forward (reference
 * src/models/linear_attention.py:53-87), no mask/dropout - the optional
 * "all-pairs QK^T volume" variant, never materialised in HBM.  Same shapes. */
oetr_status oetr_full_attention(const float *q, const float *k, const float *v,
                                int n, int L, int S, float *out, void *stream);
/* The same on the f16 matrix pipe with the fp32-class operand split of
 * OETR_DTYPE_F32_SPLIT_F16 (12 f16 MFMAs per 32x32 score tile instead of 32 f32 MFMAs;
 * inputs must be < 65504: `flags`, an optional DEVICE uint32 word, receives
 * OETR_FLAG_F16_RANGE otherwise - pass NULL to skip the check). */
oetr_status oetr_full_attention_split(const float *q, const float *k,
                                      const float *v, int n, int L, int S,
                                      float *out, uint32_t *flags, void *stream);

/* ---- neck: input_proj -> PatchMerging -> input_proj2 ------
 * The part of OETR.feature_extraction between the ResNet trunk and the hot
 * path (reference src/model.py:113-118; PatchMerging.forward,
 * src/models/backbone.py:53-67).  Separate handle: its 47.7 MB of weights are
 * independent of the hot-path handle.  GEMMs use the split-f16 arithmetic
 * (fp32-class) whatever dtype the hot-path handle was created with. */
typedef struct {
  uint32_t struct_size; /* = sizeof(oetr_neck_weights) */
  uint32_t abi_version; /* = OETR_ABI_VERSION */
  const float *input_proj_w, *input_proj_b;     /* [256][1024][1][1],[256] model.py:45-47 */
  const float *norm_w, *norm_b;                 /* [256] LayerNorm     backbone.py:37 */
  const float *reduction_w[3], *reduction_b[3]; /* [256|128|128][256][k][k], k = 4,8,16;
                                                   stride 2, pad (k-2)/2  backbone.py:39-51 */
  const float *input_proj2_w, *input_proj2_b;   /* [256][512][1][1],[256] model.py:48-50 */
} oetr_neck_weights;
typedef struct oetr_neck_ctx *oetr_neck_handle;

oetr_status oetr_neck_create(const oetr_neck_weights *w, int device,
                             oetr_neck_handle *out);
void oetr_neck_destroy(oetr_neck_handle h);
/* Workspace bytes for n_images backbone maps of hb x wb (0 on invalid shape:
 * hb, wb in [2, 400], (hb/2)*(wb/2) <= OETR_MAX_TOKENS). */
size_t oetr_neck_workspace_bytes(oetr_neck_handle h, int n_images, int hb,
                                 int wb);
/* Replaces: input_proj2(patchmerging(input_proj(x))) of
 * OETR.feature_extraction (reference src/model.py:113-118).
 *   backbone_feat [n_images][1024][hb][wb]   (NCHW, what ResnetEncoder emits)
 *   feat_out      [n_images][256][hb/2][wb/2] (NCHW, what oetr_forward takes) */
oetr_status oetr_neck_forward(oetr_neck_handle h, const float *backbone_feat,
                              int n_images, int hb, int wb, void *workspace,
                              size_t workspace_bytes, float *feat_out,
                              void *stream);
/* The same with a token-major result: tokens_out [n_images * (hb/2)*(wb/2)][256], i.e.
 * feat_out[n][c][l] stored at tokens_out[(n*L + l)*256 + c] - what oetr_token_buffers
 * hands out (same values as oetr_neck_forward, bit for bit). */
oetr_status oetr_neck_forward_tokens(oetr_neck_handle h,
                                     const float *backbone_feat, int n_images,
                                     int hb, int wb, void *workspace,
                                     size_t workspace_bytes, float *tokens_out,
                                     void *stream);
/* ABI 6: oetr_neck_forward_tokens whose range bit (OETR_FLAG_F16_RANGE) goes into a status word of the
 * CALLER'S choice instead of the neck workspace's own: `status_word` = device address of a 4-byte word,
 * normally the first word of the hot-path workspace the tokens are stored into - the forward call that
 * consumes them (oetr_forward_tokens_flagslot) then publishes ONE word for both stages and the neck needs
 * no read of its own.  The word is only ever OR-ed into. */
oetr_status oetr_neck_forward_tokens_status(oetr_neck_handle h,
                                            const float *backbone_feat, int n_images,
                                            int hb, int wb, void *workspace,
                                            size_t workspace_bytes, float *tokens_out,
                                            uint32_t *status_word, void *stream);
/* Output positions per workgroup of the fused PatchMerging conv kernel: 0 = auto
 * (default: the shape with the fewest workgroup rounds over the CUs for this problem
 * size), or 256 / 192 / 128.  Results are identical in every shape (same summation
 * order per output).  Mutates the handle. */
oetr_status oetr_neck_set_conv_rows(oetr_neck_handle h, int rows);
/* Which PatchMerging conv kernel runs: 0 = auto (default: kind 3 when the output map is
 * at least 16 wide, else the gather kernel), 1 = gather (one input pixel per output
 * position per kernel pixel), 2 = row window (each input row segment staged once per
 * kernel row and x parity; needs wo >= 16, else OETR_ERR_BAD_SHAPE at forward), 3 = row
 * window with one wave per SIMD (4-wave workgroups, accumulators in AGPRs: each weight
 * fragment fetched once per workgroup; bit-identical to kind 2).  Gather and row window
 * sum the same products in a different order (fp32 rounding-level differences).
 * Mutates the handle. */
oetr_status oetr_neck_set_conv_kernel(oetr_neck_handle h, int kind);
/* Status word of a NECK workspace (same layout and rules as oetr_query_flags /
 * oetr_read_flags_async; initialise with oetr_workspace_init): OETR_FLAG_F16_RANGE when a
 * backbone feature / intermediate reached the f16 range of its split GEMMs. */
oetr_status oetr_neck_query_flags(oetr_neck_handle h, void *workspace, void *stream,
                                  uint32_t *flags, int clear);
oetr_status oetr_neck_read_flags_async(oetr_neck_handle h, void *workspace,
                                       uint32_t *host_flags, int clear, void *stream);

/* ---- box -> crop step on the device ---------------------------
 * Replaces, for one image pair, the overlap branch of Matching.forward (reference
 * evaluation.py:82-170: boxes x overlap_scales, int truncation, the size gate) and
 * tensor_overlap_crop / patch_resize (dloc/core/utils/utils.py:476-564: integer crop
 * rectangle, target size from the larger-area image, cv2.resize INTER_CUBIC of the
 * crop x255, optional second resize to a multiple of size_divisor, /255).  The
 * reference reads the boxes back to the host and resizes with OpenCV on the CPU;
 * here the boxes stay on the GPU and the step is enqueue-only (no D2H copy, no
 * synchronisation): sizes are data dependent, so the caller provides buffers of
 * oetr_overlap_crop_capacity() floats and reads the geometry from `info` (device
 * memory) whenever it needs it.
 * Integer / ratio outputs are pinned to the reference (tests/golden/crop.npz); the
 * resize follows OpenCV's published float32 bicubic (a = -0.75) but is
 * parity-UNPINNED: cv2 is not installed where the fixtures were generated. */
typedef struct {
  int32_t valid;        /* 1: crops made; 0: gate failed, the full images passed through
                           bit for bit (evaluation.py:142-170); -1: a crop is degenerate
                           or exceeds the capacity (sizes zeroed, nothing written) - where
                           the reference would raise from cv2.resize                 */
  int32_t box[2][4];    /* scaled boxes truncated to int, xyxy (utils.py:515-516)      */
  int32_t crop_w[2], crop_h[2]; /* slice sizes (Python slicing clamps at the border) */
  int32_t new_w[2], new_h[2];   /* patch_resize output (utils.py:476-493)            */
  int32_t out_w[2], out_h[2];   /* after rounding up to size_divisor (:547-556)      */
  double ratio[2][2];   /* ratio1 / ratio2 = [[rx, ry]] as Python floats             */
  float sbox[2][4];     /* bbox * overlap_scales: the 'bbox0' / 'bbox1' the reference
                           returns (full-image box when valid == 0)                  */
} oetr_crop_info;

/* Floats per output buffer ([channels][cap_h][cap_w], cap = the larger image's size
 * rounded up to size_divisor).  0 on invalid arguments. */
size_t oetr_overlap_crop_capacity(int channels, int h1, int w1, int h2, int w2,
                                  int size_divisor, int *cap_h, int *cap_w);

/*   image1/image2 [channels][h][w]   the matcher's images (data['image0'/'image1'][0])
 *   box1/box2     device [4]         entry 0 of the OETR boxes, OETR input frame
 *   scale1/scale2 host (sx, sy)      overlap_scales0/1 (read_overlap_image, utils.py:313-318)
 *   keep_aspect   extractor_name != 'disk';  size_divisor  8 for LoFTR, else 1
 *   gate_mode     0: crops whenever every box side > 1 px; 1: 'pragueparks-val' rule
 *                 (additionally an integer size ratio > 2 between the two boxes)
 *   tmp           2 x capacity floats (device scratch between the two resize passes)
 *   out1/out2     capacity floats each; hold [channels][out_h][out_w] densely packed
 *   info          device, written by the first launch                                */
oetr_status oetr_overlap_crop(const float *image1, const float *image2, int channels,
                              int h1, int w1, int h2, int w2, const float *box1,
                              const float *box2, const float scale1[2],
                              const float scale2[2], int keep_aspect, int size_divisor,
                              int gate_mode, float *tmp, float *out1, float *out2,
                              size_t capacity_floats, oetr_crop_info *info, void *stream);

/* ---- image reader of the overlap pipeline on the device ------------
 * Replaces read_overlap_image (reference dloc/core/utils/utils.py:271-343) after the
 * decode: the matcher's frame (sizes rounded up to a multiple of 32 for 'disk', 8 for
 * 'loftr', process_resize :248-265), the OETR input frame (resize x resize, or the native
 * size for resize == -1), `scales` / `overlap_scales` as the reference's Python floats, the two
 * chained cv2.resize (INTER_LINEAR) calls, the grey conversion and the /255.
 *   oetr_overlap_frame       host arithmetic only (usable without a GPU).
 *   oetr_read_overlap_image  enqueue-only: image_bgr DEVICE [h][w][3] (uint8 when is_u8, else
 *                            float32 0..255; cv2.imread's BGR order) -> overlap_out
 *                            [h_ov][w_ov][3] in [0,1] (one slot of the [N,H,W,3] batch
 *                            oetr's forward_dummy seam takes) and inp_out [1|3][h_new][w_new]
 *                            in [0,1] (grey or planar BGR: the matcher's image, what
 *                            oetr_overlap_crop crops); tmp: h_new*w_new*3 floats of scratch.
 *                            swap_rb = 1 reverses the channel order first, as the reference
 *                            does when `align` is empty (utils.py:283-284; its grey
 *                            conversion still reads the channels as B,G,R afterwards).
 * Sizes and scales are pinned to the reference (tests/golden/reader.npz); the resize follows
 * OpenCV's published float32 INTER_LINEAR but is parity-UNPINNED (no cv2 in the build image). */
typedef enum { OETR_ALIGN_NONE = 0, OETR_ALIGN_DISK = 1, OETR_ALIGN_LOFTR = 2 } oetr_align;
oetr_status oetr_overlap_frame(int w, int h, int resize, oetr_align align, int *w_new,
                               int *h_new, int *w_ov, int *h_ov, double scales[2],
                               double overlap_scales[2]);
oetr_status oetr_read_overlap_image(const void *image_bgr, int is_u8, int h, int w,
                                    int h_new, int w_new, int h_ov, int w_ov, int grayscale,
                                    int swap_rb, float *tmp, float *overlap_out,
                                    float *inp_out, void *stream);

/* ---- measurement hook (bench.py / profiling only) -------------------------
 * A trace owns a pool of HIP events.  While attached to a handle, every
 * kernel launched by the forward entry points is bracketed by two events
 * recorded on the SAME stream the kernel is launched on (so the durations are
 * GPU-side, independent of torch's current stream).  Attaching mutates the
 * handle: do it from one thread, outside concurrent forward calls.
 * oetr_trace_summary synchronises on the recorded events and returns, per
 * kernel id, its name, number of launches and summed duration; it then resets
 * the pool.  Nothing like this exists in the reference. */
typedef struct oetr_trace *oetr_trace_handle;
#define OETR_TRACE_MAX_KERNELS 16
oetr_status oetr_trace_create(int max_launches, oetr_trace_handle *out);
void oetr_trace_destroy(oetr_trace_handle t);
oetr_status oetr_set_trace(oetr_handle h, oetr_trace_handle t /* NULL detaches */);
oetr_status oetr_neck_set_trace(oetr_neck_handle h, oetr_trace_handle t);
oetr_status oetr_trace_summary(oetr_trace_handle t, int *n_kernels,
                               const char *names[OETR_TRACE_MAX_KERNELS],
                               int launches[OETR_TRACE_MAX_KERNELS],
                               float total_ms[OETR_TRACE_MAX_KERNELS]);

#ifdef __cplusplus
}
#endif
#endif /* OETR_HIP_H_ */
