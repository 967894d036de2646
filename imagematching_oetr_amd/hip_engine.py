"""ctypes binding of ``liboetr_hip.so`` (C ABI: ``include/oetr_hip.h``).

PyTorch is used here for device memory and the current HIP stream only; all
arithmetic of the hot path happens inside the library.  Loading fails loudly:
there is no fallback implementation.
"""
import ctypes as C
import os
import time
from pathlib import Path

import torch

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / 'csrc' / 'liboetr_hip.so'

N_ENC, N_DEC, D_MODEL, N_HEAD = 8, 2, 256, 8
_f32p = C.POINTER(C.c_float)


class _EncW(C.Structure):
    _fields_ = [(n, _f32p) for n in (
        'q_proj', 'k_proj', 'v_proj', 'merge', 'mlp0', 'mlp2',
        'pre_norm_q_w', 'pre_norm_q_b', 'pre_norm_kv_w', 'pre_norm_kv_b',
        'norm2_w', 'norm2_b')]


class _MhaW(C.Structure):
    _fields_ = [(n, _f32p) for n in (
        'q_proj_w', 'q_proj_b', 'k_proj_w', 'k_proj_b', 'v_proj_w',
        'v_proj_b', 'merge')]


class _DecW(C.Structure):
    _fields_ = [('self_attn', _MhaW), ('multihead_attn', _MhaW)] + [
        (n, _f32p) for n in ('mlp0', 'mlp2', 'norm1_w', 'norm1_b', 'norm2_w',
                             'norm2_b', 'norm3_w', 'norm3_b')]


class _Weights(C.Structure):
    _fields_ = [('struct_size', C.c_uint32), ('abi_version', C.c_uint32),
                ('enc', _EncW * N_ENC), ('dec', _DecW * N_DEC)] + [
        (n, _f32p) for n in ('query_embed1', 'query_embed2', 'tlbr0_w',
                             'tlbr2_w', 'tlbr2_b', 'heat_conv_w',
                             'heat_conv_b', 'heat_gn_w', 'heat_gn_b',
                             'heat_out_w', 'heat_out_b')]


class _Stages(C.Structure):
    _fields_ = [('struct_size', C.c_uint32), ('enc_layers', C.c_int32)] + [
        (n, C.c_void_p) for n in ('hs1', 'hs2', 'memory1', 'memory2',
                                  'logits1', 'logits2', 'cxy1', 'cxy2',
                                  'tlbr1', 'tlbr2')]


class _NeckWeights(C.Structure):
    _fields_ = [('struct_size', C.c_uint32), ('abi_version', C.c_uint32)] + [
        (n, _f32p) for n in ('input_proj_w', 'input_proj_b', 'norm_w', 'norm_b')] + [
        ('reduction_w', _f32p * 3), ('reduction_b', _f32p * 3),
        ('input_proj2_w', _f32p), ('input_proj2_b', _f32p)]


class _CropInfo(C.Structure):
    _fields_ = [('valid', C.c_int32), ('box', (C.c_int32 * 4) * 2),
                ('crop_w', C.c_int32 * 2), ('crop_h', C.c_int32 * 2),
                ('new_w', C.c_int32 * 2), ('new_h', C.c_int32 * 2),
                ('out_w', C.c_int32 * 2), ('out_h', C.c_int32 * 2),
                ('ratio', (C.c_double * 2) * 2), ('sbox', (C.c_float * 4) * 2)]


ABI_VERSION = 6
# OETR_WORKSPACE_STATUS_BYTES: the block that opens a workspace - status word, the split decoder's
# call counters and exchange granules; zero it once (oetr_workspace_init), the library owns it after
WORKSPACE_STATUS_BYTES = 256 + 16 * 5 * 4 * 256 * 8
EXPORTS = (
    'oetr_last_error', 'oetr_abi_version', 'oetr_create', 'oetr_destroy',
    'oetr_workspace_bytes', 'oetr_forward', 'oetr_forward_stages',
    'oetr_feature_correlation', 'oetr_center_estimation',
    'oetr_size_regression', 'oetr_box_tlbr_to_xyxy', 'oetr_linear_attention',
    'oetr_full_attention', 'oetr_trace_create', 'oetr_trace_destroy',
    'oetr_set_trace', 'oetr_trace_summary', 'oetr_neck_create',
    'oetr_neck_destroy', 'oetr_neck_workspace_bytes', 'oetr_neck_forward',
    'oetr_neck_set_trace', 'oetr_set_encoder_tile', 'oetr_query_flags',
    'oetr_neck_query_flags', 'oetr_overlap_crop', 'oetr_overlap_crop_capacity',
    'oetr_full_attention_split', 'oetr_set_attention', 'oetr_neck_set_conv_rows',
    'oetr_linear_attention_workspace_bytes', 'oetr_neck_set_conv_kernel',
    'oetr_token_buffers', 'oetr_forward_tokens', 'oetr_neck_forward_tokens',
    'oetr_workspace_init', 'oetr_read_flags_async', 'oetr_neck_read_flags_async',
    'oetr_set_state_prereduce', 'oetr_set_tail_mode', 'oetr_set_decoder_split', 'oetr_overlap_frame', 'oetr_read_overlap_image',
    'oetr_forward_masked', 'oetr_feature_correlation_masked', 'oetr_center_estimation_masked',
    'oetr_linear_attention_masked', 'oetr_debug_decoder_fault',
    'oetr_flagslot_device_pointer', 'oetr_forward_flagslot', 'oetr_forward_tokens_flagslot',
    'oetr_neck_forward_tokens_status', 'oetr_debug_mfma_rate')

FLAG_F16_RANGE = 1   # OETR_FLAG_F16_RANGE
FLAG_EXCHANGE = 2    # OETR_FLAG_EXCHANGE: the split decoder's workgroups were not resident together
FLAG_INVALID = FLAG_F16_RANGE | FLAG_EXCHANGE   # either bit: that call's outputs are invalid
FLAG_PUBLISHED = 0x80000000   # OETR_FLAG_PUBLISHED: set in every word a *_flagslot call stores into its slot


def hot_path_keys():
    """State-dict keys (reference checkpoint names) the library consumes."""
    keys = []
    for i in range(N_ENC):
        p = f'transformer.encoder.{i}.'
        keys += [p + s for s in (
            'q_proj.weight', 'k_proj.weight', 'v_proj.weight', 'merge.weight',
            'mlp.0.weight', 'mlp.2.weight', 'pre_norm_q.weight',
            'pre_norm_q.bias', 'pre_norm_kv.weight', 'pre_norm_kv.bias',
            'norm2.weight', 'norm2.bias')]
    for i in range(N_DEC):
        p = f'transformer.decoder.layers.{i}.'
        for a in ('self_attn', 'multihead_attn'):
            keys += [p + f'{a}.{s}' for s in (
                'q_proj.weight', 'q_proj.bias', 'k_proj.weight', 'k_proj.bias',
                'v_proj.weight', 'v_proj.bias', 'merge.weight')]
        keys += [p + s for s in (
            'mlp.0.weight', 'mlp.2.weight', 'norm1.weight', 'norm1.bias',
            'norm2.weight', 'norm2.bias', 'norm3.weight', 'norm3.bias')]
    keys += ['query_embed1.weight', 'query_embed2.weight', 'tlbr_reg.0.weight',
             'tlbr_reg.2.weight', 'tlbr_reg.2.bias', 'heatmap_conv.0.weight',
             'heatmap_conv.0.bias', 'heatmap_conv.1.weight',
             'heatmap_conv.1.bias', 'heatmap_conv.3.weight',
             'heatmap_conv.3.bias']
    return keys


def neck_keys():
    """State-dict keys of the neck (input_proj, PatchMerging, input_proj2)."""
    keys = ['input_proj.weight', 'input_proj.bias', 'patchmerging.norm.weight',
            'patchmerging.norm.bias']
    for i in range(3):
        keys += [f'patchmerging.reductions.{i}.weight',
                 f'patchmerging.reductions.{i}.bias']
    return keys + ['input_proj2.weight', 'input_proj2.bias']


_lib = None


def load_library(path=None):
    """dlopen the HIP library and declare signatures.  Raises if missing.

    torch is imported first on purpose: its bundled libamdhip64.so.7 is then
    the HIP runtime the library binds to, so device pointers and streams are
    shared with torch."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path or os.environ.get('OETR_HIP_LIB', LIB_PATH))
    if not p.exists():
        raise RuntimeError(
            f'{p} not found: build it with `python -c "import __graft_entry__ '
            f'as g; g.build()"` or `make -C {_PKG_DIR / "csrc"}`. The OETR hot '
            'path has no non-HIP fallback.')
    lib = C.CDLL(str(p))
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.oetr_last_error.restype = C.c_char_p
    lib.oetr_last_error.argtypes = []
    lib.oetr_abi_version.restype = i
    lib.oetr_abi_version.argtypes = []
    lib.oetr_create.restype = i
    lib.oetr_create.argtypes = [C.POINTER(_Weights), i, i, C.POINTER(vp)]
    lib.oetr_destroy.restype = None
    lib.oetr_destroy.argtypes = [vp]
    lib.oetr_workspace_bytes.restype = sz
    lib.oetr_workspace_bytes.argtypes = [vp, i, i, i, i, i]
    fwd = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, vp, sz, vp, vp]
    lib.oetr_forward.restype = i
    lib.oetr_forward.argtypes = fwd + [vp]
    lib.oetr_forward_stages.restype = i
    lib.oetr_forward_stages.argtypes = fwd + [C.POINTER(_Stages), vp]
    lib.oetr_feature_correlation.restype = i
    lib.oetr_feature_correlation.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i,
                                             vp, sz, vp, vp, vp, vp, vp]
    lib.oetr_center_estimation.restype = i
    lib.oetr_center_estimation.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i,
                                           i, i, vp, sz, vp, vp, vp]
    # forward_dummy's masks: the same calls with (mask1, mask2) after the position tables / memories
    lib.oetr_forward_masked.restype = i
    lib.oetr_forward_masked.argtypes = fwd[:5] + [vp, vp] + fwd[5:] + [C.POINTER(_Stages), vp]
    lib.oetr_feature_correlation_masked.restype = i
    lib.oetr_feature_correlation_masked.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i,
                                                    vp, sz, vp, vp, vp, vp, vp]
    lib.oetr_center_estimation_masked.restype = i
    lib.oetr_center_estimation_masked.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i,
                                                  i, i, vp, sz, vp, vp, vp]
    lib.oetr_size_regression.restype = i
    lib.oetr_size_regression.argtypes = [vp, vp, vp, i, vp, vp, vp]
    lib.oetr_box_tlbr_to_xyxy.restype = i
    lib.oetr_box_tlbr_to_xyxy.argtypes = [vp, vp, i, i, i, vp, vp]
    lib.oetr_full_attention.restype = i
    lib.oetr_full_attention.argtypes = [vp, vp, vp, i, i, i, vp, vp]
    lib.oetr_linear_attention.restype = i
    lib.oetr_linear_attention.argtypes = [vp, vp, vp, i, i, i, vp, vp, sz, vp]
    lib.oetr_linear_attention_workspace_bytes.restype = sz
    lib.oetr_linear_attention_workspace_bytes.argtypes = [i]
    lib.oetr_linear_attention_masked.restype = i
    lib.oetr_linear_attention_masked.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp, vp, sz, vp]
    lib.oetr_full_attention_split.restype = i
    lib.oetr_full_attention_split.argtypes = [vp, vp, vp, i, i, i, vp, vp, vp]
    lib.oetr_trace_create.restype = i
    lib.oetr_trace_create.argtypes = [i, C.POINTER(vp)]
    lib.oetr_trace_destroy.restype = None
    lib.oetr_trace_destroy.argtypes = [vp]
    lib.oetr_set_trace.restype = i
    lib.oetr_set_trace.argtypes = [vp, vp]
    lib.oetr_trace_summary.restype = i
    lib.oetr_trace_summary.argtypes = [vp, C.POINTER(i), C.POINTER(C.c_char_p),
                                       C.POINTER(i), C.POINTER(C.c_float)]
    lib.oetr_set_encoder_tile.restype = i
    lib.oetr_set_encoder_tile.argtypes = [vp, i]
    lib.oetr_set_attention.restype = i
    lib.oetr_set_attention.argtypes = [vp, i]
    lib.oetr_set_state_prereduce.restype = i
    lib.oetr_set_state_prereduce.argtypes = [vp, i]
    lib.oetr_set_tail_mode.restype = i
    lib.oetr_set_tail_mode.argtypes = [vp, i]
    lib.oetr_set_decoder_split.restype = i
    lib.oetr_set_decoder_split.argtypes = [vp, i]
    lib.oetr_debug_decoder_fault.restype = i
    lib.oetr_debug_decoder_fault.argtypes = [vp, i]
    lib.oetr_neck_create.restype = i
    lib.oetr_neck_create.argtypes = [C.POINTER(_NeckWeights), i, C.POINTER(vp)]
    lib.oetr_neck_destroy.restype = None
    lib.oetr_neck_destroy.argtypes = [vp]
    lib.oetr_neck_workspace_bytes.restype = sz
    lib.oetr_neck_workspace_bytes.argtypes = [vp, i, i, i]
    lib.oetr_neck_forward.restype = i
    lib.oetr_neck_forward.argtypes = [vp, vp, i, i, i, vp, sz, vp, vp]
    lib.oetr_neck_forward_tokens.restype = i
    lib.oetr_neck_forward_tokens.argtypes = [vp, vp, i, i, i, vp, sz, vp, vp]
    lib.oetr_token_buffers.restype = i
    lib.oetr_token_buffers.argtypes = [vp, i, i, i, i, i, vp, sz] + [C.POINTER(vp)] * 4
    lib.oetr_forward_tokens.restype = i
    lib.oetr_forward_tokens.argtypes = [vp, i, i, i, i, i, i, i, i, i, vp, sz, vp, vp, vp]
    lib.oetr_neck_set_trace.restype = i
    lib.oetr_neck_set_trace.argtypes = [vp, vp]
    lib.oetr_neck_set_conv_rows.restype = i
    lib.oetr_neck_set_conv_rows.argtypes = [vp, i]
    lib.oetr_neck_set_conv_kernel.restype = i
    lib.oetr_neck_set_conv_kernel.argtypes = [vp, i]
    for name in ('oetr_query_flags', 'oetr_neck_query_flags'):
        fn = getattr(lib, name)
        fn.restype = i
        fn.argtypes = [vp, vp, vp, C.POINTER(C.c_uint32), i]      # handle, workspace, stream, out, clear
    for name in ('oetr_read_flags_async', 'oetr_neck_read_flags_async'):
        fn = getattr(lib, name)
        fn.restype = i
        fn.argtypes = [vp, vp, vp, i, vp]                         # handle, workspace, host word, clear, stream
    lib.oetr_workspace_init.restype = i
    lib.oetr_workspace_init.argtypes = [vp, sz, vp]
    # ABI 6: the status word published by the call's last kernel (no runtime dispatch behind it)
    lib.oetr_flagslot_device_pointer.restype = i
    lib.oetr_flagslot_device_pointer.argtypes = [vp, C.POINTER(vp)]
    lib.oetr_forward_flagslot.restype = i
    lib.oetr_forward_flagslot.argtypes = fwd[:5] + [vp, vp] + fwd[5:] + [vp, vp]       # ..., flag_slot, stream
    lib.oetr_forward_tokens_flagslot.restype = i
    lib.oetr_forward_tokens_flagslot.argtypes = [vp, i, i, i, i, i, i, i, i, i, vp, sz, vp, vp, vp, vp]
    lib.oetr_debug_mfma_rate.restype = i
    lib.oetr_debug_mfma_rate.argtypes = [i, C.c_double, C.POINTER(C.c_double), vp]
    lib.oetr_neck_forward_tokens_status.restype = i
    lib.oetr_neck_forward_tokens_status.argtypes = [vp, vp, i, i, i, vp, sz, vp, vp, vp]
    lib.oetr_overlap_crop_capacity.restype = sz
    lib.oetr_overlap_crop_capacity.argtypes = [i, i, i, i, i, i, C.POINTER(i), C.POINTER(i)]
    lib.oetr_overlap_crop.restype = i
    lib.oetr_overlap_crop.argtypes = [vp, vp, i, i, i, i, i, vp, vp, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), i, i, i, vp, vp, vp, sz, vp, vp]
    lib.oetr_overlap_frame.restype = i
    lib.oetr_overlap_frame.argtypes = [i, i, i, i] + [C.POINTER(i)] * 4 + [C.POINTER(C.c_double)] * 2
    lib.oetr_read_overlap_image.restype = i
    lib.oetr_read_overlap_image.argtypes = [vp, i, i, i, i, i, i, i, i, i, vp, vp, vp, vp]
    if lib.oetr_abi_version() != ABI_VERSION:
        raise RuntimeError(f'{p}: ABI version {lib.oetr_abi_version()} != '
                           f'{ABI_VERSION}')
    if path is None:
        _lib = lib
    return lib


class OetrError(RuntimeError):
    pass


class OetrRangeError(OetrError):
    """A GEMM operand left the f16 range of an f16-based precision
    (``OETR_FLAG_F16_RANGE``): the outputs of that call are invalid."""


class OetrExchangeError(OetrError):
    """The four workgroups of an image's split decoder chain were not resident together within
    its time limit (``OETR_FLAG_EXCHANGE``): the outputs of that call - and of every call that used
    the workspace while the bit stood - are invalid.  Nothing is wrong with the inputs: the status
    block has been re-initialised and the engine switched to one workgroup per image
    (``set_decoder_split(1)``), so the same call can simply be submitted again."""


def _check(lib, status, what):
    if status != 0:
        msg = lib.oetr_last_error().decode(errors='replace')
        exc = ValueError if status in (1, 2) else OetrError
        raise exc(f'{what} failed (status {status}): {msg}')


def _stream(device=None):
    """torch's current HIP stream ON ``device`` (not on the current device)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _query_flags(lib, fn, handle, ws, device, clear):
    """Synchronising read of a workspace's status word (0 when the stream has no workspace yet)."""
    if ws is None:
        return 0
    flags = C.c_uint32(0)
    with torch.cuda.device(device):
        _check(lib, fn(handle, ws.data_ptr(), _stream(device), C.byref(flags), int(bool(clear))), fn.__name__)
    return int(flags.value)


class FlagTicket:
    """A status word on its way to the host, nothing has synchronised.  Two kinds:

    * published (``oetr_forward*_flagslot``, ABI 6): the forward call's last kernel stores the word,
      with ``FLAG_PUBLISHED`` set, into a mapped pinned host word that was zeroed before the call.
      No dispatch and NO EVENT behind the batch (an event record is a marker packet on the stream:
      measured 6 us of idle chip in front of the next batch's first kernel): ``value()`` polls the
      word - by the time the next batch is submitted it has long arrived - and falls back to
      synchronising the stream the call was enqueued on if it has not after ``POLL_S``.
    * copied (``oetr_read_flags_async``): a 4-byte copy was ENQUEUED behind the calls it reports
      on; ``value()`` waits for an event recorded right behind that copy."""

    POLL_S = 0.05

    def __init__(self, slot, event, owner=None, index=None, stream=None, published=False):
        self._slot, self._event = slot, event
        self._owner, self._index = owner, index     # graph words: handed back by release()
        self._stream, self._published = stream, published

    def ready(self):
        if self._published and self._stream is not None:
            return bool(int(self._slot.item()) & FLAG_PUBLISHED)
        return self._event is None or self._event.query()

    def value(self):
        if self._event is not None:
            self._event.synchronize()
        if not self._published:
            return int(self._slot.item())
        word = int(self._slot.item()) & 0xffffffff
        if not word & FLAG_PUBLISHED and self._stream is not None:     # eager call: poll, then synchronise
            deadline, spins = time.perf_counter() + self.POLL_S, 0
            while not word & FLAG_PUBLISHED and time.perf_counter() < deadline:
                spins += 1
                if not spins & 63:
                    time.sleep(0)      # (a long wait means the device is the bottleneck: let other Python threads run)
                word = int(self._slot.item()) & 0xffffffff
            if not word & FLAG_PUBLISHED:
                self._stream.synchronize()
                word = int(self._slot.item()) & 0xffffffff
        if not word & FLAG_PUBLISHED:
            if self._stream is None:
                return 0      # captured into a graph that has not been replayed yet: nothing ran
            raise OetrError('the status word of a forward call never reached its flag slot (the call has '
                            'completed): the slot is not device-visible host memory')
        return word & ~FLAG_PUBLISHED

    def release(self):
        """Hand a word captured into a HIP graph back to its reader (the graph was discarded:
        nothing will write the word again).  No-op for eager tickets."""
        if self._owner is not None:
            self._owner._release(self._index)
            self._owner = None


class _FlagReader:
    """Pinned host words for asynchronous status reads, one set per engine: a ring for eager
    calls (each word is consumed before the ring comes round: the deferred check settles
    batch i when batch i+2 is submitted - at most two batches, i.e. four words, are pending) and words handed out to calls captured into a HIP
    graph (every replay rewrites them) until the ticket is released.  Pinning host memory is
    not allowed while a stream is capturing, so graph words come in pages allocated OUTSIDE
    capture: one page up front, another whenever an eager call finds the free list short
    (``reserve_graph_words``) - an engine that re-captures per shape never runs dry as long
    as it releases the tickets of discarded graphs or makes an eager call in between."""

    SLOTS, GRAPH_PAGE = 16, 16

    def __init__(self):
        self._words = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
        self._next = 0
        self._pages = []          # pinned pages of graph words (kept alive here)
        self._free = []           # (page, index) pairs not handed out
        self._dev_base = {}       # pinned block (host address) -> its device address
        self.reserve_graph_words(self.GRAPH_PAGE)

    def reserve_graph_words(self, n):
        """Make sure at least ``n`` graph words are free (allocates pinned pages; not while
        a stream is capturing)."""
        while len(self._free) < n:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise OetrError('graph status words must be reserved outside stream capture')
            page = torch.zeros(self.GRAPH_PAGE, dtype=torch.int32).pin_memory()
            self._pages.append(page)
            self._free += [(len(self._pages) - 1, i) for i in range(self.GRAPH_PAGE)]

    def _release(self, key):
        self._free.append(key)

    def _device_address(self, lib, slot):
        """Device address of a pinned word (``oetr_flagslot_device_pointer``; resolved once per
        pinned block, raises if torch's pinned memory is not mapped into the device)."""
        base = slot._base if slot._base is not None else slot
        key = base.data_ptr()
        dev = self._dev_base.get(key)
        if dev is None:
            out = C.c_void_p()
            _check(lib, lib.oetr_flagslot_device_pointer(key, C.byref(out)), 'oetr_flagslot_device_pointer')
            dev = self._dev_base[key] = out.value
        return dev + (slot.data_ptr() - key)

    def publish_slot(self, lib):
        """Reserve a word for a ``*_flagslot`` forward call: -> (slot, device address, owner, key).
        Hand all four to :meth:`published` right after the call was enqueued."""
        capturing = torch.cuda.is_current_stream_capturing()
        owner = key = None
        if capturing:
            if not self._free:
                raise OetrError('no free status word for a forward call captured into a HIP graph: release '
                                'the tickets of discarded graphs (OETR.hip_graph_release) or reserve more '
                                'outside capture (reserve_graph_words)')
            key = self._free.pop(0)
            slot = self._pages[key[0]][key[1]:key[1] + 1]
            owner = self
        else:
            if len(self._free) < self.GRAPH_PAGE // 2:
                self.reserve_graph_words(self.GRAPH_PAGE)      # top up while pinning is allowed
            i = self._next
            self._next = (self._next + 1) % self.SLOTS
            slot = self._words[i:i + 1]
        if not capturing:
            slot.zero_()      # (host store to pinned memory, before the call is enqueued: the ticket polls for FLAG_PUBLISHED)
        return slot, self._device_address(lib, slot), owner, key

    def published(self, slot, owner, key, device):
        """The ticket of a ``*_flagslot`` call just enqueued on the current stream of ``device``."""
        if torch.cuda.is_current_stream_capturing():
            # valid once a replay has been synchronised (every replay rewrites the word)
            return FlagTicket(slot, None, owner, key, published=True)
        return FlagTicket(slot, None, stream=torch.cuda.current_stream(device), published=True)

    def read(self, lib, fn, handle, ws, device, clear):
        capturing = torch.cuda.is_current_stream_capturing()
        owner = key = None
        if capturing:
            if not self._free:
                raise OetrError('no free status word for a read captured into a HIP graph: release the '
                                'tickets of discarded graphs (OETR.hip_graph_release) or reserve more '
                                'outside capture (reserve_graph_words)')
            key = self._free.pop(0)
            slot = self._pages[key[0]][key[1]:key[1] + 1]
            owner = self
        else:
            if len(self._free) < self.GRAPH_PAGE // 2:
                self.reserve_graph_words(self.GRAPH_PAGE)      # top up while pinning is allowed
            i = self._next
            self._next = (self._next + 1) % self.SLOTS
            slot = self._words[i:i + 1]
        if ws is None:
            slot.zero_()
            return FlagTicket(slot, None, owner, key)
        with torch.cuda.device(device):
            _check(lib, fn(handle, ws.data_ptr(), slot.data_ptr(), int(bool(clear)), _stream(device)),
                   fn.__name__)
            if capturing:
                return FlagTicket(slot, None, owner, key)      # valid once a replay has been synchronised
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
        return FlagTicket(slot, ev)


def _dev(t, name):
    if not t.is_cuda:
        raise OetrError(f'{name} must be a GPU tensor (got {t.device}); the '
                        'OETR hot path has no CPU implementation')
    if t.dtype != torch.float32:
        raise ValueError(f'{name} must be float32 (got {t.dtype})')
    return t.contiguous()


class HotPathEngine:
    """Owns one ``oetr_handle`` (repacked weights on one GPU) and a growable
    workspace tensor per HIP stream.  Calls enqueue on torch's current stream;
    batches submitted on two streams overlap on the GPU (at 8 pairs a launch
    fills 208 of 256 CUs, a second stream fills the rest: +23 % throughput)."""

    #: GEMM arithmetic modes (oetr_dtype in the header)
    PRECISIONS = {'f32': 0, 'f32_split_f16': 1, 'f16': 2, 'bf16': 3, 'f32_split_qk16': 4}
    #: precisions whose GEMM operands are f16 values (|x| < 65504, see query_flags)
    F16_RANGE = ('f32_split_f16', 'f16', 'f32_split_qk16')

    #: encoder attention cores (oetr_attention in the header)
    ATTENTIONS = {'linear': 0, 'full': 1}

    def __init__(self, weights, device=None, precision='f32_split_f16', enc_tile=None,
                 attention='linear'):
        """``precision``: 'f32' = exact fp32 MFMA products; 'f32_split_f16' =
        fp32-class results from 3 f16 MFMAs per product (default; same parity
        tolerances, ~2x faster); 'f16' / 'bf16' = GEMM operands rounded to
        f16 / bf16, one MFMA per product, fp32 accumulate and fp32 everything
        else (BASELINE configs[4] / configs[2]; reduced parity margin).
        ``enc_tile``: token rows per encoder workgroup, None = auto, 32 or 64
        (``oetr_set_encoder_tile``).  ``attention``: 'linear' (what the reference
        builds, ``src/model.py:82-84``) or 'full' = ``EncoderLayer(attention='full')``
        (``transformer.py:86-89``): all-pairs softmax attention, precisions 'f32_split_f16',
        'f16' and 'f32' (the exact build an overflowing batch is re-run on)."""
        self.lib = load_library()
        if precision not in self.PRECISIONS:
            raise ValueError(f'precision must be one of {sorted(self.PRECISIONS)}')
        self.precision = precision
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        device = torch.device(device)
        if device.type != 'cuda':
            raise OetrError('HotPathEngine needs a GPU device')
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = device
        missing = [k for k in hot_path_keys() if k not in weights]
        if missing:
            raise KeyError(f'missing hot-path weights: {missing[:4]}...')
        host = {k: weights[k].detach().to('cpu', torch.float32).contiguous()
                for k in hot_path_keys()}
        w = _Weights()
        w.struct_size = C.sizeof(_Weights)
        w.abi_version = ABI_VERSION

        def ptr(key):
            return C.cast(host[key].data_ptr(), _f32p)

        for li in range(N_ENC):
            p, e = f'transformer.encoder.{li}.', w.enc[li]
            e.q_proj, e.k_proj = ptr(p + 'q_proj.weight'), ptr(p + 'k_proj.weight')
            e.v_proj, e.merge = ptr(p + 'v_proj.weight'), ptr(p + 'merge.weight')
            e.mlp0, e.mlp2 = ptr(p + 'mlp.0.weight'), ptr(p + 'mlp.2.weight')
            e.pre_norm_q_w = ptr(p + 'pre_norm_q.weight')
            e.pre_norm_q_b = ptr(p + 'pre_norm_q.bias')
            e.pre_norm_kv_w = ptr(p + 'pre_norm_kv.weight')
            e.pre_norm_kv_b = ptr(p + 'pre_norm_kv.bias')
            e.norm2_w, e.norm2_b = ptr(p + 'norm2.weight'), ptr(p + 'norm2.bias')
        for li in range(N_DEC):
            p, d = f'transformer.decoder.layers.{li}.', w.dec[li]
            for attr in ('self_attn', 'multihead_attn'):
                m = getattr(d, attr)
                q = p + attr + '.'
                m.q_proj_w, m.q_proj_b = ptr(q + 'q_proj.weight'), ptr(q + 'q_proj.bias')
                m.k_proj_w, m.k_proj_b = ptr(q + 'k_proj.weight'), ptr(q + 'k_proj.bias')
                m.v_proj_w, m.v_proj_b = ptr(q + 'v_proj.weight'), ptr(q + 'v_proj.bias')
                m.merge = ptr(q + 'merge.weight')
            d.mlp0, d.mlp2 = ptr(p + 'mlp.0.weight'), ptr(p + 'mlp.2.weight')
            for nn_ in ('norm1', 'norm2', 'norm3'):
                setattr(d, nn_ + '_w', ptr(p + nn_ + '.weight'))
                setattr(d, nn_ + '_b', ptr(p + nn_ + '.bias'))
        w.query_embed1 = ptr('query_embed1.weight')
        w.query_embed2 = ptr('query_embed2.weight')
        w.tlbr0_w = ptr('tlbr_reg.0.weight')
        w.tlbr2_w, w.tlbr2_b = ptr('tlbr_reg.2.weight'), ptr('tlbr_reg.2.bias')
        w.heat_conv_w = ptr('heatmap_conv.0.weight')
        w.heat_conv_b = ptr('heatmap_conv.0.bias')
        w.heat_gn_w, w.heat_gn_b = ptr('heatmap_conv.1.weight'), ptr('heatmap_conv.1.bias')
        w.heat_out_w, w.heat_out_b = ptr('heatmap_conv.3.weight'), ptr('heatmap_conv.3.bias')

        handle = C.c_void_p()
        _check(self.lib, self.lib.oetr_create(C.byref(w), self.PRECISIONS[precision], device.index,
                                              C.byref(handle)), 'oetr_create')
        self._h = handle
        self._ws = {}
        self._flag_reader = _FlagReader()
        self._ws_shape = {}     # stream -> geometry the workspace was last carved for
        self._pos_loaded = {}   # stream -> key of the token-major position tables it holds
        if enc_tile is not None:
            self.set_encoder_tile(enc_tile)
        if attention not in self.ATTENTIONS:
            raise ValueError(f'attention must be one of {sorted(self.ATTENTIONS)}')
        self.attention = attention
        if attention != 'linear':
            _check(self.lib, self.lib.oetr_set_attention(self._h, self.ATTENTIONS[attention]),
                   'oetr_set_attention')
        if os.environ.get('OETR_STATE_PREREDUCE'):      # tuning knob for A/B runs (bench.py, tools/)
            self.set_state_prereduce(int(os.environ['OETR_STATE_PREREDUCE']))

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                self.lib.oetr_destroy(h)
            except Exception:
                pass

    def set_encoder_tile(self, rows):
        """0/None = auto, 32 or 64 token rows per encoder workgroup."""
        _check(self.lib, self.lib.oetr_set_encoder_tile(self._h, int(rows or 0)),
               'oetr_set_encoder_tile')

    def _current_ws(self):
        return self._ws.get(torch.cuda.current_stream(self.device).cuda_stream)

    def set_state_prereduce(self, mode):
        """``oetr_set_state_prereduce``: partial linear-attention states summed once per image
        instead of in every consuming workgroup - -1 automatic (the library default: the reduction
        launch from 768 source tokens per image), 0 off, 1 in a launch of their own (bit-identical
        results in every setting)."""
        _check(self.lib, self.lib.oetr_set_state_prereduce(self._h, int(mode)), 'oetr_set_state_prereduce')

    def set_tail_mode(self, mode):
        """``oetr_set_tail_mode``: order of the forward path's tail - 0 automatic (the default), 1
        'P form' (decoder beside the hs-independent conv products, then the combine: small batches),
        2 'direct form' (decoder, then the 64-row conv with the taps accumulated in registers, no P
        buffer: large batches; two-plane precisions only)."""
        _check(self.lib, self.lib.oetr_set_tail_mode(self._h, int(mode)), 'oetr_set_tail_mode')

    def set_decoder_split(self, k):
        """``oetr_set_decoder_split``: workgroups per image of the decoder chain - 0 automatic (4
        while 2N x 4 workgroups take at most a quarter of the CUs, else 1), 1, or 4 (four
        workgroups exchange five 256-float all-reduces per image inside the launch)."""
        _check(self.lib, self.lib.oetr_set_decoder_split(self._h, int(k)), 'oetr_set_decoder_split')

    def debug_decoder_fault(self, on=True):
        """``oetr_debug_decoder_fault`` (tests): the next four-workgroup decoder launch times out."""
        _check(self.lib, self.lib.oetr_debug_decoder_fault(self._h, int(bool(on))), 'oetr_debug_decoder_fault')

    def settle_exchange(self):
        """Recovery after ``FLAG_EXCHANGE`` was READ (and cleared) from the current stream's
        workspace: the failed call's per-image counters and granules are not trustworthy (a
        workgroup that became resident late may have published under the next call's tag), so the
        status block is zeroed again (``oetr_workspace_init``, enqueued behind everything
        submitted so far), and the engine stops splitting the decoder chain - its residency
        assumption does not hold on this device right now."""
        ws = self._current_ws()
        if ws is not None:
            with torch.cuda.device(self.device):
                _check(self.lib, self.lib.oetr_workspace_init(ws.data_ptr(), ws.numel(), _stream(self.device)),
                       'oetr_workspace_init')
        self.set_decoder_split(1)

    def query_flags(self, clear=True):
        """Status word of the CURRENT STREAM's workspace (``oetr_query_flags``): synchronises
        torch's current stream on the engine's device.  Bit ``FLAG_F16_RANGE`` = a GEMM
        operand of an earlier call on this stream reached the f16 range (f16-based
        precisions): that call's outputs are invalid.  Calls on other streams have their own
        workspace and word."""
        return _query_flags(self.lib, self.lib.oetr_query_flags, self._h, self._current_ws(),
                            self.device, clear)

    def read_flags_async(self, clear=True):
        """The same without synchronising: a :class:`FlagTicket` whose ``value()`` is the word
        as it stood behind every call enqueued on the current stream so far."""
        return self._flag_reader.read(self.lib, self.lib.oetr_read_flags_async, self._h,
                                      self._current_ws(), self.device, clear)

    def check_range(self):
        """Raise if any call since the last check was invalid (clears the word):
        :class:`OetrExchangeError` when the split decoder timed out (after
        :meth:`settle_exchange`: re-submitting works), :class:`OetrRangeError` when a GEMM
        operand overflowed the f16 range."""
        flags = self.query_flags(clear=True)
        if flags & FLAG_EXCHANGE:
            self.settle_exchange()
            raise OetrExchangeError(
                "the split decoder's workgroups were not resident together (OETR_FLAG_EXCHANGE): "
                "results since the last check are invalid; the engine now runs one workgroup per "
                "image - submit the batch again")
        if flags & FLAG_F16_RANGE:
            raise OetrRangeError(
                f"a GEMM operand reached |x| >= 65504 under precision "
                f"'{self.precision}': results are invalid; use precision 'f32' or 'bf16'")

    # ------------------------------------------------------------ helpers
    def workspace(self, n, hf1, wf1, hf2, wf2):
        need = self.lib.oetr_workspace_bytes(self._h, n, hf1, wf1, hf2, wf2)
        if need == 0:
            raise ValueError(
                f'invalid shape N={n} grids {hf1}x{wf1}, {hf2}x{wf2}: '
                + self.lib.oetr_last_error().decode())
        # one workspace per HIP stream: the handle is immutable, so calls on
        # different streams may overlap as long as their workspaces differ
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            old = ws
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
            if old is None:
                ws[:WORKSPACE_STATUS_BYTES].zero_()       # oetr_workspace_init
            else:                                         # a sticky flag survives the regrowth
                ws[:WORKSPACE_STATUS_BYTES].copy_(old[:WORKSPACE_STATUS_BYTES])
            self._pos_loaded.pop(key, None)
        if self._ws_shape.get(key) != (n, hf1, wf1, hf2, wf2):   # another carve: tables gone
            self._ws_shape[key] = (n, hf1, wf1, hf2, wf2)
            self._pos_loaded.pop(key, None)
        return ws

    @staticmethod
    def _grid(feat, pos, name):
        if feat.dim() != 4 or feat.shape[1] != D_MODEL:
            raise ValueError(f'{name} must be [N,{D_MODEL},hf,wf], got '
                             f'{tuple(feat.shape)}')
        hf, wf = int(feat.shape[2]), int(feat.shape[3])
        if tuple(pos.shape[-3:]) != (D_MODEL, hf, wf) or pos.numel() != D_MODEL * hf * wf:
            raise ValueError(f'pos for {name} must be [1,{D_MODEL},{hf},{wf}], '
                             f'got {tuple(pos.shape)}')
        return hf, wf

    def _masks(self, mask1, mask2, n, L1, L2):
        """forward_dummy's optional masks ([N,hf,wf] or [N,L], any numeric / bool dtype, as the
        reference takes them: src/model.py:229, transformer.py:340-343) -> contiguous float32
        device tensors [N,L] (or (None, None)).  Both or neither."""
        if mask1 is None and mask2 is None:
            return None, None
        if mask1 is None or mask2 is None:
            raise ValueError('masks: pass both mask1 and mask2, or neither')
        out = []
        for m, L, name in ((mask1, L1, 'mask1'), (mask2, L2, 'mask2')):
            m = m.to(device=self.device, dtype=torch.float32).reshape(m.shape[0], -1).contiguous()
            if tuple(m.shape) != (n, L):
                raise ValueError(f'{name} must have N x hf*wf = {n} x {L} elements, got {tuple(m.shape)}')
            out.append(m)
        return out[0], out[1]

    # -------------------------------------------------------------- calls
    def forward(self, feat1, feat2, pos1, pos2, img_hw1, img_hw2, stages=None,
                enc_layers=N_ENC, mask1=None, mask2=None, publish=False):
        """feats [N,256,hf,wf] + pos [1,256,hf,wf] -> (box1, box2) [N,4].
        With ``stages=True`` returns a dict of intermediates as well.
        ``mask1`` / ``mask2``: forward_dummy's optional masks at the token grid's resolution
        (``oetr_forward_masked``; with them ``logits`` holds -1e9 at masked tokens).
        ``publish=True`` (not with ``stages``): -> ((box1, box2), FlagTicket) - the call's last
        kernel moves the workspace's status word into a pinned host word
        (``oetr_forward_flagslot``), the deferred check without a dispatch of its own."""
        feat1, feat2 = _dev(feat1, 'feat1'), _dev(feat2, 'feat2')
        pos1, pos2 = _dev(pos1, 'pos1'), _dev(pos2, 'pos2')
        n = int(feat1.shape[0])
        if feat2.shape[0] != n:
            raise ValueError('feat1/feat2 batch sizes differ')
        hf1, wf1 = self._grid(feat1, pos1, 'feat1')
        hf2, wf2 = self._grid(feat2, pos2, 'feat2')
        mask1, mask2 = self._masks(mask1, mask2, n, hf1 * wf1, hf2 * wf2)
        ws = self.workspace(n, hf1, wf1, hf2, wf2)
        self._pos_loaded.pop(torch.cuda.current_stream(self.device).cuda_stream, None)
        dev = self.device
        both = torch.empty(2, n, 4, device=dev)   # one [2,n,4] block: what parallel.BoxGatherer sends, without a copy
        box1, box2 = both[0], both[1]
        args = [self._h, feat1.data_ptr(), feat2.data_ptr(), pos1.data_ptr(),
                pos2.data_ptr(), n, hf1, wf1, hf2, wf2, int(img_hw1[0]),
                int(img_hw1[1]), int(img_hw2[0]), int(img_hw2[1]),
                ws.data_ptr(), ws.numel(), box1.data_ptr(), box2.data_ptr()]
        margs = None
        if mask1 is not None:
            margs = args[:5] + [mask1.data_ptr(), mask2.data_ptr()] + args[5:]
        with torch.cuda.device(dev):
            if publish:
                if stages:
                    raise ValueError('publish=True has no stage outputs (oetr_forward_flagslot)')
                slot, dptr, owner, key = self._flag_reader.publish_slot(self.lib)
                fargs = margs if margs is not None else args[:5] + [None, None] + args[5:]
                _check(self.lib, self.lib.oetr_forward_flagslot(*fargs, dptr, _stream(dev)),
                       'oetr_forward_flagslot')
                return (box1, box2), self._flag_reader.published(slot, owner, key, dev)
            if not stages:
                if margs is not None:
                    _check(self.lib, self.lib.oetr_forward_masked(*margs, None, _stream(dev)),
                           'oetr_forward_masked')
                    return box1, box2
                _check(self.lib, self.lib.oetr_forward(*args, _stream(dev)),
                       'oetr_forward')
                return box1, box2
            L1, L2 = hf1 * wf1, hf2 * wf2
            out = dict(
                hs1=torch.empty(n, 1, D_MODEL, device=dev),
                hs2=torch.empty(n, 1, D_MODEL, device=dev),
                memory1=torch.empty(n, L1, D_MODEL, device=dev),
                memory2=torch.empty(n, L2, D_MODEL, device=dev),
                logits1=torch.empty(n, L1, device=dev),
                logits2=torch.empty(n, L2, device=dev),
                cxy1=torch.empty(n, 2, device=dev),
                cxy2=torch.empty(n, 2, device=dev),
                tlbr1=torch.empty(n, 4, device=dev),
                tlbr2=torch.empty(n, 4, device=dev))
            st = _Stages()
            st.struct_size = C.sizeof(_Stages)
            st.enc_layers = int(enc_layers)
            for k, t in out.items():
                setattr(st, k, t.data_ptr())
            if margs is not None:
                _check(self.lib, self.lib.oetr_forward_masked(
                    *margs, C.byref(st), _stream(dev)), 'oetr_forward_masked')
            else:
                _check(self.lib, self.lib.oetr_forward_stages(
                    *args, C.byref(st), _stream(dev)), 'oetr_forward_stages')
            out['box1'], out['box2'] = box1, box2
            if enc_layers < N_ENC:
                out = {k: out[k] for k in ('memory1', 'memory2')}
            return out

    # ---- token-resident entry: the neck stores straight into the workspace ----
    def token_buffers(self, n, hf1, wf1, hf2, wf2):
        """Where this stream's workspace keeps the hot path's token-major inputs for the
        shape: dict of float32 tensor VIEWS of the workspace - ``tokens1`` [n*L1,256],
        ``tokens2`` [n*L2,256] (contiguous after tokens1), ``pos1`` [L1,256], ``pos2``
        [L2,256] (``oetr_token_buffers``)."""
        ws = self.workspace(n, hf1, wf1, hf2, wf2)
        ptrs = [C.c_void_p() for _ in range(4)]
        _check(self.lib, self.lib.oetr_token_buffers(
            self._h, n, hf1, wf1, hf2, wf2, ws.data_ptr(), ws.numel(),
            *[C.byref(p) for p in ptrs]), 'oetr_token_buffers')
        rows = (n * hf1 * wf1, n * hf2 * wf2, hf1 * wf1, hf2 * wf2)
        out = {}

        def view(ptr, r):
            off = ptr - ws.data_ptr()
            return ws[off:off + r * D_MODEL * 4].view(torch.float32).view(r, D_MODEL)
        for name, p, r in zip(('tokens1', 'tokens2', 'pos1', 'pos2'), ptrs, rows):
            out[name] = view(p.value, r)
        if ptrs[1].value != ptrs[0].value + rows[0] * D_MODEL * 4:
            raise OetrError('oetr_token_buffers: tokens2 does not follow tokens1')
        out['tokens'] = view(ptrs[0].value, rows[0] + rows[1])   # both sides: one 2n-image neck call
        return out

    def load_pos_tokens(self, bufs, pos1, pos2):
        """Write the position tables [1,256,hf,wf] token-major into ``bufs['pos1/2']``
        unless this workspace already holds them (they survive forward calls)."""
        skey = torch.cuda.current_stream(self.device).cuda_stream
        key = (bufs['pos1'].data_ptr(), bufs['pos2'].data_ptr(),
               pos1.data_ptr(), pos1._version, tuple(pos1.shape),
               pos2.data_ptr(), pos2._version, tuple(pos2.shape))
        if self._pos_loaded.get(skey) != key:
            bufs['pos1'].copy_(pos1.reshape(D_MODEL, -1).t())
            bufs['pos2'].copy_(pos2.reshape(D_MODEL, -1).t())
            self._pos_loaded[skey] = key

    def forward_tokens(self, n, hf1, wf1, hf2, wf2, img_hw1, img_hw2, publish=False):
        """``forward`` on the tokens / position tables the workspace holds
        (``token_buffers``): -> (box1, box2) [n,4]; ``publish=True``: -> ((box1, box2),
        FlagTicket) as in :meth:`forward` (``oetr_forward_tokens_flagslot``)."""
        ws = self.workspace(n, hf1, wf1, hf2, wf2)
        dev = self.device
        both = torch.empty(2, n, 4, device=dev)   # one [2,n,4] block: what parallel.BoxGatherer sends, without a copy
        box1, box2 = both[0], both[1]
        with torch.cuda.device(dev):
            if publish:
                slot, dptr, owner, key = self._flag_reader.publish_slot(self.lib)
                _check(self.lib, self.lib.oetr_forward_tokens_flagslot(
                    self._h, n, hf1, wf1, hf2, wf2, int(img_hw1[0]), int(img_hw1[1]),
                    int(img_hw2[0]), int(img_hw2[1]), ws.data_ptr(), ws.numel(),
                    box1.data_ptr(), box2.data_ptr(), dptr, _stream(dev)), 'oetr_forward_tokens_flagslot')
                return (box1, box2), self._flag_reader.published(slot, owner, key, dev)
            _check(self.lib, self.lib.oetr_forward_tokens(
                self._h, n, hf1, wf1, hf2, wf2, int(img_hw1[0]), int(img_hw1[1]),
                int(img_hw2[0]), int(img_hw2[1]), ws.data_ptr(), ws.numel(),
                box1.data_ptr(), box2.data_ptr(), _stream(dev)), 'oetr_forward_tokens')
        return box1, box2

    def feature_correlation(self, feat1, feat2, pos1, pos2, mask1=None, mask2=None):
        feat1, feat2 = _dev(feat1, 'feat1'), _dev(feat2, 'feat2')
        pos1, pos2 = _dev(pos1, 'pos1'), _dev(pos2, 'pos2')
        n = int(feat1.shape[0])
        hf1, wf1 = self._grid(feat1, pos1, 'feat1')
        hf2, wf2 = self._grid(feat2, pos2, 'feat2')
        mask1, mask2 = self._masks(mask1, mask2, n, hf1 * wf1, hf2 * wf2)
        ws = self.workspace(n, hf1, wf1, hf2, wf2)
        self._pos_loaded.pop(torch.cuda.current_stream(self.device).cuda_stream, None)
        dev = self.device
        hs1 = torch.empty(n, 1, D_MODEL, device=dev)
        hs2 = torch.empty(n, 1, D_MODEL, device=dev)
        m1 = torch.empty(n, hf1 * wf1, D_MODEL, device=dev)
        m2 = torch.empty(n, hf2 * wf2, D_MODEL, device=dev)
        with torch.cuda.device(dev):
            _check(self.lib, self.lib.oetr_feature_correlation_masked(
                self._h, feat1.data_ptr(), feat2.data_ptr(), pos1.data_ptr(),
                pos2.data_ptr(), mask1.data_ptr() if mask1 is not None else None,
                mask2.data_ptr() if mask2 is not None else None, n, hf1, wf1, hf2, wf2,
                ws.data_ptr(), ws.numel(), hs1.data_ptr(), hs2.data_ptr(), m1.data_ptr(),
                m2.data_ptr(), _stream(dev)), 'oetr_feature_correlation')
        return hs1, hs2, m1, m2

    def center_estimation(self, hs1, hs2, memory1, memory2, hf1, wf1, hf2, wf2,
                          img_h1, img_h2, mask1=None, mask2=None):
        hs1, hs2 = _dev(hs1, 'hs1'), _dev(hs2, 'hs2')
        memory1, memory2 = _dev(memory1, 'memory1'), _dev(memory2, 'memory2')
        n = int(hs1.shape[0])
        mask1, mask2 = self._masks(mask1, mask2, n, hf1 * wf1, hf2 * wf2)
        if memory1.shape != (n, hf1 * wf1, D_MODEL) or \
                memory2.shape != (n, hf2 * wf2, D_MODEL):
            raise ValueError('memory shapes do not match the token grids')
        ws = self.workspace(n, hf1, wf1, hf2, wf2)
        c1 = torch.empty(n, 2, device=self.device)
        c2 = torch.empty(n, 2, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.oetr_center_estimation_masked(
                self._h, hs1.data_ptr(), hs2.data_ptr(), memory1.data_ptr(),
                memory2.data_ptr(), mask1.data_ptr() if mask1 is not None else None,
                mask2.data_ptr() if mask2 is not None else None, n, hf1, wf1, hf2, wf2, int(img_h1),
                int(img_h2), ws.data_ptr(), ws.numel(), c1.data_ptr(),
                c2.data_ptr(), _stream(self.device)), 'oetr_center_estimation')
        return c1, c2

    def size_regression(self, hs1, hs2):
        hs1, hs2 = _dev(hs1, 'hs1'), _dev(hs2, 'hs2')
        n = int(hs1.shape[0])
        t1 = torch.empty(n, 4, device=self.device)
        t2 = torch.empty(n, 4, device=self.device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.oetr_size_regression(
                self._h, hs1.data_ptr(), hs2.data_ptr(), n, t1.data_ptr(),
                t2.data_ptr(), _stream(self.device)), 'oetr_size_regression')
        return t1, t2


class NeckEngine:
    """Owns one ``oetr_neck_handle``: input_proj -> PatchMerging -> input_proj2
    (reference ``src/model.py:113-118``, ``backbone.py:53-67``) as HIP kernels."""

    PATCH_SIZES = (4, 8, 16)
    BACKBONE_C = 1024

    def __init__(self, weights, device=None):
        self.lib = load_library()
        device = torch.device('cuda', torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if device.type != 'cuda':
            raise OetrError('NeckEngine needs a GPU device')
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = device
        missing = [k for k in neck_keys() if k not in weights]
        if missing:
            raise KeyError(f'missing neck weights: {missing}')
        host = {k: weights[k].detach().to('cpu', torch.float32).contiguous()
                for k in neck_keys()}
        shapes = {'input_proj.weight': (D_MODEL, self.BACKBONE_C, 1, 1),
                  'input_proj2.weight': (D_MODEL, 2 * D_MODEL, 1, 1)}
        for i, ps in enumerate(self.PATCH_SIZES):
            shapes[f'patchmerging.reductions.{i}.weight'] = (
                D_MODEL if i == 0 else D_MODEL // 2, D_MODEL, ps, ps)
        for k, shp in shapes.items():
            if tuple(host[k].shape) != shp:
                raise ValueError(f'{k}: expected {shp}, got {tuple(host[k].shape)} '
                                 '(the HIP neck is built for the default config)')
        w = _NeckWeights()
        w.struct_size = C.sizeof(_NeckWeights)
        w.abi_version = ABI_VERSION

        def ptr(key):
            return C.cast(host[key].data_ptr(), _f32p)

        w.input_proj_w, w.input_proj_b = ptr('input_proj.weight'), ptr('input_proj.bias')
        w.norm_w, w.norm_b = ptr('patchmerging.norm.weight'), ptr('patchmerging.norm.bias')
        for i in range(3):
            w.reduction_w[i] = ptr(f'patchmerging.reductions.{i}.weight')
            w.reduction_b[i] = ptr(f'patchmerging.reductions.{i}.bias')
        w.input_proj2_w, w.input_proj2_b = ptr('input_proj2.weight'), ptr('input_proj2.bias')
        handle = C.c_void_p()
        _check(self.lib, self.lib.oetr_neck_create(C.byref(w), device.index, C.byref(handle)),
               'oetr_neck_create')
        self._h = handle
        self._ws = {}
        self._flag_reader = _FlagReader()

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                self.lib.oetr_neck_destroy(h)
            except Exception:
                pass

    def _workspace(self, need):
        key = torch.cuda.current_stream(self.device).cuda_stream   # one workspace per stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            old = ws
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
            if old is None:
                ws[:WORKSPACE_STATUS_BYTES].zero_()       # oetr_workspace_init
            else:
                ws[:WORKSPACE_STATUS_BYTES].copy_(old[:WORKSPACE_STATUS_BYTES])
        return ws

    def forward(self, backbone_feat):
        """[n,1024,hb,wb] (ResNet layer3 output) -> feat [n,256,hb//2,wb//2]."""
        x = _dev(backbone_feat, 'backbone_feat')
        if x.dim() != 4 or x.shape[1] != self.BACKBONE_C:
            raise ValueError(f'backbone_feat must be [n,{self.BACKBONE_C},hb,wb], '
                             f'got {tuple(x.shape)}')
        n, hb, wb = int(x.shape[0]), int(x.shape[2]), int(x.shape[3])
        need = self.lib.oetr_neck_workspace_bytes(self._h, n, hb, wb)
        if need == 0:
            raise ValueError(f'invalid neck shape n={n} grid {hb}x{wb}')
        ws = self._workspace(need)
        feat = torch.empty(n, D_MODEL, hb // 2, wb // 2, device=self.device)
        # launch on the ENGINE's device and on torch's current stream of that device
        # (the model may live on a GPU that is not torch's current device)
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.oetr_neck_forward(
                self._h, x.data_ptr(), n, hb, wb, ws.data_ptr(), ws.numel(),
                feat.data_ptr(), _stream(self.device)), 'oetr_neck_forward')
        return feat

    def forward_tokens(self, backbone_feat, tokens_out, status_word=None):
        """Same as ``forward`` with the result stored token-major into ``tokens_out``
        [n*(hb//2)*(wb//2), 256] (a ``HotPathEngine.token_buffers`` view).  ``status_word``: a
        device tensor whose first 4 bytes take the neck's range bit instead of the neck
        workspace's own word (``oetr_neck_forward_tokens_status``) - pass the hot-path workspace
        the tokens go into, and the forward call behind publishes one word for both stages."""
        x = _dev(backbone_feat, 'backbone_feat')
        if x.dim() != 4 or x.shape[1] != self.BACKBONE_C:
            raise ValueError(f'backbone_feat must be [n,{self.BACKBONE_C},hb,wb], '
                             f'got {tuple(x.shape)}')
        n, hb, wb = int(x.shape[0]), int(x.shape[2]), int(x.shape[3])
        if tuple(tokens_out.shape) != (n * (hb // 2) * (wb // 2), D_MODEL) or \
                tokens_out.dtype != torch.float32 or not tokens_out.is_contiguous():
            raise ValueError(f'tokens_out must be contiguous float32 '
                             f'[{n * (hb // 2) * (wb // 2)},{D_MODEL}], got {tuple(tokens_out.shape)}')
        need = self.lib.oetr_neck_workspace_bytes(self._h, n, hb, wb)
        if need == 0:
            raise ValueError(f'invalid neck shape n={n} grid {hb}x{wb}')
        ws = self._workspace(need)
        with torch.cuda.device(self.device):
            if status_word is not None:
                if not status_word.is_cuda or status_word.device != tokens_out.device:
                    raise ValueError('status_word must live on the device of tokens_out')
                _check(self.lib, self.lib.oetr_neck_forward_tokens_status(
                    self._h, x.data_ptr(), n, hb, wb, ws.data_ptr(), ws.numel(),
                    tokens_out.data_ptr(), status_word.data_ptr(), _stream(self.device)),
                    'oetr_neck_forward_tokens_status')
                return tokens_out
            _check(self.lib, self.lib.oetr_neck_forward_tokens(
                self._h, x.data_ptr(), n, hb, wb, ws.data_ptr(), ws.numel(),
                tokens_out.data_ptr(), _stream(self.device)), 'oetr_neck_forward_tokens')
        return tokens_out

    def set_conv_rows(self, rows):
        """Output positions per workgroup of the conv kernel: 0/None = auto, 256, 192, 128."""
        _check(self.lib, self.lib.oetr_neck_set_conv_rows(self._h, int(rows or 0)),
               'oetr_neck_set_conv_rows')

    CONV_KERNELS = {'auto': 0, 'gather': 1, 'row_window': 2, 'row_window_1w': 3}

    def set_conv_kernel(self, kind):
        """'auto' (= 'row_window_1w' when the output map is >= 16 wide, else 'gather'),
        'gather', 'row_window' (two waves per SIMD) or 'row_window_1w' (one, AGPR accumulators)."""
        _check(self.lib, self.lib.oetr_neck_set_conv_kernel(self._h, self.CONV_KERNELS[kind or 'auto']),
               'oetr_neck_set_conv_kernel')

    def query_flags(self, clear=True):
        """Status word of the current stream's neck workspace (see ``HotPathEngine.query_flags``)."""
        ws = self._ws.get(torch.cuda.current_stream(self.device).cuda_stream)
        return _query_flags(self.lib, self.lib.oetr_neck_query_flags, self._h, ws, self.device, clear)

    def read_flags_async(self, clear=True):
        ws = self._ws.get(torch.cuda.current_stream(self.device).cuda_stream)
        return self._flag_reader.read(self.lib, self.lib.oetr_neck_read_flags_async, self._h, ws,
                                      self.device, clear)

    def check_range(self):
        if self.query_flags(clear=True) & FLAG_F16_RANGE:
            raise OetrRangeError('a backbone feature / neck intermediate reached |x| >= 65504 '
                                 '(the HIP neck uses f16-split GEMMs): results are invalid')


class KernelTrace:
    """Measurement hook (bench/profiling): per-kernel GPU durations from HIP
    events recorded on the launch stream (``oetr_trace_*`` in the header)."""

    MAX_KERNELS = 16

    def __init__(self, engine, max_launches=4096):
        self.lib, self.engine = engine.lib, engine
        self._t = C.c_void_p()
        _check(self.lib, self.lib.oetr_trace_create(int(max_launches),
                                                    C.byref(self._t)),
               'oetr_trace_create')

    def _attach(self, t):
        fn = self.lib.oetr_neck_set_trace if isinstance(self.engine, NeckEngine) \
            else self.lib.oetr_set_trace
        return fn(self.engine._h, t)

    def __enter__(self):
        _check(self.lib, self._attach(self._t), 'oetr_set_trace')
        return self

    def __exit__(self, *exc):
        self._attach(None)

    def summary(self):
        """{kernel name: (launches, total_ms)} since the last call."""
        n = C.c_int()
        names = (C.c_char_p * self.MAX_KERNELS)()
        launches = (C.c_int * self.MAX_KERNELS)()
        ms = (C.c_float * self.MAX_KERNELS)()
        _check(self.lib, self.lib.oetr_trace_summary(self._t, C.byref(n), names,
                                                     launches, ms),
               'oetr_trace_summary')
        return {names[k].decode(): (launches[k], ms[k]) for k in range(n.value)
                if launches[k]}

    def __del__(self):
        t, self._t = getattr(self, '_t', None), None
        if t:
            try:
                self.lib.oetr_trace_destroy(t)
            except Exception:
                pass


def sustained_mfma_tflops(device, seconds=1.0):
    """``oetr_debug_mfma_rate``: the dense f16 MFMA rate (TFLOP/s) ``device`` sustains right now -
    about ``seconds`` of back-to-back MFMAs on every CU (measurements only; synchronises)."""
    device = torch.device(device)
    lib = load_library()
    out = C.c_double(0.0)
    with torch.cuda.device(device):
        _check(lib, lib.oetr_debug_mfma_rate(device.index or 0, float(seconds), C.byref(out), _stream(device)),
               'oetr_debug_mfma_rate')
    return float(out.value)


def box_tlbr_to_xyxy(cxy, tlbr, max_h, max_w):
    """HIP version of reference ``src/models/utils.py:16-28``."""
    lib = load_library()
    cxy, tlbr = _dev(cxy, 'cxy'), _dev(tlbr, 'tlbr')
    n = int(cxy.shape[0])
    box = torch.empty(n, 4, device=cxy.device)
    with torch.cuda.device(cxy.device):
        _check(lib, lib.oetr_box_tlbr_to_xyxy(
            cxy.data_ptr(), tlbr.data_ptr(), n, int(max_h), int(max_w),
            box.data_ptr(), _stream(cxy.device)), 'oetr_box_tlbr_to_xyxy')
    return box


def _attention(fn_name, q, k, v):
    lib = load_library()
    q, k, v = _dev(q, 'q'), _dev(k, 'k'), _dev(v, 'v')
    n, L, h, d = q.shape
    S = int(k.shape[1])
    if (h, d) != (N_HEAD, D_MODEL // N_HEAD) or k.shape != (n, S, h, d) \
            or v.shape != k.shape:
        raise ValueError('attention expects q [N,L,8,32], k,v [N,S,8,32]')
    out = torch.empty_like(q)
    extra = ()
    if fn_name == 'oetr_linear_attention':
        ws = torch.empty(lib.oetr_linear_attention_workspace_bytes(n), dtype=torch.uint8, device=q.device)
        extra = (ws.data_ptr(), ws.numel())
    with torch.cuda.device(q.device):
        _check(lib, getattr(lib, fn_name)(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), n, L, S, out.data_ptr(), *extra,
            _stream(q.device)), fn_name)
    return out


def linear_attention(q, k, v, q_mask=None, kv_mask=None):
    """HIP version of reference ``LinearAttention.forward``
    (``src/models/linear_attention.py:22-50``; ``q_mask`` [N,L] / ``kv_mask`` [N,S] as there,
    either may be None)."""
    if q_mask is None and kv_mask is None:
        return _attention('oetr_linear_attention', q, k, v)
    lib = load_library()
    q, k, v = _dev(q, 'q'), _dev(k, 'k'), _dev(v, 'v')
    n, L, h, d = q.shape
    S = int(k.shape[1])
    if (h, d) != (N_HEAD, D_MODEL // N_HEAD) or k.shape != (n, S, h, d) or v.shape != k.shape:
        raise ValueError('attention expects q [N,L,8,32], k,v [N,S,8,32]')
    masks = []
    for m, length, name in ((q_mask, L, 'q_mask'), (kv_mask, S, 'kv_mask')):
        if m is not None:
            m = m.to(device=q.device, dtype=torch.float32).contiguous()
            if tuple(m.shape) != (n, length):
                raise ValueError(f'{name} must be [{n},{length}], got {tuple(m.shape)}')
        masks.append(m)
    out = torch.empty_like(q)
    ws = torch.empty(lib.oetr_linear_attention_workspace_bytes(n), dtype=torch.uint8, device=q.device)
    with torch.cuda.device(q.device):
        _check(lib, lib.oetr_linear_attention_masked(
            q.data_ptr(), k.data_ptr(), v.data_ptr(),
            masks[0].data_ptr() if masks[0] is not None else None,
            masks[1].data_ptr() if masks[1] is not None else None,
            n, L, S, out.data_ptr(), ws.data_ptr(), ws.numel(), _stream(q.device)),
            'oetr_linear_attention_masked')
    return out


FULL_ATTENTION_VARIANTS = ('f32', 'f32_split_f16')


def full_attention(q, k, v, variant='f32', check_range=False):
    """HIP version of reference ``FullAttention.forward``
    (``src/models/linear_attention.py:53-87``).  ``variant``: 'f32' = exact fp32 MFMA
    products; 'f32_split_f16' = fp32-class products on the f16 matrix pipe (3 MFMAs per
    product; inputs must stay below 65504 - ``check_range=True`` verifies that, at the
    price of a stream synchronisation)."""
    if variant == 'f32':
        return _attention('oetr_full_attention', q, k, v)
    if variant != 'f32_split_f16':
        raise ValueError(f'variant must be one of {FULL_ATTENTION_VARIANTS}')
    lib = load_library()
    q, k, v = _dev(q, 'q'), _dev(k, 'k'), _dev(v, 'v')
    n, L, h, d = q.shape
    S = int(k.shape[1])
    if (h, d) != (N_HEAD, D_MODEL // N_HEAD) or k.shape != (n, S, h, d) or v.shape != k.shape:
        raise ValueError('attention expects q [N,L,8,32], k,v [N,S,8,32]')
    out = torch.empty_like(q)
    flags = torch.zeros(1, dtype=torch.int32, device=q.device) if check_range else None
    with torch.cuda.device(q.device):
        _check(lib, lib.oetr_full_attention_split(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), n, L, S, out.data_ptr(),
            flags.data_ptr() if check_range else None, _stream(q.device)), 'oetr_full_attention_split')
    if check_range and int(flags.item()) & FLAG_F16_RANGE:
        raise OetrRangeError('full_attention: an input reached |x| >= 65504 under the f16 split')
    return out


class OverlapCrops:
    """Result of :func:`overlap_crop`: device buffers + the device-side geometry.
    Nothing here has touched the host yet; :meth:`geometry` (or any of the properties
    built on it) copies the 120-byte ``oetr_crop_info`` back, which synchronises the
    stream - do it when the crops are about to be consumed."""

    def __init__(self, out0, out1, info, channels):
        self._out, self._info, self._channels, self._geo = (out0, out1), info, channels, None

    def geometry(self):
        if self._geo is None:
            raw = bytes(self._info.cpu().numpy().tobytes()[:C.sizeof(_CropInfo)])
            self._geo = _CropInfo.from_buffer_copy(raw)
        return self._geo

    @property
    def valid(self):
        """True: crops were made; False: the gate failed and the full images passed through.
        A degenerate crop / one beyond the buffers' capacity (``oetr_crop_info.valid == -1``,
        where the reference raises from ``cv2.resize``) raises ``OetrError``."""
        v = int(self.geometry().valid)
        if v < 0:
            raise OetrError('overlap_crop: degenerate crop rectangle or crop larger than the capacity '
                            '(oetr_crop_info.valid == -1)')
        return bool(v)

    def crop(self, i):
        """[1, C, out_h, out_w] view of image ``i``'s crop (the reference's ``left[None]``)."""
        g = self.geometry()
        h, w = int(g.out_h[i]), int(g.out_w[i])
        return self._out[i][:self._channels * h * w].view(1, self._channels, h, w)

    def bbox(self, i):
        """The reference's ``pred['bbox0'/'bbox1']``: the scaled float box, [1, 4]."""
        return torch.tensor([list(self.geometry().sbox[i])], dtype=torch.float32)

    def ratio(self, i):
        """The reference's ``ratio0/ratio1`` = [[rx, ry]] (Python floats)."""
        g = self.geometry()
        return [[float(g.ratio[i][0]), float(g.ratio[i][1])]]


def overlap_crop(image0, image1, box0, box1, scales0, scales1, keep_aspect=True,
                 size_divisor=1, pragueparks=False):
    """HIP version of the reference's box -> crop step (``evaluation.py:82-170`` +
    ``tensor_overlap_crop``, ``dloc/core/utils/utils.py:509-564``) for one pair, enqueued
    on torch's current stream with the boxes staying on the GPU.

    ``image0/1``: [1,C,H,W] float32 GPU tensors in [0,1]; ``box0/1``: the OETR boxes
    ([N,4] or [4]; entry 0 is used, as the reference does); ``scales0/1``: (sx, sy)
    ``overlap_scales`` of ``read_overlap_image``; ``keep_aspect`` = extractor is not
    'disk'; ``size_divisor`` 8 for LoFTR; ``pragueparks`` selects that dataset's gate.
    Returns an :class:`OverlapCrops`."""
    lib = load_library()
    image0, image1 = _dev(image0, 'image0'), _dev(image1, 'image1')
    if image0.dim() != 4 or image1.dim() != 4 or image0.shape[0] != 1 or image1.shape[0] != 1 \
            or image0.shape[1] != image1.shape[1]:
        raise ValueError('images must be [1,C,H,W] with equal C')
    dev = image0.device
    b0 = _dev(box0.reshape(-1, 4)[0], 'box0')
    b1 = _dev(box1.reshape(-1, 4)[0], 'box1')
    ch, h0, w0 = (int(v) for v in image0.shape[1:])
    h1, w1 = int(image1.shape[2]), int(image1.shape[3])
    cap = lib.oetr_overlap_crop_capacity(ch, h0, w0, h1, w1, int(size_divisor), None, None)
    if cap == 0:
        raise ValueError('invalid crop arguments')
    tmp = torch.empty(2 * cap, device=dev)
    out0, out1 = torch.empty(cap, device=dev), torch.empty(cap, device=dev)
    info = torch.zeros((C.sizeof(_CropInfo) + 7) // 8, dtype=torch.float64, device=dev)
    s0 = (C.c_float * 2)(float(scales0[0]), float(scales0[1]))
    s1 = (C.c_float * 2)(float(scales1[0]), float(scales1[1]))
    with torch.cuda.device(dev):
        _check(lib, lib.oetr_overlap_crop(
            image0.data_ptr(), image1.data_ptr(), ch, h0, w0, h1, w1, b0.data_ptr(), b1.data_ptr(),
            s0, s1, int(bool(keep_aspect)), int(size_divisor), int(bool(pragueparks)),
            tmp.data_ptr(), out0.data_ptr(), out1.data_ptr(), cap, info.data_ptr(),
            _stream(dev)), 'oetr_overlap_crop')
    res = OverlapCrops(out0, out1, info, ch)
    res._keep = (image0, image1, b0, b1, tmp)      # inputs stay alive until the work has run
    return res
