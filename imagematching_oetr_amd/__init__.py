"""MI355X-native OETR overlap-estimation hot path.

Drop-in for the inference half of TencentYoutuResearch/ImageMatching-OETR's
``src/model.py``: same ``OETR`` constructor, checkpoint keys and
``forward_dummy`` signature; the feature-correlation transformer and the
centre/size regression heads run as hand-written HIP kernels for gfx950
behind the C ABI in ``include/oetr_hip.h``.
"""
import os as _os

# Streams and hardware queues: the HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES
# hardware queues (default 4), and work of two streams that share a queue is serialised.  The throughput
# mode (model.hip_streams = 3) uses three side streams + the caller's stream = four; a fifth stream (an
# asynchronous collective on the process group's stream) halved the overlapped rate (30.5 k -> 14.7 k
# pairs/s), which is why parallel.BoxGatherer issues a BLOCKING collective on the batch's own side stream
# in that mode.  More queues are not a fix: with 8 the stream -> queue assignment depends on the order in
# which torch / RCCL / this package create their streams (28-32 k by order; profiles/r5_pg_streams.txt).

from .config import Cfg, get_cfg_defaults  # noqa: F401
from .model import OETR, build_detectors  # noqa: F401
from .pipeline import forward_pairs, forward_pairs_raw, forward_pairs_sharded  # noqa: F401
from .reader import overlap_frame, read_overlap_images  # noqa: F401
from .hip_engine import (FLAG_EXCHANGE, FLAG_F16_RANGE, FLAG_INVALID, FULL_ATTENTION_VARIANTS, HotPathEngine, KernelTrace, NeckEngine, OetrError,  # noqa: F401
                         OetrExchangeError, OetrRangeError,
                         box_tlbr_to_xyxy, full_attention, hot_path_keys,
                         linear_attention, load_library, neck_keys, overlap_crop)

__all__ = ['Cfg', 'get_cfg_defaults', 'OETR', 'build_detectors',
           'HotPathEngine', 'KernelTrace', 'NeckEngine', 'neck_keys', 'OetrError', 'OetrExchangeError', 'OetrRangeError', 'FLAG_F16_RANGE', 'FLAG_EXCHANGE', 'FLAG_INVALID', 'box_tlbr_to_xyxy', 'full_attention',
           'linear_attention', 'FULL_ATTENTION_VARIANTS', 'hot_path_keys', 'load_library', 'overlap_crop', 'forward_pairs',
           'forward_pairs_sharded', 'forward_pairs_raw', 'overlap_frame', 'read_overlap_images']
