"""MI355X-native OETR overlap-estimation hot path.

Drop-in for the inference half of TencentYoutuResearch/ImageMatching-OETR's
``src/model.py``: same ``OETR`` constructor, checkpoint keys and
``forward_dummy`` signature; the feature-correlation transformer and the
centre/size regression heads run as hand-written HIP kernels for gfx950
behind the C ABI in ``include/oetr_hip.h``.
"""
import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4)
# and work of two streams that share a queue is serialised.  The throughput mode (model.hip_streams = 3)
# uses three side streams + the caller's stream, and RCCL adds its own: five streams on four queues
# halved the overlapped rate (measured, MI355X: 30.5 k -> 14.7 k pairs/s with a process group up; 27.5 k
# with eight queues).  The variable is read when the runtime initialises - i.e. before the first HIP
# call of the process, normally after this import - and an explicit setting of the caller's wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

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
