"""Device-side image reader of the overlap pipeline (the READER half).

The reference reads ONE image per call on the host - ``read_overlap_image``
(``dloc/core/utils/utils.py:271-343``): ``cv2.imread``, two ``cv2.resize`` calls (to the
matcher's frame, then from that to the OETR input frame ``resize[0] x resize[0]``), ``/ 255``,
upload - and returns ``(image, overlap_inp, inp, scales, overlap_scales)``.  Here everything
after the decode runs on the GPU (``csrc/reader.hip`` behind ``oetr_overlap_frame`` /
``oetr_read_overlap_image``), and a batch of decoded images crosses the PCIe bus in ONE copy
from one pinned staging buffer.

Host logic only in this file: the arithmetic lives in the library (the frame sizes and scale
factors in ``oetr_overlap_frame``, usable without a GPU).
"""
import ctypes as C

import numpy as np
import torch

from .hip_engine import OetrError, _check, _stream, load_library

ALIGN = {'': 0, None: 0, 'disk': 1, 'loftr': 2}


def overlap_frame(w, h, resize=(640,), align='disk'):
    """Sizes and scale factors ``read_overlap_image(..., overlap=True)`` uses for a decoded
    ``w x h`` image (reference ``utils.py:283-309``): dict ``w_new, h_new`` (matcher frame),
    ``w_ov, h_ov`` (OETR frame), ``scales``, ``overlap_scales`` (tuples of Python floats).
    ``resize``: ``[S]`` (square S x S frame) or ``[-1]`` (native size), as the reference's
    ``--resize`` option."""
    resize = list(resize)
    if len(resize) != 1:
        raise ValueError('overlap frame: resize must be [S] or [-1] (reference utils.py:296-300)')
    if align not in ALIGN:
        raise ValueError(f"align must be one of 'disk', 'loftr', '' (got {align!r})")
    lib = load_library()
    iv = [C.c_int() for _ in range(4)]
    sc, osc = (C.c_double * 2)(), (C.c_double * 2)()
    _check(lib, lib.oetr_overlap_frame(int(w), int(h), int(resize[0]), ALIGN[align], *[C.byref(v) for v in iv],
                                       sc, osc), 'oetr_overlap_frame')
    return dict(w_new=iv[0].value, h_new=iv[1].value, w_ov=iv[2].value, h_ov=iv[3].value,
                scales=(sc[0], sc[1]), overlap_scales=(osc[0], osc[1]))


def _as_hwc(image):
    t = torch.from_numpy(image) if isinstance(image, np.ndarray) else image
    if t.dim() != 3 or t.shape[2] != 3 or t.dtype not in (torch.uint8, torch.float32):
        raise ValueError(f'decoded image must be [H,W,3] uint8 or float32 (cv2.imread layout), got '
                         f'{tuple(t.shape)} {t.dtype}')
    return t.contiguous()


class ReadImage:
    """What ``read_overlap_image`` returns, on the device: ``overlap_inp`` [1,S,S,3] (a view of
    the batch slot it was written to), ``inp`` [1,1|3,h_new,w_new], ``scales``,
    ``overlap_scales``.  (The reference's first value, the grey picture as a host array, is
    ``inp * 255`` for grayscale readers.)"""

    def __init__(self, overlap_inp, inp, frame):
        self.overlap_inp, self.inp = overlap_inp, inp
        self.scales, self.overlap_scales = frame['scales'], frame['overlap_scales']
        self.frame = frame


def read_overlap_images(images, device, resize=(640,), grayscale=True, align='disk', rotation=0):
    """Decoded BGR images (``[H,W,3]`` uint8 / float32, numpy or torch, any sizes) -> list of
    :class:`ReadImage` on ``device``.  The bytes of ALL images go through one pinned staging
    buffer and one host-to-device copy; images whose OETR frames agree in size share one
    ``[n,H,W,3]`` batch tensor (``ReadImage.overlap_inp`` are its slots, in input order within
    the group).  ``rotation`` (reference ``utils.py:322-325``): the matcher picture ``inp`` - not the
    OETR frame - is turned by k x 90 degrees counter-clockwise and an odd k swaps ``scales``
    (a permutation of the finished picture on the device: same pixels).  Enqueue-only on torch's
    current stream."""
    lib = load_library()
    device = torch.device(device)
    if device.type != 'cuda':
        raise OetrError('read_overlap_images needs a GPU device (the reader has no CPU implementation)')
    imgs = [_as_hwc(im) for im in images]
    frames = [overlap_frame(int(t.shape[1]), int(t.shape[0]), resize, align) for t in imgs]
    # one staging buffer (256-byte aligned pieces), one copy
    offs, total = [], 0
    for t in imgs:
        offs.append(total)
        total += (t.numel() * t.element_size() + 255) // 256 * 256
    dev_bytes = []
    if total:
        if all(t.is_cuda for t in imgs):
            dev_bytes = None
        else:
            stage = torch.empty(total, dtype=torch.uint8).pin_memory()
            for t, o in zip(imgs, offs):
                n = t.numel() * t.element_size()
                stage[o:o + n].copy_(t.cpu().reshape(-1).view(torch.uint8))
            dev_bytes = stage.to(device, non_blocking=True)
    groups = {}
    for i, fr in enumerate(frames):
        groups.setdefault((fr['h_ov'], fr['w_ov']), []).append(i)
    batches = {k: torch.empty(len(v), k[0], k[1], 3, device=device) for k, v in groups.items()}
    out = [None] * len(imgs)
    keep = [dev_bytes]
    with torch.cuda.device(device):
        for key, idxs in groups.items():
            for slot, i in enumerate(idxs):
                t, fr = imgs[i], frames[i]
                if dev_bytes is None:
                    src = t.to(device)
                    keep.append(src)
                    src_ptr = src.data_ptr()
                else:
                    src_ptr = dev_bytes.data_ptr() + offs[i]
                tmp = torch.empty(fr['h_new'] * fr['w_new'] * 3, device=device)
                inp = torch.empty(1, 1 if grayscale else 3, fr['h_new'], fr['w_new'], device=device)
                ov = batches[key][slot:slot + 1]
                _check(lib, lib.oetr_read_overlap_image(
                    src_ptr, int(t.dtype == torch.uint8), int(t.shape[0]), int(t.shape[1]), fr['h_new'],
                    fr['w_new'], fr['h_ov'], fr['w_ov'], int(bool(grayscale)), int(not align), tmp.data_ptr(), ov.data_ptr(),
                    inp.data_ptr(), _stream(device)), 'oetr_read_overlap_image')
                keep.append(tmp)
                if rotation % 4:
                    inp = torch.rot90(inp, int(rotation), dims=(2, 3)).contiguous()
                    if rotation % 2:
                        fr = dict(fr, scales=fr['scales'][::-1])
                res = ReadImage(ov, inp, fr)
                res._batch, res._slot, res._keep = batches[key], slot, keep
                out[i] = res
    return out
