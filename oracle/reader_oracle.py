"""CPU restatement of the reference's image reader for the overlap model (the
READER half).  TEST INFRASTRUCTURE: only tests/, smoke() and bench.py's cpu_baseline leg may
import it.

What it follows (paths relative to the reference root): ``dloc/core/utils/utils.py:271-343``
(``read_overlap_image``) with ``process_resize`` (``:248-265``):

* the decoded BGR image (``cv2.imread``) as float32 - channel order REVERSED when ``align`` is
  empty (``:283-284``; the grey conversion afterwards still treats the channels as B,G,R: a
  reference quirk, kept); ``align='disk'`` / ``'loftr'`` round the
  matcher's frame UP to a multiple of 32 / 8, otherwise it keeps the size (``:287-294``);
  ``overlap=True`` adds the OETR input frame - ``resize[0] x resize[0]`` (square, aspect NOT
  kept) or the native size for ``resize == [-1]`` (``:296-300``);
* ``scales = (w / w_new, h / h_new)``, ``overlap_scales = (w_new / w_ov, h_new / h_ov)`` as
  Python floats (``:304-309``);
* two resizes: image -> (w_new, h_new), and THAT result -> (w_ov, h_ov) (``:311-320``);
* ``overlap_inp = overlap_image[None] / 255`` ([1,H,W,3], BGR: what ``OETR.forward_dummy``
  takes), ``inp`` = the matcher's image / 255: grayscale ``[1,1,h,w]`` or colour
  ``[1,3,h,w]`` (``:326-338``).  Rotation (``:322-325``) is not restated: every caller passes 0.

Parity status: sizes, scales and shapes are pinned to the reference by
``tests/golden/reader.npz`` (``oracle/gen_golden.py`` imports ``read_overlap_image`` itself).
The pixels are **parity-unpinned**: ``cv2.resize`` (default ``INTER_LINEAR``) and
``cv2.cvtColor`` are not available in the build image; :func:`bilinear_resize` and
:func:`bgr_to_gray` restate OpenCV's published float32 algorithms (pixel centres
``(d + 0.5) * scale - 0.5``, the two taps clamped to the border, horizontal then vertical;
grey = 0.114 B + 0.587 G + 0.299 R) and are cross-checked against torch's independent
``F.interpolate(mode='bilinear', align_corners=False)`` in ``tests/test_reader_cpu.py``.
"""
import math

import numpy as np
import torch


def process_resize(w, h, resize):
    """``dloc/core/utils/utils.py:248-265``."""
    assert 0 < len(resize) <= 2
    if len(resize) == 1 and resize[0] > -1:
        scale = resize[0] / max(h, w)
        return int(round(w * scale)), int(round(h * scale))
    if len(resize) == 1 and resize[0] == -1:
        return w, h
    return resize[0], resize[1]


def overlap_frame(w, h, resize, align='disk', overlap=True):
    """Sizes and scale factors of ``read_overlap_image`` (``:283-309``) for a ``w x h`` image."""
    if align == 'disk':
        w_new, h_new = math.ceil(w / 32) * 32, math.ceil(h / 32) * 32
    elif align == 'loftr':
        w_new, h_new = math.ceil(w / 8) * 8, math.ceil(h / 8) * 8
    else:
        w_new, h_new = process_resize(w, h, [-1])
    out = {}
    if overlap:
        if len(resize) == 1 and resize[0] == -1:
            w_ov, h_ov = w, h
        else:
            w_ov, h_ov = resize[0], resize[0]
        out.update(w_ov=w_ov, h_ov=h_ov,
                   overlap_scales=(float(w_new) / float(w_ov), float(h_new) / float(h_ov)))
    else:
        w_new, h_new = process_resize(w, h, resize)
    out.update(w_new=w_new, h_new=h_new, scales=(float(w) / float(w_new), float(h) / float(h_new)))
    return out


def _axis_table(src, dst):
    scale = float(src) / float(dst)
    i0 = np.empty(dst, dtype=np.int64)
    i1 = np.empty(dst, dtype=np.int64)
    f = np.empty(dst, dtype=np.float32)
    for d in range(dst):
        fx = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(fx))
        t = np.float32(fx - np.float32(s))
        if s < 0:
            s, t = 0, np.float32(0.0)
        if s >= src - 1:
            s, t = src - 1, np.float32(0.0)
        i0[d], i1[d], f[d] = s, min(s + 1, src - 1), t
    return i0, i1, f


def bilinear_resize(img, new_w, new_h):
    """``img`` [h,w] or [h,w,c] float32 -> [new_h,new_w(,c)]: OpenCV's float32 INTER_LINEAR
    (cv2.resize argument order: width first)."""
    a = np.asarray(img, dtype=np.float32)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[:, :, None]
    h, w, _ = a.shape
    if (new_w, new_h) == (w, h):
        out = a.copy()
    else:
        x0, x1, fx = _axis_table(w, new_w)
        y0, y1, fy = _axis_table(h, new_h)
        one = np.float32(1.0)
        rows = a[:, x0, :] * (one - fx)[None, :, None] + a[:, x1, :] * fx[None, :, None]
        out = rows[y0] * (one - fy)[:, None, None] + rows[y1] * fy[:, None, None]
    out = out.astype(np.float32)
    return out[:, :, 0] if squeeze else out


def bgr_to_gray(img):
    """cv2.cvtColor(img, COLOR_BGR2GRAY) for float32 images: 0.114 B + 0.587 G + 0.299 R."""
    a = np.asarray(img, dtype=np.float32)
    return (a[:, :, 0] * np.float32(0.114) + a[:, :, 1] * np.float32(0.587) + a[:, :, 2] * np.float32(0.299)).astype(np.float32)


def read_overlap_image(image_bgr, resize, grayscale=False, align='disk', rotation=0):
    """The reader on an already decoded BGR image [h,w,3] (uint8 or float): returns
    ``dict(image, overlap_inp, inp, scales, overlap_scales)`` like ``:340-343`` (``image`` =
    the grey matcher-frame picture, 0..255).  ``rotation`` (:322-325): the matcher picture -
    not the OETR frame - is turned by k x 90 degrees counter-clockwise (``np.rot90``) after the
    resizes, and an odd k swaps ``scales``."""
    img = np.asarray(image_bgr)
    if not align:                       # utils.py:283-284: "BGR to RGB" only without alignment - the
        img = img[:, :, ::-1]           # later grey conversion still reads the channels as B,G,R
    img = img.astype(np.float32)
    h, w = img.shape[:2]
    fr = overlap_frame(w, h, resize, align, overlap=True)
    image = bilinear_resize(img, fr['w_new'], fr['h_new'])
    overlap_image = bilinear_resize(image, fr['w_ov'], fr['h_ov'])
    overlap_inp = torch.from_numpy(overlap_image[None] / 255.0).float()
    scales = fr['scales']
    if rotation != 0:
        image = np.ascontiguousarray(np.rot90(image, k=rotation))
        if rotation % 2:
            scales = scales[::-1]
    gray = bgr_to_gray(image)
    if grayscale:
        inp = torch.from_numpy(gray[None, None] / 255.0).float()
    else:
        inp = torch.from_numpy(image.transpose((2, 0, 1))[None] / 255.0).float()
    return dict(image=gray, overlap_inp=overlap_inp, inp=inp, scales=scales,
                overlap_scales=fr['overlap_scales'])
