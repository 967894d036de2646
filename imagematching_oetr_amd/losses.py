"""Box losses / metrics of the training-side ``OETR.forward`` (reference
``src/model.py:255-376``), host code on [N,4] tensors like the reference's own:

* ``obtain_overlap_bbox``  ``src/model.py:193-226``  - UNCLAMPED xyxy + cxywh boxes
* ``box_xyxy_to_cxywh``    ``src/models/utils.py:42-54``
* ``giou_loss``            ``src/losses/losses.py:113-152``
* ``bbox_oiou`` / ``oiou_loss``  ``src/losses/utils.py:107-119``, ``losses.py:107-110``
* ``bbox_iou_aligned``     ``src/losses/utils.py:69-104`` (``is_aligned=True`` branch)

They run on whatever device the boxes live on (the GPU after the HIP stages); none of
this is on the hot path.
"""
import torch


def obtain_overlap_bbox(cxy1, tlbr1, cxy2, tlbr2, hw1, hw2):
    """-> (xyxy1, xyxy2, cxywh1, cxywh2), no clamping (training-time boxes)."""
    out = []
    for cxy, tlbr, (h, w) in ((cxy1, tlbr1, hw1), (cxy2, tlbr2, hw2)):
        xyxy = torch.stack([cxy[:, 0] - tlbr[:, 1] * w, cxy[:, 1] - tlbr[:, 0] * h,
                            cxy[:, 0] + tlbr[:, 3] * w, cxy[:, 1] + tlbr[:, 2] * h], dim=1)
        cxywh = torch.cat([(xyxy[:, :2] + xyxy[:, 2:]) / 2, xyxy[:, 2:] - xyxy[:, :2]], dim=-1)
        out.append((xyxy, cxywh))
    return out[0][0], out[1][0], out[0][1], out[1][1]


def box_xyxy_to_cxywh(xyxy, max_h, max_w):
    x1, y1, x2, y2 = xyxy.unbind(-1)
    x1, x2 = x1.clamp(min=0.0, max=max_w), x2.clamp(min=0.0, max=max_w)
    y1, y2 = y1.clamp(min=0.0, max=max_h), y2.clamp(min=0.0, max=max_h)
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], dim=-1)


def _overlap_area(pred, target):
    wh = (torch.min(pred[:, 2:], target[:, 2:]) - torch.max(pred[:, :2], target[:, :2])).clamp(min=0)
    return wh[:, 0] * wh[:, 1]


def giou_loss(pred, target, eps=1e-7):
    overlap = _overlap_area(pred, target)
    ap = (pred[:, 2] - pred[:, 0]) * (pred[:, 3] - pred[:, 1])
    ag = (target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1])
    union = ap + ag - overlap + eps
    ious = overlap / union
    enclose_wh = (torch.max(pred[:, 2:], target[:, 2:]) - torch.min(pred[:, :2], target[:, :2])).clamp(min=0)
    enclose_area = enclose_wh[:, 0] * enclose_wh[:, 1] + eps
    return 1 - (ious - (enclose_area - union) / enclose_area)


def bbox_oiou(target, pred, eps=1e-7):
    """Overlap area over the TARGET's area (argument order as in the reference)."""
    return _overlap_area(pred, target) / ((target[:, 2] - target[:, 0]) * (target[:, 3] - target[:, 1]))


def oiou_loss(pred, target, eps=1e-7):
    return 1 - bbox_oiou(target, pred, eps)


def bbox_iou_aligned(a, b, eps=1e-6):
    overlap = _overlap_area(a, b)
    area1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    union = (area1 + area2 - overlap).clamp(min=eps)
    return overlap / union
