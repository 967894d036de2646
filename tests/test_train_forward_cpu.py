"""CPU side of the training-pipeline forward (reference
``src/model.py:255-376``): the box losses / metrics of ``imagematching_oetr_amd/losses.py``
against ``tests/golden/train_forward.npz`` (written by the imported reference model's own
``forward(data)``, ``oracle/gen_golden.py::gen_train_forward``) fed with the golden's
predicted boxes, and the contract checks of ``OETR.forward`` that need no GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import imagematching_oetr_amd as pkg
from imagematching_oetr_amd import losses


def test_losses_reproduce_the_reference_results(golden_dir):
    g = np.load(golden_dir / 'train_forward.npz')
    v = torch.from_numpy(g['overlap_valid'])
    gt1, gt2 = torch.from_numpy(g['overlap_box1'])[v], torch.from_numpy(g['overlap_box2'])[v]
    p1, p2 = torch.from_numpy(g['giou_pred_bbox1']), torch.from_numpy(g['giou_pred_bbox2'])
    giou = ((losses.giou_loss(p1, gt1) + losses.giou_loss(p2, gt2)) / 2).mean()
    oiou = ((losses.oiou_loss(p1, gt1) + losses.oiou_loss(p2, gt2)) / 2).mean()
    assert float(giou) == float(g['giou_iouloss']) and float(oiou) == float(g['oiou_cycle_iouloss'])
    assert float(losses.bbox_iou_aligned(p1, gt1).mean()) == float(g['giou_iou1'])
    assert float(losses.bbox_iou_aligned(p2, gt2).mean()) == float(g['giou_iou2'])
    assert float(losses.bbox_oiou(gt1, p1).mean()) == float(g['giou_oiou1'])
    assert float(losses.bbox_oiou(gt2, p2).mean()) == float(g['giou_oiou2'])
    # loc / wh losses from the same boxes (images 128x160 and 160x128)
    cx = lambda b: torch.cat([(b[:, :2] + b[:, 2:]) / 2, b[:, 2:] - b[:, :2]], -1)   # noqa: E731
    gtc1 = losses.box_xyxy_to_cxywh(gt1, max_h=128, max_w=160)
    gtc2 = losses.box_xyxy_to_cxywh(gt2, max_h=160, max_w=128)
    s1, s2 = torch.tensor([160, 128]), torch.tensor([128, 160])
    loc = F.l1_loss(cx(p1)[:, :2] / s1, gtc1[:, :2] / s1) + F.l1_loss(cx(p2)[:, :2] / s2, gtc2[:, :2] / s2)
    wh = (F.l1_loss(cx(p1)[:, 2:] / s1, gtc1[:, 2:] / s1) + F.l1_loss(cx(p2)[:, 2:] / s2, gtc2[:, 2:] / s2)) / 2
    assert abs(float(loc) - float(g['giou_loc_loss'])) < 1e-6
    assert abs(float(wh) - float(g['giou_wh_loss'])) < 1e-6


def test_obtain_overlap_bbox_is_unclamped():
    cxy = torch.tensor([[10.0, 20.0]])
    tlbr = torch.tensor([[0.5, 0.25, 0.5, 1.0]])     # top, left, bottom, right fractions
    x1, x2, c1, _ = losses.obtain_overlap_bbox(cxy, tlbr, cxy, tlbr, (100, 200), (100, 200))
    assert x1.tolist() == [[10 - 50.0, 20 - 50.0, 10 + 200.0, 20 + 50.0]]     # beyond the image: kept
    assert c1.tolist() == [[85.0, 20.0, 250.0, 100.0]]
    assert torch.equal(x1, x2)


def test_forward_contract_without_a_gpu():
    model = pkg.OETR(pkg.get_cfg_defaults().OETR)
    data = {'image1': torch.rand(1, 64, 64, 3), 'image2': torch.rand(1, 64, 64, 3),
            'overlap_valid': torch.tensor([True]),
            'overlap_box1': torch.tensor([[1.0, 1.0, 30.0, 30.0]]),
            'overlap_box2': torch.tensor([[1.0, 1.0, 30.0, 30.0]])}
    with torch.enable_grad():
        with pytest.raises(NotImplementedError, match='no backward'):
            model(data)                   # autograd on: no graph-less losses by accident
    with torch.no_grad():
        with pytest.raises(RuntimeError):     # masks (reference model.py:256-258) go the same way: no CPU route
            model.eval()(dict(data, resize_mask1=torch.ones(1, 2, 2), resize_mask2=torch.ones(1, 2, 2)))
        model.hip_attention = 'full'
        with pytest.raises(NotImplementedError, match='masks'):
            model(dict(data, resize_mask1=torch.ones(1, 2, 2), resize_mask2=torch.ones(1, 2, 2)))
        model.hip_attention = 'linear'
        model.train()
        with pytest.raises(RuntimeError):     # CPU tensors: the hot path has no CPU implementation
            model.eval()(data)
