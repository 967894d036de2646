"""GPU parity of the HIP neck (input_proj -> PatchMerging -> input_proj2,
SURVEY.md §8f.1) through the C ABI, against the reference goldens and the
oracle.  Tolerance: the reference's own fp32-vs-fp64 drift on ``feat`` is
~5e-6 (abs-max 5.7); the bound below is 10x that."""
import glob
from pathlib import Path

import numpy as np
import pytest
import torch

import imagematching_oetr_amd as pkg
from oracle import oetr_oracle as orc
from tests.test_oracle_golden import load_neck_case

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
NECK = sorted(glob.glob(str(Path(__file__).parent / 'golden' / 'neck_*.npz')))
FEAT_TOL = 5e-5


@pytest.mark.parametrize('path', NECK, ids=lambda p: p.split('neck_')[-1][:-4])
def test_neck_matches_reference_golden(gpu, path):
    g, w, bb = load_neck_case(path)
    eng = pkg.NeckEngine(w, device=gpu)
    feat = eng.forward(bb.to(gpu))
    ref = torch.from_numpy(g['feat'])
    assert feat.shape == ref.shape
    err = (feat.cpu() - ref).abs().max().item()
    assert err <= FEAT_TOL, f'feat max err {err:.3e}'
    # second call on the same workspace: bit-identical (fixed summation order)
    assert torch.equal(eng.forward(bb.to(gpu)), feat)


@pytest.mark.parametrize('n,hb,wb', [(1, 2, 2), (1, 3, 5), (2, 7, 40), (5, 16, 18), (1, 40, 99)])
def test_neck_edge_shapes_vs_oracle(gpu, n, hb, wb):
    """Tiny, odd and ragged grids (floor(h/2) outputs; kernel 16 wider than the map)."""
    w = orc.make_neck_weights(40)
    bb = orc.make_backbone_features(41 + hb, n, hb, wb)
    ref = orc.neck(bb.double(), {k: v.double() for k, v in w.items()})
    feat = pkg.NeckEngine(w, device=gpu).forward(bb.to(gpu))
    assert feat.shape == (n, 256, hb // 2, wb // 2)
    err = (feat.cpu().double() - ref).abs().max().item()
    assert err <= FEAT_TOL, f'feat max err {err:.3e}'


@pytest.mark.parametrize('n,hb,wb', [(1, 32, 32), (3, 33, 35), (2, 9, 40), (1, 40, 99), (7, 40, 40), (1, 100, 36),
                                     (1, 6, 400)])
def test_neck_conv_kernels_agree(gpu, n, hb, wb):
    """The row-window conv kernel (each input row segment staged once per kernel row and
    x parity) against the gather kernel and the fp64 oracle: every rows-per-workgroup
    shape, tiles that start mid-row, cross image boundaries and end ragged."""
    w = orc.make_neck_weights(45)
    bb = orc.make_backbone_features(46 + hb, n, hb, wb)
    ref = orc.neck(bb.double(), {k: v.double() for k, v in w.items()})
    eng = pkg.NeckEngine(w, device=gpu)
    outs = {}
    for kind in ('gather', 'row_window', 'row_window_1w'):
        eng.set_conv_kernel(kind)
        for rows in (0, 192, 128):
            eng.set_conv_rows(rows)
            feat = eng.forward(bb.to(gpu))
            err = (feat.cpu().double() - ref).abs().max().item()
            assert err <= FEAT_TOL, f'{kind}/{rows}: feat max err {err:.3e}'
            assert torch.equal(outs.setdefault(kind, feat.clone()), feat), f'{kind}: rows={rows} changed the result'
    assert (outs['gather'] - outs['row_window']).abs().max().item() <= 1e-5
    assert torch.equal(outs['row_window'], outs['row_window_1w'])   # same step order per output


def test_neck_row_window_needs_wide_maps(gpu):
    w = orc.make_neck_weights(47)
    eng = pkg.NeckEngine(w, device=gpu)
    bb = orc.make_backbone_features(48, 1, 20, 14).to(gpu)      # output map 7 wide
    auto = eng.forward(bb)                                      # auto -> gather kernel
    eng.set_conv_kernel('row_window')
    with pytest.raises(ValueError):
        eng.forward(bb)
    eng.set_conv_kernel('gather')
    assert torch.equal(eng.forward(bb), auto)
    with pytest.raises(KeyError):
        eng.set_conv_kernel('fastest')


def test_neck_bench_size_properties(gpu):
    """16 images of 40x40 (both sides of 8 pairs at 640x640): per-image independence
    (a batch equals its images run one by one) and shift of the padding row."""
    w = orc.make_neck_weights(42)
    eng = pkg.NeckEngine(w, device=gpu)
    bb = orc.make_backbone_features(43, 16, 40, 40).to(gpu)
    feat = eng.forward(bb)
    assert torch.isfinite(feat).all()
    for i in (0, 7, 15):
        assert torch.equal(eng.forward(bb[i:i + 1].contiguous())[0], feat[i])
    ref = orc.neck(bb[3:4].cpu(), w)
    assert (feat[3:4].cpu() - ref).abs().max().item() <= FEAT_TOL


def test_neck_errors(gpu):
    w = orc.make_neck_weights(44)
    eng = pkg.NeckEngine(w, device=gpu)
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(1, 1024, 1, 8, device=gpu))
    with pytest.raises(ValueError):
        eng.forward(torch.zeros(1, 512, 8, 8, device=gpu))
    with pytest.raises(pkg.OetrError):
        eng.forward(torch.zeros(1, 1024, 8, 8))
    bad = dict(w)
    bad['patchmerging.reductions.2.weight'] = torch.zeros(128, 256, 8, 8)
    with pytest.raises(ValueError):
        pkg.NeckEngine(bad, device=gpu)


def test_module_neck_hip_vs_torch_modules(gpu):
    """OETR.neck: HIP kernels vs the torch modules holding the same weights, and
    feature_extraction's batched (same-size) vs per-image paths."""
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval().to(gpu)
    g = torch.Generator().manual_seed(8)
    im1 = torch.rand(2, 256, 320, 3, generator=g).to(gpu)
    im2 = torch.rand(2, 256, 320, 3, generator=g).to(gpu)
    bb = model.backbone(im1)
    hip = model.neck(bb)
    model.hip_neck = False
    ref = model.neck(bb)
    model.hip_neck = True
    assert hip.shape == ref.shape == (2, 256, 8, 10)
    scale = ref.abs().max().item()
    assert (hip - ref).abs().max().item() <= 2e-5 * max(1.0, scale)
    f1, f2, *_ = model.feature_extraction(im1, im2)          # one batch of 4
    f1s = model.neck(model.backbone(im1))
    assert (f1 - f1s).abs().max().item() <= 2e-5 * max(1.0, scale)
    # in-place edit of a neck weight rebuilds the engine
    with torch.no_grad():
        model.input_proj2.bias.add_(0.5)
    assert (model.neck(bb) - hip - 0.5).abs().max().item() <= 1e-5
