"""forward_dummy's optional masks through the HIP path (oetr_forward_masked,
oetr_feature_correlation_masked, oetr_center_estimation_masked) against the vectors the reference's
own modules produced with masks (tests/golden/hotmask_*.npz, oracle/gen_golden.py: MASK_CASES) and
against the oracle.  Tolerances: tests/test_gpu_parity.py.  Run with `-m gpu` on an MI355X."""
import glob
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import oetr_oracle as orc
from tests.test_gpu_parity import TOL, check_stages, maxerr
from tests.test_oracle_golden import load_hot_case
from tests.test_oracle_masks import load_masks

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
MASKED = sorted(glob.glob(str(Path(__file__).parent / 'golden' / 'hotmask_*.npz')))


def engine(gpu, seed, sharpen, tile=None, **kw):
    from imagematching_oetr_amd import HotPathEngine
    return HotPathEngine(orc.make_hot_weights(seed, sharpen=sharpen), device=gpu, enc_tile=tile, **kw)


@pytest.mark.parametrize('tile', [None, 64], ids=['tile32', 'tile64'])
@pytest.mark.parametrize('path', MASKED, ids=lambda p: p.split('hotmask_')[-1][:-4])
def test_masked_hot_path_vs_reference_golden_and_oracle(path, tile, gpu):
    g, w, f1, f2 = load_hot_case(path)
    m1, m2 = load_masks(g)
    im1, im2 = tuple(int(v) for v in g['img1']), tuple(int(v) for v in g['img2'])
    p1, p2 = orc.position_table(*g['grid1']), orc.position_table(*g['grid2'])
    eng = engine(gpu, int(g['weight_seed']), bool(g['sharpen']), tile)
    dev = [t.to(gpu) for t in (f1, f2, p1, p2)]
    out = eng.forward(*dev, im1, im2, stages=True, mask1=m1, mask2=m2)
    ref = orc.hot_path(f1, f2, w, im1, im2, return_stages=True, mask1=m1, mask2=m2)
    assert eng.query_flags() == 0
    # masked logits are the fill value on both sides: compare them exactly, the rest within tolerance
    for s, m in (('1', m1), ('2', m2)):
        dead = (m.flatten(1) == 0).to(gpu)
        assert (out['logits' + s][dead] == orc.MASK_FILL).all(), 'masked logits must hold -1e9'
        assert (out['logits' + s][~dead] > -1e8).all()
    check_stages(out, ref, 'vs oracle', 'hotmask', 'f32_split_f16')
    # ... and against what the reference itself produced
    for s in ('1', '2'):
        step = int(g[f'memory{s}_step'])
        assert maxerr(out['memory' + s][:, ::step], g['memory' + s]) <= TOL['memory']
        assert maxerr(out['hs' + s], g['hs' + s]) <= TOL['hs']
        assert maxerr(out['logits' + s], g['logits' + s]) <= TOL['logits']
        assert maxerr(out['cxy' + s], g['cxy' + s]) <= TOL['cxy']
        assert maxerr(out['tlbr' + s], g['tlbr' + s]) <= TOL['tlbr']
        assert maxerr(out['box' + s], g['box' + s]) <= TOL['box']
        iou = orc.bbox_iou_aligned(out['box' + s].cpu(), torch.from_numpy(g['box' + s]))
        assert (iou >= 1 - 1e-3).all(), iou
    # encoder prefixes: after layer 0 (self) and layer 1 (cross)
    for li in (0, 1):
        pre = eng.forward(*dev, im1, im2, stages=True, enc_layers=li + 1, mask1=m1, mask2=m2)
        for s in ('1', '2'):
            step = int(g[f'enc{li}_x{s}_step'])
            e = maxerr(pre['memory' + s][:, ::step], g[f'enc{li}_x{s}'])
            assert e <= TOL['memory'], f'enc{li} x{s}: {e:.3e}'
    # plain masked forward == staged masked forward, bit for bit, repeatable, and the masks matter
    b1, b2 = eng.forward(*dev, im1, im2, mask1=m1, mask2=m2)
    assert torch.equal(b1, out['box1']) and torch.equal(b2, out['box2'])
    b1b, _ = eng.forward(*dev, im1, im2, mask1=m1.bool(), mask2=m2.to(torch.uint8))   # any dtype, like the reference
    assert torch.equal(b1, b1b)
    plain = eng.forward(*dev, im1, im2, stages=True)
    assert maxerr(plain['hs1'], out['hs1']) > 1e-3


def test_all_ones_masks_change_nothing(gpu):
    """mask == 1 everywhere multiplies by one and fills nothing: the masked kernels must return what
    the unmasked ones do, bit for bit (same operations in the same order), in both tile sizes."""
    f1, f2 = orc.make_features(31, 3, 12, 17), orc.make_features(32, 3, 9, 30)
    p1, p2 = orc.position_table(12, 17), orc.position_table(9, 30)
    dev = [t.to(gpu) for t in (f1, f2, p1, p2)]
    for tile in (None, 64):
        eng = engine(gpu, 1, True, tile)
        a = eng.forward(*dev, (384, 544), (288, 960), stages=True)
        b = eng.forward(*dev, (384, 544), (288, 960), stages=True,
                        mask1=torch.ones(3, 12, 17), mask2=torch.ones(3, 9, 30))
        for k in a:
            assert torch.equal(a[k], b[k]), (tile, k)


def test_masked_seams_and_module(gpu):
    """feature_correlation / center_estimation with masks (reference src/model.py:132-186) and the
    drop-in module's forward_dummy(image1, image2, mask1, mask2) end to end vs the oracle fed with the
    module's own features."""
    import imagematching_oetr_amd as pkg
    w = orc.make_hot_weights(2, sharpen=True)
    eng = engine(gpu, 2, True)
    f1, f2 = orc.make_features(41, 2, 10, 14), orc.make_features(42, 2, 16, 9)
    p1, p2 = orc.position_table(10, 14), orc.position_table(16, 9)
    m1, m2 = orc.make_masks(43, 2, 10, 14, 'holes'), orc.make_masks(44, 2, 16, 9, 'pad')
    im1, im2 = (320, 448), (512, 288)
    ref = orc.hot_path(f1, f2, w, im1, im2, return_stages=True, mask1=m1, mask2=m2)
    dev = [t.to(gpu) for t in (f1, f2, p1, p2)]
    hs1, hs2, mem1, mem2 = eng.feature_correlation(*dev, mask1=m1, mask2=m2)
    assert maxerr(mem1, ref['memory1']) <= TOL['memory'] and maxerr(mem2, ref['memory2']) <= TOL['memory']
    assert maxerr(hs1, ref['hs1']) <= TOL['hs'] and maxerr(hs2, ref['hs2']) <= TOL['hs']
    c1, c2 = eng.center_estimation(hs1, hs2, mem1, mem2, 10, 14, 16, 9, im1[0], im2[0], mask1=m1, mask2=m2)
    assert maxerr(c1, ref['cxy1']) <= TOL['cxy'] and maxerr(c2, ref['cxy2']) <= TOL['cxy']
    # the masks only fill logits in center_estimation: any dtype of handle takes them there
    e32 = engine(gpu, 2, True, precision='f32')
    c1f, _ = e32.center_estimation(hs1, hs2, mem1, mem2, 10, 14, 16, 9, im1[0], im2[0], mask1=m1, mask2=m2)
    assert maxerr(c1f, ref['cxy1']) <= TOL['cxy']
    # errors: one mask only, wrong shape, a dtype the encoder kernels are not built for
    with pytest.raises(ValueError, match='both'):
        eng.forward(*dev, im1, im2, mask1=m1)
    with pytest.raises(ValueError, match='elements'):
        eng.forward(*dev, im1, im2, mask1=m1, mask2=m1)
    with pytest.raises(pkg.hip_engine.OetrError, match='masks'):
        engine(gpu, 2, True, precision='bf16').forward(*dev, im1, im2, mask1=m1, mask2=m2)
    # the precision policy (Q / K / decoder-K on single f16 MFMAs) carries them too: inside the IoU bar
    for tile in (None, 64):
        pol = engine(gpu, 2, True, tile, precision='f32_split_qk16').forward(*dev, im1, im2, stages=True, mask1=m1, mask2=m2)
        for s_ in ('1', '2'):
            assert maxerr(pol['memory' + s_], ref['memory' + s_]) <= 5e-3 and maxerr(pol['hs' + s_], ref['hs' + s_]) <= 5e-3
            assert (orc.bbox_iou_aligned(pol['box' + s_].cpu(), ref['box' + s_]) >= 1 - 1e-3).all()
    with pytest.raises(pkg.hip_engine.OetrError, match='masks'):
        engine(gpu, 2, True, attention='full').forward(*dev, im1, im2, mask1=m1, mask2=m2)
    # the exact-fp32 build carries the masks too (the re-run route of a masked batch that overflowed f16)
    out32 = e32.forward(*dev, im1, im2, stages=True, mask1=m1, mask2=m2)
    check_stages(out32, ref, 'exact fp32 vs oracle', 'hotmask', 'f32')

    # drop-in module: forward_dummy with masks = the reference's signature
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    sd.update(orc.make_hot_weights(5, sharpen=True))
    model.load_state_dict(sd)
    model = model.to(gpu)
    g = torch.Generator().manual_seed(6)
    img1, img2 = torch.rand(2, 256, 320, 3, generator=g).to(gpu), torch.rand(2, 192, 256, 3, generator=g).to(gpu)
    feat1, feat2, pos1, pos2, hf1, wf1, hf2, wf2 = model.feature_extraction(img1, img2)
    mm1, mm2 = orc.make_masks(45, 2, hf1, wf1), orc.make_masks(46, 2, hf2, wf2)
    b1, b2 = model.forward_dummy(img1, img2, mm1.to(gpu), mm2.to(gpu))
    model.hip_flush()
    r1, r2 = orc.hot_path(feat1.cpu(), feat2.cpu(), orc.make_hot_weights(5, sharpen=True), (256, 320), (192, 256),
                          mask1=mm1, mask2=mm2)
    iou = orc.bbox_iou_aligned(torch.cat([b1, b2]).cpu(), torch.cat([r1, r2]))
    assert (iou >= 1 - 1e-3).all(), iou
    u1, _ = model.forward_dummy(img1, img2)
    model.hip_flush()
    assert maxerr(u1, b1) > 1e-2        # the masks moved the boxes
    # an operand out of the f16 range in a masked batch: re-run in exact fp32 WITH the masks
    big1, big2 = feat1.clone(), feat2.clone()
    big1[0, 3, 1, 1] = 3.0e5
    o1, o2 = model.boxes_from_features(big1, big2, pos1, pos2, (256, 320), (192, 256), mm1, mm2)
    model.hip_flush()
    q1, q2 = orc.hot_path(big1.cpu(), big2.cpu(), orc.make_hot_weights(5, sharpen=True), (256, 320), (192, 256),
                          mask1=mm1, mask2=mm2)
    got, want = torch.cat([o1, o2]).cpu(), torch.cat([q1, q2])
    assert maxerr(got, want) <= TOL['box'], (got, want)
    area = (want[:, 2] - want[:, 0]) * (want[:, 3] - want[:, 1])
    iou = orc.bbox_iou_aligned(got, want)
    assert (iou[area > 1] >= 1 - 1e-3).all(), iou      # (the outlier may pin image 0's box to a clamped line)


def test_masked_linear_attention_entry_vs_reference_golden(gpu, golden_dir):
    """oetr_linear_attention_masked (LinearAttention.forward with q_mask / kv_mask) against the
    reference's own outputs; either mask alone; no mask = the unmasked entry bit for bit."""
    from imagematching_oetr_amd import hip_engine
    g = np.load(golden_dir / 'attention_masked.npz')
    for (L, S) in g['cases']:
        tag = f'L{L}_S{S}'
        gen = torch.Generator().manual_seed(int(g[tag + '_seed']))
        q = (torch.rand(2, L, 8, 32, generator=gen) - 0.5) * 4
        k = (torch.rand(2, S, 8, 32, generator=gen) - 0.5) * 4
        v = (torch.rand(2, S, 8, 32, generator=gen) - 0.5) * 2
        qm = (torch.rand(2, L, generator=gen) >= 0.25).float()
        km = (torch.rand(2, S, generator=gen) >= 0.25).float()
        km[0] = 1.0
        dq, dk, dv = q.to(gpu), k.to(gpu), v.to(gpu)
        out = hip_engine.linear_attention(dq, dk, dv, q_mask=qm, kv_mask=km).reshape(2, L, 256)
        step = int(g[tag + '_step'])
        ref = torch.from_numpy(g[tag + '_lin'])
        err = (out[:, ::step].cpu() - ref).abs()
        assert (err <= 2e-5 + 1e-5 * ref.abs()).all(), f'{tag}: {float(err.max()):.3e}'
        assert (out[qm.to(gpu) == 0] == 0).all()
        only_q = hip_engine.linear_attention(dq, dk, dv, q_mask=qm)
        full = orc.linear_attention(q, k, v, q_mask=qm)
        assert maxerr(only_q, full) <= 2e-5 * max(1.0, float(full.abs().max()))
        only_kv = hip_engine.linear_attention(dq, dk, dv, kv_mask=km.bool())
        fullk = orc.linear_attention(q, k, v, kv_mask=km)
        assert maxerr(only_kv, fullk) <= 2e-5 * max(1.0, float(fullk.abs().max()))
        ones = hip_engine.linear_attention(dq, dk, dv, q_mask=torch.ones(2, L), kv_mask=torch.ones(2, S))
        assert torch.equal(ones, hip_engine.linear_attention(dq, dk, dv))


def test_fractional_and_empty_masks(gpu):
    """The reference only MULTIPLIES by the masks in the attention (any float value, linear_attention.py:37-41)
    and fills logits where ``bool(mask)`` is False (model.py:166-171): fractional weights scale, zeros fill;
    an image masked everywhere gets the uniform softmax (centre of the grid) and a zero state."""
    w = orc.make_hot_weights(3, sharpen=True)
    f1, f2 = orc.make_features(51, 3, 9, 11), orc.make_features(52, 3, 14, 6)
    p1, p2 = orc.position_table(9, 11), orc.position_table(14, 6)
    g = torch.Generator().manual_seed(9)
    m1 = torch.rand(3, 9, 11, generator=g) * (torch.rand(3, 9, 11, generator=g) > 0.2)     # weights in [0,1), ~20 % zeros
    m2 = torch.rand(3, 14, 6, generator=g).clamp_min(0.05)
    m1[2] = 0.0                                                                              # image 2 of side 1: nothing valid
    im1, im2 = (288, 352), (448, 192)
    dev = [t.to(gpu) for t in (f1, f2, p1, p2)]
    ref = orc.hot_path(f1, f2, w, im1, im2, return_stages=True, mask1=m1, mask2=m2)
    for tile in (None, 64):
        out = engine(gpu, 3, True, tile).forward(*dev, im1, im2, stages=True, mask1=m1, mask2=m2)
        check_stages(out, ref, f'weighted masks, tile {tile}', 'masks_9x11_14x6', 'f32_split_f16')
        assert torch.isfinite(out['memory1']).all() and torch.isfinite(out['hs1']).all()
        # the empty image: uniform softmax -> centre of its token grid times the stride (32)
        assert maxerr(out['cxy1'][2], torch.tensor([11 * 32 / 2.0, 9 * 32 / 2.0])) <= 1e-3


def test_masked_forward_is_enqueue_only(gpu):
    """oetr_forward_masked like oetr_forward: no synchronisation, no device-to-host copy - capturable into a HIP
    graph; a replay after the MASKS changed in place returns what an eager call on the new masks returns."""
    eng = engine(gpu, 1, True)
    f1, f2 = orc.make_features(61, 2, 12, 12).to(gpu), orc.make_features(62, 2, 8, 15).to(gpu)
    p1, p2 = orc.position_table(12, 12).to(gpu), orc.position_table(8, 15).to(gpu)
    m1 = orc.make_masks(63, 2, 12, 12, 'pad').to(gpu).flatten(1).contiguous()     # float32 [N,L] on the device: no
    m2 = orc.make_masks(64, 2, 8, 15, 'pad').to(gpu).flatten(1).contiguous()       # conversion inside the capture
    first = [t.clone() for t in eng.forward(f1, f2, p1, p2, (384, 384), (256, 480), mask1=m1, mask2=m2)]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = eng.forward(f1, f2, p1, p2, (384, 384), (256, 480), mask1=m1, mask2=m2)
    m1.copy_(orc.make_masks(65, 2, 12, 12, 'holes').flatten(1))
    graph.replay()
    torch.cuda.synchronize()
    fresh = eng.forward(f1, f2, p1, p2, (384, 384), (256, 480), mask1=m1, mask2=m2)
    assert torch.equal(captured[0], fresh[0]) and torch.equal(captured[1], fresh[1])
    assert not torch.equal(fresh[0], first[0])
    assert eng.query_flags() == 0


@pytest.mark.parametrize('tile', [None, 64], ids=['tile32', 'tile64'])
def test_masked_forward_at_the_benchmark_size(gpu, tile):
    """BASELINE configs[1]'s shape (8 pairs @640x640: 20x20 token grids, ragged last tiles) with padding-style masks."""
    w = orc.make_hot_weights(5, sharpen=True)
    f1, f2 = orc.make_features(81, 8, 20, 20), orc.make_features(82, 8, 20, 20)
    p1 = orc.position_table(20, 20)
    m1, m2 = orc.make_masks(83, 8, 20, 20, 'pad'), orc.make_masks(84, 8, 20, 20, 'holes')
    ref = orc.hot_path(f1, f2, w, (640, 640), (640, 640), return_stages=True, mask1=m1, mask2=m2)
    out = engine(gpu, 5, True, tile).forward(f1.to(gpu), f2.to(gpu), p1.to(gpu), p1.to(gpu), (640, 640), (640, 640),
                                             stages=True, mask1=m1, mask2=m2)
    check_stages(out, ref, f'8 pairs @640, tile {tile}', 'masks_8p_640', 'f32_split_f16')
