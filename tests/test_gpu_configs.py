"""BASELINE.json's configs at their REAL per-GPU sizes, in the driver-run GPU suite (VERDICT r2
item 3): not just the small goldens with a tile shape forced.

* configs[3]: 32 pairs @ 1024x1024 (32x32 tokens per image), where the auto rule picks
  64-token encoder workgroups and the workspace is 4x configs[1]'s - auto, 32- and 64-row
  tiles: bit-exact independence of pairs (permutation / slicing), finite in-range boxes, and
  IoU >= 1 - 1e-3 against the CPU oracle on a 4-pair slice;
* configs[4]'s per-GPU share: 8 pairs 640x640 vs 1280x1280 (400 vs 1600 tokens, cross
  attention with L != S) in the default mode and under the precision policy, same properties;
* configs[2]'s per-GPU share under the policy: 8 pairs @ 640x640.

Oracle cost bounds the compared slice (the CPU path needs ~0.5 s per 1024x1024 pair)."""
import pytest
import torch

from oracle import oetr_oracle as orc

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _engine(w, gpu, precision):
    """'prec' = every size-dependent rule automatic; 'prec@tile' = the encoder tile, the form of the
    tail (direct form, what batches of this size run) AND the decoder's workgroups per image pinned:
    with the rules pinned a pair's boxes do not depend on the batch around it, bit for bit - the
    automatic rules (64-row tiles once the grid exceeds the chip, direct tail from 16 000 token rows,
    four decoder workgroups per image up to 8 pairs) trade that for speed and change the fp32
    summation order only."""
    from imagematching_oetr_amd import HotPathEngine
    prec, _, tile = precision.partition('@')
    eng = HotPathEngine(w, device=gpu, precision=prec, enc_tile=int(tile) if tile else None)
    if tile:
        eng.set_tail_mode(2)
        eng.set_decoder_split(1)
    return eng


def _check_batch(eng, gpu, w, n, g1, g2, im1, im2, seed, n_oracle, perm_seed=0, slice_exact=True):
    f1, f2 = orc.make_features(seed, n, *g1), orc.make_features(seed + 1, n, *g2)
    p1, p2 = orc.position_table(*g1).to(gpu), orc.position_table(*g2).to(gpu)
    d1, d2 = f1.to(gpu), f2.to(gpu)
    b1, b2 = eng.forward(d1, d2, p1, p2, im1, im2)
    assert eng.query_flags() == 0
    for b, (h, wd) in ((b1, im1), (b2, im2)):
        assert b.shape == (n, 4) and torch.isfinite(b).all() and (b >= 0).all()
        assert (b[:, 0::2] <= wd).all() and (b[:, 1::2] <= h).all()
        assert (b[:, 2] >= b[:, 0]).all() and (b[:, 3] >= b[:, 1]).all()
    # pairs are independent: any permutation / slice of the batch gives the same boxes, bit for bit
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(perm_seed)).to(gpu)
    c1, c2 = eng.forward(d1[perm], d2[perm], p1, p2, im1, im2)
    assert torch.equal(c1, b1[perm]) and torch.equal(c2, b2[perm])
    lo = n // 3
    s1, s2 = eng.forward(d1[lo:lo + n_oracle].contiguous(), d2[lo:lo + n_oracle].contiguous(), p1, p2, im1, im2)
    if slice_exact:
        assert torch.equal(s1, b1[lo:lo + n_oracle]) and torch.equal(s2, b2[lo:lo + n_oracle])
    else:   # auto rules: the small slice runs 32-token tiles and the P form of the tail, the batch 64-token
        # tiles and the direct form - another fp32 summation order of the partial states / the nine taps
        assert float((s1 - b1[lo:lo + n_oracle]).abs().max()) <= 2e-2 and float((s2 - b2[lo:lo + n_oracle]).abs().max()) <= 2e-2
    # north_star bar against the CPU oracle on that slice
    r1, r2 = orc.hot_path(f1[lo:lo + n_oracle], f2[lo:lo + n_oracle], w, im1, im2)
    iou = torch.cat([orc.bbox_iou_aligned(s1.cpu(), r1), orc.bbox_iou_aligned(s2.cpu(), r2)])
    ref = torch.cat([r1, r2])
    area = (ref[:, 2] - ref[:, 0]) * (ref[:, 3] - ref[:, 1])
    assert (iou[area > 1] >= 1 - 1e-3).all(), iou
    return b1, b2


@pytest.mark.parametrize('precision', ['f32_split_f16', 'f32_split_f16@32', 'f32_split_f16@64', 'f32_split_qk16'])
def test_configs3_32_pairs_at_1024(gpu, precision):
    """BASELINE configs[3]: batch = 32 pairs, 1024x1024 -> 32x32 tokens per image, one GPU."""
    w = orc.make_hot_weights(3, sharpen=True)
    eng = _engine(w, gpu, precision)
    _check_batch(eng, gpu, w, 32, (32, 32), (32, 32), (1024, 1024), (1024, 1024), seed=510, n_oracle=4,
                 slice_exact='@' in precision)


def test_configs3_tile_shapes_agree(gpu):
    """auto (= 64-row at this size), 32- and 64-row encoder tiles: same boxes up to the fp32
    summation order of the per-tile partial states."""
    w = orc.make_hot_weights(3, sharpen=True)
    f1, f2 = orc.make_features(520, 32, 32, 32).to(gpu), orc.make_features(521, 32, 32, 32).to(gpu)
    p = orc.position_table(32, 32).to(gpu)
    boxes = {}
    for prec in ('f32_split_f16', 'f32_split_f16@32', 'f32_split_f16@64'):
        boxes[prec] = _engine(w, gpu, prec).forward(f1, f2, p, p, (1024, 1024), (1024, 1024))
    assert torch.equal(boxes['f32_split_f16'][0], boxes['f32_split_f16@64'][0])      # the auto rule at 2048 tiles
    d = float((boxes['f32_split_f16@32'][0] - boxes['f32_split_f16@64'][0]).abs().max())
    assert d <= 2e-2, d


@pytest.mark.parametrize('precision', ['f32_split_f16', 'f32_split_f16@64', 'f32_split_qk16'])
def test_configs4_mixed_scale_640_vs_1280(gpu, precision):
    """BASELINE configs[4]'s per-GPU share: 8 pairs, image1 640x640 (400 tokens) vs image2
    1280x1280 (1600 tokens), sharpened heads (boxes off the clamp)."""
    w = orc.make_hot_weights(5, sharpen=True)
    eng = _engine(w, gpu, precision)
    _check_batch(eng, gpu, w, 8, (20, 20), (40, 40), (640, 640), (1280, 1280), seed=530, n_oracle=3,
                 slice_exact='@' in precision)


def test_configs2_policy_8_pairs_at_640(gpu):
    """BASELINE configs[2]'s per-GPU share (64 pairs over 8 GPUs) under the precision policy."""
    w = orc.make_hot_weights(1, sharpen=True)
    eng = _engine(w, gpu, 'f32_split_qk16')
    _check_batch(eng, gpu, w, 8, (20, 20), (20, 20), (640, 640), (640, 640), seed=540, n_oracle=8)


@pytest.mark.parametrize('precision', ['f32_split_f16', 'f32_split_f16@64'])
def test_configs3_literal_64x64_tokens_4_pairs_at_2048(gpu, precision):
    """BASELINE configs[3]'s literal "HW = 64x64 correlation volume" (SURVEY 8d "Config 4"): 4 pairs @
    2048x2048 -> 4096 tokens per image through the linear path; pair independence bit for bit, boxes in
    range, IoU >= 1 - 1e-3 against the CPU oracle on a 1-pair slice (VERDICT r5 missing item 2)."""
    w = orc.make_hot_weights(4, sharpen=True)
    eng = _engine(w, gpu, precision)
    _check_batch(eng, gpu, w, 4, (64, 64), (64, 64), (2048, 2048), (2048, 2048), seed=550, n_oracle=1,
                 slice_exact='@' in precision)


def test_all_pairs_attention_at_L4096_properties(gpu):
    """The all-pairs kernel (reference FullAttention, src/models/linear_attention.py:53-87) at the literal
    4096 x 4096 volume, 8 heads: too large for the fp64 oracle in the suite's budget as a whole, so
    (i) a slice of queries against the oracle over ALL 4096 keys at the goldens' tolerance, (ii) softmax
    properties that do not depend on size - V = const gives that constant exactly to fp32 rounding,
    permuting the keys (with their values) changes nothing beyond the summation order, a query's
    result does not depend on the other queries bit for bit - and (iii) run-to-run determinism."""
    from imagematching_oetr_amd import full_attention
    L = 4096
    gen = torch.Generator().manual_seed(77)
    q = (torch.rand(1, L, 8, 32, generator=gen) - 0.5) * 4
    k = (torch.rand(1, L, 8, 32, generator=gen) - 0.5) * 4
    v = (torch.rand(1, L, 8, 32, generator=gen) - 0.5) * 2
    dq, dk, dv = q.to(gpu), k.to(gpu), v.to(gpu)
    out = full_attention(dq, dk, dv, variant='f32_split_f16', check_range=True)
    assert torch.isfinite(out).all()
    rows = torch.arange(0, L, 97)
    ref = orc.full_attention(q[:, rows].double(), k.double(), v.double())
    assert float((out[:, rows].cpu().double() - ref).abs().max()) <= 5e-6
    exact = full_attention(dq, dk, dv, variant='f32')
    assert float((out - exact).abs().max()) <= 5e-6
    # (ii) properties
    const = torch.full_like(dv, 0.625)
    assert float((full_attention(dq, dk, const, variant='f32_split_f16') - 0.625).abs().max()) <= 2e-6
    perm = torch.randperm(L, generator=gen).to(gpu)
    assert float((full_attention(dq, dk[:, perm].contiguous(), dv[:, perm].contiguous(), variant='f32_split_f16')
                  - out).abs().max()) <= 2e-6
    part = full_attention(dq[:, 1000:1300].contiguous(), dk, dv, variant='f32_split_f16')   # another query blocking (300 of 4096)
    assert float((part - out[:, 1000:1300]).abs().max()) <= 2e-6
    # (iii) the same launch again, bit for bit (the kernel's MFMA operands come straight out of VALU conversions)
    for _ in range(5):
        assert torch.equal(full_attention(dq, dk, dv, variant='f32_split_f16'), out)
