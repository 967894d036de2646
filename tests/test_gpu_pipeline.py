"""GPU tests of the two "next" rows either side of the hot path:

* the device-side box -> crop step (``oetr_overlap_crop``) against the goldens the
  reference's own functions produced (gate, scaled boxes, ratios, shapes: exact) and
  against the oracle's restatement of OpenCV's bicubic (pixels; OpenCV itself is
  parity-unpinned - cv2 is not installed);
* the batched pair front-end (``forward_pairs``) on the real model: a mixed list of
  640x640 / 640x1280 / 480x640 pairs must give, in input order, the boxes of the
  reference's per-pair loop.
"""
import numpy as np
import pytest
import torch

import imagematching_oetr_amd as pkg
from oracle import crop_oracle as cro
from oracle import oetr_oracle as orc
from tests.test_crop_cpu import load_cases

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
PIX_TOL = 2e-6      # crops live in [0,1]; same algorithm, fp32, different FMA contraction


def test_crop_step_matches_reference_goldens(gpu, golden_dir):
    n = 0
    for ci, c, im0, im1 in load_cases(golden_dir):
        res = pkg.overlap_crop(im0.to(gpu), im1.to(gpu), torch.from_numpy(c['box0']).to(gpu),
                               torch.from_numpy(c['box1']).to(gpu), tuple(c['scales0']),
                               tuple(c['scales1']), keep_aspect=bool(c['keep_aspect']),
                               size_divisor=int(c['size_divisor']), pragueparks=bool(c['pragueparks']))
        assert res.valid == bool(c['valid']), ci
        for i, s in ((0, '0'), (1, '1')):
            assert np.array_equal(res.bbox(i).numpy().reshape(-1), c['bbox' + s]), (ci, i)
            assert np.array_equal(np.float32(res.ratio(i)).astype(np.float64).reshape(-1), c['ratio' + s]), (ci, i)
            crop = res.crop(i)
            assert tuple(crop.shape) == tuple(c['out_shape' + s]), (ci, i)
            if 'crop' + s in c:
                err = float((crop.cpu() - torch.from_numpy(c['crop' + s])).abs().max())
                assert err <= PIX_TOL, (ci, i, err)
        # all cases: pixels vs the oracle
        ref = cro.overlap_crop(im0, im1, torch.from_numpy(c['box0']), torch.from_numpy(c['box1']),
                               tuple(c['scales0']), tuple(c['scales1']), bool(c['keep_aspect']),
                               int(c['size_divisor']), bool(c['pragueparks']))
        assert float((res.crop(0).cpu() - ref['crop0']).abs().max()) <= PIX_TOL
        assert float((res.crop(1).cpu() - ref['crop1']).abs().max()) <= PIX_TOL
        if not res.valid:      # gate failed: the reference hands the images back untouched - bit for bit
            assert torch.equal(res.crop(0).cpu(), im0) and torch.equal(res.crop(1).cpu(), im1), ci
        n += 1
    assert n == 8


def test_degenerate_crop_is_reported_not_passed_through(gpu):
    """A box past the image border leaves an empty slice: the reference raises from cv2.resize
    there; here ``oetr_crop_info.valid == -1`` (distinct from a failed gate) and the Python
    wrapper raises instead of handing back an empty crop."""
    im = torch.rand(1, 1, 64, 64, generator=torch.Generator().manual_seed(1)).to(gpu)
    inside = torch.tensor([[5.0, 5.0, 60.0, 60.0]], device=gpu)
    outside = torch.tensor([[70.0, 10.0, 100.0, 50.0]], device=gpu)      # x1 beyond the 64-px width
    res = pkg.overlap_crop(im, im, outside, inside, (1, 1), (1, 1))
    assert int(res.geometry().valid) == -1
    with pytest.raises(pkg.OetrError):
        res.valid


def test_crop_geometry_fuzz_is_bit_exact_vs_oracle(gpu):
    """Random boxes / scales / sizes / modes: every integer, ratio and the gate equal to
    the oracle's (which is pinned to the reference by tests/golden/crop.npz)."""
    import random
    rng = random.Random(11)
    im_cache = {}
    for case in range(60):
        hw0 = (rng.randrange(40, 200), rng.randrange(40, 200))
        hw1 = (rng.randrange(40, 200), rng.randrange(40, 200))
        for hw in (hw0, hw1):
            if hw not in im_cache:
                im_cache[hw] = torch.rand(1, 1, *hw, generator=torch.Generator().manual_seed(hw[0] * 1000 + hw[1]))
        sc0 = (rng.choice([1.0, 0.75, 0.3, 1.6]), rng.choice([1.0, 0.6, 0.25, 1.3]))
        sc1 = (rng.choice([1.0, 0.75, 0.3, 1.6]), rng.choice([1.0, 0.6, 0.25, 1.3]))

        def box(hw, sc):
            w, h = hw[1] / sc[0], hw[0] / sc[1]          # OETR-frame size that maps onto the image
            x1, y1 = rng.uniform(0, w * 0.7), rng.uniform(0, h * 0.7)
            return torch.tensor([x1, y1, x1 + rng.uniform(0.5, w * 0.6), y1 + rng.uniform(0.5, h * 0.6)])
        b0, b1 = box(hw0, sc0), box(hw1, sc1)
        keep, div, pp = rng.random() < 0.7, rng.choice([1, 1, 8]), rng.random() < 0.3
        ref = cro.overlap_crop(im_cache[hw0], im_cache[hw1], b0, b1, sc0, sc1, keep, div, pp)
        res = pkg.overlap_crop(im_cache[hw0].to(gpu), im_cache[hw1].to(gpu), b0.to(gpu), b1.to(gpu),
                               sc0, sc1, keep, div, pp)
        assert res.valid == ref['valid'], case
        for i, s in ((0, '0'), (1, '1')):
            assert torch.equal(res.bbox(i).reshape(-1), ref['bbox' + s].float().reshape(-1)), (case, i)
            assert res.ratio(i)[0] == list(ref['ratio' + s]), (case, i)
            assert tuple(res.crop(i).shape) == tuple(ref['crop' + s].shape), (case, i)
            assert float((res.crop(i).cpu() - ref['crop' + s]).abs().max()) <= PIX_TOL, (case, i)


def test_crop_step_is_enqueue_only(gpu):
    """No device-to-host copy or synchronisation inside the call: it can be captured in a
    HIP graph together with the hot path that produced the boxes, and replayed."""
    g0 = torch.Generator().manual_seed(3)
    im0, im1 = torch.rand(1, 3, 96, 128, generator=g0).to(gpu), torch.rand(1, 3, 80, 112, generator=g0).to(gpu)
    b0 = torch.tensor([[10.0, 12.0, 100.0, 80.0]], device=gpu)
    b1 = torch.tensor([[5.0, 6.0, 90.0, 70.0]], device=gpu)
    eager = pkg.overlap_crop(im0, im1, b0, b1, (1, 1), (1, 1), True, 8)
    want0 = eager.crop(0).clone()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = pkg.overlap_crop(im0, im1, b0, b1, (1, 1), (1, 1), True, 8)
    b0.copy_(torch.tensor([[20.0, 12.0, 100.0, 80.0]]))       # new boxes, same graph
    graph.replay()
    torch.cuda.synchronize()
    fresh = pkg.overlap_crop(im0, im1, b0, b1, (1, 1), (1, 1), True, 8)
    assert torch.equal(captured.crop(0), fresh.crop(0)) and not torch.equal(fresh.crop(0), want0)


def test_crop_argument_errors(gpu):
    im = torch.rand(1, 1, 32, 32, device=gpu)
    b = torch.tensor([[1.0, 1.0, 20.0, 20.0]], device=gpu)
    with pytest.raises(ValueError):
        pkg.overlap_crop(im, torch.rand(1, 3, 32, 32, device=gpu), b, b, (1, 1), (1, 1))
    with pytest.raises(pkg.OetrError):
        pkg.overlap_crop(im.cpu(), im, b, b, (1, 1), (1, 1))
    with pytest.raises(ValueError):
        pkg.overlap_crop(im, im, b, b, (1, 1), (1, 1), size_divisor=0)


def test_forward_pairs_equals_the_per_pair_loop_on_the_real_model(gpu):
    """Mixed 640x640 / 640x1280 / 480x640 pairs through trunk + HIP neck + HIP hot path
    in shape buckets: entry i must be what forward_dummy returns for pair i alone
    (reference evaluation.py:77-80).  The HIP stages are batch-invariant bit for bit; the
    torch/MIOpen trunk may pick another convolution algorithm for another batch size, so
    the end-to-end comparison allows 0.05 px (the box tolerance of the parity tests)
    while the HIP part is checked bit-exactly on identical features."""
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    sd.update(orc.make_hot_weights(5, sharpen=True))
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    g = torch.Generator().manual_seed(9)
    sizes = [((640, 640), (640, 640)), ((640, 1280), (640, 640)), ((480, 640), (480, 640)),
             ((640, 640), (640, 640)), ((640, 1280), (640, 640)), ((640, 640), (640, 640)),
             ((480, 640), (480, 640)), ((640, 640), (640, 640))]
    pairs = [(torch.rand(1, *a, 3, generator=g), torch.rand(*b, 3, generator=g)) for a, b in sizes]
    b0, b1 = pkg.forward_pairs(model, pairs, max_batch=8)
    assert b0.shape == (len(pairs), 4) and b0.device.type == 'cuda'
    exact = True
    for i, (a, b) in enumerate(pairs):
        e0, e1 = model.forward_dummy(a.to(gpu), b[None].to(gpu))
        assert float((b0[i] - e0[0]).abs().max()) <= 5e-2 and float((b1[i] - e1[0]).abs().max()) <= 5e-2, i
        exact &= torch.equal(b0[i], e0[0]) and torch.equal(b1[i], e1[0])
    print('forward_pairs bit-equal to the per-pair loop end to end:', exact)
    # the HIP part on identical features: batched == one by one, bit for bit
    idx = [i for i, s in enumerate(sizes) if s == sizes[0]]
    im0 = torch.cat([pairs[i][0] for i in idx]).to(gpu)
    im1 = torch.cat([pairs[i][1][None] for i in idx]).to(gpu)
    f0, f1, p0, p1, *_ = model.feature_extraction(im0, im1)
    full = model.boxes_from_features(f0, f1, p0, p1, (640, 640), (640, 640))
    for j in range(len(idx)):
        one = model.boxes_from_features(f0[j:j + 1].contiguous(), f1[j:j + 1].contiguous(), p0, p1,
                                        (640, 640), (640, 640))
        assert torch.equal(one[0][0], full[0][j]) and torch.equal(one[1][0], full[1][j])


def test_neck_stores_tokens_straight_into_the_hot_path(gpu):
    """forward_dummy's fused route (trunk -> ``oetr_neck_forward_tokens`` into the
    workspace -> ``oetr_forward_tokens``: no NCHW feat, no transpose launch) against the
    stepwise route on the SAME trunk output: bit-identical boxes, for equal and unequal
    image sizes, across shape changes on one workspace (the cached token-major position
    tables must be reloaded) and interleaved with the ordinary entry."""
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    sd.update(orc.make_hot_weights(6, sharpen=True))
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    g = torch.Generator().manual_seed(10)
    eng, neck = model.engine(), model.neck_engine()

    def stepwise(bb1, bb2, hw1, hw2):
        f1, f2 = neck.forward(bb1), neck.forward(bb2)
        return eng.forward(f1, f2, model.pos_encoding(f1), model.pos_encoding(f2), hw1, hw2)
    for (h1, w1), (h2, w2), n in (((640, 640), (640, 640), 3), ((640, 1280), (640, 640), 2),
                                  ((640, 640), (640, 640), 3), ((480, 640), (480, 640), 1)):
        im1 = torch.rand(n, h1, w1, 3, generator=g).to(gpu)
        im2 = torch.rand(n, h2, w2, 3, generator=g).to(gpu)
        if (h1, w1) == (h2, w2):
            bb = model.backbone(torch.cat([im1, im2]))
            bb1, bb2, both = bb[:n], bb[n:], bb
        else:
            bb1, bb2, both = model.backbone(im1), model.backbone(im2), None
        want = stepwise(bb1.contiguous(), bb2.contiguous(), (h1, w1), (h2, w2))
        got = model.boxes_from_backbone(bb1, bb2, (h1, w1), (h2, w2), both=both)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), ((h1, w1), (h2, w2))
        again = model.boxes_from_backbone(bb1, bb2, (h1, w1), (h2, w2), both=both)   # tables cached
        assert torch.equal(again[0], want[0]) and torch.equal(again[1], want[1])
    # every arithmetic / attention mode of the hot path takes the resident tokens the same way
    im1 = torch.rand(2, 480, 640, 3, generator=g).to(gpu)
    im2 = torch.rand(2, 480, 640, 3, generator=g).to(gpu)
    bb = model.backbone(torch.cat([im1, im2]))
    for prec, attn in (('f32', 'linear'), ('bf16', 'linear'), ('f16', 'linear'), ('f32_split_f16', 'full')):
        model.hip_precision, model.hip_attention = prec, attn
        model.invalidate_engine()
        eng, neck = model.engine(), model.neck_engine()
        want = stepwise(bb[:2].contiguous(), bb[2:].contiguous(), (480, 640), (480, 640))
        got = model.boxes_from_backbone(bb[:2], bb[2:], (480, 640), (480, 640), both=bb)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (prec, attn)
    model.hip_precision, model.hip_attention = 'f32_split_f16', 'linear'
    model.invalidate_engine()
    eng, neck = model.engine(), model.neck_engine()
    # the neck's token-major store is its NCHW result transposed, bit for bit
    bb = model.backbone(torch.rand(2, 480, 640, 3, generator=g).to(gpu))
    feat = neck.forward(bb)
    tok = torch.empty(2 * feat.shape[2] * feat.shape[3], 256, device=gpu)
    neck.forward_tokens(bb, tok)
    assert torch.equal(tok.view(2, -1, 256), feat.flatten(2).permute(0, 2, 1))
    with pytest.raises(ValueError):
        neck.forward_tokens(bb, tok[:-1])
    # the public switch
    im = torch.rand(2, 640, 640, 3, generator=g).to(gpu)
    fused = model.forward_dummy(im, im.flip(0))
    model.hip_fuse_neck = False
    plain = model.forward_dummy(im, im.flip(0))
    # (two trunk runs: MIOpen's convolutions are not run-to-run bit-stable; box tolerance)
    assert float((fused[0] - plain[0]).abs().max()) <= 5e-2 and float((fused[1] - plain[1]).abs().max()) <= 5e-2
    # a tripped range flag sends the fused route through the stepwise one (here: raise)
    model.hip_fuse_neck, model.hip_on_overflow = True, 'raise'
    bb = model.backbone(torch.cat([im, im]))
    model.boxes_from_backbone(bb[:2] * 1e6, bb[2:] * 1e6, (640, 640), (640, 640))   # enqueue-only ...
    with pytest.raises(pkg.OetrRangeError):
        model.hip_flush()                                                            # ... reported here
    model.hip_defer_check = False
    with pytest.raises(pkg.OetrRangeError):
        model.boxes_from_backbone(bb[:2] * 1e6, bb[2:] * 1e6, (640, 640), (640, 640))


def test_status_word_is_published_by_the_last_kernel(gpu):
    """ABI 6 (``oetr_forward_flagslot`` / ``oetr_forward_tokens_flagslot`` /
    ``oetr_neck_forward_tokens_status``): the forward call's last kernel moves the workspace's
    status word into a pinned host word and leaves 0 behind - same word ``oetr_query_flags``
    would have read, same boxes as the plain call, nothing else enqueued."""
    w = orc.make_hot_weights(7, sharpen=True)
    eng = pkg.HotPathEngine(w, device=gpu)
    f1, f2 = orc.make_features(70, 2, 8, 8).to(gpu), orc.make_features(71, 2, 5, 7).to(gpu)
    p1, p2 = orc.position_table(8, 8).to(gpu), orc.position_table(5, 7).to(gpu)
    want = eng.forward(f1, f2, p1, p2, (256, 256), (160, 224))
    assert eng.query_flags() == 0
    got, ticket = eng.forward(f1, f2, p1, p2, (256, 256), (160, 224), publish=True)
    assert ticket.value() == 0
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    # an operand beyond the f16 range: the bit arrives in the slot and the device word is clear again
    _, ticket = eng.forward(f1 * 1e6, f2, p1, p2, (256, 256), (160, 224), publish=True)
    assert ticket.value() & pkg.hip_engine.FLAG_F16_RANGE
    assert eng.query_flags(clear=False) == 0
    # ... and a clean call behind it reports clean (the slot ring moved on, the word was reset)
    got, ticket = eng.forward(f1, f2, p1, p2, (256, 256), (160, 224), publish=True)
    assert ticket.value() == 0 and torch.equal(got[0], want[0])
    # masks take the same entry
    m1, m2 = orc.make_masks(73, 2, 8, 8, 'holes'), orc.make_masks(74, 2, 5, 7, 'pad')
    wm = eng.forward(f1, f2, p1, p2, (256, 256), (160, 224), mask1=m1, mask2=m2)
    gm, ticket = eng.forward(f1, f2, p1, p2, (256, 256), (160, 224), mask1=m1, mask2=m2, publish=True)
    assert ticket.value() == 0 and torch.equal(gm[0], wm[0]) and torch.equal(gm[1], wm[1])
    with pytest.raises(ValueError):
        eng.forward(f1, f2, p1, p2, (256, 256), (160, 224), stages=True, publish=True)
    # neck -> tokens: the neck's range bit lands in the hot-path workspace's word, one slot for both stages
    neck = pkg.NeckEngine(orc.make_neck_weights(8), device=gpu)
    bb = orc.make_backbone_features(72, 4, 10, 12).to(gpu)
    for scale, bit in ((1.0, 0), (1e6, pkg.hip_engine.FLAG_F16_RANGE)):
        bufs = eng.token_buffers(2, 5, 6, 5, 6)
        eng.load_pos_tokens(bufs, orc.position_table(5, 6).to(gpu), orc.position_table(5, 6).to(gpu))
        neck.forward_tokens(bb * scale, bufs['tokens'], status_word=eng._current_ws())
        assert neck.query_flags() == 0                      # nothing went into the neck's own word
        _, ticket = eng.forward_tokens(2, 5, 6, 5, 6, (160, 192), (160, 192), publish=True)
        assert ticket.value() & pkg.hip_engine.FLAG_F16_RANGE == bit
        assert eng.query_flags(clear=False) == 0


def test_forward_dummy_to_crop_is_one_hip_graph_with_the_default_guard(gpu):
    """with the DEFAULT settings (hip_on_overflow='f32', deferred check)
    ``forward_dummy -> overlap_crop`` is enqueue-only - one batch captured into a single HIP
    graph (trunk, HIP neck, HIP hot path, the asynchronous status reads, the crop step) and
    replayed; an injected out-of-range batch afterwards is still caught and re-run one call
    later.  The boxes feed the crop with no host logic in between (reference
    evaluation.py:77-113 reads them back to the CPU at that point)."""
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    sd.update(orc.make_hot_weights(6, sharpen=True))
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    assert model.hip_on_overflow == 'f32' and model.hip_defer_check
    g = torch.Generator().manual_seed(12)
    im0 = torch.rand(1, 320, 320, 3, generator=g).to(gpu)
    im1 = torch.rand(1, 320, 320, 3, generator=g).to(gpu)
    m0 = torch.rand(1, 1, 240, 320, generator=g).to(gpu)      # the matcher's (grayscale) images
    m1 = torch.rand(1, 1, 240, 320, generator=g).to(gpu)

    def step():
        b0, b1 = model.forward_dummy(im0, im1)
        return b0, b1, pkg.overlap_crop(m0, m1, b0, b1, (1.0, 0.75), (1.0, 0.75), True, 8)
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):                 # warm-up: engines, MIOpen algorithm choices
        for _ in range(3):
            e0, e1, ecrop = step()
        model.hip_flush()
    torch.cuda.current_stream(gpu).wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        c0, c1, ccrop = step()
    assert model._pending is None and len(model._graph_tickets) >= 1
    graph.replay()
    torch.cuda.synchronize()
    model.hip_graph_check()                       # in range: no complaint
    first = (c0.clone(), c1.clone(), ccrop.crop(0).clone())
    graph.replay()
    torch.cuda.synchronize()
    # (the torch / MIOpen trunk inside the graph is not run-to-run bit-stable - atomics in its
    #  convolutions: 1e-4 px between replays - so the whole-graph comparison carries the box
    #  tolerance; the HIP part alone is captured below and must replay bit for bit)
    assert float((c0 - first[0]).abs().max()) <= 5e-2 and float((c1 - first[1]).abs().max()) <= 5e-2
    bb_static = model.backbone(torch.cat([im0, im1]))

    def hip_step():
        b0, b1 = model.boxes_from_backbone(bb_static[:1], bb_static[1:], (320, 320), (320, 320), both=bb_static)
        return b0, b1, pkg.overlap_crop(m0, m1, b0, b1, (1.0, 0.75), (1.0, 0.75), True, 8)
    with torch.cuda.stream(side):
        h0, h1, hcrop = hip_step()
        model.hip_flush()
        want = (h0.clone(), h1.clone(), hcrop.crop(0).clone(), hcrop.crop(1).clone())
    torch.cuda.current_stream(gpu).wait_stream(side)
    torch.cuda.synchronize()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2):
        g0, g1, gcrop = hip_step()
    for _ in range(2):
        graph2.replay()
        torch.cuda.synchronize()
        model.hip_graph_check()
        assert torch.equal(g0, want[0]) and torch.equal(g1, want[1])
        assert torch.equal(gcrop.crop(0), want[2]) and torch.equal(gcrop.crop(1), want[3])
    # and what the eager path produced (MIOpen may pick per-call algorithms: box tolerance)
    assert float((c0 - e0).abs().max()) <= 5e-2 and float((c1 - e1).abs().max()) <= 5e-2
    assert ccrop.valid == ecrop.valid
    # an injected out-of-range batch, eagerly, default settings: enqueued, settled one call later
    bb = model.backbone(torch.cat([im0, im1]))
    bad = model.boxes_from_backbone(bb[:1] * 1e6, bb[1:] * 1e6, (320, 320), (320, 320))
    assert model._pending is not None
    good = model.forward_dummy(im0, im1)          # submitting the next batch settles the previous one
    model.hip_flush()
    assert torch.isfinite(bad[0]).all() and torch.isfinite(bad[1]).all()
    want = model._boxes_checked(*(lambda f1, f2: (f1, f2, model.pos_encoding(f1), model.pos_encoding(f2)))(
        model._neck_torch(bb[:1] * 1e6), model._neck_torch(bb[1:] * 1e6)), (320, 320), (320, 320))
    assert float((bad[0] - want[0]).abs().max()) <= 5e-2 and float((bad[1] - want[1]).abs().max()) <= 5e-2
    assert float((good[0] - e0).abs().max()) <= 5e-2


def test_overflow_inside_a_replayed_graph_is_reported(gpu):
    """A batch captured into a HIP graph cannot be re-run from the host side: its status reads
    are graph nodes, and ``hip_graph_check()`` after a synchronised replay raises when one of
    them tripped (never a silent wrong box)."""
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    sd.update(orc.make_hot_weights(6, sharpen=True))
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    g = torch.Generator().manual_seed(13)
    im = torch.rand(2, 320, 320, 3, generator=g).to(gpu)
    bb = model.backbone(im)
    scale = torch.ones(1, device=gpu)

    def step():
        x = bb * scale
        return model.boxes_from_backbone(x[:1], x[1:], (320, 320), (320, 320), both=x)
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):
        step()
        model.hip_flush()
    torch.cuda.current_stream(gpu).wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        boxes = step()
    graph.replay()
    torch.cuda.synchronize()
    model.hip_graph_check()                      # in range
    assert torch.isfinite(boxes[0]).all()
    scale.fill_(1e6)                             # same graph, out-of-range input
    graph.replay()
    torch.cuda.synchronize()
    with pytest.raises(pkg.OetrRangeError):
        model.hip_graph_check()
    scale.fill_(1.0)
    graph.replay()
    torch.cuda.synchronize()
    model.hip_graph_check()                      # the words are rewritten by every replay


def test_training_forward_matches_the_reference_results(gpu, golden_dir):
    """``OETR.forward(data)`` on the HIP stages vs the result dict the
    REFERENCE model's forward produced on CPU for the same seeded weights and batch
    (``tests/golden/train_forward.npz``): unclamped boxes, L1 / GIoU / oIoU / cycle losses,
    IoU metrics.  Tolerances cover the torch-CPU vs MIOpen trunk (features differ ~1e-5)."""
    g = np.load(golden_dir / 'train_forward.npz')
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    sd.update(orc.make_hot_weights(int(g['hot_seed']), sharpen=True))
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    gen = torch.Generator().manual_seed(int(g['seed']))
    data = {'image1': torch.rand(3, 128, 160, 3, generator=gen), 'image2': torch.rand(3, 160, 128, 3, generator=gen)}
    assert orc.checksum(data['image1']) == list(g['in_fp'][0])
    data = {k: v.to(gpu) for k, v in data.items()}
    data['overlap_valid'] = torch.from_numpy(g['overlap_valid']).to(gpu)
    data['overlap_box1'] = torch.from_numpy(g['overlap_box1']).to(gpu)
    data['overlap_box2'] = torch.from_numpy(g['overlap_box2']).to(gpu)
    for tag, cycle, oiou in (('giou', False, False), ('giou_cycle', True, False), ('oiou_cycle', True, True)):
        model.cycle, model.oiou = cycle, oiou
        res = model(data)
        keys = ['pred_bbox1', 'pred_bbox2', 'iouloss', 'wh_loss', 'loc_loss', 'iou1', 'iou2', 'oiou1',
                'oiou2'] + (['cycle_loss'] if cycle else [])
        assert set(res) == set(keys), sorted(res)
        for k in res:
            ref = torch.from_numpy(g[f'{tag}_{k}'])
            err = float((res[k].cpu() - ref).abs().max())
            tol = 0.1 if k.startswith('pred_bbox') else 2e-3
            assert err <= tol, (tag, k, err)
    # the mask branch (reference src/model.py:256-258: `resize_mask1` in data): the reference model's own
    # forward on the same batch WITH masks (tests/golden/train_forward_masked.npz)
    gm = np.load(golden_dir / 'train_forward_masked.npz')
    m1 = orc.make_masks(int(gm['mask_seeds'][0]), 3, 4, 5, 'holes')
    m2 = orc.make_masks(int(gm['mask_seeds'][1]), 3, 5, 4, 'pad')
    assert np.array_equal(m1.numpy().astype(np.uint8), gm['resize_mask1']) and np.array_equal(m2.numpy().astype(np.uint8), gm['resize_mask2'])
    mdata = dict(data, resize_mask1=m1.to(gpu), resize_mask2=m2.to(gpu))
    for tag, cycle, oiou in (('giou', False, False), ('oiou_cycle', True, True)):
        model.cycle, model.oiou = cycle, oiou
        res = model(mdata)
        for k in res:
            ref = torch.from_numpy(gm[f'{tag}_{k}'])
            err = float((res[k].cpu() - ref).abs().max())
            tol = 0.1 if k.startswith('pred_bbox') else 2e-3
            assert err <= tol, ('masked', tag, k, err)
        assert float((res['pred_bbox1'].cpu() - torch.from_numpy(g[f'{tag}_pred_bbox1'])).abs().max()) > 0.5   # the masks matter
    model.cycle, model.oiou = False, False


def test_throughput_mode_is_the_serial_result_bit_for_bit(gpu):
    """the throughput mode as a product path.  ``model.hip_streams = 3``: consecutive
    batches alternate over three side streams with the engines' throughput settings (64-token encoder
    workgroups, direct tail), one workspace per stream, ``hip_queue_depth`` batches queued per stream (one: at
    most three in flight; two, the default: six), range checks settled in submission order.  Twelve batches of three shapes, one of them overflow-injected (its
    status word trips on a side stream while its neighbours are in flight; it is re-run in exact fp32
    and corrected in place): after hip_flush() every batch equals - bit for bit - what the same
    settings return one batch at a time; the tripped batch equals the exact-fp32 engine."""
    import imagematching_oetr_amd as pkg
    from oracle import oetr_oracle as orc
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    w = orc.make_hot_weights(5, sharpen=True)
    sd.update(w)
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    shapes = [(3, 13, 13), (8, 20, 20), (2, 10, 20)]
    batches = []
    for i in range(12):
        n, h1, h2 = shapes[i % 3]
        f1, f2 = orc.make_features(100 + i, n, h1, h1), orc.make_features(200 + i, n, h2, h2)
        if i == 4:
            f1 = f1 * 4.0e5                       # a GEMM operand beyond the f16 range
        batches.append([t.to(gpu) for t in (f1, f2, orc.position_table(h1, h1), orc.position_table(h2, h2))]
                       + [(h1 * 32, h1 * 32), (h2 * 32, h2 * 32)])
    # one batch at a time, the same engine settings
    model.hip_streams, model.hip_throughput = 1, True
    serial = []
    for b in batches:
        out = model.boxes_from_features(*b)
        model.hip_flush()
        serial.append([t.clone() for t in out])
    exact = pkg.HotPathEngine(w, device=gpu, precision='f32')
    exact.set_decoder_split(1)
    e = exact.forward(*batches[4])
    assert torch.equal(serial[4][0], e[0]) and torch.equal(serial[4][1], e[1])
    # three streams
    model.hip_streams, model.hip_throughput = 3, None
    assert model.hip_queue_depth == 2
    for rnd in range(3):
        model.hip_queue_depth = 1 if rnd == 1 else 2
        outs, most = [], 0
        for b in batches:
            outs.append(model.boxes_from_features(*b))
            most = max(most, len(model._inflight))
        assert most == (3 if rnd == 1 else 6), most
        assert model.hip_batch_stream() in model._side_streams
        model.hip_flush()
        assert len(model._inflight) == 0
        torch.cuda.synchronize()
        for i, (o, r) in enumerate(zip(outs, serial)):
            assert torch.equal(o[0], r[0]) and torch.equal(o[1], r[1]), (rnd, i)
    assert model.engine().query_flags() == 0
    # 'raise' reports the tripped batch when ITS turn comes, and leaves the queue consistent
    model.hip_on_overflow = 'raise'
    for b in batches[:4]:
        model.boxes_from_features(*b)
    model.boxes_from_features(*batches[4])
    model.boxes_from_features(*batches[5])
    with pytest.raises(pkg.OetrRangeError):
        model.hip_flush()
    model.hip_flush()                             # the batches behind it
    assert len(model._inflight) == 0
    model.hip_on_overflow = 'f32'
    # the pair front-end from images (trunk on the caller's stream, hot path on the side streams)
    g = torch.Generator().manual_seed(3)
    pairs = [(torch.rand(160, 192, 3, generator=g), torch.rand(192, 160, 3, generator=g)) for _ in range(5)] + \
            [(torch.rand(128, 128, 3, generator=g), torch.rand(128, 128, 3, generator=g)) for _ in range(4)]
    many = pkg.forward_pairs(model, pairs, max_batch=2)
    model.hip_streams, model.hip_throughput = 1, True
    one = pkg.forward_pairs(model, pairs, max_batch=2)
    # (two trunk runs: MIOpen's convolutions are not run-to-run bit-stable; box tolerance)
    assert float((many[0] - one[0]).abs().max()) <= 5e-2 and float((many[1] - one[1]).abs().max()) <= 5e-2
