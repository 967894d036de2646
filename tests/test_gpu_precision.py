"""GPU tests of the GEMM arithmetic modes beyond the fp32-class default:

* the per-GEMM-site precision POLICY ``f32_split_qk16`` (``OETR_DTYPE_F32_SPLIT_QK16``): the
  encoder's Q / K projections and the decoder's K projection on single f16 MFMAs, every other
  site fp32-class.  It is the reduced mode that MEETS the north_star bar - plain passing
  tests below, on every golden incl. the sharpened-head ones;

* the single-pass 16-bit operand modes ``f16`` (BASELINE configs[4]: "fp16 with fp32
  accumulate") and ``bf16`` (configs[2]: "bf16 MFMA attention"), gated on the
  north_star bar - boxes within 1e-3 IoU of the reference's - on every golden, with
  the drift of the intermediate tensors RECORDED (under ``OETR_DRIFT_LOG=<file>``,
  copied to ``profiles/``) and bounded by separate, wider tolerances than the fp32 ones;
* the range guard of the f16-based modes: out-of-range weights are rejected by
  ``oetr_create``, out-of-range activations set ``OETR_FLAG_F16_RANGE`` (never a silent
  wrong box), and the drop-in module re-runs such a batch in exact fp32;
* magnitude stress: features and weights far from the Xavier / +-0.5 comfort zone,
  against the fp64 oracle with tolerances scaled to the tensors' magnitude.

The reference is fp32-only (no autocast anywhere), so the reduced modes have no
reference lines of their own; their bar is the north_star's IoU >= 1 - 1e-3.
"""
import glob
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import oetr_oracle as orc

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

HOT = sorted(glob.glob(str(Path(__file__).parent / 'golden' / 'hot_*.npz')))
REDUCED = ['f16', 'f16@64', 'bf16', 'bf16@64']
# Measured on MI355X, round 2 (profiles/r3_precision_drift.json), worst case over the
# five reference goldens + the full-forward golden, either tile shape:
#            memory   hs      tlbr     min IoU (plain heads)  min IoU (sharpened heads)
#   f16      8.6e-3   1.1e-2  7.2e-4   0.9975                 0.935   (cxy off by up to 7 px)
#   bf16     7.0e-2   7.8e-2  1.2e-2   0.973                  0.883   (cxy off by up to 14 px)
# i.e. NEITHER single-pass mode meets the north_star bar (IoU >= 1 - 1e-3) on the seeded
# goldens: rounding every GEMM operand to 11 (8) mantissa bits drifts `memory` by ~1e-2
# (~7e-2) through the 8 encoder layers, which moves the sigmoid extents by ~0.5 (~5) px.
# The fp32-class default (f32_split_f16) and exact f32 do meet it (test_gpu_parity.py).
# The bounds below are ~2-3x the worst observation so that a real regression trips them.
DRIFT_TOL = {'f16': dict(memory=2.5e-2, hs=3e-2, tlbr=2e-3),
             'bf16': dict(memory=2e-1, hs=2e-1, tlbr=3e-2)}
IOU_FLOOR = {'f16': dict(plain=0.995, sharp=0.90), 'bf16': dict(plain=0.95, sharp=0.80)}
BAR_XFAIL = ('single-pass 16-bit GEMM operands miss the 1e-3 IoU bar on the seeded goldens: '
             'measured min IoU f16 0.9975 (plain heads) / 0.935 (sharpened), bf16 0.973 / 0.883 '
             '(profiles/r3_precision_drift.json); kept as an expected failure, not dropped')
# (written only when OETR_DRIFT_LOG names a file - a plain test run leaves no files behind)
DRIFT_LOG = Path(os.environ['OETR_DRIFT_LOG']) if os.environ.get('OETR_DRIFT_LOG') else None


def maxerr(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    return float((a.reshape(b.shape) - b).abs().max())


def _engine(w, gpu, precision):
    from imagematching_oetr_amd import HotPathEngine
    prec, _, tile = precision.partition('@')
    return HotPathEngine(w, device=gpu, precision=prec, enc_tile=int(tile) if tile else None)


def _record(entry):
    if DRIFT_LOG is None:
        return
    try:
        DRIFT_LOG.parent.mkdir(parents=True, exist_ok=True)
        rows = json.loads(DRIFT_LOG.read_text()) if DRIFT_LOG.exists() else []
        rows = [r for r in rows if (r['case'], r['precision']) != (entry['case'], entry['precision'])]
        rows.append(entry)
        DRIFT_LOG.write_text(json.dumps(rows, indent=1))
    except OSError:
        pass


def _golden_drift(path, precision, gpu):
    from tests.test_oracle_golden import load_hot_case
    g, w, f1, f2 = load_hot_case(path)
    im1, im2 = tuple(int(v) for v in g['img1']), tuple(int(v) for v in g['img2'])
    p1, p2 = orc.position_table(*g['grid1']), orc.position_table(*g['grid2'])
    eng = _engine(w, gpu, precision)
    out = eng.forward(f1.to(gpu), f2.to(gpu), p1.to(gpu), p2.to(gpu), im1, im2, stages=True)
    assert eng.query_flags() == 0
    drift, ious = {}, []
    for s in ('1', '2'):
        step = int(g[f'memory{s}_step'])
        drift['memory' + s] = maxerr(out['memory' + s][:, ::step], g['memory' + s])
        for key in ('hs', 'logits', 'cxy', 'tlbr', 'box'):
            drift[key + s] = maxerr(out[key + s], g[key + s])
        ref_box = torch.from_numpy(g['box' + s])
        iou = orc.bbox_iou_aligned(out['box' + s].cpu(), ref_box)
        area = (ref_box[:, 2] - ref_box[:, 0]) * (ref_box[:, 3] - ref_box[:, 1])
        ious += [float(v) for v in iou[area > 1]]
    entry = dict(case=Path(path).stem, precision=precision, sharpened_heads=bool(g['sharpen']),
                 min_iou=min(ious) if ious else None,
                 **{k: float(f'{v:.3e}') for k, v in drift.items()})
    _record(entry)
    return entry, drift, ious


@pytest.mark.parametrize('precision', REDUCED)
@pytest.mark.parametrize('path', HOT, ids=lambda p: p.split('hot_')[-1][:-4])
def test_reduced_precision_drift_is_recorded_and_bounded(path, precision, gpu):
    """16-bit operand modes vs what the REFERENCE produced (goldens): the drift of every
    stage is recorded (OETR_DRIFT_LOG=<file> -> profiles/) and the well-conditioned
    tensors (memory, hs, tlbr) plus the box IoU are bounded by the mode's own tolerances."""
    entry, drift, ious = _golden_drift(path, precision, gpu)
    mode = precision.partition('@')[0]
    for key, v in drift.items():
        if key[:-1] in DRIFT_TOL[mode]:
            assert v <= DRIFT_TOL[mode][key[:-1]], f'{precision} {key}: drift {v:.3e} ({entry})'
    floor = IOU_FLOOR[mode]['sharp' if entry['sharpened_heads'] else 'plain']
    assert all(v >= floor for v in ious), f'{precision}: IoU below its floor {floor}: {entry}'


# hot_s2's boxes are saturated at the image border in every mode (IoU 1.0 whatever the drift): it says
# nothing about the bar and is left out; on the others the failure is STRICT - a mode that starts to
# meet the bar shows up as an XPASS failure and gets promoted to a passing test.
HOT_BAR = [p for p in HOT if not Path(p).stem.startswith('hot_s2')]


@pytest.mark.xfail(reason=BAR_XFAIL, strict=True)
@pytest.mark.parametrize('precision', REDUCED)
@pytest.mark.parametrize('path', HOT_BAR, ids=lambda p: p.split('hot_')[-1][:-4])
def test_reduced_precision_meets_the_north_star_iou_bar(path, precision, gpu):
    """The north_star bar itself (IoU >= 1 - 1e-3 vs the reference's boxes) in the
    single-pass modes: fails on every golden whose boxes are not saturated at the image
    border - see BAR_XFAIL for the measured margins."""
    entry, _, ious = _golden_drift(path, precision, gpu)
    assert all(v >= 1 - 1e-3 for v in ious), f'{precision}: IoU bar missed: {entry}'


# --------------------------------------------------------------- precision policy (QK16)
# Which GEMM sites may be reduced is a measured property (profiles/r3_site_drift.json, one
# site at a time on the goldens; tools/site_drift.py emulates the same on the CPU oracle):
#   site reduced to f16 operands      worst 1 - IoU      verdict
#   Q                                 8e-5               ok   (phi(Q) enters numerator and normaliser)
#   K                                 9e-5               ok   (K only enters sums over all source tokens)
#   decoder K                         1e-5               ok
#   V                                 3e-2               no   (weight rounding is systematic across tokens)
#   merge / MLP1 / MLP2               1e-2 / 3e-2 / 1e-2 no   (feed the residual stream directly)
#   attention contractions (KV, apply) 1e-2 (bf16: 1e-1) no   - what configs[2] literally names
# Q + K + decoder K together: 1.1e-4 = the policy.
POLICY = 'f32_split_qk16'
POLICY_TOL = dict(memory=3e-4, hs=3e-4, tlbr=5e-5, cxy=0.15)   # observed 6e-5 / 6e-5 / 1e-5 / 0.035 px


@pytest.mark.parametrize('precision', [POLICY + '@32', POLICY + '@64'])
@pytest.mark.parametrize('path', HOT, ids=lambda p: p.split('hot_')[-1][:-4])
def test_precision_policy_meets_the_north_star_iou_bar(path, precision, gpu):
    """The north_star bar (IoU >= 1 - 1e-3 vs the REFERENCE's boxes) under the QK16 policy, on
    every golden - 20x20, 32x32, mixed 20x20 vs 40x40 (configs[4]'s shape), ragged grids,
    plain and sharpened heads - in both encoder workgroup shapes, plus bounds on the
    intermediate tensors.  A plain test."""
    entry, drift, ious = _golden_drift(path, precision, gpu)
    assert all(v >= 1 - 1e-3 for v in ious), f'{precision}: IoU bar missed: {entry}'
    for key, v in drift.items():
        if key[:-1] in POLICY_TOL:
            assert v <= POLICY_TOL[key[:-1]], f'{precision} {key}: drift {v:.3e} ({entry})'


def test_precision_policy_full_forward_golden_boxes(gpu, golden_dir):
    """Boxes the reference's forward_dummy produced from 640x640 images under the policy."""
    g = np.load(golden_dir / 'full_640.npz')
    eng = _engine(orc.make_hot_weights(int(g['weight_seed']), sharpen=True), gpu, POLICY)
    t = [torch.from_numpy(g[k]).to(gpu) for k in ('feat1', 'feat2', 'pos1', 'pos2')]
    b1, b2 = eng.forward(*t, (640, 640), (640, 640))
    iou = orc.bbox_iou_aligned(torch.cat([b1, b2]).cpu(),
                               torch.from_numpy(np.concatenate([g['box1'], g['box2']])))
    _record(dict(case='full_640', precision=POLICY, sharpened_heads=True, min_iou=float(iou.min()),
                 box1=maxerr(b1, g['box1']), box2=maxerr(b2, g['box2'])))
    assert (iou >= 1 - 1e-3).all(), iou


def test_precision_policy_shapes_and_limits(gpu):
    """Both encoder workgroup shapes run the policy (same sites reduced: boxes agree to the
    summation-order level); the all-pairs attention mode has no policy build."""
    from imagematching_oetr_amd import HotPathEngine, OetrError
    w = orc.make_hot_weights(1, sharpen=True)
    f1, f2 = orc.make_features(11, 2, 20, 20).to(gpu), orc.make_features(111, 2, 20, 20).to(gpu)
    p = orc.position_table(20, 20).to(gpu)
    eng = _engine(w, gpu, POLICY)
    boxes = {}
    for tile in (32, 64, 0):
        eng.set_encoder_tile(tile)
        boxes[tile] = eng.forward(f1, f2, p, p, (640, 640), (640, 640))
    assert float((boxes[32][0] - boxes[64][0]).abs().max()) <= 5e-2
    assert torch.equal(boxes[0][0], boxes[32][0])            # auto at 2 pairs: 32-token tiles
    with pytest.raises(OetrError):
        HotPathEngine(w, device=gpu, precision=POLICY, attention='full')


@pytest.mark.parametrize('precision', REDUCED)
def test_reduced_precision_full_forward_golden_boxes(gpu, golden_dir, precision):
    """Boxes the reference's forward_dummy produced from 640x640 images (real
    extraction-path features, sharpened heads): recorded, floor-checked."""
    g = np.load(golden_dir / 'full_640.npz')
    eng = _engine(orc.make_hot_weights(int(g['weight_seed']), sharpen=True), gpu, precision)
    t = [torch.from_numpy(g[k]).to(gpu) for k in ('feat1', 'feat2', 'pos1', 'pos2')]
    b1, b2 = eng.forward(*t, (640, 640), (640, 640))
    iou = orc.bbox_iou_aligned(torch.cat([b1, b2]).cpu(),
                               torch.from_numpy(np.concatenate([g['box1'], g['box2']])))
    _record(dict(case='full_640', precision=precision, sharpened_heads=True, min_iou=float(iou.min()),
                 box1=maxerr(b1, g['box1']), box2=maxerr(b2, g['box2'])))
    assert (iou >= IOU_FLOOR[precision.partition('@')[0]]['sharp']).all(), iou


@pytest.mark.parametrize('precision', ['f16', 'bf16', 'f32_split_qk16'])
def test_reduced_precision_batch_properties(gpu, precision):
    """configs[2]'s per-GPU workload (8 pairs @640x640) in the reduced modes: pairs stay
    independent (bit-exact under batch permutation / slicing, the decoder's workgroups per image
    pinned - the automatic rule gives the 3-pair slice four), results repeat bit for
    bit and stay valid boxes; the IoU against the fp32 oracle is recorded."""
    w = orc.make_hot_weights(3, sharpen=True)
    eng = _engine(w, gpu, precision)
    eng.set_decoder_split(1)
    n = 8
    f1, f2 = orc.make_features(41, n, 20, 20).to(gpu), orc.make_features(42, n, 20, 20).to(gpu)
    p = orc.position_table(20, 20).to(gpu)
    b1, b2 = eng.forward(f1, f2, p, p, (640, 640), (640, 640))
    perm = torch.tensor([3, 7, 0, 5, 1, 6, 2, 4], device=gpu)
    c1, c2 = eng.forward(f1[perm], f2[perm], p, p, (640, 640), (640, 640))
    assert torch.equal(c1, b1[perm]) and torch.equal(c2, b2[perm])
    d1, _ = eng.forward(f1[2:5], f2[2:5], p, p, (640, 640), (640, 640))
    assert torch.equal(d1, b1[2:5])
    e1, e2 = eng.forward(f1, f2, p, p, (640, 640), (640, 640))
    assert torch.equal(e1, b1) and torch.equal(e2, b2)
    for b in (b1, b2):
        assert torch.isfinite(b).all() and (b >= 0).all() and (b <= 640).all()
        assert (b[:, 2] >= b[:, 0]).all() and (b[:, 3] >= b[:, 1]).all()
    r1, r2 = orc.hot_path(f1.cpu(), f2.cpu(), w, (640, 640), (640, 640))
    iou = torch.cat([orc.bbox_iou_aligned(b1.cpu(), r1), orc.bbox_iou_aligned(b2.cpu(), r2)])
    _record(dict(case='bench_8x640_sharp_vs_oracle', precision=precision, sharpened_heads=True,
                 min_iou=float(iou.min()), mean_iou=float(iou.mean())))
    # (all-rounded modes: sanity only, their bar is tracked by the xfail test above; the policy
    #  meets the bar here too)
    assert float(iou.min()) >= (1 - 1e-3 if precision == POLICY else 0.5)


# ----------------------------------------------------------------- range guard
def test_out_of_range_weights_are_rejected_at_create(gpu):
    from imagematching_oetr_amd import HotPathEngine, OetrError
    w = orc.make_hot_weights(0)
    big = dict(w)
    big['transformer.encoder.3.mlp.0.weight'] = w['transformer.encoder.3.mlp.0.weight'].clone()
    big['transformer.encoder.3.mlp.0.weight'][5, 7] = 7.0e4
    for prec in ('f32_split_f16', 'f16', 'f32_split_qk16'):
        with pytest.raises(OetrError, match='f16 range'):
            HotPathEngine(big, device=gpu, precision=prec)
    for prec in ('bf16', 'f32'):       # representable there
        HotPathEngine(big, device=gpu, precision=prec)
    nan = dict(w)
    nan['heatmap_conv.0.weight'] = w['heatmap_conv.0.weight'].clone()
    nan['heatmap_conv.0.weight'][0, 0, 1, 1] = float('nan')
    for prec in ('f32_split_f16', 'f16', 'bf16', 'f32'):
        with pytest.raises(OetrError, match='non-finite'):
            HotPathEngine(nan, device=gpu, precision=prec)


@pytest.mark.parametrize('precision', ['f32_split_f16', 'f32_split_f16@64', 'f16', 'f32_split_qk16'])
def test_out_of_range_activations_set_the_flag(gpu, precision):
    """Features of magnitude 2e5: `memory` (the un-normalised residual stream) exceeds
    65504 where it enters the decoder K/V and conv-P GEMMs.  The call must report it;
    the flag is sticky until cleared and does not fire on in-range input."""
    from imagematching_oetr_amd import FLAG_F16_RANGE, OetrRangeError
    w = orc.make_hot_weights(2, sharpen=True)
    eng = _engine(w, gpu, precision)
    ok1, ok2 = orc.make_features(90, 2, 9, 13).to(gpu), orc.make_features(91, 2, 6, 6).to(gpu)
    p1, p2 = orc.position_table(9, 13).to(gpu), orc.position_table(6, 6).to(gpu)
    eng.forward(ok1, ok2, p1, p2, (288, 416), (192, 192))
    assert eng.query_flags() == 0
    eng.forward(ok1 * 4.0e5, ok2, p1, p2, (288, 416), (192, 192))
    assert eng.query_flags(clear=False) & FLAG_F16_RANGE
    eng.forward(ok1, ok2, p1, p2, (288, 416), (192, 192))          # sticky across good calls
    with pytest.raises(OetrRangeError):
        eng.check_range()
    assert eng.query_flags() == 0                                   # cleared by the check
    # the exact and bf16 modes have fp32 range: no flag, finite boxes
    for prec in ('f32', 'bf16'):
        e2 = _engine(w, gpu, prec)
        b1, b2 = e2.forward(ok1 * 4.0e5, ok2, p1, p2, (288, 416), (192, 192))
        assert e2.query_flags() == 0 and torch.isfinite(b1).all() and torch.isfinite(b2).all()


def _overflow_model(gpu):
    import imagematching_oetr_amd as pkg
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    w = orc.make_hot_weights(5, sharpen=True)
    sd.update(w)
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    f1, f2 = orc.make_features(95, 2, 8, 10) * 4.0e5, orc.make_features(96, 2, 10, 8)
    p1, p2 = orc.position_table(8, 10), orc.position_table(10, 8)
    return pkg, model, w, [t.to(gpu) for t in (f1, f2, p1, p2)]


@pytest.mark.parametrize('defer', [True, False])
def test_module_reruns_an_overflowing_batch_in_exact_fp32(gpu, defer):
    """Drop-in module: hip_on_overflow='f32' (default) answers an out-of-range batch
    with the exact-fp32 engine's boxes; 'raise' raises; never a silent wrong box.
    Deferred mode (default): the call itself only enqueues; the re-run lands IN the returned
    tensors when the next batch is submitted or at hip_flush()."""
    pkg, model, w, dev = _overflow_model(gpu)
    model.hip_defer_check = defer
    b1, b2 = model.boxes_from_features(*dev, (256, 320), (320, 256))
    if defer:
        assert model._pending is not None
        model.hip_flush()
    exact = pkg.HotPathEngine(w, device=gpu, precision='f32')
    exact.set_decoder_split(1)     # as the module's re-run route (waits for nobody)
    e1, e2 = exact.forward(*dev, (256, 320), (320, 256))
    assert torch.equal(b1, e1) and torch.equal(b2, e2) and torch.isfinite(b1).all()
    # ... which is NOT what the overflowing default mode produced
    raw = model.engine().forward(*dev, (256, 320), (320, 256))
    assert model.engine().query_flags() & pkg.FLAG_F16_RANGE
    del raw
    model.hip_on_overflow = 'raise'
    if defer:
        model.boxes_from_features(*dev, (256, 320), (320, 256))       # enqueues, reports later
        with pytest.raises(pkg.OetrRangeError):
            model.hip_flush()
    else:
        with pytest.raises(pkg.OetrRangeError):
            model.boxes_from_features(*dev, (256, 320), (320, 256))
    # in-range batches never leave the default engine
    model.hip_on_overflow = 'f32'
    model._engine_f32 = None
    model.boxes_from_features(dev[0] / 4.0e5, *dev[1:], (256, 320), (320, 256))
    model.hip_flush()
    assert model._engine_f32 is None


def test_deferred_check_settles_two_calls_later(gpu):
    """hip_defer_check: batch i's status word is examined when batch i+2 is submitted (round 5: one
    batch stays in flight behind the one being submitted, so the host never waits for the device) -
    the injected x4e5 batch's boxes are overwritten by the exact re-run at that point, and the
    in-range batches that follow are untouched."""
    pkg, model, w, dev = _overflow_model(gpu)
    good = [dev[0] / 4.0e5] + dev[1:]
    bad1, bad2 = model.boxes_from_features(*dev, (256, 320), (320, 256))      # trips, unnoticed so far
    assert model._pending is not None and model._engine_f32 is None
    ok1, ok2 = model.boxes_from_features(*good, (256, 320), (320, 256))       # the first stays in flight
    assert model._engine_f32 is None and len(model._inflight) == 2
    model.boxes_from_features(*good, (256, 320), (320, 256))                  # settles the first
    assert model._engine_f32 is not None and len(model._inflight) == 2
    exact = pkg.HotPathEngine(w, device=gpu, precision='f32')
    exact.set_decoder_split(1)     # as the module's re-run route (waits for nobody)
    e1, e2 = exact.forward(*dev, (256, 320), (320, 256))
    assert torch.equal(bad1, e1) and torch.equal(bad2, e2)
    model.hip_flush()
    r1, r2 = model.engine().forward(*good, (256, 320), (320, 256))
    assert torch.equal(ok1, r1) and torch.equal(ok2, r2)


def test_status_words_are_per_stream(gpu):
    """the status word lives in the per-stream workspace - an overflow on stream B
    is neither seen nor cleared by a query on stream A."""
    pkg, model, w, dev = _overflow_model(gpu)
    eng = pkg.HotPathEngine(w, device=gpu)
    good = [dev[0] / 4.0e5] + dev[1:]
    sa, sb = torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        eng.forward(*dev, (256, 320), (320, 256))          # overflows on B
    with torch.cuda.stream(sa):
        eng.forward(*good, (256, 320), (320, 256))
        assert eng.query_flags(clear=True) == 0            # A's own word: clean, and B's untouched
    with torch.cuda.stream(sb):
        assert eng.query_flags(clear=True) & pkg.FLAG_F16_RANGE
        assert eng.query_flags() == 0
    torch.cuda.synchronize()


# ------------------------------------------------------------ magnitude stress
@pytest.mark.parametrize('precision', ['f32_split_f16', 'f32_split_f16@64', 'f32'])
@pytest.mark.parametrize('scale', [1e-3, 30.0, 1e3])
def test_feature_magnitude_stress_vs_fp64_oracle(gpu, precision, scale):
    """Backbone features x{1e-3, 30, 1e3}.  The first LayerNorm removes the scale from
    the attention inputs, but the residual stream - and therefore `memory`, which enters
    the conv-P / decoder K,V GEMMs un-normalised - keeps it.  Tolerances are relative to
    each tensor's own magnitude (fp32-class modes: a few 1e-6 of abs-max)."""
    w = orc.make_hot_weights(4, sharpen=True)
    w64 = orc.cast_weights(w, torch.float64)
    f1, f2 = orc.make_features(101, 2, 12, 9, scale=scale), orc.make_features(102, 2, 7, 16, scale=scale)
    p1, p2 = orc.position_table(12, 9), orc.position_table(7, 16)
    im1, im2 = (384, 288), (224, 512)
    ref = orc.hot_path(f1.double(), f2.double(), w64, im1, im2, return_stages=True)
    eng = _engine(w, gpu, precision)
    out = eng.forward(f1.to(gpu), f2.to(gpu), p1.to(gpu), p2.to(gpu), im1, im2, stages=True)
    assert eng.query_flags() == 0
    for key, rel in (('memory1', 2e-5), ('memory2', 2e-5), ('hs1', 2e-5), ('hs2', 2e-5)):
        mag = float(ref[key].abs().max())
        e = maxerr(out[key], ref[key])
        assert e <= rel * max(mag, 1.0), f'{key} scale {scale}: err {e:.3e}, magnitude {mag:.3e}'
    for s in ('1', '2'):
        assert maxerr(out['tlbr' + s], ref['tlbr' + s]) <= 2e-5
        area = (ref['box' + s][:, 2] - ref['box' + s][:, 0]) * (ref['box' + s][:, 3] - ref['box' + s][:, 1])
        iou = orc.bbox_iou_aligned(out['box' + s].cpu().double(), ref['box' + s])
        assert (iou[area > 1] >= 1 - 1e-3).all(), (scale, iou)


@pytest.mark.parametrize('precision', ['f32_split_f16', 'f32_split_f16@64', 'f32'])
def test_weight_magnitude_stress_vs_fp64_oracle(gpu, precision):
    """A weight set with 1e-6-scale and 50x-scale matrices (Xavier elsewhere): tiny
    projections exercise the low plane of the split near the f16 subnormals, the 50x
    ones large activations ahead of the LayerNorms."""
    w = orc.make_hot_weights(6, sharpen=True)
    for key, s in (('transformer.encoder.0.q_proj.weight', 1e-6),
                   ('transformer.encoder.1.v_proj.weight', 1e-6),
                   ('transformer.encoder.2.mlp.0.weight', 50.0),
                   ('transformer.encoder.3.merge.weight', 50.0),
                   ('transformer.encoder.5.k_proj.weight', 50.0),
                   ('transformer.decoder.layers.1.multihead_attn.v_proj.weight', 50.0)):
        w[key] = w[key] * s
    w64 = orc.cast_weights(w, torch.float64)
    f1, f2 = orc.make_features(111, 2, 10, 10), orc.make_features(112, 2, 6, 20)
    p1, p2 = orc.position_table(10, 10), orc.position_table(6, 20)
    im1, im2 = (320, 320), (192, 640)
    ref = orc.hot_path(f1.double(), f2.double(), w64, im1, im2, return_stages=True)
    s32 = orc.hot_path(f1, f2, w, im1, im2, return_stages=True)
    eng = _engine(w, gpu, precision)
    out = eng.forward(f1.to(gpu), f2.to(gpu), p1.to(gpu), p2.to(gpu), im1, im2, stages=True)
    assert eng.query_flags() == 0
    for key in ('memory1', 'memory2', 'hs1', 'hs2'):
        mag = float(ref[key].abs().max())
        drift32 = maxerr(s32[key], ref[key])          # torch's own fp32 on the same graph
        e = maxerr(out[key], ref[key])
        assert e <= max(8 * drift32, 2e-5 * mag), f'{key}: err {e:.3e}, torch-fp32 drift {drift32:.3e}, |x| {mag:.3e}'
    for s in ('1', '2'):
        area = (ref['box' + s][:, 2] - ref['box' + s][:, 0]) * (ref['box' + s][:, 3] - ref['box' + s][:, 1])
        iou = orc.bbox_iou_aligned(out['box' + s].cpu().double(), ref['box' + s])
        assert (iou[area > 1] >= 1 - 1e-3).all(), iou


def test_standalone_center_estimation_survives_large_attention_weights(gpu):
    """oetr_center_estimation converts memory*att to GEMM operands; att is a 256-term
    dot product (the largest-magnitude tensor of the path).  The kernel scales att by a
    power of two per tile, so hs x 300 (att ~ 1e4 x memory ~ 10 -> 1e5 > 65504 unscaled)
    neither overflows nor loses accuracy."""
    w = orc.make_hot_weights(1, sharpen=True)
    eng = _engine(w, gpu, 'f32_split_f16')
    f1, f2 = orc.make_features(21, 2, 12, 17), orc.make_features(22, 2, 9, 30)
    st = orc.hot_path(f1, f2, w, (384, 544), (288, 960), return_stages=True)
    hs1, hs2 = st['hs1'] * 300.0, st['hs2'] * 300.0
    w64 = orc.cast_weights(w, torch.float64)
    r1, r2 = orc.center_estimation(hs1.double(), hs2.double(), st['memory1'].double(),
                                   st['memory2'].double(), 12, 17, 9, 30, 384, 288, w64)
    c1, c2 = eng.center_estimation(hs1.to(gpu), hs2.to(gpu), st['memory1'].to(gpu),
                                   st['memory2'].to(gpu), 12, 17, 9, 30, 384, 288)
    assert eng.query_flags() == 0
    assert maxerr(c1, r1) <= 5e-2 and maxerr(c2, r2) <= 5e-2


def test_neck_magnitude_and_range_flag(gpu):
    """Post-ReLU backbone maps x1e3 into the HIP neck (f16-split GEMMs) vs the fp64
    oracle, tolerance scaled to the output magnitude; x1e6 must raise the range flag."""
    import imagematching_oetr_amd as pkg
    w = orc.make_neck_weights(44)
    eng = pkg.NeckEngine(w, device=gpu)
    bb = orc.make_backbone_features(45, 2, 14, 18)
    w64 = {k: v.double() for k, v in w.items()}
    for scale in (1e-3, 1.0, 1e3):
        ref = orc.neck(bb.double() * scale, w64)
        feat = eng.forward((bb * scale).to(gpu))
        assert eng.query_flags() == 0
        mag = float(ref.abs().max())
        assert maxerr(feat, ref) <= 1e-5 * max(mag, 1.0), (scale, maxerr(feat, ref), mag)
    eng.forward((bb * 1e6).to(gpu))
    assert eng.query_flags(clear=False) & pkg.FLAG_F16_RANGE
    with pytest.raises(pkg.OetrRangeError):
        eng.check_range()


def test_engines_follow_a_model_on_a_non_current_device():
    """A model on cuda:1 while torch's current device is cuda:0: both engines
    must launch on their own device's stream."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    import imagematching_oetr_amd as pkg
    d1 = torch.device('cuda', 1)
    w, nw = orc.make_hot_weights(7, sharpen=True), orc.make_neck_weights(8)
    f1, f2 = orc.make_features(70, 2, 8, 8), orc.make_features(71, 2, 5, 7)
    p1, p2 = orc.position_table(8, 8), orc.position_table(5, 7)
    bb = orc.make_backbone_features(72, 1, 6, 10)
    with torch.cuda.device(0):
        b1, _ = pkg.HotPathEngine(w, device=d1).forward(f1.to(d1), f2.to(d1), p1.to(d1), p2.to(d1),
                                                        (256, 256), (160, 224))
        feat = pkg.NeckEngine(nw, device=d1).forward(bb.to(d1))
        torch.cuda.synchronize(d1)
    r1, _ = orc.hot_path(f1, f2, w, (256, 256), (160, 224))
    assert maxerr(b1, r1) <= 5e-2 and maxerr(feat, orc.neck(bb, nw)) <= 5e-5
