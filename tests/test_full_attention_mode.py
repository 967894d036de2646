"""attention='full' as an encoder mode (reference
``EncoderLayer(attention='full')``, ``src/models/transformer.py:86-89`` with
``FullAttention``, ``src/models/linear_attention.py:53-87``): goldens
``tests/golden/fullattn_*.npz`` come from the reference model with the attention module of
all eight encoder layers swapped (``oracle/gen_golden.py``); the CPU test pins the oracle,
the GPU test the HIP path (flash-style f16-split kernel fused into the encoder)."""
import glob
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import oetr_oracle as orc
from tests.test_oracle_golden import load_hot_case

torch.set_grad_enabled(False)
CASES = sorted(glob.glob(str(Path(__file__).parent / 'golden' / 'fullattn_*.npz')))
TOL = dict(memory=2e-4, hs=1e-4, logits=1e-3, cxy=5e-2, tlbr=1e-5, box=5e-2)


def maxerr(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    return float((a.reshape(b.shape) - b).abs().max())


@pytest.mark.parametrize('path', CASES, ids=lambda p: p.split('fullattn_')[-1][:-4])
def test_oracle_full_attention_matches_reference(path):
    g, w, f1, f2 = load_hot_case(path)
    im1, im2 = tuple(int(v) for v in g['img1']), tuple(int(v) for v in g['img2'])
    ref = orc.hot_path(f1, f2, w, im1, im2, return_stages=True, attention=orc.full_attention)
    for s in ('1', '2'):
        step = int(g[f'memory{s}_step'])
        assert maxerr(ref['memory' + s][:, ::step], g['memory' + s]) <= 5e-5
        assert maxerr(ref['hs' + s], g['hs' + s]) <= 5e-5
        assert maxerr(ref['cxy' + s], g['cxy' + s]) <= 1e-2
        assert maxerr(ref['box' + s], g['box' + s]) <= 1e-2
    # ... and it is a different function from the default linear attention
    lin = orc.hot_path(f1, f2, w, im1, im2, return_stages=True)
    assert maxerr(lin['memory1'], ref['memory1']) > 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['f32_split_f16', 'f16', 'f32'])
@pytest.mark.parametrize('path', CASES, ids=lambda p: p.split('fullattn_')[-1][:-4])
def test_hip_full_attention_mode_vs_reference_golden(path, precision, gpu):
    from imagematching_oetr_amd import HotPathEngine
    g, w, f1, f2 = load_hot_case(path)
    im1, im2 = tuple(int(v) for v in g['img1']), tuple(int(v) for v in g['img2'])
    p1, p2 = orc.position_table(*g['grid1']), orc.position_table(*g['grid2'])
    eng = HotPathEngine(w, device=gpu, precision=precision, attention='full')
    dev = [t.to(gpu) for t in (f1, f2, p1, p2)]
    out = eng.forward(*dev, im1, im2, stages=True)
    assert eng.query_flags() == 0
    if precision in ('f32_split_f16', 'f32'):  # fp32-class / exact fp32: the fp32 tolerances of the parity suite
        for s in ('1', '2'):
            step = int(g[f'memory{s}_step'])
            assert maxerr(out['memory' + s][:, ::step], g['memory' + s]) <= TOL['memory']
            for key in ('hs', 'logits', 'cxy', 'tlbr', 'box'):
                assert maxerr(out[key + s], g[key + s]) <= TOL[key], (key, s)
        for li in (0, 1):                     # encoder prefixes: self layer, cross layer
            pre = eng.forward(*dev, im1, im2, stages=True, enc_layers=li + 1)
            for s in ('1', '2'):
                step = int(g[f'enc{li}_x{s}_step'])
                assert maxerr(pre['memory' + s][:, ::step], g[f'enc{li}_x{s}']) <= TOL['memory'], (li, s)
    else:                                     # single-pass f16 GEMMs: its own drift bounds
        for s in ('1', '2'):
            step = int(g[f'memory{s}_step'])
            assert maxerr(out['memory' + s][:, ::step], g['memory' + s]) <= 2.5e-2
    b1, b2 = eng.forward(*dev, im1, im2)
    assert torch.equal(b1, out['box1']) and torch.equal(b2, out['box2'])      # repeatable, staged == plain
    # batch slicing stays bit-exact (pairs independent)
    c1, _ = eng.forward(dev[0][:1].contiguous(), dev[1][:1].contiguous(), dev[2], dev[3], im1, im2)
    assert torch.equal(c1[0], b1[0])


@pytest.mark.gpu
def test_full_attention_mode_contract(gpu):
    from imagematching_oetr_amd import HotPathEngine, OetrError
    w = orc.make_hot_weights(0)
    for prec in ('bf16', 'f32_split_qk16'):
        with pytest.raises(OetrError, match='full'):
            HotPathEngine(w, device=gpu, precision=prec, attention='full')
    with pytest.raises(ValueError):
        HotPathEngine(w, device=gpu, attention='sparse')
    # single token per image, ragged tiles, 1600 keys
    for prec in ('f32_split_f16', 'f32'):
      eng = HotPathEngine(w, device=gpu, attention='full', precision=prec)
      for g1, g2 in (((1, 1), (1, 1)), ((1, 7), (33, 1)), ((5, 5), (40, 40))):
          f1, f2 = orc.make_features(31, 2, *g1), orc.make_features(32, 2, *g2)
          p1, p2 = orc.position_table(*g1), orc.position_table(*g2)
          im1, im2 = (g1[0] * 32, g1[1] * 32), (g2[0] * 32, g2[1] * 32)
          out = eng.forward(f1.to(gpu), f2.to(gpu), p1.to(gpu), p2.to(gpu), im1, im2, stages=True)
          ref = orc.hot_path(f1, f2, w, im1, im2, return_stages=True, attention=orc.full_attention)
          assert maxerr(out['memory1'], ref['memory1']) <= TOL['memory'], (g1, g2)
          assert maxerr(out['memory2'], ref['memory2']) <= TOL['memory'], (g1, g2)
          assert maxerr(out['box1'], ref['box1']) <= TOL['box']


@pytest.mark.gpu
def test_full_attention_mode_sharp_scores_take_the_lazy_maximum_path(gpu):
    """Round 6: the encoder's all-pairs tile keeps a LAZY reference maximum (raised only when a key tile's
    maximum exceeds it by more than 2^8).  With the initialisation's weights the scores are small and that
    branch never runs after the first tile: sharpen Q and K (x 5 each: scores x 25) so that it does - 1 600
    keys = 50 tiles per query - and compare with the oracle and with the exact-fp32 build, whose tile keeps
    the exact running maximum (another code path)."""
    from imagematching_oetr_amd import HotPathEngine
    w = dict(orc.make_hot_weights(2))
    for l in range(8):
        for name in ('q_proj', 'k_proj'):
            key = f'transformer.encoder.{l}.{name}.weight'
            w[key] = w[key] * 5.0
    g1, g2 = (5, 5), (40, 40)
    f1, f2 = orc.make_features(51, 2, *g1), orc.make_features(52, 2, *g2)
    p1, p2 = orc.position_table(*g1), orc.position_table(*g2)
    im1, im2 = (160, 160), (1280, 1280)
    ref = orc.hot_path(f1.double(), f2.double(), {k: v.double() for k, v in w.items()}, im1, im2, return_stages=True,
                       attention=orc.full_attention)
    # sharp softmaxes amplify fp32 rounding: the yardstick is what torch's own fp32 does against fp64 on this case
    ref32 = orc.hot_path(f1, f2, w, im1, im2, return_stages=True, attention=orc.full_attention)
    drift = max(maxerr(ref32['memory' + s], ref['memory' + s]) for s in ('1', '2'))
    bound = max(4 * drift, 2 * TOL['memory'])
    outs = {}
    for prec in ('f32_split_f16', 'f32'):
        eng = HotPathEngine(w, device=gpu, attention='full', precision=prec)
        outs[prec] = eng.forward(f1.to(gpu), f2.to(gpu), p1.to(gpu), p2.to(gpu), im1, im2, stages=True)
        assert eng.query_flags() == 0
        for s in ('1', '2'):
            err = maxerr(outs[prec]['memory' + s], ref['memory' + s])
            assert err <= bound, (prec, s, err, drift)
    assert maxerr(outs['f32_split_f16']['memory1'], outs['f32']['memory1'].cpu()) <= 2 * bound
    # the attention really is sharp: a uniform average over the keys would give something else entirely
    lin = orc.hot_path(f1, f2, w, im1, im2, return_stages=True)
    assert maxerr(lin['memory1'], ref['memory1']) > 1e-1


@pytest.mark.gpu
def test_module_reruns_an_overflowing_full_attention_batch_in_exact_fp32(gpu):
    """``attention='full'`` has an exact-fp32 build too (``OETR_DTYPE_F32``): an out-of-range batch
    is answered with ITS boxes, as in the linear mode (reference ``transformer.py:86-89`` is fp32)."""
    import imagematching_oetr_amd as pkg
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    w = orc.make_hot_weights(5, sharpen=True)
    sd = model.state_dict(); sd.update(w); model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    model.hip_attention = 'full'
    model.invalidate_engine()
    f1, f2 = orc.make_features(41, 2, 8, 10) * 4.0e5, orc.make_features(42, 2, 10, 8)
    p1, p2 = orc.position_table(8, 10), orc.position_table(10, 8)
    dev = [t.to(gpu) for t in (f1, f2, p1, p2)]
    b1, b2 = model.boxes_from_features(*dev, (256, 320), (320, 256))
    model.hip_flush()
    exact = pkg.HotPathEngine(w, device=gpu, precision='f32', attention='full')
    exact.set_decoder_split(1)     # as the module's re-run route (waits for nobody)
    e1, e2 = exact.forward(*dev, (256, 320), (320, 256))
    assert torch.equal(b1, e1) and torch.equal(b2, e2) and torch.isfinite(b1).all()
    assert model._engine_f32 is not None and model._engine_f32.attention == 'full'
