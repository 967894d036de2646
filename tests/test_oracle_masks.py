"""forward_dummy's optional masks (reference src/model.py:229; LinearAttention q_mask / kv_mask,
linear_attention.py:37-41; the decoder's memory_mask, transformer.py:361-381; the heat map's
masked_fill, model.py:166-171): the oracle against vectors produced by the imported reference
(oracle/gen_golden.py: gen_attention_masked, MASK_CASES).  CPU only."""
import glob
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import oetr_oracle as orc
from tests.test_oracle_golden import _close, _sub, load_hot_case

torch.set_grad_enabled(False)
MASKED = sorted(glob.glob(str(Path(__file__).parent / 'golden' / 'hotmask_*.npz')))


def load_masks(g):
    n = int(g['n'])
    m1 = orc.make_masks(int(g['mask_seed']), n, *g['grid1'], kind=str(g['mask_kind']))
    m2 = orc.make_masks(int(g['mask_seed']) + 100, n, *g['grid2'], kind=str(g['mask_kind']))
    assert np.array_equal(m1.numpy().astype(np.uint8), g['mask1']), 'seeded masks differ'
    assert np.array_equal(m2.numpy().astype(np.uint8), g['mask2'])
    assert 0 < m1.mean() < 1 and 0 < m2.mean() < 1
    return m1, m2


def test_masked_linear_attention_matches_reference(golden_dir):
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
        fps = np.stack([orc.checksum(t) for t in (q, k, v, qm, km)])
        assert np.array_equal(fps, g[tag + '_in_fp']), 'seeded inputs differ'
        out = orc.linear_attention(q, k, v, q_mask=qm, kv_mask=km).reshape(2, L, 256)
        _close(_sub(out, g[tag + '_step']), g[tag + '_lin'], 1e-6, 1e-6, tag)
        assert (out[qm == 0] == 0).all()          # a masked query's message is exactly zero


@pytest.mark.parametrize('path', MASKED, ids=lambda p: p.split('hotmask_')[-1][:-4])
def test_masked_hot_path_matches_reference(path):
    g, w, f1, f2 = load_hot_case(path)
    m1, m2 = load_masks(g)
    (hf1, wf1), (hf2, wf2) = g['grid1'], g['grid2']
    p1, p2 = orc.position_table(hf1, wf1), orc.position_table(hf2, wf2)
    x1, x2, t1, t2 = orc.tokens(f1), orc.tokens(f2), orc.tokens(p1), orc.tokens(p2)
    for li in (0, 1):
        y1, y2 = orc.encoder_stack(x1, x2, t1, t2, w, n_layers=li + 1, mask1=m1.flatten(1),
                                   mask2=m2.flatten(1))
        _close(_sub(y1, g[f'enc{li}_x1_step']), g[f'enc{li}_x1'], 2e-5, 1e-5, f'enc{li} x1')
        _close(_sub(y2, g[f'enc{li}_x2_step']), g[f'enc{li}_x2'], 2e-5, 1e-5, f'enc{li} x2')
    st = orc.hot_path(f1, f2, w, tuple(g['img1']), tuple(g['img2']), return_stages=True,
                      mask1=m1, mask2=m2)
    plain = orc.hot_path(f1, f2, w, tuple(g['img1']), tuple(g['img2']), return_stages=True)
    for s, m in (('1', m1), ('2', m2)):
        _close(_sub(st['memory' + s], g[f'memory{s}_step']), g['memory' + s], 5e-5, 1e-5, 'memory' + s)
        _close(st['hs' + s].numpy(), g['hs' + s], 1e-4, 1e-5, 'hs' + s)
        _close(st['logits' + s].numpy(), g['logits' + s], 2e-3, 1e-4, 'logits' + s)
        assert (st['logits' + s][m.flatten(1) == 0] == orc.MASK_FILL).all()
        _close(st['cxy' + s].numpy(), g['cxy' + s], 2e-2, 0, 'cxy' + s)
        _close(st['tlbr' + s].numpy(), g['tlbr' + s], 1e-5, 0, 'tlbr' + s)
        _close(st['box' + s].numpy(), g['box' + s], 3e-2, 0, 'box' + s)
        iou = orc.bbox_iou_aligned(st['box' + s], torch.from_numpy(g['box' + s]))
        assert (iou >= 1 - 1e-3).all(), iou
        # the masks matter: the unmasked forward of the same inputs is somewhere else
        assert (st['hs' + s] - plain['hs' + s]).abs().max() > 1e-3
