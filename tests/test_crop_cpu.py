"""CPU: the box -> crop oracle (``oracle/crop_oracle.py``) against
``tests/golden/crop.npz`` - produced by the reference's own ``Matching.forward`` overlap
branch, ``tensor_overlap_crop`` and ``patch_resize`` (``oracle/gen_golden.py::gen_crop``).
Gate, scaled boxes, ratios and output shapes are the reference's arithmetic and must match
exactly; the crop pixels match bit for bit because the golden run used this oracle's
bicubic in place of the absent ``cv2.resize`` (OpenCV's own numerics stay unpinned)."""
import numpy as np
import pytest
import torch

from oracle import crop_oracle as cro
from oracle import oetr_oracle as orc


def load_cases(golden_dir):
    g = np.load(golden_dir / 'crop.npz')
    for ci in range(int(g['n_cases'])):
        t = f'c{ci}_'
        gen = torch.Generator().manual_seed(int(g[t + 'seed']))
        ch = int(g[t + 'channels'])
        im0 = torch.rand(1, ch, *(int(v) for v in g[t + 'hw0']), generator=gen)
        im1 = torch.rand(1, ch, *(int(v) for v in g[t + 'hw1']), generator=gen)
        assert orc.checksum(im0) == list(g[t + 'in_fp'][0]) and orc.checksum(im1) == list(g[t + 'in_fp'][1])
        yield ci, {k[len(t):]: g[k] for k in g.files if k.startswith(t)}, im0, im1


def test_oracle_reproduces_the_reference_crop_step(golden_dir):
    n = 0
    for ci, c, im0, im1 in load_cases(golden_dir):
        out = cro.overlap_crop(im0, im1, torch.from_numpy(c['box0']), torch.from_numpy(c['box1']),
                               tuple(c['scales0']), tuple(c['scales1']),
                               keep_aspect=bool(c['keep_aspect']), size_divisor=int(c['size_divisor']),
                               pragueparks=bool(c['pragueparks']))
        assert out['valid'] == bool(c['valid']), ci
        assert np.array_equal(out['bbox0'].numpy().astype(np.float32), c['bbox0']), ci
        assert np.array_equal(out['bbox1'].numpy().astype(np.float32), c['bbox1']), ci
        assert np.array_equal(np.float32(out['ratio0']).astype(np.float64), c['ratio0']), ci
        assert np.array_equal(np.float32(out['ratio1']).astype(np.float64), c['ratio1']), ci
        assert tuple(out['crop0'].shape) == tuple(c['out_shape0']), ci
        assert tuple(out['crop1'].shape) == tuple(c['out_shape1']), ci
        assert orc.checksum(out['crop0']) == list(c['out_fp'][0]), ci
        assert orc.checksum(out['crop1']) == list(c['out_fp'][1]), ci
        if 'crop0' in c:
            assert np.array_equal(out['crop0'].numpy(), c['crop0'])
        n += 1
    assert n == 8


def test_bicubic_properties():
    """Size-independent properties of the restated OpenCV bicubic: identity at equal
    size, exact on constants, weights sum to 1, monotone on ramps, replicate border."""
    g = np.random.default_rng(0)
    img = g.random((13, 17, 3), dtype=np.float32)
    assert np.array_equal(cro.bicubic_resize(img, 17, 13), img)
    const = np.full((9, 11), 3.25, dtype=np.float32)
    assert np.allclose(cro.bicubic_resize(const, 23, 31), 3.25, atol=1e-6)
    for t in (0.0, 0.25, 0.5, 0.999):
        assert abs(float(cro.cubic_weights(t).sum()) - 1.0) < 1e-6
    # (a = -0.75 is not Catmull-Rom: linear ramps are reproduced only approximately)
    ramp = np.tile(np.arange(32, dtype=np.float32), (4, 1))
    up = cro.bicubic_resize(ramp, 64, 4)
    x = (np.arange(64) + 0.5) * 0.5 - 0.5
    assert np.abs(up[0, 4:-4] - x[4:-4]).max() < 0.1 and np.all(np.diff(up[0]) >= -1e-6)
    assert up[0, 0] == ramp[0, 0] or abs(up[0, 0] - ramp[0, 0]) < 0.2     # replicate border
    assert cro.bicubic_resize(img, 5, 4).shape == (4, 5, 3)


def test_gate_and_geometry_edge_cases():
    b = torch.tensor([10.0, 10.0, 50.0, 50.0])
    assert cro.scale_and_gate(b, b, (1, 1), (1, 1))[2] is True
    assert cro.scale_and_gate(torch.tensor([10.0, 10.0, 11.9, 50.0]), b, (1, 1), (1, 1))[2] is False
    assert cro.scale_and_gate(b, torch.tensor([10.0, 10.0, 10.0, 50.0]), (1, 1), (1, 1))[2] is False   # zero width
    assert cro.scale_and_gate(b, b, (1, 1), (1, 1), pragueparks=True)[2] is False       # ratio 1
    big = torch.tensor([0.0, 0.0, 130.0, 40.0])
    assert cro.scale_and_gate(big, b, (1, 1), (1, 1), pragueparks=True)[2] is True      # 130 // 40 = 3
    geo = cro.crop_geometry((100, 100), (50, 50), torch.tensor([90.0, 90.0, 120.0, 130.0]),
                            torch.tensor([0.0, 0.0, 50.0, 50.0]), True, 8)
    assert geo[0]['crop'] == (10, 10) and geo[0]['box'] == (90, 90, 120, 130)
    assert geo[0]['new'] == (100, 100) and geo[0]['out'] == (104, 104)
    assert geo[1]['ratio'] == (2.0, 2.0) and geo[1]['out'] == (104, 104)


def test_bicubic_agrees_with_an_independent_implementation():
    """cv2 is not installed, so OpenCV's own output cannot pin :func:`bicubic_resize`; what
    can be checked is that the restatement agrees with an INDEPENDENT implementation of the
    same published algorithm - torch's ``F.interpolate(mode='bicubic',
    align_corners=False)``: cubic convolution with a = -0.75, pixel centres
    ``(d + 0.5) * scale - 0.5``, taps clamped to the border, no antialiasing.  The two
    differ only in fp32 rounding (torch evaluates all four tap polynomials, OpenCV gets the
    fourth as 1 - the others; separable passes in the other order): observed <= 7e-4 on
    the 0..255 scale the reference resizes on (3e-6 relative), up- and down-scaling."""
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    for h, w, nh, nw in ((13, 17, 31, 23), (40, 50, 20, 25), (33, 21, 64, 64), (64, 48, 17, 5), (7, 9, 7, 30)):
        img = (rng.random((h, w, 3)) * 255).astype(np.float32)
        ours = cro.bicubic_resize(img, nw, nh)
        ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(nh, nw), mode='bicubic',
                            align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(ours - ref).max() <= 2e-3, (h, w, nh, nw, np.abs(ours - ref).max())
