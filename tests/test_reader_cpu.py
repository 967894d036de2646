"""CPU checks of the READER half of the pair front end: the oracle's
restatement of ``read_overlap_image`` against goldens produced by the reference's own
function (``tests/golden/reader.npz``), the library's host-side frame arithmetic
(``oetr_overlap_frame``) against both, and the restated bilinear resize against torch's
independent implementation of the same algorithm."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import imagematching_oetr_amd as pkg
from oracle import oetr_oracle as orc
from oracle import reader_oracle as rdo


def load_reader_cases(golden_dir):
    g = np.load(golden_dir / 'reader.npz')
    for ci in range(int(g['n_cases'])):
        c = {k[len(f'c{ci}_'):]: g[k] for k in g.files if k.startswith(f'c{ci}_')}
        w, h = (int(v) for v in c['wh'])
        gen = torch.Generator().manual_seed(int(c['seed']))
        img = (torch.rand(h, w, 3, generator=gen) * 255).to(torch.uint8).numpy()
        assert orc.checksum(torch.from_numpy(img.astype(np.float32))) == list(c['in_fp'])
        yield ci, c, img


def test_oracle_reader_matches_reference_goldens(golden_dir):
    n = 0
    for ci, c, img in load_reader_cases(golden_dir):
        out = rdo.read_overlap_image(img, [int(v) for v in c['resize']], bool(c['grayscale']), str(c['align']),
                                     rotation=int(c['rotation']))
        assert out['scales'] == tuple(c['scales']) and out['overlap_scales'] == tuple(c['overlap_scales']), ci
        assert tuple(out['overlap_inp'].shape) == tuple(c['overlap_shape']), ci
        assert tuple(out['inp'].shape) == tuple(c['inp_shape']), ci
        assert orc.checksum(out['overlap_inp']) == list(c['overlap_fp']), ci
        assert orc.checksum(out['inp']) == list(c['inp_fp']), ci
        if 'inp' in c:
            assert np.array_equal(out['inp'].numpy(), c['inp']) and np.array_equal(out['overlap_inp'].numpy(), c['overlap_inp'])
        n += 1
    assert n == 8


def test_library_frame_arithmetic_is_the_references(golden_dir):
    """``oetr_overlap_frame`` (host arithmetic, no GPU): sizes and scale factors bit-equal to the
    reference's Python floats, on the goldens and on a sweep against the oracle."""
    for ci, c, img in load_reader_cases(golden_dir):
        w, h = (int(v) for v in c['wh'])
        fr = pkg.overlap_frame(w, h, [int(v) for v in c['resize']], str(c['align']))
        odd = int(c['rotation']) % 2      # (an odd rotation swaps the matcher picture's axes and `scales`)
        assert fr['scales'][::-1 if odd else 1] == tuple(c['scales']) and fr['overlap_scales'] == tuple(c['overlap_scales']), ci
        assert (1, fr['h_ov'], fr['w_ov'], 3) == tuple(c['overlap_shape']), ci
        assert (fr['h_new'], fr['w_new'])[::-1 if odd else 1] == tuple(c['inp_shape'][2:]), ci
    import random
    rng = random.Random(5)
    for _ in range(300):
        w, h = rng.randrange(1, 3000), rng.randrange(1, 3000)
        resize, align = rng.choice([[640], [512], [-1], [1024]]), rng.choice(['disk', 'loftr', ''])
        ref = rdo.overlap_frame(w, h, resize, align)
        fr = pkg.overlap_frame(w, h, resize, align)
        assert fr == {k: ref[k] for k in fr}, (w, h, resize, align)
    with pytest.raises(ValueError):
        pkg.overlap_frame(640, 480, [640, 480])
    with pytest.raises(ValueError):
        pkg.overlap_frame(0, 480)


@pytest.mark.parametrize('src,dst', [((61, 97), (64, 128)), ((240, 320), (100, 77)), ((50, 50), (200, 170))])
def test_bilinear_restatement_agrees_with_torch(src, dst):
    """The oracle's restatement of OpenCV's float32 INTER_LINEAR vs torch's independent
    implementation of the same half-pixel-centre bilinear (no antialiasing): the only
    cross-check available without cv2 (pixels are parity-unpinned, reader_oracle.py)."""
    g = torch.Generator().manual_seed(3)
    img = torch.rand(*src, 3, generator=g) * 255
    mine = rdo.bilinear_resize(img.numpy(), dst[1], dst[0])
    ref = F.interpolate(img.permute(2, 0, 1)[None], size=dst, mode='bilinear', align_corners=False)[0].permute(1, 2, 0)
    # (torch forms the source coordinate in float32, OpenCV - and the restatement - in double:
    #  weights differ by ~1e-5, times neighbour differences of up to 255 on a random image)
    assert float(np.abs(mine - ref.numpy()).max()) <= 2e-2      # of 255
    assert np.array_equal(rdo.bilinear_resize(img.numpy(), src[1], src[0]), img.numpy())
