"""GPU tests of the READER half of the pair front end: the device reader vs
the reference goldens / the oracle, and the whole chain decoded pictures -> boxes -> crops with
no host round trip between the stages."""
import numpy as np
import pytest
import torch

import imagematching_oetr_amd as pkg
from oracle import crop_oracle as cro
from oracle import oetr_oracle as orc
from oracle import reader_oracle as rdo
from tests.test_reader_cpu import load_reader_cases

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
PIX_TOL = 2e-6      # pictures live in [0,1]; same algorithm in fp32


def test_device_reader_matches_reference_goldens(gpu, golden_dir):
    n = 0
    for ci, c, img in load_reader_cases(golden_dir):
        resize, gray, align = [int(v) for v in c['resize']], bool(c['grayscale']), str(c['align'])
        rot = int(c['rotation'])
        ref = rdo.read_overlap_image(img, resize, gray, align, rotation=rot)
        for src in (img, img.astype(np.float32)):            # uint8 as decoded, or float32
            (res,) = pkg.read_overlap_images([src], gpu, resize, gray, align, rotation=rot)
            assert res.scales == tuple(c['scales']) and res.overlap_scales == tuple(c['overlap_scales']), ci
            assert tuple(res.overlap_inp.shape) == tuple(c['overlap_shape']), ci
            assert tuple(res.inp.shape) == tuple(c['inp_shape']), ci
            assert float((res.overlap_inp.cpu() - ref['overlap_inp']).abs().max()) <= PIX_TOL, ci
            assert float((res.inp.cpu() - ref['inp']).abs().max()) <= PIX_TOL, ci
            if 'inp' in c:                                    # small case: the fixture holds the pixels
                assert float((res.inp.cpu() - torch.from_numpy(c['inp'])).abs().max()) <= PIX_TOL
        n += 1
    assert n == 8


def test_reader_batches_mixed_sizes_through_one_copy(gpu):
    """Pictures of different sizes: one staging copy, every picture a slot of ONE [n,S,S,3] batch,
    each equal to its single-picture read; device-resident input works too."""
    g = torch.Generator().manual_seed(4)
    imgs = [(torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8) for h, w in ((300, 400), (123, 77), (640, 640), (480, 900))]
    res = pkg.read_overlap_images(imgs, gpu, [256], True, 'disk')
    assert all(r._batch is res[0]._batch for r in res) and tuple(res[0]._batch.shape) == (4, 256, 256, 3)
    for i, r in enumerate(res):
        assert r.overlap_inp.data_ptr() == res[0]._batch[i].data_ptr()
        (one,) = pkg.read_overlap_images([imgs[i].numpy()], gpu, [256], True, 'disk')
        assert torch.equal(one.overlap_inp, r.overlap_inp) and torch.equal(one.inp, r.inp)
        (dev,) = pkg.read_overlap_images([imgs[i].to(gpu)], gpu, [256], True, 'disk')
        assert torch.equal(dev.overlap_inp, r.overlap_inp)
    with pytest.raises(ValueError):
        pkg.read_overlap_images([torch.zeros(8, 8)], gpu)
    with pytest.raises(pkg.OetrError):
        pkg.read_overlap_images(imgs[:1], 'cpu')


def test_raw_pairs_to_crops_without_a_host_round_trip(gpu):
    """decoded pictures -> device reader -> forward_dummy -> overlap_crop: the boxes and the
    matcher's images stay on the GPU from the upload to the crops (the reference goes through
    the host twice per pair: ``read_overlap_image`` and ``tensor_overlap_crop``).  Checked
    against the per-pair loop built from the oracles around the same model."""
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    sd.update(orc.make_hot_weights(6, sharpen=True))
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    g = torch.Generator().manual_seed(21)
    sizes = [((480, 640), (600, 800)), ((333, 517), (480, 640)), ((640, 640), (240, 320))]
    raw = [((torch.rand(*a, 3, generator=g) * 255).to(torch.uint8).numpy(),
            (torch.rand(*b, 3, generator=g) * 255).to(torch.uint8).numpy()) for a, b in sizes]
    out = pkg.forward_pairs_raw(model, raw, resize=[320], grayscale=True, align='disk', max_batch=8)
    assert out['box0'].is_cuda and tuple(out['box0'].shape) == (3, 4)
    for i, (a, b) in enumerate(raw):
        ra, rb = rdo.read_overlap_image(a, [320], True, 'disk'), rdo.read_overlap_image(b, [320], True, 'disk')
        assert out['overlap_scales0'][i] == ra['overlap_scales'] and out['overlap_scales1'][i] == rb['overlap_scales']
        assert out['scales0'][i] == ra['scales'] and out['scales1'][i] == rb['scales']
        assert float((out['inp0'][i].cpu() - ra['inp']).abs().max()) <= PIX_TOL
        e0, e1 = model.forward_dummy(ra['overlap_inp'].to(gpu), rb['overlap_inp'].to(gpu))
        model.hip_flush()
        assert float((out['box0'][i] - e0[0]).abs().max()) <= 5e-2 and float((out['box1'][i] - e1[0]).abs().max()) <= 5e-2
        # the crop step straight from the device-resident results
        crops = pkg.overlap_crop(out['inp0'][i], out['inp1'][i], out['box0'][i], out['box1'][i],
                                 out['overlap_scales0'][i], out['overlap_scales1'][i], True, 1)
        ref = cro.overlap_crop(ra['inp'], rb['inp'], out['box0'][i].cpu(), out['box1'][i].cpu(),
                               ra['overlap_scales'], rb['overlap_scales'], True, 1)
        assert crops.valid == ref['valid']
        for s in (0, 1):
            assert tuple(crops.crop(s).shape) == tuple(ref[f'crop{s}'].shape)
            assert float((crops.crop(s).cpu() - ref[f'crop{s}']).abs().max()) <= 1e-5


def test_raw_pairs_native_size_frames_are_bucketed(gpu):
    """``resize=[-1]`` keeps every picture's own size as its OETR frame (reference
    utils.py:297-298): pairs are then bucketed by shape, boxes still in input order."""
    class Stub:   # forward_dummy's contract, boxes that identify the inputs - DEFERRED like
        _pending = None   # OETR.hip_defer_check: garbage until the next call / hip_flush() fixes it in place

        def parameters(self):
            return iter([torch.zeros(1, device=gpu)])

        def hip_flush(self):
            if self._pending is not None:
                for t, good in self._pending:
                    t.copy_(good)
                self._pending = None

        def forward_dummy(self, a, b):
            self.hip_flush()
            k = torch.arange(4, dtype=torch.float32, device=a.device)
            good = (a.reshape(a.shape[0], -1).mean(1, keepdim=True) + k, b.reshape(b.shape[0], -1).mean(1, keepdim=True) - k)
            out = tuple(torch.full_like(t, 7.0e4) for t in good)      # "overflowed" until settled
            self._pending = list(zip(out, good))
            return out
    g = torch.Generator().manual_seed(8)
    sizes = [((64, 96), (64, 96)), ((32, 32), (64, 96)), ((64, 96), (64, 96)), ((32, 32), (64, 96))]
    raw = [((torch.rand(*a, 3, generator=g) * 255).to(torch.uint8), (torch.rand(*b, 3, generator=g) * 255).to(torch.uint8))
           for a, b in sizes]
    out = pkg.forward_pairs_raw(Stub(), raw, resize=[-1], grayscale=False, align='', max_batch=8)
    for i, (a, b) in enumerate(raw):
        ra, rb = rdo.read_overlap_image(a.numpy(), [-1], False, ''), rdo.read_overlap_image(b.numpy(), [-1], False, '')
        assert abs(float(out['box0'][i, 0]) - float(ra['overlap_inp'].mean())) <= 1e-5, i
        assert abs(float(out['box1'][i, 0]) - float(rb['overlap_inp'].mean())) <= 1e-5, i
        assert tuple(out['inp0'][i].shape) == tuple(ra['inp'].shape) and out['overlap_scales0'][i] == (1.0, 1.0)


def test_raw_pairs_overflowing_batch_is_rerun_before_the_boxes_are_copied(gpu):
    """``forward_pairs_raw`` on the REAL model with a hot path that overflows the f16 operand
    range: forward_dummy defers its range check and re-runs the batch in exact fp32 INTO the
    tensors it returned; the result dict must hold those corrected boxes (round 3 copied the
    boxes by value right after the call and returned the out-of-range ones silently)."""
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    sd.update(orc.make_hot_weights(6, sharpen=True))
    sd['input_proj2.weight'] = sd['input_proj2.weight'] * 4.0e5     # features beyond the f16 range
    sd['input_proj2.bias'] = sd['input_proj2.bias'] * 4.0e5
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    model.hip_neck = False          # (the torch neck: only the HOT PATH's guard is under test)
    g = torch.Generator().manual_seed(5)
    raw = [((torch.rand(160, 200, 3, generator=g) * 255).to(torch.uint8).numpy(),
            (torch.rand(200, 160, 3, generator=g) * 255).to(torch.uint8).numpy()) for _ in range(5)]
    out = pkg.forward_pairs_raw(model, raw, resize=[320], grayscale=True, align='disk', max_batch=2)
    assert bool(torch.isfinite(out['box0']).all()) and bool(torch.isfinite(out['box1']).all())
    # reference: the same pictures through the exact-fp32 engine, synchronously
    model.hip_precision = 'f32'
    model.hip_defer_check = False
    for i, (a, b) in enumerate(raw):
        ra, rb = rdo.read_overlap_image(a, [320], True, 'disk'), rdo.read_overlap_image(b, [320], True, 'disk')
        e0, e1 = model.forward_dummy(ra['overlap_inp'].to(gpu), rb['overlap_inp'].to(gpu))
        assert float((out['box0'][i] - e0[0]).abs().max()) <= 5e-2, i
        assert float((out['box1'][i] - e1[0]).abs().max()) <= 5e-2, i
