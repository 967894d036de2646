"""Opt-in trunk arithmetic of the drop-in module (``hip_trunk_dtype`` / ``hip_trunk_channels_last`` /
``hip_trunk_fp32_stages``; host code by north_star - the trunk stays torch / MIOpen).  The measured
drift / speed table is profiles/r4_trunk_autocast.txt; here: the options do what they say."""
import pytest
import torch

import imagematching_oetr_amd as pkg

pytestmark = pytest.mark.gpu


def test_trunk_options(gpu):
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval().to(gpu)
    img = torch.rand(2, 128, 160, 3, device=gpu)
    def close(a, b):      # (MIOpen may pick another convolution algorithm from one call to the next: fp32 rounding only)
        return float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
    with torch.no_grad():
        model.trunk(img)
        ref = model.trunk(img)                              # defaults: the reference's fp32 sequence
        assert close(ref, model.backbone(img)) and ref.dtype == torch.float32
        # every stage pinned to fp32 under an autocast setting = the stage-by-stage path of trunk(),
        # which must be the reference's sequence (backbone.py:159-174): the same tensor
        model.hip_trunk_dtype = 'float16'
        model.hip_trunk_fp32_stages = ('layer0', 'layer1', 'layer2', 'layer3')
        assert close(model.trunk(img), ref)
        # autocast changes the numbers a little, never the shape / dtype / finiteness
        for stages in ((), ('layer2', 'layer3'), ('layer0',)):
            model.hip_trunk_fp32_stages = stages
            out = model.trunk(img)
            assert out.shape == ref.shape and out.dtype == torch.float32 and out.is_contiguous()
            assert torch.isfinite(out).all()
            rel = float((out - ref).abs().max() / ref.abs().max())
            assert 1e-4 < rel < 2e-2, (stages, rel)
        # fewer 16-bit stages, less drift
        model.hip_trunk_fp32_stages = ()
        all16 = float((model.trunk(img) - ref).abs().mean())
        model.hip_trunk_fp32_stages = ('layer1', 'layer2', 'layer3')
        one16 = float((model.trunk(img) - ref).abs().mean())
        assert one16 < all16
