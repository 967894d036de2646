"""Bounded probe: the trunk is host code and stays torch - but 93.5 % of forward_dummy is
MIOpen fp32 convolution at 0.385 of the fp32 peak, and about two thirds of ResNet-50[:layer3]'s FLOPs are 1x1
convolutions, i.e. plain GEMMs over the flattened map.  Run them as channels_last `F.linear` (hipBLASLt fp32)
instead of MIOpen convolutions, without and with BatchNorm folded into the GEMM (weight scale + bias), and
report trunk time, end-to-end pairs/s and parity against the default route.  Adopt only if end to end moves >= 15 %.

    python tools/trunk_1x1_probe.py > profiles/r6_trunk_1x1_probe.txt
"""
import sys, time, types
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
import imagematching_oetr_amd as pkg
from oracle import oetr_oracle as orc

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
sd = model.state_dict()
sd.update(orc.make_hot_weights(5, sharpen=True))
model.load_state_dict(sd, strict=True)
model = model.to(dev)
n = 8
g = torch.Generator().manual_seed(1)
im1, im2 = torch.rand(n, 640, 640, 3, generator=g).to(dev), torch.rand(n, 640, 640, 3, generator=g).to(dev)
both = torch.cat([im1, im2])


def timed(fn, reps=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def linear_1x1(conv, bn=None):
    """A 1x1 stride-1 Conv2d (optionally with its eval-mode BatchNorm folded in) as F.linear over NHWC."""
    w = conv.weight.reshape(conv.weight.shape[0], -1)
    b = None
    if bn is not None:
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = (w * s[:, None]).contiguous()
        b = (bn.bias - bn.running_mean * s).contiguous()

    def run(x):      # x: [N,C,H,W] in channels_last memory format -> same
        y = F.linear(x.permute(0, 2, 3, 1), w, b)
        return y.permute(0, 3, 1, 2)
    return run


def patch(trunk, fold):
    """Replace forward of every bottleneck: conv1 / conv3 (and a stride-1 downsample conv) as GEMMs."""
    saved = []
    for layer in (trunk.layer1, trunk.layer2, trunk.layer3):
        for blk in layer:
            c1 = linear_1x1(blk.conv1, blk.bn1 if fold else None)
            c3 = linear_1x1(blk.conv3, blk.bn3 if fold else None)
            ds = None
            if blk.downsample is not None and blk.downsample[0].stride == (1, 1):
                ds = linear_1x1(blk.downsample[0], blk.downsample[1] if fold else None)

            def fwd(self, x, c1=c1, c3=c3, ds=ds, fold=fold):
                if self.downsample is None:
                    skip = x
                elif ds is not None:
                    skip = ds(x) if fold else self.downsample[1](ds(x))
                else:
                    skip = self.downsample(x)
                y = c1(x)
                y = self.relu(y if fold else self.bn1(y))
                y = self.relu(self.bn2(self.conv2(y)))
                y = c3(y)
                if not fold:
                    y = self.bn3(y)
                return self.relu(y + skip)
            saved.append((blk, blk.forward))
            blk.forward = types.MethodType(fwd, blk)
    return saved


def unpatch(saved):
    for blk, f in saved:
        blk.forward = f


bb = model.backbone
ref = bb(both).clone()
t_ref = timed(lambda: bb(both))
model.hip_flush()
e2e_ref = timed(lambda: (model.forward_dummy(im1, im2), model.hip_flush()))
box_ref = [b.clone() for b in model.forward_dummy(im1, im2)]
model.hip_flush()
print(f'default (NCHW, MIOpen): trunk {t_ref * 1e3:.2f} ms per 16 images; forward_dummy {n / e2e_ref:.0f} pairs/s')

rows = []
for name, cl, fold, lin in (('channels_last, MIOpen for everything', True, False, False),
                            ('channels_last, 1x1 convs as F.linear', True, False, True),
                            ('channels_last, 1x1 convs as F.linear with BatchNorm folded in', True, True, True)):
    model.hip_trunk_channels_last = cl
    saved = patch(bb.encoder, fold) if lin else []
    try:
        x = both.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)

        def trunk_only():
            y = x
            if bb.cfg.NORM_INPUT:
                y = (y - 0.45) / 0.225
            return bb.layer3(bb.layer2(bb.layer1(bb.layer0(y))))
        out = trunk_only()
        err = float((out - ref).abs().max()) / float(ref.abs().max())
        t = timed(trunk_only)
        e2e = timed(lambda: (model.forward_dummy(im1, im2), model.hip_flush()))
        b = model.forward_dummy(im1, im2)
        model.hip_flush()
        iou = torch.cat([orc.bbox_iou_aligned(b[0].cpu(), box_ref[0].cpu()), orc.bbox_iou_aligned(b[1].cpu(), box_ref[1].cpu())])
        print(f'{name}: trunk {t * 1e3:.2f} ms ({t_ref / t:.2f}x); forward_dummy {n / e2e:.0f} pairs/s ({e2e_ref / e2e:.2f}x); '
              f'trunk output max rel err {err:.1e}; boxes 1 - IoU vs default {float(1 - iou.min()):.1e}')
    finally:
        unpatch(saved)
        model.hip_trunk_channels_last = False
