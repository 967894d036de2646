"""Host-side trunk settings (torch / MIOpen; not the HIP path): time per 16 images @640x640 and the
max difference of the trunk output against the default setting."""
import sys, time, copy
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval().to(dev)
img = torch.rand(16, 640, 640, 3, device=dev)
def timed(fn, it=8):
    for _ in range(3): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3, out
GF = 16 * 53.5
t, ref = timed(lambda: model.backbone(img))
print(f'default (NCHW, unfused BN)      : {t:7.2f} ms  {GF / t:6.1f} TFLOP/s', flush=True)
# eval-mode BatchNorm folded into the preceding convolution
from torch.nn.utils.fusion import fuse_conv_bn_eval
def fold(module):
    for name, child in list(module.named_children()):
        fold(child)
    names = list(module._modules)
    for a, b in zip(names, names[1:]):
        ma, mb = module._modules[a], module._modules[b]
        if isinstance(ma, torch.nn.Conv2d) and isinstance(mb, torch.nn.BatchNorm2d):
            module._modules[a] = fuse_conv_bn_eval(ma, mb)
            module._modules[b] = torch.nn.Identity()
bb2 = copy.deepcopy(model.backbone)
fold(bb2)
t, out = timed(lambda: bb2(img))
print(f'BN folded into conv             : {t:7.2f} ms  {GF / t:6.1f} TFLOP/s  max diff {float((out - ref).abs().max()):.2e} (|x| max {float(ref.abs().max()):.2f})', flush=True)
bb3 = copy.deepcopy(bb2).to(memory_format=torch.channels_last)
t, out = timed(lambda: bb3(img))
print(f'BN folded + channels_last       : {t:7.2f} ms  {GF / t:6.1f} TFLOP/s  max diff {float((out - ref).abs().max()):.2e}', flush=True)
bb4 = copy.deepcopy(model.backbone).to(memory_format=torch.channels_last)
t, out = timed(lambda: bb4(img))
print(f'channels_last (unfused)         : {t:7.2f} ms  {GF / t:6.1f} TFLOP/s  max diff {float((out - ref).abs().max()):.2e}', flush=True)
torch.backends.cudnn.benchmark = True
t0 = time.time(); bb2(img); torch.cuda.synchronize(); first = time.time() - t0
t, out = timed(lambda: bb2(img))
print(f'BN folded + MIOpen benchmark    : {t:7.2f} ms  {GF / t:6.1f} TFLOP/s  max diff {float((out - ref).abs().max()):.2e}  (first call {first:.1f} s)', flush=True)
t0 = time.time(); bb3(img); torch.cuda.synchronize(); first = time.time() - t0
t, out = timed(lambda: bb3(img))
print(f'BN folded + CL + benchmark      : {t:7.2f} ms  {GF / t:6.1f} TFLOP/s  max diff {float((out - ref).abs().max()):.2e}  (first call {first:.1f} s)', flush=True)
