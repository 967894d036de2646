import sys, time, os
sys.path.insert(0, '/root/repo')
import torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval().to(dev)
img = torch.rand(16, 640, 640, 3, device=dev)
def timed(fn, it=5):
    for _ in range(2): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3
print('default            : backbone(16 imgs) %.2f ms' % timed(lambda: model.backbone(img)))
torch.backends.cudnn.benchmark = True
t0 = time.time(); model.backbone(img); torch.cuda.synchronize(); print('  first call with benchmark=True took %.1f s' % (time.time() - t0))
print('cudnn.benchmark    : backbone(16 imgs) %.2f ms' % timed(lambda: model.backbone(img)))
m2 = model.to(memory_format=torch.channels_last)
t0 = time.time(); m2.backbone(img); torch.cuda.synchronize(); print('  first call channels_last took %.1f s' % (time.time() - t0))
print('+channels_last     : backbone(16 imgs) %.2f ms' % timed(lambda: m2.backbone(img)))
