"""Follow-up of tools/trunk_1x1_probe.py (bounded): per-op timing of the fused forms torch itself
offers for an eval-mode conv + BatchNorm + ReLU of the ResNet trunk (fp32, 16 images of 640x640):
  * aten.miopen_convolution_relu (MIOpen fusion plan conv + bias + activation, NCHW fp32) on BN-folded weights,
  * torch._addmm_activation (hipBLASLt bias + ReLU epilogue) for the 1x1 convolutions on channels_last maps,
against conv2d + batch_norm + relu_ as the module graph runs them.   python tools/trunk_fuse_probe.py
"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def case(name, n, cin, cout, hw, k, stride):
    x = torch.randn(n, cin, hw, hw, generator=g).to(dev)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    bw, bb = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    rm, rv = torch.randn(cout, generator=g).to(dev), (torch.rand(cout, generator=g) + 0.5).to(dev)
    pad = k // 2
    s = bw / torch.sqrt(rv + 1e-5)
    wf, bf = (w * s[:, None, None, None]).contiguous(), (bb - rm * s).contiguous()

    def ref():
        return F.relu_(F.batch_norm(F.conv2d(x, w, None, stride, pad), rm, rv, bw, bb, False, 0.0, 1e-5))

    def conv_only():
        return F.conv2d(x, w, None, stride, pad)

    def fused():
        return torch.ops.aten.miopen_convolution_relu(x, wf, bf, [stride, stride], [pad, pad], [1, 1], 1)

    def bias_relu():
        return F.relu_(F.conv2d(x, wf, bf, stride, pad))
    out = [f'{name}: conv+bn+relu {timed(ref):.3f} ms', f'conv alone {timed(conv_only):.3f}', f'conv(bias)+relu_ {timed(bias_relu):.3f}']
    try:
        e = float((fused() - ref()).abs().max())
        out.append(f'miopen_convolution_relu {timed(fused):.3f} (max diff {e:.1e})')
    except Exception as ex:
        out.append(f'miopen_convolution_relu FAILED {repr(ex)[:80]}')
    if k == 1 and stride == 1:
        xl = x.contiguous(memory_format=torch.channels_last)
        w2 = wf.reshape(cout, cin)

        def gemm():
            x2 = xl.permute(0, 2, 3, 1).reshape(-1, cin)
            return torch._addmm_activation(bf, x2, w2.t())

        def gemm_plain():
            return F.relu_(F.linear(xl.permute(0, 2, 3, 1), w2, bf))
        e = float((gemm().reshape(n, hw, hw, cout).permute(0, 3, 1, 2) - ref()).abs().max())
        out.append(f'_addmm_activation {timed(gemm):.3f} (max diff {e:.1e}); F.linear+relu_ {timed(gemm_plain):.3f}')
    print('  '.join(out), flush=True)


case('stem 7x7 s2 3->64 @640', 16, 3, 64, 640, 7, 2)
case('layer1 conv1 1x1 256->64 @160', 16, 256, 64, 160, 1, 1)
case('layer1 conv2 3x3 64->64 @160', 16, 64, 64, 160, 3, 1)
case('layer1 conv3 1x1 64->256 @160', 16, 64, 256, 160, 1, 1)
case('layer2 conv2 3x3 128->128 @80', 16, 128, 128, 80, 3, 1)
case('layer3 conv2 3x3 256->256 @40', 16, 256, 256, 40, 3, 1)
case('layer3 conv1 1x1 1024->256 @40', 16, 1024, 256, 40, 1, 1)
