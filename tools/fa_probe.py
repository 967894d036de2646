"""Stand-alone FullAttention kernel (csrc/attention.hip: k_full_attention_split): error against the fp64
oracle on the golden-test shapes and launch time at L = S = 1024 / 4096 (8 images), for the library
named by OETR_HIP_LIB (default: the shipped one).

    python tools/fa_probe.py            # on the GPU box
"""
import sys
import time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import imagematching_oetr_amd as pkg
from oracle import oetr_oracle as orc

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
worst = 0.0
for (n, L, S) in [(2, 400, 400), (2, 1, 400), (2, 400, 1600), (2, 1024, 1024), (1, 300, 1000), (3, 33, 95), (1, 1, 1)]:
    gen = torch.Generator().manual_seed(L * 1000 + S)
    q = (torch.rand(n, L, 8, 32, generator=gen) - 0.5) * 4
    k = (torch.rand(n, S, 8, 32, generator=gen) - 0.5) * 4
    v = (torch.rand(n, S, 8, 32, generator=gen) - 0.5) * 2
    ref = orc.full_attention(q.double(), k.double(), v.double())
    errs = {}
    for variant in ('f32', 'f32_split_f16'):
        out = pkg.full_attention(q.to(dev), k.to(dev), v.to(dev), variant=variant)
        errs[variant] = float((out.cpu().double() - ref).abs().max())
    # what torch fp32 itself does on the CPU (the reference's arithmetic)
    errs['torch_f32'] = float((orc.full_attention(q, k, v).double() - ref).abs().max())
    worst = max(worst, errs['f32_split_f16'])
    print(f'n={n} L={L} S={S}: ' + '  '.join(f'{a} {b:.2e}' for a, b in errs.items()), flush=True)
# sharper scores (larger |q.k|): the regime where P is concentrated on few keys
gen = torch.Generator().manual_seed(7)
q = (torch.rand(2, 400, 8, 32, generator=gen) - 0.5) * 12
k = (torch.rand(2, 400, 8, 32, generator=gen) - 0.5) * 12
v = (torch.rand(2, 400, 8, 32, generator=gen) - 0.5) * 2
ref = orc.full_attention(q.double(), k.double(), v.double())
for variant in ('f32', 'f32_split_f16'):
    e = float((pkg.full_attention(q.to(dev), k.to(dev), v.to(dev), variant=variant).cpu().double() - ref).abs().max())
    print(f'sharp x12 L=S=400: {variant} {e:.2e}')
print(f'worst f32_split_f16 error {worst:.2e} (golden tolerance 5e-6)')
for L in (1024, 4096):
    n = 8
    g = torch.Generator().manual_seed(5)
    q = ((torch.rand(n, L, 8, 32, generator=g) - 0.5) * 4).to(dev)
    k = ((torch.rand(n, L, 8, 32, generator=g) - 0.5) * 4).to(dev)
    v = ((torch.rand(n, L, 8, 32, generator=g) - 0.5) * 2).to(dev)
    flop = 4.0 * n * 8 * L * L * 32
    for variant in ('f32', 'f32_split_f16'):
        fn = lambda: pkg.full_attention(q, k, v, variant=variant)
        for _ in range(60):      # (behind the power controller's transient: tools/fa_each.py)
            fn()
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        peak = 157.3 if variant == 'f32' else 2500.0 / 3
        print(f'L={L} {variant}: {best * 1e3:.1f} us  {flop / best / 1e9:.1f} TFLOP/s  frac {flop / best / 1e9 / peak:.3f}', flush=True)
