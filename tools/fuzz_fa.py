"""Random-shape fuzz of the stand-alone FullAttention kernels vs the fp64 oracle (run on the GPU box):
python tools/fuzz_fa.py [seed] [cases] [all: print every case]"""
import sys, random
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from oracle import oetr_oracle as orc
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
worst = {'f32': 0.0, 'f32_split_f16': 0.0}
for case in range(ncase):
    n = rng.randrange(1, 4)
    L = rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, rng.randrange(1, 700)])
    S = rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, rng.randrange(1, 2100)])
    qs, ks = rng.choice([1, 4, 4, 12]), rng.choice([1, 4, 4, 12])
    g = torch.Generator().manual_seed(rng.randrange(1 << 30))
    q = (torch.rand(n, L, 8, 32, generator=g) - 0.5) * qs
    k = (torch.rand(n, S, 8, 32, generator=g) - 0.5) * ks
    if rng.random() < 0.3:      # scores that grow / shrink along S: the lazy maximum's rescale path
        k = k * torch.linspace(*rng.choice([(0.1, 4.0), (4.0, 0.1)]), S).view(1, S, 1, 1)
    v = (torch.rand(n, S, 8, 32, generator=g) - 0.5) * 2
    ref = orc.full_attention(q.double(), k.double(), v.double())
    drift = float((orc.full_attention(q, k, v).double() - ref).abs().max())       # torch fp32's own error on this case
    line = f'case {case}: n={n} L={L} S={S} q*{qs} k*{ks} torch_f32 {drift:.1e}'
    ok = True
    for variant in ('f32', 'f32_split_f16'):
        out = pkg.full_attention(q.to(dev), k.to(dev), v.to(dev), variant=variant)
        err = float((out.cpu().double() - ref).abs().max())
        worst[variant] = max(worst[variant], err / max(drift, 2e-7))
        line += f'  {variant} {err:.1e}'
        if not (err <= max(5e-6, 4 * drift)) or not torch.isfinite(out).all():
            ok = False
    if not ok:
        bad += 1
    if not ok or case >= ncase - 3 or len(sys.argv) > 3:
        print(('OK  ' if ok else 'BAD ') + line)
print(f'{bad} bad of {ncase}; worst error / max(torch fp32 drift, 2e-7): ' + ', '.join(f'{k} {v:.1f}x' for k, v in worst.items()))
