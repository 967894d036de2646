"""A few launches of the stand-alone FullAttention kernel (for rocprofv3 passes): python tools/fa_run.py [L] [variant] [launches]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import imagematching_oetr_amd as pkg
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variant = sys.argv[2] if len(sys.argv) > 2 else 'f32_split_f16'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(5)
q = ((torch.rand(8, L, 8, 32, generator=g) - 0.5) * 4).to(dev)
k = ((torch.rand(8, L, 8, 32, generator=g) - 0.5) * 4).to(dev)
v = ((torch.rand(8, L, 8, 32, generator=g) - 0.5) * 2).to(dev)
for _ in range(reps):
    pkg.full_attention(q, k, v, variant=variant)
torch.cuda.synchronize()
