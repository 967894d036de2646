"""Per-launch duration of the stand-alone FullAttention kernel over 60 back-to-back launches after an idle gap (the
socket power controller's transient: profiles/r6_full_attention_p1plane.txt).  python tools/fa_each.py"""
import sys
sys.path.insert(0, '/root/repo')
import torch, time
import imagematching_oetr_amd as pkg
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(5)
L = 4096
q = ((torch.rand(8, L, 8, 32, generator=g) - 0.5) * 4).to(dev)
k = ((torch.rand(8, L, 8, 32, generator=g) - 0.5) * 4).to(dev)
v = ((torch.rand(8, L, 8, 32, generator=g) - 0.5) * 2).to(dev)
torch.cuda.synchronize()
for trial in range(2):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    evs[0].record()
    for i in range(60):
        pkg.full_attention(q, k, v, variant='f32_split_f16')
        evs[i + 1].record()
    torch.cuda.synchronize()
    print('trial', trial, ' '.join(f'{evs[i].elapsed_time(evs[i+1])*1e3:.0f}' for i in range(60)))
    time.sleep(1.0)
# with host gaps: sync after every launch
ts = []
for i in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); pkg.full_attention(q, k, v, variant='f32_split_f16'); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
print('sync after each:', ' '.join(f'{t:.0f}' for t in ts))
