"""(Needs the library built from tools/r5_chain.patch: `git apply tools/r5_chain.patch && make -C imagematching_oetr_amd/csrc` - the chained
encoder launch was measured and NOT shipped: profiles/r5_launch_boundary.txt.)
Chained encoder launch (oetr_set_encoder_chain) against one launch per layer: bit-identity of every output over
repeated forwards, serial step time (interleaved regions), the time-out path.   python tools/chain_ab.py [pairs] [reps]"""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
w = model.hot_path_state()
hf = 20
f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev); f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
hw = (hf * 32, hf * 32)
base = pkg.HotPathEngine(w, device=dev)
chain = pkg.HotPathEngine(w, device=dev)
chain.set_encoder_chain(True)


def run(e):
    out = e.forward(f1, f2, pos, pos, hw, hw)
    return [t.clone() for t in out]


ref = run(base)
torch.cuda.synchronize()
bad = 0
for i in range(reps):
    f1.copy_((torch.rand(n, 256, hf, hf, device=dev) - 0.5)) if i % 10 == 9 else None
    if i % 10 == 9:
        ref = run(base)
    got = run(chain)
    if not all(torch.equal(a, b) for a, b in zip(ref, got)):
        bad += 1
torch.cuda.synchronize()
print(f'{n} pairs: {bad} of {reps} chained forwards differ from one launch per layer; flags base {base.query_flags()} chain {chain.query_flags()}')


def timed(e, steps=200):
    for _ in range(20):
        e.forward(f1, f2, pos, pos, hw, hw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        e.forward(f1, f2, pos, pos, hw, hw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


for r in range(3):
    a, b = timed(base), timed(chain)
    print(f'serial step: one launch per layer {a:.1f} us ({n / a * 1e6:.0f} pairs/s), chained {b:.1f} us ({n / b * 1e6:.0f} pairs/s)')
# the time-out path
chain.debug_chain_fault(True)
t0 = time.perf_counter(); out = run(chain); torch.cuda.synchronize(); dt = time.perf_counter() - t0
fl = chain.query_flags()
print(f'fault injected: flags {fl} (FLAG_EXCHANGE = {pkg.FLAG_EXCHANGE}) after {dt * 1e3:.1f} ms')
got = run(chain)
print('next chained forward identical:', all(torch.equal(a, b) for a, b in zip(ref, got)), 'flags', chain.query_flags())
for name, e in (('one launch per layer', base), ('chained', chain)):
    with pkg.KernelTrace(e, max_launches=2000) as tr:
        for _ in range(50):
            e.forward(f1, f2, pos, pos, hw, hw)
        torch.cuda.synchronize()
    print(name, {k: (v[0], round(v[1] / v[0] * 1e3, 2)) for k, v in tr.summary().items()})
