"""Soak of the module's throughput mode (model.hip_streams = 3, hip_queue_depth = 2: two batches queued per side stream): batches of
RANDOM shapes - every stream sees its workspace re-carved and regrown while its previous batch is still queued or running -, some of
them overflow-injected (status word trips, exact-fp32 re-run corrects the returned tensors in place), every batch compared bit for
bit with what the same settings return one batch at a time.      python tools/throughput_soak.py [seconds] [seed]
"""
import sys, time, random
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import imagematching_oetr_amd as pkg
from oracle import oetr_oracle as orc

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
sd = model.state_dict()
sd.update(orc.make_hot_weights(5, sharpen=True))
model.load_state_dict(sd, strict=True)
model = model.to(dev)
model.hip_freeze_weights = True

shapes = [(1, 5, 7), (3, 13, 13), (8, 20, 20), (2, 10, 20), (4, 32, 20), (6, 20, 9), (2, 40, 40), (8, 7, 7)]
pool = []
for i, (n, h1, h2) in enumerate(shapes):
    for trip in (False, True) if i % 3 == 1 else (False,):
        f1, f2 = orc.make_features(100 + i, n, h1, h1), orc.make_features(200 + i, n, h2, h2)
        if trip:
            f1 = f1 * 4.0e5                      # a GEMM operand beyond the f16 range
        pool.append([t.to(dev) for t in (f1, f2, orc.position_table(h1, h1), orc.position_table(h2, h2))]
                    + [(h1 * 32, h1 * 32), (h2 * 32, h2 * 32)])
# one batch at a time, the same engine settings
model.hip_streams, model.hip_throughput = 1, True
want = []
for b in pool:
    out = model.boxes_from_features(*b)
    model.hip_flush()
    want.append([t.clone() for t in out])
model.hip_streams, model.hip_throughput = 3, None
assert model.hip_queue_depth == 2
t0, batches, bad, rounds = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    order = [rng.randrange(len(pool)) for _ in range(rng.randrange(1, 40))]
    outs = [model.boxes_from_features(*pool[i]) for i in order]
    if rng.random() < 0.5:
        model.hip_flush()
    else:                                        # leave the queue full across rounds, settle through the next submits
        pass
    model.hip_flush()
    torch.cuda.synchronize()
    for i, o in zip(order, outs):
        batches += 1
        if not (torch.equal(o[0], want[i][0]) and torch.equal(o[1], want[i][1])):
            bad += 1
            print(f'DIFFERS: round {rounds} pool entry {i}', flush=True)
    rounds += 1
assert model.engine().query_flags() == 0
print(f'{bad} differing of {batches} batches in {rounds} rounds, {time.time() - t0:.0f} s '
      f'({len(pool)} pool entries, {sum(1 for b in pool if float(b[0].abs().max()) > 1e4)} overflow-injected; hip_streams = 3, hip_queue_depth = 2)')
sys.exit(1 if bad else 0)
