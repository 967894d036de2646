"""State pre-reduction launch (oetr_set_state_prereduce 0 / 1) x library variants, ONE process, interleaved
rounds: serial step (eager, one stream) and per-kernel HIP events.  tools/variants/<name>/liboetr_hip.so:
  base  shipped;  wt3  partial states without write-through (sc1) stores;  xcd  k_kv_reduce's blocks on the XCD
  that wrote the image's partials;  wt3xcd  both.
    python tools/prereduce_ab.py base wt3 xcd wt3xcd > profiles/r5_prereduce_ab.txt"""
import os, sys, time, glob, statistics
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import imagematching_oetr_amd as pkg
from imagematching_oetr_amd import hip_engine

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
w = model.hot_path_state()
names = sys.argv[1:] or sorted(Path(p).parent.name for p in glob.glob(str(REPO / 'tools/variants/*/liboetr_hip.so')))
libs = {n: hip_engine.load_library(str(REPO / 'tools' / 'variants' / n / 'liboetr_hip.so')) for n in names}


def case(n, hf, tile=None, rounds=4):
    f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
    f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
    p1 = model.pos_encoding(f1.cpu()).contiguous().to(dev)
    args = (f1, f2, p1, p1, (hf * 32, hf * 32), (hf * 32, hf * 32))
    engs = {}
    for k in names:
        for pre in (0, 1):
            hip_engine._lib = libs[k]
            e = pkg.HotPathEngine(w, device=dev, enc_tile=tile)
            e.set_state_prereduce(pre)
            engs[k, pre] = e
    step, kern, boxes = {k: [] for k in engs}, {}, {}
    for rnd in range(rounds):
        for k, e in engs.items():
            for _ in range(5):
                b = e.forward(*args)
            boxes[k] = b[0].clone()
            if rnd == 0:
                with pkg.KernelTrace(e, max_launches=1024) as tr:
                    for _ in range(20):
                        e.forward(*args)
                    torch.cuda.synchronize()
                kern[k] = ' '.join(f'{kk.replace("k_", "")}={v[1] / v[0] * 1e3:.1f}' for kk, v in tr.summary().items())
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    e.forward(*args)
                torch.cuda.synchronize()
                step[k].append((time.perf_counter() - t0) / 50 * 1e6)
    print(f'## {n} pairs {hf * 32}x{hf * 32}, encoder tile {tile or "auto"}')
    ref = boxes[names[0], 0]
    for k in engs:
        print(f'  {k[0]:8s} prereduce {k[1]}  serial step {min(step[k]):7.1f} us (median {statistics.median(step[k]):7.1f})  '
              f'boxes {"identical" if torch.equal(boxes[k], ref) else "DIFFER"}\n           {kern[k]}')
    sys.stdout.flush()


case(8, 20)
case(8, 20, tile=64)
case(1, 20)
case(16, 20)
