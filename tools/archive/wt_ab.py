"""Write-through (sc1) stores of the encoder's outputs, A/B in ONE process: every
tools/variants/<name>/liboetr_hip.so (built with -DOETR_WT=<mask>, tools/wt_variants.sh) on the same
inputs, interleaved rounds - serial step (eager, one stream), overlapped throughput (3 streams, 64-row
tiles, direct tail: bench.py's `value` setting), per-kernel HIP events, and the boxes compared bit for bit.

    python tools/wt_ab.py base wt7 ... > profiles/r4_wt_stores.txt
"""
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
prec = os.environ.get('PREC', 'f32_split_f16')
libs = {n: hip_engine.load_library(str(REPO / 'tools' / 'variants' / n / 'liboetr_hip.so')) for n in names}


def engine(name, **kw):
    hip_engine._lib = libs[name]
    return pkg.HotPathEngine(w, device=dev, precision=prec, **kw)


def case(n, hf, hf2=None, rounds=4, overlap=True):
    hf2 = hf2 or hf
    f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
    f2 = (torch.rand(n, 256, hf2, hf2) - 0.5).to(dev)
    p1 = model.pos_encoding(f1.cpu()).contiguous().to(dev)
    p2 = model.pos_encoding(f2.cpu()).contiguous().to(dev)
    args = (f1, f2, p1, p2, (hf * 32, hf * 32), (hf2 * 32, hf2 * 32))
    ser = {k: engine(k) for k in names}                       # library rules (serial line of bench.py)
    ovl = {}
    if overlap:
        for k in names:
            e = engine(k, enc_tile=64)
            e.set_tail_mode(2)
            ovl[k] = e
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    step, thr, kern, boxes = {k: [] for k in names}, {k: [] for k in names}, {k: {} for k in names}, {}
    for rnd in range(rounds):
        for k in names:
            e = ser[k]
            for _ in range(5):
                b = e.forward(*args)
            boxes[k] = (b[0].clone(), b[1].clone())
            if rnd == 0:
                with pkg.KernelTrace(e, max_launches=1024) as tr:
                    for _ in range(20):
                        e.forward(*args)
                    torch.cuda.synchronize()
                for kk, v in tr.summary().items():
                    a = kern[k].setdefault(kk, [0, 0.0]); a[0] += v[0]; a[1] += v[1]
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    e.forward(*args)
                torch.cuda.synchronize()
                step[k].append((time.perf_counter() - t0) / 50 * 1e6)
            if overlap:
                e = ovl[k]
                for i in range(12):
                    with torch.cuda.stream(streams[i % 3]):
                        e.forward(*args)
                torch.cuda.synchronize()
                for _ in range(3):
                    t0 = time.perf_counter()
                    for i in range(120):
                        with torch.cuda.stream(streams[i % 3]):
                            e.forward(*args)
                    torch.cuda.synchronize()
                    thr[k].append(120 * n / (time.perf_counter() - t0))
    print(f'## {n} pairs {hf*32}x{hf*32} vs {hf2*32}x{hf2*32}, {prec}')
    ref = boxes[names[0]]
    for k in names:
        same = torch.equal(boxes[k][0], ref[0]) and torch.equal(boxes[k][1], ref[1])
        ks = ' '.join(f'{kk.replace("k_", "")}={v[1] / v[0] * 1e3:.1f}' for kk, v in kern[k].items())
        o = f'overlapped {statistics.median(thr[k]):8.0f} pairs/s (best {max(thr[k]):8.0f})' if overlap else ''
        print(f'  {k:8s} serial step {min(step[k]):7.1f} us (median {statistics.median(step[k]):7.1f}) = {n / min(step[k]) * 1e6:7.0f} pairs/s  {o}  '
              f'boxes {"identical" if same else "DIFFER"}  flags {ser[k].query_flags()}\n           {ks}')
    sys.stdout.flush()


case(8, 20)
case(8, 20, 40)
case(32, 32, rounds=2)
case(1, 20, overlap=False)
case(4, 32, overlap=False)
