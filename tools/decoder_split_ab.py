"""oetr_set_decoder_split A/B in one process: one workgroup per image (rounds 1-3) against four
(decoder.hip: decoder_body4) - per-kernel HIP-event durations, serial step time, box / hs
differences, status word; then the split form on several streams at once (residency).

    python tools/decoder_split_ab.py > profiles/r4_decoder_split.txt
"""
import os, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import imagematching_oetr_amd as pkg

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
w = model.hot_path_state()
prec = os.environ.get('PREC', 'f32_split_f16')


def case(n, hf, hf2=None, tail=0, streams=1):
    hf2 = hf2 or hf
    f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
    f2 = (torch.rand(n, 256, hf2, hf2) - 0.5).to(dev)
    p1 = model.pos_encoding(f1.cpu()).contiguous().to(dev)
    p2 = model.pos_encoding(f2.cpu()).contiguous().to(dev)
    args = (f1, f2, p1, p2, (hf * 32, hf * 32), (hf2 * 32, hf2 * 32))
    engines = {}
    for k in (1, 4):
        e = pkg.HotPathEngine(w, device=dev, precision=prec)
        e.set_decoder_split(k)
        if tail:
            e.set_tail_mode(tail)
        engines[k] = e
    out = {}
    res = {}
    for rnd in range(3):
        for k, e in engines.items():
            for _ in range(3):
                st = e.forward(*args, stages=True)
            res[k] = {x: st[x].clone() for x in ('hs1', 'hs2', 'box1', 'box2')}
            with pkg.KernelTrace(e, max_launches=1024) as tr:
                for _ in range(20):
                    e.forward(*args)
                torch.cuda.synchronize()
            acc = out.setdefault(k, {})
            for kk, v in tr.summary().items():
                a = acc.setdefault(kk, [0, 0.0]); a[0] += v[0]; a[1] += v[1]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                e.forward(*args)
            torch.cuda.synchronize()
            acc.setdefault('_step', []).append((time.perf_counter() - t0) / 50 * 1e6)
    print(f'## {n} pairs {hf*32}x{hf*32} vs {hf2*32}x{hf2*32}, {prec}, tail mode {tail or "auto"}')
    for k in (1, 4):
        acc = out[k]
        ks = ' '.join(f'{kk.replace("k_", "")}={v[1] / v[0] * 1e3:.1f}' for kk, v in acc.items() if kk != '_step')
        print(f'  split {k}: serial step {min(acc["_step"]):7.1f} us (eager, best of 3 x 50)  flags {engines[k].query_flags()}  {ks}')
    d = {x: float((res[1][x] - res[4][x]).abs().max()) for x in res[1]}
    print(f'  max |split 4 - split 1|: hs {max(d["hs1"], d["hs2"]):.2e} (|hs| max {float(res[1]["hs1"].abs().max()):.2f})  box {max(d["box1"], d["box2"]):.2e} px')
    if streams > 1:
        # the split form on `streams` streams at once: every stream's batch must finish with a clean status word
        e = engines[4]
        ss = [torch.cuda.Stream(device=dev) for _ in range(streams)]
        ref = res[4]['box1']
        bad = 0
        t0 = time.perf_counter()
        for it in range(200):
            outs = []
            for s in ss:
                with torch.cuda.stream(s):
                    outs.append(e.forward(*args))
            for s, o in zip(ss, outs):
                s.synchronize()
                bad += int(not torch.equal(o[0], ref))
        dt = time.perf_counter() - t0
        fl = 0
        for s in ss:
            with torch.cuda.stream(s):
                fl |= e.query_flags()
        print(f'  split 4 on {streams} streams, 200 rounds: {200 * streams * n / dt:.0f} pairs/s, {bad} of {200 * streams} batches differ from the serial boxes, status words OR = {fl}')


case(8, 20, streams=3)
case(8, 20, streams=6)
case(1, 20, streams=3)
case(4, 32)
case(8, 20, 40)          # direct tail (auto): decoder exposed
case(8, 40, tail=2)
case(2, 7, 13)
