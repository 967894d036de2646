"""Soak of the overlapped (multi-stream) mode: batches alternating over S streams in the settings
bench.py's `value` uses (64-row tiles, direct tail) and in the latency settings, every batch compared
bit for bit with the boxes of a serial run of the same inputs; status words checked.

    python tools/overlap_soak.py [seconds per mode]
"""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import imagematching_oetr_amd as pkg

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
w = model.hot_path_state()


def inputs(n, hf, hf2, seed):
    g = torch.Generator().manual_seed(seed)
    f1 = (torch.rand(n, 256, hf, hf, generator=g) - 0.5).to(dev)
    f2 = (torch.rand(n, 256, hf2, hf2, generator=g) - 0.5).to(dev)
    p1 = model.pos_encoding(f1.cpu()).contiguous().to(dev)
    p2 = model.pos_encoding(f2.cpu()).contiguous().to(dev)
    return (f1, f2, p1, p2, (hf * 32, hf * 32), (hf2 * 32, hf2 * 32))


def soak(name, streams, tile, tail, split, shapes):
    eng = pkg.HotPathEngine(w, device=dev, enc_tile=tile)
    eng.set_tail_mode(tail); eng.set_decoder_split(split)
    cases = [inputs(*s, seed=100 + i) for i, s in enumerate(shapes)]
    refs = [[t.clone() for t in eng.forward(*c)] for c in cases]
    ss = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    torch.cuda.synchronize()
    t0 = time.time(); runs = bad = 0; rnd = 0
    while time.time() - t0 < budget:
        outs = []
        for si, s in enumerate(ss):
            with torch.cuda.stream(s):
                for k in range(4):                       # four batches in flight per stream
                    ci = (rnd + si + k) % len(cases)
                    outs.append((ci, eng.forward(*cases[ci])))
                    if k < 3:                            # the next forward reuses the output tensors' workspace: copy out first
                        outs[-1] = (ci, [t.clone() for t in outs[-1][1]])
        torch.cuda.synchronize()
        for ci, o in outs:
            runs += 1
            bad += int(not (torch.equal(o[0], refs[ci][0]) and torch.equal(o[1], refs[ci][1])))
        rnd += 1
    flags = 0
    for s in ss:
        with torch.cuda.stream(s):
            flags |= eng.query_flags()
    print(f'{name}: {bad} differing of {runs} batches on {streams} streams in {budget:.0f} s, status words OR = {flags}', flush=True)


S8 = [(8, 20, 20)]
MIX = [(8, 20, 20), (2, 20, 20), (3, 25, 25), (1, 32, 32), (8, 20, 40), (4, 20, 20)]
soak('value settings (64-row tiles, direct tail, split auto), 8 pairs @640', 3, 64, 2, 0, S8)
soak('latency settings (automatic rules), 8 pairs @640', 3, None, 0, 0, S8)
soak('value settings, mixed shapes', 3, 64, 2, 0, MIX)
soak('latency settings, mixed shapes', 3, None, 0, 0, MIX)
soak('decoder split 4 forced, P form, mixed shapes, 4 streams', 4, None, 1, 4, MIX)
soak('decoder split 4 forced, direct tail, mixed shapes, 4 streams', 4, 64, 2, 4, MIX)
