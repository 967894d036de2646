"""What would the encoder GEMMs lose if the two CROSS terms of the fp32-class split ran on the fp8 matrix pipe?
CPU emulation inside the fp64 oracle (the method of tools/site_drift.py, which predicts the GPU's per-site table to two
digits).  Today a product is  a.w = ah.wh + (ah.wl + al.wh)  with f16 planes, three f16 MFMAs.  The cross terms are
2^-11 of the result and only need a few significant bits: with both of their operands in fp8 (e4m3, one power-of-two scale
per 32-element block along K, what v_mfma_scale_f32_32x32x64_f8f6f4 takes) they would cost ONE f16-MFMA equivalent instead
of two (fp8 runs at twice the f16 rate) - 2 units per product instead of 3.  This script measures the drift of that
arithmetic on the seeded golden cases, next to the shipped split (cross terms in f16) and to single f16.
Test infrastructure only (imports oracle/).   python tools/fp8_cross_drift.py [out.json]
"""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, torch.nn.functional as F
import tools.site_drift as sd
from oracle import oetr_oracle as orc
torch.set_grad_enabled(False)

F8 = torch.float8_e4m3fn
F8_MAX = 448.0


def q8_blocks(t, block=32, fmt=F8, fmax=F8_MAX):
    """fp8 with one power-of-two scale per `block` elements of the last (K) dimension."""
    shp = t.shape
    k = shp[-1]
    b = t.reshape(-1, k // block, block)
    amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.floor(torch.log2(fmax / amax)))
    q = (b * scale).to(torch.float32).to(fmt).to(t.dtype) / scale
    return q.reshape(shp)


MODE = {}


def lin(site, x, w, b=None):
    mode = MODE.get(site)
    if mode is None:
        return F.linear(x, w, b)
    xh = x.to(torch.float16).to(x.dtype); wh = w.to(torch.float16).to(w.dtype)
    xl, wl = x - xh, w - wh
    if mode == 'f16':
        return F.linear(xh, wh, b)
    if mode == 'split':      # the shipped arithmetic: lo planes are f16 too
        xl16 = (xl * 2048).to(torch.float16).to(x.dtype) / 2048
        wl16 = (wl * 2048).to(torch.float16).to(w.dtype) / 2048
        return F.linear(xh, wh, b) + F.linear(xh, wl16) + F.linear(xl16, wh)
    if mode.startswith('fp8'):
        fmt, fmax = (torch.float8_e5m2, 57344.0) if mode.endswith('e5m2') else (F8, F8_MAX)
        q = lambda t: q8_blocks(t, 32, fmt, fmax)
        return F.linear(xh, wh, b) + F.linear(q(xh), q(wl)) + F.linear(q(xl), q(wh))
    if mode == 'one_cross_fp8':   # a.w = ah.wh + [ah | al].[wl ; wh] as ONE fp8 product of K = 512 (same thing, stated once)
        return F.linear(xh, wh, b) + F.linear(torch.cat([q8_blocks(xh), q8_blocks(xl)], -1), torch.cat([q8_blocks(wl), q8_blocks(wh)], -1))
    raise ValueError(mode)


sd.lin = lin
SITES = ['q', 'k', 'v', 'merge', 'mlp1', 'mlp2']

if __name__ == '__main__':
    base = [sd.run(c) for c in sd.CASES]
    rows = []

    def evaluate(name, modes):
        global MODE
        MODE.clear(); MODE.update(modes)
        worst = dict(memory=0, hs=0, cxy=0, tlbr=0, iou=1.0)
        for ci, case in enumerate(sd.CASES):
            o = sd.run(case)
            for k in ('memory', 'hs', 'cxy', 'tlbr'):
                worst[k] = max(worst[k], float((o[k] - base[ci][k]).abs().max()))
            worst['iou'] = min(worst['iou'], float(orc.bbox_iou_aligned(o['box'], base[ci]['box']).min()))
        worst['arithmetic'] = name
        rows.append(worst)
        print(f"{name:58s} mem {worst['memory']:.2e} hs {worst['hs']:.2e} cxy {worst['cxy']:.2e} tlbr {worst['tlbr']:.2e} 1-iou {1-worst['iou']:.2e}", flush=True)

    evaluate('all six encoder GEMMs: split (shipped, 3 f16 MFMAs)', {s: 'split' for s in SITES})
    evaluate('all six: cross terms fp8 e4m3, block-32 scales (2 units)', {s: 'fp8' for s in SITES})
    evaluate('all six: cross terms fp8 e5m2, block-32 scales', {s: 'fp8_e5m2' for s in SITES})
    evaluate('all six: single f16 (1 MFMA)', {s: 'f16' for s in SITES})
    evaluate('q, k single f16 (the policy) + the rest fp8 cross', dict({s: 'fp8' for s in SITES}, q='f16', k='f16'))
    for s in SITES:
        evaluate(f'{s}: fp8 cross, the rest split', dict({t: 'split' for t in SITES}, **{s: 'fp8'}))
    json.dump(rows, open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/fp8_cross_drift.json', 'w'), indent=1)
