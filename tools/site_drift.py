"""Per-GEMM-site precision study on the CPU ORACLE (fp64): round the operands of one site (or a
combination) to f16 / bf16 inside the oracle's graph and record the drift of memory / hs / cxy /
tlbr and the worst 1 - IoU of the boxes against the unrounded run, on the seeded golden cases.
Predicts the GPU table (tools/site_variants.sh -> profiles/r3_site_drift.jsonl) to two digits;
DESIGN.md 3.10.  Test infrastructure only (imports oracle/).
    python tools/site_drift.py [out.json]
"""
import sys, itertools, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, torch.nn.functional as F
from oracle import oetr_oracle as orc
torch.set_grad_enabled(False)
torch.set_num_threads(8)

def q16(t, kind):
    if kind == 'f16': return t.to(torch.float16).to(t.dtype)
    if kind == 'bf16': return t.to(torch.bfloat16).to(t.dtype)
    return t

# site policy: dict site -> (act_kind, w_kind) ; None = exact
POL = {}
def lin(site, x, w, b=None):
    a, ww = POL.get(site, (None, None))
    return F.linear(q16(x, a), q16(w, ww), b)

def linear_attention(q, k, v, eps=orc.ATTN_EPS):
    S = v.shape[1]
    fq = F.elu(q) + 1; fk = F.elu(k) + 1; vs = v / S
    sa, sb = POL.get('state', (None, None))
    kv = torch.einsum('nshd,nshv->nhdv', q16(fk, sa), q16(vs, sb))
    z = 1 / (torch.einsum('nlhd,nhd->nlh', fq, fk.sum(dim=1)) + eps)
    aa, ab = POL.get('apply', (None, None))
    return (torch.einsum('nlhd,nhdv->nlhv', q16(fq, aa), q16(kv, ab)) * z.unsqueeze(-1) * S).contiguous()

def encoder_layer(x, src, x_pos, s_pos, w, p):
    q = orc._ln(x, w, p + 'pre_norm_q') + x_pos
    kv = orc._ln(src, w, p + 'pre_norm_kv') + s_pos
    Q = orc._heads(lin('q', q, w[p + 'q_proj.weight']))
    K = orc._heads(lin('k', kv, w[p + 'k_proj.weight']))
    V = orc._heads(lin('v', kv, w[p + 'v_proj.weight']))
    msg = linear_attention(Q, K, V).reshape(x.shape)
    x = x + lin('merge', msg, w[p + 'merge.weight'])
    h = F.gelu(lin('mlp1', orc._ln(x, w, p + 'norm2'), w[p + 'mlp.0.weight']))
    return x + lin('mlp2', h, w[p + 'mlp.2.weight'])

def run(case, dtype=torch.float64):
    (tag, wseed, sharp, fseed, n, g1, g2, im1, im2) = case
    w = orc.cast_weights(orc.make_hot_weights(wseed, sharpen=sharp), dtype)
    f1 = orc.make_features(fseed, n, *g1).to(dtype); f2 = orc.make_features(fseed + 100, n, *g2).to(dtype)
    p1 = orc.position_table(*g1, dtype=dtype); p2 = orc.position_table(*g2, dtype=dtype)
    x1, x2 = orc.tokens(f1), orc.tokens(f2); pp1, pp2 = orc.tokens(p1), orc.tokens(p2)
    for i in range(8):
        p = f'transformer.encoder.{i}.'
        if i % 2 == 0:
            x1 = encoder_layer(x1, x1, pp1, pp1, w, p); x2 = encoder_layer(x2, x2, pp2, pp2, w, p)
        else:
            y1 = encoder_layer(x1, x2, pp1, pp2, w, p); y2 = encoder_layer(x2, x1, pp2, pp1, w, p); x1, x2 = y1, y2
    hs = []
    for mem, mpos, qe in ((x1, pp1, w['query_embed1.weight']), (x2, pp2, w['query_embed2.weight'])):
        qpos = qe.unsqueeze(0).repeat(n, 1, 1); tgt = torch.zeros_like(qpos)
        for i in range(2):
            tgt = orc.decoder_layer(tgt, mem, qpos, mpos, w, f'transformer.decoder.layers.{i}.')
        hs.append(tgt)
    lg1 = orc.heatmap_logits(hs[0], x1, *g1, w); lg2 = orc.heatmap_logits(hs[1], x2, *g2, w)
    c1 = orc.soft_argmax(lg1, *g1, im1[0]); c2 = orc.soft_argmax(lg2, *g2, im2[0])
    t1, t2 = orc.size_regression(hs[0], w), orc.size_regression(hs[1], w)
    b1 = orc.box_tlbr_to_xyxy(c1, t1, *im1); b2 = orc.box_tlbr_to_xyxy(c2, t2, *im2)
    return dict(memory=torch.cat([x1.flatten(), x2.flatten()]), hs=torch.cat([hs[0].flatten(), hs[1].flatten()]),
                cxy=torch.cat([c1, c2]), tlbr=torch.cat([t1, t2]), box=torch.cat([b1, b2]))

CASES = [
    ('s0_20x20', 0, False, 10, 2, (20, 20), (20, 20), (640, 640), (640, 640)),
    ('s1_20x20_sharp', 1, True, 11, 2, (20, 20), (20, 20), (640, 640), (640, 640)),
    ('s3_32x32_sharp', 3, True, 13, 2, (32, 32), (32, 32), (1024, 1024), (1024, 1024)),
    ('s4_15x20_25x10', 4, True, 14, 3, (15, 20), (25, 10), (480, 640), (800, 320)),
]
def evaluate(pol, base):
    global POL
    POL = pol
    worst = dict(memory=0, hs=0, cxy=0, tlbr=0, iou=1.0)
    for ci, case in enumerate(CASES):
        o = run(case)
        for k in ('memory', 'hs', 'cxy', 'tlbr'):
            worst[k] = max(worst[k], float((o[k] - base[ci][k]).abs().max()))
        iou = orc.bbox_iou_aligned(o['box'], base[ci]['box'])
        worst['iou'] = min(worst['iou'], float(iou.min()))
    return worst

if __name__ == '__main__':
    POL = {}
    base = [run(c) for c in CASES]
    sites = ['q', 'k', 'v', 'merge', 'mlp1', 'mlp2', 'state', 'apply']
    rows = []
    def show(name, pol):
        r = evaluate(pol, base); r['policy'] = name; rows.append(r)
        print(f"{name:40s} mem {r['memory']:.2e} hs {r['hs']:.2e} cxy {r['cxy']:.2e} tlbr {r['tlbr']:.2e} 1-iou {1-r['iou']:.2e}", flush=True)
    for kind in ('f16', 'bf16'):
        show(f'all {kind}', {s: (kind, kind) for s in sites})
        for s in sites:
            show(f'{s}: act+w {kind}', {s: (kind, kind)})
        for s in sites:
            show(f'{s}: act {kind} only', {s: (kind, None)})
        for s in sites:
            show(f'{s}: w {kind} only', {s: (None, kind)})
    show('q+k f16 (the policy, without decoder K)', {'q': ('f16', 'f16'), 'k': ('f16', 'f16')})
    show('q+k f16, state act', {'q': ('f16', 'f16'), 'k': ('f16', 'f16'), 'state': ('f16', None)})
    show('attention contractions bf16 (configs[2] literally)', {'state': ('bf16', 'bf16'), 'apply': ('bf16', 'bf16')})
    show('attention contractions f16', {'state': ('f16', 'f16'), 'apply': ('f16', 'f16')})
    json.dump(rows, open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/site_drift_cpu.json', 'w'), indent=1)
