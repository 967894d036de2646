"""Random-shape fuzz of the HIP path vs the CPU oracle (run on the GPU box)."""
import sys, random
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
from oracle import oetr_oracle as orc
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
TOL = dict(memory=2e-4, hs=1e-4, logits=1e-3, cxy=5e-2, tlbr=1e-5, box=5e-2)
engines = {}
bad = 0
for case in range(ncase):
    wseed, sharp = rng.randrange(4), rng.random() < 0.5
    prec = rng.choice(['f32_split_f16', 'f32_split_f16@64', 'f32'])
    key = (wseed, sharp, prec)
    w = orc.make_hot_weights(wseed, sharpen=sharp)
    if key not in engines:
        pr, _, tile = prec.partition('@')
        engines[key] = pkg.HotPathEngine(w, device=dev, precision=pr, enc_tile=int(tile) if tile else None)
    eng = engines[key]
    # size-dependent rules: automatic or pinned at random (tail form, decoder workgroups per image, state pre-reduction)
    tail = rng.choice([0, 0, 1, 2, 3]) if prec != 'f32' else rng.choice([0, 1])
    split = rng.choice([0, 0, 1, 4])
    pre = rng.choice([-1, -1, 0, 1])
    eng.set_tail_mode(tail); eng.set_decoder_split(split); eng.set_state_prereduce(pre)
    n = rng.randrange(1, 10)
    big = 61 if rng.random() < 0.15 else 41
    g1 = (rng.randrange(1, big), rng.randrange(1, big))
    g2 = (rng.randrange(1, 41), rng.randrange(1, 41))
    f1, f2 = orc.make_features(1000 + case, n, *g1), orc.make_features(2000 + case, n, *g2)
    p1, p2 = orc.position_table(*g1), orc.position_table(*g2)
    im1, im2 = (g1[0] * 32, g1[1] * 32), (g2[0] * 32, g2[1] * 32)
    # forward_dummy's masks on half of the cases (padding-style corners, or with random holes)
    m1 = m2 = None
    mk = rng.choice(['', '', 'pad', 'holes'])
    if mk:
        m1, m2 = orc.make_masks(3000 + case, n, *g1, kind=mk), orc.make_masks(4000 + case, n, *g2, kind=mk)
    out = eng.forward(f1.to(dev), f2.to(dev), p1.to(dev), p2.to(dev), im1, im2, stages=True, mask1=m1, mask2=m2)
    ref = orc.hot_path(f1, f2, w, im1, im2, return_stages=True, mask1=m1, mask2=m2)
    msgs = []
    for s in ('1', '2'):
        for k, tol in TOL.items():
            e = (out[k + s].cpu().double().reshape(ref[k + s].shape) - ref[k + s].double()).abs().max().item()
            if not (e <= tol):
                msgs.append(f'{k}{s}={e:.2e}')
        b, r = out['box' + s].cpu(), ref['box' + s]
        area = (r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1])
        iou = orc.bbox_iou_aligned(b, r)
        if not (iou[area > 1] >= 1 - 1e-3).all():
            msgs.append(f'iou{s}={iou.tolist()}')
    status = 'OK ' if not msgs else 'BAD'
    bad += bool(msgs)
    fl = eng.query_flags()
    if fl:
        msgs.append(f'flags={fl}'); status = 'BAD'; bad += 1
    print(f'{status} case {case}: n={n} {g1} {g2} w{wseed}{"s" if sharp else ""} {prec} tail={tail} split={split} pre={pre} masks={mk or "-"} ' + ' '.join(msgs), flush=True)
print(f'{bad} bad of {ncase}')
