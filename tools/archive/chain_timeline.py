"""(Needs the library built from tools/r5_chain.patch: `git apply tools/r5_chain.patch && make -C imagematching_oetr_amd/csrc` - the chained
encoder launch was measured and NOT shipped: profiles/r5_launch_boundary.txt.)
Kernel timeline of serial forwards with and without the chained encoder launch (rocprofv3 --kernel-trace):
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o tl -- python tools/chain_timeline.py run
    python tools/chain_timeline.py parse $OUT/tl"""
import csv, glob, os, sys, collections
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
if sys.argv[1] == 'run':
    import torch
    import imagematching_oetr_amd as pkg
    torch.set_grad_enabled(False)
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    w = model.hot_path_state()
    n, hf = 8, 20
    f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev); f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
    pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
    hw = (hf * 32, hf * 32)
    for k in (0, 1, 0, 1):
        e = pkg.HotPathEngine(w, device=dev)
        e.set_encoder_chain(bool(k))
        for _ in range(40):
            e.forward(f1, f2, pos, pos, hw, hw)
        torch.cuda.synchronize()
else:
    f = glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    ks = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'oetr' in r['Kernel_Name']]
    seq, cur = [], []
    for k in ks:
        if 'k_decoder_consts' in k[0]:
            continue
        cur.append(k)
        if 'k_heat_final' in k[0]:
            seq.append(cur); cur = []
    print(len(seq), 'forwards')
    for seg, name in ((seq[10:40], 'one launch per layer'), (seq[50:80], 'chained'), (seq[90:120], 'one launch per layer'), (seq[130:160], 'chained')):
        acc = collections.OrderedDict(); tot = []; per = []
        for j, fw in enumerate(seg):
            tot.append(fw[-1][2] - fw[0][1])
            if j: per.append(fw[0][1] - seg[j - 1][0][1])
            for i, (nm, st, en) in enumerate(fw):
                short = nm.split('(')[0][-40:]
                a = acc.setdefault((i, short), [0, 0.0, 0.0])
                a[0] += 1; a[1] += (en - st) / 1e3
                if i: a[2] += (st - fw[i - 1][2]) / 1e3
        print(f'== {name}: first kernel start -> last kernel end {sum(tot) / len(tot) / 1e3:.1f} us; forward period {sum(per) / len(per) / 1e3:.1f} us')
        for (i, short), a in acc.items():
            print(f'   {i:2d} {short:42s} dur {a[1] / a[0]:6.1f}  gap before {a[2] / a[0]:5.1f}')
