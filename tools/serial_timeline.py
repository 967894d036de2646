"""Kernel timeline of serial forwards (rocprofv3 --kernel-trace --output-format csv): per kernel the
mean duration and the mean gap to the previous kernel's end, for oetr_set_decoder_split 1 and 4.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python tools/serial_timeline.py run
    python tools/serial_timeline.py parse gpurun_out/tl
"""
import csv, glob, os, sys
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
    n, hf = int(os.environ.get('PAIRS', 8)), int(os.environ.get('HF', 20))
    f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
    f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
    pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
    hw = (hf * 32, hf * 32)
    for k in (1, 4, 1, 4):
        e = pkg.HotPathEngine(w, device=dev)
        e.set_decoder_split(k)
        for _ in range(40):
            e.forward(f1, f2, pos, pos, hw, hw)
        torch.cuda.synchronize()
else:
    f = glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ks = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'oetr' in r['Kernel_Name']]
    # split into segments by the decoder kernel's grid: identify by the position of k_decoder_convp in each forward
    import collections
    seq, cur = [], []
    for k in ks:
        if 'k_decoder_consts' in k[0]:
            continue
        cur.append(k)
        if 'k_heat_final' in k[0]:
            seq.append(cur); cur = []
    print(len(seq), 'forwards')
    for seg, name in ((seq[10:40], 'split 1'), (seq[50:80], 'split 4'), (seq[90:120], 'split 1'), (seq[130:160], 'split 4')):
        acc = collections.OrderedDict()
        tot = []
        prev_end = None
        for fw in seg:
            tot.append(fw[-1][2] - fw[0][1])
            for i, (nm, st, en) in enumerate(fw):
                short = nm.split('(')[0][-40:]
                a = acc.setdefault((i, short), [0, 0.0, 0.0])
                a[0] += 1; a[1] += (en - st) / 1e3
                if i: a[2] += (st - fw[i - 1][2]) / 1e3
        print(f'== {name}: first kernel start -> last kernel end {sum(tot) / len(tot) / 1e3:.1f} us')
        for (i, short), a in acc.items():
            print(f'   {i:2d} {short:42s} dur {a[1] / a[0]:6.1f}  gap before {a[2] / a[0]:5.1f}')
