"""Timeline of the PRODUCT path's serial step (OETR.boxes_from_features, hip_streams = 1, deferred
check): every device dispatch of a step - library kernels AND the runtime's copy / fill kernels -
with its mean duration and the mean idle gap in front of it, plus the host's enqueue time per step.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python tools/step_timeline.py run
    python tools/step_timeline.py parse gpurun_out/tl
"""
import collections
import csv
import glob
import os
import sys
import time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

if sys.argv[1] == 'run':
    import torch
    import imagematching_oetr_amd as pkg
    torch.set_grad_enabled(False)
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval().to(dev)
    model.hip_freeze_weights = True
    n, hf = int(os.environ.get('PAIRS', 8)), int(os.environ.get('HF', 20))
    f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
    f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
    pos = model.pos_encoding(f1).contiguous()
    hw = (hf * 32, hf * 32)
    steps = int(os.environ.get('STEPS', 200))
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.boxes_from_features(f1, f2, pos, pos, hw, hw)
        t1 = time.perf_counter()
        model.hip_flush()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'rep {rep}: host enqueue {1e6 * (t1 - t0) / steps:.1f} us/step, region {1e6 * (t2 - t0) / steps:.1f} us/step',
              flush=True)
else:
    f = glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ks = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
    fin = [i for i, k in enumerate(ks) if 'k_heat_final' in k[0]]
    # a step = everything from behind one k_heat_final up to and including the next
    steps = [ks[a + 1:b + 1] for a, b in zip(fin[:-1], fin[1:])]
    steps = steps[len(steps) // 3:]                  # the last two thirds: warmed up
    shape = collections.Counter(tuple(k[0] for k in s) for s in steps).most_common(1)[0][0]
    steps = [s for s in steps if tuple(k[0] for k in s) == shape]
    print(len(steps), 'steps of', len(shape), 'dispatches')
    per = []
    for i in range(1, len(steps)):
        per.append(steps[i][-1][2] - steps[i - 1][-1][2])
    print(f'k_heat_final end -> next k_heat_final end: {sum(per) / len(per) / 1e3:.1f} us')
    tot_d = tot_g = 0.0
    for j, name in enumerate(shape):
        d = sum(s[j][2] - s[j][1] for s in steps) / len(steps) / 1e3
        g = [s[j][1] - (s[j - 1][2] if j else None or 0) for s in steps] if j else None
        if j:
            gap = sum(s[j][1] - s[j - 1][2] for s in steps) / len(steps) / 1e3
        else:
            gap = sum(steps[i][0][1] - steps[i - 1][-1][2] for i in range(1, len(steps))) / (len(steps) - 1) / 1e3
        tot_d += d
        tot_g += gap
        print(f'  {j:2d} {name.split("(")[0][-52:]:52s} dur {d:6.2f}  gap before {gap:5.2f}')
    print(f'  sum of durations {tot_d:.1f} us, sum of gaps {tot_g:.1f} us')
