"""What a launch boundary costs inside the serial step, and the clock the chip holds: real-time stamps
(s_memrealtime, 100 MHz; library built with -DOETR_PHASE_TIMING=3) against cycle stamps (-DOETR_PHASE_TIMING=1).
After a full forward the stamp buffer holds phase A (slots 8..12) of the LAST <B,A> launch and phase B (slots 0..8)
of the <B,dec> launch behind it: max(slot 12) -> min(slot 0) is the idle time between the last workgroup of one
launch and the first of the next.

    tools/variants.sh timing "-DOETR_PHASE_TIMING" timing_rt "-DOETR_PHASE_TIMING=3"
    python tools/launch_boundary.py [tile]
CHAIN=1 (and the per-step lines) need the library built from tools/r5_chain.patch (the chained encoder launch: studied, not shipped).
"""
import ctypes, os, sys, subprocess
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if len(sys.argv) <= 2:
    for v in ('timing_rt', 'timing'):
        subprocess.run([sys.executable, __file__, str(tile), v], check=True)
    sys.exit(0)
variant = sys.argv[2]
os.environ['OETR_HIP_LIB'] = str(REPO / f'tools/variants/{variant}/liboetr_hip.so')
import numpy as np, torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
n = 8
f1 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev); f2 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev)
pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
eng = pkg.HotPathEngine(model.hot_path_state(), device=dev, enc_tile=tile)
import os as _os
CHAIN = _os.environ.get('CHAIN') == '1'
if CHAIN:
    eng.set_encoder_chain(True)
lib = eng.lib
rows = tile or 32
nb = 16 * (-(-400 // rows))
for _ in range(20):
    eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(50):
    eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
ev[1].record(); torch.cuda.synchronize()
step_us = ev[0].elapsed_time(ev[1]) * 1e3 / 50
buf = (ctypes.c_longlong * (16 * nb))()
lib.oetr_debug_read_tbuf.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.oetr_debug_read_tbuf(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(nb, 16).astype(np.float64)
unit = 0.01 if variant == 'timing_rt' else 1.0          # us per tick / cycles
name = 'us' if variant == 'timing_rt' else 'cycles'
print(f'--- {variant}{" CHAINED" if CHAIN else ""}: {rows}-row tiles, {nb} workgroups, serial step {step_us:.1f} us (stamped build)')
a_end, b_start, b_end = t[:, 12], t[:, 0], t[:, 8]
print(f'  <B,A> launch: phase A per workgroup {((t[:, 12] - t[:, 8]) * unit).mean():.1f} {name}; last workgroup leaves {(a_end.max() - a_end.min()) * unit:.1f} {name} after the first')
print(f'  boundary: last end of <B,A> -> first start of <B,dec>: {(b_start.min() - a_end.max()) * unit:.2f} {name}; first -> last start {(b_start.max() - b_start.min()) * unit:.2f} {name}')
print(f'  <B,dec> launch: phase B per workgroup {((b_end - b_start) * unit).mean():.1f} {name} (min {((b_end - b_start) * unit).min():.1f}, max {((b_end - b_start) * unit).max():.1f}); first start -> last phase-B end {(b_end.max() - b_start.min()) * unit:.1f} {name}')
d = (b_start - a_end) * unit
print(f'  per workgroup, end of its phase A -> start of its next step: mean {d.mean():.2f} {name} (min {d.min():.2f}, max {d.max():.2f})')
ph = np.diff(t[:, :13], axis=1) * unit
print('  phases (mean per workgroup):', ' '.join(f'{x:.1f}' for x in ph.mean(axis=0)))
if CHAIN and variant == 'timing_rt':
    lib.oetr_debug_read_tbuf_chain.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.oetr_debug_read_tbuf_chain(buf, nb) == 0
    c = np.frombuffer(buf, dtype=np.int64).reshape(nb, 16)[:, :9].astype(np.float64) * 0.01
    print('  chained launch: first step start -> last kernel end %.1f us' % (c[:, 8].max() - c[:, 0].min()))
    print('  step durations per workgroup (mean):', ' '.join('%.1f' % x for x in np.diff(c, axis=1).mean(axis=0)))
    print('  step start skew over workgroups (max - min):', ' '.join('%.1f' % (c[:, i].max() - c[:, i].min()) for i in range(9)))
