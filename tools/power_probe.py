"""Power, clock and temperature the chip holds while the hot path saturates it (rocm-smi sampled beside a running loop).
    python tools/power_probe.py [pairs] [size] [streams] [seconds]"""
import subprocess, sys, threading, time, re
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import bench
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 1
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
model, weights, f1, f2, p1, p2, hf, hf2 = bench.synthetic_inputs(n, size, size, dev)
model = model.to(dev); model.hip_freeze_weights = True
model.hip_streams = streams
hw = (size, size)
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showtemp', '--showuse'], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if re.search(r'Power|sclk|mclk|Temperature \(Sensor (junction|edge)|GPU use', l)]
            samples.append((time.perf_counter(), keep))
        except Exception as e:   # noqa
            samples.append((time.perf_counter(), [repr(e)]))
        time.sleep(0.5)


idle = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showmaxpower'], capture_output=True, text=True).stdout
print('--- idle'); print('\n'.join(l for l in idle.splitlines() if re.search(r'Power|sclk|Max', l)))
for _ in range(20):
    model.boxes_from_features(f1, f2, p1, p2, hw, hw)
model.hip_flush(); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); steps = 0
while time.perf_counter() - t0 < secs:
    for _ in range(50):
        model.boxes_from_features(f1, f2, p1, p2, hw, hw)
    steps += 50
    model.hip_flush()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
stop = True; th.join()
print(f'--- {n} pairs @{size}x{size}, {streams} stream(s): {n * steps / dt:.0f} pairs/s over {dt:.1f} s')
for t, keep in samples:
    print(f'{t - t0:5.1f} s |', ' | '.join(re.sub(r'\s+', ' ', k) for k in keep))
