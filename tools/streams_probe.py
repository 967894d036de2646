"""Throughput of the hot path with consecutive batches alternating over S HIP streams."""
import sys, time
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
f1 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev); f2 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev)
pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
for S in (1, 2, 3, 4):
    engs = [pkg.HotPathEngine(w, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    def run(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % S]):
                engs[i % S].forward(f1, f2, pos, pos, (640, 640), (640, 640))
    run(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 400; run(K); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f'streams={S}: {dt*1e3:.4f} ms/step  {n/dt:.0f} pairs/s')
