"""Throughput of the hot path with consecutive batches alternating over S HIP streams,
eager launches vs one captured hipGraph per stream."""
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
eng = pkg.HotPathEngine(w, device=dev)
K = 400
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(S)]
    def run(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % S]):
                eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
    run(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(K); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    # host-side enqueue cost alone
    t0 = time.perf_counter(); run(K); t_host = (time.perf_counter() - t0) / K; torch.cuda.synchronize()
    # graphs: one per stream
    graphs = []
    for s in streams:
        with torch.cuda.stream(s):
            eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
        graphs.append((g, out))
    def rung(k):
        for i in range(k):
            with torch.cuda.stream(streams[i % S]):
                graphs[i % S][0].replay()
    rung(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); rung(K); torch.cuda.synchronize()
    dg = (time.perf_counter() - t0) / K
    print(f'streams={S}: eager {dt*1e3:.4f} ms/step {n/dt:.0f} pairs/s (host enqueue {t_host*1e3:.3f} ms/step) | graphs {dg*1e3:.4f} ms/step {n/dg:.0f} pairs/s')
