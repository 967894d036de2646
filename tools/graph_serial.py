"""Serial step latency: stream launches vs one replayed HIP graph of the same 12 launches."""
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
n, hf = 8, 20
f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev); f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
hw = (hf * 32, hf * 32)
for prec, tile in (('f32_split_f16', 32), ('f32_split_f16', 64), ('f32_split_qk16', 32)):
    eng = pkg.HotPathEngine(model.hot_path_state(), device=dev, precision=prec, enc_tile=tile)
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        for _ in range(5): b = eng.forward(f1, f2, pos, pos, hw, hw)
        s.synchronize()
        def timed(fn, steps=200):
            s.synchronize(); t0 = time.perf_counter()
            for _ in range(steps): fn()
            s.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
        t_stream = min(timed(lambda: eng.forward(f1, f2, pos, pos, hw, hw)) for _ in range(5))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = eng.forward(f1, f2, pos, pos, hw, hw)
        g.replay(); s.synchronize()
        assert torch.equal(out[0], b[0])
        t_graph = min(timed(g.replay) for _ in range(5))
    print(f'{prec}@{tile}: stream launches {t_stream:.4f} ms/step, graph replay {t_graph:.4f} ms/step')
