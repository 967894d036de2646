"""Host cost of a submitted batch through the module (GPU box): submit into an idle device, hip_streams = 1 and 3,
plus a cProfile of 300 submits in the throughput mode.  python tools/host_cost.py"""
import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval().to(dev)
model.hip_freeze_weights = True
n, hf = 8, 20
f1 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev); f2 = (torch.rand(n, 256, hf, hf) - 0.5).to(dev)
pos = model.pos_encoding(f1).contiguous(); hw = (640, 640)
for k in (1, 3):
    model.hip_flush(); model.hip_streams = k
    for _ in range(30): model.boxes_from_features(f1, f2, pos, pos, hw, hw)
    model.hip_flush(); torch.cuda.synchronize()
    # pure host cost: submit k batches into an idle device (nothing to wait for), many times
    ts = []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k): model.boxes_from_features(f1, f2, pos, pos, hw, hw)
        ts.append((time.perf_counter() - t0) / k)
        model.hip_flush()
    ts.sort()
    print(f'hip_streams={k}: host submit {1e6*ts[len(ts)//2]:.1f} us per batch (median), min {1e6*ts[0]:.1f}')
model.hip_streams = 3
pr = cProfile.Profile(); pr.enable()
for _ in range(300): model.boxes_from_features(f1, f2, pos, pos, hw, hw)
model.hip_flush(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
