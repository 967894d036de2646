"""Where does the overlapped step go when a process group is up (world 1, RCCL)?  Variants of bench.py's step."""
import os, sys, time
sys.path.insert(0, '.')
import torch, torch.distributed as dist
import bench
import imagematching_oetr_amd as pkg
from imagematching_oetr_amd.parallel import BoxGatherer
torch.set_grad_enabled(False)
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', str(bench.free_port()))
model, weights, f1, f2, p1, p2, hf, hf2 = bench.synthetic_inputs(8, 640, 640, dev)
model = model.to(dev); model.hip_freeze_weights = True
hw = (640, 640)
model.hip_streams = 3
gs = torch.cuda.Stream(device=dev)


def plain(steps=100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        model.boxes_from_features(f1, f2, p1, p2, hw, hw)
    model.hip_flush(); torch.cuda.synchronize()
    return 8 * steps / (time.perf_counter() - t0)


plain(30)
print('before init_process_group:', [round(plain()) for _ in range(3)], flush=True)
dist.init_process_group('nccl', device_id=dev, rank=0, world_size=1)
print('after init_process_group :', [round(plain()) for _ in range(3)], flush=True)
t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
print('after first collective   :', [round(plain()) for _ in range(3)], flush=True)


BAR = [False]
G = BoxGatherer()


def region(mode, steps=100):
    g = G if BAR[0] else BoxGatherer()
    if BAR[0]:
        dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        b1, b2 = model.boxes_from_features(f1, f2, p1, p2, hw, hw)
        if mode == 'side':
            with torch.cuda.stream(model.hip_batch_stream()):
                g.submit(b1, b2)
        elif mode == 'gstream':
            gs.wait_stream(model.hip_batch_stream())
            with torch.cuda.stream(gs):
                g.submit(b1, b2)
        elif mode == 'side_nowait':
            with torch.cuda.stream(model.hip_batch_stream()):
                g._pending = None
                g.submit(b1, b2)
    model.hip_flush()
    if mode != 'none':
        with torch.cuda.stream(gs):
            g.flush()
    if BAR[0]:
        g.flush()
        dist.barrier()
    torch.cuda.synchronize()
    return 8 * steps / (time.perf_counter() - t0)


for mode in ('none', 'side', 'gstream', 'side_nowait', 'none', 'side', 'gstream'):
    region(mode, 20)
    print(mode, round(region(mode)), flush=True)
BAR[0] = True
for mode in ('none', 'side', 'none', 'side'):
    region(mode, 20)
    print('with dist.barrier + shared gatherer:', mode, round(region(mode)), flush=True)
dist.destroy_process_group()
