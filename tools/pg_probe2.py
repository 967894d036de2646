"""Which part of BoxGatherer.submit costs the overlapped step its throughput (world 1, RCCL)?"""
import os, sys, time
sys.path.insert(0, '.')
import torch, torch.distributed as dist
import bench
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', str(bench.free_port()))
model, weights, f1, f2, p1, p2, hf, hf2 = bench.synthetic_inputs(8, 640, 640, dev)
model = model.to(dev); model.hip_freeze_weights = True
hw = (640, 640)
model.hip_streams = 3
dist.init_process_group('nccl', device_id=dev, rank=0, world_size=1)
mine_s = torch.zeros(2, 8, 4, device=dev); flat_s = torch.zeros(2, 8, 4, device=dev)
gs = torch.cuda.Stream(device=dev)
ring = [(torch.zeros(2, 8, 4, device=dev), torch.zeros(2, 8, 4, device=dev)) for _ in range(8)]


def region(mode, steps=100):
    works = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        b1, b2 = model.boxes_from_features(f1, f2, p1, p2, hw, hw)
        if mode == 'none':
            continue
        with torch.cuda.stream(model.hip_batch_stream()):
            if mode == 'ctx':
                pass
            elif mode == 'stack':
                mine = torch.stack((b1, b2))
            elif mode == 'stack_empty':
                mine = torch.stack((b1, b2)); flat = torch.empty_like(mine)
            elif mode == 'copy2':                 # two copies into a persistent buffer, no collective
                mine, flat = ring[i % 8]
                mine[0].copy_(b1); mine[1].copy_(b2)
            elif mode == 'coll_static':           # the collective alone on persistent buffers
                works.append(dist.all_gather_into_tensor(flat_s, mine_s, async_op=True))
            elif mode == 'coll_sync':             # not async: the batch stream waits for the collective
                dist.all_gather_into_tensor(flat_s, mine_s)
            elif mode == 'full':
                mine = torch.stack((b1, b2)); flat = torch.empty_like(mine)
                works.append(dist.all_gather_into_tensor(flat, mine, async_op=True))
            elif mode == 'copy_d2d':              # what a world-1 all-gather is on the device
                mine, flat = ring[i % 8]
                flat.copy_(mine, non_blocking=True)
            if len(works) > 4:
                works.pop(0)
    t1 = time.perf_counter()
    model.hip_flush()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return round(8 * steps / (t2 - t0)), round((t1 - t0) / steps * 1e6, 1)


for mode in ('none', 'ctx', 'stack', 'stack_empty', 'copy2', 'copy_d2d', 'coll_static', 'coll_sync', 'full', 'none'):
    region(mode, 20)
    print('%-12s pairs/s, host us/step enqueue:' % mode, region(mode), region(mode), flush=True)
dist.destroy_process_group()
