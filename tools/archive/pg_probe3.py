"""Stream -> hardware-queue mapping under a process group: order of creation, stream priority, queue count.
usage: pg_probe3.py <order: pg_first|eng_first|no_pg> <prio: 0|-1|2 (2 = CU-mask streams)>     (GPU_MAX_HW_QUEUES from the environment)"""
import os, sys, time
sys.path.insert(0, '.')
order, prio = sys.argv[1], int(sys.argv[2])
on_stream = len(sys.argv) > 3 and sys.argv[3] == 'on_stream'
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
if prio == -1:
    model._side_streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(3)]
elif prio == 2:           # private hardware queues: streams created with a (full) CU mask are never pooled
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)(*([0xffffffff] * words))
    model._side_streams = []
    for _ in range(3):
        h = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
        assert rc == 0, rc
        model._side_streams.append(torch.cuda.ExternalStream(h.value, device=dev))


def region(gather, steps=100):
    g = BoxGatherer(on_stream=on_stream) if gather else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        b1, b2 = model.boxes_from_features(f1, f2, p1, p2, hw, hw)
        if g is not None:
            with torch.cuda.stream(model.hip_batch_stream()):
                g.submit(b1, b2)
    model.hip_flush()
    if g is not None:
        g.flush()
    torch.cuda.synchronize()
    return round(8 * steps / (time.perf_counter() - t0))


def pg():
    dist.init_process_group('nccl', device_id=dev, rank=0, world_size=1)
    t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()


if order == 'pg_first':
    pg()
region(False, 30)
if order == 'eng_first':
    pg()
res = {'none': [region(False) for _ in range(3)]}
if order != 'no_pg':
    region(True, 30)
    res['gather'] = [region(True) for _ in range(3)]
    res['none_again'] = [region(False) for _ in range(2)]
print(order, 'prio', prio, 'on_stream', on_stream, 'queues', os.environ.get('GPU_MAX_HW_QUEUES'), res, flush=True)
if order != 'no_pg':
    dist.destroy_process_group()
