"""RCCL on the driver's path.

north_star's metric is "1/2/4/8 MI355X ... RCCL over xGMI used only for the all-gather of per-pair
boxes".  The driver's GPU box has one GPU, so what can be executed there is everything EXCEPT the
second rank: ``librccl`` loaded, a process group bound to the device, ``all_gather_into_tensor`` on the
hot path's own box block, in the three stream arrangements ``BoxGatherer`` has, under
``model.hip_streams = 3`` with a batch that trips its range check in flight - bit-equal to the run
without a group, and not slower than 0.9 x of it (the regression ``profiles/r5_hw_queues.txt`` records: a
gather on a fifth stream halved the overlapped rate).  A second test launches two ranks with torchrun
and skips below two GPUs.

The reference has no inference-time collective (its only distributed code is training DDP,
``train.py:59-74``)."""
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import pytest
import torch
import torch.distributed as dist

import imagematching_oetr_amd as pkg
from imagematching_oetr_amd.parallel import BoxGatherer, forward_sharded, gather_boxes
from oracle import oetr_oracle as orc

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
REPO = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.fixture(scope='module')
def rccl_world1(gpu):
    """An RCCL process group of ONE rank, bound to the device (what bench.py brings up at N > 1)."""
    assert not dist.is_initialized()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1,
                            device_id=gpu)
    yield dist.group.WORLD
    torch.cuda.synchronize()
    dist.destroy_process_group()


def _model(gpu, seed=5):
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    sd = model.state_dict()
    sd.update(orc.make_hot_weights(seed, sharpen=True))
    model.load_state_dict(sd, strict=True)
    model = model.to(gpu)
    model.hip_freeze_weights = True
    return model


def _batches(gpu, count=9):
    shapes = [(8, 20, 20), (3, 13, 13), (2, 10, 20)]
    out = []
    for i in range(count):
        n, h1, h2 = shapes[i % 3]
        f1, f2 = orc.make_features(300 + i, n, h1, h1), orc.make_features(400 + i, n, h2, h2)
        if i == 4:
            f1 = f1 * 4.0e5                       # trips the f16 range guard while its neighbours are in flight
        out.append([t.to(gpu) for t in (f1, f2, orc.position_table(h1, h1), orc.position_table(h2, h2))]
                   + [(h1 * 32, h1 * 32), (h2 * 32, h2 * 32)])
    return out


def _librccl_mapped():
    with open('/proc/self/maps') as f:
        return any('librccl' in line for line in f)


def test_rccl_world1_gather_is_the_no_group_result_bit_for_bit(gpu, rccl_world1):
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    model = _model(gpu)
    batches = _batches(gpu)
    # reference: no gatherer, one batch at a time, the throughput settings
    model.hip_streams, model.hip_throughput = 1, True
    want = []
    for b in batches:
        out = model.boxes_from_features(*b)
        model.hip_flush()
        want.append([t.clone() for t in out])
    # the three stream arrangements of the gather under three batches in flight
    model.hip_streams, model.hip_throughput = 3, None
    for on_stream in (True, False, None):
        gat = BoxGatherer(group=rccl_world1, on_stream=on_stream, model=model)
        got = []
        for b in batches:
            b1, b2 = model.boxes_from_features(*b)
            with torch.cuda.stream(model.hip_batch_stream()):
                done = gat.submit(b1, b2)
            if done is not None:
                got.append(done)
        got += gat.flush_all()
        torch.cuda.synchronize()
        assert len(got) == len(batches), on_stream
        for i, (g, w) in enumerate(zip(got, want)):
            # batch 4 went out AFTER its exact-fp32 re-run (the gatherer waits for the model to settle it)
            assert torch.equal(g[0], w[0]) and torch.equal(g[1], w[1]), (on_stream, i)
    assert _librccl_mapped(), 'the collective did not go through librccl'
    # the blocking helper and the image-level entry
    b1, b2 = want[0]
    g1, g2 = gather_boxes(b1, b2, b1.shape[0], group=rccl_world1)
    assert torch.equal(g1, b1) and torch.equal(g2, b2)
    model.hip_streams = 1
    gen = torch.Generator().manual_seed(3)
    im1, im2 = torch.rand(3, 320, 320, 3, generator=gen).to(gpu), torch.rand(3, 320, 320, 3, generator=gen).to(gpu)
    s1, s2 = forward_sharded(model, im1, im2, group=rccl_world1)
    p1, p2 = model.forward_dummy(im1, im2)
    model.hip_flush()
    # (two trunk runs: MIOpen's convolutions are not run-to-run bit-stable; box tolerance)
    assert float((s1 - p1).abs().max()) <= 5e-2 and float((s2 - p2).abs().max()) <= 5e-2


def test_rccl_world1_gather_does_not_cost_the_overlapped_rate(gpu, rccl_world1):
    """``profiles/r5_hw_queues.txt``: the asynchronous gather was a fifth stream next to the throughput
    mode's three side streams and the caller's, shared a hardware queue and HALVED the overlapped rate
    (30.5 -> 14.7 k pairs/s); the on-stream collective must stay within 10 % of the run without a group."""
    model = _model(gpu)
    n, hf = 8, 20
    f1, f2 = orc.make_features(1, n, hf, hf).to(gpu), orc.make_features(2, n, hf, hf).to(gpu)
    pos = orc.position_table(hf, hf).to(gpu)
    hw = (hf * 32, hf * 32)
    model.hip_streams = 3

    def region(steps, gat):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            b1, b2 = model.boxes_from_features(f1, f2, pos, pos, hw, hw)
            if gat is not None:
                with torch.cuda.stream(model.hip_batch_stream()):
                    gat.submit(b1, b2)
        model.hip_flush()
        if gat is not None:
            gat.flush_all()
        torch.cuda.synchronize()
        return n * steps / (time.perf_counter() - t0)
    gat = BoxGatherer(group=rccl_world1, model=model)
    region(30, None), region(30, gat)                                     # warm both (RCCL's first call sets up channels)
    plain = max(region(200, None) for _ in range(3))
    grouped = max(region(200, gat) for _ in range(3))
    assert grouped >= 0.9 * plain, (grouped, plain)


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ['OETR_REPO'])
import imagematching_oetr_amd as pkg
from imagematching_oetr_amd.parallel import BoxGatherer, shard_bounds
from oracle import oetr_oracle as orc
torch.set_grad_enabled(False)
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
backend = os.environ.get('OETR_TEST_BACKEND', 'nccl')
dev = torch.device('cuda', int(os.environ['LOCAL_RANK']) % torch.cuda.device_count())   # (gloo dry run: both ranks on the one GPU)
torch.cuda.set_device(dev)
if backend == 'nccl':
    dist.init_process_group('nccl', device_id=dev)
else:
    dist.init_process_group(backend)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
sd = model.state_dict(); sd.update(orc.make_hot_weights(5, sharpen=True)); model.load_state_dict(sd, strict=True)
model = model.to(dev); model.hip_streams = 3
n_pairs, hf = 6, 13
f1, f2 = orc.make_features(11, n_pairs, hf, hf), orc.make_features(12, n_pairs, hf, hf)
pos = orc.position_table(hf, hf).to(dev)
lo, hi = shard_bounds(n_pairs, rank, world)
gat = BoxGatherer(model=model)
outs = []
for k in range(5):
    b1, b2 = model.boxes_from_features(f1[lo:hi].to(dev) * (1.0 + 0.1 * k), f2[lo:hi].to(dev), pos, pos, (416, 416), (416, 416))
    with torch.cuda.stream(model.hip_batch_stream()):
        d = gat.submit(b1, b2, n_pairs=n_pairs)
    if d is not None: outs.append(d)
outs += gat.flush_all()
torch.cuda.synchronize()
# every rank holds every pair's boxes; compare with the whole batch computed locally, one batch at a time
model.hip_streams = 1
for k, (g1, g2) in enumerate(outs):
    w1, w2 = model.boxes_from_features(f1.to(dev) * (1.0 + 0.1 * k), f2.to(dev), pos, pos, (416, 416), (416, 416))
    model.hip_flush()
    assert g1.shape == (n_pairs, 4)
    # (a pair's boxes depend on the batch around it in the last bits - automatic tile / tail rules: tolerance)
    assert float((g1 - w1).abs().max()) <= 5e-2 and float((g2 - w2).abs().max()) <= 5e-2, (rank, k)
dist.barrier(); dist.destroy_process_group()
sys.stdout.write(f'rank{rank}-ok-{len(outs)}\n'); sys.stdout.flush()    # (ONE write: two ranks share the pipe)
'''


@pytest.mark.parametrize('backend', ['nccl', 'gloo'])
def test_two_ranks_sharded_hot_path_and_pipelined_gather(tmp_path, backend):
    """Two ranks under torchrun: the sharded hot path in the throughput mode + the pipelined box all-gather
    (`BoxGatherer(model=...)`), every rank checking ALL pairs' boxes against the whole batch computed locally.
    'nccl': one rank per GPU over RCCL / xGMI - needs two GPUs, the driver's 1-GPU box skips it; it is the first
    thing to run on a multi-GPU node.  'gloo': the SAME worker with both ranks on the one GPU (device tensors through
    gloo) - runs everywhere and keeps the worker honest."""
    if backend == 'nccl' and torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (the same worker runs under gloo on one GPU in the other case)')
    env = dict(os.environ, OETR_REPO=str(REPO), HSA_ENABLE_IPC_MODE_LEGACY='0', OETR_TEST_BACKEND=backend)
    script = tmp_path / 'rccl_worker.py'
    script.write_text(WORKER)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'rank0-ok-' in r.stdout and 'rank1-ok-' in r.stdout, r.stdout[-2000:]
