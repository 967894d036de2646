"""N>1 path on CPU: world_size-2 gloo processes exercise the pair sharding
and the box all-gather (the only collective on the path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imagematching_oetr_amd.parallel import (bucket_by_shape, gather_boxes,
                                             shard_bounds)


def test_shard_bounds_cover_every_pair_once():
    for n in (0, 1, 5, 8, 64, 67):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_bounds(n, r, world)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
            sizes = [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0]
                     for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_bucket_by_shape_groups_mixed_scale_pairs():
    shapes = [((640, 640), (640, 640)), ((640, 640), (1280, 1280)),
              ((640, 640), (640, 640)), ((640, 640), (1280, 1280))]
    b = bucket_by_shape(shapes)
    assert list(b.values()) == [[0, 2], [1, 3]]


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(n_pairs, rank, world)
        idx = torch.arange(lo, hi, dtype=torch.float32)
        box1 = torch.stack([idx, idx + 0.25, idx + 0.5, idx + 0.75], 1)
        box2 = -box1
        g1, g2 = gather_boxes(box1, box2, n_pairs)
        q.put((rank, g1.tolist(), g2.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_pairs', [5, 8])
def test_gather_boxes_world2_gloo(n_pairs):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx = torch.arange(n_pairs, dtype=torch.float32)
    expect1 = torch.stack([idx, idx + 0.25, idx + 0.5, idx + 0.75], 1)
    for rank, g1, g2 in results:
        assert torch.equal(torch.tensor(g1), expect1), rank
        assert torch.equal(torch.tensor(g2), -expect1), rank


def _pipe_worker(rank, world, port, q, on_stream=None):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from imagematching_oetr_amd.parallel import BoxGatherer
        g = BoxGatherer(on_stream=on_stream)
        outs = []
        for k in range(3):                       # three batches of 2 pairs per rank
            b1 = torch.full((2, 4), float(10 * k + rank))
            done = g.submit(b1, -b1)
            assert (done is None) == (k == 0)
            if done is not None:
                outs.append(done[0][:, 0].tolist())
        outs.append(g.flush()[0][:, 0].tolist())
        assert g.flush() is None
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('on_stream', [None, True])
def test_pipelined_box_gatherer_world2_gloo(on_stream):
    """on_stream=True: the blocking collective of the throughput mode (ordered on the submitting stream);
    same hand-out order as the asynchronous one."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, q, on_stream)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outs in results:
        assert outs == [[0.0, 0.0, 1.0, 1.0], [10.0, 10.0, 11.0, 11.0], [20.0, 20.0, 21.0, 21.0]]


class _DeferringModel:
    """The part of ``OETR`` the gatherer talks to, with the deferred check's behaviour: ``batch()``
    returns box tensors that hold garbage if the batch "tripped" and are corrected IN PLACE when the
    batch is settled - oldest first, once ``keep`` newer batches are in flight, or at ``hip_flush()``."""

    def __init__(self, keep):
        self.keep, self._inflight = keep, []

    def _settle_down_to(self, keep):
        while len(self._inflight) > keep:
            boxes, good = self._inflight.pop(0)
            for t, g in zip(boxes, good):
                t.copy_(g)

    def batch(self, good1, good2, tripped):
        self._settle_down_to(self.keep - 1)
        out = (torch.full_like(good1, 7.0e4), torch.full_like(good2, 7.0e4)) if tripped \
            else (good1.clone(), good2.clone())
        self._inflight.append((out, (good1, good2)))
        return out

    def hip_settled(self, boxes):
        return not any(boxes is e[0][0] or boxes is e[0][1] for e in self._inflight)

    def hip_flush(self):
        self._settle_down_to(0)


def _deferred_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from imagematching_oetr_amd.parallel import BoxGatherer
        model = _DeferringModel(keep=3)                  # hip_streams = 3
        g = BoxGatherer(model=model)
        outs, issued_at = [], []
        for k in range(7):                               # seven batches of 2 pairs per rank
            good = torch.full((2, 4), float(10 * k + rank))
            b1, b2 = model.batch(good, -good, tripped=(rank == 1 and k in (1, 4)))   # rank 1 trips twice, rank 0 never
            done = g.submit(b1, b2)
            issued_at.append(len(g._waiting))
            if done is not None:
                outs.append((done[0][:, 0].tolist(), done[1][:, 0].tolist()))
        outs += [(d[0][:, 0].tolist(), d[1][:, 0].tolist()) for d in g.flush_all()]
        assert g.flush() is None and not g._waiting and not g._issued
        q.put((rank, outs, issued_at))
    finally:
        dist.destroy_process_group()


def test_gatherer_issues_a_batch_only_once_the_model_has_settled_it_world2_gloo():
    """under the deferred check a tripped batch is corrected in place AFTER it was returned;
    ``BoxGatherer(model=...)`` holds a batch's collective back until the model has settled it, so no
    rank ever receives the stale boxes - and since every rank submits and settles in the same order the
    collectives line up although only rank 1 trips."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_deferred_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outs, waiting in results:
        assert len(outs) == 7, rank
        for k, (o1, o2) in enumerate(outs):              # [rank 0's 2 pairs, rank 1's 2 pairs], settled values
            assert o1 == [10.0 * k, 10.0 * k, 10.0 * k + 1, 10.0 * k + 1], (rank, k, o1)
            assert o2 == [-v for v in o1], (rank, k)
        assert waiting == [1, 2, 3, 3, 3, 3, 3], (rank, waiting)   # three batches stay back: the ones in flight


def _unequal_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from imagematching_oetr_amd.parallel import BoxGatherer
        g = BoxGatherer()
        outs = []
        for k, n_pairs in enumerate((5, 7, 4)):      # 3+2, 4+3, then an equal 2+2 batch
            lo, hi = shard_bounds(n_pairs, rank, world)
            idx = torch.arange(lo, hi, dtype=torch.float32) + 100 * k
            b1 = torch.stack([idx, idx + 0.25, idx + 0.5, idx + 0.75], 1)
            done = g.submit(b1, -b1, n_pairs=n_pairs)
            if done is not None:
                outs.append((done[0].tolist(), done[1].tolist()))
        last = g.flush()
        outs.append((last[0].tolist(), last[1].tolist()))
        try:                                          # a shard of the wrong size is an error, not a hang
            g.submit(torch.zeros(1, 4), torch.zeros(1, 4), n_pairs=5)
            bad = False
        except ValueError:
            bad = True
        q.put((rank, outs, bad))
    finally:
        dist.destroy_process_group()


def test_pipelined_box_gatherer_unequal_shards_world2_gloo():
    """BoxGatherer with contiguous shards whose sizes differ by one: padded
    to the largest shard for the one all_gather_into_tensor, padding dropped afterwards."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_unequal_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outs, bad in results:
        assert bad, rank
        for k, n_pairs in enumerate((5, 7, 4)):
            idx = torch.arange(n_pairs, dtype=torch.float32) + 100 * k
            expect = torch.stack([idx, idx + 0.25, idx + 0.5, idx + 0.75], 1)
            assert torch.equal(torch.tensor(outs[k][0]), expect), (rank, k)
            assert torch.equal(torch.tensor(outs[k][1]), -expect), (rank, k)


class _StubModel:
    """forward_dummy with the reference's contract ([N,H,W,3] images -> two [N,4]
    boxes) computed from the images alone, so that sharding is observable.

    It DEFERS like ``OETR.forward_dummy`` under ``hip_defer_check``: the tensors it returns
    hold out-of-range garbage until ``hip_flush()`` (or the next call) corrects them in place -
    a consumer that copies the boxes before settling ships the garbage."""

    def __init__(self):
        self._pending = None

    def hip_flush(self):
        if self._pending is not None:
            for t, good in self._pending:
                t.copy_(good)
            self._pending = None

    def forward_dummy(self, image1, image2, mask1=None, mask2=None):
        self.hip_flush()
        m1 = image1.reshape(image1.shape[0], -1).mean(1, keepdim=True)
        m2 = image2.reshape(image2.shape[0], -1).mean(1, keepdim=True)
        if mask1 is not None:     # masks travel with their pairs: observable in the boxes
            assert mask1.shape[0] == image1.shape[0] and mask2.shape[0] == image2.shape[0]
            m1 = m1 + 10.0 * mask1.reshape(mask1.shape[0], -1).float().sum(1, keepdim=True)
            m2 = m2 + 10.0 * mask2.reshape(mask2.shape[0], -1).float().sum(1, keepdim=True)
        k = torch.arange(4, dtype=torch.float32)
        good = (m1 + k, m2 - k)
        out = tuple(torch.full_like(t, 7.0e4) for t in good)      # "overflowed" until settled
        self._pending = list(zip(out, good))
        return out


def _sharded_worker(rank, world, port, n_pairs, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from imagematching_oetr_amd.parallel import forward_sharded
        g = torch.Generator().manual_seed(3)          # every rank holds the full batch
        im1 = torch.rand(n_pairs, 6, 5, 3, generator=g)
        im2 = torch.rand(n_pairs, 4, 7, 3, generator=g)
        b1, b2 = forward_sharded(_StubModel(), im1, im2)
        mk1 = (torch.arange(n_pairs * 6).reshape(n_pairs, 2, 3) % 5 > 1)
        mk2 = (torch.arange(n_pairs * 4).reshape(n_pairs, 2, 2) % 3 > 0)
        c1, c2 = forward_sharded(_StubModel(), im1, im2, mask1=mk1, mask2=mk2)
        q.put((rank, b1.tolist(), b2.tolist(), c1.tolist(), c2.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_pairs', [7, 8])
def test_forward_sharded_world2_gloo(n_pairs):
    """forward_sharded = model.forward_dummy on this rank's contiguous shard + the box
    all-gather: every rank must end up with the boxes of ALL pairs, in input order,
    equal to the unsharded call."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, n_pairs, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(3)
    im1 = torch.rand(n_pairs, 6, 5, 3, generator=g)
    im2 = torch.rand(n_pairs, 4, 7, 3, generator=g)
    ref = _StubModel()
    e1, e2 = ref.forward_dummy(im1, im2)
    ref.hip_flush()
    assert float(e1.abs().max()) < 100.0      # settled values
    mk1 = (torch.arange(n_pairs * 6).reshape(n_pairs, 2, 3) % 5 > 1)
    mk2 = (torch.arange(n_pairs * 4).reshape(n_pairs, 2, 2) % 3 > 0)
    f1, f2 = ref.forward_dummy(im1, im2, mk1, mk2)
    ref.hip_flush()
    assert not torch.equal(f1, e1)
    for rank, b1, b2, c1, c2 in results:
        assert torch.equal(torch.tensor(b1), e1) and torch.equal(torch.tensor(b2), e2), rank
        assert torch.equal(torch.tensor(c1), f1) and torch.equal(torch.tensor(c2), f2), rank   # masked: sharded with the pairs


def _world8_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from imagematching_oetr_amd.parallel import BoxGatherer, forward_sharded
        from imagematching_oetr_amd.pipeline import forward_pairs_sharded
        out = {}
        for n_pairs in (64, 61):          # BASELINE configs[2]: 64 pairs over 8 GPUs = 8 per rank; and an indivisible job
            lo, hi = shard_bounds(n_pairs, rank, world)
            idx = torch.arange(lo, hi, dtype=torch.float32)
            box1 = torch.stack([idx, idx + 0.25, idx + 0.5, idx + 0.75], 1)
            g1, g2 = gather_boxes(box1, -box1, n_pairs)
            out['gather', n_pairs] = (g1.tolist(), g2.tolist(), hi - lo)
            gat = BoxGatherer()
            assert gat.submit(box1, -box1, n_pairs=n_pairs) is None
            nxt = gat.submit(box1 + 1000, -box1, n_pairs=n_pairs)        # returns batch 0
            last = gat.flush()
            out['pipe', n_pairs] = (nxt[0].tolist(), last[0].tolist())
            g = torch.Generator().manual_seed(3)                       # every rank holds the full batch
            im1, im2 = torch.rand(n_pairs, 6, 5, 3, generator=g), torch.rand(n_pairs, 4, 7, 3, generator=g)
            b1, b2 = forward_sharded(_StubModel(), im1, im2)
            out['sharded', n_pairs] = (b1.tolist(), b2.tolist())
        # BASELINE configs[4] as a mixed-scale job: two shapes interleaved, 61 pairs, chunks of 8
        g = torch.Generator().manual_seed(4)
        pairs = []
        for i in range(61):
            hw2 = (8, 8) if i % 3 else (16, 16)          # "640 vs 640" and "640 vs 1280"
            pairs.append((torch.rand(8, 8, 3, generator=g), torch.rand(*hw2, 3, generator=g)))
        m1, m2 = forward_pairs_sharded(_StubModel(), pairs, max_batch=8)
        out['mixed'] = (m1.tolist(), m2.tolist())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_world8_gloo_dry_run_of_the_scaling_job():
    """the first real 8-GPU attempt must not fail on arithmetic.  Eight gloo ranks
    run what `bench.py --gpus 8` and the pair front-end run over RCCL: `gather_boxes` and the pipelined
    `BoxGatherer` at BASELINE configs[2]'s real split (64 pairs = 8 per rank) and at an indivisible 61
    (shards of 8 and 7, padded to 8 for the one all_gather_into_tensor), `forward_sharded`, and the
    mixed-scale bucketing of configs[4] (`forward_pairs_sharded`: two shapes interleaved, every bucket
    sharded, one gather) - every rank ends with every pair's boxes in input order."""
    world, port = 8, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(r for r, _ in results) == list(range(world))
    ref = _StubModel()
    g = torch.Generator().manual_seed(4)
    pairs = []
    for i in range(61):
        hw2 = (8, 8) if i % 3 else (16, 16)
        pairs.append((torch.rand(8, 8, 3, generator=g), torch.rand(*hw2, 3, generator=g)))
    from imagematching_oetr_amd.pipeline import forward_pairs
    x1, x2 = forward_pairs(ref, pairs, max_batch=8)
    for rank, out in results:
        for n_pairs in (64, 61):
            idx = torch.arange(n_pairs, dtype=torch.float32)
            expect = torch.stack([idx, idx + 0.25, idx + 0.5, idx + 0.75], 1)
            g1, g2, mine = out['gather', n_pairs]
            assert mine == (8 if n_pairs == 64 else (8 if rank < 5 else 7)), (rank, mine)
            assert torch.equal(torch.tensor(g1), expect) and torch.equal(torch.tensor(g2), -expect), rank
            nxt, last = out['pipe', n_pairs]
            assert torch.equal(torch.tensor(nxt), expect) and torch.equal(torch.tensor(last), expect + 1000), rank
            gg = torch.Generator().manual_seed(3)
            im1, im2 = torch.rand(n_pairs, 6, 5, 3, generator=gg), torch.rand(n_pairs, 4, 7, 3, generator=gg)
            r = _StubModel()
            e1, e2 = r.forward_dummy(im1, im2)
            r.hip_flush()
            b1, b2 = out['sharded', n_pairs]
            assert torch.equal(torch.tensor(b1), e1) and torch.equal(torch.tensor(b2), e2), rank
        m1, m2 = out['mixed']
        assert torch.equal(torch.tensor(m1), x1) and torch.equal(torch.tensor(m2), x2), rank


def test_gather_is_identity_without_process_group():
    b1, b2 = torch.rand(3, 4), torch.rand(3, 4)
    g1, g2 = gather_boxes(b1, b2, 3)
    assert g1 is b1 and g2 is b2


def test_adjacent_box_block_is_a_view():
    """The engine returns box1 / box2 as the halves of one [2,n,4] block: BoxGatherer sends that block as it is
    (no stack kernel on the batch's stream); unrelated tensors are stacked."""
    from imagematching_oetr_amd.parallel import _adjacent
    both = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4)
    v = _adjacent(both[0], both[1])
    assert v.data_ptr() == both.data_ptr() and torch.equal(v, both)
    x, y = torch.rand(3, 4), torch.rand(3, 4)
    s = _adjacent(x, y)
    assert s.data_ptr() not in (x.data_ptr(), y.data_ptr()) and torch.equal(s, torch.stack((x, y)))
    assert torch.equal(_adjacent(both[1], both[0]), torch.stack((both[1], both[0])))     # wrong order: copied
