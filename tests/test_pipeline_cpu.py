"""Host logic of the batched pair front-end (SURVEY.md §8 f3,
``imagematching_oetr_amd/pipeline.py``) on CPU with a stub model: bucketing by shape,
batch size cap, input-order results equal to the reference's per-pair loop
(``evaluation.py:77-80``: one ``overlap({'image0','image1'})`` call per pair), and the
world-size-2 gloo variant that shards every bucket over the ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imagematching_oetr_amd.pipeline import forward_pairs, plan_batches

SIZES = [((64, 64), (64, 64)), ((64, 128), (64, 64)), ((48, 64), (48, 64))]


class StubModel:
    """forward_dummy with the reference contract; boxes depend on the pixels of the
    pair only, so any mixing of pairs or wrong order shows up."""

    def __init__(self):
        self.calls = []
        self._pending = None

    def hip_flush(self):
        """Deferred settle of the last batch, IN PLACE (what OETR does under hip_defer_check
        when a batch tripped the f16 range flag): until then its boxes are garbage."""
        if self._pending is not None:
            for t, good in self._pending:
                t.copy_(good)
            self._pending = None

    def forward_dummy(self, image1, image2):
        assert image1.shape[0] == image2.shape[0] and image1.shape[-1] == 3
        self.hip_flush()
        self.calls.append((tuple(image1.shape), tuple(image2.shape)))
        m1 = image1.reshape(image1.shape[0], -1).mean(1, keepdim=True)
        m2 = image2.reshape(image2.shape[0], -1).mean(1, keepdim=True)
        k = torch.arange(4, dtype=torch.float32)
        good = (m1 * image1.shape[2] + k, m2 * image2.shape[1] - k)
        out = tuple(torch.full_like(t, 7.0e4) for t in good)      # "overflowed" until settled
        self._pending = list(zip(out, good))
        return out


def make_pairs(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    pairs = []
    for i in range(n):
        (h0, w0), (h1, w1) = SIZES[(i * 7 + i // 3) % len(SIZES)]
        a = torch.rand(1, h0, w0, 3, generator=g)
        b = torch.rand(h1, w1, 3, generator=g)          # [H,W,3] form is accepted too
        pairs.append((a, b))
    return pairs


def per_pair_loop(pairs):
    m = StubModel()
    out0, out1 = [], []
    for a, b in pairs:
        a = a if a.dim() == 4 else a[None]
        b = b if b.dim() == 4 else b[None]
        b0, b1 = m.forward_dummy(a, b)
        m.hip_flush()
        out0.append(b0[0]); out1.append(b1[0])
    return torch.stack(out0), torch.stack(out1)


def test_plan_covers_every_pair_once_in_shape_buckets():
    shapes = [SIZES[i % 3] for i in range(20)] + [SIZES[0]] * 5
    for world in (1, 2, 3):
        seen = []
        for rank in range(world):
            for key, idx in plan_batches(shapes, 4, rank, world):
                assert 1 <= len(idx) <= 4
                assert all(shapes[i] == key for i in idx)
                seen += idx
        assert sorted(seen) == list(range(len(shapes)))
    with pytest.raises(ValueError):
        plan_batches(shapes, 0)


@pytest.mark.parametrize('max_batch', [1, 3, 8, 64])
def test_forward_pairs_equals_the_per_pair_loop(max_batch):
    pairs = make_pairs(23)
    model = StubModel()
    b0, b1 = forward_pairs(model, pairs, max_batch=max_batch)
    e0, e1 = per_pair_loop(pairs)
    assert torch.equal(b0, e0) and torch.equal(b1, e1)
    assert all(c[0][0] <= max_batch for c in model.calls)
    if max_batch >= 8:          # buckets really are batched
        assert max(c[0][0] for c in model.calls) >= 7
    assert len(model.calls) >= 3   # one call per shape bucket at least


def test_forward_pairs_edge_cases():
    z0, z1 = forward_pairs(StubModel(), [])
    assert z0.shape == (0, 4) and z1.shape == (0, 4)
    with pytest.raises(ValueError):
        forward_pairs(StubModel(), [(torch.rand(2, 8, 8, 3), torch.rand(1, 8, 8, 3))])
    with pytest.raises(ValueError):
        forward_pairs(StubModel(), [(torch.rand(8, 8), torch.rand(1, 8, 8, 3))])


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from imagematching_oetr_amd.pipeline import forward_pairs_sharded
        model = StubModel()
        b0, b1 = forward_pairs_sharded(model, make_pairs(n), max_batch=4)
        q.put((rank, b0.tolist(), b1.tolist(), sum(c[0][0] for c in model.calls)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n', [1, 11])
def test_forward_pairs_sharded_world2_gloo(n):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    e0, e1 = per_pair_loop(make_pairs(n))
    done = 0
    for rank, b0, b1, pairs_run in results:
        assert torch.equal(torch.tensor(b0), e0) and torch.equal(torch.tensor(b1), e1), rank
        done += pairs_run
    assert done == n            # every pair was computed exactly once across the ranks


def test_throughput_mode_in_flight_cap():
    """``hip_queue_depth`` batches queued per side stream: never fewer than one per stream, never more than seven in
    flight (two status words per batch at most, sixteen in an engine's ring) - host logic, no device."""
    import imagematching_oetr_amd as pkg
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    assert model.hip_queue_depth == 2
    want = {(1, 1): 1, (2, 1): 2, (3, 1): 3, (3, 2): 6, (3, 3): 7, (4, 2): 7, (7, 2): 7, (8, 1): 8, (8, 2): 8, (2, 0): 2}
    for (k, depth), cap in want.items():
        model.hip_queue_depth = depth
        assert model._inflight_cap(k) == cap, (k, depth)
        assert 2 * model._inflight_cap(k) <= 16
    model.hip_streams = 9
    with pytest.raises(ValueError):
        model._stream_count()
