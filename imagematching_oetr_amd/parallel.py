"""Multi-GPU use of the hot path: one process per GPU, image pairs sharded
contiguously over ranks, weights replicated, and ONE collective per batch -
the all-gather of the per-pair boxes (32 B per pair) over RCCL/xGMI.

The reference has no inference-time multi-GPU code (its only distributed code
is training DDP, ``train.py:59-74``); pairs are independent (no cross-pair
term anywhere in reference ``src/model.py:229-252``), so there is no exchange
inside the forward.  The gather is latency-bound (N=64 pairs
-> 2 KiB in total): one padded ``all_gather_into_tensor`` call, no bucketing.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_pairs, rank, world_size):
    """Contiguous [lo, hi) slice of ``n_pairs`` owned by ``rank``; the first
    ``n_pairs % world_size`` ranks hold one extra pair."""
    if not 0 <= rank < world_size:
        raise ValueError(f'rank {rank} outside world of {world_size}')
    base, extra = divmod(n_pairs, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def bucket_by_shape(shapes):
    """Group pair indices by (image1 shape, image2 shape) so that every rank
    runs equal work on mixed-scale batches (BASELINE configs[4]).  Returns
    {shape_key: [indices]} in first-seen order."""
    buckets = {}
    for i, key in enumerate(shapes):
        buckets.setdefault(tuple(key), []).append(i)
    return buckets


def gather_boxes(box1, box2, n_pairs, group=None):
    """All-gather this rank's ``[n_local,4]`` boxes into the full
    ``([n_pairs,4], [n_pairs,4])`` in shard order: ONE ``all_gather_into_tensor``
    of the (padded) ``[cap,2,4]`` shard.  The same code runs on every backend - RCCL
    on GPUs, gloo on CPU tensors in the tests - so the CPU tests execute exactly the
    collective the GPUs will."""
    if not (dist.is_available() and dist.is_initialized()):
        return box1, box2
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_pairs, rank, world)
    if box1.shape[0] != hi - lo or box2.shape[0] != hi - lo:
        raise ValueError(f'rank {rank} holds {box1.shape[0]} pairs, expected '
                         f'{hi - lo} of {n_pairs}')
    cap = -(-n_pairs // world)                       # ceil: (padded) shard size
    if n_pairs % world == 0:
        mine = torch.stack((box1, box2), dim=1)      # [cap, 2, 4], no padding needed
    else:
        mine = torch.zeros(cap, 2, 4, dtype=box1.dtype, device=box1.device)
        mine[:hi - lo, 0] = box1
        mine[:hi - lo, 1] = box2
    everyone = torch.empty(world * cap, 2, 4, dtype=box1.dtype, device=box1.device)
    dist.all_gather_into_tensor(everyone, mine.contiguous(), group=group)
    if n_pairs % world == 0:
        return everyone[:, 0].contiguous(), everyone[:, 1].contiguous()
    keep = torch.cat([
        torch.arange(r * cap, r * cap + (shard_bounds(n_pairs, r, world)[1]
                                         - shard_bounds(n_pairs, r, world)[0]))
        for r in range(world)]).to(everyone.device)
    full = everyone.index_select(0, keep)
    return full[:, 0].contiguous(), full[:, 1].contiguous()


def _adjacent(box1, box2):
    """``[2, n, 4]`` of the two box tensors: a VIEW when they are the two halves of one block (what the engine
    returns), else a stacked copy."""
    n = box1.shape[0]
    if (box1.is_contiguous() and box2.is_contiguous() and box1.shape == box2.shape and box1.dtype == box2.dtype
            and box1.device == box2.device and n > 0
            and box1.untyped_storage().data_ptr() == box2.untyped_storage().data_ptr()
            and box2.storage_offset() == box1.storage_offset() + box1.numel()):
        return torch.as_strided(box1, (2, n, 4), (n * 4, 4, 1))
    return torch.stack((box1, box2))


class BoxGatherer:
    """Pipelined box all-gather for a stream of batches: the collective of
    batch k is issued asynchronously (RCCL's own stream) and completed when
    batch k+1 is submitted, so the latency-bound gather runs under the next
    batch's compute.  ``submit`` returns the gathered boxes of
    the OLDEST batch whose collective was issued by an earlier call (None while there is
    none); ``flush`` issues and completes everything outstanding and returns the LAST
    batch's boxes (``flush_all``: every outstanding batch's, in submission order).

    ``submit(box1, box2)``: every rank holds the same number of pairs (the bench's weak
    scaling).  ``submit(box1, box2, n_pairs=N)``: the ranks hold the contiguous
    ``shard_bounds`` slices of N pairs (sizes differ by at most one); shards are padded to
    the largest and the padding dropped after the gather, like ``gather_boxes``.

    The gather copies VALUES, and ``OETR.forward_dummy`` with the deferred range check
    (``hip_defer_check``, the default) corrects a tripped batch IN PLACE later.  Three ways to
    keep stale boxes off the other ranks:

    * ``model=`` (the throughput recipe): the collective of a batch is ISSUED only once the
      model has settled that batch (``OETR.hip_settled``) - i.e. at the ``submit`` that follows
      the settling, k x ``hip_queue_depth`` batches later under ``hip_streams = k``, or at ``flush``.  Every rank
      submits and settles in the same order, so the collectives line up whatever tripped
      where; the in-place correction is ordered before the collective (the model orders the
      batch's side stream behind the caller's stream before it enqueues there again).
    * ``settle=model.hip_flush``: settle before every ``submit`` (serialises the streams).
    * neither: only valid with ``hip_on_overflow = 'ignore'`` or precisions without a range
      guard, or when the caller has flushed the model itself."""

    def __init__(self, group=None, settle=None, on_stream=None, model=None):
        """``on_stream``: False = the collective is asynchronous (the process group's own stream) and
        completed at the next ``submit`` - right for ONE stream of batches (latency mode), where it runs
        under the next batch's kernels.  True = a blocking collective, ordered behind the batch on the
        stream ``submit`` is called from - right for the throughput mode (``model.hip_streams = k``,
        submit under ``torch.cuda.stream(model.hip_batch_stream())``): that side stream's next batch is k
        batches away, the other side streams carry on, and no fifth stream joins the four that the HIP
        runtime's default of four hardware queues carries without sharing (measured, MI355X, world 1:
        32.5 k pairs/s without a gather, 32.3 k with the on-stream one, 14.9 k with the asynchronous one -
        two streams on one queue serialise; `profiles/r5_pg_streams.txt`).  None (default) = True when
        ``submit`` is called from a stream other than the device's default stream.

        The gathered tensors are ordered on the stream ``submit`` / ``flush`` ran on when it handed
        them out (that stream waited for the collective, and the buffer is recorded on it): consume
        them there, or order your stream behind that one."""
        self.settle = settle
        self.group = group
        self.on_stream = on_stream
        self.model = model
        self._waiting = []        # submitted, collective not issued yet (model has not settled them)
        self._issued = []         # (work, everyone, n_pairs, keep), oldest first

    def _finish(self, entry):
        work, everyone, n_pairs, keep = entry
        if isinstance(work, torch.cuda.Event):                 # on-stream collective: order the consumer behind it
            cur = torch.cuda.current_stream(everyone.device)
            cur.wait_event(work)
            everyone.record_stream(cur)                        # allocated on the submitting (side) stream, read here
        elif work is not None:
            work.wait()
            if everyone.is_cuda:
                everyone.record_stream(torch.cuda.current_stream(everyone.device))
        if keep is not None:                                   # unequal shards: drop the padding
            everyone = everyone.index_select(0, keep.to(everyone.device))
            return everyone[:, 0].contiguous(), everyone[:, 1].contiguous()
        return (everyone[:, 0].reshape(n_pairs, 4), everyone[:, 1].reshape(n_pairs, 4))

    def _collective(self, out, mine):
        on_stream = self.on_stream
        if on_stream is None:
            on_stream = out.is_cuda and torch.cuda.current_stream(out.device) != torch.cuda.default_stream(out.device)
        if not on_stream:
            return dist.all_gather_into_tensor(out, mine, group=self.group, async_op=True)
        dist.all_gather_into_tensor(out, mine, group=self.group, async_op=False)
        if not out.is_cuda:
            return None
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(out.device))
        return done

    def _issue(self, box1, box2, n_pairs):
        world = dist.get_world_size(self.group)
        if n_pairs is None or n_pairs % world == 0:
            mine = _adjacent(box1, box2)                              # [2, n_local, 4]
            flat = torch.empty((world * 2,) + tuple(mine.shape[1:]), dtype=mine.dtype,
                               device=mine.device)             # concatenation along dim 0
            work = self._collective(flat, mine)
            self._issued.append((work, flat.view((world, 2) + tuple(mine.shape[1:])),
                                 world * box1.shape[0], None))
            return
        rank = dist.get_rank(self.group)
        lo, hi = shard_bounds(n_pairs, rank, world)
        cap = -(-n_pairs // world)
        mine = torch.zeros(cap, 2, 4, dtype=box1.dtype, device=box1.device)
        mine[:hi - lo, 0] = box1
        mine[:hi - lo, 1] = box2
        everyone = torch.empty(world * cap, 2, 4, dtype=box1.dtype, device=box1.device)
        work = self._collective(everyone, mine)
        sizes = [shard_bounds(n_pairs, r, world) for r in range(world)]
        keep = torch.cat([torch.arange(r * cap, r * cap + (b - a)) for r, (a, b) in enumerate(sizes)])
        self._issued.append((work, everyone, n_pairs, keep))

    def _issue_ready(self, everything=False):
        """Issue the collectives of the waiting batches the model has settled, oldest first."""
        while self._waiting:
            box1, box2, n_pairs = self._waiting[0]
            if not everything and self.model is not None and not self.model.hip_settled(box1):
                break
            self._waiting.pop(0)
            self._issue(box1, box2, n_pairs)

    def submit(self, box1, box2, n_pairs=None):
        if self.settle is not None:
            self.settle()
        world = dist.get_world_size(self.group)
        if n_pairs is None or n_pairs % world == 0:
            if n_pairs is not None and box1.shape[0] * world != n_pairs:
                raise ValueError(f'this rank holds {box1.shape[0]} pairs, expected {n_pairs // world}')
        else:
            rank = dist.get_rank(self.group)
            lo, hi = shard_bounds(n_pairs, rank, world)
            if box1.shape[0] != hi - lo or box2.shape[0] != hi - lo:
                raise ValueError(f'rank {rank} holds {box1.shape[0]} pairs, expected {hi - lo} of {n_pairs}')
        done = self._finish(self._issued.pop(0)) if self._issued else None   # issued by an earlier call
        self._waiting.append((box1, box2, n_pairs))
        self._issue_ready()
        return done

    def flush_all(self):
        """Settle (``model.hip_flush`` when a model was given), issue and complete everything
        outstanding: the gathered ``(box1, box2)`` of every batch not handed out yet, oldest first."""
        if self.model is not None:
            self.model.hip_flush()
        self._issue_ready(everything=True)
        out = [self._finish(e) for e in self._issued]
        self._issued = []
        return out

    def flush(self):
        out = self.flush_all()
        return out[-1] if out else None


@torch.no_grad()
def forward_sharded(model, image1, image2, group=None, mask1=None, mask2=None):
    """``model.forward_dummy`` on this rank's contiguous shard of the batch,
    then the box all-gather: every rank returns boxes for ALL pairs.
    ``mask1`` / ``mask2``: forward_dummy's optional masks [N,hf,wf], sharded with the images."""
    n = image1.shape[0]
    if dist.is_available() and dist.is_initialized():
        lo, hi = shard_bounds(n, dist.get_rank(group), dist.get_world_size(group))
    else:
        lo, hi = 0, n
    if mask1 is not None or mask2 is not None:
        if mask1 is None or mask2 is None:
            raise ValueError('masks: pass both mask1 and mask2, or neither')
        b1, b2 = model.forward_dummy(image1[lo:hi], image2[lo:hi], mask1[lo:hi], mask2[lo:hi])
    else:
        b1, b2 = model.forward_dummy(image1[lo:hi], image2[lo:hi])
    # forward_dummy defers its f16 range check and corrects a tripped batch IN PLACE later
    # (OETR.hip_defer_check): settle it before the boxes are copied to the other ranks
    flush = getattr(model, 'hip_flush', None)
    if flush is not None:
        flush()
    return gather_boxes(b1, b2, n, group)
