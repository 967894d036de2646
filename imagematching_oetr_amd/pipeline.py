"""Batched pair front-end.

The reference feeds the overlap model ONE pair per call: ``evaluation.py:303``
and ``dloc/core/overlap_features.py:273`` loop over the pair list, each
iteration reads two images (``read_overlap_image``,
``dloc/core/utils/utils.py:271-343``: an ``[1,H,W,3]`` float tensor in [0,1]) and
calls ``self.overlap({'image0': ..., 'image1': ...})`` (``evaluation.py:77-80``,
``overlap_features.py:158-178``), using ``bbox[0]``.  Pairs are independent, and
the MI355X hot path only reaches its throughput with N >= 8 pairs per call, so
this module turns a pair *stream* into shape-bucketed batches:

* :func:`forward_pairs` - boxes for a list of pairs of arbitrary (mixed) sizes,
  in input order, computed bucket by bucket with up to ``max_batch`` pairs per
  ``forward_dummy`` call.  Per-pair contract unchanged: entry ``i`` of the result
  is what ``model.forward_dummy(image0_i, image1_i)`` returns for that pair alone
  (pairs never interact: no cross-pair term anywhere in reference
  ``src/model.py:229-252``).
* :func:`forward_pairs_sharded` - the same over a process group: every bucket is
  split contiguously over the ranks (equal shapes -> equal work), each rank runs
  its share, and ONE padded ``all_gather_into_tensor`` of the ``[n,2,4]`` boxes
  (the only collective on the path) gives every rank all boxes.

Host logic only: arithmetic stays in ``OETR.forward_dummy`` (HIP hot path).
"""
import torch
import torch.distributed as dist

from .parallel import bucket_by_shape, shard_bounds


def _as_batch1(img):
    """[H,W,3] or [1,H,W,3] -> [1,H,W,3] (the reader's layout, utils.py:329-330)."""
    if img.dim() == 3:
        img = img.unsqueeze(0)
    if img.dim() != 4 or img.shape[0] != 1 or img.shape[-1] != 3:
        raise ValueError(f'pair images must be [H,W,3] or [1,H,W,3] (NHWC), got {tuple(img.shape)}')
    return img


def plan_batches(shapes, max_batch, rank=0, world=1):
    """Deterministic work plan shared by every rank: ``[(bucket_key, [pair indices])]``.

    Pairs are bucketed by ``(shape0, shape1)`` in first-seen order; with
    ``world > 1`` each bucket is split contiguously over the ranks
    (``shard_bounds``) and only this rank's slice is kept; slices are then cut
    into chunks of at most ``max_batch`` pairs."""
    if max_batch < 1:
        raise ValueError('max_batch must be >= 1')
    plan = []
    for key, idx in bucket_by_shape(shapes).items():
        lo, hi = shard_bounds(len(idx), rank, world)
        mine = idx[lo:hi]
        for s in range(0, len(mine), max_batch):
            plan.append((key, mine[s:s + max_batch]))
    return plan


@torch.no_grad()
def _run_plan(model, pairs, plan, device):
    out = {}
    for _, idx in plan:
        im0 = torch.cat([_as_batch1(pairs[i][0]) for i in idx]).to(device, non_blocking=True)
        im1 = torch.cat([_as_batch1(pairs[i][1]) for i in idx]).to(device, non_blocking=True)
        b0, b1 = model.forward_dummy(im0, im1)
        for j, i in enumerate(idx):
            out[i] = (b0[j], b1[j])
    flush = getattr(model, 'hip_flush', None)
    if flush is not None:
        flush()            # deferred range check of the last batch (OETR.hip_defer_check)
    return out


def _model_device(model):
    try:
        return next(model.parameters()).device
    except (StopIteration, AttributeError):
        return torch.device('cpu')


@torch.no_grad()
def forward_pairs(model, pairs, max_batch=8):
    """``pairs``: sequence of ``(image0, image1)``, each ``[H,W,3]`` or ``[1,H,W,3]``
    float in [0,1] (any device; moved to the model's).  Returns ``(box0, box1)``,
    each ``[len(pairs), 4]`` xyxy pixels of the respective image, in input order."""
    if len(pairs) == 0:
        z = torch.zeros(0, 4, device=_model_device(model))
        return z, z.clone()
    shapes = [(tuple(_as_batch1(a).shape[1:3]), tuple(_as_batch1(b).shape[1:3])) for a, b in pairs]
    res = _run_plan(model, pairs, plan_batches(shapes, max_batch), _model_device(model))
    box0 = torch.stack([res[i][0] for i in range(len(pairs))])
    box1 = torch.stack([res[i][1] for i in range(len(pairs))])
    return box0, box1


@torch.no_grad()
def forward_pairs_raw(model, raw_pairs, resize=(640,), grayscale=True, align='disk', max_batch=8):
    """The reference's per-pair loop from DECODED images to boxes, batched and on the device
    (``dloc/core/overlap_features.py:158-178``: two ``read_overlap_image`` calls, then
    ``self.overlap({'image0': overlap_inp0, 'image1': overlap_inp1})``).

    ``raw_pairs``: sequence of ``(image0, image1)``, each a decoded BGR picture ``[H,W,3]``
    (uint8 or float32, numpy / torch, ANY sizes: the reader maps every picture into the same
    ``resize[0] x resize[0]`` OETR frame, so one bucket holds them all).  Per chunk of
    ``max_batch`` pairs: one pinned staging buffer and ONE host-to-device copy for all 2n
    pictures, the device reader (``reader.read_overlap_images``) writing straight into the
    ``[n,S,S,3]`` batches, one ``forward_dummy``.  Returns a dict of per-pair results in
    input order: ``box0`` / ``box1`` ``[len,4]`` device tensors (OETR frame), ``scales0/1``,
    ``overlap_scales0/1`` (lists of float pairs), ``inp0`` / ``inp1`` (lists of
    ``[1,1|3,h,w]`` device tensors: the matcher's images) - exactly what
    ``overlap_crop(inp0[i], inp1[i], box0[i], box1[i], overlap_scales0[i], overlap_scales1[i])``
    takes, with no host round trip in between."""
    from .reader import read_overlap_images
    device = _model_device(model)
    n = len(raw_pairs)
    res = dict(box0=torch.zeros(n, 4, device=device), box1=torch.zeros(n, 4, device=device),
               scales0=[None] * n, scales1=[None] * n, overlap_scales0=[None] * n,
               overlap_scales1=[None] * n, inp0=[None] * n, inp1=[None] * n)
    # Boxes are copied into `res` only AFTER the final flush: forward_dummy defers its range
    # check (OETR.hip_defer_check) and corrects a tripped batch IN PLACE in the tensors it
    # returned - at the next forward_dummy or at hip_flush() - so a by-value copy taken right
    # after the call would keep the out-of-range boxes.
    produced = []      # (result indices, b0, b1, rows of b0 / b1)
    for s in range(0, n, max(1, max_batch)):
        chunk = list(range(s, min(n, s + max_batch)))
        m = len(chunk)
        read = read_overlap_images([raw_pairs[i][0] for i in chunk] + [raw_pairs[i][1] for i in chunk],
                                   device, resize, grayscale, align)
        r0, r1 = read[:m], read[m:]
        if all(r._batch is read[0]._batch for r in read):
            # one OETR frame for every picture (always, unless resize == [-1]): the reader filled ONE
            # [2m,S,S,3] batch, image0s first - the two halves ARE forward_dummy's inputs, no copy
            b0, b1 = model.forward_dummy(read[0]._batch[:m], read[0]._batch[m:])
            produced.append((chunk, b0, b1, list(range(m))))
        else:   # native-size frames: bucket the pairs by shape
            shapes = [(tuple(a.overlap_inp.shape[1:3]), tuple(b.overlap_inp.shape[1:3])) for a, b in zip(r0, r1)]
            for _, idx in bucket_by_shape(shapes).items():
                im0, im1 = torch.cat([r0[k].overlap_inp for k in idx]), torch.cat([r1[k].overlap_inp for k in idx])
                b0, b1 = model.forward_dummy(im0, im1)
                produced.append(([chunk[k] for k in idx], b0, b1, list(range(len(idx)))))
        for k, i in enumerate(chunk):
            res['scales0'][i], res['scales1'][i] = r0[k].scales, r1[k].scales
            res['overlap_scales0'][i], res['overlap_scales1'][i] = r0[k].overlap_scales, r1[k].overlap_scales
            res['inp0'][i], res['inp1'][i] = r0[k].inp, r1[k].inp
    flush = getattr(model, 'hip_flush', None)
    if flush is not None:
        flush()
    for dst, b0, b1, rows in produced:     # settled (and, if need be, corrected) boxes -> result
        di = torch.as_tensor(dst, device=res['box0'].device)
        res['box0'][di], res['box1'][di] = b0[rows], b1[rows]
    return res


@torch.no_grad()
def forward_pairs_sharded(model, pairs, max_batch=8, group=None):
    """:func:`forward_pairs` over a process group (one rank per GPU): every rank holds
    the same pair list, computes its shard of every shape bucket and receives the
    boxes of ALL pairs through one all-gather.  Without an initialised process
    group this is :func:`forward_pairs`."""
    if not (dist.is_available() and dist.is_initialized()):
        return forward_pairs(model, pairs, max_batch)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(pairs)
    shapes = [(tuple(_as_batch1(a).shape[1:3]), tuple(_as_batch1(b).shape[1:3])) for a, b in pairs]
    owned = [[i for _, idx in plan_batches(shapes, max_batch, r, world) for i in idx]
             for r in range(world)]                       # same on every rank
    assert sorted(i for o in owned for i in o) == list(range(n))
    device = _model_device(model)
    res = _run_plan(model, pairs, plan_batches(shapes, max_batch, rank, world), device)
    cap = max(1, max(len(o) for o in owned))              # padded shard size
    mine = torch.zeros(cap, 2, 4, dtype=torch.float32, device=device)
    for slot, i in enumerate(owned[rank]):
        mine[slot, 0], mine[slot, 1] = res[i]
    everyone = torch.empty(world * cap, 2, 4, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    src = torch.empty(n, dtype=torch.long)
    for r in range(world):
        for slot, i in enumerate(owned[r]):
            src[i] = r * cap + slot
    full = everyone.index_select(0, src.to(device))
    return full[:, 0].contiguous(), full[:, 1].contiguous()
