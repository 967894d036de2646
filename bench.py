#!/usr/bin/env python
"""Benchmark of the OETR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (feature maps -> overlap boxes:
feature-correlation transformer + centre/size heads, reference
``src/model.py:239-252``) over one batch of synthetic 640x640 pairs per GPU,
with the backbone feature maps already resident in HBM.  Workload = BASELINE
configs[1]: batch of 8 pairs, 640x640 (20x20 = 400 tokens per image), fp32.
With N > 1 (torchrun, one rank per GPU) every rank runs its own batch of 8
(weak scaling) and each step ends with the RCCL all-gather of the per-pair
boxes - the only collective on the path.

The K timed steps are run three times:
  1. ``value`` / ``ms_per_step``: consecutive steps alternate over ``--streams``
     HIP streams (default 3) with 64-token encoder workgroups, so the next
     batch's kernels fill the CUs a batch of 8 pairs leaves idle.  Every step
     still pushes its whole batch through the whole path inside the region.
  2. ``serial``: the same steps strictly one after the other on one stream
     (library-default tile shape) - the batch latency.
  3. the traced pass: serial, with HIP events recorded by the library around
     every launch on its launch stream -> ``roofline`` for the dominant kernel
     (k_encoder<B,A>, 7 of the 13 launches of a step) and ``kernels_us``.

Rank 0 prints ONE JSON line.  ``cpu_baseline`` is the oracle (torch CPU
restatement of the reference) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

PAIR_GFLOP_640 = 8.365            # BASELINE.md §4, hot path per 640x640 pair
ENC_FLOP_PER_TOKEN = 16 * 256 * 256 + 4 * 256 * 32   # one B;A launch
F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
F16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense f16/bf16 MFMA
DOMINANT = 'k_encoder<B,A>'


def synthetic_inputs(pairs, size, size2, device):
    """Random-init weights of the architecture + uniform features with the
    spread of the real extraction path (std ~0.29).  No oracle involved."""
    import imagematching_oetr_amd as pkg
    torch.manual_seed(0)
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    weights = {k: v.detach().clone() for k, v in model.hot_path_state().items()}
    hf, hf2 = size // 32, size2 // 32
    g = torch.Generator().manual_seed(1 + int(os.environ.get('RANK', 0)))
    feat1 = (torch.rand(pairs, 256, hf, hf, generator=g) - 0.5).to(device)
    feat2 = (torch.rand(pairs, 256, hf2, hf2, generator=g) - 0.5).to(device)
    pos1 = model.pos_encoding(feat1.cpu()).contiguous().to(device)
    pos2 = model.pos_encoding(feat2.cpu()).contiguous().to(device)
    return model, weights, feat1, feat2, pos1, pos2, hf, hf2


def cpu_baseline(weights, feat1, feat2, size, size2, budget_s=12.0):
    """The oracle timed on the host cores (bounded sample, rank 0 only).
    torch's intra-op pool is tried at a few sizes first (small tensors stop
    scaling long before a 100+-core host is full; more threads only add
    synchronisation cost) and the fastest one is used and reported."""
    from oracle import oetr_oracle as orc
    f1, f2 = feat1.cpu(), feat2.cpu()
    w = {k: v.cpu() for k, v in weights.items()}
    ncpu = os.cpu_count() or 1
    run = lambda: orc.hot_path(f1, f2, w, (size, size), (size2, size2))
    best_t, best_dt = 1, float('inf')
    for threads in sorted({t for t in (4, 8, 16, 32, 64) if t <= ncpu} | {min(ncpu, 8)}):
        torch.set_num_threads(threads)
        run()
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        if dt < best_dt:
            best_t, best_dt = threads, dt
    torch.set_num_threads(best_t)
    boxes = run()
    iters, t0 = 0, time.perf_counter()
    while True:
        run()
        iters += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or iters >= 400:
            break
    return dict(value=round(f1.shape[0] * iters / dt, 2), unit='image-pairs/s',
                cores=best_t, kind='port',
                sample=f'{iters} batches of {f1.shape[0]} pairs @ {size}x{size}'
                       + (f' vs {size2}x{size2} ' if size2 != size else ' ')
                       +
                       f'(hot path only, features precomputed) in {dt:.1f} s; '
                       f'oracle/oetr_oracle.py on torch CPU, {best_t} intra-op '
                       f'threads (best of 4..64 on a {ncpu}-CPU host)'), boxes


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the latest committed
    rocprofv3 PMC passes (profiles/*_pmc_fetch.csv / *_pmc_write.csv, made by
    tools/profile_round.sh with this same bench command; counters cannot be
    read from inside the process).  Per MI355X_MICROARCH.md §HBM: FETCH_SIZE
    and WRITE_SIZE are in KiB and FETCH_SIZE under-reports wide coalesced
    reads by 2x on gfx950, so bytes = (2*FETCH + WRITE) * 1024."""
    import csv
    import glob
    out = {}
    for kind in ('fetch', 'write'):
        files = sorted(glob.glob(str(REPO / 'profiles' / f'*_pmc_{kind}.csv')))
        if not files:
            return None
        with open(files[-1]) as f:
            for row in csv.DictReader(f):
                if kernel_substr in row['kernel']:
                    out[kind] = float(row['FETCH_SIZE' if kind == 'fetch' else 'WRITE_SIZE'])
    if len(out) != 2:
        return None
    return int((2 * out['fetch'] + out['write']) * 1024)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--pairs-per-gpu', type=int, default=8)
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--size2', type=int, default=None,
                    help='side of image2 (default: same as --size); BASELINE configs[4] = 1280')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true')
    ap.add_argument('--precision', default='f32_split_f16', choices=['f32_split_f16', 'f32'],
                    help='GEMM arithmetic: 3 f16 MFMAs per fp32 product (default) or exact f32 MFMA')
    ap.add_argument('--no-trace', action='store_true',
                    help='do not record per-kernel events in the timed region')
    ap.add_argument('--streams', type=int, default=3,
                    help='HIP streams the consecutive steps (batches) alternate over: a step of 8 '
                         'pairs fills 208 of 256 CUs, the next batch on a second stream fills the '
                         'rest.  1 = strictly serial steps (also always reported)')
    ap.add_argument('--enc-tile', type=int, default=0, choices=[0, 32, 64],
                    help='token rows per encoder workgroup (0: 64 while batches overlap on several '
                         'streams - fewest CU-microseconds per token - and the library default, '
                         '32 at this size, for the serial / traced passes)')
    args = ap.parse_args()

    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus and rank == 0:
        print(f'[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with '
              f'torchrun --nproc-per-node {args.gpus}', file=sys.stderr)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    # (dry runs of the N>1 logic on a 1-GPU box: OETR_BENCH_BACKEND=gloo maps every
    #  rank onto the GPUs that exist; the real launch is one rank per GPU over RCCL)
    backend = os.environ.get('OETR_BENCH_BACKEND', 'nccl')
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    import imagematching_oetr_amd as pkg
    from imagematching_oetr_amd.parallel import BoxGatherer
    torch.set_grad_enabled(False)
    n = args.pairs_per_gpu
    size2 = args.size2 or args.size
    model, weights, feat1, feat2, pos, pos2, hf, hf2 = synthetic_inputs(n, args.size, size2, device)
    eng = pkg.HotPathEngine(weights, device=device, precision=args.precision)
    hw, hw2 = (args.size, args.size), (size2, size2)
    n_total = n * world

    gatherer = BoxGatherer() if world > 1 else None
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]

    def step(i=0, ns=1):
        # consecutive steps alternate over the streams (one workspace per stream in
        # the engine); every step is a full batch of n pairs through the whole path
        with torch.cuda.stream(streams[i % ns]):
            b1, b2 = eng.forward(feat1, feat2, pos, pos2, hw, hw2)
            if gatherer is not None:
                # the all-gather of this batch's boxes runs on RCCL's stream under the
                # next batch's kernels; it is completed at the next submit / the flush
                gatherer.submit(b1, b2)
        return b1, b2

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(ns):
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, ns)
        if gatherer is not None:
            gatherer.flush()            # last batch's gather is inside the timed region
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    split = args.precision == 'f32_split_f16'
    tile_overlap = args.enc_tile or (64 if (n_streams > 1 and split) else 0)
    eng.set_encoder_tile(args.enc_tile)
    for i in range(args.warmup):
        step(i, 1)
    eng.set_encoder_tile(tile_overlap)
    for i in range(args.warmup):
        step(i, n_streams)
    elapsed = timed_region(n_streams)     # -> value (no instrumentation)
    eng.set_encoder_tile(args.enc_tile)
    elapsed_serial = timed_region(1) if n_streams > 1 else elapsed
    kern, elapsed_traced = {}, None
    if not args.no_trace:
        # Same K steps again, on ONE stream (kernel durations are only meaningful
        # when launches do not share the chip), with the library's per-kernel HIP
        # events recorded on its launch stream.  The events themselves cost ~9% of
        # a step, so this pass feeds `roofline` only; its wall time is reported too.
        with pkg.KernelTrace(eng, max_launches=16 * args.steps + 64) as trace:
            elapsed_traced = timed_region(1)
        kern = trace.summary()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed
    out = {
        'metric': f'image-pairs/sec @{args.size}x{args.size} (OETR hot path: '
                  'feature correlation + overlap regression, features resident in HBM)',
        'value': round(value, 1), 'unit': 'image-pairs/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_step, 4), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'gemm_mode': ('fp32-class products from 3 f16 MFMAs (a=ah+al/2^11 split), fp32 accumulate'
                      if args.precision == 'f32_split_f16' else 'exact fp32 MFMA'),
        'data': 'synthetic',
        'config': {'workload': (f'BASELINE configs[1]: ' if (n, args.size, size2) == (8, 640, 640) else '')
                               + f'batch={n} pairs/GPU, {args.size}x{args.size}'
                               + (f' vs {size2}x{size2}' if size2 != args.size else '')
                               + f' -> {hf}x{hf}' + (f' / {hf2}x{hf2}' if hf2 != hf else '')
                               + ' tokens/image, C=256, 8 enc + 2 dec layers, fp32',
                   'pairs_per_gpu': n, 'global_pairs': n_total,
                   'streams': n_streams,
                   'encoder_tile_rows': tile_overlap or 'auto',
                   'tokens_per_image': hf * hf,
                   'parallelism': f'pairs sharded over {world} rank(s); '
                                  'all-gather of boxes only'},
        'hot_path_tflops': round(value * PAIR_GFLOP_640 * (hf * hf + hf2 * hf2) / 800 / 1e3, 2),
        'hot_path_frac_of_mfma_peak': round(
            value * PAIR_GFLOP_640 * (hf * hf + hf2 * hf2) / 800 / 1e3
            / (F16_MFMA_PEAK_TFLOPS / 3 if args.precision == 'f32_split_f16' else F32_MFMA_PEAK_TFLOPS), 4),
        # the same K steps strictly one after the other on one stream (batch latency;
        # encoder tile = library default, which is also what the traced pass below runs)
        'serial': {'ms_per_step': round(elapsed_serial / args.steps * 1e3, 4),
                   'pairs_per_s': round(n_total * args.steps / elapsed_serial, 1)},
    }
    if kern and DOMINANT in kern:
        launches, total_ms = kern[DOMINANT]
        avg_ms = total_ms / launches
        flop = ENC_FLOP_PER_TOKEN * n * (hf * hf + hf2 * hf2)   # tokens of both sides (algorithmic)
        # `achieved` is ALGORITHMIC fp32 FLOP/s.  In split mode every algorithmic
        # product costs 3 f16 MFMA products, so the MFMA roof for this scheme is the
        # dense f16 peak / 3; in exact mode it is the f32 MFMA peak.
        peak = F16_MFMA_PEAK_TFLOPS / 3 if split else F32_MFMA_PEAK_TFLOPS
        ach = flop / (avg_ms * 1e-3) / 1e12
        out['roofline'] = {
            'kernel': DOMINANT, 'bound': 'mfma', 'achieved': round(ach, 2),
            'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
            'peak_basis': ('dense f16 MFMA 2500 TFLOP/s / 3 products per fp32 product'
                           if split else 'f32 MFMA (v_mfma_f32_32x32x2_f32) 157.3 TFLOP/s'),
            'executed_mfma_tflops': round(ach * (3 if split else 1), 2),
            'frac_of_f32_mfma_peak': round(ach / F32_MFMA_PEAK_TFLOPS, 4),
            'traffic': pmc_traffic_bytes('k_encoderILb1ELi0E') if (n, args.size, size2) == (8, 640, 640) else None,
            'avg_launch_us': round(avg_ms * 1e3, 2), 'launches': launches,
            'flop_per_launch': flop,
            'share_of_step': round(total_ms / (elapsed_traced * 1e3), 4),
            'traced_ms_per_step': round(elapsed_traced / args.steps * 1e3, 4)}
        out['kernels_us'] = {k: [v[0] // args.steps, round(v[1] / v[0] * 1e3, 2)]
                             for k, v in kern.items()}
    if not args.no_cpu_baseline and world == 1:   # host-core baseline: rank 0 at N=1 only
        base, ref_boxes = cpu_baseline(weights, feat1, feat2, args.size, size2)
        out['cpu_baseline'] = base
        from oracle import oetr_oracle as orc
        mine = eng.forward(feat1, feat2, pos, pos2, hw, hw2)
        iou = torch.cat([orc.bbox_iou_aligned(mine[0].cpu(), ref_boxes[0]),
                         orc.bbox_iou_aligned(mine[1].cpu(), ref_boxes[1])])
        out['iou_vs_cpu_min'] = round(float(iou.min()), 6)
        out['speedup_vs_cpu'] = round(value / base['value'], 1)
    if not args.no_e2e and world == 1:
        try:     # whole forward_dummy incl. the PyTorch/MIOpen backbone (host code)
            model = model.to(device)
            g = torch.Generator().manual_seed(2)
            im1 = torch.rand(n, args.size, args.size, 3, generator=g).to(device)
            im2 = torch.rand(n, size2, size2, 3, generator=g).to(device)
            for _ in range(3):
                model.forward_dummy(im1, im2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                model.forward_dummy(im1, im2)
            torch.cuda.synchronize()
            out['end_to_end_pairs_per_s'] = round(n * reps / (time.perf_counter() - t1), 1)
            # front end of that forward: neck (HIP) per batch of 2N backbone maps
            bbf = model.backbone(torch.cat([im1, im2])) if args.size == size2 else model.backbone(im1)
            with pkg.KernelTrace(model.neck_engine()) as ntr:
                for _ in range(reps):
                    model.neck(bbf)
                torch.cuda.synchronize()
            out['neck_kernels_us'] = {k: round(v[1] / v[0] * 1e3, 1) for k, v in ntr.summary().items()}
            out['neck_images'] = int(bbf.shape[0])
        except Exception as e:  # host-side extras must never break the bench line
            out['end_to_end_error'] = repr(e)[:200]
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
