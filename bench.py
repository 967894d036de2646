#!/usr/bin/env python
"""Benchmark of the OETR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--precision P]

A *step* is one pass of the hot path (feature maps -> overlap boxes:
feature-correlation transformer + centre/size heads, reference
``src/model.py:239-252``) over one batch of synthetic 640x640 pairs per GPU,
with the backbone feature maps already resident in HBM.  Default workload =
BASELINE configs[1]: batch of 8 pairs, 640x640 (20x20 = 400 tokens per image),
fp32 (``--precision f32_split_f16``: fp32-class products from f16 MFMAs).
``--precision f32_split_qk16`` (the per-GEMM-site precision policy, the reduced mode
that meets the 1e-3 IoU bar) is the per-GPU share of configs[2] (64 pairs over 8 GPUs),
``--precision f32_split_qk16 --size2 1280`` that of configs[4]; ``bf16`` / ``f16`` are the
all-rounded single-pass modes (they miss the bar: reported for comparison only).

N > 1: one rank per GPU.  Started WITHOUT a launcher (``python bench.py --gpus
4``) the script re-executes itself under ``torch.distributed.run`` with N ranks;
started under torchrun it checks WORLD_SIZE == N.  It exits non-zero rather than
report fewer GPUs than asked for.  Every rank runs its own batch (weak scaling)
and each step ends with the RCCL all-gather of the per-pair boxes - the only
collective on the path; ``n_gpus`` in the output is the process group's size.

Timed regions (each: barrier + synchronize, exactly K steps, barrier +
synchronize, max over ranks) are repeated ``--repeats`` times and the MEDIAN
region is reported (min / max alongside): a 20-step region lasts ~7 ms, too
short to be stable alone.  Regions:
  1. ``value`` / ``ms_per_step``: consecutive steps alternate over ``--streams``
     HIP streams (default 3) with 64-token encoder workgroups, so the next
     batch's kernels fill the CUs a batch of 8 pairs leaves idle.  Every step
     still pushes its whole batch through the whole path inside the region.
  2. ``serial``: the same steps strictly one after the other on one stream
     (library-default tile shape, 32 token rows at this size) - batch latency.
  3. traced passes (serial, HIP events recorded by the library around every
     launch on its launch stream): ``roofline`` describes the dominant kernel
     IN THE SHAPE THAT PRODUCED ``value`` (64-token workgroups); the 32-token
     kernel of the serial mode is under ``serial.roofline``.
  4. ``exact_f32``: the same workload on an OETR_DTYPE_F32 handle (true fp32 MFMA
     products), timed the same way, so the strict-fp32 figure is driver-timed too.

Rank 0 prints ONE JSON line.  ``cpu_baseline`` is the oracle (torch CPU
restatement of the reference) timed on this box's host cores (rank 0, N=1).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

PAIR_GFLOP_640 = 8.365            # BASELINE.md §4, hot path per 640x640 pair
ENC_FLOP_PER_TOKEN = 16 * 256 * 256 + 4 * 256 * 32   # one B;A launch
F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
F16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense f16/bf16 MFMA
DOMINANT = 'k_encoder<B,A>'
MODE_ID = {'f32': 0, 'f32_split_f16': 1, 'f16': 2, 'bf16': 3, 'f32_split_qk16': 1}   # GM_* id in mangled kernel names
POLICY_ID = {'f32_split_qk16': 1}

# What MI355X SUSTAINS on back-to-back dense f16 MFMAs at its socket power cap - the roof a power-bound kernel can
# actually reach - is MEASURED in the run that quotes it (oetr_debug_mfma_rate, ~1 s, N = 1): CALIB['f16_sustained']
# stays None (and the second yardstick is left out) when the calibration did not run.  Round 5 carried a constant here
# (1 642 TFLOP/s, one box, a stand-alone probe: profiles/r5_energy_prices.txt).
CALIB = {'f16_sustained': None, 'event_pair_us': 0.0}
# MFMA products executed per algorithmic product, and the pipe they run on
MFMA_COST = {'f32': (1, F32_MFMA_PEAK_TFLOPS, 'f32 MFMA (v_mfma_f32_32x32x2_f32) 157.3 TFLOP/s'),
             'f32_split_f16': (3, F16_MFMA_PEAK_TFLOPS,
                               'dense f16 MFMA 2500 TFLOP/s / 3 MFMA products per fp32-class product'),
             # policy: of the 8 GEMM units of a B;A launch Q and K run 1 MFMA per product, the other six 3
             'f32_split_qk16': (2.5, F16_MFMA_PEAK_TFLOPS,
                                'dense f16 MFMA 2500 TFLOP/s / 2.5 MFMA products per algorithmic product '
                                '(Q, K: 1; V, merge, MLP: 3)'),
             'f16': (1, F16_MFMA_PEAK_TFLOPS, 'dense f16 MFMA 2500 TFLOP/s'),
             'bf16': (1, F16_MFMA_PEAK_TFLOPS, 'dense bf16 MFMA 2500 TFLOP/s')}
GEMM_MODE_TEXT = {
    'f32': 'exact fp32 MFMA',
    'f32_split_f16': 'fp32-class products from 3 f16 MFMAs (a=ah+al/2^11 split), fp32 accumulate',
    'f32_split_qk16': 'per-GEMM-site precision policy: Q / K / decoder-K projections on single f16 MFMAs, every '
                      'other site fp32-class (3 f16 MFMAs per product); meets the 1e-3 IoU bar',
    'f16': 'GEMM operands rounded to f16, one MFMA per product, fp32 accumulate / LN / softmax / residual',
    'bf16': 'GEMM operands rounded to bf16, one MFMA per product, fp32 accumulate / LN / softmax / residual'}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--repeats', type=int, default=9,
                    help='timed regions of --steps steps each; the median region is reported')
    ap.add_argument('--pairs-per-gpu', type=int, default=8)
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--size2', type=int, default=None,
                    help='side of image2 (default: same as --size); BASELINE configs[4] = 1280')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-e2e', action='store_true')
    ap.add_argument('--no-power', action='store_true', help='skip the rocm-smi power / clock samples (N=1; about 6 s)')
    ap.add_argument('--no-exact-f32', action='store_true',
                    help='skip the exact-fp32 comparison leg (N=1, default precision only)')
    ap.add_argument('--precision', default='f32_split_f16', choices=sorted(MODE_ID),
                    help='GEMM arithmetic: 3 f16 MFMAs per fp32 product (default, fp32-class), exact '
                         'f32 MFMA, or operands rounded to f16 / bf16 (one MFMA per product)')
    ap.add_argument('--no-trace', action='store_true',
                    help='do not record per-kernel events (no roofline block)')
    ap.add_argument('--streams', type=int, default=3,
                    help='HIP streams the consecutive steps (batches) alternate over: a step of 8 '
                         'pairs fills 208 of 256 CUs, the next batch on a second stream fills the '
                         'rest.  1 = strictly serial steps (also always reported)')
    ap.add_argument('--enc-tile', type=int, default=0, choices=[0, 32, 64],
                    help='token rows per encoder workgroup (0: 64 while batches overlap on several '
                         'streams - fewest CU-microseconds per token - and the library default, '
                         '32 at this size, for the serial pass)')
    ap.add_argument('--attention', default='linear', choices=['linear', 'full'],
                    help="encoder attention core: 'linear' (the reference's default model) or 'full' "
                         "(EncoderLayer(attention='full'): all-pairs softmax attention, flash style)")
    ap.add_argument('--kernel', default=None, choices=[None, 'full_attention'],
                    help='micro-benchmark of one stand-alone kernel instead of the hot path')
    ap.add_argument('--L', type=int, default=1024, help='--kernel full_attention: tokens per image')
    ap.add_argument('--decoder-split-overlap', type=int, default=0, choices=[0, 1, 4],
                    help='oetr_set_decoder_split for the overlapped (multi-stream) run (0 = the library rule; A/Bs)')
    ap.add_argument('--tail-mode-overlap', type=int, default=2, choices=[0, 1, 2, 3],
                    help='oetr_set_tail_mode for the overlapped (multi-stream) run, two-plane precisions: 2 = direct form '
                         '(decoder, then the 64-row conv: fewest CU-microseconds - no P buffer, no combine - which is what '
                         'counts when other streams fill the chip; +2.4 %% at 8 pairs @640x640), 0 = the library rule '
                         '(latency: P form below 16 000 token rows)')
    ap.add_argument('--prereduce-overlap', type=int, default=-1, choices=[-1, 0, 1],
                    help='oetr_set_state_prereduce for the overlapped run (-1 = the library rule; A/Bs)')
    ap.add_argument('--decoder-split', type=int, default=0, choices=[0, 1, 4],
                    help='oetr_set_decoder_split for the serial run (0 = the library rule; A/Bs)')
    ap.add_argument('--no-rccl-world1', action='store_true',
                    help='N = 1: skip the regions with an RCCL process group of one rank (roofline.other_configs.c1_rccl_world1)')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the short driver-timed regions of BASELINE configs[2] / [3] / [4] (N=1, default '
                         'workload only; about 20 s)')
    ap.add_argument('--gather-on-stream', default='auto', choices=['auto', '0', '1'],
                    help="N > 1: BoxGatherer(on_stream=...) - 'auto' = blocking collective on the batch's side stream in "
                         'the throughput mode, asynchronous on the process group\'s stream in the serial mode (A/B knob)')
    ap.add_argument('--workload', default='uniform', choices=['uniform', 'mixed'],
                    help="'mixed' = BASELINE configs[4] as a mixed-scale job: every rank holds the same global "
                         'pair list (640x640 vs 640x640 and 640x640 vs 1280x1280, --pairs-per-gpu of each per '
                         'GPU), buckets it by (L1, L2), runs its contiguous shard of every bucket and all-gathers '
                         'the boxes per bucket; reports the per-rank step-time spread')
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start N ranks ourselves."""
    import torch
    backend = os.environ.get('OETR_BENCH_BACKEND', 'nccl')
    have = torch.cuda.device_count()
    if backend == 'nccl' and have < args.gpus:
        print(f'[bench] --gpus {args.gpus} but only {have} GPU(s) visible: refusing to run a '
              f'smaller job under that label', file=sys.stderr)
        return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def synthetic_inputs(pairs, size, size2, device):
    """Random-init weights of the architecture + uniform features with the
    spread of the real extraction path (std ~0.29).  No oracle involved."""
    import torch
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
    import torch
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
                cores=best_t, threads=best_t, host_cpus=ncpu, kind='port',
                sample=f'{iters} batches of {f1.shape[0]} pairs @ {size}x{size}'
                       + (f' vs {size2}x{size2} ' if size2 != size else ' ')
                       + f'(hot path only, features precomputed) in {dt:.1f} s; '
                       f'oracle/oetr_oracle.py on torch CPU, {best_t} intra-op '
                       f'threads (best of 4..64 on a {ncpu}-CPU host)'), boxes


def power_sample(run_steps, seconds=2.5):
    """Socket power, power cap and shader clock while `run_steps(k)` keeps the hot path running (rocm-smi sampled
    from a thread beside it, OUTSIDE every timed region).  The hot path runs the chip at its power cap - the clock
    it holds is what the cap allows - so the line records both.  None if rocm-smi is not there."""
    import re
    import shutil
    import subprocess
    import threading
    if shutil.which('rocm-smi') is None:
        return None
    samples, stop = [], []

    def smi(*flags):
        return subprocess.run(['rocm-smi', '-d', '0', *flags], capture_output=True, text=True, timeout=15).stdout

    def sampler():
        while not stop:
            try:
                out = smi('--showpower', '--showclocks')
                w = re.search(r'Socket Graphics Package Power \(W\):\s*([0-9.]+)', out)
                c = re.search(r'sclk clock level:[^(]*\((\d+)Mhz\)', out)
                if w and c:
                    samples.append((float(w.group(1)), int(c.group(1))))
            except Exception:
                return
    try:
        cap = re.search(r'Max Graphics Package Power \(W\):\s*([0-9.]+)', smi('--showmaxpower'))
        th = threading.Thread(target=sampler, daemon=True)
        t0 = time.perf_counter()
        run_steps(20)
        th.start()
        while time.perf_counter() - t0 < seconds:
            run_steps(20)
        stop.append(1)
        th.join(timeout=20)
    except Exception:
        return None
    if len(samples) < 2:
        return None
    samples = samples[1:]                      # (the first sample straddles the ramp)
    watts = sorted(w for w, _ in samples)
    clocks = sorted(c for _, c in samples)
    return {'socket_w': watts[len(watts) // 2], 'cap_w': float(cap.group(1)) if cap else None,
            'sclk_mhz': clocks[len(clocks) // 2], 'samples': len(samples), 'source': 'rocm-smi beside a %.1f-s run' % seconds}


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the latest COMMITTED rocprofv3 PMC passes
    (profiles/*_pmc_fetch.csv / *_pmc_write.csv, made by tools/profile_round.sh with this
    bench command on the same workload; the counters cannot be read from inside the
    process, so this is the builder's measurement replayed - `traffic_source` names the
    files).  Per MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in KiB and
    FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, so
    bytes = (2*FETCH + WRITE) * 1024.  Returns (bytes, source) or (None, None)."""
    import csv
    import glob
    out, src = {}, []
    for kind in ('fetch', 'write'):
        files = sorted(glob.glob(str(REPO / 'profiles' / f'*_pmc_{kind}.csv')))
        if not files:
            return None, None
        with open(files[-1]) as f:
            for row in csv.DictReader(f):
                if kernel_substr in row['kernel']:
                    out[kind] = float(row['FETCH_SIZE' if kind == 'fetch' else 'WRITE_SIZE'])
        src.append('profiles/' + Path(files[-1]).name)
    if len(out) != 2:
        return None, None
    return int((2 * out['fetch'] + out['write']) * 1024), ' + '.join(src)


def mangled_encoder(tile, mode_id, policy=0, attention='linear'):
    """Substring of the B;A encoder kernel's mangled name in rocprofv3 CSVs."""
    # (template arguments: HAS_B, TAIL, MODE, [waves, FULL,] policy, MASKED = false)
    if tile == 64:
        return f'k_encoder64ILb1ELi0ELi{mode_id}ELi{policy}ELb0EE'
    if mode_id == 1 and attention == 'linear':   # two-plane dtypes: the 32-row kernel on the 64-row kernel's body (encoder.hip: k_encoder32m)
        return f'k_encoder32mILb1ELi0ELi{mode_id}ELi{policy}ELb0EE'
    return f'k_encoderILb1ELi0ELi{mode_id}ELi{4 if mode_id == 0 else 8}ELb0ELi{policy}ELb0EE'


def roofline_block(kern, precision, tokens, tile, steps, traced_s, standard_workload, extra_flop=0, grids=None,
                   attention='linear'):
    if not kern or DOMINANT not in kern:
        return None
    launches, total_ms = kern[DOMINANT]
    raw_ms = total_ms / launches
    # the library brackets every launch with two HIP events on its stream; an event pair with NOTHING between
    # measures CALIB['event_pair_us'] on this box (main(): median of 200), and about three quarters of that is
    # what a bracket adds to the kernel it surrounds (the closing marker is fetched while the kernel still
    # runs) - same-process comparison with rocprofv3, profiles/r6_event_bracket.txt: k_encoder32m<B,A> 39.99 us
    # from the brackets, 35.59 us from rocprofv3, empty pair 5.64 us -> 0.78.  `frac` uses the net figure (the
    # one comparable with the committed rocprofv3 summaries), `frac_events_raw` the bracket as it reads
    avg_ms = max(raw_ms - 0.75 * CALIB['event_pair_us'] * 1e-3, 0.5 * raw_ms)
    flop = ENC_FLOP_PER_TOKEN * tokens + extra_flop   # algorithmic, both sides
    cost, pipe_peak, basis = MFMA_COST[precision]
    # `achieved` is ALGORITHMIC FLOP/s; the MFMA roof for a scheme that spends `cost`
    # MFMA products per algorithmic product is the pipe's dense peak / cost.
    peak = pipe_peak / cost
    ach = flop / (avg_ms * 1e-3) / 1e12
    sustained = CALIB['f16_sustained'] if pipe_peak == F16_MFMA_PEAK_TFLOPS else None
    block = {
        'kernel': ('k_encoder64<B,A>' if tile == 64 else
                   'k_encoder32m<B,A>' if precision in ('f32_split_f16', 'f32_split_qk16') and attention == 'linear'
                   else 'k_encoder<B,A>') + f' [{precision}]',
        'tile_rows': tile, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': round(peak, 1),
        'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'peak_basis': basis,
        'executed_mfma_tflops': round(ach * cost, 2),
        # (second yardstick, f16-pipe modes: the dense-MFMA rate the chip sustains at its power cap - measured, not nominal)
        **({'sustained_peak_measured': round(sustained / cost, 1),
            'frac_of_sustained_peak': round(ach / (sustained / cost), 4)} if sustained else {}),
        'avg_launch_us': round(avg_ms * 1e3, 2), 'avg_launch_us_events_raw': round(raw_ms * 1e3, 2),
        'event_pair_us': round(CALIB['event_pair_us'], 2), 'event_bracket_overhead_us': round(0.75 * CALIB['event_pair_us'], 2),
        'frac_events_raw': round(flop / (raw_ms * 1e-3) / 1e12 / peak, 4),
        'launches': launches, 'flop_per_launch': flop,
        'share_of_step': round(total_ms / (traced_s * 1e3), 4),
        'traced_ms_per_step': round(traced_s / steps * 1e3, 4)}
    if grids:   # one workgroup per CU (LDS): how much of the chip a launch of this batch can occupy
        n_pairs, l1, l2 = grids
        wgs = n_pairs * (-(-l1 // tile) + -(-l2 // tile))
        block['workgroups_per_launch'] = wgs
        block['cu_share'] = round(min(wgs, 256) / 256, 4)
        block['note'] = ('frac is this kernel ALONE on the chip; its launch has %d workgroups for 256 CUs '
                         '(one per CU), the overlapped streams fill the rest - the chip-level figure is '
                         'hot_path_frac_of_mfma_peak' % wgs)
    traffic, src = (pmc_traffic(mangled_encoder(tile, MODE_ID[precision], POLICY_ID.get(precision, 0), attention))
                    if standard_workload else (None, None))
    block['traffic'] = traffic
    if traffic is not None:
        block['traffic_source'] = src + ' (committed rocprofv3 PMC passes of this command, not read in-run)'
    return block


def bench_full_attention(args, device):
    """--kernel full_attention: the stand-alone flash-style FullAttention kernel
    (reference src/models/linear_attention.py:53-87), n = --pairs-per-gpu images, L = S."""
    import torch
    import imagematching_oetr_amd as pkg
    n, L = args.pairs_per_gpu, args.L
    g = torch.Generator().manual_seed(5)
    q = ((torch.rand(n, L, 8, 32, generator=g) - 0.5) * 4).to(device)
    k = ((torch.rand(n, L, 8, 32, generator=g) - 0.5) * 4).to(device)
    v = ((torch.rand(n, L, 8, 32, generator=g) - 0.5) * 2).to(device)
    flop = 4.0 * n * 8 * L * L * 32          # QK^T + PV, 2*MAC
    out = {'metric': f'FullAttention (all-pairs {L}x{L} softmax(QK^T)V, 8 heads x 32) launches/s',
           'unit': 'TFLOP/s (algorithmic)', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
           'higher_is_better': True, 'data': 'synthetic', 'vs_baseline': None,
           'config': {'workload': f'{n} images, L=S={L}, 8 heads, D=32, fp32 in/out'}}
    variants = {}
    for name in pkg.FULL_ATTENTION_VARIANTS:
        fn = lambda: pkg.full_attention(q, k, v, variant=name)
        for _ in range(args.warmup):
            fn()
        regions = []
        for _ in range(args.repeats):
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(args.steps):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            regions.append(ev0.elapsed_time(ev1) / args.steps)
        ms = statistics.median(regions)
        cost, pipe_peak, basis = MFMA_COST['f32' if name == 'f32' else 'f32_split_f16']
        ach = flop / (ms * 1e-3) / 1e12
        variants[name] = {'avg_launch_us': round(ms * 1e3, 2), 'achieved': round(ach, 2),
                          'peak': round(pipe_peak / cost, 1), 'frac': round(ach / (pipe_peak / cost), 4),
                          'unit': 'TFLOP/s', 'bound': 'mfma', 'peak_basis': basis}
    best = max(variants, key=lambda k_: variants[k_]['achieved'])
    out['value'] = variants[best]['achieved']
    out['ms_per_step'] = variants[best]['avg_launch_us'] / 1e3
    out['dtype'] = 'f32'
    out['roofline'] = dict(kernel=f'k_full_attention [{best}]', **variants[best])
    out['variants'] = variants
    print(json.dumps(out))



TRUNK_GFLOP_PER_IMAGE_640 = 53.5     # ResNet-50 conv1..layer3 at 640x640


def end_to_end(args, model, device, n, size2, pkg):
    """`forward_dummy` from images (reference src/model.py:229-252) with the default settings -
    enqueue-only, deferred range check - and where its time goes: torch / MIOpen trunk (host
    code by north_star), HIP neck, HIP hot path (each timed alone with events); plus the
    reference's per-pair calling pattern through the batched front end, and the same
    forward on the host cores."""
    import torch
    res = {}
    model = model.to(device)
    model.hip_precision = args.precision
    g = torch.Generator().manual_seed(2)
    im1 = torch.rand(n, args.size, args.size, 3, generator=g).to(device)
    im2 = torch.rand(n, size2, size2, 3, generator=g).to(device)

    def timed(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        model.hip_flush()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    reps = 10
    t_all = timed(lambda: model.forward_dummy(im1, im2), reps)
    res['end_to_end_pairs_per_s'] = round(n / t_all, 1)
    model.hip_defer_check = False          # the round-2 behaviour: one stream sync per call
    res['end_to_end_pairs_per_s_sync_check'] = round(n / timed(lambda: model.forward_dummy(im1, im2), reps), 1)
    model.hip_defer_check = True
    # throughput mode of the product path: the hot path of batch i on side stream i mod 3 under the trunk of batch i+1
    model.hip_streams, model.hip_throughput = 3, None
    res['end_to_end_pairs_per_s_streams3'] = round(n / timed(lambda: model.forward_dummy(im1, im2), reps), 1)
    model.hip_flush()
    model.hip_streams = 1
    # the reference's calling pattern: a stream of single pairs, bucketed by shape into batches of n
    pair_list = [(im1[i:i + 1], im2[i:i + 1]) for i in range(n)] * 2
    t_fp = timed(lambda: pkg.forward_pairs(model, pair_list, max_batch=n), 5)
    res['end_to_end_forward_pairs_per_s'] = round(len(pair_list) / t_fp, 1)
    # stage split (same-size pairs: one 2N-image trunk + neck call)
    same = args.size == size2
    imgs = torch.cat([im1, im2]) if same else im1
    t_trunk = timed(lambda: model.backbone(imgs), reps)
    bbf = model.backbone(imgs)
    n_img = int(bbf.shape[0])
    with pkg.KernelTrace(model.neck_engine()) as ntr:
        t_neck = timed(lambda: model.neck_engine().forward(bbf), reps)
    neck_us = {k: round(v[1] / v[0] * 1e3, 1) for k, v in ntr.summary().items()}
    gflop = TRUNK_GFLOP_PER_IMAGE_640 * (args.size / 640.0) ** 2 * n_img
    res['end_to_end'] = {
        'ms_per_batch': round(t_all * 1e3, 3),
        'trunk_ms': round(t_trunk * 1e3, 3), 'trunk_images': n_img,
        'neck_ms': round(t_neck * 1e3, 3), 'neck_kernels_us': neck_us,
        'hot_ms_serial': None,      # filled by the caller from the serial pass
        'trunk_tflops': round(gflop / t_trunk / 1e3, 1),
        'trunk_frac_of_f32_peak': round(gflop / t_trunk / 1e3 / F32_MFMA_PEAK_TFLOPS, 3),
        'trunk_share': round(t_trunk * (1 if same else 2) / t_all, 3),
        'note': 'trunk = torch / MIOpen fp32 convolutions (host code by north_star; channels_last, BN folding and '
                'MIOpen benchmark mode measured within 4 % of this: profiles/r3_trunk_probe.txt)'}
    return res


def cpu_full_forward(model, size, budget_s=20.0):
    """the whole forward_dummy (trunk + neck + hot path) on the host cores - the
    oracle's hot path behind this repo's torch trunk/neck modules (host code either way) -
    for N = 1 and N = 8 pairs, bounded sample."""
    import torch
    import imagematching_oetr_amd as pkg
    from oracle import oetr_oracle as orc
    m = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()       # (the GPU model's engines hold ctypes handles)
    m.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    w = {k: v.detach().cpu() for k, v in m.hot_path_state().items()}
    ncpu = os.cpu_count() or 1
    torch.set_num_threads(min(ncpu, 32))
    res = {}
    for n in (1, 8):
        g = torch.Generator().manual_seed(2)
        im1, im2 = torch.rand(n, size, size, 3, generator=g), torch.rand(n, size, size, 3, generator=g)

        def run():
            f = m._neck_torch(m.backbone(torch.cat([im1, im2])))
            return orc.hot_path(f[:n], f[n:], w, (size, size), (size, size))
        run()
        it, t0 = 0, time.perf_counter()
        while True:
            run()
            it += 1
            dt = time.perf_counter() - t0
            if dt > budget_s / 2 or it >= 20:
                break
        res[f'n{n}_pairs_per_s'] = round(n * it / dt, 3)
        res[f'n{n}_sample'] = f'{it} forward(s) of {n} pair(s) in {dt:.1f} s'
    res['cores'] = min(ncpu, 32)
    res['kind'] = 'port'
    return res


def _pair_flop(l1, l2):
    """Algorithmic FLOP of one B;A encoder launch per pair."""
    return ENC_FLOP_PER_TOKEN * (l1 + l2)


def other_configs(args, device, pkg):
    """BASELINE configs[2] / [3] / [4] under the SAME clock as `value`: short
    regions - median of 5 regions of `steps` steps, batches alternating over --streams streams like
    `value` - each with the roofline of its dominant kernel from a traced serial pass and an IoU
    check of a 2-pair slice against the CPU oracle OUTSIDE the timed regions.  Bounded to ~20 s."""
    import torch
    from oracle import oetr_oracle as orc
    t_start = time.perf_counter()
    cases = [
        # key, pairs, size1, size2, precision, forced tile (0 = auto), steps
        ('configs[3] 32 pairs @1024x1024, auto tile', 32, 1024, 1024, 'f32_split_f16', 0, 8),
        ('configs[3] 32 pairs @1024x1024, 32-row tile (LDS-tile sweep)', 32, 1024, 1024, 'f32_split_f16', 32, 8),
        ('configs[3] 32 pairs @1024x1024, 64-row tile (LDS-tile sweep)', 32, 1024, 1024, 'f32_split_f16', 64, 8),
        ('configs[4] share: 8 pairs 640x640 vs 1280x1280, default precision', 8, 640, 1280, 'f32_split_f16', 0, 16),
        ('configs[4] share: 8 pairs 640x640 vs 1280x1280, precision policy', 8, 640, 1280, 'f32_split_qk16', 0, 16),
        ('configs[2] share: 8 pairs @640x640, precision policy', 8, 640, 640, 'f32_split_qk16', 0, 40),
        # the literal "HW = 64x64 correlation volume" of configs[3] (SURVEY 8d "Config 4"): 4096 tokens per image
        ('configs[3] literal 64x64 tokens: 4 pairs @2048x2048', 4, 2048, 2048, 'f32_split_f16', 0, 6),
    ]
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
    out, inputs, engines = {}, {}, {}
    for key, n, s1, s2, prec, tile, steps in cases:
        ik = (n, s1, s2)
        if ik not in inputs:
            inputs[ik] = synthetic_inputs(n, s1, s2, device)
        model, weights, f1, f2, p1, p2, hf, hf2 = inputs[ik]
        if prec not in engines:
            engines[prec] = pkg.HotPathEngine(weights, device=device, precision=prec)
        eng = engines[prec]
        hw, hw2 = (s1, s1), (s2, s2)
        L1, L2 = hf * hf, hf2 * hf2
        eff_tile = tile or (64 if n_streams > 1 else 0)     # like `value`: 64-row tiles while batches overlap
        eng.set_encoder_tile(eff_tile)
        eng.set_tail_mode(args.tail_mode_overlap if n_streams > 1 else 0)   # ... and the direct tail

        def region(ns, k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(k):
                with torch.cuda.stream(streams[i % ns]):
                    eng.forward(f1, f2, p1, p2, hw, hw2)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        region(n_streams, 2)
        over = statistics.median(region(n_streams, steps) for _ in range(5))
        eng.set_encoder_tile(tile)                           # serial: the forced tile, or the library's choice
        eng.set_tail_mode(0)
        region(1, 2)
        ser = statistics.median(region(1, steps) for _ in range(5))
        eng.set_encoder_tile(eff_tile)
        with pkg.KernelTrace(eng, max_launches=24 * steps + 64) as tr:
            t_tr = region(1, steps)
            kern = tr.summary()
        cost, pipe_peak, _ = MFMA_COST[prec]
        pair_gflop = PAIR_GFLOP_640 * (L1 + L2) / 800
        rec = {'pairs_per_s': round(n * steps / over, 1), 'ms_per_step': round(over / steps * 1e3, 4),
               'serial_pairs_per_s': round(n * steps / ser, 1), 'steps': steps, 'streams': n_streams,
               'precision': prec, 'encoder_tile_rows': eff_tile or 'auto',
               'tail_mode_overlapped': args.tail_mode_overlap if n_streams > 1 else 0,
               'hot_path_frac_of_mfma_peak': round(n * steps / over * pair_gflop / 1e3 / (pipe_peak / cost), 4)}
        used_tile = eff_tile or 64      # (auto on one stream picks 64 rows whenever the 32-row grid exceeds the chip)
        rb = roofline_block(kern, prec, n * (L1 + L2), used_tile, steps, t_tr, False, 0, grids=(n, L1, L2))
        if rb:
            rec['roofline'] = {k: rb[k] for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac',
                                                  'avg_launch_us', 'launches', 'workgroups_per_launch',
                                                  'cu_share', 'traffic')}
        # parity on a 2-pair slice, outside every timed region
        b1, b2 = eng.forward(f1, f2, p1, p2, hw, hw2)
        torch.cuda.synchronize()
        w = {k: v.cpu() for k, v in weights.items()}
        r1, r2 = orc.hot_path(f1[:2].cpu(), f2[:2].cpu(), w, hw, hw2)
        iou = torch.cat([orc.bbox_iou_aligned(b1[:2].cpu(), r1), orc.bbox_iou_aligned(b2[:2].cpu(), r2)])
        rec['iou_vs_cpu_min_2pairs'] = round(float(iou.min()), 6)
        if eng.precision in eng.F16_RANGE:
            rec['f16_range_flag'] = eng.query_flags()
        out[key] = rec
    # the all-pairs kernel itself (reference FullAttention, linear_attention.py:53-87) at the same literal size:
    # L = S = 4096, 8 images, 8 heads x 32 - softmax(QK^T / sqrt D) V, the 4096 x 4096 volume never materialised
    try:
        g = torch.Generator().manual_seed(5)
        fq = ((torch.rand(8, 4096, 8, 32, generator=g) - 0.5) * 4).to(device)
        fk = ((torch.rand(8, 4096, 8, 32, generator=g) - 0.5) * 4).to(device)
        fv = ((torch.rand(8, 4096, 8, 32, generator=g) - 0.5) * 2).to(device)
        # (40 launches first: after an idle gap the socket's power controller overshoots - launches 5-8 of a burst take
        #  700+ us where the first took 575 - and needs ~50 launches, 25 ms, to settle at the sustained 515-520 us:
        #  tools/fa_each.py, profiles/r6_full_attention_p1plane.txt)
        for _ in range(40):
            pkg.full_attention(fq, fk, fv, variant='f32_split_f16')
        times = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                pkg.full_attention(fq, fk, fv, variant='f32_split_f16')
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / 10)
        ms = statistics.median(times)
        flop = 4.0 * 8 * 8 * 4096 * 4096 * 32
        cost, pipe_peak, basis = MFMA_COST['f32_split_f16']
        out['full_attention L=S=4096, 8 images'] = {
            'kernel': 'k_full_attention_split [f32_split_f16]', 'avg_launch_us': round(ms * 1e3, 1),
            'achieved': round(flop / ms / 1e9, 1), 'peak': round(pipe_peak / cost, 1), 'unit': 'TFLOP/s',
            'frac': round(flop / ms / 1e9 / (pipe_peak / cost), 4), 'bound': 'mfma', 'peak_basis': basis,
            'flop_per_launch': flop, 'timing': 'median of 5 back-to-back regions of 10 launches behind 40 warm-up launches, HIP events around each region'}
        del fq, fk, fv
    except Exception as e:
        out['full_attention L=S=4096, 8 images'] = {'error': repr(e)[:200]}
    out['_seconds'] = round(time.perf_counter() - t_start, 1)
    out['_note'] = ('each entry: median of 5 regions of `steps` steps, batches alternating over `streams` HIP streams '
                    '(64-row encoder tiles unless forced), `serial_pairs_per_s` = the same steps on one stream; '
                    'roofline = dominant kernel from a traced serial pass (HIP events on the launch stream); IoU of a '
                    '2-pair slice vs the CPU oracle outside the timed regions')
    return out


def bench_mixed(args, device, world, rank, use_pg, pkg):
    """--workload mixed: BASELINE configs[4] as a mixed-scale multi-GPU job (bucket by
    (L1, L2) BEFORE sharding so every rank runs equal work).  The global list holds
    world * --pairs-per-gpu pairs of 640x640 vs 640x640 and as many of 640x640 vs 1280x1280,
    interleaved; every rank buckets it, takes its contiguous `shard_bounds` slice of each bucket,
    runs one hot-path batch per bucket per step and all-gathers the boxes per bucket (BoxGatherer,
    asynchronous, under the next bucket's kernels).  Reported: whole-job pairs/s (max over ranks)
    and the per-rank step-time spread."""
    import torch
    import torch.distributed as dist
    from imagematching_oetr_amd.parallel import BoxGatherer, bucket_by_shape, shard_bounds
    n = args.pairs_per_gpu
    s_small, s_big = args.size, args.size2 or 2 * args.size
    shapes = []
    for i in range(n * world):
        shapes += [((s_small, s_small), (s_small, s_small)), ((s_small, s_small), (s_big, s_big))]
    buckets = bucket_by_shape(shapes)                  # same on every rank (first-seen order)
    work, eng = [], None
    for key, idx in buckets.items():
        lo, hi = shard_bounds(len(idx), rank, world)
        (h1, _), (h2, _) = key
        model, weights, f1, f2, p1, p2, hf, hf2 = synthetic_inputs(hi - lo, h1, h2, device)
        if eng is None:
            eng = pkg.HotPathEngine(weights, device=device, precision=args.precision)
        work.append(dict(key=key, n_bucket=len(idx), n_local=hi - lo, f1=f1, f2=f2, p1=p1, p2=p2,
                         hw=(h1, h1), hw2=(h2, h2), tokens=hf * hf + hf2 * hf2))
    gatherer = BoxGatherer(on_stream={'auto': None, '0': False, '1': True}[args.gather_on_stream]) if use_pg else None     # on-stream collective when submitted from a side stream
    streams = [torch.cuda.Stream(device=device) for _ in range(max(1, args.streams))]
    eng.set_encoder_tile(args.enc_tile or (64 if len(streams) > 1 else 0))

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    def region(k):
        barrier()
        t0 = time.perf_counter()
        j = 0
        for _ in range(k):
            for wk in work:
                with torch.cuda.stream(streams[j % len(streams)]):
                    b1, b2 = eng.forward(wk['f1'], wk['f2'], wk['p1'], wk['p2'], wk['hw'], wk['hw2'])
                    if gatherer is not None:
                        gatherer.submit(b1, b2, n_pairs=wk['n_bucket'])
                j += 1
        if gatherer is not None:
            gatherer.flush()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0       # this rank's own time (before the closing barrier)
        barrier()
        dt = time.perf_counter() - t0
        if use_pg:
            t = torch.tensor([dt, mine], device=device, dtype=torch.float64)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            return max(float(e[0]) for e in every), [float(e[1]) for e in every]
        return dt, [mine]
    region(max(1, args.warmup))
    regs = sorted((region(args.steps) for _ in range(max(1, args.repeats))), key=lambda r: r[0])
    dt, per_rank = regs[len(regs) // 2]
    if rank != 0:
        return None
    pairs_step = sum(wk['n_bucket'] for wk in work)
    cost, pipe_peak, _ = MFMA_COST[args.precision]
    gflop_step = sum(wk['n_bucket'] * PAIR_GFLOP_640 * wk['tokens'] / 800 for wk in work)
    return {
        'metric': f'image-pairs/sec, mixed-scale job ({s_small}x{s_small} vs {s_small}x{s_small} and vs '
                  f'{s_big}x{s_big}; OETR hot path, features resident in HBM)',
        'value': round(pairs_step * args.steps / dt, 1), 'unit': 'image-pairs/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if args.precision in ('f32', 'f32_split_f16') else args.precision,
        'gemm_mode': GEMM_MODE_TEXT[args.precision], 'data': 'synthetic',
        'config': {'workload': f'BASELINE configs[4] mixed-scale: {n * world} pairs {s_small}^2 vs {s_small}^2 + '
                               f'{n * world} pairs {s_small}^2 vs {s_big}^2 per step, bucketed by (L1, L2) then '
                               f'sharded contiguously over {world} rank(s), {args.precision}',
                   'buckets': [{'shapes': [list(k[0]), list(k[1])], 'global_pairs': wk['n_bucket'],
                                'pairs_this_rank': wk['n_local']} for k, wk in zip(buckets, work)],
                   'streams': len(streams), 'parallelism': f'every bucket sharded over {world} rank(s); one '
                                                           'all-gather of [n_local,2,4] boxes per bucket'},
        'rank_step_ms': {'per_rank': [round(t / args.steps * 1e3, 4) for t in per_rank],
                         'min': round(min(per_rank) / args.steps * 1e3, 4),
                         'max': round(max(per_rank) / args.steps * 1e3, 4),
                         'spread': round((max(per_rank) - min(per_rank)) / max(per_rank), 4)},
        'hot_path_frac_of_mfma_peak': round(args.steps / dt * gflop_step / 1e3 / (pipe_peak / cost) / world, 4),
    }


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(respawn_under_torchrun(args))

    if int(os.environ.get('RANK', 0)) != 0:
        # only rank 0 reports: keep the other ranks' C-level stdout (RCCL's version banner is
        # printed through C stdio at exit) away from the one JSON line
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        if rank == 0:
            print(f'[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with '
                  f'`python bench.py --gpus {args.gpus}` (self-spawning) or torchrun '
                  f'--nproc-per-node {args.gpus}', file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    # (dry runs of the N>1 logic on a 1-GPU box: OETR_BENCH_BACKEND=gloo maps every
    #  rank onto the GPUs that exist; the real launch is one rank per GPU over RCCL)
    backend = os.environ.get('OETR_BENCH_BACKEND', 'nccl')
    if backend == 'nccl' and torch.cuda.device_count() < world:
        if rank == 0:
            print(f'[bench] {world} ranks but {torch.cuda.device_count()} GPU(s)', file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    # OETR_BENCH_FORCE_PG=1: bring the process group up even at world size 1, so that the RCCL
    # code path (librccl load, device binding, BoxGatherer's asynchronous all_gather_into_tensor,
    # the timing all-reduce) executes on a 1-GPU box exactly as it does at N > 1
    force_pg = os.environ.get('OETR_BENCH_FORCE_PG', '0') == '1'
    use_pg = world > 1 or force_pg
    if use_pg:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        kw = {} if 'RANK' in os.environ else dict(rank=0, world_size=1)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device, **kw)
        else:
            dist.init_process_group(backend, **kw)
        world = dist.get_world_size()          # what actually came up
    torch.set_grad_enabled(False)

    if args.kernel == 'full_attention':
        return bench_full_attention(args, device)

    import imagematching_oetr_amd as pkg
    from imagematching_oetr_amd.parallel import BoxGatherer
    if args.workload == 'mixed':
        res = bench_mixed(args, device, world, rank, use_pg, pkg)
        if use_pg:
            if rank == 0:
                res['process_group'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size()}
            dist.destroy_process_group()
        if rank == 0:
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            print(json.dumps(res), flush=True)
        return
    n = args.pairs_per_gpu
    size2 = args.size2 or args.size
    model, weights, feat1, feat2, pos, pos2, hf, hf2 = synthetic_inputs(n, args.size, size2, device)
    hw, hw2 = (args.size, args.size), (size2, size2)
    n_total = n * world
    tokens = n * (hf * hf + hf2 * hf2)
    standard = (n, args.size, size2) == (8, 640, 640) and args.attention == 'linear'
    L1, L2 = hf * hf, hf2 * hf2
    # attention='full' adds QK^T and PV: 4*L*S*C per encoder call and image (self / cross average)
    extra_flop = 0 if args.attention == 'linear' else 4 * 256 * n * (L1 * L1 + L2 * L2 + 2 * L1 * L2) // 2

    # (mutable: the RCCL-at-world-1 region of an N = 1 run brings a group up for one measurement, below)
    pg = {'use': use_pg,
          'gatherer': BoxGatherer(on_stream={'auto': None, '0': False, '1': True}[args.gather_on_stream]) if use_pg else None}
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]

    def barrier():
        if pg['use']:
            dist.barrier()
        torch.cuda.synchronize()

    # Every timed step goes through the PRODUCT path: OETR.boxes_from_features (what forward_dummy
    # calls behind the trunk) - the model picks the stream (hip_streams: throughput mode, batch i on
    # side stream i mod k with the engines' throughput settings) and enqueues the deferred range
    # check behind every batch; hip_flush() at the end of the region settles them in order.
    model.to(device)
    model.hip_freeze_weights = True           # (documented knob: skip the per-call parameter identity check)
    models = {}

    def model_for(precision):
        if precision not in models:
            if not models:
                m = model
            else:      # a second module on the same weights (the engines hold ctypes handles: no deepcopy)
                m = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
                m.load_state_dict(model.state_dict())
                m = m.to(device)
                m.hip_freeze_weights = True
            m.hip_precision = precision
            m.hip_attention = args.attention if precision == args.precision else 'linear'
            m.hip_enc_tile = args.enc_tile or None
            models[precision] = m
        return models[precision]

    def configure(m, ns, overlapped):
        """Stream count + engine settings of a region, OUTSIDE the timed part (the setters need an idle
        device).  Defaults = the model's own policy; the --*-overlap / --decoder-split flags are A/B knobs."""
        m.hip_flush()
        m.hip_streams = ns
        m.hip_throughput = bool(overlapped)
        eng = m.engine()
        half = eng.precision != 'f32'
        if overlapped:
            if args.tail_mode_overlap != 2 and eng.precision in ('f32_split_f16', 'f32_split_qk16'):
                eng.set_tail_mode(args.tail_mode_overlap)
            if eng.attention == 'linear':
                eng.set_state_prereduce(args.prereduce_overlap)
            m.hip_decoder_split = args.decoder_split_overlap or None
        else:
            if eng.attention == 'linear':
                eng.set_state_prereduce(-1)
            m.hip_decoder_split = args.decoder_split or None
        return eng, (args.enc_tile or (64 if (overlapped and half and eng.attention == 'linear') else 0))

    step_spread = {}

    def run_mode(m, ns):
        """One timed region: exactly --steps steps over `ns` streams."""
        def step(i):
            # consecutive steps alternate over the model's streams (one workspace per stream in
            # the engine); every step is a full batch of n pairs through the whole path
            b1, b2 = m.boxes_from_features(feat1, feat2, pos, pos2, hw, hw2)
            gatherer = pg['gatherer']
            if gatherer is not None:
                # the all-gather of a batch's boxes is ISSUED once the model has settled that batch's deferred
                # check (BoxGatherer(model=...): 2k batches later, from this submit - no rank ever receives boxes
                # that are corrected afterwards): on that batch's own side stream in the throughput mode (a
                # blocking collective there - the other side streams carry on), on RCCL's stream under the
                # next batch's kernels in the serial mode; completed at a later submit / the flush
                gatherer.model = m
                with torch.cuda.stream(m.hip_batch_stream()):
                    gatherer.submit(b1, b2)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        m.hip_flush()                   # every batch's range check settled, in submission order
        if pg['gatherer'] is not None:
            pg['gatherer'].flush()      # the last batches' gathers are inside the timed region
        barrier()                       # closing barrier + synchronise of the bracket: inside the timed region
        dt = time.perf_counter() - t0   # (under RCCL the barrier collective's own host latency is 0.2-0.45 ms: 4-9 % of a
                                        #  20-step region of 5 ms, nothing at --steps 200; profiles/r5_pg_streams.txt)
        if pg['use']:
            t = torch.tensor([dt, -dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            step_spread['last'] = (float(-t[1].item()), float(t[0].item()))   # (fastest, slowest) rank
            dt = float(t[0].item())
        return dt

    def warm(m, ns):
        saved, args.steps = args.steps, max(1, args.warmup)
        run_mode(m, ns)
        args.steps = saved

    def repeated(m, ns):
        regions = sorted(run_mode(m, ns) for _ in range(max(1, args.repeats)))
        return statistics.median(regions), regions[0], regions[-1]

    def traced(m, eng):
        with pkg.KernelTrace(eng, max_launches=24 * args.steps + 64) as trace:
            dt = run_mode(m, 1)
        return trace.summary(), dt

    def measure(precision, with_serial_trace):
        m = model_for(precision)
        eng, _ = configure(m, 1, False)
        res = {'engine': eng, 'model': m}
        warm(m, 1)
        eng, tile_overlap = configure(m, n_streams, n_streams > 1)
        res['tile_overlap'] = tile_overlap
        warm(m, n_streams)
        res['overlap'] = repeated(m, n_streams)              # -> value (no instrumentation)
        res['rank_spread'] = step_spread.get('last')
        if not args.no_trace:
            # same K steps on ONE stream (kernel durations only mean something when launches
            # do not share the chip), in the tile shape that produced `value`, with the
            # library's per-kernel HIP events recorded on its launch stream
            configure(m, 1, n_streams > 1)
            res['trace_overlap_shape'] = traced(m, eng)
        configure(m, 1, False)
        res['serial'] = repeated(m, 1) if n_streams > 1 else res['overlap']
        if not args.no_trace and with_serial_trace and tile_overlap != (args.enc_tile or 0):
            res['trace_serial_shape'] = traced(m, eng)
        return res

    if world == 1 and not use_pg and not args.no_trace:
        # the two yardsticks that belong to THIS box and THIS run: what an empty HIP-event bracket reads, and the
        # dense f16 MFMA rate the chip sustains at its power cap (with the clock it held, from rocm-smi beside it)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
        torch.cuda.synchronize()
        for a, b in evs:
            a.record()
            b.record()
        torch.cuda.synchronize()
        CALIB['event_pair_us'] = statistics.median(a.elapsed_time(b) for a, b in evs) * 1e3
        if not args.no_power:
            try:
                from imagematching_oetr_amd.hip_engine import sustained_mfma_tflops
                rates = []
                CALIB['f16_sustained_power'] = power_sample(lambda k: rates.append(sustained_mfma_tflops(device, 0.5)),
                                                            seconds=1.5)
                CALIB['f16_sustained'] = statistics.median(rates) if rates else None
            except Exception as e:
                print(f'[bench] MFMA-rate calibration failed: {e!r}', file=sys.stderr)
    main_res = measure(args.precision, with_serial_trace=True)
    eng = main_res['engine']
    power = None
    if world == 1 and not use_pg and not args.no_power:
        # what the chip draws and clocks while each mode runs (not timed; rank 0 at N = 1 only)
        m = main_res['model']

        def keep_running(ns, overlapped):
            def go(k):
                for _ in range(k):
                    m.boxes_from_features(feat1, feat2, pos, pos2, hw, hw2)
                m.hip_flush()
                torch.cuda.synchronize()
            return go
        configure(m, n_streams, n_streams > 1)
        power = {'overlapped': power_sample(keep_running(n_streams, True))}
        configure(m, 1, False)
        power['serial'] = power_sample(keep_running(1, False))
    exact_res = None
    if args.precision == 'f32_split_f16' and not args.no_exact_f32 and args.attention == 'linear':
        exact_res = measure('f32', with_serial_trace=False)
    # RCCL on the N = 1 line: the same two regions with a process group of ONE rank up -
    # librccl loaded and bound to the device, the box all-gather of every batch issued by BoxGatherer, the
    # bracket's dist.barrier() and the timing all-reduce executed - exactly the code an N > 1 launch runs.
    # 5 x --steps per region (the barrier collective's own host latency, 0.2-0.45 ms, is a fixed cost per
    # region: profiles/r5_pg_streams.txt).  RCCL's banner goes through C stdio: stdout is parked on stderr.
    rccl_w1 = None
    if world == 1 and not use_pg and standard and args.precision == 'f32_split_f16' and not args.no_rccl_world1:
        saved_fd, saved_steps = os.dup(1), args.steps
        try:
            sys.stdout.flush()
            os.dup2(2, 1)
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{free_port()}', rank=0, world_size=1,
                                    device_id=device)
            pg['use'], pg['gatherer'] = True, BoxGatherer()
            m = main_res['model']
            args.steps = 5 * saved_steps
            configure(m, n_streams, n_streams > 1)
            warm(m, n_streams)
            ov = repeated(m, n_streams)
            configure(m, 1, False)
            warm(m, 1)
            se = repeated(m, 1)
            args.steps = saved_steps
            pg['use'], pg['gatherer'] = False, None
            configure(m, n_streams, n_streams > 1)       # the same 5 x regions without the group: the comparison
            args.steps = 5 * saved_steps
            warm(m, n_streams)
            ov0 = repeated(m, n_streams)
            with open('/proc/self/maps') as f:
                mapped = any('librccl' in line for line in f)
            rccl_w1 = {'pairs_per_s': round(n_total * args.steps / ov[0], 1),
                       'serial_pairs_per_s': round(n_total * args.steps / se[0], 1),
                       'pairs_per_s_no_group_same_region': round(n_total * args.steps / ov0[0], 1),
                       'steps_per_region': args.steps, 'librccl_mapped': mapped,
                       'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                       'collective': 'all_gather_into_tensor of the [2,n,4] box block per batch, issued once the '
                                     'batch\'s deferred check has settled (BoxGatherer(model=...))'}
        except Exception as e:
            rccl_w1 = {'error': repr(e)[:300]}
        finally:
            args.steps = saved_steps
            pg['use'], pg['gatherer'] = False, None
            try:
                if dist.is_initialized():
                    torch.cuda.synchronize()
                    dist.destroy_process_group()
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    if rank != 0:
        if use_pg:
            dist.destroy_process_group()
        return

    elapsed, e_min, e_max = main_res['overlap']
    ms_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed
    cost, pipe_peak, _ = MFMA_COST[args.precision]
    pair_gflop = PAIR_GFLOP_640 * (hf * hf + hf2 * hf2) / 800
    tile_overlap = main_res['tile_overlap']
    dtype = {'f32': 'f32', 'f32_split_f16': 'f32 (3xf16-split operands, fp32 accumulate)', 'f16': 'f16', 'bf16': 'bf16',
             'f32_split_qk16': 'f32 (3xf16-split operands, fp32 accumulate; Q/K projections single f16)'}[args.precision]
    tag = ''
    if standard and args.precision in ('f32', 'f32_split_f16'):
        tag = 'BASELINE configs[1]: '
    elif standard and args.precision == 'f32_split_qk16':
        tag = 'BASELINE configs[2] per-GPU share (64 pairs / 8 GPUs; reduced-precision policy inside the IoU bar): '
    elif standard and args.precision == 'bf16':
        tag = 'all-rounded bf16 (misses the IoU bar; comparison only), configs[2] shape: '
    elif (n, args.size, size2) == (8, 640, 1280) and args.precision == 'f32_split_qk16':
        tag = 'BASELINE configs[4] per-GPU share (reduced-precision policy inside the IoU bar): '
    elif (n, args.size, size2) == (8, 640, 1280) and args.precision == 'f16':
        tag = 'all-rounded f16 (misses the IoU bar; comparison only), configs[4] shape: '
    out = {
        'metric': f'image-pairs/sec @{args.size}x{args.size} (OETR hot path: '
                  'feature correlation + overlap regression, features resident in HBM)',
        'value': round(value, 1), 'unit': 'image-pairs/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_step, 4), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': dtype,
        'gemm_mode': GEMM_MODE_TEXT[args.precision],
        'data': 'synthetic',
        'timing': {'repeats': args.repeats, 'statistic': 'median region of --steps steps',
                   'ms_per_step_min': round(e_min / args.steps * 1e3, 4),
                   'ms_per_step_max': round(e_max / args.steps * 1e3, 4)},
        'config': {'workload': tag + f'batch={n} pairs/GPU, {args.size}x{args.size}'
                               + (f' vs {size2}x{size2}' if size2 != args.size else '')
                               + f' -> {hf}x{hf}' + (f' / {hf2}x{hf2}' if hf2 != hf else '')
                               + f' tokens/image, C=256, 8 enc + 2 dec layers, {args.precision}'
                               + (', attention=full' if args.attention == 'full' else ''),
                   'pairs_per_gpu': n, 'global_pairs': n_total,
                   'streams': n_streams,
                   'api': 'OETR.boxes_from_features with model.hip_streams = %d (product path; module default of two batches '
                          'queued per stream; deferred range check per batch, settled by hip_flush inside the timed region)' % n_streams,
                   'world_size': world, 'gpu_max_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'),
                   'rank_step_ms_fastest_slowest': ([round(v / args.steps * 1e3, 4) for v in main_res['rank_spread']]
                                                     if main_res.get('rank_spread') else None),
                   'encoder_tile_rows': tile_overlap or 'auto',
                   'tail_mode': ({0: 'library rule', 1: 'P form', 2: 'direct form (throughput setting for overlapped streams)',
                                  3: 'direct form, per-tap staging'}[args.tail_mode_overlap]
                                 if n_streams > 1 and args.precision in ('f32_split_f16', 'f32_split_qk16') else 'library rule'),
                   'tokens_per_image': hf * hf,
                   'parallelism': f'pairs sharded over {world} rank(s); '
                                  'all-gather of boxes only'},
        'hot_path_tflops': round(value * pair_gflop / 1e3, 2),
        'hot_path_frac_of_mfma_peak': round(value * pair_gflop / 1e3 / (pipe_peak / cost), 4),
    }
    if power and (power.get('overlapped') or power.get('serial')):
        # the roof that binds: the hot path runs the socket at its power cap in both modes, and the shader clock is
        # what the cap leaves (MFMA peaks in `roofline` are quoted at the nominal 2.4 GHz)
        out['power'] = power
    s_med, s_min, s_max = main_res['serial']
    # the same K steps strictly one after the other on one stream (batch latency;
    # encoder tile = library default)
    out['serial'] = {'ms_per_step': round(s_med / args.steps * 1e3, 4),
                     'pairs_per_s': round(n_total * args.steps / s_med, 1),
                     'ms_per_step_min': round(s_min / args.steps * 1e3, 4),
                     'ms_per_step_max': round(s_max / args.steps * 1e3, 4)}
    if 'trace_overlap_shape' in main_res:
        kern, t_s = main_res['trace_overlap_shape']
        rb = roofline_block(kern, args.precision, tokens, tile_overlap or 32, args.steps, t_s, standard, extra_flop,
                            grids=(n, hf * hf, hf2 * hf2), attention=args.attention)
        if rb:
            out['roofline'] = rb
            if out.get('power', {}).get('overlapped'):   # (kept keys of the driver's record: roofline, config)
                pw = out['power']['overlapped']
                rb['socket_w_cap_w_sclk_mhz'] = [pw['socket_w'], pw['cap_w'], pw['sclk_mhz']]
                rb['power_note'] = ('socket power / cap / shader clock while this mode runs (rocm-smi beside a 2.5-s run, outside the '
                                    'timed regions); `peak` is quoted at the nominal 2400 MHz')
            out['kernels_us'] = {k: [v[0] // args.steps, round(v[1] / v[0] * 1e3, 2)]
                                 for k, v in kern.items()}
            out['kernels_us_sum'] = round(sum(v[1] for v in kern.values()) / args.steps * 1e3, 1)
    if 'trace_serial_shape' in main_res:
        kern, t_s = main_res['trace_serial_shape']
        rb = roofline_block(kern, args.precision, tokens, 64 if args.precision in POLICY_ID else (args.enc_tile or 32), args.steps, t_s, standard, extra_flop,
                            grids=(n, hf * hf, hf2 * hf2), attention=args.attention)
        if rb:
            out['serial']['roofline'] = rb
            out['serial']['kernels_us'] = {k: [v[0] // args.steps, round(v[1] / v[0] * 1e3, 2)]
                                           for k, v in kern.items()}
            # (per-kernel HIP events inflate each launch by their own cost: the sum exceeds the untraced step)
            out['serial']['kernels_us_sum'] = round(sum(v[1] for v in kern.values()) / args.steps * 1e3, 1)
    if exact_res is not None:
        x_med, x_min, x_max = exact_res['overlap']
        xs_med = exact_res['serial'][0]
        out['exact_f32'] = {'gemm_mode': GEMM_MODE_TEXT['f32'],
                            'pairs_per_s': round(n_total * args.steps / x_med, 1),
                            'ms_per_step': round(x_med / args.steps * 1e3, 4),
                            'ms_per_step_min': round(x_min / args.steps * 1e3, 4),
                            'ms_per_step_max': round(x_max / args.steps * 1e3, 4),
                            'serial_pairs_per_s': round(n_total * args.steps / xs_med, 1)}
        if 'trace_overlap_shape' in exact_res:
            kern, t_s = exact_res['trace_overlap_shape']
            rb = roofline_block(kern, 'f32', tokens, 32, args.steps, t_s, standard)
            if rb:
                out['exact_f32']['roofline'] = rb
    if 'roofline' in out:
        # the stored record keeps `roofline` and `config` whole and drops other top-level keys: what a reader of that
        # record needs beside the headline goes in here, compact
        r = out['roofline']
        r['serial'] = [out['serial']['pairs_per_s'], out['serial']['ms_per_step']]          # [pairs/s, ms per step]
        if exact_res is not None:
            xr = out['exact_f32'].get('roofline') or {}
            # the strict-fp32 build (v_mfma_f32_32x32x2_f32 products): [pairs/s overlapped, pairs/s serial, frac of the 157.3-TFLOP/s fp32 MFMA peak]
            r['exact_f32'] = [out['exact_f32']['pairs_per_s'], out['exact_f32']['serial_pairs_per_s'], xr.get('frac')]
        if CALIB.get('f16_sustained'):
            pw = CALIB.get('f16_sustained_power') or {}
            r['sustained_f16_mfma_measured'] = {'tflops': round(CALIB['f16_sustained'], 1),
                                                'socket_w_cap_w_sclk_mhz': [pw.get('socket_w'), pw.get('cap_w'), pw.get('sclk_mhz')],
                                                'how': 'oetr_debug_mfma_rate: every CU on back-to-back v_mfma_f32_32x32x16_f16 for 3 x 0.5 s '
                                                       'in this run, rocm-smi beside it; nominal dense peak 2500'}
        if rccl_w1 is not None:
            r.setdefault('other_configs', {})['c1_rccl_world1'] = (
                [rccl_w1['pairs_per_s'], rccl_w1['serial_pairs_per_s'], rccl_w1['pairs_per_s_no_group_same_region'],
                 rccl_w1['librccl_mapped']] if 'error' not in rccl_w1 else rccl_w1)
    if rccl_w1 is not None:
        out['rccl_world1'] = rccl_w1
    if not args.no_cpu_baseline and world == 1:   # host-core baseline: rank 0 at N=1 only
        base, ref_boxes = cpu_baseline(weights, feat1, feat2, args.size, size2)
        out['cpu_baseline'] = base
        from oracle import oetr_oracle as orc
        mine = eng.forward(feat1, feat2, pos, pos2, hw, hw2)
        iou = torch.cat([orc.bbox_iou_aligned(mine[0].cpu(), ref_boxes[0]),
                         orc.bbox_iou_aligned(mine[1].cpu(), ref_boxes[1])])
        out['iou_vs_cpu_min'] = round(float(iou.min()), 6)
        out['speedup_vs_cpu'] = round(value / base['value'], 1)
    if eng.precision in eng.F16_RANGE:
        out['f16_range_flag'] = eng.query_flags()
    if not args.no_e2e and world == 1:
        try:     # whole forward_dummy incl. the PyTorch/MIOpen backbone (host code)
            out.update(end_to_end(args, model, device, n, size2, pkg))
            out['end_to_end']['hot_ms_serial'] = out['serial']['ms_per_step']
            if not args.no_cpu_baseline and args.size == size2:
                out['cpu_baseline_full_forward'] = cpu_full_forward(model, args.size)
        except Exception as e:  # host-side extras must never break the bench line
            out['end_to_end_error'] = repr(e)[:200]
    if (not args.no_other_configs and world == 1 and standard and args.precision == 'f32_split_f16'
            and not args.no_trace):
        try:     # configs[2] / [3] / [4] under the same clock (short regions, ~20 s)
            oc = other_configs(args, device, pkg)
            out['other_configs'] = oc
            if 'roofline' in out:
                def compact(prefix):
                    for k, v in oc.items():
                        if k.startswith(prefix) and isinstance(v, dict):
                            r = v.get('roofline', {})
                            return [v['pairs_per_s'], r.get('frac'), r.get('avg_launch_us'), v['serial_pairs_per_s']]
                    return None
                # [pairs/s overlapped, frac of the dominant kernel's roof, its average launch us, pairs/s serial]
                out['roofline'].setdefault('other_configs', {}).update({
                    'c3_32p_1024': compact('configs[3] 32 pairs @1024x1024, auto'),
                    'c3_32p_1024_tile32': compact('configs[3] 32 pairs @1024x1024, 32-row'),
                    'c4_8p_640v1280': compact('configs[4] share: 8 pairs 640x640 vs 1280x1280, default'),
                    'c4_policy': compact('configs[4] share: 8 pairs 640x640 vs 1280x1280, precision policy'),
                    'c2_policy': compact('configs[2] share'),
                    'c3_4p_2048': compact('configs[3] literal 64x64 tokens')})
                fa = oc.get('full_attention L=S=4096, 8 images') or {}
                # [TFLOP/s algorithmic, frac of the 3-MFMA split roof, launch us]
                out['roofline']['other_configs']['full_attention_L4096'] = (
                    [fa.get('achieved'), fa.get('frac'), fa.get('avg_launch_us')] if 'error' not in fa else fa)
        except Exception as e:
            out['other_configs_error'] = repr(e)[:300]
    if use_pg:
        out['process_group'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                                'forced_at_world_1': bool(force_pg and world == 1),
                                'collective': 'all_gather_into_tensor of [n_local,2,4] boxes per step (BoxGatherer: blocking on the batch\'s side stream in the throughput mode, asynchronous on the group\'s stream in the serial mode)'}
    if use_pg:
        dist.destroy_process_group()
    # RCCL prints a version banner through C stdio: flush it BEFORE the one JSON line, which is
    # then the last line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
