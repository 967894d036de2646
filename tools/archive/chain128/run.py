#!/usr/bin/env python3
"""Gate-1 driver for tools/chain128/chain128.hip (GPU box): correctness of the token-split
merge -> LN2 -> MLP chain against an fp64 restatement of the same lines of the reference
(src/models/transformer.py:131-142), then timing with the chip full.

  python tools/chain128/run.py [--variants 0,1,2] [--rounds 4] [--iters 20] [--out gpurun_out/chain128.json]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
C, FF = 256, 512
FLOP_PER_TOKEN = 2 * (C * C + C * FF + FF * C)   # merge + MLP1 + MLP2 = 655 360
SPLIT_ROOF = 2500.0 / 3.0                        # TFLOP/s algorithmic: three f16 MFMAs per product


def frag(W, mt, s):
    """A-operand fragment (m-tile mt, k16-step s) of W[out][in]: [64 lanes][8 slots] fp32."""
    lane = np.arange(64)
    rows = 32 * mt + (lane & 31)
    i = np.arange(8)
    cols = 16 * s + 8 * (i[None, :] >> 2) + 4 * (lane[:, None] >> 5) + (i[None, :] & 3)
    return W[rows[:, None], cols]


def planes(w, PL):
    wh = w.astype(np.float16)
    wl = ((w - wh.astype(np.float32)) * 2048.0).astype(np.float16)
    if PL == 3:
        whs = (wh.astype(np.float32) * 2048.0).astype(np.float16)
        assert np.all(np.isfinite(whs.astype(np.float32)))
        return [whs, wh, wl]
    return [wh, wl]


def pack_stream(Wm, W1, W2, PL, G, R):
    frs = []
    for s in range(16):
        for mt in range(8):
            frs.append(frag(Wm, mt, s))

    def m1(cp):   # chunk pair: hidden channels 64 cp .. 64 cp + 63
        for s in range(16):
            for m in range(2):
                frs.append(frag(W1, 2 * cp + m, s))

    def m2(cp):
        for ks in range(4):
            for mt in range(8):
                frs.append(frag(W2, mt, 4 * cp + ks))

    m1(0)
    for cp in range(1, 8):
        m1(cp)
        m2(cp - 1)
    m2(7)
    assert len(frs) == 640 and len(frs) % G == 0
    out = []
    for f in frs:
        for p in planes(f, PL):
            out.append(p.reshape(-1).view(np.uint16))
    buf = np.concatenate(out)
    pad = np.zeros(G * R * PL * 512 + 4096, np.uint16)   # groups issued past the end of the stream
    return np.concatenate([buf, pad])


def reference(msg, x, Wm, W1, W2, g, b, dtype):
    msg, x, Wm, W1, W2, g, b = [torch.from_numpy(a).to(dtype) for a in (msg, x, Wm, W1, W2, g, b)]
    x1 = x + msg @ Wm.T
    ln = torch.nn.functional.layer_norm(x1, (C,), g, b, 1e-5)
    hid = torch.nn.functional.gelu(ln @ W1.T)
    return (x1 + hid @ W2.T).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,1,2,3,4,5")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="")
    ap.add_argument("--flags", default="0,1")
    ap.add_argument("--nwg", default="0", help="workgroup counts to time (0 = CUs x rounds)")
    args = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(HERE, "libchain128.so"))
    lib.chain128_run.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.chain128_run.restype = ctypes.c_int
    dev = torch.device("cuda:0")
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    rng = np.random.default_rng(5)
    Wm = rng.uniform(-1, 1, (C, C)).astype(np.float32) / 16.0
    W1 = rng.uniform(-1, 1, (FF, C)).astype(np.float32) / 16.0
    W2 = rng.uniform(-1, 1, (C, FF)).astype(np.float32) / np.sqrt(FF).astype(np.float32)
    g = (1.0 + 0.2 * rng.standard_normal(C)).astype(np.float32)
    b = (0.1 * rng.standard_normal(C)).astype(np.float32)
    results = []
    for v in [int(t) for t in args.variants.split(",")]:
        PL, G, R = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if lib.chain128_variant_info(v, ctypes.byref(PL), ctypes.byref(G), ctypes.byref(R)) != 0:
            continue
        PL, G, R = PL.value, G.value, R.value
        ws = torch.from_numpy(pack_stream(Wm, W1, W2, PL, G, R).view(np.int16)).to(dev)
        gd, bd = torch.from_numpy(g).to(dev), torch.from_numpy(b).to(dev)
        rec = {"variant": v, "PL": PL, "G": G, "R": R}
        # ---- correctness: 3 workgroups + a ragged one
        T = 3 * 128 + 40
        msg = rng.standard_normal((T, C)).astype(np.float32)
        x = rng.standard_normal((T, C)).astype(np.float32)
        md, xd = torch.from_numpy(msg).to(dev), torch.from_numpy(x).to(dev)
        yd = torch.full((T, C), float("nan"), device=dev)
        st = torch.cuda.current_stream().cuda_stream
        rc = lib.chain128_run(v, md.data_ptr(), xd.data_ptr(), yd.data_ptr(), ws.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                              T, None, st, 0)
        torch.cuda.synchronize()
        y = yd.cpu().numpy()
        ref64 = reference(msg, x, Wm, W1, W2, g, b, torch.float64)
        ref32 = reference(msg, x, Wm, W1, W2, g, b, torch.float32)
        err = float(np.max(np.abs(y - ref64)))
        err32 = float(np.max(np.abs(ref32 - ref64)))
        rec.update(rc=rc, max_abs_err_vs_fp64=err, torch_fp32_cpu_err_vs_fp64=err32, finite=bool(np.all(np.isfinite(y))),
                   rms_err=float(np.sqrt(np.mean((y - ref64) ** 2))), rms_err32=float(np.sqrt(np.mean((ref32 - ref64) ** 2))))
        print(json.dumps(rec), flush=True)
        if not (err < 20 * err32 + 1e-5) and v < 6:
            bad = np.argwhere(np.abs(y - ref64) > 20 * err32 + 1e-5)
            print("  MISMATCH rows/cols (first 10):", bad[:10].tolist(), "n_bad", len(bad), flush=True)
            results.append(rec)
            continue
        # ---- timing: `nwg` workgroups (default: the chip full, `rounds` workgroups per CU), with / without global I/O
        for nwg, flags in [(n, f) for n in [int(t) for t in args.nwg.split(",")] for f in [int(t) for t in args.flags.split(',')]]:
            nwg = nwg if nwg > 0 else ncu * args.rounds
            T = 128 * nwg
            md = torch.randn((T, C), device=dev)
            xd = torch.randn((T, C), device=dev)
            yd = torch.empty((T, C), device=dev)
            tb = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)

            def run(tbuf=None):
                return lib.chain128_run(v, md.data_ptr(), xd.data_ptr(), yd.data_ptr(), ws.data_ptr(), gd.data_ptr(),
                                        bd.data_ptr(), T, tbuf, st, flags)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            times = []
            for _ in range(args.iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                e1.synchronize()
                times.append(e0.elapsed_time(e1) * 1e3)
            us = float(np.median(times))
            tf = T * FLOP_PER_TOKEN / us * 1e-6
            r2 = dict(variant=v, nwg=nwg, no_io=flags, us_median=round(us, 2), us_min=round(float(np.min(times)), 2),
                      tflops=round(tf, 1), frac_of_split_roof=round(tf / SPLIT_ROOF, 4))
            run(tb.data_ptr())
            torch.cuda.synchronize()
            t = tb.cpu().numpy().reshape(-1, 16)[:, :6].astype(np.float64)
            d = np.diff(t, axis=1)
            r2["phase_cycles_median"] = dict(zip(["prologue", "merge", "ln2", "mlp", "store"], np.median(d, axis=0).tolist()))
            r2["wg_cycles_median"] = float(np.median(t[:, 5] - t[:, 0]))
            print(json.dumps(r2), flush=True)
            rec.setdefault("timing", []).append(r2)
        results.append(rec)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(results, f, indent=1)


if __name__ == "__main__":
    sys.exit(main())
