"""Per-phase cycles of the FIRST encoder launch, k_encoder<A> (phase A only: input tile, LN-A, Q / K / V GEMMs,
KV state), beside the same phases inside a <B,A> launch - what the NCHW gather of the reference's layout costs
(VERDICT r5 item 1b).  Library built with -DOETR_PHASE_TIMING (tools/variants.sh timing "-DOETR_PHASE_TIMING").

    python tools/phase_timing_a.py [tile]        # tile = 32 (latency shape, default) or 64
"""
import ctypes, os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
os.environ['OETR_HIP_LIB'] = str(REPO / 'tools/variants/timing/liboetr_hip.so')
import numpy as np, torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 8
f1 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev); f2 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev)
pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
eng = pkg.HotPathEngine(model.hot_path_state(), device=dev, precision='f32_split_f16', enc_tile=tile)
lib = eng.lib
lib.oetr_debug_read_tbuf.argtypes = [ctypes.c_void_p, ctypes.c_int]
nb = 16 * (13 if tile == 32 else 7)


def stamps():
    buf = (ctypes.c_longlong * (16 * nb))()
    assert lib.oetr_debug_read_tbuf(buf, nb) == 0
    return np.frombuffer(buf, dtype=np.int64).reshape(nb, 16).astype(np.float64)


def show(title, t, cols):
    print(title)
    for name, a, b in cols:
        d = t[:, b] - t[:, a]
        print(f'  {name:44s} {d.mean():9.0f}  (min {d.min():9.0f} max {d.max():9.0f})')


for _ in range(3):
    eng.forward(f1, f2, pos, pos, (640, 640), (640, 640), stages=True, enc_layers=3)
torch.cuda.synchronize()
# the stamps of a slot are overwritten by every launch: after a 3-layer run the LAST launch is <B> (stamps 0..8), the one
# before it <B,A> - whose phase-A stamps 9..12 are still in place (stamp 8 = end of phase B, overwritten by <B>: skip it)
tBA = stamps()
# the first launch's own stamps: 13 (start) and 14 (tile in LDS, token-major copy stored) are written by <A> only;
# run ONE layer so that no <B,A> launch rewrites 9..12 (the <B> launch behind <A> writes 0..8 only)
eng.forward(f1, f2, pos, pos, (640, 640), (640, 640), stages=True, enc_layers=1)
torch.cuda.synchronize()
t1 = stamps()
print(f'tile {tile} rows, {nb} workgroups; cycles (s_memtime, 100 MHz-independent shader clock counter)')
show('<B,A> launch, phase A (from the LN-A start = stamp 9 ... relative):', tBA,
     [('Q GEMM', 9, 10), ('K GEMM + phi(Q) store, V GEMM', 10, 11), ('KV state + stores', 11, 12)])
show('<A> launch (first launch of a forward):', t1,
     [('kernel start -> tile in LDS + token-major store', 13, 14), ('LN-A', 14, 9), ('Q GEMM', 9, 10),
      ('K GEMM + phi(Q) store, V GEMM', 10, 11), ('KV state + stores', 11, 12), ('whole workgroup', 13, 12)])
# the same launch fed token-major (oetr_forward_tokens: what the HIP neck hands over) - no transpose on the way in
bufs = eng.token_buffers(n, 20, 20, 20, 20)
eng.load_pos_tokens(bufs, pos, pos)
tok = torch.cat([f1.flatten(2).permute(0, 2, 1).reshape(-1, 256), f2.flatten(2).permute(0, 2, 1).reshape(-1, 256)]).contiguous()
for _ in range(2):
    bufs['tokens'].copy_(tok)
    eng.forward_tokens(n, 20, 20, 20, 20, (640, 640), (640, 640))
torch.cuda.synchronize()
# (a full forward: 9..12 now belong to the last <B,A>; 13 / 14 to the first launch)
t2 = stamps()
show('<A> launch on token-major input (oetr_forward_tokens):', t2, [('kernel start -> tile in LDS', 13, 14)])
