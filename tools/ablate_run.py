"""Timing attribution on the GPU with the -DOETR_ABLATE library
(tools/ablate.sh).  Results are WRONG numerically by construction; only the
per-kernel durations matter."""
import os, sys, json
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
os.environ['OETR_HIP_LIB'] = str(REPO / 'tools' / 'ablate' / 'liboetr_hip.so')
import torch
import imagematching_oetr_amd as pkg

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
PREC = os.environ.get('ABL_PREC', 'f32_split_f16')
eng = pkg.HotPathEngine(model.hot_path_state(), device=dev, precision=PREC, enc_tile=32)
print('precision', PREC)
n = 8
f1 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev)
f2 = (torch.rand(n, 256, 20, 20) - 0.5).to(dev)
pos = model.pos_encoding(f1.cpu()).contiguous().to(dev)
NAMES = {1: 'kvreduce->1tile', 2: 'no-gelu', 4: 'no-elu', 8: 'no-LN', 16: 'no-GEMM', 32: 'no-store', 64: 'no-xload', 128: 'no-wload', 256: 'no-attn', 512: 'no-kvstate'}
TILE = int(os.environ.get('ABL_TILE', 32))
eng.set_encoder_tile(TILE)
CUM = [(n + 1) << 16 for n in range(13)] if os.environ.get('ABL_CUM') else []
FLAGS = CUM or [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 32, 64, 128, 256, 512, 1 + 64, 16, 16 + 128, 128 + 1 + 64, 1023 - 16 - 128, 1023]
for flags in FLAGS:
    os.environ['OETR_ABLATE'] = str(flags)
    for _ in range(5):
        eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
    with pkg.KernelTrace(eng, max_launches=2048) as tr:
        for _ in range(30):
            eng.forward(f1, f2, pos, pos, (640, 640), (640, 640))
        torch.cuda.synchronize()
    s = tr.summary()
    label = '+'.join(v for k, v in NAMES.items() if flags & k) or 'baseline'
    if flags >> 16:
        label = f'exit at phase boundary {(flags >> 16) - 1}'
    print(f'{flags:3d} {label:45s} ' + ' '.join(f'{k}={v[1]/v[0]*1e3:.1f}' for k, v in s.items()
          if k.startswith('k_enc')), flush=True)
