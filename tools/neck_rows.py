"""PatchMerging conv A/B: kernel (gather / row window) x rows per workgroup (256 / 192 / 128 / auto)."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import torch
import imagematching_oetr_amd as pkg
torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
for n, hb in ((16, 40), (64, 40), (16, 64), (2, 80)):
    bb = torch.relu(torch.randn(n, 1024, hb, hb, device=dev))
    eng = pkg.NeckEngine({k: v for k, v in model.state_dict().items() if k in pkg.neck_keys()}, device=dev)
    refs = {}
    line = f'n={n} {hb}x{hb}:'
    for kind, rows in [(k, r) for _ in range(2) for k in ('gather', 'row_window', 'row_window_1w') for r in (256, 192, 128, 0)]:
        if kind != 'gather' and rows == 256:
            continue
        eng.set_conv_kernel(kind)
        eng.set_conv_rows(rows)
        for _ in range(3):
            out = eng.forward(bb)
        with pkg.KernelTrace(eng) as tr:
            for _ in range(10):
                eng.forward(bb)
            torch.cuda.synchronize()
        ks = {k: v[1] / v[0] * 1e3 for k, v in tr.summary().items()}
        ref = refs.setdefault(kind, out.clone())
        assert torch.equal(out, ref), 'rows-per-workgroup changed the result'
        err = (out - refs['gather']).abs().max().item() / refs['gather'].abs().max().item()
        line += f'  {kind[:2] + kind[-2:] if kind.endswith("1w") else kind[:2]}/{rows or "auto"}: {ks["k_neck_conv"]:.1f} us (d={err:.1e})'
    print(line, flush=True)
