"""CPU-only checks of the C-ABI library: it loads, exports every symbol the
header declares, and host-side argument validation works without a GPU.
No compute calls here."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

import imagematching_oetr_amd as pkg
from imagematching_oetr_amd import hip_engine

REPO = Path(__file__).resolve().parents[1]


def header_functions():
    text = (REPO / 'include' / 'oetr_hip.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(oetr_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = pkg.load_library()
    names = header_functions()
    assert len(names) == 53, names
    assert set(names) == set(hip_engine.EXPORTS)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/oetr_hip.h but not exported'
    assert lib.oetr_abi_version() == hip_engine.ABI_VERSION


def test_struct_layout_matches_header_sizes():
    # 8 encoder layers x 12 ptrs, 2 decoder layers x (7+7+8) ptrs, 11 tail ptrs
    n_ptr = 8 * 12 + 2 * 22 + 11
    assert ctypes.sizeof(hip_engine._Weights) == 8 + 8 * n_ptr
    assert ctypes.sizeof(hip_engine._Stages) == 8 + 8 * 10
    # neck: 4 + 3 + 3 + 2 pointers
    assert ctypes.sizeof(hip_engine._NeckWeights) == 8 + 8 * 12


def test_workspace_query_and_shape_errors_need_no_gpu():
    lib = pkg.load_library()
    b = lib.oetr_workspace_bytes(None, 8, 20, 20, 20, 20)
    assert b > 2 * 8 * 2 * 400 * 256 * 4          # at least x and phi(Q)
    assert b % 256 == 0
    assert lib.oetr_workspace_bytes(None, 8, 101, 100, 20, 20) == 0   # > 100x100 tokens
    assert b'invalid shape' in lib.oetr_last_error()
    assert lib.oetr_workspace_bytes(None, 0, 20, 20, 20, 20) == 0
    # ragged grids grow monotonically
    assert lib.oetr_workspace_bytes(None, 8, 20, 20, 40, 40) > b


def test_neck_workspace_query_and_errors_need_no_gpu():
    lib = pkg.load_library()
    b = lib.oetr_neck_workspace_bytes(None, 16, 40, 40)
    # X planes (2 x f16) + 21 partial slabs
    assert b >= 2 * (16 * 1600 + 1) * 256 * 2 + 16 * 400 * (256 + 4 * 128 + 16 * 128) * 4
    assert b % 256 == 0
    assert lib.oetr_neck_workspace_bytes(None, 1, 1, 40) == 0      # no output row
    assert lib.oetr_neck_workspace_bytes(None, 1, 201, 202) == 0   # > 100x100 tokens out
    assert lib.oetr_neck_workspace_bytes(None, 0, 40, 40) == 0
    st = lib.oetr_neck_forward(None, None, 1, 40, 40, None, 0, None, None)
    assert st == 1 and b'NULL' in lib.oetr_last_error()
    h = ctypes.c_void_p()
    w = hip_engine._NeckWeights()
    w.struct_size = 4
    assert lib.oetr_neck_create(ctypes.byref(w), 0, ctypes.byref(h)) == 1
    assert b'size/ABI' in lib.oetr_last_error()
    w.struct_size = ctypes.sizeof(hip_engine._NeckWeights)
    w.abi_version = hip_engine.ABI_VERSION
    assert lib.oetr_neck_create(ctypes.byref(w), 0, ctypes.byref(h)) == 1
    assert b'is NULL' in lib.oetr_last_error()


def test_null_arguments_are_rejected_not_crashed():
    lib = pkg.load_library()
    assert lib.oetr_set_encoder_tile(None, 64) == 1 and b'NULL' in lib.oetr_last_error()
    st = lib.oetr_forward(None, None, None, None, None, 1, 20, 20, 20, 20, 640,
                          640, 640, 640, None, 0, None, None, None)
    assert st == 1 and b'NULL' in lib.oetr_last_error()
    h = ctypes.c_void_p()
    assert lib.oetr_create(None, 0, 0, ctypes.byref(h)) == 1
    w = hip_engine._Weights()
    w.struct_size = 4                               # wrong size
    assert lib.oetr_create(ctypes.byref(w), 0, 0, ctypes.byref(h)) == 1
    assert lib.oetr_box_tlbr_to_xyxy(None, None, 0, 1, 1, None, None) == 1
    # the masked entries (ABI 4) validate like the ones they extend
    st = lib.oetr_forward_masked(None, None, None, None, None, None, None, 1, 20, 20, 20, 20, 640,
                                 640, 640, 640, None, 0, None, None, None, None)
    assert st == 1 and b'NULL' in lib.oetr_last_error()
    assert lib.oetr_feature_correlation_masked(None, None, None, None, None, None, None, 1, 20, 20, 20, 20,
                                               None, 0, None, None, None, None, None) == 1
    assert lib.oetr_center_estimation_masked(None, None, None, None, None, None, None, 1, 20, 20, 20, 20,
                                             640, 640, None, 0, None, None, None) == 1
    assert lib.oetr_linear_attention_masked(None, None, None, None, None, 1, 4, 4, None, None, 0, None) == 1
    # the flag-slot entries (ABI 6): a NULL slot is refused before anything else is looked at
    st = lib.oetr_forward_flagslot(None, None, None, None, None, None, None, 1, 20, 20, 20, 20, 640,
                                   640, 640, 640, None, 0, None, None, None, None)
    assert st == 1 and b'flag_slot' in lib.oetr_last_error()
    st = lib.oetr_forward_tokens_flagslot(None, 1, 20, 20, 20, 20, 640, 640, 640, 640, None, 0, None, None,
                                          None, None)
    assert st == 1 and b'flag_slot' in lib.oetr_last_error()
    assert lib.oetr_neck_forward_tokens_status(None, None, 1, 40, 40, None, 0, None, None, None) == 1
    assert b'status_word' in lib.oetr_last_error()
    out = ctypes.c_void_p()
    assert lib.oetr_flagslot_device_pointer(None, ctypes.byref(out)) == 1
    assert lib.oetr_debug_mfma_rate(0, 1.0, None, None) == 1


def test_product_path_has_no_cpu_fallback():
    """The drop-in module must refuse to run the hot path off-GPU."""
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    img = torch.rand(1, 64, 64, 3)
    with pytest.raises(RuntimeError, match='GPU'):
        model.forward_dummy(img, img)
    with pytest.raises(ValueError, match='both mask1 and mask2'):      # masks: both or neither
        model.forward_dummy(img, img, mask1=torch.ones(1, 2, 2))
    with pytest.raises(RuntimeError, match='GPU'):                      # ... and no CPU route with them either
        model.forward_dummy(img, img, mask1=torch.ones(1, 2, 2), mask2=torch.ones(1, 2, 2))
    model.hip_precision = 'bf16'
    with pytest.raises(NotImplementedError, match='masks'):             # built for the default arithmetic
        model.forward_dummy(img, img, mask1=torch.ones(1, 2, 2), mask2=torch.ones(1, 2, 2))
    model.hip_precision = 'f32_split_f16'
    with torch.no_grad(), pytest.raises(KeyError):   # training forward: needs the reference's data keys
        model({'image1': img})
    with pytest.raises(ValueError):
        cfg = pkg.get_cfg_defaults().OETR
        cfg.MODEL = 'oetr_fc'
        pkg.build_detectors(cfg)
    with pytest.raises(hip_engine.OetrError, match='GPU'):
        hip_engine.box_tlbr_to_xyxy(torch.zeros(1, 2), torch.zeros(1, 4), 4, 4)


def test_product_code_never_imports_the_oracle():
    for f in (REPO / 'imagematching_oetr_amd').rglob('*.py'):
        assert 'oracle' not in f.read_text().replace('no oracle', ''), f


def test_host_feature_extraction_shapes():
    """Host (torch) side: stride-32 token grid, 256 channels, batch-broadcast
    position window (reference src/model.py:109-130)."""
    model = pkg.OETR(pkg.get_cfg_defaults().OETR).eval()
    with torch.no_grad():
        f1, f2, p1, p2, hf1, wf1, hf2, wf2 = model.feature_extraction(
            torch.rand(1, 128, 160, 3), torch.rand(1, 96, 64, 3))
    assert f1.shape == (1, 256, 4, 5) and f2.shape == (1, 256, 3, 2)
    assert p1.shape == (1, 256, 4, 5) and (hf1, wf1, hf2, wf2) == (4, 5, 3, 2)
