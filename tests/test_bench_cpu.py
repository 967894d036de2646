"""bench.py's host-side helpers on a box without a GPU: nothing here may need one, and the optional extras
(power samples) must degrade to None instead of breaking the one JSON line."""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))


def test_power_sample_degrades_without_a_gpu():
    import bench
    calls = []
    t0 = time.time()
    out = bench.power_sample(lambda k: (calls.append(k), time.sleep(0.02)), seconds=0.5)
    assert out is None or {'socket_w', 'cap_w', 'sclk_mhz'} <= set(out)     # None here (no device behind rocm-smi)
    assert calls and time.time() - t0 < 60


def test_roofline_block_carries_both_yardsticks():
    """Nominal peak (MI355X_MICROARCH.md) and - when the run calibrated it (oetr_debug_mfma_rate) - the
    dense-MFMA rate the chip sustains at its power cap; the launch time is the HIP-event bracket minus the
    empty bracket measured in the same run, with the raw figure stated beside it."""
    import bench
    kern = {bench.DOMINANT: (7, 7 * 0.045)}            # 7 launches of 45 us
    saved = dict(bench.CALIB)
    try:
        bench.CALIB.update(f16_sustained=None, event_pair_us=0.0)
        rb = bench.roofline_block(kern, 'f32_split_f16', 6400, 64, 1, 0.5e-3, False, grids=(8, 400, 400))
        assert rb['peak'] == round(bench.F16_MFMA_PEAK_TFLOPS / 3, 1)
        assert 'sustained_peak_measured' not in rb and rb['frac'] == rb['frac_events_raw']   # nothing calibrated: nothing claimed
        bench.CALIB.update(f16_sustained=1600.0, event_pair_us=4.0)
        rb = bench.roofline_block(kern, 'f32_split_f16', 6400, 64, 1, 0.5e-3, False, grids=(8, 400, 400))
        assert rb['sustained_peak_measured'] == round(1600.0 / 3, 1)
        assert abs(rb['frac_of_sustained_peak'] - rb['achieved'] / (1600.0 / 3)) < 1e-3
        assert rb['avg_launch_us'] == 42.0 and rb['avg_launch_us_events_raw'] == 45.0 and rb['event_bracket_overhead_us'] == 3.0 and rb['event_pair_us'] == 4.0
        assert abs(rb['frac'] / rb['frac_events_raw'] - 45.0 / 42.0) < 1e-3
        assert rb['workgroups_per_launch'] == 112 and rb['traffic'] is None
        rb32 = bench.roofline_block(kern, 'f32', 6400, 32, 1, 0.5e-3, False)
        assert 'sustained_peak_measured' not in rb32        # (measured for the f16 pipe only)
    finally:
        bench.CALIB.clear()
        bench.CALIB.update(saved)
