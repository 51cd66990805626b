"""No GPU needed: the roofline denominators bench.py reports are the figures of SURVEY.md §8d."""
import importlib.util
import os

from tests import helpers as H


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(H.ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_match_survey():
    from prism_amd.config import baseline_config
    b = _bench()
    c3 = baseline_config(2)
    # c3: P = 201 430 parameters, no target network, cap2 = 131 072 (17 levels): 5.79 MB per step
    assert b.algorithmic_bytes(c3, 201_430, 0, 17) == 823_552 + 21_504 + 107_520 + 4_834_320 + 1_028
    c4 = baseline_config(3)
    # c4: B = 512, P = 1 544 210 with a target network: 45.15 MB per step
    got = b.algorithmic_bytes(c4, 1_544_210, 1_544_210, 17)
    assert got == 1_647_104 + 43_008 + 215_040 + 37_061_040 + 6_176_840 + 2_052


def test_gemm_flops_match_survey():
    from prism_amd.config import baseline_config
    b = _bench()
    fl = b.kernel_flops(baseline_config(2), 256)
    # forward: 4096 rows x (2*64*1024 + 2*1024*128 + 2*128*6); backward: 2048 rows x 2*1024*(64 + 128 + 128)
    # (dW of phi, dW and dX of the trunk; the phi columns the kernel recomputes are not algorithmic work)
    assert fl["fwd_tile_kernel"] == 4096 * (2 * 64 * 1024 + 2 * 1024 * 128 + 2 * 128 * 6)
    assert fl["iqn_bwd_kernel"] == 2048 * 2 * 1024 * 320
    assert abs(sum(fl.values()) / 1e9 - 2.97) < 0.02     # 2.97 GFLOP per c3 step
    assert b.FP32_MFMA_PEAK_TFLOPS == 157.3 and b.HBM_PEAK_GBS == 8000.0
