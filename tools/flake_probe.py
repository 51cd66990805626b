"""Diagnostic: repeat one parametrisation of tests/test_gpu_step.py to look for run-to-run differences."""
import sys, traceback
sys.path.insert(0, ".")
from tests.test_gpu_step import test_fused_and_graph_equal_unfused as t
cases = [dict(base=3, target_update_period=3), dict(), dict(base=1, use_layer_norm=True, use_double_q_learning=True, use_target_network=True)]
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for c in cases:
        try:
            t(dict(c))
        except AssertionError as e:
            bad += 1
            print("FAIL rep", rep, c, str(e)[:600])
print("failures:", bad)
