"""ORACLE — test infrastructure only.

CPU restatements of the reference's replay-sample -> TD-update -> priority-writeback path.
Nothing under ``prism_amd`` may import this package; only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg do, and only as the checker.
"""
