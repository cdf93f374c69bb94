"""usage: python tools/time_call_overhead.py  -- wall time of aps_step(n) calls at config 2 as a function of n (diagnostic:
the fixed cost of a call = launch + completion latency, and the per-step slope)"""
import sys, time, importlib
sys.path.insert(0, ".")
import numpy as np
import bench
capi = importlib.import_module(bench.PKG + ".capi")
w = dict(bench.WORK)
h = bench.make_handle(capi, w, method="tiles")
h.set_state(*bench.initial_state(w))
h.step(200)
for n in (1, 2, 4, 8, 16, 20, 32, 64, 128, 256):
    h.step(n)                                             # captures the exact-count graph if there is one
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); h.step(n); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    print(f"n={n:4d}  median {np.median(ts):8.1f} us  min {ts.min():8.1f}  per step {np.median(ts) / n:6.2f}  info {h.step_info()}")
h.close()
