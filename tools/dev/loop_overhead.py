import os, sys, importlib, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
capi = importlib.import_module(bench.PKG + ".capi")
w = dict(bench.WORK)
h = bench.make_handle(capi, w, method="tiles")
h.set_state(*bench.initial_state(w))
h.step(201)
for n in (10, 11, 20, 21, 40, 41, 80, 160, 320, 640):
    ms = [h.step_loop_timed(n)[0] for _ in range(7)]
    print(n, "steps: kernel us", round(float(np.median(ms)) * 1e3, 2), "per step", round(float(np.median(ms)) * 1e3 / n, 3), flush=True)
import time
for n in (20, 200):
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); h.step(n); ts.append((time.perf_counter() - t0) * 1e6)
    print(n, "steps: wall us (incl. sync)", round(float(np.median(ts)), 1), "per step", round(float(np.median(ts)) / n, 3))
h.close()
