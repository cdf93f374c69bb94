"""Diagnostic: does a short resident-loop launch run slower per step because of what precedes it (idle GPU, clocks) or because of its own first steps?"""
import os, sys, importlib, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
capi = importlib.import_module(bench.PKG + ".capi")
w = dict(bench.WORK)
h = bench.make_handle(capi, w, method="tiles")
h.set_state(*bench.initial_state(w))
h.step(201)
def timed(n, reps):
    t0 = time.perf_counter()
    for _ in range(reps): h.step(n)                            # (aps_step waits for its launch)
    return (time.perf_counter() - t0) / reps * 1e6
for n, reps in ((20, 1), (20, 10), (20, 100), (640, 1), (20, 100)):
    print(n, "steps x", reps, "calls:", round(timed(n, reps), 1), "us per call", round(timed(n, reps) / n, 3), "per step")
# the same 20-step kernel right after a long one (GPU busy and clocked up)
h.step(5000)
ms = [h.step_loop_timed(20)[0] for _ in range(5)]
print("20-step kernels right after 5000 steps:", [round(m * 1e3, 1) for m in ms])
time.sleep(0.5)
ms = [h.step_loop_timed(20)[0] for _ in range(5)]
print("20-step kernels after 0.5 s idle:", [round(m * 1e3, 1) for m in ms])
h.close()
