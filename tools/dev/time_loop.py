"""Development aid: us per step of the resident loop vs one launch per step at config 2 for a frame geometry.
Usage (GPU box): APS_TS_R=5 [APS_TS_OWN=..] python tools/dev/time_loop.py [extra -D flags]"""
import importlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd"
R = os.environ.get("APS_TS_R", "5")
extra = sys.argv[1:]
lib = f"/tmp/libaps_time_loop_{R}_{'_'.join(x.strip('-D') for x in extra)}.so"
if not os.path.exists(lib):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DAPS_DEV_RS=" + R, *extra,
                    "-I", os.path.join(ROOT, "include"), "-o", lib, os.path.join(ROOT, PKG, "csrc", "aps_hip.hip")], check=True)
capi = importlib.import_module(PKG + ".capi")
capi.LIB_PATH = lib
import bench
w = dict(bench.WORK)
if os.environ.get("DBG_FP32"): w["fp32"] = True
final = {}
for loop in (True, False):
    h = bench.make_handle(capi, w, method="tiles")
    h.set_resident_loop(loop)
    h.set_state(*bench.initial_state(w))
    h.step(201)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); h.step(2001); ts.append((time.perf_counter() - t0) / 2001 * 1e6)
    print("R", R, "OWN", os.environ.get("APS_TS_OWN"), "loop" if loop else "per-step", h.loop_info()[:2], "us/step", [round(x, 2) for x in ts])
    final[loop] = h.get_state() + h.get_lattice()
    h.close()
print("loop == per-step (state and lattice arrays):", all(np.array_equal(x, y) for x, y in zip(final[True], final[False])))
