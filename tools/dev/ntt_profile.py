"""Development aid (GPU box): per-launch times of the tile kernel and of the convolution's launches at config 5."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
capi = importlib.import_module(bench.PKG + ".capi")
w = dict(bench.EXTRA["config5"])
h = bench.make_handle(capi, w, method="tiles")
h.set_state(*bench.initial_state(w))
h.step(64)
t0 = time.perf_counter(); h.step(128); print("us/step", (time.perf_counter() - t0) / 128 * 1e6)
prof = h.step_profile(20)
print({k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items() if v[1]}, "us per launch")
info = h.ntt_info()
print(info, "ntt us per launch", info["prof_ms"] / max(info["prof_launches"], 1) * 1e3, "per step", info["prof_ms"] / 20 * 1e3)
h.close()
