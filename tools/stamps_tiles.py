"""Diagnostic: build the library with -DAPS_STAMPS into /tmp and print per-phase cycles of tile_step.
Usage (GPU box): python tools/stamps_lattice.py [-DAPS_...]"""
import ctypes as C, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd"
workload = "config2"
extra = sys.argv[1:]
if extra and not extra[0].startswith("-"):
    workload, extra = extra[0], extra[1:]
lib = f"/tmp/libaps_stamps_tiles_{'_'.join(x.strip('-D') for x in extra)}.so"
if os.environ.get("APS_LIB"):                      # a stamps build made beforehand (tools/build_variant.sh ... -DAPS_STAMPS)
    lib = os.environ["APS_LIB"]
else:
  subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DAPS_STAMPS",
                *extra, "-I", os.path.join(ROOT, "include"), "-o", lib, os.path.join(ROOT, PKG, "csrc", "aps_hip.hip")], check=True)
capi = importlib.import_module(PKG + ".capi")
capi.LIB_PATH = lib
import bench
w = dict(bench.WORK) if workload == "config2" else dict(bench.EXTRA[workload])
h = bench.make_handle(capi, w, method="tiles")
for e in range(len(w.get("betas", [0]))):
    h.set_state(*bench.initial_state(w), ensemble=e)
h.step(200)
prof = h.step_profile(20)
print({k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items() if v[1]}, "us per launch (event-bracketed)")
buf = np.zeros(8 * 4096, dtype=np.uint64)
fn = h.lib.aps_debug_stamps
fn.restype = C.c_int
fn(h._h, buf.ctypes.data_as(C.c_void_p), C.c_int64(len(buf)))
st = buf.reshape(-1, 8)
idx = np.flatnonzero(st[:, 4] > 0)
st = st[idx].astype(float)
print("workgroups", len(st))
for k, name in (1, "prologue+stage"), (2, "copy lists"), (3, "sweep"), (7, "reduce+propose"), (0, "resolve+store"), (4, "total"):
    print(f"  {name:12s} mean {st[:, k].mean():9.0f} cyc   median {np.median(st[:, k]):9.0f}   max {st[:, k].max():9.0f}")
rs, re = st[:, 6], st[:, 5]
print(f"  start spread {rs.max() - rs.min():.0f} ticks(10ns)  end spread {re.max() - re.min():.0f}  first start -> last end {re.max() - rs.min():.0f}")
print(f"  WG lifetime in ticks: mean {(re - rs).mean():.0f} max {(re - rs).max():.0f}")
life = re - rs
order = np.argsort(-life)[:12]
print("  slowest WGs (blockIdx, lifetime ticks, start-first, phases resolve/stage/copy/sweep/propose cycles):")
for o in order:
    print(f"    {idx[o]:5d} {life[o]:6.0f} {rs[o] - rs.min():5.0f}  {st[o, 0]:7.0f} {st[o, 1]:7.0f} {st[o, 2]:7.0f} {st[o, 3]:7.0f}  {st[o, 7]:4.0f}")
print("  lifetime percentiles (ticks): ", {q: float(np.percentile(life, q)) for q in (5, 25, 50, 75, 90, 95, 99)})
h.close()
