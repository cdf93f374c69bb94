"""Diagnostic: build the library with -DAPS_LOOP_STAMPS into /tmp and print where a workgroup of the resident loop
(csrc/tile_loop.hpp) spends an iteration.  Usage (GPU box): python tools/stamps_loop.py [steps] [-DAPS_...]"""
import ctypes as C, importlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd"
extra = sys.argv[1:]
steps = 201
if extra and not extra[0].startswith("-"):
    steps, extra = int(extra[0]) | 1, extra[1:]
lib = f"/tmp/libaps_stamps_loop_{'_'.join(x.strip('-D') for x in extra)}.so"
if os.environ.get("APS_LIB"):                      # a stamps build made beforehand (tools/build_variant.sh ... -DAPS_LOOP_STAMPS -DAPS_DEV_RS=7)
    lib = os.environ["APS_LIB"]
else:
  subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DAPS_LOOP_STAMPS",
                "-DAPS_DEV_RS=" + os.environ.get("APS_TS_R", "5"), *extra, "-I", os.path.join(ROOT, "include"), "-o", lib,
                os.path.join(ROOT, PKG, "csrc", "aps_hip.hip")], check=True)
capi = importlib.import_module(PKG + ".capi")
capi.LIB_PATH = lib
import bench
w = dict(bench.WORK)
h = bench.make_handle(capi, w, method="tiles")
h.set_state(*bench.initial_state(w))
h.step(201)
t0 = time.perf_counter(); h.step(steps); dt = time.perf_counter() - t0
print("loop_info", h.loop_info(), f"{dt / steps * 1e6:.2f} us per step (wall, stamps build)")
buf = np.zeros(8 * 4096, dtype=np.uint64)
fn = h.lib.aps_debug_stamps
fn.restype = C.c_int
fn(h._h, buf.ctypes.data_as(C.c_void_p), C.c_int64(len(buf)))
allw = buf.reshape(-1, 4, 16).astype(float) / steps          # [tile][wave][16]
idxt = np.flatnonzero(allw[:, 0, 6] > 0)
allw = allw[idxt]
NAMES = ((2, "barrier F + random numbers"), (0, "wait for the records"), (8, "intake (pooling)"), (9, "barrier S"), (10, "sweep"), (11, "ds_add into the field"),
         (7, "barrier B"), (3, "proposals (D)"), (4, "exclusion (E)"), (5, "hand over + ask ahead"), (6, "total"))
print("workgroups", len(allw), "; cycles per iteration (s_memtime), per wave: mean over tiles [wave 0, 1, 2, 3]")
for k, name in NAMES:
    print(f"  {name:28s} " + "  ".join(f"{allw[:, wv, k].mean():9.1f}" for wv in range(4)) + f"   | max over waves, mean over tiles {allw[:, :, k].max(axis=1).mean():9.1f}")
st = np.zeros((len(allw), 8))
st[:, :8] = allw[:, 0, :8]
idx = idxt
for k, name in ((1, "intake, barrier S, sweep (wave 0)"),):
    print(f"  {name:24s} mean {st[:, k].mean():9.1f}   median {np.median(st[:, k]):9.1f}   min {st[:, k].min():9.1f}   max {st[:, k].max():9.1f}")
worst = np.argsort(-st[:, 1])[:8]
print("  longest intake .. sweep: tiles", idx[worst], st[worst, 1].round(0), " their waits", st[worst, 0].round(0))
best = np.argsort(st[:, 0])[:8]
print("  shortest waits: tiles", idx[best], st[best, 0].round(0), "their intake+sweep", st[best, 1].round(0))
h.close()
