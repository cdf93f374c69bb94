"""Diagnostic: build the library with -DAPS_STAMPS into /tmp and print per-phase cycle shares of pair_propose.
Usage (GPU box): python tools/stamps.py [WAVES]"""
import ctypes as C, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd"
waves = int(sys.argv[1]) if len(sys.argv) > 1 else 4
extra = sys.argv[2:]
lib = f"/tmp/libaps_stamps_{waves}_{'_'.join(x.strip('-D') for x in extra)}.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DAPS_STAMPS",
                f"-DAPS_WAVES={waves}", *extra, "-I", os.path.join(ROOT, "include"), "-o", lib,
                os.path.join(ROOT, PKG, "csrc", "aps_hip.hip")], check=True)
capi = importlib.import_module(PKG + ".capi")
capi.LIB_PATH = lib
import bench
w = dict(bench.WORK)
h = bench.make_handle(capi, w)
h.set_state(*bench.initial_state(w))
h.step(20)
ms, n, pairs = h.step_timed(20)
print(f"WAVES={waves} {extra}: pair kernel {ms/n*1e3:.1f} us/launch, pairs/launch {pairs/n:.3g}")
pn = np.zeros(1564, dtype=np.uint32)
h.lib.aps_debug_plan_n.restype = C.c_int
h.lib.aps_debug_plan_n(h._h, pn.ctypes.data_as(C.c_void_p), C.c_int64(len(pn)))
print("plan_n: min", pn.min(), "mean", pn.mean(), "max", pn.max(), "first", pn[:6], "last", pn[-4:])
buf = np.zeros(8 * 4096, dtype=np.uint64)
fn = h.lib.aps_debug_stamps
fn.restype = C.c_int
fn(h._h, buf.ctypes.data_as(C.c_void_p), C.c_int64(len(buf)))
st = buf.reshape(-1, 8)
st = st[st[:, 4] > 0]
tot = st[:, 4].astype(float)
print("workgroups", len(st), "items/WG mean", st[:, 3].mean(), "max", st[:, 3].max())
for k, name in enumerate(("fetch", "accumulate", "epilogue")):
    print(f"  {name:10s} mean {st[:, k].mean():10.0f} cyc  share {100 * st[:, k].sum() / tot.sum():5.1f}%  per item {st[:, k].sum() / st[:, 3].sum():8.0f}")
print(f"  prologue-of-accumulate (wait for target/plan loads): per item {st[:, 6].sum() / st[:, 3].sum():8.0f} cyc")
print(f"  total      mean {tot.mean():10.0f} cyc   min {tot.min():.0f} max {tot.max():.0f}")
rt = st[:, 5].astype(float)
print(f"  end-time spread (100MHz ticks): {rt.max() - rt.min():.0f}")
rs = st[:, 6].astype(float)
print(f"  start-time spread (100MHz ticks): {rs.max() - rs.min():.0f}; WGs starting > 1000 ticks after the first: {(rs - rs.min() > 1000).sum()}")
h.close()
