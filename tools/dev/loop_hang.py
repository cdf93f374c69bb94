"""Development aid: which wait of the resident loop runs out at config 2 (APS_LOOP_DEBUG build in /tmp)."""
import ctypes as C, importlib, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd"
lib = "/tmp/libaps_loop_debug.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DAPS_LOOP_DEBUG",
                "-DAPS_DEV_RS=5", "-I", os.path.join(ROOT, "include"), "-o", lib, os.path.join(ROOT, PKG, "csrc", "aps_hip.hip")], check=True)
capi = importlib.import_module(PKG + ".capi")
capi.LIB_PATH = lib
import bench
os.environ["APS_LOOP_TIMEOUT_MS"] = "50"
w = dict(bench.WORK)
for n in [int(x) for x in os.environ.get("DBG_N", "11,21,41,81,201").split(",")]:
    h = bench.make_handle(capi, w, method="tiles")
    h.set_state(*bench.initial_state(w))
    t0 = time.perf_counter(); h.step(n); dt = time.perf_counter() - t0
    info = h.loop_info()
    print("n", n, info[:2], f"{dt / n * 1e6:.2f} us/step")
    if info[1] == -1:
        buf = np.zeros(8 * 4096, dtype=np.uint64)
        fn = h.lib.aps_debug_stamps; fn.restype = C.c_int
        fn(h._h, buf.ctypes.data_as(C.c_void_p), C.c_int64(len(buf)))
        st = buf.reshape(-1, 8)
        rows = np.flatnonzero(st[:, 0] == 0xDEAD)
        rec = int(st[rows[0], 7]) if len(rows) else 1
        print("  waits that ran out:", len(rows))
        for r in rows[:24]:
            off = int(st[r, 3]); buf_i, rest = divmod(off, 633 * rec); b, slot = divmod(rest, rec)
            print(f"   tile {r} it {int(st[r, 1])} wave {int(st[r, 2])} waits on buffer {buf_i} tile {b} slot {slot}: saw tag {int(st[r, 4]) >> 32} word {int(st[r, 4]) & 0xFFFFFFFF:#x}, wants {int(st[r, 5])}, missing lanes {int(st[r, 6]):#018x}")
    h.close()
