"""Build libaps_hip.so (HIP kernels + C ABI) for gfx950, in-tree next to this file.

    python build.py [--force] [--save-temps]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract: the rate and
probability code is a fixed sequence of IEEE operations shared with the CPU oracle."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "aps_hip.hip")
SRCS = [SRC, os.path.join(HERE, "csrc", "pde_hip.hip"), os.path.join(HERE, "csrc", "gillespie_hip.hip"),
        os.path.join(HERE, "csrc", "gillespie_big_hip.hip")]   # stepper; PDE solver; exact event loop
HDR = os.path.join(ROOT, "include", "aps.h")
HDRS = [HDR, os.path.join(ROOT, "include", "pde.h"), os.path.join(HERE, "csrc", "aps_common.hpp"), os.path.join(HERE, "csrc", "tile_step.hpp"), os.path.join(HERE, "csrc", "tile_dense.hpp"), os.path.join(HERE, "csrc", "tile_loop.hpp"), os.path.join(HERE, "csrc", "ntt_conv.hpp"),
        os.path.join(ROOT, "include", "gillespie.h")]
LIB = os.path.join(HERE, "libaps_hip.so")
ARCH = "gfx950"


def build(force=False, save_temps=False, verbose=False):
    newest = max(os.path.getmtime(f) for f in SRCS + HDRS)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    cmd = ["hipcc", f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"), "-o", LIB] + SRCS
    if os.environ.get("APS_DEV_RS"):           # development: only one frame size of the tile kernels (builds in a fraction of the time)
        cmd += ["-DAPS_DEV_RS=" + os.environ["APS_DEV_RS"]] + (["-DAPS_LOOP_DEBUG"] if os.environ.get("APS_LOOP_DEBUG") else [])
    if save_temps:
        tmp = os.path.join(HERE, "csrc", "_temps")
        os.makedirs(tmp, exist_ok=True)
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=os.path.join(HERE, "csrc"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, verbose=True))
