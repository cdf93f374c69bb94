"""ctypes binding of libaps_hip.so (the C ABI declared in include/aps.h).

There is no CPU fallback: if the shared library is missing or no GPU is present the constructor of
`Handle` raises.  Loading the library itself works without a GPU (used by the CPU test that checks
the exported symbols against the header)."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("APS_LIB") or os.path.join(HERE, "libaps_hip.so")     # APS_LIB: a tuning build (tools/build_variant.sh)
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "aps.h")

APS_OK, APS_ERR_ARG, APS_ERR_HIP, APS_ERR_STATE, APS_ERR_NODEVICE = 0, -1, -2, -3, -4
METHODS = {"auto": 0, "pairs": 1, "lattice": 2, "tiles": 3}          # APS_METHOD_* of include/aps.h
KERNELS = ("pair_accumulate", "propose", "claim", "apply", "plan_tiles", "propose_lattice", "field_update", "tile_step")


class ApsError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libaps_hip error {code}: {text}")
        self.code = code


class ApsParams(C.Structure):
    """struct aps_params of include/aps.h, field for field."""
    _fields_ = [
        ("L", C.c_int32), ("K", C.c_int32), ("periodic", C.c_int32), ("minus_anchor", C.c_int32),
        ("immobilize", C.c_int32), ("suppress_flip", C.c_int32), ("crowding", C.c_int32),
        ("n_ensembles", C.c_int32), ("n_particles", C.c_int64), ("sigma_grid", C.c_double),
        ("rate_diffusion", C.c_double), ("rate_active", C.c_double), ("k_on", C.c_double),
        ("k_off", C.c_double), ("k_exit", C.c_double), ("dt", C.c_double), ("seed", C.c_uint64),
        ("beta", C.POINTER(C.c_double)), ("anchor_mask", C.POINTER(C.c_uint8)), ("device", C.c_int32),
        ("rank", C.c_int32), ("world", C.c_int32), ("sort_by_site", C.c_int32),
        ("ensemble_base", C.c_int32), ("method", C.c_int32), ("fp32", C.c_int32), ("halo_interval", C.c_int32),
    ]


_lib = None


def header_symbols():
    """Function names declared in include/aps.h."""
    with open(HEADER_PATH) as fh:
        text = fh.read()
    return sorted(set(re.findall(r"\b(aps_[a-z_0-9]+)\s*\(", text)))


def _share_hip_runtime_with_torch():
    """One HIP/HSA runtime per process.  The PyTorch wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7, the name libaps_hip.so links against).  If our library pulled in the system copy first,
    a later `import torch` would start a SECOND runtime and find no GPUs.  Loading torch's copy first makes the
    dynamic linker resolve our NEEDED entry to it, whatever the import order (torch.distributed is the
    multi-GPU plumbing).  APS_SYSTEM_HIP_RUNTIME=1 opts out (never combine that with torch)."""
    if "torch" in sys.modules or os.environ.get("APS_SYSTEM_HIP_RUNTIME") == "1":
        return None
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    return C.CDLL(cand, mode=C.RTLD_GLOBAL) if os.path.exists(cand) else None


def load():
    """dlopen the library (no GPU needed for this) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python {os.path.join(HERE, 'build.py')}` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    P = C.POINTER
    protos = {
        "aps_device_count": (C.c_int, []),
        "aps_last_error": (C.c_char_p, [vp]),
        "aps_create": (C.c_int, [P(ApsParams), P(vp)]),
        "aps_destroy": (None, [vp]),
        "aps_set_stream": (C.c_int, [vp, vp]),
        "aps_set_state": (C.c_int, [vp, i32, vp, vp, vp, vp, i64]),
        "aps_get_state": (C.c_int, [vp, i32, vp, vp, vp, vp, i64]),
        "aps_pair_accumulate": (C.c_int, [vp, i32, vp, vp, vp, i64]),
        "aps_step": (C.c_int, [vp, i64]),
        "aps_propose": (C.c_int, [vp]),
        "aps_commit": (C.c_int, [vp]),
        "aps_exchange_buffer": (C.c_int, [vp, P(vp), P(i64), P(i64), P(i64)]),
        "aps_bind_exchange_buffer": (C.c_int, [vp, vp, i64]),
        "aps_observe": (C.c_int, [vp, i32, vp, vp, vp]),
        "aps_field_from_counts": (C.c_int, [vp, i32, vp, vp, vp]),
        "aps_time": (C.c_int, [vp, P(dbl), P(i64)]),
        "aps_get_exits": (C.c_int, [vp, i32, vp, i64, P(i64)]),
        "aps_get_table": (C.c_int, [vp, vp, i32, P(i32), P(i32)]),
        "aps_resort": (C.c_int, [vp]),
        "aps_step_timed": (C.c_int, [vp, i64, P(dbl), P(i64), P(dbl)]),
        "aps_step_profile": (C.c_int, [vp, i64, vp, vp]),
        "aps_step_info": (C.c_int, [vp, P(i64), P(i64)]),
        "aps_set_resident_loop": (C.c_int, [vp, i32]),
        "aps_step_loop_timed": (C.c_int, [vp, i64, P(dbl), P(i64)]),
        "aps_loop_info": (C.c_int, [vp, P(i64), P(i32), C.c_char_p, i32]),
        "aps_copy_bandwidth": (C.c_int, [vp, i64, i32, P(dbl)]),
        "aps_lattice_accumulate": (C.c_int, [vp, i32, vp, vp, vp, i64]),
        "aps_get_lattice": (C.c_int, [vp, i32, vp, vp, vp]),
        "aps_method": (C.c_int, [vp]),
        "aps_mark_reference": (C.c_int, [vp, i32]),
        "aps_observe_scalars": (C.c_int, [vp, i32, i32, i32, i32, vp, vp]),
        "aps_observe_scalars_all": (C.c_int, [vp, i32, vp, vp, vp]),
        "aps_event_overhead": (C.c_int, [vp, i32, P(dbl)]),
        "aps_observe_structure": (C.c_int, [vp, i32, i32, vp]),
        "aps_observe_bins": (C.c_int, [vp, i32, i32, vp, vp]),
        "aps_rates_from_field": (C.c_int, [vp, i32, vp, vp, vp, i64, vp, vp, vp, vp]),
        "aps_comm_unique_id": (C.c_int, [vp]),
        "aps_comm_init": (C.c_int, [vp, vp]),
        "aps_comm_ranks": (C.c_int, [vp, P(i32)]),
        "aps_owned_sites": (C.c_int, [vp, P(i32), P(i32)]),
        "aps_halo_copy": (C.c_int, [vp, vp]),
        "aps_halo_pack": (C.c_int, [vp, i32, vp, i64, P(i64)]),
        "aps_halo_unpack": (C.c_int, [vp, i32, vp, i64]),
        "aps_halo_info": (C.c_int, [vp, P(i32), P(i32), P(i32)]),
        "aps_comm_selftest": (C.c_int, [vp, i64]),
        "aps_halo_sizes": (C.c_int, [vp, P(i64), P(i64)]),
        "aps_ipc_export": (C.c_int, [vp, vp]),
        "aps_ipc_connect": (C.c_int, [vp, vp, vp]),
        "aps_exchange_kind": (C.c_int, [vp]),
        "aps_set_flip_table": (C.c_int, [vp, vp, i32]),
        "aps_ntt_info": (C.c_int, [vp, P(i32), P(i32), P(dbl), P(i64)]),
        "aps_ntt_launches": (C.c_int, [vp]),
        "aps_tiles_info": (C.c_int, [vp, P(i32), P(i32), P(i32), P(i32)]),
    }
    lenient = os.environ.get("APS_LIB_LENIENT") == "1"     # tuning tools that load an older build of the library for A/B timing
    for name, (res, args) in protos.items():
        if lenient and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib._aps_protos = protos
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Handle:
    """One aps_handle: E ensembles of n particles on one GPU (optionally one shard of a multi-GPU job)."""

    def __init__(self, *, L, K, periodic, sigma_grid, rate_diffusion, rate_active, beta, dt, seed,
                 n_particles, minus_anchor=True, immobilize=True, suppress_flip=True, crowding=False,
                 k_on=0.0, k_off=0.0, k_exit=0.0, anchor_mask=None, device=0, rank=0, world=1,
                 sort_by_site=True, ensemble_base=0, method="auto", fp32=False, halo_interval=0):
        self.lib = load()
        self._h = C.c_void_p()
        betas = np.atleast_1d(np.asarray(beta, dtype=np.float64)).copy()
        self.E, self.n, self.L, self.K = len(betas), int(n_particles), int(L), int(K)
        self.dt = float(dt)
        mask = None if anchor_mask is None else np.ascontiguousarray(anchor_mask, dtype=np.uint8)
        if mask is not None and not mask.any():
            mask = None
        par = ApsParams(L=L, K=K, periodic=int(bool(periodic)), minus_anchor=int(bool(minus_anchor)),
                        immobilize=int(bool(immobilize)), suppress_flip=int(bool(suppress_flip)),
                        crowding=int(bool(crowding)), n_ensembles=self.E, n_particles=self.n,
                        sigma_grid=float(sigma_grid), rate_diffusion=float(rate_diffusion),
                        rate_active=float(rate_active), k_on=float(k_on), k_off=float(k_off),
                        k_exit=float(k_exit), dt=self.dt, seed=int(seed) & (2 ** 64 - 1),
                        beta=betas.ctypes.data_as(C.POINTER(C.c_double)),
                        anchor_mask=None if mask is None else mask.ctypes.data_as(C.POINTER(C.c_uint8)),
                        device=int(device), rank=int(rank), world=int(world),
                        sort_by_site=int(bool(sort_by_site)), ensemble_base=int(ensemble_base), method=METHODS[method],
                        fp32=int(bool(fp32)), halo_interval=int(halo_interval))
        rc = self.lib.aps_create(C.byref(par), C.byref(self._h))
        if rc != APS_OK:
            raise ApsError(rc, self.lib.aps_last_error(None).decode())
        self.method = {1: "pairs", 2: "lattice", 3: "tiles"}[self.lib.aps_method(self._h)]

    # -- plumbing
    def _ck(self, rc):
        if rc != APS_OK:
            raise ApsError(rc, self.lib.aps_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.aps_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, raw_stream):
        self._ck(self.lib.aps_set_stream(self._h, C.c_void_p(raw_stream)))

    # -- state
    def set_state(self, pos, sigma, bound=None, alive=None, ensemble=0):
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        sigma = np.ascontiguousarray(sigma, dtype=np.int8)
        bound = None if bound is None else np.ascontiguousarray(bound, dtype=np.uint8)
        alive = None if alive is None else np.ascontiguousarray(alive, dtype=np.uint8)
        self._ck(self.lib.aps_set_state(self._h, ensemble, _ptr(pos), _ptr(sigma), _ptr(bound), _ptr(alive), len(pos)))
        self._n_set = getattr(self, "_n_set", {})
        self._n_set[ensemble] = len(pos)

    def get_state(self, ensemble=0):
        n = self._n_set[ensemble]
        pos, sigma = np.zeros(n, np.int32), np.zeros(n, np.int8)
        bound, alive = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        self._ck(self.lib.aps_get_state(self._h, ensemble, _ptr(pos), _ptr(sigma), _ptr(bound), _ptr(alive), n))
        return pos, sigma, bound, alive

    def pair_accumulate(self, ensemble=0):
        n = self._n_set[ensemble]
        S, W, occ4 = np.zeros(n), np.zeros(n), np.zeros((n, 4), np.int32)
        self._ck(self.lib.aps_pair_accumulate(self._h, ensemble, _ptr(S), _ptr(W), _ptr(occ4), n))
        return S, W, occ4

    def lattice_accumulate(self, ensemble=0):
        """S, W, occ4 per particle read from the maintained lattice arrays (lattice formulation only)."""
        n = self._n_set[ensemble]
        S, W, occ4 = np.zeros(n), np.zeros(n), np.zeros((n, 4), np.int32)
        self._ck(self.lib.aps_lattice_accumulate(self._h, ensemble, _ptr(S), _ptr(W), _ptr(occ4), n))
        return S, W, occ4

    def get_lattice(self, ensemble=0):
        """The maintained lattice arrays: W = tot_conv, S = s_conv (weight-grid units), occ = particles per site."""
        W, S, occ = np.zeros(self.L), np.zeros(self.L), np.zeros(self.L, np.int32)
        self._ck(self.lib.aps_get_lattice(self._h, ensemble, _ptr(W), _ptr(S), _ptr(occ)))
        return W, S, occ

    # -- stepping
    def step(self, nsteps=1):
        self._ck(self.lib.aps_step(self._h, int(nsteps)))

    def propose(self):
        self._ck(self.lib.aps_propose(self._h))

    def commit(self):
        self._ck(self.lib.aps_commit(self._h))

    def exchange_buffer(self):
        ptr, tot, off, mine = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.lib.aps_exchange_buffer(self._h, C.byref(ptr), C.byref(tot), C.byref(off), C.byref(mine)))
        return ptr.value, tot.value, off.value, mine.value

    def bind_exchange_buffer(self, dev_ptr, nbytes):
        self._ck(self.lib.aps_bind_exchange_buffer(self._h, C.c_void_p(dev_ptr), int(nbytes)))

    def comm_init(self, id128: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(id128)
        self._ck(self.lib.aps_comm_init(self._h, C.cast(buf, C.c_void_p)))

    def owned_sites(self):
        """[lo, hi) of the sites this handle steps (the whole lattice unless it is a site-sharded tiles handle)."""
        lo, hi = C.c_int32(), C.c_int32()
        self._ck(self.lib.aps_owned_sites(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def halo_from(self, neighbour):
        """Copy the halo this handle needs from a neighbour rank's handle on the same device (between propose and commit)."""
        self._ck(self.lib.aps_halo_copy(self._h, neighbour._h))

    def halo_info(self):
        """(steps per halo exchange, steps since the last one, exchange due between this propose and its commit)"""
        k, age, due = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._ck(self.lib.aps_halo_info(self._h, C.byref(k), C.byref(age), C.byref(due)))
        return int(k.value), int(age.value), bool(due.value)

    def halo_sizes(self):
        """((first block, last block) this rank sends, (right neighbour's first, left neighbour's last) it receives), bytes"""
        snd, rcv = (C.c_int64 * 2)(), (C.c_int64 * 2)()
        self._ck(self.lib.aps_halo_sizes(self._h, snd, rcv))
        return (int(snd[0]), int(snd[1])), (int(rcv[0]), int(rcv[1]))

    def halo_pack(self, side):
        """This rank's first (side 0) / last (side 1) halo block as a uint8 array (between propose and commit)."""
        n = C.c_int64()
        self._ck(self.lib.aps_halo_pack(self._h, int(side), None, 0, C.byref(n)))
        buf = np.zeros(max(n.value, 1), np.uint8)
        self._ck(self.lib.aps_halo_pack(self._h, int(side), _ptr(buf), len(buf), C.byref(n)))
        return buf[:n.value]

    def halo_unpack(self, from_side, data):
        """Store a neighbour's block: from_side 0 = the RIGHT neighbour's first block, 1 = the LEFT neighbour's last block."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        self._ck(self.lib.aps_halo_unpack(self._h, int(from_side), _ptr(data), len(data)))

    def comm_selftest(self, nbytes=1 << 16):
        """nbytes from this rank to itself through the halo's transport calls (ncclSend / ncclRecv in one group)."""
        self._ck(self.lib.aps_comm_selftest(self._h, int(nbytes)))

    def ipc_export(self) -> bytes:
        """Peer-store transport of the halo, step 1: allocate this rank's landing buffers and return the 256-byte blob its two
        neighbour ranks need (any transport may carry it: torch.distributed gloo, a pipe, a file)."""
        buf = (C.c_uint8 * 256)()
        self._ck(self.lib.aps_ipc_export(self._h, C.cast(buf, C.c_void_p)))
        return bytes(buf)

    def ipc_connect(self, left_blob, right_blob):
        """Step 2: map the neighbours' landing buffers (None where a reflecting wall is the neighbour); afterwards `step` moves
        the halo itself -- packed blocks stored straight into the neighbour's memory, one arrival word per block."""
        keep = [None if b is None else (C.c_uint8 * 256).from_buffer_copy(b) for b in (left_blob, right_blob)]
        self._ck(self.lib.aps_ipc_connect(self._h, *[None if k is None else C.cast(k, C.c_void_p) for k in keep]))

    def set_flip_table(self, table):
        """A caller's flip_rate_fn tabulated over m in [-1, 1]: table[2][n + 1], row 0 sigma = +1, row 1 sigma = -1 (None: back to
        the Curie-Weiss rate); every rate evaluation of this handle interpolates it linearly (aps_set_flip_table)."""
        if table is None:
            self._ck(self.lib.aps_set_flip_table(self._h, None, 0))
            return
        tab = np.ascontiguousarray(table, dtype=np.float64)
        assert tab.ndim == 2 and tab.shape[0] == 2 and tab.shape[1] >= 2
        self._ck(self.lib.aps_set_flip_table(self._h, _ptr(tab), tab.shape[1] - 1))

    def ntt_info(self):
        """dict(on, log2_m, prof_ms, prof_launches, launches): does this handle update the field by the exact convolution (aps_ntt_info)"""
        on, m, ms, n = C.c_int32(), C.c_int32(), C.c_double(), C.c_int64()
        self._ck(self.lib.aps_ntt_info(self._h, C.byref(on), C.byref(m), C.byref(ms), C.byref(n)))
        return dict(on=bool(on.value), log2_m=m.value, prof_ms=ms.value, prof_launches=n.value, launches=int(self.lib.aps_ntt_launches(self._h)))

    def tiles_info(self):
        """dict(frame_sites, owned_sites, n_tiles, table_in_lds) of a tiles handle (aps_tiles_info)."""
        v = [C.c_int32() for _ in range(4)]
        self._ck(self.lib.aps_tiles_info(self._h, *[C.byref(x) for x in v]))
        return dict(frame_sites=v[0].value, owned_sites=v[1].value, n_tiles=v[2].value, table_in_lds=bool(v[3].value))

    def exchange_kind(self):
        """How `step` moves the halo of a sharded handle: "ipc-peer", "rccl" or "none" (the caller moves it)."""
        return {0: "none", 1: "rccl", 2: "ipc-peer"}[self.lib.aps_exchange_kind(self._h)]

    def comm_ranks(self):
        n = C.c_int32()
        self._ck(self.lib.aps_comm_ranks(self._h, C.byref(n)))
        return n.value

    def step_timed(self, nsteps):
        ms, n, pairs = C.c_double(), C.c_int64(), C.c_double()
        self._ck(self.lib.aps_step_timed(self._h, int(nsteps), C.byref(ms), C.byref(n), C.byref(pairs)))
        return ms.value, n.value, pairs.value

    def step_profile(self, nsteps):
        """{kernel: (summed ms, launches)} over nsteps steps launched one by one with HIP events around each."""
        ms, cnt = np.zeros(len(KERNELS)), np.zeros(len(KERNELS), np.int64)
        self._ck(self.lib.aps_step_profile(self._h, int(nsteps), _ptr(ms), _ptr(cnt)))
        return {k: (float(m), int(c)) for k, m, c in zip(KERNELS, ms, cnt)}

    def step_info(self):
        """(steps replayed from hipGraphs, steps launched kernel by kernel) of the last step() call."""
        g, k = C.c_int64(), C.c_int64()
        self._ck(self.lib.aps_step_info(self._h, C.byref(g), C.byref(k)))
        return g.value, k.value

    def set_resident_loop(self, on=True):
        """Let step() run its steps inside one launch when every tile of the grid is resident at once (default), or not."""
        self._ck(self.lib.aps_set_resident_loop(self._h, 1 if on else 0))

    def step_loop_timed(self, nsteps):
        """step(nsteps) with start/stop events attached to the resident loop's dispatch: (kernel ms, steps it took)."""
        ms, n = C.c_double(), C.c_int64()
        self._ck(self.lib.aps_step_loop_timed(self._h, int(nsteps), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def loop_info(self):
        """(steps of the last step() call taken inside the resident loop, state, reason when it is not used);
        state: 1 usable, 0 not eligible, -1 a call gave up and was repeated with one launch per step, -2 not looked at yet."""
        n, st, why = C.c_int64(), C.c_int32(), C.create_string_buffer(256)
        self._ck(self.lib.aps_loop_info(self._h, C.byref(n), C.byref(st), why, 256))
        return n.value, st.value, why.value.decode()

    def copy_bandwidth(self, nbytes=1 << 30, reps=5):
        """GB/s (read + written) of a plain streaming copy kernel on this handle's device."""
        r = C.c_double()
        self._ck(self.lib.aps_copy_bandwidth(self._h, int(nbytes), int(reps), C.byref(r)))
        return r.value

    def event_overhead(self, reps=50):
        """Elapsed time (ms) the HIP events report for an empty bracket on the handle's stream."""
        ms = C.c_double()
        self._ck(self.lib.aps_event_overhead(self._h, int(reps), C.byref(ms)))
        return ms.value

    SCALARS = ("n", "sum_sigma", "sum_pos", "n_wall", "max_pos", "n_range", "attempts", "blocked", "sum_d", "sum_d2", "n_d")

    def mark_reference(self, ensemble=0):
        """Remember the current positions of ensemble `ensemble` as the origin of the displacement sums."""
        self._ck(self.lib.aps_mark_reference(self._h, ensemble))

    def observe_scalars(self, ensemble=0, x_wall=0, range_lo=0, range_hi=-1, block_table=None):
        """Integer sums over the live particles (see include/aps.h) as a dict of Python ints."""
        out = np.zeros(len(self.SCALARS), np.int64)
        tab = None if block_table is None else np.ascontiguousarray(block_table, dtype=np.uint8)
        if tab is not None and tab.size != (self.K + 1) ** 2:
            raise ValueError("block_table must have (K+1)*(K+1) entries")
        self._ck(self.lib.aps_observe_scalars(self._h, ensemble, int(x_wall), int(range_lo), int(range_hi), _ptr(tab), _ptr(out)))
        return {k: int(v) for k, v in zip(self.SCALARS, out)}

    def observe_scalars_all(self, x_wall=0, ranges=None, block_table=None):
        """observe_scalars for every ensemble of the handle in one pass; `ranges` = [E][2] (lo, hi) per ensemble.
        Returns a list of dicts."""
        out = np.zeros((self.E, len(self.SCALARS)), np.int64)
        tab = None if block_table is None else np.ascontiguousarray(block_table, dtype=np.uint8)
        rng = None if ranges is None else np.ascontiguousarray(ranges, dtype=np.int32).reshape(self.E, 2)
        self._ck(self.lib.aps_observe_scalars_all(self._h, int(x_wall), _ptr(rng), _ptr(tab), _ptr(out)))
        return [{k: int(v) for k, v in zip(self.SCALARS, row)} for row in out]

    def observe_bins(self, nbins, ensemble=0):
        """(plus, minus) live particles per bin of ceil(L / nbins) consecutive sites, counted on the device."""
        cp, cm = np.zeros(nbins, np.int64), np.zeros(nbins, np.int64)
        self._ck(self.lib.aps_observe_bins(self._h, ensemble, int(nbins), _ptr(cp), _ptr(cm)))
        return cp, cm

    def observe_structure(self, ensemble=0, k_max=25):
        """(n live, sum count^2 over sites, sum m, sum m^2 over sites, [k_max][2] Fourier sums of the site histogram)."""
        k_max = int(min(k_max, self.L))
        out = np.zeros(4 + 2 * k_max)
        self._ck(self.lib.aps_observe_structure(self._h, ensemble, k_max, _ptr(out)))
        return int(out[0]), float(out[1]), float(out[2]), float(out[3]), out[4:].reshape(k_max, 2)

    # -- observation
    def observe(self, ensemble=0, want_field=True):
        cp, cm = np.zeros(self.L, np.int64), np.zeros(self.L, np.int64)
        m = np.zeros(self.L) if want_field else None
        self._ck(self.lib.aps_observe(self._h, ensemble, _ptr(cp), _ptr(cm), _ptr(m)))
        return cp, cm, m

    def field_from_counts(self, counts_p, counts_m, ensemble=0):
        cp = np.ascontiguousarray(counts_p, dtype=np.int64)
        cm = np.ascontiguousarray(counts_m, dtype=np.int64)
        m = np.zeros(self.L)
        self._ck(self.lib.aps_field_from_counts(self._h, ensemble, _ptr(cp), _ptr(cm), _ptr(m)))
        return m

    def rates_from_field(self, pos, sigma, bound, m_field, counts_p, counts_m, ensemble=0):
        """dict of the nine rate vectors of step_gillespie's rate section for caller-supplied arrays."""
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        sigma = np.ascontiguousarray(sigma, dtype=np.int8)
        bound = np.ascontiguousarray(bound, dtype=np.uint8)
        m_field = np.ascontiguousarray(m_field, dtype=np.float64)
        cp = np.ascontiguousarray(counts_p, dtype=np.int64)
        cm = np.ascontiguousarray(counts_m, dtype=np.int64)
        out = np.zeros((9, len(pos)))
        self._ck(self.lib.aps_rates_from_field(self._h, ensemble, _ptr(pos), _ptr(sigma), _ptr(bound), len(pos), _ptr(m_field),
                                               _ptr(cp), _ptr(cm), _ptr(out)))
        return dict(zip(("diff", "act", "flip", "bind", "unbind", "exit", "left", "right", "total"), out))

    def time(self):
        t, k = C.c_double(), C.c_int64()
        self._ck(self.lib.aps_time(self._h, C.byref(t), C.byref(k)))
        return t.value, k.value

    def exits(self, ensemble=0):
        n = C.c_int64()
        self._ck(self.lib.aps_get_exits(self._h, ensemble, None, 0, C.byref(n)))
        rows = np.zeros((max(n.value, 1), 3))
        self._ck(self.lib.aps_get_exits(self._h, ensemble, _ptr(rows), len(rows), C.byref(n)))
        return rows[:n.value]

    def table(self):
        tlen, q = C.c_int32(), C.c_int32()
        self._ck(self.lib.aps_get_table(self._h, None, 0, C.byref(tlen), C.byref(q)))
        out = np.zeros(max(tlen.value, 1))
        self._ck(self.lib.aps_get_table(self._h, _ptr(out), len(out), C.byref(tlen), C.byref(q)))
        return out[:tlen.value], q.value

    def resort(self):
        self._ck(self.lib.aps_resort(self._h))


def device_count():
    return load().aps_device_count()


def comm_unique_id() -> bytes:
    """128-byte RCCL unique id (call on one rank, broadcast the bytes to the others).
    The library dlopens librccl.so.1 lazily; import torch first so that it resolves to the copy torch loaded."""
    lib = load()
    buf = (C.c_uint8 * 128)()
    rc = lib.aps_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != APS_OK:
        raise ApsError(rc, lib.aps_last_error(None).decode())
    return bytes(buf)
