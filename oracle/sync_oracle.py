"""ORACLE (test infrastructure): ctypes face of oracle/sync_oracle.c.

`SyncOracle` carries a lattice-gas state through fixed-dt steps on the CPU with the same Philox
counters, the same weight table and the same commit rule as the HIP path, so that the integer state
(pos, spin, bound, alive) can be compared bit for bit, step by step.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
import numpy as np

from . import build as _build
from .gillespie_numpy import LatticeGasParams

_lib = None


class OrcParams(C.Structure):
    _fields_ = [("L", C.c_int32), ("K", C.c_int32), ("periodic", C.c_int32), ("field_mode", C.c_int32),
                ("tlen", C.c_int32), ("minus_anchor", C.c_int32), ("immobilize", C.c_int32),
                ("suppress_flip", C.c_int32), ("crowding", C.c_int32), ("flip_n", C.c_int32),
                ("rate_diffusion", C.c_double), ("rate_active", C.c_double), ("beta", C.c_double),
                ("k_on", C.c_double), ("k_off", C.c_double), ("k_exit", C.c_double), ("dt", C.c_double),
                ("seed", C.c_uint64), ("ensemble", C.c_uint32), ("reserved2", C.c_uint32), ("flip_tab", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
        _lib.orc_exp.restype = C.c_double
        _lib.orc_exp.argtypes = [C.c_double]
        _lib.orc_build_table.restype = C.c_int32
        _lib.orc_sync_step.restype = C.c_int32
        _lib.orc_sync_commit.restype = C.c_int32
        _lib.orc_sync_run.restype = C.c_int32
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def philox4x32_10(ctr, key):
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


def det_exp(x):
    f = lib().orc_exp
    return np.array([f(float(v)) for v in np.atleast_1d(x)])


def build_table(sigma_grid, L, K, periodic, sum_bits=51):
    """Quantised, image-folded weight table -> (table f64[tlen], q).  sum_bits = 51: the exact binary64 field; 29: the grid
    of the library's 32-bit integer field (`fp32` mode)."""
    cap = L + 2
    buf = np.zeros(cap, dtype=np.float64)
    q = C.c_int32(0)
    n = lib().orc_build_table_bits(C.c_double(float(sigma_grid)), C.c_int32(L), C.c_int32(K),
                                   C.c_int32(int(bool(periodic))), C.c_int32(int(sum_bits)), _p(buf), C.c_int32(cap), C.byref(q))
    assert n >= 0
    return buf[:n].copy(), int(q.value)


class SyncOracle:
    def __init__(self, par: LatticeGasParams, dt: float, seed: int, ensemble: int = 0, raw_table=None, sum_bits=51, flip_table=None):
        """flip_table: [2][n + 1] tabulation of a custom flip_rate_fn (sigma = +1 / -1 over m in [-1, 1]), interpolated linearly --
        the product's device modes do the same (include/aps.h: aps_set_flip_table)."""
        self.par = par
        self.flip_table = None if flip_table is None else np.ascontiguousarray(flip_table, dtype=np.float64)
        self.dt = float(dt)
        if par.sigma_kernel > 0:
            if raw_table is None:
                self.table, self.q = build_table(par.sigma_grid, par.L, par.K, par.periodic, sum_bits)
            else:                                   # unquantised weights: proves the formula itself
                self.table, self.q = np.ascontiguousarray(raw_table, dtype=np.float64), None
            mode = 1
        else:
            self.table, self.q, mode = np.zeros(1), None, 0
        self.c = OrcParams(L=par.L, K=par.K, periodic=int(par.periodic), field_mode=mode,
                           tlen=len(self.table), minus_anchor=int(par.minus_anchor),
                           immobilize=int(par.immobilize_when_anchored),
                           suppress_flip=int(par.suppress_flip_when_bound),
                           crowding=int(par.crowding_suppresses_rates), rate_diffusion=par.rate_diffusion,
                           rate_active=par.rate_active, beta=par.beta, k_on=par.k_on, k_off=par.k_off,
                           k_exit=par.k_exit, dt=self.dt, seed=int(seed) & (2**64 - 1), ensemble=int(ensemble),
                           flip_n=0 if self.flip_table is None else self.flip_table.shape[1] - 1,
                           flip_tab=None if self.flip_table is None else self.flip_table.ctypes.data)
        self.anchor = np.ascontiguousarray(par.is_anchor_site, dtype=np.uint8)
        self.step_index = 0
        self.exit_log = np.zeros((0, 3))
        self.set_state(np.zeros(0, np.int32), np.zeros(0, np.int8))

    # ------------------------------------------------------------------ state
    def set_state(self, pos, spin, bound=None, alive=None):
        n = len(pos)
        self.pos = np.ascontiguousarray(pos, dtype=np.int32).copy()
        self.spin = np.ascontiguousarray(spin, dtype=np.int8).copy()
        self.bound = (np.zeros(n, np.uint8) if bound is None
                      else np.ascontiguousarray(bound, dtype=np.uint8).copy())
        self.alive = (np.ones(n, np.uint8) if alive is None
                      else np.ascontiguousarray(alive, dtype=np.uint8).copy())
        self._exit_buf = np.zeros((max(n, 1), 3), dtype=np.float64)
        self._n_exit = C.c_int64(0)

    @property
    def n(self):
        return len(self.pos)

    # ------------------------------------------------------------------ pieces
    def pair_sums(self):
        n = self.n
        S, W, occ4 = np.zeros(n), np.zeros(n), np.zeros((n, 4), dtype=np.int32)
        lib().orc_pair_sums(C.byref(self.c), _p(self.table), C.c_int64(n), _p(self.pos), _p(self.spin),
                            _p(self.alive), _p(S), _p(W), _p(occ4))
        return S, W, occ4

    def field_sites(self):
        L = self.par.L
        cp, cm, m = np.zeros(L, np.int32), np.zeros(L, np.int32), np.zeros(L)
        S, W = np.zeros(L), np.zeros(L)
        lib().orc_field_sites(C.byref(self.c), _p(self.table), C.c_int64(self.n), _p(self.pos),
                              _p(self.spin), _p(self.alive), _p(cp), _p(cm), _p(m), _p(S), _p(W))
        self.last_site_sums = (S, W)
        return cp, cm, m

    def rates_from_field(self, m_field, use_libm=True):
        n = self.n
        out = np.zeros((9, n))
        m_field = np.ascontiguousarray(m_field, dtype=np.float64)
        lib().orc_rates(C.byref(self.c), _p(self.anchor), C.c_int64(n), _p(self.pos), _p(self.spin),
                        _p(self.bound), _p(m_field), C.c_int32(int(use_libm)), _p(out))
        names = ("diff", "act", "flip", "bind", "unbind", "exit", "left", "right", "total")
        return dict(zip(names, out))

    # ------------------------------------------------------------------ stepping
    def step(self, want_detail=False):
        n = self.n
        prop = np.zeros(n, np.uint8)
        acc = np.zeros(n, np.uint8)
        S, W = np.zeros(n), np.zeros(n)
        rc = lib().orc_sync_step(C.byref(self.c), _p(self.table), _p(self.anchor), C.c_int64(n),
                                 _p(self.pos), _p(self.spin), _p(self.bound), _p(self.alive),
                                 C.c_uint64(self.step_index), _p(prop), _p(acc), _p(S), _p(W),
                                 _p(self._exit_buf), C.c_int64(len(self._exit_buf)), C.byref(self._n_exit))
        assert rc == 0
        self.step_index += 1
        if want_detail:
            return dict(prop=prop, accepted=acc, S=S, W=W)

    def propose(self, lo, hi, prop):
        """Proposal bytes of particles lo..hi-1 into prop[lo:hi] (uint8 array of length n)."""
        lib().orc_sync_propose(C.byref(self.c), _p(self.table), _p(self.anchor), C.c_int64(self.n), _p(self.pos),
                               _p(self.spin), _p(self.bound), _p(self.alive), C.c_uint64(self.step_index),
                               C.c_int64(lo), C.c_int64(hi), _p(prop), None, None)

    def commit(self, prop):
        rc = lib().orc_sync_commit(C.byref(self.c), C.c_int64(self.n), _p(self.pos), _p(self.spin), _p(self.bound),
                                   _p(self.alive), C.c_uint64(self.step_index), _p(prop), None, _p(self._exit_buf),
                                   C.c_int64(len(self._exit_buf)), C.byref(self._n_exit))
        assert rc == 0
        self.step_index += 1

    def run(self, nsteps):
        rc = lib().orc_sync_run(C.byref(self.c), _p(self.table), _p(self.anchor), C.c_int64(self.n),
                                _p(self.pos), _p(self.spin), _p(self.bound), _p(self.alive),
                                C.c_uint64(self.step_index), C.c_int64(int(nsteps)), _p(self._exit_buf),
                                C.c_int64(len(self._exit_buf)), C.byref(self._n_exit))
        assert rc == 0
        self.step_index += int(nsteps)

    @property
    def time(self):
        return self.step_index * self.dt

    def exits(self):
        """(time, position, particle index) rows in the order they were logged."""
        return self._exit_buf[:self._n_exit.value].copy()
