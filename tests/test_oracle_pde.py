"""CPU suite: oracle/pde_numpy.py (restatement of the reference's IMEXPDE) against fixture G6, which holds seeded
runs of the reference itself for all four boundary-condition / active-model combinations.  Bar: bit-exact
(same legacy np.random call sequence, same scipy/numpy routines)."""
import numpy as np

from oracle.pde_numpy import PdeOracle


def test_pde_oracle_reproduces_reference_runs_bit_for_bit(golden):
    g = golden("g6_pde.npz")
    kw, init = g.meta["kw"], g.meta["init"]
    for idx, c in enumerate(g.meta["cases"]):
        pde = PdeOracle(bc=c["bc"], active_model=c["active_model"], gaussian_kernel=c["gaussian_kernel"], seed=c["seed"], **kw)
        pde.initialize(mode=c["init"], **init)
        pre = f"c{idx}_"
        assert np.array_equal(pde.rho_p, g[pre + "rho_p0"]) and np.array_equal(pde.rho_m, g[pre + "rho_m0"]), c["tag"]
        pde.solve()
        out = pde.get_output()
        for k in ("rho_p", "rho_m", "m_series", "var_series", "snapshots", "times"):
            assert np.array_equal(out[k], g[pre + k]), (c["tag"], k)
        for k in ("v_eff_series", "D_eff_series"):
            assert np.array_equal(out[k], g[pre + k], equal_nan=True), (c["tag"], k)
        assert np.array_equal(out["fft_amp"][-1], g[pre + "fft_amp_last"])
        assert np.array_equal(pde.tracers_unwrapped, g[pre + "tracers"])
        assert np.array_equal(pde.tracer_state, g[pre + "tracer_state"])
