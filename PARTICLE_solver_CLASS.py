"""Drop-in module name of the reference: `from PARTICLE_solver_CLASS import ParticleSystem`
(PARTICLE_solver_BIOLOGY_EXCLUSION*.py line 12/13).  Re-exports the MI355X-backed class.

Same constructor keywords, attributes and `run()` result dictionary as the reference.  A driver left unchanged gets the
reference's own dynamics: `run()` is the exact one-event-per-iteration loop (ref :511-516) resident on the GPU
(`mode="gillespie_gpu"`; random numbers from Philox instead of the caller's Generator, so agreement is in distribution --
fixture G4 without a bias allowance; `mode="gillespie"` draws from `rng` in the reference's order and reproduces seeded
trajectories bit for bit, and is the default when a custom `flip_rate_fn` is given).  The fixed-`dt` synchronous stepper
(`mode="sync"`, or simply passing `dt=`) is the high-throughput path, first order in dt.  What still differs: `m_local_list[k]`
is the field of the observed state, K <= 32 and L <= 2^25."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module(
    "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd")
ParticleSystem = _pkg.ParticleSystem
