"""Drop-in module name of the reference: `from PARTICLE_solver_CLASS import ParticleSystem`
(PARTICLE_solver_BIOLOGY_EXCLUSION*.py line 12/13).  Re-exports the MI355X-backed class.

Same constructor keywords, attributes and `run()` result dictionary as the reference.  What differs when a driver is left
unchanged: `run()` advances in fixed steps `dt` (synchronous scheme, first order in dt against the reference's exact
Gillespie dynamics; `mode="gillespie_gpu"` or `mode="gillespie"` give the exact dynamics), `m_local_list[k]` is the field of
the observed state, a custom `flip_rate_fn` needs `mode="gillespie"`, K <= 32 and L <= 2^25."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module(
    "hydrodynamic-limits-of-active-particle-systems-with-mean-field-interactions_amd")
ParticleSystem = _pkg.ParticleSystem
