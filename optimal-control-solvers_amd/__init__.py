"""optimal-control-solvers_amd: MI355X-native batched RK4 state / discrete-adjoint /
forward-backward-sweep kernels behind the OCProblem / Integrator / Control plugin
surface of DrDanRyan/Optimal-Control-Solvers.

The directory name is not a Python identifier; load it with
`__graft_entry__.load_package()` (registers it as module `ocs_amd`).
"""
from . import _lib  # noqa: F401  fails loudly when libocs.so is missing
from ._lib import OcsError  # noqa: F401


def tracing_enabled():
    """True if a roctx marker library was found: every compute entry point then opens a named range"""
    return bool(_lib.lib.ocs_tracing_enabled())

from .problem import OCProblem, TestOCProblem, LogisticProblem, LQProblem, UserProblem  # noqa: F401
from .integrator import Integrator, RK4Integrator, RK4InfiniteIntegrator, trajectory_status_dev  # noqa: F401
from .control import Control, PWLinearControl, PWConstantControl, ChebyshevControl  # noqa: F401
from .interp import vectorInterpolant, vectorInterpolant_dev, heval, linspace  # noqa: F401
from .solvers import (nlp_objective, nlp_objective_dev, single_shooting, single_shooting_batch,  # noqa: F401
                      compute_equilibrium, compute_equilibrium_dev)
from .sweep import fb_sweep, fb_sweep_batch, fb_sweep_dev, fb_sweep_path, compute_x_lam, compute_x_lam_J, compute_J  # noqa: F401
from . import distributed  # noqa: F401
from .multi import MultiDevice  # noqa: F401


def make_from_symbolic(*args, **kwargs):
    """prob = make_from_symbolic(symObjective, symStateRHS, nStates, nControls, params, bounds)
    (functions/make_from_symbolic.m); SymPy is imported on first use (symbolic.py)."""
    from .symbolic import make_from_symbolic as _m
    return _m(*args, **kwargs)
