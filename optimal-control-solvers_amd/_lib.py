"""ctypes binding of libocs.so (the C-ABI declared in include/ocs.h).

There is no CPU fallback: if the shared library has not been built the import fails, and
every compute call fails with OCS_ERR_NO_DEVICE when no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import torch  # noqa: F401  (first: makes torch's libamdhip64.so.7 the process-wide HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OCS_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "libocs.so")  # override: tuning builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ocs.h")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
vp = C.c_void_p


class OcsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libocs error {code}: {msg}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    return C.CDLL(LIB_PATH)


lib = _load()

_SIG = {
    # name: (restype, argtypes)
    "ocs_version": (C.c_char_p, []),
    "ocs_last_error": (C.c_char_p, []),
    "ocs_device_count": (C.c_int, [ip]),
    "ocs_set_device": (C.c_int, [C.c_int]),
    "ocs_synchronize": (C.c_int, []),
    "ocs_problem_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, dp, C.c_int, dp]),
    "ocs_fb_sweep_path": (C.c_int, [vp]),
    "ocs_problem_create_from_source": (C.c_int, [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int, dp, C.c_int, dp, C.c_int]),
    "ocs_problem_check_source": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ocs_problem_destroy": (C.c_int, [vp]),
    "ocs_problem_dims": (C.c_int, [vp, ip, ip]),
    "ocs_problem_set_batch_params": (C.c_int, [vp, C.c_int, ip, C.c_int, dp]),
    "ocs_problem_F": (C.c_int, [vp, C.c_int, dp, dp, dp, dp]),
    "ocs_problem_dFdx_times_vec": (C.c_int, [vp, C.c_int, dp, dp, dp, dp, dp]),
    "ocs_problem_dFdu_times_vec": (C.c_int, [vp, C.c_int, dp, dp, dp, dp, dp]),
    "ocs_problem_ControlChar": (C.c_int, [vp, C.c_int, dp, dp, dp, dp]),
    "ocs_compute_equilibrium": (C.c_int, [vp, C.c_int, C.c_double, dp, dp, dp, dp, dp, dp, ip]),
    "ocs_compute_equilibrium_dev": (C.c_int, [vp, C.c_int, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ocs_trajectory_status_dev": (C.c_int, [C.c_int, vp, vp, vp]),
    "ocs_integrator_trajectory_status": (C.c_int, [vp, C.c_int, ip]),
    "ocs_tracing_enabled": (C.c_int, []),
    "ocs_rk4_create": (C.c_int, [C.POINTER(vp), dp, C.c_int]),
    "ocs_integrator_destroy": (C.c_int, [vp]),
    "ocs_integrator_set_mapping": (C.c_int, [vp, C.c_int]),
    "ocs_control_set_fusion": (C.c_int, [vp, C.c_int]),
    "ocs_integrator_nsteps": (C.c_int, [vp, ip]),
    "ocs_integrator_t": (C.c_int, [vp, dp]),
    "ocs_integrator_h": (C.c_int, [vp, dp]),
    "ocs_compute_states": (C.c_int, [vp, vp, C.c_int, dp, dp, dp, dp]),
    "ocs_compute_adjoints": (C.c_int, [vp, vp, C.c_int, dp, dp, dp, dp]),
    "ocs_compute_states_dev": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, vp]),
    "ocs_compute_adjoints_dev": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, vp]),
    "ocs_rk4inf_create": (C.c_int, [C.POINTER(vp), dp, C.c_int, dp, C.c_int, dp, C.c_int]),
    "ocs_control_create": (C.c_int, [C.POINTER(vp), C.c_int, dp, C.c_int, C.c_int, C.c_int]),
    "ocs_control_destroy": (C.c_int, [vp]),
    "ocs_control_dims": (C.c_int, [vp, ip, ip, ip]),
    "ocs_control_basis": (C.c_int, [vp, dp]),
    "ocs_control_points": (C.c_int, [vp, dp]),
    "ocs_control_compute_u": (C.c_int, [vp, C.c_int, dp, dp]),
    "ocs_control_compute_dJdv": (C.c_int, [vp, C.c_int, dp, dp]),
    "ocs_control_compute_u_dev": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "ocs_control_compute_dJdv_dev": (C.c_int, [vp, C.c_int, vp, vp, vp]),
    "ocs_control_compute_initial_v": (C.c_int, [vp, dp, C.c_int, dp]),
    "ocs_control_compute_nlp_bounds": (C.c_int, [vp, dp, dp, dp]),
    "ocs_control_eval_uFunc": (C.c_int, [vp, dp, C.c_int, dp, dp]),
    "ocs_interp": (C.c_int, [C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, dp, dp]),
    "ocs_interp_dev": (C.c_int, [C.c_int, C.c_int, C.c_int, dp, vp, C.c_int, dp, vp, C.c_int, vp]),
    "ocs_nlp_objective": (C.c_int, [vp, vp, vp, C.c_int, dp, dp, C.c_int, ip, dp, dp]),
    "ocs_nlp_objective_dev": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, C.c_int, ip, vp, vp, vp]),
    "ocs_fbs_default_options": (C.c_int, [vp]),
    "ocs_ss_default_options": (C.c_int, [vp]),
    "ocs_single_shooting_batch_dev": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, C.c_int, ip, dp, dp, vp, vp, vp, vp, vp, vp]),
    "ocs_compute_x_lam": (C.c_int, [vp, vp, C.c_int, dp, dp, dp, dp, dp]),
    "ocs_compute_x_lam_dev": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp]),
    "ocs_fb_sweep": (C.c_int, [vp, vp, C.c_int, dp, vp, dp, dp, dp, dp, dp, dp, ip, dp]),
    "ocs_fb_sweep_dev": (C.c_int, [vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ocs_multi_create": (C.c_int, [C.POINTER(vp), ip, C.c_int]),
    "ocs_multi_destroy": (C.c_int, [vp]),
    "ocs_multi_size": (C.c_int, [vp]),
    "ocs_multi_device": (C.c_int, [vp, C.c_int]),
    "ocs_multi_shard": (C.c_int, [vp, C.c_int, C.c_int, ip, ip]),
    "ocs_multi_compute_states": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, dp, dp, dp, dp, dp]),
    "ocs_multi_compute_adjoints": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, dp, dp, dp, dp]),
    "ocs_multi_nlp_objective": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int, dp, dp, C.c_int, ip,
                                          dp, dp, dp]),
    "ocs_multi_fb_sweep": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.c_int, dp, vp, dp, dp, dp, dp, dp, dp, ip, dp, dp]),
    "ocs_multi_has_communicator": (C.c_int, [vp]),
    "ocs_multi_stream": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "ocs_multi_synchronize": (C.c_int, [vp]),
    "ocs_multi_stats": (C.c_int, [vp, dp]),
    "ocs_multi_compute_states_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), ip, C.POINTER(vp), C.POINTER(vp),
                                               C.POINTER(vp), C.POINTER(vp), C.c_int]),
    "ocs_multi_compute_adjoints_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), ip, C.POINTER(vp), C.POINTER(vp),
                                                 C.POINTER(vp), C.POINTER(vp)]),
    "ocs_multi_nlp_objective_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), ip, C.POINTER(vp),
                                              C.POINTER(vp), C.c_int, ip, C.POINTER(vp), C.POINTER(vp), C.c_int]),
    "ocs_multi_fb_sweep_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), ip, C.POINTER(vp), vp, C.POINTER(vp),
                                         C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int]),
    "ocs_device_malloc": (C.c_int, [C.POINTER(vp), C.c_ulong]),
    "ocs_device_free": (C.c_int, [vp]),
    "ocs_device_upload": (C.c_int, [vp, vp, C.c_ulong, vp]),
    "ocs_device_download": (C.c_int, [vp, vp, C.c_ulong, vp]),
    "ocs_copy_dev": (C.c_int, [vp, vp, C.c_long, vp]),
    "ocs_to_batch_minor_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "ocs_to_traj_major_dev": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
}

for _name, (_res, _args) in _SIG.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


class FbsOptions(C.Structure):
    """struct ocs_fbs_options (include/ocs.h)."""
    _fields_ = [("uRelTol", C.c_double), ("uAbsTol", C.c_double), ("nSWEEPS", C.c_int),
                ("nERROR_PTS", C.c_int), ("nINTERP_PTS", C.c_int), ("fused_update_off", C.c_int), ("nWINDOWS", C.c_int),
                ("cost_row", C.c_int), ("uRelax", C.c_double)]


class SsOptions(C.Structure):
    """struct ocs_ss_options (include/ocs.h)."""
    _fields_ = [("TolX", C.c_double), ("TolFun", C.c_double), ("MaxIter", C.c_int), ("memory", C.c_int),
                ("maxBacktracks", C.c_int)]


def declared_symbols():
    """Every function include/ocs.h declares (used by the export test)."""
    with open(HEADER_PATH) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ocs_[A-Za-z0-9_]+)\s*\(", src)))


def check(rc):
    """Negative status -> exception; positive (numerical condition) is returned to the caller."""
    if rc < 0:
        raise OcsError(rc, lib.ocs_last_error().decode())
    return rc
