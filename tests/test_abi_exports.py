"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports
every symbol include/ocs.h declares, and fails loudly (no CPU fallback) on compute calls."""
import ctypes as C
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    g.build()
    return g.load_package()


def test_library_exports_every_declared_symbol(pkg):
    from ocs_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(_lib.lib, n)]
    assert not missing, missing
    # and the binding table covers the header
    assert sorted(_lib._SIG) == names
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for n in names:
        assert f" T {n}" in out


def test_header_is_plain_c(tmp_path):
    # MATLAB's loadlibrary needs a header a C compiler accepts
    src = tmp_path / "t.c"
    src.write_text('#include "ocs.h"\nint main(void){return OCS_OK;}\n')
    import os
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-std=c89", "-pedantic", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "t.o")],
                   check=True)


def test_host_side_grid_and_argument_validation(pkg):
    import torch
    from ocs_amd import _lib
    lib = _lib.lib
    h = C.c_void_p()
    ts = np.array([0.0, 0.5, 2.0])
    assert lib.ocs_rk4_create(C.byref(h), ts.ctypes.data_as(_lib.dp), 3) == 0
    n = C.c_int()
    lib.ocs_integrator_nsteps(h, C.byref(n))
    t = np.empty(5)
    lib.ocs_integrator_t(h, t.ctypes.data_as(_lib.dp))
    assert n.value == 2 and np.array_equal(t, [0.0, 0.25, 0.5, 1.25, 2.0])  # RK4Integrator.m:21-24
    lib.ocs_integrator_destroy(h)
    assert lib.ocs_rk4_create(C.byref(h), ts.ctypes.data_as(_lib.dp), 1) == -2
    assert b"tspan" in lib.ocs_last_error()
    bad = np.array([0.0, 1.0, 1.0])
    assert lib.ocs_rk4_create(C.byref(h), bad.ctypes.data_as(_lib.dp), 3) == -1
    par, bnd = np.array([1.5, 3.0, 0.05]), np.array([0.0, 1.0])
    assert lib.ocs_problem_create(C.byref(h), 99, 1, 1, par.ctypes.data_as(_lib.dp), 3, bnd.ctypes.data_as(_lib.dp)) == -6
    assert lib.ocs_problem_create(C.byref(h), 1, 2, 1, par.ctypes.data_as(_lib.dp), 3, bnd.ctypes.data_as(_lib.dp)) == -2
    if not torch.cuda.is_available():
        # no GPU: compute must fail loudly, never fall back to a CPU path
        prob = pkg.TestOCProblem({"c": 1.5, "m": 3.0, "r": 0.05}, [[0.0, 1.0]])
        g = pkg.RK4Integrator(np.linspace(0, 1, 11))
        with pytest.raises(pkg.OcsError) as e:
            g.compute_states(prob, [1.0], np.zeros((1, 21)))
        assert e.value.code == -4


def test_matlab_shims_only_call_exported_symbols(pkg):
    """matlab/*.m (the reference-side binding, unverifiable without MATLAB): every calllib target must be a symbol the
    header declares and the library exports, and every shim must subclass / replace what it says it does."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "matlab", "*.m")))
    assert {os.path.basename(f) for f in files} >= {"GpuTestOCProblem.m", "GpuRK4Integrator.m", "GpuPWLinearControl.m",
                                                    "gpu_fb_sweep.m", "gpu_compute_equilibrium.m", "ocs_check.m",
                                                    "ocs_load.m", "gpu_nlp_objective.m"}
    header = open(os.path.join(root, "include", "ocs.h")).read()
    declared = set(re.findall(r"\b(ocs_[a-z0-9_]+)\s*\(", header))
    called = set()
    for f in files:
        called |= set(re.findall(r"calllib\('libocs',\s*'(ocs_[a-z0-9_]+)'", open(f).read()))
    assert called and called <= declared, sorted(called - declared)
    lib = C.CDLL(os.path.join(root, "optimal-control-solvers_amd", "lib", "libocs.so"))
    for name in called:
        assert hasattr(lib, name), name
    assert "classdef GpuRK4Integrator < Integrator" in open(os.path.join(root, "matlab", "GpuRK4Integrator.m")).read()
    assert "classdef GpuPWLinearControl < Control" in open(os.path.join(root, "matlab", "GpuPWLinearControl.m")).read()
    assert "classdef GpuTestOCProblem < OCProblem" in open(os.path.join(root, "matlab", "GpuTestOCProblem.m")).read()
