"""Register / scratch / LDS use of the kernels a hipRTC plugin gets, compiled OFFLINE with hipcc the way csrc/ocs_jit.cpp
assembles the program (no GPU needed): python scripts/offline_plugin_isa.py [ring6]  [kernel name expression ...]
Writes /tmp/ocs_offline/<name>.s and prints .vgpr_count / scratch / LDS per kernel."""
import os, re, subprocess, sys, importlib.util
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
CSRC = os.path.join(ROOT, 'optimal-control-solvers_amd', 'csrc')
sys.path.insert(0, os.path.join(ROOT, 'tests'))
spec = importlib.util.spec_from_file_location("symbolic", os.path.join(ROOT, 'optimal-control-solvers_amd', 'symbolic.py'))
sym = importlib.util.module_from_spec(spec); spec.loader.exec_module(sym)
import user_problems as up

which = sys.argv[1] if len(sys.argv) > 1 else "ring6"
if which == "ring6":
    gg, f, vals = up.ring6_symbolic(sym)
    gen = sym.generate(gg, f, 6, 3, vals, [[0.0, 1.0]] * 3)
    nS, nC = 6, 3
else:
    raise SystemExit("unknown problem")
kernels = sys.argv[2:] or ["ocs::k_costate<ocs::UserP, 4>", "ocs::k_control_grid<ocs::UserP>", "ocs::k_control_pts<ocs::UserP>",
                           "ocs::k_forward<ocs::UserP, 1, 4, true, false>", "ocs::k_backward<ocs::UserP, 1, 4, true, true, false>"]
rowsep = bool(gen["row_separable"]); fold = rowsep and gen["has_control_char"] and gen["control_from_costate"]
npar = len(gen["params"])
src = "#include <hip/hip_runtime.h>\n"
src += f"#define OCS_USER_NS {nS}\n#define OCS_USER_NC {nC}\n#define OCS_USER_NPAR {npar}\n"
if gen["has_control_char"]: src += "#define OCS_USER_HAS_CONTROLCHAR 1\n"
if rowsep: src += "#define OCS_USER_ROWSEP 1\n"
if fold: src += "#define OCS_USER_CC_NOX 1\n"
src += '#include "ocs_device_common.hpp"\nconstexpr int NS = OCS_USER_NS, NC = OCS_USER_NC, NPAR = OCS_USER_NPAR;\n'
src += "typedef const double* OCS_PARAMS;\n" if (npar <= 16 and not rowsep) else "typedef ocs::uniform_ptr OCS_PARAMS;\n"
src += gen["source"]
src += '\n#include "ocs_user_functor.hpp"\n#include "ocs_rk4_kernels.hpp"\n#include "ocs_fbs_device.hpp"\n'
for h in sys.argv[0:0]: pass
extra = os.environ.get("EXTRA_INCLUDES", "")
for h in extra.split(","):
    if h: src += f'#include "{h}"\n'
for k in kernels:
    src += f"template __global__ void {k}(" + "decltype(ocs::first_arg(&" + k + ")));\n" if False else ""
# explicit instantiation needs the argument type: take the address instead
for i, k in enumerate(kernels):
    src += f"auto* ocs_keep_{i} = &{k};\n"
out = "/tmp/ocs_offline"; os.makedirs(out, exist_ok=True)
path = os.path.join(out, f"{which}.hip")
open(path, "w").write(src)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", f"-I{CSRC}", "-S",
       "--cuda-device-only", "-o", os.path.join(out, f"{which}.s"), path]
subprocess.run(cmd, check=True)
txt = open(os.path.join(out, f"{which}.s")).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    body = m.group(2)
    g = lambda key: (re.search(rf"\.amdhsa_{key} (\S+)", body) or [None, "?"])[1]
    print(f"{m.group(1)[:90]:90s} vgpr {g('next_free_vgpr'):>4s} accum_offset {g('accum_offset'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
