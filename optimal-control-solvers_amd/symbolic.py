"""Symbolic front-end for OCProblem plugins (reference: functions/make_from_symbolic.m:1-114).

The reference derives its Gen-1 problem struct from two symbolic expressions with MATLAB's Symbolic Toolbox:
    H = objective + lam * stateRHS,  adjointRHS = -grad_x H,  dHdu = grad_u H,  ControlChar = solve(dHdu = 0, u)
clamped to the control bounds (:11-23, :111), and turns them into function handles with matlabFunction (:26-31).
Here the same derivation runs in SymPy and ends in *device source* for the hipRTC plugin path
(ocs_problem_create_from_source, csrc/ocs_user_functor.hpp): the Gen-2 methods F, dFdx_times_vec, dFdu_times_vec of
OCProblem/OCProblem.m:8-21 (so RK4Integrator / single_shooting run on the problem) plus ocs_ControlChar (so fb_sweep
does).  The generator picks the fastest plugin form the expressions allow:

  * row functions (ocs_row_*), when nC = 1, nS in {1, 2, 4}, every state equation reads its own state only and the
    objective integrand is a sum of per-state terms: the wave-specialised state pass and the scan adjoint pass;
    with flag bit 2 (`control_from_costate`) when ControlChar does not read x and dF_r/dy_r, dq_r/dy_r do not read u:
    fb_sweep's two-kernel sweep;
  * the three full-vector methods otherwise;
  * a time coefficient tabulated once per grid point (ocs_row_tcoef / OCS_USER_TCOEF, OCS_USER_CC_TCOEF) when the time
    enters the methods through ONE sub-expression of t and the parameters (e.g. exp(-r t)).

Symbols follow the reference (:4-8): t, x1..xn, lam1..lamn, u1..um -- `symbols(nStates, nControls, params)` returns them.
Nothing here touches the GPU except `make_from_symbolic(..., create=True)`, which compiles the source and returns a
UserProblem; `generate` alone needs SymPy only (and UserProblem.check_source compiles without a GPU)."""
from __future__ import annotations

import numpy as np
import sympy as sp
from sympy.printing.c import C99CodePrinter


def symbols(nStates, nControls, params=()):
    """t, x, lam, u and the parameter symbols as make_from_symbolic.m:3-8 names them: x1.., lam1.., u1..; one symbol
    per parameter name.  Returns (t, x, lam, u, p) with x, lam, u lists and p a dict name -> Symbol."""
    t = sp.Symbol("t", real=True)
    x = [sp.Symbol(f"x{i + 1}", real=True) for i in range(nStates)]
    lam = [sp.Symbol(f"lam{i + 1}", real=True) for i in range(nStates)]
    u = [sp.Symbol(f"u{i + 1}", real=True) for i in range(nControls)]
    p = {name: sp.Symbol(name, real=True) for name in params}
    return t, x, lam, u, p


class _DevicePrinter(C99CodePrinter):
    """C for the device: small integer powers as products (pow(x, 2) is a libm call per evaluation), everything double."""

    def _print_Pow(self, expr):
        b, e = expr.as_base_exp()
        if e.is_Integer and 2 <= int(e) <= 4:
            s = self.parenthesize(b, 1000)   # atoms stay bare, everything else is parenthesised
            return "(" + " * ".join([s] * int(e)) + ")"
        if e.is_Integer and -4 <= int(e) <= -1:
            s = self.parenthesize(b, 1000)
            return "(1.0 / (" + " * ".join([s] * int(-e)) + "))"
        return super()._print_Pow(expr)

    def _print_Integer(self, expr):
        return f"{int(expr)}.0"

    def _print_Rational(self, expr):
        return f"({int(expr.p)}.0 / {int(expr.q)}.0)"


def _emit(outputs, names, indent="  "):
    """C statements assigning outputs[k] (SymPy expressions in already-renamed symbols) to names[k], with common
    sub-expressions factored out."""
    pr = _DevicePrinter()
    rep, red = sp.cse(list(outputs), symbols=sp.numbered_symbols("w_"), optimizations="basic")
    lines = [f"{indent}const double {pr.doprint(s)} = {pr.doprint(e)};" for s, e in rep]
    lines += [f"{indent}{n} = {pr.doprint(e)};" for n, e in zip(names, red)]
    return "\n".join(lines)


def _time_subexpressions(exprs, t, state_like):
    """The maximal sub-expressions through which t enters: they contain t and no symbol of `state_like` (x, lam, u)."""
    found = set()

    def walk(e):
        if not e.has(t):
            return
        if not (e.free_symbols & state_like):
            found.add(e)
            return
        for a in e.args:
            walk(a)
    for e in exprs:
        walk(sp.sympify(e))
    return found


def _hoist(exprs, t, state_like, tc):
    """If t enters `exprs` through exactly one sub-expression E(t, params), returns (E, exprs with E -> tc); if t does not
    enter at all, (t, exprs); else (None, exprs): no hoisting."""
    subs = _time_subexpressions(exprs, t, state_like)
    if not subs:
        return t, list(exprs)
    if len(subs) == 1:
        E = next(iter(subs))
        out = [sp.sympify(e).xreplace({E: tc}) for e in exprs]
        if not any(o.has(t) for o in out):
            return E, out
    # products like 2 c exp(-r t): try the common exponential / the bare t-dependent factor
    cands = set()
    for s in subs:
        cands |= {a for a in sp.preorder_traversal(s) if a.has(t) and not (a.free_symbols & state_like) and
                  isinstance(a, (sp.exp, sp.Pow, sp.Symbol, sp.sin, sp.cos))}
    for E in sorted(cands, key=lambda a: -sp.count_ops(a)):
        out = [sp.sympify(e).xreplace({E: tc}) for e in exprs]
        if not any(o.has(t) for o in out):
            return E, out
    return None, list(exprs)


def generate(symObjective, symStateRHS, nStates, nControls, params, bounds=None, want_control_char=True,
             allow_rows=True):
    """Derives the optimality system (make_from_symbolic.m:11-23) and writes the plugin source.

    symObjective: the integrand; symStateRHS: sequence of nStates expressions; params: dict name -> value (ordered: the
    parameter block of the plugin is its values in this order).  Returns a dict with
      source, has_control_char, row_separable, control_from_costate, params (ndarray), form ("rows" / "vector"),
      tcoef, cc_tcoef (the hoisted sub-expressions or None),
      H, adjointRHS, dHdu, ControlChar (SymPy, unclamped; ControlChar None if solve found no unique solution)."""
    nS, nC = int(nStates), int(nControls)
    names = list(params)
    t, x, lam, u, psym = symbols(nS, nC, names)
    f = [sp.sympify(e) for e in (symStateRHS if isinstance(symStateRHS, (list, tuple)) else list(symStateRHS))]
    if len(f) != nS:
        raise ValueError(f"symStateRHS has {len(f)} entries, nStates = {nS}")
    g = sp.sympify(symObjective)
    known = {t, *x, *u, *psym.values()}
    for e in [g, *f]:
        extra = e.free_symbols - known
        if extra:
            raise ValueError(f"unknown symbols {sorted(map(str, extra))}: use symbols(nStates, nControls, params)")
    H = g + sum(l * fi for l, fi in zip(lam, f))                       # :11
    adjointRHS = [-sp.diff(H, xi) for xi in x]                         # :14
    dHdu = [sp.diff(H, uj) for uj in u]                                # :17
    cc = None
    if want_control_char:                                              # :20-23
        try:
            sol = sp.solve(dHdu, u, dict=True)
        except NotImplementedError:
            sol = []
        if len(sol) == 1 and all(uj in sol[0] for uj in u) and not any(sol[0][uj].has(*u) for uj in u):
            cc = [sp.simplify(sol[0][uj]) for uj in u]

    state_like = {*x, *lam, *u}
    tc = sp.Symbol("tc", real=True)
    pmap = {psym[n]: sp.Symbol(f"p[{k}]") for k, n in enumerate(names)}

    # Gen-2 methods: columns of dF/dy' v and dF/du' v with v = [v_1 .. v_nS, v_last]
    v = [sp.Symbol(f"v[{i}]") for i in range(nS + 1)]
    dFdx = [sum(sp.diff(f[i], x[j]) * v[i] for i in range(nS)) + sp.diff(g, x[j]) * v[nS] for j in range(nS)]
    dFdu = [sum(sp.diff(f[i], u[j]) * v[i] for i in range(nS)) + sp.diff(g, u[j]) * v[nS] for j in range(nC)]

    # ---- row-separable form? ------------------------------------------------------------------------------------
    rows_ok = allow_rows and nC == 1 and nS in (1, 2, 4) and len(names) <= 16
    q = None
    if rows_ok:
        rows_ok = all(not f[i].has(*(x[:i] + x[i + 1:])) for i in range(nS))
    if rows_ok:
        zero = {xi: 0 for xi in x}
        try:
            base = g.xreplace(zero)
            q = []
            for i in range(nS):
                others = {xj: 0 for j, xj in enumerate(x) if j != i}
                q.append(g.xreplace(others) - base + (base if i == 0 else 0))
            rows_ok = sp.simplify(sum(q) - g) == 0 and all(sp.simplify(e).is_finite is not False for e in q)
        except Exception:
            rows_ok = False
    tcoef = cc_tcoef = None
    src = []
    src.append("// generated by optimal-control-solvers_amd/symbolic.py from symbolic f, g (functions/make_from_symbolic.m)\n"
               f"// parameters: [{', '.join(names)}]")
    if rows_ok:
        row_exprs = []
        for i in range(nS):
            row_exprs += [f[i], q[i], sp.diff(f[i], x[i]), sp.diff(q[i], x[i]), sp.diff(f[i], u[0]), sp.diff(q[i], u[0])]
        E, hoisted = _hoist(row_exprs, t, state_like, tc)
        if E is None:
            rows_ok = False
        else:
            tcoef = E
    control_from_costate = False
    if rows_ok:
        yy, uu = sp.Symbol("y"), sp.Symbol("u")

        def ren(e, i):
            return sp.sympify(e).xreplace({x[i]: yy, u[0]: uu, **pmap})

        def switch(kind, body_of):
            cases = "\n".join(f"    case {i}: {{\n{body_of(i)}\n      break;\n    }}" for i in range(nS))
            return f"  switch (r) {{\n{cases}\n    default: break;\n  }}"
        src.append(f"__device__ double ocs_row_tcoef(double t, OCS_PARAMS p) {{ return {_DevicePrinter().doprint(sp.sympify(tcoef).xreplace(pmap))}; }}")
        src.append("__device__ double ocs_row_F(double tc, double y, double u, OCS_PARAMS p, int r) {\n  double o = 0.0;\n" +
                   switch("F", lambda i: _emit([ren(hoisted[6 * i + 0], i)], ["o"], "      ")) + "\n  return o;\n}")
        src.append("__device__ double ocs_row_q(double tc, double y, double u, OCS_PARAMS p, int r) {\n  double o = 0.0;\n" +
                   switch("q", lambda i: _emit([ren(hoisted[6 * i + 1], i)], ["o"], "      ")) + "\n  return o;\n}")
        src.append("__device__ void ocs_row_dFdy(double tc, double y, double u, OCS_PARAMS p, int r, double* dF, double* dq) {\n" +
                   switch("dy", lambda i: _emit([ren(hoisted[6 * i + 2], i), ren(hoisted[6 * i + 3], i)], ["*dF", "*dq"], "      ")) + "\n}")
        src.append("__device__ void ocs_row_dFdu(double tc, double y, double u, OCS_PARAMS p, int r, double* dF, double* dq) {\n" +
                   switch("du", lambda i: _emit([ren(hoisted[6 * i + 4], i), ren(hoisted[6 * i + 5], i)], ["*dF", "*dq"], "      ")) + "\n}")
        dy_reads_u = any(sp.sympify(hoisted[6 * i + k]).has(u[0]) for i in range(nS) for k in (2, 3))
        control_from_costate = cc is not None and not any(c.has(*x) for c in cc) and not dy_reads_u
    else:
        ymap = {x[i]: sp.Symbol(f"y[{i}]") for i in range(nS)}
        umap = {u[j]: sp.Symbol(f"u[{j}]") for j in range(nC)}
        allm = [*f, g, *dFdx, *dFdu]
        E, hoisted = _hoist(allm, t, state_like | set(v), tc)
        targ = "t"
        if E is not None and E != t:
            tcoef = E
            src.append("#define OCS_USER_TCOEF 1\n__device__ double ocs_tcoef(double t, OCS_PARAMS p) { return "
                       f"{_DevicePrinter().doprint(sp.sympify(E).xreplace(pmap))}; }}")
            allm, targ = hoisted, "tc"

        def ren(e):
            return sp.sympify(e).xreplace({**ymap, **umap, **pmap})
        F_e, dx_e, du_e = allm[:nS + 1], allm[nS + 1:2 * nS + 1], allm[2 * nS + 1:]
        src.append(f"__device__ void ocs_F(double {targ}, const double* y, const double* u, OCS_PARAMS p, double* f) {{\n" +
                   _emit([ren(e) for e in F_e], [f"f[{i}]" for i in range(nS + 1)]) + "\n}")
        src.append(f"__device__ void ocs_dFdx_times_vec(double {targ}, const double* y, const double* u, OCS_PARAMS p, "
                   "const double* v, double* g) {\n" + _emit([ren(e) for e in dx_e], [f"g[{i}]" for i in range(nS)]) + "\n}")
        src.append(f"__device__ void ocs_dFdu_times_vec(double {targ}, const double* y, const double* u, OCS_PARAMS p, "
                   "const double* v, double* g) {\n" + _emit([ren(e) for e in du_e], [f"g[{j}]" for j in range(nC)]) + "\n}")
    if cc is not None:
        xmap = {x[i]: sp.Symbol(f"x[{i}]") for i in range(nS)}
        lmap = {lam[i]: sp.Symbol(f"lam[{i}]") for i in range(nS)}
        E, hoisted = _hoist(cc, t, state_like, tc)
        targ, exprs = "t", cc
        if E is not None and E != t:
            cc_tcoef = E
            src.append("#define OCS_USER_CC_TCOEF 1\n__device__ double ocs_cc_tcoef(double t, OCS_PARAMS p) { return "
                       f"{_DevicePrinter().doprint(sp.sympify(E).xreplace(pmap))}; }}")
            targ, exprs = "tc", hoisted
        body = _emit([sp.sympify(e).xreplace({**xmap, **lmap, **pmap}) for e in exprs], [f"const double s{j}" for j in range(nC)])
        clamp = "\n".join(f"  u[{j}] = fmin(ub[{j}], fmax(lb[{j}], s{j}));" for j in range(nC))   # :111
        src.append(f"__device__ void ocs_ControlChar(double {targ}, const double* x, const double* lam, OCS_PARAMS p, "
                   "const double* lb, const double* ub, double* u) {\n" + body + "\n" + clamp + "\n}")
    return {"source": "\n".join(src) + "\n", "has_control_char": cc is not None, "row_separable": bool(rows_ok),
            "control_from_costate": bool(control_from_costate), "form": "rows" if rows_ok else "vector",
            "params": np.asarray([float(params[n]) for n in names], dtype=np.float64), "param_names": names,
            "tcoef": tcoef, "cc_tcoef": cc_tcoef, "nS": nS, "nC": nC,
            "H": H, "adjointRHS": adjointRHS, "dHdu": dHdu, "ControlChar": cc, "objective": g, "stateRHS": f,
            "bounds": None if bounds is None else np.asarray(bounds, dtype=np.float64).reshape(nC, 2)}


class Gen1Functions:
    """The function handles of make_from_symbolic.m:33-38 as NumPy callables (lambdify in the place of matlabFunction):
    objective(t, x, u), stateRHS(t, x, u), adjointRHS(t, x, lam, u), dHdu(t, x, lam, u), ControlChar(t, x, lam) with the
    clamp of :111; x, lam, u are n x k arrays, t has k entries.  Host-side, for inspection and tests."""

    def __init__(self, gen):
        nS, nC, names = gen["nS"], gen["nC"], gen["param_names"]
        t, x, lam, u, psym = symbols(nS, nC, names)
        pv = [float(v) for v in gen["params"]]
        ps = [psym[n] for n in names]
        self._b = gen["bounds"]

        def fn(exprs, args):
            f = sp.lambdify([*args, *ps], list(exprs), "numpy")

            def call(*a):
                tt = np.atleast_1d(np.asarray(a[0], dtype=np.float64))
                cols = [np.atleast_2d(np.asarray(m, dtype=np.float64)) for m in a[1:]]
                flat = [tt] + [row for m in cols for row in m]
                out = f(*flat, *pv)
                return np.vstack([np.broadcast_to(np.asarray(o, dtype=np.float64), tt.shape) for o in out])
            return call
        self.objective = fn([gen["objective"]], [t, *x, *u])
        self.stateRHS = fn(gen["stateRHS"], [t, *x, *u])
        self.adjointRHS = fn(gen["adjointRHS"], [t, *x, *lam, *u])
        self.dHdu = fn(gen["dHdu"], [t, *x, *lam, *u])
        if gen["ControlChar"] is not None:
            raw = fn(gen["ControlChar"], [t, *x, *lam])

            def cc(tt, xx, ll):
                val = raw(tt, xx, ll)
                return val if self._b is None else np.minimum(self._b[:, 1:2], np.maximum(self._b[:, 0:1], val))
            self.ControlChar = cc
        self.ControlBounds = self._b


class NumpyTwin:
    """The Gen-2 plugin methods of OCProblem/OCProblem.m:8-21 -- F, dFdx_times_vec, dFdu_times_vec -- as NumPy callables built
    from the symbolic Jacobians with lambdify (columns vectorised, same signatures as oracle/np_twin's problem classes):
    what the generated device source is tested against (oracle/np_twin.RK4IntegratorNP integrates it)."""

    def __init__(self, gen):
        nS, nC, names = gen["nS"], gen["nC"], gen["param_names"]
        t, x, lam, u, psym = symbols(nS, nC, names)
        self.nS, self.nC = nS, nC
        pv = [float(v) for v in gen["params"]]
        ps = [psym[n] for n in names]
        f, g = gen["stateRHS"], gen["objective"]
        args = [t, *x, *u, *ps]
        self._F = sp.lambdify(args, [*f, g], "numpy")
        self._Jx = sp.lambdify(args, [[sp.diff(e, xj) for xj in x] for e in [*f, g]], "numpy")   # (nS+1) x nS
        self._Ju = sp.lambdify(args, [[sp.diff(e, uj) for uj in u] for e in [*f, g]], "numpy")   # (nS+1) x nC
        self._pv = pv

    def _call(self, fn, t, y, u):
        t = np.atleast_1d(np.asarray(t, dtype=np.float64))
        y, u = np.atleast_2d(y), np.atleast_2d(u)
        return t, fn(t, *[y[i] for i in range(self.nS)], *[u[j] for j in range(self.nC)], *self._pv)

    def F(self, t, y, u):
        t, out = self._call(self._F, t, y, u)
        return np.vstack([np.broadcast_to(np.asarray(o, dtype=np.float64), t.shape) for o in out])

    def _contract(self, M, v, t, ncols):
        out = np.zeros((ncols, t.size))
        for i in range(self.nS + 1):
            for j in range(ncols):
                out[j] += np.broadcast_to(np.asarray(M[i][j], dtype=np.float64), t.shape) * v[i]
        return out

    def dFdx_times_vec(self, t, y, u, v):
        t, M = self._call(self._Jx, t, y, u)
        g = self._contract(M, np.atleast_2d(v), t, self.nS)
        return np.vstack([g, np.zeros((1, t.size))])

    def dFdu_times_vec(self, t, y, u, v):
        t, M = self._call(self._Ju, t, y, u)
        return self._contract(M, np.atleast_2d(v), t, self.nC)


def make_from_symbolic(symObjective, symStateRHS, nStates, nControls, params, bounds, create=True, **kw):
    """prob = make_from_symbolic(symObjective, symStateRHS, nStates, nControls, params, bounds)   make_from_symbolic.m:1-2

    Returns a UserProblem compiled from the generated source (needs the GPU library; `create=False` returns the
    generation dict only).  The object carries `.generated` (that dict) and `.gen1` (the reference's function handles as
    NumPy callables)."""
    gen = generate(symObjective, symStateRHS, nStates, nControls, params, bounds, **kw)
    if not create:
        return gen
    from .problem import UserProblem
    prob = UserProblem(gen["source"], gen["nS"], gen["nC"], gen["params"], gen["bounds"],
                       has_control_char=gen["has_control_char"], row_separable=gen["row_separable"],
                       control_from_costate=gen["control_from_costate"])
    prob.generated = gen
    prob.gen1 = Gen1Functions(gen)
    prob.numpy_twin = NumpyTwin(gen)
    return prob
