"""The batch axis over the GPUs of one node behind the C-ABI (include/ocs.h, ocs_multi_*): a single process
hands whole batches to MultiDevice, which cuts them into contiguous blocks, runs the one-device entry points on every
device concurrently (no data-path exchange) and reduces [sum J, count] / (min J, argmin) over RCCL.

The reference has no batch axis (tests/solve_test_problem.m:37 integrates one trajectory per call); arrays here are the
MATLAB-shaped host arrays of the one-device calls with the batch as trailing dimension."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from .problem import _f, _p


def _harr(objs):
    return (C.c_void_p * len(objs))(*[o._h for o in objs])


def _parr(tensors):
    """array of device pointers, one per device (None -> NULL array)"""
    if tensors is None:
        return None
    return (C.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class MultiDevice:
    def __init__(self, devices=None, n=None):
        """devices: list of HIP device ids (default: the first n, default all)."""
        if devices is None:
            cnt = C.c_int()
            check(lib.ocs_device_count(C.byref(cnt)))
            devices = list(range(n or cnt.value))
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        self._h = C.c_void_p()
        check(lib.ocs_multi_create(C.byref(self._h), arr, len(self.devices)))

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:   # (lib is None while the interpreter shuts down)
            lib.ocs_multi_destroy(self._h)
            self._h = None

    @property
    def size(self):
        return lib.ocs_multi_size(self._h)

    @property
    def has_communicator(self):
        """True: the reductions run over RCCL; False: on the host (librccl missing / OCS_MULTI_NO_RCCL=1 / init failure)."""
        return check(lib.ocs_multi_has_communicator(self._h)) == 1

    def stream(self, k):
        """hipStream_t (as int) the kernels of block k are enqueued on."""
        st = C.c_void_p()
        check(lib.ocs_multi_stream(self._h, k, C.byref(st)))
        return st.value or 0

    def synchronize(self):
        check(lib.ocs_multi_synchronize(self._h))

    def stats(self):
        """The reductions a *_dev call with reduce=True enqueued: waits for them."""
        st = np.empty(4)
        check(lib.ocs_multi_stats(self._h, _p(st)))
        return self._stats(st)

    # ---- device-resident blocks: lists of torch tensors, entry k on device self.devices[k], batch-minor -----------------
    @staticmethod
    def _counts(tensors):
        return (C.c_int * len(tensors))(*[int(t.shape[-1]) for t in tensors])

    def compute_states_dev(self, integs, probs, x0, u, x, J, reduce=False):
        """x0[k] [nS][B_k], u[k] [2N+1][nC][B_k] -> x[k] [N+1][nAug][B_k] (or None), J[k] [B_k]; asynchronous."""
        return check(lib.ocs_multi_compute_states_dev(self._h, _harr(integs), _harr(probs), self._counts(x0), _parr(x0),
                                                      _parr(u), _parr(x), _parr(J), int(bool(reduce))))

    def compute_adjoints_dev(self, integs, probs, u, lam, dJdu, lamT=None):
        return check(lib.ocs_multi_compute_adjoints_dev(self._h, _harr(integs), _harr(probs), self._counts(u), _parr(u),
                                                        _parr(lamT), _parr(lam), _parr(dJdu)))

    def nlp_objective_dev(self, integs, probs, ctrls, x0, v, J, dJdv, FreeInitStates=(), reduce=False):
        fis = (C.c_int * len(FreeInitStates))(*FreeInitStates) if len(FreeInitStates) else None
        return check(lib.ocs_multi_nlp_objective_dev(self._h, _harr(integs), _harr(probs), _harr(ctrls), self._counts(v),
                                                     _parr(x0), _parr(v), len(FreeInitStates), fis, _parr(J), _parr(dJdv),
                                                     int(bool(reduce))))

    def fb_sweep_dev(self, integs, probs, x0, options=None, reduce=False):
        """One fb_sweep per device block, concurrently; returns per-device dicts of tensors as sweep.fb_sweep_dev."""
        import torch
        from .sweep import _options
        o, _ = _options(options)
        outs = []
        for k, x0k in enumerate(x0):
            N, B, dev = integs[k].nSTEPS, x0k.shape[-1], x0k.device
            nS, nC = probs[k].nS, probs[k].nC
            outs.append({"xaug": torch.empty((N + 1, nS + 1, B), dtype=torch.float64, device=dev),
                         "lam": torch.empty((N + 1, nS, B), dtype=torch.float64, device=dev),
                         "u": torch.empty((o.nINTERP_PTS, nC, B), dtype=torch.float64, device=dev),
                         "J": torch.empty(B, dtype=torch.float64, device=dev),
                         "sweeps": torch.zeros(B, dtype=torch.int32, device=dev),
                         "maxChange": torch.empty((o.nSWEEPS, B), dtype=torch.float64, device=dev)})
        import torch as _t
        _t.cuda.synchronize()   # the outputs were allocated on torch's stream, the kernels run on the handle's streams
        rc = check(lib.ocs_multi_fb_sweep_dev(self._h, _harr(integs), _harr(probs), self._counts(x0), _parr(x0), C.byref(o),
                                              *[_parr([r[n] for r in outs]) for n in ("xaug", "lam", "u", "J", "sweeps", "maxChange")],
                                              int(bool(reduce))))
        for r in outs:
            r["status"] = rc
        return outs

    def shard(self, batch, k):
        lo, hi = C.c_int(), C.c_int()
        check(lib.ocs_multi_shard(self._h, batch, k, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def replicate(self, make):
        """One handle per device: make() is called with that device current (ocs_set_device)."""
        out = []
        for d in self.devices:
            check(lib.ocs_set_device(d))
            out.append(make())
        check(lib.ocs_set_device(self.devices[0]))
        return out

    @staticmethod
    def _stats(s):
        return {"sum_J": s[0], "count": int(s[1]), "min_J": s[2], "argmin": int(s[3])}

    def compute_states(self, integs, probs, x0, u, want_x=True):
        """[x, J] = compute_states(obj, prob, x0, u) (RK4Integrator.m:28-56): x0 nS x B, u nC x (2N+1) x B."""
        x0, u = _f(np.atleast_2d(x0)), _f(u)
        B, nS, N = u.shape[-1], probs[0].nS, integs[0].nSTEPS
        x = np.empty((nS + 1, N + 1, B), order="F") if want_x else None
        J, st = np.empty(B), np.empty(4)
        rc = check(lib.ocs_multi_compute_states(self._h, _harr(integs), _harr(probs), B, _p(x0), _p(u),
                                                _p(x) if want_x else None, _p(J), _p(st)))
        return x, J, self._stats(st), rc

    def compute_adjoints(self, integs, probs, u, lamT=None, want_dJdu=True):
        u = _f(u)
        B, nS, N, nC = u.shape[-1], probs[0].nS, integs[0].nSTEPS, u.shape[0]
        lam = np.empty((nS + 1, N + 1, B), order="F")
        dJdu = np.empty((nC, 2 * N + 1, B), order="F") if want_dJdu else None
        lt = _f(lamT) if lamT is not None else None
        check(lib.ocs_multi_compute_adjoints(self._h, _harr(integs), _harr(probs), B, _p(u), _p(lt) if lt is not None else None,
                                             _p(lam), _p(dJdu) if want_dJdu else None))
        return lam, dJdu

    def nlp_objective(self, integs, probs, ctrls, x0, v, FreeInitStates=()):
        """[J, dJdv] = nlpObjective(v) (single_shooting.m:137-150): v (nV + nFree) x B, x0 nS x B (updated in place)."""
        x0, v = _f(np.atleast_2d(x0)), _f(np.atleast_2d(v))
        B = v.shape[1]
        fis = (C.c_int * len(FreeInitStates))(*FreeInitStates) if len(FreeInitStates) else None
        J, dJdv, st = np.empty(B), np.empty_like(v), np.empty(4)
        rc = check(lib.ocs_multi_nlp_objective(self._h, _harr(integs), _harr(probs), _harr(ctrls), B, _p(x0), _p(v),
                                               len(FreeInitStates), fis, _p(J), _p(dJdv), _p(st)))
        return J, dJdv, x0, self._stats(st), rc

    def fb_sweep(self, integs, probs, x0, options=None):
        """soln = fb_sweep(prob, x0, tspan, options) (fb_sweep.m) for a batch: sample arrays as sweep.fb_sweep_batch."""
        from .sweep import _options
        o, _ = _options(options)
        x0 = _f(np.atleast_2d(x0))
        B, nS, N = x0.shape[1], probs[0].nS, integs[0].nSTEPS
        nC = probs[0].ControlBounds.shape[0]
        x, lam = np.empty((nS, N + 1, B), order="F"), np.empty((nS, N + 1, B), order="F")
        uI = np.empty((nC, o.nINTERP_PTS, B), order="F")
        J, sweeps, st = np.empty(B), np.zeros(B, dtype=np.int32), np.empty(4)
        mc = np.empty((o.nSWEEPS, B), order="F")
        rc = check(lib.ocs_multi_fb_sweep(self._h, _harr(integs), _harr(probs), B, _p(x0), C.byref(o), None, None, _p(x),
                                          _p(lam), _p(uI), _p(J), sweeps.ctypes.data_as(_lib.ip), _p(mc), _p(st)))
        return {"x": x, "lam": lam, "u": uI, "J": J, "sweeps": sweeps, "maxChange": mc, "status": rc, "stats": self._stats(st)}
