"""ctypes binding of oracle/libsofacontrol_cpu.so -- the native CPU twin of the hot path (csrc/sofacontrol_cpu.cpp).

Test / baseline infrastructure like everything under oracle/: used by tests/test_cpu_twin.py and by the cpu_baseline
leg of bench.py; the product package never loads it."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libsofacontrol_cpu.so')
_dp = C.POINTER(C.c_double)


class _Problem(C.Structure):
    _fields_ = [('N', C.c_int), ('n_x', C.c_int), ('n_u', C.c_int), ('n_z', C.c_int),
                ('H', _dp), ('Qz', _dp), ('R', _dp), ('Qzf', _dp), ('x_scale', _dp),
                ('nU', C.c_int), ('UA', _dp), ('Ub', _dp), ('nX', C.c_int), ('XA', _dp), ('Xb', _dp),
                ('nXf', C.c_int), ('XfA', _dp), ('Xfb', _dp), ('tr_active', C.c_int)]


class _Model(C.Structure):
    _fields_ = [('P', C.c_int), ('r', C.c_int), ('m', C.c_int), ('w_q', C.c_double), ('w_v', C.c_double),
                ('q', _dp), ('v', _dp), ('Ac', _dp), ('Bc', _dp), ('dc', _dp), ('Ad', _dp), ('Bd', _dp), ('dd', _dp)]


class _GParams(C.Structure):
    _fields_ = [('delta0', C.c_double), ('omega0', C.c_double), ('rho', C.c_double), ('beta_fail', C.c_double),
                ('gamma_fail', C.c_double), ('epsilon', C.c_double), ('omega_max', C.c_double), ('convg_thresh', C.c_double),
                ('max_gusto_iters', C.c_int)]


_lib = None


def build():
    subprocess.check_call(['make', '-C', _HERE, '-s'])


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        if _lib.scpu_version() < 3:          # a stale build from an earlier round
            build()
            _lib = C.CDLL(LIB_PATH)
    return _lib


def _a(x):
    return None if x is None else np.ascontiguousarray(x, dtype=np.float64)


def _p(x):
    return None if x is None else x.ctypes.data_as(_dp)


def _problem(N, H, Qz, R, Qzf=None, U=None, X=None, Xf=None, x_scale=None, tr_active=True):
    keep = [_a(H), _a(Qz), _a(R), _a(Qzf), _a(x_scale)]
    polys = []
    for t in (U, X, Xf):
        A, b = (None, None) if t is None else (_a(t[0]), _a(t[1]))
        polys += [A, b]
    keep += polys
    nz, n = keep[0].shape
    pr = _Problem(N, n, keep[2].shape[0], nz, _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]), _p(keep[4]),
                  0 if polys[0] is None else polys[0].shape[0], _p(polys[0]), _p(polys[1]),
                  0 if polys[2] is None else polys[2].shape[0], _p(polys[2]), _p(polys[3]),
                  0 if polys[4] is None else polys[4].shape[0], _p(polys[4]), _p(polys[5]), 1 if tr_active else 0)
    return pr, keep


def project(U, ref, X, threads=1):
    U, ref, X = _a(U), _a(ref), _a(X)
    out = np.empty((X.shape[0], U.shape[1]))
    lib().scpu_project(_p(U), C.c_int64(U.shape[0]), C.c_int(U.shape[1]), _p(ref), _p(X), C.c_int64(X.shape[0]), _p(out), C.c_int(threads))
    return out


ALGO = {'riccati': 0, 'condensed': 1}


def locp_solve(N, H, Qz, R, Ad, Bd, dd, x0, xk, delta, omega, z=None, zf=None, u_des=None, Qzf=None, U=None, X=None, Xf=None,
               x_scale=None, tr_active=True, algo='riccati'):
    """algo 'riccati': stage-wise Riccati interior point (oracle/riccati_ipm.py) with the trust-region prescreen;
    'condensed': the device kernel's control flow -- condensed interior point (oracle/condensed_ipm.py) for the QP without
    its trust-region rows, the Riccati interior point of the full QP when that minimiser leaves the trust region."""
    pr, keep = _problem(N, H, Qz, R, Qzf, U, X, Xf, x_scale, tr_active)
    Ad, Bd, dd, x0, xk, z, zf, u_des = map(_a, (Ad, Bd, dd, x0, xk, z, zf, u_des))
    n, m = Bd.shape[1], Bd.shape[2]
    x = np.empty((N + 1, n)); u = np.empty((N, m)); s = np.empty(N + 1)
    J = C.c_double(); it = C.c_int()
    st = lib().scpu_locp_solve_algo(C.byref(pr), _p(Ad), _p(Bd), _p(dd), _p(x0), _p(xk), C.c_double(delta), C.c_double(omega), _p(z),
                                    _p(zf), _p(u_des), _p(x), _p(u), _p(s), C.byref(J), C.byref(it), C.c_int(ALGO[algo]))
    return x, u, s, J.value, dict(status=st, iters=it.value)


def gusto_solve(model, Ad, Bd, dd, H, N, dt, Qz, R, x0, u_init, x_init, z=None, zf=None, u_des=None, Qzf=None, U=None, X=None,
                Xf=None, x_char=None, f_char=None, threads=1, max_trace=0, algo='riccati', **kw):
    """oracle.gusto.solve for a batch (leading axis) of rollouts; returns xopt, uopt, iters, trace.  `algo` as locp_solve."""
    from .gusto import DEFAULTS
    par = dict(DEFAULTS); par.update(kw)
    pr, keep = _problem(N, H, Qz, R, Qzf, U, X, Xf, None, True)
    t = [_a(model[k]) for k in ('q', 'v', 'A_c', 'B_c', 'd_c')] + [_a(Ad), _a(Bd), _a(dd)]
    mo = _Model(t[0].shape[0], t[0].shape[1], t[3].shape[2], float(model['w_q']), float(model['w_v']), *[_p(a) for a in t])
    gp = _GParams(par['delta0'], par['omega0'], par['rho'], par['beta_fail'], par['gamma_fail'], par['epsilon'], par['omega_max'],
                  par['convg_thresh'], int(par['max_gusto_iters']))
    x0, u_init, x_init, z, zf, u_des, x_char, f_char = map(_a, (x0, u_init, x_init, z, zf, u_des, x_char, f_char))
    if x0.ndim == 1:
        x0, u_init, x_init = x0[None], u_init[None], x_init[None]
        z = None if z is None else z[None]
        zf = None if zf is None else zf[None]
        u_des = None if u_des is None else u_des[None]
    B, n = x0.shape
    m = u_init.shape[2]
    xo = np.empty((B, N + 1, n)); uo = np.empty((B, N, m)); iters = np.empty(B, dtype=np.int32)
    trace = np.full((B, max(1, max_trace), 4), np.nan)
    lib().scpu_gusto_solve_algo(C.byref(mo), C.byref(pr), C.byref(gp), C.c_double(dt), C.c_int64(B), _p(x0), _p(u_init), _p(x_init), _p(z),
                                _p(zf), _p(u_des), _p(x_char), _p(f_char), _p(xo), _p(uo), iters.ctypes.data_as(C.POINTER(C.c_int32)),
                                _p(trace) if max_trace > 0 else None, C.c_int(max_trace), C.c_int(threads), C.c_int(ALGO[algo]))
    return xo, uo, iters, trace


class _IlqrParams(C.Structure):
    _fields_ = [('max_iter', C.c_int), ('epsilon', C.c_double), ('alpha0', C.c_double), ('alpha_scaling', C.c_double), ('improv_lb', C.c_double),
                ('improv_ub', C.c_double), ('alpha_min', C.c_double), ('counter_limit', C.c_int), ('rho0', C.c_double), ('drho0', C.c_double),
                ('rho_scaling', C.c_double), ('rho_increase_fp', C.c_double), ('rho_max', C.c_double), ('rho_min', C.c_double),
                ('include_input_var_constraint', C.c_int), ('do_linesearch', C.c_int), ('regularize', C.c_int), ('state_regularization', C.c_int)]


def _ilqr_params(**kw):
    from .lqr import ILQRParams as P
    g = lambda k: kw.get(k, getattr(P, k))
    return _IlqrParams(int(g('max_iter')), g('epsilon'), g('alpha0'), g('alpha_scaling'), g('improv_lb'), g('improv_ub'), g('alpha_min'),
                       int(g('counter_limit')), g('rho0'), g('drho0'), g('rho_scaling'), g('rho_increase_fp'), g('rho_max'), g('rho_min'),
                       int(bool(g('include_input_var_constraint'))), int(bool(g('do_linesearch'))), int(bool(g('regularize'))),
                       int(bool(g('state_regularization'))))


def _ilqr_out(B, N, n, m):
    return (np.empty((B, N + 1, n)), np.empty((B, N, m)), np.empty((B, N, m, n)), np.empty(B), np.empty(B, dtype=np.int32))


def ilqr_tpwl(model, Ad, Bd, dd, H, z_ref, Q, R, Qf, N, x0, z_target, u_warm=None, u_last=None, threads=1, **params):
    """oracle.lqr.ILQR.solve for a batch (leading axis) of problems on the nearest-point TPWL model; returns x, u, K, cost, iters."""
    t = [_a(model[k]) for k in ('q', 'v', 'A_c', 'B_c', 'd_c')] + [_a(Ad), _a(Bd), _a(dd)]
    mo = _Model(t[0].shape[0], t[0].shape[1], t[3].shape[2], float(model['w_q']), float(model['w_v']), *[_p(a) for a in t])
    H, z_ref, Q, R, Qf, x0, z_target, u_warm, u_last = map(_a, (H, z_ref, Q, R, Qf, x0, z_target, u_warm, u_last))
    if x0.ndim == 1:
        x0, z_target = x0[None], z_target[None]
        u_warm = None if u_warm is None else u_warm[None]
        u_last = None if u_last is None else u_last[None]
    B, n = x0.shape
    m = t[3].shape[2]
    x, u, K, cost, iters = _ilqr_out(B, N, n, m)
    par = _ilqr_params(**params)
    lib().scpu_ilqr_tpwl(C.byref(mo), _p(H), _p(z_ref), C.c_int(H.shape[0]), _p(Q), _p(R), _p(Qf), C.byref(par), C.c_int(N), C.c_int64(B), _p(x0),
                         _p(z_target), _p(u_warm), _p(u_last), _p(x), _p(u), _p(K), _p(cost), iters.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int(threads))
    return x, u, K, cost, iters


SSM_MODES = {'fe': 1, 'be': 2, 'bil': 3}


def ilqr_ssm(n, m, rom_order, ssm_order, r_coeff, Bc, w_coeff, z_ref, H, method, dt, Q, R, Qf, N, x0, z_target, u_warm=None, u_last=None,
             threads=1, **params):
    """oracle.lqr.ILQRGeneric.solve over oracle.ssm (continuous model discretised per `method`) for a batch of problems."""
    r_coeff, Bc, w_coeff, z_ref, H, Q, R, Qf, x0, z_target, u_warm, u_last = map(_a, (r_coeff, Bc, w_coeff, z_ref, H, Q, R, Qf, x0, z_target, u_warm, u_last))
    if x0.ndim == 1:
        x0, z_target = x0[None], z_target[None]
        u_warm = None if u_warm is None else u_warm[None]
        u_last = None if u_last is None else u_last[None]
    B = x0.shape[0]
    no = w_coeff.shape[0]
    x, u, K, cost, iters = _ilqr_out(B, N, n, m)
    par = _ilqr_params(**params)
    lib().scpu_ilqr_ssm(C.c_int(n), C.c_int(m), C.c_int(no), C.c_int(rom_order), C.c_int(ssm_order), _p(r_coeff), _p(Bc), _p(w_coeff), _p(z_ref),
                        _p(H), C.c_int(SSM_MODES[method]), C.c_double(dt), _p(Q), _p(R), _p(Qf), C.byref(par), C.c_int(N), C.c_int64(B), _p(x0),
                        _p(z_target), _p(u_warm), _p(u_last), _p(x), _p(u), _p(K), _p(cost), iters.ctypes.data_as(C.POINTER(C.c_int32)),
                        C.c_int(threads))
    return x, u, K, cost, iters
