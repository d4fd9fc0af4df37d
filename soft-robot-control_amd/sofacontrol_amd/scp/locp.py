"""LOCP: the horizon QP of GuSTO, solved on the device -- surface of sofacontrol/scp/locp.py:9-203.

The reference builds the QP in cvxpy and hands it to OSQP/GUROBI; here `solve()` launches the
Riccati-structured interior-point kernel (csrc/locp_dev.h).  Same problem data (locp.py:218-342), same
`update / solve / get_solution` protocol, `J*` without the 1/2 factor like cvxpy reports it."""
import ctypes as C
import os
import time

import numpy as np

from .. import _lib


class _Stats:
    def __init__(self, solve_time, iters):
        self.solve_time = solve_time
        self.num_iters = iters


def _poly(p):
    if p is None:
        return 0, None, None
    A = np.ascontiguousarray(p.A, dtype=np.float64)
    b = np.ascontiguousarray(p.b, dtype=np.float64)
    return A.shape[0], A, b


def make_problem(N, H, Qz, R, Qzf=None, U=None, X=None, Xf=None, dU=None, x_scale=None, tr_active=True):
    """Fill a slocp_problem (include/sofacontrol_hip.h); returns (struct, keepalive list)."""
    H = _lib.f64(H); Qz = _lib.f64(Qz); R = _lib.f64(R)
    Qzf = _lib.f64(Qzf)
    xs = _lib.f64(x_scale)
    nU, UA, Ub = _poly(U)
    nX, XA, Xb = _poly(X)
    nXf, XfA, Xfb = _poly(Xf)
    ndU, dUA, dUb = _poly(dU)
    p = _lib.SLocpProblem(N, H.shape[1], R.shape[0], Qz.shape[0], _lib.dptr(H), _lib.dptr(Qz), _lib.dptr(R),
                          _lib.dptr(Qzf), _lib.dptr(xs), nU, _lib.dptr(UA), _lib.dptr(Ub), nX, _lib.dptr(XA),
                          _lib.dptr(Xb), nXf, _lib.dptr(XfA), _lib.dptr(Xfb), ndU, _lib.dptr(dUA), _lib.dptr(dUb),
                          1 if tr_active else 0)
    return p, [H, Qz, R, Qzf, xs, UA, Ub, XA, Xb, XfA, Xfb, dUA, dUb]


def _rate_start(dU):
    """The increment e_1 the augmented dynamics start from (u_0 has no predecessor in the reference's constraint set,
    locp.py:305-308, so e_1 is a constant of the QP and only has to satisfy dU.A e_1 <= dU.b with room to spare): zero
    when that is feasible, otherwise the Chebyshev centre of the rate polyhedron (one small LP at construction)."""
    A, b = np.asarray(dU.A, dtype=np.float64), np.asarray(dU.b, dtype=np.float64)
    m = A.shape[1]
    if np.all(b >= 0):
        return np.zeros(m)
    from scipy.optimize import linprog
    nrm = np.linalg.norm(A, axis=1)
    res = linprog(np.concatenate((np.zeros(m), [-1.0])), A_ub=np.hstack((A, nrm[:, None])), b_ub=b,
                  bounds=[(None, None)] * m + [(0.0, 1e6)], method='highs')
    if res.status != 0 or res.x[-1] <= 0:
        raise ValueError('dU: the rate polyhedron dU.A e <= dU.b has no interior point')
    return res.x[:m]


def _rate_dynamics(Aa, Ba, da, off, m, e1):
    """p_{k+1} = u_k, e_{k+1} = u_k - p_k (k >= 1), e_1 = e1, for the extra states [p; e] at offset `off`."""
    I = np.eye(m)
    Ba[:, off:off + m] = I
    Ba[1:, off + m:off + 2 * m] = I
    Aa[1:, off + m:off + 2 * m, off:off + m] = -I
    da[0, off + m:off + 2 * m] = e1


class LOCP:
    def __init__(self, N, H, Qz, R, Qzf=None, U=None, X=None, Xf=None, dU=None, verbose=False, warm_start=True,
                 x_char=None, **kwargs):
        self.N = N
        self.H = np.asarray(H, dtype=np.float64)
        self.Qz, self.R, self.Qzf = Qz, R, Qzf
        self.U, self.X, self.Xf, self.dU = U, X, Xf, dU
        self.verbose = verbose
        self.warm_start = warm_start
        self.nonlinear_observer = kwargs.pop('nonlinear_observer', False)
        self.n_x = self.H.shape[1]
        self.n_z = Qz.shape[0]
        self.n_u = R.shape[0]
        self.x_scale = np.ones(self.n_x) if x_char is None else 1. / np.abs(x_char)
        self.tr_active = kwargs.pop('is_tr_active', True)
        self._du_aug = False
        self._init_nullspace(kwargs.pop('input_nullspace', None))
        if self.nonlinear_observer:
            self._init_augmented(N, Qz, R, Qzf, U, X, Xf, dU, kwargs)
            return
        if dU is not None:
            self.solver_args = kwargs
            self._init_rate_augmented(N, Qz, R, Qzf, U, X, Xf, dU)
            return
        self.solver_args = kwargs      # OSQP/GUROBI settings have no meaning here; kept for signature parity
        self._prob, self._keep = make_problem(N, self.H, Qz, R, Qzf, U, X, Xf, dU, self.x_scale, self.tr_active)
        self._data = None
        self._sol = None

    def _init_augmented(self, N, Qz, R, Qzf, U, X, Xf, dU, kwargs):
        """Nonlinear-observer branch (locp.py:84-88, 231-245, 312-329): z_k = Hd_k x_k + cd_k with per-stage
        (Hd_k, cd_k).  The kernel's QP has one constant performance matrix, so the stage outputs are carried
        as extra states: xa_k = [x_k; zeta_k], zeta_{k+1} = Hd_{k+1} (A_k x_k + B_k u_k + d_k) + cd_{k+1},
        H_a = [0 I], X_a = [0 X.A], Xf_a = [Xf.A 0], zero trust-region scale on zeta -- the same QP in
        (x, u, s) after eliminating zeta."""
        from ..utils import Polyhedron
        if Qzf is not None and np.any(self.H != 0):
            raise NotImplementedError('terminal cost through a non-zero constant H together with a nonlinear '
                                      'observer (locp.py:251-252) is not covered')
        self.solver_args = kwargs
        n, nz, m = self.n_x, self.n_z, self.n_u
        ne = 0 if dU is None else 2 * m                 # + [u_prev; du] for input-rate rows (see _init_rate_augmented)
        Ha = np.hstack((np.zeros((nz, n)), np.eye(nz), np.zeros((nz, ne))))
        rows, rhs = [], []
        if X is not None:
            rows.append(np.hstack((np.zeros((X.A.shape[0], n)), X.A, np.zeros((X.A.shape[0], ne))))); rhs.append(np.asarray(X.b, dtype=np.float64))
        if dU is not None:
            rows.append(np.hstack((np.zeros((dU.A.shape[0], n + nz + m)), np.asarray(dU.A, dtype=np.float64)))); rhs.append(np.asarray(dU.b, dtype=np.float64))
            self._e1 = _rate_start(dU)
        Xa = Polyhedron(np.vstack(rows), np.concatenate(rhs)) if rows else None
        Xfa = None if Xf is None else Polyhedron(np.hstack((Xf.A, np.zeros((Xf.A.shape[0], nz + ne)))), Xf.b)
        xs = np.concatenate((self.x_scale, np.zeros(nz + ne)))
        self._ne = ne
        self._prob, self._keep = make_problem(N, Ha, Qz, R, None, U, Xa, Xfa, None, xs, self.tr_active)
        self._data = None
        self._sol = None

    def _init_rate_augmented(self, N, Qz, R, Qzf, U, X, Xf, dU):
        """Input-rate constraints dU.A (u_{k+1} - u_k) <= dU.b, k = 0..N-2 (locp.py:305-308) couple consecutive
        stages; the stage-structured kernel takes pure state rows and pure input rows.  The previous input and the
        input increment are carried as extra states: xa_k = [x_k; p_k; e_k], p_{k+1} = u_k, e_{k+1} = u_k - p_k
        (e_1 = 0: u_0 has no predecessor in the reference's constraint set), and the rate rows become the state
        rows [0 0 dU.A] xa_k <= dU.b on k = 1..N -- the same QP in (x, u, s).  The row at k = 1 acts on the constant e_1
        (_rate_start: zero, or an interior point of the rate polyhedron when dU.b has negative entries)."""
        from ..utils import Polyhedron
        self._e1 = _rate_start(dU)
        n, m, nz = self.n_x, self.n_u, self.n_z
        na = n + 2 * m
        Ha = np.hstack((self.H, np.zeros((nz, 2 * m))))
        rows = [np.hstack((np.zeros((dU.A.shape[0], n + m)), np.asarray(dU.A, dtype=np.float64)))]
        rhs = [np.asarray(dU.b, dtype=np.float64)]
        if X is not None:
            rows.insert(0, np.hstack((np.asarray(X.A, dtype=np.float64), np.zeros((X.A.shape[0], 2 * m)))))
            rhs.insert(0, np.asarray(X.b, dtype=np.float64))
        Xa = Polyhedron(np.vstack(rows), np.concatenate(rhs))
        Xfa = None if Xf is None else Polyhedron(np.hstack((Xf.A, np.zeros((Xf.A.shape[0], 2 * m)))), Xf.b)
        xs = np.concatenate((self.x_scale, np.zeros(2 * m)))
        self._du_aug = True
        self._na = na
        self._prob, self._keep = make_problem(N, Ha, Qz, R, Qzf, U, Xa, Xfa, None, xs, self.tr_active)
        self._data = None
        self._sol = None

    def _update_rate_augmented(self, Ad, Bd, dd, x0, xk, z, zf, u):
        N, n, m = self.N, self.n_x, self.n_u
        na = self._na
        Ad = np.asarray(Ad).reshape(N, n, n); Bd = np.asarray(Bd).reshape(N, n, m); dd = np.asarray(dd).reshape(N, n)
        Aa = np.zeros((N, na, na)); Ba = np.zeros((N, na, m)); da = np.zeros((N, na))
        Aa[:, :n, :n] = Ad
        Ba[:, :n] = Bd
        da[:, :n] = dd
        _rate_dynamics(Aa, Ba, da, n, m, self._e1)
        x0 = np.asarray(x0).reshape(n)
        xka = np.zeros((N + 1, na))
        if xk is not None:
            xka[:, :n] = np.asarray(xk).reshape(N + 1, n)
        self._data = dict(Ad=_lib.f64(Aa), Bd=_lib.f64(Ba), dd=_lib.f64(da),
                          x0=_lib.f64(np.concatenate((x0, np.zeros(2 * m)))), xk=_lib.f64(xka),
                          z=None if z is None else _lib.f64(np.ravel(z)),
                          zf=None if (self.Qzf is None or zf is None) else _lib.f64(zf),
                          u=None if u is None else _lib.f64(np.ravel(u)))

    def _update_augmented(self, Ad, Bd, dd, x0, xk, z, u, Hd, cd):
        N, n, m, nz = self.N, self.n_x, self.n_u, self.n_z
        Ad = np.asarray(Ad).reshape(N, n, n); Bd = np.asarray(Bd).reshape(N, n, m); dd = np.asarray(dd).reshape(N, n)
        Hd = np.asarray(Hd).reshape(N + 1, nz, n); cd = np.asarray(cd).reshape(N + 1, nz)
        ne = self._ne
        na = n + nz + ne
        Aa = np.zeros((N, na, na)); Ba = np.zeros((N, na, m)); da = np.zeros((N, na))
        Aa[:, :n, :n] = Ad
        Aa[:, n:n + nz, :n] = np.einsum('kij,kjl->kil', Hd[1:], Ad)
        Ba[:, :n] = Bd
        Ba[:, n:n + nz] = np.einsum('kij,kjl->kil', Hd[1:], Bd)
        da[:, :n] = dd
        da[:, n:n + nz] = np.einsum('kij,kj->ki', Hd[1:], dd) + cd[1:]
        if ne:
            _rate_dynamics(Aa, Ba, da, n + nz, m, self._e1)
        x0 = np.asarray(x0).reshape(n)
        xk = np.asarray(xk).reshape(N + 1, n)
        self._data = dict(Ad=_lib.f64(Aa), Bd=_lib.f64(Ba), dd=_lib.f64(da),
                          x0=_lib.f64(np.concatenate((x0, Hd[0] @ x0 + cd[0], np.zeros(ne)))),
                          xk=_lib.f64(np.hstack((xk, np.einsum('kij,kj->ki', Hd, xk) + cd, np.zeros((N + 1, ne))))),
                          z=None if z is None else _lib.f64(np.ravel(z)), zf=None,
                          u=None if u is None else _lib.f64(np.ravel(u)))

    def update(self, Ad, Bd, dd, x0, xk, delta, omega, z=None, zf=None, u=None, full=True, **kwargs):
        """locp.py:98-173.  full=False only changes delta / omega (locp.py:139-141)."""
        if self.nonlinear_observer and (full or self._data is None):
            self._zf_const = 0.0 if (self.Qzf is None or zf is None) else float(np.asarray(zf) @ self.Qzf @ np.asarray(zf))
            self._update_augmented(Ad, Bd, dd, x0, xk, z, u, kwargs.get('Hd'), kwargs.get('cd'))
        elif self._du_aug and (full or self._data is None):
            self._update_rate_augmented(Ad, Bd, dd, x0, xk, z, zf, u)
        elif full or self._data is None:
            N, n, m = self.N, self.n_x, self.n_u
            self._data = dict(
                Ad=_lib.f64(np.asarray(Ad).reshape(N, n, n)), Bd=_lib.f64(np.asarray(Bd).reshape(N, n, m)),
                dd=_lib.f64(np.asarray(dd).reshape(N, n)), x0=_lib.f64(np.asarray(x0).reshape(n)),
                xk=None if xk is None else _lib.f64(np.asarray(xk).reshape(N + 1, n)),
                z=None if z is None else _lib.f64(np.ravel(z)),
                zf=None if (self.Qzf is None or zf is None) else _lib.f64(zf),
                u=None if u is None else _lib.f64(np.ravel(u)))
        self._delta = np.array([float(delta)])
        self._omega = np.array([float(omega)])

    def _init_nullspace(self, input_nullspace):
        """locp.py:70-71, 258-261: J += || tile(input_nullspace, N) @ u ||_2 (not squared).  np.tile repeats along the last axis: a
        vector v gives |sum_k v . u_k|, a matrix M (k x n_u) gives || M sum_k u_k ||_2.  Not a QP -- but
        ||g|| = max_{||mu|| <= 1} mu' g, and for a fixed mu the term is LINEAR in u, i.e. a shift of the desired input of the same
        device QP:  (u - ud)' R (u - ud) + c' u = (u - ud + R^-1 c / 2)' R (.) + c' ud - c' R^-1 c / 4  with c = M' mu.  solve()
        maximises the concave dual over the unit ball (_solve_nullspace); every evaluation is one solve of the resident plan."""
        self.input_nullspace = input_nullspace
        self._ns = None
        if input_nullspace is None:
            return
        M = np.atleast_2d(np.asarray(input_nullspace, dtype=np.float64))
        if M.ndim != 2 or M.shape[1] != self.n_u:
            raise ValueError('input_nullspace: expected (n_u,) or (k, n_u), got %s' % (np.shape(input_nullspace),))
        R = np.asarray(self.R, dtype=np.float64)
        try:
            np.linalg.cholesky(R)
        except np.linalg.LinAlgError:
            raise ValueError('input_nullspace needs a positive definite R (the term becomes a shift of the desired input)')
        self._ns = dict(M=M, Rinv=np.linalg.inv(R))
        self.nullspace_stats = None

    def _solve_nullspace(self):
        """Dual maximisation of the input_nullspace term.  g(mu) = M sum_k u*_k(mu) is the gradient of the concave dual d(mu) =
        min_u f(u) + mu' M sum_k u_k; the duality gap ||g|| - mu' g of the pair (u*(mu), mu) bounds the suboptimality of u*(mu)
        (every u*(mu) is feasible), so the loop stops on it.  One row: g is a monotone scalar function of mu in [-1, 1] -- the end
        point on the side of g(0), or a bracketed secant / bisection root.  Several rows: projected gradient ascent on the unit
        ball with Barzilai-Borwein steps and a monotone safeguard (the dual of a QP is piecewise quadratic)."""
        ns = self._ns
        M, Rinv = ns['M'], ns['Rinv']
        N, m = self.N, self.n_u
        k = M.shape[0]
        d = self._data
        ud = np.zeros((N, m)) if d.get('u') is None else np.asarray(d['u'], dtype=np.float64).reshape(N, m)
        tol = float(os.environ.get('SRH_NULLSPACE_TOL', 1e-10))
        calls = [0]
        t_solve = [0.0]

        def evaluate(mu):
            c = M.T @ mu
            J, ok, st = self._solve_once(_lib.f64(np.ravel(ud - 0.5 * (Rinv @ c))))
            calls[0] += 1
            if not ok:
                return None
            t_solve[0] += st.solve_time
            x, u, sl = self._sol
            su = u.sum(axis=0)
            g = M @ su
            f = J - float(c @ su) + float(c @ ud.sum(axis=0)) - 0.25 * N * float(c @ Rinv @ c)     # the cost without the term
            ng = float(np.linalg.norm(g))
            return dict(mu=mu, g=g, f=f, primal=f + ng, dual=f + float(mu @ g), gap=ng - float(mu @ g), sol=self._sol, iters=st.num_iters)

        def good(e):
            return e['gap'] <= tol * max(1.0, abs(e['primal']))

        best = e0 = evaluate(np.zeros(k))
        if e0 is None:
            return np.inf, False, None
        if not good(e0):
            if k == 1:
                s0 = 1.0 if e0['g'][0] > 0 else -1.0
                # expansion from mu = 0 towards the end point s0: the first trial is the Newton step of the dual with the input rows
                # and the state cost ignored (d g / d mu = -N M R^-1 M' / 2: the inputs cannot respond more than that, so the trial
                # stays short of the root), then secant extrapolations through the last two points, at least doubling
                curv = 0.5 * N * float(M[0] @ Rinv @ M[0])
                prev, e1 = e0, None
                t = e0['g'][0] / curv if curv > 0 else s0
                for it in range(8):
                    t = s0 if abs(t) >= 1.0 or it == 7 else t
                    e1 = evaluate(np.array([t]))
                    if e1 is None:
                        return np.inf, False, None
                    if e1['gap'] < best['gap']:
                        best = e1
                    if good(e1) or e1['g'][0] * s0 < 0 or abs(t) >= 1.0:
                        break
                    dg = e1['g'][0] - prev['g'][0]
                    step = -e1['g'][0] * (e1['mu'][0] - prev['mu'][0]) / dg if dg * s0 < 0 else 2.0 * (e1['mu'][0] - prev['mu'][0])
                    if abs(step) < abs(e1['mu'][0] - prev['mu'][0]):
                        step = 2.0 * (e1['mu'][0] - prev['mu'][0])
                    prev, t = e1, e1['mu'][0] + 1.5 * step               # (overshoot: the aim is a sign change)
                e0b = prev
                if not good(e1) and e1['g'][0] * s0 < 0:            # the sign of g changes inside: its root is the optimum
                    # g is monotone and piecewise linear in mu (the active set of the QP is constant on each piece): the secant of the
                    # bracket with the Illinois rule (an end point kept twice has its value halved) -- exact as soon as the bracket lies
                    # inside one piece; a bisection step when the secant leaves the bracket or the bracket stops shrinking
                    a, ga, b, gb = e0b['mu'][0], e0b['g'][0], e1['mu'][0], e1['g'][0]
                    width = abs(b - a)
                    for it in range(60):
                        t = (a * gb - b * ga) / (gb - ga)
                        if not (min(a, b) < t < max(a, b)) or (it % 4 == 3 and abs(b - a) > 0.5 * width):
                            t = 0.5 * (a + b)
                        if it % 4 == 3:
                            width = abs(b - a)
                        e = evaluate(np.array([t]))
                        if e is None:
                            return np.inf, False, None
                        if e['gap'] < best['gap']:
                            best = e
                        gt = e['g'][0]
                        if good(e) or abs(b - a) <= 1e-15 or gt == 0.0:
                            break
                        if gt * gb < 0:
                            a, ga = b, gb                               # the root is between the last two points
                        else:
                            ga *= 0.5                                   # same side as before: Illinois
                        b, gb = t, gt
            else:
                proj = lambda v: v / max(1.0, float(np.linalg.norm(v)))
                cur = e0
                stalled = 0
                step = 1.0 / max(float(np.linalg.norm(e0['g'])), 1e-300)      # the first step reaches the sphere
                for it in range(200):
                    mu_n = proj(cur['mu'] + step * cur['g'])
                    e = evaluate(mu_n)
                    if e is None:
                        return np.inf, False, None
                    # overshoot: the dual must not fall -- by more than what the QP solves resolve (their objectives carry ~1e-9
                    # relative; near the optimum the differences are smaller than that and the gradient alone steers)
                    if e['dual'] < cur['dual'] - 1e-8 * max(1.0, abs(cur['dual'])) and step > 1e-12:
                        step *= 0.25
                        continue
                    if e['gap'] < 0.9 * best['gap']:
                        stalled = 0
                    else:
                        stalled += 1
                    if e['gap'] < best['gap']:
                        best = e
                    if good(e) or stalled > 12:                                # (stalled: the gap sits at what the QP solves resolve)
                        break
                    dm, dg = e['mu'] - cur['mu'], e['g'] - cur['g']
                    curv = -float(dm @ dg)
                    if curv > 0:
                        step = float(dm @ dm) / curv                           # Barzilai-Borwein
                    elif float(np.linalg.norm(dm)) == 0.0:
                        break                                                  # pinned on the sphere with g along mu: optimal
                    cur = e
        self._sol = best['sol']
        self.nullspace_stats = dict(mu=best['mu'], gap=best['gap'], qp_solves=calls[0], term=float(np.linalg.norm(best['g'])))
        return best['primal'], True, _Stats(t_solve[0], best['iters'])

    def solve(self):
        """locp.py:175-190: returns (Jstar, success, stats)."""
        if self._ns is not None:
            return self._solve_nullspace()
        return self._solve_once(None)

    def _solve_once(self, u_des):
        """One solve of the resident QP; u_des (flat N n_u, or None: the desired input of update()) -- the only per-solve change the
        input_nullspace loop makes."""
        d = self._data
        N, n, m = self.N, self.n_x, self.n_u
        if self.nonlinear_observer:
            n = n + self.n_z + self._ne
        if self._du_aug:
            n = self._na
        x = np.empty((N + 1, n)); u = np.empty((N, m)); s = np.empty(N + 1)
        J = np.empty(1); status = np.empty(1, dtype=np.int32); iters = np.empty(1, dtype=np.int32)
        L = _lib.lib()
        if getattr(self, '_plan', None) is None:
            # resident QP (slocp_plan_*): constants, horizon and work buffers are created once per LOCP object
            self._plan = C.c_void_p()
            _lib.check(L.slocp_plan_create(C.byref(self._plan), C.byref(self._prob), C.c_int64(1)), 'slocp_plan_create')
            self._resident = None
        if d.get('xk') is None and self.tr_active:
            raise RuntimeError('LOCP.solve: xk is required when the trust region is active')
        # the horizon arrays go up only when update(full=True) replaced them (locp.py:139-141: a delta / omega update keeps them)
        fresh = self._resident is not d
        t0 = time.time()
        _lib.check(L.slocp_plan_solve(self._plan, _lib.dptr(d['Ad']) if fresh else None, _lib.dptr(d['Bd']) if fresh else None,
                                      _lib.dptr(d['dd']) if fresh else None, _lib.dptr(d['x0']),
                                      _lib.dptr(d['xk']) if fresh else None, _lib.dptr(self._delta), _lib.dptr(self._omega),
                                      _lib.dptr(d['z']), _lib.dptr(d['zf']), _lib.dptr(d['u'] if u_des is None else u_des), _lib.dptr(x), _lib.dptr(u),
                                      _lib.dptr(s), _lib.dptr(J), _lib.iptr(status), _lib.iptr(iters)), 'slocp_plan_solve')
        self._resident = d
        t1 = time.time()
        if status[0] == 0:
            if self.nonlinear_observer:
                x = np.ascontiguousarray(x[:, :self.n_x])
                J[0] += self._zf_const
            if self._du_aug:
                x = np.ascontiguousarray(x[:, :self.n_x])
            self._sol = (x, u, s if self.tr_active else None)
            return float(J[0]), True, _Stats(t1 - t0, int(iters[0]))
        return np.inf, False, None

    @property
    def kernel_info(self):
        """Kernel family / instantiation of this QP's resident plan and the number of QPs of the last solve the lean
        kernel handed to the fused one (slocp_plan_info); None before the first solve created the plan."""
        if getattr(self, '_plan', None) is None:
            return None
        info = _lib.SrhKernelInfo()
        _lib.check(_lib.lib().slocp_plan_info(self._plan, C.byref(info)), 'slocp_plan_info')
        return info.as_dict()

    def get_solution(self):
        """locp.py:192-203."""
        return self._sol

    def __del__(self):
        try:
            if getattr(self, '_plan', None):
                _lib.lib().slocp_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass
