"""Oracle: the LOCP horizon QP (test infrastructure only).

Parity status.  The QP *statement* is PINNED to the reference: `build_qp` assembles the stacked QP over
w = [x_0..x_N ; u_0..u_{N-1} ; s_0..s_N] of sofacontrol/scp/locp.py:218-342 and
tests/test_oracle_golden.py::test_locp_statement_matches_reference_locp_py holds its objective value and every
constraint residual to the outputs of the reference's OWN locp.py, executed in the build container through an
evaluating cvxpy stand-in (tests/golden/_cvxpy_eval.py -> fixture g14_locp.npz: 17 cases covering trust region
on/off, U, X, Xf, dU, u_des, Qzf/zf, nonlinear observer; ten seeded points + the optimum each, 1e-12).
The *solver* the reference hands that QP to (cvxpy -> OSQP | GUROBI, locp.py:181,216) is third-party, not
vendored, versions not pinned (requirements.txt:1-5) and not installable here; but the QP has a unique (x, u)
solution (R > 0 and x is an affine function of u), so any correct solver pins the answer:

* `solve_exact`   -- primal-dual interior point on the full sparse KKT system (scipy.sparse
                     spsolve), run to ~1e-10; returns multipliers for `kkt_certificate`.
* `solve_eq_only` -- closed-form dense KKT solve when no inequality is present / active.
* `solve_osqp`    -- restatement of the published OSQP ADMM algorithm (Stellato et al. 2020,
                     Alg. 1 with Ruiz equilibration and adaptive rho), the reference's default
                     solver, at cvxpy's default tolerances -- shows what accuracy the reference
                     itself delivers.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class QPData:
    pass


def build_qp(N, H, Qz, R, Ad, Bd, dd, x0, xk, delta, omega, z=None, u_des=None, Qzf=None, zf=None,
             U=None, X=None, Xf=None, dU=None, x_scale=None, tr_active=True, Hd=None, cd=None):
    """Stacked QP  min w'Pq w + c'w + c0  s.t.  E w = e,  G w <= h   (objective WITHOUT 1/2, as
    cp.quad_form, locp.py:226,248).  U/X/Xf/dU are (A, b) tuples or None.

    Index ranges follow locp.py: dynamics 287; TR 290-297 (all k=0..N, Fortran-order reshape so
    column k is x_k); U 300-303 (k<N); dU 305-308 (k<N-1); X 330-333 (k=1..N); Xf 336-337;
    x_0 = x0 340.  Terminal cost: locp.py:252 slices x[N*n_z:], which only type-checks when
    n_x == n_z, where it equals x_N -- restated as a cost on x_N.

    Hd (N+1 x n_z x n_x), cd (N+1 x n_z): per-stage observer linearisation of the nonlinear-observer
    branch (locp.py:231-245 objective, 312-329 state constraints X.A (Hd_k x_k + cd_k) <= X.b, k=1..N);
    the terminal cost keeps the constant H (locp.py:251-252).
    """
    n, m = Bd[0].shape
    nz = Qz.shape[0]
    nx_tot, nu_tot = (N + 1) * n, N * m
    ns = (N + 1) if tr_active else 0
    nw = nx_tot + nu_tot + ns
    ox, ou, os_ = 0, nx_tot, nx_tot + nu_tot
    z = np.zeros((N + 1, nz)) if z is None else np.asarray(z).reshape(N + 1, nz)
    u_des = np.zeros((N, m)) if u_des is None else np.asarray(u_des).reshape(N, m)
    xs = np.ones(n) if x_scale is None else x_scale

    Pq = sp.lil_matrix((nw, nw))
    c = np.zeros(nw)
    c0 = 0.0
    for k in range(N + 1):
        Hk = H if Hd is None else np.asarray(Hd[k])
        zk = z[k] if cd is None else z[k] - np.asarray(cd[k])
        Pq[ox + k * n: ox + (k + 1) * n, ox + k * n: ox + (k + 1) * n] = Hk.T @ Qz @ Hk
        c[ox + k * n: ox + (k + 1) * n] = -2.0 * Hk.T @ Qz @ zk
        c0 += zk @ Qz @ zk
    if Qzf is not None:
        zf_ = np.zeros(nz) if zf is None else zf
        Pq[ox + N * n: ox + (N + 1) * n, ox + N * n: ox + (N + 1) * n] += H.T @ Qzf @ H
        c[ox + N * n: ox + (N + 1) * n] += -2.0 * H.T @ Qzf @ zf_
        c0 += zf_ @ Qzf @ zf_
    for k in range(N):
        Pq[ou + k * m: ou + (k + 1) * m, ou + k * m: ou + (k + 1) * m] = R
        c[ou + k * m: ou + (k + 1) * m] = -2.0 * R @ u_des[k]
        c0 += u_des[k] @ R @ u_des[k]
    if tr_active:
        c[os_:] = omega

    E = sp.lil_matrix(((N + 1) * n, nw))
    e = np.zeros((N + 1) * n)
    for k in range(N):
        r0 = k * n
        E[r0:r0 + n, ox + (k + 1) * n: ox + (k + 2) * n] = np.eye(n)
        E[r0:r0 + n, ox + k * n: ox + (k + 1) * n] = -np.asarray(Ad[k])
        E[r0:r0 + n, ou + k * m: ou + (k + 1) * m] = -np.asarray(Bd[k])
        e[r0:r0 + n] = np.asarray(dd[k])
    E[N * n:(N + 1) * n, ox: ox + n] = np.eye(n)
    e[N * n:] = x0

    G_list = []
    h_list = []
    if tr_active:
        for k in range(N + 1):
            Gk = sp.lil_matrix((2 * n + 1, nw))
            hk = np.zeros(2 * n + 1)
            for i in range(n):
                Gk[2 * i, ox + k * n + i] = xs[i]
                Gk[2 * i, os_ + k] = -1.0
                hk[2 * i] = delta + xs[i] * xk[k, i]
                Gk[2 * i + 1, ox + k * n + i] = -xs[i]
                Gk[2 * i + 1, os_ + k] = -1.0
                hk[2 * i + 1] = delta - xs[i] * xk[k, i]
            Gk[2 * n, os_ + k] = -1.0
            G_list.append(Gk)
            h_list.append(hk)
    if U is not None:
        UA, Ub = U
        for k in range(N):
            Gk = sp.lil_matrix((UA.shape[0], nw))
            Gk[:, ou + k * m: ou + (k + 1) * m] = UA
            G_list.append(Gk)
            h_list.append(np.asarray(Ub, dtype=float))
    if dU is not None:
        dA, db = dU
        for k in range(N - 1):
            Gk = sp.lil_matrix((dA.shape[0], nw))
            Gk[:, ou + (k + 1) * m: ou + (k + 2) * m] = dA
            Gk[:, ou + k * m: ou + (k + 1) * m] = -dA
            G_list.append(Gk)
            h_list.append(np.asarray(db, dtype=float))
    if X is not None:
        XA, Xb = X
        for k in range(1, N + 1):
            Gk = sp.lil_matrix((XA.shape[0], nw))
            if Hd is None:
                Gk[:, ox + k * n: ox + (k + 1) * n] = XA
                h_list.append(np.asarray(Xb, dtype=float))
            else:
                Gk[:, ox + k * n: ox + (k + 1) * n] = XA @ np.asarray(Hd[k])
                h_list.append(np.asarray(Xb, dtype=float) - XA @ np.asarray(cd[k]))
            G_list.append(Gk)
    if Xf is not None:
        XA, Xb = Xf
        Gk = sp.lil_matrix((XA.shape[0], nw))
        Gk[:, ox + N * n: ox + (N + 1) * n] = XA
        G_list.append(Gk)
        h_list.append(np.asarray(Xb, dtype=float))

    qp = QPData()
    qp.N, qp.n, qp.m, qp.ns = N, n, m, ns
    qp.Pq = sp.csc_matrix(Pq)
    qp.c, qp.c0 = c, c0
    qp.E, qp.e = sp.csc_matrix(E), e
    if G_list:
        qp.G = sp.csc_matrix(sp.vstack([g.tocsr() for g in G_list]))
        qp.h = np.concatenate(h_list)
    else:
        qp.G = sp.csc_matrix((0, nw))
        qp.h = np.zeros(0)
    return qp


def objective(qp, w):
    return float(w @ (qp.Pq @ w) + qp.c @ w + qp.c0)


def split(qp, w):
    N, n, m = qp.N, qp.n, qp.m
    x = w[:(N + 1) * n].reshape(N + 1, n)
    u = w[(N + 1) * n:(N + 1) * n + N * m].reshape(N, m)
    s = w[(N + 1) * n + N * m:] if qp.ns else None
    return x, u, s


def solve_eq_only(qp):
    """Exact dense KKT solve ignoring inequalities (valid when none is active)."""
    nw, ne = qp.Pq.shape[0], qp.E.shape[0]
    # slack columns have zero Hessian: pin them with a unit diagonal (they are decoupled here)
    P2 = (2.0 * qp.Pq).tolil()
    if qp.ns:
        for j in range(nw - qp.ns, nw):
            P2[j, j] = 1.0
    c = qp.c.copy()
    if qp.ns:
        c[nw - qp.ns:] = 0.0
    K = sp.bmat([[P2, qp.E.T], [qp.E, None]], format='csc')
    sol = spla.spsolve(K, np.concatenate((-c, qp.e)))
    return sol[:nw], sol[nw:]


def solve_exact(qp, tol=1e-12, max_iter=200, verbose=False):
    """Mehrotra predictor-corrector primal-dual interior point on the full sparse KKT system.

    Returns w, (y, lam), info.  Independent of the product's Riccati-structured solver: the Newton
    systems here are solved as one sparse indefinite system by SuperLU.
    """
    P = (2.0 * qp.Pq).tocsc()
    q, E, e, G, h = qp.c, qp.E, qp.e, qp.G, qp.h
    nw, ne, ng = P.shape[0], E.shape[0], G.shape[0]
    if ng == 0:
        w, y = solve_eq_only(qp)
        return w, (y, np.zeros(0)), dict(iters=0, mu=0.0)
    GT = G.T.tocsc()
    ET = E.T.tocsc()
    # starting point (the heuristic of CVXOPT's coneqp): least-squares point of the KKT system
    # with unit scaling, then shift slacks / multipliers into the positive orthant
    K0 = sp.bmat([[P + GT @ G + 1e-13 * sp.eye(nw), ET], [E, -1e-13 * sp.eye(ne)]], format='csc')
    sol = spla.splu(K0).solve(np.concatenate((-q + GT @ h, e)))
    w, y = sol[:nw], sol[nw:]
    zz = G @ w - h
    t = -zz
    if t.min() <= 0:
        t = t + (1.0 - t.min())
    lam = zz.copy()
    if lam.min() <= 0:
        lam = lam + (1.0 - lam.min())
    scale_d = max(1.0, np.abs(q).max())
    scale_p = max(1.0, np.abs(h).max() if ng else 1.0, np.abs(e).max())
    info = {}
    for it in range(max_iter):
        r_d = P @ w + q + ET @ y + GT @ lam
        r_e = E @ w - e
        r_g = G @ w + t - h
        mu = float(lam @ t) / ng
        if verbose:
            print(it, np.abs(r_d).max(), np.abs(r_e).max(), np.abs(r_g).max(), mu)
        # the complementarity gap is driven to `tol`; the linear residuals are limited by the accuracy
        # of the sparse LU on a KKT matrix whose weights lam/t span >20 decades near the solution
        rtol_lin = max(tol, 1e-8)
        if (np.abs(r_d).max() <= rtol_lin * scale_d and np.abs(r_e).max() <= rtol_lin * scale_p and
                np.abs(r_g).max() <= rtol_lin * scale_p and mu <= tol):
            break
        D = lam / t
        Phi = P + GT @ sp.diags(D) @ G + 1e-13 * sp.eye(nw)
        K = sp.bmat([[Phi, ET], [E, -1e-13 * sp.eye(ne)]], format='csc')
        K_true = sp.bmat([[P + GT @ sp.diags(D) @ G, ET], [E, None]], format='csc')
        lu = spla.splu(K)

        def newton(r_c):
            rhs1 = -r_d - GT @ ((-r_c + lam * r_g) / t)
            rhs = np.concatenate((rhs1, -r_e))
            sol = lu.solve(rhs)
            for _ in range(3):      # iterative refinement against the unregularised KKT matrix
                sol = sol + lu.solve(rhs - K_true @ sol)
            dw, dy = sol[:nw], sol[nw:]
            dt = -r_g - G @ dw
            dlam = (-r_c - lam * dt) / t
            return dw, dy, dt, dlam

        def step_len(v, dv):
            neg = dv < 0
            return float(np.min(-v[neg] / dv[neg])) if neg.any() else np.inf

        dw, dy, dt, dlam = newton(lam * t)
        a_aff = min(1.0, step_len(t, dt), step_len(lam, dlam))
        mu_aff = float((lam + a_aff * dlam) @ (t + a_aff * dt)) / ng
        sigma = (mu_aff / mu) ** 3 if mu > 0 else 0.0
        dw, dy, dt, dlam = newton(lam * t + dt * dlam - sigma * mu)
        # one common primal/dual step (QP: the dual residual couples w and the multipliers)
        a = min(1.0, 0.99 * min(step_len(t, dt), step_len(lam, dlam)))    # stay strictly interior
        w = w + a * dw
        t = t + a * dt
        y = y + a * dy
        lam = lam + a * dlam
        if not (np.isfinite(mu) and np.all(np.isfinite(w))):
            break
    info['iters'] = it
    info['mu'] = mu
    info['status'] = 'optimal' if it < max_iter - 1 and np.isfinite(mu) else 'failed'
    return w, (y, lam), info


def kkt_certificate(qp, w, y, lam):
    """Residuals of the KKT conditions of the stacked QP (all should be ~0):
    stationarity, equality, inequality violation, negative multipliers, complementarity."""
    P = 2.0 * qp.Pq
    stat = P @ w + qp.c + qp.E.T @ y + (qp.G.T @ lam if lam.size else 0.0)
    g = qp.G @ w - qp.h if lam.size else np.zeros(0)
    return dict(stationarity=float(np.abs(stat).max()),
                equality=float(np.abs(qp.E @ w - qp.e).max()),
                ineq_violation=float(np.maximum(g, 0).max()) if g.size else 0.0,
                dual_negativity=float(np.maximum(-lam, 0).max()) if lam.size else 0.0,
                complementarity=float(np.abs(lam * g).max()) if g.size else 0.0)


def solve_osqp(qp, eps_abs=1e-5, eps_rel=1e-5, max_iter=10000, rho=0.1, sigma=1e-6, alpha=1.6,
               scaling_iters=10, adaptive_rho_interval=50):
    """Restatement of OSQP (Stellato et al. 2020): Alg. 1 ADMM, Ruiz equilibration (Alg. 2),
    adaptive rho (sec. 5.2), equality rows weighted 1e3*rho.  Tolerances default to the values
    cvxpy passes for OSQP (1e-5); max_iter as cvxpy (10000).  No polishing.
    Standard form: min 1/2 x'Px + q'x  s.t.  l <= A x <= u."""
    P = (2.0 * qp.Pq).tocsc()
    q = qp.c.copy()
    A = sp.vstack([qp.E, qp.G]).tocsc() if qp.G.shape[0] else qp.E.tocsc()
    ne = qp.E.shape[0]
    l = np.concatenate((qp.e, -np.inf * np.ones(qp.G.shape[0])))
    u = np.concatenate((qp.e, qp.h))
    n, m = P.shape[0], A.shape[0]
    # Ruiz equilibration
    D = np.ones(n); Ev = np.ones(m); cscale = 1.0
    Ps, As, qs = P.copy(), A.copy(), q.copy()
    for _ in range(scaling_iters):
        colP = np.abs(Ps).max(axis=0).toarray().ravel()
        colA = np.abs(As).max(axis=0).toarray().ravel()
        dn = np.maximum(colP, colA)
        dn[dn < 1e-4] = 1.0
        dn = 1.0 / np.sqrt(np.minimum(dn, 1e4))
        rowA = np.abs(As).max(axis=1).toarray().ravel()
        rowA[rowA < 1e-4] = 1.0
        en = 1.0 / np.sqrt(np.minimum(rowA, 1e4))
        Dm, Em = sp.diags(dn), sp.diags(en)
        Ps = (Dm @ Ps @ Dm).tocsc()
        As = (Em @ As @ Dm).tocsc()
        qs = dn * qs
        D *= dn; Ev *= en
        colmean = np.abs(Ps).max(axis=0).toarray().ravel().mean()
        gam = 1.0 / max(colmean, np.abs(qs).max(), 1e-4)
        gam = min(max(gam, 1e-4), 1e4)
        Ps = Ps * gam; qs = qs * gam; cscale *= gam
    ls, us = Ev * l, Ev * u
    is_eq = np.zeros(m, dtype=bool); is_eq[:ne] = True

    def factor(rho):
        rv = np.where(is_eq, 1e3 * rho, rho)
        K = sp.bmat([[Ps + sigma * sp.eye(n), As.T], [As, -sp.diags(1.0 / rv)]], format='csc')
        return spla.splu(K), rv

    lu, rv = factor(rho)
    x = np.zeros(n); zv = np.zeros(m); y = np.zeros(m)
    status = 'max_iter'
    for it in range(1, max_iter + 1):
        sol = lu.solve(np.concatenate((sigma * x - qs, zv - y / rv)))
        xt, nu = sol[:n], sol[n:]
        zt = zv + (nu - y) / rv
        x = alpha * xt + (1 - alpha) * x
        zh = alpha * zt + (1 - alpha) * zv
        z_new = np.minimum(np.maximum(zh + y / rv, ls), us)
        y = y + rv * (zh - z_new)
        zv = z_new
        if it % 10 == 0 or it == 1:
            # unscaled residuals
            xu = D * x
            Ax = (A @ xu)
            zu = zv / Ev
            yu = Ev * y / cscale
            r_p = np.abs(Ax - zu).max()
            Px = P @ xu; Aty = A.T @ yu
            r_d = np.abs(Px + q + Aty).max()
            e_p = eps_abs + eps_rel * max(np.abs(Ax).max(), np.abs(zu).max())
            e_d = eps_abs + eps_rel * max(np.abs(Px).max(), np.abs(Aty).max(), np.abs(q).max())
            if r_p <= e_p and r_d <= e_d:
                status = 'solved'
                break
            if it % adaptive_rho_interval == 0:
                num = r_p / max(np.abs(Ax).max(), np.abs(zu).max(), 1e-12)
                den = r_d / max(np.abs(Px).max(), np.abs(Aty).max(), np.abs(q).max(), 1e-12)
                new_rho = float(np.clip(rho * np.sqrt(num / max(den, 1e-12)), 1e-6, 1e6))
                if new_rho > 5 * rho or new_rho < rho / 5:
                    rho = new_rho
                    lu, rv = factor(rho)
    w = D * x
    ydual = Ev * y / cscale
    return w, (ydual[:ne], ydual[ne:]), dict(iters=it, status=status)


# ---------------------------------------------------------------------------------------------------------------
# input_nullspace (locp.py:70-71, 258-261): J += || tile(input_nullspace, N) @ u ||_2 -- NOT squared.  np.tile repeats
# the array along its last axis, so a vector v (n_u,) gives the scalar |sum_k v . u_k| (the use the reference's driver
# sketches: "nullspace of V^T H", diamond_SSM.py:258-259) and a matrix M (k x n_u) gives || M sum_k u_k ||_2.
# ---------------------------------------------------------------------------------------------------------------
def nullspace_rows(qp, input_nullspace):
    """tile(input_nullspace, N) as a (k x nw) matrix acting on the stacked w (zero outside the u block)."""
    M = np.atleast_2d(np.asarray(input_nullspace, dtype=float))
    N, n, m = qp.N, qp.n, qp.m
    W = np.zeros((M.shape[0], qp.Pq.shape[0]))
    W[:, (N + 1) * n:(N + 1) * n + N * m] = np.tile(M, N)
    return W


def nullspace_term(qp, input_nullspace, w):
    """The value locp.py:261 adds to the objective."""
    return float(np.linalg.norm(nullspace_rows(qp, input_nullspace) @ w))


def add_abs_epigraph(qp, input_nullspace):
    """One-row input_nullspace: |a' w| = min t s.t. a' w <= t, -a' w <= t -- the SAME problem as a QP with one more
    variable (appended behind the slacks; zero Hessian, unit cost).  Returns the extended QPData."""
    W = nullspace_rows(qp, input_nullspace)
    assert W.shape[0] == 1, 'the epigraph form is a QP for a single row only'
    nw = qp.Pq.shape[0]
    q2 = QPData()
    q2.N, q2.n, q2.m, q2.ns = qp.N, qp.n, qp.m, qp.ns
    q2.Pq = sp.bmat([[qp.Pq, None], [None, sp.csc_matrix((1, 1))]], format='csc')
    q2.c, q2.c0 = np.concatenate((qp.c, [1.0])), qp.c0
    q2.E, q2.e = sp.hstack((qp.E, sp.csc_matrix((qp.E.shape[0], 1))), format='csc'), qp.e
    rows = sp.csc_matrix(np.vstack((np.append(W[0], -1.0), np.append(-W[0], -1.0))))
    q2.G = sp.vstack((sp.hstack((qp.G, sp.csc_matrix((qp.G.shape[0], 1)))), rows), format='csc')
    q2.h = np.concatenate((qp.h, [0.0, 0.0]))
    return q2


def nullspace_certificate(qp, input_nullspace, w, mu):
    """Sufficient optimality conditions of  min f(w) + ||W w||_2  for a feasible w and a multiplier mu of the norm
    (||v|| = max_{||mu|| <= 1} mu' v): ||mu|| <= 1, w minimises f + mu' W w over the feasible set (checked by an exact solve of that
    QP), and the gap ||W w|| - mu' W w is zero.  For every feasible w':  f(w') + ||W w'|| >= f(w') + mu' W w' >= f(w) + mu' W w
    = f(w) + ||W w|| - gap, so `gap` bounds the suboptimality of w.  Returns dict(mu_norm, gap, inner_dw, inner_dJ)."""
    import copy
    W = nullspace_rows(qp, input_nullspace)
    mu = np.atleast_1d(np.asarray(mu, dtype=float))
    q2 = copy.copy(qp)
    q2.c = qp.c + W.T @ mu
    wi, _, info = solve_exact(q2)
    assert info['status'] == 'optimal', info
    g = W @ w
    return dict(mu_norm=float(np.linalg.norm(mu)), gap=float(np.linalg.norm(g) - mu @ g),
                inner_dw=float(np.abs(wi - w).max()), inner_dJ=float(objective(q2, w) - objective(q2, wi)))


def solve_with_nullspace(qp, input_nullspace, tol=1e-10, max_outer=60):
    """min f(w) + ||W w||_2 over the QP's feasible set; returns w (without auxiliary variables), the objective, info (with the
    multiplier `mu` of the norm).
    One row: the epigraph QP (exact).  Several rows (a second-order-cone term), by cases:
      (A) the optimum sits in the kink W w = 0: the QP with those equality rows; optimal iff their multiplier nu has ||nu|| <= 1
          (0 in grad f + W' d||.||(0));
      (B) otherwise the norm is smooth at the optimum: Newton's method on f + ||W w|| -- the second-order model of the norm at w_i is
          ghat' W w + w' Hphi w / 2 with Hphi = W' (I - ghat ghat') W / ||g|| (Hphi w_i = 0), a convex QP per step, exact line
          search by bisection on the segment to its minimiser (the feasible set is convex)."""
    import copy
    W = nullspace_rows(qp, input_nullspace)
    k = W.shape[0]
    nw = qp.Pq.shape[0]
    if k == 1:
        w2, (y2, lam2), info = solve_exact(add_abs_epigraph(qp, input_nullspace))
        w = w2[:nw]
        return w, objective(qp, w) + float(np.linalg.norm(W @ w)), dict(info, gap=abs(w2[nw] - abs(float(W[0] @ w))),
                                                                        mu=np.array([lam2[-2] - lam2[-1]]))
    # (A)
    qa = copy.copy(qp)
    qa.E = sp.vstack((qp.E, sp.csc_matrix(W)), format='csc')
    qa.e = np.concatenate((qp.e, np.zeros(k)))
    try:
        wa, (ya, _), ia = solve_exact(qa)
        nu = ya[-k:]
        if ia['status'] == 'optimal' and np.linalg.norm(nu) <= 1.0 + 1e-9:
            return wa, objective(qp, wa) + float(np.linalg.norm(W @ wa)), dict(ia, case='kink', mu=nu)
    except RuntimeError:
        pass                                            # (the rows can be infeasible together with the constraints: case B then)
    # (B)
    w, _, info = solve_exact(qp)
    phi = lambda w_: objective(qp, w_) + float(np.linalg.norm(W @ w_))
    for it in range(max_outer):
        g = W @ w
        ng = float(np.linalg.norm(g))
        if ng <= 1e-9 * max(1.0, float(np.abs(w).max()) * float(np.abs(W).max())):
            # the iterates run into the kink although (A) found ||nu|| > 1: the multipliers of W w = 0 are not unique there (active
            # rows of the QP span the same directions) -- a degenerate instance this restatement does not resolve
            raise RuntimeError('solve_with_nullspace: degenerate kink (W w -> 0 with non-unique multipliers)')
        gh = g / ng
        qn = copy.copy(qp)
        qn.Pq = (qp.Pq + sp.csc_matrix(0.5 * (W.T @ (np.eye(k) - np.outer(gh, gh)) @ W) / ng)).tocsc()
        qn.c = qp.c + W.T @ gh
        wn, _, info = solve_exact(qn)
        assert info['status'] == 'optimal', info
        d = wn - w
        lo, hi = 0.0, 1.0                               # minimiser of the convex phi(w + a d) on [0, 1] by its derivative's sign
        dphi = lambda a: float(d @ (2.0 * (qp.Pq @ (w + a * d)) + qp.c) + (W @ d) @ (W @ (w + a * d)) / np.linalg.norm(W @ (w + a * d)))
        if dphi(1.0) <= 0.0:
            a = 1.0
        else:
            for _ in range(60):
                a = 0.5 * (lo + hi)
                if dphi(a) > 0.0:
                    hi = a
                else:
                    lo = a
            a = 0.5 * (lo + hi)
        w_new = w + a * d
        done = phi(w) - phi(w_new) <= tol * max(1.0, abs(phi(w)))
        w = w_new
        if done:
            break
    g = W @ w
    return w, phi(w), dict(info, case='smooth', outer=it + 1, mu=g / np.linalg.norm(g))
