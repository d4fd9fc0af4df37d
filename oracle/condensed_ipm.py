"""Oracle / CPU port: the CONDENSED interior point for the LOCP QP (test infrastructure only).

numpy statement of the algorithm of the HIP kernel `csrc/locp_cond.h`: the same Mehrotra predictor-corrector
iteration as oracle/riccati_ipm.py on the same QP (sofacontrol/scp/locp.py:218-342) with the trust-region rows
prescreened away (see riccati_ipm / locp_dev.h: the relaxed minimiser is the minimiser of the full QP whenever it
lies inside the trust region) -- but with the states eliminated instead of carried:

  x_k = xfree_k + sum_{j<k} Phi(k, j+1) B_j u_j          (dynamics, x_0 = x0)

The cost (locp.py:226-252) and the state rows (X, Xf: locp.py:330-337) see the state only through a few OUTPUT
directions:  2 H^T Qz H = Cq^T Cq,  X.A,  Xf.A  all lie in the row space of  C_o (p x n)  (Diamond: the tip x / y
rows, p = 2).  With y_k = C_o x_k = yfree_k + sum_{j<k} G[k][j] u_j,  G[k][j] = C_o Phi(k, j+1) B_j (p x m), the QP
becomes one in u (N m variables) with block-diagonal input terms and output terms through the block lower-triangular
G (N p x N m).  A Newton system of the interior point is then

  M du = -g ,   M = blkdiag(2R + U.A^T D_u U.A) + G^T blkdiag(S_k) G ,   S_k = Tc^T Tc + Tx^T D_x,k Tx   (p x p)

`newton='primal'` factors M (N m x N m) by Cholesky; `newton='output'` solves the equivalent output-space system
(I + Gs Dinv Gs^T) v = ... of size N p (Gs = rows of G scaled by chol(S_k)) -- what the kernel does (`refine` > 0 adds
steps of iterative refinement on the primal residual, which the kernel does not: its reduced dual residual passes
through ~1e-5 relative mid-way and reaches the 1e-9 stopping level with the last iterations, sometimes one or two
iterations after the refined variant).  All variants give the Newton direction of riccati_ipm up to rounding.

One-off per QP: G by the adjoint recursion  Psi_j = [C_o ; Psi_{j+1}] A_j,  G[:, j] = Psi_j B_j  (N^2/2 products of a
p-row block with A_j instead of N factorisations of n x n matrices per interior-point iteration).
"""
import numpy as np

from . import riccati_ipm as ripm


def output_basis(p, tol=1e-10):
    """C_o (p_o x n, orthonormal rows) spanning the rows of Cq (2 H^T Qz H = Cq^T Cq), Cqf, X.A, Xf.A, and the
    coefficient matrices with  Cq = Tc C_o,  Cqf = Tcf C_o,  X.A = Tx C_o,  Xf.A = Txf C_o."""
    n = p.n

    def sqrt_rows(Q):
        if Q is None:
            return np.zeros((0, n))
        w, V = np.linalg.eigh(0.5 * (Q + Q.T))
        keep = w > 1e-13 * max(1e-300, np.abs(w).max())
        return (np.sqrt(2.0 * w[keep])[:, None] * V[:, keep].T) @ p.H
    Cq, Cqf = sqrt_rows(p.Qz), sqrt_rows(p.Qzf)
    XA = p.X[0] if p.X is not None else np.zeros((0, n))
    XfA = p.Xf[0] if p.Xf is not None else np.zeros((0, n))
    # the linear cost term -2 H^T Qz z lies in the row space of Cq as well
    stack = np.vstack([Cq, Cqf, XA, XfA])
    if stack.shape[0] == 0:
        return np.zeros((0, n)), Cq, Cqf, XA, XfA
    nrm = np.linalg.norm(stack, axis=1)
    _, sv, Vt = np.linalg.svd(stack / np.maximum(nrm, 1e-300)[:, None], full_matrices=False)
    po = int((sv > tol * sv[0]).sum())
    Co = Vt[:po]
    return Co, Cq @ Co.T, Cqf @ Co.T, XA @ Co.T, XfA @ Co.T


def chol_psd(S):
    """Lower Cholesky factor of a symmetric positive SEMI-definite matrix (S = L L^T): a pivot that has cancelled to
    (numerically) nothing gets a zero column.  The output blocks S_k = Tc^T Tc + Tx^T D Tx lose rank in floating point
    when an output direction is weighted only by state rows whose weights D -> 0 (inactive rows late in the iteration)."""
    n = S.shape[0]
    L = np.zeros_like(S)
    dmax = max(np.abs(np.diag(S)).max(), 1e-300)
    for i in range(n):
        for j in range(i + 1):
            v = S[i, j] - L[i, :j] @ L[j, :j]
            if i == j:
                L[i, i] = np.sqrt(v) if v > 1e-14 * dmax else 0.0
            else:
                L[i, j] = v / L[j, j] if L[j, j] > 0.0 else 0.0
    return L


def condense(p, Co):
    """xfree (N+1, n), yfree (N+1, p_o) and G as a dense (N+1, p_o, N, m) array (G[k, :, j, :] = 0 for j >= k)."""
    N, n, m = p.N, p.n, p.m
    po = Co.shape[0]
    xf = np.zeros((N + 1, n)); xf[0] = p.x0
    for k in range(N):
        xf[k + 1] = p.A[k] @ xf[k] + p.d[k]
    G = np.zeros((N + 1, po, N, m))
    Psi = np.zeros((0, n))                      # rows C_o Phi(k, j+1), k = j+1 .. N (newest first)
    for j in range(N - 1, -1, -1):
        Psi = np.vstack([Co, Psi @ p.A[j + 1]]) if j + 1 < N else Co.copy()
        blk = Psi @ p.B[j]                      # ((N - j) p_o x m): k = j+1 .. N
        G[j + 1:, :, j, :] = blk.reshape(N - j, po, m)
    return xf, xf @ Co.T, G


WARM_FLOOR = 1e-2      # warm start: slacks and multipliers are pushed at least this far from zero (kernel: locp_lean.h, twin: cond_solve)


def solve(p, tol=1e-12, max_iter=60, reg=1e-8, newton='output', refine=0, verbose=False, warm=None, dy_from_system=True):
    """The QP of `p` (riccati_ipm.Problem) WITHOUT its trust-region rows.  Returns x, u, J (objective without the
    omega * s term), info (iters, status, `inside`: whether the minimiser satisfies the trust region of p, and `final`:
    the last iterate (u, slacks, multipliers)).

    dy_from_system = False keeps the product dy = G du (what the kernels do when a constant output block is only semi-definite;
    tests compare the two forms).
    warm = the `final` of the previous QP of the same SCP solve (round 4): the interior point then starts from that
    point instead of Mehrotra's -- u as it is, every slack t = max(-g(u), WARM_FLOOR) from the row values of THIS QP, every
    multiplier max(lambda_prev, WARM_FLOOR) -- and skips the initial Newton system.  Successive QPs of an SCP solve differ
    by their linearisation point only and share most of their active set: 7-9 interior-point iterations instead of 15-18
    on the BASELINE C2 / C5 problems, same minimiser (the stopping rule is unchanged).  A warm-started solve that does not
    reach the tolerances is repeated cold."""
    N, n, m = p.N, p.n, p.m
    Co, Tc, Tcf, Tx, Txf = output_basis(p)
    po = Co.shape[0]
    xfree, yfree, G = condense(p, Co)
    Gm = G.reshape((N + 1) * po, N * m)         # rows (k, a), columns (j, b)
    UA, Ub = (p.U if p.U is not None else (np.zeros((0, m)), np.zeros(0)))
    nU, nX, nXf = UA.shape[0], Tx.shape[0], Txf.shape[0]
    Xb = p.X[1] if p.X is not None else np.zeros(0)
    Xfb = p.Xf[1] if p.Xf is not None else np.zeros(0)
    ng = N * nU + N * nX + nXf
    # constant pieces: cost in output space  1/2 y^T Sc y + l_k^T y  per stage (k = 0 is constant: x_0 fixed)
    Sc = Tc.T @ Tc
    ScN = Sc + Tcf.T @ Tcf
    lin = np.zeros((N + 1, po))
    for k in range(N + 1):
        g0 = -2.0 * p.H.T @ (p.Qz @ p.z[k])
        if k == N and p.Qzf is not None:
            g0 = g0 - 2.0 * p.H.T @ (p.Qzf @ p.zf)
        lin[k] = Co @ g0

    def outputs(u):
        return yfree + (Gm @ u.ravel()).reshape(N + 1, po)

    def xrows_T(k):                              # rows of x-stage k in output coordinates
        return np.vstack([Tx, Txf]) if k == N else Tx

    def xrows_b(k):
        return np.concatenate([Xb, Xfb]) if k == N else Xb

    def grad_parts(u, y, wx, wu):
        """Gradient wrt u of  cost + sum rows^T w  (w = rho for the Newton right-hand side, lam for the residual)."""
        gy = np.zeros((N + 1, po))
        for k in range(1, N + 1):
            gy[k] = (ScN if k == N else Sc) @ y[k] + lin[k] + xrows_T(k).T @ wx[k]
        gu = np.stack([p.Ru @ (u[k] - p.ud[k]) + UA.T @ wu[k] for k in range(N)])
        return gu.ravel() + Gm.T @ gy.ravel()

    def spd_small(S):
        L = np.zeros_like(S); dmax = np.abs(np.diag(S)).max()
        for i in range(S.shape[0]):
            for j in range(i + 1):
                v = S[i, j] - L[i, :j] @ L[j, :j]
                if i == j:
                    if not v > 1e-8 * dmax:
                        return False
                    L[i, i] = np.sqrt(v)
                else:
                    L[i, j] = v / L[j, j]
        return True
    # both constant output blocks positive definite: every Ls_k is invertible, and dy = G du follows from the solved system
    # itself (round 4; kernels: ql::newton_back / qpc::newton_solve, twin: direction_y) -- with w = ks v the output-space system
    # reads (I + Ls^T Ky Ls) w = Ls^T G t, Ky = G D^-1 G^T, hence G du = G t - Ky Ls w = Ls^-T w: no second product with G, and
    # the error of the K solve is not multiplied by K on its way into dy
    ls_pd = dy_from_system and newton == 'output' and refine == 0 and spd_small(Sc) and spd_small(ScN)
    last = {}

    def direction_y(du):
        if not ls_pd:
            return (Gm @ du.ravel()).reshape(N + 1, po)
        dy = np.zeros((N + 1, po))
        for k in range(1, N + 1):
            dy[k] = np.linalg.solve(last['Ls'][k].T, last['w'][(k - 1) * po:k * po])
        return dy

    def newton_solve(Du, Dx, rhs):
        """du with  M du = rhs."""
        Dblk = [p.Ru + UA.T @ (Du[k][:, None] * UA) for k in range(N)]
        Sk = [None] + [(ScN if k == N else Sc) + xrows_T(k).T @ (Dx[k][:, None] * xrows_T(k)) for k in range(1, N + 1)]
        if newton == 'primal':
            M = np.zeros((N * m, N * m))
            for k in range(N):
                M[k * m:(k + 1) * m, k * m:(k + 1) * m] = Dblk[k]
            for k in range(1, N + 1):
                Gk = Gm[k * po:(k + 1) * po]
                M += Gk.T @ Sk[k] @ Gk
            L = np.linalg.cholesky(M)
            return np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        # output space: M = D + Gs^T Gs with Gs = blkdiag(Ls_k^T) G, S_k = Ls_k Ls_k^T
        Ls = [None] + [chol_psd(Sk[k]) for k in range(1, N + 1)]
        Gs = np.vstack([Ls[k].T @ Gm[k * po:(k + 1) * po] for k in range(1, N + 1)])          # (N p x N m)
        Ld = [np.linalg.cholesky(Dblk[k]) for k in range(N)]
        Gd = np.hstack([np.linalg.solve(Ld[k], Gs[:, k * m:(k + 1) * m].T).T for k in range(N)])   # Gs Ld^-T
        K = np.eye(N * po) + Gd @ Gd.T
        ks = 1.0 / np.sqrt(np.diag(K))              # factored under a symmetric scaling to unit diagonal (as the kernel)
        Kc = np.linalg.cholesky(ks[:, None] * K * ks[None, :])

        def Dinv(v):
            out = np.empty_like(v)
            for k in range(N):
                out[k * m:(k + 1) * m] = np.linalg.solve(Ld[k].T, np.linalg.solve(Ld[k], v[k * m:(k + 1) * m]))
            return out

        def apply_M(v):
            out = Gs.T @ (Gs @ v)
            for k in range(N):
                out[k * m:(k + 1) * m] += Dblk[k] @ v[k * m:(k + 1) * m]
            return out

        def woodbury(r):
            t = Dinv(r)
            v = ks * np.linalg.solve(Kc.T, np.linalg.solve(Kc, ks * (Gs @ t)))
            last['w'], last['Ls'] = v, Ls
            return t - Dinv(Gs.T @ v)
        du = woodbury(rhs)
        for _ in range(refine):
            du = du + woodbury(rhs - apply_M(du))
        return du

    def row_vals(u, y):
        gx = [None] + [xrows_T(k) @ y[k] - xrows_b(k) for k in range(1, N + 1)]
        gu = [UA @ u[k] - Ub for k in range(N)]
        return gx, gu

    def row_dirs(du, dy):
        return [None] + [xrows_T(k) @ dy[k] for k in range(1, N + 1)], [UA @ du[k] for k in range(N)]

    def cat(ax, au):
        return np.concatenate([a for a in ax[1:]] + list(au))

    def finish(u, it, status, mu=0.0):
        x = np.zeros((N + 1, n)); x[0] = p.x0
        for k in range(N):
            x[k + 1] = p.A[k] @ x[k] + p.B[k] @ u[k] + p.d[k]
        s = np.zeros(N + 1)
        J = p.objective(x, u, s) - (p.omega * 0.0)
        if p.tr:
            J = J - p.omega * np.sum(s)
        inside = True
        if p.tr:
            inside = bool(np.max(np.abs(p.xs * (x[1:] - p.xk[1:]))) <= p.delta)
        return x, u, J, dict(iters=it, status=status, mu=mu, inside=inside)

    u = np.zeros((N, m))
    y = outputs(u)
    zero_x = [None] + [np.zeros(xrows_T(k).shape[0]) for k in range(1, N + 1)]
    zero_u = [np.zeros(nU) for _ in range(N)]
    if ng == 0:
        du = newton_solve(zero_u, zero_x, -grad_parts(u, y, zero_x, zero_u))
        return finish(u + du.reshape(N, m), 0, 'optimal')
    if warm is not None:
        # the previous QP's point: slacks from this QP's rows, multipliers kept, both away from zero
        u = warm['u'].copy()
        y = outputs(u)
        gx, gu = row_vals(u, y)
        tx = [None] + [np.maximum(-g, WARM_FLOOR) for g in gx[1:]]; tu = [np.maximum(-g, WARM_FLOOR) for g in gu]
        lx = [None] + [np.maximum(l, WARM_FLOOR) for l in warm['lx'][1:]]; lu = [np.maximum(l, WARM_FLOOR) for l in warm['lu']]
    else:
        # starting point: unit weights, gradient shifts = row values (as riccati_ipm)
        gx, gu = row_vals(u, y)
        one_x = [None] + [np.ones_like(g) for g in gx[1:]]
        one_u = [np.ones(nU) for _ in range(N)]
        du = newton_solve(one_u, one_x, -grad_parts(u, y, gx, gu)).reshape(N, m)
        u = u + du
        y = outputs(u)
        gx, gu = row_vals(u, y)
        allg = cat(gx, gu)
        sh_t = (1.0 + allg.max()) if allg.max() >= 0 else 0.0
        sh_l = (1.0 - allg.min()) if allg.min() <= 0 else 0.0
        tx = [None] + [-g + sh_t for g in gx[1:]]; tu = [-g + sh_t for g in gu]
        lx = [None] + [g + sh_l for g in gx[1:]]; lu = [g + sh_l for g in gu]
    scale_d = max(1.0, p.omega, np.abs(p.grad_x(1, np.zeros(n))).max())
    scale_p = max(1.0, abs(p.delta), np.abs(Ub).max() if nU else 1.0)
    dreg = reg / scale_d
    status, mu, it = 'max_iter', 0.0, 0

    def maxstep(v, dv):
        neg = dv < 0
        return float(np.min(-v[neg] / dv[neg])) if neg.any() else np.inf
    try:                                  # a factorisation numpy refuses (non-finite or indefinite system) = the kernels' status 2
        for it in range(max_iter):
            gx, gu = row_vals(u, y)
            rgx = [None] + [gx[k] + tx[k] for k in range(1, N + 1)]
            rgu = [gu[k] + tu[k] for k in range(N)]
            mu = (sum(float(lx[k] @ tx[k]) for k in range(1, N + 1)) + sum(float(lu[k] @ tu[k]) for k in range(N))) / ng
            ex = [None] + [tx[k] + dreg * lx[k] for k in range(1, N + 1)]
            eu = [tu[k] + dreg * lu[k] for k in range(N)]
            Dx = [None] + [lx[k] / ex[k] for k in range(1, N + 1)]
            Du = [lu[k] / eu[k] for k in range(N)]
            rhox = [None] + [lx[k] + (lx[k] * rgx[k] - lx[k] * tx[k]) / ex[k] for k in range(1, N + 1)]
            rhou = [lu[k] + (lu[k] * rgu[k] - lu[k] * tu[k]) / eu[k] for k in range(N)]
            rd = float(np.abs(grad_parts(u, y, lx, lu)).max())
            rp = float(np.abs(cat(rgx, rgu)).max())
            if verbose:
                print(it, 'rd %.3e rp %.3e mu %.3e' % (rd, rp, mu))
            if rd <= max(tol, 1e-9) * scale_d and rp <= max(tol, 1e-9) * scale_p and mu <= tol:
                status = 'optimal'
                break
            du = newton_solve(Du, Dx, -grad_parts(u, y, rhox, rhou)).reshape(N, m)
            dy = direction_y(du)
            ax, au = row_dirs(du, dy)
            dlx = [None] + [(-lx[k] * tx[k] + lx[k] * (rgx[k] + ax[k])) / ex[k] for k in range(1, N + 1)]
            dlu = [(-lu[k] * tu[k] + lu[k] * (rgu[k] + au[k])) / eu[k] for k in range(N)]
            dtx = [None] + [-rgx[k] - ax[k] + dreg * dlx[k] for k in range(1, N + 1)]
            dtu = [-rgu[k] - au[k] + dreg * dlu[k] for k in range(N)]
            T, DT, Lm, DL = cat(tx, tu), cat(dtx, dtu), cat(lx, lu), cat(dlx, dlu)
            a_aff = min(1.0, maxstep(T, DT), maxstep(Lm, DL))
            mu_aff = float((Lm + a_aff * DL) @ (T + a_aff * DT)) / ng
            sigma = (mu_aff / mu) ** 3 if mu > 0 else 0.0
            rcx = [None] + [lx[k] * tx[k] + dtx[k] * dlx[k] - sigma * mu for k in range(1, N + 1)]
            rcu = [lu[k] * tu[k] + dtu[k] * dlu[k] - sigma * mu for k in range(N)]
            rhox = [None] + [lx[k] + (lx[k] * rgx[k] - rcx[k]) / ex[k] for k in range(1, N + 1)]
            rhou = [lu[k] + (lu[k] * rgu[k] - rcu[k]) / eu[k] for k in range(N)]
            du = newton_solve(Du, Dx, -grad_parts(u, y, rhox, rhou)).reshape(N, m)
            dy = direction_y(du)
            ax, au = row_dirs(du, dy)
            dlx = [None] + [(-rcx[k] + lx[k] * (rgx[k] + ax[k])) / ex[k] for k in range(1, N + 1)]
            dlu = [(-rcu[k] + lu[k] * (rgu[k] + au[k])) / eu[k] for k in range(N)]
            dtx = [None] + [-rgx[k] - ax[k] + dreg * dlx[k] for k in range(1, N + 1)]
            dtu = [-rgu[k] - au[k] + dreg * dlu[k] for k in range(N)]
            T, DT, Lm, DL = cat(tx, tu), cat(dtx, dtu), cat(lx, lu), cat(dlx, dlu)
            a = min(1.0, 0.99 * min(maxstep(T, DT), maxstep(Lm, DL)))
            u = u + a * du
            y = y + a * dy
            tx = [None] + [tx[k] + a * dtx[k] for k in range(1, N + 1)]
            tu = [tu[k] + a * dtu[k] for k in range(N)]
            lx = [None] + [lx[k] + a * dlx[k] for k in range(1, N + 1)]
            lu = [lu[k] + a * dlu[k] for k in range(N)]
            if not np.isfinite(mu):
                status = 'failed'
                break
    except np.linalg.LinAlgError:
        status = 'failed'
    if warm is not None and status != 'optimal':
        return solve(p, tol, max_iter, reg, newton, refine, verbose, warm=None, dy_from_system=dy_from_system)       # a warm start that stalls: again from Mehrotra's point
    res = finish(u, it, status, mu)
    res[3]['final'] = dict(u=u.copy(), lx=lx, lu=lu)
    res[3]['warm'] = warm is not None
    return res
