// CPU twin of the hot path -- TEST / BASELINE INFRASTRUCTURE ONLY (part of oracle/: never loaded by the product package;
// only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py use it).
//
// Native (C++17, no BLAS, IEEE: no -ffast-math) restatement of
//   * the POD projection            out = (X - ref) U                       sofacontrol/mor/pod.py:39-54
//   * the nearest-point TPWL model  argmin_i w_q |q_i - q| + w_v |v_i - v|  sofacontrol/tpwl/tpwl.py:160-168, 236-270
//   * the LOCP horizon QP           sofacontrol/scp/locp.py:218-342, solved like oracle/riccati_ipm.py: Mehrotra
//     predictor-corrector interior point, Newton systems by a backward Riccati recursion over the horizon, with the
//     trust-region prescreen of the device kernel (the QP without its 2n+1 trust-region rows per stage first; the full
//     QP only when that minimiser leaves the trust region)
//   * the GuSTO outer loop          sofacontrol/scp/gusto.py:283-487 (oracle/gusto.py)
//   * iLQR                          sofacontrol/lqr/ilqr.py:27-300 + lqr/config.py (oracle/lqr.py: ILQR / ILQRGeneric) on the
//     nearest-point TPWL model and on the SSM polynomial model (sofacontrol/SSM/ssm.py:158-301, oracle/ssm.py)
// so that bench.py can put an honest native number next to the GPU one: single thread, and all cores with one rollout
// per thread (`threads` = the number actually used).  tests/test_cpu_twin.py holds it to the numpy oracle.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

using vec = std::vector<double>;
constexpr double INF = std::numeric_limits<double>::infinity();

// ---- small dense helpers (row-major)
inline void matvec(const double *A, int r, int c, const double *x, double *y) {          // y = A x
    for (int i = 0; i < r; ++i) {
        double s = 0.0;
        const double *a = A + (size_t)i * c;
        for (int j = 0; j < c; ++j) s += a[j] * x[j];
        y[i] = s;
    }
}
inline void matTvec_add(const double *A, int r, int c, const double *x, double *y) {     // y += A^T x
    for (int i = 0; i < r; ++i) {
        const double xi = x[i];
        const double *a = A + (size_t)i * c;
        for (int j = 0; j < c; ++j) y[j] += a[j] * xi;
    }
}
// C (r x c) = A (r x k) B (k x c)
inline void matmul(const double *A, const double *B, int r, int k, int c, double *C) {
    std::fill(C, C + (size_t)r * c, 0.0);
    for (int i = 0; i < r; ++i)
        for (int p = 0; p < k; ++p) {
            const double a = A[(size_t)i * k + p];
            const double *b = B + (size_t)p * c;
            double *cr = C + (size_t)i * c;
            for (int j = 0; j < c; ++j) cr[j] += a * b[j];
        }
}
// C (c1 x c2) = A^T B with A (r x c1), B (r x c2)
inline void matTmul(const double *A, const double *B, int r, int c1, int c2, double *C) {
    std::fill(C, C + (size_t)c1 * c2, 0.0);
    for (int p = 0; p < r; ++p) {
        const double *a = A + (size_t)p * c1, *b = B + (size_t)p * c2;
        for (int i = 0; i < c1; ++i) {
            const double ai = a[i];
            double *cr = C + (size_t)i * c2;
            for (int j = 0; j < c2; ++j) cr[j] += ai * b[j];
        }
    }
}
// in-place lower Cholesky of an m x m SPD matrix; false if not positive definite
inline bool cholesky(double *A, int m) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * m + j];
            for (int k = 0; k < j; ++k) s -= A[i * m + k] * A[j * m + k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                A[i * m + i] = std::sqrt(s);
            } else {
                A[i * m + j] = s / A[j * m + j];
            }
        }
    return true;
}
// X (m x c) <- -(L L^T)^-1 B  column by column
inline void chol_solve_neg(const double *L, int m, const double *B, int c, double *X) {
    vec y(m);
    for (int col = 0; col < c; ++col) {
        for (int i = 0; i < m; ++i) {
            double s = B[(size_t)i * c + col];
            for (int k = 0; k < i; ++k) s -= L[i * m + k] * y[k];
            y[i] = s / L[i * m + i];
        }
        for (int i = m - 1; i >= 0; --i) {
            double s = y[i];
            for (int k = i + 1; k < m; ++k) s -= L[k * m + i] * y[k];
            y[i] = s / L[i * m + i];
        }
        for (int i = 0; i < m; ++i) X[(size_t)i * c + col] = -y[i];
    }
}

struct Model {          // nearest-point TPWL tables (oracle/tpwl.py dict)
    int P, r, n, m;
    double w_q, w_v;
    const double *q, *v;            // (P x r)
    const double *Ac, *Bc, *dc;     // continuous (P x n x n), (P x n x m), (P x n)
    const double *Ad, *Bd, *dd;     // discrete tables
};

int nearest(const Model &M, const double *x) {
    int best = 0;
    double bd = INF;
    for (int i = 0; i < M.P; ++i) {
        double sq = 0.0, sv = 0.0;
        for (int j = 0; j < M.r; ++j) { const double e = M.q[(size_t)i * M.r + j] - x[M.r + j]; sq += e * e; }
        for (int j = 0; j < M.r; ++j) { const double e = M.v[(size_t)i * M.r + j] - x[j]; sv += e * e; }
        const double d = M.w_q * std::sqrt(sq) + M.w_v * std::sqrt(sv);
        if (d < bd) { bd = d; best = i; }
    }
    return best;
}

struct Problem {        // the QP data (oracle/riccati_ipm.py: Problem)
    int N, n, m, nz, nU, nX, nXf;
    const double *H, *Qz, *R, *Qzf;                 // Qzf may be null
    const double *UA, *Ub, *XA, *Xb, *XfA, *Xfb;
    const double *xs;                               // (n) trust-region scaling
    vec Qx, QxN, Ru;                                // 2 H^T Qz H (+ terminal), 2 R
    std::vector<const double *> A, B, d;            // stage dynamics (N pointers)
    const double *x0, *xk;                          // xk (N+1 x n)
    const double *z, *zf, *ud;                      // z (N+1 x nz) or null, zf (nz) or null, ud (N x m) or null
    double delta, omega;
    bool tr;
};

void problem_consts(Problem &p) {
    const int n = p.n, nz = p.nz, m = p.m;
    vec QH((size_t)nz * n), QfH((size_t)nz * n, 0.0);
    matmul(p.Qz, p.H, nz, nz, n, QH.data());
    p.Qx.assign((size_t)n * n, 0.0); p.QxN.assign((size_t)n * n, 0.0);
    matTmul(p.H, QH.data(), nz, n, n, p.Qx.data());
    for (auto &e : p.Qx) e *= 2.0;
    p.QxN = p.Qx;
    if (p.Qzf) {
        matmul(p.Qzf, p.H, nz, nz, n, QfH.data());
        vec t((size_t)n * n);
        matTmul(p.H, QfH.data(), nz, n, n, t.data());
        for (size_t e = 0; e < t.size(); ++e) p.QxN[e] += 2.0 * t[e];
    }
    p.Ru.assign(p.R, p.R + (size_t)m * m);
    for (auto &e : p.Ru) e *= 2.0;
}

void grad_x(const Problem &p, int k, const double *x, double *g) {          // riccati_ipm.Problem.grad_x
    const int n = p.n, nz = p.nz;
    matvec(p.Qx.data(), n, n, x, g);
    vec t(nz, 0.0), t2(nz);
    if (p.z) {
        matvec(p.Qz, nz, nz, p.z + (size_t)k * nz, t.data());
        for (int i = 0; i < n; ++i) { double s = 0.0; for (int a = 0; a < nz; ++a) s += p.H[(size_t)a * n + i] * t[a]; g[i] -= 2.0 * s; }
    }
    if (k == p.N && p.Qzf) {
        vec e(nz);
        matvec(p.H, nz, n, x, e.data());
        for (int a = 0; a < nz; ++a) e[a] -= p.zf ? p.zf[a] : 0.0;
        matvec(p.Qzf, nz, nz, e.data(), t2.data());
        for (int i = 0; i < n; ++i) { double s = 0.0; for (int a = 0; a < nz; ++a) s += p.H[(size_t)a * n + i] * t2[a]; g[i] += 2.0 * s; }
    }
}

double objective(const Problem &p, const vec &x, const vec &u, const vec &s) {
    const int N = p.N, n = p.n, m = p.m, nz = p.nz;
    double J = 0.0;
    vec e(nz), t(nz);
    for (int k = 0; k <= N; ++k) {
        matvec(p.H, nz, n, &x[(size_t)k * n], e.data());
        if (p.z) for (int a = 0; a < nz; ++a) e[a] -= p.z[(size_t)k * nz + a];
        matvec(p.Qz, nz, nz, e.data(), t.data());
        for (int a = 0; a < nz; ++a) J += e[a] * t[a];
    }
    if (p.Qzf) {
        matvec(p.H, nz, n, &x[(size_t)N * n], e.data());
        for (int a = 0; a < nz; ++a) e[a] -= p.zf ? p.zf[a] : 0.0;
        matvec(p.Qzf, nz, nz, e.data(), t.data());
        for (int a = 0; a < nz; ++a) J += e[a] * t[a];
    }
    vec ue(m), tu(m);
    for (int k = 0; k < N; ++k) {
        for (int a = 0; a < m; ++a) ue[a] = u[(size_t)k * m + a] - (p.ud ? p.ud[(size_t)k * m + a] : 0.0);
        matvec(p.R, m, m, ue.data(), tu.data());
        for (int a = 0; a < m; ++a) J += ue[a] * tu[a];
    }
    if (p.tr) for (int k = 0; k <= N; ++k) J += p.omega * s[k];
    return J;
}

// inequality rows owned by x_k (k >= 1): Ax x + as s <= h   (riccati_ipm._rows_x)
struct XRows { int nr; vec Ax, as, h; };
XRows rows_x(const Problem &p, int k) {
    const int n = p.n;
    XRows r;
    r.nr = (p.tr ? 2 * n + 1 : 0) + p.nX + (k == p.N ? p.nXf : 0);
    r.Ax.assign((size_t)r.nr * n, 0.0); r.as.assign(r.nr, 0.0); r.h.assign(r.nr, 0.0);
    int o = 0;
    if (p.tr) {
        for (int i = 0; i < n; ++i) { r.Ax[(size_t)(o + i) * n + i] = p.xs[i]; r.as[o + i] = -1.0; r.h[o + i] = p.delta + p.xs[i] * p.xk[(size_t)k * n + i]; }
        o += n;
        for (int i = 0; i < n; ++i) { r.Ax[(size_t)(o + i) * n + i] = -p.xs[i]; r.as[o + i] = -1.0; r.h[o + i] = p.delta - p.xs[i] * p.xk[(size_t)k * n + i]; }
        o += n;
        r.as[o] = -1.0; o += 1;
    }
    for (int i = 0; i < p.nX; ++i) { std::copy(p.XA + (size_t)i * n, p.XA + (size_t)(i + 1) * n, &r.Ax[(size_t)(o + i) * n]); r.h[o + i] = p.Xb[i]; }
    o += p.nX;
    if (k == p.N) for (int i = 0; i < p.nXf; ++i) { std::copy(p.XfA + (size_t)i * n, p.XfA + (size_t)(i + 1) * n, &r.Ax[(size_t)(o + i) * n]); r.h[o + i] = p.Xfb[i]; }
    return r;
}

struct Info { int iters; int status; double mu; };      // status 0 optimal, 1 max_iter, 2 failed

// oracle/riccati_ipm.py: solve().  x (N+1 x n), u (N x m), s (N+1).
Info ipm_solve(const Problem &p, vec &x, vec &u, vec &s, double *J_out, double tol = 1e-12, int max_iter = 60, double reg = 1e-8) {
    const int N = p.N, n = p.n, m = p.m, nU = p.nU;
    std::vector<XRows> rows(N + 1);
    for (int k = 1; k <= N; ++k) rows[k] = rows_x(p, k);
    int ng = N * nU;
    for (int k = 1; k <= N; ++k) ng += rows[k].nr;
    auto rollout = [&](const vec &uu, vec &xx) {
        std::copy(p.x0, p.x0 + n, xx.begin());
        vec t(n);
        for (int k = 0; k < N; ++k) {
            matvec(p.A[k], n, n, &xx[(size_t)k * n], t.data());
            for (int i = 0; i < n; ++i) {
                double sB = 0.0;
                for (int a = 0; a < m; ++a) sB += p.B[k][(size_t)i * m + a] * uu[(size_t)k * m + a];
                xx[(size_t)(k + 1) * n + i] = t[i] + sB + p.d[k][i];
            }
        }
    };
    // per-row state: x rows per stage, u rows per stage
    std::vector<vec> tx(N + 1), lx(N + 1), Dx(N + 1), rhox(N + 1), rgx(N + 1), dtx(N + 1), dlx(N + 1), rcx(N + 1), ex(N + 1);
    std::vector<vec> tu(N), lu(N), Du(N), rhou(N), rgu(N), dtu(N), dlu(N), rcu(N), eu(N);
    for (int k = 1; k <= N; ++k) for (auto *v : {&tx[k], &lx[k], &Dx[k], &rhox[k], &rgx[k], &dtx[k], &dlx[k], &rcx[k], &ex[k]}) v->assign(rows[k].nr, 0.0);
    for (int k = 0; k < N; ++k) for (auto *v : {&tu[k], &lu[k], &Du[k], &rhou[k], &rgu[k], &dtu[k], &dlu[k], &rcu[k], &eu[k]}) v->assign(nU, 0.0);
    // storage of the factorisation
    vec K((size_t)N * m * n), kff((size_t)N * m), Lq((size_t)N * m * m), dx((size_t)(N + 1) * n), du((size_t)N * m), ds(N + 1);
    std::vector<vec> celim_c(N + 1);
    vec celim_H(N + 1, 1.0), celim_g(N + 1, 0.0);

    // Newton system by a backward Riccati recursion.  use_lam: also the reduced dual residual with the multipliers.
    auto newton = [&](bool factor, bool use_lam, double *rd_out) -> bool {
        vec P((size_t)n * n), pv(n), adj(n), Hxx((size_t)n * n), gx(n), gxd(n), W((size_t)n * n), G((size_t)n * m), Quu((size_t)m * m),
            Qux((size_t)m * n), Qu(m), gu(m), gud(m), t1(n), Pn((size_t)n * n), t2(std::max(n, m));
        double rd = 0.0;
        for (int k = N; k >= 0; --k) {
            if (k >= 1) {
                const XRows &r = rows[k];
                grad_x(p, k, &x[(size_t)k * n], gx.data());
                gxd = gx;
                matTvec_add(r.Ax.data(), r.nr, n, rhox[k].data(), gx.data());
                if (use_lam) {
                    matTvec_add(r.Ax.data(), r.nr, n, lx[k].data(), gxd.data());
                    if (p.tr) { double sl = p.omega; for (int i = 0; i < r.nr; ++i) sl += r.as[i] * lx[k][i]; rd = std::max(rd, std::fabs(sl)); }
                } else {
                    gxd = gx;
                }
                if (factor) {
                    Hxx = (k == N) ? p.QxN : p.Qx;
                    for (int i = 0; i < r.nr; ++i) {           // Ax^T D Ax (rows are sparse for the trust region: skip zeros)
                        const double dw = Dx[k][i];
                        const double *a = &r.Ax[(size_t)i * n];
                        for (int c1 = 0; c1 < n; ++c1) {
                            if (a[c1] == 0.0) continue;
                            const double f = dw * a[c1];
                            for (int c2 = 0; c2 < n; ++c2) Hxx[(size_t)c1 * n + c2] += f * a[c2];
                        }
                    }
                }
                if (p.tr) {
                    double gs = p.omega, Hss = 0.0;
                    vec c(n, 0.0);
                    for (int i = 0; i < r.nr; ++i) {
                        gs += r.as[i] * rhox[k][i];
                        Hss += r.as[i] * Dx[k][i] * r.as[i];
                        const double f = Dx[k][i] * r.as[i];
                        if (f != 0.0) for (int c1 = 0; c1 < n; ++c1) c[c1] += r.Ax[(size_t)i * n + c1] * f;
                    }
                    celim_c[k] = c; celim_H[k] = Hss; celim_g[k] = gs;
                    if (factor) {
                        // eliminate s_k; the diagonal of diag(hd) - c c^T / Hss without cancellation (see the port)
                        for (int i = 0; i < n; ++i)
                            for (int j = 0; j < n; ++j) if (i != j) Hxx[(size_t)i * n + j] -= c[i] * c[j] / Hss;
                        for (int i = 0; i < n; ++i) {
                            const double Dp = Dx[k][i], Dm = Dx[k][n + i];
                            const double naive = p.xs[i] * p.xs[i] * (Dp + Dm);
                            const double stable = p.xs[i] * p.xs[i] * ((Dp + Dm) * (Hss - (Dp + Dm)) + 4.0 * Dp * Dm) / Hss;
                            Hxx[(size_t)i * n + i] += stable - naive;
                        }
                    }
                    for (int i = 0; i < n; ++i) gx[i] -= c[i] * gs / Hss;
                }
            }
            if (k == N) {
                if (factor) P = Hxx;
                pv = gx; adj = gxd;
                continue;
            }
            // input stage k
            for (int a = 0; a < m; ++a) {
                double s1 = 0.0;
                for (int b = 0; b < m; ++b) s1 += p.Ru[(size_t)a * m + b] * (u[(size_t)k * m + b] - (p.ud ? p.ud[(size_t)k * m + b] : 0.0));
                gu[a] = s1; gud[a] = s1;
            }
            for (int r = 0; r < nU; ++r)
                for (int a = 0; a < m; ++a) {
                    gu[a] += p.UA[(size_t)r * m + a] * rhou[k][r];
                    if (use_lam) gud[a] += p.UA[(size_t)r * m + a] * lu[k][r];
                }
            if (!use_lam) gud = gu;
            const double *A = p.A[k], *B = p.B[k];
            // Qu = gu + B^T pv ; reduced gradient wrt u_k with the multipliers
            for (int a = 0; a < m; ++a) {
                double s1 = gu[a], s2 = gud[a];
                for (int i = 0; i < n; ++i) { s1 += B[(size_t)i * m + a] * pv[i]; s2 += B[(size_t)i * m + a] * adj[i]; }
                Qu[a] = s1;
                rd = std::max(rd, std::fabs(s2));
            }
            double *Kk = &K[(size_t)k * m * n], *Lk = &Lq[(size_t)k * m * m];
            if (factor) {
                matmul(P.data(), A, n, n, n, W.data());
                matmul(P.data(), B, n, n, m, G.data());
                matTmul(B, G.data(), n, m, m, Quu.data());
                for (int e = 0; e < m * m; ++e) Quu[e] += p.Ru[e];
                for (int r = 0; r < nU; ++r) {
                    const double dw = Du[k][r];
                    for (int a = 0; a < m; ++a) for (int b = 0; b < m; ++b) Quu[(size_t)a * m + b] += p.UA[(size_t)r * m + a] * dw * p.UA[(size_t)r * m + b];
                }
                matTmul(B, W.data(), n, m, n, Qux.data());
                std::copy(Quu.begin(), Quu.end(), Lk);
                if (!cholesky(Lk, m)) return false;
                chol_solve_neg(Lk, m, Qux.data(), n, Kk);
            }
            chol_solve_neg(Lk, m, Qu.data(), 1, &kff[(size_t)k * m]);
            if (k >= 1) {
                if (factor) {
                    // P = Hxx + A^T W + Qux^T K, symmetrised
                    matTmul(A, W.data(), n, n, n, Pn.data());
                    for (int i = 0; i < n; ++i)
                        for (int j = 0; j < n; ++j) {
                            double s1 = Hxx[(size_t)i * n + j] + Pn[(size_t)i * n + j];
                            for (int a = 0; a < m; ++a) s1 += Qux[(size_t)a * n + i] * Kk[(size_t)a * n + j];
                            P[(size_t)i * n + j] = s1;
                        }
                    for (int i = 0; i < n; ++i)
                        for (int j = i + 1; j < n; ++j) { const double v = 0.5 * (P[(size_t)i * n + j] + P[(size_t)j * n + i]); P[(size_t)i * n + j] = P[(size_t)j * n + i] = v; }
                }
                // pv = gx + A^T pv + K^T Qu ; adj = gxd + A^T adj
                std::fill(t1.begin(), t1.end(), 0.0);
                matTvec_add(A, n, n, pv.data(), t1.data());
                for (int j = 0; j < n; ++j) { double s1 = gx[j] + t1[j]; for (int a = 0; a < m; ++a) s1 += Kk[(size_t)a * n + j] * Qu[a]; t1[j] = s1; }
                std::fill(t2.begin(), t2.end(), 0.0);
                matTvec_add(A, n, n, adj.data(), t2.data());
                for (int j = 0; j < n; ++j) adj[j] = gxd[j] + t2[j];
                pv = t1;
            }
        }
        // forward
        std::fill(dx.begin(), dx.end(), 0.0);
        ds.assign(N + 1, 0.0);
        vec t(n);
        for (int k = 0; k < N; ++k) {
            const double *Kk = &K[(size_t)k * m * n];
            for (int a = 0; a < m; ++a) {
                double s1 = kff[(size_t)k * m + a];
                for (int j = 0; j < n; ++j) s1 += Kk[(size_t)a * n + j] * dx[(size_t)k * n + j];
                du[(size_t)k * m + a] = s1;
            }
            matvec(p.A[k], n, n, &dx[(size_t)k * n], t.data());
            for (int i = 0; i < n; ++i) {
                double sB = 0.0;
                for (int a = 0; a < m; ++a) sB += p.B[k][(size_t)i * m + a] * du[(size_t)k * m + a];
                dx[(size_t)(k + 1) * n + i] = t[i] + sB;
            }
            if (p.tr) {
                double cd = 0.0;
                for (int i = 0; i < n; ++i) cd += celim_c[k + 1][i] * dx[(size_t)(k + 1) * n + i];
                ds[k + 1] = -(celim_g[k + 1] + cd) / celim_H[k + 1];
            }
        }
        if (rd_out) *rd_out = rd;
        return true;
    };
    auto row_vals = [&](const vec &xx, const vec &uu, const vec &ss, std::vector<vec> &gxo, std::vector<vec> &guo) {
        for (int k = 1; k <= N; ++k) {
            const XRows &r = rows[k];
            for (int i = 0; i < r.nr; ++i) {
                double s1 = r.as[i] * ss[k] - r.h[i];
                const double *a = &r.Ax[(size_t)i * n];
                for (int j = 0; j < n; ++j) s1 += a[j] * xx[(size_t)k * n + j];
                gxo[k][i] = s1;
            }
        }
        for (int k = 0; k < N; ++k)
            for (int r = 0; r < nU; ++r) {
                double s1 = -p.Ub[r];
                for (int a = 0; a < m; ++a) s1 += p.UA[(size_t)r * m + a] * uu[(size_t)k * m + a];
                guo[k][r] = s1;
            }
    };
    auto row_dirs = [&](std::vector<vec> &axo, std::vector<vec> &auo) {
        for (int k = 1; k <= N; ++k) {
            const XRows &r = rows[k];
            for (int i = 0; i < r.nr; ++i) {
                double s1 = r.as[i] * ds[k];
                const double *a = &r.Ax[(size_t)i * n];
                for (int j = 0; j < n; ++j) s1 += a[j] * dx[(size_t)k * n + j];
                axo[k][i] = s1;
            }
        }
        for (int k = 0; k < N; ++k)
            for (int r = 0; r < nU; ++r) {
                double s1 = 0.0;
                for (int a = 0; a < m; ++a) s1 += p.UA[(size_t)r * m + a] * du[(size_t)k * m + a];
                auo[k][r] = s1;
            }
    };
    auto s0 = [&]() {
        if (!p.tr) return 0.0;
        double v = 0.0;
        for (int i = 0; i < n; ++i) v = std::max(v, std::fabs(p.xs[i] * (p.x0[i] - p.xk[i])));
        return std::max(0.0, v - p.delta);
    };
    x.assign((size_t)(N + 1) * n, 0.0); u.assign((size_t)N * m, 0.0); s.assign(N + 1, 0.0);
    s[0] = s0();
    rollout(u, x);
    Info info{0, 1, 0.0};
    std::vector<vec> gx(N + 1), gu(N), ax(N + 1), au(N);
    for (int k = 1; k <= N; ++k) { gx[k].assign(rows[k].nr, 0.0); ax[k].assign(rows[k].nr, 0.0); }
    for (int k = 0; k < N; ++k) { gu[k].assign(nU, 0.0); au[k].assign(nU, 0.0); }
    if (ng == 0) {
        if (!newton(true, false, nullptr)) { info.status = 2; return info; }
        for (size_t e = 0; e < u.size(); ++e) u[e] += du[e];
        rollout(u, x);
        if (J_out) *J_out = objective(p, x, u, s);
        info.status = 0;
        return info;
    }
    // starting point: unit weights, gradient shifts = row values, then shift into the positive orthant
    row_vals(x, u, s, gx, gu);
    for (int k = 1; k <= N; ++k) { std::fill(Dx[k].begin(), Dx[k].end(), 1.0); rhox[k] = gx[k]; }
    for (int k = 0; k < N; ++k) { std::fill(Du[k].begin(), Du[k].end(), 1.0); rhou[k] = gu[k]; }
    if (!newton(true, false, nullptr)) { info.status = 2; return info; }
    for (size_t e = 0; e < x.size(); ++e) x[e] += dx[e];
    for (size_t e = 0; e < u.size(); ++e) u[e] += du[e];
    for (int k = 0; k <= N; ++k) s[k] += ds[k];
    s[0] = s0();
    row_vals(x, u, s, gx, gu);
    double zmin = INF, zmax = -INF;
    for (int k = 1; k <= N; ++k) for (double g : gx[k]) { zmin = std::min(zmin, g); zmax = std::max(zmax, g); }
    for (int k = 0; k < N; ++k) for (double g : gu[k]) { zmin = std::min(zmin, g); zmax = std::max(zmax, g); }
    const double sh_t = zmax >= 0.0 ? 1.0 + zmax : 0.0, sh_l = zmin <= 0.0 ? 1.0 - zmin : 0.0;
    for (int k = 1; k <= N; ++k) for (int i = 0; i < rows[k].nr; ++i) { tx[k][i] = -gx[k][i] + sh_t; lx[k][i] = gx[k][i] + sh_l; }
    for (int k = 0; k < N; ++k) for (int i = 0; i < nU; ++i) { tu[k][i] = -gu[k][i] + sh_t; lu[k][i] = gu[k][i] + sh_l; }
    vec g1(n), zero(n, 0.0);
    grad_x(p, 1, zero.data(), g1.data());
    double scale_d = std::max(1.0, p.omega), scale_p = std::max(1.0, std::fabs(p.delta));
    for (double g : g1) scale_d = std::max(scale_d, std::fabs(g));
    for (int r = 0; r < nU; ++r) scale_p = std::max(scale_p, std::fabs(p.Ub[r]));
    const double dreg = reg / scale_d;
    auto each_row = [&](auto f) {            // f(t, lam, rg, D, rho, rc, dt, dlam, e, a) over every row
        for (int k = 1; k <= N; ++k) for (int i = 0; i < rows[k].nr; ++i) f(tx[k][i], lx[k][i], rgx[k][i], Dx[k][i], rhox[k][i], rcx[k][i], dtx[k][i], dlx[k][i], ex[k][i], ax[k][i], gx[k][i]);
        for (int k = 0; k < N; ++k) for (int i = 0; i < nU; ++i) f(tu[k][i], lu[k][i], rgu[k][i], Du[k][i], rhou[k][i], rcu[k][i], dtu[k][i], dlu[k][i], eu[k][i], au[k][i], gu[k][i]);
    };
    double mu = 0.0;
    int it = 0;
    for (it = 0; it < max_iter; ++it) {
        row_vals(x, u, s, gx, gu);
        double musum = 0.0, rp = 0.0;
        each_row([&](double &t, double &lam, double &rg, double &D, double &rho, double &, double &, double &, double &e, double &, double &g) {
            rg = g + t;
            musum += lam * t;
            e = t + dreg * lam;
            D = lam / e;
            rho = lam + (lam * rg - lam * t) / e;
            rp = std::max(rp, std::fabs(rg));
        });
        mu = musum / ng;
        double rd = 0.0;
        if (!newton(true, true, &rd)) { info.status = 2; break; }
        if (rd <= std::max(tol, 1e-9) * scale_d && rp <= std::max(tol, 1e-9) * scale_p && mu <= tol) { info.status = 0; break; }
        row_dirs(ax, au);
        double a_aff = 1.0;
        each_row([&](double &t, double &lam, double &rg, double &, double &, double &, double &dt, double &dl, double &e, double &a, double &) {
            dl = (-lam * t + lam * (rg + a)) / e;
            dt = -rg - a + dreg * dl;
            if (dt < 0.0) a_aff = std::min(a_aff, -t / dt);
            if (dl < 0.0) a_aff = std::min(a_aff, -lam / dl);
        });
        double ma = 0.0;
        each_row([&](double &t, double &lam, double &, double &, double &, double &, double &dt, double &dl, double &, double &, double &) { ma += (lam + a_aff * dl) * (t + a_aff * dt); });
        const double mu_aff = ma / ng;
        const double sigma = mu > 0.0 ? std::pow(mu_aff / mu, 3.0) : 0.0;
        each_row([&](double &t, double &lam, double &rg, double &, double &rho, double &rc, double &dt, double &dl, double &e, double &, double &) {
            rc = lam * t + dt * dl - sigma * mu;
            rho = lam + (lam * rg - rc) / e;
        });
        if (!newton(false, false, nullptr)) { info.status = 2; break; }
        row_dirs(ax, au);
        double amax = INF;
        each_row([&](double &t, double &lam, double &rg, double &, double &, double &rc, double &dt, double &dl, double &e, double &a, double &) {
            dl = (-rc + lam * (rg + a)) / e;
            dt = -rg - a + dreg * dl;
            if (dt < 0.0) amax = std::min(amax, -t / dt);
            if (dl < 0.0) amax = std::min(amax, -lam / dl);
        });
        const double a = std::min(1.0, 0.99 * amax);
        for (size_t e = 0; e < x.size(); ++e) x[e] += a * dx[e];
        for (size_t e = 0; e < u.size(); ++e) u[e] += a * du[e];
        for (int k = 0; k <= N; ++k) s[k] += a * ds[k];
        s[0] = s0();
        each_row([&](double &t, double &lam, double &, double &, double &, double &, double &dt, double &dl, double &, double &, double &) { t += a * dt; lam += a * dl; });
        if (!std::isfinite(mu)) { info.status = 2; break; }
    }
    rollout(u, x);
    if (J_out) *J_out = objective(p, x, u, s);
    info.iters = it; info.mu = mu;
    return info;
}


// ------------------------------------------------------------------ condensed (output-space) interior point
// oracle/condensed_ipm.py (newton = 'output'), i.e. the algorithm of the device kernel csrc/locp_cond.h / gusto_cond.hip:
// the QP WITHOUT its trust-region rows with the states eliminated, Newton systems through the Woodbury form
// K = I + Gd Gd^T (N p_o square).  Plain loops over the causal (block lower-triangular) structure of G, no BLAS.

// eigen-decomposition of a small symmetric matrix (cyclic Jacobi): A = V diag(w) V^T, eigenvectors in the columns of V
void jacobi_eig(vec A, int n, vec &w, vec &V) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) off += A[(size_t)a * n + b] * A[(size_t)a * n + b];
        if (off < 1e-300) break;
        for (int a = 0; a < n; ++a)
            for (int b = a + 1; b < n; ++b) {
                if (std::fabs(A[(size_t)a * n + b]) < 1e-300) continue;
                const double th = (A[(size_t)b * n + b] - A[(size_t)a * n + a]) / (2.0 * A[(size_t)a * n + b]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) {
                    const double ka = A[(size_t)k * n + a], kb = A[(size_t)k * n + b];
                    A[(size_t)k * n + a] = cs * ka - sn * kb; A[(size_t)k * n + b] = sn * ka + cs * kb;
                }
                for (int k = 0; k < n; ++k) {
                    const double ak = A[(size_t)a * n + k], bk = A[(size_t)b * n + k];
                    A[(size_t)a * n + k] = cs * ak - sn * bk; A[(size_t)b * n + k] = sn * ak + cs * bk;
                }
                for (int k = 0; k < n; ++k) {
                    const double ka = V[(size_t)k * n + a], kb = V[(size_t)k * n + b];
                    V[(size_t)k * n + a] = cs * ka - sn * kb; V[(size_t)k * n + b] = sn * ka + cs * kb;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
}

struct CondBasis {          // condensed_ipm.output_basis: rows of C_o and everything expressed in them
    int po = 0;
    bool ok = false;
    vec Co, Sc, ScN, Tx, Txf;          // (po x n), (po x po), (po x po), (nX x po), (nXf x po)
};

CondBasis cond_basis(const Problem &p) {
    const int n = p.n, nz = p.nz;
    CondBasis cb;
    auto sqrt_rows = [&](const double *Q) {
        vec out;
        if (!Q) return out;
        vec Qs((size_t)nz * nz), w, V;
        for (int a = 0; a < nz; ++a) for (int b = 0; b < nz; ++b) Qs[(size_t)a * nz + b] = 0.5 * (Q[(size_t)a * nz + b] + Q[(size_t)b * nz + a]);
        jacobi_eig(Qs, nz, w, V);
        double wmax = 0.0;
        for (double v : w) wmax = std::max(wmax, std::fabs(v));
        for (int e = 0; e < nz; ++e) {
            if (!(w[e] > 1e-13 * std::max(1e-300, wmax))) continue;
            const double sc = std::sqrt(2.0 * w[e]);
            for (int j = 0; j < n; ++j) {
                double v = 0.0;
                for (int a = 0; a < nz; ++a) v += V[(size_t)a * nz + e] * p.H[(size_t)a * n + j];
                out.push_back(sc * v);
            }
        }
        return out;
    };
    const vec Cq = sqrt_rows(p.Qz), Cqf = sqrt_rows(p.Qzf);
    const int ncq = (int)(Cq.size() / n), ncqf = (int)(Cqf.size() / n), rows = ncq + ncqf + p.nX + p.nXf;
    if (rows == 0) return cb;
    auto rowp = [&](int r) -> const double * {
        if (r < ncq) return &Cq[(size_t)r * n];
        r -= ncq;
        if (r < ncqf) return &Cqf[(size_t)r * n];
        r -= ncqf;
        if (r < p.nX) return p.XA + (size_t)r * n;
        return p.XfA + (size_t)(r - p.nX) * n;
    };
    vec Gm((size_t)n * n, 0.0), w, V;
    for (int r = 0; r < rows; ++r) {
        const double *row = rowp(r);
        double nr2 = 0.0;
        for (int j = 0; j < n; ++j) nr2 += row[j] * row[j];
        if (nr2 <= 0.0) continue;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Gm[(size_t)i * n + j] += row[i] * row[j] / nr2;
    }
    jacobi_eig(Gm, n, w, V);
    std::vector<int> ord(n);
    for (int i = 0; i < n; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return w[a] > w[b]; });
    int po = 0;
    while (po < n && w[ord[po]] > 1e-12 * w[ord[0]] && w[ord[po]] > 0.0) ++po;
    cb.po = po;
    cb.Co.resize((size_t)po * n);
    for (int a = 0; a < po; ++a) for (int j = 0; j < n; ++j) cb.Co[(size_t)a * n + j] = V[(size_t)j * n + ord[a]];
    auto project = [&](const double *src, int nr) {
        vec T((size_t)nr * po, 0.0);
        for (int r = 0; r < nr; ++r)
            for (int a = 0; a < po; ++a) {
                double v = 0.0;
                for (int j = 0; j < n; ++j) v += src[(size_t)r * n + j] * cb.Co[(size_t)a * n + j];
                T[(size_t)r * po + a] = v;
            }
        return T;
    };
    const vec Tc = project(Cq.data(), ncq), Tcf = project(Cqf.data(), ncqf);
    cb.Tx = project(p.XA, p.nX); cb.Txf = project(p.XfA, p.nXf);
    cb.Sc.assign((size_t)po * po, 0.0); cb.ScN.assign((size_t)po * po, 0.0);
    for (int a = 0; a < po; ++a)
        for (int b = 0; b < po; ++b) {
            double v = 0.0, vf = 0.0;
            for (int r = 0; r < ncq; ++r) v += Tc[(size_t)r * po + a] * Tc[(size_t)r * po + b];
            for (int r = 0; r < ncqf; ++r) vf += Tcf[(size_t)r * po + a] * Tcf[(size_t)r * po + b];
            cb.Sc[(size_t)a * po + b] = v; cb.ScN[(size_t)a * po + b] = v + vf;
        }
    cb.ok = po >= 1 && po <= 8;
    return cb;
}

// lower Cholesky factor of a positive SEMI-definite matrix (condensed_ipm.chol_psd): cancelled pivots get a zero column
inline void chol_psd(double *S, int n) {
    double dmax = 1e-300;
    for (int i = 0; i < n; ++i) dmax = std::max(dmax, std::fabs(S[(size_t)i * n + i]));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double v = S[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) v -= S[(size_t)i * n + k] * S[(size_t)j * n + k];
            if (i == j) S[(size_t)i * n + i] = v > 1e-14 * dmax ? std::sqrt(v) : 0.0;
            else S[(size_t)i * n + j] = S[(size_t)j * n + j] > 0.0 ? v / S[(size_t)j * n + j] : 0.0;
        }
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) S[(size_t)i * n + j] = 0.0;
}

// The QP of p without its trust-region rows.  x, u, s as ipm_solve; *inside: the minimiser satisfies p's trust region.
// the iterate a converged condensed solve leaves for the next QP of the same SCP solve (condensed_ipm.solve: warm)
struct Warm { bool valid = false; vec u, lam; };
constexpr double WARM_FLOOR = 1e-2;

Info cond_solve(const Problem &p, const CondBasis &cb, vec &x, vec &u, vec &s, double *J_out, bool *inside_out,
                double tol = 1e-12, int max_iter = 60, double reg = 1e-8, Warm *ws = nullptr) {
    const int N = p.N, n = p.n, m = p.m, nz = p.nz, nU = p.nU, nX = p.nX, nXf = p.nXf, po = cb.po;
    const int NP = N * po, NM = N * m, ng = N * nU + N * nX + nXf;
    auto nrx = [&](int k) { return nX + (k == N ? nXf : 0); };                  // state rows of stage k = 1..N
    auto Trow = [&](int k, int r) { return r < nX ? &cb.Tx[(size_t)r * po] : &cb.Txf[(size_t)(r - nX) * po]; };
    auto brow = [&](int k, int r) { return r < nX ? p.Xb[r] : p.Xfb[r - nX]; };
    const int RX = nX + nXf;                                                     // row stride per x stage
    // ---- condensation: free response, G (NP x NM; row (k-1) po + a, column j m + b; zero for j >= k)
    vec xf((size_t)(N + 1) * n), yf(NP), G((size_t)NP * NM, 0.0);
    std::copy(p.x0, p.x0 + n, xf.begin());
    for (int k = 0; k < N; ++k) {
        matvec(p.A[k], n, n, &xf[(size_t)k * n], &xf[(size_t)(k + 1) * n]);
        for (int i = 0; i < n; ++i) xf[(size_t)(k + 1) * n + i] += p.d[k][i];
    }
    for (int k = 1; k <= N; ++k) matvec(cb.Co.data(), po, n, &xf[(size_t)k * n], &yf[(size_t)(k - 1) * po]);
    {
        vec Psi((size_t)NP * n, 0.0), t(n);                  // rows C_o Phi(k, j+1) of stage k at (k-1) po
        for (int j = N - 1; j >= 0; --j) {
            for (int i = (j + 1) * po; i < NP; ++i) {         // stages k >= j+2: times A_{j+1}
                std::fill(t.begin(), t.end(), 0.0);
                matTvec_add(p.A[j + 1], n, n, &Psi[(size_t)i * n], t.data());
                std::copy(t.begin(), t.end(), &Psi[(size_t)i * n]);
            }
            std::copy(cb.Co.begin(), cb.Co.end(), &Psi[(size_t)j * po * n]);
            for (int i = j * po; i < NP; ++i) {
                const double *ps = &Psi[(size_t)i * n];
                double *g = &G[(size_t)i * NM + (size_t)j * m];
                for (int c = 0; c < n; ++c) { const double pc = ps[c]; const double *b = p.B[j] + (size_t)c * m; for (int a = 0; a < m; ++a) g[a] += pc * b[a]; }
            }
        }
    }
    // linear cost term in output space
    vec lin(NP, 0.0);
    {
        vec t(nz), g0(n);
        for (int k = 1; k <= N; ++k) {
            std::fill(g0.begin(), g0.end(), 0.0);
            if (p.z) { matvec(p.Qz, nz, nz, p.z + (size_t)k * nz, t.data()); for (int i = 0; i < n; ++i) for (int a = 0; a < nz; ++a) g0[i] -= 2.0 * p.H[(size_t)a * n + i] * t[a]; }
            if (k == N && p.Qzf && p.zf) { matvec(p.Qzf, nz, nz, p.zf, t.data()); for (int i = 0; i < n; ++i) for (int a = 0; a < nz; ++a) g0[i] -= 2.0 * p.H[(size_t)a * n + i] * t[a]; }
            matvec(cb.Co.data(), po, n, g0.data(), &lin[(size_t)(k - 1) * po]);
        }
    }
    auto jmax_of_row = [&](int i) { return i / po + 1; };       // row i = (k-1) po + a reaches the stages j < k
    auto G_times = [&](const vec &uu, vec &yy) {                 // yy = G uu
        for (int i = 0; i < NP; ++i) {
            const double *g = &G[(size_t)i * NM];
            const int ce = jmax_of_row(i) * m;
            double sacc = 0.0;
            for (int c = 0; c < ce; ++c) sacc += g[c] * uu[c];
            yy[i] = sacc;
        }
    };
    auto GT_add = [&](const vec &yy, vec &uu) {                  // uu += G^T yy
        for (int i = 0; i < NP; ++i) {
            const double *g = &G[(size_t)i * NM];
            const int ce = jmax_of_row(i) * m;
            const double yi = yy[i];
            for (int c = 0; c < ce; ++c) uu[c] += g[c] * yi;
        }
    };
    // ---- rows: x rows (k-1) RX + r (k = 1..N), then u rows N RX + k nU + r
    const int NR = N * RX + N * nU;
    vec tt(NR, 0.0), lam(NR, 0.0), rg(NR, 0.0), D(NR, 0.0), rho(NR, 0.0), rc(NR, 0.0), dtv(NR, 0.0), dl(NR, 0.0), ev(NR, 1.0), gval(NR, 0.0), aval(NR, 0.0);
    std::vector<char> live(NR, 0);
    for (int k = 1; k <= N; ++k) for (int r = 0; r < nrx(k); ++r) live[(size_t)(k - 1) * RX + r] = 1;
    for (int e = N * RX; e < NR; ++e) live[e] = 1;
    auto row_apply = [&](const vec &yy, const vec &uu, vec &out, bool with_h) {
        for (int k = 1; k <= N; ++k)
            for (int r = 0; r < nrx(k); ++r) {
                const double *T = Trow(k, r);
                double v = with_h ? -brow(k, r) : 0.0;
                for (int a = 0; a < po; ++a) v += T[a] * yy[(size_t)(k - 1) * po + a];
                out[(size_t)(k - 1) * RX + r] = v;
            }
        for (int k = 0; k < N; ++k)
            for (int r = 0; r < nU; ++r) {
                double v = with_h ? -p.Ub[r] : 0.0;
                for (int a = 0; a < m; ++a) v += p.UA[(size_t)r * m + a] * uu[(size_t)k * m + a];
                out[(size_t)N * RX + (size_t)k * nU + r] = v;
            }
    };
    // total gradient wrt u of cost + rows^T w
    vec gy(NP), gtot(NM);
    auto grad = [&](const vec &uu, const vec &yy, const vec &w, vec &out) {
        for (int k = 1; k <= N; ++k) {
            const vec &S = (k == N) ? cb.ScN : cb.Sc;
            for (int a = 0; a < po; ++a) {
                double v = lin[(size_t)(k - 1) * po + a];
                for (int b = 0; b < po; ++b) v += S[(size_t)a * po + b] * yy[(size_t)(k - 1) * po + b];
                for (int r = 0; r < nrx(k); ++r) v += Trow(k, r)[a] * w[(size_t)(k - 1) * RX + r];
                gy[(size_t)(k - 1) * po + a] = v;
            }
        }
        for (int k = 0; k < N; ++k)
            for (int a = 0; a < m; ++a) {
                double v = 0.0;
                for (int b = 0; b < m; ++b) v += p.Ru[(size_t)a * m + b] * (uu[(size_t)k * m + b] - (p.ud ? p.ud[(size_t)k * m + b] : 0.0));
                for (int r = 0; r < nU; ++r) v += p.UA[(size_t)r * m + a] * w[(size_t)N * RX + (size_t)k * nU + r];
                out[(size_t)k * m + a] = v;
            }
        GT_add(gy, out);
    };
    // ---- Newton systems  (blkdiag(Dblk) + G^T blkdiag(S) G) du = rhs  in output space
    vec Ld((size_t)N * m * m), Ls((size_t)N * po * po), Gd((size_t)NP * NM), K((size_t)NP * NP), ks(NP), tv(NM), yv(NP), wv(NP), t2(NM);
    auto factor = [&]() -> bool {
        for (int k = 0; k < N; ++k) {
            double *L = &Ld[(size_t)k * m * m];
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b) {
                    double v = p.Ru[(size_t)a * m + b];
                    for (int r = 0; r < nU; ++r) v += p.UA[(size_t)r * m + a] * D[(size_t)N * RX + (size_t)k * nU + r] * p.UA[(size_t)r * m + b];
                    L[(size_t)a * m + b] = v;
                }
            if (!cholesky(L, m)) return false;
        }
        for (int k = 1; k <= N; ++k) {
            double *L = &Ls[(size_t)(k - 1) * po * po];
            const vec &S = (k == N) ? cb.ScN : cb.Sc;
            for (int a = 0; a < po; ++a)
                for (int b = 0; b < po; ++b) {
                    double v = S[(size_t)a * po + b];
                    for (int r = 0; r < nrx(k); ++r) v += Trow(k, r)[a] * D[(size_t)(k - 1) * RX + r] * Trow(k, r)[b];
                    L[(size_t)a * po + b] = v;
                }
            chol_psd(L, po);
        }
        // Gd = blkdiag(Ls^T) G blkdiag(Ld^-T): row mixing per output stage, forward substitution per input block
        vec tmp(m);
        for (int k = 1; k <= N; ++k) {
            const double *L = &Ls[(size_t)(k - 1) * po * po];
            const int ce = k * m;
            for (int a = 0; a < po; ++a) {
                double *out = &Gd[((size_t)(k - 1) * po + a) * NM];
                std::fill(out, out + NM, 0.0);
                for (int a2 = a; a2 < po; ++a2) {            // (Ls^T)[a][a2] = Ls[a2][a]
                    const double l = L[(size_t)a2 * po + a];
                    if (l == 0.0) continue;
                    const double *g = &G[((size_t)(k - 1) * po + a2) * NM];
                    for (int c = 0; c < ce; ++c) out[c] += l * g[c];
                }
                for (int j = 0; j < k; ++j) {                // row block j: solve Ld_j y = block^T
                    const double *Lj = &Ld[(size_t)j * m * m];
                    double *blk = out + (size_t)j * m;
                    for (int i = 0; i < m; ++i) {
                        double v = blk[i];
                        for (int q = 0; q < i; ++q) v -= Lj[(size_t)i * m + q] * tmp[q];
                        tmp[i] = v / Lj[(size_t)i * m + i];
                    }
                    for (int i = 0; i < m; ++i) blk[i] = tmp[i];
                }
            }
        }
        for (int i = 0; i < NP; ++i) {
            const double *gi = &Gd[(size_t)i * NM];
            for (int j = 0; j <= i; ++j) {
                const double *gj = &Gd[(size_t)j * NM];
                const int ce = jmax_of_row(j) * m;           // the shorter of the two rows (j <= i)
                double v = (i == j) ? 1.0 : 0.0;
                for (int c = 0; c < ce; ++c) v += gi[c] * gj[c];
                K[(size_t)i * NP + j] = v;
            }
        }
        for (int i = 0; i < NP; ++i) ks[i] = 1.0 / std::sqrt(K[(size_t)i * NP + i]);
        for (int i = 0; i < NP; ++i) for (int j = 0; j <= i; ++j) K[(size_t)i * NP + j] *= ks[i] * ks[j];
        return cholesky(K.data(), NP);
    };
    auto Dinv = [&](vec &v) {
        for (int k = 0; k < N; ++k) {
            const double *L = &Ld[(size_t)k * m * m];
            double *b = &v[(size_t)k * m];
            for (int i = 0; i < m; ++i) { double s1 = b[i]; for (int q = 0; q < i; ++q) s1 -= L[(size_t)i * m + q] * b[q]; b[i] = s1 / L[(size_t)i * m + i]; }
            for (int i = m - 1; i >= 0; --i) { double s1 = b[i]; for (int q = i + 1; q < m; ++q) s1 -= L[(size_t)q * m + i] * b[q]; b[i] = s1 / L[(size_t)i * m + i]; }
        }
    };
    auto Ls_apply = [&](vec &v, bool transpose) {            // per stage: v_k <- Ls_k v_k or Ls_k^T v_k
        vec o(po);
        for (int k = 0; k < N; ++k) {
            const double *L = &Ls[(size_t)k * po * po];
            double *b = &v[(size_t)k * po];
            for (int a = 0; a < po; ++a) {
                double s1 = 0.0;
                if (transpose) { for (int c = a; c < po; ++c) s1 += L[(size_t)c * po + a] * b[c]; }
                else { for (int c = 0; c <= a; ++c) s1 += L[(size_t)a * po + c] * b[c]; }
                o[a] = s1;
            }
            for (int a = 0; a < po; ++a) b[a] = o[a];
        }
    };
    // S*_k positive definite (both constant output blocks): every Ls_k is invertible and dy = G du follows from the solved
    // system itself -- with w = ks v: (I + Ls^T Ky Ls) w = Ls^T G t, Ky = G D^-1 G^T, so G du = G t - Ky Ls w = Ls^-T w
    // (condensed_ipm.py: newton_solve; kernels: ql::newton_back, qpc::newton_solve)
    auto spd_small = [&](const vec &S) {
        vec L(S);
        double dmax = 0.0;
        for (int a = 0; a < po; ++a) dmax = std::max(dmax, std::fabs(S[(size_t)a * po + a]));
        for (int i = 0; i < po; ++i)
            for (int j = 0; j <= i; ++j) {
                double v = L[(size_t)i * po + j];
                for (int q = 0; q < j; ++q) v -= L[(size_t)i * po + q] * L[(size_t)j * po + q];
                if (i == j) { if (!(v > 1e-8 * dmax)) return false; L[(size_t)i * po + i] = std::sqrt(v); }
                else L[(size_t)i * po + j] = v / L[(size_t)j * po + j];
            }
        return true;
    };
    const bool ls_pd = spd_small(cb.Sc) && spd_small(cb.ScN);
    // du = M^-1 rhs:  t = D^-1 rhs;  K v = Ls^T G t;  du = t - D^-1 G^T Ls v;  wv = ks v (for dy_from_w)
    auto newton_solve = [&](const vec &rhs, vec &duo) {
        tv = rhs;
        Dinv(tv);
        G_times(tv, yv);
        Ls_apply(yv, true);
        for (int i = 0; i < NP; ++i) yv[i] *= ks[i];
        for (int i = 0; i < NP; ++i) { double s1 = yv[i]; for (int q = 0; q < i; ++q) s1 -= K[(size_t)i * NP + q] * yv[q]; yv[i] = s1 / K[(size_t)i * NP + i]; }
        for (int i = NP - 1; i >= 0; --i) { double s1 = yv[i]; for (int q = i + 1; q < NP; ++q) s1 -= K[(size_t)q * NP + i] * yv[q]; yv[i] = s1 / K[(size_t)i * NP + i]; }
        for (int i = 0; i < NP; ++i) yv[i] *= ks[i];
        wv = yv;
        Ls_apply(yv, false);
        std::fill(t2.begin(), t2.end(), 0.0);
        GT_add(yv, t2);
        Dinv(t2);
        for (int e = 0; e < NM; ++e) duo[e] = tv[e] - t2[e];
    };
    auto direction_y = [&](const vec &duv, vec &dyv) {          // dy = G du
        if (!ls_pd) { G_times(duv, dyv); return; }
        for (int k = 0; k < N; ++k) {                           // Ls_k^T dy_k = w_k (back substitution, Ls lower)
            const double *L = &Ls[(size_t)k * po * po];
            for (int a = po - 1; a >= 0; --a) {
                double s1 = wv[(size_t)k * po + a];
                for (int c = a + 1; c < po; ++c) s1 -= L[(size_t)c * po + a] * dyv[(size_t)k * po + c];
                dyv[(size_t)k * po + a] = s1 / L[(size_t)a * po + a];
            }
        }
    };
    auto finish = [&](int it, int status, double mu) {
        x.assign((size_t)(N + 1) * n, 0.0);
        std::copy(p.x0, p.x0 + n, x.begin());
        vec t(n);
        for (int k = 0; k < N; ++k) {
            matvec(p.A[k], n, n, &x[(size_t)k * n], t.data());
            for (int i = 0; i < n; ++i) {
                double sB = 0.0;
                for (int a = 0; a < m; ++a) sB += p.B[k][(size_t)i * m + a] * u[(size_t)k * m + a];
                x[(size_t)(k + 1) * n + i] = t[i] + sB + p.d[k][i];
            }
        }
        s.assign(N + 1, 0.0);
        Problem q = p;
        q.tr = false;
        if (J_out) *J_out = objective(q, x, u, s);
        bool inside = true;
        if (p.tr) {
            double md = 0.0;
            for (int k = 1; k <= N; ++k) for (int i = 0; i < n; ++i) md = std::max(md, std::fabs(p.xs[i] * (x[(size_t)k * n + i] - p.xk[(size_t)k * n + i])));
            inside = md <= p.delta;
        }
        if (inside_out) *inside_out = inside;
        return Info{it, status, mu};
    };
    u.assign(NM, 0.0);
    vec y = yf, du(NM), dy(NP), rhs(NM);
    auto neg_grad = [&](const vec &w) { grad(u, y, w, gtot); for (int e = 0; e < NM; ++e) rhs[e] = -gtot[e]; };
    if (ng == 0) {
        if (!factor()) return finish(0, 2, 0.0);
        neg_grad(rho);
        newton_solve(rhs, du);
        for (int e = 0; e < NM; ++e) u[e] += du[e];
        return finish(0, 0, 0.0);
    }
    const bool warm = ws != nullptr && ws->valid && (int)ws->u.size() == NM && (int)ws->lam.size() == NR;
    if (warm) {
        // the previous QP's point (condensed_ipm.solve, warm): slacks from this QP's rows, multipliers kept, both away from zero
        u = ws->u;
        G_times(u, y);
        for (int i = 0; i < NP; ++i) y[i] += yf[i];
        row_apply(y, u, gval, true);
        for (int e = 0; e < NR; ++e) if (live[e]) { tt[e] = std::max(-gval[e], WARM_FLOOR); lam[e] = std::max(ws->lam[e], WARM_FLOOR); }
    } else {
    // starting point: unit weights, gradient shifts = row values
    row_apply(y, u, gval, true);
    for (int e = 0; e < NR; ++e) { D[e] = live[e] ? 1.0 : 0.0; rho[e] = live[e] ? gval[e] : 0.0; }
    if (!factor()) return finish(0, 2, 0.0);
    neg_grad(rho);
    newton_solve(rhs, du);
    for (int e = 0; e < NM; ++e) u[e] += du[e];
    G_times(u, y);
    for (int i = 0; i < NP; ++i) y[i] += yf[i];
    row_apply(y, u, gval, true);
    double zmin = INF, zmax = -INF;
    for (int e = 0; e < NR; ++e) if (live[e]) { zmin = std::min(zmin, gval[e]); zmax = std::max(zmax, gval[e]); }
    const double sh_t = zmax >= 0.0 ? 1.0 + zmax : 0.0, sh_l = zmin <= 0.0 ? 1.0 - zmin : 0.0;
    for (int e = 0; e < NR; ++e) if (live[e]) { tt[e] = -gval[e] + sh_t; lam[e] = gval[e] + sh_l; }
    }
    vec g1(n), zero(n, 0.0);
    grad_x(p, 1, zero.data(), g1.data());
    double scale_d = std::max(1.0, p.omega), scale_p = std::max(1.0, std::fabs(p.delta));
    for (double g : g1) scale_d = std::max(scale_d, std::fabs(g));
    for (int r = 0; r < nU; ++r) scale_p = std::max(scale_p, std::fabs(p.Ub[r]));
    const double dreg = reg / scale_d;
    int status = 1, it = 0;
    double mu = 0.0;
    auto maxstep = [&]() {
        double a = INF;
        for (int e = 0; e < NR; ++e) if (live[e]) {
            if (dtv[e] < 0.0) a = std::min(a, -tt[e] / dtv[e]);
            if (dl[e] < 0.0) a = std::min(a, -lam[e] / dl[e]);
        }
        return a;
    };
    for (it = 0; it < max_iter; ++it) {
        row_apply(y, u, gval, true);
        double musum = 0.0, rp = 0.0;
        for (int e = 0; e < NR; ++e) if (live[e]) {
            rg[e] = gval[e] + tt[e];
            musum += lam[e] * tt[e];
            ev[e] = tt[e] + dreg * lam[e];
            D[e] = lam[e] / ev[e];
            rho[e] = lam[e] + (lam[e] * rg[e] - lam[e] * tt[e]) / ev[e];
            rp = std::max(rp, std::fabs(rg[e]));
        }
        mu = musum / ng;
        grad(u, y, lam, gtot);
        double rd = 0.0;
        for (int e = 0; e < NM; ++e) rd = std::max(rd, std::fabs(gtot[e]));
        if (rd <= std::max(tol, 1e-9) * scale_d && rp <= std::max(tol, 1e-9) * scale_p && mu <= tol) { status = 0; break; }
        if (!factor()) { status = 2; break; }
        neg_grad(rho);
        newton_solve(rhs, du);
        direction_y(du, dy);
        row_apply(dy, du, aval, false);
        for (int e = 0; e < NR; ++e) if (live[e]) {
            dl[e] = (-lam[e] * tt[e] + lam[e] * (rg[e] + aval[e])) / ev[e];
            dtv[e] = -rg[e] - aval[e] + dreg * dl[e];
        }
        const double a_aff = std::min(1.0, maxstep());
        double ma = 0.0;
        for (int e = 0; e < NR; ++e) if (live[e]) ma += (lam[e] + a_aff * dl[e]) * (tt[e] + a_aff * dtv[e]);
        const double mu_aff = ma / ng;
        const double sigma = mu > 0.0 ? std::pow(mu_aff / mu, 3.0) : 0.0;
        for (int e = 0; e < NR; ++e) if (live[e]) {
            rc[e] = lam[e] * tt[e] + dtv[e] * dl[e] - sigma * mu;
            rho[e] = lam[e] + (lam[e] * rg[e] - rc[e]) / ev[e];
        }
        neg_grad(rho);
        newton_solve(rhs, du);
        direction_y(du, dy);
        row_apply(dy, du, aval, false);
        for (int e = 0; e < NR; ++e) if (live[e]) {
            dl[e] = (-rc[e] + lam[e] * (rg[e] + aval[e])) / ev[e];
            dtv[e] = -rg[e] - aval[e] + dreg * dl[e];
        }
        const double a = std::min(1.0, 0.99 * maxstep());
        for (int e = 0; e < NM; ++e) u[e] += a * du[e];
        for (int i = 0; i < NP; ++i) y[i] += a * dy[i];
        for (int e = 0; e < NR; ++e) if (live[e]) { tt[e] += a * dtv[e]; lam[e] += a * dl[e]; }
        if (!std::isfinite(mu)) { status = 2; break; }
    }
    if (warm && status != 0) {                       // a warm start that stalls: again from Mehrotra's point
        ws->valid = false;
        return cond_solve(p, cb, x, u, s, J_out, inside_out, tol, max_iter, reg, ws);
    }
    if (ws != nullptr && status == 0) { ws->valid = true; ws->u = u; ws->lam = lam; }
    return finish(it, status, mu);
}

// The QP with the device kernel's control flow: without the trust-region rows first; the full QP only if that
// minimiser leaves the trust region (dropping satisfied constraints cannot change an optimum).
// algo 0: stage-wise Riccati interior point throughout.  algo 1: what the device kernel does -- the condensed interior point
// for the QP without its trust-region rows (accepted when it converges inside the trust region), the Riccati interior point
// of the full QP otherwise.
Info qp_solve(Problem &p, vec &x, vec &u, vec &s, double *J, int algo = 0, const CondBasis *cb = nullptr, Warm *ws = nullptr) {
    auto s0 = [&]() {
        double v = 0.0;
        for (int i = 0; i < p.n; ++i) v = std::max(v, std::fabs(p.xs[i] * (p.x0[i] - p.xk[i])));
        return std::max(0.0, v - p.delta);
    };
    if (algo == 1 && cb && cb->ok) {
        bool inside = true;
        Info a = cond_solve(p, *cb, x, u, s, J, &inside, 1e-12, 60, 1e-8, ws);
        if (a.status == 0 && inside) {
            if (p.tr) { s[0] = s0(); if (J) *J += p.omega * s[0]; }
            return a;
        }
        return ipm_solve(p, x, u, s, J);
    }
    if (p.tr) {
        Problem q = p;
        q.tr = false;
        Info a = ipm_solve(q, x, u, s, J);
        if (a.status == 0) {
            double md = 0.0;
            for (int k = 1; k <= p.N; ++k)
                for (int i = 0; i < p.n; ++i) md = std::max(md, std::fabs(p.xs[i] * (x[(size_t)k * p.n + i] - p.xk[(size_t)k * p.n + i])));
            if (md <= p.delta) {
                s.assign(p.N + 1, 0.0);
                s[0] = s0();
                if (J) *J += p.omega * s[0];
                return a;
            }
        }
    }
    return ipm_solve(p, x, u, s, J);
}

struct GustoPar { double delta0, omega0, rho, beta_fail, gamma_fail, epsilon, omega_max, convg_thresh; int max_iters; };

// one rollout: oracle/gusto.py _loop with the nearest-point TPWL model.  Returns the number of SCP iterations.
int gusto_one(const Model &M, Problem base, const GustoPar &par, double dt, const double *fs, const double *x0, const double *u_init,
              const double *x_init, double *xopt, double *uopt, double *trace, int max_trace, int algo = 0, const CondBasis *cb = nullptr) {
    const int N = base.N, n = base.n, m = base.m;
    vec xk(x_init, x_init + (size_t)(N + 1) * n), uk(u_init, u_init + (size_t)N * m);
    std::vector<int> idx(N), idx2(N);
    auto linearise = [&](const vec &xx, std::vector<int> &id) { for (int k = 0; k < N; ++k) id[k] = nearest(M, &xx[(size_t)k * n]); };
    linearise(xk, idx);
    double delta = par.delta0, omega = par.omega0, J_prev = INF, d_prev = INF, o_prev = INF;
    bool converged = false;
    int itr = 0;
    vec x, u, s;
    Warm warm_state;                                 // every QP after the first one the condensed path finished starts from that one's iterate
    while (itr <= par.max_iters && !converged && omega <= par.omega_max) {
        Problem p = base;
        p.x0 = x0; p.xk = xk.data(); p.delta = delta; p.omega = omega;
        p.A.resize(N); p.B.resize(N); p.d.resize(N);
        for (int k = 0; k < N; ++k) { p.A[k] = M.Ad + (size_t)idx[k] * n * n; p.B[k] = M.Bd + (size_t)idx[k] * n * m; p.d[k] = M.dd + (size_t)idx[k] * n; }
        double J = 0.0;
        const Info inf = qp_solve(p, x, u, s, &J, algo, cb, &warm_state);
        if (inf.status != 0) break;
        double md = 0.0;
        for (int k = 0; k <= N; ++k) for (int i = 0; i < n; ++i) md = std::max(md, std::fabs(p.xs[i] * (x[(size_t)k * n + i] - xk[(size_t)k * n + i])));
        const bool tr_ok = !(md - delta > par.epsilon);
        bool new_solution = false;
        double rho_k = -1.0;
        const double d_cur = delta, o_cur = omega;
        if (tr_ok) {
            linearise(x, idx2);
            double err = 0.0, app = 0.0;
            for (int k = 0; k < N; ++k) {
                const double *Ak = M.Ac + (size_t)idx[k] * n * n, *Bk = M.Bc + (size_t)idx[k] * n * m, *dk = M.dc + (size_t)idx[k] * n;
                const double *An = M.Ac + (size_t)idx2[k] * n * n, *Bn = M.Bc + (size_t)idx2[k] * n * m, *dn = M.dc + (size_t)idx2[k] * n;
                double e2 = 0.0, a2 = 0.0;
                for (int r = 0; r < n; ++r) {
                    double fk = dk[r], fl = 0.0, f = dn[r];
                    for (int c = 0; c < n; ++c) {
                        const double xo = xk[(size_t)k * n + c], xn = x[(size_t)k * n + c];
                        fk += Ak[(size_t)r * n + c] * xo; fl += Ak[(size_t)r * n + c] * (xn - xo); f += An[(size_t)r * n + c] * xn;
                    }
                    for (int c = 0; c < m; ++c) {
                        const double uo = uk[(size_t)k * m + c], un = u[(size_t)k * m + c];
                        fk += Bk[(size_t)r * m + c] * uo; fl += Bk[(size_t)r * m + c] * (un - uo); f += Bn[(size_t)r * m + c] * un;
                    }
                    const double fa = fk + fl, de = fs[r] * (f - fa), da = fs[r] * fa;
                    e2 += de * de; a2 += da * da;
                }
                err += dt * std::sqrt(e2); app += dt * std::sqrt(a2);
            }
            rho_k = err / (J + app);
            if (rho_k > par.rho && itr != 1) {
                delta = par.beta_fail * delta;
            } else {
                if (d_prev == delta && o_prev == omega && J_prev <= J) delta = par.beta_fail * delta;
                d_prev = delta; J_prev = J; o_prev = omega;
                double viol = 0.0;
                for (int k = 0; k <= N && base.nX > 0; ++k) {
                    double v2 = 0.0;
                    for (int r = 0; r < base.nX; ++r) {
                        double v = -base.Xb[r];
                        for (int j = 0; j < n; ++j) v += base.XA[(size_t)r * n + j] * x[(size_t)k * n + j];
                        v = std::max(v, 0.0); v2 += v * v;
                    }
                    viol = std::max(viol, std::sqrt(v2));
                }
                const bool X_ok = !(viol > par.epsilon);
                if (!X_ok) omega = par.gamma_fail * omega;
                double dsum = 0.0;
                for (int k = 0; k <= N; ++k) {
                    double v2 = 0.0;
                    for (int j = 0; j < n; ++j) { const double e = p.xs[j] * (x[(size_t)k * n + j] - xk[(size_t)k * n + j]); v2 += e * e; }
                    dsum += std::sqrt(v2);
                }
                converged = ((1.0 / N) * ((1.0 / n) * dsum) <= par.convg_thresh) && X_ok;
                new_solution = true;
            }
        } else {
            omega = par.gamma_fail * omega;
        }
        if (trace && itr < max_trace) { double *t = trace + (size_t)itr * 4; t[0] = J; t[1] = d_cur; t[2] = o_cur; t[3] = rho_k; }
        ++itr;
        if (new_solution) {
            xk = x; uk = u;
            if (par.max_iters >= 1) linearise(xk, idx);
        }
    }
    std::copy(xk.begin(), xk.end(), xopt);
    std::copy(uk.begin(), uk.end(), uopt);
    return itr;
}


// ------------------------------------------------------------------ iLQR (oracle/lqr.py: ILQR.solve; ilqr.py:27-300)
struct IlqrPar {
    int max_iter; double epsilon, alpha0, alpha_scaling, improv_lb, improv_ub, alpha_min; int counter_limit;
    double rho0, drho0, rho_scaling, rho_increase_fp, rho_max, rho_min;
    int include_input_var_constraint, do_linesearch, regularize, state_regularization;
};

// in-place Gauss-Jordan inverse with partial pivoting (first maximum), A (n x n) -> Ainv; false if singular
inline bool gj_inverse(vec &A, int n, vec &Ainv) {
    Ainv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) Ainv[(size_t)i * n + i] = 1.0;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(A[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i) { const double v = std::fabs(A[(size_t)i * n + k]); if (v > best) { best = v; p = i; } }
        if (!(best > 0.0)) return false;
        if (p != k) for (int j = 0; j < n; ++j) { std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]); std::swap(Ainv[(size_t)k * n + j], Ainv[(size_t)p * n + j]); }
        const double d = A[(size_t)k * n + k];
        for (int j = 0; j < n; ++j) { A[(size_t)k * n + j] /= d; Ainv[(size_t)k * n + j] /= d; }
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const double f = A[(size_t)i * n + k];
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) { A[(size_t)i * n + j] -= f * A[(size_t)k * n + j]; Ainv[(size_t)i * n + j] -= f * Ainv[(size_t)k * n + j]; }
        }
    }
    return true;
}

// SSM polynomial model (oracle/ssm.py): graded monomial basis, analytic derivatives, discretisation of ssm.py:279-301
struct SsmModel {
    int n, m, no, nr, ns, order_r, order_s;
    const double *R, *B, *W, *z_ref;          // r_coeff (n x nr), B (n x m), w_coeff (no x ns), z_ref (no)
    std::vector<int> Er, Es, par_r, var_r, par_s, var_s;      // exponents, parent monomial / multiplied variable
    std::vector<int> nzk, nzq, nze, nzoff;    // per variable j: monomials with e_kj > 0, the index of e_k - 1_j (-2: constant), e_kj
    int mode; double dt;                      // 1 fe, 2 be, 3 bil (SSM_FE / SSM_BE / SSM_BIL)
};
inline void exponents(int dim, int order, std::vector<int> &out) {
    std::vector<int> cur(dim, 0);
    struct Rec { std::vector<int> &out, &cur; int dim;
        void go(int pos, int left) {
            if (pos == dim - 1) { cur[pos] = left; out.insert(out.end(), cur.begin(), cur.end()); return; }
            for (int e = left; e >= 0; --e) { cur[pos] = e; go(pos + 1, left - e); }
        } } rec{out, cur, dim};
    for (int deg = 1; deg <= order; ++deg) rec.go(0, deg);
}
inline int find_monomial(const std::vector<int> &E, int dim, const std::vector<int> &e) {
    const int nm = (int)(E.size() / dim);
    for (int j = 0; j < nm; ++j) if (std::equal(e.begin(), e.end(), E.begin() + (size_t)j * dim)) return j;
    return -1;
}
inline void basis_tables(const std::vector<int> &E, int dim, std::vector<int> &par, std::vector<int> &var) {
    const int nm = (int)(E.size() / dim);
    par.assign(nm, -1); var.assign(nm, 0);
    for (int j = 0; j < nm; ++j) {
        std::vector<int> e(E.begin() + (size_t)j * dim, E.begin() + (size_t)(j + 1) * dim);
        int deg = 0, last = 0;
        for (int i = 0; i < dim; ++i) { deg += e[i]; if (e[i] > 0) last = i; }
        var[j] = last;
        if (deg > 1) { e[last] -= 1; par[j] = find_monomial(E, dim, e); }
    }
}
inline void ssm_prepare(SsmModel &S) {
    exponents(S.n, S.order_r, S.Er); exponents(S.no, S.order_s, S.Es);
    S.nr = (int)(S.Er.size() / S.n); S.ns = (int)(S.Es.size() / S.no);
    basis_tables(S.Er, S.n, S.par_r, S.var_r); basis_tables(S.Es, S.no, S.par_s, S.var_s);
    S.nzoff.assign(S.n + 1, 0);
    for (int j = 0; j < S.n; ++j) {
        for (int k = 0; k < S.nr; ++k) {
            const int ekj = S.Er[(size_t)k * S.n + j];
            if (ekj == 0) continue;
            std::vector<int> e(S.Er.begin() + (size_t)k * S.n, S.Er.begin() + (size_t)(k + 1) * S.n);
            int deg = 0;
            for (int v : e) deg += v;
            e[j] -= 1;
            S.nzk.push_back(k); S.nze.push_back(ekj); S.nzq.push_back(deg == 1 ? -2 : find_monomial(S.Er, S.n, e));
        }
        S.nzoff[j + 1] = (int)S.nzk.size();
    }
}
inline void ssm_phi(const std::vector<int> &par, const std::vector<int> &var, const double *x, double *phi) {
    const int nm = (int)par.size();
    for (int j = 0; j < nm; ++j) phi[j] = (par[j] < 0 ? 1.0 : phi[par[j]]) * x[var[j]];       // parents come first (graded order)
}
// (A_d, B_d, d_d) at (x, u): continuous Jacobians of f = R phi(x) + B u, d = f - A x - B u, then the discretisation
inline void ssm_lin(const SsmModel &S, const double *x, const double *u, double *A, double *Bm, double *d, vec &phi, vec &tmp) {
    const int n = S.n, m = S.m, nr = S.nr;
    phi.resize(nr);
    ssm_phi(S.par_r, S.var_r, x, phi.data());
    vec Ac((size_t)n * n, 0.0), f(n), dc(n);
    for (int j = 0; j < n; ++j)
        for (int t = S.nzoff[j]; t < S.nzoff[j + 1]; ++t) {
            const double dv = (double)S.nze[t] * (S.nzq[t] == -2 ? 1.0 : phi[S.nzq[t]]);
            const int k = S.nzk[t];
            for (int i = 0; i < n; ++i) Ac[(size_t)i * n + j] += S.R[(size_t)i * nr + k] * dv;
        }
    for (int i = 0; i < n; ++i) {
        double s = 0.0, bu = 0.0, ax = 0.0;
        for (int k = 0; k < nr; ++k) s += S.R[(size_t)i * nr + k] * phi[k];
        for (int k = 0; k < m; ++k) bu += S.B[(size_t)i * m + k] * u[k];
        for (int k = 0; k < n; ++k) ax += Ac[(size_t)i * n + k] * x[k];
        f[i] = s + bu; dc[i] = f[i] - ax - bu;
    }
    if (S.mode == 1) {
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = (i == j ? 1.0 : 0.0) + S.dt * Ac[(size_t)i * n + j];
        for (int e = 0; e < n * m; ++e) Bm[e] = S.dt * S.B[e];
        for (int i = 0; i < n; ++i) d[i] = S.dt * dc[i];
        return;
    }
    const double h = S.mode == 2 ? S.dt : 0.5 * S.dt;
    vec M1((size_t)n * n), M2, M3(Ac), M4, Ad((size_t)n * n), sep((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) M1[(size_t)i * n + j] = (i == j ? 1.0 : 0.0) - h * Ac[(size_t)i * n + j];
    gj_inverse(M1, n, M2);
    gj_inverse(M3, n, M4);
    if (S.mode == 3) {
        vec T((size_t)n * n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) T[(size_t)i * n + j] = (i == j ? 1.0 : 0.0) + h * Ac[(size_t)i * n + j];
        matmul(T.data(), M2.data(), n, n, n, Ad.data());
    } else {
        Ad = M2;
    }
    vec AmI(Ad);
    for (int i = 0; i < n; ++i) AmI[(size_t)i * n + i] -= 1.0;
    matmul(M4.data(), AmI.data(), n, n, n, sep.data());
    std::copy(Ad.begin(), Ad.end(), A);
    matmul(sep.data(), S.B, n, n, m, Bm);
    matvec(sep.data(), n, n, dc.data(), d);
}
inline void ssm_out(const SsmModel &S, const double *x, double *z, vec &phi) {           // z = W phi_s(x) + z_ref
    phi.resize(S.ns);
    ssm_phi(S.par_s, S.var_s, x, phi.data());
    for (int i = 0; i < S.no; ++i) {
        double s = 0.0;
        for (int k = 0; k < S.ns; ++k) s += S.W[(size_t)i * S.ns + k] * phi[k];
        z[i] = s + S.z_ref[i];
    }
}

// One problem.  lin(x, u, A, B, d): discrete Jacobians at (x, u); out(x, z): the performance output (absolute).  H: the constant
// output Jacobian of the cost derivatives (nz x n).  Returns the iteration count (-1: gave up, Q~_uu never positive definite).
template <class Lin, class Out>
int ilqr_one(int N, int n, int m, int nz, const double *H, const double *Q, const double *R, const double *Qf, const IlqrPar &p,
             const double *x0, const double *zt, const double *u_warm, const double *u_last, Lin lin, Out out, double *x, double *u,
             double *K, double *cost_out) {
    const size_t nn = (size_t)n * n, nm = (size_t)n * m;
    vec A(N * nn), B(N * nm), dd((size_t)N * n), A2(N * nn), B2(N * nm), d2((size_t)N * n);
    vec x2((size_t)(N + 1) * n), u2((size_t)N * m), kff((size_t)N * m), Qu((size_t)N * m), Quu((size_t)N * m * m), z(nz), dz(nz);
    vec HtQH(nn), HtQfH(nn), QH((size_t)nz * n), QfH((size_t)nz * n), ulast(m, 0.0);
    if (u_last) ulast.assign(u_last, u_last + m);
    matmul(Q, H, nz, nz, n, QH.data()); matmul(Qf, H, nz, nz, n, QfH.data());
    matTmul(H, QH.data(), nz, n, n, HtQH.data()); matTmul(H, QfH.data(), nz, n, n, HtQfH.data());
    double rho = p.rho0, drho = p.drho0;
    auto reg = [&](bool increase) {
        if (increase) { drho = std::max(drho * p.rho_scaling, p.rho_scaling); rho = std::max(rho * drho, p.rho_min); if (rho > p.rho_max) rho = p.rho_max; }
        else { const double dh = std::min(drho / p.rho_scaling, 1.0 / p.rho_scaling); rho = rho * dh; if (rho <= p.rho_min) rho = p.rho_min; }
    };
    auto quad = [&](const double *M, const double *v, int k) { double s = 0.0; for (int i = 0; i < k; ++i) { double r = 0.0; for (int j = 0; j < k; ++j) r += M[(size_t)i * k + j] * v[j]; s += v[i] * r; } return s; };
    // forward pass from (xp, up) with gains (Kg, kg, alpha) into (xo, uo, Ao, Bo, do_); returns the cost (ilqr.py:117-162)
    auto forward = [&](const double *xp, const double *up, double alpha, const double *Kg, const double *kg, double *xo, double *uo,
                       double *Ao, double *Bo, double *do_) {
        double cost = 0.0;
        vec du(m);
        std::copy(x0, x0 + n, xo);
        for (int t = 0; t < N; ++t) {
            const double *xt = xo + (size_t)t * n;
            double *ut = uo + (size_t)t * m;
            for (int r = 0; r < m; ++r) {
                double v = up[(size_t)t * m + r];
                if (kg) v += alpha * kg[(size_t)t * m + r];
                if (Kg) for (int j = 0; j < n; ++j) v += Kg[((size_t)t * m + r) * n + j] * (xt[j] - xp[(size_t)t * n + j]);
                ut[r] = v;
            }
            out(xt, z.data());
            for (int a = 0; a < nz; ++a) dz[a] = z[a] - zt[(size_t)t * nz + a];
            for (int r = 0; r < m; ++r) du[r] = ut[r] - (p.include_input_var_constraint ? (t == 0 ? ulast[r] : uo[(size_t)(t - 1) * m + r]) : 0.0);
            cost += 0.5 * quad(Q, dz.data(), nz) + 0.5 * quad(R, du.data(), m);
            double *At = Ao + (size_t)t * nn, *Bt = Bo + (size_t)t * nm, *dt_ = do_ + (size_t)t * n;
            lin(xt, ut, At, Bt, dt_);
            double *xn = xo + (size_t)(t + 1) * n;
            for (int i = 0; i < n; ++i) {
                double s = dt_[i];
                for (int j = 0; j < n; ++j) s += At[(size_t)i * n + j] * xt[j];
                for (int j = 0; j < m; ++j) s += Bt[(size_t)i * m + j] * ut[j];
                xn[i] = s;
            }
        }
        out(xo + (size_t)N * n, z.data());
        for (int a = 0; a < nz; ++a) dz[a] = z[a] - zt[(size_t)N * nz + a];
        return cost + 0.5 * quad(Qf, dz.data(), nz);
    };
    // backward pass (ilqr.py:219-300); false: gave up
    auto backward = [&]() {
        vec P(nn), pv(n), PA(nn), PB(nm), Qxx(nn), Qux(nm), Quut((size_t)m * m), Quxt(nm), Qx(n), Lc((size_t)m * m), Kt(nm), kt(m), tmp(n), cu(m), g(nz);
        int restarts = 0;
        while (true) {
            out(x + (size_t)N * n, z.data());
            for (int a = 0; a < nz; ++a) dz[a] = z[a] - zt[(size_t)N * nz + a];
            matvec(Qf, nz, nz, dz.data(), g.data());
            std::fill(pv.begin(), pv.end(), 0.0); matTvec_add(H, nz, n, g.data(), pv.data());
            P = HtQfH;
            bool restart = false;
            for (int t = N - 1; t >= 0; --t) {
                const double *At = A.data() + (size_t)t * nn, *Bt = B.data() + (size_t)t * nm, *xt = x + (size_t)t * n, *ut = u + (size_t)t * m;
                out(xt, z.data());
                for (int a = 0; a < nz; ++a) dz[a] = z[a] - zt[(size_t)t * nz + a];
                matvec(Q, nz, nz, dz.data(), g.data());
                std::fill(Qx.begin(), Qx.end(), 0.0); matTvec_add(H, nz, n, g.data(), Qx.data());          // c_x
                matTvec_add(At, n, n, pv.data(), Qx.data());                                                  // + A' p
                for (int r = 0; r < m; ++r) {
                    double s = 0.0;
                    for (int q = 0; q < m; ++q) s += R[(size_t)r * m + q] * (ut[q] - (p.include_input_var_constraint ? (t == 0 ? ulast[q] : u[(size_t)(t - 1) * m + q]) : 0.0));
                    cu[r] = s;
                }
                double *Qut = Qu.data() + (size_t)t * m, *Quu_t = Quu.data() + (size_t)t * m * m;
                for (int r = 0; r < m; ++r) Qut[r] = cu[r];
                matTvec_add(Bt, n, m, pv.data(), Qut);                                                        // Q_u = c_u + B' p
                matmul(P.data(), At, n, n, n, PA.data()); matmul(P.data(), Bt, n, n, m, PB.data());
                matTmul(At, PA.data(), n, n, n, Qxx.data());
                for (size_t e = 0; e < nn; ++e) Qxx[e] += HtQH[e];
                matTmul(Bt, PB.data(), n, m, m, Quu_t);
                for (int e = 0; e < m * m; ++e) Quu_t[e] += R[e];
                matTmul(Bt, PA.data(), n, m, n, Qux.data());
                std::copy(Quu_t, Quu_t + m * m, Quut.begin()); Quxt = Qux;
                if (p.regularize && p.state_regularization) {
                    vec BB((size_t)m * m), BA(nm);
                    matTmul(Bt, Bt, n, m, m, BB.data()); matTmul(Bt, At, n, m, n, BA.data());
                    for (int e = 0; e < m * m; ++e) Quut[e] += rho * BB[e];
                    for (size_t e = 0; e < nm; ++e) Quxt[e] += rho * BA[e];
                } else if (p.regularize) {
                    for (int r = 0; r < m; ++r) Quut[(size_t)r * m + r] += rho;
                }
                Lc = Quut;
                if (!cholesky(Lc.data(), m)) {
                    if (!p.regularize || ++restarts > 100) return false;
                    reg(true); restart = true; break;
                }
                double *Kk = K + (size_t)t * nm;
                chol_solve_neg(Lc.data(), m, Quxt.data(), n, Kk);                                             // K = -Q~uu^-1 Q~ux
                chol_solve_neg(Lc.data(), m, Qut, 1, kt.data());                                              // k = -Q~uu^-1 Q_u
                std::copy(kt.begin(), kt.end(), kff.begin() + (size_t)t * m);
                // p = Q_x + K' Quu k + K' Q_u + Q_ux' k ;  P = Q_xx + K' Quu K + K' Q_ux + Q_ux' K
                vec QuuK(nm), Quuk(m);
                matmul(Quu_t, Kk, m, m, n, QuuK.data()); matvec(Quu_t, m, m, kt.data(), Quuk.data());
                for (int i = 0; i < n; ++i) {
                    double s = Qx[i];
                    for (int r = 0; r < m; ++r) s += Kk[(size_t)r * n + i] * (Quuk[r] + Qut[r]) + Qux[(size_t)r * n + i] * kt[r];
                    tmp[i] = s;
                }
                pv = tmp;
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) {
                        double s = Qxx[(size_t)i * n + j];
                        for (int r = 0; r < m; ++r) s += Kk[(size_t)r * n + i] * (QuuK[(size_t)r * n + j] + Qux[(size_t)r * n + j]) + Qux[(size_t)r * n + i] * Kk[(size_t)r * n + j];
                        P[(size_t)i * n + j] = s;
                    }
            }
            if (restart) continue;
            reg(false);
            return true;
        }
    };
    std::fill(x2.begin(), x2.end(), 0.0);
    std::copy(x0, x0 + n, x2.begin());
    if (u_warm) std::copy(u_warm, u_warm + (size_t)N * m, u2.begin()); else std::fill(u2.begin(), u2.end(), 0.0);
    double cost = forward(x2.data(), u2.data(), 1.0, nullptr, nullptr, x, u, A.data(), B.data(), dd.data());
    int failed_counter = 0, it = 0;
    bool converged = false;
    while (!converged && it <= p.max_iter) {
        if (!backward()) { it = -1; break; }
        const double prev = cost;
        double alpha = p.alpha0, nc = cost;
        bool improved = false, failed = false;
        while (!improved && !failed) {
            improved = true;
            nc = forward(x, u, alpha, K, kff.data(), x2.data(), u2.data(), A2.data(), B2.data(), d2.data());
            double dc = 0.0;
            for (int t = 0; t < N; ++t) {
                const double *kt = kff.data() + (size_t)t * m;
                double s1 = 0.0;
                for (int r = 0; r < m; ++r) s1 += kt[r] * Qu[(size_t)t * m + r];
                dc += alpha * s1 + alpha * alpha * 0.5 * quad(Quu.data() + (size_t)t * m * m, kt, m);
            }
            const double ratio = (nc - prev) / dc;
            if (p.do_linesearch && (ratio <= p.improv_lb || ratio > p.improv_ub)) {
                alpha *= p.alpha_scaling; improved = false;
                if (alpha < p.alpha_min) { reg(true); rho += p.rho_increase_fp; failed = true; }
            }
        }
        if (!failed) {
            std::copy(x2.begin(), x2.end(), x); std::copy(u2.begin(), u2.end(), u);
            A.swap(A2); B.swap(B2); dd.swap(d2);
            cost = nc;
            converged = (prev - cost) < p.epsilon && (prev - cost) >= 0.0;
            failed_counter = 0;
        } else if (++failed_counter >= p.counter_limit) {
            converged = true;
        }
        ++it;
    }
    if (cost_out) *cost_out = cost;
    return it;
}

template <typename F>
void parallel_for(int64_t count, int threads, F f) {
    threads = std::max(1, std::min<int>(threads, (int)count));
    if (threads == 1) { for (int64_t i = 0; i < count; ++i) f(i); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([=]() { for (int64_t i = t; i < count; i += threads) f(i); });
    for (auto &th : pool) th.join();
}

}  // namespace

extern "C" {

struct scpu_problem {       // mirrors slocp_problem (include/sofacontrol_hip.h)
    int N, n_x, n_u, n_z;
    const double *H, *Qz, *R, *Qzf, *x_scale;
    int nU; const double *UA, *Ub;
    int nX; const double *XA, *Xb;
    int nXf; const double *XfA, *Xfb;
    int tr_active;
};
struct scpu_model { int P, r, m; double w_q, w_v; const double *q, *v, *Ac, *Bc, *dc, *Ad, *Bd, *dd; };
struct scpu_gusto_params { double delta0, omega0, rho, beta_fail, gamma_fail, epsilon, omega_max, convg_thresh; int max_gusto_iters; };

int scpu_version(void) { return 3; }

// out (B x r) = (X (B x n_f) - ref) U (n_f x r), rows split over `threads`
int scpu_project(const double *U, int64_t n_f, int r, const double *ref, const double *X, int64_t B, double *out, int threads) {
    parallel_for(B, threads, [&](int64_t b) {
        const double *x = X + (size_t)b * n_f;
        double acc[64];
        for (int j = 0; j < r; ++j) acc[j] = 0.0;
        for (int64_t i = 0; i < n_f; ++i) {
            const double e = x[i] - ref[i];
            const double *urow = U + (size_t)i * r;
            for (int j = 0; j < r; ++j) acc[j] += e * urow[j];
        }
        for (int j = 0; j < r; ++j) out[(size_t)b * r + j] = acc[j];
    });
    return 0;
}

static Problem make_problem(const scpu_problem *pr, const vec &ones) {
    Problem p{};
    p.N = pr->N; p.n = pr->n_x; p.m = pr->n_u; p.nz = pr->n_z; p.nU = pr->nU; p.nX = pr->nX; p.nXf = pr->nXf;
    p.H = pr->H; p.Qz = pr->Qz; p.R = pr->R; p.Qzf = pr->Qzf;
    p.UA = pr->UA; p.Ub = pr->Ub; p.XA = pr->XA; p.Xb = pr->Xb; p.XfA = pr->XfA; p.Xfb = pr->Xfb;
    p.xs = pr->x_scale ? pr->x_scale : ones.data();
    p.tr = pr->tr_active != 0;
    problem_consts(p);
    return p;
}

// one QP (LOCP.update + solve): per-stage Ad (N x n x n), Bd, dd given explicitly.  status 0 = optimal.
// algo: 0 Riccati interior point (with the trust-region prescreen), 1 condensed interior point first (as the device kernel).
int scpu_locp_solve_algo(const scpu_problem *pr, const double *Ad, const double *Bd, const double *dd, const double *x0, const double *xk,
                         double delta, double omega, const double *z, const double *zf, const double *u_des, double *x, double *u,
                         double *s, double *J, int *iters, int algo) {
    vec ones(pr->n_x, 1.0);
    Problem p = make_problem(pr, ones);
    const int N = p.N, n = p.n, m = p.m;
    p.A.resize(N); p.B.resize(N); p.d.resize(N);
    for (int k = 0; k < N; ++k) { p.A[k] = Ad + (size_t)k * n * n; p.B[k] = Bd + (size_t)k * n * m; p.d[k] = dd + (size_t)k * n; }
    p.x0 = x0; p.xk = xk; p.z = z; p.zf = zf; p.ud = u_des; p.delta = delta; p.omega = omega;
    vec xv, uv, sv;
    double Jv = 0.0;
    CondBasis cb;
    if (algo == 1) cb = cond_basis(p);
    const Info inf = qp_solve(p, xv, uv, sv, &Jv, algo, &cb);
    std::copy(xv.begin(), xv.end(), x); std::copy(uv.begin(), uv.end(), u);
    if (s) std::copy(sv.begin(), sv.end(), s);
    if (J) *J = Jv;
    if (iters) *iters = inf.iters;
    return inf.status;
}
int scpu_locp_solve(const scpu_problem *pr, const double *Ad, const double *Bd, const double *dd, const double *x0, const double *xk,
                    double delta, double omega, const double *z, const double *zf, const double *u_des, double *x, double *u,
                    double *s, double *J, int *iters) {
    return scpu_locp_solve_algo(pr, Ad, Bd, dd, x0, xk, delta, omega, z, zf, u_des, x, u, s, J, iters, 0);
}

// GuSTO.solve for `batch` independent rollouts, one per thread at a time.  iters (batch): SCP iterations per rollout.
int scpu_gusto_solve_algo(const scpu_model *mo, const scpu_problem *pr, const scpu_gusto_params *gp, double dt, int64_t batch,
                          const double *x0, const double *u_init, const double *x_init, const double *z, const double *zf,
                          const double *u_des, const double *x_char, const double *f_char, double *xopt, double *uopt, int32_t *iters,
                          double *trace, int max_trace, int threads, int algo) {
    const int n = pr->n_x, m = pr->n_u, N = pr->N, nz = pr->n_z;
    vec xs(n, 1.0), fs(n, 1.0);
    if (x_char) for (int i = 0; i < n; ++i) xs[i] = 1.0 / std::fabs(x_char[i]);
    if (f_char) for (int i = 0; i < n; ++i) fs[i] = 1.0 / std::fabs(f_char[i]);
    scpu_problem p2 = *pr;
    p2.x_scale = xs.data();
    vec ones(n, 1.0);
    const Problem base = make_problem(&p2, ones);
    Model M{mo->P, mo->r, 2 * mo->r, mo->m, mo->w_q, mo->w_v, mo->q, mo->v, mo->Ac, mo->Bc, mo->dc, mo->Ad, mo->Bd, mo->dd};
    GustoPar par{gp->delta0, gp->omega0, gp->rho, gp->beta_fail, gp->gamma_fail, gp->epsilon, gp->omega_max, gp->convg_thresh, gp->max_gusto_iters};
    CondBasis cb;
    if (algo == 1) cb = cond_basis(base);
    parallel_for(batch, threads, [&](int64_t b) {
        Problem pb = base;
        pb.z = z ? z + (size_t)b * (N + 1) * nz : nullptr;
        pb.zf = zf ? zf + (size_t)b * nz : nullptr;
        pb.ud = u_des ? u_des + (size_t)b * N * m : nullptr;
        iters[b] = gusto_one(M, pb, par, dt, fs.data(), x0 + (size_t)b * n, u_init + (size_t)b * N * m, x_init + (size_t)b * (N + 1) * n,
                             xopt + (size_t)b * (N + 1) * n, uopt + (size_t)b * N * m, trace ? trace + (size_t)b * max_trace * 4 : nullptr, max_trace, algo, &cb);
    });
    return 0;
}
int scpu_gusto_solve(const scpu_model *mo, const scpu_problem *pr, const scpu_gusto_params *gp, double dt, int64_t batch,
                     const double *x0, const double *u_init, const double *x_init, const double *z, const double *zf,
                     const double *u_des, const double *x_char, const double *f_char, double *xopt, double *uopt, int32_t *iters,
                     double *trace, int max_trace, int threads) {
    return scpu_gusto_solve_algo(mo, pr, gp, dt, batch, x0, u_init, x_init, z, zf, u_des, x_char, f_char, xopt, uopt, iters, trace,
                                 max_trace, threads, 0);
}

struct scpu_ilqr_params {
    int max_iter; double epsilon, alpha0, alpha_scaling, improv_lb, improv_ub, alpha_min; int counter_limit;
    double rho0, drho0, rho_scaling, rho_increase_fp, rho_max, rho_min;
    int include_input_var_constraint, do_linesearch, regularize, state_regularization;
};
static IlqrPar ilqr_par(const scpu_ilqr_params *p) {
    return IlqrPar{p->max_iter, p->epsilon, p->alpha0, p->alpha_scaling, p->improv_lb, p->improv_ub, p->alpha_min, p->counter_limit, p->rho0,
                   p->drho0, p->rho_scaling, p->rho_increase_fp, p->rho_max, p->rho_min, p->include_input_var_constraint, p->do_linesearch,
                   p->regularize, p->state_regularization};
}
// iLQR on the prediscretised nearest-point TPWL model (oracle/lqr.py: ILQR): z = H x + z_ref; `batch` problems, one per thread
int scpu_ilqr_tpwl(const scpu_model *mo, const double *H, const double *z_ref, int n_z, const double *Q, const double *R, const double *Qf,
                   const scpu_ilqr_params *par, int N, int64_t batch, const double *x0, const double *z_target, const double *u_warm,
                   const double *u_last, double *x, double *u, double *K, double *cost, int32_t *iters, int threads) {
    Model M{mo->P, mo->r, 2 * mo->r, mo->m, mo->w_q, mo->w_v, mo->q, mo->v, mo->Ac, mo->Bc, mo->dc, mo->Ad, mo->Bd, mo->dd};
    const int n = M.n, m = M.m;
    const IlqrPar p = ilqr_par(par);
    parallel_for(batch, threads, [&](int64_t b) {
        auto lin = [&](const double *xx, const double *, double *A, double *B, double *d) {
            const size_t i = (size_t)nearest(M, xx);
            std::copy(M.Ad + i * n * n, M.Ad + (i + 1) * n * n, A); std::copy(M.Bd + i * n * m, M.Bd + (i + 1) * n * m, B);
            std::copy(M.dd + i * n, M.dd + (i + 1) * n, d);
        };
        auto out = [&](const double *xx, double *z) { matvec(H, n_z, n, xx, z); for (int a = 0; a < n_z; ++a) z[a] += z_ref[a]; };
        iters[b] = ilqr_one(N, n, m, n_z, H, Q, R, Qf, p, x0 + (size_t)b * n, z_target + (size_t)b * (N + 1) * n_z,
                            u_warm ? u_warm + (size_t)b * N * m : nullptr, u_last ? u_last + (size_t)b * m : nullptr, lin, out,
                            x + (size_t)b * (N + 1) * n, u + (size_t)b * N * m, K + (size_t)b * N * m * n, cost ? cost + b : nullptr);
    });
    return 0;
}
// iLQR on the SSM polynomial model (oracle/lqr.py: ILQRGeneric over oracle/ssm.py): mode 1 fe / 2 be / 3 bil, z = W phi_s(x) + z_ref,
// constant H for the cost Jacobians
int scpu_ilqr_ssm(int n, int m, int no, int rom_order, int ssm_order, const double *r_coeff, const double *B, const double *w_coeff,
                  const double *z_ref, const double *H, int mode, double dt, const double *Q, const double *R, const double *Qf,
                  const scpu_ilqr_params *par, int N, int64_t batch, const double *x0, const double *z_target, const double *u_warm,
                  const double *u_last, double *x, double *u, double *K, double *cost, int32_t *iters, int threads) {
    SsmModel S{};
    S.n = n; S.m = m; S.no = no; S.order_r = rom_order; S.order_s = ssm_order; S.R = r_coeff; S.B = B; S.W = w_coeff; S.z_ref = z_ref;
    S.mode = mode; S.dt = dt;
    ssm_prepare(S);
    const IlqrPar p = ilqr_par(par);
    parallel_for(batch, threads, [&](int64_t b) {
        vec phi, tmp, phis;
        auto lin = [&](const double *xx, const double *uu, double *A, double *Bm, double *d) { ssm_lin(S, xx, uu, A, Bm, d, phi, tmp); };
        auto out = [&](const double *xx, double *z) { ssm_out(S, xx, z, phis); };
        iters[b] = ilqr_one(N, n, m, no, H, Q, R, Qf, p, x0 + (size_t)b * n, z_target + (size_t)b * (N + 1) * no,
                            u_warm ? u_warm + (size_t)b * N * m : nullptr, u_last ? u_last + (size_t)b * m : nullptr, lin, out,
                            x + (size_t)b * (N + 1) * n, u + (size_t)b * N * m, K + (size_t)b * N * m * n, cost ? cost + b : nullptr);
    });
    return 0;
}

}  // extern "C"
