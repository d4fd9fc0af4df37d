// LOCP (horizon QP) and GuSTO (SCP outer loop) on the device: one workgroup per problem / rollout.
// Reference: sofacontrol/scp/locp.py (QP), sofacontrol/scp/gusto.py:283-487 (outer loop),
// sofacontrol/scp/models/tpwl.py:32-58 (model adapter).
#include "tpwl_host.h"
#include "locp_dev.h"
#include <cstdlib>

namespace {

constexpr int NTHREADS = 512;

struct GustoPar {
    double delta0, omega0, rho, beta_fail, gamma_fail, epsilon, omega_max, convg_thresh, dt;
    int max_iters, max_trace;
};

// ------------------------------------------------------------------ generic LOCP kernel
struct LocpBatch {
    const double *Ad, *AdT, *Bd, *BdT, *dd;     // (batch x N x ...)
    const double *x0, *xk, *delta, *omega, *z, *zf, *ud;
    double *x, *u, *s, *J;
    int32_t *status, *iters;
    double *work;
    size_t work_stride;
    double *dbg;
};

template <bool SPLIT, int MSEL, int NSEL>
__global__ __launch_bounds__(NTHREADS) void locp_kernel(QPDims d, QPConst c, LocpBatch b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    qp::specialise<MSEL, NSEL>(d);             // compile-time n_u (and n_x) for everything inlined below
    QPLds L;
    qp_lds_carve(L, (lptr)smem, d, NTHREADS);
    qp_lds_init(L, d, c);
    const size_t p = blockIdx.x;
    const size_t N = d.N, n = d.n, m = d.m;
    QPWork w;
    gptr wbase = (gptr)(b.work + p * b.work_stride);
    qp_carve(w, wbase, d);
    QPDyn dyn{(cgptr)(b.Ad + p * N * n * n), (cgptr)(b.AdT + p * N * n * n), (cgptr)(b.Bd + p * N * n * m),
              (cgptr)(b.BdT + p * N * n * m), (cgptr)(b.dd + p * N * n), (cgiptr)nullptr};
    QPData q{(cgptr)(b.x0 + p * n), (cgptr)(b.xk + p * (N + 1) * n), (cgptr)(b.z ? b.z + p * (N + 1) * d.nz : nullptr),
             (cgptr)(b.zf ? b.zf + p * d.nz : nullptr), (cgptr)(b.ud ? b.ud + p * N * m : nullptr), b.delta[p], b.omega[p],
             (gptr)((b.dbg && p == 0) ? b.dbg : nullptr)};
    double J;
    int it;
    const int st = qp::solve<SPLIT, MSEL, NSEL>(d, c, dyn, q, wbase, L, &J, &it, true, w);
    for (int e = threadIdx.x; e < (N + 1) * n; e += blockDim.x) b.x[p * (N + 1) * n + e] = w.x[e];
    for (int e = threadIdx.x; e < N * m; e += blockDim.x) b.u[p * N * m + e] = w.u[e];
    for (int e = threadIdx.x; e <= N; e += blockDim.x) b.s[p * (N + 1) + e] = w.s[e];
    if (threadIdx.x == 0) { b.J[p] = J; b.status[p] = st; b.iters[p] = it; }
}

__global__ void transpose_batch_kernel(const double *__restrict__ src, int64_t count, int rows, int cols,
                                       double *__restrict__ dst) {
    const int64_t b = blockIdx.x;
    if (b >= count) return;
    for (int e = threadIdx.x; e < rows * cols; e += blockDim.x) {
        const int i = e / cols, j = e - i * cols;
        dst[b * rows * cols + (size_t)j * rows + i] = src[b * rows * cols + e];
    }
}

// ------------------------------------------------------------------ GuSTO kernel (TPWL model)
struct GustoBatch {
    const double *x0, *u_init, *x_init, *z, *zf, *ud;
    const double *fs;                   // 1/|f_char| (n)
    double *xopt, *uopt, *zopt;
    int32_t *iters, *status;
    double *trace;
    double *work;                       // per problem: [qp work | xk | uk | ints]
    size_t work_stride;
    const int32_t *order;               // workgroup -> rollout (longest expected solve first), or null
    int32_t *last_iters;                // SCP iterations of this solve per rollout: the next solve's dispatch key
};

// Longest-processing-time-first dispatch: the rollouts of a receding-horizon batch need 1..max SCP iterations each
// and a workgroup owns its CU for the whole solve, so the tail of a launch is set by whichever long solves start
// last.  The previous solve of the same rollout predicts its length; a counting sort on those iteration counts
// (descending) gives the workgroup -> rollout map of the next launch.  Results do not depend on the order.
__global__ __launch_bounds__(1024) void lpt_order_kernel(const int32_t *__restrict__ key, int64_t batch, int32_t *__restrict__ order) {
    __shared__ int cnt[1024];
    const int tid = threadIdx.x;
    cnt[tid] = 0;
    __syncthreads();
    for (int64_t i = tid; i < batch; i += 1024) atomicAdd(&cnt[1023 - min(max(key[i], 0), 1023)], 1);
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int q = 0; q < 1024; ++q) { const int c = cnt[q]; cnt[q] = run; run += c; }
    }
    __syncthreads();
    for (int64_t i = tid; i < batch; i += 1024) order[atomicAdd(&cnt[1023 - min(max(key[i], 0), 1023)], 1)] = (int32_t)i;
}

template <bool SPLIT, int MSEL, int NSEL>
__global__ __launch_bounds__(NTHREADS) void gusto_kernel(QPDims d, QPConst c, TpwlDev T, GustoPar par, GustoBatch b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    qp::specialise<MSEL, NSEL>(d);             // compile-time n_u (and n_x) for everything inlined below
    QPLds L;
    qp_lds_carve(L, (lptr)smem, d, NTHREADS);
    qp_lds_init(L, d, c);
    const size_t p = b.order ? (size_t)b.order[blockIdx.x] : (size_t)blockIdx.x;
    const int N = d.N, n = d.n, m = d.m, nz = d.nz;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    gptr base = (gptr)(b.work + p * b.work_stride);
    QPWork w;
    qp_carve(w, base, d);
    gptr xk = base + qp_work_doubles(d);
    gptr uk = xk + (size_t)(N + 1) * n;
    gptr accb = uk + (size_t)N * m;                    // 2*N doubles: per-stage error / approx
    giptr idx = (giptr)(accb + 2 * (size_t)N);
    giptr idx2 = idx + N;

    cgptr x0 = (cgptr)(b.x0 + p * n);
    for (int e = tid; e < (N + 1) * n; e += nt) xk[e] = b.x_init[p * (size_t)(N + 1) * n + e];
    for (int e = tid; e < N * m; e += nt) uk[e] = b.u_init[p * (size_t)N * m + e];
    __syncthreads();
    tpwl::nearest_many(T, xk, n, N, idx);

    QPDyn dyn{T.Ad, T.AdT, T.Bd, T.BdT, T.dd, (cgiptr)idx};
    double delta = par.delta0, omega = par.omega0;
    double J_prev = INFINITY, d_prev = INFINITY, o_prev = INFINITY;
    bool converged = false;
    int itr = 0, status = 0;
    while (itr <= par.max_iters && !converged && omega <= par.omega_max) {
        QPData q{x0, xk, (cgptr)(b.z ? b.z + p * (size_t)(N + 1) * nz : nullptr), (cgptr)(b.zf ? b.zf + p * nz : nullptr),
                 (cgptr)(b.ud ? b.ud + p * (size_t)N * m : nullptr), delta, omega, (gptr)nullptr};
        double J;
        int qit;
        const int st = qp::solve<SPLIT, MSEL, NSEL>(d, c, dyn, q, base, L, &J, &qit, true, w);
        if (st != 0) { status = 1; break; }          // gusto.py:357-365: keep the previous iterate
        // trust region test (gusto.py:174-183)
        double md = 0.0;
        for (int e = tid; e < (N + 1) * n; e += nt) md = fmax(md, fabs(c.xs[e % n] * (w.x[e] - xk[e])));
        md = wg::reduce(md, 1, L.red);
        const bool tr_ok = !(md - delta > par.epsilon);
        bool new_solution = false;
        double rho_k = -1.0;
        const double d_cur = delta, o_cur = omega;
        if (tr_ok) {
            // model accuracy (gusto.py:203-223) with continuous nearest-point dynamics
            tpwl::nearest_many(T, w.x, n, N, idx2);
            for (int i = wave; i < N; i += nw) {
                const size_t ia = idx[i], ib = idx2[i];
                double e2 = 0.0, a2 = 0.0;
                for (int r = lane; r < n; r += 64) {
                    double fk = T.dc[ia * n + r], fl = 0.0, f = T.dc[ib * n + r];
                    cgptr Ak = T.AcT + ia * n * n, An = T.AcT + ib * n * n;
                    for (int cidx = 0; cidx < n; ++cidx) {
                        const double xo = xk[(size_t)i * n + cidx], xn = w.x[(size_t)i * n + cidx];
                        const double a = Ak[(size_t)cidx * n + r];
                        fk = fma(a, xo, fk);
                        fl = fma(a, xn - xo, fl);
                        f = fma(An[(size_t)cidx * n + r], xn, f);
                    }
                    cgptr Bk = T.BcT + ia * m * n, Bn = T.BcT + ib * m * n;
                    for (int cidx = 0; cidx < m; ++cidx) {
                        const double uo = uk[(size_t)i * m + cidx], un = w.u[(size_t)i * m + cidx];
                        const double bb = Bk[(size_t)cidx * n + r];
                        fk = fma(bb, uo, fk);
                        fl = fma(bb, un - uo, fl);
                        f = fma(Bn[(size_t)cidx * n + r], un, f);
                    }
                    const double fa = fk + fl;
                    const double fsr = b.fs[r];
                    const double de = fsr * (f - fa), da = fsr * fa;
                    e2 = fma(de, de, e2);
                    a2 = fma(da, da, a2);
                }
                e2 = wg::wave_sum(e2);
                a2 = wg::wave_sum(a2);
                if (lane == 0) { accb[2 * i] = par.dt * sqrt(e2); accb[2 * i + 1] = par.dt * sqrt(a2); }
            }
            __syncthreads();
            double err = 0.0, app = 0.0;      // sequential sums in stage order, as the reference loop
            for (int i = 0; i < N; ++i) { err += accb[2 * i]; app += accb[2 * i + 1]; }
            rho_k = err / (J + app);
            if (rho_k > par.rho && itr != 1) {
                delta = par.beta_fail * delta;
            } else {
                if (d_prev == delta && o_prev == omega && J_prev <= J) delta = par.beta_fail * delta;
                d_prev = delta; J_prev = J; o_prev = omega;
                // state-constraint violation (gusto.py:185-201): all k = 0..N
                double viol = 0.0;
                if (d.nX > 0) {
                    for (int k = tid; k <= N; k += nt) {
                        double v2 = 0.0;
                        for (int r = 0; r < d.nX; ++r) {
                            double v = -c.Xb[r];
                            for (int j = 0; j < n; ++j) v = fma(c.XA[(size_t)r * n + j], w.x[(size_t)k * n + j], v);
                            v = fmax(v, 0.0);
                            v2 = fma(v, v, v2);
                        }
                        viol = fmax(viol, sqrt(v2));
                    }
                    viol = wg::reduce(viol, 1, L.red);
                }
                const bool X_ok = !(viol > par.epsilon);
                if (!X_ok) omega = par.gamma_fail * omega;
                // convergence (gusto.py:150-161)
                double ds = 0.0;
                for (int k = wave; k <= N; k += nw) {
                    double v2 = 0.0;
                    for (int j = lane; j < n; j += 64) {
                        const double e = c.xs[j] * (w.x[(size_t)k * n + j] - xk[(size_t)k * n + j]);
                        v2 = fma(e, e, v2);
                    }
                    v2 = wg::wave_sum(v2);
                    if (lane == 0) ds += sqrt(v2);
                }
                ds = wg::reduce(ds, 0, L.red);
                const double dsol = (1.0 / N) * ((1.0 / n) * ds);
                converged = (dsol <= par.convg_thresh) && X_ok;
                new_solution = true;
            }
        } else {
            omega = par.gamma_fail * omega;
        }
        if (b.trace && itr < par.max_trace && tid == 0) {
            double *tr = b.trace + (p * par.max_trace + itr) * 4;
            tr[0] = J; tr[1] = d_cur; tr[2] = o_cur; tr[3] = rho_k;
        }
        ++itr;
        if (new_solution) {
            __syncthreads();
            for (int e = tid; e < (N + 1) * n; e += nt) xk[e] = w.x[e];
            for (int e = tid; e < N * m; e += nt) uk[e] = w.u[e];
            __syncthreads();
            if (par.max_iters >= 1) tpwl::nearest_many(T, xk, n, N, idx);
        }
    }
    if (status == 0) {
        if (omega > par.omega_max) status = 2;
        else if (itr - 1 > par.max_iters) status = 3;
    }
    __syncthreads();
    for (int e = tid; e < (N + 1) * n; e += nt) b.xopt[p * (size_t)(N + 1) * n + e] = xk[e];
    for (int e = tid; e < N * m; e += nt) b.uopt[p * (size_t)N * m + e] = uk[e];
    for (int e = tid; e < (N + 1) * nz; e += nt) {
        const int k = e / nz, a = e - k * nz;
        double v = 0.0;
        for (int j = 0; j < n; ++j) v = fma(c.H[a * n + j], xk[(size_t)k * n + j], v);
        b.zopt[p * (size_t)(N + 1) * nz + e] = v;
    }
    if (tid == 0) { b.iters[p] = itr; b.status[p] = status; if (b.last_iters) b.last_iters[p] = itr; }
}

// ------------------------------------------------------------------ host side: constants
struct QPConstHost {
    srh::DevBuf H, Qz, Qzf, R, xs, UA, Ub, XA, Xb, XfA, Xfb, Qx, QxN, HtQz2, HtQzf2, R2, Cq;
    QPDims dims{};
    QPConst view() const {
        QPConst c{};
        auto g = [](const srh::DevBuf &b) { return (cgptr)b.as<double>(); };
        c.H = g(H); c.Qz = g(Qz); c.Qzf = g(Qzf); c.R = g(R);
        c.xs = g(xs); c.UA = g(UA); c.Ub = g(Ub); c.XA = g(XA);
        c.Xb = g(Xb); c.XfA = g(XfA); c.Xfb = g(Xfb); c.Qx = g(Qx);
        c.QxN = g(QxN); c.HtQz2 = g(HtQz2); c.HtQzf2 = g(HtQzf2);
        c.R2 = g(R2);
        c.Cq = g(Cq);
        return c;
    }
};

// eigen-decomposition of a small symmetric matrix (cyclic Jacobi): A = V diag(w) V^T, V columns
static void jacobi_eig(std::vector<double> A, int n, std::vector<double> &w, std::vector<double> &V) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                if (fabs(A[p * n + q]) < 1e-300) continue;
                const double th = (A[q * n + q] - A[p * n + p]) / (2.0 * A[p * n + q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = cs * akp - sn * akq; A[k * n + q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = cs * apk - sn * aqk; A[q * n + k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = cs * vkp - sn * vkq; V[k * n + q] = sn * vkp + cs * vkq;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

int build_consts(const slocp_problem *pr, QPConstHost &C) {
    SRH_REQUIRE(pr && pr->H && pr->Qz && pr->R, "LOCP: H, Qz and R are required");
    const int N = pr->N, n = pr->n_x, m = pr->n_u, nz = pr->n_z;
    SRH_REQUIRE(N >= 1 && n >= 1 && n <= 128 && m >= 1 && m <= 16 && nz >= 1 && nz <= 16,
                "LOCP: need 1 <= N, 1 <= n_x <= 128, 1 <= n_u <= 16, 1 <= n_z <= 16");
    SRH_REQUIRE(pr->ndU == 0, "LOCP: dU (input-rate) constraints are not supported by the device solver yet");
    SRH_REQUIRE(pr->nU >= 0 && pr->nX >= 0 && pr->nXf >= 0, "LOCP: negative constraint count");
    SRH_REQUIRE(pr->nX + pr->nXf <= 32 && pr->nU <= 64, "LOCP: at most 32 state rows (X + Xf) and 64 input rows per stage");
    SRH_REQUIRE(pr->nU == 0 || (pr->UA && pr->Ub), "LOCP: U.A / U.b missing");
    SRH_REQUIRE(pr->nX == 0 || (pr->XA && pr->Xb), "LOCP: X.A / X.b missing");
    SRH_REQUIRE(pr->nXf == 0 || (pr->XfA && pr->Xfb), "LOCP: Xf.A / Xf.b missing");
    QPDims &d = C.dims;
    d.N = N; d.n = n; d.m = m; d.nz = nz; d.nU = pr->nU; d.nX = pr->nX; d.nXf = pr->nXf;
    d.tr = pr->tr_active ? 1 : 0;
    d.NPa = (n + m + 15) & ~15;
    d.ld = d.NPa + 1;
    d.mp = (m + 3) & ~3;
    d.NK = (n + 3) & ~3;
    d.NE4 = (m + pr->nX + 3) & ~3;
    d.split = 0; d.WR = 0; d.nzr = 0; d.RC = 0; d.RW = 0;
    // 2 H^T Qz H = Cq^T Cq: constant extra rows of the Gram product when the panels still fit in LDS
    std::vector<double> Cq;
    {
        std::vector<double> Qs((size_t)nz * nz), wv, V;
        for (int a = 0; a < nz; ++a) for (int b = 0; b < nz; ++b) Qs[a * nz + b] = 0.5 * (pr->Qz[a * nz + b] + pr->Qz[b * nz + a]);
        jacobi_eig(Qs, nz, wv, V);
        double wmax = 0.0;
        for (double x : wv) wmax = std::max(wmax, fabs(x));
        for (int e = 0; e < nz; ++e) {
            if (wv[e] <= 1e-13 * wmax) continue;
            const double sc = sqrt(2.0 * wv[e]);
            for (int j = 0; j < n; ++j) {
                double v = 0.0;
                for (int a = 0; a < nz; ++a) v += V[a * nz + e] * pr->H[a * n + j];
                Cq.push_back(sc * v);
            }
        }
    }
    const int nzr_full = (int)(Cq.size() / n);
    // layout: whole W panel in LDS when it fits, else the split variant (W holds 48 rows at a time)
    auto layout = [&](int split, int nzr) {
        QPDims t = d;
        t.split = split;
        t.nzr = nzr;
        t.RC = std::max(t.NK + m + pr->nX, (n + 15) & ~15);
        t.RW = std::max((n + 15) & ~15, t.NK + std::max(t.NE4, (m + 3) & ~3));
        if (nzr > 0) t.RW = std::max(t.RW, (t.RC + nzr + 3) & ~3);
        t.WR = split ? 48 : t.RW;
        return t;
    };
    {
        const size_t lim = 160 * 1024;
        QPDims best = layout(0, 0);
        bool found = false;
        for (int split = 0; split < 2 && !found; ++split) {
            for (int nzr : {nzr_full, 0}) {
                QPDims t = layout(split, nzr);
                const int KE = (nzr ? ((t.RC + nzr + 3) & ~3) : t.NK + t.NE4) - t.NK;
                if (split && (t.NK <= 48 || t.NK > 96 || KE > 48)) continue;
                if (qp_lds_bytes(t, NTHREADS) <= lim) { best = t; found = true; break; }
            }
        }
        d = best;
    }
    d.nrx = d.tr * (2 * n + 1) + d.nX;
    d.RX = d.nrx + d.nXf;
    d.NR = N * d.RX + N * d.nU;
    d.ng = N * d.nrx + d.nXf + N * d.nU;
    d.max_iter = 60;
    d.tol = 1e-12;
    d.reg = 1e-8;
    std::vector<double> Qx(n * n), QxN(n * n), Ht2(n * nz), Htf2(n * nz, 0.0), R2(m * m), xs(n, 1.0);
    std::vector<double> QzH(nz * n), QzfH(nz * n, 0.0);
    for (int a = 0; a < nz; ++a)
        for (int j = 0; j < n; ++j) {
            double v = 0.0, vf = 0.0;
            for (int b = 0; b < nz; ++b) {
                v += pr->Qz[a * nz + b] * pr->H[b * n + j];
                if (pr->Qzf) vf += pr->Qzf[a * nz + b] * pr->H[b * n + j];
            }
            QzH[a * n + j] = v; QzfH[a * n + j] = vf;
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double v = 0.0, vf = 0.0;
            for (int a = 0; a < nz; ++a) { v += pr->H[a * n + i] * QzH[a * n + j]; vf += pr->H[a * n + i] * QzfH[a * n + j]; }
            Qx[i * n + j] = 2.0 * v; QxN[i * n + j] = 2.0 * (v + vf);
        }
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < nz; ++a) {
            double v = 0.0, vf = 0.0;
            for (int b = 0; b < nz; ++b) {
                v += pr->H[b * n + i] * pr->Qz[b * nz + a];
                if (pr->Qzf) vf += pr->H[b * n + i] * pr->Qzf[b * nz + a];
            }
            Ht2[i * nz + a] = 2.0 * v; Htf2[i * nz + a] = 2.0 * vf;
        }
    for (int e = 0; e < m * m; ++e) R2[e] = 2.0 * pr->R[e];
    if (pr->x_scale) xs.assign(pr->x_scale, pr->x_scale + n);
    int rc;
    const double zero = 0.0;
    auto up = [&](srh::DevBuf &b, const double *src, size_t cnt) { return cnt ? b.upload(src, sizeof(double) * cnt) : b.upload(&zero, sizeof(double)); };
    if ((rc = up(C.H, pr->H, (size_t)nz * n)) || (rc = up(C.Qz, pr->Qz, (size_t)nz * nz)) || (rc = up(C.R, pr->R, (size_t)m * m)) ||
        (rc = up(C.xs, xs.data(), n)) || (rc = up(C.UA, pr->UA, (size_t)pr->nU * m)) || (rc = up(C.Ub, pr->Ub, pr->nU)) ||
        (rc = up(C.XA, pr->XA, (size_t)pr->nX * n)) || (rc = up(C.Xb, pr->Xb, pr->nX)) ||
        (rc = up(C.XfA, pr->XfA, (size_t)pr->nXf * n)) || (rc = up(C.Xfb, pr->Xfb, pr->nXf)) ||
        (rc = up(C.Qx, Qx.data(), (size_t)n * n)) || (rc = up(C.QxN, QxN.data(), (size_t)n * n)) ||
        (rc = up(C.HtQz2, Ht2.data(), (size_t)n * nz)) || (rc = up(C.HtQzf2, Htf2.data(), (size_t)n * nz)) ||
        (rc = up(C.R2, R2.data(), (size_t)m * m)) || (rc = up(C.Cq, Cq.data(), Cq.size())))
        return rc;
    if (pr->Qzf && (rc = up(C.Qzf, pr->Qzf, (size_t)nz * nz))) return rc;
    return SRH_OK;
}

int set_lds_limit(const void *kernel, size_t bytes) {
    SRH_REQUIRE(bytes <= 160 * 1024, "LOCP: problem too large for LDS (%zu bytes needed, 160 KiB available)", bytes);
    SRH_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SRH_OK;
}

// Kernel variants by (split panel, n_u, n_x): instantiations for the reference's 4- and 8-cable robots, with n_x fixed
// as well for the benchmark's r = 30 and the shipped r = 36 Diamond model; the all-sizes kernel otherwise.
// Further shapes are a build-time list:  make EXTRA="'-DSRH_QP_EXTRA_VARIANTS(X)=X(false,8,44)X(true,4,80)'"
// (split panel is true for 64 < n_x <= 96).
#ifndef SRH_QP_EXTRA_VARIANTS
#define SRH_QP_EXTRA_VARIANTS(X)
#endif
#define SRH_QP_VARIANTS(X)                                                                  \
    SRH_QP_EXTRA_VARIANTS(X)                                                                \
    X(false, 4, 60) X(false, 8, 60) X(true, 4, 72)                                          \
    X(false, 4, 0) X(false, 8, 0) X(false, 0, 0) X(true, 4, 0) X(true, 8, 0) X(true, 0, 0)
inline bool variant_matches(const QPDims &d, bool sp, int msel, int nsel) {
    return (d.split != 0) == sp && (msel == 0 || d.m == msel) && (nsel == 0 || d.n == nsel);
}
const void *gusto_entry(const QPDims &d) {
#define X(SP, M, NX) if (variant_matches(d, SP, M, NX)) return (const void *)gusto_kernel<SP, M, NX>;
    SRH_QP_VARIANTS(X)
#undef X
    return nullptr;
}
const void *locp_entry(const QPDims &d) {
#define X(SP, M, NX) if (variant_matches(d, SP, M, NX)) return (const void *)locp_kernel<SP, M, NX>;
    SRH_QP_VARIANTS(X)
#undef X
    return nullptr;
}

}  // namespace

// ------------------------------------------------------------------ GuSTO plan (resident solver)
struct sgusto_plan {
    stpwl *model = nullptr;
    QPConstHost C;
    GustoPar par{};
    int64_t batch = 0;
    srh::DevBuf fs, work, x0, u_init, x_init, z, zf, ud, xopt, uopt, zopt, iters, status, trace, order, last_iters;
    bool have_last = false;             // a previous solve left its iteration counts
    size_t work_stride = 0;
    size_t lds = 0;
    bool has_z = false, has_zf = false, has_ud = false;
};

extern "C" {

void sgusto_default_params(sgusto_params *p) {
    if (!p) return;
    p->delta0 = 1e4; p->omega0 = 1.0; p->rho = 0.1; p->beta_fail = 0.5; p->gamma_fail = 5.0;
    p->epsilon = 0.01; p->omega_max = 1e10; p->convg_thresh = 0.1; p->max_gusto_iters = 500;
}

int slocp_solve(const slocp_problem *prob, int64_t batch, const double *Ad, const double *Bd, const double *dd,
                const double *x0, const double *xk, const double *delta, const double *omega, const double *z,
                const double *zf, const double *u_des, double *x, double *u, double *s, double *J,
                int32_t *status, int32_t *iters) {
    SRH_REQUIRE(prob && Ad && Bd && dd && x0 && delta && omega && x && u && J && status,
                "slocp_solve: null argument");
    SRH_REQUIRE(batch >= 0, "slocp_solve: negative batch");
    SRH_REQUIRE(!prob->tr_active || xk, "slocp_solve: xk is required when the trust region is active");
    if (batch == 0) return SRH_OK;
    QPConstHost C;
    int rc = build_consts(prob, C);
    if (rc) return rc;
    const QPDims &d = C.dims;
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz;
    srh::DevBuf dA, dAT, dB, dBT, dD, dx0, dxk, ddel, dom, dz, dzf, dud, ox, ou, os, oJ, ost, oit, work;
    std::vector<double> xk0;
    if (!xk) xk0.assign(batch * (N + 1) * n, 0.0);
    if ((rc = dA.upload(Ad, sizeof(double) * batch * N * n * n)) || (rc = dAT.alloc(sizeof(double) * batch * N * n * n)) ||
        (rc = dB.upload(Bd, sizeof(double) * batch * N * n * m)) || (rc = dBT.alloc(sizeof(double) * batch * N * n * m)) ||
        (rc = dD.upload(dd, sizeof(double) * batch * N * n)) || (rc = dx0.upload(x0, sizeof(double) * batch * n)) ||
        (rc = dxk.upload(xk ? xk : xk0.data(), sizeof(double) * batch * (N + 1) * n)) ||
        (rc = ddel.upload(delta, sizeof(double) * batch)) || (rc = dom.upload(omega, sizeof(double) * batch)) ||
        (rc = ox.alloc(sizeof(double) * batch * (N + 1) * n)) || (rc = ou.alloc(sizeof(double) * batch * N * m)) ||
        (rc = os.alloc(sizeof(double) * batch * (N + 1))) || (rc = oJ.alloc(sizeof(double) * batch)) ||
        (rc = ost.alloc(sizeof(int32_t) * batch)) || (rc = oit.alloc(sizeof(int32_t) * batch)))
        return rc;
    if (z && (rc = dz.upload(z, sizeof(double) * batch * (N + 1) * nz))) return rc;
    if (zf && (rc = dzf.upload(zf, sizeof(double) * batch * nz))) return rc;
    if (u_des && (rc = dud.upload(u_des, sizeof(double) * batch * N * m))) return rc;
    const size_t stride = (qp_work_doubles(d) + 3) & ~(size_t)3;
    if ((rc = work.alloc(sizeof(double) * stride * batch))) return rc;
    transpose_batch_kernel<<<(unsigned)(batch * N), 256>>>(dA.as<double>(), batch * N, (int)n, (int)n, dAT.as<double>());
    transpose_batch_kernel<<<(unsigned)(batch * N), 256>>>(dB.as<double>(), batch * N, (int)n, (int)m, dBT.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    LocpBatch b{dA.as<double>(), dAT.as<double>(), dB.as<double>(), dBT.as<double>(), dD.as<double>(), dx0.as<double>(),
                dxk.as<double>(), ddel.as<double>(), dom.as<double>(), z ? dz.as<double>() : nullptr,
                zf ? dzf.as<double>() : nullptr, u_des ? dud.as<double>() : nullptr, ox.as<double>(), ou.as<double>(),
                os.as<double>(), oJ.as<double>(), ost.as<int32_t>(), oit.as<int32_t>(), work.as<double>(), stride, nullptr};
    srh::DevBuf dbg;
    const bool want_dbg = getenv("SRH_LOCP_TRACE") != nullptr;
    if (want_dbg) { if ((rc = dbg.alloc(sizeof(double) * 8 * 64))) return rc; (void)hipMemset(dbg.p, 0, sizeof(double) * 8 * 64); b.dbg = dbg.as<double>(); }
    const size_t lds = qp_lds_bytes(d, NTHREADS);
    if ((rc = set_lds_limit(locp_entry(d), lds))) return rc;
    {
        bool launched = false;
#define X(SP, M, NX) if (!launched && variant_matches(d, SP, M, NX)) { locp_kernel<SP, M, NX><<<(unsigned)batch, NTHREADS, lds>>>(d, C.view(), b); launched = true; }
        SRH_QP_VARIANTS(X)
#undef X
    }
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipDeviceSynchronize());
    if (want_dbg) {
        std::vector<double> t(8 * 64);
        dbg.download(t.data(), sizeof(double) * 8 * 64);
        fprintf(stderr, "[locp] time (shader clocks): init %.0f rows %.0f prepass %.0f ricc_full %.0f ricc_vec %.0f final %.0f\n", t[8*62], t[8*62+1], t[8*62+2], t[8*62+3], t[8*62+4], t[8*62+5]);
        fprintf(stderr, "[locp] riccati laps (SRH_PROFILE build): load %.0f W %.0f BtW %.0f Qu %.0f gain %.0f AtW+finish %.0f vec-backward %.0f forward %.0f\n",
                t[8*63], t[8*63+1], t[8*63+2], t[8*63+3], t[8*63+4], t[8*63+5], t[8*63+6], t[8*63+7]);
        fprintf(stderr, "[locp] riccati phases (shader clocks): load %.0f gemm1 %.0f gemm2 %.0f matvec %.0f chol+K %.0f Pnew %.0f (vec/other %.0f) fwd %.0f\n", t[8*63], t[8*63+1], t[8*63+2], t[8*63+3], t[8*63+4], t[8*63+5], t[8*63+6], t[8*63+7]);
        for (int i = 0; i < 62 && (t[8 * i + 3] != 0.0); ++i)
            fprintf(stderr, "[locp] it %2d mu %.3e rd %.3e rp %.3e (sd %.2e sp %.2e) a_aff %.3e sigma %.3e a %.3e\n", i, t[8 * i], t[8 * i + 1], t[8 * i + 2], t[8 * i + 3], t[8 * i + 4], t[8 * i + 5], t[8 * i + 6], t[8 * i + 7]);
    }
    if ((rc = ox.download(x, sizeof(double) * batch * (N + 1) * n)) || (rc = ou.download(u, sizeof(double) * batch * N * m)) ||
        (rc = oJ.download(J, sizeof(double) * batch)) || (rc = ost.download(status, sizeof(int32_t) * batch)))
        return rc;
    if (s && (rc = os.download(s, sizeof(double) * batch * (N + 1)))) return rc;
    if (iters && (rc = oit.download(iters, sizeof(int32_t) * batch))) return rc;
    return SRH_OK;
}

int sgusto_plan_create(sgusto_plan_t **out, stpwl_t *h, const slocp_problem *prob, const sgusto_params *par,
                       double dt, int64_t batch, const double *x_char, const double *f_char, int max_trace) {
    SRH_REQUIRE(out && h && prob && par, "sgusto_plan_create: null argument");
    SRH_REQUIRE(h->has_discrete, "sgusto_plan_create: model has not been pre-discretised (TPWLGuSTO.pre_discretize)");
    SRH_REQUIRE(prob->n_x == h->n && prob->n_u == h->m, "sgusto_plan_create: problem / model dimension mismatch");
    SRH_REQUIRE(batch > 0, "sgusto_plan_create: batch must be positive");
    sgusto_plan *pl = new sgusto_plan();
    pl->model = h;
    pl->batch = batch;
    slocp_problem p2 = *prob;
    std::vector<double> xs(h->n, 1.0), fs(h->n, 1.0);
    if (x_char) for (int i = 0; i < h->n; ++i) xs[i] = 1.0 / fabs(x_char[i]);
    if (f_char) for (int i = 0; i < h->n; ++i) fs[i] = 1.0 / fabs(f_char[i]);
    if (x_char) p2.x_scale = xs.data();
    int rc = build_consts(&p2, pl->C);
    if (rc) { delete pl; return rc; }
    const QPDims &d = pl->C.dims;
    pl->par = GustoPar{par->delta0, par->omega0, par->rho, par->beta_fail, par->gamma_fail, par->epsilon,
                       par->omega_max, par->convg_thresh, dt, par->max_gusto_iters, max_trace};
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz;
    size_t doubles = qp_work_doubles(d) + (N + 1) * n + N * m + 2 * N + (2 * N + 1) / 2 + 8;
    pl->work_stride = (doubles + 3) & ~(size_t)3;
    pl->lds = qp_lds_bytes(d, NTHREADS);
    if ((rc = pl->fs.upload(fs.data(), sizeof(double) * n)) || (rc = pl->work.alloc(sizeof(double) * pl->work_stride * batch)) ||
        (rc = pl->x0.alloc(sizeof(double) * batch * n)) || (rc = pl->u_init.alloc(sizeof(double) * batch * N * m)) ||
        (rc = pl->x_init.alloc(sizeof(double) * batch * (N + 1) * n)) || (rc = pl->z.alloc(sizeof(double) * batch * (N + 1) * nz)) ||
        (rc = pl->zf.alloc(sizeof(double) * batch * nz)) || (rc = pl->ud.alloc(sizeof(double) * batch * N * m)) ||
        (rc = pl->xopt.alloc(sizeof(double) * batch * (N + 1) * n)) || (rc = pl->uopt.alloc(sizeof(double) * batch * N * m)) ||
        (rc = pl->zopt.alloc(sizeof(double) * batch * (N + 1) * nz)) || (rc = pl->iters.alloc(sizeof(int32_t) * batch)) ||
        (rc = pl->status.alloc(sizeof(int32_t) * batch)) || (rc = pl->order.alloc(sizeof(int32_t) * batch)) ||
        (rc = pl->last_iters.alloc(sizeof(int32_t) * batch)) ||
        (rc = pl->trace.alloc(sizeof(double) * batch * (size_t)std::max(1, max_trace) * 4)) ||
        (rc = set_lds_limit(gusto_entry(d), pl->lds))) {
        delete pl;
        return rc;
    }
    *out = pl;
    return SRH_OK;
}

int sgusto_plan_destroy(sgusto_plan_t *pl) {
    delete pl;
    return SRH_OK;
}

int sgusto_plan_set_max_iters(sgusto_plan_t *pl, int max_gusto_iters) {
    SRH_REQUIRE(pl, "sgusto_plan_set_max_iters: null plan");
    pl->par.max_iters = max_gusto_iters;
    return SRH_OK;
}

int sgusto_plan_solve_dev(sgusto_plan_t *pl, const double *x0, const double *u_init, const double *x_init,
                          const double *z, const double *zf, const double *u_des, double *xopt, double *uopt,
                          double *zopt, int32_t *iters, int32_t *status, double *trace, void *stream) {
    SRH_REQUIRE(pl && x0 && u_init && x_init && xopt && uopt && zopt && iters && status,
                "sgusto_plan_solve_dev: null argument");
    GustoBatch b{x0, u_init, x_init, z, zf, u_des, pl->fs.as<double>(), xopt, uopt, zopt, iters, status, trace,
                 pl->work.as<double>(), pl->work_stride, nullptr, pl->last_iters.as<int32_t>()};
    if (pl->have_last && pl->batch > 256 && !getenv("SRH_GUSTO_NO_LPT")) {      // more rollouts than CUs: order matters
        lpt_order_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(pl->last_iters.as<int32_t>(), pl->batch, pl->order.as<int32_t>());
        b.order = pl->order.as<int32_t>();
    }
    pl->have_last = true;
    GustoPar par = pl->par;
    if (!trace) par.max_trace = 0;
    {
        const QPDims &d = pl->C.dims;
        bool launched = false;
#define X(SP, M, NX) if (!launched && variant_matches(d, SP, M, NX)) { gusto_kernel<SP, M, NX><<<(unsigned)pl->batch, NTHREADS, pl->lds, (hipStream_t)stream>>>(d, pl->C.view(), pl->model->view(), par, b); launched = true; }
        SRH_QP_VARIANTS(X)
#undef X
    }
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

int sgusto_plan_solve(sgusto_plan_t *pl, const double *x0, const double *u_init, const double *x_init,
                      const double *z, const double *zf, const double *u_des, double *xopt, double *uopt,
                      double *zopt, int32_t *iters, int32_t *status, double *trace) {
    SRH_REQUIRE(pl && x0 && u_init && x_init && xopt && uopt && zopt, "sgusto_plan_solve: null argument");
    const QPDims &d = pl->C.dims;
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz, B = pl->batch;
    SRH_CHECK_HIP(hipMemcpy(pl->x0.p, x0, sizeof(double) * B * n, hipMemcpyHostToDevice));
    SRH_CHECK_HIP(hipMemcpy(pl->u_init.p, u_init, sizeof(double) * B * N * m, hipMemcpyHostToDevice));
    SRH_CHECK_HIP(hipMemcpy(pl->x_init.p, x_init, sizeof(double) * B * (N + 1) * n, hipMemcpyHostToDevice));
    if (z) SRH_CHECK_HIP(hipMemcpy(pl->z.p, z, sizeof(double) * B * (N + 1) * nz, hipMemcpyHostToDevice));
    if (zf) SRH_CHECK_HIP(hipMemcpy(pl->zf.p, zf, sizeof(double) * B * nz, hipMemcpyHostToDevice));
    if (u_des) SRH_CHECK_HIP(hipMemcpy(pl->ud.p, u_des, sizeof(double) * B * N * m, hipMemcpyHostToDevice));
    int rc = sgusto_plan_solve_dev(pl, pl->x0.as<double>(), pl->u_init.as<double>(), pl->x_init.as<double>(),
                                   z ? pl->z.as<double>() : nullptr, zf ? pl->zf.as<double>() : nullptr,
                                   u_des ? pl->ud.as<double>() : nullptr, pl->xopt.as<double>(), pl->uopt.as<double>(),
                                   pl->zopt.as<double>(), pl->iters.as<int32_t>(), pl->status.as<int32_t>(),
                                   trace ? pl->trace.as<double>() : nullptr, nullptr);
    if (rc) return rc;
    SRH_CHECK_HIP(hipDeviceSynchronize());
    if ((rc = pl->xopt.download(xopt, sizeof(double) * B * (N + 1) * n)) || (rc = pl->uopt.download(uopt, sizeof(double) * B * N * m)) ||
        (rc = pl->zopt.download(zopt, sizeof(double) * B * (N + 1) * nz)))
        return rc;
    if (iters && (rc = pl->iters.download(iters, sizeof(int32_t) * B))) return rc;
    if (status && (rc = pl->status.download(status, sizeof(int32_t) * B))) return rc;
    if (trace && (rc = pl->trace.download(trace, sizeof(double) * B * pl->par.max_trace * 4))) return rc;
    return SRH_OK;
}

int sgusto_solve(stpwl_t *h, const slocp_problem *prob, const sgusto_params *par, double dt, int64_t batch,
                 const double *x0, const double *u_init, const double *x_init, const double *z, const double *zf,
                 const double *u_des, const double *x_char, const double *f_char, double *xopt, double *uopt,
                 double *zopt, int32_t *iters, int32_t *status, double *trace, int max_trace) {
    sgusto_plan *pl = nullptr;
    int rc = sgusto_plan_create(&pl, h, prob, par, dt, batch, x_char, f_char, trace ? max_trace : 0);
    if (rc) return rc;
    rc = sgusto_plan_solve(pl, x0, u_init, x_init, z, zf, u_des, xopt, uopt, zopt, iters, status, trace);
    sgusto_plan_destroy(pl);
    return rc;
}

}  // extern "C"
