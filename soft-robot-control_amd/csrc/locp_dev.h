// Device-side solver for the LOCP horizon QP (sofacontrol/scp/locp.py:218-342): Mehrotra
// predictor-corrector interior point whose Newton systems are solved by a backward Riccati
// factorisation over the horizon (dynamics + x_0 = x0 eliminated) -- one workgroup per QP, the
// cost-to-go matrix and the stage Jacobian resident in LDS.  The numpy statement of exactly this
// algorithm is oracle/riccati_ipm.py; the QP itself is pinned by oracle/locp.py.
//
// Stage layout: x_0 is fixed (its trust-region slack s_0 has the closed form
// max(0, ||xs*(x0-xbar_0)||_inf - delta)); stage k = 0..N-1 owns u_k and the U rows; x_k (k = 1..N)
// owns s_k, the 2n+1 trust-region rows, the X rows and (k = N) the Xf rows.
#pragma once
#include "dev_la.h"

struct QPDims {
    int N, n, m, nz, nU, nX, nXf, tr;
    int ld;    // leading dimension of n x n LDS matrices (multiple of 4)
    int mp;    // leading dimension of n x m LDS matrices (multiple of 4)
    int nrx;   // inequality rows owned by x_k, k < N:  tr*(2n+1) + nX
    int RX;    // row stride per x stage: nrx + nXf
    int NR;    // total rows: N*RX + N*nU
    int ng;    // active rows: N*nrx + nXf + N*nU
    int max_iter;
    double tol;
    double reg;   // dual (proximal) regularisation of the Newton systems, relative to the dual scale
};

struct QPConst {                       // shared by the whole batch (HBM/L2 resident)
    const double *H, *Qz, *Qzf, *R;    // (nz x n), (nz x nz), (nz x nz)|null, (m x m)
    const double *xs;                  // (n) trust-region scaling
    const double *UA, *Ub, *XA, *Xb, *XfA, *Xfb;
    const double *Qx, *QxN;            // 2 H^T Qz H, (+ 2 H^T Qzf H)      (n x n)
    const double *HtQz2, *HtQzf2;      // 2 H^T Qz, 2 H^T Qzf               (n x nz)
    const double *R2;                  // 2 R
};

struct QPDyn {                         // stage dynamics: matrix k at base + idx[k]*size (idx null: k)
    const double *A, *AT, *B, *BT, *d;
    const int *idx;
    __device__ __forceinline__ size_t sel(int k) const { return idx ? (size_t)idx[k] : (size_t)k; }
};

struct QPData {                        // one problem
    const double *x0, *xk, *z, *zf, *ud;   // xk (N+1 x n); z (N+1 x nz)|null; zf (nz)|null; ud (N x m)|null
    double delta, omega;
    double *dbg;                           // optional per-iteration trace (8 doubles per iteration) or null
};

struct QPWork {                        // per-problem scratch in HBM/L2 (doubles)
    double *x, *u, *s, *dx, *du, *ds;
    double *t, *lam, *rg, *D, *rho, *rc, *dt, *dlam;      // NR each
    double *hd, *cv, *gx, *gxd;                           // (N+1) x n   (index k = 1..N used)
    double *Hss, *gs;                                     // (N+1)
    double *Huu, *gu, *gud;                               // N x m x m, N x m, N x m
    double *K, *Qinv, *kff;                               // N x m x n, N x m x m (Cholesky factors of Quu), N x m
    double *ez;                                           // (N+1) x nz
};

__host__ __device__ inline size_t qp_work_doubles(const QPDims &d) {
    const size_t N = d.N, n = d.n, m = d.m;
    return (N + 1) * n * 2 + N * m * 2 + (N + 1) * 2 + 8 * (size_t)d.NR + 4 * (N + 1) * n + 2 * (N + 1) +
           N * m * m + 2 * N * m + N * m * n + N * m * m + N * m + (N + 1) * d.nz + 64;
}

__device__ inline void qp_carve(QPWork &w, double *base, const QPDims &d) {
    const size_t N = d.N, n = d.n, m = d.m, NR = d.NR;
    double *p = base;
    auto take = [&](size_t c) { double *q = p; p += c; return q; };
    w.x = take((N + 1) * n); w.dx = take((N + 1) * n);
    w.u = take(N * m); w.du = take(N * m);
    w.s = take(N + 1); w.ds = take(N + 1);
    w.t = take(NR); w.lam = take(NR); w.rg = take(NR); w.D = take(NR);
    w.rho = take(NR); w.rc = take(NR); w.dt = take(NR); w.dlam = take(NR);
    w.hd = take((N + 1) * n); w.cv = take((N + 1) * n); w.gx = take((N + 1) * n); w.gxd = take((N + 1) * n);
    w.Hss = take(N + 1); w.gs = take(N + 1);
    w.Huu = take(N * m * m); w.gu = take(N * m); w.gud = take(N * m);
    w.K = take(N * m * n); w.Qinv = take(N * m * m); w.kff = take(N * m);
    w.ez = take((N + 1) * d.nz);
}

struct QPLds {                         // LDS carve (doubles unless noted)
    double *P, *W, *A;                 // n x ld each
    double *B, *G;                     // n x mp
    double *Qux, *Km;                  // m x ld
    double *Quu, *Qinv, *Lb;           // m x m
    double *pv, *adj, *v1, *v2, *v3;   // n each
    double *Qu, *kf, *rdu;             // m each (padded)
    double *part;                      // blockDim
    double *red;                       // 16
    int *flag;                         // 4 ints
};

__host__ __device__ inline size_t qp_lds_bytes(const QPDims &d, int nthreads) {
    size_t c = 3 * (size_t)d.n * d.ld + 2 * (size_t)d.n * d.mp + 2 * (size_t)d.m * d.ld + 3 * 16 * 16 +
               5 * (size_t)d.ld + 3 * 16 + nthreads + 16 + 4;
    return c * sizeof(double);
}

__device__ inline void qp_lds_carve(QPLds &L, double *base, const QPDims &d, int nthreads) {
    double *p = base;
    auto take = [&](size_t c) { double *q = p; p += c; return q; };
    L.P = take((size_t)d.n * d.ld); L.W = take((size_t)d.n * d.ld); L.A = take((size_t)d.n * d.ld);
    L.B = take((size_t)d.n * d.mp); L.G = take((size_t)d.n * d.mp);
    L.Qux = take((size_t)d.m * d.ld); L.Km = take((size_t)d.m * d.ld);
    L.Quu = take(256); L.Qinv = take(256); L.Lb = take(256);
    L.pv = take(d.ld); L.adj = take(d.ld); L.v1 = take(d.ld); L.v2 = take(d.ld); L.v3 = take(d.ld);
    L.Qu = take(16); L.kf = take(16); L.rdu = take(16);
    L.part = take(nthreads);
    L.red = take(16);
    L.flag = reinterpret_cast<int *>(take(4));
}

namespace qp {

// ------------------------------------------------------------------ inequality rows
// value of row r of x-stage k (1..N) applied to the vector (vx, sv):  a_x . vx + a_s * sv
__device__ __forceinline__ double xrow_dot(const QPDims &d, const QPConst &c, int k, int r, const double *vx,
                                           double sv) {
    const int n = d.n;
    if (d.tr) {
        if (r < n) return c.xs[r] * vx[r] - sv;
        if (r < 2 * n) return -c.xs[r - n] * vx[r - n] - sv;
        if (r == 2 * n) return -sv;
        r -= 2 * n + 1;
    }
    const double *row = (r < d.nX) ? c.XA + (size_t)r * n : c.XfA + (size_t)(r - d.nX) * n;
    double acc = 0.0;
    for (int j = 0; j < n; ++j) acc = fma(row[j], vx[j], acc);
    return acc;
}
__device__ __forceinline__ double xrow_h(const QPDims &d, const QPConst &c, const QPData &q, int k, int r) {
    const int n = d.n;
    if (d.tr) {
        if (r < n) return q.delta + c.xs[r] * q.xk[(size_t)k * n + r];
        if (r < 2 * n) return q.delta - c.xs[r - n] * q.xk[(size_t)k * n + r - n];
        if (r == 2 * n) return 0.0;
        r -= 2 * n + 1;
    }
    return (r < d.nX) ? c.Xb[r] : c.Xfb[r - d.nX];
}
__device__ __forceinline__ int xrows_of(const QPDims &d, int k) { return d.nrx + (k == d.N ? d.nXf : 0); }

// iterate over all active rows: f(rowIndex, isU, k, r).  x rows first, then u rows.
template <typename F>
__device__ __forceinline__ void for_rows(const QPDims &d, F f) {
    const int nxr = d.N * d.RX;
    for (int e = threadIdx.x; e < nxr; e += blockDim.x) {
        const int k = e / d.RX + 1, r = e - (k - 1) * d.RX;
        if (r < xrows_of(d, k)) f(e, false, k, r);
    }
    const int nur = d.N * d.nU;
    for (int e = threadIdx.x; e < nur; e += blockDim.x) {
        const int k = e / d.nU, r = e - k * d.nU;
        f(nxr + e, true, k, r);
    }
}

// a . w for every row with w = (vx, vs, vu); out[row]
__device__ inline void rows_apply(const QPDims &d, const QPConst &c, const double *vx, const double *vs,
                                  const double *vu, double *out) {
    for_rows(d, [&](int row, bool isU, int k, int r) {
        if (!isU) {
            out[row] = xrow_dot(d, c, k, r, vx + (size_t)k * d.n, vs[k]);
        } else {
            const double *ua = c.UA + (size_t)r * d.m, *uk = vu + (size_t)k * d.m;
            double acc = 0.0;
            for (int j = 0; j < d.m; ++j) acc = fma(ua[j], uk[j], acc);
            out[row] = acc;
        }
    });
}
__device__ __forceinline__ double row_h(const QPDims &d, const QPConst &c, const QPData &q, bool isU, int k, int r) {
    return isU ? c.Ub[r] : xrow_h(d, c, q, k, r);
}

// ------------------------------------------------------------------ rollout x = f(u)
__device__ inline void rollout(const QPDims &d, const QPDyn &dyn, const QPData &q, const double *u, double *x,
                               QPLds &L) {
    const int n = d.n, m = d.m;
    for (int e = threadIdx.x; e < n; e += blockDim.x) { L.v1[e] = q.x0[e]; x[e] = q.x0[e]; }
    __syncthreads();
    for (int k = 0; k < d.N; ++k) {
        const size_t i = dyn.sel(k);
        for (int e = threadIdx.x; e < m; e += blockDim.x) L.Qu[e] = u[(size_t)k * m + e];
        __syncthreads();
        wg::matTvec(L.v2, dyn.AT + i * n * n, n, n, n, L.v1, dyn.d + i * n, L.part);
        wg::matTvec(L.v2, dyn.BT + i * m * n, n, m, n, L.Qu, L.v2, L.part);
        for (int e = threadIdx.x; e < n; e += blockDim.x) { L.v1[e] = L.v2[e]; x[(size_t)(k + 1) * n + e] = L.v2[e]; }
        __syncthreads();
    }
}

// s_0 = max(0, ||xs (x0 - xbar_0)||_inf - delta)   (every thread returns it)
__device__ inline double slack0(const QPDims &d, const QPConst &c, const QPData &q, QPLds &L) {
    if (!d.tr) return 0.0;
    double v = 0.0;
    for (int e = threadIdx.x; e < d.n; e += blockDim.x) v = fmax(v, fabs(c.xs[e] * (q.x0[e] - q.xk[e])));
    v = wg::reduce(v, 1, L.red);
    return fmax(0.0, v - q.delta);
}

// objective value (without the 1/2, as cvxpy reports): locp.py:218-263
__device__ inline double objective(const QPDims &d, const QPConst &c, const QPData &q, const double *x,
                                   const double *u, const double *s, QPLds &L) {
    double acc = 0.0;
    const int n = d.n, nz = d.nz, m = d.m;
    for (int k = threadIdx.x; k <= d.N; k += blockDim.x) {
        double e[16];
        for (int a = 0; a < nz; ++a) {
            double v = q.z ? -q.z[(size_t)k * nz + a] : 0.0;
            for (int j = 0; j < n; ++j) v = fma(c.H[a * n + j], x[(size_t)k * n + j], v);
            e[a] = v;
        }
        for (int a = 0; a < nz; ++a)
            for (int b = 0; b < nz; ++b) acc = fma(e[a] * c.Qz[a * nz + b], e[b], acc);
        if (k == d.N && c.Qzf) {
            for (int a = 0; a < nz; ++a) e[a] += (q.z ? q.z[(size_t)k * nz + a] : 0.0) - (q.zf ? q.zf[a] : 0.0);
            for (int a = 0; a < nz; ++a)
                for (int b = 0; b < nz; ++b) acc = fma(e[a] * c.Qzf[a * nz + b], e[b], acc);
        }
        if (k < d.N) {
            double ue[16];
            for (int a = 0; a < m; ++a) ue[a] = u[(size_t)k * m + a] - (q.ud ? q.ud[(size_t)k * m + a] : 0.0);
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b) acc = fma(ue[a] * c.R[a * m + b], ue[b], acc);
        }
        if (d.tr) acc += q.omega * s[k];
    }
    return wg::reduce(acc, 0, L.red);
}

// ------------------------------------------------------------------ stage pre-pass
// From the row weights D and gradient shifts rho (and, for the dual residual, the multipliers lam)
// build per-stage Hessian / gradient pieces with the slack s_k eliminated.
__device__ inline void stage_prepass(const QPDims &d, const QPConst &c, const QPData &q, QPWork &w, bool with_dual) {
    const int n = d.n, nz = d.nz, m = d.m, N = d.N;
    // e_k = H x_k - z_k
    for (int e = threadIdx.x; e < (N + 1) * nz; e += blockDim.x) {
        const int k = e / nz, a = e - k * nz;
        double v = q.z ? -q.z[e] : 0.0;
        for (int j = 0; j < n; ++j) v = fma(c.H[a * n + j], w.x[(size_t)k * n + j], v);
        w.ez[e] = v;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int k = 1 + wave; k <= N; k += nw) {
        const double *Dk = w.D + (size_t)(k - 1) * d.RX, *rk = w.rho + (size_t)(k - 1) * d.RX;
        const double *lk = w.lam + (size_t)(k - 1) * d.RX;
        const int nxrows = (k == N) ? d.nX + d.nXf : d.nX;
        const int xoff = d.tr ? 2 * n + 1 : 0;
        double sumD = 0.0, sumR = 0.0, sumL = 0.0;
        for (int i = lane; i < n; i += 64) {
            double g = 0.0;
            for (int a = 0; a < nz; ++a) g = fma(c.HtQz2[i * nz + a], w.ez[(size_t)k * nz + a], g);
            if (k == N && c.Qzf) {
                for (int a = 0; a < nz; ++a) {
                    const double ef = w.ez[(size_t)k * nz + a] + (q.z ? q.z[(size_t)k * nz + a] : 0.0) - (q.zf ? q.zf[a] : 0.0);
                    g = fma(c.HtQzf2[i * nz + a], ef, g);
                }
            }
            double gd = g, hd = 0.0, cc = 0.0;
            if (d.tr) {
                const double dp = Dk[i], dm = Dk[n + i], xs = c.xs[i];
                hd = xs * xs * (dp + dm);      // finalised below once Hss is known
                cc = -xs * (dp - dm);
                g += xs * (rk[i] - rk[n + i]);
                gd += xs * (lk[i] - lk[n + i]);
                sumD += dp + dm;
                sumR += rk[i] + rk[n + i];
                sumL += lk[i] + lk[n + i];
            }
            for (int r = 0; r < nxrows; ++r) {
                const double a = (r < d.nX) ? c.XA[(size_t)r * n + i] : c.XfA[(size_t)(r - d.nX) * n + i];
                g = fma(a, rk[xoff + r], g);
                gd = fma(a, lk[xoff + r], gd);
            }
            w.hd[(size_t)k * n + i] = hd;
            w.cv[(size_t)k * n + i] = cc;
            w.gx[(size_t)k * n + i] = g;
            w.gxd[(size_t)k * n + i] = gd;
        }
        if (d.tr) {
            sumD = wg::wave_sum(sumD) + Dk[2 * n];
            sumR = wg::wave_sum(sumR) + rk[2 * n];
            sumL = wg::wave_sum(sumL) + lk[2 * n];
            const double Hss = sumD, gs = q.omega - sumR;
            for (int i = lane; i < n; i += 64) {
                w.gx[(size_t)k * n + i] -= w.cv[(size_t)k * n + i] * gs / Hss;
                // diagonal of diag(hd) - c c^T / Hss without cancellation:
                //   (D+ + D-) - (D+ - D-)^2 / Hss = [(D+ + D-) (Hss - D+ - D-) + 4 D+ D-] / Hss
                const double dp = Dk[i], dm = Dk[n + i], xs = c.xs[i];
                w.hd[(size_t)k * n + i] = xs * xs * ((dp + dm) * (Hss - (dp + dm)) + 4.0 * dp * dm) / Hss;
            }
            if (lane == 0) {
                w.Hss[k] = Hss;
                w.gs[k] = gs;
                // stationarity residual of s_k with the true multipliers, parked in gud slot via Hss? keep in ds
                w.ds[k] = q.omega - sumL;
            }
        }
    }
    // input stages
    const double *Du = w.D + (size_t)N * d.RX, *ru = w.rho + (size_t)N * d.RX, *lu = w.lam + (size_t)N * d.RX;
    for (int e = threadIdx.x; e < N * m * m; e += blockDim.x) {
        const int k = e / (m * m), ab = e - k * m * m, a = ab / m, b = ab - a * m;
        double v = c.R2[ab];
        for (int r = 0; r < d.nU; ++r) v = fma(c.UA[r * m + a] * Du[(size_t)k * d.nU + r], c.UA[r * m + b], v);
        w.Huu[e] = v;
    }
    for (int e = threadIdx.x; e < N * m; e += blockDim.x) {
        const int k = e / m, a = e - k * m;
        double v = 0.0;
        for (int b = 0; b < m; ++b) v = fma(c.R2[a * m + b], w.u[(size_t)k * m + b] - (q.ud ? q.ud[(size_t)k * m + b] : 0.0), v);
        double g = v, gd = v;
        for (int r = 0; r < d.nU; ++r) {
            g = fma(c.UA[r * m + a], ru[(size_t)k * d.nU + r], g);
            gd = fma(c.UA[r * m + a], lu[(size_t)k * d.nU + r], gd);
        }
        w.gu[e] = g;
        w.gud[e] = gd;
    }
    __syncthreads();
}

// ------------------------------------------------------------------ Riccati solve of one Newton system
// full = factorise (stores K_k, Quu_k^-1) and solve; !full = re-solve with new gradients only.
// Returns false on a non-positive-definite Quu.  rd_out: max |reduced dual residual| (with_dual).
__device__ inline bool riccati_solve(const QPDims &d, const QPConst &c, const QPDyn &dyn, QPWork &w, QPLds &L,
                                     bool full, bool with_dual, double *rd_out) {
    const int n = d.n, m = d.m, N = d.N, ld = d.ld, mp = d.mp;
    const int tid = threadIdx.x, nt = blockDim.x;
    double rd = 0.0;
    // ---- terminal stage
    {
        const int k = N;
        if (full) {
            const double Hss = d.tr ? w.Hss[k] : 1.0;
            const int nxrows = d.nX + d.nXf, xoff = d.tr ? 2 * n + 1 : 0;
            const double *Dk = w.D + (size_t)(k - 1) * d.RX + xoff;
            for (int e = tid; e < n * n; e += nt) {
                const int i = e / n, j = e - i * n;
                double v = c.QxN[e];
                if (d.tr) {
                    if (i == j) v += w.hd[(size_t)k * n + i];
                    else v -= w.cv[(size_t)k * n + i] * w.cv[(size_t)k * n + j] / Hss;
                }
                for (int r = 0; r < nxrows; ++r) {
                    const double *row = (r < d.nX) ? c.XA + (size_t)r * n : c.XfA + (size_t)(r - d.nX) * n;
                    v = fma(row[i] * Dk[r], row[j], v);
                }
                L.P[i * ld + j] = v;
            }
        }
        for (int e = tid; e < n; e += nt) { L.pv[e] = w.gx[(size_t)k * n + e]; L.adj[e] = w.gxd[(size_t)k * n + e]; }
        __syncthreads();
    }
    for (int k = N - 1; k >= 0; --k) {
        const size_t sel = dyn.sel(k);
        const double *Ag = dyn.A + sel * n * n, *Bg = dyn.B + sel * n * m;
        // stage A, B into LDS (A only needed for the matrix part and the transposed mat-vecs)
        for (int e = tid; e < n * n; e += nt) { const int i = e / n, j = e - i * n; L.A[i * ld + j] = Ag[e]; }
        for (int e = tid; e < n * mp; e += nt) { const int i = e / mp, j = e - i * mp; L.B[e] = j < m ? Bg[i * m + j] : 0.0; }
        if (full) {
            for (int e = tid; e < m * m; e += nt) L.Quu[e] = w.Huu[(size_t)k * m * m + e];
        } else {
            for (int e = tid; e < m * n; e += nt) { const int a = e / n, j = e - a * n; L.Km[a * ld + j] = w.K[(size_t)k * m * n + e]; }
            for (int e = tid; e < m * m; e += nt) L.Qinv[e] = w.Qinv[(size_t)k * m * m + e];
        }
        __syncthreads();
        if (full) {
            wg::gemm<false>(L.W, ld, L.P, ld, L.A, ld, n, n, n);     // W = P A
            wg::gemm<false>(L.G, mp, L.P, ld, L.B, mp, n, m, n);     // G = P B
            // Quu += B^T G ; Qux = B^T W
            for (int e = tid; e < m * m + m * n; e += nt) {
                if (e < m * m) {
                    const int a = e / m, b = e - a * m;
                    double v = L.Quu[e];
                    for (int i = 0; i < n; ++i) v = fma(L.B[i * mp + a], L.G[i * mp + b], v);
                    L.Quu[e] = v;
                } else {
                    const int f = e - m * m, a = f / n, j = f - a * n;
                    double v = 0.0;
                    for (int i = 0; i < n; ++i) v = fma(L.B[i * mp + a], L.W[i * ld + j], v);
                    L.Qux[a * ld + j] = v;
                }
            }
        }
        // Qu = gu + B^T pv ; dual residual wrt u_k = gud + B^T adj
        for (int e = tid; e < 2 * m; e += nt) {
            const int a = e % m;
            const double *vec = e < m ? L.pv : L.adj;
            double v = e < m ? w.gu[(size_t)k * m + a] : w.gud[(size_t)k * m + a];
            for (int i = 0; i < n; ++i) v = fma(L.B[i * mp + a], vec[i], v);
            if (e < m) L.Qu[a] = v; else L.rdu[a] = v;
        }
        __syncthreads();
        if (with_dual && tid < m) rd = fmax(rd, fabs(L.rdu[tid]));
        if (full) {
            // Quu = L L^T (kept in L.Qinv); K = -Quu^-1 Qux by triangular solves, one state column per
            // thread (backward stable -- an explicit inverse loses the weakly curved input directions
            // once the active-bound weights reach 1e12)
            if (!wg::chol_factor(L.Quu, L.Qinv, m, L.flag, true)) return false;
            for (int j = tid; j < n; j += nt) {
                wg::chol_solve_neg(L.Qinv, m, L.Qux + j, ld, L.Km + j, ld);
                for (int a = 0; a < m; ++a) w.K[((size_t)k * m + a) * n + j] = L.Km[a * ld + j];
            }
            for (int e = tid; e < m * m; e += nt) w.Qinv[(size_t)k * m * m + e] = L.Qinv[e];
        }
        if (tid == 0) {
            wg::chol_solve_neg(L.Qinv, m, L.Qu, 1, L.kf, 1);
            for (int a = 0; a < m; ++a) w.kff[(size_t)k * m + a] = L.kf[a];
        }
        __syncthreads();
        if (k >= 1) {
            if (full) {
                wg::gemm<true>(L.P, ld, L.A, ld, L.W, ld, n, n, n);   // T = A^T W  (P no longer needed)
                // P_k = sym(T) + Qx + diag(hd) - c c^T/Hss + X^T D X + sym(Qux^T K)
                const double Hss = d.tr ? w.Hss[k] : 1.0;
                const int xoff = d.tr ? 2 * n + 1 : 0;
                const double *Dk = w.D + (size_t)(k - 1) * d.RX + xoff;
                for (int e = tid; e < n * n; e += nt) {
                    const int i = e / n, j = e - i * n;
                    if (j < i) continue;
                    double v = 0.5 * (L.P[i * ld + j] + L.P[j * ld + i]) + c.Qx[e];
                    double kk = 0.0;
                    for (int a = 0; a < m; ++a) kk += L.Qux[a * ld + i] * L.Km[a * ld + j] + L.Qux[a * ld + j] * L.Km[a * ld + i];
                    v += 0.5 * kk;
                    if (d.tr) {
                        if (i == j) v += w.hd[(size_t)k * n + i];
                        else v -= w.cv[(size_t)k * n + i] * w.cv[(size_t)k * n + j] / Hss;
                    }
                    for (int r = 0; r < d.nX; ++r) v = fma(c.XA[(size_t)r * n + i] * Dk[r], c.XA[(size_t)r * n + j], v);
                    L.W[i * ld + j] = v;       // W is free: use it as the output buffer (no read/write race)
                    L.W[j * ld + i] = v;
                }
            }
            // pv_new = gx + A^T pv + K^T Qu ; adj_new = gxd + A^T adj
            for (int e = tid; e < 2 * n; e += nt) {
                const int j = e % n;
                const bool first = e < n;
                const double *vec = first ? L.pv : L.adj;
                double v = first ? w.gx[(size_t)k * n + j] : w.gxd[(size_t)k * n + j];
                for (int i = 0; i < n; ++i) v = fma(L.A[i * ld + j], vec[i], v);
                if (first) {
                    for (int a = 0; a < m; ++a) v = fma(L.Km[a * ld + j], L.Qu[a], v);
                    L.v1[j] = v;
                } else {
                    L.v2[j] = v;
                }
            }
            __syncthreads();
            if (full) {
                for (int e = tid; e < n * n; e += nt) { const int i = e / n, j = e - i * n; L.P[i * ld + j] = L.W[i * ld + j]; }
            }
            for (int e = tid; e < n; e += nt) { L.pv[e] = L.v1[e]; L.adj[e] = L.v2[e]; }
            __syncthreads();
        }
    }
    if (with_dual) {
        // stationarity of the slacks s_k (parked in ds by the pre-pass)
        if (d.tr) for (int k = 1 + tid; k <= N; k += nt) rd = fmax(rd, fabs(w.ds[k]));
        rd = wg::reduce(rd, 1, L.red);
        if (rd_out) *rd_out = rd;
    }
    // ---- forward sweep: dx_0 = 0
    for (int e = tid; e < n; e += nt) { L.v1[e] = 0.0; w.dx[e] = 0.0; }
    if (tid == 0) w.ds[0] = 0.0;
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    for (int k = 0; k < N; ++k) {
        const size_t sel = dyn.sel(k);
        // du = K dx + kff  (one wave per output row, lanes over the state)
        for (int a = wave; a < m; a += nw) {
            double v = 0.0;
            for (int j = lane; j < n; j += 64) v = fma(w.K[((size_t)k * m + a) * n + j], L.v1[j], v);
            v = wg::wave_sum(v);
            if (lane == 0) { v += w.kff[(size_t)k * m + a]; L.kf[a] = v; w.du[(size_t)k * m + a] = v; }
        }
        __syncthreads();
        wg::matTvec(L.v2, dyn.AT + sel * n * n, n, n, n, L.v1, nullptr, L.part);
        wg::matTvec(L.v2, dyn.BT + sel * m * n, n, m, n, L.kf, L.v2, L.part);
        for (int e = tid; e < n; e += nt) { L.v1[e] = L.v2[e]; w.dx[(size_t)(k + 1) * n + e] = L.v2[e]; }
        if (d.tr && wave == 0) {
            double v = 0.0;
            for (int j = lane; j < n; j += 64) v = fma(w.cv[(size_t)(k + 1) * n + j], L.v2[j], v);
            v = wg::wave_sum(v);
            if (lane == 0) w.ds[k + 1] = -(w.gs[k + 1] + v) / w.Hss[k + 1];
        }
        __syncthreads();
    }
    if (!d.tr) { for (int k = tid; k <= N; k += nt) w.ds[k] = 0.0; __syncthreads(); }
    return true;
}

// largest step keeping t + a dt >= 0 and lam + a dlam >= 0 (not clamped to 1)
__device__ inline double max_step(const QPDims &d, const QPWork &w, QPLds &L) {
    double a = 1e300;
    for_rows(d, [&](int row, bool, int, int) {
        const double dt = w.dt[row], dl = w.dlam[row];
        if (dt < 0.0) a = fmin(a, -w.t[row] / dt);
        if (dl < 0.0) a = fmin(a, -w.lam[row] / dl);
    });
    return wg::reduce(a, 2, L.red);
}

// Solve one QP.  Results in w.x, w.u, w.s.  Returns status: 0 optimal, 1 max iterations, 2 numerical failure.
__device__ inline int solve(const QPDims &d, const QPConst &c, const QPDyn &dyn, const QPData &q, QPWork &w,
                            QPLds &L, double *J_out, int *iters_out) {
    const int N = d.N, n = d.n, m = d.m;
    const int tid = threadIdx.x, nt = blockDim.x;
    const double s0 = slack0(d, c, q, L);
    for (int e = tid; e < N * m; e += nt) w.u[e] = 0.0;
    for (int e = tid; e <= N; e += nt) w.s[e] = (e == 0) ? s0 : 0.0;
    __syncthreads();
    rollout(d, dyn, q, w.u, w.x, L);
    int status = 1, it = 0;
    if (d.ng == 0) {
        for (int e = tid; e < d.NR; e += nt) { w.D[e] = 0.0; w.rho[e] = 0.0; w.lam[e] = 0.0; }
        __syncthreads();
        stage_prepass(d, c, q, w, false);
        if (!riccati_solve(d, c, dyn, w, L, true, false, nullptr)) status = 2;
        else {
            for (int e = tid; e < N * m; e += nt) w.u[e] += w.du[e];
            __syncthreads();
            status = 0;
        }
    } else {
        // ---- starting point: unit weights, rho = a.w - h  (least-squares point), then shift
        rows_apply(d, c, w.x, w.s, w.u, w.rg);
        __syncthreads();
        for_rows(d, [&](int row, bool isU, int k, int r) {
            const double g = w.rg[row] - row_h(d, c, q, isU, k, r);
            w.D[row] = 1.0; w.rho[row] = g; w.lam[row] = 0.0;
        });
        __syncthreads();
        stage_prepass(d, c, q, w, false);
        bool ok = riccati_solve(d, c, dyn, w, L, true, false, nullptr);
        if (!ok) status = 2;
        if (ok) {
            for (int e = tid; e < (N + 1) * n; e += nt) w.x[e] += w.dx[e];
            for (int e = tid; e < N * m; e += nt) w.u[e] += w.du[e];
            for (int e = tid; e <= N; e += nt) w.s[e] = (e == 0) ? s0 : w.s[e] + w.ds[e];
            __syncthreads();
            rows_apply(d, c, w.x, w.s, w.u, w.rg);
            __syncthreads();
            double zmin = INFINITY, zmax = -INFINITY;
            for_rows(d, [&](int row, bool isU, int k, int r) {
                const double g = w.rg[row] - row_h(d, c, q, isU, k, r);
                w.rg[row] = g;
                zmin = fmin(zmin, g); zmax = fmax(zmax, g);
            });
            zmin = wg::reduce(zmin, 2, L.red);
            zmax = wg::reduce(zmax, 1, L.red);
            const double sh_t = zmax >= 0.0 ? 1.0 + zmax : 0.0, sh_l = zmin <= 0.0 ? 1.0 - zmin : 0.0;
            for_rows(d, [&](int row, bool, int, int) {
                const double g = w.rg[row];
                w.t[row] = -g + sh_t; w.lam[row] = g + sh_l;
            });
            __syncthreads();
            // scales for the stopping test (as oracle/riccati_ipm.py)
            double sd = 1.0, sp = 1.0;
            for (int e = tid; e < n; e += nt) {
                double g = 0.0;
                if (q.z) for (int a = 0; a < d.nz; ++a) g = fma(c.HtQz2[e * d.nz + a], -q.z[d.nz + a], g);
                sd = fmax(sd, fabs(g));
            }
            for (int e = tid; e < d.nU; e += nt) sp = fmax(sp, fabs(c.Ub[e]));
            sd = fmax(wg::reduce(sd, 1, L.red), q.omega);
            sp = fmax(wg::reduce(sp, 1, L.red), fabs(q.delta));
            const double dreg = d.reg / sd;
            bool near_opt = false;
            for (it = 0; it < d.max_iter; ++it) {
                // residuals, weights, predictor shifts
                rows_apply(d, c, w.x, w.s, w.u, w.rg);
                __syncthreads();
                double musum = 0.0, rp = 0.0;
                for_rows(d, [&](int row, bool isU, int k, int r) {
                    const double g = w.rg[row] - row_h(d, c, q, isU, k, r);
                    const double t = w.t[row], lam = w.lam[row];
                    const double rg = g + t;
                    w.rg[row] = rg;
                    const double D = lam / (t + dreg * lam);     // regularised weight (see oracle/riccati_ipm.py)
                    w.D[row] = D;
                    w.rho[row] = D * (rg + dreg * lam);
                    musum += lam * t;
                    rp = fmax(rp, fabs(rg));
                });
                const double mu = wg::reduce(musum, 0, L.red) / d.ng;
                rp = wg::reduce(rp, 1, L.red);
                stage_prepass(d, c, q, w, true);
                double rd = 0.0;
                // a factorisation that breaks down in the last digits of an already converged iterate
                // (weights D = lam/t up to 1e13) is accepted at the looser 1e-8 certificate
                if (!riccati_solve(d, c, dyn, w, L, true, true, &rd)) { status = near_opt ? 0 : 2; break; }
                if (!(mu == mu)) { status = near_opt ? 0 : 5; break; }
                if (!(rd == rd)) { status = near_opt ? 0 : 6; break; }
                if (q.dbg && threadIdx.x == 0) { double *g = q.dbg + 8 * it; g[0] = mu; g[1] = rd; g[2] = rp; g[3] = sd; g[4] = sp; }
                const double ltol = fmax(d.tol, 1e-9);     // linear residuals: round-off floor (see the port)
                if (rd <= ltol * sd && rp <= ltol * sp && mu <= d.tol) { status = 0; break; }
                near_opt = (rd <= 1e-8 * sd && rp <= 1e-8 * sp && mu <= 1e-8);
                // predictor direction on the rows
                rows_apply(d, c, w.dx, w.ds, w.du, w.dt);
                __syncthreads();
                for_rows(d, [&](int row, bool, int, int) {
                    const double t = w.t[row], lam = w.lam[row], rga = w.rg[row] + w.dt[row];
                    const double dl = (-lam * t + lam * rga) / (t + dreg * lam);
                    w.dlam[row] = dl;
                    w.dt[row] = -rga + dreg * dl;
                });
                __syncthreads();
                const double a_aff = fmin(1.0, max_step(d, w, L));
                double ma = 0.0;
                for_rows(d, [&](int row, bool, int, int) {
                    ma += (w.lam[row] + a_aff * w.dlam[row]) * (w.t[row] + a_aff * w.dt[row]);
                });
                const double mu_aff = wg::reduce(ma, 0, L.red) / d.ng;
                const double sig = mu > 0.0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
                for_rows(d, [&](int row, bool, int, int) {
                    const double t = w.t[row], lam = w.lam[row];
                    const double rc = lam * t + w.dt[row] * w.dlam[row] - sig * mu;
                    w.rc[row] = rc;
                    w.rho[row] = lam + (lam * w.rg[row] - rc) / (t + dreg * lam);
                });
                __syncthreads();
                stage_prepass(d, c, q, w, false);
                riccati_solve(d, c, dyn, w, L, false, false, nullptr);
                rows_apply(d, c, w.dx, w.ds, w.du, w.dt);
                __syncthreads();
                for_rows(d, [&](int row, bool, int, int) {
                    const double t = w.t[row], lam = w.lam[row], rga = w.rg[row] + w.dt[row];
                    const double dl = (-w.rc[row] + lam * rga) / (t + dreg * lam);
                    w.dlam[row] = dl;
                    w.dt[row] = -rga + dreg * dl;
                });
                __syncthreads();
                const double a = fmin(1.0, 0.99 * max_step(d, w, L));   // stay strictly interior
                if (q.dbg && threadIdx.x == 0) { double *g = q.dbg + 8 * it; g[5] = a_aff; g[6] = sig; g[7] = a; }
                for (int e = tid; e < (N + 1) * n; e += nt) w.x[e] += a * w.dx[e];
                for (int e = tid; e < N * m; e += nt) w.u[e] += a * w.du[e];
                for (int e = tid; e <= N; e += nt) w.s[e] = (e == 0) ? s0 : w.s[e] + a * w.ds[e];
                for_rows(d, [&](int row, bool, int, int) {
                    w.t[row] += a * w.dt[row];
                    w.lam[row] += a * w.dlam[row];
                });
                __syncthreads();
            }
        }
    }
    // final consistency: x is exactly the rollout of u
    rollout(d, dyn, q, w.u, w.x, L);
    const double J = objective(d, c, q, w.x, w.u, w.s, L);
    if (J_out) *J_out = J;
    if (iters_out) *iters_out = it;
    return status;
}

}  // namespace qp
