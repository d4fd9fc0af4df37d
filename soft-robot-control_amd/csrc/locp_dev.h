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
    int ld;    // leading dimension of the LDS matrices = NPa = roundup16(n + m)
    int mp;    // unused (kept for layout stability)
    int NK;    // roundup4(n): K extent of the MFMA products (zero padded rows)
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
    double *P;                         // ld x ld : cost-to-go P_{k+1}, then the stage Gram matrix M = [A|B]^T P [A|B]
    double *AB;                        // roundup16(n) x ld : [A_k | B_k], zero padded
    double *W;                         // roundup16(n) x ld : P [A_k | B_k]
    double *Km;                        // m x ld  : feedback gain of the stage
    double *Quu, *Lc;                  // 16 x 16 : Quu and its Cholesky factor
    double *pv, *adj, *v1, *v2, *hdv, *cvv;   // ld each
    double *ypv, *yadj;                // ld each : [A|B]^T pv, [A|B]^T adj
    double *Qu, *kf, *rdu;             // 16 each
    double *Hm;                        // nz x ld : H
    double *HtQ;                       // ld x 16 : 2 H^T Qz
    double *XAl;                       // (nX + nXf) x ld
    double *Dx;                        // 32      : X-row weights of the stage
    double *part;                      // blockDim
    double *red;                       // 16
    int *flag;                         // 4 ints
};

__host__ __device__ inline size_t qp_lds_bytes(const QPDims &d, int nthreads) {
    const size_t nk16 = (size_t)((d.n + 15) & ~15);   // W / AB rows: whole MFMA tiles are stored
    size_t c = (size_t)d.ld * d.ld + 2 * nk16 * d.ld + (size_t)d.m * d.ld + 2 * 256 + 8 * (size_t)d.ld + 3 * 16 +
               (size_t)d.nz * d.ld + (size_t)d.ld * 16 + (size_t)(d.nX + d.nXf) * d.ld + 32 + nthreads + 16 + 4;
    return c * sizeof(double);
}

__device__ inline void qp_lds_carve(QPLds &L, double *base, const QPDims &d, int nthreads) {
    double *p = base;
    auto take = [&](size_t c) { double *q = p; p += c; return q; };
    const size_t nk16 = (size_t)((d.n + 15) & ~15);
    L.P = take((size_t)d.ld * d.ld); L.AB = take(nk16 * d.ld); L.W = take(nk16 * d.ld);
    L.Km = take((size_t)d.m * d.ld);
    L.Quu = take(256); L.Lc = take(256);
    L.pv = take(d.ld); L.adj = take(d.ld); L.v1 = take(d.ld); L.v2 = take(d.ld); L.hdv = take(d.ld); L.cvv = take(d.ld);
    L.ypv = take(d.ld); L.yadj = take(d.ld);
    L.Qu = take(16); L.kf = take(16); L.rdu = take(16);
    L.Hm = take((size_t)d.nz * d.ld); L.HtQ = take((size_t)d.ld * 16);
    L.XAl = take((size_t)(d.nX + d.nXf) * d.ld);
    L.Dx = take(32);
    L.part = take(nthreads);
    L.red = take(16);
    L.flag = reinterpret_cast<int *>(take(4));
}

// one-off per kernel: constants into LDS, zero the padding of the MFMA operands
__device__ inline void qp_lds_init(QPLds &L, const QPDims &d, const QPConst &c) {
    const int tid = threadIdx.x, nt = blockDim.x, n = d.n, ld = d.ld;
    for (int e = tid; e < d.ld * d.ld; e += nt) L.P[e] = 0.0;
    for (int e = tid; e < ((d.n + 15) & ~15) * d.ld; e += nt) { L.AB[e] = 0.0; L.W[e] = 0.0; }
    for (int e = tid; e < d.nz * n; e += nt) { const int a = e / n, j = e - a * n; L.Hm[a * ld + j] = c.H[e]; }
    for (int e = tid; e < n * d.nz; e += nt) { const int i = e / d.nz, a = e - i * d.nz; L.HtQ[i * 16 + a] = c.HtQz2[e]; }
    for (int e = tid; e < (d.nX + d.nXf) * n; e += nt) {
        const int r = e / n, j = e - r * n;
        L.XAl[r * ld + j] = r < d.nX ? c.XA[(size_t)r * n + j] : c.XfA[(size_t)(r - d.nX) * n + j];
    }
    __syncthreads();
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

// ------------------------------------------------------------------ f64 MFMA product on LDS operands
typedef double qp_d4 __attribute__((ext_vector_type(4)));

// C[i][j] = sum_{k<K} Lm[k][i] * Rm[k][j]  for i < 16*MT, j < 16*NTl.  Lm, Rm: k-major rows (K x ld) in
// LDS, K a multiple of 4 (zero padded).  Rows i >= vrows of C are stored as exact zeros.
// v_mfma_f64_16x16x4: A lane l holds Lm^T[i=l&15][k=l>>4], B lane holds Rm[k=l>>4][j=l&15];
// D reg q of lane l is C[row = (l>>4) + 4q][col = l&15].
__device__ inline void mfma_atb(double *C, const double *Lm, const double *Rm, int K, int MT, int NTl, int ld,
                                int vrows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    const int ntiles = MT * NTl;
    for (int t0 = wave; t0 < ntiles; t0 += 2 * nw) {
        const int t1 = t0 + nw;
        const bool has1 = t1 < ntiles;
        const int ti0 = t0 / NTl, tj0 = t0 - ti0 * NTl;
        const int ti1 = has1 ? t1 / NTl : ti0, tj1 = has1 ? t1 - ti1 * NTl : tj0;
        const double *la0 = Lm + kk * ld + 16 * ti0 + l16, *rb0 = Rm + kk * ld + 16 * tj0 + l16;
        const double *la1 = Lm + kk * ld + 16 * ti1 + l16, *rb1 = Rm + kk * ld + 16 * tj1 + l16;
        qp_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < K; k0 += 4) {
            const double a0 = la0[k0 * ld], b0 = rb0[k0 * ld];
            const double a1 = la1[k0 * ld], b1 = rb1[k0 * ld];
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r0 = 16 * ti0 + kk + 4 * q;
            C[r0 * ld + 16 * tj0 + l16] = r0 < vrows ? acc0[q] : 0.0;
            if (has1) {
                const int r1 = 16 * ti1 + kk + 4 * q;
                C[r1 * ld + 16 * tj1 + l16] = r1 < vrows ? acc1[q] : 0.0;
            }
        }
    }
    __syncthreads();
}

// stage Hessian of x_k added to the (i,j) entry: Qx (+ terminal) + slack-eliminated trust region + X rows
__device__ __forceinline__ double stage_hess(const QPDims &d, const QPConst &c, const QPLds &L, int k, int i, int j,
                                             double Hss, int nxrows) {
    double v = 0.0;
    for (int a = 0; a < d.nz; ++a) v = fma(L.HtQ[i * 16 + a], L.Hm[a * d.ld + j], v);     // 2 H^T Qz H
    if (k == d.N && c.Qzf) v += c.QxN[(size_t)i * d.n + j] - c.Qx[(size_t)i * d.n + j];
    if (d.tr) {
        if (i == j) v += L.hdv[i];
        else v -= L.cvv[i] * L.cvv[j] / Hss;
    }
    for (int r = 0; r < nxrows; ++r) v = fma(L.XAl[r * d.ld + i] * L.Dx[r], L.XAl[r * d.ld + j], v);
    return v;
}

// ------------------------------------------------------------------ Riccati solve of one Newton system
// full = factorise (stores K_k and the Cholesky factor of Quu_k) and solve; !full = re-solve with new
// gradients only.  Returns false on a non-positive-definite Quu.  rd_out: max |reduced dual residual|.
//
// Per stage (full): with AB = [A_k | B_k] (n x (n+m)) the two f64-MFMA products
//     W = P_{k+1} AB            (n x (n+m))
//     M = AB^T W = [[A^T P A, A^T P B], [B^T P A, B^T P B]]
// give Qxx, Qux and Quu in one Gram matrix; P_k = sym(M_xx) + H_k + sym(Qux^T K) with K = -Quu^-1 Qux by
// Cholesky solves.  P, AB, W stay in LDS for the whole horizon; A_k, B_k stream from the (L2 resident)
// TPWL tables.
__device__ inline bool riccati_solve(const QPDims &d, const QPConst &c, const QPDyn &dyn, QPWork &w, QPLds &L,
                                     bool full, bool with_dual, double *rd_out) {
    const int n = d.n, m = d.m, N = d.N, ld = d.ld, NK = d.NK;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int xoff = d.tr ? 2 * n + 1 : 0;
    double rd = 0.0;
    // ---- terminal stage
    {
        const int k = N;
        const int nxrows = d.nX + d.nXf;
        for (int e = tid; e < n; e += nt) {
            L.pv[e] = w.gx[(size_t)k * n + e];
            L.adj[e] = w.gxd[(size_t)k * n + e];
            L.hdv[e] = w.hd[(size_t)k * n + e];
            L.cvv[e] = w.cv[(size_t)k * n + e];
        }
        if (tid < nxrows) L.Dx[tid] = w.D[(size_t)(k - 1) * d.RX + xoff + tid];
        __syncthreads();
        if (full) {
            const double Hss = d.tr ? w.Hss[k] : 1.0;
            for (int e = tid; e < n * n; e += nt) {
                const int i = e / n, j = e - i * n;
                if (j < i) continue;
                const double v = stage_hess(d, c, L, k, i, j, Hss, nxrows);
                L.P[i * ld + j] = v;
                L.P[j * ld + i] = v;
            }
            for (int e = tid; e < (NK - n) * ld; e += nt) L.P[n * ld + e] = 0.0;
        }
        __syncthreads();
    }
    for (int k = N - 1; k >= 0; --k) {
        const size_t sel = dyn.sel(k);
        const double *Ag = dyn.A + sel * n * n, *Bg = dyn.B + sel * n * m;
        if (full) {
            // stage [A | B] into LDS (pad rows / columns stay zero)
            for (int e = tid; e < n * n; e += nt) { const int i = e / n, j = e - i * n; L.AB[i * ld + j] = Ag[e]; }
            for (int e = tid; e < n * m; e += nt) { const int i = e / m, j = e - i * m; L.AB[i * ld + n + j] = Bg[e]; }
            for (int e = tid; e < m * m; e += nt) L.Quu[e] = w.Huu[(size_t)k * m * m + e];
            if (k >= 1) {
                for (int e = tid; e < n; e += nt) { L.hdv[e] = w.hd[(size_t)k * n + e]; L.cvv[e] = w.cv[(size_t)k * n + e]; }
                if (tid < d.nX) L.Dx[tid] = w.D[(size_t)(k - 1) * d.RX + xoff + tid];
            }
            __syncthreads();
            mfma_atb(L.W, L.P, L.AB, NK, (n + 15) >> 4, ld >> 4, ld, n);         // W = P [A|B]
            mfma_atb(L.P, L.AB, L.W, NK, ld >> 4, ld >> 4, ld, ld);              // M = [A|B]^T W
            // [A|B]^T pv and [A|B]^T adj : (n+m) outputs each, 4 partial sums per output
            {
                const int nout = 2 * (n + m), S = nt / (2 * ld) > 0 ? nt / (2 * ld) : 1;
                const int o = tid % (2 * ld), sl = tid / (2 * ld);
                if (sl < S && o < 2 * ld) {
                    const int col = o % ld;
                    const double *vec = o < ld ? L.pv : L.adj;
                    double acc = 0.0;
                    if (col < n + m) for (int i = sl; i < n; i += S) acc = fma(L.AB[i * ld + col], vec[i], acc);
                    L.part[sl * 2 * ld + o] = acc;
                }
                __syncthreads();
                if (tid < 2 * ld) {
                    double r = 0.0;
                    for (int q = 0; q < S; ++q) r += L.part[q * 2 * ld + tid];
                    if (tid < ld) L.ypv[tid] = r; else L.yadj[tid - ld] = r;
                }
                (void)nout;
            }
            __syncthreads();
            if (tid < m) {
                L.Qu[tid] = w.gu[(size_t)k * m + tid] + L.ypv[n + tid];
                L.rdu[tid] = w.gud[(size_t)k * m + tid] + L.yadj[n + tid];
            }
            for (int e = tid; e < m * m; e += nt) { const int a = e / m, b = e - a * m; L.Quu[e] += L.P[(n + a) * ld + n + b]; }
            __syncthreads();
            if (with_dual && tid < m) rd = fmax(rd, fabs(L.rdu[tid]));
            // Quu = Lc Lc^T; K = -Quu^-1 Qux by triangular solves, one state column per thread (backward
            // stable: an explicit inverse loses the weakly curved input directions once the active-bound
            // weights reach 1e12).  Qux = M[n.., 0..n).
            if (!wg::chol_factor(L.Quu, L.Lc, m, L.flag, true)) return false;
            for (int j = tid; j < n; j += nt) {
                wg::chol_solve_neg(L.Lc, m, L.P + n * ld + j, ld, L.Km + j, ld);
                for (int a = 0; a < m; ++a) w.K[((size_t)k * m + a) * n + j] = L.Km[a * ld + j];
            }
            for (int e = tid; e < m * m; e += nt) w.Qinv[(size_t)k * m * m + e] = L.Lc[e];
            if (tid == 0) {
                wg::chol_solve_neg(L.Lc, m, L.Qu, 1, L.kf, 1);
                for (int a = 0; a < m; ++a) w.kff[(size_t)k * m + a] = L.kf[a];
            }
            __syncthreads();
            if (k >= 1) {
                // P_k (in place over M): each unordered pair (i,j) is owned by one thread
                const double Hss = d.tr ? w.Hss[k] : 1.0;
                for (int e = tid; e < n * n; e += nt) {
                    const int i = e / n, j = e - i * n;
                    if (j < i) continue;
                    double v = 0.5 * (L.P[i * ld + j] + L.P[j * ld + i]) + stage_hess(d, c, L, k, i, j, Hss, d.nX);
                    double kk = 0.0;
                    for (int a = 0; a < m; ++a)
                        kk += L.P[(n + a) * ld + i] * L.Km[a * ld + j] + L.P[(n + a) * ld + j] * L.Km[a * ld + i];
                    v += 0.5 * kk;
                    L.W[i * ld + j] = v;  // staged in W (free after the second product), copied below
                    if (i != j) L.W[j * ld + i] = v;
                }
                // pv_new = gx + A^T pv + K^T Qu ; adj_new = gxd + A^T adj
                for (int e = tid; e < 2 * n; e += nt) {
                    const int j = e % n;
                    if (e < n) {
                        double v = w.gx[(size_t)k * n + j] + L.ypv[j];
                        for (int a = 0; a < m; ++a) v = fma(L.Km[a * ld + j], L.Qu[a], v);
                        L.v1[j] = v;
                    } else {
                        L.v2[j] = w.gxd[(size_t)k * n + j] + L.yadj[j];
                    }
                }
                __syncthreads();
                for (int e = tid; e < n * n; e += nt) { const int i = e / n, j = e - i * n; L.P[i * ld + j] = L.W[i * ld + j]; }
                for (int e = tid; e < (NK - n) * ld; e += nt) L.P[n * ld + e] = 0.0;
                for (int e = tid; e < n; e += nt) { L.pv[e] = L.v1[e]; L.adj[e] = L.v2[e]; }
                __syncthreads();
            }
        } else {
            // vector-only re-solve: A^T pv, B^T pv straight from the L2-resident tables
            for (int e = tid; e < m * n; e += nt) { const int a = e / n, j = e - a * n; L.Km[a * ld + j] = w.K[(size_t)k * m * n + e]; }
            for (int e = tid; e < m * m; e += nt) L.Lc[e] = w.Qinv[(size_t)k * m * m + e];
            wg::matTvec(L.ypv, Ag, n, n, n, L.pv, nullptr, L.part);
            for (int a = wave; a < m; a += nw) {
                double v = 0.0;
                for (int i = lane; i < n; i += 64) v = fma(Bg[i * m + a], L.pv[i], v);
                v = wg::wave_sum(v);
                if (lane == 0) L.Qu[a] = v + w.gu[(size_t)k * m + a];
            }
            __syncthreads();
            if (tid == 0) {
                wg::chol_solve_neg(L.Lc, m, L.Qu, 1, L.kf, 1);
                for (int a = 0; a < m; ++a) w.kff[(size_t)k * m + a] = L.kf[a];
            }
            if (k >= 1) {
                for (int j = tid; j < n; j += nt) {
                    double v = w.gx[(size_t)k * n + j] + L.ypv[j];
                    for (int a = 0; a < m; ++a) v = fma(L.Km[a * ld + j], L.Qu[a], v);
                    L.v1[j] = v;
                }
            }
            __syncthreads();
            if (k >= 1) { for (int e = tid; e < n; e += nt) L.pv[e] = L.v1[e]; }
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
    long long tm[6] = {0, 0, 0, 0, 0, 0};
    long long t_last = wall_clock64();
    auto lap = [&](int slot) { const long long now = wall_clock64(); tm[slot] += now - t_last; t_last = now; };
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
            lap(0);
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
                lap(1);
                stage_prepass(d, c, q, w, true);
                lap(2);
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
                lap(3);
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
                lap(1);
                stage_prepass(d, c, q, w, false);
                lap(2);
                riccati_solve(d, c, dyn, w, L, false, false, nullptr);
                lap(4);
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
    lap(1);
    // final consistency: x is exactly the rollout of u
    rollout(d, dyn, q, w.u, w.x, L);
    const double J = objective(d, c, q, w.x, w.u, w.s, L);
    lap(5);
    if (q.dbg && threadIdx.x == 0) for (int i = 0; i < 6; ++i) q.dbg[8 * 62 + i] = (double)tm[i];
    if (J_out) *J_out = J;
    if (iters_out) *iters_out = it;
    return status;
}

// Trust-region prescreen.  The 2n+1 trust-region rows per stage are 90 % of the inequality rows, yet with
// GuSTO's delta (1e4 initially) they are almost never active.  Solve the QP WITHOUT them first: if the
// minimiser satisfies ||xs (x_k - xbar_k)||_inf <= delta for every k, then (x, u, s = s_0 e_0) is feasible
// for the full QP at the same cost and therefore its minimiser (dropping satisfied constraints cannot
// change an optimum; the slack cost omega * s >= 0 is minimal at 0) -- identical result, 10x fewer rows.
// Otherwise the full QP is solved.
__device__ inline int solve_prescreen(const QPDims &d, const QPConst &c, const QPDyn &dyn, const QPData &q,
                                      double *work_base, QPWork &w, QPLds &L, double *J_out, int *iters_out) {
    if (d.tr) {
        QPDims dn = d;
        dn.tr = 0;
        dn.nrx = d.nX;
        dn.RX = dn.nrx + d.nXf;
        dn.NR = d.N * dn.RX + d.N * d.nU;
        dn.ng = d.N * dn.nrx + d.nXf + d.N * d.nU;
        QPWork wn;
        qp_carve(wn, work_base, dn);
        double J;
        int it;
        const int st = solve(dn, c, dyn, q, wn, L, &J, &it);
        if (st == 0) {
            double md = 0.0;
            for (int e = threadIdx.x + d.n; e < (d.N + 1) * d.n; e += blockDim.x)
                md = fmax(md, fabs(c.xs[e % d.n] * (wn.x[e] - q.xk[e])));
            md = wg::reduce(md, 1, L.red);
            if (md <= q.delta) {
                const double s0 = slack0(d, c, q, L);
                for (int e = threadIdx.x; e <= d.N; e += blockDim.x) w.s[e] = (e == 0) ? s0 : 0.0;
                __syncthreads();
                if (J_out) *J_out = J + q.omega * s0;
                if (iters_out) *iters_out = it;
                return 0;
            }
        }
    }
    return solve(d, c, dyn, q, w, L, J_out, iters_out);
}

}  // namespace qp
