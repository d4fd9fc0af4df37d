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
#include <type_traits>

struct QPDims {
    int N, n, m, nz, nU, nX, nXf, tr;
    int ld;    // leading dimension of the LDS matrices = NPa + 1 (odd: row AND column accesses conflict-free)
    int NPa;   // roundup16(n + m): extent of the stage Gram matrix
    int split; // 1: the W panel holds half of the rows at a time (n_x > 64, see riccati_solve); 0: whole W in LDS
    int WR;    // rows of the W panel (RW, or 48 in split mode)
    int NK;    // roundup4(n): K extent of the MFMA products (zero padded rows)
    int NE4;   // roundup4(m + nX): extra Gram rows (-Y, +Y from the gain; sqrt(D) X rows)
    int RW;    // rows of the AB / W panels
    int nzr;   // rank of Qz folded into the Gram product as constant extra rows (0: Qx added element-wise)
    int RC;    // first constant extra row (>= NK + m + nX and >= roundup16(n))
    int nrx;   // inequality rows owned by x_k, k < N:  tr*(2n+1) + nX
    int RX;    // row stride per x stage: nrx + nXf
    int NR;    // total rows: N*RX + N*nU
    int ng;    // active rows: N*nrx + nXf + N*nU
    int max_iter;
    double tol;
    double reg;   // dual (proximal) regularisation of the Newton systems, relative to the dual scale
    int cond;  // 1: the condensed (output-space) interior point of locp_cond.h serves the QP without trust-region rows
    int po;    // output directions spanning Cq, X.A, Xf.A (rows of C_o)
    int KT;    // 16 x 16 tiles along N * po
    long long qc_off;   // offset (doubles) of the condensed path's HBM block (QCWork) from the problem's work base
    int diagD;          // 1: 2R + U.A^T D U.A is diagonal for every weight vector D (R diagonal, one entry per U.A row)
    int lean;           // 1: the lean condensed kernels (locp_lean.h: packed G resident in LDS) serve this problem
    int lean_j0;        // first stage whose packed G^T rows live in LDS (the stages before it stay in the L2 block)
    int lean_half;      // 1: the HALF-SIZE lean workgroup (round 6): 256 threads, <= 80 KB of LDS, so that two rollouts share a CU -- every packed
                        // row of G in the L2 block (lean_j0 = N), inverses of the diagonal tiles in place of the tiles, Theta^T condensed in
                        // two column passes (ql::sizes / lds_carve / condense; lean.hip: the <.., N, N, ..> instantiations)
    int ls_pd;          // 1 (p_o = 2): the constant output blocks S*_k are positive definite, so every S_k = S*_k + T^T D_x T has an
                        // invertible Cholesky factor Ls_k -- the lean Newton solve then gets dy from the solved system itself
                        // (ql::newton_back) instead of a second product with G
};

namespace qp {
// Kernel instantiations for a fixed n_u (MSEL) and n_x (NSEL): overwrite the runtime copies with the constants (and
// the layout values that follow from them alone, as build_consts computes them) so that everything inlined below
// sees compile-time extents.
template <int MSEL, int NSEL>
__device__ __forceinline__ void specialise(QPDims &d) {
    if constexpr (MSEL > 0) d.m = MSEL;
    if constexpr (NSEL > 0) {
        d.n = NSEL;
        d.NK = (NSEL + 3) & ~3;
        if constexpr (MSEL > 0) { d.NPa = (NSEL + MSEL + 15) & ~15; d.ld = d.NPa + 1; }
    }
}
}  // namespace qp

struct QPConst {                       // shared by the whole batch (HBM/L2 resident)
    cgptr H, Qz, Qzf, R;               // (nz x n), (nz x nz), (nz x nz)|null, (m x m)
    cgptr xs;                          // (n) trust-region scaling
    cgptr UA, Ub, XA, Xb, XfA, Xfb;
    cgptr Qx, QxN;                     // 2 H^T Qz H, (+ 2 H^T Qzf H)      (n x n)
    cgptr HtQz2, HtQzf2;               // 2 H^T Qz, 2 H^T Qzf               (n x nz)
    cgptr R2;                          // 2 R
    cgptr Cq;                          // (nzr x n) with 2 H^T Qz H = Cq^T Cq, or null
    // condensed path (locp_cond.h): output basis C_o (po x n, orthonormal rows) and everything expressed in it
    cgptr Co;                          // (po x n)
    cgptr Sc, ScN;                     // (po x po): 2 H^T Qz H = C_o^T Sc C_o (+ the terminal 2 H^T Qzf H for ScN)
    cgptr Tx, Txf;                     // (nX x po), (nXf x po): X.A = Tx C_o, Xf.A = Txf C_o
    cgptr Cz2, Czf2;                   // (po x nz): C_o H^T 2 Qz, C_o H^T 2 Qzf
    cgiptr gram_sched;                 // lean kernels: tile tasks of the Gram product, ql::GRAM_TASKS x {I, J0, nJ, 0}, longest first (pulled by the waves)
};

struct QPDyn {                         // stage dynamics: matrix k at base + idx[k]*size (idx null: k)
    cgptr A, AT, B, BT, d;
    cgiptr idx;
    __device__ __forceinline__ size_t sel(int k) const { return idx ? (size_t)idx[k] : (size_t)k; }
};

struct QPData {                        // one problem
    cgptr x0, xk, z, zf, ud;               // xk (N+1 x n); z (N+1 x nz)|null; zf (nz)|null; ud (N x m)|null
    double delta, omega;
    gptr dbg;                              // optional per-iteration trace (8 doubles per iteration) or null
};

struct QPWork {                        // per-problem scratch in HBM/L2 (doubles)
    gptr x, u, s, dx, du, ds;
    gptr t, lam, rg, D, rho, rc, dt, dlam;                // NR each
    gptr hd, cv, gx, gxd;                                 // (N+1) x n   (index k = 1..N used)
    gptr Hss, gs;                                         // (N+1)
    gptr Huu, gu, gud;                                    // N x m x m, N x m, N x m
    gptr K, Qinv, kff;                                    // N x m x n, N x m x m (Cholesky factors of Quu), N x m
    gptr ez;                                              // (N+1) x nz
    gptr dump;                                            // 64: target of the stores of lanes that own no result
    gptr tprof;                                        // optional phase timers (debug) or null
};

__host__ __device__ inline size_t qp_work_doubles(const QPDims &d) {
    const size_t N = d.N, n = d.n, m = d.m;
    return (N + 1) * n * 2 + N * m * 2 + (N + 1) * 2 + 8 * (size_t)d.NR + 4 * (N + 1) * n + 2 * (N + 1) +
           N * m * m + 2 * N * m + N * m * n + N * m * m + N * m + (N + 1) * d.nz + 64;
}

__device__ inline void qp_carve(QPWork &w, gptr base, const QPDims &d) {
    const size_t N = d.N, n = d.n, m = d.m, NR = d.NR;
    gptr p = base;
    auto take = [&](size_t c) { gptr q = p; p += c; return q; };
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
    w.dump = take(64);
    w.tprof = (gptr)nullptr;
}

struct QPLds {                         // LDS carve (doubles unless noted)
    lptr P;                         // roundup16(n) x ld : cost-to-go P_{k+1} / P_k
    lptr AB;                        // RW x ld : [A_k | B_k], zero padded; rows NK.. = extra Gram rows (left factor)
    lptr W;                         // RW x ld : P [A_k | B_k]; rows NK.. = extra Gram rows (right factor)
    lptr QUX;                       // 16 x ld : [Qux | B^T P B] = B^T W
    lptr Km;                        // m x ld  : feedback gain of the stage
    lptr Quu, Lc;                  // 16 x 16 : Quu and its Cholesky factor
    lptr pv, adj, v1, v2, v3, hdv, cvv;   // ld each
    lptr ypv, yadj;                // ld each : [A|B]^T pv, [A|B]^T adj
    lptr Qu, kf, rdu;             // 16 each
    lptr XAl;                       // (nX + nXf) x ld
    lptr Dx;                        // 32      : X-row weights of the stage
    lptr sDx;                       // 32      : their square roots (extra Gram rows)
    lptr part;                      // blockDim
    lptr red;                       // 16
    liptr flag;                         // 4 ints
    liptr idxl;                         // N ints: region of every stage (copy of dyn.idx, or 0..N-1)
    int psel;                           // region whose [A | B] is in the panel (register copy, same in every thread)
    lptr base;                          // start of the workgroup's dynamic LDS (the condensed path carves it its own way)
    bool ready;                         // the Riccati layout holds its constants (qp_lds_init ran and nothing clobbered it)
};

__host__ __device__ inline size_t qp_lds_bytes(const QPDims &d, int nthreads) {
    const size_t nk16 = d.split ? (size_t)d.NK : (size_t)((d.n + 15) & ~15);   // whole MFMA tiles are stored (masked in split mode)
    size_t c = nk16 * d.ld + ((size_t)d.RW + d.WR) * d.ld + (size_t)d.m * d.ld + 2 * 256 + 9 * (size_t)d.ld + 3 * 16 +
               (size_t)(d.nX + d.nXf) * d.ld + 64 + nthreads + 16 + 4 + (size_t)(d.N / 2 + 2);
    return c * sizeof(double);
}

__device__ inline void qp_lds_carve(QPLds &L, lptr base, const QPDims &d, int nthreads) {
    lptr p = base;
    L.base = base;
    L.ready = false;
    L.psel = -1;
    auto take = [&](size_t c) { lptr q = p; p += c; return q; };
    const size_t nk16 = d.split ? (size_t)d.NK : (size_t)((d.n + 15) & ~15);
    L.P = take(nk16 * d.ld); L.AB = take((size_t)d.RW * d.ld); L.W = take((size_t)d.WR * d.ld);
    L.QUX = L.AB + (size_t)d.NK * d.ld;     // [Qux | B^T P B] lands in the extra-row slots of the left panel (rows NK..NK+m)
    L.Km = take((size_t)d.m * d.ld);
    L.Quu = take(256); L.Lc = take(256);
    L.pv = take(d.ld); L.adj = take(d.ld); L.v1 = take(d.ld); L.v2 = take(d.ld); L.v3 = take(d.ld); L.hdv = take(d.ld); L.cvv = take(d.ld);
    L.ypv = take(d.ld); L.yadj = take(d.ld);
    L.Qu = take(16); L.kf = take(16); L.rdu = take(16);
    L.XAl = take((size_t)(d.nX + d.nXf) * d.ld);
    L.Dx = take(32);
    L.sDx = take(32);
    L.part = take(nthreads);
    L.red = take(16);
    L.flag = (liptr)take(4);
    L.idxl = (liptr)take((size_t)(d.N / 2 + 2));
}

// one-off per kernel: constants into LDS, zero the padding of the MFMA operands
__device__ inline void qp_lds_init(QPLds &L, const QPDims &d, const QPConst &c) {
    const int tid = SRH_TID, nt = blockDim.x, n = d.n, ld = d.ld;
    // the whole carve starts from zeros (the MFMA operands rely on zero padding; later rounds of a launch inherit the
    // LDS of the previous workgroup), and the BARRIER below orders the zeroing before the constant rows written next
    // by other threads -- without it a late zeroing thread could wipe a Cq / XA entry: an inexact Hessian the
    // interior point corrects, visible only as a run-to-run difference of 1e-10 (tools/determinism_probe.py)
    {
        const size_t total = qp_lds_bytes(d, nt) / sizeof(double);
        for (size_t e = tid; e < total; e += nt) L.P[e] = 0.0;
        __syncthreads();
    }
    L.psel = -1;
    L.ready = true;
    for (int e = tid; e < d.nzr * n; e += nt) {
        const int r = e / n, j = e - r * n;
        L.AB[(d.RC + r) * ld + j] = c.Cq[e];
        if (!d.split) L.W[(d.RC + r) * ld + j] = c.Cq[e];
    }
    for (int e = tid; e < (d.nX + d.nXf) * n; e += nt) {
        const int r = e / n, j = e - r * n;
        L.XAl[r * ld + j] = r < d.nX ? c.XA[(size_t)r * n + j] : c.XfA[(size_t)(r - d.nX) * n + j];
    }
    __syncthreads();
}

#ifdef SRH_PROFILE
#define SRH_LAP(x) lap(x)
#else
#define SRH_LAP(x) ((void)0)
#endif

namespace qp {

// ------------------------------------------------------------------ inequality rows
// value of row r of x-stage k (1..N) applied to the vector (vx, sv):  a_x . vx + a_s * sv
__device__ __forceinline__ double xrow_dot(const QPDims &d, const QPConst &c, int k, int r, cgptr vx,
                                           double sv) {
    const int n = d.n;
    if (d.tr) {
        if (r < n) return c.xs[r] * vx[r] - sv;
        if (r < 2 * n) return -c.xs[r - n] * vx[r - n] - sv;
        if (r == 2 * n) return -sv;
        r -= 2 * n + 1;
    }
    cgptr row = (r < d.nX) ? c.XA + (size_t)r * n : c.XfA + (size_t)(r - d.nX) * n;
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
    for (int e = SRH_TID; e < nxr; e += blockDim.x) {
        const int k = e / d.RX + 1, r = e - (k - 1) * d.RX;
        if (r < xrows_of(d, k)) f(e, false, k, r);
    }
    const int nur = d.N * d.nU;
    for (int e = SRH_TID; e < nur; e += blockDim.x) {
        const int k = e / d.nU, r = e - k * d.nU;
        f(nxr + e, true, k, r);
    }
}

// a . w for every row with w = (vx, vs, vu); out[row]
__device__ __forceinline__ void rows_apply(const QPDims &d, const QPConst &c, cgptr vx, cgptr vs, cgptr vu, gptr out) {
    for_rows(d, [&](int row, bool isU, int k, int r) {
        if (!isU) {
            out[row] = xrow_dot(d, c, k, r, vx + (size_t)k * d.n, vs[k]);
        } else {
            cgptr ua = c.UA + (size_t)r * d.m, uk = vu + (size_t)k * d.m;
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
__device__ __forceinline__ void rollout(const QPDims &d, const QPDyn &dyn, const QPData &q, cgptr u, gptr x, QPLds &L) {
    const int n = d.n, m = d.m;
    for (int e = SRH_TID; e < n; e += blockDim.x) { L.v1[e] = q.x0[e]; x[e] = q.x0[e]; }
    __syncthreads();
    for (int k = 0; k < d.N; ++k) {
        const size_t i = (size_t)L.idxl[k];
        for (int e = SRH_TID; e < m; e += blockDim.x) L.Qu[e] = u[(size_t)k * m + e];
        __syncthreads();
        wg::matTvec(L.v2, dyn.AT + i * n * n, n, n, n, L.v1, dyn.d + i * n, L.part);
        wg::matTvec(L.v2, dyn.BT + i * m * n, n, m, n, L.Qu, (clptr)L.v2, L.part);
        for (int e = SRH_TID; e < n; e += blockDim.x) { L.v1[e] = L.v2[e]; x[(size_t)(k + 1) * n + e] = L.v2[e]; }
        __syncthreads();
    }
}

// s_0 = max(0, ||xs (x0 - xbar_0)||_inf - delta)   (every thread returns it)
__device__ __forceinline__ double slack0(const QPDims &d, const QPConst &c, const QPData &q, QPLds &L) {
    if (!d.tr) return 0.0;
    double v = 0.0;
    for (int e = SRH_TID; e < d.n; e += blockDim.x) v = fmax(v, fabs(c.xs[e] * (q.x0[e] - q.xk[e])));
    v = wg::reduce(v, 1, L.red);
    return fmax(0.0, v - q.delta);
}

// objective value (without the 1/2, as cvxpy reports): locp.py:218-263
__device__ __forceinline__ double objective(const QPDims &d, const QPConst &c, const QPData &q, cgptr x, cgptr u,
                                       cgptr s, QPLds &L) {
    double acc = 0.0;
    const int n = d.n, nz = d.nz, m = d.m;
    for (int k = SRH_TID; k <= d.N; k += blockDim.x) {
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
__device__ __forceinline__ void stage_prepass(const QPDims &d, const QPConst &c, const QPData &q, QPWork &w, bool with_dual) {
    const int n = d.n, nz = d.nz, m = d.m, N = d.N;
    // e_k = H x_k - z_k
    for (int e = SRH_TID; e < (N + 1) * nz; e += blockDim.x) {
        const int k = e / nz, a = e - k * nz;
        double v = q.z ? -q.z[e] : 0.0;
        for (int j = 0; j < n; ++j) v = fma(c.H[a * n + j], w.x[(size_t)k * n + j], v);
        w.ez[e] = v;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(SRH_TID >> 6), lane = SRH_TID & 63, nw = blockDim.x >> 6;
    for (int k = 1 + wave; k <= N; k += nw) {
        cgptr Dk = w.D + (size_t)(k - 1) * d.RX, rk = w.rho + (size_t)(k - 1) * d.RX;
        cgptr lk = w.lam + (size_t)(k - 1) * d.RX;
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
    cgptr Du = w.D + (size_t)N * d.RX, ru = w.rho + (size_t)N * d.RX, lu = w.lam + (size_t)N * d.RX;
    for (int e = SRH_TID; e < N * m * m; e += blockDim.x) {
        const int k = e / (m * m), ab = e - k * m * m, a = ab / m, b = ab - a * m;
        double v = c.R2[ab];
        for (int r = 0; r < d.nU; ++r) v = fma(c.UA[r * m + a] * Du[(size_t)k * d.nU + r], c.UA[r * m + b], v);
        w.Huu[e] = v;
    }
    for (int e = SRH_TID; e < N * m; e += blockDim.x) {
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

using wg::qp_d4;
using wg::mfma_atb;

// Register-resident variant for products that are accumulated over several passes (split mode): tile
// t = wave + r * (number of waves), r < NR, of the MT x NTl tile grid lives in acc[r].
//   acc[r] += sum_{k<K} Lm[k][i] * Rm[k][j]      (no barrier inside)
template <int NR>
__device__ __forceinline__ void mfma_acc(qp_d4 (&acc)[NR], clptr Lm, clptr Rm, int K, int MT, int NTl, int ld) {
    static_assert(NR % 2 == 0, "tiles are processed in pairs");
    const int wave = __builtin_amdgcn_readfirstlane(SRH_TID >> 6), lane = SRH_TID & 63, nw = blockDim.x >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    const int ntiles = MT * NTl;
#pragma unroll
    for (int r = 0; r < NR; r += 2) {
        const int t0 = wave + r * nw, t1 = t0 + nw;
        if (t0 < ntiles) {                             // wave-uniform
            const bool has1 = t1 < ntiles;
            const int ti0 = t0 / NTl, tj0 = t0 - ti0 * NTl;
            const int ti1 = has1 ? t1 / NTl : ti0, tj1 = has1 ? t1 - ti1 * NTl : tj0;
            clptr la0 = Lm + kk * ld + 16 * ti0 + l16, rb0 = Rm + kk * ld + 16 * tj0 + l16;
            clptr la1 = Lm + kk * ld + 16 * ti1 + l16, rb1 = Rm + kk * ld + 16 * tj1 + l16;
            qp_d4 c0 = acc[r], c1 = acc[r + 1];
            for (int k0 = 0; k0 < K; k0 += 4) {
                const double a0 = la0[k0 * ld], b0 = rb0[k0 * ld];
                const double a1 = la1[k0 * ld], b1 = rb1[k0 * ld];
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c0, 0, 0, 0);
                if (has1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c1, 0, 0, 0);
            }
            acc[r] = c0;
            acc[r + 1] = c1;
        }
    }
}

// C rows < srows of the accumulated tiles (no barrier inside)
template <int NR>
__device__ __forceinline__ void mfma_put(const qp_d4 (&acc)[NR], lptr C, int ldc, int MT, int NTl, int srows) {
    const int wave = __builtin_amdgcn_readfirstlane(SRH_TID >> 6), lane = SRH_TID & 63, nw = blockDim.x >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    const int ntiles = MT * NTl;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int t = wave + r * nw;
        if (t >= ntiles) continue;
        const int ti = t / NTl, tj = t - ti * NTl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r0 = 16 * ti + kk + 4 * q;
            if (r0 < srows) C[r0 * ldc + 16 * tj + l16] = acc[r][q];
        }
    }
}

// stage Hessian of x_k added to the (i,j) entry: Qx (+ terminal) + slack-eliminated trust region + X rows
__device__ __forceinline__ double stage_hess(const QPDims &d, const QPConst &c, const QPLds &L, int k, int i, int j,
                                             double Hss, int nxrows) {
    double v = 0.0;
    for (int a = 0; a < d.nz; ++a) v = fma(c.HtQz2[i * d.nz + a], c.H[a * d.n + j], v);     // 2 H^T Qz H (terminal stage only)
    if (k == d.N && c.Qzf) v += c.QxN[(size_t)i * d.n + j] - c.Qx[(size_t)i * d.n + j];
    if (d.tr) {
        if (i == j) v += L.hdv[i];
        else v -= L.cvv[i] * L.cvv[j] / Hss;
    }
    for (int r = 0; r < nxrows; ++r) v = fma(L.XAl[r * d.ld + i] * L.Dx[r], L.XAl[r * d.ld + j], v);
    return v;
}


// Gain of stage k from Quu = Lc Lc^T (Cholesky in registers, recomputed by every thread):
//   Y = Lc^-1 Qux (m x n), K_k = -Lc^-T Y, kff_k = -Quu^-1 Qu.
// Besides K (LDS + HBM) the thread of column j writes the extra Gram rows used by the second product:
//   left  panel rows NK..   : -Y[:, j],  +sqrt(D_r) XA[r][j]
//   right panel rows NK..   : +Y[:, j],  +sqrt(D_r) XA[r][j]
// so that  A^T W + left^T right = A^T P A - Qux^T Quu^-1 Qux + X^T D X  comes out of one MFMA product,
// exactly symmetric in the added terms.  Column n / n+1 of the right panel receive pv / adj (rows < n).
template <int M, bool SPLIT>
__device__ __forceinline__ bool stage_gain_t(const QPDims &d, QPWork &w, QPLds &L, int k, bool extras) {
    const int n = d.n, ld = d.ld, NK = d.NK, tid = SRH_TID, nt = blockDim.x;
    const int wb = SPLIT ? 0 : NK;            // first extra row of the right panel
    double Lr[M * M], inv[M];
    double dmax = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) dmax = fmax(dmax, fabs(L.Quu[i * M + i]));
    bool ok = wg::chol_reg<M>(L.Quu, M, 0.0, Lr, inv);
    // a breakdown caused by round-off in a nearly singular Quu is retried with a growing diagonal shift
    // (inexact Newton step; the interior-point iteration corrects it)
    double shift = 1e-14 * dmax;
    for (int attempt = 0; attempt < 7 && !ok; ++attempt, shift *= 100.0) ok = wg::chol_reg<M>(L.Quu, M, shift, Lr, inv);
    if (!ok) return false;          // uniform: every thread factors the same matrix
    // the extra rows that do not depend on the gain -- sqrt(D) X and the pv / adj columns -- are written by other
    // threads (other waves when n_x < 128) while the first n_x + 1 threads run the column solves
    if (extras) {
        const int o1 = nt >= 4 * 128 ? 128 : 0, o2 = nt >= 4 * 128 ? 256 : 0;
        for (int j = tid - o1; j >= 0 && j < n; j += nt)
            for (int r = 0; r < d.nX; ++r) {
                const double v = L.sDx[r] * L.XAl[r * ld + j];
                L.AB[(NK + M + r) * ld + j] = v;
                L.W[(wb + M + r) * ld + j] = v;
            }
        if (!SPLIT)
            for (int j = tid - o2; j >= 0 && j < n; j += nt) {
                L.W[j * ld + n] = L.pv[j];
                L.W[j * ld + n + 1] = L.adj[j];
            }
    }
    for (int j = tid; j <= n; j += nt) {
        clptr b = j < n ? L.QUX + j : L.Qu;
        const int bs = j < n ? ld : 1;
        double y[M], x[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double sum = b[i * bs];
#pragma unroll
            for (int q = 0; q < i; ++q) sum -= Lr[i * M + q] * y[q];
            y[i] = sum * inv[i];
        }
#pragma unroll
        for (int i = M - 1; i >= 0; --i) {
            double sum = y[i];
#pragma unroll
            for (int q = i + 1; q < M; ++q) sum -= Lr[q * M + i] * x[q];
            x[i] = sum * inv[i];
        }
        if (j < n) {
#pragma unroll
            for (int a = 0; a < M; ++a) {
                L.Km[a * ld + j] = -x[a];
                w.K[((size_t)k * M + a) * n + j] = -x[a];
            }
            if (extras) {
#pragma unroll
                for (int a = 0; a < M; ++a) {
                    L.AB[(NK + a) * ld + j] = -y[a];
                    L.W[(wb + a) * ld + j] = y[a];
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < M; ++a) { L.kf[a] = -x[a]; w.kff[(size_t)k * M + a] = -x[a]; }
        }
    }
    // (compile-time indices only: a runtime index would push the factor into scratch memory)
#pragma unroll
    for (int e = 0; e < M * M; ++e)
        if (tid == e) w.Qinv[(size_t)k * M * M + e] = (e % M <= e / M) ? Lr[e] : 0.0;
    __syncthreads();
    return true;
}

// MSEL > 0: the kernel is an instantiation for exactly n_u = MSEL (the reference's robots: 4 and 8 cables) and carries
// only that gain routine -- the all-sizes kernel pays for the register Cholesky of every n_u <= 8 in its allocation
// (3.5 % on the Diamond shape); MSEL = 0: any n_u.
template <bool SPLIT, int MSEL>
__device__ __forceinline__ bool stage_gain(const QPDims &d, QPWork &w, QPLds &L, int k, bool extras) {
    if constexpr (MSEL > 0) return stage_gain_t<MSEL, SPLIT>(d, w, L, k, extras);
    switch (d.m) {
        case 1: return stage_gain_t<1, SPLIT>(d, w, L, k, extras);
        case 2: return stage_gain_t<2, SPLIT>(d, w, L, k, extras);
        case 3: return stage_gain_t<3, SPLIT>(d, w, L, k, extras);
        case 4: return stage_gain_t<4, SPLIT>(d, w, L, k, extras);
        case 5: return stage_gain_t<5, SPLIT>(d, w, L, k, extras);
        case 6: return stage_gain_t<6, SPLIT>(d, w, L, k, extras);
        case 7: return stage_gain_t<7, SPLIT>(d, w, L, k, extras);
        case 8: return stage_gain_t<8, SPLIT>(d, w, L, k, extras);
        default: break;
    }
    // m > 8: factor once in LDS (thread 0), per-column solves from LDS
    const int n = d.n, m = d.m, ld = d.ld, NK = d.NK, tid = SRH_TID, nt = blockDim.x;
    const int wb = SPLIT ? 0 : NK;
    if (!wg::chol_factor(L.Quu, L.Lc, m, L.flag, true)) return false;
    for (int j = tid; j <= n; j += nt) {
        clptr b = j < n ? L.QUX + j : L.Qu;
        const int bs = j < n ? ld : 1;
        // forward substitution into the extra-row slots / a scratch column of W, then back substitution
        lptr yc = j < n ? L.W + wb * ld + j : L.v3;
        const int ys = j < n ? ld : 1;
        for (int i = 0; i < m; ++i) {
            double sum = b[i * bs];
            for (int q = 0; q < i; ++q) sum -= L.Lc[i * m + q] * yc[q * ys];
            yc[i * ys] = sum / L.Lc[i * m + i];
        }
        lptr xc = j < n ? L.Km + j : L.kf;
        const int xs = j < n ? ld : 1;
        for (int i = m - 1; i >= 0; --i) {
            double sum = yc[i * ys];
            for (int q = i + 1; q < m; ++q) sum -= L.Lc[q * m + i] * xc[q * xs];
            xc[i * xs] = sum / L.Lc[i * m + i];
        }
        if (j < n) {
            for (int a = 0; a < m; ++a) {
                const double xv = -xc[a * xs];
                xc[a * xs] = xv;
                w.K[((size_t)k * m + a) * n + j] = xv;
                if (extras) L.AB[(NK + a) * ld + j] = -yc[a * ys];
            }
            if (extras) {
                for (int r = 0; r < d.nX; ++r) {
                    const double v = L.sDx[r] * L.XAl[r * ld + j];
                    L.AB[(NK + m + r) * ld + j] = v;
                    L.W[(wb + m + r) * ld + j] = v;
                }
                if (!SPLIT) {
                    L.W[j * ld + n] = L.pv[j];
                    L.W[j * ld + n + 1] = L.adj[j];
                }
            }
        } else {
            for (int a = 0; a < m; ++a) { L.kf[a] = -L.kf[a]; w.kff[(size_t)k * m + a] = L.kf[a]; }
        }
    }
    for (int e = tid; e < m * m; e += nt) w.Qinv[(size_t)k * m * m + e] = L.Lc[e];
    __syncthreads();
    return true;
}

// ---- the [A | B] panel of the current TPWL region stays in LDS: consecutive stages of a horizon mostly
// share their region, so the 30 KB table read (and its L2 latency) is paid only when the region changes;
// the vector sweeps take A v, A^T v, B u, B^T v from the panel instead of streaming the tables again.
// Returns true when the panel was (re)written: the caller places the barrier that publishes it (the factorising pass
// has one after its own stage loads anyway; the vector sweeps only pay it on a reload).  Which region the panel holds
// is tracked in a per-thread copy (L.psel, identical in every thread): no LDS flag, no barrier to protect it.
__device__ __forceinline__ bool panel_load(const QPDims &d, const QPDyn &dyn, QPLds &L, int k) {
    const int n = d.n, m = d.m, ld = d.ld, tid = SRH_TID, nt = blockDim.x;
    // both values are the same in every lane: readfirstlane moves them to scalar registers, the branch below (and the
    // caller's barrier on it) is then a scalar branch, not exec-mask control flow
    const int sel = __builtin_amdgcn_readfirstlane(L.idxl[k]);
    if (dyn.idx != nullptr && __builtin_amdgcn_readfirstlane(L.psel) == sel) return false;
    cgptr Ag = dyn.A + (size_t)sel * n * n, Bg = dyn.B + (size_t)sel * n * m;
    const int NPa = d.NPa;
    const int jj = tid % NPa, ii0 = tid / NPa, rstep = nt / NPa;
    if (ii0 < rstep && jj < n + m) {
        cgptr src = jj < n ? Ag + jj : Bg + (jj - n);
        const int stride = jj < n ? n : m;
        for (int i = ii0; i < n; i += rstep) L.AB[i * ld + jj] = src[(size_t)i * stride];
    }
    L.psel = sel;
    return true;
}

// y[j] = sum_{i<n} AB[i][j] v[i], j < n + m      ( [A^T v ; B^T v] )
__device__ __forceinline__ void panel_T_vec(const QPDims &d, QPLds &L, clptr v, lptr y) {
    const int n = d.n, m = d.m, ld = d.ld, NPa = d.NPa, tid = SRH_TID, nt = blockDim.x;
    const int S = nt / NPa > 0 ? nt / NPa : 1;
    const int col = tid % NPa, sl = tid / NPa;
    if (sl < S) {
        double acc = 0.0;
        if (col < n + m) for (int i = sl; i < n; i += S) acc = fma(L.AB[i * ld + col], v[i], acc);
        L.part[sl * NPa + col] = acc;
    }
    __syncthreads();
    if (tid < NPa) {
        double r = 0.0;
        for (int q = 0; q < S; ++q) r += L.part[q * NPa + tid];
        y[tid] = r;
    }
    __syncthreads();
}

// ---- the two vector sweeps of a Newton system (see riccati_solve): ONE barrier per stage, wave-local DPP sums, and
// a software pipeline over the stages: everything a stage reads from HBM/L2 (gain rows, gradients) is requested one
// stage ahead.  For that to pay, every global access inside the loops is UNCONDITIONAL (clamped indices; lanes that
// own no result store to a dump slot): loads and stores share the in-order vmcnt counter, and behind a conditional
// access the compiler can only wait with vmcnt(0), i.e. for the write acknowledgement of the previous stage's stores.
//
// backward, stage k:  Qu = B^T pv + gu  by every wave itself (GA lanes per input row);  columns col = wave + nw c of
//   pv_k = gx + A^T pv + K^T Qu  (lane (c, g) of 8 x 8: rows g, g + 8, ..; lane g < m adds K[g][col] Qu[g]);
//   Qu is parked in kff[k] and turned into kff = -Quu^-1 Qu for all stages at once after the sweep.
template <int MSEL, int NSEL>
__device__ __forceinline__ void vector_sweep_back(const QPDims &d, const QPDyn &dyn, QPWork &w, QPLds &L) {
    const int n = d.n, m = d.m, N = d.N, ld = d.ld, tid = SRH_TID, nt = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6;
    constexpr int NP = 2;                                   // column passes of 64: n <= 128
    lptr pv = L.pv, pn = L.v3;
    const int c = lane >> 3, g8 = lane & 7;
    const int gk = g8 < m ? g8 : m - 1;                     // clamped gain row of this lane
    auto run = [&](auto ga_tag) {
        constexpr int GA = decltype(ga_tag)::value;
        const int a = lane / GA, g = lane % GA;
        const bool va = a < m;
        const int ac = va ? a : m - 1;
        struct Regs { double gu, kk[NP], gx[NP]; };
        auto fetch = [&](int k, Regs &r) {
            r.gu = w.gu[(size_t)k * m + ac];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int col = wave + nw * c + 64 * q;
                const int cc = col < n ? col : n - 1;
                if (64 * q < n) {                            // uniform
                    r.kk[q] = w.K[((size_t)k * m + gk) * n + cc];
                    r.gx[q] = w.gx[(size_t)k * n + cc];
                }
            }
        };
        auto stage = [&](int k, const Regs &cur, Regs &nxt) {
            if (panel_load(d, dyn, L, k)) __syncthreads();
            double p = 0.0;
            if (va)
                for (int i = g; i < n; i += GA) p = fma(L.AB[i * ld + n + a], pv[i], p);
            p = wg::group_sum<GA>(p) + cur.gu;
            if (g == 0 && va) L.Qu[a] = p;                   // every wave stores the same value
            fetch(k >= 1 ? k - 1 : 0, nxt);
            *(va ? w.kff + (size_t)k * m + a : w.dump + lane) = p;
            __builtin_amdgcn_wave_barrier();                  // Qu written and read by this wave: LDS is in order
            if (k >= 1) {
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (64 * q < n) {                        // uniform
                        const int col = wave + nw * c + 64 * q;
                        double acc = 0.0;
                        if (col < n) {
                            for (int i = g8; i < n; i += 8) acc = fma(L.AB[i * ld + col], pv[i], acc);
                            if (g8 < m) acc = fma(cur.kk[q], L.Qu[g8], acc);
                            for (int a2 = g8 + 8; a2 < m; a2 += 8) acc = fma(w.K[((size_t)k * m + a2) * n + col], L.Qu[a2], acc);
                            if (g8 == 0) acc += cur.gx[q];
                        }
                        acc = wg::group_sum<8>(acc);
                        if (g8 == 0 && col < n) pn[col] = acc;
                    }
                }
            }
            __syncthreads();
            lptr t = pv; pv = pn; pn = t;
        };
        Regs ra, rb;
        fetch(N - 1, ra);
        int k = N - 1;
        for (; k >= 1; k -= 2) { stage(k, ra, rb); stage(k - 1, rb, ra); }
        if (k == 0) stage(0, ra, rb);
    };
    if (m <= 4) run(std::integral_constant<int, 16>{});
    else if (m <= 8) run(std::integral_constant<int, 8>{});
    else run(std::integral_constant<int, 4>{});
    // kff_k = -Quu_k^-1 Qu_k, one thread per stage (the factor and Qu from HBM/L2)
    for (int k = tid; k < N; k += nt) wg::chol_solve_neg(w.Qinv + (size_t)k * m * m, m, w.kff + (size_t)k * m, 1, w.kff + (size_t)k * m, 1);
    __syncthreads();
}

// forward, stage k:  du = K dx + kff by every wave itself into xu = [dx_k ; du_k];  rows r = wave + nw c of
//   dx_{k+1} = [A | B] xu  (lane (c, g) of 8 x 8: columns g, g + 8, ..).  The slack steps follow after the sweep.
template <int MSEL, int NSEL>
__device__ __forceinline__ void vector_sweep_fwd(const QPDims &d, const QPDyn &dyn, QPWork &w, QPLds &L) {
    const int n = d.n, m = d.m, N = d.N, ld = d.ld, tid = SRH_TID, nt = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6, nm = n + m;
    for (int e = tid; e < n; e += nt) { L.v1[e] = 0.0; w.dx[e] = 0.0; }
    __syncthreads();
    lptr xu = L.v1, xn = L.v2;
    const int c = lane >> 3, g8 = lane & 7;
    auto run = [&](auto ga_tag) {
        constexpr int GA = decltype(ga_tag)::value;
        constexpr int KQ = NSEL > 0 ? (NSEL + GA - 1) / GA : 1;      // gain entries per lane (pipelined when n is fixed)
        const int a = lane / GA, g = lane % GA;
        const bool va = a < m;
        const int ac = va ? a : m - 1;
        struct Regs { double kk[KQ], kf; };
        auto fetch = [&](int k, Regs &r) {
            if constexpr (NSEL > 0) {
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    const int j = g + GA * q;
                    r.kk[q] = w.K[((size_t)k * m + ac) * n + (j < n ? j : n - 1)];
                }
                r.kf = w.kff[(size_t)k * m + ac];
            }
        };
        auto stage = [&](int k, const Regs &cur, Regs &nxt) {
            if (panel_load(d, dyn, L, k)) __syncthreads();
            double p = 0.0;
            if constexpr (NSEL > 0) {
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    const int j = g + GA * q;
                    p = fma(cur.kk[q], j < n ? xu[j] : 0.0, p);
                }
                p = wg::group_sum<GA>(p) + cur.kf;
            } else {
                for (int j = g; j < n; j += GA) p = fma(w.K[((size_t)k * m + ac) * n + j], xu[j], p);
                p = wg::group_sum<GA>(p) + w.kff[(size_t)k * m + ac];
            }
            if (g == 0 && va) xu[n + a] = p;                 // every wave stores the same value
            fetch(k + 1 < N ? k + 1 : k, nxt);
            *(va ? w.du + (size_t)k * m + a : w.dump + lane) = p;
            __builtin_amdgcn_wave_barrier();                  // xu[n..] written and read by this wave: LDS is in order
            for (int r = wave + nw * c; r < ((n + 63) & ~63); r += 64) {   // uniform trip count
                double acc = 0.0;
                if (r < n)
                    for (int j = g8; j < nm; j += 8) acc = fma(L.AB[r * ld + j], xu[j], acc);
                acc = wg::group_sum<8>(acc);
                if (g8 == 0 && r < n) xn[r] = acc;
                *(r < n ? w.dx + (size_t)(k + 1) * n + r : w.dump + lane) = acc;
            }
            __syncthreads();
            lptr t = xu; xu = xn; xn = t;
        };
        Regs ra, rb;
        fetch(0, ra);
        int k = 0;
        for (; k + 1 < N; k += 2) { stage(k, ra, rb); stage(k + 1, rb, ra); }
        if (k < N) stage(k, ra, rb);
    };
    if (m <= 4) run(std::integral_constant<int, 16>{});
    else if (m <= 8) run(std::integral_constant<int, 8>{});
    else run(std::integral_constant<int, 4>{});
    // slack steps ds_k = -(gs_k + cv_k . dx_k) / Hss_k from the stored dx (one wave per stage), or zero
    if (tid == 0) w.ds[0] = 0.0;
    if (d.tr) {
        for (int k = 1 + wave; k <= N; k += nw) {
            double v = 0.0;
            for (int j = lane; j < n; j += 64) v = fma(w.cv[(size_t)k * n + j], w.dx[(size_t)k * n + j], v);
            v = wg::wave_sum(v);
            if (lane == 0) w.ds[k] = -(w.gs[k] + v) / w.Hss[k];
        }
    } else {
        for (int k = 1 + tid; k <= N; k += nt) w.ds[k] = 0.0;
    }
    __syncthreads();
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
template <bool SPLIT, int MSEL, int NSEL>
__device__ __forceinline__ bool riccati_solve(const QPDims &d, const QPConst &c, const QPDyn &dyn, QPWork &w, QPLds &L,
                                     bool full, bool with_dual, double *rd_out) {
    const int n = d.n, m = MSEL > 0 ? MSEL : d.m, N = d.N, ld = d.ld, NK = d.NK, NPa = d.NPa, n16 = (d.n + 15) & ~15;
    const int tid = SRH_TID, nt = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6;
    const int xoff = d.tr ? 2 * n + 1 : 0;
    double rd = 0.0;
#ifdef SRH_PROFILE
    long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tl = clock64();
    auto lap = [&](int sl) { const long long now = clock64(); tp[sl] += now - tl; tl = now; };
#endif
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
    if (!full) vector_sweep_back<MSEL, NSEL>(d, dyn, w, L);   // re-solve with the stored gains / factors
    // What a factorising stage reads from HBM/L2 (Huu, the trust-region terms, the X-row weights, the gradients) is
    // requested one stage ahead: one value of each kind per thread (n <= blockDim, m^2 <= blockDim), consumed at the
    // start of the next stage, a whole stage of MFMA work later.
    // the Qu phase needs no result of the B^T W product: with the whole W panel it runs on the upper half of the
    // workgroup while the first waves multiply the (four or five) tiles of that product
    const int qbase = (!SPLIT && nt >= 512) ? 256 : 0;
    struct StageIn { double Huu, hd, cv, Dx, g1, g2; };
    auto stage_fetch = [&](int k, StageIn &r) {
        if (tid < m * m) r.Huu = w.Huu[(size_t)k * m * m + tid];
        if (k >= 1) {
            if (tid < n) { r.hd = w.hd[(size_t)k * n + tid]; r.cv = w.cv[(size_t)k * n + tid]; }
            if (tid < d.nX) r.Dx = w.D[(size_t)(k - 1) * d.RX + xoff + tid];
            if (tid < 2 * n) r.g1 = tid < n ? w.gx[(size_t)k * n + tid] : w.gxd[(size_t)k * n + tid - n];
        }
        // thread qbase + 8 o owns output o of the Qu phase (o < m: Qu, else the dual residual): its gradient entry
        if ((tid & 7) == 0 && tid >= qbase && ((tid - qbase) >> 3) < 2 * m) { const int o = (tid - qbase) >> 3; r.g2 = o < m ? w.gu[(size_t)k * m + o] : w.gud[(size_t)k * m + o - m]; }
    };
    StageIn sin{0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, snx{0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (full) stage_fetch(N - 1, sin);
    for (int k = N - 1; k >= 0 && full; --k) {
        {
            const bool reloaded = panel_load(d, dyn, L, k);  // published by the barrier below
            if (tid < m * m) L.Quu[tid] = sin.Huu;
            if (k >= 1) {
                if (tid < n) { L.hdv[tid] = sin.hd; L.cvv[tid] = sin.cv; }
                if (tid < d.nX) { L.Dx[tid] = sin.Dx; L.sDx[tid] = sqrt(sin.Dx); }
            }
            const double gin1 = sin.g1, gin2 = sin.g2;
            // Qu = gu + B^T pv, dual residual wrt u_k = gud + B^T adj (8 lanes per output, DPP sums)
            auto qu_phase = [&]() {
                if (tid >= qbase && ((tid - qbase) >> 3) < 2 * m) {          // whole groups of 8 lanes
                    const int o = (tid - qbase) >> 3, g8 = tid & 7, a = o < m ? o : o - m;
                    clptr vec = o < m ? L.pv : L.adj;
                    double v = 0.0;
                    for (int i = g8; i < n; i += 8) v = fma(L.AB[i * ld + n + a], vec[i], v);
                    v = wg::group_sum<8>(v);
                    if (g8 == 0) {
                        if (o < m) L.Qu[a] = v + gin2;
                        else L.rdu[a] = v + gin2;
                    }
                }
            };
            if (k >= 1) stage_fetch(k - 1, snx);
            // what was just written (Quu, hd, cv, D) is first read behind the barrier that ends the W product; the
            // product itself reads P (complete since the barrier of the previous stage's tail) and the panel: a barrier
            // is needed here only for a new panel, for K-padding rows of P, and in split mode
            if (SPLIT || reloaded || NK > n) __syncthreads();
            SRH_LAP(0);
            // split mode (n_x > 64: P, AB and W do not fit LDS together): W = P [A|B] is produced 48 rows at a
            // time; the Gram products that contract over the rows of W accumulate in MFMA registers across the
            // two halves, so only half of W is ever resident
            qp_d4 accP[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
            if constexpr (SPLIT) {
                qp_d4 accQ[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
                for (int h = 0; h < 2; ++h) {
                    const int r0 = 48 * h;
                    const int rows = h == 0 ? (NK < 48 ? NK : 48) : NK - 48;
                    if (rows <= 0) break;
                    const int rt = (rows + 15) >> 4;
                    mfma_atb(L.W, ld, L.P + r0, L.AB, NK, rt, NPa >> 4, ld, n - r0, 16 * rt);   // W_h = P[r0.., :] [A|B]
                    mfma_acc<2>(accQ, L.AB + (size_t)r0 * ld + n, L.W, rows, 1, NPa >> 4, ld);
                    if (k >= 1) mfma_acc<4>(accP, L.AB + (size_t)r0 * ld, L.W, rows, n16 >> 4, n16 >> 4, ld);
                    __syncthreads();
                }
                SRH_LAP(1);
                mfma_put<2>(accQ, L.QUX, ld, 1, NPa >> 4, m);
                __syncthreads();
            } else {
                mfma_atb(L.W, ld, L.P, L.AB, NK, n16 >> 4, NPa >> 4, ld, n);          // W = P [A|B]
                SRH_LAP(1);
                if (qbase) qu_phase();                                                 // upper waves, beside the product below
                mfma_atb(L.QUX, ld, L.AB + n, L.W, NK, 1, NPa >> 4, ld, 16, m);        // [Qux | B^T P B] = B^T W
            }
            SRH_LAP(2);
            if (qbase == 0) qu_phase();
            // Quu += B^T P B
            for (int e = tid; e < m * m; e += nt) { const int a = e / m, b = e - a * m; L.Quu[e] += L.QUX[a * ld + n + b]; }
            __syncthreads();
            if (with_dual && tid < m) rd = fmax(rd, fabs(L.rdu[tid]));
            SRH_LAP(3);
            if (!stage_gain<SPLIT, MSEL>(d, w, L, k, k >= 1)) return false;
            SRH_LAP(4);
            if (k >= 1) {
                // P_k = A^T W + (extra rows: -Y^T Y + X^T D X); columns n, n+1 deliver A^T pv, A^T adj
                const int K2 = d.nzr ? ((d.RC + d.nzr + 3) & ~3) : NK + d.NE4;
                if constexpr (SPLIT) {
                    // extra Gram rows: the gain wrote +Y / sqrt(D) X into W rows 0..; constant Cq rows copied from
                    // the left panel, the rest of the K extent zeroed; A^T pv, A^T adj by panel mat-vecs
                    const int KE = K2 - NK;
                    for (int e = tid; e < (KE - m - d.nX) * n16; e += nt) {
                        const int r = m + d.nX + e / n16, j = e % n16;
                        const bool cq = d.nzr && NK + r >= d.RC && NK + r < d.RC + d.nzr;
                        L.W[r * ld + j] = cq ? L.AB[(NK + r) * ld + j] : 0.0;
                    }
                    __syncthreads();
                    mfma_acc<4>(accP, L.AB + (size_t)NK * ld, L.W, KE, n16 >> 4, n16 >> 4, ld);
                    mfma_put<4>(accP, L.P, ld, n16 >> 4, n16 >> 4, n);
                    panel_T_vec(d, L, L.pv, L.ypv);
                    panel_T_vec(d, L, L.adj, L.yadj);
                } else {
                    mfma_atb(L.P, ld, L.AB, L.W, K2, n16 >> 4, ((n + 2 + 15) & ~15) >> 4, ld, n16);
                }
                // finish in place (only when something is left to add): the constant 2 H^T Qz H unless it was
                // folded into the product as extra rows, and the slack-eliminated trust region; symmetrised.
                // Each unordered pair (i,j) is owned by one thread.
                if (d.tr || d.nzr == 0) {
                    const double Hss = d.tr ? w.Hss[k] : 1.0;
                    const int sh = 32 - __clz(n - 1), msk = (1 << sh) - 1;          // j = e & msk, i = e >> sh
                    for (int e = tid; e < (n << sh); e += nt) {
                        const int i = e >> sh, j = e & msk;
                        if (j < i || j >= n) continue;
                        double v = 0.5 * (L.P[i * ld + j] + L.P[j * ld + i]);
                        if (d.nzr == 0) v += c.Qx[(size_t)i * n + j];
                        if (d.tr) {
                            if (i == j) v += L.hdv[i];
                            else v -= L.cvv[i] * L.cvv[j] / Hss;
                        }
                        L.P[i * ld + j] = v;
                        L.P[j * ld + i] = v;
                    }
                }
                // pv_new = gx + A^T pv + K^T Qu ; adj_new = gxd + A^T adj.  When P needed no finishing and has no K-padding
                // rows, nothing of this stage reads pv / adj any more and the next reader (the Qu phase) sits behind
                // the barrier of the next W product: they are written in place and the stage ends without a barrier.
                const bool direct = !SPLIT && !(d.tr || d.nzr == 0) && NK == n;        // uniform
                lptr pvo = direct ? L.pv : L.v1, ado = direct ? L.adj : L.v2;
                for (int e = tid; e < 2 * n; e += nt) {
                    const int j = e % n;
                    if (e < n) {
                        double v = gin1 + (SPLIT ? L.ypv[j] : L.P[j * ld + n]);
                        for (int a = 0; a < m; ++a) v = fma(L.Km[a * ld + j], L.Qu[a], v);
                        pvo[j] = v;
                    } else {
                        ado[j] = gin1 + (SPLIT ? L.yadj[j] : L.P[j * ld + n + 1]);
                    }
                }
                if (!direct) {
                    __syncthreads();
                    for (int e = tid; e < (NK - n) * ld; e += nt) L.P[n * ld + e] = 0.0;
                    for (int e = tid; e < n; e += nt) { L.pv[e] = L.v1[e]; L.adj[e] = L.v2[e]; }
                }
            }
            sin = snx;
            SRH_LAP(5);
        }
    }
    if (with_dual) {
        // stationarity of the slacks s_k (parked in ds by the pre-pass)
        if (d.tr) for (int k = 1 + tid; k <= N; k += nt) rd = fmax(rd, fabs(w.ds[k]));
        rd = wg::reduce(rd, 1, L.red);
        if (rd_out) *rd_out = rd;
    }
    SRH_LAP(6);
    vector_sweep_fwd<MSEL, NSEL>(d, dyn, w, L);
    SRH_LAP(7);
#ifdef SRH_PROFILE
    if (w.tprof && tid == 0) for (int i = 0; i < 8; ++i) w.tprof[i] += (double)tp[i];
#endif
    return true;
}

// largest step keeping t + a dt >= 0 and lam + a dlam >= 0 (not clamped to 1)
__device__ __forceinline__ double max_step(const QPDims &d, const QPWork &w, QPLds &L) {
    double a = 1e300;
    for_rows(d, [&](int row, bool, int, int) {
        const double dt = w.dt[row], dl = w.dlam[row];
        if (dt < 0.0) a = fmin(a, -w.t[row] / dt);
        if (dl < 0.0) a = fmin(a, -w.lam[row] / dl);
    });
    return wg::reduce(a, 2, L.red);
}

}  // namespace qp
#include "locp_cond.h"
namespace qp {

// Solve one QP.  Results in w.x, w.u, w.s.  Returns status: 0 optimal, 1 max iterations, 2 numerical failure.
//
// One loop drives every Newton system through a single (inlined) copy of the pre-pass and the Riccati
// solve: mode INIT computes the least-squares starting point (unit weights), PRED the affine-scaling
// direction (factorisation + solve), CORR the Mehrotra corrector (re-solve with the stored factors).
// With `prescreen` the trust-region rows are dropped first (see below) and the full QP is only solved
// when the relaxed minimiser leaves the trust region.
template <bool SPLIT, int MSEL, int NSEL>
__device__ __forceinline__ int solve(const QPDims &dfull, const QPConst &c, const QPDyn &dyn, const QPData &q, gptr work_base,
                                     QPLds &L, double *J_out, int *iters_out, bool prescreen, QPWork &wout, bool full_first = false,
                                     bool warm = false, int *pass_out = nullptr, bool warm_full = false) {
    // warm_full (round 6): the caller's PREVIOUS QP was finished by the FULL pass (trust-region rows active) and this one goes
    // straight to the full pass as well (full_first) -- a rejected SCP step: same linearisation, delta halved or omega raised,
    // gusto.py:383-402 / locp.py:139-141 `update(full=False)` -- so w.u, w.s and w.lam of the full layout still hold that QP's
    // minimiser and multipliers: start from them like the relaxed pass does (the tail of an uncapped solve is one rollout's chain of
    // such QPs: 22 interior-point iterations each from the cold start).  A warm attempt that fails is repeated cold.
    // warm (round 4, as ql::ipm_box): the caller's PREVIOUS QP was finished by the relaxed Riccati pass below, so w.u and
    // w.lam of that layout still hold its minimiser and multipliers -- the relaxed pass of this QP starts from them
    // (t = max(-g(u), floor), lam = max(lam, floor), no starting system); a warm attempt that fails is repeated cold.
    // pass_out: -1 condensed path, 0 relaxed Riccati pass, 1 full QP -- what produced the result.
    constexpr double WARM_FLOOR = 1e-2;
    int tid = SRH_TID;                                       // re-read at the top of every interior-point iteration (dev_la.h: SRH_TID)
    const int nt = blockDim.x;
    int status = 1, it = 0;
    if (pass_out) *pass_out = -1;
    double J = 0.0;
    const int npass = (prescreen && dfull.tr) ? 2 : 1;
    // full_first: the caller expects the trust region to bind (the previous QP of this rollout ended on its boundary, or a
    // lean kernel has just found this QP's relaxed minimiser outside): skip the relaxed attempts -- the full QP has the same
    // minimiser either way, the relaxed solves (~6 ms at C2) would only be discarded
    int first_pass = (full_first && npass == 2) ? 1 : 0;
    // ---- fast path: the QP without its trust-region rows by the condensed interior point (locp_cond.h).  Accepted when
    // it converges and (trust region present) its minimiser lies inside the trust region -- the same argument as the
    // prescreen below; otherwise the stage-wise Riccati solve of the full QP follows.
    if (dfull.cond && (npass == 2 || !dfull.tr) && first_pass == 0) {
        const int st = qpc::solve<MSEL, NSEL>(dfull, c, dyn, q, work_base, L.base, L, &it, wout);
        warm = false;                                  // the condensed attempt has overwritten w.u / w.lam: what follows starts cold
        qp_lds_carve(L, L.base, dfull, nt);            // back to the Riccati layout (its constants are gone: ready = false)
        if (q.dbg && tid == 0) { q.dbg[8 * 61] = 1.0; q.dbg[8 * 61 + 1] = (double)st; q.dbg[8 * 61 + 2] = (double)it; }
#ifdef SRH_PROFILE
        if (st != 0 && tid == 0) printf("[qp] block %d: condensed path status %d after %d iterations -> Riccati path\n", (int)blockIdx.x, st, it);
#endif
        if (st == 0) {
            QPDims d0 = dfull;
            d0.tr = 0;
            QPWork w = wout;
            const int N = d0.N, n = d0.n;
            const double s0 = slack0(dfull, c, q, L);
            for (int k = tid; k < N; k += nt) L.idxl[k] = dyn.idx ? dyn.idx[k] : k;
            for (int e = tid; e <= N; e += nt) w.s[e] = (e == 0) ? s0 : 0.0;
            __syncthreads();
            rollout(d0, dyn, q, w.u, w.x, L);
            J = objective(d0, c, q, w.x, w.u, w.s, L);
            bool inside = true;
            if (dfull.tr) {
                double md = 0.0;
                for (int e = tid + n; e < (N + 1) * n; e += nt) md = fmax(md, fabs(c.xs[e % n] * (w.x[e] - q.xk[e])));
                md = wg::reduce(md, 1, L.red);
                inside = md <= q.delta;
                J += q.omega * s0;
            }
            if (q.dbg && tid == 0) q.dbg[8 * 61 + 3] = inside ? 1.0 : 0.0;
            if (inside) {
                if (J_out) *J_out = J;
                if (iters_out) *iters_out = it;
                return 0;
            }
#ifdef SRH_PROFILE
            if (tid == 0) printf("[qp] block %d: condensed minimiser outside the trust region -> full QP on the Riccati path\n", (int)blockIdx.x);
#endif
            first_pass = npass - 1;                      // outside the trust region: straight to the full QP
        }
    }
    if (!L.ready) qp_lds_init(L, dfull, c);
    for (int pass = first_pass; pass < npass; ++pass) {
        // Trust-region prescreen.  The 2n+1 trust-region rows per stage are 90 % of the inequality rows, yet
        // with GuSTO's delta (1e4 initially) they are almost never active.  Pass 0 solves the QP WITHOUT
        // them: if its minimiser satisfies ||xs (x_k - xbar_k)||_inf <= delta for every k then
        // (x, u, s = s_0 e_0) is feasible for the full QP at the same cost and therefore its minimiser
        // (dropping satisfied constraints cannot change an optimum; the slack cost omega*s >= 0 is minimal
        // at 0) -- identical result, 10x fewer rows.  Otherwise pass 1 solves the full QP.
        QPDims d = dfull;
        if (npass == 2 && pass == 0) {
            d.tr = 0;
            d.nrx = d.nX;
            d.RX = d.nrx + d.nXf;
            d.NR = d.N * d.RX + d.N * d.nU;
            d.ng = d.N * d.nrx + d.nXf + d.N * d.nU;
        }
        QPWork w;
        qp_carve(w, work_base, d);
        wout = w;
        if (pass_out) *pass_out = pass;
        const bool warm_now = (warm && npass == 2 && pass == 0) || (warm_full && npass == 2 && pass == 1 && first_pass == 1);
        warm = false; warm_full = false;               // (a repeated pass starts cold)
        const int N = d.N, n = d.n, m = d.m;
        w.tprof = q.dbg ? q.dbg + 8 * 63 : (gptr)nullptr;
#ifdef SRH_PROFILE
        long long tm[6] = {0, 0, 0, 0, 0, 0};
        long long t_last = clock64();
        auto lap = [&](int slot) { const long long now = clock64(); tm[slot] += now - t_last; t_last = now; };
#endif
        const double s0 = slack0(dfull, c, q, L);
        if (!warm_now) for (int e = tid; e < N * m; e += nt) w.u[e] = 0.0;
        for (int e = tid; e <= N; e += nt) w.s[e] = (e == 0) ? s0 : ((warm_now && d.tr) ? fmax(w.s[e], 0.0) : 0.0);
        for (int k = tid; k < N; k += nt) L.idxl[k] = dyn.idx ? dyn.idx[k] : k;
        __syncthreads();
        rollout(d, dyn, q, w.u, w.x, L);
        status = 1;
        it = 0;
        enum { INIT = 0, PRED = 1, CORR = 2 };
        int mode = INIT;
        double mu = 0.0, rp = 0.0, sig = 0.0, sd = 1.0, sp = 1.0, dreg = 0.0;
        bool near_opt = false;
        // scales for the stopping test (as oracle/riccati_ipm.py)
        auto scales = [&]() {
            for (int e = tid; e < n; e += nt) {
                double g = 0.0;
                if (q.z) for (int a = 0; a < d.nz; ++a) g = fma(c.HtQz2[e * d.nz + a], -q.z[d.nz + a], g);
                sd = fmax(sd, fabs(g));
            }
            for (int e = tid; e < d.nU; e += nt) sp = fmax(sp, fabs(c.Ub[e]));
            sd = fmax(wg::reduce(sd, 1, L.red), q.omega);
            sp = fmax(wg::reduce(sp, 1, L.red), fabs(q.delta));
            dreg = d.reg / sd;
        };
        if (warm_now && d.ng > 0) {
            rows_apply(d, c, w.x, w.s, w.u, w.rg);
            __syncthreads();
            for_rows(d, [&](int row, bool isU, int k, int r) {
                const double g = w.rg[row] - row_h(d, c, q, isU, k, r);
                w.t[row] = fmax(-g, WARM_FLOOR);
                w.lam[row] = fmax(w.lam[row], WARM_FLOOR);
            });
            __syncthreads();
            scales();
            mode = PRED;
        }
        while (true) {
            tid = SRH_TID;
            // ---------------- rows: weights D and gradient shifts rho for this Newton system
            if (mode != CORR) {
                rows_apply(d, c, w.x, w.s, w.u, w.rg);
                __syncthreads();
            }
            if (mode == INIT) {
                for_rows(d, [&](int row, bool isU, int k, int r) {
                    const double g = w.rg[row] - row_h(d, c, q, isU, k, r);
                    w.D[row] = d.ng ? 1.0 : 0.0; w.rho[row] = g; w.lam[row] = 0.0;
                });
            } else if (mode == PRED) {
                double musum = 0.0, rpm = 0.0;
                for_rows(d, [&](int row, bool isU, int k, int r) {
                    const double g = w.rg[row] - row_h(d, c, q, isU, k, r);
                    const double t = w.t[row], lam = w.lam[row];
                    const double rg = g + t;
                    w.rg[row] = rg;
                    const double D = lam / (t + dreg * lam);     // regularised weight (see oracle/riccati_ipm.py)
                    w.D[row] = D;
                    w.rho[row] = D * (rg + dreg * lam);
                    musum += lam * t;
                    rpm = fmax(rpm, fabs(rg));
                });
                mu = wg::reduce(musum, 0, L.red) / d.ng;
                rp = wg::reduce(rpm, 1, L.red);
            } else {
                for_rows(d, [&](int row, bool, int, int) {
                    const double t = w.t[row], lam = w.lam[row];
                    const double rc = lam * t + w.dt[row] * w.dlam[row] - sig * mu;
                    w.rc[row] = rc;
                    w.rho[row] = lam + (lam * w.rg[row] - rc) / (t + dreg * lam);
                });
            }
            __syncthreads();
            SRH_LAP(1);
            // ---------------- Newton system
            stage_prepass(d, c, q, w, mode == PRED);
            SRH_LAP(2);
            double rd = 0.0;
            const bool ok = riccati_solve<SPLIT, MSEL, NSEL>(d, c, dyn, w, L, mode != CORR, mode == PRED, &rd);
            if (mode == CORR) SRH_LAP(4); else SRH_LAP(3);
            // ---------------- use the direction
            if (mode == INIT) {
                if (!ok) { status = 2; break; }
                for (int e = tid; e < (N + 1) * n; e += nt) w.x[e] += w.dx[e];
                for (int e = tid; e < N * m; e += nt) w.u[e] += w.du[e];
                for (int e = tid; e <= N; e += nt) w.s[e] = (e == 0) ? s0 : w.s[e] + w.ds[e];
                __syncthreads();
                if (d.ng == 0) { status = 0; break; }      // no inequality rows: the Newton point is the solution
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
                scales();
                mode = PRED;
                continue;
            }
            if (mode == PRED) {
                // a factorisation that breaks down in the last digits of an already converged iterate
                // (weights D = lam/t up to 1e13) is accepted at the looser 1e-8 certificate
                if (!ok) { status = near_opt ? 0 : 2; break; }
                if (!(mu == mu)) { status = near_opt ? 0 : 5; break; }
                if (!(rd == rd)) { status = near_opt ? 0 : 6; break; }
                if (q.dbg && tid == 0) { gptr g = q.dbg + 8 * it; g[0] = mu; g[1] = rd; g[2] = rp; g[3] = sd; g[4] = sp; }
                const double ltol = fmax(d.tol, 1e-9);     // linear residuals: round-off floor (see the port)
                if (rd <= ltol * sd && rp <= ltol * sp && mu <= d.tol) { status = 0; break; }
                near_opt = (rd <= 1e-8 * sd && rp <= 1e-8 * sp && mu <= 1e-8);
                if (it >= d.max_iter) { status = 1; break; }
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
                sig = mu > 0.0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
                if (q.dbg && tid == 0) { gptr g = q.dbg + 8 * it; g[5] = a_aff; g[6] = sig; }
                mode = CORR;
                continue;
            }
            // mode == CORR: step
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
            if (q.dbg && tid == 0) { gptr g = q.dbg + 8 * it; g[7] = a; }
            for (int e = tid; e < (N + 1) * n; e += nt) w.x[e] += a * w.dx[e];
            for (int e = tid; e < N * m; e += nt) w.u[e] += a * w.du[e];
            for (int e = tid; e <= N; e += nt) w.s[e] = (e == 0) ? s0 : w.s[e] + a * w.ds[e];
            for_rows(d, [&](int row, bool, int, int) {
                w.t[row] += a * w.dt[row];
                w.lam[row] += a * w.dlam[row];
            });
            __syncthreads();
            ++it;
            mode = PRED;
        }
        SRH_LAP(1);
        if (warm_now && status != 0) { --pass; continue; }      // the warm attempt did not reach the tolerances: the same pass from the cold start
        // final consistency: x is exactly the rollout of u
        rollout(d, dyn, q, w.u, w.x, L);
        J = objective(d, c, q, w.x, w.u, w.s, L);
        SRH_LAP(5);
#ifdef SRH_PROFILE
        if (q.dbg && tid == 0) for (int i = 0; i < 6; ++i) q.dbg[8 * 62 + i] = (double)tm[i];
#endif
        if (npass == 2 && pass == 0) {
            if (status != 0) continue;               // relaxed QP not solved: go to the full QP
            double md = 0.0;
            for (int e = tid + n; e < (N + 1) * n; e += nt) md = fmax(md, fabs(c.xs[e % n] * (w.x[e] - q.xk[e])));
            md = wg::reduce(md, 1, L.red);
            if (md <= q.delta) {                     // trust region inactive: (x, u, s0 e_0) solves the full QP
                J += q.omega * s0;
                break;
            }
        }
    }
    if (J_out) *J_out = J;
    if (iters_out) *iters_out = it;
    return status;
}

}  // namespace qp
