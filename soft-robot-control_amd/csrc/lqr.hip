// Riccati recursions and iLQR on the device: one workgroup per problem, the cost-to-go matrix in LDS.
// Reference: sofacontrol/lqr/traj_tracking_lqr.py:18-48 (TV-LQR), sofacontrol/lqr/lqr.py:6-21 (fixed-point
// DARE), sofacontrol/lqr/ilqr.py:27-300 + sofacontrol/lqr/config.py (iLQR).
#include "tpwl_host.h"
#include "ssm_host.h"

namespace {

constexpr int NT = 512;

// ---- tiny dense helpers (row-major, any address space, runtime sizes; one output per thread-iteration)
// C (M x N) = alpha * op(A) * op(B) + beta * C0 ; op = transpose flag.  Ends with __syncthreads().
template <bool TA, bool TB, typename CP, typename AP, typename BP>
__device__ inline void mm(CP C, int ldc, AP A, int lda, BP B, int ldb, int M, int N, int K) {
    for (int e = SRH_TID; e < M * N; e += blockDim.x) {
        const int i = e / N, j = e - i * N;
        double acc = 0.0;
        int k = 0;
        // 8 independent operand pairs in flight per trip: with one or two waves per SIMD the dependent
        // load -> fma chain of a rolled loop is bound by the LDS / L2 latency of every single k
        for (; k + 8 <= K; k += 8) {
            double av[8], bv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                av[q] = TA ? A[(k + q) * lda + i] : A[i * lda + k + q];
                bv[q] = TB ? B[j * ldb + k + q] : B[(k + q) * ldb + j];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = fma(av[q], bv[q], acc);
        }
        for (; k < K; ++k) {
            const double a = TA ? A[k * lda + i] : A[i * lda + k];
            const double b = TB ? B[j * ldb + k] : B[k * ldb + j];
            acc = fma(a, b, acc);
        }
        C[i * ldc + j] = acc;
    }
    __syncthreads();
}

// Cholesky (thread 0) + solve helper for m <= 16 on LDS
__device__ inline bool chol16(lptr Q, lptr Lb, int m, liptr flag) { return wg::chol_factor(Q, Lb, m, flag, false); }

struct LqrLds {
    lptr P, W, T;          // n x n
    lptr PB, BK;           // n x m (PB), m x n (K / Qux)
    lptr Quu, Lc, Kt, Kk;  // m x m, m x m, m x n, m x n
    lptr v1, v2, v3, v4;   // n
    lptr u1, u2;           // m (16)
    lptr red;
    liptr flag;
};

// nn: doubles of each of the three square work matrices (n * n; 256 when the MFMA panels replace them)
__host__ __device__ inline size_t lqr_lds_doubles(int n, int m, size_t nn = 0) {
    if (nn == 0) nn = (size_t)n * n;
    return 3 * nn + 2 * (size_t)n * m + 2 * 256 + 2 * (size_t)m * n + 4 * (size_t)n + 32 + 16 + 4;
}

__device__ inline void lqr_carve(LqrLds &L, lptr base, int n, int m, size_t nn = 0) {
    if (nn == 0) nn = (size_t)n * n;
    lptr p = base;
    auto take = [&](size_t c) { lptr q = p; p += c; return q; };
    L.P = take(nn); L.W = take(nn); L.T = take(nn);
    L.PB = take((size_t)n * m); L.BK = take((size_t)n * m);
    L.Quu = take(256); L.Lc = take(256);
    L.Kt = take((size_t)m * n); L.Kk = take((size_t)m * n);
    L.v1 = take(n); L.v2 = take(n); L.v3 = take(n); L.v4 = take(n);
    L.u1 = take(16); L.u2 = take(16);
    L.red = take(16);
    L.flag = (liptr)take(4);
}

// K = -(R + B^T P B)^-1 B^T P A   into L.Kk (m x n); uses L.W = P A, L.PB = P B.  false if not PD.
template <typename AP, typename BP, typename RP>
__device__ inline bool lqr_gain(LqrLds &L, AP A, BP B, RP R, int n, int m) {
    mm<false, false>(L.W, n, L.P, n, A, n, n, n, n);        // W = P A
    mm<false, false>(L.PB, m, L.P, n, B, m, n, m, n);       // PB = P B
    mm<true, false>(L.Quu, m, B, m, L.PB, m, m, m, n);      // B^T P B
    for (int e = SRH_TID; e < m * m; e += blockDim.x) L.Quu[e] += R[e];
    mm<true, false>(L.Kt, n, B, m, L.W, n, m, n, n);        // B^T P A
    if (!chol16(L.Quu, L.Lc, m, L.flag)) return false;
    for (int j = SRH_TID; j < n; j += blockDim.x) wg::chol_solve_neg(L.Lc, m, L.Kt + j, n, L.Kk + j, n);
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------ TV-LQR (traj_tracking_lqr.py:18-48)
__global__ __launch_bounds__(NT) void tvlqr_kernel(const double *A, const double *B, const int *idx, int steps, int n,
                                                   int m, const double *Q, const double *R, double *K, double *P,
                                                   int *status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LqrLds L;
    lqr_carve(L, (lptr)smem, n, m);
    cgptr Ag = (cgptr)A, Bg = (cgptr)B, Qg = (cgptr)Q, Rg = (cgptr)R;
    gptr Kg = (gptr)K, Pg = (gptr)P;
    for (int e = SRH_TID; e < n * n; e += blockDim.x) { L.P[e] = Qg[e]; if (Pg) Pg[(size_t)steps * n * n + e] = Qg[e]; }
    __syncthreads();
    int st = 0;
    for (int i = steps - 1; i >= 0; --i) {
        const size_t sel = idx ? (size_t)idx[i] : (size_t)i;
        cgptr Ai = Ag + sel * n * n, Bi = Bg + sel * n * m;
        if (!lqr_gain(L, Ai, Bi, Rg, n, m)) { st = 2; break; }
        for (int e = SRH_TID; e < m * n; e += blockDim.x) Kg[(size_t)i * m * n + e] = L.Kk[e];
        // Acl = A + B K (into T) ; P = Q + K^T R K + Acl^T P Acl
        for (int e = SRH_TID; e < n * n; e += blockDim.x) {
            const int r = e / n, c = e - r * n;
            double v = Ai[e];
            for (int a = 0; a < m; ++a) v = fma(Bi[r * m + a], L.Kk[a * n + c], v);
            L.T[e] = v;
        }
        __syncthreads();
        mm<false, false>(L.W, n, L.P, n, L.T, n, n, n, n);          // W = P Acl
        mm<false, false>(L.Kt, n, Rg, m, L.Kk, n, m, n, m);         // R K
        // new P (reads T, W, Kk, Kt; writes P only)
        for (int e = SRH_TID; e < n * n; e += blockDim.x) {
            const int r = e / n, c = e - r * n;
            double v = Qg[e];
            for (int a = 0; a < m; ++a) v = fma(L.Kk[a * n + r], L.Kt[a * n + c], v);
            for (int k = 0; k < n; ++k) v = fma(L.T[k * n + r], L.W[k * n + c], v);
            L.P[e] = v;
            if (Pg) Pg[(size_t)i * n * n + e] = v;
        }
        __syncthreads();
    }
    if (SRH_TID == 0) *status = st;
}

// ------------------------------------------------------------------ fixed-point DARE (lqr.py:6-21)
__global__ __launch_bounds__(NT) void dare_fp_kernel(const double *A, const double *B, int n, int m, const double *Q,
                                                     const double *R, double tol, int max_iter, double *Lout,
                                                     double *Pout, int *iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LqrLds L;
    lqr_carve(L, (lptr)smem, n, m);
    const size_t p = blockIdx.x;
    cgptr Ag = (cgptr)A + p * n * n, Bg = (cgptr)B + p * n * m, Qg = (cgptr)Q, Rg = (cgptr)R;
    for (int e = SRH_TID; e < n * n; e += blockDim.x) L.P[e] = 0.0;
    __syncthreads();
    // L = solve(R + B'PB, B'PA) with P = 0 (no sign): zeros; L_old = inf
    bool first = true;
    int it = 0;
    // BK holds the previous gain (m x n) -- needs m*n <= n*m doubles
    for (int e = SRH_TID; e < m * n; e += blockDim.x) L.BK[e] = 0.0;
    __syncthreads();
    while (it < max_iter) {
        // gain of the current P:  Kk = -(R + B'PB)^-1 B'PA  ; W = P A, Kt = B'PA
        if (!lqr_gain(L, Ag, Bg, Rg, n, m)) break;
        if (!first) {
            double d2 = 0.0;
            for (int e = SRH_TID; e < m * n; e += blockDim.x) { const double d = L.Kk[e] - L.BK[e]; d2 = fma(d, d, d2); }
            d2 = wg::reduce(d2, 0, L.red);
            if (sqrt(d2) <= tol) break;
        }
        first = false;
        for (int e = SRH_TID; e < m * n; e += blockDim.x) L.BK[e] = L.Kk[e];
        // P <- A'PA - A'PB (R+B'PB)^-1 B'PA + Q = A'W + (B'PA)' Kk + Q
        for (int e = SRH_TID; e < n * n; e += blockDim.x) {
            const int r = e / n, c = e - r * n;
            double v = Qg[e];
            for (int k = 0; k < n; ++k) v = fma(Ag[k * n + r], L.W[k * n + c], v);
            for (int a = 0; a < m; ++a) v = fma(L.Kt[a * n + r], L.Kk[a * n + c], v);
            L.T[e] = v;
        }
        __syncthreads();
        for (int e = SRH_TID; e < n * n; e += blockDim.x) L.P[e] = L.T[e];
        __syncthreads();
        ++it;
    }
    for (int e = SRH_TID; e < m * n; e += blockDim.x) Lout[p * m * n + e] = L.Kk[e];
    for (int e = SRH_TID; e < n * n; e += blockDim.x) Pout[p * n * n + e] = L.P[e];
    if (SRH_TID == 0 && iters) iters[p] = it;
}

// ------------------------------------------------------------------ DARE by structure-preserving doubling
// lqr.py:24-31 (`dare`: scipy.linalg.solve_discrete_are in the reference; the scp controller calls it for every TPWL
// point at start-up, tpwl/controllers.py:238-246).  A fixed-point Riccati iteration converges like rho(A_cl)^2k and
// crawls on lightly damped points; the doubling algorithm (SDA) squares the closed-loop transition per step:
//   G = B R^-1 B^T, H = Q;   W = I + G H,  [V1 V2] = W^-1 [A G]
//   A <- A V1,   G <- G + A V2 A^T,   H <- H + A^T (H V1)            ->   H converges quadratically to P
// (11-12 steps at the Diamond size where the fixed point needs ~900).  One workgroup per (A, B) pair; five n x n
// slots (W / scratch, A -> V1, G -> V2, H, A_next) in LDS when they fit (n <= 62), else in the per-problem HBM
// workspace next to the copies of A_k, G_k that the products read through L2; W^-1 by Gauss-Jordan elimination with
// partial pivoting (physical row swaps) on the tableau [W | A | G]; all products on the VALU (mm), generic pointers.
struct SdaTail {
    lptr Rq, Lc, Bt, Yn;     // R (m x m), its Cholesky factor, B^T (m x n), -R^-1 B^T (m x n)
    lptr fcol, prow, jrow;   // Gauss-Jordan: multipliers (n), pivot row (3n), old row j (3n)
    lptr red;
    liptr flag, ipiv;
};

__host__ __device__ inline size_t sda_tail_doubles(int n, int m) { return 512 + 2 * (size_t)m * n + 7 * (size_t)n + 16 + 8; }

__global__ __launch_bounds__(NT) void dare_sda_kernel(const double *A, const double *B, int n, int m, const double *Q,
                                                      const double *R, double tol, int max_iter, double *work,
                                                      int lds_slots, double *Lout, double *Pout, int *iters,
                                                      int *status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const size_t p = blockIdx.x;
    const int ld = n | 1, tid = SRH_TID, nt = blockDim.x;
    const size_t nn = (size_t)n * ld;
    double *wk = work + p * (7 * nn);
    double *gA = wk, *gG = wk + nn;
    double *sm = (double *)smem;
    double *S1, *S2, *S3, *S4, *S5;
    lptr tail;
    if (lds_slots) {
        S1 = sm; S2 = sm + nn; S3 = sm + 2 * nn; S4 = sm + 3 * nn; S5 = sm + 4 * nn;
        tail = (lptr)smem + 5 * nn;
    } else {
        S1 = wk + 2 * nn; S2 = wk + 3 * nn; S3 = wk + 4 * nn; S4 = wk + 5 * nn; S5 = wk + 6 * nn;
        tail = (lptr)smem;
    }
    SdaTail T;
    {
        lptr q = tail;
        auto take = [&](size_t c) { lptr r0 = q; q += c; return r0; };
        T.Rq = take(256); T.Lc = take(256); T.Bt = take((size_t)m * n); T.Yn = take((size_t)m * n);
        T.fcol = take(n); T.prow = take(3 * (size_t)n); T.jrow = take(3 * (size_t)n); T.red = take(16);
        T.flag = (liptr)take(4); T.ipiv = (liptr)take(4);
    }
    cgptr Ag = (cgptr)A + p * n * n, Bg = (cgptr)B + p * n * m, Qg = (cgptr)Q, Rg = (cgptr)R;
    int st = 0, it = 0;

    // ---- G0 = B R^-1 B^T, H0 = Q, A0 = A
    for (int e = tid; e < m * m; e += nt) T.Rq[e] = Rg[e];
    for (int e = tid; e < m * n; e += nt) T.Bt[e] = Bg[(e % n) * m + e / n];
    __syncthreads();
    if (!wg::chol_factor(T.Rq, T.Lc, m, T.flag, false)) st = 2;
    if (st == 0) {
        for (int j = tid; j < n; j += nt) wg::chol_solve_neg(T.Lc, m, T.Bt + j, n, T.Yn + j, n);
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) {
            const int r = e / n, c = e - r * n;
            double g = 0.0;
            for (int a = 0; a < m; ++a) g = fma(-T.Bt[a * n + r], T.Yn[a * n + c], g);
            S3[r * ld + c] = g; gG[r * ld + c] = g;
            const double av = Ag[e];
            S2[r * ld + c] = av; gA[r * ld + c] = av;
            S4[r * ld + c] = Qg[e];
        }
        __syncthreads();
    }
    while (st == 0 && it < max_iter) {
        // W = I + G H
        mm<false, false>(S1, ld, S3, ld, S4, ld, n, n, n);
        for (int e = tid; e < n; e += nt) S1[e * ld + e] += 1.0;
        __syncthreads();
        // [V1 V2] = W^-1 [A G]: Gauss-Jordan with partial pivoting on [S1 | S2 | S3]
        for (int j = 0; j < n && st == 0; ++j) {
            if (tid < 64) {
                double best = -1.0;
                int bi = j;
                for (int i = j + tid; i < n; i += 64) {
                    const double v = fabs(S1[i * ld + j]);
                    if (v > best) { best = v; bi = i; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const double ob = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
                if (tid == 0) { T.ipiv[0] = bi; T.ipiv[1] = (best > 1e-300 && best < 1e300) ? 1 : 0; }
            }
            __syncthreads();
            const int pv = T.ipiv[0];
            if (T.ipiv[1] == 0) { st = 3; break; }
            // snapshot: pivot row (old row pv), old row j, multipliers of every row as they will sit after the swap
            for (int c = tid; c < 3 * n; c += nt) {
                double *blk = c < n ? S1 : (c < 2 * n ? S2 : S3);
                const int cc = c < n ? c : (c < 2 * n ? c - n : c - 2 * n);
                T.prow[c] = blk[pv * ld + cc];
                T.jrow[c] = blk[j * ld + cc];
            }
            for (int i = tid; i < n; i += nt) T.fcol[i] = S1[(i == pv ? j : i) * ld + j];
            __syncthreads();
            const double rp = 1.0 / T.prow[j];
            for (int e = tid; e < 3 * n * n; e += nt) {
                const int i = e / (3 * n), c = e - i * 3 * n;
                double *blk = c < n ? S1 : (c < 2 * n ? S2 : S3);
                const int cc = c < n ? c : (c < 2 * n ? c - n : c - 2 * n);
                const double pr = T.prow[c] * rp;
                double v;
                if (i == j) v = pr;
                else {
                    const double src = (i == pv) ? T.jrow[c] : blk[i * ld + cc];
                    v = fma(-T.fcol[i], pr, src);
                }
                blk[i * ld + cc] = v;
            }
            __syncthreads();
        }
        if (st != 0) break;
        mm<false, false>(S5, ld, gA, ld, S2, ld, n, n, n);            // A_next = A V1
        mm<false, false>(S1, ld, gA, ld, S3, ld, n, n, n);            // T2 = A V2
        mm<false, true>(S3, ld, S1, ld, gA, ld, n, n, n);             // T2 A^T  (V2 is dead)
        for (int e = tid; e < n * n; e += nt) { const int r = e / n, c = e - r * n; S3[r * ld + c] += gG[r * ld + c]; }
        mm<false, false>(S1, ld, S4, ld, S2, ld, n, n, n);            // T3 = H V1
        mm<true, false>(S2, ld, gA, ld, S1, ld, n, n, n);             // A^T T3  (V1 is dead)
        double dmax = 0.0, hmax = 0.0;
        for (int e = tid; e < n * n; e += nt) {
            const int r = e / n, c = e - r * n;
            const double d = S2[r * ld + c], h = S4[r * ld + c] + d;
            S4[r * ld + c] = h;
            dmax = fmax(dmax, fabs(d)); hmax = fmax(hmax, fabs(h));
            const double an = S5[r * ld + c];
            S2[r * ld + c] = an; gA[r * ld + c] = an;
            gG[r * ld + c] = S3[r * ld + c];
        }
        dmax = wg::reduce(dmax, 1, T.red);
        hmax = wg::reduce(hmax, 1, T.red);
        __syncthreads();
        ++it;
        if (!(dmax == dmax) || !(hmax < 1e300)) { st = 3; break; }
        if (dmax <= tol * hmax) break;
    }
    if (st == 0 && it >= max_iter) st = 1;
    for (int e = tid; e < n * n; e += nt) { const int r = e / n, c = e - r * n; Pout[p * n * n + e] = S4[r * ld + c]; }
    __syncthreads();
    // gain K = -(R + B^T P B)^-1 B^T P A from the converged P (the fixed-point kernel's routine, its own LDS carve)
    LqrLds L;
    lqr_carve(L, (lptr)smem, n, m);
    for (int e = tid; e < n * n; e += nt) L.P[e] = Pout[p * n * n + e];
    __syncthreads();
    if (!lqr_gain(L, Ag, Bg, Rg, n, m)) { if (st == 0) st = 2; }
    for (int e = tid; e < m * n; e += nt) Lout[p * m * n + e] = L.Kk[e];
    if (tid == 0) { if (iters) iters[p] = it; status[p] = st; }
}

// K = -Q~uu^-1 Q~ux (columns of L.BK -> L.Kk), k = -Q~uu^-1 Q_u (L.u2 -> L.u1) with the Cholesky factor of the
// tiny Q~uu recomputed in registers by every thread (no serial thread-0 phase, no barrier before the solves).
// Returns false (uniformly) if Q~uu is not positive definite.
template <int M>
__device__ __forceinline__ bool ilqr_gain_t(LqrLds &L, int n) {
    double Lr[M * M], inv[M];
    if (!wg::chol_reg<M>(L.Quu, M, 0.0, Lr, inv)) return false;
    for (int j = SRH_TID; j <= n; j += blockDim.x) {
        if (j < n) wg::chol_solve_neg_reg<M>(Lr, inv, L.BK + j, n, L.Kk + j, n);
        else wg::chol_solve_neg_reg<M>(Lr, inv, L.u2, 1, L.u1, 1);
    }
    __syncthreads();
    return true;
}

__device__ __forceinline__ bool ilqr_gain(LqrLds &L, int n, int m) {
    switch (m) {
        case 1: return ilqr_gain_t<1>(L, n);
        case 2: return ilqr_gain_t<2>(L, n);
        case 3: return ilqr_gain_t<3>(L, n);
        case 4: return ilqr_gain_t<4>(L, n);
        case 5: return ilqr_gain_t<5>(L, n);
        case 6: return ilqr_gain_t<6>(L, n);
        case 7: return ilqr_gain_t<7>(L, n);
        case 8: return ilqr_gain_t<8>(L, n);
        default: break;
    }
    if (!chol16(L.Quu, L.Lc, m, L.flag)) return false;
    for (int j = SRH_TID; j <= n; j += blockDim.x) {
        if (j < n) wg::chol_solve_neg(L.Lc, m, L.BK + j, n, L.Kk + j, n);
        else wg::chol_solve_neg(L.Lc, m, L.u2, 1, L.u1, 1);
    }
    __syncthreads();
    return true;
}

// ------------------------------------------------------------------ iLQR (ilqr.py:27-300)
struct IlqrArgs {
    int N, n, m, nz;
    silqr_params par;
    const double *x0, *z_target, *u_warm, *u_last, *Q, *R, *Qf;
    double *x, *u, *K, *cost;
    int *iters;
    double *work;          // per problem: x2 (N+1)n, u2 N m, kff N m, Qu N m, Quu N m m, K2 N m n
    int *iwork;            // per problem: idx N, idx2 N
    size_t work_stride;
    double *lin;           // SSM model only, per problem: 2 x N x (n n + n m + n) per-step (A, B, d)
    int ssm_mode;          // SSM model only: discretisation mode (ssm_dev.h)
    int jl_cap;            // SSM model only: slots per compact derivative list (ssm::jacobian_list_cap; 0: dense derivative table)
    int stage_ab;          // 1: the backward pass stages (A_t, B_t) in LDS
    size_t panel_off;      // doubles from the start of LDS to the MFMA panels
    int mfma;              // 1: backward pass on f64 MFMA products over padded LDS panels (n_x + n_u panels fit LDS)
    double dt;
};

// MFMA product for the backward pass when the [A_t | B_t] panel does not fit LDS (64 < n_x: the shipped Diamond basis,
// n_x = 72): C[i][j] = sum_k L(k, i) R(k, j) like wg::mfma_atb, but an operand is either an LDS panel or the pair
// (A_t, B_t) read in place from the L2-resident tables -- AB(k, c) = A[k][c] for c < n, B[k][c - n] for c < n + m, else 0,
// rows k >= n zero; `coff` shifts the columns (coff = n: the B block).  Rows >= vrows of C are stored as zeros, rows
// >= srows not at all.  Six k-steps of operands are requested before their MFMAs.
struct IlqrOp {
    clptr p; int ld;                 // LDS panel (p != null)
    cgptr A, B; int n, m, coff;      // or the stage's Jacobians in HBM / L2
    __device__ __forceinline__ double get(int k, int c) const {
        if (p) return p[(size_t)k * ld + c];
        const int cc = c + coff;
        if (k >= n) return 0.0;
        return cc < n ? A[(size_t)k * n + cc] : (cc < n + m ? B[(size_t)k * m + cc - n] : 0.0);
    }
};

__device__ __forceinline__ void ilqr_mm(lptr C, int ldc, const IlqrOp &Lo, const IlqrOp &Ro, int K, int MT, int NTl, int vrows,
                                        int srows) {
    const int wave = __builtin_amdgcn_readfirstlane(SRH_TID >> 6), lane = SRH_TID & 63, nw = blockDim.x >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    for (int t = wave; t < MT * NTl; t += nw) {
        const int ti = t / NTl, tj = t - ti * NTl;
        const int ci = 16 * ti + l16, cj = 16 * tj + l16;
        wg::qp_d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < K; k0 += 24) {
            double av[6], bv[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const bool in = k0 + 4 * q < K;
                const int k = in ? k0 + 4 * q + kk : 0;
                av[q] = in ? Lo.get(k, ci) : 0.0;
                bv[q] = in ? Ro.get(k, cj) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (k0 + 4 * q < K) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 16 * ti + kk + 4 * q;
            if (i < srows) C[(size_t)i * ldc + cj] = i < vrows ? acc[q] : 0.0;
        }
    }
    __syncthreads();
}

// MODEL 0: prediscretised nearest-neighbour TPWL tables (T); MODEL 1: SSM polynomial model (S), linearised
// and discretised at every step of the forward pass (ilqr.py:155: model.get_jacobians(x[t], u=u[t], dt)).
// NSEL / MSEL > 0: instantiation for exactly that n_x / n_u (compile-time extents: the index arithmetic and the small
// loops of the passes fold); 0: any size.
// NTH: threads per workgroup = per problem.  512 (NT) for the robot-sized models; 64 for small ones (n_x + n_u <= 32: the C3 SSM
// shape, n_x = 10, n_u = 8): a 10 x 10 problem has no work for eight waves, every phase boundary is a workgroup barrier that seven
// of them spend waiting, and ONE wave needs no barrier at all (its LDS queue is in order; s_barrier of a one-wave workgroup is
// free).  Same code, same per-element arithmetic in the same order: the wave-specialised phases (`tid >= O1`) run one after the
// other on the single wave, the only workgroup reduction whose tree depends on the wave count (the expected cost decrease) sums
// per 64 stages in stage order in both forms.
template <int MODEL, int NSEL, int MSEL, int NTH = NT>
__global__ __launch_bounds__(NTH) void ilqr_kernel(TpwlDev T, SsmDev S, IlqrArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int O1 = NTH > 64 ? 64 : 0, O2 = NTH > 64 ? 128 : 0;      // first thread of the second / third wave-specialised phase
    const int N = a.N, n = NSEL > 0 ? NSEL : a.n, m = MSEL > 0 ? MSEL : a.m, nz = a.nz;
    LqrLds L;
    lqr_carve(L, (lptr)smem, n, m, a.mfma ? 256 : 0);
    lptr QH = L.red + 20;                     // placed after the carve: Q H (nz x n); c_xx = H^T (Q H) on the fly
    lptr zt = QH + (size_t)16 * n;            // nz scratch (16)
    lptr part = zt + 16;                      // blockDim
    // SSM scratch (MODEL 1): polynomial work space, per-step (A, B, d), observed output
    ssm::Work sw;
    lptr zs = part + NTH, xl = zs + 16;        // observed output, a state vector (both models)
    lptr Al = xl + n, Bl = Al + (size_t)n * n, dl = Bl + (size_t)n * m;      // MODEL 1 only (not allocated for MODEL 0)
    SsmLds ST;
    if constexpr (MODEL == 1) {
        ssm::carve(sw, dl + n, S);
        // coefficient rows and exponent tables of the polynomial model into LDS, once per kernel
        ssm::stage(ST, dl + n + ssm::work_doubles(S.n, S.m, S.no, S.nr, S.ns), S, a.ssm_mode == SSM_DISCRETE_MAP, a.jl_cap);
    }
    // MFMA backward pass: padded panels P / G, [A|B], W = P [A|B] (NPa x ld each) and B^T [A|B] (16 x ld)
    const int NPa = (n + m + 15) & ~15, ldp = NPa + 1, NK4 = (n + 3) & ~3, n16 = (n + 15) & ~15;
    // mfma == 1: P / G, [A|B], W panels of NPa rows.  mfma == 2 (they do not fit: n_x > 64): P / G with n + m rows, W with
    // NK4 rows, no [A|B] panel -- the products read the stage's Jacobians in place (ilqr_mm)
    const bool abg = a.mfma == 2;
    const int prow = abg ? n + m : NPa;
    lptr Pm = (lptr)smem + a.panel_off, ABm = Pm + (size_t)prow * ldp, Wm = abg ? ABm : ABm + (size_t)NPa * ldp,
         RBm = Wm + (size_t)(abg ? NK4 : NPa) * ldp;
    lptr Hl = RBm + (size_t)16 * ldp, qz = Hl + (size_t)16 * n;      // H (nz x n) and Q (z - z*) in LDS (MFMA path)
    cgptr Hm = MODEL == 0 ? T.H : S.H;
    cgptr zref = MODEL == 0 ? T.z_ref : S.z_ref;
    const size_t lstride = (size_t)n * n + (size_t)n * m + n;
    const size_t p = blockIdx.x;
    int tid = SRH_TID;                                  // re-read at the top of every stage of the passes (dev_la.h: SRH_TID)
    const int nt = blockDim.x;
    cgptr x0 = (cgptr)a.x0 + p * n, ztar = (cgptr)a.z_target + p * (size_t)(N + 1) * nz;
    cgptr Qg = (cgptr)a.Q, Rg = (cgptr)a.R, Qfg = (cgptr)a.Qf;
    gptr X = (gptr)a.x + p * (size_t)(N + 1) * n, U = (gptr)a.u + p * (size_t)N * m;
    gptr Kout = (gptr)a.K + p * (size_t)N * m * n;
    gptr wk = (gptr)a.work + p * a.work_stride;
    gptr X2 = wk, U2 = X2 + (size_t)(N + 1) * n, kff = U2 + (size_t)N * m, Qu = kff + (size_t)N * m;
    gptr Quu = Qu + (size_t)N * m, K2 = Quu + (size_t)N * m * m;
    // SSM model: z(x_t) - z*_t of the accepted trajectory (ZC) and of the forward pass in flight (ZC2), (N + 1) x nz each -- the
    // backward pass takes them from here instead of evaluating the output polynomial of every x_t a second time (same values:
    // the forward pass has just computed them from the same states)
    gptr ZC = K2 + (size_t)N * m * n, ZC2 = ZC + (size_t)(N + 1) * nz;
    giptr idx = (giptr)a.iwork + p * 2 * (size_t)N, idx2 = idx + N;
    const silqr_params &P_ = a.par;
    gptr lin = MODEL == 1 ? (gptr)a.lin + p * 2 * (size_t)N * lstride : (gptr) nullptr;
    gptr lin2 = MODEL == 1 ? lin + (size_t)N * lstride : (gptr) nullptr;

    // z(x) - z*_t into zt for the state xq (LDS); all threads, ends with a sync
    auto zerr = [&](clptr xq, int t) {
        if constexpr (MODEL == 0) {
            if (tid < nz) {
                double v = zref[tid] - ztar[(size_t)t * nz + tid];
                for (int j = 0; j < n; ++j) v = fma(Hm[tid * n + j], xq[j], v);
                zt[tid] = v;
            }
            __syncthreads();
        } else {
            ssm::observe_l(S, ST, xq, sw, zs);
            if (tid < nz) zt[tid] = zs[tid] + zref[tid] - ztar[(size_t)t * nz + tid];
            __syncthreads();
        }
    };

    // forward pass (ilqr.py:117-162): from (xp, up) with gains (Kg, kg, alpha) into (xo, uo, io); returns cost
#ifdef SRH_PROFILE
    long long dprof[6] = {0, 0, 0, 0, 0, 0};
    sw.prof = dprof;
    long long fp[6] = {0, 0, 0, 0, 0, 0};
    long long fpl = 0;
#define FP_T0() fpl = clock64()
#define FP_LAP(i) do { const long long now_ = clock64(); fp[i] += now_ - fpl; fpl = now_; } while (0)
#else
#define FP_T0() ((void)0)
#define FP_LAP(i) ((void)0)
#endif
    auto forward = [&](cgptr xp, cgptr up, double alpha, cgptr Kg, cgptr kg, gptr xo, gptr uo, giptr io,
                       gptr lo, gptr zco) -> double {
        double cost = 0.0;
        for (int e = tid; e < n; e += nt) { L.v1[e] = x0[e]; xo[e] = x0[e]; }
        __syncthreads();
        for (int t = 0; t < N; ++t) {
            tid = SRH_TID;
            FP_T0();
            // u_t = u_prev + alpha k + K (x - x_prev)
            if (tid < m) {
                double v = up[(size_t)t * m + tid];
                if (kg) v += alpha * kg[(size_t)t * m + tid];
                if (Kg) for (int j = 0; j < n; ++j) v = fma(Kg[((size_t)t * m + tid) * n + j], L.v1[j] - xp[(size_t)t * n + j], v);
                L.u1[tid] = v;
                uo[(size_t)t * m + tid] = v;
            }
            if constexpr (MODEL == 0) {
                if (tid >= O1 && tid < O1 + 64) {
                    const int i = tpwl::nearest_wave(T, (clptr)L.v1);
                    if (tid == O1) { io[t] = i; *L.flag = i; }
                }
                // z - z*  (z = H x + z_ref)
                if (tid >= O2 && tid < O2 + nz) {
                    const int r = tid - O2;
                    double v = zref[r] - ztar[(size_t)t * nz + r];
                    for (int j = 0; j < n; ++j) v = fma(Hm[r * n + j], L.v1[j], v);
                    zt[r] = v;
                }
                __syncthreads();
            } else {
                __syncthreads();
                FP_LAP(0);
                zerr((clptr)L.v1, t);
                if (tid < nz) zco[(size_t)t * nz + tid] = zt[tid];
                FP_LAP(1);
                ssm::jacobians_l(S, ST, a.ssm_mode == SSM_DISCRETE_MAP, (clptr)L.v1, (clptr)L.u1, sw, Al, n, Bl, dl);
                FP_LAP(2);
                ssm::discretize(S, a.ssm_mode, a.dt, sw, Al, n, Bl, dl);
                FP_LAP(3);
            }
            if (tid < 64) {          // step cost (ilqr.py:168-176): the terms of both quadratic forms over the lanes of wave 0
                auto dui = [&](int r) {
                    if (!P_.include_input_var_constraint) return (double)L.u1[r];            // u' R u (ilqr.py:150-152)
                    return L.u1[r] - (t == 0 ? (a.u_last ? a.u_last[p * m + r] : 0.0) : uo[(size_t)(t - 1) * m + r]);
                };
                double c = 0.0;
                for (int e = tid; e < nz * nz + m * m; e += 64) {
                    if (e < nz * nz) {
                        c = fma(zt[e / nz] * Qg[e], zt[e % nz], c);
                    } else {
                        const int f = e - nz * nz;
                        c = fma(dui(f / m) * Rg[f], dui(f % m), c);
                    }
                }
                c = wg::wave_sum(c);
                cost += 0.5 * c;     // every lane of wave 0 carries the running cost; thread 0 publishes it
            }
            if constexpr (MODEL == 0) {
                const size_t i = (size_t)*L.flag;
                wg::matTvec(L.v2, T.AdT + i * n * n, n, n, n, (clptr)L.v1, T.dd + i * n, part);
                wg::matTvec(L.v2, T.BdT + i * m * n, n, m, n, (clptr)L.u1, (clptr)L.v2, part);
            } else {
                gptr lt = lo + (size_t)t * lstride;
                for (int e = tid; e < n * n; e += nt) lt[e] = Al[e];
                for (int e = tid; e < n * m; e += nt) lt[(size_t)n * n + e] = Bl[e];
                for (int e = tid; e < n; e += nt) lt[(size_t)n * n + (size_t)n * m + e] = dl[e];
                for (int i = tid; i < n; i += nt) {
                    double ax = 0.0, bu = 0.0;
                    for (int j = 0; j < n; ++j) ax = fma(Al[i * n + j], L.v1[j], ax);
                    for (int j = 0; j < m; ++j) bu = fma(Bl[i * m + j], L.u1[j], bu);
                    L.v2[i] = ax + bu + dl[i];
                }
                __syncthreads();
            }
            for (int e = tid; e < n; e += nt) { L.v1[e] = L.v2[e]; xo[(size_t)(t + 1) * n + e] = L.v2[e]; }
            __syncthreads();
            FP_LAP(4);
        }
        zerr((clptr)L.v1, N);
        if constexpr (MODEL == 1) { if (tid < nz) zco[(size_t)N * nz + tid] = zt[tid]; }
        if (tid < 64) {
            double c = 0.0;
            for (int e = tid; e < nz * nz; e += 64) c = fma(zt[e / nz] * Qfg[e], zt[e % nz], c);
            c = wg::wave_sum(c);
            cost += 0.5 * c;
            if (tid == 0) L.red[15] = cost;
        }
        __syncthreads();
        cost = L.red[15];
        __syncthreads();
        return cost;
    };

    double rho = P_.rho0, drho = P_.drho0;
    auto reg_update = [&](bool increase) {       // ilqr.py:198-217 (decrease keeps drho: the reference's typo)
        if (increase) {
            drho = fmax(drho * P_.rho_scaling, P_.rho_scaling);
            rho = fmax(rho * drho, P_.rho_min);
            if (rho > P_.rho_max) rho = P_.rho_max;
        } else {
            const double dh = fmin(drho / P_.rho_scaling, 1.0 / P_.rho_scaling);
            rho = rho * dh;
            if (rho <= P_.rho_min) rho = P_.rho_min;
        }
    };

    // The reference restarts the backward pass with a larger rho for as long as Q~_uu is not positive definite
    // (ilqr.py:276-287) -- for ever when the trajectory has gone non-finite (an unstable discretisation: every Cholesky
    // of NaNs fails).  On the device that would hang the GPU: rho saturates at rho_max after ~30 increases, so after
    // MAX_RESTARTS consecutive failures of one backward pass the problem is given up (iters = -1, cost as it stands).
    constexpr int MAX_RESTARTS = 100;
    int restarts = 0;
    bool diverged = false;
    // c_xx = H^T Q H (constant), terminal uses Qf
    // backward pass (ilqr.py:219-300) on the trajectory (X, U, idx); writes K2? no: Kout, kff, Qu, Quu
    auto backward = [&]() {
        while (true) {
            const double rho_b = P_.regularize ? rho : 0.0;            // config.py:9 / ilqr.py:264-274
            const bool sreg = P_.state_regularization != 0;            // config.py:31: rho (B'B, B'A) or rho I
            // terminal: p = H^T Qf (z - z*), P = H^T Qf H
            for (int e = tid; e < n; e += nt) xl[e] = X[(size_t)N * n + e];
            __syncthreads();
            zerr((clptr)xl, N);
            for (int e = tid; e < n * n; e += nt) {
                const int r = e / n, c = e - r * n;
                double v = 0.0;
                for (int s = 0; s < nz; ++s) {
                    double q1 = 0.0;
                    for (int s2 = 0; s2 < nz; ++s2) q1 = fma(Qfg[s * nz + s2], Hm[s2 * n + c], q1);
                    v = fma(Hm[s * n + r], q1, v);
                }
                L.P[e] = v;
            }
            for (int e = tid; e < nz * n; e += nt) {
                const int s = e / n, c = e - s * n;
                double q2 = 0.0;
                for (int s2 = 0; s2 < nz; ++s2) q2 = fma(Qg[s * nz + s2], Hm[s2 * n + c], q2);
                QH[e] = q2;
            }
            for (int e = tid; e < n; e += nt) {
                double v = 0.0;
                for (int s = 0; s < nz; ++s) { double q1 = 0.0; for (int s2 = 0; s2 < nz; ++s2) q1 = fma(Qfg[s * nz + s2], zt[s2], q1); v = fma(Hm[s * n + e], q1, v); }
                L.v1[e] = v;     // p
            }
            __syncthreads();
            // one backward stage on (A_t, B_t) -- in LDS when the staging panels fit (a.stage_ab), else straight from
            // the HBM tables / per-step linearisations; returns false when Q~_uu is not positive definite
            auto stage = [&](auto At, auto Bt, int t) -> bool {
                // c_x = H^T Q (z - z*), c_u = R (u_t - u_{t-1})
                for (int e = tid; e < n; e += nt) xl[e] = X[(size_t)t * n + e];
                __syncthreads();
                zerr((clptr)xl, t);
                if (tid >= O1 && tid < O1 + m) {
                    const int r = tid - O1;
                    double v = 0.0;
                    for (int s = 0; s < m; ++s) {
                        const double du = U[(size_t)t * m + s] - (!P_.include_input_var_constraint ? 0.0 :
                                          (t == 0 ? (a.u_last ? a.u_last[p * m + s] : 0.0) : U[(size_t)(t - 1) * m + s]));
                        v = fma(Rg[r * m + s], du, v);
                    }
                    L.u1[r] = v;        // c_u
                }
                __syncthreads();
                mm<false, false>(L.W, n, L.P, n, At, n, n, n, n);        // P A
                mm<false, false>(L.PB, m, L.P, n, Bt, m, n, m, n);       // P B
                // Q_uu = R + B'PB ; Q~_uu = Q_uu + rho B'B ; Q_ux = B'PA ; Q~_ux = Q_ux + rho B'A
                for (int e = tid; e < m * m; e += nt) {
                    const int r = e / m, c = e - r * m;
                    double v = Rg[e], bb = 0.0;
                    for (int k = 0; k < n; ++k) { v = fma(Bt[k * m + r], L.PB[k * m + c], v); bb = fma(Bt[k * m + r], Bt[k * m + c], bb); }
                    Quu[(size_t)t * m * m + e] = v;
                    L.Quu[e] = v + (sreg ? rho_b * bb : (r == c ? rho_b : 0.0));
                }
                for (int e = tid; e < m * n; e += nt) {
                    const int r = e / n, c = e - r * n;
                    double v = 0.0, ba = 0.0;
                    for (int k = 0; k < n; ++k) { v = fma(Bt[k * m + r], L.W[k * n + c], v); ba = fma(Bt[k * m + r], At[k * n + c], ba); }
                    L.Kt[e] = v;                      // Q_ux
                    L.BK[e] = v + (sreg ? rho_b * ba : 0.0);           // Q~_ux
                }
                // Q_x = c_x + A'p ; Q_u = c_u + B'p
                for (int e = tid; e < n + m; e += nt) {
                    if (e < n) {
                        double v = 0.0;
                        for (int s = 0; s < nz; ++s) { double q1 = 0.0; for (int s2 = 0; s2 < nz; ++s2) q1 = fma(Qg[s * nz + s2], zt[s2], q1); v = fma(Hm[s * n + e], q1, v); }
                        for (int k = 0; k < n; ++k) v = fma(At[k * n + e], L.v1[k], v);
                        L.v2[e] = v;
                    } else {
                        const int r = e - n;
                        double v = L.u1[r];
                        for (int k = 0; k < n; ++k) v = fma(Bt[k * m + r], L.v1[k], v);
                        L.u2[r] = v;
                        Qu[(size_t)t * m + r] = v;
                    }
                }
                __syncthreads();
                if (!chol16(L.Quu, L.Lc, m, L.flag)) {            // not PD: raise rho, restart (ilqr.py:276-287)
                    reg_update(true);
                    return false;
                }
                for (int j = tid; j <= n; j += nt) {
                    if (j < n) wg::chol_solve_neg(L.Lc, m, L.BK + j, n, L.Kk + j, n);
                    else wg::chol_solve_neg(L.Lc, m, L.u2, 1, L.u1, 1);           // k (feed-forward) in u1
                }
                __syncthreads();
                for (int e = tid; e < m * n; e += nt) Kout[(size_t)t * m * n + e] = L.Kk[e];
                if (tid < m) kff[(size_t)t * m + tid] = L.u1[tid];
                // p = Q_x + K'Quu k + K'Q_u + Q_ux' k ; P = Q_xx + K'Quu K + K'Q_ux + Q_ux'K
                // QK = Quu K (m x n) into PB region (m*n <= n*m)
                for (int e = tid; e < m * n; e += nt) {
                    const int r = e / n, c = e - r * n;
                    double v = 0.0;
                    for (int s = 0; s < m; ++s) v = fma(Quu[(size_t)t * m * m + r * m + s], L.Kk[s * n + c], v);
                    L.PB[e] = v;
                }
                if (tid < m) {
                    double v = 0.0;
                    for (int s = 0; s < m; ++s) v = fma(Quu[(size_t)t * m * m + tid * m + s], L.u1[s], v);
                    zt[tid] = v;        // Quu k   (zt has 16 slots)
                }
                __syncthreads();
                for (int e = tid; e < n * n; e += nt) {
                    const int r = e / n, c = e - r * n;
                    double v = 0.0;
                    for (int s = 0; s < nz; ++s) v = fma(Hm[s * n + r], QH[s * n + c], v);
                    {
                        int k = 0;
                        for (; k + 8 <= n; k += 8) {
                            double av[8], bv[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) { av[q] = At[(k + q) * n + r]; bv[q] = L.W[(k + q) * n + c]; }
#pragma unroll
                            for (int q = 0; q < 8; ++q) v = fma(av[q], bv[q], v);
                        }
                        for (; k < n; ++k) v = fma(At[k * n + r], L.W[k * n + c], v);
                    }
                    for (int s = 0; s < m; ++s) {
                        v = fma(L.Kk[s * n + r], L.PB[s * n + c], v);
                        v = fma(L.Kk[s * n + r], L.Kt[s * n + c], v);
                        v = fma(L.Kt[s * n + r], L.Kk[s * n + c], v);
                    }
                    L.T[e] = v;
                }
                for (int e = tid; e < n; e += nt) {
                    double v = L.v2[e];
                    for (int s = 0; s < m; ++s) {
                        v = fma(L.Kk[s * n + e], zt[s], v);
                        v = fma(L.Kk[s * n + e], L.u2[s], v);
                        v = fma(L.Kt[s * n + e], L.u1[s], v);
                    }
                    L.v3[e] = v;
                }
                __syncthreads();
                for (int e = tid; e < n * n; e += nt) L.P[e] = L.T[e];
                for (int e = tid; e < n; e += nt) L.v1[e] = L.v3[e];
                __syncthreads();
                return true;
            };
            bool restart = false;
            for (int t = N - 1; t >= 0; --t) {
                tid = SRH_TID;
                cgptr Ag, Bg;
                if constexpr (MODEL == 0) {
                    const size_t i = (size_t)idx[t];
                    Ag = T.Ad + i * n * n; Bg = T.Bd + i * n * m;
                } else {
                    Ag = (cgptr)lin + (size_t)t * lstride; Bg = Ag + (size_t)n * n;
                }
                bool ok;
                if (a.stage_ab) {
                    for (int e = tid; e < n * n; e += nt) Al[e] = Ag[e];
                    for (int e = tid; e < n * m; e += nt) Bl[e] = Bg[e];
                    __syncthreads();
                    ok = stage((clptr)Al, (clptr)Bl, t);
                } else {
                    ok = stage(Ag, Bg, t);
                }
                if (!ok) { restart = true; break; }
            }
            if (restart) { if (!P_.regularize || ++restarts > MAX_RESTARTS) { diverged = true; break; } continue; }
            reg_update(false);
            break;
        }
    };

    // backward pass on f64 MFMA products (same algebra as `backward`): with AB = [A_t | B_t]
    //   W = P AB;  G = AB^T W = [[A'PA, A'PB], [B'PA, B'PB]] (written over P);  RB = B^T AB = [B'A | B'B]
    // give Q_xx - c_xx, Q_ux, Q_uu - R and the rho-regularised variants; the rank-n_u corrections of P stay VALU.
    auto backward_m = [&]() {
        while (true) {
            const double rho_b = P_.regularize ? rho : 0.0;            // config.py:9 / ilqr.py:264-274
            const bool sreg = P_.state_regularization != 0;            // config.py:31: rho (B'B, B'A) or rho I
            for (int e = tid; e < n; e += nt) xl[e] = X[(size_t)N * n + e];
            for (int e = tid; e < prow * ldp; e += nt) Pm[e] = 0.0;
            if (!abg) for (int e = tid; e < NPa * ldp; e += nt) ABm[e] = 0.0;
            __syncthreads();
            if constexpr (MODEL == 1) { if (tid < nz) zt[tid] = ZC[(size_t)N * nz + tid]; __syncthreads(); }
            else zerr((clptr)xl, N);
            for (int e = tid; e < n * n; e += nt) {
                const int r = e / n, c = e - r * n;
                double v = 0.0;
                for (int s = 0; s < nz; ++s) {
                    double q1 = 0.0;
                    for (int s2 = 0; s2 < nz; ++s2) q1 = fma(Qfg[s * nz + s2], Hm[s2 * n + c], q1);
                    v = fma(Hm[s * n + r], q1, v);
                }
                Pm[r * ldp + c] = v;
            }
            for (int e = tid; e < nz * n; e += nt) {
                const int s = e / n, c = e - s * n;
                double q2 = 0.0;
                for (int s2 = 0; s2 < nz; ++s2) q2 = fma(Qg[s * nz + s2], Hm[s2 * n + c], q2);
                QH[e] = q2;
                Hl[e] = Hm[e];
            }
            for (int e = tid; e < n; e += nt) {
                double v = 0.0;
                for (int s = 0; s < nz; ++s) { double q1 = 0.0; for (int s2 = 0; s2 < nz; ++s2) q1 = fma(Qfg[s * nz + s2], zt[s2], q1); v = fma(Hm[s * n + e], q1, v); }
                L.v1[e] = v;     // p
            }
            __syncthreads();
            bool restart = false;
            int psel = -1;                                     // TPWL region whose [A | B] is in the panel
            for (int t = N - 1; t >= 0; --t) {
                tid = SRH_TID;
                cgptr Ag, Bg;
                bool load_panel = true;
                if constexpr (MODEL == 0) {
                    // consecutive steps of a trajectory mostly share their nearest TPWL point: the 30 KB panel is
                    // rewritten only when the region changes (uniform: every thread reads the same idx[t])
                    const int sel = idx[t];
                    load_panel = sel != psel;
                    psel = sel;
                    Ag = T.Ad + (size_t)sel * n * n; Bg = T.Bd + (size_t)sel * n * m;
                } else {
                    Ag = (cgptr)lin + (size_t)t * lstride; Bg = Ag + (size_t)n * n;
                }
                if (load_panel && !abg) {
                    for (int e = tid; e < n * n; e += nt) ABm[(e / n) * ldp + e % n] = Ag[e];
                    for (int e = tid; e < n * m; e += nt) ABm[(e / m) * ldp + n + e % m] = Bg[e];
                }
                for (int e = tid; e < n; e += nt) xl[e] = X[(size_t)t * n + e];
                __syncthreads();
                if constexpr (MODEL == 0) {
                    // z - z* = H x + z_ref - z*: one wave per output row, H from its LDS copy
                    for (int r = tid >> 6; r < nz; r += nt >> 6) {
                        double v = 0.0;
                        for (int j = tid & 63; j < n; j += 64) v = fma(Hl[r * n + j], xl[j], v);
                        v = wg::wave_sum(v);
                        if ((tid & 63) == 0) zt[r] = v + zref[r] - ztar[(size_t)t * nz + r];
                    }
                    __syncthreads();
                } else {
                    if (tid < nz) zt[tid] = ZC[(size_t)t * nz + tid];      // z(x_t) - z*_t as the forward pass computed it
                    __syncthreads();
                }
                if (tid >= O1 && tid < O1 + m) {
                    const int r = tid - O1;
                    double v = 0.0;
                    for (int s = 0; s < m; ++s) {
                        const double du = U[(size_t)t * m + s] - (!P_.include_input_var_constraint ? 0.0 :
                                          (t == 0 ? (a.u_last ? a.u_last[p * m + s] : 0.0) : U[(size_t)(t - 1) * m + s]));
                        v = fma(Rg[r * m + s], du, v);
                    }
                    L.u1[r] = v;        // c_u
                }
                if (tid < nz) {
                    double q1 = 0.0;
                    for (int s2 = 0; s2 < nz; ++s2) q1 = fma(Qg[tid * nz + s2], zt[s2], q1);
                    qz[tid] = q1;
                }
                if (!abg) {
                    wg::mfma_atb(Wm, ldp, Pm, ABm, NK4, n16 >> 4, NPa >> 4, ldp, n);              // W = P [A|B]
                    wg::mfma_atb(Pm, ldp, ABm, Wm, NK4, NPa >> 4, NPa >> 4, ldp, NPa);           // G = [A|B]^T W  (over P)
                    wg::mfma_atb(RBm, ldp, ABm + n, ABm, NK4, 1, NPa >> 4, ldp, 16, m);           // [B'A | B'B]
                } else {
                    const IlqrOp ab{nullptr, 0, Ag, Bg, n, m, 0}, bb{nullptr, 0, Ag, Bg, n, m, n};
                    const IlqrOp pp{Pm, ldp, nullptr, nullptr, 0, 0, 0}, ww{Wm, ldp, nullptr, nullptr, 0, 0, 0};
                    ilqr_mm(Wm, ldp, pp, ab, NK4, n16 >> 4, NPa >> 4, n, NK4);
                    ilqr_mm(Pm, ldp, ab, ww, NK4, NPa >> 4, NPa >> 4, NPa, n + m);
                    ilqr_mm(RBm, ldp, bb, ab, NK4, 1, NPa >> 4, 16, m);
                }
                // Q_uu = R + B'PB ; Q~_uu = Q_uu + rho B'B ; Q_ux = B'PA ; Q~_ux = Q_ux + rho B'A
                for (int e = tid; e < m * m; e += nt) {
                    const int r = e / m, c = e - r * m;
                    const double v = Rg[e] + Pm[(n + r) * ldp + n + c];
                    Quu[(size_t)t * m * m + e] = v;
                    L.T[e] = v;                       // unregularised Q_uu stays in LDS for the updates below
                    L.Quu[e] = v + (sreg ? rho_b * RBm[r * ldp + n + c] : (r == c ? rho_b : 0.0));
                }
                for (int e = tid; e < m * n; e += nt) {
                    const int r = e / n, c = e - r * n;
                    const double v = Pm[(n + r) * ldp + c];
                    L.Kt[e] = v;                      // Q_ux
                    L.BK[e] = v + (sreg ? rho_b * RBm[r * ldp + c] : 0.0);
                }
                // Q_x = c_x + A'p ; Q_u = c_u + B'p   (columns of the panel: odd leading dimension, conflict free)
                for (int e = tid; e < n + m; e += nt) {
                    double v0 = 0.0, v1a = 0.0, v2a = 0.0, v3a = 0.0;
                    int k = 0;
                    if (!abg) {
                        for (; k + 4 <= n; k += 4) {
                            v0 = fma(ABm[k * ldp + e], L.v1[k], v0);
                            v1a = fma(ABm[(k + 1) * ldp + e], L.v1[k + 1], v1a);
                            v2a = fma(ABm[(k + 2) * ldp + e], L.v1[k + 2], v2a);
                            v3a = fma(ABm[(k + 3) * ldp + e], L.v1[k + 3], v3a);
                        }
                        for (; k < n; ++k) v0 = fma(ABm[k * ldp + e], L.v1[k], v0);
                    } else {
                        // column e of [A | B] from the tables: the same four partial sums, eight entries in flight
                        cgptr col = e < n ? Ag + e : Bg + (e - n);
                        const int cs = e < n ? n : m;
                        for (; k + 8 <= n; k += 8) {
                            double c8[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) c8[q] = col[(size_t)(k + q) * cs];
                            v0 = fma(c8[0], L.v1[k], v0); v1a = fma(c8[1], L.v1[k + 1], v1a);
                            v2a = fma(c8[2], L.v1[k + 2], v2a); v3a = fma(c8[3], L.v1[k + 3], v3a);
                            v0 = fma(c8[4], L.v1[k + 4], v0); v1a = fma(c8[5], L.v1[k + 5], v1a);
                            v2a = fma(c8[6], L.v1[k + 6], v2a); v3a = fma(c8[7], L.v1[k + 7], v3a);
                        }
                        for (; k + 4 <= n; k += 4) {
                            v0 = fma(col[(size_t)k * cs], L.v1[k], v0); v1a = fma(col[(size_t)(k + 1) * cs], L.v1[k + 1], v1a);
                            v2a = fma(col[(size_t)(k + 2) * cs], L.v1[k + 2], v2a); v3a = fma(col[(size_t)(k + 3) * cs], L.v1[k + 3], v3a);
                        }
                        for (; k < n; ++k) v0 = fma(col[(size_t)k * cs], L.v1[k], v0);
                    }
                    double v = (v0 + v1a) + (v2a + v3a);
                    if (e < n) {
                        for (int s = 0; s < nz; ++s) v = fma(Hl[s * n + e], qz[s], v);     // c_x = H^T (Q (z - z*))
                        L.v2[e] = v;
                    } else {
                        v += L.u1[e - n];
                        L.u2[e - n] = v;
                        Qu[(size_t)t * m + e - n] = v;
                    }
                }
                __syncthreads();
                // the Q_ux rows of G are consumed: rows n.. of P must be zero again (K padding of the next product)
                for (int e = tid; e < (prow - n) * ldp; e += nt) Pm[n * ldp + e] = 0.0;
                if (!ilqr_gain(L, n, m)) {                        // not PD: raise rho, restart (ilqr.py:276-287)
                    __syncthreads();
                    reg_update(true);
                    restart = true;
                    break;
                }
                for (int e = tid; e < m * n; e += nt) Kout[(size_t)t * m * n + e] = L.Kk[e];
                if (tid < m) kff[(size_t)t * m + tid] = L.u1[tid];
                // QK = Quu K (m x n) into PB ; zt = Quu k
                for (int e = tid; e < m * n; e += nt) {
                    const int r = e / n, c = e - r * n;
                    double v = 0.0;
                    for (int s = 0; s < m; ++s) v = fma(L.T[r * m + s], L.Kk[s * n + c], v);
                    L.PB[e] = v + L.Kt[e];            // Quu K + Q_ux
                }
                if (tid < m) {
                    double v = 0.0;
                    for (int s = 0; s < m; ++s) v = fma(L.T[tid * m + s], L.u1[s], v);
                    zt[tid] = v;
                }
                __syncthreads();
                // P = c_xx + A'PA + K'Quu K + K'Q_ux + Q_ux'K  (in place over the A'PA block of G)
                for (int e = tid; e < n * n; e += nt) {
                    const int r = e / n, c = e - r * n;
                    double v = Pm[r * ldp + c];
                    for (int s = 0; s < nz; ++s) v = fma(Hl[s * n + r], QH[s * n + c], v);
                    for (int s = 0; s < m; ++s) {
                        v = fma(L.Kk[s * n + r], L.PB[s * n + c], v);        // K'(Quu K + Q_ux)
                        v = fma(L.Kt[s * n + r], L.Kk[s * n + c], v);        // Q_ux' K
                    }
                    Pm[r * ldp + c] = v;
                }
                for (int e = tid; e < n; e += nt) {
                    double v = L.v2[e];
                    for (int s = 0; s < m; ++s) {
                        v = fma(L.Kk[s * n + e], zt[s], v);
                        v = fma(L.Kk[s * n + e], L.u2[s], v);
                        v = fma(L.Kt[s * n + e], L.u1[s], v);
                    }
                    L.v3[e] = v;
                }
                __syncthreads();
                for (int e = tid; e < n; e += nt) L.v1[e] = L.v3[e];
                __syncthreads();
            }
            if (restart) { if (!P_.regularize || ++restarts > MAX_RESTARTS) { diverged = true; break; } continue; }
            reg_update(false);
            break;
        }
    };

    // ---- ilqr_computation (ilqr.py:27-107)
    for (int e = tid; e < (N + 1) * n; e += nt) X2[e] = (e < n) ? x0[e] : 0.0;
    for (int e = tid; e < N * m; e += nt) U2[e] = a.u_warm ? a.u_warm[p * (size_t)N * m + e] : 0.0;
    __syncthreads();
#ifdef SRH_PROFILE
    long long tf = 0, tb = 0, tc0 = clock64();
    int nf = 0;
#define IL_T0() tc0 = clock64()
#define IL_ACC(v) v += clock64() - tc0
#else
#define IL_T0() ((void)0)
#define IL_ACC(v) ((void)0)
#endif
    double cost = forward((cgptr)X2, (cgptr)U2, 1.0, (cgptr)nullptr, (cgptr)nullptr, X, U, idx, lin, ZC);
    int failed_counter = 0, it = 0;
    bool converged = false;
    while (!converged && it <= P_.max_iter) {
        IL_T0();
        restarts = 0;
        if (a.mfma) backward_m(); else backward();
        IL_ACC(tb);
        if (diverged) break;
        const double prev_cost = cost;
        double alpha = P_.alpha0, new_cost = cost;
        bool improved = false, failed = false;
        while (!improved && !failed) {
            improved = true;
            IL_T0();
            new_cost = forward((cgptr)X, (cgptr)U, alpha, (cgptr)Kout, (cgptr)kff, X2, U2, idx2, lin2, ZC2);
            IL_ACC(tf);
#ifdef SRH_PROFILE
            ++nf;
#endif
            // expected decrease (ilqr.py:71-73): the stages in blocks of 64 -- one wave sum per block, the blocks added in stage
            // order (N <= 512: exactly what wg::reduce over one stage per thread does; the one-wave form walks the same blocks)
            auto dc_stage = [&](int t) {
                double s1 = 0.0, s2 = 0.0;
                for (int r = 0; r < m; ++r) {
                    s1 = fma(kff[(size_t)t * m + r], Qu[(size_t)t * m + r], s1);
                    double q = 0.0;
                    for (int s = 0; s < m; ++s) q = fma(Quu[(size_t)t * m * m + r * m + s], kff[(size_t)t * m + s], q);
                    s2 = fma(kff[(size_t)t * m + r], q, s2);
                }
                return alpha * s1 + alpha * alpha * 0.5 * s2;
            };
            double dc = 0.0;
            if constexpr (NTH == 64) {
                for (int t0 = 0; t0 < N; t0 += 64) dc += wg::wave_sum(t0 + tid < N ? 0.0 + dc_stage(t0 + tid) : 0.0);
            } else {
                for (int t = tid; t < N; t += nt) dc += dc_stage(t);
                dc = wg::reduce(dc, 0, L.red);
            }
            const double ratio = (new_cost - prev_cost) / dc;
            if (P_.do_linesearch && (ratio <= P_.improv_lb || ratio > P_.improv_ub)) {      // ilqr.py:75: without it alpha0 is taken
                alpha = P_.alpha_scaling * alpha;
                improved = false;
                if (alpha < P_.alpha_min) {
                    reg_update(true);
                    rho += P_.rho_increase_fp;
                    failed = true;
                }
            }
        }
        if (!failed) {
            __syncthreads();
            for (int e = tid; e < (N + 1) * n; e += nt) X[e] = X2[e];
            for (int e = tid; e < N * m; e += nt) U[e] = U2[e];
            if constexpr (MODEL == 0) {
                for (int e = tid; e < N; e += nt) idx[e] = idx2[e];
            } else {
                gptr tmp = lin; lin = lin2; lin2 = tmp;
                tmp = ZC; ZC = ZC2; ZC2 = tmp;
            }
            __syncthreads();
            cost = new_cost;
            converged = ((prev_cost - cost) < P_.epsilon) && ((prev_cost - cost) >= 0.0);
            failed_counter = 0;
        } else {
            ++failed_counter;
            if (failed_counter >= P_.counter_limit) converged = true;
        }
        ++it;
    }
    if (tid == 0) { a.cost[p] = cost; a.iters[p] = diverged ? -1 : it; }
#ifdef SRH_PROFILE
    if (tid == 0 && p == 0) printf("[ilqr] forward laps per step: control %.0f zerr %.0f jacobians %.0f discretize %.0f cost+store+update %.0f\n",
                                   (double)fp[0] / (nf + 1) / N, (double)fp[1] / (nf + 1) / N, (double)fp[2] / (nf + 1) / N, (double)fp[3] / (nf + 1) / N, (double)fp[4] / (nf + 1) / N);
    if (tid == 0 && p == 0) printf("[ilqr] discretize laps per step: build %.0f inverses %.0f sep %.0f products %.0f copy %.0f\n",
                                   (double)dprof[0] / (nf + 1) / N, (double)dprof[1] / (nf + 1) / N, (double)dprof[2] / (nf + 1) / N, (double)dprof[3] / (nf + 1) / N, (double)dprof[4] / (nf + 1) / N);
    if (tid == 0 && p == 0) printf("[ilqr] problem 0: %d iterations, %d forward passes %lld clocks (%.0f per step), backward %lld clocks (%.0f per stage); mfma %d\n",
                                   it, nf, tf, (double)tf / (nf > 0 ? nf : 1) / N, tb, (double)tb / (it > 0 ? it : 1) / N, a.mfma);
#endif
}

}  // namespace

extern "C" {

void silqr_default_params(silqr_params *p) {
    if (!p) return;
    p->max_iter = 50; p->epsilon = 0.1; p->alpha0 = 1.0; p->alpha_scaling = 0.5; p->improv_lb = 1e-4;
    p->improv_ub = 100.0; p->alpha_min = 5e-2; p->counter_limit = 5; p->rho0 = 0.0; p->drho0 = 0.0;
    p->rho_scaling = 1.5; p->rho_increase_fp = 10.0; p->rho_max = 1e5; p->rho_min = 1e-3;
    p->include_input_var_constraint = 1; p->do_linesearch = 1; p->regularize = 1; p->state_regularization = 1;
}

static int tvlqr_impl(const double *dA, const double *dB, const int *didx, int n_steps, int n, int m, const double *Q,
                      const double *R, double *K, double *P) {
    srh::DevBuf dQ, dR, dK, dP, dS;
    int rc;
    if ((rc = dQ.upload(Q, sizeof(double) * n * n)) || (rc = dR.upload(R, sizeof(double) * m * m)) ||
        (rc = dK.alloc(sizeof(double) * (size_t)n_steps * m * n)) || (rc = dP.alloc(sizeof(double) * (size_t)(n_steps + 1) * n * n)) ||
        (rc = dS.alloc(sizeof(int))))
        return rc;
    const size_t lds = srh::lds_request(lqr_lds_doubles(n, m) * sizeof(double));
    SRH_REQUIRE(lds <= 160 * 1024, "sric_tvlqr: state dimension too large for LDS");
    SRH_CHECK_HIP(hipFuncSetAttribute((const void *)tvlqr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    tvlqr_kernel<<<1, NT, lds>>>(dA, dB, didx, n_steps, n, m, dQ.as<double>(), dR.as<double>(), dK.as<double>(),
                                 dP.as<double>(), dS.as<int>());
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    int st = 0;
    if ((rc = dS.download(&st, sizeof(int)))) return rc;
    if (st != 0) { srh::set_error("sric_tvlqr: R + B'PB is not positive definite"); return SRH_ENUMERIC; }
    if ((rc = dK.download(K, sizeof(double) * (size_t)n_steps * m * n))) return rc;
    if (P) return dP.download(P, sizeof(double) * (size_t)(n_steps + 1) * n * n);
    return SRH_OK;
}

int sric_tvlqr(const double *A, const double *B, int n_steps, int n_x, int n_u, const double *Q, const double *R,
               double *K, double *P) {
    SRH_REQUIRE(A && B && Q && R && K, "sric_tvlqr: null argument");
    SRH_REQUIRE(n_steps > 0 && n_x > 0 && n_u > 0 && n_u <= 16, "sric_tvlqr: bad dimensions");
    srh::DevBuf dA, dB;
    int rc;
    if ((rc = dA.upload(A, sizeof(double) * (size_t)n_steps * n_x * n_x)) || (rc = dB.upload(B, sizeof(double) * (size_t)n_steps * n_x * n_u)))
        return rc;
    return tvlqr_impl(dA.as<double>(), dB.as<double>(), nullptr, n_steps, n_x, n_u, Q, R, K, P);
}

int sric_tvlqr_tpwl(stpwl_t *h, const double *xbar, int n_steps, const double *Q, const double *R, double *K, double *P) {
    SRH_REQUIRE(h && xbar && Q && R && K, "sric_tvlqr_tpwl: null argument");
    SRH_REQUIRE(h->has_discrete, "sric_tvlqr_tpwl: model has not been pre-discretised");
    srh::DevBuf dX, dI;
    int rc;
    if ((rc = dX.upload(xbar, sizeof(double) * (size_t)n_steps * h->n)) || (rc = dI.alloc(sizeof(int32_t) * n_steps))) return rc;
    if ((rc = stpwl_nearest_dev(h, dX.as<double>(), n_steps, dI.as<int32_t>(), nullptr))) return rc;
    return tvlqr_impl(h->Ad.as<double>(), h->Bd.as<double>(), dI.as<int>(), n_steps, h->n, h->m, Q, R, K, P);
}

int sric_dare_fixed_point(const double *A, const double *B, int64_t batch, int n_x, int n_u, const double *Q,
                          const double *R, double tol, int max_iter, double *L, double *P, int32_t *iters) {
    SRH_REQUIRE(A && B && Q && R && L && P, "sric_dare_fixed_point: null argument");
    SRH_REQUIRE(batch > 0 && n_x > 0 && n_u > 0 && n_u <= 16, "sric_dare_fixed_point: bad dimensions");
    srh::DevBuf dA, dB, dQ, dR, dL, dP, dI;
    int rc;
    if ((rc = dA.upload(A, sizeof(double) * batch * n_x * n_x)) || (rc = dB.upload(B, sizeof(double) * batch * n_x * n_u)) ||
        (rc = dQ.upload(Q, sizeof(double) * n_x * n_x)) || (rc = dR.upload(R, sizeof(double) * n_u * n_u)) ||
        (rc = dL.alloc(sizeof(double) * batch * n_u * n_x)) || (rc = dP.alloc(sizeof(double) * batch * n_x * n_x)) ||
        (rc = dI.alloc(sizeof(int32_t) * batch)))
        return rc;
    const size_t lds = srh::lds_request(lqr_lds_doubles(n_x, n_u) * sizeof(double));
    SRH_REQUIRE(lds <= 160 * 1024, "sric_dare_fixed_point: state dimension too large for LDS");
    SRH_CHECK_HIP(hipFuncSetAttribute((const void *)dare_fp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dare_fp_kernel<<<(unsigned)batch, NT, lds>>>(dA.as<double>(), dB.as<double>(), n_x, n_u, dQ.as<double>(), dR.as<double>(),
                                                 tol, max_iter, dL.as<double>(), dP.as<double>(), dI.as<int>());
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if ((rc = dL.download(L, sizeof(double) * batch * n_u * n_x)) || (rc = dP.download(P, sizeof(double) * batch * n_x * n_x))) return rc;
    if (iters) return dI.download(iters, sizeof(int32_t) * batch);
    return SRH_OK;
}

int sric_dare(const double *A, const double *B, int64_t batch, int n_x, int n_u, const double *Q, const double *R,
              double tol, int max_iter, double *L, double *P, int32_t *iters) {
    SRH_REQUIRE(A && B && Q && R && L && P, "sric_dare: null argument");
    SRH_REQUIRE(batch > 0 && n_x > 0 && n_u > 0 && n_u <= 16, "sric_dare: bad dimensions");
    const int ld = n_x | 1;
    const size_t nn = (size_t)n_x * ld;
    srh::DevBuf dA, dB, dQ, dR, dL, dP, dI, dS, dW;
    int rc;
    if ((rc = dA.upload(A, sizeof(double) * batch * n_x * n_x)) || (rc = dB.upload(B, sizeof(double) * batch * n_x * n_u)) ||
        (rc = dQ.upload(Q, sizeof(double) * n_x * n_x)) || (rc = dR.upload(R, sizeof(double) * n_u * n_u)) ||
        (rc = dL.alloc(sizeof(double) * batch * n_u * n_x)) || (rc = dP.alloc(sizeof(double) * batch * n_x * n_x)) ||
        (rc = dI.alloc(sizeof(int32_t) * batch)) || (rc = dS.alloc(sizeof(int32_t) * batch)) ||
        (rc = dW.alloc(sizeof(double) * batch * 7 * nn)))
        return rc;
    const size_t gain_lds = lqr_lds_doubles(n_x, n_u) * sizeof(double);
    const size_t tail = sda_tail_doubles(n_x, n_u) * sizeof(double);
    SRH_REQUIRE(gain_lds <= 160 * 1024 && tail <= 160 * 1024, "sric_dare: state dimension too large for LDS");
    const int lds_slots = (5 * nn * sizeof(double) + tail <= 160 * 1024 && !getenv("SRH_DARE_HBM_SLOTS")) ? 1 : 0;
    const size_t lds = srh::lds_request(std::max(gain_lds, (lds_slots ? 5 * nn * sizeof(double) : 0) + tail));
    SRH_CHECK_HIP(hipFuncSetAttribute((const void *)dare_sda_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dare_sda_kernel<<<(unsigned)batch, NT, lds>>>(dA.as<double>(), dB.as<double>(), n_x, n_u, dQ.as<double>(), dR.as<double>(),
                                                  tol, max_iter, dW.as<double>(), lds_slots, dL.as<double>(), dP.as<double>(),
                                                  dI.as<int>(), dS.as<int>());
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    std::vector<int32_t> st((size_t)batch);
    if ((rc = dS.download(st.data(), sizeof(int32_t) * batch))) return rc;
    for (int64_t i = 0; i < batch; ++i)
        if (st[i] != 0) {
            srh::set_error("sric_dare: problem %lld: %s", (long long)i,
                           st[i] == 1 ? "no convergence within max_iter doubling steps"
                                      : (st[i] == 2 ? "R or R + B^T P B is not positive definite" : "singular I + G H (not stabilisable / detectable?)"));
            return SRH_ENUMERIC;
        }
    if ((rc = dL.download(L, sizeof(double) * batch * n_u * n_x)) || (rc = dP.download(P, sizeof(double) * batch * n_x * n_x))) return rc;
    if (iters) return dI.download(iters, sizeof(int32_t) * batch);
    return SRH_OK;
}

static int ilqr_impl(stpwl_t *ht, sssm_t *hs, int ssm_mode, double dt, int N, int64_t batch, const double *x0,
                     const double *z_target, const double *u_warm, const double *u_last, const double *Q,
                     const double *R, const double *Qf, const silqr_params *p, double *x, double *u, double *K,
                     double *cost, int32_t *iters) {
    const int n = ht ? ht->n : hs->n, m = ht ? ht->m : hs->m, nz = ht ? ht->nz : hs->no;
    SRH_REQUIRE(nz <= 16 && m <= 16, "silqr_solve: n_z and n_u must be <= 16");
    silqr_params par;
    if (p) par = *p; else silqr_default_params(&par);
    srh::DevBuf d0, dz, duw, dul, dQ, dR, dQf, ox, ou, oK, oc, oi, work, iwork, lin;
    int rc;
    const size_t stride = (size_t)(N + 1) * n + (size_t)N * m * 3 + (size_t)N * m * m + (size_t)N * m * n + 2 * (size_t)(N + 1) * nz + 8;
    const size_t lstride = (size_t)n * n + (size_t)n * m + n;
    if ((rc = d0.upload(x0, sizeof(double) * batch * n)) || (rc = dz.upload(z_target, sizeof(double) * batch * (N + 1) * nz)) ||
        (rc = dQ.upload(Q, sizeof(double) * nz * nz)) || (rc = dR.upload(R, sizeof(double) * m * m)) ||
        (rc = dQf.upload(Qf, sizeof(double) * nz * nz)) || (rc = ox.alloc(sizeof(double) * batch * (N + 1) * n)) ||
        (rc = ou.alloc(sizeof(double) * batch * N * m)) || (rc = oK.alloc(sizeof(double) * batch * N * m * n)) ||
        (rc = oc.alloc(sizeof(double) * batch)) || (rc = oi.alloc(sizeof(int32_t) * batch)) ||
        (rc = work.alloc(sizeof(double) * stride * batch)) || (rc = iwork.alloc(sizeof(int32_t) * 2 * N * batch)))
        return rc;
    if (hs && (rc = lin.alloc(sizeof(double) * 2 * N * lstride * batch))) return rc;
    if (u_warm && (rc = duw.upload(u_warm, sizeof(double) * batch * N * m))) return rc;
    if (u_last && (rc = dul.upload(u_last, sizeof(double) * batch * m))) return rc;
    IlqrArgs a{N, n, m, nz, par, d0.as<double>(), dz.as<double>(), u_warm ? duw.as<double>() : nullptr,
               u_last ? dul.as<double>() : nullptr, dQ.as<double>(), dR.as<double>(), dQf.as<double>(), ox.as<double>(),
               ou.as<double>(), oK.as<double>(), oc.as<double>(), oi.as<int>(), work.as<double>(), iwork.as<int>(), stride,
               hs ? lin.as<double>() : nullptr, ssm_mode, 0, 0, 0, 0, dt};
    if (hs && !getenv("SRH_SSM_DENSE_JACOBIAN")) {
        const std::vector<int> er = ssm_exponents(hs->n, hs->order_r);
        a.jl_cap = ssm::jacobian_list_cap(er.data(), hs->nr, hs->n);
    }
    const size_t tail = 20 + (size_t)16 * n + 16 + NT + 16 + n +
                        (hs ? lstride + ssm::work_doubles(hs->n, hs->m, hs->no, hs->nr, hs->ns) + ssm::lds_tab_doubles(hs->n, hs->no, hs->nr, hs->ns, a.jl_cap) : 0);
    // preferred: backward pass on f64 MFMA products over three padded (NPa x ld) panels + one 16-row panel
    const size_t NPa = (size_t)((n + m + 15) & ~15), ldp = NPa + 1;
    const size_t mf_off = lqr_lds_doubles(n, m, 256) + tail;
    const size_t mf_lds = (mf_off + 3 * NPa * ldp + 16 * ldp + (size_t)16 * n + 16) * sizeof(double);
    size_t lds;
    // when those do not fit (n_x > 64): P / G with n + m rows, W with NK4 rows, [A|B] read in place (ilqr_mm)
    const size_t NK4h = (size_t)((n + 3) & ~3);
    const size_t mf2_lds = (mf_off + ((size_t)n + m + NK4h + 16) * ldp + (size_t)16 * n + 16) * sizeof(double);
    if (mf_lds <= 160 * 1024 && !getenv("SRH_ILQR_NO_MFMA")) {
        a.mfma = 1;
        a.panel_off = mf_off;
        a.stage_ab = hs ? 1 : 0;
        lds = mf_lds;
    } else if (mf2_lds <= 160 * 1024 && !getenv("SRH_ILQR_NO_MFMA")) {
        a.mfma = 2;
        a.panel_off = mf_off;
        a.stage_ab = hs ? 1 : 0;
        lds = mf2_lds;
    } else {
        lds = (lqr_lds_doubles(n, m) + tail) * sizeof(double);
        // VALU fallback: the backward pass reads (A_t, B_t) n times per stage: stage them in LDS when they fit
        if (hs) a.stage_ab = 1;
        else if (lds + sizeof(double) * lstride <= 160 * 1024) { a.stage_ab = 1; lds += sizeof(double) * lstride; }
    }
    SRH_REQUIRE(lds <= 160 * 1024, "silqr_solve: state dimension too large for LDS (%zu bytes)", lds);
    lds = srh::lds_request(lds);
    // variants: the reference's robots at the benchmark / shipped basis sizes (TPWL), the C3 SSM shape; else all sizes
    // (model, n_x, n_u, threads)
#define SRH_ILQR_VARIANTS(X) X(1, 10, 8, 64) X(0, 60, 4, NT) X(0, 60, 8, NT) X(0, 72, 4, NT) X(1, 10, 8, NT) X(0, 0, 0, NT) X(1, 0, 0, NT)
    {
        const int model = ht ? 0 : 1;
        // SRH_ILQR_THREADS=64: one wave per problem for the C3 shape (A/B runs; measured SLOWER than eight waves -- 21 ms against
        // 10.4 ms -- because the parallel phases, 110 dot products of length 285 and 2850 table entries per step, then run on 64
        // lanes: DESIGN.md section 13); results bit-identical to the 512-thread form
        const char *force = getenv("SRH_ILQR_THREADS");
        const int threads = (force && atoi(force) == 64 && hs && a.mfma == 1 && n == 10 && m == 8 && N <= 512) ? 64 : NT;
        bool launched = false;
#define X(MD, NX, MU, TH)                                                                                                       \
    if (!launched && model == MD && (NX == 0 || n == NX) && (MU == 0 || m == MU) && threads == TH) {                            \
        SRH_CHECK_HIP(hipFuncSetAttribute((const void *)ilqr_kernel<MD, NX, MU, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        ilqr_kernel<MD, NX, MU, TH><<<(unsigned)batch, TH, lds>>>(ht ? ht->view() : TpwlDev{}, hs ? hs->view() : SsmDev{}, a);  \
        launched = true;                                                                                                        \
    }
        SRH_ILQR_VARIANTS(X)
#undef X
        SRH_REQUIRE(launched, "silqr_solve: no kernel variant for model %d, n_x %d, n_u %d, %d threads", model, n, m, threads);
    }
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if ((rc = ox.download(x, sizeof(double) * batch * (N + 1) * n)) || (rc = ou.download(u, sizeof(double) * batch * N * m)) ||
        (rc = oK.download(K, sizeof(double) * batch * N * m * n)))
        return rc;
    if (cost && (rc = oc.download(cost, sizeof(double) * batch))) return rc;
    if (iters && (rc = oi.download(iters, sizeof(int32_t) * batch))) return rc;
    return SRH_OK;
}

int silqr_solve(stpwl_t *h, int N, int64_t batch, const double *x0, const double *z_target, const double *u_warm,
                const double *u_last, const double *Q, const double *R, const double *Qf, const silqr_params *p,
                double *x, double *u, double *K, double *cost, int32_t *iters) {
    SRH_REQUIRE(h && x0 && z_target && Q && R && Qf && x && u && K, "silqr_solve: null argument");
    SRH_REQUIRE(h->has_discrete, "silqr_solve: model has not been pre-discretised");
    SRH_REQUIRE(h->nz > 0, "silqr_solve: Need to set output or meas. model");
    SRH_REQUIRE(N > 0 && batch > 0, "silqr_solve: bad dimensions");
    return ilqr_impl(h, nullptr, 0, 0.0, N, batch, x0, z_target, u_warm, u_last, Q, R, Qf, p, x, u, K, cost, iters);
}

int silqr_solve_ssm(sssm_t *h, int mode, double dt, int N, int64_t batch, const double *x0, const double *z_target,
                    const double *u_warm, const double *u_last, const double *Q, const double *R, const double *Qf,
                    const silqr_params *p, double *x, double *u, double *K, double *cost, int32_t *iters) {
    SRH_REQUIRE(h && x0 && z_target && Q && R && Qf && x && u && K, "silqr_solve_ssm: null argument");
    SRH_REQUIRE(mode >= SSM_FE && mode <= SSM_DISCRETE_MAP, "self.discr_method must be in [fe, be, bil, zoh]");
    SRH_REQUIRE(mode != SSM_DISCRETE_MAP || h->has_discrete, "silqr_solve_ssm: model has no discrete map");
    SRH_REQUIRE(h->n == h->no, "silqr_solve_ssm: the reduced -> observed map needs n_x == n_o");
    SRH_REQUIRE(N > 0 && batch > 0, "silqr_solve_ssm: bad dimensions");
    return ilqr_impl(nullptr, h, mode, dt, N, batch, x0, z_target, u_warm, u_last, Q, R, Qf, p, x, u, K, cost, iters);
}

}  // extern "C"
