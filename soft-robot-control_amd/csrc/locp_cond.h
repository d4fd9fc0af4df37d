// Condensed (output-space) interior point for the LOCP QP without its trust-region rows -- the fast path of qp::solve.
//
// The cost (sofacontrol/scp/locp.py:226-252) and the state rows X / Xf (locp.py:330-337) see the state x_k only through
// a few OUTPUT directions: 2 H^T Qz H = Cq^T Cq, X.A and Xf.A all lie in the row space of C_o (po x n; Diamond and Trunk
// drivers: the tip x / y rows, po = 2).  Eliminating the states with the dynamics,
//     y_k = C_o x_k = yfree_k + sum_{j<k} G[k][j] u_j ,   G[k][j] = C_o A_{k-1} ... A_{j+1} B_j   (po x m),
// leaves a QP in u alone (N m variables) whose interior-point Newton systems are
//     M du = -g ,  M = blkdiag(D_j) + G^T blkdiag(S_k) G ,  D_j = 2R + U.A^T D_u U.A ,  S_k = Tc^T Tc + Tx^T D_x,k Tx .
// They are solved in OUTPUT space (Woodbury): with D_j = Ld_j Ld_j^T, S_k = Ls_k Ls_k^T, Gd = Ls^T G Ld^-T,
//     K = I + Gd Gd^T = R^T R    (N po x N po: 100 x 100 for both robots at N = 50),
//     t = -D^-1 (gu + G^T gy) ,   K v = Ls^T G t ,   du = t - D^-1 G^T Ls v ,   dy = G du
// (gu / gy: input / output part of the gradient; the total gradient is formed in u-space BEFORE D^-1 amplifies it --
// short-cuts through (K - I) lose that cancellation): two products with G and two with G^T per Newton system, G
// streaming from L2 (180 - 360 KB, ~20 B/clock per CU: tools/probes/mv_probe.hip), everything else in LDS.
// Per interior-point iteration: one Gram product (f64 MFMA, accumulated in registers over 32-row slabs of the scaled
// G^T staged through LDS), one tile Cholesky of K in LDS (diagonal 16 x 16 tiles factored and inverted in registers with
// v_readlane broadcasts, panel / trailing updates on MFMA) -- ~2.5 MFLOP instead of the ~50 MFLOP of a backward
// Riccati factorisation with n_x = 60 states.  Once per QP: G by the adjoint recursion
//     Theta_{j-1} = [C_o ; Theta_j A_j] ,  G[:, j] = Theta_j B_j
// as one MFMA product per stage with the same LDS panel [A_j | B_j] the Riccati path uses.
// The slack / multiplier / residual of every inequality row live in the REGISTERS of the thread that owns the row
// (<= 2 rows per thread); only the weights D, the gradient shifts rho and the multipliers go through L2 for the
// per-stage sums.
//
// The iteration (Mehrotra predictor-corrector, starting point, regularised weights, stopping rule) is the one of
// qp::solve / oracle/riccati_ipm.py; the numpy statement of THIS file is oracle/condensed_ipm.py (newton='output').
// If the minimiser leaves the trust region, or a factorisation breaks down, qp::solve falls back to the stage-wise
// Riccati solve of the full QP.
#pragma once

struct QCWork {                        // per-problem scratch in HBM/L2 (doubles), behind the QPWork block
    // (N m) x ldG : GT[(j,b)][(k-1) po + a] = G[k][a][j][b]; zero where k <= j.  Streams from L2 at ~20 B/clock per CU
    // (tools/probes/mv_probe.hip).  A packed triangular store was tried: its index arithmetic cost more than the bytes.
    gptr GT;
};

__host__ __device__ inline int qc_ldg(const QPDims &d) { return 16 * d.KT; }
__host__ __device__ inline size_t qc_work_doubles(const QPDims &d) {
    if (!d.cond) return 0;
    return (size_t)d.N * d.m * qc_ldg(d) + 8;
}
__device__ inline void qc_carve(QCWork &w, gptr base, const QPDims &d) { w.GT = base; }

namespace qpc {

constexpr int TS = 17;                 // row stride of a 16 x 16 LDS tile (odd: row and column accesses conflict free)
constexpr int TSZ = 16 * TS;
constexpr int SR = 32;                 // rows (contraction length) of a Gram slab
constexpr int QR = 2;                  // inequality rows per thread (register resident)

struct Lds {
    lptr A;        // [A | B] panel (NK x ld) while condensing; Gram slab (SR x ldG); u-space temporaries otherwise
    lptr B;        // Theta^T (NK x ldT) while condensing; upper tiles of K / its Cholesky factor afterwards
    lptr Rinv;     // KT tiles: inverses of the diagonal tiles of the factor
    lptr Ldi;      // diagD: N x m, 1 / sqrt(D_j[b][b]);  else N x m x m : Ld_j^-1 (lower)
    lptr Ls;       // N x po x po : Ls_k (lower), index k - 1
    lptr u, du;                        // N m
    lptr y, dy, yf;                    // ldG, index (k-1) po + a, k = 1..N
    lptr ya, yb, yc, yd, yg;           // y-space temporaries (ldG)
    lptr ks;                           // 1 / sqrt(diag K): the symmetric scaling under which K is factored
    lptr UA, Tx;                       // (nU x m), ((nX + nXf) x po): row coefficients
    lptr v1, v2, Qu, red;
    liptr flag, idxl;
    liptr goff;                        // N ints: start of the rows (j, .) in the packed G^T
    lptr ta, tb, tc, part;             // aliases inside region A: u-space temporaries (N m), reduction scratch
};

struct Sizes { size_t regA, regB, rinv, ldi, ls, nm4, ldG, ua, tx, ld, idx; };
__host__ __device__ inline Sizes sizes(const QPDims &d, int nthreads) {
    Sizes s;
    // the panel and Theta^T need rows up to the contraction extent roundup4(n) only (NK; zero padded)
    const size_t nk = (size_t)d.NK, nm = (size_t)d.N * d.m;
    s.ldG = qc_ldg(d);
    s.nm4 = (nm + 3) & ~(size_t)3;
    s.regA = nk * d.ld > (size_t)SR * s.ldG ? nk * d.ld : (size_t)SR * s.ldG;
    const size_t tmp = 3 * s.nm4 + (size_t)nthreads;
    if (s.regA < tmp) s.regA = tmp;
    const size_t tiles = (size_t)d.KT * (d.KT + 1) / 2 * TSZ;
    s.regB = nk * (s.ldG + 1) > tiles ? nk * (s.ldG + 1) : tiles;
    s.rinv = (size_t)d.KT * TSZ;
    s.ldi = d.diagD ? s.nm4 : nm * d.m;
    s.ls = (size_t)d.N * d.po * d.po;
    s.ua = ((size_t)d.nU * d.m + 3) & ~(size_t)3;
    s.tx = ((size_t)(d.nX + d.nXf) * d.po + 3) & ~(size_t)3;
    s.ld = ((size_t)d.ld + 3) & ~(size_t)3;
    s.idx = ((size_t)(d.N / 2 + 2) + 3) & ~(size_t)3;
    return s;
}
__host__ __device__ inline size_t lds_doubles(const QPDims &d, int nthreads) {
    const Sizes s = sizes(d, nthreads);
    return s.regA + s.regB + s.rinv + s.ldi + s.ls + 2 * s.nm4 + 9 * s.ldG + s.ua + s.tx + 2 * s.ld + 16 + 16 + 4 + 2 * s.idx;
}
__device__ inline void lds_carve(Lds &L, lptr base, const QPDims &d, int nthreads) {
    const Sizes s = sizes(d, nthreads);
    lptr p = base;
    auto take = [&](size_t c) { lptr q = p; p += c; return q; };
    L.A = take(s.regA); L.B = take(s.regB); L.Rinv = take(s.rinv); L.Ldi = take(s.ldi); L.Ls = take(s.ls);
    L.u = take(s.nm4); L.du = take(s.nm4);
    L.y = take(s.ldG); L.dy = take(s.ldG); L.yf = take(s.ldG);
    L.ya = take(s.ldG); L.yb = take(s.ldG); L.yc = take(s.ldG); L.yd = take(s.ldG); L.yg = take(s.ldG); L.ks = take(s.ldG);
    L.UA = take(s.ua); L.Tx = take(s.tx);
    L.v1 = take(s.ld); L.v2 = take(s.ld); L.Qu = take(16); L.red = take(16);
    L.flag = (liptr)take(4);
    L.idxl = (liptr)take(s.idx);
    L.goff = (liptr)take(s.idx);
    L.ta = L.A; L.tb = L.A + s.nm4; L.tc = L.A + 2 * s.nm4; L.part = L.A + 3 * s.nm4;
}

#ifdef SRH_PROFILE
struct Prof { long long t[24]; long long last; };
#define QC_SUB(P, x) do { const long long now_ = clock64(); (P).t[x] += now_ - (P).last; (P).last = now_; } while (0)
#else
struct Prof { };
#define QC_SUB(P, x) ((void)0)
#endif

__device__ __forceinline__ double readlane_d(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int g_c0(const QPDims &d, int j) { return ((j * d.po) >> 4) << 4; }
__device__ __forceinline__ int tile_index(int I, int J, int KT) { return I * KT - I * (I - 1) / 2 + (J - I); }   // I <= J

// ------------------------------------------------------------------ products with G (streams from HBM/L2)
// yv[i] = sum_{rows (j,b), j <= i / po} GT[row][i] uv[row]   (uv, yv in LDS).  512 threads = 4 row groups x 128 columns;
// every thread issues its loads in blocks of CH independent requests (the latency of L2, not its bandwidth, is what a
// one-load-per-trip loop would pay).
__device__ __forceinline__ void g_times(const QPDims &d, const QCWork &w, Lds &L, clptr uv, lptr yv) {
    constexpr int CH = 16, CW = 128;
    const int ldG = qc_ldg(d), m = d.m, po = d.po, nm = d.N * m, tid = SRH_TID, nt = blockDim.x;
    const int G = nt / CW, col = tid % CW, grp = tid / CW;
    const int cc = col < ldG ? col : ldG - 1;
    int rmax = (cc / po + 1) * m;                          // rows (j,b) with j <= col / po reach this column
    if (rmax > nm) rmax = nm;
    const int cmax = min(ldG - 1, (col | 63));             // wave-uniform bound: the largest column of this wave
    int rwave = (cmax / po + 1) * m;
    if (rwave > nm) rwave = nm;
    double acc = 0.0;
    cgptr g = w.GT + cc;
    for (int r0 = grp; r0 < rwave; r0 += G * CH) {
        double gv[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t) { const int r = r0 + G * t; gv[t] = g[(size_t)(r < nm ? r : nm - 1) * ldG]; }
#pragma unroll
        for (int t = 0; t < CH; ++t) { const int r = r0 + G * t; const double uu = uv[r < nm ? r : nm - 1]; acc = fma(gv[t], r < rmax ? uu : 0.0, acc); }
    }
    L.part[grp * CW + col] = acc;
    __syncthreads();
    if (tid < ldG) {
        double s = 0.0;
        for (int q = 0; q < G; ++q) s += L.part[q * CW + tid];
        yv[tid] = s;
    }
    __syncthreads();
}

// out1[row] = sum_i GT[row][i] y1[i]  (and out2 with y2 when y2 != null): 8 lanes per row, 16 independent loads each
__device__ __forceinline__ void gT_times(const QPDims &d, const QCWork &w, Lds &L, clptr y1, clptr y2, lptr out1, lptr out2) {
    const int ldG = qc_ldg(d), m = d.m, po = d.po, nm = d.N * m, tid = SRH_TID, nt = blockDim.x;
    const int g8 = tid & 7;
    for (int r0 = 0; r0 < nm; r0 += nt / 8) {              // uniform trip count
        const int r = r0 + (tid >> 3), rc = r < nm ? r : nm - 1;
        const int q0 = ((rc / m) * po) >> 3;               // first 8-column group the row reaches
        cgptr g = w.GT + (size_t)rc * ldG + g8;
        double gv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int qq = 8 * q < ldG ? q : 0; gv[q] = g[8 * (qq >= q0 ? qq : q0)]; }
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (8 * q < ldG) {                              // uniform
                const double gq = q >= q0 ? gv[q] : 0.0;
                a1 = fma(gq, y1[8 * q + g8], a1);
                if (y2) a2 = fma(gq, y2[8 * q + g8], a2);
            }
        }
        a1 = wg::group_sum<8>(a1);
        if (y2) a2 = wg::group_sum<8>(a2);
        if (g8 == 0 && r < nm) { out1[r] = a1; if (y2) out2[r] = a2; }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ condensation (once per QP)
// Free response: x (N+1 x n) must hold the zero-input rollout; yf = C_o x.
// G by the adjoint recursion; Theta^T lives in L.B (NK x ldT, column i = (k-1) po + a), the stage panel in L.A.
template <int MSEL, int NSEL>
__device__ __forceinline__ void condense(const QPDims &d, const QPConst &c, const QPDyn &dyn, cgptr x, QCWork &w, Lds &L) {
    const int N = d.N, n = d.n, m = d.m, po = d.po, ld = d.ld, KT = d.KT, ldG = qc_ldg(d), ldT = ldG + 1;
    const int nk = d.NK, NPa = d.NPa;
    const int tid = SRH_TID, nt = blockDim.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    for (int e = tid; e < ldG; e += nt) {
        double v = 0.0;
        if (e < N * po) {
            const int k = e / po + 1, a = e - (k - 1) * po;
            for (int j = 0; j < n; ++j) v = fma(c.Co[(size_t)a * n + j], x[(size_t)k * n + j], v);
        }
        L.yf[e] = v;
    }
    for (int e = tid; e < nk * ldT; e += nt) L.B[e] = 0.0;
    for (int e = tid; e < nk * ld; e += nt) L.A[e] = 0.0;
    __syncthreads();
    // a QPLds view for panel_load: panel in region A
    QPLds P{};
    P.AB = L.A; P.idxl = L.idxl; P.psel = -1;
    const int MT = NPa >> 4;                         // row tiles of the product: [Theta A | Theta B]^T has n + m rows
    const int KS = (n + 3) >> 2;                     // k-steps of 4 over the contraction (rows of A); zero padded
    for (int j = N - 1; j >= 0; --j) {
        // every wave is done with the previous panel (its MFMAs have issued their LDS reads) before it is replaced
        const int sel = __builtin_amdgcn_readfirstlane(L.idxl[j]);
        const bool reload = dyn.idx == nullptr || __builtin_amdgcn_readfirstlane(P.psel) != sel;
        if (reload) __syncthreads();
        if (qp::panel_load(d, dyn, P, j)) __syncthreads();
        const int t_first = (j * po) >> 4;           // first active column tile (columns i >= j po)
        for (int ti = wave; ti < KT; ti += nw) {
            if (ti < t_first) {                      // structurally zero part of the rows (j, b) of GT
                for (int e = lane; e < m * 16; e += 64) w.GT[((size_t)j * m + (e >> 4)) * ldG + 16 * ti + (e & 15)] = 0.0;
                continue;
            }
            // new columns of stage k = j + 1: C_o Phi(j+1, j+1) = C_o   (owned by exactly one tile each)
            for (int e = lane; e < po * n; e += 64) {
                const int a = e / n, r = e - a * n, i = j * po + a;
                if ((i >> 4) == ti) L.B[r * ldT + i] = c.Co[(size_t)a * n + r];
            }
            __builtin_amdgcn_wave_barrier();
            // this wave owns columns 16 ti .. 16 ti + 15 of Theta^T: operands to registers, products, write back in place
            constexpr int KSMAX = NSEL > 0 ? (NSEL + 3) / 4 : 32;
            double bop[KSMAX];
#pragma unroll
            for (int s = 0; s < KSMAX; ++s) bop[s] = s < KS ? L.B[(4 * s + kk) * ldT + 16 * ti + l16] : 0.0;
            for (int ci = 0; ci < MT; ++ci) {
                wg::qp_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KSMAX; ++s)
                    if (s < KS) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(L.A[(4 * s + kk) * ld + 16 * ci + l16], bop[s], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 16 * ci + kk + 4 * q, i = 16 * ti + l16;
                    if (row < n) L.B[row * ldT + i] = acc[q];                                   // Theta_{j-1}^T
                    else if (row < n + m) w.GT[((size_t)j * m + (row - n)) * ldG + i] = acc[q];     // G[:, j]^T
                }
            }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ per-stage pieces of a Newton system
// gu (N m), gy (ldG) from the row weights `wrow` = rho (right-hand side) or lam (dual residual), read from L2
__device__ __forceinline__ void gradients(const QPDims &d, const QPConst &c, const QPData &q, Lds &L, cgptr wrow, lptr gu, lptr gy) {
    const int N = d.N, m = d.m, po = d.po, nz = d.nz, ldG = qc_ldg(d), tid = SRH_TID, nt = blockDim.x;
    for (int e = tid; e < ldG; e += nt) {
        double g = 0.0;
        if (e < N * po) {
            const int k = e / po + 1, a = e - (k - 1) * po;
            cgptr S = (k == N) ? c.ScN : c.Sc;
            for (int b = 0; b < po; ++b) g = fma(S[a * po + b], L.y[(k - 1) * po + b], g);
            if (q.z) for (int b = 0; b < nz; ++b) g = fma(-c.Cz2[a * nz + b], q.z[(size_t)k * nz + b], g);
            if (k == N && c.Qzf && q.zf) for (int b = 0; b < nz; ++b) g = fma(-c.Czf2[a * nz + b], q.zf[b], g);
            const int nr = qp::xrows_of(d, k);
            cgptr wr = wrow + (size_t)(k - 1) * d.RX;
            for (int r = 0; r < nr; ++r) g = fma(L.Tx[r * po + a], wr[r], g);
        }
        gy[e] = g;
    }
    cgptr wu = wrow + (size_t)N * d.RX;
    for (int e = tid; e < N * m; e += nt) {
        const int k = e / m, a = e - k * m;
        double g = 0.0;
        for (int b = 0; b < m; ++b) g = fma(c.R2[a * m + b], L.u[k * m + b] - (q.ud ? q.ud[(size_t)k * m + b] : 0.0), g);
        for (int r = 0; r < d.nU; ++r) g = fma(L.UA[r * m + a], wu[(size_t)k * d.nU + r], g);
        gu[e] = g;
    }
    __syncthreads();
}

// in-place Cholesky factor (lower, row-major m x m; the strict upper part is zeroed) of a tiny SPD matrix in LDS, by one
// thread.  false: not positive definite.  semidef: a pivot that cancelled to (numerically) nothing gets a zero column
// instead (oracle/condensed_ipm.py: chol_psd) -- the output blocks S_k lose rank in floating point when an output
// direction is weighted only by state rows whose weights go to zero.
__device__ __forceinline__ bool small_chol(lptr A, int m, bool semidef = false) {
    double dmax = 0.0;
    for (int i = 0; i < m; ++i) dmax = fmax(dmax, fabs(A[i * m + i]));
    for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j) {
            double sum = A[i * m + j];
            for (int k = 0; k < j; ++k) sum = fma(-A[i * m + k], A[j * m + k], sum);
            if (i == j) {
                if (semidef) { A[i * m + i] = sum > 1e-14 * dmax ? sqrt(sum) : 0.0; continue; }
                if (!(sum > 1e-300 * dmax)) return false;
                A[i * m + i] = sqrt(sum);
            } else {
                const double piv = A[j * m + j];
                A[i * m + j] = piv > 0.0 ? sum / piv : 0.0;
            }
        }
    for (int i = 0; i < m; ++i)
        for (int j = i + 1; j < m; ++j) A[i * m + j] = 0.0;
    return true;
}

// in-place inverse of a lower-triangular matrix, column by column: column cc of X = L^-1 needs L[i][cc..i] only -- its
// own column (each entry read once, right before it is overwritten) and columns to its right, which still hold L.
__device__ __forceinline__ void tri_inverse(lptr A, int m) {
    for (int cc = 0; cc < m; ++cc) {
        const double xcc = 1.0 / A[cc * m + cc];
        for (int i = cc + 1; i < m; ++i) {
            double sum = A[i * m + cc] * xcc;
            for (int k = cc + 1; k < i; ++k) sum = fma(A[i * m + k], A[k * m + cc], sum);
            A[i * m + cc] = -sum / A[i * m + i];
        }
        A[cc * m + cc] = xcc;
    }
}

// D_j = 2R + U.A^T D_u U.A -> Ld_j^-1 (diagD: the reciprocal square roots of its diagonal);  S_k = S*_k + T^T D_x T -> Ls_k.
// Returns false when one of them is not positive definite.  Dw = the weights D (L2).
__device__ __forceinline__ bool stage_factors(const QPDims &d, const QPConst &c, cgptr Dw, Lds &L) {
    const int N = d.N, m = d.m, po = d.po, tid = SRH_TID, nt = blockDim.x;
    cgptr Du = Dw + (size_t)N * d.RX;
    if (tid == 0) L.flag[0] = 1;
    __syncthreads();
    bool ok = true;
    if (d.diagD) {
        for (int e = tid; e < N * m; e += nt) {
            const int k = e / m, b = e - k * m;
            double v = c.R2[b * m + b];
            for (int r = 0; r < d.nU; ++r) { const double a = L.UA[r * m + b]; v = fma(a * Du[(size_t)k * d.nU + r], a, v); }
            ok = ok && (v > 0.0);
            L.Ldi[e] = 1.0 / sqrt(v);
        }
    } else {
        for (int e = tid; e < N * m * m; e += nt) {
            const int k = e / (m * m), ab = e - k * m * m, a = ab / m, b = ab - a * m;
            double v = c.R2[ab];
            for (int r = 0; r < d.nU; ++r) v = fma(L.UA[r * m + a] * Du[(size_t)k * d.nU + r], L.UA[r * m + b], v);
            L.Ldi[e] = v;
        }
    }
    for (int e = tid; e < N * po * po; e += nt) {
        const int k = e / (po * po) + 1, ab = e - (k - 1) * po * po, a = ab / po, b = ab - a * po;
        double v = (k == N ? c.ScN : c.Sc)[ab];
        const int nr = qp::xrows_of(d, k);
        cgptr Dk = Dw + (size_t)(k - 1) * d.RX;
        for (int r = 0; r < nr; ++r) v = fma(L.Tx[r * po + a] * Dk[r], L.Tx[r * po + b], v);
        L.Ls[e] = v;
    }
    __syncthreads();
    // one thread per stage; the output blocks from thread 256 on (other SIMDs than the input blocks)
    if (!d.diagD && tid < N) {
        lptr A = L.Ldi + (size_t)tid * m * m;
        ok = small_chol(A, m);
        if (ok) tri_inverse(A, m);
    }
    const int t2 = tid - (nt >= 512 ? 256 : 64);
    if (t2 >= 0 && t2 < N) ok = ok && small_chol(L.Ls + (size_t)t2 * po * po, po, true);
    if (!ok) L.flag[0] = 0;
    __syncthreads();
    return L.flag[0] != 0;
}

// ------------------------------------------------------------------ Gram matrix K = I + Gd Gd^T -> upper tiles in L.B
// Slab = rows (j, b) of Gd^T for the stages j0 .. j0 + cj - 1.  One thread per (stage, column i): it applies Ld_j^-1 to
// its column of the G^T block and -- po == 2 -- mixes the two columns of an output stage by Ls_k with one DPP swap
// (otherwise a second pass over the slab does).  With a compile-time n_u the loads of slab s + 1 are issued before the
// MFMAs of slab s.
template <int MSEL>
__device__ __forceinline__ void gram(const QPDims &d, const QCWork &w, Lds &L) {
    constexpr bool PIPE = MSEL > 0;
    constexpr int MB = PIPE ? MSEL : 16;                                   // bound of the register block
    constexpr int IT = PIPE ? (SR / MB * 128 + 511) / 512 : 1;             // (stage, column) items per thread and slab
    const int N = d.N, m = d.m, po = d.po, KT = d.KT, ldG = qc_ldg(d), NP = N * po;
    const int tid = SRH_TID, nt = blockDim.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    const int cj = SR / m > 0 ? SR / m : 1;        // stages per slab
    const int rows_used = (cj * m + 3) & ~3;
    const int ntiles = KT * (KT + 1) / 2;
    wg::qp_d4 acc[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    // tile t = wave + nw * slot -> (I, J), I <= J (row-major over the upper triangle)
    int tI[4], tJ[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        int t = wave + nw * sl, I = 0;
        if (t >= ntiles) { tI[sl] = -1; tJ[sl] = 0; continue; }
        while (t >= KT - I) { t -= KT - I; ++I; }
        tI[sl] = I; tJ[sl] = I + t;
    }
    for (int e = tid; e < SR * ldG; e += nt) L.A[e] = 0.0;      // rows beyond cj m stay zero
    // one item: column i of the G^T block of stage j (m values), zero where the block is structurally zero
    auto fetch_item = [&](int e, int j0, double (&g)[MB]) {
        const int jj = e / ldG, i = e - jj * ldG, j = j0 + jj;
        const bool live = jj < cj && j < N && i < NP && i >= j * po;            // k = i / po + 1 > j
        const int jc = j < N ? j : N - 1, ic = jj < cj ? i : 0;
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            const double v = w.GT[((size_t)jc * m + (b < m ? b : 0)) * ldG + ic];
            g[b] = (live && b < m) ? v : 0.0;
        }
    };
    auto store_item = [&](int e, int j0, const double (&g)[MB]) {
        const int jj = e / ldG, i = e - jj * ldG, j = j0 + jj;
        const int jc = j < N ? j : N - 1;
        double t[MB];
        if (d.diagD) {
#pragma unroll
            for (int b = 0; b < MB; ++b) t[b] = b < m ? g[b] * L.Ldi[jc * m + b] : 0.0;
        } else {
            clptr Li = L.Ldi + (size_t)jc * m * m;
#pragma unroll
            for (int b = 0; b < MB; ++b) {
                double s = 0.0;
                if (b < m) {
#pragma unroll
                    for (int b2 = 0; b2 < MB; ++b2) if (b2 <= b) s = fma(Li[b * m + b2], g[b2], s);
                }
                t[b] = s;
            }
        }
        if (po == 2) {
            // columns (k, 0) and (k, 1) sit in adjacent lanes: out_0 = t_0 L00 + t_1 L10, out_1 = t_1 L11
            const int kq = (i < NP ? i : 0) >> 1, a = i & 1;
            clptr Lk = L.Ls + (size_t)kq * 4;
            const double l_own = a == 0 ? Lk[0] : Lk[3], l_oth = a == 0 ? Lk[2] : 0.0;
#pragma unroll
            for (int b = 0; b < MB; ++b) {
                const double tp = wg::dpp_mov<0xB1>(t[b]);             // quad_perm [1,0,3,2]: the partner lane
                if (b < m && jj < cj) L.A[(size_t)(jj * m + b) * ldG + i] = fma(tp, l_oth, t[b] * l_own);
            }
        } else {
#pragma unroll
            for (int b = 0; b < MB; ++b) if (b < m && jj < cj) L.A[(size_t)(jj * m + b) * ldG + i] = t[b];
        }
    };
    double g[IT][MB];
    if constexpr (PIPE) {
#pragma unroll
        for (int it = 0; it < IT; ++it) fetch_item(tid + nt * it, 0, g[it]);
    }
    __syncthreads();
    for (int j0 = 0; j0 < N; j0 += cj) {
        // ---- scale and store the slab
        if constexpr (PIPE) {
#pragma unroll
            for (int it = 0; it < IT; ++it) store_item(tid + nt * it, j0, g[it]);
            if (j0 + cj < N) {                             // in flight while this slab is multiplied
#pragma unroll
                for (int it = 0; it < IT; ++it) fetch_item(tid + nt * it, j0 + cj, g[it]);
            }
        } else {
            for (int e0 = 0; e0 < cj * ldG; e0 += nt) {    // uniform trip count (the DPP swap needs whole waves)
                fetch_item(e0 + tid, j0, g[0]);
                store_item(e0 + tid, j0, g[0]);
            }
        }
        __syncthreads();
        if (po != 2) {
            // mix the po columns of every output stage by Ls_k (lower): out_a = sum_{a' >= a} t_a' Ls[a'][a]
            for (int e = tid; e < cj * m * N; e += nt) {
                const int row = e / N, kq = e - row * N;
                lptr p = L.A + (size_t)row * ldG + kq * po;
                clptr Lk = L.Ls + (size_t)kq * po * po;
                double v[4];
                for (int a = 0; a < po; ++a) v[a] = p[a];
                for (int a = 0; a < po; ++a) {
                    double s = 0.0;
                    for (int a2 = a; a2 < po; ++a2) s = fma(v[a2], Lk[a2 * po + a], s);
                    p[a] = s;
                }
            }
            __syncthreads();
        }
        // ---- accumulate the upper tiles.  (Advancing the four tiles of a wave together, one k-step at a time with the
        // operands preloaded, measured slower inside the fused kernels: they sit at the 256-VGPR limit.)
        const int first = (j0 * po) >> 4;              // column tiles below hold only zeros in this slab
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            if (tI[sl] >= first) {
                clptr pa = L.A + kk * ldG + 16 * tI[sl] + l16, pb = L.A + kk * ldG + 16 * tJ[sl] + l16;
                for (int s = 0; s < rows_used; s += 4)
                    acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(size_t)s * ldG], pb[(size_t)s * ldG], acc[sl], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- K = I + acc, scaled symmetrically to a unit diagonal, into the tile store (region B: Theta^T is no longer
    // needed).  The scaling matters: active state rows put weights up to 1e12 into single stages of Ls, and the tile
    // factorisation below multiplies by explicit inverses of its diagonal tiles -- accurate only as far as those tiles
    // are reasonably conditioned (without it the reduced dual residual stalls at 1e-5 relative on the C2 QPs).
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        if (tI[sl] < 0 || tI[sl] != tJ[sl]) continue;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            if (kk + 4 * qd == l16) L.ks[16 * tI[sl] + l16] = 1.0 / sqrt(acc[sl][qd] + 1.0);
    }
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        if (tI[sl] < 0) continue;
        lptr T = L.B + (size_t)tile_index(tI[sl], tJ[sl], KT) * TSZ;
        const double sc = L.ks[16 * tJ[sl] + l16];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int r = kk + 4 * qd;
            const double v = acc[sl][qd] + ((tI[sl] == tJ[sl] && r == l16) ? 1.0 : 0.0);
            T[r * TS + l16] = v * L.ks[16 * tI[sl] + r] * sc;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ tile Cholesky K = R^T R (upper), in place
// diagonal tile: factor + inverse in registers on ONE wave -- the critical path of the factorisation.  Lane c of every row of 16
// lanes holds column c of the tile (a[r] = row r) and column c of the identity (b[r]); the elimination is a sequence of ROW
// operations M (M A = R, so M = R^-T) that carries the identity to M = Rinv^T -- no separate back-substitution for the inverse.
// The multiplier of a row operation is entry r of the scaled pivot row, i.e. a[s] of lane r: it reaches the FMA through the DPP
// network (`v_fmac_f64_dpp ... row_newbcast:r`, one instruction) instead of a v_readlane pair, a hazard s_nop and an FMA through
// an SGPR pair (rounds 2-5).  On one wave an f64 VALU instruction costs ~8 shader clocks whether it depends on the one before or
// not, the readlane-fed FMA ~18 (tools/probes/dpp_rate_probe.hip): the factorisation is bound by its instruction count, 5.1 k ->
// 3.7 k clocks per tile (tools/probes/chol16_probe.hip).  The four rows of 16 lanes do the same work on the same data (a broadcast
// cannot cross rows; splitting the rows of the tile over them costs more in pivot-row exchanges than it saves).
// The inline asm carries no wait states: tools/check_dpp_hazard.py (part of the guarded build) places the s_nop the final
// instruction order needs in front of a DPP read.  All 64 lanes must be active.
// acc <- acc - src[lane R of this row of 16 lanes] * oth
template <int R>
__device__ __forceinline__ void fnma_bc(double &acc, double src, double oth) {
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(oth), "n"(R));
}
template <int R>
__device__ __forceinline__ double mov_bc(double src) {
    double o;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(o) : "v"(src), "n"(R));
    return o;
}
// 1 / sqrt(p) for a positive normal p: v_rsq_f64 and the third-order correction of the library's rsqrt (without its class test:
// the caller flags a pivot that is not positive)
__device__ __forceinline__ double rsq3(double p) {
    const double y0 = __builtin_amdgcn_rsq(p), e = fma(-(p * y0), y0, 1.0);
    return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
template <int S, int R, int NPC>
__device__ __forceinline__ void chol16_rows(double (&a)[16], double (&b)[16]) {
    if constexpr (R < NPC) {
        fnma_bc<R>(a[R], a[S], a[S]);
        fnma_bc<R>(b[R], a[S], b[S]);
        chol16_rows<S, R + 1, NPC>(a, b);
    }
}
// A pivot that is not positive shows in the LAST scale factor: rsq of a negative number is NaN, of zero infinite, and either
// reaches every later pivot through the row operations -- one test at the end instead of one per pivot.
template <int S, int NPC>
__device__ __forceinline__ void chol16_steps(double (&a)[16], double (&b)[16], double y, double &ylast) {
    if constexpr (S < NPC) {
        a[S] *= y;
        b[S] *= y;
        double yn = 0.0;
        if constexpr (S + 1 < NPC) {
            fnma_bc<S + 1>(a[S + 1], a[S], a[S]);
            yn = rsq3(mov_bc<S + 1>(a[S + 1]));
            fnma_bc<S + 1>(b[S + 1], a[S], b[S]);
        } else ylast = y;
        chol16_rows<S, S + 2, NPC>(a, b);
        chol16_steps<S + 1, NPC>(a, b, yn, ylast);
    }
}
// STORE_T: write the factor back (qpc::r_times / rT_times read the diagonal tiles; the lean kernels only ever use Rinv).
// NPC: only the leading NPC x NPC block of the tile is not the identity (short horizons, ql::ipm_wave: N p_o < 16 outputs): the
// pivots and row operations beyond it are no-ops and are left out -- 45 row operations instead of 120 at NPC = 10.
template <bool STORE_T = true, int NPC = 16>
__device__ __forceinline__ bool chol16(lptr T, lptr Rinv) {
    static_assert(NPC >= 1 && NPC <= 16, "chol16: 1 <= NPC <= 16");
    const int lane = SRH_TID & 63, c = lane & 15, grp = lane >> 4;
    double a[16], b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a[r] = T[r * TS + c]; b[r] = r == c ? 1.0 : 0.0; }
    double ylast = 0.0;
    chol16_steps<0, NPC>(a, b, rsq3(mov_bc<0>(a[0])), ylast);
    const bool ok = ylast > 0.0 && ylast < INFINITY;
    if (grp == 0) {
        if constexpr (STORE_T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[r * TS + c] = (r <= c) ? a[r] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Rinv[c * TS + r] = b[r];          // b[r] = M[r][c] = Rinv[c][r]
    }
    return ok;
}

// T <- T - Ra^T Rb (one 16 x 16 x 16 product)
__device__ __forceinline__ void tile_update(lptr T, clptr Ra, clptr Rb, int l16, int kk) {
    wg::qp_d4 acc;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) acc[qd] = T[(kk + 4 * qd) * TS + l16];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ra[(4 * s + kk) * TS + l16], Rb[(4 * s + kk) * TS + l16], acc, 0, 0, 0);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) T[(kk + 4 * qd) * TS + l16] = acc[qd];
}

// Right-looking over tile rows.  Wave 0 owns the critical path: it updates the next diagonal tile first and factors it
// while the other waves finish the trailing update of the step.
__device__ __forceinline__ bool tile_cholesky(const QPDims &d, Lds &L) {
    const int KT = d.KT, tid = SRH_TID, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = blockDim.x >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    if (wave == 0) {
        const bool ok = chol16(L.B, L.Rinv);
        if (lane == 0) L.flag[1] = ok ? 1 : 0;
    }
    __syncthreads();
    for (int J = 0; J < KT; ++J) {
        if (L.flag[1] == 0) return false;
        // panel: R_JJ' = Rinv^T K_JJ'   (J' > J), one tile per wave
        clptr Ri = L.Rinv + (size_t)J * TSZ;
        for (int Jp = J + 1 + wave; Jp < KT; Jp += nw) {
            lptr T = L.B + (size_t)tile_index(J, Jp, KT) * TSZ;
            double av[4], bv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { av[s] = Ri[(4 * s + kk) * TS + l16]; bv[s] = T[(4 * s + kk) * TS + l16]; }
            wg::qp_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) T[(kk + 4 * qd) * TS + l16] = acc[qd];
        }
        __syncthreads();
        if (J + 1 >= KT) break;
        // trailing update: K_IK -= R_JI^T R_JK  for J < I <= K.  Tile (J+1, J+1) by wave 0, which then factors it at once
        const int rem = KT - J - 1, ntr = rem * (rem + 1) / 2;
        if (wave == 0) {
            clptr Ra = L.B + (size_t)tile_index(J, J + 1, KT) * TSZ;
            lptr T = L.B + (size_t)tile_index(J + 1, J + 1, KT) * TSZ;
            tile_update(T, Ra, Ra, l16, kk);
            __builtin_amdgcn_wave_barrier();
            const bool ok = chol16(T, L.Rinv + (size_t)(J + 1) * TSZ);
            if (lane == 0) L.flag[1] = ok ? 1 : 0;
        } else {
            for (int t = wave; t < ntr; t += nw - 1) {          // tiles 1 .. ntr-1 over waves 1 .. nw-1 (tile 0 is wave 0's)
                int tt = t, Ir = 0;
                while (tt >= rem - Ir) { tt -= rem - Ir; ++Ir; }
                const int I = J + 1 + Ir, Kc = I + tt;
                tile_update(L.B + (size_t)tile_index(I, Kc, KT) * TSZ, L.B + (size_t)tile_index(J, I, KT) * TSZ,
                            L.B + (size_t)tile_index(J, Kc, KT) * TSZ, l16, kk);
            }
        }
        __syncthreads();
    }
    return L.flag[1] != 0;
}

// ---- products with the factor (tile rows / columns over the waves; vectors of 16 KT entries in LDS)
// out = R x   (R upper: tile row J needs tiles (J, J') for J' >= J)
__device__ __forceinline__ void r_times(const QPDims &d, Lds &L, clptr x, lptr out) {
    const int KT = d.KT, tid = SRH_TID, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = blockDim.x >> 6;
    const int c = lane & 15, part = lane >> 4;
    for (int J = wave; J < KT; J += nw) {
        double acc = 0.0;
        for (int Jp = J; Jp < KT; ++Jp) {
            clptr T = L.B + (size_t)tile_index(J, Jp, KT) * TSZ;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) { const int k = 4 * part + kq; acc = fma(T[c * TS + k], x[16 * Jp + k], acc); }
        }
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (part == 0) out[16 * J + c] = acc;
    }
    __syncthreads();
}
// out = R^T x   (tile column J needs tiles (I, J) for I <= J)
__device__ __forceinline__ void rT_times(const QPDims &d, Lds &L, clptr x, lptr out) {
    const int KT = d.KT, tid = SRH_TID, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = blockDim.x >> 6;
    const int c = lane & 15, part = lane >> 4;
    for (int J = wave; J < KT; J += nw) {
        double acc = 0.0;
        for (int I = 0; I <= J; ++I) {
            clptr T = L.B + (size_t)tile_index(I, J, KT) * TSZ;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) { const int k = 4 * part + kq; acc = fma(T[k * TS + c], x[16 * I + k], acc); }
        }
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (part == 0) out[16 * J + c] = acc;
    }
    __syncthreads();
}

// v <- K^-1 v (in place, LDS vector of 16 KT entries) by wave 0; ends with a barrier
__device__ __forceinline__ void k_solve(const QPDims &d, Lds &L, lptr v) {
    const int KT = d.KT, tid = SRH_TID, lane = tid & 63;
    if (tid < 64) {
        const int c = lane & 15, part = lane >> 4;
        // forward: R^T z = v
        for (int J = 0; J < KT; ++J) {
            double acc = 0.0;
            for (int I = 0; I < J; ++I) {
                clptr T = L.B + (size_t)tile_index(I, J, KT) * TSZ;
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) { const int k = 4 * part + kq; acc = fma(T[k * TS + c], v[16 * I + k], acc); }
            }
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            const double tmp = v[16 * J + c] - acc;
            __builtin_amdgcn_wave_barrier();
            if (part == 0) L.Qu[c] = tmp;
            __builtin_amdgcn_wave_barrier();
            clptr Ri = L.Rinv + (size_t)J * TSZ;
            double z = 0.0;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) { const int k = 4 * part + kq; z = fma(Ri[k * TS + c], L.Qu[k], z); }     // Rinv^T
            z += __shfl_xor(z, 16, 64);
            z += __shfl_xor(z, 32, 64);
            __builtin_amdgcn_wave_barrier();
            if (part == 0) v[16 * J + c] = z;
            __builtin_amdgcn_wave_barrier();
        }
        // backward: R x = z
        for (int J = KT - 1; J >= 0; --J) {
            double acc = 0.0;
            for (int Jp = J + 1; Jp < KT; ++Jp) {
                clptr T = L.B + (size_t)tile_index(J, Jp, KT) * TSZ;
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) { const int k = 4 * part + kq; acc = fma(T[c * TS + k], v[16 * Jp + k], acc); }
            }
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            const double tmp = v[16 * J + c] - acc;
            __builtin_amdgcn_wave_barrier();
            if (part == 0) L.Qu[c] = tmp;
            __builtin_amdgcn_wave_barrier();
            clptr Ri = L.Rinv + (size_t)J * TSZ;
            double z = 0.0;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) { const int k = 4 * part + kq; z = fma(Ri[c * TS + k], L.Qu[k], z); }     // Rinv
            z += __shfl_xor(z, 16, 64);
            z += __shfl_xor(z, 32, 64);
            __builtin_amdgcn_wave_barrier();
            if (part == 0) v[16 * J + c] = z;
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
}

// r_j <- D_j^-1 r_j   (in place on an LDS u-space vector)
__device__ __forceinline__ void dinv_apply(const QPDims &d, Lds &L, lptr r) {
    const int m = d.m, tid = SRH_TID, nt = blockDim.x;
    if (d.diagD) {
        for (int e = tid; e < d.N * m; e += nt) { const double s = L.Ldi[e]; r[e] *= s * s; }
    } else if (tid < d.N) {
        clptr Li = L.Ldi + (size_t)tid * m * m;
        lptr v = r + (size_t)tid * m;
        for (int i = m - 1; i >= 0; --i) {               // y = Li v, bottom row first (in place)
            double s = 0.0;
            for (int k = 0; k <= i; ++k) s = fma(Li[i * m + k], v[k], s);
            v[i] = s;
        }
        for (int i = 0; i < m; ++i) {                    // v = Li^T y, top row first (in place)
            double s = 0.0;
            for (int k = i; k < m; ++k) s = fma(Li[k * m + i], v[k], s);
            v[i] = s;
        }
    }
    __syncthreads();
}

// per output stage: out_k = Ls_k in_k (FWD), Ls_k^T in_k (TR), Ls_k^-1 in_k (INV), Ls_k^-T in_k (INVT); padding zeroed
enum { LS_FWD = 0, LS_TR = 1, LS_INV = 2, LS_INVT = 3 };
template <int OP>
__device__ __forceinline__ void ls_apply(const QPDims &d, Lds &L, clptr in, lptr out) {
    const int N = d.N, po = d.po, ldG = qc_ldg(d), tid = SRH_TID, nt = blockDim.x;
    if (tid < N) {
        clptr Lk = L.Ls + (size_t)tid * po * po;
        double v[4] = {0.0, 0.0, 0.0, 0.0}, o[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < 4; ++a) if (a < po) v[a] = in[tid * po + a];
        if (OP == LS_FWD) {
#pragma unroll
            for (int a = 0; a < 4; ++a) if (a < po) { double s = 0.0;
#pragma unroll
                for (int b = 0; b < 4; ++b) if (b <= a) s = fma(Lk[a * po + b], v[b], s);
                o[a] = s; }
        } else if (OP == LS_TR) {
#pragma unroll
            for (int a = 0; a < 4; ++a) if (a < po) { double s = 0.0;
#pragma unroll
                for (int b = 0; b < 4; ++b) if (b >= a && b < po) s = fma(Lk[b * po + a], v[b], s);
                o[a] = s; }
        } else if (OP == LS_INV) {
#pragma unroll
            for (int a = 0; a < 4; ++a) if (a < po) { double s = v[a];
#pragma unroll
                for (int b = 0; b < 4; ++b) if (b < a) s = fma(-Lk[a * po + b], o[b], s);
                o[a] = s / Lk[a * po + a]; }
        } else {
#pragma unroll
            for (int a = 3; a >= 0; --a) if (a < po) { double s = v[a];
#pragma unroll
                for (int b = 0; b < 4; ++b) if (b > a && b < po) s = fma(-Lk[b * po + a], o[b], s);
                o[a] = s / Lk[a * po + a]; }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) if (a < po) out[tid * po + a] = o[a];
    }
    for (int e = N * po + tid; e < ldG; e += nt) out[e] = 0.0;
    __syncthreads();
}

// Newton direction.  In: gu in L.ta, gy in L.ya.  Out: du in L.du, dy = G du in L.dy.  If gyd != null the reduced dual
// residual max |gud + G^T gyd| (gud in L.tb) is returned through *rd -- its G^T pass rides along with the first one.
// Order of operations: the total gradient g = gu + G^T gy is formed in u-space FIRST (it is small near the solution; the
// two parts are not), and only then amplified by D^-1 -- algebraically equivalent short-cuts through (K - I) lose the
// cancellation and leave the dual residual at 1e-7.
__device__ __forceinline__ void newton_solve(const QPDims &d, QCWork &qw, Lds &L, clptr gyd, double *rd, Prof &pf) {
    const int nm = d.N * d.m, ldG = qc_ldg(d), tid = SRH_TID, nt = blockDim.x;
    QC_SUB(pf, 8);
    gT_times(d, qw, L, L.ya, gyd, L.du, gyd ? L.tc : (lptr) nullptr);            // du (scratch) = G^T gy; tc = G^T gyd
    QC_SUB(pf, 9);
    if (gyd) {
        double r = 0.0;
        for (int e = tid; e < nm; e += nt) r = fmax(r, fabs(L.tb[e] + L.tc[e]));
        *rd = wg::reduce(r, 1, L.red);
    }
    for (int e = tid; e < nm; e += nt) L.ta[e] = -(L.ta[e] + L.du[e]);        // rhs = -g
    __syncthreads();
    dinv_apply(d, L, L.ta);                                                   // t = D^-1 rhs
    QC_SUB(pf, 10);
    g_times(d, qw, L, L.ta, L.yb);                                            // G t
    QC_SUB(pf, 11);
    ls_apply<LS_TR>(d, L, L.yb, L.yc);                                        // Ls^T G t
    for (int e = tid; e < ldG; e += nt) L.yc[e] *= L.ks[e];                   // K^-1 = ks (ks K ks)^-1 ks
    __syncthreads();
    k_solve(d, L, L.yc);                                                      // v
    for (int e = tid; e < ldG; e += nt) L.yc[e] *= L.ks[e];
    __syncthreads();
    QC_SUB(pf, 12);
    ls_apply<LS_FWD>(d, L, L.yc, L.yd);                                       // Ls v
    gT_times(d, qw, L, L.yd, (clptr) nullptr, L.du, (lptr) nullptr);             // G^T Ls v
    QC_SUB(pf, 13);
    dinv_apply(d, L, L.du);
    for (int e = tid; e < nm; e += nt) L.du[e] = L.ta[e] - L.du[e];
    __syncthreads();
    QC_SUB(pf, 14);
    // dy = G du.  With w = ks v (in L.yc) the solved system reads (I + Ls^T Ky Ls) w = Ls^T G t, Ky = G D^-1 G^T, so that
    // G du = G t - Ky Ls w = Ls^-T w: a back substitution per output stage instead of the product (d.ls_pd: every Ls_k is
    // invertible), and the error of the K solve does not pass through K on its way into dy
    if (d.ls_pd) ls_apply<LS_INVT>(d, L, L.yc, L.dy);
    else g_times(d, qw, L, L.du, L.dy);
    QC_SUB(pf, 15);
}

// ------------------------------------------------------------------ the solve
// Results: w.u (the caller rolls the states out).  Returns 0 optimal, 1 max iterations, 2 numerical failure.
template <int MSEL, int NSEL>
__device__ __forceinline__ int solve(const QPDims &dfull, const QPConst &c, const QPDyn &dyn, const QPData &q, gptr work_base,
                                     lptr smem, QPLds &Lq, int *iters_out, QPWork &wout) {
    int tid = SRH_TID;                                       // re-read at the top of every interior-point iteration (dev_la.h: SRH_TID)
    const int nt = blockDim.x;
    QPDims d = dfull;                               // the QP without its trust-region rows
    d.tr = 0;
    d.nrx = d.nX;
    d.RX = d.nrx + d.nXf;
    d.NR = d.N * d.RX + d.N * d.nU;
    d.ng = d.N * d.nrx + d.nXf + d.N * d.nU;
    QPWork w;
    qp_carve(w, work_base, d);
    wout = w;
    QCWork qw;
    qc_carve(qw, work_base + dfull.qc_off, dfull);
    Lds L;
    lds_carve(L, smem, d, nt);
    const int N = d.N, m = d.m, po = d.po, nm = N * m, ldG = qc_ldg(d);
    // a QPLds view on this carve for the shared helpers (rollout, reductions)
    Lq.v1 = L.v1; Lq.v2 = L.v2; Lq.Qu = L.Qu; Lq.part = L.part; Lq.red = L.red; Lq.idxl = L.idxl; Lq.flag = L.flag;
    Prof pf;
#ifdef SRH_PROFILE
    long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 24; ++i) pf.t[i] = 0;
    long long tq_last = clock64();
    auto qlap = [&](int slot) { const long long now = clock64(); tq[slot] += now - tq_last; tq_last = now; };
#define QC_LAP(x) qlap(x)
#else
#define QC_LAP(x) ((void)0)
#endif
    for (int e = tid; e < nm; e += nt) { w.u[e] = 0.0; L.u[e] = 0.0; }
    for (int e = tid; e <= N; e += nt) w.s[e] = 0.0;
    for (int k = tid; k < N; k += nt) L.idxl[k] = dyn.idx ? dyn.idx[k] : k;
    for (int e = tid; e < d.nU * m; e += nt) L.UA[e] = c.UA[e];
    for (int e = tid; e < (d.nX + d.nXf) * po; e += nt) L.Tx[e] = e < d.nX * po ? c.Tx[e] : c.Txf[e - d.nX * po];
    __syncthreads();
    qp::rollout(d, dyn, q, w.u, w.x, Lq);
    QC_LAP(0);                                      // set-up + zero-input rollout
    condense<MSEL, NSEL>(d, c, dyn, w.x, qw, L);
    for (int e = tid; e < ldG; e += nt) { L.y[e] = L.yf[e]; L.dy[e] = 0.0; }
    __syncthreads();
    QC_LAP(2);                                      // condensation
    // ---- the rows of this thread: slot e = tid + nt q;  x rows (k-1) RX + r, then u rows N RX + k nU + r
    bool rv[QR], ru[QR];
    int rk[QR], rr[QR];
    double rh[QR], rt[QR], rlam[QR], rrg[QR], rrc[QR], rdt[QR], rdl[QR];
#pragma unroll
    for (int qi = 0; qi < QR; ++qi) {
        const int e = tid + nt * qi, nxs = N * d.RX;
        rt[qi] = rlam[qi] = rrg[qi] = rrc[qi] = rdt[qi] = rdl[qi] = 0.0;
        if (e < nxs) {
            rk[qi] = e / d.RX + 1; rr[qi] = e - (rk[qi] - 1) * d.RX; ru[qi] = false;
            rv[qi] = rr[qi] < qp::xrows_of(d, rk[qi]);
        } else {
            const int e2 = e - nxs;
            ru[qi] = true; rv[qi] = e2 < N * d.nU;
            rk[qi] = rv[qi] ? e2 / d.nU : 0; rr[qi] = rv[qi] ? e2 - rk[qi] * d.nU : 0;
        }
        rh[qi] = rv[qi] ? qp::row_h(d, c, q, ru[qi], rk[qi], rr[qi]) : 0.0;
    }
    auto row_val = [&](int qi, clptr vy, clptr vu) {              // a_row . (vy, vu)
        double acc = 0.0;
        if (!ru[qi]) { for (int a = 0; a < po; ++a) acc = fma(L.Tx[rr[qi] * po + a], vy[(rk[qi] - 1) * po + a], acc); }
        else { for (int j = 0; j < m; ++j) acc = fma(L.UA[rr[qi] * m + j], vu[rk[qi] * m + j], acc); }
        return acc;
    };
    int status = 1, it = 0;
    enum { INIT = 0, PRED = 1, CORR = 2 };
    int mode = INIT;
    double mu = 0.0, rp = 0.0, sig = 0.0, sd = 1.0, sp = 1.0, dreg = 0.0;
    bool near_opt = false;
    while (true) {
        tid = SRH_TID;
        QC_LAP(7);
        // ---------------- rows: weights D and gradient shifts rho of this Newton system -> L2 for the stage sums
        double musum = 0.0, rpm = 0.0;
#pragma unroll
        for (int qi = 0; qi < QR; ++qi) {
            if (!rv[qi]) continue;
            const int e = tid + nt * qi;
            if (mode == INIT) {
                const double g = row_val(qi, L.y, L.u) - rh[qi];
                w.D[e] = 1.0; w.rho[e] = g; rlam[qi] = 0.0;
            } else if (mode == PRED) {
                const double g = row_val(qi, L.y, L.u) - rh[qi];
                const double t = rt[qi], lam = rlam[qi], rg = g + t;
                rrg[qi] = rg;
                const double D = lam / (t + dreg * lam);
                w.D[e] = D; w.rho[e] = D * (rg + dreg * lam); w.lam[e] = lam;
                musum += lam * t;
                rpm = fmax(rpm, fabs(rg));
            } else {
                const double t = rt[qi], lam = rlam[qi];
                const double rc = lam * t + rdt[qi] * rdl[qi] - sig * mu;
                rrc[qi] = rc;
                w.rho[e] = lam + (lam * rrg[qi] - rc) / (t + dreg * lam);
            }
        }
        if (mode == PRED) {
            mu = wg::reduce(musum, 0, L.red) / d.ng;
            rp = wg::reduce(rpm, 1, L.red);
        }
        __syncthreads();
        QC_LAP(1);
        // ---------------- Newton system
        double rd = 0.0;
        bool ok = true;
        if (mode != CORR) {
            ok = stage_factors(d, c, w.D, L);
            QC_LAP(3);
            if (ok) {
                gram<MSEL>(d, qw, L);
                QC_LAP(4);
                ok = tile_cholesky(d, L);
                QC_LAP(5);
            }
        }
#ifdef SRH_PROFILE
        pf.last = clock64();
#endif
        if (ok) {
            if (mode == PRED) gradients(d, c, q, L, w.lam, L.tb, L.yg);          // multipliers: the dual residual
            gradients(d, c, q, L, w.rho, L.ta, L.ya);
            newton_solve(d, qw, L, mode == PRED ? (clptr)L.yg : (clptr) nullptr, &rd, pf);
        }
        QC_LAP(6);
        // ---------------- use the direction
        if (mode == INIT) {
            if (!ok) { status = 2; break; }
            for (int e = tid; e < nm; e += nt) L.u[e] += L.du[e];
            for (int e = tid; e < ldG; e += nt) L.y[e] += L.dy[e];
            __syncthreads();
            if (d.ng == 0) { status = 0; break; }
            double zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
            for (int qi = 0; qi < QR; ++qi) {
                if (!rv[qi]) continue;
                const double g = row_val(qi, L.y, L.u) - rh[qi];
                rrg[qi] = g;
                zmin = fmin(zmin, g); zmax = fmax(zmax, g);
            }
            zmin = wg::reduce(zmin, 2, L.red);
            zmax = wg::reduce(zmax, 1, L.red);
            const double sh_t = zmax >= 0.0 ? 1.0 + zmax : 0.0, sh_l = zmin <= 0.0 ? 1.0 - zmin : 0.0;
#pragma unroll
            for (int qi = 0; qi < QR; ++qi) { rt[qi] = -rrg[qi] + sh_t; rlam[qi] = rrg[qi] + sh_l; }
            for (int e = tid; e < d.n; e += nt) {
                double g = 0.0;
                if (q.z) for (int a = 0; a < d.nz; ++a) g = fma(c.HtQz2[e * d.nz + a], -q.z[d.nz + a], g);
                sd = fmax(sd, fabs(g));
            }
            for (int e = tid; e < d.nU; e += nt) sp = fmax(sp, fabs(c.Ub[e]));
            sd = fmax(wg::reduce(sd, 1, L.red), q.omega);
            sp = fmax(wg::reduce(sp, 1, L.red), fabs(q.delta));
            dreg = d.reg / sd;
            mode = PRED;
            continue;
        }
        // row directions  dt, dlam  and the largest step that keeps t, lam positive
        double amax = 1e300;
        if (ok) {
#pragma unroll
            for (int qi = 0; qi < QR; ++qi) {
                if (!rv[qi]) continue;
                const double t = rt[qi], lam = rlam[qi], rga = rrg[qi] + row_val(qi, L.dy, L.du);
                const double dl = ((mode == PRED ? -lam * t : -rrc[qi]) + lam * rga) / (t + dreg * lam);
                const double dtv = -rga + dreg * dl;
                rdl[qi] = dl; rdt[qi] = dtv;
                if (dtv < 0.0) amax = fmin(amax, -t / dtv);
                if (dl < 0.0) amax = fmin(amax, -lam / dl);
            }
        }
        amax = wg::reduce(amax, 2, L.red);
        if (mode == PRED) {
            if (!ok) { status = near_opt ? 0 : 2; break; }
            if (!(mu == mu)) { status = near_opt ? 0 : 5; break; }
            if (!(rd == rd)) { status = near_opt ? 0 : 6; break; }
            if (q.dbg && tid == 0) { gptr g = q.dbg + 8 * it; g[0] = mu; g[1] = rd; g[2] = rp; g[3] = sd; g[4] = sp; }
            const double ltol = fmax(d.tol, 1e-9);
            if (rd <= ltol * sd && rp <= ltol * sp && mu <= d.tol) { status = 0; break; }
            near_opt = (rd <= 1e-8 * sd && rp <= 1e-8 * sp && mu <= 1e-8);
            if (it >= d.max_iter) { status = 1; break; }
            const double a_aff = fmin(1.0, amax);
            double ma = 0.0;
#pragma unroll
            for (int qi = 0; qi < QR; ++qi)
                if (rv[qi]) ma += (rlam[qi] + a_aff * rdl[qi]) * (rt[qi] + a_aff * rdt[qi]);
            const double mu_aff = wg::reduce(ma, 0, L.red) / d.ng;
            sig = mu > 0.0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
            if (q.dbg && tid == 0) { gptr g = q.dbg + 8 * it; g[5] = a_aff; g[6] = sig; }
            mode = CORR;
            continue;
        }
        // mode == CORR: step
        if (!ok) { status = 2; break; }
        const double a = fmin(1.0, 0.99 * amax);
        if (q.dbg && tid == 0) { gptr g = q.dbg + 8 * it; g[7] = a; }
        for (int e = tid; e < nm; e += nt) L.u[e] += a * L.du[e];
        for (int e = tid; e < ldG; e += nt) L.y[e] += a * L.dy[e];
#pragma unroll
        for (int qi = 0; qi < QR; ++qi) { rt[qi] += a * rdt[qi]; rlam[qi] += a * rdl[qi]; }
        __syncthreads();
        ++it;
        mode = PRED;
    }
    __syncthreads();
    for (int e = tid; e < nm; e += nt) w.u[e] = L.u[e];
    __syncthreads();
#ifdef SRH_PROFILE
    if (q.dbg && tid == 0) for (int i = 0; i < 8; ++i) { q.dbg[8 * 60 + i] = (double)tq[i]; q.dbg[8 * 59 + i] = (double)pf.t[8 + i]; }
#endif
    if (iters_out) *iters_out = it;
    return status;
}

}  // namespace qpc
