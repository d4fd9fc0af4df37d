// Condensed (output-space) interior point for the LOCP QP without its trust-region rows -- the fast path of qp::solve.
//
// The cost (sofacontrol/scp/locp.py:226-252) and the state rows X / Xf (locp.py:330-337) see the state x_k only through
// a few OUTPUT directions: 2 H^T Qz H = Cq^T Cq, X.A and Xf.A all lie in the row space of C_o (po x n; Diamond and Trunk
// drivers: the tip x / y rows, po = 2).  Eliminating the states with the dynamics,
//     y_k = C_o x_k = yfree_k + sum_{j<k} G[k][j] u_j ,   G[k][j] = C_o A_{k-1} ... A_{j+1} B_j   (po x m),
// leaves a QP in u alone (N m variables) whose interior-point Newton systems are
//     M du = rhs ,  M = blkdiag(2R + U.A^T D_u U.A) + G^T blkdiag(S_k) G ,  S_k = Tc^T Tc + Tx^T D_x,k Tx  (po x po).
// They are solved in OUTPUT space (Woodbury): with D_j = Ld_j Ld_j^T, S_k = Ls_k Ls_k^T, Gd = Ls^T G Ld^-T,
//     K = I + Gd Gd^T   (N po x N po: 100 x 100 for both robots at N = 50),   K v = Ls^T G D^-1 rhs ,
//     du = D^-1 (rhs - G^T Ls v) .
// Per interior-point iteration: one Gram product (f64 MFMA, accumulated in registers over slabs of the scaled G staged
// through LDS), one tile Cholesky of K in LDS (diagonal 16 x 16 tiles factored and inverted in registers with
// v_readlane broadcasts, panel / trailing updates on MFMA), and a handful of mat-vecs with G -- ~2.5 MFLOP instead of
// the ~50 MFLOP of a backward Riccati factorisation with n_x = 60 states.  Once per QP: G by the adjoint recursion
//     Theta_{j-1} = [C_o ; Theta_j A_j] ,  G[:, j] = Theta_j B_j
// as one MFMA product per stage with the same LDS panel [A_j | B_j] the Riccati path uses.
//
// The iteration (Mehrotra predictor-corrector, starting point, regularised weights, stopping rule) is the one of
// qp::solve / oracle/riccati_ipm.py; the numpy statement of THIS file is oracle/condensed_ipm.py (newton='output').
// If the minimiser leaves the trust region, or a factorisation breaks down, qp::solve falls back to the stage-wise
// Riccati solve of the full QP.
#pragma once

struct QCWork {                        // per-problem scratch in HBM/L2 (doubles), behind the QPWork block
    gptr GT;                           // (N m) x ldG : GT[(j,b)][(k-1) po + a] = G[k][a][j][b]; zero where k <= j
    gptr yf, y, dy;                    // (N+1) x po : free response, current outputs, output step (index k po + a)
};

__host__ __device__ inline int qc_ldg(const QPDims &d) { return 16 * d.KT; }
__host__ __device__ inline size_t qc_work_doubles(const QPDims &d) {
    if (!d.cond) return 0;
    return (size_t)d.N * d.m * qc_ldg(d) + 3 * (size_t)(d.N + 1) * d.po + 8;
}
__device__ inline void qc_carve(QCWork &w, gptr base, const QPDims &d) {
    gptr p = base;
    auto take = [&](size_t c) { gptr q = p; p += c; return q; };
    w.GT = take((size_t)d.N * d.m * qc_ldg(d));
    w.yf = take((size_t)(d.N + 1) * d.po); w.y = take((size_t)(d.N + 1) * d.po); w.dy = take((size_t)(d.N + 1) * d.po);
}

namespace qpc {

constexpr int TS = 17;                 // row stride of a 16 x 16 LDS tile (odd: row and column accesses conflict free)
constexpr int TSZ = 16 * TS;
constexpr int SR = 32;                 // rows (contraction length) of a Gram slab

struct Lds {
    lptr A;        // [A | B] panel (n16 x ld) while condensing; Gram slab (SR x ldG) afterwards
    lptr B;        // Theta^T (n16 x ldT) while condensing; upper tiles of K / its Cholesky factor afterwards
    lptr Rinv;     // KT tiles: inverses of the diagonal tiles of the factor
    lptr Ldi;      // N x m x m : Ld_j^-1 (lower)
    lptr Ls;       // N x po x po : Ls_k (lower), index k - 1
    lptr ua, ub, uc;           // u-space vectors (N m)
    lptr ya, yb, yc, yd;       // y-space vectors (ldG; index (k-1) po + a)
    lptr v1, v2, Qu, part, red;
    liptr flag, idxl;
};

__host__ __device__ inline size_t lds_doubles(const QPDims &d, int nthreads) {
    // the panel and Theta^T need rows up to the contraction extent roundup4(n) only (NK; zero padded)
    const size_t n16 = (size_t)d.NK, ldG = qc_ldg(d), ldT = ldG + 1, nm = (size_t)d.N * d.m;
    const size_t regA = n16 * d.ld > (size_t)SR * ldG ? n16 * d.ld : (size_t)SR * ldG;
    const size_t tiles = (size_t)d.KT * (d.KT + 1) / 2 * TSZ;
    const size_t regB = n16 * ldT > tiles ? n16 * ldT : tiles;
    return regA + regB + (size_t)d.KT * TSZ + nm * d.m + (size_t)d.N * d.po * d.po + 3 * ((nm + 3) & ~(size_t)3) + 4 * ldG +
           2 * (size_t)d.ld + 16 + nthreads + 16 + 4 + (size_t)(d.N / 2 + 2);
}

__device__ inline void lds_carve(Lds &L, lptr base, const QPDims &d, int nthreads) {
    lptr p = base;
    auto take = [&](size_t c) { lptr q = p; p += c; return q; };
    const size_t n16 = (size_t)d.NK, ldG = qc_ldg(d), ldT = ldG + 1, nm = (size_t)d.N * d.m, nm4 = (nm + 3) & ~(size_t)3;
    const size_t regA = n16 * d.ld > (size_t)SR * ldG ? n16 * d.ld : (size_t)SR * ldG;
    const size_t tiles = (size_t)d.KT * (d.KT + 1) / 2 * TSZ;
    const size_t regB = n16 * ldT > tiles ? n16 * ldT : tiles;
    L.A = take(regA); L.B = take(regB); L.Rinv = take((size_t)d.KT * TSZ);
    L.Ldi = take(nm * d.m); L.Ls = take((size_t)d.N * d.po * d.po);
    L.ua = take(nm4); L.ub = take(nm4); L.uc = take(nm4);
    L.ya = take(ldG); L.yb = take(ldG); L.yc = take(ldG); L.yd = take(ldG);
    L.v1 = take(d.ld); L.v2 = take(d.ld); L.Qu = take(16); L.part = take(nthreads); L.red = take(16);
    L.flag = (liptr)take(4);
    L.idxl = (liptr)take((size_t)(d.N / 2 + 2));
}

__device__ __forceinline__ double readlane_d(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int tile_index(int I, int J, int KT) { return I * KT - I * (I - 1) / 2 + (J - I); }   // I <= J

// row coefficients of x-stage k, row r, in output coordinates
__device__ __forceinline__ cgptr xrow_T(const QPDims &d, const QPConst &c, int r) {
    return r < d.nX ? c.Tx + (size_t)r * d.po : c.Txf + (size_t)(r - d.nX) * d.po;
}

// out[row] = a_row . (vy, vu)     (d.tr == 0 in this mode: the x rows are the X / Xf rows)
__device__ __forceinline__ void rows_apply(const QPDims &d, const QPConst &c, cgptr vy, cgptr vu, gptr out) {
    const int po = d.po, m = d.m;
    qp::for_rows(d, [&](int row, bool isU, int k, int r) {
        double acc = 0.0;
        if (!isU) {
            cgptr t = xrow_T(d, c, r);
            for (int a = 0; a < po; ++a) acc = fma(t[a], vy[(size_t)k * po + a], acc);
        } else {
            cgptr ua = c.UA + (size_t)r * m, uk = vu + (size_t)k * m;
            for (int j = 0; j < m; ++j) acc = fma(ua[j], uk[j], acc);
        }
        out[row] = acc;
    });
}

// ------------------------------------------------------------------ mat-vecs with G (HBM/L2 resident)
// yv[i] (+)= sum_rows GT[row][i] uv[row]   (uv in LDS, yv in LDS; row (j,b) only reaches columns i >= j po)
__device__ __forceinline__ void g_times(const QPDims &d, const QCWork &w, Lds &L, clptr uv, lptr yv) {
    const int ldG = qc_ldg(d), m = d.m, po = d.po, nm = d.N * m, tid = threadIdx.x, nt = blockDim.x;
    const int CW = 128, G = nt / CW;                       // column lanes per group, row groups
    const int col = tid % CW, grp = tid / CW;
    double acc = 0.0;
    if (col < ldG && grp < G) {
        int rmax = (col / po + 1) * m;                     // rows (j,b) with j <= col / po
        if (rmax > nm) rmax = nm;
        cgptr g = w.GT + col;
#pragma unroll 4
        for (int r = grp; r < rmax; r += G) acc = fma(g[(size_t)r * ldG], uv[r], acc);
    }
    if (grp < G) L.part[grp * CW + col] = acc;
    __syncthreads();
    if (tid < ldG) {
        double s = 0.0;
        for (int q = 0; q < G; ++q) s += L.part[q * CW + tid];
        yv[tid] = s;
    }
    __syncthreads();
}

// out1[row] = sum_i GT[row][i] y1[i]  (and out2 with y2 when y2 != null): 8 lanes per row
__device__ __forceinline__ void gT_times(const QPDims &d, const QCWork &w, clptr y1, clptr y2, lptr out1, lptr out2) {
    const int ldG = qc_ldg(d), m = d.m, po = d.po, nm = d.N * m, tid = threadIdx.x, nt = blockDim.x;
    const int g8 = tid & 7;
    for (int r0 = 0; r0 < nm; r0 += nt / 8) {              // uniform trip count
        const int r = r0 + (tid >> 3);
        double a1 = 0.0, a2 = 0.0;
        if (r < nm) {
            const int i0 = (r / m) * po;                   // first column the row reaches
            cgptr g = w.GT + (size_t)r * ldG;
            for (int i = (i0 & ~7) + g8; i < ldG; i += 8) {
                const double gv = g[i];
                a1 = fma(gv, y1[i], a1);
                if (y2) a2 = fma(gv, y2[i], a2);
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
// G by the adjoint recursion; Theta^T lives in L.B (n16 x ldT, column i = (k-1) po + a), the stage panel in L.A.
template <int MSEL, int NSEL>
__device__ __forceinline__ void condense(const QPDims &d, const QPConst &c, const QPDyn &dyn, cgptr x, QCWork &w, Lds &L) {
    const int N = d.N, n = d.n, m = d.m, po = d.po, ld = d.ld, KT = d.KT, ldG = qc_ldg(d), ldT = ldG + 1;
    const int n16 = d.NK, NPa = d.NPa;              // rows of the panel / of Theta^T: the contraction extent
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    for (int e = tid; e < (N + 1) * po; e += nt) {
        const int k = e / po, a = e - k * po;
        double v = 0.0;
        for (int j = 0; j < n; ++j) v = fma(c.Co[(size_t)a * n + j], x[(size_t)k * n + j], v);
        w.yf[e] = v;
    }
    for (int e = tid; e < n16 * ldT; e += nt) L.B[e] = 0.0;
    for (int e = tid; e < n16 * ld; e += nt) L.A[e] = 0.0;
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
            double bop[(NSEL > 0 ? (NSEL + 3) / 4 : 32)];
            constexpr int KSMAX = NSEL > 0 ? (NSEL + 3) / 4 : 32;
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
// gu (N m) -> L.ua, gy (ldG) -> L.ya from the weights `wx` / `wu` = rho (right-hand side) or lam (dual residual)
__device__ __forceinline__ void gradients(const QPDims &d, const QPConst &c, const QPData &q, const QPWork &w, const QCWork &qw,
                                          cgptr wrow, lptr gu, lptr gy) {
    const int N = d.N, m = d.m, po = d.po, nz = d.nz, ldG = qc_ldg(d), tid = threadIdx.x, nt = blockDim.x;
    for (int e = tid; e < ldG; e += nt) {
        double g = 0.0;
        if (e < N * po) {
            const int k = e / po + 1, a = e - (k - 1) * po;
            cgptr S = (k == N) ? c.ScN : c.Sc;
            for (int b = 0; b < po; ++b) g = fma(S[a * po + b], qw.y[(size_t)k * po + b], g);
            if (q.z) for (int b = 0; b < nz; ++b) g = fma(-c.Cz2[a * nz + b], q.z[(size_t)k * nz + b], g);
            if (k == N && c.Qzf && q.zf) for (int b = 0; b < nz; ++b) g = fma(-c.Czf2[a * nz + b], q.zf[b], g);
            const int nr = qp::xrows_of(d, k);
            cgptr wr = wrow + (size_t)(k - 1) * d.RX;
            for (int r = 0; r < nr; ++r) g = fma(xrow_T(d, c, r)[a], wr[r], g);
        }
        gy[e] = g;
    }
    cgptr wu = wrow + (size_t)N * d.RX;
    for (int e = tid; e < N * m; e += nt) {
        const int k = e / m, a = e - k * m;
        double g = 0.0;
        for (int b = 0; b < m; ++b) g = fma(c.R2[a * m + b], w.u[(size_t)k * m + b] - (q.ud ? q.ud[(size_t)k * m + b] : 0.0), g);
        for (int r = 0; r < d.nU; ++r) g = fma(c.UA[r * m + a], wu[(size_t)k * d.nU + r], g);
        gu[e] = g;
    }
    __syncthreads();
}

// in-place Cholesky factor (lower, row-major m x m; the strict upper part is zeroed) of a tiny SPD matrix in LDS, by one
// thread.  false: not positive definite.
__device__ __forceinline__ bool small_chol(lptr A, int m) {
    double dmax = 0.0;
    for (int i = 0; i < m; ++i) dmax = fmax(dmax, fabs(A[i * m + i]));
    for (int i = 0; i < m; ++i)
        for (int j = 0; j <= i; ++j) {
            double sum = A[i * m + j];
            for (int k = 0; k < j; ++k) sum = fma(-A[i * m + k], A[j * m + k], sum);
            if (i == j) {
                if (!(sum > 1e-300 * dmax)) return false;
                A[i * m + i] = sqrt(sum);
            } else {
                A[i * m + j] = sum / A[j * m + j];
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

// D_j = 2R + U.A^T D_u U.A -> Ld_j^-1 ;  S_k = S*_k + T^T D_x T -> Ls_k.  Returns false when one of them is not PD.
__device__ __forceinline__ bool stage_factors(const QPDims &d, const QPConst &c, const QPWork &w, Lds &L) {
    const int N = d.N, m = d.m, po = d.po, tid = threadIdx.x, nt = blockDim.x;
    cgptr Du = w.D + (size_t)N * d.RX;
    for (int e = tid; e < N * m * m; e += nt) {
        const int k = e / (m * m), ab = e - k * m * m, a = ab / m, b = ab - a * m;
        double v = c.R2[ab];
        for (int r = 0; r < d.nU; ++r) v = fma(c.UA[r * m + a] * Du[(size_t)k * d.nU + r], c.UA[r * m + b], v);
        L.Ldi[e] = v;
    }
    for (int e = tid; e < N * po * po; e += nt) {
        const int k = e / (po * po) + 1, ab = e - (k - 1) * po * po, a = ab / po, b = ab - a * po;
        double v = (k == N ? c.ScN : c.Sc)[ab];
        const int nr = qp::xrows_of(d, k);
        cgptr Dk = w.D + (size_t)(k - 1) * d.RX;
        for (int r = 0; r < nr; ++r) { cgptr t = xrow_T(d, c, r); v = fma(t[a] * Dk[r], t[b], v); }
        L.Ls[e] = v;
    }
    if (tid == 0) L.flag[0] = 1;
    __syncthreads();
    bool ok = true;
    // one thread per stage; the input blocks on the first waves, the output blocks from thread 256 on (other SIMDs)
    if (tid < N) {
        lptr A = L.Ldi + (size_t)tid * m * m;
        ok = small_chol(A, m);
        if (ok) tri_inverse(A, m);
    }
    const int t2 = tid - (nt >= 512 ? 256 : 64);
    if (t2 >= 0 && t2 < N) ok = small_chol(L.Ls + (size_t)t2 * po * po, po);
    if (!ok) L.flag[0] = 0;
    __syncthreads();
    return L.flag[0] != 0;
}

// ------------------------------------------------------------------ Gram matrix K = I + Gd Gd^T -> upper tiles in L.B
template <int MSEL>
__device__ __forceinline__ void gram(const QPDims &d, const QCWork &w, Lds &L) {
    constexpr int MB = MSEL > 0 ? MSEL : 16;       // bound of the register block (rows of a G^T block)
    const int N = d.N, m = d.m, po = d.po, KT = d.KT, ldG = qc_ldg(d);
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, nw = nt >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    const int cj = SR / m > 0 ? SR / m : 1;        // stages per slab
    const int rows_used = cj * m;
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
    for (int e = tid; e < SR * ldG; e += nt) L.A[e] = 0.0;      // K padding columns / unused rows stay zero
    __syncthreads();
    for (int j0 = 0; j0 < N; j0 += cj) {
        // ---- fill: block (j, k) of G^T (m x po) -> Ld_j^-1 . blk . Ls_k
        for (int e = tid; e < cj * N; e += nt) {
            const int jj = e / N, k = e - jj * N + 1, j = j0 + jj;
            lptr dst = L.A + (size_t)(jj * m) * ldG + (k - 1) * po;
            if (j >= N || k <= j) {
                for (int b = 0; b < m; ++b)
                    for (int a = 0; a < po; ++a) dst[b * ldG + a] = 0.0;
                continue;
            }
            clptr Li = L.Ldi + (size_t)j * m * m, Lk = L.Ls + (size_t)(k - 1) * po * po;
            for (int a = 0; a < po; ++a) {               // one column of the block at a time: t = Ld^-1 g
                double g[MB];
#pragma unroll
                for (int b = 0; b < MB; ++b) g[b] = b < m ? w.GT[((size_t)j * m + b) * ldG + (k - 1) * po + a] : 0.0;
#pragma unroll
                for (int b = MB - 1; b >= 0; --b) {      // lower-triangular product, bottom row first (in place)
                    if (b < m) {
                        double s = 0.0;
#pragma unroll
                        for (int b2 = 0; b2 < MB; ++b2) if (b2 <= b) s = fma(Li[b * m + b2], g[b2], s);
                        g[b] = s;
                    }
                }
                // times Ls_k (lower): out[:, a2] += t * Ls[a][a2] for a2 <= a
#pragma unroll
                for (int b = 0; b < MB; ++b)
                    if (b < m)
                        for (int a2 = 0; a2 <= a; ++a2) {
                            const double add = g[b] * Lk[a * po + a2];
                            dst[b * ldG + a2] = (a == a2 ? 0.0 : dst[b * ldG + a2]) + add;   // first touch of column a2 is a == a2
                        }
            }
        }
        __syncthreads();
        // ---- accumulate the upper tiles
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
    // ---- K = I + acc into the tile store (region B: Theta^T is no longer needed)
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        if (tI[sl] < 0) continue;
        lptr T = L.B + (size_t)tile_index(tI[sl], tJ[sl], KT) * TSZ;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int r = kk + 4 * qd;
            T[r * TS + l16] = acc[sl][qd] + ((tI[sl] == tJ[sl] && r == l16) ? 1.0 : 0.0);
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ tile Cholesky K = R^T R (upper), in place
// diagonal tile: factor + inverse in registers, lane c (mod 16) holds column c; values of other columns by v_readlane
__device__ __forceinline__ bool chol16(lptr T, lptr Rinv) {
    const int c = threadIdx.x & 15;
    double a[16], x[16], dinv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = T[r * TS + c];
    bool ok = true;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const double piv = readlane_d(a[s], s);
        ok = ok && (piv > 0.0);
        const double di = rsqrt(piv);
        // one Newton step on the reciprocal square root: full double accuracy
        const double di2 = di * (1.5 - 0.5 * piv * di * di);
        dinv[s] = di2;
        a[s] *= di2;
#pragma unroll
        for (int r = s + 1; r < 16; ++r) a[r] = fma(-readlane_d(a[s], r), a[s], a[r]);
    }
#pragma unroll
    for (int r = 15; r >= 0; --r) {
        double sum = (r == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = r + 1; k < 16; ++k) sum = fma(-readlane_d(a[r], k), x[k], sum);
        x[r] = (r <= c) ? sum * dinv[r] : 0.0;
    }
    if ((threadIdx.x & 63) < 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            T[r * TS + c] = (r <= c) ? a[r] : 0.0;
            Rinv[r * TS + c] = x[r];
        }
    }
    return ok;
}

__device__ __forceinline__ bool tile_cholesky(const QPDims &d, Lds &L) {
    const int KT = d.KT, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    for (int J = 0; J < KT; ++J) {
        if (wave == 0) {
            const bool ok = chol16(L.B + (size_t)tile_index(J, J, KT) * TSZ, L.Rinv + (size_t)J * TSZ);
            if (lane == 0) L.flag[1] = ok ? 1 : 0;
        }
        __syncthreads();
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
        // trailing update: K_IK -= R_JI^T R_JK  for J < I <= K
        const int rem = KT - J - 1, ntr = rem * (rem + 1) / 2;
        for (int t = wave; t < ntr; t += nw) {
            int tt = t, Ir = 0;
            while (tt >= rem - Ir) { tt -= rem - Ir; ++Ir; }
            const int I = J + 1 + Ir, Kc = I + tt;
            clptr Ra = L.B + (size_t)tile_index(J, I, KT) * TSZ, Rb = L.B + (size_t)tile_index(J, Kc, KT) * TSZ;
            lptr T = L.B + (size_t)tile_index(I, Kc, KT) * TSZ;
            wg::qp_d4 acc;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) acc[qd] = T[(kk + 4 * qd) * TS + l16];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ra[(4 * s + kk) * TS + l16], Rb[(4 * s + kk) * TS + l16], acc, 0, 0, 0);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) T[(kk + 4 * qd) * TS + l16] = acc[qd];
        }
        __syncthreads();
    }
    return true;
}

// v <- K^-1 v (in place, LDS vector of 16 KT entries) by wave 0; ends with a barrier
__device__ __forceinline__ void k_solve(const QPDims &d, Lds &L, lptr v) {
    const int KT = d.KT, tid = threadIdx.x, lane = tid & 63;
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

// t_j <- D_j^-1 r_j = Ld^-T (Ld^-1 r_j)   (in place on an LDS u-space vector; one thread per stage)
__device__ __forceinline__ void dinv_apply(const QPDims &d, Lds &L, lptr r) {
    const int m = d.m, tid = threadIdx.x;
    if (tid < d.N) {
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

// Newton direction for the right-hand side -(gu + G^T gy) held as gu in L.ua, gy in L.ya:
//   du -> w.du (global) and L.uc, dy -> qw.dy.  The factors (stage_factors, gram, tile_cholesky) must be current.
__device__ __forceinline__ void newton_solve(const QPDims &d, QPWork &w, QCWork &qw, Lds &L) {
    const int N = d.N, m = d.m, po = d.po, nm = N * m, ldG = qc_ldg(d), tid = threadIdx.x, nt = blockDim.x;
    gT_times(d, qw, L.ya, (clptr) nullptr, L.ub, (lptr) nullptr);              // ub = G^T gy
    for (int e = tid; e < nm; e += nt) L.ub[e] = -(L.ua[e] + L.ub[e]);          // rhs
    __syncthreads();
    for (int e = tid; e < nm; e += nt) L.uc[e] = L.ub[e];
    __syncthreads();
    dinv_apply(d, L, L.uc);                                                     // t = D^-1 rhs
    g_times(d, qw, L, L.uc, L.yb);                                              // yb = G t
    if (tid < N) {                                                              // yc_k = Ls_k^T yb_k
        clptr Lk = L.Ls + (size_t)tid * po * po;
        for (int a = 0; a < po; ++a) {
            double s = 0.0;
            for (int b = a; b < po; ++b) s = fma(Lk[b * po + a], L.yb[tid * po + b], s);
            L.yc[tid * po + a] = s;
        }
    }
    for (int e = N * po + tid; e < ldG; e += nt) L.yc[e] = 0.0;
    __syncthreads();
    k_solve(d, L, L.yc);                                                        // v
    if (tid < N) {                                                              // yb_k = Ls_k v_k
        clptr Lk = L.Ls + (size_t)tid * po * po;
        for (int a = 0; a < po; ++a) {
            double s = 0.0;
            for (int b = 0; b <= a; ++b) s = fma(Lk[a * po + b], L.yc[tid * po + b], s);
            L.yb[tid * po + a] = s;
        }
    }
    for (int e = N * po + tid; e < ldG; e += nt) L.yb[e] = 0.0;
    __syncthreads();
    gT_times(d, qw, L.yb, (clptr) nullptr, L.ub, (lptr) nullptr);              // ub = G^T Ls v
    dinv_apply(d, L, L.ub);
    for (int e = tid; e < nm; e += nt) { const double v = L.uc[e] - L.ub[e]; L.uc[e] = v; w.du[e] = v; }
    __syncthreads();
    g_times(d, qw, L, L.uc, L.yd);                                              // dy = G du
    for (int e = tid; e < (N + 1) * po; e += nt) qw.dy[e] = e < po ? 0.0 : L.yd[e - po];
    __syncthreads();
}

// ------------------------------------------------------------------ the solve
// Results: w.u (and w.x by the caller's final rollout).  Returns 0 optimal, 1 max iterations, 2 numerical failure.
template <int MSEL, int NSEL>
__device__ __forceinline__ int solve(const QPDims &dfull, const QPConst &c, const QPDyn &dyn, const QPData &q, gptr work_base,
                                     lptr smem, QPLds &Lq, int *iters_out, QPWork &wout) {
    const int tid = threadIdx.x, nt = blockDim.x;
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
    const int N = d.N, m = d.m, po = d.po, nm = N * m;
    // a QPLds view on this carve for the shared helpers (rollout, reductions)
    Lq.v1 = L.v1; Lq.v2 = L.v2; Lq.Qu = L.Qu; Lq.part = L.part; Lq.red = L.red; Lq.idxl = L.idxl; Lq.flag = L.flag;
    for (int e = tid; e < nm; e += nt) w.u[e] = 0.0;
    for (int e = tid; e <= N; e += nt) w.s[e] = 0.0;
    for (int k = tid; k < N; k += nt) L.idxl[k] = dyn.idx ? dyn.idx[k] : k;
    __syncthreads();
    qp::rollout(d, dyn, q, w.u, w.x, Lq);
    condense<MSEL, NSEL>(d, c, dyn, w.x, qw, L);
    for (int e = tid; e < (N + 1) * po; e += nt) { qw.y[e] = qw.yf[e]; qw.dy[e] = 0.0; }
    __syncthreads();
    int status = 1, it = 0;
    enum { INIT = 0, PRED = 1, CORR = 2 };
    int mode = INIT;
    double mu = 0.0, rp = 0.0, sig = 0.0, sd = 1.0, sp = 1.0, dreg = 0.0;
    bool near_opt = false;
#ifdef SRH_PROFILE
    long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tq_last = clock64();
    auto qlap = [&](int slot) { const long long now = clock64(); tq[slot] += now - tq_last; tq_last = now; };
#define QC_LAP(x) qlap(x)
#else
#define QC_LAP(x) ((void)0)
#endif
    QC_LAP(0);                                      // rollout + condensation
    while (true) {
        QC_LAP(7);
        if (mode != CORR) {
            rows_apply(d, c, qw.y, w.u, w.rg);
            __syncthreads();
        }
        if (mode == INIT) {
            qp::for_rows(d, [&](int row, bool isU, int k, int r) {
                const double g = w.rg[row] - qp::row_h(d, c, q, isU, k, r);
                w.D[row] = d.ng ? 1.0 : 0.0; w.rho[row] = g; w.lam[row] = 0.0;
            });
        } else if (mode == PRED) {
            double musum = 0.0, rpm = 0.0;
            qp::for_rows(d, [&](int row, bool isU, int k, int r) {
                const double g = w.rg[row] - qp::row_h(d, c, q, isU, k, r);
                const double t = w.t[row], lam = w.lam[row];
                const double rg = g + t;
                w.rg[row] = rg;
                const double D = lam / (t + dreg * lam);
                w.D[row] = D;
                w.rho[row] = D * (rg + dreg * lam);
                musum += lam * t;
                rpm = fmax(rpm, fabs(rg));
            });
            mu = wg::reduce(musum, 0, L.red) / d.ng;
            rp = wg::reduce(rpm, 1, L.red);
        } else {
            qp::for_rows(d, [&](int row, bool, int, int) {
                const double t = w.t[row], lam = w.lam[row];
                const double rc = lam * t + w.dt[row] * w.dlam[row] - sig * mu;
                w.rc[row] = rc;
                w.rho[row] = lam + (lam * w.rg[row] - rc) / (t + dreg * lam);
            });
        }
        __syncthreads();
        QC_LAP(1);                                  // rows
        // ---------------- Newton system
        double rd = 0.0;
        bool ok = true;
        if (mode == PRED) {
            // reduced dual residual: gradient of the Lagrangian wrt u with the true multipliers
            gradients(d, c, q, w, qw, w.lam, L.ua, L.ya);
            gT_times(d, qw, L.ya, (clptr) nullptr, L.ub, (lptr) nullptr);
            for (int e = tid; e < nm; e += nt) rd = fmax(rd, fabs(L.ua[e] + L.ub[e]));
            rd = wg::reduce(rd, 1, L.red);
        }
        QC_LAP(2);                                  // dual residual
        if (mode != CORR) {
            ok = stage_factors(d, c, w, L);
            QC_LAP(3);
            if (ok) {
                gram<MSEL>(d, qw, L);
                QC_LAP(4);
                ok = tile_cholesky(d, L);
                QC_LAP(5);
            }
        }
        if (ok) {
            gradients(d, c, q, w, qw, w.rho, L.ua, L.ya);
            newton_solve(d, w, qw, L);
        }
        QC_LAP(6);                                  // gradients + Newton solve
        // ---------------- use the direction
        if (mode == INIT) {
            if (!ok) { status = 2; break; }
            for (int e = tid; e < nm; e += nt) w.u[e] += w.du[e];
            for (int e = tid; e < (N + 1) * po; e += nt) qw.y[e] += qw.dy[e];
            __syncthreads();
            if (d.ng == 0) { status = 0; break; }
            rows_apply(d, c, qw.y, w.u, w.rg);
            __syncthreads();
            double zmin = INFINITY, zmax = -INFINITY;
            qp::for_rows(d, [&](int row, bool isU, int k, int r) {
                const double g = w.rg[row] - qp::row_h(d, c, q, isU, k, r);
                w.rg[row] = g;
                zmin = fmin(zmin, g); zmax = fmax(zmax, g);
            });
            zmin = wg::reduce(zmin, 2, L.red);
            zmax = wg::reduce(zmax, 1, L.red);
            const double sh_t = zmax >= 0.0 ? 1.0 + zmax : 0.0, sh_l = zmin <= 0.0 ? 1.0 - zmin : 0.0;
            qp::for_rows(d, [&](int row, bool, int, int) {
                const double g = w.rg[row];
                w.t[row] = -g + sh_t; w.lam[row] = g + sh_l;
            });
            __syncthreads();
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
        if (mode == PRED) {
            if (!ok) { status = near_opt ? 0 : 2; break; }
            if (!(mu == mu)) { status = near_opt ? 0 : 5; break; }
            if (!(rd == rd)) { status = near_opt ? 0 : 6; break; }
            if (q.dbg && tid == 0) { gptr g = q.dbg + 8 * it; g[0] = mu; g[1] = rd; g[2] = rp; g[3] = sd; g[4] = sp; }
            const double ltol = fmax(d.tol, 1e-9);
            if (rd <= ltol * sd && rp <= ltol * sp && mu <= d.tol) { status = 0; break; }
            near_opt = (rd <= 1e-8 * sd && rp <= 1e-8 * sp && mu <= 1e-8);
            if (it >= d.max_iter) { status = 1; break; }
            rows_apply(d, c, qw.dy, w.du, w.dt);
            __syncthreads();
            qp::for_rows(d, [&](int row, bool, int, int) {
                const double t = w.t[row], lam = w.lam[row], rga = w.rg[row] + w.dt[row];
                const double dl = (-lam * t + lam * rga) / (t + dreg * lam);
                w.dlam[row] = dl;
                w.dt[row] = -rga + dreg * dl;
            });
            __syncthreads();
            const double a_aff = fmin(1.0, qp::max_step(d, w, Lq));
            double ma = 0.0;
            qp::for_rows(d, [&](int row, bool, int, int) {
                ma += (w.lam[row] + a_aff * w.dlam[row]) * (w.t[row] + a_aff * w.dt[row]);
            });
            const double mu_aff = wg::reduce(ma, 0, L.red) / d.ng;
            sig = mu > 0.0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
            if (q.dbg && tid == 0) { gptr g = q.dbg + 8 * it; g[5] = a_aff; g[6] = sig; }
            mode = CORR;
            continue;
        }
        // mode == CORR: step
        rows_apply(d, c, qw.dy, w.du, w.dt);
        __syncthreads();
        qp::for_rows(d, [&](int row, bool, int, int) {
            const double t = w.t[row], lam = w.lam[row], rga = w.rg[row] + w.dt[row];
            const double dl = (-w.rc[row] + lam * rga) / (t + dreg * lam);
            w.dlam[row] = dl;
            w.dt[row] = -rga + dreg * dl;
        });
        __syncthreads();
        const double a = fmin(1.0, 0.99 * qp::max_step(d, w, Lq));
        if (q.dbg && tid == 0) { gptr g = q.dbg + 8 * it; g[7] = a; }
        for (int e = tid; e < nm; e += nt) w.u[e] += a * w.du[e];
        for (int e = tid; e < (N + 1) * po; e += nt) qw.y[e] += a * qw.dy[e];
        qp::for_rows(d, [&](int row, bool, int, int) {
            w.t[row] += a * w.dt[row];
            w.lam[row] += a * w.dlam[row];
        });
        __syncthreads();
        ++it;
        mode = PRED;
    }
    QC_LAP(7);                                      // steps (row directions, step lengths, updates)
#ifdef SRH_PROFILE
    if (q.dbg && tid == 0) for (int i = 0; i < 8; ++i) q.dbg[8 * 60 + i] = (double)tq[i];
#endif
    if (iters_out) *iters_out = it;
    return status;
}

}  // namespace qpc
