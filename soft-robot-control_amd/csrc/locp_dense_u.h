// The LOCP QP without its trust-region rows for SHORT horizons, on ONE wave, in the space of the inputs (round 6).
//
// ql::ipm_wave (locp_lean.h) is the one-wave interior point of the reference drivers' closed-loop horizons, but it lives in the output space
// of a cost with p_o = 2 outputs (both robots' TPWL drivers weigh two tip coordinates).  The reference's SSM hardware driver weighs THREE
// (examples/hardware/diamond_SSM.py:322-326: x, y, z of the end effector) and then falls back to the eight-wave condensed interior point:
// 160 k clocks per iteration for a problem with 12 unknowns.  For N n_u <= 16 the whole QP fits one 16 x 16 tile in the space of u whatever
// the cost looks like:
//     minimise  (u - ud)^T Rb (u - ud) + sum_k (H x_k - z_k)^T Qz (H x_k - z_k)     x = x_free + S u            (locp.py:218-263, no 1/2)
//     s.t.      C u <= h          rows: U.A u_k <= U.b (locp.py:300-303),  X.A x_k <= X.b, k = 1..N (locp.py:330-333) through S
// with the dense Hessian Hq = 2 (Rb + G^T Qb G), G = H S (N n_z x N n_u), and the interior point of ql::ipm_box / ipm_wave -- same starting
// point (unit-weight Newton step, shifts), weights D = lambda / (t + dreg lambda), Mehrotra predictor / corrector with step 0.99, stopping
// rule and warm start -- on the Newton systems (Hq + C^T D C) du = -(grad + C^T rho): one MFMA Gram product per factorisation (operands: the
// rows' coefficients in the MFMA layout, constant per QP, times sqrt(D) from LDS), Jacobi scaling, qpc::chol16, two 16 x 16 products per
// solve.  One row per lane (<= 64 rows), everything else in a few KB of LDS; the other waves of the workgroup wait at the barrier behind it.
// The per-stage matrices come from global memory (gusto_ssm.hip writes them per SCP iteration; per-stage or region-indexed: QPDyn).
#pragma once
#include "locp_lean.h"

namespace qdu {

constexpr int NU = 16;                 // inputs of the QP (N n_u <= 16): one tile
constexpr int NYM = 48;                // N n_z <= 48
constexpr int TS = qpc::TS;
constexpr int CS = 17;                 // row stride of C in LDS (odd: lane = row reads without bank conflicts)

// LDS copies of the per-stage [A_k | B_k | d_k], of H and of X.A: the set-up's dot products would otherwise wait for L2 once per term
__host__ __device__ inline size_t stage_doubles(const QPDims &d) {
    return (size_t)d.N * d.n * (d.n + d.m + 1) + (size_t)d.nz * d.n + (size_t)d.nX * d.n;
}
__host__ __device__ inline bool applies(const QPDims &d) {
    return d.N * d.m <= NU && d.N * d.nz <= NYM && d.N * (d.nU + d.nX) <= 64 && d.nXf == 0 && d.n <= 64 && d.N <= 8 && d.nz <= 16 && d.m <= 16 &&
           stage_doubles(d) <= 6144;            // the stage matrices, H and X.A are staged in LDS (48 KB at most): small models
}
// LDS doubles: S (2 x n x 16) | G (NYM x 16) | C (64 x 17) | W = Qb G (NYM x 16) | Hq, M, Rinv tiles (3 x 16 x 17) | xf ((N + 1) n) | vectors
__host__ __device__ inline size_t lds_doubles(const QPDims &d) {
    return 2 * (size_t)d.n * 16 + 2 * (size_t)NYM * 16 + 64 * CS + 3 * 16 * TS + (size_t)(d.N + 1) * d.n + NYM + 8 * 64 + 16 + stage_doubles(d);
}

// returns 0: w.x / w.u hold the minimiser of the FULL QP (converged, inside the trust region), J_out its objective; 100: the minimiser of
// the relaxed QP leaves the trust region; anything else: interior point not converged.  `lam` (64 doubles of the rollout's work block): the
// multipliers of the last converged solve (warm != 0 starts from w.u and them).
__device__ __forceinline__ int solve(const QPDims &d, const QPConst &c, const QPDyn &dyn, const QPData &q, QPWork &w, lptr lds, gptr lam,
                                     double *J_out, int *it_out, int warm, double *dbg = nullptr) {
    const int tid = SRH_TID;
    const int N = d.N, n = d.n, m = d.m, nz = d.nz, nu = N * m, ny = N * nz, nrU = N * d.nU, nr = nrU + N * d.nX;
    // the arrays of the interior point first, at compile-time offsets from one base (one address register, the rest in the instructions'
    // offset fields); the arrays whose size follows the model (set-up and tail only) behind them
    lptr Gl = lds, Cl = Gl + NYM * 16, Hl = Cl + 64 * CS, Tl = Hl + 16 * TS, Rl = Tl + 16 * TS, ey = Rl + 16 * TS, ul = ey + NYM, dul = ul + 64,
         rhs = dul + 64, tv = rhs + 64, rhol = tv + 64, laml = rhol + 64, sdl = laml + 64, scl = sdl + 64, res = scl + 64, Wl = res + 16;
    lptr Sa = Wl + NYM * 16, Sb = Sa + (size_t)n * 16, xf = Sb + (size_t)n * 16;
    lptr Asl = xf + (size_t)(N + 1) * n, Bsl = Asl + (size_t)N * n * n, dsl = Bsl + (size_t)N * n * m, Hsl = dsl + (size_t)N * n, Xsl = Hsl + (size_t)nz * n;
    int status = 1, it = 0;
    double Jv = 0.0;
#ifndef QDU_CLOCKS
#define QDU_CLOCKS 0                   // 1: phase clocks of the solve through res[4..13] / dbg (22 more scalar registers: a measuring build only)
#endif
#if QDU_CLOCKS
    long long lp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, lc = clock64();
#define QDU_LAP(i) do { const long long now_ = clock64(); lp[i] += now_ - lc; lc = now_; } while (0)
#else
#define QDU_LAP(i) do { } while (0)
#endif
    if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
        const int lane = tid, l16 = lane & 15, kk = lane >> 4;
        auto fence = [&]() { ql::wave_fence(); };
        // the stage matrices, H and X.A into LDS (coalesced, all loads in flight at once)
        for (int k = 0; k < N; ++k) {
            cgptr Ag = dyn.A + dyn.sel(k) * (size_t)n * n, Bg = dyn.B + dyn.sel(k) * (size_t)n * m, dg = dyn.d + dyn.sel(k) * (size_t)n;
            for (int e = lane; e < n * n; e += 64) Asl[(size_t)k * n * n + e] = Ag[e];
            for (int e = lane; e < n * m; e += 64) Bsl[(size_t)k * n * m + e] = Bg[e];
            for (int e = lane; e < n; e += 64) dsl[(size_t)k * n + e] = dg[e];
        }
        for (int e = lane; e < nz * n; e += 64) Hsl[e] = c.H[e];
        for (int e = lane; e < d.nX * n; e += 64) Xsl[e] = c.XA[e];
        auto Ak = [&](int k) { return (clptr)(Asl + (size_t)k * n * n); };
        auto Bk = [&](int k) { return (clptr)(Bsl + (size_t)k * n * m); };
        auto dk = [&](int k) { return (clptr)(dsl + (size_t)k * n); };
        // ---- free response xf_k (u = 0) and the sensitivities S_k = d x_k / d u (n x 16, column e = (stage j, input b)), stage by stage;
        //      G rows (k, a) = H S_k, state rows of stage k = X.A S_k
        for (int e = lane; e < n; e += 64) xf[e] = q.x0[e];
        for (int e = lane; e < n * 16; e += 64) Sa[e] = 0.0;
        for (int e = lane; e < 64 * CS; e += 64) Cl[e] = 0.0;
        for (int e = lane; e < NYM * 16; e += 64) Gl[e] = 0.0;
        fence();
        lptr Sc = Sa, Sn = Sb;
        for (int k = 0; k < N; ++k) {
            clptr A = Ak(k), B = Bk(k), dd = dk(k);
            for (int i = lane; i < n; i += 64) {
                double v = dd[i];
#pragma unroll 4
                for (int j = 0; j < n; ++j) v = fma(A[(size_t)i * n + j], xf[(size_t)k * n + j], v);
                xf[(size_t)(k + 1) * n + i] = v;
            }
            // S_{k+1} = A_k S_k + B_k at the columns of stage k: lane (column e = l16, rows i = kk, kk + 4, ...)
            for (int i = kk; i < n; i += 4) {
                double v = 0.0;
                if (l16 < nu) {
#pragma unroll 4
                    for (int j = 0; j < n; ++j) v = fma(A[(size_t)i * n + j], Sc[j * 16 + l16], v);
                    if (l16 / m == k) v += B[(size_t)i * m + (l16 - k * m)];
                }
                Sn[i * 16 + l16] = v;
            }
            fence();
            for (int a = kk; a < nz; a += 4) {                             // G rows of stage k + 1
                double v = 0.0;
#pragma unroll 4
                for (int j = 0; j < n; ++j) v = fma(Hsl[(size_t)a * n + j], Sn[j * 16 + l16], v);
                Gl[(k * nz + a) * 16 + l16] = v;
            }
            for (int r = kk; r < d.nX; r += 4) {                           // state rows of stage k + 1
                double v = 0.0;
#pragma unroll 4
                for (int j = 0; j < n; ++j) v = fma(Xsl[(size_t)r * n + j], Sn[j * 16 + l16], v);
                Cl[(nrU + k * d.nX + r) * CS + l16] = v;
            }
            lptr t_ = Sc; Sc = Sn; Sn = t_;
            fence();
        }
        // input rows: U.A on the inputs of their stage
        for (int e = lane; e < nrU * 16; e += 64) {
            const int r = e >> 4, col = e & 15, k = r / d.nU, rr = r - k * d.nU;
            Cl[r * CS + col] = (col < nu && col / m == k) ? c.UA[(size_t)rr * m + (col - k * m)] : 0.0;
        }
        // output errors of the free response, e_y = H xf_k - z_k (k = 1..N), and the constant of the objective (k = 0)
        for (int e = lane; e < ny; e += 64) {
            const int k = e / nz + 1, a = e - (k - 1) * nz;
            double v = q.z ? -q.z[(size_t)k * nz + a] : 0.0;
#pragma unroll 4
            for (int j = 0; j < n; ++j) v = fma(Hsl[(size_t)a * n + j], xf[(size_t)k * n + j], v);
            ey[e] = v;
        }
        fence();
        QDU_LAP(0);
        // ---- this lane's row (right-hand side; its coefficients and the MFMA operand slice of C stay in LDS: 32 registers per lane that the
        //      kernel around this function does not have), Hq in the accumulator layout
        const bool isrow = lane < nr;
        double hr = 0.0;
        if (lane < nrU) hr = c.Ub[lane % d.nU];
        else if (isrow) {
            const int rl = lane - nrU, k = rl / d.nX + 1, rr = rl - (k - 1) * d.nX;
            double v = c.Xb[rr];
#pragma unroll 4
            for (int j = 0; j < n; ++j) v = fma(-Xsl[(size_t)rr * n + j], xf[(size_t)k * n + j], v);
            hr = v;
        }
        // W = Qb G (column l16, rows of this lane's k-group), Hq[i][l16] = 2 (sum_ya G[ya][i] W[ya] + R)
        for (int ya = kk; ya < ny; ya += 4) {
            const int k = ya / nz, a = ya - k * nz;
            double v = 0.0;
            for (int b = 0; b < nz; ++b) v = fma(c.Qz[a * nz + b], Gl[(k * nz + b) * 16 + l16], v);
            Wl[ya * 16 + l16] = v;
        }
        fence();
        wg::qp_d4 hq;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int i = kk + 4 * qd;
            double v = 0.0;
            for (int ya = 0; ya < ny; ++ya) v = fma(Gl[ya * 16 + i], Wl[ya * 16 + l16], v);
            if (i < nu && l16 < nu && i / m == l16 / m) v += c.R[(i % m) * m + (l16 % m)];
            hq[qd] = (i < nu && l16 < nu) ? 2.0 * v : 0.0;
            Hl[i * TS + l16] = hq[qd];
        }
        // g0[e] = 2 (G^T Qb e_y - Rb ud)
        double g0 = 0.0;
        if (lane < nu) {
            double v = 0.0;
            for (int ya = 0; ya < ny; ++ya) v = fma(Wl[ya * 16 + lane], ey[ya], v);        // (Qb G)^T e_y = G^T Qb e_y  (Qz symmetric)
            if (q.ud) { const int k = lane / m, b = lane - k * m; for (int b2 = 0; b2 < m; ++b2) v = fma(-c.R[b * m + b2], q.ud[(size_t)k * m + b2], v); }
            g0 = 2.0 * v;
        }
        fence();
        // ---- helpers on the 16-vectors in LDS
        auto hq_times = [&](clptr v) -> double {                           // (Hq v)[lane], lane < 16
            double s0 = 0.0, s1 = 0.0;
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) { s0 = fma(Hl[lane * TS + e], v[e], s0); s1 = fma(Hl[lane * TS + e + 1], v[e + 1], s1); }
            }
            return s0 + s1;
        };
        auto row_dot = [&](clptr v) -> double {                            // c_row . v
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int e = 0; e < 16; e += 2) { s0 = fma(Cl[lane * CS + e], v[e], s0); s1 = fma(Cl[lane * CS + e + 1], v[e + 1], s1); }
            return s0 + s1;
        };
        auto ct_times = [&](clptr rv) -> double {                          // (C^T rv)[l16] on every lane; rv: 64 row values in LDS
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int s = 0; s < 16; s += 2) { a0 = fma(Cl[(4 * s + kk) * CS + l16], rv[4 * s + kk], a0); a1 = fma(Cl[(4 * s + 4 + kk) * CS + l16], rv[4 * (s + 1) + kk], a1); }
            double a = a0 + a1;
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            return a;
        };
        // M = Hq + C^T diag(sdl^2) C, Jacobi-scaled, factored: Rl = inverse of the factor, scl = the scaling; returns the pivots' verdict
        auto factor = [&]() -> bool {
            // four independent accumulation chains (a dependent MFMA waits out the 64-cycle pipeline of the one before it)
            wg::qp_d4 acc = hq, acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0}, acc3 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 16; s += 4) {
                const double a0 = Cl[(4 * s + kk) * CS + l16] * sdl[4 * s + kk], a1 = Cl[(4 * s + 4 + kk) * CS + l16] * sdl[4 * s + 4 + kk];
                const double a2 = Cl[(4 * s + 8 + kk) * CS + l16] * sdl[4 * s + 8 + kk], a3 = Cl[(4 * s + 12 + kk) * CS + l16] * sdl[4 * s + 12 + kk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a2, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, a3, acc3, 0, 0, 0);
            }
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) acc[qd] = (acc[qd] + acc1[qd]) + (acc2[qd] + acc3[qd]);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int i = kk + 4 * qd;
                double v = acc[qd];
                if (i >= nu || l16 >= nu) v = (i == l16) ? 1.0 : 0.0;      // padding: the identity
                Tl[i * TS + l16] = v;
            }
            fence();
            if (lane < 16) scl[lane] = qpc::rsq3(Tl[lane * TS + lane]);
            fence();
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) { const int i = kk + 4 * qd; Tl[i * TS + l16] *= scl[i] * scl[l16]; }
            fence();
            const bool ok = qpc::chol16<false>(Tl, Rl);
            fence();
            return ok;
        };
        // dul = M^-1 rhs  (M^-1 = S Rinv Rinv^T S with Rl[c][r] = Rinv[c][r]: qpc::chol16)
        auto msolve = [&]() {
            const int cI = lane >> 2, part = lane & 3;
            double t = 0.0;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) t = fma(Rl[(4 * part + kq) * TS + cI], rhs[4 * part + kq] * scl[4 * part + kq], t);
            t = wg::group_sum<4>(t);
            fence();
            if (part == 0) tv[cI] = t;
            fence();
            double x = 0.0;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) x = fma(Rl[cI * TS + 4 * part + kq], tv[4 * part + kq], x);
            x = wg::group_sum<4>(x);
            fence();
            if (part == 0) dul[cI] = x * scl[cI];
            fence();
        };
        QDU_LAP(1);
        // ---- the interior point (ql::ipm_box's iteration)
        constexpr double WARM_FLOOR = 1e-2;
        const double ng = (double)nr;
        if (lane < 16) ul[lane] = (warm && lane < nu) ? w.u[lane] : 0.0;
        fence();
        double tr_ = 1.0, lr = 0.0, rg = 0.0, rc = 0.0, dtr = 0.0, dlr = 0.0;
        enum { INIT = 0, PRED = 1, CORR = 2 };
        int mode = INIT;
        double mu = 0.0, rp = 0.0, sig = 0.0, sd = 1.0, sp = 1.0, dreg = 0.0;
        bool near_opt = false;
        auto scales = [&]() {
            for (int e = lane; e < n; e += 64) {
                double gq = 0.0;
                if (q.z) for (int a = 0; a < nz; ++a) gq = fma(c.HtQz2[e * nz + a], -q.z[nz + a], gq);
                sd = fmax(sd, fabs(gq));
            }
            for (int e = lane; e < d.nU; e += 64) sp = fmax(sp, fabs(c.Ub[e]));
            sd = fmax(wg::wave_max(sd), q.omega);
            sp = fmax(wg::wave_max(sp), fabs(q.delta));
            dreg = d.reg / sd;
        };
        if (warm && nr > 0) {
            if (isrow) { tr_ = fmax(-(row_dot(ul) - hr), WARM_FLOOR); lr = fmax(lam[lane], WARM_FLOOR); }
            scales();
            mode = PRED;
        }
        bool ok = true;
        while (true) {
            double Dw = 0.0, rho = 0.0, musum = 0.0, rpm = 0.0;
            if (isrow) {
                const double gq = mode != CORR ? row_dot(ul) - hr : 0.0;          // (the corrector's rows use the predictor's residuals)
                if (mode == INIT) { Dw = 1.0; rho = gq; lr = 0.0; }
                else if (mode == PRED) {
                    rg = gq + tr_;
                    Dw = lr / (tr_ + dreg * lr);
                    rho = Dw * (rg + dreg * lr);
                    musum = lr * tr_;
                    rpm = fabs(rg);
                } else {
                    rc = lr * tr_ + dtr * dlr - sig * mu;
                    rho = lr + (lr * rg - rc) / (tr_ + dreg * lr);
                }
            }
            if (mode != CORR) sdl[lane] = sqrt(Dw);
            rhol[lane] = rho;
            if (mode == PRED) laml[lane] = isrow ? lr : 0.0;
            fence();
            QDU_LAP(6);
            if (mode == PRED) { mu = wg::wave_sum(musum) / ng; rp = wg::wave_max(rpm); }
            QDU_LAP(7);
            // gradient of the cost at u, right-hand side, dual residual
            const double gcost = hq_times(ul) + g0;
            const double ctr = ct_times(rhol);
            double rd = 0.0;
            if (mode == PRED) {
                const double ctl = ct_times(laml);
                rd = wg::wave_max((lane < nu) ? fabs(gcost + ctl) : 0.0);
            }
            if (lane < 16) rhs[lane] = lane < nu ? -(gcost + ctr) : 0.0;
            fence();
            QDU_LAP(8);
            if (mode != CORR) ok = factor();
            QDU_LAP(3);
            if (ok) msolve();
            QDU_LAP(4);
            if (mode == INIT) {
                if (!ok) { status = 2; break; }
                if (lane < 16) ul[lane] += dul[lane];
                fence();
                if (nr == 0) { status = 0; break; }
                double zmin = INFINITY, zmax = -INFINITY, gq = 0.0;
                if (isrow) { gq = row_dot(ul) - hr; zmin = gq; zmax = gq; }
                zmin = wg::wave_min(zmin); zmax = wg::wave_max(zmax);
                const double sh_t = zmax >= 0.0 ? 1.0 + zmax : 0.0, sh_l = zmin <= 0.0 ? 1.0 - zmin : 0.0;
                tr_ = -gq + sh_t; lr = gq + sh_l;
                scales();
                mode = PRED;
                continue;
            }
            double amax = 1e300;
            if (ok && isrow) {
                const double rga = rg + row_dot(dul);
                const double dl = ((mode == PRED ? -lr * tr_ : -rc) + lr * rga) / (tr_ + dreg * lr);
                const double dtv = -rga + dreg * dl;
                dlr = dl; dtr = dtv;
                if (dtv < 0.0) amax = fmin(amax, -tr_ / dtv);
                if (dl < 0.0) amax = fmin(amax, -lr / dl);
            }
            amax = wg::wave_min(amax);
            QDU_LAP(9);
            if (mode == PRED) {
                if (!ok) { status = near_opt ? 0 : 2; break; }
                if (!(mu == mu)) { status = near_opt ? 0 : 5; break; }
                if (!(rd == rd)) { status = near_opt ? 0 : 6; break; }
                const double ltol = fmax(d.tol, 1e-9);
                if (rd <= ltol * sd && rp <= ltol * sp && mu <= d.tol) { status = 0; break; }
                near_opt = (rd <= 1e-8 * sd && rp <= 1e-8 * sp && mu <= 1e-8);
                if (it >= d.max_iter) { status = 1; break; }
                const double a_aff = fmin(1.0, amax);
                const double ma = isrow ? (lr + a_aff * dlr) * (tr_ + a_aff * dtr) : 0.0;
                const double mu_aff = wg::wave_sum(ma) / ng;
                sig = mu > 0.0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
                mode = CORR;
                continue;
            }
            if (!ok) { status = 2; break; }
            const double a = fmin(1.0, 0.99 * amax);
            if (lane < 16) ul[lane] += a * dul[lane];
            if (isrow) { tr_ += a * dtr; lr += a * dlr; }
            fence();
            ++it;
            mode = PRED;
        }
        QDU_LAP(2);
        // ---- results: u, the multipliers for the next warm start, the trajectory by a rollout of the minimiser, the objective
        if (lane < nu) w.u[lane] = ul[lane];
        if (status == 0 && isrow) lam[lane] = lr;
        for (int e = lane; e < n; e += 64) { xf[e] = q.x0[e]; w.x[e] = q.x0[e]; }
        fence();
        for (int k = 0; k < N; ++k) {
            clptr A = Ak(k), B = Bk(k), dd = dk(k);
            for (int i = lane; i < n; i += 64) {
                double v = dd[i];
#pragma unroll 4
                for (int j = 0; j < n; ++j) v = fma(A[(size_t)i * n + j], xf[(size_t)k * n + j], v);
                for (int b = 0; b < m; ++b) v = fma(B[(size_t)i * m + b], ul[k * m + b], v);
                xf[(size_t)(k + 1) * n + i] = v;
                w.x[(size_t)(k + 1) * n + i] = v;
            }
            fence();
        }
        // J = sum_k (H x_k - z_k)^T Qz (.) + sum_k (u_k - ud_k)^T R (.): lane = stage
        double acc = 0.0;
        if (lane <= N) {
            const int k = lane;
            double e[16];
            for (int a = 0; a < nz; ++a) {
                double v = q.z ? -q.z[(size_t)k * nz + a] : 0.0;
#pragma unroll 4
                for (int j = 0; j < n; ++j) v = fma(Hsl[(size_t)a * n + j], xf[(size_t)k * n + j], v);
                e[a] = v;
            }
            for (int a = 0; a < nz; ++a) for (int b = 0; b < nz; ++b) acc = fma(e[a] * c.Qz[a * nz + b], e[b], acc);
            if (k < N) {
                double ue[16];
                for (int a = 0; a < m; ++a) ue[a] = ul[k * m + a] - (q.ud ? q.ud[(size_t)k * m + a] : 0.0);
                for (int a = 0; a < m; ++a) for (int b = 0; b < m; ++b) acc = fma(ue[a] * c.R[a * m + b], ue[b], acc);
            }
        }
        Jv = wg::wave_sum(acc);
        // trust region of the full QP: slack of stage 0 in closed form, the other stages must lie inside (qp::solve's prescreen)
        double md = 0.0, m0 = 0.0;
        for (int e = lane; e < (N + 1) * n; e += 64) {
            const double v = fabs(c.xs[e % n] * (xf[e] - q.xk[e]));
            if (e < n) m0 = fmax(m0, v); else md = fmax(md, v);
        }
        md = wg::wave_max(md); m0 = wg::wave_max(m0);
        if (d.tr) {
            const double s0 = fmax(0.0, m0 - q.delta);
            Jv += q.omega * s0;
            if (status == 0 && !(md <= q.delta)) status = 100;
        }
        QDU_LAP(5);
        if (lane == 0) { res[0] = (double)status; res[1] = (double)it; res[2] = Jv;
#if QDU_CLOCKS
            for (int i = 0; i < 10; ++i) res[4 + i] = (double)lp[i];
#endif
        }
    }
    __syncthreads();
    status = (int)res[0];
    it = (int)res[1];
    Jv = res[2];
    __syncthreads();
    if (J_out) *J_out = Jv;
    if (it_out) *it_out = it;
#if QDU_CLOCKS
    if (dbg) for (int i = 0; i < 10; ++i) dbg[i] = res[4 + i];
#endif
    return status;
}

#undef QDU_LAP
}  // namespace qdu
