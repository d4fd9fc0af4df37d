// Host side shared by the LOCP and GuSTO translation units (scp.hip, gusto.hip): problem constants, the LDS layout
// search, the list of kernel variants.  Everything has internal linkage: each unit carries its own copy.
#pragma once
#include "tpwl_host.h"
#include "locp_dev.h"
#include <cstdlib>

namespace {

constexpr int NTHREADS = 512;

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

}  // namespace
