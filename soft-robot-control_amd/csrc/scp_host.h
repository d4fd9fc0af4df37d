// Host side shared by the LOCP and GuSTO translation units (scp.hip, gusto.hip): problem constants, the LDS layout
// search, the list of kernel variants.  Everything has internal linkage: each unit carries its own copy.
#pragma once
#include "tpwl_host.h"
#include "locp_lean.h"
#include <algorithm>
#include <cstdlib>

#ifndef SRH_QC_MAX_M
#define SRH_QC_MAX_M 16
#endif

namespace {

constexpr int NTHREADS = 512;

struct QPConstHost {
    srh::DevBuf H, Qz, Qzf, R, xs, UA, Ub, XA, Xb, XfA, Xfb, Qx, QxN, HtQz2, HtQzf2, R2, Cq, Co, Sc, ScN, Tx, Txf, Cz2, Czf2, gram_sched;
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
        c.Co = g(Co); c.Sc = g(Sc); c.ScN = g(ScN); c.Tx = g(Tx); c.Txf = g(Txf); c.Cz2 = g(Cz2); c.Czf2 = g(Czf2);
        c.gram_sched = (cgiptr)gram_sched.as<int>();
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

// Gram tile tasks of the lean kernels: tile row I (KT - I upper tiles, k-steps = min(N, 8 (I + 1)) m / 4 each) is cut into chunks
// of at most 4 tiles that share the A operand; the chunks go to the waves longest first onto the least loaded wave (waves w and
// w + 4 share a SIMD and its MFMA pipe: a quarter of the partner's load counts).  The cost of a chunk of nJ tiles over ks k-steps
// is what the per-wave clocks of the profile build say (round 5, C2, on the kernel without scratch: t ~ 0.034 nJ ks + 0.07 ks +
// 0.75 nJ k clocks -- the MFMAs, the operand loads + 1 / D product of a k-step, the epilogue of a tile; the MFMA pipe of a SIMD is shared
// by its two waves and is not what a wave waits for): in MFMA units nJ ks + 2.1 ks + 22 nJ.  The chunk size
// of every tile row is searched exhaustively (4^KT greedy assignments, KT <= 8: milliseconds at plan creation) for the smallest
// maximum load -- the one-size rule of round 4 left the busiest wave 24 % above the mean.  Output: GRAM_TASKS x {I, J0, nJ, 0}, longest first, zeros behind the last task.
static bool lean_gram_schedule(int N, int m, int KT, int nw, std::vector<int> &out) {
    struct Task { int I, J0, nJ; double cost; };
    if (KT < 1 || KT > 8 || nw < 1 || nw > 16) return false;
    auto ksteps = [&](int I) { return std::min(N, 8 * (I + 1)) * (m / 4); };
    auto cost = [&](int I, int nJ) { const double ks = ksteps(I); return nJ * ks + 2.1 * ks + 22.0 * nJ; };
    double best_max = 0.0, best_sum = 0.0;
    bool have = false;
    std::vector<Task> tasks;
    std::vector<int> cur;
    int combos = 1;
    for (int I = 0; I < KT; ++I) combos *= 4;
    for (int code = 0; code < combos; ++code) {
        tasks.clear();
        int cd = code;
        for (int I = 0; I < KT; ++I) {
            const int per = 1 + (cd & 3);
            cd >>= 2;
            for (int J = I; J < KT; J += per) { const int nJ = std::min(per, KT - J); tasks.push_back({I, J, nJ, cost(I, nJ)}); }
        }
        if ((int)tasks.size() > 4 * nw || (int)tasks.size() >= ql::GRAM_TASKS) continue;
        std::stable_sort(tasks.begin(), tasks.end(), [](const Task &a, const Task &b) { return a.cost > b.cost; });
        double load[16] = {0.0};
        int cnt[16] = {0};
        cur.assign((size_t)nw * 16, 0);
        bool ok = true;
        for (const Task &t : tasks) {
            int best = -1;
            double bl = 0.0;
            for (int w = 0; w < nw; ++w) {
                if (cnt[w] >= 4) continue;
                const double key = load[w] + (nw > 4 ? 0.25 * load[(w + nw / 2) % nw] : 0.0);
                if (best < 0 || key < bl) { best = w; bl = key; }
            }
            if (best < 0) { ok = false; break; }
            int *o = &cur[((size_t)best * 4 + cnt[best]) * 4];
            o[0] = t.I; o[1] = t.J0; o[2] = t.nJ; o[3] = 0;
            load[best] += t.cost;
            ++cnt[best];
        }
        if (!ok) continue;
        double mx = 0.0, sum = 0.0;
        for (int w = 0; w < nw; ++w) { mx = std::max(mx, load[w]); sum += load[w]; }
        if (!have || mx < best_max - 1e-9 || (mx < best_max + 1e-9 && sum < best_sum - 1e-9)) {
            have = true; best_max = mx; best_sum = sum;
            // what the kernel gets: the chunks of this cut, longest first -- its waves PULL them (ql::gram); the greedy assignment above
            // only ranks the cuts
            out.assign((size_t)ql::GRAM_TASKS * 4, 0);
            for (size_t t = 0; t < tasks.size(); ++t) { out[4 * t] = tasks[t].I; out[4 * t + 1] = tasks[t].J0; out[4 * t + 2] = tasks[t].nJ; }
        }
    }
    return have;
}

// want_half: the caller's batch is larger than the chip has CUs -- lay the problem out for the half-size lean workgroup where it fits (two
// rollouts per CU: throughput; a single rollout is faster on the full-size workgroup).  SRH_LEAN_HALF=1 / 0 in the environment overrides.
int build_consts(const slocp_problem *pr, QPConstHost &C, bool want_half = false) {
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
    d.cond = 0; d.po = 0; d.KT = 0; d.qc_off = 0; d.diagD = 0; d.lean = 0; d.lean_j0 = 0; d.ls_pd = 0; d.lean_half = 0;
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
    // ---- condensed path (locp_cond.h): an orthonormal basis C_o of the row space of [Cq; Cqf; X.A; Xf.A] and the
    // cost / constraint rows expressed in it.  Enabled when the space is small (po <= 4, N po <= 128) and the LDS
    // carve fits; SRH_QP_NO_COND=1 switches it off (A/B runs, tests of the Riccati path).
    {
        std::vector<double> Cqf;
        if (pr->Qzf) {
            std::vector<double> Qs((size_t)nz * nz), wv, V;
            for (int a = 0; a < nz; ++a) for (int b = 0; b < nz; ++b) Qs[a * nz + b] = 0.5 * (pr->Qzf[a * nz + b] + pr->Qzf[b * nz + a]);
            jacobi_eig(Qs, nz, wv, V);
            double wmax = 0.0;
            for (double x : wv) wmax = std::max(wmax, fabs(x));
            for (int e = 0; e < nz; ++e) {
                if (wv[e] <= 1e-13 * wmax) continue;
                const double sc = sqrt(2.0 * wv[e]);
                for (int j = 0; j < n; ++j) {
                    double v = 0.0;
                    for (int a = 0; a < nz; ++a) v += V[a * nz + e] * pr->H[a * n + j];
                    Cqf.push_back(sc * v);
                }
            }
        }
        const int ncq = (int)(Cq.size() / n), ncqf = (int)(Cqf.size() / n);
        const int rows = ncq + ncqf + pr->nX + pr->nXf;
        std::vector<double> M((size_t)rows * n);
        auto rowp = [&](int r) -> const double * {
            if (r < ncq) return Cq.data() + (size_t)r * n;
            r -= ncq;
            if (r < ncqf) return Cqf.data() + (size_t)r * n;
            r -= ncqf;
            if (r < pr->nX) return pr->XA + (size_t)r * n;
            return pr->XfA + (size_t)(r - pr->nX) * n;
        };
        for (int r = 0; r < rows; ++r) std::copy(rowp(r), rowp(r) + n, M.begin() + (size_t)r * n);
        // Gram matrix of the normalised rows; its eigenvectors with non-negligible eigenvalue span the row space
        std::vector<double> Gm((size_t)n * n, 0.0), wv, V;
        for (int r = 0; r < rows; ++r) {
            double nr2 = 0.0;
            for (int j = 0; j < n; ++j) nr2 += M[(size_t)r * n + j] * M[(size_t)r * n + j];
            if (nr2 <= 0.0) continue;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) Gm[(size_t)i * n + j] += M[(size_t)r * n + i] * M[(size_t)r * n + j] / nr2;
        }
        int po = 0;
        std::vector<double> Co;
        if (rows > 0) {
            jacobi_eig(Gm, n, wv, V);
            std::vector<int> ord(n);
            for (int i = 0; i < n; ++i) ord[i] = i;
            std::sort(ord.begin(), ord.end(), [&](int a, int b) { return wv[a] > wv[b]; });
            const double wmax = wv[ord[0]];
            // (eigenvalues of a Gram matrix carry an absolute error ~1e-16 wmax: a direction whose relative singular value is
            // below 1e-6 counts as dependent -- the reconstruction check below disables the path if that loses a row)
            while (po < n && wv[ord[po]] > 1e-12 * wmax && wv[ord[po]] > 0.0) ++po;
            Co.resize((size_t)po * n);
            for (int a = 0; a < po; ++a)
                for (int j = 0; j < n; ++j) Co[(size_t)a * n + j] = V[(size_t)j * n + ord[a]];
        }
        auto project = [&](const double *rowsrc, int nr, std::vector<double> &T) {      // T = rows Co^T  (nr x po)
            T.assign((size_t)std::max(1, nr * po), 0.0);
            for (int r = 0; r < nr; ++r)
                for (int a = 0; a < po; ++a) {
                    double v = 0.0;
                    for (int j = 0; j < n; ++j) v += rowsrc[(size_t)r * n + j] * Co[(size_t)a * n + j];
                    T[(size_t)r * po + a] = v;
                }
        };
        // the inequality rows live in registers, qpc::QR per thread
        bool ok = po >= 1 && po <= 4 && N * po <= 128 && m <= SRH_QC_MAX_M && !getenv("SRH_QP_NO_COND") &&
                  N * (pr->nX + pr->nXf) + N * pr->nU <= qpc::QR * NTHREADS;
        {   // box-type input rows and a diagonal R make every D_j diagonal: scalar scalings instead of m x m factors
            bool diag = true;
            for (int a = 0; a < m && diag; ++a) for (int b = 0; b < m; ++b) if (a != b && pr->R[a * m + b] != 0.0) { diag = false; break; }
            for (int r = 0; r < pr->nU && diag; ++r) {
                int nnz = 0;
                for (int b = 0; b < m; ++b) nnz += pr->UA[r * m + b] != 0.0;
                if (nnz > 1) diag = false;
            }
            d.diagD = diag && !getenv("SRH_QP_NO_DIAGD") ? 1 : 0;
        }
        std::vector<double> Tc, Tcf, Tx, Txf, Sc, ScN, Cz2, Czf2;
        if (ok) {
            project(Cq.data(), ncq, Tc); project(Cqf.data(), ncqf, Tcf);
            project(pr->XA, pr->nX, Tx); project(pr->XfA, pr->nXf, Txf);
            // the basis must reproduce every row (it does by construction; guards the rank threshold)
            double err = 0.0, scale = 0.0;
            auto check = [&](const double *rowsrc, int nr, const std::vector<double> &T) {
                for (int r = 0; r < nr; ++r)
                    for (int j = 0; j < n; ++j) {
                        double v = 0.0;
                        for (int a = 0; a < po; ++a) v += T[(size_t)r * po + a] * Co[(size_t)a * n + j];
                        err = std::max(err, fabs(v - rowsrc[(size_t)r * n + j]));
                        scale = std::max(scale, fabs(rowsrc[(size_t)r * n + j]));
                    }
            };
            check(Cq.data(), ncq, Tc); check(Cqf.data(), ncqf, Tcf); check(pr->XA, pr->nX, Tx); check(pr->XfA, pr->nXf, Txf);
            ok = err <= 1e-10 * std::max(scale, 1e-300);
        }
        if (ok) {
            Sc.assign((size_t)po * po, 0.0); ScN.assign((size_t)po * po, 0.0);
            for (int a = 0; a < po; ++a)
                for (int b = 0; b < po; ++b) {
                    double v = 0.0, vf = 0.0;
                    for (int r = 0; r < ncq; ++r) v += Tc[(size_t)r * po + a] * Tc[(size_t)r * po + b];
                    for (int r = 0; r < ncqf; ++r) vf += Tcf[(size_t)r * po + a] * Tcf[(size_t)r * po + b];
                    Sc[a * po + b] = v; ScN[a * po + b] = v + vf;
                }
            Cz2.assign((size_t)po * nz, 0.0); Czf2.assign((size_t)po * nz, 0.0);
            for (int a = 0; a < po; ++a)
                for (int b = 0; b < nz; ++b) {
                    double v = 0.0, vf = 0.0;
                    for (int i = 0; i < n; ++i) { v += Co[(size_t)a * n + i] * Ht2[(size_t)i * nz + b]; vf += Co[(size_t)a * n + i] * Htf2[(size_t)i * nz + b]; }
                    Cz2[a * nz + b] = v; Czf2[a * nz + b] = vf;
                }
            d.po = po;
            d.KT = (N * po + 15) / 16;
            d.cond = 1;
            if (qpc::lds_doubles(d, NTHREADS) * sizeof(double) > (size_t)160 * 1024) { d.cond = 0; d.po = 0; d.KT = 0; }
            // both constant output blocks positive definite (Cholesky with pivots above 1e-8 of the largest diagonal entry, as
            // oracle/condensed_ipm.py: spd_small): the Newton solves take dy from the solved system (qpc::newton_solve)
            auto spd_small = [&](const std::vector<double> &S) {
                std::vector<double> Lc(S);
                double dmax = 0.0;
                for (int a = 0; a < po; ++a) dmax = std::max(dmax, std::fabs(S[(size_t)a * po + a]));
                for (int i = 0; i < po; ++i)
                    for (int j = 0; j <= i; ++j) {
                        double v = Lc[(size_t)i * po + j];
                        for (int q = 0; q < j; ++q) v -= Lc[(size_t)i * po + q] * Lc[(size_t)j * po + q];
                        if (i == j) { if (!(v > 1e-8 * dmax)) return false; Lc[(size_t)i * po + i] = std::sqrt(v); }
                        else Lc[(size_t)i * po + j] = v / Lc[(size_t)j * po + j];
                    }
                return true;
            };
            d.ls_pd = (d.cond && spd_small(Sc) && spd_small(ScN)) ? 1 : 0;
        }
        // ---- lean kernels (locp_lean.h): p_o = 2, diagonal input Hessians, n_u = 4 or 8 (the split-panel shapes too: the lean
        // kernels have no W panel; what they hand over goes to the split fused kernel);
        // j0 = the smallest first LDS-resident stage for which the carve fits; the Gram tile tasks of the 8 waves
        if (d.cond && d.po == 2 && d.diagD && (m == 4 || m == 8) && !getenv("SRH_QP_NO_LEAN")) {
            int j0 = -1;
            for (int t = 0; t <= N; ++t)
                if (ql::lds_doubles(d, NTHREADS, t) * sizeof(double) <= (size_t)160 * 1024) { j0 = t; break; }
            // A/B knob (read when the constants are built): SRH_LEAN_J0=k keeps the packed rows of the stages < k in the L2 block even
            // where the LDS has room for them -- with SRH_LEAN_NO_FIXED=1 (run-time layouts) this measures what streaming G from L2
            // costs the interior point (round 6: the price of any layout that gives up LDS for a second resident workgroup)
            if (const char *e = getenv("SRH_LEAN_J0")) { if (j0 >= 0) j0 = std::min(N - 1, std::max(j0, atoi(e))); }
            // The half-size workgroup (round 6; chosen for batches above the CU count, SRH_LEAN_HALF=0 / 1 overrides): 256 threads and <= 80 KB of LDS so that TWO
            // rollouts share a CU -- every packed row of G in the L2 block (j0 = N), a thread owns an input and a state-row slot
            // (ql::ipm_box4), Theta^T condensed in two column passes.  Only where the layout fits and lean.hip has the instantiation.
            bool half = false;
            if (const char *eh = getenv("SRH_LEAN_HALF")) want_half = atoi(eh) != 0;
            // (only the shapes lean.hip instantiates the half-size kernels for: BASELINE C2 -- Diamond r = 30, N = 50, U box, 4 state rows)
            const bool half_shape = m == 4 && n == 60 && N == 50 && pr->nX == 4 && pr->nXf == 0 && pr->n_z == 6 && d.po == 2;
            // (the half-size kernels exist as fixed-layout instantiations only: SRH_LEAN_NO_FIXED=1 rules them out with the other fixed layouts)
            if (want_half && half_shape && getenv("SRH_LEAN_NO_FIXED") == nullptr && j0 >= 0 && j0 < N && pr->nU == 2 * m) {
                QPDims dh = d;
                dh.lean_half = 1; dh.lean_j0 = N;
                const int RXh = pr->nX + pr->nXf, GXh = RXh == 0 ? 1 : (RXh <= 2 ? 2 : (RXh <= 4 ? 4 : 8));
                const int MTh = dh.NPa / 16, wide = dh.KT - ql::half_split_tile(dh.KT);
                const ql::Sizes sh = ql::sizes(dh, 256, N);
                half = ql::lds_doubles(dh, 256, N) * sizeof(double) <= (size_t)80 * 1024 && wide * MTh <= 4 * ql::CONDENSE_SLOTS &&
                       N * m <= 256 && N * GXh <= 256 && RXh <= 8 && (N * m) % GXh == 0 &&
                       sh.regX >= (size_t)2 * (N + 1) * n + (size_t)2 * N * m + (size_t)pr->nX * n &&
                       ql::half_l2_off(dh) + 16 * (size_t)dh.KT + (size_t)(N / 2 + 2) <= qc_work_doubles(dh);
                if (half) { d.lean_half = 1; j0 = N; }
            }
            std::vector<int> sched;
            // (the SCP loop of the lean GuSTO kernel stages both trajectories in the K-tile area between two QPs: csrc/lean.hip)
            const int lthreads = half ? 256 : NTHREADS;
            const bool stage_fits = j0 >= 0 && ql::sizes(d, lthreads, j0).regX >= (size_t)2 * (N + 1) * n + (size_t)2 * N * m + (size_t)pr->nX * n;
            if (j0 >= 0 && (j0 < N || half) && stage_fits && (half || ql::condense_fits(d, NTHREADS / 64)) && lean_gram_schedule(N, m, d.KT, lthreads / 64, sched)) {
                d.lean = 1;
                d.lean_j0 = j0;

                // rows next to their sums (ql::ipm_box): the reference's HyperRectangle layout of the input rows (rows 2 b,
                // 2 b + 1 act on input b alone, utils.py:390-414), at most 8 state rows per stage, everything in 512 threads
                bool box = pr->nU == 2 * m && !getenv("SRH_QP_NO_BOX");
                for (int r = 0; r < pr->nU && box; ++r)
                    for (int b = 0; b < m; ++b) {
                        const bool want = b == r / 2;
                        if ((pr->UA[r * m + b] != 0.0) != want) { box = false; break; }
                    }
                const int RXa = pr->nX + pr->nXf;
                const int GX = RXa == 0 ? 1 : (RXa <= 2 ? 2 : (RXa <= 4 ? 4 : 8));
                if (box && RXa <= 8 && N * m + N * GX <= NTHREADS && (N * m) % GX == 0) d.lean = 2;
                if ((rc = C.gram_sched.upload(sched.data(), sizeof(int) * sched.size()))) return rc;
            }
        }
        if (d.cond) {
            if ((rc = up(C.Co, Co.data(), Co.size())) || (rc = up(C.Sc, Sc.data(), Sc.size())) || (rc = up(C.ScN, ScN.data(), ScN.size())) ||
                (rc = up(C.Tx, Tx.data(), (size_t)pr->nX * po)) || (rc = up(C.Txf, Txf.data(), (size_t)pr->nXf * po)) ||
                (rc = up(C.Cz2, Cz2.data(), Cz2.size())) || (rc = up(C.Czf2, Czf2.data(), Czf2.size())))
                return rc;
        }
    }
    return SRH_OK;
}

// dynamic LDS of a QP / GuSTO kernel: the larger of the two layouts
inline size_t qp_kernel_lds_bytes(const QPDims &d) {
    const size_t a = qp_lds_bytes(d, NTHREADS), b = d.cond ? qpc::lds_doubles(d, NTHREADS) * sizeof(double) : 0;
    return srh::lds_request(std::max(a, b));
}
inline size_t lean_kernel_lds_bytes(const QPDims &d) {
    if (d.lean_half) return srh::lds_request(ql::lds_doubles(d, 256, d.lean_j0) * sizeof(double));     // (<= 80 KB by construction; whole allocation granules: 81 920 B, two per CU)
    return srh::lds_request(ql::lds_doubles(d, NTHREADS, d.lean_j0) * sizeof(double));
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
