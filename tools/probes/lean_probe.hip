// Microbenchmark of the phases of the lean condensed interior point (csrc/locp_lean.h) on ONE workgroup with synthetic data
// at the C2 (n_u = 4) or C5 (n_u = 8) shape: shader clocks per call, averaged over REPS calls.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I soft-robot-control_amd/csrc tools/probes/lean_probe.hip -o gpurun_variants/lean_probe
#include "scp_host.h"
#include "gram_chol_pipeline.h"
#include "gram_exp.h"
#include "g_times_pairs.h"
#include "gT_times_fixed.h"
#include <cstdio>
#include <vector>
#include <random>

namespace srh { void set_error(const char *, ...) {} void *pool_take(size_t b) { void *p = nullptr; (void)hipMalloc(&p, b); return p; } void pool_give(void *p, size_t) { (void)hipFree(p); } }

#ifndef PROBE_M
#define PROBE_M 4
#endif
constexpr int REPS = 20;

__global__ __launch_bounds__(512) void probe(QPDims d, QPConst c, QPDyn dyn, double *work, double *x0, long long *out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    qp::specialise<PROBE_M, 60>(d);
    ql::Lds L;
    ql::lds_carve(L, (lptr)smem, d, 512);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int N = d.N, m = d.m, NP = N * d.po, ldG = 16 * d.KT, nm = N * m;
    gptr base = (gptr)work;
    gptr gh = base + d.qc_off;
    ql::GPack g{(cgptr)gh, (clptr)L.Gt, d.lean_j0, m, NP};
    // synthetic contents
    const size_t total = ql::lds_doubles(d, 512, d.lean_j0);
    for (size_t e = tid; e < total; e += nt) ((lptr)smem)[e] = 0.001 * (double)((e * 2654435761u) % 1000) - 0.5;
    for (int e = tid; e < ql::goff(d.lean_j0, m, NP) + 64; e += nt) gh[e] = 0.001 * (double)((e * 40503u) % 1000) - 0.5;
    __syncthreads();
    for (int e = tid; e < nm; e += nt) { L.Ldi[e] = 1.0 + 0.001 * e; L.u[e] = 0.01 * e; }
    for (int e = tid; e < N * 4; e += nt) L.Ls[e] = (e & 3) == 1 ? 0.0 : 1.0 + 0.01 * (e & 3);
    for (int e = tid; e < ldG + ql::YPAD; e += nt) { L.ya[e] = e < NP ? 0.01 * e : 0.0; L.yd[e] = L.ya[e]; L.yg[e] = L.ya[e]; }
    for (int k = tid; k < N; k += nt) L.idxl[k] = dyn.idx ? dyn.idx[k] : k;
    __syncthreads();
    // ---- self-check of the products against naive loops over the packed store (max abs error -> out[32..])
    {
        auto Gat = [&](int j, int b, int i) -> double {              // G^T[(j,b)][i], zero outside the stored part
            if (i < 2 * j || i >= NP) return 0.0;
            const int at = ql::goff(j, m, NP) + b * (NP - 2 * j) + (i - 2 * j);
            return j < d.lean_j0 ? (double)gh[at] : (double)L.Gt[at - ql::goff(d.lean_j0, m, NP)];
        };
        double *chk = work + 200000;                                 // scratch far behind everything else
        ql::g_times<PROBE_M>(d, g, L, L.u, L.yb);
        double e0 = 0.0;
        for (int i = tid; i < NP; i += nt) {
            double sref = 0.0;
            for (int j = 0; j < N; ++j) for (int b = 0; b < m; ++b) sref += Gat(j, b, i) * L.u[j * m + b];
            e0 = fmax(e0, fabs(sref - L.yb[i]) / (1.0 + fabs(sref)));
        }
        e0 = wg::reduce(e0, 1, L.red);
        {   // the balanced form against the same naive sums
            auto Wa = ql::all_waves();
            ql::g_times_pairs<PROBE_M, false>(d, g, L, L.u, L.yc, Wa);
            double e0b = 0.0;
            for (int i = tid; i < ldG; i += nt) {
                double sref = 0.0;
                if (i < NP) for (int j = 0; j < N; ++j) for (int b = 0; b < m; ++b) sref += Gat(j, b, i) * L.u[j * m + b];
                e0b = fmax(e0b, fabs(sref - L.yc[i]) / (1.0 + fabs(sref)));
            }
            e0b = wg::reduce(e0b, 1, L.red);
            if (tid < 2) L.flag[4 + tid] = 0;
            __syncthreads();
            auto Wh = ql::half_waves(L.flag + 4);
            if (ql::half_of_wave(tid >> 6) == 1) ql::g_times_pairs<PROBE_M, true>(d, g, L, L.u, L.yd, Wh);
            __syncthreads();
            double e0c = 0.0;
            for (int i = tid; i < ldG; i += nt) e0c = fmax(e0c, fabs(L.yd[i] - L.yc[i]));
            e0c = wg::reduce(e0c, 1, L.red);
            if (tid == 0) { ((double *)out)[42] = e0b; ((double *)out)[43] = e0c; }
            for (int e = tid; e < ldG + ql::YPAD; e += nt) L.yd[e] = L.ya[e];
            __syncthreads();
        }
        ql::gT_times<PROBE_M>(d, g, L, L.ya, L.yg, L.du, L.tc);
        double e1 = 0.0;
        for (int r = tid; r < nm; r += nt) {
            const int j = r / m, b = r % m;
            double s1 = 0.0, s2 = 0.0;
            for (int i = 0; i < NP; ++i) { s1 += Gat(j, b, i) * L.ya[i]; s2 += Gat(j, b, i) * L.yg[i]; }
            e1 = fmax(e1, fmax(fabs(s1 - L.du[r]) / (1.0 + fabs(s1)), fabs(s2 - L.tc[r]) / (1.0 + fabs(s2))));
        }
        e1 = wg::reduce(e1, 1, L.red);
        // Gram: K = I + Ls^T (G D^-1 G^T) Ls scaled to a unit diagonal, against a naive evaluation (upper tiles)
        ql::gram<PROBE_M>(d, c, g, L);
        double e2 = 0.0;
        for (int e = tid; e < NP * NP; e += nt) {
            const int i1 = e / NP, i2 = e % NP;
            if (i1 > i2) continue;
            // (Ls^T Ky Ls)[i1][i2] = sum_{a1,a2} Ls[k1][a1][i1&1] Ky[2k1+a1][2k2+a2] Ls[k2][a2][i2&1]
            const int k1 = i1 >> 1, k2 = i2 >> 1;
            double v = 0.0;
            for (int a1 = 0; a1 < 2; ++a1) for (int a2 = 0; a2 < 2; ++a2) {
                const double l1 = L.Ls[k1 * 4 + a1 * 2 + (i1 & 1)], l2 = L.Ls[k2 * 4 + a2 * 2 + (i2 & 1)];
                if (l1 == 0.0 || l2 == 0.0) continue;
                double ky = 0.0;
                for (int j = 0; j < N; ++j) for (int b = 0; b < m; ++b) { const double s = L.Ldi[j * m + b]; ky += Gat(j, b, 2 * k1 + a1) * Gat(j, b, 2 * k2 + a2) * s * s; }
                v += l1 * ky * l2;
            }
            if (i1 == i2) v += 1.0;
            chk[e] = v;
        }
        __syncthreads();
        for (int e = tid; e < NP * NP; e += nt) {
            const int i1 = e / NP, i2 = e % NP;
            if (i1 > i2) continue;
            const double ref = chk[e] / sqrt(chk[i1 * NP + i1] * chk[i2 * NP + i2]);
            const int I = i1 >> 4, J = i2 >> 4;
            const double got = L.B[(size_t)qpc::tile_index(I, J, d.KT) * ql::TSZ + (i1 & 15) * ql::TS + (i2 & 15)];
            e2 = fmax(e2, fabs(ref - got));
        }
        e2 = wg::reduce(e2, 1, L.red);
        if (tid == 0) { ((double *)out)[32] = e0; ((double *)out)[33] = e1; ((double *)out)[34] = e2; }
        __syncthreads();
        // rollout against qp::rollout (u = the work block's leading doubles scaled down)
        {
            const int n = d.n;
            gptr xa = (gptr)(work + 210000), xb = xa + (N + 1) * n, uu = xb + (N + 1) * n;
            for (int e = tid; e < nm; e += nt) uu[e] = 0.01 * (double)(e % 17) - 0.05;
            __syncthreads();
            ql::rollout<PROBE_M, 60>(d, dyn, (cgptr)x0, (cgptr)uu, xa, L);
            __syncthreads();
            QPLds Lq{};
            Lq.v1 = L.v1; Lq.v2 = L.v2; Lq.Qu = L.Qu; Lq.part = L.part; Lq.red = L.red; Lq.idxl = L.idxl; Lq.flag = L.flag;
            QPData qd{(cgptr)x0, (cgptr)xa, (cgptr) nullptr, (cgptr) nullptr, (cgptr) nullptr, 1e4, 1.0, (gptr) nullptr};
            qp::rollout(d, dyn, qd, (cgptr)uu, xb, Lq);
            double e3 = 0.0;
            for (int e = tid; e < (N + 1) * n; e += nt) e3 = fmax(e3, fabs(xa[e] - xb[e]) / (1.0 + fabs(xb[e])));
            e3 = wg::reduce(e3, 1, L.red);
            // condensation against qpc::condense (dense G^T in the L2 block): a qpc view of the same LDS regions
            qpc::Lds Lc = L;
            Lc.A = L.panel;
            QCWork qw; qw.GT = (gptr)(work + 230000);
            qpc::condense<PROBE_M, 60>(d, c, dyn, (cgptr)xb, qw, Lc);
            __syncthreads();
            // copy the dense result aside, then the packed condensation (it overwrites LDS regions only)
            ql::condense<PROBE_M, 60>(d, c, dyn, (cgptr)xb, gh, L);
            __syncthreads();
            double e4 = 0.0;
            for (int e = tid; e < nm * NP; e += nt) {
                const int r = e / NP, i = e % NP, j = r / m, b = r % m;
                const double ref = qw.GT[(size_t)r * ldG + i];
                e4 = fmax(e4, fabs(ref - Gat(j, b, i)) / (1.0 + fabs(ref)));
            }
            e4 = wg::reduce(e4, 1, L.red);
            if (tid == 0) { ((double *)out)[35] = e3; ((double *)out)[36] = e4; }
            __syncthreads();
        }
    }
    {   // the pipelined Gram + Cholesky (ql::gram_chol) against the two-phase form (gram, then qpc::tile_cholesky) on the same data
        double *keep = work + 260000;
        const int nt_ = d.KT * (d.KT + 1) / 2;
        ql::gram<PROBE_M>(d, c, g, L);
        const bool ok1 = qpc::tile_cholesky(d, L);
        __syncthreads();
        for (int e = tid; e < nt_ * ql::TSZ; e += nt) keep[e] = L.B[e];
        for (int e = tid; e < d.KT * ql::TSZ; e += nt) keep[nt_ * ql::TSZ + e] = L.Rinv[e];
        for (int e = tid; e < ldG; e += nt) keep[(nt_ + d.KT) * ql::TSZ + e] = L.ks[e];
        __syncthreads();
        const bool ok2 = ql::gram_chol<PROBE_M>(d, g, L);
        double e5 = 0.0, e6 = 0.0, e7 = 0.0;
        for (int e = tid; e < nt_ * ql::TSZ; e += nt) { const int rr = (e % ql::TSZ) / ql::TS, cc = (e % ql::TSZ) % ql::TS; if (cc < 16 && rr < 16) e5 = fmax(e5, fabs(keep[e] - L.B[e])); }
        for (int e = tid; e < d.KT * ql::TSZ; e += nt) { const int cc = (e % ql::TSZ) % ql::TS; if (cc < 16) e6 = fmax(e6, fabs(keep[nt_ * ql::TSZ + e] - L.Rinv[e])); }
        for (int e = tid; e < ldG; e += nt) e7 = fmax(e7, fabs(keep[(nt_ + d.KT) * ql::TSZ + e] - L.ks[e]));
        e5 = wg::reduce(e5, 1, L.red); e6 = wg::reduce(e6, 1, L.red); e7 = wg::reduce(e7, 1, L.red);
        if (tid == 0) { ((double *)out)[37] = e5; ((double *)out)[38] = e6; ((double *)out)[39] = e7; ((double *)out)[40] = (ok1 ? 1.0 : 0.0) + (ok2 ? 2.0 : 0.0); }
        __syncthreads();
    }
    long long t0, t1;
    int slot = 0;
#define TIME(...)                                                      \
    __syncthreads(); t0 = clock64();                                   \
    for (int r = 0; r < REPS; ++r) { __VA_ARGS__; }                           \
    __syncthreads(); t1 = clock64(); if (tid == 0) out[slot] = (t1 - t0) / REPS; ++slot;
    TIME(ql::g_times<PROBE_M>(d, g, L, L.u, L.yb));                                               // 0
    TIME(ql::gT_times<PROBE_M>(d, g, L, L.ya, (clptr) nullptr, L.du, (lptr) nullptr));            // 1
    TIME(ql::gT_times<PROBE_M>(d, g, L, L.ya, L.yg, L.du, L.tc));                                 // 2
    TIME(ql::gram<PROBE_M>(d, c, g, L));                                                          // 3
    // a well conditioned K for the factorisation: diagonal 4, off-diagonal small (the Gram product above left something else)
    auto fillK = [&]() {
        const int ntl = d.KT * (d.KT + 1) / 2;
        for (int e = tid; e < ntl * 256; e += nt) {
            int t = e >> 8, I = 0, tt = t;
            while (tt >= d.KT - I) { tt -= d.KT - I; ++I; }
            const int J = I + tt, r = (e >> 4) & 15, cc = e & 15;
            L.B[(size_t)t * ql::TSZ + r * ql::TS + cc] = (I == J && r == cc) ? 4.0 : 0.01 * (double)(((16 * I + r) * 31 + (16 * J + cc) * 17) % 13) / 13.0;
        }
        __syncthreads();
    };
    long long tch = 0;
    for (int r = 0; r < REPS; ++r) { fillK(); __syncthreads(); t0 = clock64(); qpc::tile_cholesky(d, L); __syncthreads(); tch += clock64() - t0; }
    if (tid == 0) out[slot] = tch / REPS; ++slot;                                              // 4
    TIME(qpc::k_solve(d, L, L.yc));                                                            // 5
    { long long tc = 0; for (int r = 0; r < REPS; ++r) { fillK(); __syncthreads(); t0 = clock64(); if (tid < 64) qpc::chol16(L.B, L.Rinv); __syncthreads(); tc += clock64() - t0; }
      if (tid == 0) out[20] = tc / REPS; }
    { long long tc = 0; for (int r = 0; r < REPS; ++r) { fillK(); __syncthreads(); t0 = clock64(); if (tid < 64) qpc::tile_update(L.B + 7 * ql::TSZ, L.B + ql::TSZ, L.B + ql::TSZ, tid & 15, (tid & 63) >> 4); __syncthreads(); tc += clock64() - t0; }
      if (tid == 0) out[21] = tc / REPS; }
    {   // chol16 itself: R^T R against the tile it was given, Rinv R against the identity
        fillK();
        double *Acopy = work + 290000;
        for (int e = tid; e < 256; e += nt) { const int i = e >> 4, j = e & 15; if (j < i) L.B[i * ql::TS + j] = L.B[j * ql::TS + i]; }      // fillK's tile is not symmetric
        __syncthreads();
        for (int e = tid; e < 256; e += nt) Acopy[e] = L.B[(e >> 4) * ql::TS + (e & 15)];
        __syncthreads();
        if (tid < 64) qpc::chol16(L.B, L.Rinv);
        __syncthreads();
        double ec = 0.0, ei = 0.0;
        for (int e = tid; e < 256; e += nt) {
            const int i = e >> 4, j = e & 15;
            double s1 = 0.0, s2 = 0.0;
            for (int k = 0; k < 16; ++k) { s1 += L.B[k * ql::TS + i] * L.B[k * ql::TS + j]; s2 += L.Rinv[i * ql::TS + k] * L.B[k * ql::TS + j]; }
            ec = fmax(ec, fabs(s1 - Acopy[i * 16 + j])); ei = fmax(ei, fabs(s2 - (i == j ? 1.0 : 0.0)));
        }
        ec = wg::reduce(ec, 1, L.red); ei = wg::reduce(ei, 1, L.red);
        if (tid == 0) { ((double *)out)[48] = ec; ((double *)out)[49] = ei; }
        __syncthreads();
    }
    TIME(qpc::stage_factors(d, c, (cgptr)(base + 4096), L));                                   // 6
    TIME(ql::rollout<PROBE_M, 60>(d, dyn, (cgptr)x0, (cgptr) nullptr, base, L));                  // 7
    TIME(ql::condense<PROBE_M, 60>(d, c, dyn, (cgptr)base, gh, L));                               // 8
    TIME(qpc::dinv_apply(d, L, L.ta));                                                         // 9
    TIME(qpc::ls_apply<qpc::LS_TR>(d, L, L.yb, L.yc));                                         // 10
    { double v = tid; TIME(v = wg::reduce(v, 1, L.red)); if (v < 0) out[63] = 1; }           // 11
    TIME(__syncthreads());                                                                     // 12
    {   // where the Gram fill's clocks go: parts switched off, per-wave clocks up to the first barrier (out[64 ..])
        long long *wt = out + 64;
        long long tg[5];
        __syncthreads(); t0 = clock64(); for (int r = 0; r < REPS; ++r) ql::gram_exp<PROBE_M, 0>(d, c, g, L, wt); __syncthreads(); tg[0] = (clock64() - t0) / REPS;
        __syncthreads(); t0 = clock64(); for (int r = 0; r < REPS; ++r) ql::gram_exp<PROBE_M, 1>(d, c, g, L, wt + 8); __syncthreads(); tg[1] = (clock64() - t0) / REPS;
        __syncthreads(); t0 = clock64(); for (int r = 0; r < REPS; ++r) ql::gram_exp<PROBE_M, 2>(d, c, g, L, wt + 16); __syncthreads(); tg[2] = (clock64() - t0) / REPS;
        __syncthreads(); t0 = clock64(); for (int r = 0; r < REPS; ++r) ql::gram_exp<PROBE_M, 4>(d, c, g, L, wt + 24); __syncthreads(); tg[3] = (clock64() - t0) / REPS;
        __syncthreads(); t0 = clock64(); for (int r = 0; r < REPS; ++r) ql::gram_exp<PROBE_M, 7>(d, c, g, L, wt + 32); __syncthreads(); tg[4] = (clock64() - t0) / REPS;
        if (tid == 0) for (int i = 0; i < 5; ++i) out[56 + i] = tg[i];
        __syncthreads();
    }
    TIME(ql::gram_chol<PROBE_M>(d, g, L));                                                        // 13
    TIME(ql::gram<PROBE_M>(d, c, g, L); qpc::tile_cholesky(d, L));                                // 14
    TIME(ql::unit_tiles(d, L));                                                                   // 15
    TIME(ql::k_solve_unit(d, L, L.yc));                                                           // 16
    { auto Wa = ql::all_waves(); __syncthreads(); t0 = clock64(); for (int r = 0; r < REPS; ++r) ql::g_times_pairs<PROBE_M, false>(d, g, L, L.u, L.yb, Wa);
      __syncthreads(); t1 = clock64(); if (tid == 0) out[19] = (t1 - t0) / REPS; }
    {   // the fixed-layout product (N = 50 and the probe's j0 as compile-time constants) against the run-time form
        constexpr int PJ0 = PROBE_M == 4 ? 7 : 24;
        if (d.lean_j0 == PJ0 && d.N == 50) {
            ql::GPackT<50, PJ0> gf{g.gh, g.gt, g.j0, g.m, g.NP};
            ql::g_times<PROBE_M>(d, g, L, L.u, L.yb);
            ql::g_times<PROBE_M>(d, gf, L, L.u, L.yc);
            double ef = 0.0;
            for (int i = tid; i < ldG; i += nt) ef = fmax(ef, fabs(L.yb[i] - L.yc[i]) / (1.0 + fabs(L.yb[i])));
            ef = wg::reduce(ef, 1, L.red);
            if (tid < 2) L.flag[4 + tid] = 0;
            __syncthreads();
            auto Wh = ql::half_waves(L.flag + 4);
            if (ql::half_of_wave(tid >> 6) == 1) ql::g_times<PROBE_M, true>(d, gf, L, L.u, L.yd, Wh);
            __syncthreads();
            double eh = 0.0;
            for (int i = tid; i < ldG; i += nt) eh = fmax(eh, fabs(L.yd[i] - L.yc[i]));
            eh = wg::reduce(eh, 1, L.red);
            __syncthreads(); t0 = clock64(); for (int r = 0; r < REPS; ++r) ql::g_times<PROBE_M>(d, gf, L, L.u, L.yb);
            __syncthreads(); t1 = clock64();
            if (tid == 0) { out[22] = (t1 - t0) / REPS; ((double *)out)[44] = ef; ((double *)out)[45] = eh; }
            // gT_times, fixed layout against the run-time form (two right-hand sides), whole workgroup and half set
            ql::gT_times<PROBE_M>(d, g, L, L.ya, L.yg, L.du, L.tc);
            { auto Wq = ql::all_waves(); ql::gT_times_fixed_or_not<PROBE_M, false>(d, gf, L, L.ya, L.yg, L.ta, L.tb, Wq); }
            double eg = 0.0;
            for (int e = tid; e < nm; e += nt) eg = fmax(eg, fmax(fabs(L.du[e] - L.ta[e]), fabs(L.tc[e] - L.tb[e])) / (1.0 + fabs(L.du[e])));
            eg = wg::reduce(eg, 1, L.red);
            if (tid < 2) L.flag[4 + tid] = 0;
            __syncthreads();
            auto Wg = ql::half_waves(L.flag + 4);
            if (ql::half_of_wave(tid >> 6) == 1) ql::gT_times_fixed_or_not<PROBE_M, true>(d, gf, L, L.ya, (clptr) nullptr, L.tb, (lptr) nullptr, Wg);
            __syncthreads();
            double eg2 = 0.0;
            for (int e = tid; e < nm; e += nt) eg2 = fmax(eg2, fabs(L.tb[e] - L.ta[e]));
            eg2 = wg::reduce(eg2, 1, L.red);
            __syncthreads(); t0 = clock64(); for (int r = 0; r < REPS; ++r) { auto Wq = ql::all_waves(); ql::gT_times_fixed_or_not<PROBE_M, false>(d, gf, L, L.ya, (clptr) nullptr, L.du, (lptr) nullptr, Wq); }
            __syncthreads(); t1 = clock64(); const long long tg1 = (t1 - t0) / REPS;
            __syncthreads(); t0 = clock64(); for (int r = 0; r < REPS; ++r) { auto Wq = ql::all_waves(); ql::gT_times_fixed_or_not<PROBE_M, false>(d, gf, L, L.ya, L.yg, L.du, L.tc, Wq); }
            __syncthreads(); t1 = clock64();
            if (tid == 0) { out[23] = tg1; out[24] = (t1 - t0) / REPS; ((double *)out)[46] = eg; ((double *)out)[47] = eg2; }
        }
    }
    // the factorisation on one SIMD half of the workgroup (counters in LDS, no set-wide barrier) while the other half waits; checked
    // against qpc::tile_cholesky on the same tiles
    {
        double *keep = work + 260000;
        const int nt_ = d.KT * (d.KT + 1) / 2;
        fillK(); qpc::tile_cholesky(d, L); __syncthreads();
        for (int e = tid; e < nt_ * ql::TSZ; e += nt) keep[e] = L.B[e];
        for (int e = tid; e < d.KT * ql::TSZ; e += nt) keep[nt_ * ql::TSZ + e] = L.Rinv[e];
        __syncthreads();
        long long tc = 0;
        for (int r = 0; r < REPS; ++r) {
            fillK(); if (tid < 4) L.flag[4 + tid] = 0; __syncthreads(); t0 = clock64(); auto W = ql::half_waves(L.flag + 4);
            if (ql::half_of_wave(tid >> 6) == 0) ql::tile_cholesky_set(d, L, W, L.flag + 5);
            __syncthreads(); tc += clock64() - t0;
        }
        if (tid == 0) out[17] = tc / REPS;
        double e5 = 0.0;
        for (int e = tid; e < nt_ * ql::TSZ; e += nt) { const int cc = (e % ql::TSZ) % ql::TS; if (cc < 16) e5 = fmax(e5, fabs(keep[e] - L.B[e])); }
        for (int e = tid; e < d.KT * ql::TSZ; e += nt) { const int cc = (e % ql::TSZ) % ql::TS; if (cc < 16) e5 = fmax(e5, fabs(keep[nt_ * ql::TSZ + e] - L.Rinv[e])); }
        e5 = wg::reduce(e5, 1, L.red);
        if (tid == 0) ((double *)out)[41] = e5;
    }
    { long long tc = 0; for (int r = 0; r < REPS; ++r) { if (tid < 4) L.flag[4 + tid] = 0; __syncthreads(); t0 = clock64(); auto W = ql::half_waves(L.flag + 4);
        if (ql::half_of_wave(tid >> 6) == 1) { for (int q = 0; q < 16; ++q) W.sync(); } __syncthreads(); tc += clock64() - t0; }
      if (tid == 0) out[18] = tc / REPS / 16; }
}

int main() {
    slocp_problem pr{};
    const int N = 50, n = 60, m = PROBE_M, nz = 6, P = 64;
    std::vector<double> H(nz * n, 0.0), Qz(nz * nz, 0.0), R(m * m, 0.0), UA(2 * m * m, 0.0), Ub(2 * m), XA(4 * n, 0.0), Xb(4, 50.0);
    for (int a = 0; a < 3; ++a) { H[a * n + 10 + a] = 1.0; H[(3 + a) * n + 40 + a] = 1.0; }
    Qz[3 * nz + 3] = 100.0; Qz[4 * nz + 4] = 100.0;
    for (int a = 0; a < m; ++a) { R[a * m + a] = 1e-5; UA[(2 * a) * m + a] = 1.0; UA[(2 * a + 1) * m + a] = -1.0; Ub[2 * a] = 1500.0; Ub[2 * a + 1] = 0.0; }
    XA[0 * n + 40] = 1; XA[1 * n + 40] = -1; XA[2 * n + 41] = 1; XA[3 * n + 41] = -1;
    pr.N = N; pr.n_x = n; pr.n_u = m; pr.n_z = nz; pr.H = H.data(); pr.Qz = Qz.data(); pr.R = R.data();
    pr.nU = 2 * m; pr.UA = UA.data(); pr.Ub = Ub.data(); pr.tr_active = 1;
    if (PROBE_M == 4) { pr.nX = 4; pr.XA = XA.data(); pr.Xb = Xb.data(); }
    QPConstHost C;
    if (build_consts(&pr, C)) { printf("build_consts failed\n"); return 1; }
    QPDims &d = C.dims;
    printf("lean %d j0 %d KT %d po %d lds %zu\n", d.lean, d.lean_j0, d.KT, d.po, lean_kernel_lds_bytes(d));
    if (!d.lean) return 1;
    d.qc_off = (long long)((qp_work_doubles(d) + 3) & ~(size_t)3);
    const size_t stride = (size_t)d.qc_off + qc_work_doubles(d) + 64 + 300000;
    std::mt19937 rng(1);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> Ad((size_t)P * n * n), Bd((size_t)P * n * m), dd((size_t)P * n), AdT(Ad.size()), BdT(Bd.size()), wk(stride), x0(n);
    for (int p = 0; p < P; ++p) {
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { const double v = (i == j ? 0.9 : 0.0) + 0.01 * nd(rng); Ad[(size_t)p * n * n + i * n + j] = v; AdT[(size_t)p * n * n + j * n + i] = v; }
        for (int i = 0; i < n; ++i) for (int j = 0; j < m; ++j) { const double v = 0.1 * nd(rng); Bd[(size_t)p * n * m + i * m + j] = v; BdT[(size_t)p * n * m + j * n + i] = v; }
        for (int i = 0; i < n; ++i) dd[(size_t)p * n + i] = 0.01 * nd(rng);
    }
    for (auto &v : wk) v = 0.5 + 0.1 * nd(rng);
    for (auto &v : x0) v = nd(rng);
    std::vector<int> idx(N);
    for (int k = 0; k < N; ++k) idx[k] = (k / 8) % P;          // the region changes every 8 stages (12 % of the stages)
    double *dA, *dAT, *dB, *dBT, *dD, *dW, *dx0; int *dI; long long *dout;
    hipMalloc(&dA, Ad.size() * 8); hipMalloc(&dAT, Ad.size() * 8); hipMalloc(&dB, Bd.size() * 8); hipMalloc(&dBT, Bd.size() * 8);
    hipMalloc(&dD, dd.size() * 8); hipMalloc(&dW, wk.size() * 8); hipMalloc(&dx0, n * 8); hipMalloc(&dI, N * 4); hipMalloc(&dout, 128 * 8);
    hipMemcpy(dA, Ad.data(), Ad.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dAT, AdT.data(), Ad.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bd.data(), Bd.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dBT, BdT.data(), Bd.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dD, dd.data(), dd.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dW, wk.data(), wk.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dx0, x0.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dI, idx.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0, 128 * 8);
    QPDyn dyn{(cgptr)dA, (cgptr)dAT, (cgptr)dB, (cgptr)dBT, (cgptr)dD, (cgiptr)dI};
    const size_t lds = lean_kernel_lds_bytes(d);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        probe<<<1, 512, lds>>>(d, C.view(), dyn, dW, dx0, dout);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); return 1; }
    }
    long long out[128];
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    const char *names[] = {"g_times", "gT_times(1)", "gT_times(2)", "gram", "tile_cholesky", "k_solve", "stage_factors", "rollout", "condense",
                           "dinv_apply", "ls_apply", "wg::reduce", "barrier", "gram_chol", "gram+cholesky", "unit_tiles", "k_solve_unit", "chol 4 waves", "set sync (4 w)"};
    for (int i = 0; i < 19; ++i) printf("%-14s %8lld clocks\n", names[i], out[i]);
    printf("self-check: g_times %.2e gT_times %.2e gram %.2e rollout %.2e condense %.2e\n", ((double *)out)[32], ((double *)out)[33], ((double *)out)[34], ((double *)out)[35], ((double *)out)[36]);
    printf("chol16 (one wave) %lld, tile_update (one wave) %lld; chol16: max |R^T R - A| %.2e, max |Rinv R - I| %.2e\n", out[20], out[21], ((double *)out)[48], ((double *)out)[49]);
    printf("g_times_pairs: %lld clocks; vs naive sums %.2e, half set vs whole workgroup %.2e\n", out[19], ((double *)out)[42], ((double *)out)[43]);
    printf("g_times_fixed: %lld clocks; vs g_times %.2e, half set vs whole workgroup %.2e\n", out[22], ((double *)out)[44], ((double *)out)[45]);
    printf("gT_times_fixed: %lld / %lld clocks (one / two right-hand sides); vs gT_times %.2e, half set vs whole workgroup %.2e\n", out[23], out[24], ((double *)out)[46], ((double *)out)[47]);
    printf("tile_cholesky_set (4 waves, LDS counters) vs qpc::tile_cholesky: max |d| %.2e\n", ((double *)out)[41]);
    printf("gram_exp: full %lld, no head %lld, no epilogue %lld, no LDS ranges %lld, none of the three %lld clocks\n", out[56], out[57], out[58], out[59], out[60]);
    for (int v = 0; v < 5; ++v) { printf("  per-wave clocks to the first barrier (variant %d):", v); for (int w8 = 0; w8 < 8; ++w8) printf(" %lld", out[64 + 8 * v + w8]); printf("\n"); }
    printf("gram_chol vs gram + tile_cholesky: max |dR| %.2e max |dRinv| %.2e max |dks| %.2e ok flags %.0f (3 = both)\n", ((double *)out)[37], ((double *)out)[38],
           ((double *)out)[39], ((double *)out)[40]);
    return 0;
}
