// Experimental copy of ql::gram (locp_lean.h) with parts switched off by FLAGS (1: no L2 head ranges, 2: no epilogue, 4: no LDS
// ranges) and per-wave clocks up to the first barrier -- where the 37.7 k clocks of the Gram fill go.  Probe only.
#pragma once
namespace ql {
template <int MSEL, int FLAGS, class GP>
__device__ __forceinline__ void gram_exp(const QPDims &d, const QPConst &c, const GP &g, Lds &L, long long *wt) {
    const long long tw0 = clock64();
    static_assert(MSEL == 4 || MSEL == 8, "lean Gram: n_u = 4 or 8");
    constexpr int M = MSEL, SPS = M / 4;                       // k-steps per stage
    const int N = d.N, KT = d.KT, NP = g.NP;
    const int tid = threadIdx.x, nt = blockDim.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l16 = lane & 15, kk = lane >> 4;
    const int goff0 = goff(g.j0, M, NP);
    lptr w2 = L.tc;                                            // 1 / D per packed row
    for (int e = tid; e < N * M; e += nt) { const double s = L.Ldi[e]; w2[e] = s * s; }
    __syncthreads();
    // this wave's tasks: all descriptors requested at once (one L2 latency instead of one per task)
    int tI[4], tJ0[4], tnJ[4];
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) {
        cgiptr task = c.gram_sched + (wave * 4 + slot) * 4;
        tI[slot] = task[0]; tJ0[slot] = task[1]; tnJ[slot] = task[2];
    }
#pragma unroll
    for (int slot = 0; slot < 4; ++slot) {
        tI[slot] = __builtin_amdgcn_readfirstlane(tI[slot]); tJ0[slot] = __builtin_amdgcn_readfirstlane(tJ0[slot]);
        tnJ[slot] = __builtin_amdgcn_readfirstlane(tnJ[slot]);
    }
    for (int slot = 0; slot < 4; ++slot) {
        const int I = tI[slot], J0 = tJ0[slot], nJ = tnJ[slot];
        if (nJ == 0) break;
        wg::qp_d4 acc[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        const int jend = min(N, 8 * (I + 1));                  // stages with 2 j < 16 (I + 1)
        const int jfull = min(jend, 8 * I + 1);                // stages with 2 j <= 16 I: every lane of the row is live
        const int ia = 16 * I + l16;
        const bool diag0 = J0 == I;                            // tile 0 of the task is the diagonal tile
        const int dJ = 16 * (J0 - I);                          // B operand of tile t: 16 (dJ / 16 + t) doubles behind A's
        // stages [jb, je) from `src` (indexed by the global packed offset); MASK: triangular part of the tile row; NJ tiles.
        // No masks for the padding columns i >= NP of the last tile row / column: what they read (finite or not) only
        // reaches the padding rows / columns of K, which are overwritten below.  1 / D goes onto the shared A operand.
        auto run = [&](auto src, int jb, int je, auto MASK, auto NJ) {
            constexpr bool masked = decltype(MASK)::value;
            constexpr int nj = decltype(NJ)::value;
            int R = goff(jb, M, NP) + kk * (NP - 2 * jb) - 2 * jb + ia;           // R(jb, b = kk) + column of the A operand
            int dl = M * (NP - 2 * jb) - 2 * kk - 2;
            constexpr int UN = 4 / SPS;                                      // stages per full trip: 4 k-steps
            auto trip = [&](int j0, auto UNS) {                              // UNS stages = UNS * SPS k-steps
                constexpr int uns = decltype(UNS)::value, KS = uns * SPS;
                double av[KS], bv[KS][nj];
#pragma unroll
                for (int us = 0; us < uns; ++us) {
                    const int j = j0 + us;
#pragma unroll
                    for (int sub = 0; sub < SPS; ++sub) {
                        const int u = us * SPS + sub;
                        const int base = R + 4 * sub * (NP - 2 * j);
                        av[u] = src[base] * w2[j * M + 4 * sub + kk];
#pragma unroll
                        for (int t = 0; t < nj; ++t) bv[u][t] = src[base + dJ + 16 * t];
                        if constexpr (masked) {
                            const bool va = ia >= 2 * j;
                            av[u] = va ? av[u] : 0.0;
                            if (diag0) bv[u][0] = va ? bv[u][0] : 0.0;
                        }
                    }
                    R += dl; dl -= 2 * M;
                }
#pragma unroll
                for (int u = 0; u < KS; ++u)
#pragma unroll
                    for (int t = 0; t < nj; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u][t], acc[t], 0, 0, 0);
            };
            int j0 = jb;
            for (; j0 + UN <= je; j0 += UN) trip(j0, std::integral_constant<int, UN>{});
            for (; j0 < je; ++j0) trip(j0, std::integral_constant<int, 1>{});
        };
        // four ranges: {L2 head, LDS} x {full, triangular}
        auto task_body = [&](auto NJ) {
            const int hf = min(jfull, g.j0), he = min(jend, g.j0);
            if constexpr (!(FLAGS & 1)) {
            if (hf > 0) run(g.gh, 0, hf, std::false_type{}, NJ);
            if (he > hf) run(g.gh, hf, he, std::true_type{}, NJ);
            }
            if constexpr (!(FLAGS & 4)) {
            if (jfull > g.j0) run(g.gt - goff0, g.j0, jfull, std::false_type{}, NJ);
            if (jend > max(jfull, g.j0)) run(g.gt - goff0, max(jfull, g.j0), jend, std::true_type{}, NJ);
            }
        };
        if (nJ == 4) task_body(std::integral_constant<int, 4>{});
        else if (nJ == 3) task_body(std::integral_constant<int, 3>{});
        else if (nJ == 2) task_body(std::integral_constant<int, 2>{});
        else task_body(std::integral_constant<int, 1>{});
        // ---- Ls on both sides, + I, raw tile to the store; the diagonal feeds the Jacobi scaling
        if constexpr (FLAGS & 2) { double sacc = 0.0; for (int t = 0; t < 4; ++t) for (int q = 0; q < 4; ++q) sacc += acc[t][q]; if (sacc == 1.2345e300) L.ks[0] = sacc; }
        else
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t >= nJ) continue;
            const int J = J0 + t;
            const int gjc = 16 * J + l16, kb = min(gjc >> 1, N - 1), bc = gjc & 1;
            clptr Lb = L.Ls + (size_t)kb * 4;
            const double cb_own = bc == 0 ? Lb[0] : Lb[3], cb_oth = bc == 0 ? Lb[2] : 0.0;
            lptr T = L.B + (size_t)qpc::tile_index(I, J, KT) * TSZ;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = kk + 4 * q, gi = 16 * I + r, ka = min(gi >> 1, N - 1), ar = gi & 1;
                clptr La = L.Ls + (size_t)ka * 4;
                double v = (gi < NP && gjc < NP) ? acc[t][q] : 0.0;      // padding rows / columns: exactly the identity
                const double vp = wg::dpp_mov<0xB1>(v);                 // the other column of the output stage
                v = fma(vp, cb_oth, v * cb_own);                        // (Ky Ls)
                const double vr = __shfl_xor(v, 16, 64);                // the other row of the output stage (kk ^ 1)
                v = ar == 0 ? fma(La[2], vr, La[0] * v) : La[3] * v;    // Ls^T (Ky Ls)
                const bool dg = I == J && r == l16;
                v += dg ? 1.0 : 0.0;
                const double ri = rsqrt(dg ? v : 1.0);
                if (dg) L.ks[gi] = ri * (1.5 - 0.5 * v * ri * ri);       // one Newton step: full double accuracy
                T[r * TS + l16] = v;
            }
        }
    }
    if (wt && lane == 0) wt[wave] = clock64() - tw0;
    __syncthreads();
    // ---- symmetric scaling to a unit diagonal (see qpc::gram for why it matters): every wave scales the tiles it wrote
    for (int slot = 0; slot < 4; ++slot) {
        const int I = tI[slot], J0 = tJ0[slot], nJ = tnJ[slot];
        if (nJ == 0) break;
        for (int t = 0; t < nJ; ++t) {
            lptr T = L.B + (size_t)qpc::tile_index(I, J0 + t, KT) * TSZ;
            const double sc = L.ks[16 * (J0 + t) + l16];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int r = kk + 4 * q; T[r * TS + l16] *= L.ks[16 * I + r] * sc; }
        }
    }
    __syncthreads();
}

}  // namespace ql
