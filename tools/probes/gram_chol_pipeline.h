// NOT part of the library: the pipelined Gram fill + tile Cholesky of round 4 (measured, not adopted -- DESIGN.md section 12).
// Bit-identical factor to ql::gram + qpc::tile_cholesky, and no faster: 95.0 k clocks against 90.7 k at the C2 shape
// (tools/probes/lean_probe.hip).  One tile of K costs a wave ~8-10 k clocks (L2 head loads 1.5 k, Ls / scaling epilogue 2.7 k,
// 130 clocks per k-step), twice a chol16: the pipeline is bound by the tile a wave builds per step, not by the chain.
// Included by lean_probe.hip only.
#pragma once
namespace ql {

// ------------------------------------------------------------------ Gram fill and tile Cholesky as ONE pipeline (round 4)
// gram() + qpc::tile_cholesky() above are two phases: every tile of K is formed (37 k clocks at C2, all waves), then factored
// right-looking with wave 0 on the critical path (seven 16 x 16 diagonal factorisations, 35 k of 52 k clocks) while the other
// seven waves mostly wait.  Nothing in a tile of K depends on the factorisation, so the fill can run UNDER the factorisation:
// left-looking by tile rows, wave J owns tile column J and keeps the tile it is working on in its MFMA accumulators
//     step I:   wave 0: chol16 of K_II (LDS)                 | wave J > I: pending (I+1, J) = Gram part, Ls, scaling,
//                                                            |             minus R_s,I+1^T R_sJ for the finished rows s < I
//               barrier
//               wave J > I: R_IJ = Rinv_I^T K_IJ   (K_IJ straight from the accumulators: the C layout of
//                                                   v_mfma_f64_16x16x4 is its own B operand) -> LDS
//               barrier
//               wave J > I: pending (I+1, J) -= R_I,I+1^T R_IJ;  wave I+1 hands the finished diagonal tile to wave 0 (LDS)
//               barrier
// A tile is written to LDS once (as part of the factor) instead of once per trailing update.  The Jacobi scaling ks of K
// needs the diagonal of every diagonal tile before any off-diagonal tile can be scaled: the diagonal tiles are formed first
// (wave J: tile (J, J)), each scaled by its own diagonal.  Same K, same factor as gram() + tile_cholesky() up to the order of
// the additions inside a tile.  KT <= 8 (one wave per tile column).

// acc += (G W G^T) tile (I, J), J >= I: the A operand (columns 16 I .. of the packed rows, times 1 / D) and the B operand
// (columns 16 J ..) straight from the packed store, stages 2 j < 16 (I + 1) only
template <int MSEL>
__device__ __forceinline__ void gram_tile_acc(const GPack &g, clptr w2, int N, int I, int J, int l16, int kk, wg::qp_d4 &acc) {
    static_assert(MSEL == 4 || MSEL == 8, "lean Gram: n_u = 4 or 8");
    constexpr int M = MSEL, SPS = M / 4;
    const int NP = g.NP;
    const int goff0 = goff(g.j0, M, NP);
    const int jend = min(N, 8 * (I + 1)), jfull = min(jend, 8 * I + 1);
    const int ia = 16 * I + l16, dJ = 16 * (J - I);
    const bool diag = I == J;
    auto run = [&](auto src, int jb, int je, auto MASK) {
        constexpr bool masked = decltype(MASK)::value;
        int R = goff(jb, M, NP) + kk * (NP - 2 * jb) - 2 * jb + ia;
        int dl = M * (NP - 2 * jb) - 2 * kk - 2;
        constexpr int UN = 8 / SPS;                                          // stages per full trip: 8 k-steps, 16 loads in flight
        auto trip = [&](int j0, auto UNS) {
            constexpr int uns = decltype(UNS)::value, KS = uns * SPS;
            double av[KS], bv[KS];
#pragma unroll
            for (int us = 0; us < uns; ++us) {
                const int j = j0 + us;
#pragma unroll
                for (int sub = 0; sub < SPS; ++sub) {
                    const int u = us * SPS + sub;
                    const int base = R + 4 * sub * (NP - 2 * j);
                    av[u] = src[base] * w2[j * M + 4 * sub + kk];
                    bv[u] = src[base + dJ];
                    if constexpr (masked) {
                        const bool va = ia >= 2 * j;
                        av[u] = va ? av[u] : 0.0;
                        if (diag) bv[u] = va ? bv[u] : 0.0;
                    }
                }
                R += dl; dl -= 2 * M;
            }
#pragma unroll
            for (int u = 0; u < KS; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        };
        int j0 = jb;
        for (; j0 + UN <= je; j0 += UN) trip(j0, std::integral_constant<int, UN>{});
        for (; j0 < je; ++j0) trip(j0, std::integral_constant<int, 1>{});
    };
    const int hf = min(jfull, g.j0), he = min(jend, g.j0);
    if (hf > 0) run(g.gh, 0, hf, std::false_type{});
    if (he > hf) run(g.gh, hf, he, std::true_type{});
    if (jfull > g.j0) run(g.gt - goff0, g.j0, jfull, std::false_type{});
    if (jend > max(jfull, g.j0)) run(g.gt - goff0, max(jfull, g.j0), jend, std::true_type{});
}

// acc (raw G W G^T tile) -> Ls^T acc Ls (+ I on the diagonal tile); padding rows / columns (index >= NP): the identity
__device__ __forceinline__ void gram_tile_ls(Lds &L, int N, int NP, int I, int J, int l16, int kk, wg::qp_d4 &acc) {
    const int gjc = 16 * J + l16, kb = min(gjc >> 1, N - 1), bc = gjc & 1;
    clptr Lb = L.Ls + (size_t)kb * 4;
    const double cb_own = bc == 0 ? Lb[0] : Lb[3], cb_oth = bc == 0 ? Lb[2] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = kk + 4 * q, gi = 16 * I + r, ka = min(gi >> 1, N - 1), ar = gi & 1;
        clptr La = L.Ls + (size_t)ka * 4;
        double v = (gi < NP && gjc < NP) ? acc[q] : 0.0;
        const double vp = wg::dpp_mov<0xB1>(v);                 // the other column of the output stage
        v = fma(vp, cb_oth, v * cb_own);                        // (Ky Ls)
        const double vr = __shfl_xor(v, 16, 64);                // the other row of the output stage (kk ^ 1)
        v = ar == 0 ? fma(La[2], vr, La[0] * v) : La[3] * v;    // Ls^T (Ky Ls)
        if (I == J && r == l16) v += 1.0;
        acc[q] = v;
    }
}

// acc -= Ra^T Rb (tiles of the factor in LDS)
__device__ __forceinline__ void tile_sub(wg::qp_d4 &acc, clptr Ra, clptr Rb, int l16, int kk) {
    double av[4], bv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { av[s] = -Ra[(4 * s + kk) * TS + l16]; bv[s] = Rb[(4 * s + kk) * TS + l16]; }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
}

template <int MSEL>
__device__ __forceinline__ bool gram_chol(const QPDims &d, const GPack &g, Lds &L) {
    constexpr int M = MSEL;
    const int N = d.N, KT = d.KT, NP = g.NP;
    const int tid = threadIdx.x, nt = blockDim.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l16 = lane & 15, kk = lane >> 4;
    lptr w2 = L.tc;                                            // 1 / D per packed row
    for (int e = tid; e < N * M; e += nt) { const double s = L.Ldi[e]; w2[e] = s * s; }
    __syncthreads();
    const bool owner = wave < KT;                              // wave J owns tile column J (wave 0: the diagonal factorisations)
    auto tile = [&](int I, int J) -> lptr { return L.B + (size_t)qpc::tile_index(I, J, KT) * TSZ; };
    auto put = [&](lptr T, const wg::qp_d4 &a) {
#pragma unroll
        for (int q = 0; q < 4; ++q) T[(kk + 4 * q) * TS + l16] = a[q];
    };
    auto scale = [&](int I, int J, wg::qp_d4 &a) {
        const double sc = L.ks[16 * J + l16];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] *= L.ks[16 * I + kk + 4 * q] * sc;
    };
    // ---- diagonal tiles: K_JJ, its own Jacobi scaling, scaled tile to LDS
    if (owner) {
        const int J = wave;
        wg::qp_d4 a = {0.0, 0.0, 0.0, 0.0};
        gram_tile_acc<MSEL>(g, w2, N, J, J, l16, kk, a);
        gram_tile_ls(L, N, NP, J, J, l16, kk, a);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (kk + 4 * q == l16) {
                const double v = a[q], ri = rsqrt(v);
                L.ks[16 * J + l16] = ri * (1.5 - 0.5 * v * ri * ri);     // one Newton step: full double accuracy
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        scale(J, J, a);
        put(tile(J, J), a);
    }
    __syncthreads();
    // ---- tile row 0 (needs every ks): pending in the accumulators of its column's wave
    wg::qp_d4 P1 = {0.0, 0.0, 0.0, 0.0}, P2 = {0.0, 0.0, 0.0, 0.0};
    if (owner && wave >= 1) {
        gram_tile_acc<MSEL>(g, w2, N, 0, wave, l16, kk, P1);
        gram_tile_ls(L, N, NP, 0, wave, l16, kk, P1);
        scale(0, wave, P1);
    }
    bool ok = true;
    for (int I = 0; I < KT; ++I) {
        if (wave == 0) {
            ok = qpc::chol16(tile(I, I), L.Rinv + (size_t)I * TSZ) && ok;
        } else if (owner && wave >= I + 1 && I + 1 < KT) {
            const int J = wave;
            if (J == I + 1) {
                clptr T = tile(J, J);
#pragma unroll
                for (int q = 0; q < 4; ++q) P2[q] = T[(kk + 4 * q) * TS + l16];
            } else {
                P2 = {0.0, 0.0, 0.0, 0.0};
                gram_tile_acc<MSEL>(g, w2, N, I + 1, J, l16, kk, P2);
                gram_tile_ls(L, N, NP, I + 1, J, l16, kk, P2);
                scale(I + 1, J, P2);
            }
            for (int s = 0; s < I; ++s) tile_sub(P2, tile(s, I + 1), tile(s, J), l16, kk);
        }
        __syncthreads();
        if (I + 1 >= KT) break;
        if (owner && wave > I) {                               // panel: R_IJ = Rinv_I^T K_IJ
            clptr Ri = L.Rinv + (size_t)I * TSZ;
            double av[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) av[s] = Ri[(4 * s + kk) * TS + l16];
            wg::qp_d4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) r = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], P1[s], r, 0, 0, 0);
            put(tile(I, wave), r);
        }
        __syncthreads();
        if (owner && wave >= I + 1) {
            tile_sub(P2, tile(I, I + 1), tile(I, wave), l16, kk);
            if (wave == I + 1) put(tile(I + 1, I + 1), P2);
            P1 = P2;
        }
        __syncthreads();
    }
    if (tid == 0) L.flag[1] = ok ? 1 : 0;
    __syncthreads();
    return L.flag[1] != 0;
}


}  // namespace ql
