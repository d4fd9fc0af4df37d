// Device view of a TPWL model (tables resident in HBM/L2) and workgroup-cooperative helpers.
// Reference: sofacontrol/tpwl/tpwl.py (nearest-neighbour TPWL), sofacontrol/scp/models/tpwl.py.
#pragma once
#include "dev_la.h"

struct TpwlDev {
    int P, r, n, m, nz;
    double w_q, w_v;
    cgptr qT, vT;            // (r x P) transposed point tables (coalesced over points)
    cgptr u;                          // (P x m)
    cgptr Ac, Bc, dc;       // continuous tables (P x n x n), (P x n x m), (P x n)
    cgptr AcT;                // (P x n x n) transposed
    cgptr Ad, Bd, dd;       // discrete tables or null
    cgptr AdT;                // transposed discrete A
    cgptr BdT;                // (P x m x n) transposed discrete B
    cgptr BcT;                // (P x m x n)
    cgptr H, z_ref;          // (nz x n), (nz) or null
};

namespace tpwl {

// argmin_i w_q ||q_i - q|| + w_v ||v_i - v||, first minimum (np.argmin), for the state x (LDS or
// global, x = [v; q]).  Executed by ONE wave (64 lanes); every lane returns the index.
template <typename XP>
__device__ inline int nearest_wave(const TpwlDev &T, XP x) {
    const int lane = SRH_TID & 63;
    double best = INFINITY;
    int besti = 0x7fffffff;
    // sum_j (tab[j][i] - x[xoff + j])^2 in the order j = 0, 1, ...: sixteen table / state loads are requested before the
    // first FMA (a rolled load -> FMA loop pays the L2 latency r times per point; same sums, same order)
    auto sqdist = [&](cgptr tab, int xoff, int i) {
        double sq = 0.0;
        for (int j0 = 0; j0 < T.r; j0 += 16) {
            double tv[16], xv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int j = j0 + q < T.r ? j0 + q : T.r - 1;
                tv[q] = tab[j * T.P + i];
                xv[q] = x[xoff + j];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (j0 + q < T.r) { const double e = tv[q] - xv[q]; sq = fma(e, e, sq); }
            }
        }
        return sq;
    };
    for (int i0 = 0; i0 < T.P; i0 += 64) {
        const int i = i0 + lane;
        double d = INFINITY;
        if (i < T.P) {
            d = T.w_q * sqrt(sqdist(T.qT, T.r, i));
            if (T.w_v != 0.0) d += T.w_v * sqrt(sqdist(T.vT, 0, i));
        }
        if (d < best) { best = d; besti = i; }
    }
    // wave argmin with smallest-index tie break
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    return besti;
}

// nearest point for `count` states X (count x n, LDS or global) -> idx (LDS/global int array).
// All waves of the workgroup participate; ends with __syncthreads().
// Position-weighted tables of at most 64 points and 32 coordinates (every shipped model: w_v = 0, P = 64, r = 30 / 36 is the general
// path): lane i keeps ITS point in registers for all the states of the call, the coordinates of a state arrive with one coalesced load
// (requested one state ahead) and are broadcast lane by lane -- nearest_wave pays four L2 round trips, a library sqrt and eighteen
// ds_bpermute per state (6-8 k clocks; 60 k per SCP iteration at C2).  Same sums in the same order, same first minimum.
template <typename XP, typename IP>
__device__ inline void nearest_many(const TpwlDev &T, XP X, int ldx, int count, IP idx) {
    const int wave = __builtin_amdgcn_readfirstlane(SRH_TID >> 6), nw = blockDim.x >> 6, lane = SRH_TID & 63;
    if (T.w_v == 0.0 && T.P <= 64 && T.r <= 32 && count >= 2 * nw) {            // (fewer states than two rounds: the table load does not pay)
        const int r = T.r;
        const bool live = lane < T.P;
        const int pl = live ? lane : T.P - 1, xl = lane < r ? lane : r - 1;
        double tq[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) tq[j] = T.qT[(j < r ? j : r - 1) * T.P + pl];     // unconditional (clamped) loads: all in flight at once
        int k = wave;
        double xv = (double)X[(size_t)(k < count ? k : count - 1) * ldx + r + xl];
        while (k < count) {
            const int kn = k + nw;
            const double xn = (double)X[(size_t)(kn < count ? kn : count - 1) * ldx + r + xl];             // the next state, while this one is worked on
            double sq = 0.0;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (j < r) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(xv), j), hi = __builtin_amdgcn_readlane(__double2hiint(xv), j);
                    const double e = tq[j] - __hiloint2double(hi, lo);
                    sq = fma(e, e, sq);
                }
            }
            const double dist = live ? T.w_q * sqrt(sq) : INFINITY;
            const double dmin = wg::wave_min(dist);
            const int imin = (int)wg::wave_min(dist == dmin ? (double)lane : 1e9);        // first minimum (np.argmin)
            if (lane == 0) idx[k] = imin < T.P ? imin : 0;          // (a state that is not a number: no minimum)
            k = kn;
            xv = xn;
        }
    } else {
        for (int k = wave; k < count; k += nw) {
            const int i = nearest_wave(T, X + (size_t)k * ldx);
            if (lane == 0) idx[k] = i;
        }
    }
    __syncthreads();
}

}  // namespace tpwl
