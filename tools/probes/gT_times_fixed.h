// NOT part of the library: the fixed-layout form of out = G^T y of round 4 (measured, not adopted -- DESIGN.md section 12).
// Alone on a workgroup it is the faster product (C2: 5.4 k -> 3.0 k clocks, two right-hand sides 7.3 k -> 4.5 k; C5: 9.4 k ->
// 5.4 k, tools/probes/lean_probe.hip); inlined four times into the SCP kernel (whole workgroup / half set x one / two right-hand
// sides) it costs the kernel 120 B more scratch per lane and 18 KB of code, and the kernel as a whole gets SLOWER: C2 140.0 k ->
// 122.8 k SCP iterations/s, one rollout alone 1.70 -> 1.78 ms per SCP iteration; requesting the loads of a row 4 or 7 at a
// time instead of all 13 changes nothing (124.5 k / 125.7 k).  The fixed-layout form of y = G u (g_times_fixed) has no such
// cost and stays in the library.
#pragma once
namespace ql {

// out = G^T y for a fixed-layout instantiation (compile-time horizon NF and first resident stage J0F): the passes over the rows
// and the trips along a row are unrolled completely -- every load of a pass is an immediate offset from the lane's row start,
// all of them requested before the first FMA (gT_times below walks the same rows with run-time trip counts, four loads at a time).
// C2: 5.4 k -> 3.0 k clocks (two right-hand sides: 7.3 k -> 4.5 k), C5: 9.4 k -> 5.4 k (tools/probes/lean_probe.hip).
template <int MSEL, bool HALF, int NF, int J0F, bool TWO, class GP>
__device__ __forceinline__ void gT_times_fixed(const GP &g, clptr y1, clptr y2, lptr out1, lptr out2, Waves<HALF> &W) {
    constexpr int M = MSEL, NP = 2 * NF, NM = NF * M, RH = (J0F < NF ? J0F : NF) * M;
    constexpr int goff0 = M * (J0F * NP - J0F * (J0F - 1));
    const int tid = W.tid, g8 = tid & 7;
    constexpr int RPP = (HALF ? 256 : 512) / 8;
    // rows [RB, RB + RPP) clipped to RE, from `src` (indexed by the global packed offset minus SHIFT)
    auto pass = [&](auto src, auto RB_, auto RE_, auto SHIFT_) {
        constexpr int RB = decltype(RB_)::value, RE = decltype(RE_)::value, SHIFT = decltype(SHIFT_)::value;
        constexpr int NQ = (NP - 2 * (RB / M) + 7) >> 3;               // trips of the longest row of the pass
        const int r = RB + (tid >> 3), rc = r < RE ? r : RE - 1;
        const int j = rc / M, b = rc - j * M, len = NP - 2 * j;
        auto pg = src + (goff(j, M, NP) + b * len + g8 - SHIFT);
        clptr p1 = y1 + 2 * j + g8, p2 = (TWO ? y2 : y1) + 2 * j + g8;
        constexpr int CQ = NQ;                 // loads requested together (registers: 2-3 doubles each)
        double a1 = 0.0, c1 = 0.0, a2 = 0.0, c2 = 0.0;
        srh_static_for<0, (NQ + CQ - 1) / CQ>([&](auto C_) {
            constexpr int Q0 = decltype(C_)::value * CQ, Q1 = Q0 + CQ < NQ ? Q0 + CQ : NQ;
            double gv[CQ], ya[CQ], yb[TWO ? CQ : 1];
#pragma unroll
            for (int q = Q0; q < Q1; ++q) { gv[q - Q0] = pg[8 * q]; ya[q - Q0] = p1[8 * q]; if constexpr (TWO) yb[q - Q0] = p2[8 * q]; }
#pragma unroll
            for (int q = Q0; q < Q1; ++q) {
                if (q & 1) c1 = fma(gv[q - Q0], ya[q - Q0], c1); else a1 = fma(gv[q - Q0], ya[q - Q0], a1);
                if constexpr (TWO) { if (q & 1) c2 = fma(gv[q - Q0], yb[q - Q0], c2); else a2 = fma(gv[q - Q0], yb[q - Q0], a2); }
            }
        });
        a1 = wg::group_sum<8>(a1 + c1);
        if constexpr (TWO) a2 = wg::group_sum<8>(a2 + c2);
        if (g8 == 0 && r < RE) { out1[r] = a1; if constexpr (TWO) out2[r] = a2; }
    };
    if constexpr (RH > 0)
        srh_static_for<0, (RH + RPP - 1) / RPP>([&](auto K_) {
            pass(g.gh, std::integral_constant<int, decltype(K_)::value * RPP>{}, std::integral_constant<int, RH>{}, std::integral_constant<int, 0>{});
        });
    if constexpr (NM > RH)
        srh_static_for<0, (NM - RH + RPP - 1) / RPP>([&](auto K_) {
            pass(g.gt, std::integral_constant<int, RH + decltype(K_)::value * RPP>{}, std::integral_constant<int, NM>{}, std::integral_constant<int, goff0>{});
        });
    W.sync();
}


// the dispatch gT_times would use
template <int MSEL, bool HALF, class GP>
__device__ __forceinline__ void gT_times_fixed_or_not(const QPDims &d, const GP &g, Lds &L, clptr y1, clptr y2, lptr out1, lptr out2, Waves<HALF> &W) {
    if constexpr (GP::NF > 0) {
        if (y2) gT_times_fixed<MSEL, HALF, GP::NF, GP::J0F, true>(g, y1, y2, out1, out2, W);
        else gT_times_fixed<MSEL, HALF, GP::NF, GP::J0F, false>(g, y1, y2, out1, out2, W);
    } else {
        gT_times<MSEL, HALF>(d, g, L, y1, y2, out1, out2, W);
    }
}
}  // namespace ql
