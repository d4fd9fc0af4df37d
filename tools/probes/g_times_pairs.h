// NOT part of the library: the "balanced" form of the product y = G u of round 4 (measured, not adopted -- DESIGN.md section 12):
// column i and column NP - 1 - i together always have n_u (N + 1) entries; a pair gets 2 n_u lanes (input b, stage parity h).
// Correct, and no faster than ql::g_times (8.3 k against 7.9 k clocks at C2, 17.0 k against 15.7 k at C5): the products are not
// bound by the imbalance of the columns but by the instructions per stage -- what the fixed-layout form (g_times_fixed) removes.
#pragma once
namespace ql {

// The same product with the work BALANCED over the lanes.  Column i of G has n_u (i / 2 + 1) entries: in g_times() the lanes
// of the last columns walk all N stages while the first columns' lanes finish at once, and the four input classes of a
// column meet through an LDS round trip.  Column i and column NP - 1 - i TOGETHER always have n_u (N + 1) entries: a pair
// gets 2 n_u lanes -- lane = (input b, stage parity h) -- every lane walks the stages j = h, h + 2, ... of BOTH columns in
// one loop (the two packed addresses differ by a constant, the u values are shared), ~N / 2 stages instead of N, and the
// 2 n_u partial sums of a column meet in DPP adds.  Trips of 8 stages; the masks are compiled out of the trips in which
// every lane of the wave is inside its column (long column) / no lane is (short column).
template <int MSEL, bool HALF, class GP>
__device__ __forceinline__ void g_times_pairs(const QPDims &d, const GP &g, Lds &L, clptr uv, lptr yv, Waves<HALF> &W) {
    constexpr int M = MSEL, LP = 2 * M, CH = 8, WPAIRS = 64 / LP;
    const int ldG = 16 * d.KT, NP = g.NP, N = d.N, tid = W.tid, nt = W.nt;
    const int npairs = NP >> 1, PP = nt / LP;
    const int gl = tid % LP, b = gl % M, h = gl / M;
    const int goff0 = goff(g.j0, M, NP), nm1 = N * M - 1;
    for (int q0 = 0; q0 < npairs; q0 += PP) {
        const int q = q0 + tid / LP;
        const bool live = q < npairs;
        const int qc = live ? q : npairs - 1;
        const int chi = NP - 1 - qc, clo = qc, dc = chi - clo;
        const int jhi = live ? chi >> 1 : -1, jlo = live ? clo >> 1 : -1;          // this lane's last stage of either column
        // wave-uniform: the pairs of this wave are wq .. wq + WPAIRS - 1 (clipped)
        const int wq = __builtin_amdgcn_readfirstlane(q0 + (tid & ~63) / LP);
        const int wql = min(wq + WPAIRS - 1, npairs - 1);
        const int jhi_any = wq < npairs ? (NP - 1 - wq) >> 1 : -1;                 // some lane needs stages up to here (long column)
        const int jhi_all = wq + WPAIRS - 1 < npairs ? (NP - 1 - wql) >> 1 : -1;   // every lane needs stages up to here
        const int jlo_any = wq < npairs ? wql >> 1 : -1;                           // (short column)
        double ahi[2] = {0.0, 0.0}, alo[2] = {0.0, 0.0};
        // stages js, js + 2, ... (js of parity h) up to jend (inclusive, wave-uniform bound) from `src`
        auto stream = [&](auto src, int jfirst, int jend) {
            int js = jfirst + ((jfirst ^ h) & 1);                                  // first stage >= jfirst of this lane's parity
            int R = goff(js, M, NP) + b * (NP - 2 * js) + chi - 2 * js;
            int dl = M * (2 * NP - 4 * js - 2) - 4 * b - 4;
            int ui = js * M + b;
            auto trip = [&](auto HIM, auto LO) {
                constexpr bool himask = decltype(HIM)::value, lo = decltype(LO)::value;
                double gh[CH], gq[CH], uu[CH];
#pragma unroll
                for (int t = 0; t < CH; ++t) {
                    gh[t] = src[R];
                    if constexpr (lo) gq[t] = src[R - dc];
                    uu[t] = uv[min(ui, nm1)];
                    R += dl; dl -= 8 * M; ui += 2 * M;
                }
#pragma unroll
                for (int t = 0; t < CH; ++t) {
                    const int j = js + 2 * t;
                    if constexpr (himask) ahi[t & 1] = fma((j <= jhi && j <= jend) ? gh[t] : 0.0, uu[t], ahi[t & 1]);
                    else ahi[t & 1] = fma(gh[t], uu[t], ahi[t & 1]);
                    if constexpr (lo) alo[t & 1] = fma((j <= jlo && j <= jend) ? gq[t] : 0.0, uu[t], alo[t & 1]);
                }
                js += 2 * CH;
            };
            for (int jt = jfirst; jt <= jend; jt += 2 * CH) {
                // the trip covers the stages jt .. jt + 2 CH of either parity
                const bool full = jt + 2 * CH <= min(jhi_all, jend);               // uniform: no lane leaves its long column
                const bool lo = jt <= jlo_any;                                     // uniform: some lane is still in its short column
                if (full) { if (lo) trip(std::false_type{}, std::true_type{}); else trip(std::false_type{}, std::false_type{}); }
                else { if (lo) trip(std::true_type{}, std::true_type{}); else trip(std::true_type{}, std::false_type{}); }
            }
        };
        if (g.j0 > 0 && jhi_any >= 0) stream(g.gh, 0, min(g.j0 - 1, jhi_any));
        if (jhi_any >= g.j0) stream(g.gt - goff0, g.j0, jhi_any);
        const double shi = wg::group_sum<LP>(ahi[0] + ahi[1]), slo = wg::group_sum<LP>(alo[0] + alo[1]);
        if (live && gl == 0) { yv[chi] = shi; yv[clo] = slo; }
    }
    for (int e = NP + tid; e < ldG; e += nt) yv[e] = 0.0;
    W.sync();
}


}  // namespace ql
