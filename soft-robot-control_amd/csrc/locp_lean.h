// Lean condensed interior point for the LOCP QP (sofacontrol/scp/locp.py:218-342) -- the kernel the SCP loop spends its
// time in, as its own code path: no stage-wise Riccati solver inlined next to it (the fused kernel of locp_dev.h +
// locp_cond.h sits at 256 VGPRs with ~1.5 KB of scratch per lane for that reason), and the block lower-triangular
//     G[k][j] = C_o Phi(k, j+1) B_j          (N p_o x N m; locp_cond.h has the mathematics)
// RESIDENT IN LDS in packed form instead of streaming from L2 four to six times per Newton system.
//
// Packed G^T: row (j, b) (input b of stage j) holds only its structurally non-zero columns i = p_o j .. N p_o - 1, rows in
// the order (j, b); p_o = 2:  off(j) = m (j NP - j (j - 1)) doubles, element (j, b, i) at off(j) + b (NP - 2 j) + i - 2 j.
// The rows of the stages j >= j0 live in LDS (`Gt`), the first j0 stages -- the longest rows -- in the problem's L2 block
// (`gh`); j0 is the smallest value for which the carve fits into the 160 KB of a CU (7 at the Diamond shape: 61 of 82 KB
// in LDS; the Trunk's eight inputs leave room for the last 27 of 50 stages).
//
// What changes against locp_cond.h, phase by phase (same iteration, same iterates up to rounding):
//   condensation   writes G packed (LDS / L2 head) -- no zero fill of the structurally empty part
//   Gram matrix    K = I + Ls^T (G D^-1 G^T) Ls: MFMA operands straight from the packed rows (weights applied to the B
//                  operand in registers), causal k-ranges (a tile row stops at the last stage that reaches it), tiles
//                  distributed by a host-built schedule; Ls and the Jacobi scaling applied to the accumulators
//   G / G^T        products from LDS (+ the L2 head)
// Requirements (checked on the host: QPDims::lean): the condensed path applies, p_o = 2, diagonal input Hessians
// (diagD), n_u a multiple of 4.  Everything else stays on the fused kernel.
#pragma once
#include "locp_dev.h"

// compile-time loop: f(integral_constant<int, I>) for I = B .. E - 1
template <int B, int E, class F>
__device__ __forceinline__ void srh_static_for(F &&f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); srh_static_for<B + 1, E>(f); }
}

namespace ql {

using qpc::TS;
using qpc::TSZ;
using qpc::QR;
using qpc::Prof;

struct Lds : qpc::Lds {
    lptr Gt;           // packed G^T rows of the stages j >= j0
    lptr panel;        // [A | B] panel while condensing (aliases Rinv / the u-space temporaries)
    // half-size layout (QPDims::lean_half): the free response yf and the region sequence of the condensation live in the problem's L2
    // block behind the packed G (read once per QP), Rinv == B marks "inverse of a diagonal tile in the tile's own place" (rinv_tile)
    gptr yfg;
    giptr goffg;
    int ypad;          // zeros behind ya / yd / yg (gT_times): 48, or 24 when a pass has 32 rows (256 threads)
    int ldTc;          // row pitch of Theta^T while condensing: 16 KT + 1, or the widest column pass + 1
};
// inverse of the J-th diagonal tile of the factor
__device__ __forceinline__ lptr rinv_tile(const Lds &L, int J, int KT) {
    return L.Rinv == L.B ? L.B + (size_t)qpc::tile_index(J, J, KT) * TSZ : L.Rinv + (size_t)J * TSZ;
}

__host__ __device__ inline int goff(int j, int m, int NP) { return m * (j * NP - j * (j - 1)); }     // p_o = 2
constexpr int YPAD = 48;               // zeros behind the y-space vectors that feed gT_times (see there)
// condense(): item slots per wave.  A stage has at most KT * (NPa / 16) MFMA items spread over the 8 waves; the host only
// enables the lean path when they fit (scp_host.h:build_consts) -- an item past the last slot would silently never be written.
constexpr int CONDENSE_SLOTS = 5;
constexpr int GRAM_TASKS = 32;         // entries of QPConst::gram_sched ({I, J0, nJ, 0}, longest first, nJ = 0 behind the last one)
__host__ __device__ inline bool condense_fits(const QPDims &d, int nwaves) { return d.KT * (d.NPa / 16) <= nwaves * CONDENSE_SLOTS; }

struct Sizes { size_t regX, thetaT, tiles, rinv, nm4, gt, ldi, ls, ldG, ua, tx, ld, idx, ypad, ldTc, nyv, nvv, nidx; };
// first column tile of the second condensation pass of the half-size layout (pass A: tiles [T, KT), pass B: [0, T))
__host__ __device__ inline int half_split_tile(int KT) { return KT / 2; }
__host__ __device__ inline Sizes sizes(const QPDims &d, int nthreads, int j0) {
    Sizes s;
    const bool half = d.lean_half != 0;
    const size_t nk = (size_t)d.NK, nm = (size_t)d.N * d.m;
    s.ldG = 16 * (size_t)d.KT;
    s.nm4 = (nm + 3) & ~(size_t)3;
    s.ldTc = half ? 16 * (size_t)(d.KT - half_split_tile(d.KT)) + 1 : s.ldG + 1;
    s.thetaT = nk * s.ldTc;
    s.tiles = (size_t)d.KT * (d.KT + 1) / 2 * TSZ;
    s.rinv = half ? 0 : (size_t)d.KT * TSZ;
    const size_t a = s.thetaT + nk * d.ld, b = s.tiles + s.rinv + 3 * s.nm4 + (size_t)nthreads;
    s.regX = ((a > b ? a : b) + 3) & ~(size_t)3;
    const int NP = d.N * d.po;
    s.gt = ((size_t)(goff(d.N, d.m, NP) - goff(j0, d.m, NP)) + 3) & ~(size_t)3;
    s.ldi = s.nm4;
    s.ls = (size_t)d.N * d.po * d.po;
    s.ua = half ? 0 : ((size_t)d.nU * d.m + 3) & ~(size_t)3;             // (ipm_box reads its row coefficients from global memory)
    s.tx = half ? 0 : ((size_t)(d.nX + d.nXf) * d.po + 3) & ~(size_t)3;
    s.ld = ((size_t)d.ld + 3) & ~(size_t)3;
    s.idx = ((size_t)(d.N / 2 + 2) + 3) & ~(size_t)3;
    s.ypad = half ? 24 : 48;
    s.nyv = half ? 7 : 9;                   // y-space vectors of their own (half: yf in L2, yb shares yd's place)
    s.nvv = half ? 0 : 2;                   // rollout vectors of their own (half: in du's place, dead while a rollout runs)
    s.nidx = half ? 1 : 2;                  // int arrays of their own (half: goff in L2)
    return s;
}
__host__ __device__ inline size_t lds_doubles(const QPDims &d, int nthreads, int j0) {
    const Sizes s = sizes(d, nthreads, j0);
    return s.regX + s.gt + s.ldi + s.ls + 2 * s.nm4 + s.nyv * s.ldG + 3 * s.ypad + s.ua + s.tx + s.nvv * s.ld + 16 + 16 + 4 + s.nidx * s.idx;
}
// doubles of the half-size layout's homes in the L2 block behind the packed G: [pad (YPAD) | yf (16 KT) | goff (N ints)]
__host__ __device__ inline size_t half_l2_off(const QPDims &d) { return (size_t)goff(d.N, d.m, d.N * d.po) + 48; }
__device__ inline void lds_carve(Lds &L, lptr base, const QPDims &d, int nthreads, gptr work_base = nullptr) {
    const Sizes s = sizes(d, nthreads, d.lean_j0);
    const bool half = d.lean_half != 0;
    lptr p = base;
    auto take = [&](size_t c) { lptr q = p; p += c; return q; };
    lptr X = take(s.regX);
    // interior-point view: K tiles | inverses of the diagonal tiles | u-space temporaries | reduction scratch
    L.B = X; L.Rinv = half ? X : X + s.tiles;
    L.ta = X + s.tiles + s.rinv; L.tb = L.ta + s.nm4; L.tc = L.tb + s.nm4; L.part = L.tc + s.nm4;
    // condensation view: Theta^T (in L.B) | [A | B] panel
    L.panel = X + s.thetaT;
    L.A = L.panel;
    L.Gt = take(s.gt);
    L.Ldi = take(s.ldi); L.Ls = take(s.ls);
    L.u = take(s.nm4); L.du = take(s.nm4);
    L.y = take(s.ldG); L.dy = take(s.ldG);
    L.yf = half ? (lptr) nullptr : take(s.ldG);
    // ya, yd, yg feed gT_times: `ypad` zeros behind each (written once by ipm, never touched again)
    L.ya = take(s.ldG + s.ypad); L.yd = take(s.ldG + s.ypad); L.yg = take(s.ldG + s.ypad);
    L.yb = half ? L.yd : take(s.ldG);          // (yb = G t is dead once yc = ks Ls^T yb is formed; yd is written behind the K solve)
    L.yc = take(s.ldG); L.ks = take(s.ldG);
    L.UA = take(s.ua); L.Tx = take(s.tx);
    if (half) { L.v1 = L.du; L.v2 = L.du + s.ld; } else { L.v1 = take(s.ld); L.v2 = take(s.ld); }
    L.Qu = take(16); L.red = take(16);
    L.flag = (liptr)take(4);
    L.idxl = (liptr)take(s.idx);
    L.goff = half ? (liptr) nullptr : (liptr)take(s.idx);
    L.ypad = (int)s.ypad;
    L.ldTc = (int)s.ldTc;
    L.yfg = nullptr; L.goffg = nullptr;
    if (half && work_base != nullptr) {
        gptr hb = work_base + d.qc_off + half_l2_off(d);
        L.yfg = hb;
        L.goffg = (giptr)(hb + s.ldG);
    }
}

// packed G^T: head rows in the problem's L2 block, the rest in LDS
// NF_ > 0: the horizon and the first LDS-resident stage are compile-time constants of the kernel instantiation (the fixed
// layouts of lean.hip) -- the products can then address every stage with immediate offsets (g_times_fixed)
template <int NF_ = 0, int J0F_ = 0>
struct GPackT {
    cgptr gh;          // rows of the stages j < j0
    clptr gt;          // rows of the stages j >= j0, offset so that gt[goff(j) + ...] addresses stage j
    int j0, m, NP;
    static constexpr int NF = NF_, J0F = J0F_;
};
using GPack = GPackT<0, 0>;

// ------------------------------------------------------------------ wave sets
// The factorisation of K is a chain of seven one-wave 16 x 16 factorisations with short all-wave phases in between (35 k of
// its 52 k clocks at C2 belong to wave 0 alone), and the first half of the Newton solve that follows -- G^T g_y, the dual
// residual, D^-1, G t, Ls^T -- does not depend on the factor.  ipm_box() therefore splits the workgroup for that stretch:
// one half of the waves factors, the other runs the products, and they meet again at an s_barrier.  The halves are taken BY
// SIMD (a workgroup's waves w and w + 4 share one: set 0 = waves 0, 1, 4, 5, set 1 = waves 2, 3, 6, 7), so that the wave
// with the one-wave factorisations shares its SIMD with a wave of its own set, which waits most of the time, and not with
// a wave that streams the products.  Inside the stretch a set cannot use
// s_barrier (it counts every wave of the workgroup): a set synchronises on its own monotone arrival counter in LDS -- each
// wave adds one and waits for the next multiple of the set size (the LDS pipe is in order per wave, so a wave's earlier
// LDS writes have landed when its arrival is seen; the workgroup-scope fences keep the compiler from moving accesses
// across and drain the global-memory counters).  HALF = false: the whole workgroup, plain s_barrier.
template <bool HALF>
struct Waves {
    int tid, nt, wave, nw;             // thread / wave index inside the set, set sizes
    liptr ctr;                         // HALF: arrival counter of this set (zeroed by the caller before the split)
    int target;
    bool sleepy;                       // HALF: s_sleep between polls (the set whose waits are long; the other polls back to back)
    __device__ __forceinline__ void sync() {
        if constexpr (!HALF) {
            __syncthreads();
        } else {
            target += nw;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if ((SRH_TID & 63) == 0) {
                __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) { if (sleepy) __builtin_amdgcn_s_sleep(1); }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
};
__device__ __forceinline__ Waves<false> all_waves() {
    return Waves<false>{(int)SRH_TID, (int)blockDim.x, __builtin_amdgcn_readfirstlane((int)SRH_TID >> 6), (int)blockDim.x >> 6, (liptr) nullptr, 0, false};
}
// the two halves of an 8-wave workgroup by SIMD: set (w >> 1) & 1, wave (w & 1) + 2 (w >> 2) inside it; ctr0: the arrival counter
// of set 1 (set 0 synchronises through the counters of tile_cholesky_set)
// (a 4-wave workgroup -- the half-size layout, one wave per SIMD -- splits 2 + 2: waves 0, 1 and waves 2, 3)
// The one-wave phases of a 4-wave workgroup (tile factorisations, block substitutions) run on its SERIAL WAVE: wave 0, or wave 2 when the
// workgroup finds its wave 0 in the second wave slot of its SIMD -- i.e. when another workgroup got to this CU first (two half-size
// workgroups share a CU, one wave of each per SIMD: with both serial chains on wave 0 they would share ONE SIMD's issue slots while three
// SIMDs idle).  serial_wave_pick() is called once per kernel by every wave; the choice is kept in L.Qu[15].
__device__ __forceinline__ void serial_wave_pick(Lds &L, bool enable = true) {
    if (SRH_TID == 0) {
        const unsigned slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID [3:0]: wave slot inside the SIMD
        L.Qu[15] = (enable && blockDim.x == 256 && (slot & 1u)) ? 2.0 : 0.0;
    }
}
__device__ __forceinline__ int serial_wave(const Lds &L) {
    return blockDim.x == 256 ? __builtin_amdgcn_readfirstlane((int)L.Qu[15]) : 0;
}
__device__ __forceinline__ int half_of_wave(int w) { return (w >> 1) & 1; }
__device__ __forceinline__ Waves<true> half_waves(liptr ctr0, int sw = 0) {
    const int w = __builtin_amdgcn_readfirstlane((int)SRH_TID >> 6) ^ sw, which = half_of_wave(w);
    const bool four = blockDim.x == 256;
    const int lw = four ? (w & 1) : (w & 1) + 2 * (w >> 2), sz = four ? 2 : 4;
    return Waves<true>{lw * 64 + ((int)SRH_TID & 63), sz * 64, lw, sz, ctr0, 0, which == 1};      // (only set 1 calls sync(): one counter)
}

// ------------------------------------------------------------------ rollout x_{k+1} = A_k x_k + B_k u_k + d_k  (u null: zero inputs)
// One dot product of length n + m per state row through the [A | B] panel in LDS (reloaded only when the TPWL region
// changes: ~12 % of the stages), ONE barrier per stage: 8 lanes per row, lane (il, s) of a 16-lane DPP row takes the columns
// 2 s + h + 16 q of row 2 r + il (bank = (row + column) mod 16: conflict free), three DPP additions finish the row.
// The stage vector [x_k ; u_k ; 0] is double buffered in v1 / v2.  (qp::rollout: two L2-fed products and four barriers
// per stage, 7.6 k clocks per stage against ~0.8 k.)
// XS: the trajectory is ALSO written to xs (LDS: the start of the Theta^T area, free during a rollout) for the tests that follow it --
// objective, trust region, and the SCP loop's tests in the GuSTO kernel.  (A flag, not a null test: the area starts at LDS offset 0,
// which is what a null pointer of address space 3 compares equal to.)
template <int MSEL, int NSEL, bool XS = false, int WG = 8>
__device__ __forceinline__ void rollout(const QPDims &d, const QPDyn &dyn, cgptr x0, cgptr u, gptr x, Lds &L, lptr xs = nullptr) {
    const int N = d.N, n = d.n, m = d.m, ld = d.ld, NPa = d.NPa, nk = d.NK;
    const int tid = SRH_TID, nt = blockDim.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l = lane & 15, il = l >> 3, sl = l & 7;
    const int i_lo = 8 * wave + 2 * (lane >> 4) + il;                 // + 8 rows per wave per row block (n_x > 64, or a 4-wave workgroup: more passes)
    constexpr int rows_pass = 8 * WG;
    constexpr bool ONE_BLOCK = NSEL > 0 && NSEL <= rows_pass;
    const int vlen = (int)(((size_t)ld + 3) & ~(size_t)3);
    lptr va = L.v1, vb = L.v2;
    for (int e = tid; e < nk * ld; e += nt) L.panel[e] = 0.0;
    for (int e = tid; e < vlen; e += nt) {
        const double v = e < n ? x0[e] : (e < n + m && u ? u[e - n] : 0.0);
        va[e] = v; vb[e] = 0.0;
        if (e < n) { x[e] = v; if constexpr (XS) xs[e] = v; }
    }
    __syncthreads();
    QPLds P{};
    P.AB = L.panel; P.idxl = L.idxl; P.psel = -1;
    if (qp::panel_load(d, dyn, P, 0)) __syncthreads();
    const int nq = NPa >> 4;
    for (int k = 0; k < N; ++k) {
        const int sel = __builtin_amdgcn_readfirstlane(L.idxl[k]);
        const double un = (tid < m && u && k + 1 < N) ? u[(size_t)(k + 1) * m + tid] : 0.0;
        for (int ib = 0; ib < (ONE_BLOCK ? 1 : n); ib += rows_pass) {
            const int i = i_lo + ib, ic = i < n ? i : n - 1;
            const double dk = dyn.d[(size_t)sel * n + ic];
            clptr row = L.panel + ic * ld + 2 * sl;
            double acc = 0.0;
            for (int q = 0; q < nq; ++q) {
                acc = fma(row[16 * q], va[2 * sl + 16 * q], acc);
                acc = fma(row[16 * q + 1], va[2 * sl + 16 * q + 1], acc);
            }
            acc = wg::group_sum<8>(acc) + dk;
            if (sl == 0 && i < n) { vb[i] = acc; x[(size_t)(k + 1) * n + i] = acc; if constexpr (XS) xs[(size_t)(k + 1) * n + i] = acc; }
        }
        if (tid < m) vb[n + tid] = un;
        __syncthreads();
        if (k + 1 < N) {
            const int nsel = __builtin_amdgcn_readfirstlane(L.idxl[k + 1]);
            if (dyn.idx == nullptr || nsel != sel) { if (qp::panel_load(d, dyn, P, k + 1)) __syncthreads(); }
        }
        lptr t = va; va = vb; vb = t;
    }
}

// ------------------------------------------------------------------ condensation (once per QP)
// As qpc::condense (adjoint recursion Theta_{j-1} = [C_o ; Theta_j A_j], G[:, j] = Theta_j B_j, one MFMA product per
// stage with the [A_j | B_j] panel in LDS); G goes to the packed store.
template <int MSEL, int NSEL>
__device__ __forceinline__ void condense(const QPDims &d, const QPConst &c, const QPDyn &dyn, cgptr x, gptr gh, Lds &L) {
    const int N = d.N, n = d.n, m = d.m, po = d.po, ld = d.ld, KT = d.KT, ldG = 16 * d.KT, ldT = ldG + 1;
    const int nk = d.NK, NPa = d.NPa, NP = N * po, j0 = d.lean_j0;
    const int tid = SRH_TID, nt = blockDim.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    for (int e = tid; e < ldG; e += nt) {
        double v = 0.0;
        if (e < NP) {
            const int k = e / po + 1, a = e - (k - 1) * po;
            for (int j = 0; j < n; ++j) v = fma(c.Co[(size_t)a * n + j], x[(size_t)k * n + j], v);
        }
        L.yf[e] = v;
    }
    for (int e = tid; e < nk * ldT; e += nt) L.B[e] = 0.0;
    for (int e = tid; e < nk * ld; e += nt) L.panel[e] = 0.0;
    // gT_times runs its lanes past the end of a row (times zeros of y): finite values behind the last L2-resident row
    for (int e = tid; e < YPAD; e += nt) gh[goff(j0, m, NP) + e] = 0.0;
    __syncthreads();
    QPLds P{};
    P.AB = L.panel; P.idxl = L.idxl; P.psel = -1;
    const int MT = NPa >> 4;
    const int KS = (n + 3) >> 2;
    const int goff0 = goff(j0, m, NP);
    // Work items of a stage = (column tile ti of Theta^T that already has columns, row tile ci of [A | B]^T): spread over
    // ALL waves (a wave per column tile left most SIMDs idle late in the recursion, when few output stages are "born").
    // Phase A: every wave multiplies its items (operands from LDS, results in registers); barrier; phase B: results into
    // Theta^T / the packed G, the C_o columns of the next output stage, the panel of the next stage if its region differs.
    (void)qp::panel_load(d, dyn, P, N - 1);
    for (int e = tid; e < po * n; e += nt) { const int a = e / n, r = e - a * n; L.B[r * ldT + (N - 1) * po + a] = c.Co[(size_t)a * n + r]; }
    __syncthreads();
    constexpr int KSMAX = NSEL > 0 ? (NSEL + 3) / 4 : 32;
    for (int j = N - 1; j >= 0; --j) {
        const int t_first = (j * po) >> 4;
        const int count = (KT - t_first) * MT;
        const int len = NP - po * j, gj = goff(j, m, NP);
        constexpr int RMAX = CONDENSE_SLOTS;             // item slots per wave: KT * MT <= 8 * 5 items over 8 waves (condense_fits)
        wg::qp_d4 acc[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            acc[r] = {0.0, 0.0, 0.0, 0.0};
            const int it = wave + nw * r;
            if (it < count) {
                const int ti = t_first + it / MT, ci = it - (it / MT) * MT;
                double aop[KSMAX], bop[KSMAX];
#pragma unroll
                for (int s = 0; s < KSMAX; ++s) {
                    aop[s] = s < KS ? L.panel[(4 * s + kk) * ld + 16 * ci + l16] : 0.0;
                    bop[s] = s < KS ? L.B[(4 * s + kk) * ldT + 16 * ti + l16] : 0.0;
                }
#pragma unroll
                for (int s = 0; s < KSMAX; ++s)
                    if (s < KS) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[s], bop[s], acc[r], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int it = wave + nw * r;
            if (it < count) {
                const int ti = t_first + it / MT, ci = it - (it / MT) * MT;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 16 * ci + kk + 4 * q, i = 16 * ti + l16;
                    if (i >= po * j && i < NP) {                 // columns of the output stages k > j ("born")
                        if (row < n) {
                            L.B[row * ldT + i] = acc[r][q];
                        } else if (row < n + m) {
                            const int at = gj + (row - n) * len + (i - po * j);
                            if (j < j0) gh[at] = acc[r][q]; else L.Gt[at - goff0] = acc[r][q];
                        }
                    }
                }
            }
        }
        if (j > 0) {
            for (int e = tid; e < po * n; e += nt) { const int a = e / n, r = e - a * n; L.B[r * ldT + (j - 1) * po + a] = c.Co[(size_t)a * n + r]; }
            (void)qp::panel_load(d, dyn, P, j - 1);
        }
        __syncthreads();
    }
}

// The condensation for the half-size layout (QPDims::lean_half): Theta^T has room for about half of its N p_o columns (Lds::ldTc), so the
// adjoint recursion runs twice -- pass A over the column tiles [T, KT) from stage N - 1 down, pass B over the tiles [0, T) from the last stage
// that still reaches them (columns < 16 T are "born" at the stages j <= (16 T - 1) / p_o) -- each with the products of condense() restricted
// to its tiles.  1.5 x the MFMA work of the one-pass form; G goes to the L2 block whole (lean_j0 = N), yf and the pad behind G as well.
template <int MSEL, int NSEL>
__device__ __forceinline__ void condense_half(const QPDims &d, const QPConst &c, const QPDyn &dyn, cgptr x, gptr gh, Lds &L) {
    const int N = d.N, n = d.n, m = d.m, po = d.po, ld = d.ld, KT = d.KT, ldG = 16 * d.KT, ldT = L.ldTc;
    const int nk = d.NK, NPa = d.NPa, NP = N * po;
    const int tid = SRH_TID, nt = blockDim.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, nw = nt >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    for (int e = tid; e < ldG; e += nt) {
        double v = 0.0;
        if (e < NP) {
            const int k = e / po + 1, a = e - (k - 1) * po;
            for (int j = 0; j < n; ++j) v = fma(c.Co[(size_t)a * n + j], x[(size_t)k * n + j], v);
        }
        L.yfg[e] = v;
    }
    for (int e = tid; e < nk * ld; e += nt) L.panel[e] = 0.0;
    // gT_times runs its lanes past the end of a row (times zeros of y): finite values behind the last row
    for (int e = tid; e < YPAD; e += nt) gh[goff(N, m, NP) + e] = 0.0;
    QPLds P{};
    P.AB = L.panel; P.idxl = L.idxl; P.psel = -1;
    const int MT = NPa >> 4;
    const int KS = (n + 3) >> 2;
    const int Tsplit = half_split_tile(KT);
    constexpr int KSMAX = NSEL > 0 ? (NSEL + 3) / 4 : 32;
    for (int pass = 0; pass < 2; ++pass) {
        const int T0 = pass == 0 ? Tsplit : 0, T1 = pass == 0 ? KT : Tsplit;
        if (T1 <= T0) continue;
        const int c0 = 16 * T0, c1 = min(16 * T1, NP);                  // this pass's columns [c0, c1)
        const int jstart = min(N - 1, (c1 - 1) / po);                   // the last stage that reaches them
        __syncthreads();
        for (int e = tid; e < nk * ldT; e += nt) L.B[e] = 0.0;
        __syncthreads();
        (void)qp::panel_load(d, dyn, P, jstart);
        for (int e = tid; e < po * n; e += nt) {
            const int a = e / n, r = e - a * n, i = jstart * po + a;
            if (i >= c0 && i < c1) L.B[r * ldT + (i - c0)] = c.Co[(size_t)a * n + r];
        }
        __syncthreads();
        for (int j = jstart; j >= 0; --j) {
            const int t_first = max(T0, (j * po) >> 4);
            const int count = (T1 - t_first) * MT;
            const int len = NP - po * j, gj = goff(j, m, NP);
            constexpr int RMAX = CONDENSE_SLOTS;             // item slots per wave: (T1 - T0) MT <= 4 * 5 items over 4 waves (scp_host.h)
            wg::qp_d4 acc[RMAX];
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                acc[r] = {0.0, 0.0, 0.0, 0.0};
                const int it = wave + nw * r;
                if (it < count) {
                    const int ti = t_first + it / MT, ci = it - (it / MT) * MT;
                    double aop[KSMAX], bop[KSMAX];
#pragma unroll
                    for (int s = 0; s < KSMAX; ++s) {
                        aop[s] = s < KS ? L.panel[(4 * s + kk) * ld + 16 * ci + l16] : 0.0;
                        bop[s] = s < KS ? L.B[(4 * s + kk) * ldT + 16 * (ti - T0) + l16] : 0.0;
                    }
#pragma unroll
                    for (int s = 0; s < KSMAX; ++s)
                        if (s < KS) acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[s], bop[s], acc[r], 0, 0, 0);
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const int it = wave + nw * r;
                if (it < count) {
                    const int ti = t_first + it / MT, ci = it - (it / MT) * MT;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 16 * ci + kk + 4 * q, i = 16 * ti + l16;
                        if (i >= po * j && i < c1) {                 // columns of the output stages k > j ("born") inside this pass
                            if (row < n) L.B[row * ldT + (i - c0)] = acc[r][q];
                            else if (row < n + m) gh[gj + (row - n) * len + (i - po * j)] = acc[r][q];
                        }
                    }
                }
            }
            if (j > 0) {
                for (int e = tid; e < po * n; e += nt) {
                    const int a = e / n, r = e - a * n, i = (j - 1) * po + a;
                    if (i >= c0 && i < c1) L.B[r * ldT + (i - c0)] = c.Co[(size_t)a * n + r];
                }
                (void)qp::panel_load(d, dyn, P, j - 1);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ products with G
// Row (j, b) of the packed G^T starts at goff(j) + b (NP - 2 j); with R(j, b) = that start - 2 j, element i sits at R + i and
//     R(j + 1, b) = R(j, b) + m (NP - 2 j) - 2 b - 2
// -- the loops below walk the stages with two integer additions instead of re-deriving the offsets.

// The same product for a kernel instantiation whose horizon NF and first LDS-resident stage J0F are compile-time constants
// (same thread mapping as g_times below: thread = (column, input class), partial sums of the classes through LDS).  The
// stage loop is unrolled completely: element (j, b, i) of the packed store sits at
//     goff(j) - 2 j  +  (b NP + i)  -  2 j b
// -- an immediate, a per-lane base, and a term that only needs one subtraction of the wave-uniform 2 b per stage -- so a stage
// costs an address subtraction, two LDS reads (G and u, the latter at an immediate offset from a per-wave base) and one FMA,
// plus a select in the triangular part; g_times spends about ten instructions per stage on the same (index arithmetic of
// the run-time loop, clamps and masks of its eight-stage trips).  Stages in blocks of sixteen; a block no lane of the wave
// needs is skipped, a block every lane needs whole runs without the selects.  C2: 7.9 k -> 5.5 k clocks, C5: 15.7 k -> 11.6 k.
template <int MSEL, bool HALF, int NF, int J0F, class GP>
__device__ __forceinline__ void g_times_fixed(const GP &g, Lds &L, clptr uv, lptr yv, int ldG, Waves<HALF> &W) {
    constexpr int M = MSEL, CW = 128, NP = 2 * NF;
    constexpr int BLK = 16;                 // stages per block: 16 + 16 loads in flight (8: 6.2 k clocks at C2, 16: 5.5 k)
    constexpr int goff0 = M * (J0F * NP - J0F * (J0F - 1));
    const int tid = W.tid, nt = W.nt;
    const int GR = nt / CW, col = tid % CW, grp = __builtin_amdgcn_readfirstlane(tid / CW);
    const int cc = col < NP ? col : NP - 1;
    const int jlast = col < NP ? cc >> 1 : -1;                        // last stage that reaches this lane's column
    const int c_lo = col & ~63, c_hi = min(NP - 1, col | 63);
    const int jall = __builtin_amdgcn_readfirstlane((col | 63) < NP ? c_lo >> 1 : -1);      // every lane of the wave: stages 0 .. jall
    const int jany = __builtin_amdgcn_readfirstlane(c_lo < NP ? c_hi >> 1 : -1);            // some lane of the wave: stages 0 .. jany
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = grp; b < M; b += GR) {                               // wave-uniform
        clptr pu = uv + b;
        // stages [JB, JE) from `pg` (pointing at element (0, b, column) of its store minus SHIFT), MASK: triangular part
        auto block = [&](auto pg, auto JB_, auto JE_, auto SHIFT_, auto MASK) {
            constexpr int JB = decltype(JB_)::value, JE = decltype(JE_)::value, SHIFT = decltype(SHIFT_)::value;
            constexpr bool masked = decltype(MASK)::value;
            double gv[JE - JB], uu[JE - JB];
#pragma unroll
            for (int j = JB; j < JE; ++j) {
                gv[j - JB] = pg[(M * (j * NP - j * (j - 1)) - 2 * j - SHIFT) - 2 * j * b];
                uu[j - JB] = pu[j * M];
            }
#pragma unroll
            for (int j = JB; j < JE; ++j) {
                if constexpr (masked) acc[j & 3] = fma(j <= jlast ? gv[j - JB] : 0.0, uu[j - JB], acc[j & 3]);
                else acc[j & 3] = fma(gv[j - JB], uu[j - JB], acc[j & 3]);
            }
        };
        auto span = [&](auto pg, auto LO_, auto HI_, auto SHIFT_) {   // compile-time range [LO, HI) in blocks of BLK
            constexpr int LO = decltype(LO_)::value, HI = decltype(HI_)::value;
            srh_static_for<0, (HI - LO + BLK - 1) / BLK>([&](auto K_) {
                constexpr int JB = LO + decltype(K_)::value * BLK, JE = JB + BLK < HI ? JB + BLK : HI;
                if (JB <= jany) {                                     // uniform
                    if (JE - 1 <= jall) block(pg, std::integral_constant<int, JB>{}, std::integral_constant<int, JE>{}, SHIFT_, std::false_type{});
                    else block(pg, std::integral_constant<int, JB>{}, std::integral_constant<int, JE>{}, SHIFT_, std::true_type{});
                }
            });
        };
        if constexpr (J0F > 0) span(g.gh + (b * NP + cc), std::integral_constant<int, 0>{}, std::integral_constant<int, (J0F < NF ? J0F : NF)>{}, std::integral_constant<int, 0>{});
        if constexpr (J0F < NF) span(g.gt + (b * NP + cc), std::integral_constant<int, J0F>{}, std::integral_constant<int, NF>{}, std::integral_constant<int, goff0>{});
    }
    lptr part = L.part;
    part[grp * CW + col] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    W.sync();
    if (tid < ldG) {
        double s = 0.0;
        if (tid < NP) for (int q = 0; q < GR; ++q) s += part[q * CW + tid];
        yv[tid] = s;
    }
    W.sync();
}

// yv[i] = sum_{rows (j,b), 2 j <= i} G^T[(j,b)][i] uv[(j,b)]: thread = (column, input class b mod 4).  Stages j <= jlo (the
// last stage that reaches EVERY column of the wave) need no mask; lanes read past "their" rows only inside LDS / the L2
// block (the select discards what they get).  8 loads of G and of u per trip before the FMAs.
template <int MSEL, bool HALF, class GP>
__device__ __forceinline__ void g_times(const QPDims &d, const GP &g, Lds &L, clptr uv, lptr yv, Waves<HALF> &W) {
    if constexpr (GP::NF > 0) { g_times_fixed<MSEL, HALF, GP::NF, GP::J0F>(g, L, uv, yv, 16 * d.KT, W); return; }
    constexpr int M = MSEL, CH = 8, CW = 128;
    const int ldG = 16 * d.KT, NP = g.NP, N = d.N, tid = W.tid, nt = W.nt;
    const int GR = nt / CW, col = tid % CW, grp = tid / CW;          // GR = 4
    const int cc = col < NP ? col : NP - 1;
    const int jcnt = col < NP ? min(N, cc / 2 + 1) : 0;              // stages that reach this column
    // wave-uniform: stages that reach every column of the wave / any column of the wave
    const int c_lo = col & ~63, c_hi = min(NP - 1, col | 63);
    const int jall = __builtin_amdgcn_readfirstlane((col | 63) < NP ? min(N, c_lo / 2 + 1) : 0);
    const int jany = __builtin_amdgcn_readfirstlane(c_lo < NP ? min(N, c_hi / 2 + 1) : 0);
    const int goff0 = goff(g.j0, M, NP);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};            // four independent chains (a dependent f64 FMA issues every ~13 clocks)
    for (int b = grp; b < M; b += GR) {
        // stages [jb, je) of input b; MASK: lanes past their last stage contribute zero
        auto pass = [&](auto src, int jb, int je, auto MASK) {
            constexpr bool masked = decltype(MASK)::value;
            int R = goff(jb, M, NP) + b * (NP - 2 * jb) - 2 * jb + cc, dl = M * (NP - 2 * jb) - 2 * b - 2;
            for (int j0 = jb; j0 < je; j0 += CH) {
                double gv[CH], uu[CH];
#pragma unroll
                for (int t = 0; t < CH; ++t) {
                    gv[t] = src[R];
                    uu[t] = uv[min(j0 + t, N - 1) * M + b];
                    R += dl; dl -= 2 * M;
                }
#pragma unroll
                for (int t = 0; t < CH; ++t) {
                    const int j = j0 + t;
                    bool ok = j < je;                                 // uniform: the tail of the last trip
                    if constexpr (masked) ok = ok && j < jcnt;
                    acc[t & 3] = fma(ok ? gv[t] : 0.0, uu[t], acc[t & 3]);     // (what masked lanes read may be anything)
                }
            }
        };
        auto span = [&](auto src, int lo, int hi) {                   // stages [lo, hi) of one store
            const int mid = max(lo, min(hi, jall));
            if (mid > lo) pass(src, lo, mid, std::false_type{});
            if (hi > mid) pass(src, mid, hi, std::true_type{});
        };
        span(g.gh, 0, min(g.j0, jany));
        span(g.gt - goff0, g.j0, jany);
    }
    lptr part = L.part;                                     // GR * CW slots: 512 for the workgroup, 256 for a half set
    part[grp * CW + col] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    W.sync();
    if (tid < ldG) {
        double s = 0.0;
        if (tid < NP) for (int q = 0; q < GR; ++q) s += part[q * CW + tid];
        yv[tid] = s;
    }
    W.sync();
}
template <int MSEL, class GP>
__device__ __forceinline__ void g_times(const QPDims &d, const GP &g, Lds &L, clptr uv, lptr yv) {
    auto W = all_waves();
    g_times<MSEL, false>(d, g, L, uv, yv, W);
}

// out1[row] = sum_{i >= 2 j} G^T[row][i] y1[i]  (and out2 with y2 when y2 != null): 8 lanes per row, 64 rows per pass; the
// L2-resident rows (stages < j0) in passes of their own.  No masks: y1 / y2 must be zero from index NP up to NP + YPAD - 1
// (the lanes run to the length of the longest row of the pass; what they read of G past the end of a row is the next
// rows' data -- finite -- times those zeros).
template <int MSEL, bool HALF, class GP>
__device__ __forceinline__ void gT_times(const QPDims &d, const GP &g, Lds &L, clptr y1, clptr y2, lptr out1, lptr out2, Waves<HALF> &W) {
    constexpr int M = MSEL;
    const int NP = g.NP, nm = d.N * M, tid = W.tid, nt = W.nt;
    const int g8 = tid & 7, rpp = nt / 8;
    const int goff0 = goff(g.j0, M, NP);
    auto rows = [&](auto src, int rb, int re, auto TWO) {           // rows [rb, re) from `src` (indexed by the global offset)
        constexpr bool two = decltype(TWO)::value;
        for (int r0 = rb; r0 < re; r0 += rpp) {
            const int r = r0 + (tid >> 3), rc = r < re ? r : re - 1;
            const int j = rc / M, b = rc - j * M, len = NP - 2 * j;
            const int at = goff(j, M, NP) + b * len + g8;
            clptr p1 = y1 + 2 * j + g8, p2 = (two ? y2 : y1) + 2 * j + g8;
            const int nq = (NP - 2 * (r0 / M) + 7) >> 3;             // trips of the longest row of this pass (uniform)
            double a1 = 0.0, a2 = 0.0, c1 = 0.0, c2 = 0.0;         // two chains per output
            int q = 0;
            for (; q + 4 <= nq; q += 4) {
                double gv[4], ya[4], yb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { gv[t] = src[at + 8 * (q + t)]; ya[t] = p1[8 * (q + t)]; if (two) yb[t] = p2[8 * (q + t)]; }
                a1 = fma(gv[0], ya[0], a1); c1 = fma(gv[1], ya[1], c1); a1 = fma(gv[2], ya[2], a1); c1 = fma(gv[3], ya[3], c1);
                if (two) { a2 = fma(gv[0], yb[0], a2); c2 = fma(gv[1], yb[1], c2); a2 = fma(gv[2], yb[2], a2); c2 = fma(gv[3], yb[3], c2); }
            }
            for (; q < nq; ++q) {
                const double gq = src[at + 8 * q];
                a1 = fma(gq, p1[8 * q], a1);
                if (two) a2 = fma(gq, p2[8 * q], a2);
            }
            a1 += c1; a2 += c2;
            a1 = wg::group_sum<8>(a1);
            if (two) a2 = wg::group_sum<8>(a2);
            if (g8 == 0 && r < re) { out1[r] = a1; if (two) out2[r] = a2; }
        }
    };
    const int rh = min(nm, g.j0 * M);
    if (y2) {
        if (rh > 0) rows(g.gh, 0, rh, std::true_type{});
        if (nm > rh) rows(g.gt - goff0, rh, nm, std::true_type{});
    } else {
        if (rh > 0) rows(g.gh, 0, rh, std::false_type{});
        if (nm > rh) rows(g.gt - goff0, rh, nm, std::false_type{});
    }
    W.sync();
}
template <int MSEL, class GP>
__device__ __forceinline__ void gT_times(const QPDims &d, const GP &g, Lds &L, clptr y1, clptr y2, lptr out1, lptr out2) {
    auto W = all_waves();
    gT_times<MSEL, false>(d, g, L, y1, y2, out1, out2, W);
}

// ------------------------------------------------------------------ Gram matrix K = I + Ls^T (G D^-1 G^T) Ls -> upper tiles
// One task = up to 4 tiles (I, J0 .. J0 + nJ - 1) of one tile row: the A operand (columns 16 I .. of the packed rows) is
// shared by the tiles of the task; k-steps = 4 packed rows = the inputs 4 sub .. 4 sub + 3 of one stage, and a tile row only
// runs over the stages that reach it: 2 j < 16 (I + 1).  Stages with 2 j <= 16 I reach every lane of the tile row (no
// masks); the last seven are triangular.  sched: per wave 4 tasks x {I, J0, nJ, 0} (host-built, longest first onto the
// least loaded SIMD; nJ = 0: no more tasks).
#ifdef SRH_PROFILE
__device__ long long g_prof_waves[16];         // per-wave clocks of the Gram fill (products, epilogue), cumulative over the launch's workgroup 0
#endif
template <int MSEL, class GP>
__device__ __forceinline__ void gram(const QPDims &d, const QPConst &c, const GP &g, Lds &L) {
    static_assert(MSEL == 4 || MSEL == 8, "lean Gram: n_u = 4 or 8");
    constexpr int M = MSEL, SPS = M / 4;                       // k-steps per stage
    const int N = d.N, KT = d.KT, NP = g.NP;
    const int tid = SRH_TID, nt = blockDim.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l16 = lane & 15, kk = lane >> 4;
    const int goff0 = goff(g.j0, M, NP);
    lptr w2 = L.tc;                                            // 1 / D per packed row
#ifdef SRH_PROFILE
    long long gl[6] = {0, 0, 0, 0, 0, 0}, g0 = clock64(), g1;   // 1/D pass + barrier, descriptors, products, epilogue, barrier, scaling + barrier  (wave 0 -> L.Qu[4..7], L.Qu[13..14])
#define GR_LAP(i) do { g1 = clock64(); gl[i] += g1 - g0; g0 = g1; } while (0)
#else
#define GR_LAP(i) ((void)0)
#endif
    // The tile tasks are PULLED: the list (longest first, scp_host.h) goes into LDS beside a counter, a wave that has finished a task
    // takes the next one -- the static assignment of round 5a left the busiest wave 40 % behind the first one to finish whatever
    // cost model placed the tasks (the waves of a SIMD share its MFMA pipe, the L2 head of G is slower than its LDS tail, ...).
    // The list lives in the reduction scratch of the products (L.part: free during the Gram fill).  Which wave computes a tile
    // does not change its bits.
    liptr tl = (liptr)L.part;                                  // [0] next task, [4 + 4 t ..] = {I, J0, nJ, 0} of task t
    if (tid < GRAM_TASKS * 4) tl[4 + tid] = c.gram_sched[tid];
    if (tid == 0) tl[0] = 0;
    for (int e = tid; e < N * M; e += nt) { const double s = L.Ldi[e]; w2[e] = s * s; }
    __syncthreads();
    GR_LAP(0);
    while (true) {
        int tnext = 0;
        if (lane == 0) tnext = __hip_atomic_fetch_add(tl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tnext = __builtin_amdgcn_readfirstlane(tnext);
        if (tnext >= GRAM_TASKS) break;
        const int I = __builtin_amdgcn_readfirstlane(tl[4 + 4 * tnext]), J0 = __builtin_amdgcn_readfirstlane(tl[5 + 4 * tnext]),
                  nJ = __builtin_amdgcn_readfirstlane(tl[6 + 4 * tnext]);
        GR_LAP(1);
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
#ifndef SRH_GRAM_KS
#define SRH_GRAM_KS 4
#endif
            constexpr int UN = SRH_GRAM_KS / SPS;                            // stages per full trip: SRH_GRAM_KS k-steps
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
            if (hf > 0) run(g.gh, 0, hf, std::false_type{}, NJ);
            if (he > hf) run(g.gh, hf, he, std::true_type{}, NJ);
            if (jfull > g.j0) run(g.gt - goff0, g.j0, jfull, std::false_type{}, NJ);
            if (jend > max(jfull, g.j0)) run(g.gt - goff0, max(jfull, g.j0), jend, std::true_type{}, NJ);
        };
        if (nJ == 4) task_body(std::integral_constant<int, 4>{});
        else if (nJ == 3) task_body(std::integral_constant<int, 3>{});
        else if (nJ == 2) task_body(std::integral_constant<int, 2>{});
        else task_body(std::integral_constant<int, 1>{});
        GR_LAP(2);
        // ---- Ls on both sides, + I, raw tile to the store; the diagonal feeds the Jacobi scaling
        // (the factors of the rows are the same for every tile of the task; the row r ^ 1 of an output stage sits in the neighbouring
        // row of 16 lanes: wg::xor16, two v_permlane16_swap instead of a ds_bpermute round trip; the scale factors of the Jacobi
        // scaling only where the tile is a diagonal one -- round 5: 13 k of the 40 k clocks of a Gram fill were this epilogue)
        double la0[4], la2[4], la3[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gi = 16 * I + kk + 4 * q, ka = min(gi >> 1, N - 1);
            clptr La = L.Ls + (size_t)ka * 4;
            la0[q] = La[0]; la2[q] = La[2]; la3[q] = La[3];
        }
        const bool ar0 = (kk & 1) == 0;                                  // row of the output stage: 16 I + 4 q are even
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
                const int r = kk + 4 * q, gi = 16 * I + r;
                double v = (gi < NP && gjc < NP) ? acc[t][q] : 0.0;      // padding rows / columns: exactly the identity
                const double vp = wg::dpp_mov<0xB1>(v);                 // the other column of the output stage
                v = fma(vp, cb_oth, v * cb_own);                        // (Ky Ls)
                const double vr = wg::xor16(v);                         // the other row of the output stage (kk ^ 1)
                v = ar0 ? fma(la2[q], vr, la0[q] * v) : la3[q] * v;     // Ls^T (Ky Ls)
                if (I == J) {                                           // (the whole wave)
                    const bool dg = r == l16;
                    v += dg ? 1.0 : 0.0;
                    const double ri = qpc::rsq3(dg ? v : 1.0);
                    if (dg) L.ks[gi] = ri;
                }
                T[r * TS + l16] = v;
            }
        }
        GR_LAP(3);
    }
    __syncthreads();
    GR_LAP(4);
    // ---- symmetric scaling to a unit diagonal (see qpc::gram for why it matters): the upper tiles over the waves
    for (int t = wave, I = 0, rowlen = KT; t < KT * (KT + 1) / 2; t += nt >> 6) {
        int tt = t;
        I = 0; rowlen = KT;
        while (tt >= rowlen) { tt -= rowlen; ++I; --rowlen; }
        const int J = I + tt;
        lptr T = L.B + (size_t)qpc::tile_index(I, J, KT) * TSZ;
        const double sc = L.ks[16 * J + l16];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int r = kk + 4 * q; T[r * TS + l16] *= L.ks[16 * I + r] * sc; }
    }
    __syncthreads();
    GR_LAP(5);
#ifdef SRH_PROFILE
    if (tid == 0) { for (int i = 0; i < 4; ++i) L.Qu[4 + i] += (double)gl[i]; L.Qu[13] += (double)gl[4]; L.Qu[14] += (double)gl[5]; }
    if (lane == 0 && blockIdx.x == 0) { g_prof_waves[wave] += gl[2]; g_prof_waves[8 + wave] += gl[3]; }
#endif
#undef GR_LAP
}

// ------------------------------------------------------------------ tile Cholesky on one wave set, without set-wide barriers
// qpc::tile_cholesky synchronises the whole workgroup twice per tile row; its critical path is wave 0 (a 16 x 16 factorisation
// per row, ~4.7 k clocks each) and every barrier on that path is paid in full.  Here wave 0 of the set never waits for a
// barrier: the dependencies are three monotone counters in LDS (F[0]: trailing updates finished, one arrival per worker and
// row; F[1]: inverses of the diagonal tiles published by wave 0; F[2]: panel tiles of a row written, one arrival per wave),
//     wave 0:   [wait: the updates of the rows before are in]  R_J,J+1 = Rinv_J^T K_J,J+1   -> F[2]
//               K_J+1,J+1 -= R_J,J+1^T R_J,J+1;  chol16                                     -> F[1]
//     workers:  [wait F[1]: Rinv_J; wait F[0]: the rows before]  their panel tiles R_JJ'    -> F[2]
//               [wait F[2]: the whole panel]  their trailing tiles                           -> F[0]
// and what wave 0 waits for (the workers' trailing updates of the row before) has normally finished long before it asks: the
// workers run one row behind, beside the factorisation of the next diagonal tile.  A wave's LDS writes reach the LDS before
// its own later arrival (the LDS queue of a wave is in order), the fences keep the compiler from moving accesses across.
// No early exit on a failed pivot (NaNs are harmless, the caller tests L.flag[1] after the halves have joined).
__device__ __forceinline__ void set_signal(liptr c) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((SRH_TID & 63) == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// SLEEP: the workers' waits (one row behind wave 0, long): s_sleep between polls -- same time as polling back to back
// (58.7 vs 58.7-59.2 ms per 4096 rollouts), fewer instructions issued beside the wave that factorises
template <bool SLEEP = false>
__device__ __forceinline__ void set_wait(liptr c, int v) {
    if ((SRH_TID & 63) == 0) { while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) { if (SLEEP) __builtin_amdgcn_s_sleep(4); } }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ wg::qp_d4 panel_tile(Lds &L, int KT, int J, int Jp, int l16, int kk) {      // R_JJ' = Rinv_J^T K_JJ' (returned as well)
    clptr Ri = rinv_tile(L, J, KT);
    lptr T = L.B + (size_t)qpc::tile_index(J, Jp, KT) * TSZ;
    double av[4], bv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { av[s] = Ri[(4 * s + kk) * TS + l16]; bv[s] = T[(4 * s + kk) * TS + l16]; }
    wg::qp_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) T[(kk + 4 * qd) * TS + l16] = acc[qd];
    return acc;
}
// T <- T - P^T P with P in the registers panel_tile() returned: the accumulator layout of the MFMA (lane (l16, kk): rows kk + 4 q) IS its
// operand layout (k-step s: row 4 s + kk) -- the same products in the same order as qpc::tile_update(T, P, P), without reading P back
__device__ __forceinline__ void tile_update_reg(lptr T, const wg::qp_d4 &P, int l16, int kk) {
    wg::qp_d4 acc;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) acc[qd] = T[(kk + 4 * qd) * TS + l16];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-P[s], P[s], acc, 0, 0, 0);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) T[(kk + 4 * qd) * TS + l16] = acc[qd];
}
// F: three LDS counters, zero on entry.  Returns nothing: L.flag[1] = 1 iff every pivot was positive (valid after the caller's barrier).
__device__ __forceinline__ void tile_cholesky_set(const QPDims &d, Lds &L, const Waves<true> &W, liptr F, liptr Fnext) {
    const int KT = d.KT, wave = W.wave, lane = W.tid & 63, nwk = W.nw - 1;
    const int l16 = lane & 15, kk = lane >> 4;
    liptr Ftrail = F, Frinv = F + 1, Fpanel = F + 2;
    if (wave == 0) {
#ifdef SRH_PROFILE
        long long cl[5] = {0, 0, 0, 0, 0}, c0 = clock64(), c1;            // wait, panel, update, factor, signals (L.Qu[8..12])
#define TC_LAP(i) do { c1 = clock64(); cl[i] += c1 - c0; c0 = c1; } while (0)
#else
#define TC_LAP(i) ((void)0)
#endif
        bool ok = qpc::chol16<false>(L.B, rinv_tile(L, 0, KT));
        TC_LAP(3);
        set_signal(Frinv);
        TC_LAP(4);
        for (int J = 0; J + 1 < KT; ++J) {
            set_wait(Fnext, 2 * J);                               // (J, J + 1) and (J + 1, J + 1) carry the updates of row J - 1
            TC_LAP(0);
            const wg::qp_d4 P = panel_tile(L, KT, J, J + 1, l16, kk);
            TC_LAP(1);
            set_signal(Fpanel);
            TC_LAP(4);
            lptr T = L.B + (size_t)qpc::tile_index(J + 1, J + 1, KT) * TSZ;
            __builtin_amdgcn_wave_barrier();
            tile_update_reg(T, P, l16, kk);
            __builtin_amdgcn_wave_barrier();
            TC_LAP(2);
            ok = qpc::chol16<false>(T, rinv_tile(L, J + 1, KT)) && ok;
            TC_LAP(3);
            set_signal(Frinv);
            TC_LAP(4);
        }
        if (lane == 0) L.flag[1] = ok ? 1 : 0;
#ifdef SRH_PROFILE
        if (lane == 0) for (int i = 0; i < 5; ++i) L.Qu[8 + i] = (double)cl[i];
#endif
#undef TC_LAP
    } else {
        for (int J = 0; J + 1 < KT; ++J) {
            set_wait<true>(Frinv, J + 1);
            set_wait<true>(Ftrail, nwk * J);
            for (int Jp = J + 2 + (wave - 1); Jp < KT; Jp += nwk) panel_tile(L, KT, J, Jp, l16, kk);      // (J, J + 1) is wave 0's
            set_signal(Fpanel);
            set_wait<true>(Fpanel, (nwk + 1) * (J + 1));
            // tiles 1 .. ntr-1 over the workers (tile 0 = (J+1, J+1) is wave 0's).  The two tiles wave 0 needs for the NEXT row -- tile 1 =
            // (J+1, J+2) and tile `rem` = (J+2, J+2) -- are the first tiles of workers 1 and 2 (tile `rem` changes places with tile 2),
            // each announced on its own: wave 0 does not wait for the whole trailing update of a row (20 tiles on three waves in row 0).
            const int rem = KT - J - 1, ntr = rem * (rem + 1) / 2, wB = nwk >= 2 ? 2 : 1;
            for (int t = wave; t < ntr; t += nwk) {
                int tt = (t == 2 && rem > 2) ? rem : ((t == rem && rem > 2) ? 2 : t), Ir = 0;
                const bool first = (rem >= 2) && (t == 1 || tt == rem);
                while (tt >= rem - Ir) { tt -= rem - Ir; ++Ir; }
                const int I = J + 1 + Ir, Kc = I + tt;
                qpc::tile_update(L.B + (size_t)qpc::tile_index(I, Kc, KT) * TSZ, L.B + (size_t)qpc::tile_index(J, I, KT) * TSZ,
                                 L.B + (size_t)qpc::tile_index(J, Kc, KT) * TSZ, l16, kk);
                if (first) set_signal(Fnext);
            }
            set_signal(Ftrail);
        }
    }
}

// ------------------------------------------------------------------ K^-1 v with the factor in unit-block-diagonal form
// qpc::k_solve substitutes through the 2 KT block rows on one wave with a diagonal solve, two LDS round trips and four
// ds_bpermute sums per block row (23 k clocks per right-hand side, two or three per factorisation).  Here the factor is
// rewritten once per factorisation as R = Dh Uh, Dh = blockdiag(R_JJ), Uh_IJ = Rinv_I R_IJ (unit block diagonal, in place
// of R_IJ: one MFMA product per tile, all waves), so that
//     K^-1 v = Uh^-1 ( Dh^-1 Dh^-T ( Uh^-T v ) )
// and both block substitutions have NO diagonal solve: right-looking on one wave, the partial sums of every block row that
// is still open live in registers (the updates of the rows further down are independent work that covers the latency of
// the one LDS round trip per step), the 4 partial sums of an entry sit in one quad (two DPP adds).  The block-diagonal
// scaling in the middle is one tile per wave.  (An explicit inverse T = R^-1 with x = T T^T v was measured as well: 2.5 k
// clocks per solve, but its residual costs the interior point an iteration on some QPs -- slower overall.)
__device__ __forceinline__ void unit_tiles(const QPDims &d, Lds &L) {
    const int KT = d.KT, tid = SRH_TID, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l16 = lane & 15, kk = lane >> 4, noff = KT * (KT - 1) / 2;
    for (int t = wave; t < noff; t += (int)(blockDim.x >> 6)) {
        int tt = t, I = 0;
        while (tt >= KT - 1 - I) { tt -= KT - 1 - I; ++I; }
        const int J = I + 1 + tt;
        lptr T = L.B + (size_t)qpc::tile_index(I, J, KT) * TSZ;
        clptr Ri = rinv_tile(L, I, KT);
        double av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { av[s] = Ri[l16 * TS + 4 * s + kk]; bv[s] = T[(4 * s + kk) * TS + l16]; }
        wg::qp_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) T[(kk + 4 * q) * TS + l16] = acc[q];
    }
    __syncthreads();
}

// v <- K^-1 v (scaled K; v: 16 KT doubles of LDS).  KT <= 8; KTC > 0 fixes KT at compile time (no branch between the tile
// updates of a step: their LDS reads are issued together instead of one tile at a time).
template <int KTC>
__device__ __forceinline__ void k_solve_unit_impl(const QPDims &d, Lds &L, lptr v) {
    constexpr int KMAX = KTC > 0 ? KTC : 8;
    const int KT = KTC > 0 ? KTC : d.KT, tid = SRH_TID, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c = lane >> 2, part = lane & 3;
    const int sw = serial_wave(L);
    if (wave == sw) {                                                    // y = Uh^-T v
        double acc[KMAX];
#pragma unroll
        for (int J = 0; J < KMAX; ++J) acc[J] = 0.0;
#pragma unroll
        for (int I = 0; I < KMAX; ++I) {
            if (KTC > 0 || I < KT) {
                const double yI = v[16 * I + c] - wg::group_sum<4>(acc[I]);
                __builtin_amdgcn_wave_barrier();
                if (part == 0) v[16 * I + c] = yI;
                __builtin_amdgcn_wave_barrier();
                const double y0 = v[16 * I + 4 * part], y1 = v[16 * I + 4 * part + 1], y2 = v[16 * I + 4 * part + 2], y3 = v[16 * I + 4 * part + 3];
#pragma unroll
                for (int J = I + 1; J < KMAX; ++J) {
                    if (KTC > 0 || J < KT) {
                        clptr T = L.B + (size_t)qpc::tile_index(I, J, KT) * TSZ + (4 * part) * TS + c;
                        acc[J] = fma(T[0], y0, fma(T[TS], y1, fma(T[2 * TS], y2, fma(T[3 * TS], y3, acc[J]))));
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int J = wave; J < KT; J += (int)(blockDim.x >> 6)) {            // w_J = Rinv_J (Rinv_J^T y_J): a tile per wave (4-wave workgroups: two rounds)
        clptr Ri = rinv_tile(L, J, KT);
        double t = 0.0;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) t = fma(Ri[(4 * part + kq) * TS + c], v[16 * J + 4 * part + kq], t);
        t = wg::group_sum<4>(t);
        __builtin_amdgcn_wave_barrier();
        if (part == 0) v[16 * J + c] = t;
        __builtin_amdgcn_wave_barrier();
        double w = 0.0;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) w = fma(Ri[c * TS + 4 * part + kq], v[16 * J + 4 * part + kq], w);
        w = wg::group_sum<4>(w);
        __builtin_amdgcn_wave_barrier();
        if (part == 0) v[16 * J + c] = w;
    }
    __syncthreads();
    if (wave == sw) {                                                    // x = Uh^-1 w
        double acc[KMAX];
#pragma unroll
        for (int J = 0; J < KMAX; ++J) acc[J] = 0.0;
#pragma unroll
        for (int Jp = KMAX - 1; Jp >= 0; --Jp) {
            if (KTC > 0 || Jp < KT) {
                const double xJ = v[16 * Jp + c] - wg::group_sum<4>(acc[Jp]);
                __builtin_amdgcn_wave_barrier();
                if (part == 0) v[16 * Jp + c] = xJ;
                __builtin_amdgcn_wave_barrier();
                const double x0 = v[16 * Jp + 4 * part], x1 = v[16 * Jp + 4 * part + 1], x2 = v[16 * Jp + 4 * part + 2], x3 = v[16 * Jp + 4 * part + 3];
#pragma unroll
                for (int J = 0; J < Jp; ++J) {
                    clptr T = L.B + (size_t)qpc::tile_index(J, Jp, KT) * TSZ + c * TS + 4 * part;
                    acc[J] = fma(T[0], x0, fma(T[1], x1, fma(T[2], x2, fma(T[3], x3, acc[J]))));
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void k_solve_unit(const QPDims &d, Lds &L, lptr v) {
    if (d.KT == 7) k_solve_unit_impl<7>(d, L, v);            // N p_o = 100: the benchmark horizons
    else k_solve_unit_impl<0>(d, L, v);
}

// Newton direction (qpc::newton_solve with the products from the packed store), in two halves:
//   front (needs the gradients, D, Ls and the scaling ks of K -- NOT the factor): t = -D^-1 (g_u + G^T g_y), the reduced dual
//          residual max |g_ud + G^T g_yd| when gyd != null (into L.Qu[0]), yc = ks Ls^T G t;  runs on any wave set
//   back  (needs the factor): v = K^-1 yc, du = t - D^-1 G^T Ls ks v, dy = G du
// keep_gt / reuse_gt (problems without state rows: the y-space gradient g_y = L.ya is the cost's alone and does not change between
// the predictor and the corrector of an iteration): the predictor leaves G^T g_y in L.tb (free once the dual residual has been
// taken), the corrector reads it there instead of repeating the product.
template <int MSEL, bool HALF, class GP>
__device__ __forceinline__ void newton_front(const QPDims &d, const GP &g, Lds &L, clptr gyd, Waves<HALF> &W, Prof &pf,
                                             bool keep_gt = false, bool reuse_gt = false) {
    const int N = d.N, nm = N * d.m, ldG = 16 * d.KT, tid = W.tid, nt = W.nt;
    QC_SUB(pf, 8);
    if (!reuse_gt) gT_times<MSEL, HALF>(d, g, L, L.ya, gyd, L.du, gyd ? L.tc : (lptr) nullptr, W);
    clptr gty = reuse_gt ? (clptr)L.tb : (clptr)L.du;
    QC_SUB(pf, 9);
    if (gyd) {
        double r = 0.0;
        for (int e = tid; e < nm; e += nt) r = fmax(r, fabs(L.tb[e] + L.tc[e]));
        r = wg::wave_max(r);
        if ((tid & 63) == 0) L.red[W.wave] = r;
    }
    for (int e = tid; e < nm; e += nt) {                                                                            // t = -D^-1 g (diagonal D)
        const double sd = L.Ldi[e], gt = gty[e];
        L.ta[e] = -(L.ta[e] + gt) * (sd * sd);
        if (keep_gt) L.tb[e] = gt;
    }
    W.sync();
    if (gyd && tid == 0) { double r = L.red[0]; for (int i = 1; i < W.nw; ++i) r = fmax(r, L.red[i]); L.Qu[0] = r; }
    QC_SUB(pf, 10);
    g_times<MSEL, HALF>(d, g, L, L.ta, L.yb, W);
    QC_SUB(pf, 11);
    // yc = ks (Ls^T yb) per output stage (p_o = 2, Ls lower: out_0 = L00 v0 + L10 v1, out_1 = L11 v1); padding zeroed
    for (int k = tid; k < ldG / 2; k += nt) {
        double o0 = 0.0, o1 = 0.0;
        if (k < N) {
            clptr Lk = L.Ls + (size_t)k * 4;
            const double v0 = L.yb[2 * k], v1 = L.yb[2 * k + 1];
            o0 = fma(Lk[2], v1, Lk[0] * v0) * L.ks[2 * k];
            o1 = Lk[3] * v1 * L.ks[2 * k + 1];
        }
        L.yc[2 * k] = o0; L.yc[2 * k + 1] = o1;
    }
    W.sync();
}

template <int MSEL, class GP>
__device__ __forceinline__ void newton_back(const QPDims &d, const GP &g, Lds &L, Prof &pf) {
    const int N = d.N, nm = N * d.m, ldG = 16 * d.KT, tid = SRH_TID, nt = blockDim.x;
    k_solve_unit(d, L, L.yc);
    // yd = Ls (ks v) per output stage: out_0 = L00 v0, out_1 = L10 v0 + L11 v1.
    // dy: with w = ks v the solved system reads (I + Ls^T Ky Ls) w = Ls^T yb, Ky = G D^-1 G^T, yb = G t, so that
    //     dy = G du = yb - Ky Ls w = Ls^-T w
    // -- one 2 x 2 back substitution per output stage instead of the product G du (d.ls_pd: every Ls_k is invertible; what the
    // two forms differ by is the residual of the K solve)
    const bool dy_here = d.ls_pd != 0;
    for (int k = tid; k < ldG / 2; k += nt) {
        double o0 = 0.0, o1 = 0.0, e0 = 0.0, e1 = 0.0;
        if (k < N) {
            clptr Lk = L.Ls + (size_t)k * 4;
            const double v0 = L.yc[2 * k] * L.ks[2 * k], v1 = L.yc[2 * k + 1] * L.ks[2 * k + 1];
            o0 = Lk[0] * v0;
            o1 = fma(Lk[3], v1, Lk[2] * v0);
            e1 = v1 / Lk[3];
            e0 = fma(-Lk[2], e1, v0) / Lk[0];
        }
        L.yd[2 * k] = o0; L.yd[2 * k + 1] = o1;
        if (dy_here) { L.dy[2 * k] = e0; L.dy[2 * k + 1] = e1; }
    }
    __syncthreads();
    QC_SUB(pf, 12);
    gT_times<MSEL>(d, g, L, L.yd, (clptr) nullptr, L.du, (lptr) nullptr);
    QC_SUB(pf, 13);
    for (int e = tid; e < nm; e += nt) { const double sd = L.Ldi[e]; L.du[e] = L.ta[e] - L.du[e] * (sd * sd); }
    __syncthreads();
    QC_SUB(pf, 14);
    if (!dy_here) g_times<MSEL>(d, g, L, L.du, L.dy);
    QC_SUB(pf, 15);
}

// the whole solve on the whole workgroup (the general-row interior point; corrector solves)
template <int MSEL, class GP>
__device__ __forceinline__ void newton_solve(const QPDims &d, const GP &g, Lds &L, clptr gyd, double *rd, Prof &pf, bool reuse_gt = false) {
    auto W = all_waves();
    newton_front<MSEL, false>(d, g, L, gyd, W, pf, false, reuse_gt);
    if (gyd) *rd = L.Qu[0];
    newton_back<MSEL>(d, g, L, pf);
}

// ------------------------------------------------------------------ the QP without its trust-region rows
// Results: w.u, and -- after the final rollout of solve_qp below -- w.x.  Returns 0 optimal, 1 max iterations, 2 numerical failure.
template <int MSEL, int NSEL>
__device__ __forceinline__ int ipm(const QPDims &dfull, const QPConst &c, const QPDyn &dyn, const QPData &q, gptr work_base,
                                   Lds &L, QPLds &Lq, int *iters_out, QPWork &wout, long long *prof) {
    int tid = SRH_TID;                                       // re-read at the top of every interior-point iteration (dev_la.h: SRH_TID)
    const int nt = blockDim.x;
    QPDims d = dfull;
    d.tr = 0;
    d.nrx = d.nX;
    d.RX = d.nrx + d.nXf;
    d.NR = d.N * d.RX + d.N * d.nU;
    d.ng = d.N * d.nrx + d.nXf + d.N * d.nU;
    QPWork w;
    qp_carve(w, work_base, d);
    wout = w;
    gptr gh = work_base + dfull.qc_off;
    const int N = d.N, m = d.m, po = d.po, nm = N * m, ldG = 16 * d.KT, NP = N * po;
    GPack g{(cgptr)gh, (clptr)(L.Gt), d.lean_j0, m, NP};
    Prof pf;
#ifdef SRH_PROFILE
    for (int i = 0; i < 24; ++i) pf.t[i] = 0;
    long long tq_last = clock64();
    auto qlap = [&](int slot) { const long long now = clock64(); prof[slot] += now - tq_last; tq_last = now; };
#define QL_LAP(x) qlap(x)
#else
#define QL_LAP(x) ((void)0)
#endif
    for (int e = tid; e < nm; e += nt) { w.u[e] = 0.0; L.u[e] = 0.0; }
    static_assert(YPAD == 48, "the carve reserves 48 doubles behind ya, yd, yg");
    for (int e = tid; e < YPAD; e += nt) { L.ya[ldG + e] = 0.0; L.yd[ldG + e] = 0.0; L.yg[ldG + e] = 0.0; }
    for (int e = tid; e <= N; e += nt) w.s[e] = 0.0;
    for (int k = tid; k < N; k += nt) L.idxl[k] = dyn.idx ? dyn.idx[k] : k;
    for (int e = tid; e < d.nU * m; e += nt) L.UA[e] = c.UA[e];
    for (int e = tid; e < (d.nX + d.nXf) * po; e += nt) L.Tx[e] = e < d.nX * po ? c.Tx[e] : c.Txf[e - d.nX * po];
    __syncthreads();
    // The condensation (G, free response) depends on the linearisation only: when the region sequence of this QP equals
    // the one G was built from (a rejected SCP step: only delta / omega change, gusto.py:341 `update(full=new)`; or an
    // accepted step whose trajectory stays in the same regions) it is still in LDS / the L2 block.  `x0` does not change
    // inside a solve; the single-QP kernel (no region index) always condenses.
    bool reuse = false;
    if (dyn.idx != nullptr) {
        int same = L.flag[2];
        for (int k = tid; k < N; k += nt) same = same && (L.goff[k] == L.idxl[k]);
        if (tid == 0) L.flag[3] = 1;                       // (__syncthreads_and brings static LDS of its own: the carve uses all 160 KB)
        __syncthreads();
        if (!same) L.flag[3] = 0;
        __syncthreads();
        reuse = L.flag[3] != 0;
    }
    if (!reuse) {
    rollout<MSEL, NSEL>(d, dyn, q.x0, (cgptr) nullptr, w.x, L);
    QL_LAP(0);
    condense<MSEL, NSEL>(d, c, dyn, w.x, gh, L);
    for (int k = tid; k < N; k += nt) L.goff[k] = L.idxl[k];
    if (tid == 0) L.flag[2] = 1;
    }
    for (int e = tid; e < ldG; e += nt) { L.y[e] = L.yf[e]; L.dy[e] = 0.0; }
    __syncthreads();
    QL_LAP(2);
    bool rv[QR], ru[QR];
    int rk[QR], rr[QR];
    double rh[QR], rt[QR], rlam[QR], rrg[QR], rrc[QR], rdt[QR], rdl[QR];
#pragma unroll
    for (int qi = 0; qi < QR; ++qi) {
        const int e = tid + nt * qi, nxs = N * d.RX;
        rt[qi] = rlam[qi] = rrg[qi] = rrc[qi] = rdt[qi] = rdl[qi] = 0.0;
        if (e < nxs) {
            rk[qi] = e / d.RX + 1; rr[qi] = e - (rk[qi] - 1) * d.RX; ru[qi] = false;
            rv[qi] = rr[qi] < qp::xrows_of(d, rk[qi]);
        } else {
            const int e2 = e - nxs;
            ru[qi] = true; rv[qi] = e2 < N * d.nU;
            rk[qi] = rv[qi] ? e2 / d.nU : 0; rr[qi] = rv[qi] ? e2 - rk[qi] * d.nU : 0;
        }
        rh[qi] = rv[qi] ? qp::row_h(d, c, q, ru[qi], rk[qi], rr[qi]) : 0.0;
    }
    auto row_val = [&](int qi, clptr vy, clptr vu) {
        double acc = 0.0;
        if (!ru[qi]) { for (int a = 0; a < po; ++a) acc = fma(L.Tx[rr[qi] * po + a], vy[(rk[qi] - 1) * po + a], acc); }
        else { for (int j = 0; j < m; ++j) acc = fma(L.UA[rr[qi] * m + j], vu[rk[qi] * m + j], acc); }
        return acc;
    };
    int status = 1, it = 0;
    enum { INIT = 0, PRED = 1, CORR = 2 };
    int mode = INIT;
    double mu = 0.0, rp = 0.0, sig = 0.0, sd = 1.0, sp = 1.0, dreg = 0.0;
    bool near_opt = false;
    while (true) {
        tid = SRH_TID;
        QL_LAP(7);
        double musum = 0.0, rpm = 0.0;
#pragma unroll
        for (int qi = 0; qi < QR; ++qi) {
            if (!rv[qi]) continue;
            const int e = tid + nt * qi;
            if (mode == INIT) {
                const double gq = row_val(qi, L.y, L.u) - rh[qi];
                w.D[e] = 1.0; w.rho[e] = gq; rlam[qi] = 0.0;
            } else if (mode == PRED) {
                const double gq = row_val(qi, L.y, L.u) - rh[qi];
                const double t = rt[qi], lam = rlam[qi], rg = gq + t;
                rrg[qi] = rg;
                const double D = lam / (t + dreg * lam);
                w.D[e] = D; w.rho[e] = D * (rg + dreg * lam); w.lam[e] = lam;
                musum += lam * t;
                rpm = fmax(rpm, fabs(rg));
            } else {
                const double t = rt[qi], lam = rlam[qi];
                const double rc = lam * t + rdt[qi] * rdl[qi] - sig * mu;
                rrc[qi] = rc;
                w.rho[e] = lam + (lam * rrg[qi] - rc) / (t + dreg * lam);
            }
        }
        if (mode == PRED) {
            mu = wg::reduce(musum, 0, L.red) / d.ng;
            rp = wg::reduce(rpm, 1, L.red);
        }
        __syncthreads();
        QL_LAP(1);
        double rd = 0.0;
        bool ok = true;
        if (mode != CORR) {
            ok = qpc::stage_factors(d, c, w.D, L);
            QL_LAP(3);
            if (ok) {
                gram<MSEL>(d, c, g, L);
                QL_LAP(4);
                ok = qpc::tile_cholesky(d, L);
                if (ok) unit_tiles(d, L);
                QL_LAP(5);
            }
        }
#ifdef SRH_PROFILE
        pf.last = clock64();
#endif
        if (ok) {
            if (mode == PRED) qpc::gradients(d, c, q, L, w.lam, L.tb, L.yg);
            qpc::gradients(d, c, q, L, w.rho, L.ta, L.ya);
            newton_solve<MSEL>(d, g, L, mode == PRED ? (clptr)L.yg : (clptr) nullptr, &rd, pf);
        }
        QL_LAP(6);
        if (mode == INIT) {
            if (!ok) { status = 2; break; }
            for (int e = tid; e < nm; e += nt) L.u[e] += L.du[e];
            for (int e = tid; e < ldG; e += nt) L.y[e] += L.dy[e];
            __syncthreads();
            if (d.ng == 0) { status = 0; break; }
            double zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
            for (int qi = 0; qi < QR; ++qi) {
                if (!rv[qi]) continue;
                const double gq = row_val(qi, L.y, L.u) - rh[qi];
                rrg[qi] = gq;
                zmin = fmin(zmin, gq); zmax = fmax(zmax, gq);
            }
            zmin = wg::reduce(zmin, 2, L.red);
            zmax = wg::reduce(zmax, 1, L.red);
            const double sh_t = zmax >= 0.0 ? 1.0 + zmax : 0.0, sh_l = zmin <= 0.0 ? 1.0 - zmin : 0.0;
#pragma unroll
            for (int qi = 0; qi < QR; ++qi) { rt[qi] = -rrg[qi] + sh_t; rlam[qi] = rrg[qi] + sh_l; }
            for (int e = tid; e < d.n; e += nt) {
                double gq = 0.0;
                if (q.z) for (int a = 0; a < d.nz; ++a) gq = fma(c.HtQz2[e * d.nz + a], -q.z[d.nz + a], gq);
                sd = fmax(sd, fabs(gq));
            }
            for (int e = tid; e < d.nU; e += nt) sp = fmax(sp, fabs(c.Ub[e]));
            sd = fmax(wg::reduce(sd, 1, L.red), q.omega);
            sp = fmax(wg::reduce(sp, 1, L.red), fabs(q.delta));
            dreg = d.reg / sd;
            mode = PRED;
            continue;
        }
        double amax = 1e300;
        if (ok) {
#pragma unroll
            for (int qi = 0; qi < QR; ++qi) {
                if (!rv[qi]) continue;
                const double t = rt[qi], lam = rlam[qi], rga = rrg[qi] + row_val(qi, L.dy, L.du);
                const double dl = ((mode == PRED ? -lam * t : -rrc[qi]) + lam * rga) / (t + dreg * lam);
                const double dtv = -rga + dreg * dl;
                rdl[qi] = dl; rdt[qi] = dtv;
                if (dtv < 0.0) amax = fmin(amax, -t / dtv);
                if (dl < 0.0) amax = fmin(amax, -lam / dl);
            }
        }
        amax = wg::reduce(amax, 2, L.red);
        if (mode == PRED) {
            if (!ok) { status = near_opt ? 0 : 2; break; }
            if (!(mu == mu)) { status = near_opt ? 0 : 5; break; }
            if (!(rd == rd)) { status = near_opt ? 0 : 6; break; }
            if (q.dbg && tid == 0) { gptr gd = q.dbg + 8 * it; gd[0] = mu; gd[1] = rd; gd[2] = rp; gd[3] = sd; gd[4] = sp; }
            const double ltol = fmax(d.tol, 1e-9);
            if (rd <= ltol * sd && rp <= ltol * sp && mu <= d.tol) { status = 0; break; }
            near_opt = (rd <= 1e-8 * sd && rp <= 1e-8 * sp && mu <= 1e-8);
            if (it >= d.max_iter) { status = 1; break; }
            const double a_aff = fmin(1.0, amax);
            double ma = 0.0;
#pragma unroll
            for (int qi = 0; qi < QR; ++qi)
                if (rv[qi]) ma += (rlam[qi] + a_aff * rdl[qi]) * (rt[qi] + a_aff * rdt[qi]);
            const double mu_aff = wg::reduce(ma, 0, L.red) / d.ng;
            sig = mu > 0.0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
            if (q.dbg && tid == 0) { gptr gd = q.dbg + 8 * it; gd[5] = a_aff; gd[6] = sig; }
            mode = CORR;
            continue;
        }
        if (!ok) { status = 2; break; }
        const double a = fmin(1.0, 0.99 * amax);
        if (q.dbg && tid == 0) { gptr gd = q.dbg + 8 * it; gd[7] = a; }
        for (int e = tid; e < nm; e += nt) L.u[e] += a * L.du[e];
        for (int e = tid; e < ldG; e += nt) L.y[e] += a * L.dy[e];
#pragma unroll
        for (int qi = 0; qi < QR; ++qi) { rt[qi] += a * rdt[qi]; rlam[qi] += a * rdl[qi]; }
        __syncthreads();
        ++it;
        mode = PRED;
    }
    __syncthreads();
    for (int e = tid; e < nm; e += nt) w.u[e] = L.u[e];
    __syncthreads();
#ifdef SRH_PROFILE
    for (int i = 0; i < 8; ++i) prof[8 + i] += pf.t[8 + i];
#endif
    if (iters_out) *iters_out = it;
    return status;
}

// ------------------------------------------------------------------ the same interior point, rows next to their sums
// ipm() above keeps qpc::solve's row handling: every inequality row in the registers of "some" thread, its weight D and
// gradient shift rho through L2 arrays, then stage_factors / gradients re-read them per stage -- three L2 round trips and
// ~35 k clocks per Newton system that do no arithmetic to speak of.  ipm_box() maps the rows so that the sums are LOCAL:
//   * input rows: the reference's HyperRectangle layout (rows 2 b and 2 b + 1 bound input b, utils.py:390-414): thread
//     (k, b) owns both, so D_kb = 2 R_bb + a0^2 D0 + a1^2 D1 and the input gradient need no other thread;
//   * state rows of stage k: GX = 1, 2, 4 or 8 adjacent lanes; S_k = S* + sum_r D_r t_r t_r^T and the output gradient by
//     DPP sums, the 2 x 2 (semidefinite) Cholesky factor in closed form by the first lane.
// Same iteration, same quantities (sums in another order: rounding-level differences).  QPDims::lean == 2 selects it.
__device__ __forceinline__ void reduce2(double &a, int opa, double &b, int opb, lptr scratch) {
    auto wr = [](double v, int op) { return op == 0 ? wg::wave_sum(v) : (op == 1 ? wg::wave_max(v) : wg::wave_min(v)); };
    const double wa = wr(a, opa), wb = wr(b, opb);
    const int wave = __builtin_amdgcn_readfirstlane(SRH_TID >> 6), nw = blockDim.x >> 6;
    __syncthreads();
    if ((SRH_TID & 63) == 0) { scratch[wave] = wa; scratch[8 + wave] = wb; }
    __syncthreads();
    auto comb = [](double x, double y, int op) { return op == 0 ? x + y : (op == 1 ? fmax(x, y) : fmin(x, y)); };
    double ra = scratch[0], rb = scratch[8];
    for (int i = 1; i < nw; ++i) { ra = comb(ra, scratch[i], opa); rb = comb(rb, scratch[8 + i], opb); }
    a = ra; b = rb;
}

template <int G>
__device__ __forceinline__ double gsum(double v) {
    if constexpr (G == 1) return v;
    else if constexpr (G == 2) return v + wg::dpp_mov<0xB1>(v);
    else return wg::group_sum<G>(v);
}

template <int MSEL, int NSEL, int GX, int NST = 0, int J0SEL = 0>
__device__ __forceinline__ int ipm_box(const QPDims &dfull, const QPConst &c, const QPDyn &dyn, const QPData &q, gptr work_base,
                                       Lds &L, int *iters_out, QPWork &wout, long long *prof, int warm_mode = 0) {
    // warm_mode: 0 cold start, 1 warm start (below), 2 warm start whose multipliers are replaced by +inf -- a test knob
    // (GustoPar::poison_warm, SRH_LEAN_POISON_WARM=1 at plan creation) that makes the warm attempt fail so that the caller's cold
    // retry runs under a test
    const bool warm = warm_mode != 0, poison = warm_mode == 2;
    int tid = SRH_TID;                                       // re-read at the top of every interior-point iteration (dev_la.h: SRH_TID)
    const int nt = blockDim.x;
    QPDims d = dfull;
    d.tr = 0;
    d.nrx = d.nX;
    d.RX = d.nrx + d.nXf;
    d.NR = d.N * d.RX + d.N * d.nU;
    d.ng = d.N * d.nrx + d.nXf + d.N * d.nU;
    QPWork w;
    qp_carve(w, work_base, d);
    wout = w;
    gptr gh = work_base + dfull.qc_off;
    const int N = d.N, m = d.m, nm = N * m, ldG = 16 * d.KT, NP = 2 * N, nz = d.nz;
    GPackT<NST, J0SEL> g{(cgptr)gh, (clptr)(L.Gt), d.lean_j0, m, NP};
    Prof pf;
#ifdef SRH_PROFILE
    for (int i = 0; i < 24; ++i) pf.t[i] = 0;
    long long tq_last = clock64();
    auto qlap = [&](int slot) { const long long now = clock64(); prof[slot] += now - tq_last; tq_last = now; };
#define QB_LAP(x) qlap(x)
#else
#define QB_LAP(x) ((void)0)
#endif
    // warm (round 4): the previous QP of this SCP solve left its minimiser in w.u and its multipliers in w.lam -- this QP differs
    // by its linearisation point only and shares most of the active set: the interior point starts from that point (u as it
    // is, slacks from THIS QP's rows, multipliers kept, both at least WARM_FLOOR from zero) and skips the initial Newton system.
    // 7-9 interior-point iterations instead of 15-18 at C2 / C5, same minimiser (oracle/condensed_ipm.py: solve(warm=...)).
    constexpr double WARM_FLOOR = 1e-2;
    if (warm) { for (int e = tid; e < nm; e += nt) L.u[e] = w.u[e]; }
    else { for (int e = tid; e < nm; e += nt) { w.u[e] = 0.0; L.u[e] = 0.0; } }
    for (int e = tid; e < ldG + YPAD; e += nt) { L.ya[e] = 0.0; L.yd[e] = 0.0; L.yg[e] = 0.0; }     // padding stays zero for good
    for (int e = tid; e <= N; e += nt) w.s[e] = 0.0;
    for (int k = tid; k < N; k += nt) L.idxl[k] = dyn.idx ? dyn.idx[k] : k;
#ifdef SRH_PROFILE
    if (tid < 16) L.Qu[tid] = 0.0;
#endif
    if (tid < 5) L.flag[3 + tid] = 0;                          // counters of the two wave sets (Waves, tile_cholesky_set; [3]: idle between the region tests of two QPs)
    __syncthreads();
    bool reuse = false;
    if (dyn.idx != nullptr) {
        int same = L.flag[2];
        for (int k = tid; k < N; k += nt) same = same && (L.goff[k] == L.idxl[k]);
        if (tid == 0) L.flag[3] = 1;
        __syncthreads();
        if (!same) L.flag[3] = 0;
        __syncthreads();
        reuse = L.flag[3] != 0;
        __syncthreads();
        if (tid == 0) L.flag[3] = 0;                           // tile_cholesky_set's early-tile counter from here on
    }
    if (!reuse) {
        rollout<MSEL, NSEL>(d, dyn, q.x0, (cgptr) nullptr, w.x, L);
        QB_LAP(0);
        condense<MSEL, NSEL>(d, c, dyn, w.x, gh, L);
        for (int k = tid; k < N; k += nt) L.goff[k] = L.idxl[k];
        if (tid == 0) L.flag[2] = 1;
    }
    if (warm) {                                               // y = y_free + G u
        __syncthreads();
        g_times<MSEL>(d, g, L, L.u, L.dy);
        for (int e = tid; e < ldG; e += nt) { L.y[e] = L.yf[e] + L.dy[e]; }
        __syncthreads();
        for (int e = tid; e < ldG; e += nt) L.dy[e] = 0.0;
    } else {
        for (int e = tid; e < ldG; e += nt) { L.y[e] = L.yf[e]; L.dy[e] = 0.0; }
    }
    __syncthreads();
    QB_LAP(2);
    // ---- this thread's rows
    const bool isu = tid < nm;
    const int tx = tid - nm;
    const bool isx = tx >= 0 && tx < N * GX;
    const int ku = isu ? tid / m : 0, bu = isu ? tid - ku * m : 0;
    const int kx = isx ? tx / GX + 1 : 1, rx = isx ? tx - (kx - 1) * GX : 0;        // stage 1..N, row slot
    const int nrk = d.nX + (kx == N ? d.nXf : 0);
    const bool xrow = isx && rx < nrk;                                               // the slot holds a state row
    const bool xlead = isx && rx == 0;                                               // first lane of the stage group
    double ca[2] = {0.0, 0.0}, rh[2] = {0.0, 0.0};                                   // row coefficients / right-hand sides
    if (isu) {
        ca[0] = c.UA[(size_t)(2 * bu) * m + bu]; ca[1] = c.UA[(size_t)(2 * bu + 1) * m + bu];
        rh[0] = c.Ub[2 * bu]; rh[1] = c.Ub[2 * bu + 1];
    } else if (xrow) {
        cgptr T = rx < d.nX ? c.Tx + (size_t)rx * 2 : c.Txf + (size_t)(rx - d.nX) * 2;
        ca[0] = T[0]; ca[1] = T[1];
        rh[0] = rx < d.nX ? c.Xb[rx] : c.Xfb[rx - d.nX];
    }
    const double r2bb = isu ? c.R2[bu * m + bu] : 0.0;
    const double udv = (isu && q.ud) ? q.ud[(size_t)ku * m + bu] : 0.0;
    // constant part of the output gradient of this stage (lead lane): -Cz2 z_k (- Czf2 zf at k = N)
    double gc0 = 0.0, gc1 = 0.0, s00 = 0.0, s01 = 0.0, s11 = 0.0;
    if (xlead) {
        cgptr S = (kx == N) ? c.ScN : c.Sc;
        s00 = S[0]; s01 = S[1]; s11 = S[3];
        if (q.z) for (int b = 0; b < nz; ++b) { gc0 = fma(-c.Cz2[b], q.z[(size_t)kx * nz + b], gc0); gc1 = fma(-c.Cz2[nz + b], q.z[(size_t)kx * nz + b], gc1); }
        if (kx == N && c.Qzf && q.zf) for (int b = 0; b < nz; ++b) { gc0 = fma(-c.Czf2[b], q.zf[b], gc0); gc1 = fma(-c.Czf2[nz + b], q.zf[b], gc1); }
    }
    const int nrow = isu ? 2 : (xrow ? 1 : 0);
    const int lslot = isu ? 2 * tid : 2 * nm + (kx - 1) * (d.nX + d.nXf) + rx;       // this thread's rows in w.lam (NR = N (nX + nXf) + N nU entries)
    double rt[2] = {0.0, 0.0}, rlam[2] = {0.0, 0.0}, rrg[2] = {0.0, 0.0}, rrc[2] = {0.0, 0.0}, rdt[2] = {0.0, 0.0}, rdl[2] = {0.0, 0.0};
    // a_row . (y, u) for row slot s2
    auto row_val = [&](int s2, clptr vy, clptr vu) -> double {
        if (isu) return ca[s2] * vu[tid];
        return fma(ca[1], vy[(kx - 1) * 2 + 1], ca[0] * vy[(kx - 1) * 2]);
    };
    int status = 1, it = 0;
    enum { INIT = 0, PRED = 1, CORR = 2 };
    int mode = INIT;
    double mu = 0.0, rp = 0.0, sig = 0.0, sd = 1.0, sp = 1.0, dreg = 0.0;
    bool near_opt = false;
    const bool ya_const = d.nX + d.nXf == 0;                  // no state rows: L.ya is the same in the predictor and the corrector (newton_front)
    auto scales = [&]() {                                     // residual scales and the dual regularisation (once per QP)
        for (int e = tid; e < d.n; e += nt) {
            double gq = 0.0;
            if (q.z) for (int a = 0; a < nz; ++a) gq = fma(c.HtQz2[e * nz + a], -q.z[nz + a], gq);
            sd = fmax(sd, fabs(gq));
        }
        for (int e = tid; e < d.nU; e += nt) sp = fmax(sp, fabs(c.Ub[e]));
        reduce2(sd, 1, sp, 1, L.red);
        sd = fmax(sd, q.omega);
        sp = fmax(sp, fabs(q.delta));
        dreg = d.reg / sd;
    };
    if (warm && d.ng > 0) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 >= nrow) continue;
            const double gq = row_val(s2, L.y, L.u) - rh[s2];
            rt[s2] = fmax(-gq, WARM_FLOOR);
            rlam[s2] = poison ? INFINITY : fmax(w.lam[lslot + s2], WARM_FLOOR);
        }
        scales();
        mode = PRED;
    }
    while (true) {
        tid = SRH_TID;
        QB_LAP(7);
        // ---------------- rows -> weights, gradient shifts, and their per-stage sums, all in place
        double musum = 0.0, rpm = 0.0;
        double Dw[2] = {0.0, 0.0}, rho[2] = {0.0, 0.0};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 >= nrow) continue;
            if (mode == INIT) {
                const double gq = row_val(s2, L.y, L.u) - rh[s2];
                Dw[s2] = 1.0; rho[s2] = gq; rlam[s2] = 0.0;
            } else if (mode == PRED) {
                const double gq = row_val(s2, L.y, L.u) - rh[s2];
                const double t = rt[s2], lam = rlam[s2], rg = gq + t;
                rrg[s2] = rg;
                Dw[s2] = lam / (t + dreg * lam);
                rho[s2] = Dw[s2] * (rg + dreg * lam);
                musum += lam * t;
                rpm = fmax(rpm, fabs(rg));
            } else {
                const double t = rt[s2], lam = rlam[s2];
                const double rc = lam * t + rdt[s2] * rdl[s2] - sig * mu;
                rrc[s2] = rc;
                rho[s2] = lam + (lam * rrg[s2] - rc) / (t + dreg * lam);
            }
        }
        if (isu) {
            const double du0 = r2bb * (L.u[tid] - udv);
            if (mode != CORR) {
                double v = fma(ca[0] * Dw[0], ca[0], r2bb);
                v = fma(ca[1] * Dw[1], ca[1], v);
                L.Ldi[tid] = 1.0 / sqrt(v);
            }
            L.ta[tid] = fma(ca[1], rho[1], fma(ca[0], rho[0], du0));
            if (mode == PRED) L.tb[tid] = fma(ca[1], rlam[1], fma(ca[0], rlam[0], du0));
        }
        if (isx) {                                           // whole stage groups: the DPP sums see every lane of a group
            const double y0 = L.y[(kx - 1) * 2], y1 = L.y[(kx - 1) * 2 + 1];
            if (mode != CORR) {
                const double a00 = gsum<GX>(ca[0] * Dw[0] * ca[0]), a01 = gsum<GX>(ca[0] * Dw[0] * ca[1]), a11 = gsum<GX>(ca[1] * Dw[0] * ca[1]);
                if (xlead) {
                    const double S00 = s00 + a00, S01 = s01 + a01, S11 = s11 + a11;
                    const double dmax = fmax(fabs(S00), fabs(S11));
                    const double l00 = S00 > 1e-14 * dmax ? sqrt(S00) : 0.0;
                    const double l10 = l00 > 0.0 ? S01 / l00 : 0.0;
                    const double v = fma(-l10, l10, S11);
                    const double l11 = v > 1e-14 * dmax ? sqrt(v) : 0.0;
                    lptr Lk = L.Ls + (size_t)(kx - 1) * 4;
                    Lk[0] = l00; Lk[1] = 0.0; Lk[2] = l10; Lk[3] = l11;
                }
            }
            const double r0 = gsum<GX>(ca[0] * rho[0]), r1 = gsum<GX>(ca[1] * rho[0]);
            double l0 = 0.0, l1 = 0.0;
            if (mode == PRED) { l0 = gsum<GX>(ca[0] * rlam[0]); l1 = gsum<GX>(ca[1] * rlam[0]); }
            if (xlead) {
                const double c0 = fma(s01, y1, s00 * y0) + gc0, c1 = fma(s11, y1, s01 * y0) + gc1;
                L.ya[(kx - 1) * 2] = c0 + r0; L.ya[(kx - 1) * 2 + 1] = c1 + r1;
                if (mode == PRED) { L.yg[(kx - 1) * 2] = c0 + l0; L.yg[(kx - 1) * 2 + 1] = c1 + l1; }
            }
        }
        if (mode == PRED) {
            reduce2(musum, 0, rpm, 1, L.red);
            mu = musum / d.ng;
            rp = rpm;
        }
        __syncthreads();
        QB_LAP(1);
        // ---------------- Newton system
        double rd = 0.0;
        bool ok = true;
        if (mode != CORR) {
            QB_LAP(3);
            gram<MSEL>(d, c, g, L);
            QB_LAP(4);
            // the factorisation (waves 0-3: a chain of one-wave 16 x 16 factorisations) beside the half of the Newton solve that
            // does not need the factor (waves 4-7), see Waves
            const clptr gyd = mode == PRED ? (clptr)L.yg : (clptr) nullptr;
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SRH_PROFILE
            pf.last = clock64();
#endif
#ifdef SRH_PROFILE
            const long long tsplit = clock64();
#endif
            auto W = half_waves(L.flag + 4);
            if (half_of_wave(wv) == 0) {
                tile_cholesky_set(d, L, W, L.flag + 5, L.flag + 3);
#ifdef SRH_PROFILE
                if (tid == 0) L.Qu[2] = (double)(clock64() - tsplit);
#endif
            } else {
                newton_front<MSEL, true>(d, g, L, gyd, W, pf, ya_const && mode == PRED);
#ifdef SRH_PROFILE
                if (W.tid == 0) L.Qu[3] = (double)(clock64() - tsplit);
#endif
            }
            __syncthreads();
#ifdef SRH_PROFILE
            prof[16] += (long long)L.Qu[2]; prof[17] += (long long)L.Qu[3];
            for (int i = 0; i < 5; ++i) prof[27 + i] += (long long)L.Qu[8 + i];
#endif
            ok = L.flag[1] != 0;
            if (tid < 5) L.flag[3 + tid] = 0;                  // the factorising half's early-tile counter, the product half's arrival counter and the factorising half's three, for the next split
            if (gyd) rd = L.Qu[0];
            if (ok) unit_tiles(d, L);
            QB_LAP(5);
#ifdef SRH_PROFILE
            pf.last = clock64();
#endif
            if (ok) newton_back<MSEL>(d, g, L, pf);
        } else {
#ifdef SRH_PROFILE
            pf.last = clock64();
#endif
            newton_solve<MSEL>(d, g, L, (clptr) nullptr, &rd, pf, ya_const);
        }
        QB_LAP(6);
        // ---------------- use the direction
        if (mode == INIT) {
            if (!ok) { status = 2; break; }
            for (int e = tid; e < nm; e += nt) L.u[e] += L.du[e];
            for (int e = tid; e < ldG; e += nt) L.y[e] += L.dy[e];
            __syncthreads();
            if (d.ng == 0) { status = 0; break; }
            double zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (s2 >= nrow) continue;
                const double gq = row_val(s2, L.y, L.u) - rh[s2];
                rrg[s2] = gq;
                zmin = fmin(zmin, gq); zmax = fmax(zmax, gq);
            }
            reduce2(zmin, 2, zmax, 1, L.red);
            const double sh_t = zmax >= 0.0 ? 1.0 + zmax : 0.0, sh_l = zmin <= 0.0 ? 1.0 - zmin : 0.0;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { rt[s2] = -rrg[s2] + sh_t; rlam[s2] = rrg[s2] + sh_l; }
            scales();
            mode = PRED;
            continue;
        }
        double amax = 1e300, dummy = 0.0;
        if (ok) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (s2 >= nrow) continue;
                const double t = rt[s2], lam = rlam[s2], rga = rrg[s2] + row_val(s2, L.dy, L.du);
                const double dl = ((mode == PRED ? -lam * t : -rrc[s2]) + lam * rga) / (t + dreg * lam);
                const double dtv = -rga + dreg * dl;
                rdl[s2] = dl; rdt[s2] = dtv;
                if (dtv < 0.0) amax = fmin(amax, -t / dtv);
                if (dl < 0.0) amax = fmin(amax, -lam / dl);
            }
        }
        QB_LAP(18);
        reduce2(amax, 2, dummy, 0, L.red);
        QB_LAP(19);
        if (mode == PRED) {
            if (!ok) { status = near_opt ? 0 : 2; break; }
            if (!(mu == mu)) { status = near_opt ? 0 : 5; break; }
            if (!(rd == rd)) { status = near_opt ? 0 : 6; break; }
            if (q.dbg && tid == 0) { gptr gd = q.dbg + 8 * it; gd[0] = mu; gd[1] = rd; gd[2] = rp; gd[3] = sd; gd[4] = sp; }
            const double ltol = fmax(d.tol, 1e-9);
            if (rd <= ltol * sd && rp <= ltol * sp && mu <= d.tol) { status = 0; break; }
            near_opt = (rd <= 1e-8 * sd && rp <= 1e-8 * sp && mu <= 1e-8);
            if (it >= d.max_iter) { status = 1; break; }
            const double a_aff = fmin(1.0, amax);
            double ma = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                if (s2 < nrow) ma += (rlam[s2] + a_aff * rdl[s2]) * (rt[s2] + a_aff * rdt[s2]);
            QB_LAP(20);
            reduce2(ma, 0, dummy, 0, L.red);
            QB_LAP(21);
            const double mu_aff = ma / d.ng;
            sig = mu > 0.0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
            if (q.dbg && tid == 0) { gptr gd = q.dbg + 8 * it; gd[5] = a_aff; gd[6] = sig; }
            mode = CORR;
            continue;
        }
        if (!ok) { status = 2; break; }
        const double a = fmin(1.0, 0.99 * amax);
        if (q.dbg && tid == 0) { gptr gd = q.dbg + 8 * it; gd[7] = a; }
        for (int e = tid; e < nm; e += nt) L.u[e] += a * L.du[e];
        for (int e = tid; e < ldG; e += nt) L.y[e] += a * L.dy[e];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { rt[s2] += a * rdt[s2]; rlam[s2] += a * rdl[s2]; }
        __syncthreads();
        ++it;
        mode = PRED;
    }
    __syncthreads();
    for (int e = tid; e < nm; e += nt) w.u[e] = L.u[e];
    if (status == 0) {                                        // what the next QP of this solve starts from (warm)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) if (s2 < nrow) w.lam[lslot + s2] = rlam[s2];
    }
    __syncthreads();
#ifdef SRH_PROFILE
    for (int i = 0; i < 8; ++i) prof[8 + i] += pf.t[8 + i];
    prof[24] += it; prof[25] += 1; prof[26] += warm ? 1 : 0;
    if (tid == 0 && blockIdx.x == 0)
        printf("lean gram laps (this QP, wave 0): 1/D+barrier %.0f descriptors %.0f products %.0f epilogue %.0f barrier %.0f scaling+barrier %.0f\n",
               L.Qu[4], L.Qu[5], L.Qu[6], L.Qu[7], L.Qu[13], L.Qu[14]);
    if (tid == 0 && blockIdx.x == 0) {
        printf("lean gram per wave (cumulative) products:");
        for (int i = 0; i < 8; ++i) printf(" %lld", g_prof_waves[i]);
        printf("  epilogue:");
        for (int i = 0; i < 8; ++i) printf(" %lld", g_prof_waves[8 + i]);
        printf("\n");
    }
#endif
    if (iters_out) *iters_out = it;
    return status;
}

// ------------------------------------------------------------------ ipm_box for the HALF-SIZE workgroup (4 waves, <= 80 KB of LDS)
// The same iteration as ipm_box() -- same starting points, weights, stopping rule, warm start, same phase order -- for a workgroup of 256
// threads that shares its CU with a second rollout (QPDims::lean_half, lean.hip: the <.., N, N, ..> instantiations).  What differs:
//   * a thread owns BOTH an input (its two box rows) and a state-row slot (ipm_box gives them to different threads: N m + N GX <= 512);
//     the per-stage sums of the state rows are the same DPP sums over GX adjacent lanes;
//   * the free response yf and the region sequence of the condensation come from the problem's L2 block (Lds::yfg / goffg), every packed
//     row of G from there as well (lean_j0 = N), the inverses of the diagonal tiles sit in the tiles' own places (rinv_tile);
//   * the condensation runs in two column passes (condense_half), rollouts take two row passes (rollout<.., WG = 4>).
// Sums over rows are taken in another order than ipm_box's (reduce2 over 4 waves, two roles per thread): rounding-level differences.
template <int MSEL, int NSEL, int GX, int NST = 0, int J0SEL = 0>
__device__ __forceinline__ int ipm_box4(const QPDims &dfull, const QPConst &c, const QPDyn &dyn, const QPData &q, gptr work_base,
                                        Lds &L, int *iters_out, QPWork &wout, long long *prof, int warm_mode = 0) {
    const bool warm = warm_mode != 0, poison = warm_mode == 2;
    int tid = SRH_TID;                                       // re-read at the top of every interior-point iteration (dev_la.h: SRH_TID)
    const int nt = blockDim.x;
    QPDims d = dfull;
    d.tr = 0;
    d.nrx = d.nX;
    d.RX = d.nrx + d.nXf;
    d.NR = d.N * d.RX + d.N * d.nU;
    d.ng = d.N * d.nrx + d.nXf + d.N * d.nU;
    QPWork w;
    qp_carve(w, work_base, d);
    wout = w;
    gptr gh = work_base + dfull.qc_off;
    const int N = d.N, m = d.m, nm = N * m, ldG = 16 * d.KT, NP = 2 * N, nz = d.nz;
    GPackT<NST, J0SEL> g{(cgptr)gh, (clptr)(L.Gt), d.lean_j0, m, NP};
    Prof pf;
    constexpr double WARM_FLOOR = 1e-2;
    if (warm) { for (int e = tid; e < nm; e += nt) L.u[e] = w.u[e]; }
    else { for (int e = tid; e < nm; e += nt) { w.u[e] = 0.0; L.u[e] = 0.0; } }
    for (int e = tid; e < ldG + L.ypad; e += nt) { L.ya[e] = 0.0; L.yd[e] = 0.0; L.yg[e] = 0.0; }     // padding stays zero for good
    for (int e = tid; e <= N; e += nt) w.s[e] = 0.0;
    for (int k = tid; k < N; k += nt) L.idxl[k] = dyn.idx ? dyn.idx[k] : k;
    if (tid < 5) L.flag[3 + tid] = 0;                          // counters of the two wave sets (Waves, tile_cholesky_set)
    __syncthreads();
    bool reuse = false;
    if (dyn.idx != nullptr) {
        int same = L.flag[2];
        for (int k = tid; k < N; k += nt) same = same && (L.goffg[k] == L.idxl[k]);
        if (tid == 0) L.flag[3] = 1;
        __syncthreads();
        if (!same) L.flag[3] = 0;
        __syncthreads();
        reuse = L.flag[3] != 0;
        __syncthreads();
        if (tid == 0) L.flag[3] = 0;                           // tile_cholesky_set's early-tile counter from here on
    }
    if (!reuse) {
        rollout<MSEL, NSEL, false, 4>(d, dyn, q.x0, (cgptr) nullptr, w.x, L);
        condense_half<MSEL, NSEL>(d, c, dyn, w.x, gh, L);
        for (int k = tid; k < N; k += nt) L.goffg[k] = L.idxl[k];
        if (tid == 0) L.flag[2] = 1;
        __syncthreads();
    }
    if (warm) {                                               // y = y_free + G u
        __syncthreads();
        g_times<MSEL>(d, g, L, L.u, L.dy);
        for (int e = tid; e < ldG; e += nt) { L.y[e] = L.yfg[e] + L.dy[e]; }
        __syncthreads();
        for (int e = tid; e < ldG; e += nt) L.dy[e] = 0.0;
    } else {
        for (int e = tid; e < ldG; e += nt) { L.y[e] = L.yfg[e]; L.dy[e] = 0.0; }
    }
    __syncthreads();
    // ---- this thread's rows: the two box rows of input `tid`, and one state-row slot
    const bool isu = tid < nm;
    const bool isx = tid < N * GX;
    const int ku = isu ? tid / m : 0, bu = isu ? tid - ku * m : 0;
    const int kx = isx ? tid / GX + 1 : 1, rx = isx ? tid - (kx - 1) * GX : 0;        // stage 1..N, row slot
    const int nrk = d.nX + (kx == N ? d.nXf : 0);
    const bool xrow = isx && rx < nrk;                                               // the slot holds a state row
    const bool xlead = isx && rx == 0;                                               // first lane of the stage group
    double cu[2] = {0.0, 0.0}, hu[2] = {0.0, 0.0}, cx[2] = {0.0, 0.0}, hx = 0.0;
    if (isu) {
        cu[0] = c.UA[(size_t)(2 * bu) * m + bu]; cu[1] = c.UA[(size_t)(2 * bu + 1) * m + bu];
        hu[0] = c.Ub[2 * bu]; hu[1] = c.Ub[2 * bu + 1];
    }
    if (xrow) {
        cgptr T = rx < d.nX ? c.Tx + (size_t)rx * 2 : c.Txf + (size_t)(rx - d.nX) * 2;
        cx[0] = T[0]; cx[1] = T[1];
        hx = rx < d.nX ? c.Xb[rx] : c.Xfb[rx - d.nX];
    }
    const double r2bb = isu ? c.R2[bu * m + bu] : 0.0;
    const double udv = (isu && q.ud) ? q.ud[(size_t)ku * m + bu] : 0.0;
    double gc0 = 0.0, gc1 = 0.0, s00 = 0.0, s01 = 0.0, s11 = 0.0;
    if (xlead) {
        cgptr S = (kx == N) ? c.ScN : c.Sc;
        s00 = S[0]; s01 = S[1]; s11 = S[3];
        if (q.z) for (int b = 0; b < nz; ++b) { gc0 = fma(-c.Cz2[b], q.z[(size_t)kx * nz + b], gc0); gc1 = fma(-c.Cz2[nz + b], q.z[(size_t)kx * nz + b], gc1); }
        if (kx == N && c.Qzf && q.zf) for (int b = 0; b < nz; ++b) { gc0 = fma(-c.Czf2[b], q.zf[b], gc0); gc1 = fma(-c.Czf2[nz + b], q.zf[b], gc1); }
    }
    const int lsu = 2 * tid, lsx = 2 * nm + (kx - 1) * (d.nX + d.nXf) + rx;           // this thread's rows in w.lam (layout of ipm_box)
    double tu[2] = {0.0, 0.0}, lu[2] = {0.0, 0.0}, gu[2] = {0.0, 0.0}, ru[2] = {0.0, 0.0}, dtu[2] = {0.0, 0.0}, dlu[2] = {0.0, 0.0};
    double txr = 0.0, lxr = 0.0, gxr = 0.0, rcx = 0.0, dtx = 0.0, dlx = 0.0;
    auto yval = [&](clptr vy) -> double { return fma(cx[1], vy[(kx - 1) * 2 + 1], cx[0] * vy[(kx - 1) * 2]); };
    int status = 1, it = 0;
    enum { INIT = 0, PRED = 1, CORR = 2 };
    int mode = INIT;
    double mu = 0.0, rp = 0.0, sig = 0.0, sd = 1.0, sp = 1.0, dreg = 0.0;
    bool near_opt = false;
    const bool ya_const = d.nX + d.nXf == 0;                  // no state rows: L.ya is the same in the predictor and the corrector (newton_front)
    auto scales = [&]() {                                     // residual scales and the dual regularisation (once per QP)
        for (int e = tid; e < d.n; e += nt) {
            double gq = 0.0;
            if (q.z) for (int a = 0; a < nz; ++a) gq = fma(c.HtQz2[e * nz + a], -q.z[nz + a], gq);
            sd = fmax(sd, fabs(gq));
        }
        for (int e = tid; e < d.nU; e += nt) sp = fmax(sp, fabs(c.Ub[e]));
        reduce2(sd, 1, sp, 1, L.red);
        sd = fmax(sd, q.omega);
        sp = fmax(sp, fabs(q.delta));
        dreg = d.reg / sd;
    };
    if (warm && d.ng > 0) {
        if (isu) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                tu[s2] = fmax(-(cu[s2] * L.u[tid] - hu[s2]), WARM_FLOOR);
                lu[s2] = poison ? INFINITY : fmax(w.lam[lsu + s2], WARM_FLOOR);
            }
        }
        if (xrow) {
            txr = fmax(-(yval(L.y) - hx), WARM_FLOOR);
            lxr = poison ? INFINITY : fmax(w.lam[lsx], WARM_FLOOR);
        }
        scales();
        mode = PRED;
    }
    while (true) {
        tid = SRH_TID;
        // ---------------- rows -> weights, gradient shifts, and their per-stage sums, all in place
        double musum = 0.0, rpm = 0.0;
        double Du[2] = {0.0, 0.0}, rhu[2] = {0.0, 0.0}, Dx = 0.0, rhx = 0.0;
        if (isu) {
            const double uv = L.u[tid];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                if (mode == INIT) {
                    Du[s2] = 1.0; rhu[s2] = cu[s2] * uv - hu[s2]; lu[s2] = 0.0;
                } else if (mode == PRED) {
                    const double gq = cu[s2] * uv - hu[s2], t = tu[s2], lam = lu[s2], rg = gq + t;
                    gu[s2] = rg;
                    Du[s2] = lam / (t + dreg * lam);
                    rhu[s2] = Du[s2] * (rg + dreg * lam);
                    musum += lam * t;
                    rpm = fmax(rpm, fabs(rg));
                } else {
                    const double t = tu[s2], lam = lu[s2], rc = lam * t + dtu[s2] * dlu[s2] - sig * mu;
                    ru[s2] = rc;
                    rhu[s2] = lam + (lam * gu[s2] - rc) / (t + dreg * lam);
                }
            }
        }
        if (xrow) {
            if (mode == INIT) {
                Dx = 1.0; rhx = yval(L.y) - hx; lxr = 0.0;
            } else if (mode == PRED) {
                const double gq = yval(L.y) - hx, t = txr, lam = lxr, rg = gq + t;
                gxr = rg;
                Dx = lam / (t + dreg * lam);
                rhx = Dx * (rg + dreg * lam);
                musum += lam * t;
                rpm = fmax(rpm, fabs(rg));
            } else {
                const double t = txr, lam = lxr, rc = lam * t + dtx * dlx - sig * mu;
                rcx = rc;
                rhx = lam + (lam * gxr - rc) / (t + dreg * lam);
            }
        }
        if (isu) {
            const double du0 = r2bb * (L.u[tid] - udv);
            if (mode != CORR) {
                double v = fma(cu[0] * Du[0], cu[0], r2bb);
                v = fma(cu[1] * Du[1], cu[1], v);
                L.Ldi[tid] = 1.0 / sqrt(v);
            }
            L.ta[tid] = fma(cu[1], rhu[1], fma(cu[0], rhu[0], du0));
            if (mode == PRED) L.tb[tid] = fma(cu[1], lu[1], fma(cu[0], lu[0], du0));
        }
        if (isx) {                                           // whole stage groups: the DPP sums see every lane of a group
            const double y0 = L.y[(kx - 1) * 2], y1 = L.y[(kx - 1) * 2 + 1];
            if (mode != CORR) {
                const double a00 = gsum<GX>(cx[0] * Dx * cx[0]), a01 = gsum<GX>(cx[0] * Dx * cx[1]), a11 = gsum<GX>(cx[1] * Dx * cx[1]);
                if (xlead) {
                    const double S00 = s00 + a00, S01 = s01 + a01, S11 = s11 + a11;
                    const double dmax = fmax(fabs(S00), fabs(S11));
                    const double l00 = S00 > 1e-14 * dmax ? sqrt(S00) : 0.0;
                    const double l10 = l00 > 0.0 ? S01 / l00 : 0.0;
                    const double v = fma(-l10, l10, S11);
                    const double l11 = v > 1e-14 * dmax ? sqrt(v) : 0.0;
                    lptr Lk = L.Ls + (size_t)(kx - 1) * 4;
                    Lk[0] = l00; Lk[1] = 0.0; Lk[2] = l10; Lk[3] = l11;
                }
            }
            const double r0 = gsum<GX>(cx[0] * rhx), r1 = gsum<GX>(cx[1] * rhx);
            double l0 = 0.0, l1 = 0.0;
            if (mode == PRED) { l0 = gsum<GX>(cx[0] * (xrow ? lxr : 0.0)); l1 = gsum<GX>(cx[1] * (xrow ? lxr : 0.0)); }
            if (xlead) {
                const double c0 = fma(s01, y1, s00 * y0) + gc0, c1 = fma(s11, y1, s01 * y0) + gc1;
                L.ya[(kx - 1) * 2] = c0 + r0; L.ya[(kx - 1) * 2 + 1] = c1 + r1;
                if (mode == PRED) { L.yg[(kx - 1) * 2] = c0 + l0; L.yg[(kx - 1) * 2 + 1] = c1 + l1; }
            }
        }
        if (mode == PRED) {
            reduce2(musum, 0, rpm, 1, L.red);
            mu = musum / d.ng;
            rp = rpm;
        }
        __syncthreads();
        // ---------------- Newton system
        double rd = 0.0;
        bool ok = true;
        if (mode != CORR) {
            gram<MSEL>(d, c, g, L);
            // the factorisation (waves 0-1: a chain of one-wave 16 x 16 factorisations) beside the half of the Newton solve that
            // does not need the factor (waves 2-3), see Waves
            const clptr gyd = mode == PRED ? (clptr)L.yg : (clptr) nullptr;
            const int sw = serial_wave(L);
            const int wv = __builtin_amdgcn_readfirstlane(tid >> 6) ^ sw;
            auto W = half_waves(L.flag + 4, sw);
            if (half_of_wave(wv) == 0) tile_cholesky_set(d, L, W, L.flag + 5, L.flag + 3);
            else newton_front<MSEL, true>(d, g, L, gyd, W, pf, ya_const && mode == PRED);
            __syncthreads();
            ok = L.flag[1] != 0;
            if (tid < 5) L.flag[3 + tid] = 0;
            if (gyd) rd = L.Qu[0];
            if (ok) unit_tiles(d, L);
            if (ok) newton_back<MSEL>(d, g, L, pf);
        } else {
            newton_solve<MSEL>(d, g, L, (clptr) nullptr, &rd, pf, ya_const);
        }
        // ---------------- use the direction
        if (mode == INIT) {
            if (!ok) { status = 2; break; }
            for (int e = tid; e < nm; e += nt) L.u[e] += L.du[e];
            for (int e = tid; e < ldG; e += nt) L.y[e] += L.dy[e];
            __syncthreads();
            if (d.ng == 0) { status = 0; break; }
            double zmin = INFINITY, zmax = -INFINITY;
            if (isu) {
                const double uv = L.u[tid];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) { gu[s2] = cu[s2] * uv - hu[s2]; zmin = fmin(zmin, gu[s2]); zmax = fmax(zmax, gu[s2]); }
            }
            if (xrow) { gxr = yval(L.y) - hx; zmin = fmin(zmin, gxr); zmax = fmax(zmax, gxr); }
            reduce2(zmin, 2, zmax, 1, L.red);
            const double sh_t = zmax >= 0.0 ? 1.0 + zmax : 0.0, sh_l = zmin <= 0.0 ? 1.0 - zmin : 0.0;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { tu[s2] = -gu[s2] + sh_t; lu[s2] = gu[s2] + sh_l; }
            txr = -gxr + sh_t; lxr = gxr + sh_l;
            scales();
            mode = PRED;
            continue;
        }
        double amax = 1e300, dummy = 0.0;
        if (ok) {
            if (isu) {
                const double duv = L.du[tid];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const double t = tu[s2], lam = lu[s2], rga = gu[s2] + cu[s2] * duv;
                    const double dl = ((mode == PRED ? -lam * t : -ru[s2]) + lam * rga) / (t + dreg * lam);
                    const double dtv = -rga + dreg * dl;
                    dlu[s2] = dl; dtu[s2] = dtv;
                    if (dtv < 0.0) amax = fmin(amax, -t / dtv);
                    if (dl < 0.0) amax = fmin(amax, -lam / dl);
                }
            }
            if (xrow) {
                const double t = txr, lam = lxr, rga = gxr + yval(L.dy);
                const double dl = ((mode == PRED ? -lam * t : -rcx) + lam * rga) / (t + dreg * lam);
                const double dtv = -rga + dreg * dl;
                dlx = dl; dtx = dtv;
                if (dtv < 0.0) amax = fmin(amax, -t / dtv);
                if (dl < 0.0) amax = fmin(amax, -lam / dl);
            }
        }
        reduce2(amax, 2, dummy, 0, L.red);
        if (mode == PRED) {
            if (!ok) { status = near_opt ? 0 : 2; break; }
            if (!(mu == mu)) { status = near_opt ? 0 : 5; break; }
            if (!(rd == rd)) { status = near_opt ? 0 : 6; break; }
            const double ltol = fmax(d.tol, 1e-9);
            if (rd <= ltol * sd && rp <= ltol * sp && mu <= d.tol) { status = 0; break; }
            near_opt = (rd <= 1e-8 * sd && rp <= 1e-8 * sp && mu <= 1e-8);
            if (it >= d.max_iter) { status = 1; break; }
            const double a_aff = fmin(1.0, amax);
            double ma = 0.0;
            if (isu) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) ma += (lu[s2] + a_aff * dlu[s2]) * (tu[s2] + a_aff * dtu[s2]);
            }
            if (xrow) ma += (lxr + a_aff * dlx) * (txr + a_aff * dtx);
            reduce2(ma, 0, dummy, 0, L.red);
            const double mu_aff = ma / d.ng;
            sig = mu > 0.0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
            mode = CORR;
            continue;
        }
        if (!ok) { status = 2; break; }
        const double a = fmin(1.0, 0.99 * amax);
        for (int e = tid; e < nm; e += nt) L.u[e] += a * L.du[e];
        for (int e = tid; e < ldG; e += nt) L.y[e] += a * L.dy[e];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { tu[s2] += a * dtu[s2]; lu[s2] += a * dlu[s2]; }
        txr += a * dtx; lxr += a * dlx;
        __syncthreads();
        ++it;
        mode = PRED;
    }
    __syncthreads();
    for (int e = tid; e < nm; e += nt) w.u[e] = L.u[e];
    if (status == 0) {                                        // what the next QP of this solve starts from (warm)
        if (isu) { w.lam[lsu] = lu[0]; w.lam[lsu + 1] = lu[1]; }
        if (xrow) w.lam[lsx] = lxr;
    }
    __syncthreads();
    if (iters_out) *iters_out = it;
    return status;
}

// ------------------------------------------------------------------ short horizons: the interior point on ONE wave
// The reference's closed-loop drivers replan N = 5 (examples/diamond/diamond.py:309-316) and N = 3 (examples/hardware/diamond.py:
// 393-399): N p_o <= 16 outputs -- K is ONE 16 x 16 tile -- and N n_u <= 64 inputs.  ipm_box() spends ~80 k clocks per interior-point
// iteration on such a problem (profiles/r05_lean_phase_clocks.json: N = 5) although nothing in it is larger than a tile: some 35
// workgroup barriers, each with seven waves that have nothing to do.  Here the iteration runs on wave 0 alone, without a single
// workgroup barrier (the other waves wait at the one behind it):
//   * lane e < N m owns input e = (stage, input): its value, its two box rows and their slacks / multipliers, its row of G^T
//     (16 registers) -- G^T y is a lane-local dot product with the 16 broadcast entries of y;
//   * lane l < N GX also owns a state-row slot (stage, slot) like the state-row threads of ipm_box (per-stage sums by DPP);
//   * lane (i, kk) of 16 x 4 holds G in the MFMA layout G^T[4 s + kk][i], s < N m / 4: G u = one FMA per k-step + the sum over kk,
//     and Ky = G D^-1 G^T is N m / 4 MFMA instructions whose A and B operands are THE SAME registers;
//   * y-space vectors live in LDS (16 entries), exchanged under a wave-level fence (the LDS queue of a wave is in order).
// Same iteration as ipm_box (same starting points, weights, stopping rule, warm start): sums in another order, rounding-level
// differences.  Requirements (lean_matches): KT == 1, N m <= 64, N GX <= 64, the whole packed G in LDS (lean_j0 == 0).
// what one wave needs between an LDS write and the read of it by another lane: nothing from the hardware (the LDS queue of a
// wave is in order), only that the compiler keeps the accesses on their sides -- a wavefront-scope fence waits for no counter
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MSEL, int NSEL, int GX>
__device__ __forceinline__ int ipm_wave(const QPDims &dfull, const QPConst &c, const QPDyn &dyn, const QPData &q, gptr work_base,
                                        Lds &L, int *iters_out, QPWork &wout, long long *prof, int warm_mode = 0) {
    static_assert(MSEL == 4 || MSEL == 8, "one-wave interior point: n_u = 4 or 8");
    const bool warm = warm_mode != 0, poison = warm_mode == 2;
    int tid = SRH_TID;                                       // re-read at the top of every interior-point iteration (dev_la.h: SRH_TID)
    const int nt = blockDim.x;
    QPDims d = dfull;
    d.tr = 0;
    d.nrx = d.nX;
    d.RX = d.nrx + d.nXf;
    d.NR = d.N * d.RX + d.N * d.nU;
    d.ng = d.N * d.nrx + d.nXf + d.N * d.nU;
    QPWork w;
    qp_carve(w, work_base, d);
    wout = w;
    gptr gh = work_base + dfull.qc_off;
    constexpr int M = MSEL, SMAX = 2 * M;                      // k-steps of a product with G: N m / 4 <= 8 m / 4 (N <= 8)
    const int N = d.N, nm = N * M, ldG = 16, NP = 2 * N, nz = d.nz;
#ifdef SRH_PROFILE
    long long tq_last = clock64();
    auto qlap = [&](int slot) { const long long now = clock64(); prof[slot] += now - tq_last; tq_last = now; };
#define QW_LAP(x) qlap(x)
#else
#define QW_LAP(x) ((void)0)
#endif
    // ---- the preamble of ipm_box: starting point, condensation (all waves)
    constexpr double WARM_FLOOR = 1e-2;
    if (warm) { for (int e = tid; e < nm; e += nt) L.u[e] = w.u[e]; }
    else { for (int e = tid; e < nm; e += nt) { w.u[e] = 0.0; L.u[e] = 0.0; } }
    for (int e = tid; e < ldG + YPAD; e += nt) { L.ya[e] = 0.0; L.yd[e] = 0.0; L.yg[e] = 0.0; }
    for (int e = tid; e <= N; e += nt) w.s[e] = 0.0;
    for (int k = tid; k < N; k += nt) L.idxl[k] = dyn.idx ? dyn.idx[k] : k;
    __syncthreads();
    bool reuse = false;
    if (dyn.idx != nullptr) {
        int same = L.flag[2];
        for (int k = tid; k < N; k += nt) same = same && (L.goff[k] == L.idxl[k]);
        if (tid == 0) L.flag[3] = 1;
        __syncthreads();
        if (!same) L.flag[3] = 0;
        __syncthreads();
        reuse = L.flag[3] != 0;
    }
    if (!reuse) {
        rollout<MSEL, NSEL>(d, dyn, q.x0, (cgptr) nullptr, w.x, L);
        QW_LAP(0);
        condense<MSEL, NSEL>(d, c, dyn, w.x, gh, L);
        for (int k = tid; k < N; k += nt) L.goff[k] = L.idxl[k];
        if (tid == 0) L.flag[2] = 1;
    }
    __syncthreads();
    QW_LAP(2);
    int status = 1, it = 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wave == 0) {
        int lane = tid, i16 = lane & 15, kk = lane >> 4;          // re-derived at the top of every iteration (SRH_TID)
        const int S = nm >> 2;                                     // k-steps of the products with G (n_u is a multiple of 4)
        // ---- G in registers, twice: by input row (gR) and in the MFMA operand layout (gM)
        const bool isu = lane < nm;
        const int ku = isu ? lane / M : 0, bu = isu ? lane - ku * M : 0;
        auto g_at = [&](int e, int i) -> double {                 // G^T[e][i] from the packed rows (structural zeros: i < 2 j)
            const int j = e / M, b = e - j * M;
            return (i >= 2 * j && i < NP) ? L.Gt[goff(j, M, NP) + b * (NP - 2 * j) + (i - 2 * j)] : 0.0;
        };
        double gR[16], gM[SMAX];
#pragma unroll
        for (int i = 0; i < 16; ++i) gR[i] = isu ? g_at(lane, i) : 0.0;
#pragma unroll
        for (int s2 = 0; s2 < SMAX; ++s2) gM[s2] = s2 < S ? g_at(4 * s2 + kk, i16) : 0.0;
        // ---- this lane's rows: the two box rows of input `lane`, and one state-row slot
        const bool isx = lane < N * GX;
        const int kx = isx ? lane / GX + 1 : 1, rx = isx ? lane - (kx - 1) * GX : 0;
        const int nrk = d.nX + (kx == N ? d.nXf : 0);
        const bool xrow = isx && rx < nrk, xlead = isx && rx == 0;
        double cu[2] = {0.0, 0.0}, hu[2] = {0.0, 0.0}, cx[2] = {0.0, 0.0}, hx = 0.0;
        if (isu) {
            cu[0] = c.UA[(size_t)(2 * bu) * M + bu]; cu[1] = c.UA[(size_t)(2 * bu + 1) * M + bu];
            hu[0] = c.Ub[2 * bu]; hu[1] = c.Ub[2 * bu + 1];
        }
        if (xrow) {
            cgptr T = rx < d.nX ? c.Tx + (size_t)rx * 2 : c.Txf + (size_t)(rx - d.nX) * 2;
            cx[0] = T[0]; cx[1] = T[1];
            hx = rx < d.nX ? c.Xb[rx] : c.Xfb[rx - d.nX];
        }
        const double r2bb = isu ? c.R2[bu * M + bu] : 0.0;
        const double udv = (isu && q.ud) ? q.ud[(size_t)ku * M + bu] : 0.0;
        double gc0 = 0.0, gc1 = 0.0, s00 = 0.0, s01 = 0.0, s11 = 0.0;
        if (xlead) {
            cgptr Sm = (kx == N) ? c.ScN : c.Sc;
            s00 = Sm[0]; s01 = Sm[1]; s11 = Sm[3];
            if (q.z) for (int b = 0; b < nz; ++b) { gc0 = fma(-c.Cz2[b], q.z[(size_t)kx * nz + b], gc0); gc1 = fma(-c.Cz2[nz + b], q.z[(size_t)kx * nz + b], gc1); }
            if (kx == N && c.Qzf && q.zf) for (int b = 0; b < nz; ++b) { gc0 = fma(-c.Czf2[b], q.zf[b], gc0); gc1 = fma(-c.Czf2[nz + b], q.zf[b], gc1); }
        }
        const int lsu = 2 * lane, lsx = 2 * nm + (kx - 1) * (d.nX + d.nXf) + rx;           // this lane's rows in w.lam (layout of ipm_box)
        double u_r = isu ? L.u[lane] : 0.0, du_r = 0.0, ldi = 0.0, w2 = 0.0, ta_r = 0.0, tb_r = 0.0;
        double tu[2] = {0.0, 0.0}, lu[2] = {0.0, 0.0}, gu[2] = {0.0, 0.0}, ru[2] = {0.0, 0.0}, dtu[2] = {0.0, 0.0}, dlu[2] = {0.0, 0.0};
        double tx = 0.0, lx = 0.0, gx = 0.0, rcx = 0.0, dtx = 0.0, dlx = 0.0;
        // y = y_free (+ G u when warm), dy = 0: lanes (i, kk) with the sum over kk
        auto g_times_w = [&](clptr uv) -> double {              // (G uv)[i16] on every lane; uv: N m entries of LDS
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < SMAX; s2 += 2) {
                if (s2 < S) a0 = fma(gM[s2], uv[4 * s2 + kk], a0);
                if (s2 + 1 < S) a1 = fma(gM[s2 + 1], uv[4 * (s2 + 1) + kk], a1);
            }
            double a = a0 + a1;
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            return a;
        };
        {
            double y0 = L.yf[i16];
            if (warm) y0 += g_times_w(L.u);
            if (lane < 16) { L.y[lane] = lane < NP ? y0 : 0.0; L.dy[lane] = 0.0; }
        }
        wave_fence();
        auto yval = [&](clptr vy) -> double { return fma(cx[1], vy[(kx - 1) * 2 + 1], cx[0] * vy[(kx - 1) * 2]); };
        enum { INIT = 0, PRED = 1, CORR = 2 };
        int mode = INIT;
        double mu = 0.0, rp = 0.0, sig = 0.0, sd = 1.0, sp = 1.0, dreg = 0.0;
        bool near_opt = false;
        auto scales = [&]() {
            for (int e = lane; e < d.n; e += 64) {
                double gq = 0.0;
                if (q.z) for (int a = 0; a < nz; ++a) gq = fma(c.HtQz2[e * nz + a], -q.z[nz + a], gq);
                sd = fmax(sd, fabs(gq));
            }
            for (int e = lane; e < d.nU; e += 64) sp = fmax(sp, fabs(c.Ub[e]));
            sd = fmax(wg::wave_max(sd), q.omega);
            sp = fmax(wg::wave_max(sp), fabs(q.delta));
            dreg = d.reg / sd;
        };
        if (warm && d.ng > 0) {
            if (isu) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    tu[s2] = fmax(-(cu[s2] * u_r - hu[s2]), WARM_FLOOR);
                    lu[s2] = poison ? INFINITY : fmax(w.lam[lsu + s2], WARM_FLOOR);
                }
            }
            if (xrow) {
                tx = fmax(-(yval(L.y) - hx), WARM_FLOOR);
                lx = poison ? INFINITY : fmax(w.lam[lsx], WARM_FLOOR);
            }
            scales();
            mode = PRED;
        }
        while (true) {
            tid = SRH_TID; lane = tid; i16 = lane & 15; kk = lane >> 4;
            // ---------------- rows -> weights, gradient shifts, per-stage sums
            double musum = 0.0, rpm = 0.0;
            double Du[2] = {0.0, 0.0}, rhu[2] = {0.0, 0.0}, Dx = 0.0, rhx = 0.0;
            if (isu) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (mode == INIT) {
                        Du[s2] = 1.0; rhu[s2] = cu[s2] * u_r - hu[s2]; lu[s2] = 0.0;
                    } else if (mode == PRED) {
                        const double gq = cu[s2] * u_r - hu[s2], t = tu[s2], lam = lu[s2], rg = gq + t;
                        gu[s2] = rg;
                        Du[s2] = lam / (t + dreg * lam);
                        rhu[s2] = Du[s2] * (rg + dreg * lam);
                        musum += lam * t;
                        rpm = fmax(rpm, fabs(rg));
                    } else {
                        const double t = tu[s2], lam = lu[s2], rc = lam * t + dtu[s2] * dlu[s2] - sig * mu;
                        ru[s2] = rc;
                        rhu[s2] = lam + (lam * gu[s2] - rc) / (t + dreg * lam);
                    }
                }
                const double du0 = r2bb * (u_r - udv);
                if (mode != CORR) {
                    double v = fma(cu[0] * Du[0], cu[0], r2bb);
                    v = fma(cu[1] * Du[1], cu[1], v);
                    ldi = 1.0 / sqrt(v);
                    w2 = ldi * ldi;
                    L.tc[lane] = w2;                             // for the A operands of the Gram product
                }
                ta_r = fma(cu[1], rhu[1], fma(cu[0], rhu[0], du0));
                if (mode == PRED) tb_r = fma(cu[1], lu[1], fma(cu[0], lu[0], du0));
            }
            if (xrow) {
                if (mode == INIT) {
                    Dx = 1.0; rhx = yval(L.y) - hx; lx = 0.0;
                } else if (mode == PRED) {
                    const double gq = yval(L.y) - hx, t = tx, lam = lx, rg = gq + t;
                    gx = rg;
                    Dx = lam / (t + dreg * lam);
                    rhx = Dx * (rg + dreg * lam);
                    musum += lam * t;
                    rpm = fmax(rpm, fabs(rg));
                } else {
                    const double t = tx, lam = lx, rc = lam * t + dtx * dlx - sig * mu;
                    rcx = rc;
                    rhx = lam + (lam * gx - rc) / (t + dreg * lam);
                }
            }
            {   // whole stage groups (every lane executes the DPP sums; lanes without a row contribute zeros)
                const double y0 = isx ? L.y[(kx - 1) * 2] : 0.0, y1 = isx ? L.y[(kx - 1) * 2 + 1] : 0.0;
                if (mode != CORR) {
                    const double a00 = gsum<GX>(cx[0] * Dx * cx[0]), a01 = gsum<GX>(cx[0] * Dx * cx[1]), a11 = gsum<GX>(cx[1] * Dx * cx[1]);
                    if (xlead) {
                        const double S00 = s00 + a00, S01 = s01 + a01, S11 = s11 + a11;
                        const double dmax = fmax(fabs(S00), fabs(S11));
                        const double l00 = S00 > 1e-14 * dmax ? sqrt(S00) : 0.0;
                        const double l10 = l00 > 0.0 ? S01 / l00 : 0.0;
                        const double v = fma(-l10, l10, S11);
                        const double l11 = v > 1e-14 * dmax ? sqrt(v) : 0.0;
                        lptr Lk = L.Ls + (size_t)(kx - 1) * 4;
                        Lk[0] = l00; Lk[1] = 0.0; Lk[2] = l10; Lk[3] = l11;
                    }
                }
                const double r0 = gsum<GX>(cx[0] * rhx), r1 = gsum<GX>(cx[1] * rhx);
                double l0 = 0.0, l1 = 0.0;
                if (mode == PRED) { l0 = gsum<GX>(cx[0] * lx); l1 = gsum<GX>(cx[1] * lx); }
                if (xlead) {
                    const double c0 = fma(s01, y1, s00 * y0) + gc0, c1 = fma(s11, y1, s01 * y0) + gc1;
                    L.ya[(kx - 1) * 2] = c0 + r0; L.ya[(kx - 1) * 2 + 1] = c1 + r1;
                    if (mode == PRED) { L.yg[(kx - 1) * 2] = c0 + l0; L.yg[(kx - 1) * 2 + 1] = c1 + l1; }
                }
            }
            if (mode == PRED) {
                mu = wg::wave_sum(musum) / d.ng;
                rp = wg::wave_max(rpm);
            }
            wave_fence();
            // ---------------- K = I + Ls^T (G D^-1 G^T) Ls, scaled to a unit diagonal, and its factor (one tile)
            bool ok = true;
            if (mode != CORR) {
                wg::qp_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s2 = 0; s2 < SMAX; ++s2)
                    if (s2 < S) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(gM[s2] * L.tc[4 * s2 + kk], gM[s2], acc, 0, 0, 0);
                const int kb = min(i16 >> 1, N - 1), bc = i16 & 1;
                clptr Lb = L.Ls + (size_t)kb * 4;
                const double cb_own = bc == 0 ? Lb[0] : Lb[3], cb_oth = bc == 0 ? Lb[2] : 0.0;
                double kv[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int r = kk + 4 * qd, ka = min(r >> 1, N - 1), ar = r & 1;
                    clptr La = L.Ls + (size_t)ka * 4;
                    double v = (r < NP && i16 < NP) ? acc[qd] : 0.0;
                    const double vp = wg::dpp_mov<0xB1>(v);
                    v = fma(vp, cb_oth, v * cb_own);
                    const double vr = wg::xor16(v);
                    v = ar == 0 ? fma(La[2], vr, La[0] * v) : La[3] * v;
                    const bool dg = r == i16;
                    v += dg ? 1.0 : 0.0;
                    const double ri = qpc::rsq3(dg ? v : 1.0);
                    if (dg) L.ks[r] = ri;
                    kv[qd] = v;
                }
                wave_fence();
                const double sc = L.ks[i16];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) { const int r = kk + 4 * qd; L.B[r * TS + i16] = kv[qd] * (L.ks[r] * sc); }
                wave_fence();
                // (the padding rows / columns of a short horizon are the identity: only the leading N p_o pivots do anything)
                ok = NP <= 6 ? qpc::chol16<false, 6>(L.B, L.Rinv) : (NP <= 10 ? qpc::chol16<false, 10>(L.B, L.Rinv) : qpc::chol16<false>(L.B, L.Rinv));
                wave_fence();
            }
            // ---------------- Newton direction
            double rd = 0.0;
            if (ok) {
                double gty = 0.0, gtd = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) gty = fma(gR[i], L.ya[i], gty);
                if (mode == PRED) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) gtd = fma(gR[i], L.yg[i], gtd);
                    rd = wg::wave_max(isu ? fabs(tb_r + gtd) : 0.0);
                }
                const double tt = -(ta_r + gty) * w2;
                if (isu) L.ta[lane] = tt;
                wave_fence();
                const double yb = g_times_w(L.ta);
                {   // yc = ks Ls^T yb per output stage
                    const double yo = wg::dpp_mov<0xB1>(yb);
                    const int k = i16 >> 1;
                    double o = 0.0;
                    if (k < N) {
                        clptr Lk = L.Ls + (size_t)k * 4;
                        o = (i16 & 1) == 0 ? fma(Lk[2], yo, Lk[0] * yb) * L.ks[i16] : Lk[3] * yb * L.ks[i16];
                    }
                    if (lane < 16) L.yc[lane] = o;
                }
                wave_fence();
                {   // v = K^-1 yc = Rinv (Rinv^T yc)   (k_solve_unit, one tile)
                    const int cc = lane >> 2, part = lane & 3;
                    double t = 0.0;
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) t = fma(L.Rinv[(4 * part + kq) * TS + cc], L.yc[4 * part + kq], t);
                    t = wg::group_sum<4>(t);
                    wave_fence();
                    if (part == 0) L.yc[cc] = t;
                    wave_fence();
                    double wv = 0.0;
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) wv = fma(L.Rinv[cc * TS + 4 * part + kq], L.yc[4 * part + kq], wv);
                    wv = wg::group_sum<4>(wv);
                    wave_fence();
                    if (part == 0) L.yc[cc] = wv;
                    wave_fence();
                }
                if (lane < 8) {                                   // yd = Ls (ks v), dy = Ls^-T (ks v) per output stage
                    const int k = lane;
                    double o0 = 0.0, o1 = 0.0, e0 = 0.0, e1 = 0.0;
                    if (k < N) {
                        clptr Lk = L.Ls + (size_t)k * 4;
                        const double v0 = L.yc[2 * k] * L.ks[2 * k], v1 = L.yc[2 * k + 1] * L.ks[2 * k + 1];
                        o0 = Lk[0] * v0;
                        o1 = fma(Lk[3], v1, Lk[2] * v0);
                        e1 = v1 / Lk[3];
                        e0 = fma(-Lk[2], e1, v0) / Lk[0];
                    }
                    L.yd[2 * k] = o0; L.yd[2 * k + 1] = o1;
                    if (d.ls_pd) { L.dy[2 * k] = e0; L.dy[2 * k + 1] = e1; }
                }
                wave_fence();
                double gyd = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) gyd = fma(gR[i], L.yd[i], gyd);
                du_r = tt - gyd * w2;
                if (!d.ls_pd) {                                   // dy = G du
                    if (isu) L.ta[lane] = du_r;
                    wave_fence();
                    const double dyv = g_times_w(L.ta);
                    if (lane < 16) L.dy[lane] = lane < NP ? dyv : 0.0;
                    wave_fence();
                }
            }
            // ---------------- use the direction
            if (mode == INIT) {
                if (!ok) { status = 2; break; }
                u_r += du_r;
                if (lane < 16) L.y[lane] += L.dy[lane];
                wave_fence();
                if (d.ng == 0) { status = 0; break; }
                double zmin = INFINITY, zmax = -INFINITY;
                if (isu) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) { gu[s2] = cu[s2] * u_r - hu[s2]; zmin = fmin(zmin, gu[s2]); zmax = fmax(zmax, gu[s2]); }
                }
                if (xrow) { gx = yval(L.y) - hx; zmin = fmin(zmin, gx); zmax = fmax(zmax, gx); }
                zmin = wg::wave_min(zmin); zmax = wg::wave_max(zmax);
                const double sh_t = zmax >= 0.0 ? 1.0 + zmax : 0.0, sh_l = zmin <= 0.0 ? 1.0 - zmin : 0.0;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) { tu[s2] = -gu[s2] + sh_t; lu[s2] = gu[s2] + sh_l; }
                tx = -gx + sh_t; lx = gx + sh_l;
                scales();
                mode = PRED;
                continue;
            }
            double amax = 1e300;
            if (ok) {
                if (isu) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const double t = tu[s2], lam = lu[s2], rga = gu[s2] + cu[s2] * du_r;
                        const double dl = ((mode == PRED ? -lam * t : -ru[s2]) + lam * rga) / (t + dreg * lam);
                        const double dtv = -rga + dreg * dl;
                        dlu[s2] = dl; dtu[s2] = dtv;
                        if (dtv < 0.0) amax = fmin(amax, -t / dtv);
                        if (dl < 0.0) amax = fmin(amax, -lam / dl);
                    }
                }
                if (xrow) {
                    const double t = tx, lam = lx, rga = gx + yval(L.dy);
                    const double dl = ((mode == PRED ? -lam * t : -rcx) + lam * rga) / (t + dreg * lam);
                    const double dtv = -rga + dreg * dl;
                    dlx = dl; dtx = dtv;
                    if (dtv < 0.0) amax = fmin(amax, -t / dtv);
                    if (dl < 0.0) amax = fmin(amax, -lam / dl);
                }
            }
            amax = wg::wave_min(amax);
            if (mode == PRED) {
                if (!ok) { status = near_opt ? 0 : 2; break; }
                if (!(mu == mu)) { status = near_opt ? 0 : 5; break; }
                if (!(rd == rd)) { status = near_opt ? 0 : 6; break; }
                if (q.dbg && lane == 0) { gptr gd = q.dbg + 8 * it; gd[0] = mu; gd[1] = rd; gd[2] = rp; gd[3] = sd; gd[4] = sp; }
                const double ltol = fmax(d.tol, 1e-9);
                if (rd <= ltol * sd && rp <= ltol * sp && mu <= d.tol) { status = 0; break; }
                near_opt = (rd <= 1e-8 * sd && rp <= 1e-8 * sp && mu <= 1e-8);
                if (it >= d.max_iter) { status = 1; break; }
                const double a_aff = fmin(1.0, amax);
                double ma = 0.0;
                if (isu) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) ma += (lu[s2] + a_aff * dlu[s2]) * (tu[s2] + a_aff * dtu[s2]);
                }
                if (xrow) ma += (lx + a_aff * dlx) * (tx + a_aff * dtx);
                const double mu_aff = wg::wave_sum(ma) / d.ng;
                sig = mu > 0.0 ? (mu_aff / mu) * (mu_aff / mu) * (mu_aff / mu) : 0.0;
                if (q.dbg && lane == 0) { gptr gd = q.dbg + 8 * it; gd[5] = a_aff; gd[6] = sig; }
                mode = CORR;
                continue;
            }
            if (!ok) { status = 2; break; }
            const double a = fmin(1.0, 0.99 * amax);
            if (q.dbg && lane == 0) { gptr gd = q.dbg + 8 * it; gd[7] = a; }
            u_r += a * du_r;
            if (lane < 16) L.y[lane] += a * L.dy[lane];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { tu[s2] += a * dtu[s2]; lu[s2] += a * dlu[s2]; }
            tx += a * dtx; lx += a * dlx;
            wave_fence();
            ++it;
            mode = PRED;
        }
        // ---- results: the minimiser, and (converged) the multipliers the next QP of this solve starts from
        if (isu) {
            w.u[lane] = u_r;
            if (status == 0) { w.lam[lsu] = lu[0]; w.lam[lsu + 1] = lu[1]; }
        }
        if (xrow && status == 0) w.lam[lsx] = lx;
        if (lane == 0) { L.red[0] = (double)status; L.red[1] = (double)it; }
    }
    __syncthreads();
    status = (int)L.red[0];
    it = (int)L.red[1];
    __syncthreads();
    QW_LAP(6);
#ifdef SRH_PROFILE
    prof[24] += it; prof[25] += 1; prof[26] += warm ? 1 : 0;
#endif
    if (iters_out) *iters_out = it;
    return status;
}

// Objective of the minimiser (qp::objective, locp.py:218-263): e = H x - z by one thread per (stage, output) -- 60 products each --
// instead of one thread walking the n_z n_x = 360 products of its stage (28 k clocks per QP whatever the horizon:
// profiles/r05_lean_phase_clocks.json), then one thread per stage for the small quadratic forms: the same sums in the same order.
// Round 5: x may be the LDS copy the rollout left (XP = clptr), and the output matrix goes through LDS as well (hs: n_z n_x doubles
// behind that copy): a thread's 60 products then read LDS only -- from global memory they were 60 loads issued one product at a time.
template <typename XP>
__device__ __forceinline__ double objective_par(const QPDims &d, const QPConst &c, const QPData &q, XP x, cgptr u, cgptr s, gptr ez, QPLds &L, bool hl = false, lptr hs = nullptr) {
    const int n = d.n, nz = d.nz, m = d.m, N = d.N;
    // hl: objective_lds_doubles(d) doubles of LDS at hs: the output matrix, the weights Qz / Qzf / R and the output errors -- the second
    // phase's n_z^2 + n_u^2 products then read LDS as well (from global memory: one dependent load per product on a handful of threads)
    lptr Hl = hs, Qzl = Hl + (size_t)nz * n, Qfl = Qzl + nz * nz, Rl = Qfl + nz * nz, ezl = Rl + m * m;
    if (hl) {
        for (int e = SRH_TID; e < nz * n; e += blockDim.x) Hl[e] = c.H[e];
        for (int e = SRH_TID; e < nz * nz; e += blockDim.x) { Qzl[e] = c.Qz[e]; Qfl[e] = c.Qzf ? c.Qzf[e] : 0.0; }
        for (int e = SRH_TID; e < m * m; e += blockDim.x) Rl[e] = c.R[e];
        __syncthreads();
    }
    for (int e = SRH_TID; e < (N + 1) * nz; e += blockDim.x) {
        const int k = e / nz, a = e - k * nz;
        double v = q.z ? -q.z[e] : 0.0;
        if (hl) {
#pragma unroll 12
            for (int j = 0; j < n; ++j) v = fma(Hl[a * n + j], x[(size_t)k * n + j], v);       // (twelve products' reads in flight: a thread walks 60 of them)
        } else {
#pragma unroll 12
            for (int j = 0; j < n; ++j) v = fma(c.H[a * n + j], x[(size_t)k * n + j], v);
        }
        ez[e] = v;
        if (hl) ezl[e] = v;
    }
    __syncthreads();
    double acc = 0.0;
    for (int k = SRH_TID; k <= N; k += blockDim.x) {
        double e[16];
        if (hl) { for (int a = 0; a < nz; ++a) e[a] = ezl[(size_t)k * nz + a]; }
        else { for (int a = 0; a < nz; ++a) e[a] = ez[(size_t)k * nz + a]; }
        for (int a = 0; a < nz; ++a)
            for (int b = 0; b < nz; ++b) acc = fma(e[a] * (hl ? (double)Qzl[a * nz + b] : (double)c.Qz[a * nz + b]), e[b], acc);
        if (k == N && c.Qzf) {
            for (int a = 0; a < nz; ++a) e[a] += (q.z ? q.z[(size_t)k * nz + a] : 0.0) - (q.zf ? q.zf[a] : 0.0);
            for (int a = 0; a < nz; ++a)
                for (int b = 0; b < nz; ++b) acc = fma(e[a] * (hl ? (double)Qfl[a * nz + b] : (double)c.Qzf[a * nz + b]), e[b], acc);
        }
        if (k < N) {
            double ue[16];
            for (int a = 0; a < m; ++a) ue[a] = u[(size_t)k * m + a] - (q.ud ? q.ud[(size_t)k * m + a] : 0.0);
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b) acc = fma(ue[a] * (hl ? (double)Rl[a * m + b] : (double)c.R[a * m + b]), ue[b], acc);
        }
        if (d.tr) acc += q.omega * s[k];
    }
    return wg::reduce(acc, 0, L.red);
}
__host__ __device__ inline size_t objective_lds_doubles(const QPDims &d) { return (size_t)d.nz * d.n + 2 * (size_t)d.nz * d.nz + (size_t)d.m * d.m + (size_t)(d.N + 1) * d.nz; }

// The QP as the SCP loop needs it: condensed interior point, states by a rollout of the minimiser, objective, trust-region
// test.  Returns 0 when the result IS the minimiser of the full QP (converged, inside the trust region); anything else
// means "hand this QP to the fused kernel" (status of the interior point, or 100 = minimiser outside the trust region).
template <int MSEL, int NSEL, int GXSEL, int NST = 0, int J0SEL = 0>
__device__ __forceinline__ int solve_qp(const QPDims &dfull, const QPConst &c, const QPDyn &dyn, const QPData &q, gptr work_base,
                                        Lds &L, double *J_out, int *iters_out, QPWork &wout, long long *prof, int warm = 0) {
    const int tid = SRH_TID, nt = blockDim.x;
    QPLds Lq{};
    Lq.v1 = L.v1; Lq.v2 = L.v2; Lq.Qu = L.Qu; Lq.part = L.part; Lq.red = L.red; Lq.idxl = L.idxl; Lq.flag = L.flag;
    int it = 0;
    int st;
    // GXSEL > 0: box-structured input rows, rows next to their sums, GXSEL lanes per stage for the state rows (one variant per
    // kernel: both interior points in one kernel thrash the instruction cache -- measured -8 % on everything)
    // NST < 0: the short-horizon instantiations (one tile of K): the interior point on one wave
    // NST > 0 and J0SEL == NST: the half-size workgroup (4 waves, every packed row of G in L2)
    constexpr bool HALFWG = NST > 0 && J0SEL == NST;
    if constexpr (NST < 0) st = ipm_wave<MSEL, NSEL, GXSEL>(dfull, c, dyn, q, work_base, L, &it, wout, prof, warm);
    else if constexpr (HALFWG) st = ipm_box4<MSEL, NSEL, GXSEL, NST, J0SEL>(dfull, c, dyn, q, work_base, L, &it, wout, prof, warm);
    else if constexpr (GXSEL > 0) st = ipm_box<MSEL, NSEL, GXSEL, NST, J0SEL>(dfull, c, dyn, q, work_base, L, &it, wout, prof, warm);
    else st = ipm<MSEL, NSEL>(dfull, c, dyn, q, work_base, L, Lq, &it, wout, prof);
    if (iters_out) *iters_out = it;
    if (st != 0) return st;
#ifdef SRH_PROFILE
    long long tail_last = clock64();
#endif
    QPDims d0 = dfull;
    d0.tr = 0;
    QPWork w = wout;
    const int N = d0.N, n = d0.n;
    const double s0 = qp::slack0(dfull, c, q, Lq);
    for (int k = tid; k < N; k += nt) L.idxl[k] = dyn.idx ? dyn.idx[k] : k;
    for (int e = tid; e <= N; e += nt) w.s[e] = (e == 0) ? s0 : 0.0;
    __syncthreads();
    lptr xs = L.B;                                        // (N + 1) n_x doubles at the start of the Theta^T area: free from here on
    rollout<MSEL, NSEL, true, HALFWG ? 4 : 8>(d0, dyn, q.x0, (cgptr)w.u, w.x, L, xs);
    __syncthreads();
#ifdef SRH_PROFILE
    { const long long now_ = clock64(); prof[22] += now_ - tail_last; tail_last = now_; }
#endif
    // (with x and H read from global memory the Diamond fixed layouts were better off with the one-thread-per-stage form -- 54.5 against
    // 55.0 ms per 4096 rollouts --; with the rollout's LDS copy of the trajectory every layout takes the form below)
    double J;
#ifdef SRH_LEAN_OBJECTIVE_SERIAL
    if constexpr (NST <= 0 || MSEL == 8) J = objective_par(d0, c, q, (cgptr)w.x, w.u, w.s, w.ez, Lq);
    else J = qp::objective(d0, c, q, w.x, w.u, w.s, Lq);
#else
    // (the output matrix behind the trajectory copy when the Theta^T area has the room: always at the shipped shapes)
    const bool hfit = (size_t)(N + 1) * n + objective_lds_doubles(d0) <= (size_t)d0.NK * (16 * (size_t)d0.KT + 1);
    J = objective_par(d0, c, q, (clptr)xs, w.u, w.s, w.ez, Lq, hfit, xs + (size_t)(N + 1) * n);
#endif
    bool inside = true;
    if (dfull.tr) {
        double md = 0.0;
        for (int e = tid + n; e < (N + 1) * n; e += nt) md = fmax(md, fabs(c.xs[e % n] * (xs[e] - q.xk[e])));
        md = wg::reduce(md, 1, L.red);
        inside = md <= q.delta;
        J += q.omega * s0;
    }
#ifdef SRH_PROFILE
    { const long long now_ = clock64(); prof[23] += now_ - tail_last; }
#endif
    if (J_out) *J_out = J;
    return inside ? 0 : 100;
}

}  // namespace ql

