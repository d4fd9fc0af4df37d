// The lean condensed kernels (locp_lean.h): one LOCP QP per workgroup, and the whole GuSTO loop (sofacontrol/scp/gusto.py:
// 283-487) per workgroup around it.  A QP this path cannot finish -- interior point not converged, or its minimiser
// outside the trust region (the full QP with its 2 n_x + 1 trust-region rows per stage is needed) -- is handed over: the
// rollout's SCP state is written to a resume record in its work block and the fused kernel (gusto.hip, mode 2) continues it.
#include "scp_types.h"

namespace {

static_assert(NTHREADS == 512, "ql::half_waves: the two wave sets are the SIMD halves of an 8-wave workgroup");

#ifdef SRH_PROFILE
#define GU_LAP(i) do { __syncthreads(); const long long now_ = clock64(); gup[i] += now_ - gul; gul = now_; } while (0)
#else
#define GU_LAP(i) ((void)0)
#endif

// NST > 0: an instantiation for ONE problem layout -- horizon NST, box input rows, NXR state rows, n_z = 6, the first
// LDS-resident stage J0SEL -- whose sizes are compile-time constants: the LDS carve becomes immediate offsets from one base
// (no pointer per array in scalar registers), trip counts and the packed-G offsets fold.  The host only selects it for
// problems whose run-time dimensions equal these (lean_matches); measured on the benchmark: 108 k -> 120 k SCP iterations/s.
template <int MSEL, int GXSEL, int NST, int J0SEL, int NXR>
__device__ __forceinline__ void fix_problem(QPDims &d) {
    if constexpr (NST > 0) {
        static_assert(GXSEL > 0, "fixed-layout variants use the box-row interior point");
        d.N = NST; d.po = 2; d.KT = (2 * NST + 15) / 16;
        d.nU = 2 * MSEL; d.nX = NXR; d.nXf = 0; d.nz = 6;
        d.cond = 1; d.diagD = 1; d.lean = 2; d.lean_j0 = J0SEL;
        d.lean_half = (J0SEL == NST) ? 1 : 0;           // the half-size workgroup: 256 threads, <= 80 KB of LDS (locp_lean.h: ipm_box4)
    }
}

// threads of an instantiation: 256 for the half-size layouts (NST > 0 and J0SEL == NST: two workgroups per CU), else NTHREADS
// (second launch bound: two waves per SIMD -- implied by 512 threads, what lets two 256-thread workgroups share a CU: <= 256 registers)
constexpr int lean_threads(int nst, int j0) { return (nst > 0 && j0 == nst) ? 256 : NTHREADS; }

template <int MSEL, int NSEL, int GXSEL, int NST, int J0SEL, int NXR>
__global__ __launch_bounds__(lean_threads(NST, J0SEL), 2) void gusto_lean_kernel(QPDims d, QPConst c, TpwlDev T, GustoPar par, GustoBatch b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long prof[32] = {0};
#ifdef SRH_PROFILE
    long long gup[8] = {0}, gul = clock64();
#endif
    qp::specialise<MSEL, NSEL>(d);
    fix_problem<MSEL, GXSEL, NST, J0SEL, NXR>(d);
    if constexpr (NST > 0) d.tr = 1;                   // GuSTO always carries the trust region
    const size_t p = b.order ? (size_t)b.order[blockIdx.x] : (size_t)blockIdx.x;
    ql::Lds L;
    ql::lds_carve(L, (lptr)smem, d, lean_threads(NST, J0SEL), (gptr)(b.work + p * b.work_stride));
    if (SRH_TID == 0) L.flag[2] = 0;               // no condensation in LDS yet (ql::ipm)
    ql::serial_wave_pick(L, (par.poison_warm & 4) != 0);
    const int N = d.N, n = d.n, m = d.m, nz = d.nz;
    int tid = SRH_TID;                                 // re-read at the top of every SCP iteration (dev_la.h: SRH_TID)
    const int nt = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6;
    int lane = tid & 63;
    gptr base = (gptr)(b.work + p * b.work_stride);
    QPWork w;
    qp_carve(w, base, d);
    const GustoWork gw = gusto_work(d);
    gptr xk = base + gw.xk, uk = base + gw.uk, accb = base + gw.acc, rec = base + gw.rec;
    giptr idx = (giptr)(base + gw.idx);
    giptr idx2 = idx + N;

    cgptr x0 = (cgptr)(b.x0 + p * n);
    cgptr zp = (cgptr)(b.z ? b.z + p * (size_t)(N + 1) * nz : nullptr);
    cgptr zfp = (cgptr)(b.zf ? b.zf + p * nz : nullptr);
    cgptr udp = (cgptr)(b.ud ? b.ud + p * (size_t)N * m : nullptr);
    if (NST < 0 || b.host_args) {
        // zero-copy solves (gusto.hip) pass pinned HOST pointers: x0, the targets and the desired inputs would be read across PCIe by every
        // QP and interior-point iteration -- keep copies in the work block (the fused kernel finds them there when it resumes a rollout)
        gptr x0c = base + gw.x0c, zc = base + gw.zc, zfc = base + gw.zfc, udc = base + gw.udc;
        for (int e = tid; e < n; e += nt) x0c[e] = x0[e];
        if (zp) for (int e = tid; e < (N + 1) * nz; e += nt) zc[e] = zp[e];
        if (zfp) for (int e = tid; e < nz; e += nt) zfc[e] = zfp[e];
        if (udp) for (int e = tid; e < N * m; e += nt) udc[e] = udp[e];
        x0 = (cgptr)x0c;
        if (zp) zp = (cgptr)zc;
        if (zfp) zfp = (cgptr)zfc;
        if (udp) udp = (cgptr)udc;
    }
    for (int e = tid; e < (N + 1) * n; e += nt) xk[e] = b.x_init[p * (size_t)(N + 1) * n + e];
    for (int e = tid; e < N * m; e += nt) uk[e] = b.u_init[p * (size_t)N * m + e];
    __syncthreads();
    tpwl::nearest_many(T, xk, n, N, idx);
    GU_LAP(0);

    QPDyn dyn{T.Ad, T.AdT, T.Bd, T.BdT, T.dd, (cgiptr)idx};
    double delta = par.delta0, omega = par.omega0;
    double J_prev = INFINITY, d_prev = INFINITY, o_prev = INFINITY;
    // have_warm: w.u / w.lam of the work block hold a converged QP -- of this solve, or (warm_across) of the rollout's previous solve
    bool converged = false, handed_over = false, have_warm = GXSEL > 0 && par.warm_across != 0 && rec[8] == 1.0;
    int itr = 0, status = 0;
    while (itr <= par.max_iters && !converged && omega <= par.omega_max) {
        tid = SRH_TID; lane = tid & 63;
        QPData q{x0, xk, zp, zfp, udp, delta, omega, (gptr)nullptr};
        double J;
        int qit;
        GU_LAP(1);
        // every QP after the first one this kernel finished starts from that one's minimiser and multipliers (ql::ipm_box: warm);
        // a warm-started interior point that does not reach the tolerances is repeated from Mehrotra's point
        int st = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool warm = GXSEL > 0 && have_warm && attempt == 0;
            st = ql::solve_qp<MSEL, NSEL, GXSEL, NST, J0SEL>(d, c, dyn, q, base, L, &J, &qit, w, prof, warm ? ((par.poison_warm & 1) ? 2 : 1) : 0);
            if (st == 0 || st == 100 || !warm) break;
        }
        // test knob (SRH_LEAN_FORCE_HANDOVER=k at plan creation): hand SCP iteration k to the fused kernel as if its relaxed minimiser
        // had left the trust region -- the full QP the fused kernel then solves has the same minimiser, so the solve must come out the
        // same through the hand-over record, the resume launch and the host paths around them
        if (st == 0 && (par.poison_warm >> 4) - 1 == itr) st = 100;
        have_warm = st == 0;
        GU_LAP(2);
        if (st != 0) {                               // the fused kernel takes this rollout from here
            if (tid == 0) {
                rec[0] = 1.0; rec[1] = delta; rec[2] = omega; rec[3] = J_prev; rec[4] = d_prev; rec[5] = o_prev; rec[6] = (double)itr;
                rec[7] = (double)st;                  // 100: relaxed minimiser outside the trust region (the fused kernel skips its own relaxed attempts)
                rec[8] = 0.0;                         // (the fused kernel carves the block differently: nothing to start the next solve from)
                if (b.handed_over) atomicAdd(b.handed_over, 1);
            }
            handed_over = true;
            break;
        }
        // The tests of an SCP iteration (trust region, nearest points, model accuracy, state rows, convergence) and the copy of an
        // accepted step all walk the new and the old trajectory: both go into the K-tile area of the LDS once (dead between two QPs;
        // 2 (N + 1) n_x + 2 N n_u doubles fit wherever the lean layout does: scp_host.h) instead of being fetched from L2 by every one
        // of them, stage by stage, in loops whose loads depend on nothing but were issued one product at a time.
        lptr Xn = L.B, Xo = Xn + (size_t)(N + 1) * n, Un = Xo + (size_t)(N + 1) * n, Uo = Un + (size_t)N * m;
        // (the new trajectory is there already: the QP's final rollout wrote it, ql::solve_qp)
        for (int e = tid; e < (N + 1) * n; e += nt) Xo[e] = xk[e];
        for (int e = tid; e < N * m; e += nt) { Un[e] = w.u[e]; Uo[e] = uk[e]; }
        lptr XAl = Uo + (size_t)N * m;                          // the state rows' matrix behind them (scp_host.h checks the room)
        const bool xal = d.nX > 0;
        if (xal) for (int e = tid; e < d.nX * n; e += nt) XAl[e] = c.XA[e];
        __syncthreads();
        // trust region test (gusto.py:174-183)
        double md = 0.0;
        for (int e = tid; e < (N + 1) * n; e += nt) md = fmax(md, fabs(c.xs[e % n] * (Xn[e] - Xo[e])));
        md = wg::reduce(md, 1, L.red);
        const bool tr_ok = !(md - delta > par.epsilon);
        bool new_solution = false;
        double rho_k = -1.0;
        const double d_cur = delta, o_cur = omega;
        if (tr_ok) {
            // model accuracy (gusto.py:203-223) with continuous nearest-point dynamics
            GU_LAP(3);
            tpwl::nearest_many(T, (clptr)Xn, n, N, idx2);
            GU_LAP(4);
            // One wave per stage, lane = row of the three products.  The stage vectors (old / new state and input) come from the staged
            // copies in LDS instead of being fetched from L2 by every lane and every column; the matrix columns are requested sixteen at a time; when the new point lies in the region of the old one -- most stages -- its matrices
            // are the ones already loaded.  Same products in the same order (round 5: 207 k -> clocks per SCP iteration at C2 in
            // profiles/r05_lean_phase_clocks.json; every stage paid eight dependent L2 round trips).
            for (int i = wave; i < N; i += nw) {
                const size_t ia = (size_t)__builtin_amdgcn_readfirstlane(idx[i]), ib = (size_t)__builtin_amdgcn_readfirstlane(idx2[i]);
                clptr xo_ = Xo + (size_t)i * n, xn_ = Xn + (size_t)i * n, uo_ = Uo + (size_t)i * m, un_ = Un + (size_t)i * m;
                double e2 = 0.0, a2 = 0.0;
                auto rows = [&](auto SAME) {
                    constexpr bool same = decltype(SAME)::value;
                    for (int r = lane; r < n; r += 64) {
                        double fk = T.dc[ia * n + r], fl = 0.0, f = T.dc[ib * n + r];
                        cgptr Ak = T.AcT + ia * n * n, An = T.AcT + ib * n * n;
                        constexpr int CH = same ? 32 : 16;           // columns in flight (registers: one matrix or two)
                        for (int c0 = 0; c0 < n; c0 += CH) {
                            double av[CH], anv[same ? 1 : CH];
#pragma unroll
                            for (int qq = 0; qq < CH; ++qq) {
                                const int cc = c0 + qq < n ? c0 + qq : n - 1;
                                av[qq] = Ak[(size_t)cc * n + r];
                                if constexpr (!same) anv[qq] = An[(size_t)cc * n + r];
                            }
#pragma unroll
                            for (int qq = 0; qq < CH; ++qq) {
                                if (c0 + qq < n) {
                                    const double xo = xo_[c0 + qq], xn = xn_[c0 + qq];
                                    fk = fma(av[qq], xo, fk);
                                    fl = fma(av[qq], xn - xo, fl);
                                    f = fma(same ? av[qq] : anv[same ? 0 : qq], xn, f);
                                }
                            }
                        }
                        cgptr Bk = T.BcT + ia * m * n, Bn = T.BcT + ib * m * n;
                        for (int c0 = 0; c0 < m; c0 += 8) {
                            double bv[8], bnv[8];
#pragma unroll
                            for (int qq = 0; qq < 8; ++qq) {
                                const int cc = c0 + qq < m ? c0 + qq : m - 1;
                                bv[qq] = Bk[(size_t)cc * n + r];
                                if constexpr (!same) bnv[qq] = Bn[(size_t)cc * n + r];
                            }
#pragma unroll
                            for (int qq = 0; qq < 8; ++qq) {
                                if (c0 + qq < m) {
                                    const double uo = uo_[c0 + qq], un = un_[c0 + qq];
                                    fk = fma(bv[qq], uo, fk);
                                    fl = fma(bv[qq], un - uo, fl);
                                    f = fma(same ? bv[qq] : bnv[qq], un, f);
                                }
                            }
                        }
                        const double fa = fk + fl;
                        const double fsr = b.fs[r];
                        const double de = fsr * (f - fa), da = fsr * fa;
                        e2 = fma(de, de, e2);
                        a2 = fma(da, da, a2);
                    }
                };
                if (ia == ib) rows(std::true_type{}); else rows(std::false_type{});
                e2 = wg::wave_sum(e2);
                a2 = wg::wave_sum(a2);
                if (lane == 0) { accb[2 * i] = par.dt * sqrt(e2); accb[2 * i + 1] = par.dt * sqrt(a2); }
            }
            __syncthreads();
            GU_LAP(5);
            double err = 0.0, app = 0.0;      // sequential sums in stage order, as the reference loop
            for (int i = 0; i < N; ++i) { err += accb[2 * i]; app += accb[2 * i + 1]; }
            rho_k = err / (J + app);
            if (rho_k > par.rho && itr != 1) {
                delta = par.beta_fail * delta;
            } else {
                if (d_prev == delta && o_prev == omega && J_prev <= J) delta = par.beta_fail * delta;
                d_prev = delta; J_prev = J; o_prev = omega;
                // state-constraint violation (gusto.py:185-201): all k = 0..N
                double viol = 0.0;
                if (d.nX > 0 && d.nX <= nz) {
                    // one thread per (stage, row) for the n_x products of a row (one thread per stage walked n_X n_x = 240 of them),
                    // then one thread per stage for the norm: the same sums in the same order
                    gptr vr = w.ez;                      // (N + 1) n_z doubles of the QP's work block, free here
                    __syncthreads();
                    for (int e = tid; e < (N + 1) * d.nX; e += nt) {
                        const int k = e / d.nX, r = e - k * d.nX;
                        double v = -c.Xb[r];
                        if (xal) {
#pragma unroll 12
                            for (int j = 0; j < n; ++j) v = fma(XAl[r * n + j], Xn[(size_t)k * n + j], v);
                        } else {
#pragma unroll 12
                            for (int j = 0; j < n; ++j) v = fma(c.XA[(size_t)r * n + j], Xn[(size_t)k * n + j], v);
                        }
                        vr[e] = fmax(v, 0.0);
                    }
                    __syncthreads();
                    for (int k = tid; k <= N; k += nt) {
                        double v2 = 0.0;
                        for (int r = 0; r < d.nX; ++r) { const double v = vr[(size_t)k * d.nX + r]; v2 = fma(v, v, v2); }
                        viol = fmax(viol, sqrt(v2));
                    }
                    viol = wg::reduce(viol, 1, L.red);
                } else if (d.nX > 0) {
                    for (int k = tid; k <= N; k += nt) {
                        double v2 = 0.0;
                        for (int r = 0; r < d.nX; ++r) {
                            double v = -c.Xb[r];
                            for (int j = 0; j < n; ++j) v = fma(c.XA[(size_t)r * n + j], Xn[(size_t)k * n + j], v);
                            v = fmax(v, 0.0);
                            v2 = fma(v, v, v2);
                        }
                        viol = fmax(viol, sqrt(v2));
                    }
                    viol = wg::reduce(viol, 1, L.red);
                }
                const bool X_ok = !(viol > par.epsilon);
                if (!X_ok) omega = par.gamma_fail * omega;
                // convergence (gusto.py:150-161)
                double ds = 0.0;
                for (int k = wave; k <= N; k += nw) {
                    double v2 = 0.0;
                    for (int j = lane; j < n; j += 64) {
                        const double e = c.xs[j] * (Xn[(size_t)k * n + j] - Xo[(size_t)k * n + j]);
                        v2 = fma(e, e, v2);
                    }
                    v2 = wg::wave_sum(v2);
                    if (lane == 0) ds += sqrt(v2);
                }
                ds = wg::reduce(ds, 0, L.red);
                const double dsol = (1.0 / N) * ((1.0 / n) * ds);
                converged = (dsol <= par.convg_thresh) && X_ok;
                new_solution = true;
            }
        } else {
            omega = par.gamma_fail * omega;
        }
        if (b.trace && itr < par.max_trace && tid == 0) {
            double *tr = b.trace + (p * par.max_trace + itr) * 4;
            tr[0] = J; tr[1] = d_cur; tr[2] = o_cur; tr[3] = rho_k;
        }
        ++itr;
        GU_LAP(6);
        if (new_solution) {
            __syncthreads();
            for (int e = tid; e < (N + 1) * n; e += nt) xk[e] = Xn[e];
            for (int e = tid; e < N * m; e += nt) uk[e] = Un[e];
            __syncthreads();
            // the nearest points of the accepted trajectory are the ones the model-accuracy test found for it above (idx2 of
            // w.x, same function, same data: gusto.py:466 recomputes them)
            if (par.max_iters >= 1) { for (int k = tid; k < N; k += nt) idx[k] = idx2[k]; __syncthreads(); }
        }
        GU_LAP(7);
    }
#ifdef SRH_PROFILE
    if (tid == 0 && blockIdx.x == 0) {
        printf("lean gusto clocks (%d iterations): init+nearest %lld loop-top %lld qp %lld tr-test %lld nearest(new) %lld accuracy %lld tests %lld accept+nearest %lld\n",
               itr, gup[0], gup[1], gup[2], gup[3], gup[4], gup[5], gup[6], gup[7]);
        printf("lean qp laps: setup+rollout %lld rows %lld condense %lld stage-factors %lld gram %lld cholesky %lld grad+newton %lld steps %lld\n",
               prof[0], prof[1], prof[2], prof[3], prof[4], prof[5], prof[6], prof[7]);
        printf("lean newton laps: gradients %lld gT(1) %lld rhs+dinv %lld g(1) %lld k_solve %lld gT(2) %lld du %lld g(2) %lld\n",
               prof[8], prof[9], prof[10], prof[11], prof[12], prof[13], prof[14], prof[15]);
        printf("lean step laps: step rows (both modes) %lld reduce(amax) %lld affine mu rows %lld reduce(mu_aff) %lld\n", prof[18], prof[19], prof[20], prof[21]);
        printf("lean qp tail: rollout-of-minimiser %lld objective+tr-test %lld ipm-iterations %lld qps %lld warm-qps %lld\n",
               prof[22], prof[23], prof[24], prof[25], prof[26]);
        printf("lean split (factorisation on waves 0-3 beside the front of the Newton solve on waves 4-7): factorisation %lld front %lld\n", prof[16], prof[17]);
        printf("lean factor chain (wave 0): wait %lld panel %lld update %lld factor %lld signals %lld\n", prof[27], prof[28], prof[29], prof[30], prof[31]);
    }
#endif
    if (handed_over) {
        if (tid == 0) { b.iters[p] = itr; b.status[p] = LEAN_PENDING; }
        return;
    }
    if (status == 0) {
        if (omega > par.omega_max) status = 2;
        else if (itr - 1 > par.max_iters) status = 3;
    }
    __syncthreads();
    for (int e = tid; e < (N + 1) * n; e += nt) b.xopt[p * (size_t)(N + 1) * n + e] = xk[e];
    for (int e = tid; e < N * m; e += nt) b.uopt[p * (size_t)N * m + e] = uk[e];
    for (int e = tid; e < (N + 1) * nz; e += nt) {
        const int k = e / nz, a = e - k * nz;
        double v = 0.0;
        for (int j = 0; j < n; ++j) v = fma(c.H[a * n + j], xk[(size_t)k * n + j], v);
        b.zopt[p * (size_t)(N + 1) * nz + e] = v;
    }
    if (tid == 0) { rec[0] = 0.0; rec[8] = have_warm ? 1.0 : 0.0; b.iters[p] = itr; b.status[p] = status; if (b.last_iters) b.last_iters[p] = itr; if (b.Jopt) b.Jopt[p] = J_prev; }
}

template <int MSEL, int NSEL, int GXSEL, int NST, int J0SEL, int NXR>
__global__ __launch_bounds__(lean_threads(NST, J0SEL), 2) void locp_lean_kernel(QPDims d, QPConst c, LocpBatch b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long prof[32] = {0};
    qp::specialise<MSEL, NSEL>(d);
    fix_problem<MSEL, GXSEL, NST, J0SEL, NXR>(d);
    const size_t p = blockIdx.x;
    ql::Lds L;
    ql::lds_carve(L, (lptr)smem, d, lean_threads(NST, J0SEL), (gptr)(b.work + p * b.work_stride));
    if (SRH_TID == 0) L.flag[2] = 0;
    ql::serial_wave_pick(L);
    const size_t N = d.N, n = d.n, m = d.m;
    QPWork w;
    gptr wbase = (gptr)(b.work + p * b.work_stride);
    QPDyn dyn{(cgptr)(b.Ad + p * N * n * n), (cgptr)(b.AdT + p * N * n * n), (cgptr)(b.Bd + p * N * n * m),
              (cgptr)(b.BdT + p * N * n * m), (cgptr)(b.dd + p * N * n), (cgiptr)nullptr};
    QPData q{(cgptr)(b.x0 + p * n), (cgptr)(b.xk + p * (N + 1) * n), (cgptr)(b.z ? b.z + p * (N + 1) * d.nz : nullptr),
             (cgptr)(b.zf ? b.zf + p * d.nz : nullptr), (cgptr)(b.ud ? b.ud + p * N * m : nullptr), b.delta[p], b.omega[p],
             (gptr)((b.dbg && p == 0) ? b.dbg : nullptr)};
    double J = 0.0;
    int it = 0;
    const int st = ql::solve_qp<MSEL, NSEL, GXSEL, NST, J0SEL>(d, c, dyn, q, wbase, L, &J, &it, w, prof);
    if (q.dbg && SRH_TID == 0) {
        q.dbg[8 * 61] = 2.0; q.dbg[8 * 61 + 1] = (double)st; q.dbg[8 * 61 + 2] = (double)it; q.dbg[8 * 61 + 3] = st == 100 ? 0.0 : 1.0;
#ifdef SRH_PROFILE
        for (int i = 0; i < 8; ++i) { q.dbg[8 * 60 + i] = (double)prof[i]; q.dbg[8 * 59 + i] = (double)prof[8 + i]; }
#endif
    }
    if (st != 0) {
        if (SRH_TID == 0) { b.status[p] = LEAN_PENDING; b.iters[p] = it; if (b.handed_over) atomicAdd(b.handed_over, 1); }
        return;
    }
    for (int e = SRH_TID; e < (N + 1) * n; e += blockDim.x) b.x[p * (N + 1) * n + e] = w.x[e];
    for (int e = SRH_TID; e < N * m; e += blockDim.x) b.u[p * N * m + e] = w.u[e];
    for (int e = SRH_TID; e <= N; e += blockDim.x) b.s[p * (N + 1) + e] = w.s[e];
    if (SRH_TID == 0) { b.J[p] = J; b.status[p] = 0; b.iters[p] = it; }
}

// instantiated shapes: the reference's 4- and 8-cable robots at the benchmark's r = 30 with the row layout their drivers use
// (U box; Diamond: 4 state rows, Trunk: none; Diamond at the shipped r = 36 basis: n_x = 72), then n_x fixed / free with the
// general row handling (GX = 0)
// (M, NX, GX, NST, J0, NXR); NST = -1: short horizons (N p_o <= 16), the interior point on one wave (ql::ipm_wave).
// First the layouts with every size fixed (BASELINE C2: Diamond, N = 50, 4 state rows; C5: Trunk,
// N = 50, no state rows; the Diamond at its shipped r = 36 basis), then the run-time-horizon forms
// (a development build may pass its own, shorter list: tools/build_lean_dev.sh compiles the benchmark layouts only)
#ifndef SRH_LEAN_VARIANTS
#define SRH_LEAN_VARIANTS(X) X(4, 60, 4, 50, 50, 4) X(4, 60, 4, 50, 7, 4) X(8, 60, 1, 50, 24, 0) X(4, 72, 4, 50, 18, 4) \
    X(4, 60, 4, -1, 0, 0) X(4, 60, 1, -1, 0, 0) X(4, 72, 4, -1, 0, 0) X(8, 60, 1, -1, 0, 0) \
    X(4, 60, 4, 0, 0, 0) X(4, 60, 1, 0, 0, 0) X(8, 60, 1, 0, 0, 0) X(4, 72, 4, 0, 0, 0) X(4, 60, 0, 0, 0, 0) X(8, 60, 0, 0, 0, 0) X(4, 0, 0, 0, 0, 0) X(8, 0, 0, 0, 0, 0)
#endif
inline int lean_gx(const QPDims &d) {
    if (d.lean != 2) return 0;
    const int RXa = d.nX + d.nXf;
    return RXa == 0 ? 1 : (RXa <= 2 ? 2 : (RXa <= 4 ? 4 : 8));
}
inline bool lean_matches(const QPDims &d, bool allow_fixed, int msel, int nsel, int gx, int nst, int j0, int nxr) {
    if (d.m != msel || (nsel != 0 && d.n != nsel)) return false;
    if (!(gx == 0 || gx == lean_gx(d))) return false;
    // a problem laid out for the half-size workgroup (scp_host.h: SRH_LEAN_HALF=1) runs its own instantiations only, and nothing else runs them
    const bool half_inst = nst > 0 && j0 == nst;
    if ((d.lean_half != 0) != half_inst) return false;
    if (nst == 0) return true;
    // nst < 0: the short-horizon form (ql::ipm_wave): K is one tile, one lane per input and per state-row slot, the whole packed
    // G in LDS; SRH_LEAN_NO_WAVE=1 at plan creation skips it (A/B runs, tests of both forms)
    if (nst < 0) return gx > 0 && d.KT == 1 && d.N * d.m <= 64 && d.N * gx <= 64 && d.lean_j0 == 0 && d.po == 2 && getenv("SRH_LEAN_NO_WAVE") == nullptr;
    return allow_fixed && d.N == nst && d.lean_j0 == j0 && d.nX == nxr && d.nXf == 0 && d.nz == 6 && d.po == 2 && d.nU == 2 * msel;
}

}  // namespace

// The instantiation a problem runs: the first row of SRH_LEAN_VARIANTS whose compile-time sizes equal the problem's
// (decided ONCE, when a plan is created -- SRH_LEAN_NO_FIXED=1 in the environment at that time skips the fixed-layout rows
// for A/B runs; nothing is read from the environment per launch).  args: <n_u, n_x, GX, N, j0, state rows>.
int lean_select(const QPDims &d, int args[6]) {
    const bool allow_fixed = getenv("SRH_LEAN_NO_FIXED") == nullptr;
    int idx = 0;
#define X(M, NX, GX, NST, J0, NXR) if (lean_matches(d, allow_fixed, M, NX, GX, NST, J0, NXR)) { \
        if (args) { args[0] = M; args[1] = NX; args[2] = GX; args[3] = NST; args[4] = J0; args[5] = NXR; } return idx; } ++idx;
    SRH_LEAN_VARIANTS(X)
#undef X
    return -1;
}

int lean_prepare(int variant, size_t lds) {
    SRH_REQUIRE(lds <= 160 * 1024, "lean kernels: %zu bytes of LDS needed, 160 KiB available", lds);
    int idx = 0;
#define X(M, NX, GX, NST, J0, NXR) if (idx++ == variant) { \
        SRH_CHECK_HIP(hipFuncSetAttribute((const void *)gusto_lean_kernel<M, NX, GX, NST, J0, NXR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        SRH_CHECK_HIP(hipFuncSetAttribute((const void *)locp_lean_kernel<M, NX, GX, NST, J0, NXR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        return SRH_OK; }
    SRH_LEAN_VARIANTS(X)
#undef X
    SRH_REQUIRE(false, "lean kernels: no variant %d", variant);
    return SRH_OK;
}

int lean_launch_gusto(int variant, const QPDims &d, const QPConst &c, const TpwlDev &T, const GustoPar &par, const GustoBatch &b, unsigned grid,
                      size_t lds, hipStream_t stream) {
    QPDims dd = d;
    int idx = 0;
#define X(M, NX, GX, NST, J0, NXR) if (idx++ == variant) { if (GX == 0) dd.lean = 1; gusto_lean_kernel<M, NX, GX, NST, J0, NXR><<<grid, lean_threads(NST, J0), lds, stream>>>(dd, c, T, par, b); SRH_CHECK_HIP(hipGetLastError()); return SRH_OK; }
    SRH_LEAN_VARIANTS(X)
#undef X
    SRH_REQUIRE(false, "lean kernels: no variant %d", variant);
    return SRH_OK;
}

int lean_launch_locp(int variant, const QPDims &d, const QPConst &c, const LocpBatch &b, unsigned grid, size_t lds, hipStream_t stream) {
    QPDims dd = d;
    int idx = 0;
#define X(M, NX, GX, NST, J0, NXR) if (idx++ == variant) { if (GX == 0) dd.lean = 1; locp_lean_kernel<M, NX, GX, NST, J0, NXR><<<grid, lean_threads(NST, J0), lds, stream>>>(dd, c, b); SRH_CHECK_HIP(hipGetLastError()); return SRH_OK; }
    SRH_LEAN_VARIANTS(X)
#undef X
    SRH_REQUIRE(false, "lean kernels: no variant %d", variant);
    return SRH_OK;
}
