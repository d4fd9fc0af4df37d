// GuSTO (SCP outer loop, sofacontrol/scp/gusto.py:283-487) on the device: one workgroup per rollout runs the whole
// solve, each SCP iteration's QP through the device interior point of locp_dev.h; the resident plan API.
#include "scp_types.h"
#include <chrono>

namespace {

// Longest-processing-time-first dispatch: the rollouts of a receding-horizon batch need 1..max SCP iterations each
// and a workgroup owns its CU for the whole solve, so the tail of a launch is set by whichever long solves start
// last.  The previous solve of the same rollout predicts its length; a counting sort on those iteration counts
// (descending) gives the workgroup -> rollout map of the next launch.  Results do not depend on the order.
__global__ __launch_bounds__(1024) void lpt_order_kernel(const int32_t *__restrict__ key, int64_t batch, int32_t *__restrict__ order) {
    __shared__ int cnt[1024];
    const int tid = SRH_TID;
    cnt[tid] = 0;
    __syncthreads();
    for (int64_t i = tid; i < batch; i += 1024) atomicAdd(&cnt[1023 - min(max(key[i], 0), 1023)], 1);
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int q = 0; q < 1024; ++q) { const int c = cnt[q]; cnt[q] = run; run += c; }
    }
    __syncthreads();
    for (int64_t i = tid; i < batch; i += 1024) order[atomicAdd(&cnt[1023 - min(max(key[i], 0), 1023)], 1)] = (int32_t)i;
}

#ifdef SRH_PROFILE
#define GU_LAP(i) do { __syncthreads(); const long long now_ = clock64(); gup[i] += now_ - gul; gul = now_; } while (0)
#else
#define GU_LAP(i) ((void)0)
#endif

template <bool SPLIT, int MSEL, int NSEL>
__global__ __launch_bounds__(NTHREADS) void gusto_kernel(QPDims d, QPConst c, TpwlDev T, GustoPar par, GustoBatch b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef SRH_PROFILE
    long long gup[8] = {0}, gul = clock64();
#endif
    qp::specialise<MSEL, NSEL>(d);             // compile-time n_u (and n_x) for everything inlined below
    QPLds L;
    qp_lds_carve(L, (lptr)smem, d, NTHREADS);          // qp::solve fills the constants of the layout it uses
    const size_t p = b.order ? (size_t)b.order[blockIdx.x] : (size_t)blockIdx.x;
    const int N = d.N, n = d.n, m = d.m, nz = d.nz;
    int tid = SRH_TID;                                 // re-read at the top of every SCP iteration (dev_la.h: SRH_TID)
    const int nt = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6;
    int lane = tid & 63;
    gptr base = (gptr)(b.work + p * b.work_stride);
    QPWork w;
    qp_carve(w, base, d);
    const GustoWork gw = gusto_work(d);
    gptr xk = base + gw.xk, uk = base + gw.uk;
    gptr accb = base + gw.acc;                         // 2*N doubles: per-stage error / approx
    gptr rec = base + gw.rec;                          // resume record of a rollout handed over by the lean kernel
    giptr idx = (giptr)(base + gw.idx);
    giptr idx2 = idx + N;

    cgptr x0 = (cgptr)(b.x0 + p * n);
    cgptr zp = (cgptr)(b.z ? b.z + p * (size_t)(N + 1) * nz : nullptr);
    cgptr zfp = (cgptr)(b.zf ? b.zf + p * nz : nullptr);
    cgptr udp = (cgptr)(b.ud ? b.ud + p * (size_t)N * m : nullptr);
    if (b.host_args) {
        // zero-copy solve: the arguments sit in pinned host memory -- work on copies in the work block (mode 2: the lean kernel made them)
        gptr x0c = base + gw.x0c, zc = base + gw.zc, zfc = base + gw.zfc, udc = base + gw.udc;
        if (b.mode != 2) {
            for (int e = tid; e < n; e += nt) x0c[e] = x0[e];
            if (zp) for (int e = tid; e < (N + 1) * nz; e += nt) zc[e] = zp[e];
            if (zfp) for (int e = tid; e < nz; e += nt) zfc[e] = zfp[e];
            if (udp) for (int e = tid; e < N * m; e += nt) udc[e] = udp[e];
        }
        x0 = (cgptr)x0c;
        if (zp) zp = (cgptr)zc;
        if (zfp) zfp = (cgptr)zfc;
        if (udp) udp = (cgptr)udc;
    }
    double delta = par.delta0, omega = par.omega0;
    double J_prev = INFINITY, d_prev = INFINITY, o_prev = INFINITY;
    bool converged = false;
    int itr = 0, status = 0;
    bool tr_hot = false;                               // expect the trust region to bind in the next QP (see qp::solve, full_first)
    bool warm_relaxed = false;                         // the previous QP ended in the relaxed Riccati pass (qp::solve, warm)
    bool warm_full = false;                            // the previous QP ended in the full pass (trust-region rows): qp::solve, warm_full
    if (b.mode == 2) {
        // only the rollouts a lean launch (lean.hip) could not finish: xk, uk, idx are where it left them
        if (rec[0] != 1.0) return;
        delta = rec[1]; omega = rec[2]; J_prev = rec[3]; d_prev = rec[4]; o_prev = rec[5]; itr = (int)rec[6];
        tr_hot = rec[7] == 100.0;                      // the lean kernel found this QP's relaxed minimiser outside the trust region
        __syncthreads();
        if (tid == 0) rec[0] = 0.0;
    } else {
        for (int e = tid; e < (N + 1) * n; e += nt) xk[e] = b.x_init[p * (size_t)(N + 1) * n + e];
        for (int e = tid; e < N * m; e += nt) uk[e] = b.u_init[p * (size_t)N * m + e];
        __syncthreads();
        tpwl::nearest_many(T, xk, n, N, idx);
    }
    GU_LAP(0);

    QPDyn dyn{T.Ad, T.AdT, T.Bd, T.BdT, T.dd, (cgiptr)idx};
    while (itr <= par.max_iters && !converged && omega <= par.omega_max) {
        tid = SRH_TID; lane = tid & 63;
        QPData q{x0, xk, zp, zfp, udp, delta, omega, (gptr)nullptr};
        double J;
        int qit;
        GU_LAP(1);
        int qpass = -1;
        const int st = qp::solve<SPLIT, MSEL, NSEL>(d, c, dyn, q, base, L, &J, &qit, true, w, tr_hot, warm_relaxed, &qpass,
                                                    warm_full && tr_hot && par.warm_full != 0);
        warm_relaxed = st == 0 && qpass == 0;        // the next QP's relaxed Riccati pass may start from this one's (u, lambda)
        warm_full = st == 0 && qpass == 1;           // ... and its full pass from this one's (u, s, lambda) when it goes there directly
        GU_LAP(2);
        if (st != 0) { status = 1; break; }          // gusto.py:357-365: keep the previous iterate
        // trust region test (gusto.py:174-183)
        double md = 0.0;
        for (int e = tid; e < (N + 1) * n; e += nt) md = fmax(md, fabs(c.xs[e % n] * (w.x[e] - xk[e])));
        md = wg::reduce(md, 1, L.red);
        const bool tr_ok = !(md - delta > par.epsilon);
        const bool on_boundary = md >= delta * (1.0 - 1e-9);      // this QP's minimiser used the whole trust region
        bool new_solution = false;
        double rho_k = -1.0;
        const double d_cur = delta, o_cur = omega;
        if (tr_ok) {
            // model accuracy (gusto.py:203-223) with continuous nearest-point dynamics
            GU_LAP(3);
            tpwl::nearest_many(T, w.x, n, N, idx2);
            GU_LAP(4);
            // (sixteen columns in flight; when the new point lies in the region of the old one its matrices are the ones already loaded:
            // same products in the same order -- the lean kernel's form, lean.hip, without its LDS copies of the trajectories)
            for (int i = wave; i < N; i += nw) {
                const size_t ia = (size_t)__builtin_amdgcn_readfirstlane(idx[i]), ib = (size_t)__builtin_amdgcn_readfirstlane(idx2[i]);
                double e2 = 0.0, a2 = 0.0;
                auto rows = [&](auto SAME) {
                    constexpr bool same = decltype(SAME)::value;
                    for (int r = lane; r < n; r += 64) {
                        double fk = T.dc[ia * n + r], fl = 0.0, f = T.dc[ib * n + r];
                        cgptr Ak = T.AcT + ia * n * n, An = T.AcT + ib * n * n;
                        for (int c0 = 0; c0 < n; c0 += 16) {
                            double av[16], anv[16], xo[16], xn[16];
#pragma unroll
                            for (int q = 0; q < 16; ++q) {
                                const int cc = c0 + q < n ? c0 + q : n - 1;
                                av[q] = Ak[(size_t)cc * n + r];
                                if constexpr (!same) anv[q] = An[(size_t)cc * n + r];
                                xo[q] = xk[(size_t)i * n + cc]; xn[q] = w.x[(size_t)i * n + cc];
                            }
#pragma unroll
                            for (int q = 0; q < 16; ++q) {
                                if (c0 + q < n) {
                                    fk = fma(av[q], xo[q], fk);
                                    fl = fma(av[q], xn[q] - xo[q], fl);
                                    f = fma(same ? av[q] : anv[q], xn[q], f);
                                }
                            }
                        }
                        cgptr Bk = T.BcT + ia * m * n, Bn = T.BcT + ib * m * n;
                        for (int c0 = 0; c0 < m; c0 += 8) {
                            double bv[8], bnv[8], uo[8], un[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int cc = c0 + q < m ? c0 + q : m - 1;
                                bv[q] = Bk[(size_t)cc * n + r];
                                if constexpr (!same) bnv[q] = Bn[(size_t)cc * n + r];
                                uo[q] = uk[(size_t)i * m + cc]; un[q] = w.u[(size_t)i * m + cc];
                            }
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                if (c0 + q < m) {
                                    fk = fma(bv[q], uo[q], fk);
                                    fl = fma(bv[q], un[q] - uo[q], fl);
                                    f = fma(same ? bv[q] : bnv[q], un[q], f);
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
                if (d.nX > 0) {
                    for (int k = tid; k <= N; k += nt) {
                        double v2 = 0.0;
                        for (int r = 0; r < d.nX; ++r) {
                            double v = -c.Xb[r];
                            for (int j = 0; j < n; ++j) v = fma(c.XA[(size_t)r * n + j], w.x[(size_t)k * n + j], v);
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
                        const double e = c.xs[j] * (w.x[(size_t)k * n + j] - xk[(size_t)k * n + j]);
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
            if (par.poison_warm & 2) tr[3] = (double)(qit + 1000 * (qpass + 1));     // debug (SRH_GUSTO_TRACE_QIT=1): interior-point iterations + 1000 (pass + 1)
        }
        // the next QP keeps this linearisation point when the step was rejected (smaller delta or larger omega, same relaxed
        // minimiser): if this one already ended on the boundary of its trust region the next one is certain to bind
        tr_hot = on_boundary && !new_solution;
        ++itr;
        GU_LAP(6);
        if (new_solution) {
            __syncthreads();
            for (int e = tid; e < (N + 1) * n; e += nt) xk[e] = w.x[e];
            for (int e = tid; e < N * m; e += nt) uk[e] = w.u[e];
            __syncthreads();
            // the nearest points of the accepted trajectory are the ones the model-accuracy test found for it above (idx2 of
            // w.x, same function, same data: gusto.py:466 recomputes them)
            if (par.max_iters >= 1) { for (int k = tid; k < N; k += nt) idx[k] = idx2[k]; __syncthreads(); }
        }
        GU_LAP(7);
    }
#ifdef SRH_PROFILE
    if (tid == 0 && blockIdx.x == 0)
        printf("gusto clocks (%d iterations): init+nearest %lld loop-top %lld qp %lld tr-test %lld nearest(new) %lld accuracy %lld tests %lld accept+nearest %lld\n",
               itr, gup[0], gup[1], gup[2], gup[3], gup[4], gup[5], gup[6], gup[7]);
#endif
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
    if (tid == 0) { b.iters[p] = itr; b.status[p] = status; if (b.last_iters) b.last_iters[p] = itr; if (b.Jopt) b.Jopt[p] = J_prev; }
}


const void *gusto_entry(const QPDims &d) {
#define X(SP, M, NX) if (variant_matches(d, SP, M, NX)) return (const void *)gusto_kernel<SP, M, NX>;
    SRH_QP_VARIANTS(X)
#undef X
    return nullptr;
}

}  // namespace

// ------------------------------------------------------------------ GuSTO plan (resident solver)
struct sgusto_plan {
    stpwl *model = nullptr;
    QPConstHost C;
    GustoPar par{};
    int64_t batch = 0;
    srh::DevBuf fs, work, x0, u_init, x_init, z, zf, ud, xopt, uopt, zopt, iters, status, trace, order, last_iters, handed, Jopt;
    bool have_last = false;             // a previous solve left its iteration counts
    size_t work_stride = 0;
    size_t lds = 0, lean_lds = 0;
    bool lean = false;                  // the lean condensed kernel runs first, the fused kernel takes what it hands over
    int lean_variant = -1;              // its instantiation (lean_select), fixed when the plan is created
    int lean_args[6] = {0, 0, 0, 0, 0, 0};
    bool use_lpt = true;                // longest-expected-rollout-first dispatch for batch > #CUs (SRH_GUSTO_NO_LPT=1 at creation: off)
    bool solved = false;
    bool has_z = false, has_zf = false, has_ud = false;
    // asynchronous requests (sgusto_plan_solve_begin / _done / _end): own stream, completion event, pinned staging
    hipStream_t astream = nullptr;
    hipEvent_t adone = nullptr, abegin = nullptr;       // timing enabled: the device-side duration of a request
    double last_ms = -1.0;
    char *pin = nullptr;               // one pinned block: [inputs | outputs]
    size_t pin_bytes = 0;
    int host_handed = -1;              // zero-copy solves: rollouts the last solve handed to the fused kernel (counted on the host), else -1
    bool pending = false, want_trace = false, launching = false;
    ~sgusto_plan() {
        if (pending && adone) (void)hipEventSynchronize(adone);
        if (adone) (void)hipEventDestroy(adone);
        if (abegin) (void)hipEventDestroy(abegin);
        if (astream) (void)hipStreamDestroy(astream);
        if (pin) (void)hipHostFree(pin);
    }
};

namespace {
// layout of the pinned staging block of a plan (byte offsets)
struct PinLayout {
    size_t x0, u_init, x_init, z, zf, ud, xopt, uopt, zopt, iters, status, trace, total;
};
PinLayout pin_layout(const sgusto_plan *pl) {
    const QPDims &d = pl->C.dims;
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz, B = pl->batch, D = sizeof(double);
    PinLayout L{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 63) & ~(size_t)63; return at; };
    L.x0 = take(D * B * n); L.u_init = take(D * B * N * m); L.x_init = take(D * B * (N + 1) * n);
    L.z = take(D * B * (N + 1) * nz); L.zf = take(D * B * nz); L.ud = take(D * B * N * m);
    L.xopt = take(D * B * (N + 1) * n); L.uopt = take(D * B * N * m); L.zopt = take(D * B * (N + 1) * nz);
    L.iters = take(sizeof(int32_t) * B); L.status = take(sizeof(int32_t) * B);
    L.trace = take(D * B * (size_t)std::max(1, pl->par.max_trace) * 4);
    L.total = o;
    return L;
}
}  // namespace

extern "C" {

void sgusto_default_params(sgusto_params *p) {
    if (!p) return;
    p->delta0 = 1e4; p->omega0 = 1.0; p->rho = 0.1; p->beta_fail = 0.5; p->gamma_fail = 5.0;
    p->epsilon = 0.01; p->omega_max = 1e10; p->convg_thresh = 0.1; p->max_gusto_iters = 500;
}


int sgusto_plan_create(sgusto_plan_t **out, stpwl_t *h, const slocp_problem *prob, const sgusto_params *par,
                       double dt, int64_t batch, const double *x_char, const double *f_char, int max_trace) {
    SRH_REQUIRE(out && h && prob && par, "sgusto_plan_create: null argument");
    SRH_REQUIRE(h->has_discrete, "sgusto_plan_create: model has not been pre-discretised (TPWLGuSTO.pre_discretize)");
    SRH_REQUIRE(prob->n_x == h->n && prob->n_u == h->m, "sgusto_plan_create: problem / model dimension mismatch");
    SRH_REQUIRE(batch > 0, "sgusto_plan_create: batch must be positive");
    sgusto_plan *pl = new sgusto_plan();
    pl->model = h;
    pl->batch = batch;
    slocp_problem p2 = *prob;
    std::vector<double> xs(h->n, 1.0), fs(h->n, 1.0);
    if (x_char) for (int i = 0; i < h->n; ++i) xs[i] = 1.0 / fabs(x_char[i]);
    if (f_char) for (int i = 0; i < h->n; ++i) fs[i] = 1.0 / fabs(f_char[i]);
    if (x_char) p2.x_scale = xs.data();
    // (more rollouts than CUs: the half-size lean workgroup where the problem has its shape -- two rollouts per CU)
    int rc = build_consts(&p2, pl->C, batch > 256);
    if (rc) { delete pl; return rc; }
    QPDims &d = pl->C.dims;
    pl->par = GustoPar{par->delta0, par->omega0, par->rho, par->beta_fail, par->gamma_fail, par->epsilon,
                       par->omega_max, par->convg_thresh, dt, par->max_gusto_iters, max_trace, 0,
                       (getenv("SRH_LEAN_POISON_WARM") != nullptr ? 1 : 0) | (getenv("SRH_GUSTO_TRACE_QIT") != nullptr ? 2 : 0) | (getenv("SRH_LEAN_SERIAL_WAVE") != nullptr ? 4 : 0) |
                       (getenv("SRH_LEAN_FORCE_HANDOVER") != nullptr ? (atoi(getenv("SRH_LEAN_FORCE_HANDOVER")) + 1) << 4 : 0),
                       getenv("SRH_GUSTO_WARM_FULL") != nullptr ? 1 : 0};
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz;
    size_t doubles = gusto_work(d).end;
    doubles = (doubles + 3) & ~(size_t)3;
    d.qc_off = (long long)doubles;                 // the condensed path's block sits behind the SCP loop's own arrays
    doubles += qc_work_doubles(d);
    pl->work_stride = (doubles + 3) & ~(size_t)3;
    pl->lds = qp_kernel_lds_bytes(d);
    if ((rc = pl->fs.upload(fs.data(), sizeof(double) * n)) || (rc = pl->work.alloc(sizeof(double) * pl->work_stride * batch)) ||
        (rc = pl->x0.alloc(sizeof(double) * batch * n)) || (rc = pl->u_init.alloc(sizeof(double) * batch * N * m)) ||
        (rc = pl->x_init.alloc(sizeof(double) * batch * (N + 1) * n)) || (rc = pl->z.alloc(sizeof(double) * batch * (N + 1) * nz)) ||
        (rc = pl->zf.alloc(sizeof(double) * batch * nz)) || (rc = pl->ud.alloc(sizeof(double) * batch * N * m)) ||
        (rc = pl->xopt.alloc(sizeof(double) * batch * (N + 1) * n)) || (rc = pl->uopt.alloc(sizeof(double) * batch * N * m)) ||
        (rc = pl->zopt.alloc(sizeof(double) * batch * (N + 1) * nz)) || (rc = pl->iters.alloc(sizeof(int32_t) * batch)) ||
        (rc = pl->status.alloc(sizeof(int32_t) * batch)) || (rc = pl->order.alloc(sizeof(int32_t) * batch)) ||
        (rc = pl->last_iters.alloc(sizeof(int32_t) * batch)) || (rc = pl->handed.alloc(sizeof(int32_t))) || (rc = pl->Jopt.alloc(sizeof(double) * batch)) ||
        (rc = pl->trace.alloc(sizeof(double) * batch * (size_t)std::max(1, max_trace) * 4)) ||
        (rc = set_lds_limit(gusto_entry(d), pl->lds))) {
        delete pl;
        return rc;
    }
    // the resume / warm-start records of the work blocks start as "nothing there"
    if (hipMemset(pl->work.p, 0, sizeof(double) * pl->work_stride * batch) != hipSuccess || hipMemset(pl->handed.p, 0, sizeof(int32_t)) != hipSuccess) {
        delete pl;
        SRH_REQUIRE(false, "sgusto_plan_create: hipMemset of the hand-over counter failed");
    }
    pl->use_lpt = getenv("SRH_GUSTO_NO_LPT") == nullptr;
    if (d.lean && !getenv("SRH_GUSTO_NO_LEAN")) {
        pl->lean_variant = lean_select(d, pl->lean_args);
        if (pl->lean_variant < 0) {
            const int n_u = d.m;                   // (d refers into the plan)
            delete pl;
            SRH_REQUIRE(false, "sgusto_plan_create: no lean kernel instantiation for n_u = %d", n_u);
        }
        pl->lean_lds = lean_kernel_lds_bytes(d);
        if ((rc = lean_prepare(pl->lean_variant, pl->lean_lds))) { delete pl; return rc; }
        pl->lean = true;
    }
    *out = pl;
    return SRH_OK;
}

int sgusto_plan_destroy(sgusto_plan_t *pl) {
    delete pl;
    return SRH_OK;
}

int sgusto_plan_set_warm_across(sgusto_plan_t *pl, int on) {
    SRH_REQUIRE(pl, "sgusto_plan_set_warm_across: null plan");
    pl->par.warm_across = on ? 1 : 0;
    return SRH_OK;
}

/* Whether a plan can honour sgusto_plan_set_warm_across: only the lean kernels with the box-row interior point (GX > 0) keep the
 * minimiser / multipliers of a rollout's last QP in its work block; fused-only plans, the general-row lean variants and
 * SRH_GUSTO_NO_LEAN plans start every solve cold whatever the flag says (and so does a rollout whose previous solve was handed to
 * the fused kernel).  *active = 1: requested AND honoured. */
int sgusto_plan_warm_across_active(const sgusto_plan_t *pl, int *active) {
    SRH_REQUIRE(pl && active, "sgusto_plan_warm_across_active: null argument");
    *active = (pl->par.warm_across != 0 && pl->lean && pl->lean_args[2] > 0) ? 1 : 0;
    return SRH_OK;
}

int sgusto_plan_set_max_iters(sgusto_plan_t *pl, int max_gusto_iters) {
    SRH_REQUIRE(pl, "sgusto_plan_set_max_iters: null plan");
    pl->par.max_iters = max_gusto_iters;
    return SRH_OK;
}

int sgusto_plan_variant(const sgusto_plan_t *pl, int *split, int *n_u_fixed, int *n_x_fixed) {
    SRH_REQUIRE(pl, "sgusto_plan_variant: null plan");
    const QPDims &d = pl->C.dims;
#define X(SP, M, NX) if (variant_matches(d, SP, M, NX)) { if (split) *split = SP ? 1 : 0; if (n_u_fixed) *n_u_fixed = M; if (n_x_fixed) *n_x_fixed = NX; return SRH_OK; }
    SRH_QP_VARIANTS(X)
#undef X
    SRH_REQUIRE(false, "sgusto_plan_variant: no kernel variant matches");
    return SRH_OK;
}

// phase 0: everything (lean launch + the fused kernel behind it for what the lean kernel hands over -- or the fused kernel alone);
// phase 1: the lean launch alone, without the hand-over counter (the caller looks at the status words itself: zero-copy solves);
// phase 2: the fused kernel in resume mode alone (after phase 1 found handed-over rollouts)
static int solve_dev_impl(sgusto_plan_t *pl, const double *x0, const double *u_init, const double *x_init,
                          const double *z, const double *zf, const double *u_des, double *xopt, double *uopt,
                          double *zopt, int32_t *iters, int32_t *status, double *trace, void *stream, int phase);

int sgusto_plan_solve_dev(sgusto_plan_t *pl, const double *x0, const double *u_init, const double *x_init,
                          const double *z, const double *zf, const double *u_des, double *xopt, double *uopt,
                          double *zopt, int32_t *iters, int32_t *status, double *trace, void *stream) {
    return solve_dev_impl(pl, x0, u_init, x_init, z, zf, u_des, xopt, uopt, zopt, iters, status, trace, stream, 0);
}

static int solve_dev_impl(sgusto_plan_t *pl, const double *x0, const double *u_init, const double *x_init,
                          const double *z, const double *zf, const double *u_des, double *xopt, double *uopt,
                          double *zopt, int32_t *iters, int32_t *status, double *trace, void *stream, int phase) {
    SRH_REQUIRE(pl && x0 && u_init && x_init && xopt && uopt && zopt && iters && status,
                "sgusto_plan_solve_dev: null argument");
    if (phase == 0) pl->host_handed = -1;
    // an asynchronous request in flight (sgusto_plan_solve_begin) is using the plan's work blocks on its own stream
    SRH_REQUIRE(!pl->pending || (pl->astream && stream == (void *)pl->astream && pl->launching),
                "sgusto_plan_solve_dev: an asynchronous request is in flight on this plan (call sgusto_plan_solve_end first)");
    GustoBatch b{x0, u_init, x_init, z, zf, u_des, pl->fs.as<double>(), xopt, uopt, zopt, iters, status, trace,
                 pl->work.as<double>(), pl->work_stride, nullptr, pl->last_iters.as<int32_t>(), 0, pl->handed.as<int32_t>(), pl->Jopt.as<double>(),
                 phase != 0 ? 1 : 0};        // phases 1 / 2 = the zero-copy host path: pinned host arguments
    if (pl->have_last && pl->batch > 256 && pl->use_lpt) {      // more rollouts than CUs: order matters
        lpt_order_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(pl->last_iters.as<int32_t>(), pl->batch, pl->order.as<int32_t>());
        b.order = pl->order.as<int32_t>();
    }
    pl->have_last = true;
    GustoPar par = pl->par;
    if (!trace) par.max_trace = 0;
    if (pl->lean) {
        // lean condensed kernel over every rollout; the fused kernel then continues the handed-over ones (its other
        // workgroups leave at once)
        if (phase != 2) {
            if (phase == 0) SRH_CHECK_HIP(hipMemsetAsync(pl->handed.p, 0, sizeof(int32_t), (hipStream_t)stream));
            else b.handed_over = nullptr;
            int rc = lean_launch_gusto(pl->lean_variant, pl->C.dims, pl->C.view(), pl->model->view(), par, b, (unsigned)pl->batch, pl->lean_lds,
                                       (hipStream_t)stream);
            if (rc) return rc;
        }
        b.mode = 2;
        b.order = nullptr;
    }
    if (!(pl->lean && phase == 1)) {
        const QPDims &d = pl->C.dims;
        bool launched = false;
#define X(SP, M, NX) if (!launched && variant_matches(d, SP, M, NX)) { gusto_kernel<SP, M, NX><<<(unsigned)pl->batch, NTHREADS, pl->lds, (hipStream_t)stream>>>(d, pl->C.view(), pl->model->view(), par, b); launched = true; }
        SRH_QP_VARIANTS(X)
#undef X
    }
    SRH_CHECK_HIP(hipGetLastError());
    pl->solved = true;
    return SRH_OK;
}

int sgusto_plan_costs(sgusto_plan_t *pl, double *J) {
    SRH_REQUIRE(pl && J, "sgusto_plan_costs: null argument");
    SRH_REQUIRE(pl->solved, "sgusto_plan_costs: no solve yet");
    SRH_REQUIRE(!pl->pending, "sgusto_plan_costs: an asynchronous request is in flight on this plan (call sgusto_plan_solve_end first)");
    if (pl->astream) SRH_CHECK_HIP(hipStreamSynchronize(pl->astream));
    SRH_CHECK_HIP(hipDeviceSynchronize());          // the _dev form may have run on any stream
    return pl->Jopt.download(J, sizeof(double) * pl->batch);
}

int sgusto_plan_info(sgusto_plan_t *pl, srh_kernel_info *info) {
    SRH_REQUIRE(pl && info, "sgusto_plan_info: null argument");
    SRH_REQUIRE(!pl->pending, "sgusto_plan_info: an asynchronous request is in flight on this plan (call sgusto_plan_solve_end first)");
    memset(info, 0, sizeof(*info));
    const QPDims &d = pl->C.dims;
    info->family = pl->lean ? 1 : 0;
    for (int i = 0; i < 6; ++i) info->lean_args[i] = pl->lean ? pl->lean_args[i] : 0;
#define X(SP, M, NX) if (info->fused_args[2] == -1 && variant_matches(d, SP, M, NX)) { info->fused_args[0] = SP ? 1 : 0; info->fused_args[1] = M; info->fused_args[2] = NX; }
    info->fused_args[2] = -1;
    SRH_QP_VARIANTS(X)
#undef X
    info->lds_bytes_lean = pl->lean ? (int32_t)pl->lean_lds : 0;
    info->lds_bytes_fused = (int32_t)pl->lds;
    info->threads = (pl->lean && d.lean_half) ? 256 : NTHREADS;      // (the half-size lean workgroup: two rollouts per CU)
    info->handed_over = -1;
    if (pl->solved) {
        // the last solve may sit on a caller's non-blocking stream (sgusto_plan_solve_dev): wait for the device, as
        // slocp_plan_info and sgusto_plan_costs do
        SRH_CHECK_HIP(hipDeviceSynchronize());
        if (pl->host_handed >= 0) info->handed_over = pl->host_handed;
        else SRH_CHECK_HIP(hipMemcpy(&info->handed_over, pl->handed.p, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (!pl->lean) info->handed_over = 0;
    }
    return SRH_OK;
}

int sgusto_plan_solve(sgusto_plan_t *pl, const double *x0, const double *u_init, const double *x_init,
                      const double *z, const double *zf, const double *u_des, double *xopt, double *uopt,
                      double *zopt, int32_t *iters, int32_t *status, double *trace) {
    SRH_REQUIRE(pl && x0 && u_init && x_init && xopt && uopt && zopt, "sgusto_plan_solve: null argument");
    SRH_REQUIRE(!pl->pending, "sgusto_plan_solve: an asynchronous request is in flight on this plan (call sgusto_plan_solve_end first)");
    const QPDims &d = pl->C.dims;
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz, B = pl->batch;
    // SRH_TRACE_SOLVE=<ms>: report the host-side segments of a call that takes longer than <ms> (attribution of latency outliers)
    static const double trace_ms = getenv("SRH_TRACE_SOLVE") ? atof(getenv("SRH_TRACE_SOLVE")) : -1.0;
    using clk = std::chrono::steady_clock;
    const auto t_in = clk::now();
    // Small batches (one receding-horizon solve at a time: the reference's closed-loop use, scp/ros.py:94-127): ZERO-COPY.  The
    // arguments go through one pinned, device-visible host block -- the kernels read the few KB of inputs and write the results
    // across PCIe themselves -- so that a solve is memcpy + ONE launch + ONE stream synchronisation instead of nine synchronous
    // hipMemcpy calls, a memset and two launches (~0.25 ms of runtime calls around a 0.5 ms kernel at the drivers' horizons).  The
    // fused kernel is launched only if a status word says a rollout was handed over.  SRH_GUSTO_NO_ZEROCOPY=1: the copying form.
    const bool no_zc = getenv("SRH_GUSTO_NO_ZEROCOPY") != nullptr;          // (per call: tests switch it)
    const PinLayout PL = pin_layout(pl);
    if (!no_zc && PL.total <= ((size_t)1 << 20)) {
        if (!pl->pin) {
            SRH_CHECK_HIP(hipHostMalloc((void **)&pl->pin, PL.total, hipHostMallocDefault));
            pl->pin_bytes = PL.total;
        }
        char *dp = nullptr;
        SRH_CHECK_HIP(hipHostGetDevicePointer((void **)&dp, pl->pin, 0));
        const size_t D = sizeof(double);
        memcpy(pl->pin + PL.x0, x0, D * B * n);
        memcpy(pl->pin + PL.u_init, u_init, D * B * N * m);
        memcpy(pl->pin + PL.x_init, x_init, D * B * (N + 1) * n);
        if (z) memcpy(pl->pin + PL.z, z, D * B * (N + 1) * nz);
        if (zf) memcpy(pl->pin + PL.zf, zf, D * B * nz);
        if (u_des) memcpy(pl->pin + PL.ud, u_des, D * B * N * m);
        auto dv = [&](size_t off) { return reinterpret_cast<double *>(dp + off); };
        int32_t *st_host = reinterpret_cast<int32_t *>(pl->pin + PL.status);
        auto launch = [&](int phase) {
            return solve_dev_impl(pl, dv(PL.x0), dv(PL.u_init), dv(PL.x_init), z ? dv(PL.z) : nullptr, zf ? dv(PL.zf) : nullptr,
                                  u_des ? dv(PL.ud) : nullptr, dv(PL.xopt), dv(PL.uopt), dv(PL.zopt), reinterpret_cast<int32_t *>(dp + PL.iters),
                                  reinterpret_cast<int32_t *>(dp + PL.status), trace ? dv(PL.trace) : nullptr, nullptr, phase);
        };
        int rc = launch(pl->lean ? 1 : 0);
        if (rc) return rc;
        const auto t_launch = clk::now();
        SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
        pl->host_handed = 0;
        if (pl->lean) {
            int handed = 0;
            for (size_t bi = 0; bi < B; ++bi) handed += st_host[bi] == LEAN_PENDING;
            pl->host_handed = handed;
            if (handed > 0) {
                if ((rc = launch(2))) return rc;
                SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
                pl->host_handed = handed;                      // (solve_dev_impl leaves it alone in phase 2)
            }
        }
        const auto t_sync = clk::now();
        memcpy(xopt, pl->pin + PL.xopt, D * B * (N + 1) * n);
        memcpy(uopt, pl->pin + PL.uopt, D * B * N * m);
        memcpy(zopt, pl->pin + PL.zopt, D * B * (N + 1) * nz);
        if (iters) memcpy(iters, pl->pin + PL.iters, sizeof(int32_t) * B);
        if (status) memcpy(status, pl->pin + PL.status, sizeof(int32_t) * B);
        if (trace) memcpy(trace, pl->pin + PL.trace, D * B * pl->par.max_trace * 4);
        if (trace_ms >= 0.0) {
            const auto ms = [](clk::time_point a, clk::time_point b2) { return std::chrono::duration<double, std::milli>(b2 - a).count(); };
            const auto t_out = clk::now();
            if (ms(t_in, t_out) > trace_ms)
                fprintf(stderr, "[sgusto_plan_solve, zero-copy] %.3f ms: staging + launch %.3f, wait for the kernels %.3f, copy out %.3f\n",
                        ms(t_in, t_out), ms(t_in, t_launch), ms(t_launch, t_sync), ms(t_sync, t_out));
        }
        return SRH_OK;
    }
    SRH_CHECK_HIP(hipMemcpy(pl->x0.p, x0, sizeof(double) * B * n, hipMemcpyHostToDevice));
    SRH_CHECK_HIP(hipMemcpy(pl->u_init.p, u_init, sizeof(double) * B * N * m, hipMemcpyHostToDevice));
    SRH_CHECK_HIP(hipMemcpy(pl->x_init.p, x_init, sizeof(double) * B * (N + 1) * n, hipMemcpyHostToDevice));
    if (z) SRH_CHECK_HIP(hipMemcpy(pl->z.p, z, sizeof(double) * B * (N + 1) * nz, hipMemcpyHostToDevice));
    if (zf) SRH_CHECK_HIP(hipMemcpy(pl->zf.p, zf, sizeof(double) * B * nz, hipMemcpyHostToDevice));
    if (u_des) SRH_CHECK_HIP(hipMemcpy(pl->ud.p, u_des, sizeof(double) * B * N * m, hipMemcpyHostToDevice));
    int rc = sgusto_plan_solve_dev(pl, pl->x0.as<double>(), pl->u_init.as<double>(), pl->x_init.as<double>(),
                                   z ? pl->z.as<double>() : nullptr, zf ? pl->zf.as<double>() : nullptr,
                                   u_des ? pl->ud.as<double>() : nullptr, pl->xopt.as<double>(), pl->uopt.as<double>(),
                                   pl->zopt.as<double>(), pl->iters.as<int32_t>(), pl->status.as<int32_t>(),
                                   trace ? pl->trace.as<double>() : nullptr, nullptr);
    if (rc) return rc;
    const auto t_launch = clk::now();
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    const auto t_sync = clk::now();
    if ((rc = pl->xopt.download(xopt, sizeof(double) * B * (N + 1) * n)) || (rc = pl->uopt.download(uopt, sizeof(double) * B * N * m)) ||
        (rc = pl->zopt.download(zopt, sizeof(double) * B * (N + 1) * nz)))
        return rc;
    if (iters && (rc = pl->iters.download(iters, sizeof(int32_t) * B))) return rc;
    if (status && (rc = pl->status.download(status, sizeof(int32_t) * B))) return rc;
    if (trace && (rc = pl->trace.download(trace, sizeof(double) * B * pl->par.max_trace * 4))) return rc;
    if (trace_ms >= 0.0) {
        const auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t_out = clk::now();
        if (ms(t_in, t_out) > trace_ms)
            fprintf(stderr, "[sgusto_plan_solve] %.3f ms: uploads + launches %.3f, wait for the kernels %.3f, downloads %.3f\n", ms(t_in, t_out),
                    ms(t_in, t_launch), ms(t_launch, t_sync), ms(t_sync, t_out));
    }
    return SRH_OK;
}

int sgusto_plan_prepare_async(sgusto_plan_t *pl) {
    SRH_REQUIRE(pl, "sgusto_plan_prepare_async: null plan");
    if (pl->astream) return SRH_OK;
    const PinLayout L = pin_layout(pl);
    SRH_CHECK_HIP(hipStreamCreateWithFlags(&pl->astream, hipStreamNonBlocking));
    SRH_CHECK_HIP(hipEventCreate(&pl->adone));
    SRH_CHECK_HIP(hipEventCreate(&pl->abegin));
    if (!pl->pin) {
        SRH_CHECK_HIP(hipHostMalloc((void **)&pl->pin, L.total, hipHostMallocDefault));
        pl->pin_bytes = L.total;
    }
    return SRH_OK;
}

int sgusto_plan_solve_begin(sgusto_plan_t *pl, const double *x0, const double *u_init, const double *x_init,
                            const double *z, const double *zf, const double *u_des, int want_trace) {
    SRH_REQUIRE(pl && x0 && u_init && x_init, "sgusto_plan_solve_begin: null argument");
    SRH_REQUIRE(!pl->pending, "sgusto_plan_solve_begin: a request is already in flight (call sgusto_plan_solve_end first)");
    const QPDims &d = pl->C.dims;
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz, B = pl->batch, D = sizeof(double);
    const PinLayout L = pin_layout(pl);
    { int rc0 = sgusto_plan_prepare_async(pl); if (rc0) return rc0; }
    // host -> pinned (the caller's arrays may change as soon as this returns), pinned -> HBM on the plan's stream
    auto stage = [&](size_t off, const double *src, size_t bytes, void *dst) -> hipError_t {
        memcpy(pl->pin + off, src, bytes);
        return hipMemcpyAsync(dst, pl->pin + off, bytes, hipMemcpyHostToDevice, pl->astream);
    };
    // from here on work is enqueued on the plan's stream: on any error the stream is drained before returning, so that
    // nothing is still writing into the pinned block / the plan's buffers while `pending` is false
    auto body = [&]() -> int {
        SRH_CHECK_HIP(hipEventRecord(pl->abegin, pl->astream));
        SRH_CHECK_HIP(stage(L.x0, x0, D * B * n, pl->x0.p));
        SRH_CHECK_HIP(stage(L.u_init, u_init, D * B * N * m, pl->u_init.p));
        SRH_CHECK_HIP(stage(L.x_init, x_init, D * B * (N + 1) * n, pl->x_init.p));
        if (z) SRH_CHECK_HIP(stage(L.z, z, D * B * (N + 1) * nz, pl->z.p));
        if (zf) SRH_CHECK_HIP(stage(L.zf, zf, D * B * nz, pl->zf.p));
        if (u_des) SRH_CHECK_HIP(stage(L.ud, u_des, D * B * N * m, pl->ud.p));
        pl->want_trace = want_trace != 0 && pl->par.max_trace > 0;
        pl->launching = true;
        int rc = sgusto_plan_solve_dev(pl, pl->x0.as<double>(), pl->u_init.as<double>(), pl->x_init.as<double>(),
                                       z ? pl->z.as<double>() : nullptr, zf ? pl->zf.as<double>() : nullptr,
                                       u_des ? pl->ud.as<double>() : nullptr, pl->xopt.as<double>(), pl->uopt.as<double>(),
                                       pl->zopt.as<double>(), pl->iters.as<int32_t>(), pl->status.as<int32_t>(),
                                       pl->want_trace ? pl->trace.as<double>() : nullptr, (void *)pl->astream);
        pl->launching = false;
        if (rc) return rc;
        auto back = [&](size_t off, const void *src, size_t bytes) {
            return hipMemcpyAsync(pl->pin + off, src, bytes, hipMemcpyDeviceToHost, pl->astream);
        };
        SRH_CHECK_HIP(back(L.xopt, pl->xopt.p, D * B * (N + 1) * n));
        SRH_CHECK_HIP(back(L.uopt, pl->uopt.p, D * B * N * m));
        SRH_CHECK_HIP(back(L.zopt, pl->zopt.p, D * B * (N + 1) * nz));
        SRH_CHECK_HIP(back(L.iters, pl->iters.p, sizeof(int32_t) * B));
        SRH_CHECK_HIP(back(L.status, pl->status.p, sizeof(int32_t) * B));
        if (pl->want_trace) SRH_CHECK_HIP(back(L.trace, pl->trace.p, D * B * (size_t)pl->par.max_trace * 4));
        SRH_CHECK_HIP(hipEventRecord(pl->adone, pl->astream));
        return SRH_OK;
    };
    const int rcb = body();
    if (rcb) {
        pl->launching = false;
        (void)hipStreamSynchronize(pl->astream);
        return rcb;
    }
    pl->pending = true;
    return SRH_OK;
}

int sgusto_plan_solve_done(sgusto_plan_t *pl, int *done) {
    SRH_REQUIRE(pl && done, "sgusto_plan_solve_done: null argument");
    if (!pl->pending) { *done = 1; return SRH_OK; }
    const hipError_t e = hipEventQuery(pl->adone);
    if (e == hipErrorNotReady) { *done = 0; return SRH_OK; }
    SRH_CHECK_HIP(e);
    *done = 1;
    return SRH_OK;
}

int sgusto_plan_solve_end(sgusto_plan_t *pl, double *xopt, double *uopt, double *zopt, int32_t *iters, int32_t *status,
                          double *trace) {
    SRH_REQUIRE(pl && xopt && uopt && zopt, "sgusto_plan_solve_end: null argument");
    SRH_REQUIRE(pl->pending, "sgusto_plan_solve_end: no request in flight");
    SRH_CHECK_HIP(hipEventSynchronize(pl->adone));
    pl->pending = false;
    { float ms = -1.0f; if (hipEventElapsedTime(&ms, pl->abegin, pl->adone) == hipSuccess) pl->last_ms = ms; else pl->last_ms = -1.0; }
    const QPDims &d = pl->C.dims;
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz, B = pl->batch, D = sizeof(double);
    const PinLayout L = pin_layout(pl);
    memcpy(xopt, pl->pin + L.xopt, D * B * (N + 1) * n);
    memcpy(uopt, pl->pin + L.uopt, D * B * N * m);
    memcpy(zopt, pl->pin + L.zopt, D * B * (N + 1) * nz);
    if (iters) memcpy(iters, pl->pin + L.iters, sizeof(int32_t) * B);
    if (status) memcpy(status, pl->pin + L.status, sizeof(int32_t) * B);
    if (trace && pl->want_trace) memcpy(trace, pl->pin + L.trace, D * B * (size_t)pl->par.max_trace * 4);
    return SRH_OK;
}

int sgusto_plan_last_async_ms(sgusto_plan_t *pl, double *ms) {
    SRH_REQUIRE(pl && ms, "sgusto_plan_last_async_ms: null argument");
    *ms = pl->last_ms;
    return SRH_OK;
}

int sgusto_solve(stpwl_t *h, const slocp_problem *prob, const sgusto_params *par, double dt, int64_t batch,
                 const double *x0, const double *u_init, const double *x_init, const double *z, const double *zf,
                 const double *u_des, const double *x_char, const double *f_char, double *xopt, double *uopt,
                 double *zopt, int32_t *iters, int32_t *status, double *trace, int max_trace) {
    sgusto_plan *pl = nullptr;
    int rc = sgusto_plan_create(&pl, h, prob, par, dt, batch, x_char, f_char, trace ? max_trace : 0);
    if (rc) return rc;
    rc = sgusto_plan_solve(pl, x0, u_init, x_init, z, zf, u_des, xopt, uopt, zopt, iters, status, trace);
    sgusto_plan_destroy(pl);
    return rc;
}


}  // extern "C"
