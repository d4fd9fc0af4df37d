// LOCP (horizon QP) and GuSTO (SCP outer loop) on the device: one workgroup per problem / rollout.
// Reference: sofacontrol/scp/locp.py (QP), sofacontrol/scp/gusto.py:283-487 (outer loop),
// sofacontrol/scp/models/tpwl.py:32-58 (model adapter).
// This unit: the batched QP entry point (slocp_solve); the GuSTO kernel and plan live in gusto.hip (two units: the
// kernel variants of each compile in parallel).
#include <memory>
#include "scp_types.h"

namespace {

template <bool SPLIT, int MSEL, int NSEL>
__global__ __launch_bounds__(NTHREADS) void locp_kernel(QPDims d, QPConst c, LocpBatch b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    qp::specialise<MSEL, NSEL>(d);             // compile-time n_u (and n_x) for everything inlined below
    QPLds L;
    qp_lds_carve(L, (lptr)smem, d, NTHREADS);          // qp::solve fills the constants of the layout it uses
    const size_t p = blockIdx.x;
    if (b.only_pending && b.status[p] != LEAN_PENDING) return;      // the lean kernel finished this one
    const size_t N = d.N, n = d.n, m = d.m;
    QPWork w;
    gptr wbase = (gptr)(b.work + p * b.work_stride);
    qp_carve(w, wbase, d);
    QPDyn dyn{(cgptr)(b.Ad + p * N * n * n), (cgptr)(b.AdT + p * N * n * n), (cgptr)(b.Bd + p * N * n * m),
              (cgptr)(b.BdT + p * N * n * m), (cgptr)(b.dd + p * N * n), (cgiptr)nullptr};
    QPData q{(cgptr)(b.x0 + p * n), (cgptr)(b.xk + p * (N + 1) * n), (cgptr)(b.z ? b.z + p * (N + 1) * d.nz : nullptr),
             (cgptr)(b.zf ? b.zf + p * d.nz : nullptr), (cgptr)(b.ud ? b.ud + p * N * m : nullptr), b.delta[p], b.omega[p],
             (gptr)((b.dbg && p == 0) ? b.dbg : nullptr)};
    double J;
    int it;
    const int st = qp::solve<SPLIT, MSEL, NSEL>(d, c, dyn, q, wbase, L, &J, &it, true, w);
    for (int e = SRH_TID; e < (N + 1) * n; e += blockDim.x) b.x[p * (N + 1) * n + e] = w.x[e];
    for (int e = SRH_TID; e < N * m; e += blockDim.x) b.u[p * N * m + e] = w.u[e];
    for (int e = SRH_TID; e <= N; e += blockDim.x) b.s[p * (N + 1) + e] = w.s[e];
    if (SRH_TID == 0) { b.J[p] = J; b.status[p] = st; b.iters[p] = it; }
}


__global__ void transpose_batch_kernel(const double *__restrict__ src, int64_t count, int rows, int cols,
                                       double *__restrict__ dst) {
    const int64_t b = blockIdx.x;
    if (b >= count) return;
    for (int e = SRH_TID; e < rows * cols; e += blockDim.x) {
        const int i = e / cols, j = e - i * cols;
        dst[b * rows * cols + (size_t)j * rows + i] = src[b * rows * cols + e];
    }
}


const void *locp_entry(const QPDims &d) {
#define X(SP, M, NX) if (variant_matches(d, SP, M, NX)) return (const void *)locp_kernel<SP, M, NX>;
    SRH_QP_VARIANTS(X)
#undef X
    return nullptr;
}


}  // namespace

extern "C" {

int slocp_condensed_info(const slocp_problem *prob, int *enabled, int *n_outputs, int *diag_input_hessian) {
    SRH_REQUIRE(prob, "slocp_condensed_info: null problem");
    QPConstHost C;
    int rc = build_consts(prob, C);
    if (rc) return rc;
    if (enabled) *enabled = C.dims.cond;
    if (n_outputs) *n_outputs = C.dims.po;
    if (diag_input_hessian) *diag_input_hessian = C.dims.diagD;
    return SRH_OK;
}

// ---- a resident LOCP: constants, horizon buffers, work blocks and result buffers live as long as the plan (what a host loop
// around the device QP -- GuSTO over an SSM or weighting-mode model, linear MPC, dU problems -- calls once per SCP iteration)
struct slocp_plan {
    QPConstHost C;
    int64_t batch = 0;
    size_t stride = 0;
    bool have_horizon = false, solved = false;
    bool lean = false;                  // the lean condensed kernel runs first (decided at creation)
    int lean_variant = -1;
    int lean_args[6] = {0, 0, 0, 0, 0, 0};
    size_t lean_lds = 0;
    srh::DevBuf handed;
    srh::DevBuf dA, dAT, dB, dBT, dD, dx0, dxk, ddel, dom, dz, dzf, dud, ox, ou, os, oJ, ost, oit, work, dbg;
};

// lean kernel, then the fused kernel for what the lean one hands over; everything on `st`, no synchronisation
static int slocp_plan_launch(slocp_plan *pl, const double *A, const double *B, const double *dv, bool new_horizon, const double *x0,
                             const double *xk, const double *delta, const double *omega, const double *z, const double *zf,
                             const double *ud, double *x, double *u, double *sl, double *J, int32_t *status, int32_t *iters,
                             hipStream_t st) {
    QPDims &d = pl->C.dims;
    const size_t N = d.N, n = d.n, m = d.m;
    const int64_t batch = pl->batch;
    int rc;
    if (new_horizon) {
        transpose_batch_kernel<<<(unsigned)(batch * N), 256, 0, st>>>(A, batch * N, (int)n, (int)n, pl->dAT.as<double>());
        transpose_batch_kernel<<<(unsigned)(batch * N), 256, 0, st>>>(B, batch * N, (int)n, (int)m, pl->dBT.as<double>());
        SRH_CHECK_HIP(hipGetLastError());
    }
    LocpBatch b{A, pl->dAT.as<double>(), B, pl->dBT.as<double>(), dv, x0, xk, delta, omega, z, zf, ud, x, u, sl, J, status, iters,
                pl->work.as<double>(), pl->stride, nullptr, 0, pl->handed.as<int32_t>()};
    const bool want_dbg = getenv("SRH_LOCP_TRACE") != nullptr;
    if (want_dbg) {
        if (!pl->dbg.p && (rc = pl->dbg.alloc(sizeof(double) * 8 * 64))) return rc;
        SRH_CHECK_HIP(hipMemsetAsync(pl->dbg.p, 0, sizeof(double) * 8 * 64, st));
        b.dbg = pl->dbg.as<double>();
    }
    const size_t lds = qp_kernel_lds_bytes(d);
    if ((rc = set_lds_limit(locp_entry(d), lds))) return rc;
    if (pl->lean) {
        // lean condensed kernel first; the fused kernel below then only takes what it could not finish (trust region
        // active at the minimiser, interior point not converged)
        const size_t llds = pl->lean_lds;
        SRH_CHECK_HIP(hipMemsetAsync(pl->handed.p, 0, sizeof(int32_t), st));
        if ((rc = lean_launch_locp(pl->lean_variant, d, pl->C.view(), b, (unsigned)batch, llds, st))) return rc;
        b.only_pending = 1;
        if (want_dbg) {
            SRH_CHECK_HIP(hipStreamSynchronize(st));
            std::vector<double> t(8 * 64);
            pl->dbg.download(t.data(), sizeof(double) * 8 * 64);
            fprintf(stderr, "[locp lean] j0 %d lds %zu; status %.0f iters %.0f inside %.0f\n", d.lean_j0, llds, t[8*61+1], t[8*61+2], t[8*61+3]);
            fprintf(stderr, "[locp lean] laps (SRH_PROFILE build): setup+rollout %.0f rows %.0f condense %.0f stage-factors %.0f gram %.0f cholesky %.0f grad+newton %.0f steps %.0f\n",
                    t[8*60], t[8*60+1], t[8*60+2], t[8*60+3], t[8*60+4], t[8*60+5], t[8*60+6], t[8*60+7]);
            fprintf(stderr, "[locp lean] newton laps: gradients %.0f gT_times(1) %.0f rhs+dinv %.0f g_times(1) %.0f k_solve %.0f gT_times(2) %.0f du %.0f g_times(2) %.0f\n",
                    t[8*59], t[8*59+1], t[8*59+2], t[8*59+3], t[8*59+4], t[8*59+5], t[8*59+6], t[8*59+7]);
            for (int i = 0; i < 59 && (t[8 * i + 3] != 0.0); ++i)
                fprintf(stderr, "[locp lean] it %2d mu %.3e rd %.3e rp %.3e (sd %.2e sp %.2e) a_aff %.3e sigma %.3e a %.3e\n", i, t[8 * i], t[8 * i + 1], t[8 * i + 2], t[8 * i + 3], t[8 * i + 4], t[8 * i + 5], t[8 * i + 6], t[8 * i + 7]);
            (void)hipMemset(pl->dbg.p, 0, sizeof(double) * 8 * 64);
        }
    }
    {
        bool launched = false;
#define X(SP, M, NX) if (!launched && variant_matches(d, SP, M, NX)) { locp_kernel<SP, M, NX><<<(unsigned)batch, NTHREADS, lds, st>>>(d, pl->C.view(), b); launched = true; }
        SRH_QP_VARIANTS(X)
#undef X
    }
    SRH_CHECK_HIP(hipGetLastError());
    pl->solved = true;
    if (want_dbg) {
        SRH_CHECK_HIP(hipStreamSynchronize(st));
        std::vector<double> t(8 * 64);
        pl->dbg.download(t.data(), sizeof(double) * 8 * 64);
        fprintf(stderr, "[locp] cond %d diagD %d po %d KT %d lds %zu; condensed path ran %.0f status %.0f iters %.0f inside %.0f\n", d.cond, d.diagD, d.po, d.KT, lds, t[8*61], t[8*61+1], t[8*61+2], t[8*61+3]);
        fprintf(stderr, "[locp] condensed laps (SRH_PROFILE build): setup+rollout %.0f rows %.0f condense %.0f stage-factors %.0f gram %.0f cholesky %.0f grad+newton %.0f steps %.0f\n",
                t[8*60], t[8*60+1], t[8*60+2], t[8*60+3], t[8*60+4], t[8*60+5], t[8*60+6], t[8*60+7]);
        fprintf(stderr, "[locp] newton laps: gradients %.0f gT_times(1) %.0f rhs+dinv %.0f g_times(1) %.0f k_solve %.0f gT_times(2) %.0f du %.0f g_times(2) %.0f\n",
                t[8*59], t[8*59+1], t[8*59+2], t[8*59+3], t[8*59+4], t[8*59+5], t[8*59+6], t[8*59+7]);
        fprintf(stderr, "[locp] time (shader clocks): init %.0f rows %.0f prepass %.0f ricc_full %.0f ricc_vec %.0f final %.0f\n", t[8*62], t[8*62+1], t[8*62+2], t[8*62+3], t[8*62+4], t[8*62+5]);
        fprintf(stderr, "[locp] riccati laps (SRH_PROFILE build): load %.0f W %.0f BtW %.0f Qu %.0f gain %.0f AtW+finish %.0f vec-backward %.0f forward %.0f\n",
                t[8*63], t[8*63+1], t[8*63+2], t[8*63+3], t[8*63+4], t[8*63+5], t[8*63+6], t[8*63+7]);
        fprintf(stderr, "[locp] riccati phases (shader clocks): load %.0f gemm1 %.0f gemm2 %.0f matvec %.0f chol+K %.0f Pnew %.0f (vec/other %.0f) fwd %.0f\n", t[8*63], t[8*63+1], t[8*63+2], t[8*63+3], t[8*63+4], t[8*63+5], t[8*63+6], t[8*63+7]);
        for (int i = 0; i < 62 && (t[8 * i + 3] != 0.0); ++i)
            fprintf(stderr, "[locp] it %2d mu %.3e rd %.3e rp %.3e (sd %.2e sp %.2e) a_aff %.3e sigma %.3e a %.3e\n", i, t[8 * i], t[8 * i + 1], t[8 * i + 2], t[8 * i + 3], t[8 * i + 4], t[8 * i + 5], t[8 * i + 6], t[8 * i + 7]);
    }
    return SRH_OK;
}

int slocp_plan_create(slocp_plan_t **out, const slocp_problem *prob, int64_t batch) {
    SRH_REQUIRE(out && prob, "slocp_plan_create: null argument");
    SRH_REQUIRE(batch > 0, "slocp_plan_create: batch must be positive");
    std::unique_ptr<slocp_plan> pl(new slocp_plan());
    int rc = build_consts(prob, pl->C);
    if (rc) return rc;
    pl->batch = batch;
    QPDims &d = pl->C.dims;
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz, D = sizeof(double), B = (size_t)batch;
    d.qc_off = (long long)((qp_work_doubles(d) + 3) & ~(size_t)3);
    pl->stride = ((size_t)d.qc_off + qc_work_doubles(d) + 3) & ~(size_t)3;
    if ((rc = pl->dA.alloc(D * B * N * n * n)) || (rc = pl->dAT.alloc(D * B * N * n * n)) || (rc = pl->dB.alloc(D * B * N * n * m)) ||
        (rc = pl->dBT.alloc(D * B * N * n * m)) || (rc = pl->dD.alloc(D * B * N * n)) || (rc = pl->dx0.alloc(D * B * n)) ||
        (rc = pl->dxk.alloc(D * B * (N + 1) * n)) || (rc = pl->ddel.alloc(D * B)) || (rc = pl->dom.alloc(D * B)) ||
        (rc = pl->dz.alloc(D * B * (N + 1) * nz)) || (rc = pl->dzf.alloc(D * B * nz)) || (rc = pl->dud.alloc(D * B * N * m)) ||
        (rc = pl->ox.alloc(D * B * (N + 1) * n)) || (rc = pl->ou.alloc(D * B * N * m)) || (rc = pl->os.alloc(D * B * (N + 1))) ||
        (rc = pl->oJ.alloc(D * B)) || (rc = pl->ost.alloc(sizeof(int32_t) * B)) || (rc = pl->oit.alloc(sizeof(int32_t) * B)) ||
        (rc = pl->work.alloc(D * pl->stride * B)) || (rc = pl->handed.alloc(sizeof(int32_t))))
        return rc;
    SRH_CHECK_HIP(hipMemset(pl->dxk.p, 0, D * B * (N + 1) * n));          // no trust region: its centre is never read, but defined
    SRH_CHECK_HIP(hipMemset(pl->handed.p, 0, sizeof(int32_t)));
    if (d.lean && !getenv("SRH_LOCP_NO_LEAN")) {
        pl->lean_variant = lean_select(d, pl->lean_args);
        SRH_REQUIRE(pl->lean_variant >= 0, "slocp_plan_create: no lean kernel instantiation for n_u = %d", d.m);
        pl->lean_lds = lean_kernel_lds_bytes(d);
        if ((rc = lean_prepare(pl->lean_variant, pl->lean_lds))) return rc;
        pl->lean = true;
    }
    *out = pl.release();
    return SRH_OK;
}

void slocp_plan_destroy(slocp_plan_t *pl) { delete pl; }

int slocp_plan_info(slocp_plan_t *pl, srh_kernel_info *info) {
    SRH_REQUIRE(pl && info, "slocp_plan_info: null argument");
    memset(info, 0, sizeof(*info));
    const QPDims &d = pl->C.dims;
    info->family = pl->lean ? 1 : 0;
    for (int i = 0; i < 6; ++i) info->lean_args[i] = pl->lean ? pl->lean_args[i] : 0;
    info->fused_args[2] = -1;
#define X(SP, M, NX) if (info->fused_args[2] == -1 && variant_matches(d, SP, M, NX)) { info->fused_args[0] = SP ? 1 : 0; info->fused_args[1] = M; info->fused_args[2] = NX; }
    SRH_QP_VARIANTS(X)
#undef X
    info->lds_bytes_lean = pl->lean ? (int32_t)pl->lean_lds : 0;
    info->lds_bytes_fused = (int32_t)qp_kernel_lds_bytes(d);
    info->threads = NTHREADS;
    info->handed_over = -1;
    if (pl->solved) {
        SRH_CHECK_HIP(hipDeviceSynchronize());      // the _dev form may have run on any stream
        SRH_CHECK_HIP(hipMemcpy(&info->handed_over, pl->handed.p, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (!pl->lean) info->handed_over = 0;
    }
    return SRH_OK;
}

int slocp_plan_solve(slocp_plan_t *pl, const double *Ad, const double *Bd, const double *dd, const double *x0, const double *xk,
                     const double *delta, const double *omega, const double *z, const double *zf, const double *u_des, double *x,
                     double *u, double *s, double *J, int32_t *status, int32_t *iters) {
    SRH_REQUIRE(pl && x0 && delta && omega && x && u && J && status, "slocp_plan_solve: null argument");
    const bool new_horizon = Ad != nullptr;
    SRH_REQUIRE((Ad != nullptr) == (Bd != nullptr) && (Ad != nullptr) == (dd != nullptr), "slocp_plan_solve: Ad, Bd, dd come together (all NULL: keep the resident horizon)");
    SRH_REQUIRE(new_horizon || pl->have_horizon, "slocp_plan_solve: no horizon resident yet");
    const QPDims &d = pl->C.dims;
    SRH_REQUIRE(!d.tr || xk || pl->have_horizon, "slocp_plan_solve: xk is required when the trust region is active");
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz, D = sizeof(double), B = (size_t)pl->batch;
    auto up = [&](srh::DevBuf &b, const void *src, size_t bytes) -> int {
        SRH_CHECK_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
        return SRH_OK;
    };
    int rc;
    if (new_horizon && ((rc = up(pl->dA, Ad, D * B * N * n * n)) || (rc = up(pl->dB, Bd, D * B * N * n * m)) || (rc = up(pl->dD, dd, D * B * N * n)))) return rc;
    if ((rc = up(pl->dx0, x0, D * B * n)) || (rc = up(pl->ddel, delta, D * B)) || (rc = up(pl->dom, omega, D * B))) return rc;
    if (xk && (rc = up(pl->dxk, xk, D * B * (N + 1) * n))) return rc;
    if (z && (rc = up(pl->dz, z, D * B * (N + 1) * nz))) return rc;
    if (zf && (rc = up(pl->dzf, zf, D * B * nz))) return rc;
    if (u_des && (rc = up(pl->dud, u_des, D * B * N * m))) return rc;
    if ((rc = slocp_plan_launch(pl, pl->dA.as<double>(), pl->dB.as<double>(), pl->dD.as<double>(), new_horizon, pl->dx0.as<double>(),
                                pl->dxk.as<double>(), pl->ddel.as<double>(), pl->dom.as<double>(), z ? pl->dz.as<double>() : nullptr,
                                zf ? pl->dzf.as<double>() : nullptr, u_des ? pl->dud.as<double>() : nullptr, pl->ox.as<double>(),
                                pl->ou.as<double>(), pl->os.as<double>(), pl->oJ.as<double>(), pl->ost.as<int32_t>(), pl->oit.as<int32_t>(),
                                nullptr)))
        return rc;
    pl->have_horizon = true;
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if ((rc = pl->ox.download(x, D * B * (N + 1) * n)) || (rc = pl->ou.download(u, D * B * N * m)) || (rc = pl->oJ.download(J, D * B)) ||
        (rc = pl->ost.download(status, sizeof(int32_t) * B)))
        return rc;
    if (s && (rc = pl->os.download(s, D * B * (N + 1)))) return rc;
    if (iters && (rc = pl->oit.download(iters, sizeof(int32_t) * B))) return rc;
    return SRH_OK;
}

int slocp_plan_solve_dev(slocp_plan_t *pl, const double *Ad_dev, const double *Bd_dev, const double *dd_dev, const double *x0_dev,
                         const double *xk_dev, const double *delta_dev, const double *omega_dev, const double *z_dev, const double *zf_dev,
                         const double *ud_dev, double *x_dev, double *u_dev, double *s_dev, double *J_dev, int32_t *status_dev,
                         int32_t *iters_dev, void *stream) {
    SRH_REQUIRE(pl && Ad_dev && Bd_dev && dd_dev && x0_dev && delta_dev && omega_dev && x_dev && u_dev && J_dev && status_dev,
                "slocp_plan_solve_dev: null argument");
    SRH_REQUIRE(!pl->C.dims.tr || xk_dev, "slocp_plan_solve_dev: xk is required when the trust region is active");
    return slocp_plan_launch(pl, Ad_dev, Bd_dev, dd_dev, true, x0_dev, xk_dev ? xk_dev : pl->dxk.as<double>(), delta_dev, omega_dev, z_dev,
                             zf_dev, ud_dev, x_dev, u_dev, s_dev ? s_dev : pl->os.as<double>(), J_dev, status_dev,
                             iters_dev ? iters_dev : pl->oit.as<int32_t>(), (hipStream_t)stream);
}

int slocp_solve(const slocp_problem *prob, int64_t batch, const double *Ad, const double *Bd, const double *dd,
                const double *x0, const double *xk, const double *delta, const double *omega, const double *z,
                const double *zf, const double *u_des, double *x, double *u, double *s, double *J,
                int32_t *status, int32_t *iters) {
    SRH_REQUIRE(prob && Ad && Bd && dd && x0 && delta && omega && x && u && J && status,
                "slocp_solve: null argument");
    SRH_REQUIRE(batch >= 0, "slocp_solve: negative batch");
    SRH_REQUIRE(!prob->tr_active || xk, "slocp_solve: xk is required when the trust region is active");
    if (batch == 0) return SRH_OK;
    slocp_plan *pl = nullptr;
    int rc = slocp_plan_create(&pl, prob, batch);
    if (rc) return rc;
    rc = slocp_plan_solve(pl, Ad, Bd, dd, x0, xk, delta, omega, z, zf, u_des, x, u, s, J, status, iters);
    slocp_plan_destroy(pl);
    return rc;
}


}  // extern "C"
