// LOCP (horizon QP) and GuSTO (SCP outer loop) on the device: one workgroup per problem / rollout.
// Reference: sofacontrol/scp/locp.py (QP), sofacontrol/scp/gusto.py:283-487 (outer loop),
// sofacontrol/scp/models/tpwl.py:32-58 (model adapter).
// This unit: the batched QP entry point (slocp_solve); the GuSTO kernel and plan live in gusto.hip (two units: the
// kernel variants of each compile in parallel).
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
    for (int e = threadIdx.x; e < (N + 1) * n; e += blockDim.x) b.x[p * (N + 1) * n + e] = w.x[e];
    for (int e = threadIdx.x; e < N * m; e += blockDim.x) b.u[p * N * m + e] = w.u[e];
    for (int e = threadIdx.x; e <= N; e += blockDim.x) b.s[p * (N + 1) + e] = w.s[e];
    if (threadIdx.x == 0) { b.J[p] = J; b.status[p] = st; b.iters[p] = it; }
}


__global__ void transpose_batch_kernel(const double *__restrict__ src, int64_t count, int rows, int cols,
                                       double *__restrict__ dst) {
    const int64_t b = blockIdx.x;
    if (b >= count) return;
    for (int e = threadIdx.x; e < rows * cols; e += blockDim.x) {
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

int slocp_solve(const slocp_problem *prob, int64_t batch, const double *Ad, const double *Bd, const double *dd,
                const double *x0, const double *xk, const double *delta, const double *omega, const double *z,
                const double *zf, const double *u_des, double *x, double *u, double *s, double *J,
                int32_t *status, int32_t *iters) {
    SRH_REQUIRE(prob && Ad && Bd && dd && x0 && delta && omega && x && u && J && status,
                "slocp_solve: null argument");
    SRH_REQUIRE(batch >= 0, "slocp_solve: negative batch");
    SRH_REQUIRE(!prob->tr_active || xk, "slocp_solve: xk is required when the trust region is active");
    if (batch == 0) return SRH_OK;
    QPConstHost C;
    int rc = build_consts(prob, C);
    if (rc) return rc;
    QPDims &d = C.dims;
    const size_t N = d.N, n = d.n, m = d.m, nz = d.nz;
    srh::DevBuf dA, dAT, dB, dBT, dD, dx0, dxk, ddel, dom, dz, dzf, dud, ox, ou, os, oJ, ost, oit, work;
    std::vector<double> xk0;
    if (!xk) xk0.assign(batch * (N + 1) * n, 0.0);
    if ((rc = dA.upload(Ad, sizeof(double) * batch * N * n * n)) || (rc = dAT.alloc(sizeof(double) * batch * N * n * n)) ||
        (rc = dB.upload(Bd, sizeof(double) * batch * N * n * m)) || (rc = dBT.alloc(sizeof(double) * batch * N * n * m)) ||
        (rc = dD.upload(dd, sizeof(double) * batch * N * n)) || (rc = dx0.upload(x0, sizeof(double) * batch * n)) ||
        (rc = dxk.upload(xk ? xk : xk0.data(), sizeof(double) * batch * (N + 1) * n)) ||
        (rc = ddel.upload(delta, sizeof(double) * batch)) || (rc = dom.upload(omega, sizeof(double) * batch)) ||
        (rc = ox.alloc(sizeof(double) * batch * (N + 1) * n)) || (rc = ou.alloc(sizeof(double) * batch * N * m)) ||
        (rc = os.alloc(sizeof(double) * batch * (N + 1))) || (rc = oJ.alloc(sizeof(double) * batch)) ||
        (rc = ost.alloc(sizeof(int32_t) * batch)) || (rc = oit.alloc(sizeof(int32_t) * batch)))
        return rc;
    if (z && (rc = dz.upload(z, sizeof(double) * batch * (N + 1) * nz))) return rc;
    if (zf && (rc = dzf.upload(zf, sizeof(double) * batch * nz))) return rc;
    if (u_des && (rc = dud.upload(u_des, sizeof(double) * batch * N * m))) return rc;
    d.qc_off = (long long)((qp_work_doubles(d) + 3) & ~(size_t)3);
    const size_t stride = ((size_t)d.qc_off + qc_work_doubles(d) + 3) & ~(size_t)3;
    if ((rc = work.alloc(sizeof(double) * stride * batch))) return rc;
    transpose_batch_kernel<<<(unsigned)(batch * N), 256>>>(dA.as<double>(), batch * N, (int)n, (int)n, dAT.as<double>());
    transpose_batch_kernel<<<(unsigned)(batch * N), 256>>>(dB.as<double>(), batch * N, (int)n, (int)m, dBT.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    LocpBatch b{dA.as<double>(), dAT.as<double>(), dB.as<double>(), dBT.as<double>(), dD.as<double>(), dx0.as<double>(),
                dxk.as<double>(), ddel.as<double>(), dom.as<double>(), z ? dz.as<double>() : nullptr,
                zf ? dzf.as<double>() : nullptr, u_des ? dud.as<double>() : nullptr, ox.as<double>(), ou.as<double>(),
                os.as<double>(), oJ.as<double>(), ost.as<int32_t>(), oit.as<int32_t>(), work.as<double>(), stride, nullptr, 0};
    srh::DevBuf dbg;
    const bool want_dbg = getenv("SRH_LOCP_TRACE") != nullptr;
    if (want_dbg) { if ((rc = dbg.alloc(sizeof(double) * 8 * 64))) return rc; (void)hipMemset(dbg.p, 0, sizeof(double) * 8 * 64); b.dbg = dbg.as<double>(); }
    const size_t lds = qp_kernel_lds_bytes(d);
    if ((rc = set_lds_limit(locp_entry(d), lds))) return rc;
    if (d.lean && !getenv("SRH_LOCP_NO_LEAN")) {
        // lean condensed kernel first; the fused kernel below then only takes what it could not finish (trust region
        // active at the minimiser, interior point not converged)
        const size_t llds = lean_kernel_lds_bytes(d);
        if ((rc = lean_prepare(d, llds)) || (rc = lean_launch_locp(d, C.view(), b, (unsigned)batch, llds, nullptr))) return rc;
        b.only_pending = 1;
        if (want_dbg) {
            SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
            std::vector<double> t(8 * 64);
            dbg.download(t.data(), sizeof(double) * 8 * 64);
            fprintf(stderr, "[locp lean] j0 %d lds %zu; status %.0f iters %.0f inside %.0f\n", d.lean_j0, llds, t[8*61+1], t[8*61+2], t[8*61+3]);
            fprintf(stderr, "[locp lean] laps (SRH_PROFILE build): setup+rollout %.0f rows %.0f condense %.0f stage-factors %.0f gram %.0f cholesky %.0f grad+newton %.0f steps %.0f\n",
                    t[8*60], t[8*60+1], t[8*60+2], t[8*60+3], t[8*60+4], t[8*60+5], t[8*60+6], t[8*60+7]);
            fprintf(stderr, "[locp lean] newton laps: gradients %.0f gT_times(1) %.0f rhs+dinv %.0f g_times(1) %.0f k_solve %.0f gT_times(2) %.0f du %.0f g_times(2) %.0f\n",
                    t[8*59], t[8*59+1], t[8*59+2], t[8*59+3], t[8*59+4], t[8*59+5], t[8*59+6], t[8*59+7]);
            for (int i = 0; i < 59 && (t[8 * i + 3] != 0.0); ++i)
                fprintf(stderr, "[locp lean] it %2d mu %.3e rd %.3e rp %.3e (sd %.2e sp %.2e) a_aff %.3e sigma %.3e a %.3e\n", i, t[8 * i], t[8 * i + 1], t[8 * i + 2], t[8 * i + 3], t[8 * i + 4], t[8 * i + 5], t[8 * i + 6], t[8 * i + 7]);
            (void)hipMemset(dbg.p, 0, sizeof(double) * 8 * 64);
        }
    }
    {
        bool launched = false;
#define X(SP, M, NX) if (!launched && variant_matches(d, SP, M, NX)) { locp_kernel<SP, M, NX><<<(unsigned)batch, NTHREADS, lds>>>(d, C.view(), b); launched = true; }
        SRH_QP_VARIANTS(X)
#undef X
    }
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if (want_dbg) {
        std::vector<double> t(8 * 64);
        dbg.download(t.data(), sizeof(double) * 8 * 64);
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
    if ((rc = ox.download(x, sizeof(double) * batch * (N + 1) * n)) || (rc = ou.download(u, sizeof(double) * batch * N * m)) ||
        (rc = oJ.download(J, sizeof(double) * batch)) || (rc = ost.download(status, sizeof(int32_t) * batch)))
        return rc;
    if (s && (rc = os.download(s, sizeof(double) * batch * (N + 1)))) return rc;
    if (iters && (rc = oit.download(iters, sizeof(int32_t) * batch))) return rc;
    return SRH_OK;
}


}  // extern "C"
