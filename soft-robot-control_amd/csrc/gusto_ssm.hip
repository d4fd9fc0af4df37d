// GuSTO on an SSM polynomial model, the whole solve inside ONE kernel launch: the loop the reference's hardware driver runs every
// control period (examples/hardware/diamond_SSM.py:353-361: N = 3, max_gusto_iters = 0 -- one QP per call, a 40 ms budget).
// Reference: sofacontrol/scp/gusto.py:283-487 (loop), sofacontrol/scp/models/ssm.py (adapter), sofacontrol/SSM/ssm.py:198-235
// (Jacobians of the polynomial maps, observer linearisation), sofacontrol/scp/locp.py:231-245, 312-329 (the QP with per-stage
// output maps z_k = H_k x_k + c_k).
//
// Rounds 2-5 served this model through a host loop: per SCP iteration two batched linearisation calls, an upload of the horizon,
// the QP kernel, a download, three more device calls for the model-accuracy test -- ~1.2 ms of round trips around a 6-state,
// 3-stage QP.  Here one workgroup per rollout does what gusto.hip's kernel does for a TPWL model, with the table gather replaced by
// the analytic linearisation of ssm_dev.h:
//   linearise   (A_k, B_k, d_k) = discretised Jacobians of f = R phi(x) + B u at (xbar_k, ubar_k), k < N  (ssm::linearize), and
//               (H_k, c_k) = observer linearisation at xbar_k, k <= N (ssm::observe).  The QP kernel has ONE constant performance
//               matrix, so the stage outputs are carried as extra states (the layout the host path has used since round 2,
//               sofacontrol_amd/scp/locp.py:_init_augmented): xa_k = [x_k ; zeta_k], zeta_{k+1} = H_{k+1}(A_k x_k + B_k u_k + d_k)
//               + c_{k+1}, H_a = [0 I], zero trust-region scale on zeta -- the same QP in (x, u, s).
//   QP          qp::solve (locp_dev.h / locp_cond.h) on the per-stage matrices just written to the rollout's work block
//   tests       trust region, model accuracy (continuous Jacobians at the old and at the new point, gusto.py:203-223), state rows,
//               convergence, acceptance -- gusto.py:371-473 with the same rules and order as gusto_kernel.
// The linearisation scratch and the QP's layouts share the workgroup's LDS (they never live at the same time).
#include "scp_types.h"
#include "ssm_host.h"
#include "locp_dense_u.h"

#include <memory>

namespace {

struct SsmGustoBatch {
    const double *x0, *u_init, *x_init, *z, *ud;
    const double *fs;                   // 1 / |f_char| (n)
    const double *Hm;                   // model.H (n_z x n): zopt = H xopt (gusto.py:486)
    const double *XA, *Xb;              // the state polyhedron as GuSTO.state_constraints_violated applies it (nXv x n), or null
    int nXv;
    double *xopt, *uopt, *zopt;
    int32_t *iters, *status;
    double *trace;
    double *work;
    size_t work_stride;
    int n;                              // model state dimension (the QP's state: n + n_obs)
    int mode;                           // discretisation of the model (SSM_FE ... SSM_DISCRETE_MAP)
    double *Jopt;
    int host_args;                      // the arguments sit in pinned host memory (zero-copy solve): copies in the work block
};

// offsets (doubles) of the SCP loop's arrays behind the QP's own work arrays
struct SsmGustoWork { size_t xk, uk, A, AT, B, BT, dd, xka, x0a, acc, x0c, zc, udc, Ac, fk, rec, lamd, end; };
__host__ __device__ inline SsmGustoWork ssm_gusto_work(const QPDims &d, int n) {
    SsmGustoWork g;
    const size_t N = d.N, na = d.n, m = d.m;
    g.xk = qp_work_doubles(d);
    g.uk = g.xk + (N + 1) * n;
    g.A = g.uk + N * m;
    g.AT = g.A + N * na * na;
    g.B = g.AT + N * na * na;
    g.BT = g.B + N * na * m;
    g.dd = g.BT + N * na * m;
    g.xka = g.dd + N * na;
    g.x0a = g.xka + (N + 1) * na;
    g.acc = g.x0a + na;
    g.x0c = g.acc + 2 * N;
    g.zc = g.x0c + n;
    g.udc = g.zc + (N + 1) * d.nz;
    g.Ac = g.udc + N * m;                       // continuous Jacobians and f of the current iterate (model-accuracy test, table path)
    g.fk = g.Ac + N * (size_t)n * n;
    g.rec = g.fk + N * (size_t)n;               // [0] = 1: w.u / w.lam hold a converged lean QP of this rollout's previous solve (warm_across)
    g.lamd = g.rec + 4;                         // 64 multipliers of the dense one-wave QP (qdu::solve) for its warm start
    g.end = g.lamd + 64;
    return g;
}

__host__ __device__ inline size_t ssm_gusto_scratch_doubles(const SsmDev &S) {
    const size_t n = S.n, m = S.m, no = S.no;
    return ssm::work_doubles(S.n, S.m, S.no, S.nr, S.ns) + 2 * (n + m) + 2 * (n * n + n * m + n) + 2 * no + no * n + 8;
}

// The model's maps for ONE WAVE (lanes 0..63, tables in LDS, no workgroup barrier): the N linearisations and N + 1 observer linearisations of a
// trajectory are independent of each other and each is far too small for a workgroup (n = 6: 36 Jacobian entries over 83 monomials), so
// every wave takes one of them -- ssm_dev.h's workgroup forms cost ~25 barriers per stage, taken one stage after the other.  Same formulas
// as ssm::basis_l / jacobians_l / discretize / observe; sums over the monomials in another split (rounding-level differences).
namespace ssmw {
__device__ __forceinline__ void fence() { ql::wave_fence(); }
__device__ inline void basis(const liptr ex, const liptr par, const liptr var, const liptr dm, const int *lv, int order, int nmon, int dim,
                             clptr x, lptr phi, lptr D, int lane) {
    for (int dg = 0; dg < order; ++dg) {
        for (int j = lv[dg] + lane; j < lv[dg + 1]; j += 64) { const int pj = par[j]; phi[j] = (pj < 0 ? 1.0 : phi[pj]) * x[var[j]]; }
        fence();
    }
    if (D != nullptr) {
        for (int e = lane; e < nmon * dim; e += 64) { const int qd = dm[e]; D[e] = qd == -1 ? 0.0 : (double)ex[e] * (qd >= 0 ? phi[qd] : 1.0); }
        fence();
    }
}
// continuous (A, B, d) of f = R phi(x) + B u at (x, u); w.f = f(x, u)
__device__ inline void jacobians(const SsmDev &S, const SsmLds &T, clptr x, clptr u, ssm::Work &w, lptr A, lptr Bm, lptr d, int lane) {
    const int n = S.n, m = S.m, nr = S.nr;
    basis(T.er, T.pr, T.vr, T.dmr, T.lvr, S.order_r, nr, n, x, w.phi, w.D, lane);
    const int g4 = lane & 3;
    for (int o0 = 0; o0 < n * (n + 1); o0 += 16) {
        const int o = o0 + (lane >> 2);
        const bool live = o < n * (n + 1);
        const int i = live ? o / (n + 1) : 0, j = live ? o - i * (n + 1) : 0;
        double acc = 0.0;
        if (live) {
            clptr r = T.R + (size_t)i * nr;
            clptr bb = j < n ? w.D + j : w.phi;
            const int bs = j < n ? n : 1;
            for (int k = g4; k < nr; k += 4) acc = fma(r[k], bb[(size_t)k * bs], acc);
        }
        acc = wg::group_sum<4>(acc);
        if (g4 == 0 && live) { if (j < n) A[i * n + j] = acc; else w.f[i] = acc; }
    }
    for (int e = lane; e < n * m; e += 64) Bm[e] = T.Bg[e];
    fence();
    for (int i = lane; i < n; i += 64) {
        double ax = 0.0, bu = 0.0;
        for (int k = 0; k < n; ++k) ax = fma(A[i * n + k], x[k], ax);
        for (int k = 0; k < m; ++k) bu = fma(T.Bg[i * m + k], u[k], bu);
        const double f = w.f[i] + bu;
        d[i] = f - ax - bu;
        w.f[i] = f;
    }
    fence();
}
// (A, B, d) continuous -> discrete, in place (ssm.py:279-301; ssm::discretize)
__device__ inline void discretize(const SsmDev &S, int mode, double dt, ssm::Work &w, lptr A, lptr Bm, lptr d, int lane) {
    const int n = S.n, m = S.m;
    if (mode == SSM_CONT || mode == SSM_DISCRETE_MAP) return;
    if (mode == SSM_FE) {
        for (int e = lane; e < n * n; e += 64) { const int i = e / n, j = e % n; A[e] = (i == j ? 1.0 : 0.0) + dt * A[e]; }
        for (int e = lane; e < n * m; e += 64) Bm[e] = dt * Bm[e];
        for (int e = lane; e < n; e += 64) d[e] = dt * d[e];
        fence();
        return;
    }
    const int ld = n | 1;
    const double h = mode == SSM_BE ? dt : 0.5 * dt;
    for (int e = lane; e < n * n; e += 64) {
        const int i = e / n, j = e % n;
        w.M1[i * ld + j] = (i == j ? 1.0 : 0.0) - h * A[e];
        w.M3[i * ld + j] = A[e];
    }
    fence();
    ssm::inverse_wave(w.M1, w.M2, n, ld);
    ssm::inverse_wave(w.M3, w.M4, n, ld);
    fence();
    if (mode == SSM_BIL) {
        for (int e = lane; e < n * n; e += 64) {
            const int i = e / n, j = e % n;
            double sacc = 0.0;
            for (int k = 0; k < n; ++k) sacc = fma((i == k ? 1.0 : 0.0) + h * A[i * n + k], w.M2[k * ld + j], sacc);
            w.M1[i * ld + j] = sacc;
        }
        fence();
        for (int e = lane; e < n * n; e += 64) w.M2[(e / n) * ld + e % n] = w.M1[(e / n) * ld + e % n];
        fence();
    }
    for (int e = lane; e < n * n; e += 64) {                      // M3 = sep = inv(A_c) (A_d - I)
        const int i = e / n, j = e % n;
        double sacc = 0.0;
        for (int k = 0; k < n; ++k) sacc = fma(w.M4[i * ld + k], w.M2[k * ld + j] - (k == j ? 1.0 : 0.0), sacc);
        w.M3[i * ld + j] = sacc;
    }
    fence();
    for (int e = lane; e < n * n; e += 64) A[e] = w.M2[(e / n) * ld + e % n];
    for (int i = lane; i < n; i += 64) {
        double sacc = 0.0;
        for (int k = 0; k < n; ++k) sacc = fma(w.M3[i * ld + k], d[k], sacc);
        w.f[i] = sacc;
    }
    for (int e = lane; e < n * m; e += 64) {
        const int i = e / m, j = e % m;
        double sacc = 0.0;
        for (int k = 0; k < n; ++k) sacc = fma(w.M3[i * ld + k], Bm[k * m + j], sacc);
        w.M1[i * ld + j] = sacc;
    }
    fence();
    for (int e = lane; e < n * m; e += 64) Bm[e] = w.M1[(e / m) * ld + e % m];
    for (int e = lane; e < n; e += 64) d[e] = w.f[e];
    fence();
}
// z = C(x), H = dC/dx (no x n), c = z - H x  (ssm.py:220-235)
__device__ inline void observe(const SsmDev &S, const SsmLds &T, clptr x, ssm::Work &w, lptr z, lptr Hj, lptr cc, int lane) {
    const int n = S.n, ns = S.ns, no = S.no;
    basis(T.es, T.ps, T.vs, T.dms, T.lvs, S.order_s, ns, no, x, w.phi, w.D, lane);
    for (int e = lane; e < no * (n + 1); e += 64) {
        const int i = e / (n + 1), j = e - i * (n + 1);
        clptr wr = T.W + (size_t)i * ns;
        double acc = 0.0;
        if (j < n) for (int k = 0; k < ns; ++k) acc = fma(wr[k], w.D[(size_t)k * no + j], acc);
        else for (int k = 0; k < ns; ++k) acc = fma(wr[k], w.phi[k], acc);
        if (j < n) Hj[i * n + j] = acc; else z[i] = acc;
    }
    fence();
    for (int i = lane; i < no; i += 64) {
        double sx = 0.0;
        for (int k = 0; k < n; ++k) sx = fma(Hj[i * n + k], x[k], sx);
        cc[i] = z[i] - sx;
    }
    fence();
}
}  // namespace ssmw

// GXL > 0: the QP without its trust-region rows runs on the lean one-wave interior point first (ql::ipm_wave: K is one 16 x 16 tile at the
// driver's N = 3, 31 k clocks per interior-point iteration against 70 k of the eight-wave forms -- DESIGN.md section 13); qp::solve takes over
// when that minimiser leaves the trust region or the interior point does not converge.  GXL = lanes per stage for the state rows (ql::ipm_box).
// Four waves per rollout (one per SIMD): each gets the 512 registers of its SIMD lane, so what the register allocator cannot keep lands in the
// accumulation registers instead of in scratch memory -- the models here are small (n <= 64, N <= 8), none of the phases has work for eight waves.
// The LDS layouts stay those of an NTHREADS workgroup (the host sizes them that way).
constexpr int SSM_THREADS = 256;
template <bool SPLIT, int MSEL, int GXL>
__global__ __launch_bounds__(SSM_THREADS) void gusto_ssm_kernel(QPDims d, QPConst c, SsmDev S, GustoPar par, SsmGustoBatch b, int red_off, int tab_off, int dense_u, int task_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    qp::specialise<MSEL, 0>(d);
    long long prof[32] = {0};
    QPLds L;
    qp_lds_carve(L, (lptr)smem, d, NTHREADS);
    const size_t p = blockIdx.x;
    const int N = d.N, na = d.n, m = d.m, nz = d.nz, n = b.n, no = na - n;       // no = 0: linear output map (model.H), else S.no
    int tid = SRH_TID;
    const int nt = blockDim.x;
    gptr base = (gptr)(b.work + p * b.work_stride);
    QPWork w;
    qp_carve(w, base, d);
    const SsmGustoWork gw = ssm_gusto_work(d, n);
    gptr xk = base + gw.xk, uk = base + gw.uk, Ag = base + gw.A, ATg = base + gw.AT, Bg = base + gw.B, BTg = base + gw.BT;
    gptr ddg = base + gw.dd, xka = base + gw.xka, x0a = base + gw.x0a, accb = base + gw.acc;
    // linearisation scratch (aliases the QP's LDS; qp::solve re-initialises its layout when it finds `ready` false)
    ssm::Work sw;
    ssm::carve(sw, (lptr)smem, S);
    lptr xs = (lptr)smem + ssm::work_doubles(S.n, S.m, S.no, S.nr, S.ns);
    lptr us = xs + n, Al = us + m, Bl = Al + (size_t)n * n, dl = Bl + (size_t)n * m, zs = dl + n, cs = zs + S.no, Hl = cs + S.no;
    lptr A2 = Hl + (size_t)S.no * n, B2 = A2 + (size_t)n * n, d2 = B2 + (size_t)n * m, x2 = d2 + n, u2 = x2 + n;
    lptr red = (lptr)smem + red_off;        // reduction scratch behind both layouts
    // The model's coefficient and exponent tables in LDS behind everything else, staged ONCE per launch (ssm::stage, as the iLQR kernel does):
    // with the rows of R read from L2 inside latency-bound dot products one linearisation is ~25 k clocks, from LDS a few thousand; the
    // real-time iteration of the hardware driver does N of them + N + 1 observer linearisations + N dynamics evaluations per call.
    // Only for the fe / be / bil discretisations (they and the model-accuracy test share the continuous coefficients); tab_off = 0: global path.
    const bool tab = tab_off > 0;
    SsmLds T{};
    if (tab) ssm::stage(T, (lptr)smem + tab_off, S, false, 0);
    gptr Acg = base + gw.Ac, fkg = base + gw.fk;
    // z = C(x), H = dC/dx, c = z - H x at xs (ssm::observe with the tables in LDS)
    auto observe_tab = [&]() {
        const int ns = S.ns, sno = S.no;
        ssm::basis_l(T.es, T.ps, T.vs, T.dms, T.lvs, S.order_s, ns, sno, xs, sw.phi, sw.D);
        for (int e = tid; e < sno * (n + 1); e += nt) {
            const int i = e / (n + 1), j = e - i * (n + 1);
            clptr wr = T.W + (size_t)i * ns;
            double acc = 0.0;
            if (j < n) for (int k = 0; k < ns; ++k) acc = fma(wr[k], sw.D[(size_t)k * sno + j], acc);
            else for (int k = 0; k < ns; ++k) acc = fma(wr[k], sw.phi[k], acc);
            if (j < n) Hl[i * n + j] = acc; else zs[i] = acc;
        }
        __syncthreads();
        for (int i = tid; i < sno; i += nt) {
            double sx = 0.0;
            for (int k = 0; k < n; ++k) sx = fma(Hl[i * n + k], xs[k], sx);
            cs[i] = zs[i] - sx;
        }
        __syncthreads();
    };

    cgptr x0 = (cgptr)(b.x0 + p * n);
    cgptr zp = (cgptr)(b.z ? b.z + p * (size_t)(N + 1) * nz : nullptr);
    cgptr udp = (cgptr)(b.ud ? b.ud + p * (size_t)N * m : nullptr);
    if (b.host_args) {
        gptr x0c = base + gw.x0c, zc = base + gw.zc, udc = base + gw.udc;
        for (int e = tid; e < n; e += nt) x0c[e] = x0[e];
        if (zp) for (int e = tid; e < (N + 1) * nz; e += nt) zc[e] = zp[e];
        if (udp) for (int e = tid; e < N * m; e += nt) udc[e] = udp[e];
        x0 = (cgptr)x0c;
        if (zp) zp = (cgptr)zc;
        if (udp) udp = (cgptr)udc;
    }
    for (int e = tid; e < (N + 1) * n; e += nt) xk[e] = b.x_init[p * (size_t)(N + 1) * n + e];
    for (int e = tid; e < N * m; e += nt) uk[e] = b.u_init[p * (size_t)N * m + e];
    __syncthreads();

    // (A, B, d)_k and (H, c)_k of the trajectory (xk, uk) -> the augmented per-stage matrices of the QP, both orientations
    // task t < N: linearisation of stage t; task N + k: observer linearisation at xbar_k (k = 0..N).  Scratch of task t at smem + t * task_stride:
    // [ssm::Work | x (n) | u (m) | LIN: A (n n), B (n m), d (n)  /  OBS: H (no n), c (no), z (no)]
    const size_t wdb = ssm::work_doubles(S.n, S.m, S.no, S.nr, S.ns);
    auto task_base = [&](int t) { return (lptr)smem + (size_t)t * task_stride; };
    auto linearise_par = [&]() {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nt >> 6, lane = tid & 63;
        const int ntask = no > 0 ? 2 * N + 1 : N;
        for (int t = wave; t < ntask; t += nwv) {
            lptr tb = task_base(t);
            ssm::Work tw;
            ssm::carve(tw, tb, S);
            lptr tx = tb + wdb, tu = tx + n, r0 = tu + m;
            if (t < N) {
                for (int e = lane; e < n; e += 64) tx[e] = xk[(size_t)t * n + e];
                for (int e = lane; e < m; e += 64) tu[e] = uk[(size_t)t * m + e];
                ssmw::fence();
                lptr tA = r0, tB = tA + (size_t)n * n, td = tB + (size_t)n * m;
                ssmw::jacobians(S, T, tx, tu, tw, tA, tB, td, lane);
                for (int e = lane; e < n * n; e += 64) Acg[(size_t)t * n * n + e] = tA[e];
                for (int e = lane; e < n; e += 64) fkg[(size_t)t * n + e] = tw.f[e];
                ssmw::fence();
                ssmw::discretize(S, b.mode, par.dt, tw, tA, tB, td, lane);
            } else {
                const int k = t - N;
                for (int e = lane; e < n; e += 64) tx[e] = xk[(size_t)k * n + e];
                ssmw::fence();
                lptr tH = r0, tc = tH + (size_t)S.no * n, tz = tc + S.no;
                ssmw::observe(S, T, tx, tw, tz, tH, tc, lane);
            }
        }
        __syncthreads();
        // the augmented stage matrices from the tasks' results
        auto Hof = [&](int k) { return task_base(N + k) + wdb + n + m; };                    // H_k; c_k behind it
        for (int e = tid; e < na; e += nt) {
            double v0, vk;
            if (e < n) { v0 = x0[e]; vk = xk[e]; }
            else {
                clptr H0 = Hof(0), c0 = H0 + (size_t)S.no * n;
                v0 = c0[e - n]; vk = c0[e - n];
                for (int j = 0; j < n; ++j) { v0 = fma(H0[(e - n) * n + j], x0[j], v0); vk = fma(H0[(e - n) * n + j], xk[j], vk); }
            }
            x0a[e] = v0; xka[e] = vk;
        }
        for (int e = tid; e < N * na * na; e += nt) {
            const int k = e / (na * na), r = e - k * na * na, i = r / na, j = r - i * na;
            clptr tA = task_base(k) + wdb + n + m;
            double v = 0.0;
            if (j < n) {
                if (i < n) v = tA[i * n + j];
                else { clptr Hn = Hof(k + 1); for (int l = 0; l < n; ++l) v = fma(Hn[(i - n) * n + l], tA[l * n + j], v); }
            }
            Ag[e] = v; ATg[(size_t)k * na * na + (size_t)j * na + i] = v;
        }
        for (int e = tid; e < N * na * m; e += nt) {
            const int k = e / (na * m), r = e - k * na * m, i = r / m, j = r - i * m;
            clptr tB = task_base(k) + wdb + n + m + (size_t)n * n;
            double v = 0.0;
            if (i < n) v = tB[i * m + j];
            else { clptr Hn = Hof(k + 1); for (int l = 0; l < n; ++l) v = fma(Hn[(i - n) * n + l], tB[l * m + j], v); }
            Bg[e] = v; BTg[(size_t)k * na * m + (size_t)j * na + i] = v;
        }
        for (int e = tid; e < N * na; e += nt) {
            const int k = e / na, i = e - k * na;
            clptr td = task_base(k) + wdb + n + m + (size_t)n * n + (size_t)n * m;
            double v, xv;
            if (i < n) { v = td[i]; xv = xk[(size_t)(k + 1) * n + i]; }
            else {
                clptr Hn = Hof(k + 1), cn = Hn + (size_t)S.no * n;
                v = cn[i - n]; xv = cn[i - n];
                for (int l = 0; l < n; ++l) { v = fma(Hn[(i - n) * n + l], td[l], v); xv = fma(Hn[(i - n) * n + l], xk[(size_t)(k + 1) * n + l], xv); }
            }
            ddg[e] = v;
            xka[(size_t)(k + 1) * na + i] = xv;
        }
        __syncthreads();
    };
    auto linearise_all = [&]() {
        if (tab && task_stride > 0) { linearise_par(); return; }
        if (no > 0) {
            for (int e = tid; e < n; e += nt) xs[e] = xk[e];
            __syncthreads();
            if (tab) observe_tab(); else ssm::observe(S, xs, sw, zs, Hl, cs);
            for (int e = tid; e < na; e += nt) {
                double v0, vk;
                if (e < n) { v0 = x0[e]; vk = xs[e]; }
                else {
                    v0 = cs[e - n];
                    for (int j = 0; j < n; ++j) v0 = fma(Hl[(e - n) * n + j], x0[j], v0);
                    vk = cs[e - n];
                    for (int j = 0; j < n; ++j) vk = fma(Hl[(e - n) * n + j], xs[j], vk);
                }
                x0a[e] = v0; xka[e] = vk;
            }
        } else {
            for (int e = tid; e < n; e += nt) { x0a[e] = x0[e]; xka[e] = xk[e]; }
        }
        __syncthreads();
        for (int k = 0; k < N; ++k) {
            for (int e = tid; e < n; e += nt) xs[e] = xk[(size_t)k * n + e];
            for (int e = tid; e < m; e += nt) us[e] = uk[(size_t)k * m + e];
            __syncthreads();
            if (tab) {
                ssm::jacobians_l(S, T, false, xs, us, sw, Al, n, Bl, dl);          // continuous (A, B, d) and f(xbar_k, ubar_k) in sw.f
                for (int e = tid; e < n * n; e += nt) Acg[(size_t)k * n * n + e] = Al[e];
                for (int e = tid; e < n; e += nt) fkg[(size_t)k * n + e] = sw.f[e];
                __syncthreads();
                ssm::discretize(S, b.mode, par.dt, sw, Al, n, Bl, dl);
            } else {
                ssm::linearize(S, b.mode, par.dt, xs, us, sw, Al, n, Bl, dl);
            }
            if (no > 0) {
                for (int e = tid; e < n; e += nt) xs[e] = xk[(size_t)(k + 1) * n + e];
                __syncthreads();
                if (tab) observe_tab(); else ssm::observe(S, xs, sw, zs, Hl, cs);
            }
            gptr Ak = Ag + (size_t)k * na * na, ATk = ATg + (size_t)k * na * na, Bk = Bg + (size_t)k * na * m, BTk = BTg + (size_t)k * na * m;
            for (int e = tid; e < na * na; e += nt) {
                const int i = e / na, j = e - i * na;
                double v = 0.0;
                if (j < n) {
                    if (i < n) v = Al[i * n + j];
                    else for (int l = 0; l < n; ++l) v = fma(Hl[(i - n) * n + l], Al[l * n + j], v);
                }
                Ak[e] = v; ATk[(size_t)j * na + i] = v;
            }
            for (int e = tid; e < na * m; e += nt) {
                const int i = e / m, j = e - i * m;
                double v = 0.0;
                if (i < n) v = Bl[i * m + j];
                else for (int l = 0; l < n; ++l) v = fma(Hl[(i - n) * n + l], Bl[l * m + j], v);
                Bk[e] = v; BTk[(size_t)j * na + i] = v;
            }
            for (int i = tid; i < na; i += nt) {
                double v;
                if (i < n) v = dl[i];
                else { v = 0.0; for (int l = 0; l < n; ++l) v = fma(Hl[(i - n) * n + l], dl[l], v); v += cs[i - n]; }
                ddg[(size_t)k * na + i] = v;
                double xv;
                if (i < n) xv = xk[(size_t)(k + 1) * n + i];
                else { xv = cs[i - n]; for (int l = 0; l < n; ++l) xv = fma(Hl[(i - n) * n + l], xs[l], xv); }
                xka[(size_t)(k + 1) * na + i] = xv;
            }
            __syncthreads();
        }
    };
    long long lap0 = clock64(), lap_lin = 0, lap_qp = 0, lap_tests = 0;
    linearise_all();
    lap_lin += clock64() - lap0;

    QPDyn dyn{(cgptr)Ag, (cgptr)ATg, (cgptr)Bg, (cgptr)BTg, (cgptr)ddg, (cgiptr)nullptr};
    gptr rec = base + gw.rec;
    // have_warm: the work block holds the minimiser and multipliers of a converged lean QP -- of this solve, or (warm_across: the reference's
    // warm_start=True keeps its solver state between solves, locp.py:181) of the rollout's previous solve
    bool have_warm = (GXL > 0 || dense_u) && par.warm_across != 0 && rec[0] == 1.0;
    gptr lamd = base + gw.lamd;
    double qdbg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};              // phase clocks of the last dense one-wave QP (SRH_GUSTO_TRACE_QIT=1: trace row 1)
    double delta = par.delta0, omega = par.omega0;
    double J_prev = INFINITY, d_prev = INFINITY, o_prev = INFINITY;
    bool converged = false, tr_hot = false;
    int itr = 0, status = 0;
    while (itr <= par.max_iters && !converged && omega <= par.omega_max) {
        tid = SRH_TID;
        QPData q{(cgptr)x0a, (cgptr)xka, zp, (cgptr)nullptr, udp, delta, omega, (gptr)nullptr};
        double J;
        int qit, qpass = -1;
        __syncthreads();
        lap0 = clock64();
        int st = -1;
        if (dense_u && !tr_hot) {
            // N n_u <= 16: the QP without its trust-region rows in the space of the inputs on one wave, whatever the cost's rank (locp_dense_u.h)
            for (int attempt = 0; attempt < 2; ++attempt) {
                const bool warm = have_warm && attempt == 0;
                st = qdu::solve(d, c, dyn, q, w, (lptr)smem, lamd, &J, &qit, warm ? 1 : 0, qdbg);
                if (st == 0 || st == 100 || !warm) break;
            }
            have_warm = st == 0;
        }
        if constexpr (GXL > 0) {
            if (!tr_hot && !dense_u) {
                ql::Lds LL;
                ql::lds_carve(LL, (lptr)smem, d, NTHREADS);
                if (tid == 0) LL.flag[2] = 0;              // (the LDS was used by the linearisation: nothing condensed is left)
                __syncthreads();
                for (int attempt = 0; attempt < 2; ++attempt) {      // a warm start that does not reach the tolerances is repeated cold (lean.hip)
                    const bool warm = have_warm && attempt == 0;
                    st = ql::solve_qp<MSEL, 0, GXL, -1, 0>(d, c, dyn, q, base, LL, &J, &qit, w, prof, warm ? 1 : 0);
                    __syncthreads();
                    if (st == 0 || st == 100 || !warm) break;
                    if (tid == 0) LL.flag[2] = 0;
                    __syncthreads();
                }
                have_warm = st == 0;
            }
        }
        if (st != 0) {
            have_warm = false;                             // (qp::solve carves the work block its own way)
            qp_lds_carve(L, (lptr)smem, d, NTHREADS);      // the linearisation / the tests / the lean attempt used the LDS: the QP starts from its own layout
            st = qp::solve<SPLIT, MSEL, 0>(d, c, dyn, q, base, L, &J, &qit, true, w, tr_hot || st == 100, false, &qpass);
        }
        if (st != 0) { status = 1; break; }                // gusto.py:357-365: keep the previous iterate
        __syncthreads();
        lap_qp += clock64() - lap0; lap0 = clock64();
        // trust region (gusto.py:174-183) on the model's own states
        double md = 0.0;
        for (int e = tid; e < (N + 1) * n; e += nt) {
            const int k = e / n, j = e - k * n;
            md = fmax(md, fabs(c.xs[j] * (w.x[(size_t)k * na + j] - xk[e])));
        }
        md = wg::reduce(md, 1, red);
        const bool tr_ok = !(md - delta > par.epsilon);
        const bool on_boundary = md >= delta * (1.0 - 1e-9);
        bool new_solution = false;
        double rho_k = -1.0;
        const double d_cur = delta, o_cur = omega;
        if (tr_ok) {
            // model accuracy (gusto.py:203-223): f = A x + B u + d with the CONTINUOUS Jacobians at each point (models/ssm.py:35-54)
            if (tab && task_stride > 0) {
                // one wave per stage: monomials at the new point, f = R phi + B u on one lane per row, against the stored (A, B, f) of the old point
                const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nt >> 6, lane = tid & 63;
                for (int i = wave; i < N; i += nwv) {
                    lptr tb = task_base(i);
                    ssm::Work tw;
                    ssm::carve(tw, tb, S);
                    lptr txo = tb + wdb, tuo = txo + n, txn = tuo + m, tun = txn + n;
                    for (int e = lane; e < n; e += 64) { txo[e] = xk[(size_t)i * n + e]; txn[e] = w.x[(size_t)i * na + e]; }
                    for (int e = lane; e < m; e += 64) { tuo[e] = uk[(size_t)i * m + e]; tun[e] = w.u[(size_t)i * m + e]; }
                    ssmw::fence();
                    ssmw::basis(T.er, T.pr, T.vr, T.dmr, T.lvr, S.order_r, S.nr, n, txn, tw.phi, (lptr) nullptr, lane);
                    double de = 0.0, da = 0.0;
                    if (lane < n) {
                        const int r = lane;
                        clptr rr = T.R + (size_t)r * S.nr;
                        double f0 = 0.0, f1 = 0.0, fl = 0.0;
                        int k = 0;
                        for (; k + 1 < S.nr; k += 2) { f0 = fma(rr[k], tw.phi[k], f0); f1 = fma(rr[k + 1], tw.phi[k + 1], f1); }
                        if (k < S.nr) f0 = fma(rr[k], tw.phi[k], f0);
                        double bf = 0.0, bl = 0.0;
                        for (int j = 0; j < m; ++j) { bf = fma(T.Bg[r * m + j], tun[j], bf); bl = fma(T.Bg[r * m + j], tun[j] - tuo[j], bl); }
                        for (int j = 0; j < n; ++j) fl = fma(Acg[(size_t)i * n * n + r * n + j], txn[j] - txo[j], fl);
                        const double fv = (f0 + f1) + bf, fa = fkg[(size_t)i * n + r] + fl + bl;
                        const double fsr = b.fs[r];
                        de = fsr * (fv - fa); da = fsr * fa;
                    }
                    const double e2 = wg::wave_sum(de * de), a2 = wg::wave_sum(da * da);
                    if (lane == 0) { accb[2 * i] = par.dt * sqrt(e2); accb[2 * i + 1] = par.dt * sqrt(a2); }
                }
                __syncthreads();
            }
            for (int i = (tab && task_stride > 0) ? N : 0; i < N; ++i) {
                for (int e = tid; e < n; e += nt) { xs[e] = xk[(size_t)i * n + e]; x2[e] = w.x[(size_t)i * na + e]; }
                for (int e = tid; e < m; e += nt) { us[e] = uk[(size_t)i * m + e]; u2[e] = w.u[(size_t)i * m + e]; }
                __syncthreads();
                if (tab) {
                    // f at the new point = R phi(x) + B u straight from the tables; the Jacobians and f of the old point are the ones the
                    // linearisation of this iterate stored (same point: gusto.py:212 re-evaluates them)
                    ssm::basis_l(T.er, T.pr, T.vr, T.dmr, T.lvr, S.order_r, S.nr, n, x2, sw.phi, (lptr) nullptr);
                    // one lane per row of f (n <= 64: the first wave), the two sums of squares by a wave reduction
                    if (tid < 64) {
                        double de = 0.0, da = 0.0;
                        if (tid < n) {
                            const int r = tid;
                            clptr rr = T.R + (size_t)r * S.nr;
                            double f0 = 0.0, f1 = 0.0, fl = 0.0;
                            int k = 0;
                            for (; k + 1 < S.nr; k += 2) { f0 = fma(rr[k], sw.phi[k], f0); f1 = fma(rr[k + 1], sw.phi[k + 1], f1); }
                            if (k < S.nr) f0 = fma(rr[k], sw.phi[k], f0);
                            double bf = 0.0, bl = 0.0;
                            for (int j = 0; j < m; ++j) { bf = fma(T.Bg[r * m + j], u2[j], bf); bl = fma(T.Bg[r * m + j], u2[j] - us[j], bl); }
                            for (int j = 0; j < n; ++j) fl = fma(Acg[(size_t)i * n * n + r * n + j], x2[j] - xs[j], fl);
                            const double fv = (f0 + f1) + bf, fa = fkg[(size_t)i * n + r] + fl + bl;
                            const double fsr = b.fs[r];
                            de = fsr * (fv - fa); da = fsr * fa;
                        }
                        const double e2 = wg::wave_sum(de * de), a2 = wg::wave_sum(da * da);
                        if (tid == 0) { accb[2 * i] = par.dt * sqrt(e2); accb[2 * i + 1] = par.dt * sqrt(a2); }
                    }
                    __syncthreads();
                    continue;
                }
                ssm::linearize(S, SSM_CONT, 0.0, xs, us, sw, Al, n, Bl, dl);
                ssm::linearize(S, SSM_CONT, 0.0, x2, u2, sw, A2, n, B2, d2);
                if (tid == 0) {
                    double e2 = 0.0, a2 = 0.0;
                    for (int r = 0; r < n; ++r) {
                        double fk = 0.0, fl = 0.0, f = 0.0;
                        for (int j = 0; j < n; ++j) {
                            fk = fma(Al[r * n + j], xs[j], fk);
                            fl = fma(Al[r * n + j], x2[j] - xs[j], fl);
                            f = fma(A2[r * n + j], x2[j], f);
                        }
                        double bk = 0.0, bl = 0.0, bf = 0.0;
                        for (int j = 0; j < m; ++j) {
                            bk = fma(Bl[r * m + j], us[j], bk);
                            bl = fma(Bl[r * m + j], u2[j] - us[j], bl);
                            bf = fma(B2[r * m + j], u2[j], bf);
                        }
                        const double fkv = fk + bk + dl[r], fv = f + bf + d2[r];
                        const double fa = fkv + fl + bl;
                        const double fsr = b.fs[r];
                        const double de = fsr * (fv - fa), da = fsr * fa;
                        e2 = fma(de, de, e2);
                        a2 = fma(da, da, a2);
                    }
                    accb[2 * i] = par.dt * sqrt(e2); accb[2 * i + 1] = par.dt * sqrt(a2);
                }
                __syncthreads();
            }
            double err = 0.0, app = 0.0;
            for (int i = 0; i < N; ++i) { err += accb[2 * i]; app += accb[2 * i + 1]; }
            rho_k = err / (J + app);
            if (rho_k > par.rho && itr != 1) {
                delta = par.beta_fail * delta;
            } else {
                if (d_prev == delta && o_prev == omega && J_prev <= J) delta = par.beta_fail * delta;
                d_prev = delta; J_prev = J; o_prev = omega;
                // state-constraint violation (gusto.py:185-201): X applied to the states, all k = 0..N
                double viol = 0.0;
                if (b.nXv > 0) {
                    for (int k = tid; k <= N; k += nt) {
                        double v2 = 0.0;
                        for (int r = 0; r < b.nXv; ++r) {
                            double v = -b.Xb[r];
                            for (int j = 0; j < n; ++j) v = fma(b.XA[(size_t)r * n + j], w.x[(size_t)k * na + j], v);
                            v = fmax(v, 0.0);
                            v2 = fma(v, v, v2);
                        }
                        viol = fmax(viol, sqrt(v2));
                    }
                    viol = wg::reduce(viol, 1, red);
                }
                const bool X_ok = !(viol > par.epsilon);
                if (!X_ok) omega = par.gamma_fail * omega;
                // convergence (gusto.py:150-161)
                double ds = 0.0;
                for (int k = tid; k <= N; k += nt) {
                    double v2 = 0.0;
                    for (int j = 0; j < n; ++j) {
                        const double e = c.xs[j] * (w.x[(size_t)k * na + j] - xk[(size_t)k * n + j]);
                        v2 = fma(e, e, v2);
                    }
                    ds += sqrt(v2);
                }
                ds = wg::reduce(ds, 0, red);
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
            if (par.poison_warm & 2) {                     // debug (SRH_GUSTO_TRACE_QIT=1): shader clocks of the phases + interior-point iterations
                lap_tests += clock64() - lap0;
                tr[0] = (double)lap_lin; tr[1] = (double)lap_qp; tr[2] = (double)lap_tests; tr[3] = (double)(qit + 1000 * (qpass + 1));
                if (par.max_trace >= 4) for (int i = 0; i < 10; ++i) tr[4 + i] = qdbg[i];     // rows 1..3 of the trace (a one-iteration solve leaves them free)
            }
        }
        tr_hot = on_boundary && !new_solution;
        ++itr;
        if (new_solution) {
            __syncthreads();
            for (int e = tid; e < (N + 1) * n; e += nt) { const int k = e / n, j = e - k * n; xk[e] = w.x[(size_t)k * na + j]; }
            for (int e = tid; e < N * m; e += nt) uk[e] = w.u[e];
            __syncthreads();
            if (par.max_iters >= 1) linearise_all();          // gusto.py:458-473
        }
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
        for (int j = 0; j < n; ++j) v = fma(b.Hm[a * n + j], xk[(size_t)k * n + j], v);
        b.zopt[p * (size_t)(N + 1) * nz + e] = v;
    }
    if (tid == 0) { rec[0] = have_warm ? 1.0 : 0.0; b.iters[p] = itr; b.status[p] = status; if (b.Jopt) b.Jopt[p] = J_prev; }
}

}  // namespace

struct sgusto_ssm_plan {
    sssm *model = nullptr;
    QPConstHost C;
    GustoPar par{};
    int64_t batch = 0;
    int n = 0, mode = 0, nXv = 0, max_trace = 0;
    srh::DevBuf fs, Hm, XA, Xb, work, Jopt;
    size_t work_stride = 0, lds = 0;
    int red_off = 0, tab_off = 0;       // (doubles) reduction scratch / model tables behind the aliased layouts (tab_off = 0: tables stay in L2)
    int lean_gx = 0;                    // > 0: the lean one-wave interior point runs first (template argument GXL of the kernel)
    int dense_u = 0;                    // 1: the dense one-wave QP in the space of the inputs runs first (N n_u <= 16: qdu::solve)
    int task_stride = 0;                // > 0 (doubles): one wave per linearisation task, each with its own LDS scratch of this size
    char *pin = nullptr;                // one pinned, device-visible block: [inputs | outputs]
    size_t pin_bytes = 0;
    bool solved = false;
    ~sgusto_ssm_plan() { if (pin) (void)hipHostFree(pin); }
};

namespace {
struct SsmPin { size_t x0, u_init, x_init, z, ud, xopt, uopt, zopt, iters, status, trace, total; };
SsmPin ssm_pin_layout(const sgusto_ssm_plan *pl) {
    const QPDims &d = pl->C.dims;
    const size_t N = d.N, n = pl->n, m = d.m, nz = d.nz, B = pl->batch, D = sizeof(double);
    SsmPin L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 63) & ~(size_t)63; return at; };
    L.x0 = take(D * B * n); L.u_init = take(D * B * N * m); L.x_init = take(D * B * (N + 1) * n); L.z = take(D * B * (N + 1) * nz);
    L.ud = take(D * B * N * m); L.xopt = take(D * B * (N + 1) * n); L.uopt = take(D * B * N * m); L.zopt = take(D * B * (N + 1) * nz);
    L.iters = take(sizeof(int32_t) * B); L.status = take(sizeof(int32_t) * B); L.trace = take(D * B * (size_t)std::max(1, pl->max_trace) * 4);
    L.total = o;
    return L;
}

int ssm_gusto_launch(sgusto_ssm_plan *pl, const SsmGustoBatch &b, hipStream_t st) {
    const QPDims &d = pl->C.dims;
    bool launched = false;
#define X(SP, M, GX) if (!launched && (d.split != 0) == SP && (M == 0 || d.m == M) && pl->lean_gx == GX) { \
        SRH_CHECK_HIP(hipFuncSetAttribute((const void *)gusto_ssm_kernel<SP, M, GX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds)); \
        gusto_ssm_kernel<SP, M, GX><<<(unsigned)pl->batch, SSM_THREADS, pl->lds, st>>>(d, pl->C.view(), pl->model->view(), pl->par, b, pl->red_off, pl->tab_off, pl->dense_u, pl->task_stride); launched = true; }
    X(false, 4, 1) X(false, 4, 2) X(false, 8, 1) X(false, 4, 0) X(false, 8, 0) X(false, 0, 0) X(true, 0, 0)
#undef X
    SRH_REQUIRE(launched, "sgusto_ssm: no kernel variant");
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}
}  // namespace

extern "C" {

/* GuSTO(SSMGuSTO(model), ...) as a resident plan (gusto.py:54-147 + models/ssm.py): `prob` is the QP in the AUGMENTED state
 * [x ; zeta] (n_x = n + n_obs, H = [0 I], x_scale zero on zeta: sofacontrol_amd/scp/locp.py) when the model's output map is nonlinear
 * (n_obs = the model's n_o), or the plain QP (n_x = n) with the constant H otherwise; `mode` = discretisation (sssm_linearize);
 * Hm (n_z x n) = model.H for zopt; XA / Xb (nX x n) = the state polyhedron as gusto.py:185-201 applies it to the states, or NULL. */
int sgusto_ssm_plan_create(sgusto_ssm_plan_t **out, sssm_t *model, const slocp_problem *prob, const sgusto_params *par, double dt, int mode,
                           int64_t batch, const double *f_char, const double *Hm, int nX, const double *XA, const double *Xb, int max_trace) {
    SRH_REQUIRE(out && model && prob && par && Hm, "sgusto_ssm_plan_create: null argument");
    SRH_REQUIRE(batch > 0 && max_trace >= 0, "sgusto_ssm_plan_create: bad batch / max_trace");
    SRH_REQUIRE(prob->n_u == model->m, "sgusto_ssm_plan_create: n_u %d != the model's %d", prob->n_u, model->m);
    SRH_REQUIRE(prob->n_x == model->n || prob->n_x == model->n + model->no,
                "sgusto_ssm_plan_create: the QP's state has %d entries; the model has n = %d states (+ %d outputs when they are carried)",
                prob->n_x, model->n, model->no);
    SRH_REQUIRE(prob->Qzf == nullptr && prob->ndU == 0, "sgusto_ssm_plan_create: terminal cost / rate rows are not on this path");
    SRH_REQUIRE(mode >= SSM_FE && mode <= SSM_DISCRETE_MAP, "sgusto_ssm_plan_create: mode %d", mode);
    SRH_REQUIRE(mode != SSM_DISCRETE_MAP || model->has_discrete, "sgusto_ssm_plan_create: the model has no discrete map");
    SRH_REQUIRE(nX == 0 || (XA && Xb), "sgusto_ssm_plan_create: nX > 0 needs XA and Xb");
    std::unique_ptr<sgusto_ssm_plan> pl(new sgusto_ssm_plan());
    pl->model = model;
    pl->batch = batch;
    pl->n = model->n;
    pl->mode = mode;
    pl->nXv = nX;
    pl->max_trace = max_trace;
    int rc = build_consts(prob, pl->C);
    if (rc) return rc;
    QPDims &d = pl->C.dims;
    // the lean one-wave interior point when the problem has its shape (lean.hip: lean_matches for NST < 0) and an instantiation exists
    {
        const int RXa = d.nX + d.nXf, gx = RXa == 0 ? 1 : (RXa <= 2 ? 2 : (RXa <= 4 ? 4 : 8));
        const bool shape = d.lean == 2 && d.KT == 1 && d.N * d.m <= 64 && d.N * gx <= 64 && d.lean_j0 == 0 && d.po == 2 && !d.split;
        const bool inst = (d.m == 4 && (gx == 1 || gx == 2)) || (d.m == 8 && gx == 1);
        pl->lean_gx = (shape && inst && !getenv("SRH_GUSTO_SSM_NO_LEAN")) ? gx : 0;
    }
    pl->par = GustoPar{par->delta0, par->omega0, par->rho, par->beta_fail, par->gamma_fail, par->epsilon,
                       par->omega_max, par->convg_thresh, dt, par->max_gusto_iters, max_trace, 0, getenv("SRH_GUSTO_TRACE_QIT") != nullptr ? 2 : 0, 0};
    const size_t n = model->n, nz = d.nz;
    size_t doubles = (ssm_gusto_work(d, (int)n).end + 3) & ~(size_t)3;
    d.qc_off = (long long)doubles;
    doubles += qc_work_doubles(d);
    pl->work_stride = (doubles + 3) & ~(size_t)3;
    const SsmDev S = model->view();
    const size_t a = qp_kernel_lds_bytes(d), s2 = ssm_gusto_scratch_doubles(S) * sizeof(double);
    pl->dense_u = (qdu::applies(d) && prob->Qzf == nullptr && !getenv("SRH_GUSTO_SSM_NO_DENSE")) ? 1 : 0;
    if (pl->dense_u) pl->lean_gx = 0;               // (one one-wave QP per kernel: the instantiation without the lean interior point is the smaller one)
    const size_t a2 = pl->lean_gx ? lean_kernel_lds_bytes(d) : 0;
    const size_t a3 = pl->dense_u ? qdu::lds_doubles(d) * sizeof(double) : 0;
    const size_t body = (std::max(std::max(std::max(a, a2), a3), s2) + 15) & ~(size_t)15;
    pl->red_off = (int)(body / sizeof(double));
    size_t total = body + 16 * sizeof(double);
    const size_t tabs = (ssm::lds_tab_doubles(S.n, S.no, S.nr, S.ns, 0) + 8) * sizeof(double);
    if (mode != SSM_DISCRETE_MAP && model->n <= 64 && total + tabs <= (size_t)160 * 1024 && !getenv("SRH_GUSTO_SSM_NO_TABLES")) {
        // one wave per linearisation task when 2 N + 1 scratch areas fit in front of the tables (they alias the QP's layouts like the rest)
        const size_t nn = model->n, mm = model->m, noo = model->no;
        size_t stride = ssm::work_doubles(S.n, S.m, S.no, S.nr, S.ns) + 2 * (nn + mm) + std::max(nn * nn + nn * mm + nn, noo * nn + 2 * noo) + 4;
        stride = (stride + 3) & ~(size_t)3;
        const size_t ntask = d.n > model->n ? 2 * (size_t)d.N + 1 : (size_t)d.N;
        size_t need = ntask * stride * sizeof(double);
        if (!getenv("SRH_GUSTO_SSM_SERIAL_LIN") && std::max(total, need + 16 * sizeof(double)) + tabs <= (size_t)160 * 1024) {
            pl->task_stride = (int)stride;
            if (need + 16 * sizeof(double) > total) {
                const size_t body2 = (need + 15) & ~(size_t)15;
                pl->red_off = (int)(body2 / sizeof(double));
                total = body2 + 16 * sizeof(double);
            }
        }
        pl->tab_off = (int)(total / sizeof(double));
        total += tabs;
    }
    pl->lds = srh::lds_request(total);
    SRH_REQUIRE(pl->lds <= 160 * 1024, "sgusto_ssm_plan_create: %zu bytes of LDS needed, 160 KiB available", pl->lds);
    std::vector<double> fs(n, 1.0);
    if (f_char) for (size_t i = 0; i < n; ++i) fs[i] = 1.0 / fabs(f_char[i]);
    if ((rc = pl->fs.upload(fs.data(), sizeof(double) * n)) || (rc = pl->Hm.upload(Hm, sizeof(double) * nz * n)) ||
        (rc = pl->work.alloc(sizeof(double) * pl->work_stride * batch)) || (rc = pl->Jopt.alloc(sizeof(double) * batch)))
        return rc;
    if (nX > 0 && ((rc = pl->XA.upload(XA, sizeof(double) * nX * n)) || (rc = pl->Xb.upload(Xb, sizeof(double) * nX)))) return rc;
    SRH_CHECK_HIP(hipMemset(pl->work.p, 0, sizeof(double) * pl->work_stride * batch));
    const SsmPin PL = ssm_pin_layout(pl.get());
    SRH_CHECK_HIP(hipHostMalloc((void **)&pl->pin, PL.total, hipHostMallocDefault));
    pl->pin_bytes = PL.total;
    *out = pl.release();
    return SRH_OK;
}

int sgusto_ssm_plan_destroy(sgusto_ssm_plan_t *pl) { delete pl; return SRH_OK; }

/* sgusto_plan_set_warm_across for the SSM plan: the first QP of a solve starts from the minimiser / multipliers the rollout's previous solve left
 * (the reference's warm_start=True, locp.py:181); only where the lean one-wave interior point runs. */
int sgusto_ssm_plan_set_warm_across(sgusto_ssm_plan_t *pl, int on) {
    SRH_REQUIRE(pl, "sgusto_ssm_plan_set_warm_across: null plan");
    pl->par.warm_across = (on && (pl->lean_gx > 0 || pl->dense_u)) ? 1 : 0;
    return SRH_OK;
}

int sgusto_ssm_plan_set_max_iters(sgusto_ssm_plan_t *pl, int max_gusto_iters) {
    SRH_REQUIRE(pl, "sgusto_ssm_plan_set_max_iters: null plan");
    pl->par.max_iters = max_gusto_iters;
    return SRH_OK;
}

/* Device-pointer form (asynchronous on `stream`): x0 (batch x n), u_init (batch x N x n_u), x_init (batch x (N+1) x n),
 * z (batch x (N+1) x n_z) or NULL, u_des or NULL -> xopt, uopt, zopt, iters, status, trace (batch x max_trace x 4) or NULL. */
int sgusto_ssm_plan_solve_dev(sgusto_ssm_plan_t *pl, const double *x0, const double *u_init, const double *x_init, const double *z,
                              const double *u_des, double *xopt, double *uopt, double *zopt, int32_t *iters, int32_t *status, double *trace,
                              void *stream) {
    SRH_REQUIRE(pl && x0 && u_init && x_init && xopt && uopt && zopt && iters && status, "sgusto_ssm_plan_solve_dev: null argument");
    SsmGustoBatch b{x0, u_init, x_init, z, u_des, pl->fs.as<double>(), pl->Hm.as<double>(), pl->nXv ? pl->XA.as<double>() : nullptr,
                    pl->nXv ? pl->Xb.as<double>() : nullptr, pl->nXv, xopt, uopt, zopt, iters, status, trace, pl->work.as<double>(),
                    pl->work_stride, pl->n, pl->mode, pl->Jopt.as<double>(), 0};
    GustoPar keep = pl->par;
    if (!trace) pl->par.max_trace = 0;
    const int rc = ssm_gusto_launch(pl, b, (hipStream_t)stream);
    pl->par = keep;
    pl->solved = rc == SRH_OK;
    return rc;
}

/* Host-pointer form: the arguments go through the plan's pinned block (memcpy in, ONE launch, ONE stream synchronisation, memcpy
 * out) -- the kernel copies what it reads more than once into its work block. */
int sgusto_ssm_plan_solve(sgusto_ssm_plan_t *pl, const double *x0, const double *u_init, const double *x_init, const double *z,
                          const double *u_des, double *xopt, double *uopt, double *zopt, int32_t *iters, int32_t *status, double *trace) {
    SRH_REQUIRE(pl && x0 && u_init && x_init && xopt && uopt && zopt, "sgusto_ssm_plan_solve: null argument");
    const QPDims &d = pl->C.dims;
    const size_t N = d.N, n = pl->n, m = d.m, nz = d.nz, B = pl->batch, D = sizeof(double);
    const SsmPin PL = ssm_pin_layout(pl);
    char *dp = nullptr;
    SRH_CHECK_HIP(hipHostGetDevicePointer((void **)&dp, pl->pin, 0));
    memcpy(pl->pin + PL.x0, x0, D * B * n);
    memcpy(pl->pin + PL.u_init, u_init, D * B * N * m);
    memcpy(pl->pin + PL.x_init, x_init, D * B * (N + 1) * n);
    if (z) memcpy(pl->pin + PL.z, z, D * B * (N + 1) * nz);
    if (u_des) memcpy(pl->pin + PL.ud, u_des, D * B * N * m);
    auto dv = [&](size_t off) { return reinterpret_cast<double *>(dp + off); };
    const bool want_trace = trace != nullptr && pl->max_trace > 0;
    SsmGustoBatch b{dv(PL.x0), dv(PL.u_init), dv(PL.x_init), z ? dv(PL.z) : nullptr, u_des ? dv(PL.ud) : nullptr, pl->fs.as<double>(),
                    pl->Hm.as<double>(), pl->nXv ? pl->XA.as<double>() : nullptr, pl->nXv ? pl->Xb.as<double>() : nullptr, pl->nXv,
                    dv(PL.xopt), dv(PL.uopt), dv(PL.zopt), reinterpret_cast<int32_t *>(dp + PL.iters), reinterpret_cast<int32_t *>(dp + PL.status),
                    want_trace ? dv(PL.trace) : nullptr, pl->work.as<double>(), pl->work_stride, pl->n, pl->mode, pl->Jopt.as<double>(), 1};
    GustoPar keep = pl->par;
    if (!want_trace) pl->par.max_trace = 0;
    if (want_trace) for (size_t i = 0; i < B * (size_t)pl->max_trace * 4; ++i) reinterpret_cast<double *>(pl->pin + PL.trace)[i] = NAN;
    const int rc = ssm_gusto_launch(pl, b, nullptr);
    pl->par = keep;
    if (rc) return rc;
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    pl->solved = true;
    memcpy(xopt, pl->pin + PL.xopt, D * B * (N + 1) * n);
    memcpy(uopt, pl->pin + PL.uopt, D * B * N * m);
    memcpy(zopt, pl->pin + PL.zopt, D * B * (N + 1) * nz);
    if (iters) memcpy(iters, pl->pin + PL.iters, sizeof(int32_t) * B);
    if (status) memcpy(status, pl->pin + PL.status, sizeof(int32_t) * B);
    if (want_trace) memcpy(trace, pl->pin + PL.trace, D * B * (size_t)pl->max_trace * 4);
    return SRH_OK;
}

}  // extern "C"
