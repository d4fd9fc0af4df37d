// Euclidean projection onto a polyhedron {p : A p <= b}: min 1/2 |p - x|^2 s.t. A p <= b.
// Reference: sofacontrol/utils.py:364-407 (Polyhedron(with_reproject=True).project_to_polyhedron: the same QP handed to
// OSQP with P = I, q = -x); used on measurements that left the admissible set, SSM/controllers.py:96-97.
// One wave per point, Mehrotra predictor-corrector interior point: lane i owns constraint i (slack, multiplier, row
// of A in registers), the n-vectors are kept uniform in every lane; the n x n normal matrix I + A^T D A is summed
// with wave reductions and factored in LDS (n <= 16, rows <= 64: these are output-space boxes and a few facets).
#include "common.h"
#include "dev_la.h"

namespace {

constexpr int PN = 16;      // max dimension
constexpr int PM = 64;      // max rows

struct PolyArgs {
    const double *A, *b, *X;
    double *out;
    int *status;
    int mc, n;
    double tol;
    int max_iter;
};

__global__ __launch_bounds__(64) void poly_project_kernel(PolyArgs a) {
    __shared__ double M[PN * PN], Lc[PN * PN], rhs[PN], sol[PN];
    __shared__ int flag;
    const int lane = threadIdx.x, n = a.n, mc = a.mc;
    const bool act = lane < mc;
    const double *x = a.X + (size_t)blockIdx.x * n;
    double ar[PN], p[PN];
#pragma unroll
    for (int k = 0; k < PN; ++k) {
        ar[k] = (act && k < n) ? a.A[lane * n + k] : 0.0;
        p[k] = k < n ? x[k] : 0.0;
    }
    const double bi = act ? a.b[lane] : 1.0;
    auto adot = [&](const double (&v)[PN]) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < PN; ++k) s = fma(ar[k], v[k], s);
        return s;
    };
    // scale of the data for the stopping rule
    double xs = 0.0;
#pragma unroll
    for (int k = 0; k < PN; ++k) xs = fmax(xs, fabs(p[k]));
    const double scale = fmax(1.0, fmax(xs, wg::wave_max(act ? fabs(bi) : 0.0)));
    double viol = act ? adot(p) - bi : -1.0;
    int st = 0, it = 0;
    if (!(wg::wave_max(viol) > 0.0)) {          // already inside: the projection is x itself
        if (lane < n) a.out[(size_t)blockIdx.x * n + lane] = x[lane];
        if (lane == 0) a.status[blockIdx.x] = 0;
        return;
    }
    double s = act ? fmax(-viol, 1e-2 * scale) : 1.0, lam = act ? fmax(viol, 1e-2 * scale) : 0.0;
    // one Newton solve: (I + A^T D A) dp = -r_d - A^T (D r_p - r_c / s), then d lam, d s; r_c passed in
    double rd[PN];
    auto newton = [&](double rp, double rc, double &dlam, double &ds, double (&dp)[PN]) -> bool {
        const double d = act ? lam / s : 0.0;
        const double w = act ? d * rp - rc / s : 0.0;
        for (int r = 0; r < n; ++r) {
            for (int c = r; c < n; ++c) {
                const double v = wg::wave_sum(d * ar[r] * ar[c]);
                if (lane == 0) { M[r * n + c] = v + (r == c ? 1.0 : 0.0); M[c * n + r] = M[r * n + c]; }
            }
            const double g = wg::wave_sum(ar[r] * w);
            if (lane == 0) rhs[r] = rd[r] + g;          // solve gives -(M^-1) rhs
        }
        __syncthreads();
        if (!wg::chol_factor((clptr)M, (lptr)Lc, n, (liptr)&flag, true)) return false;
        if (lane == 0) wg::chol_solve_neg((clptr)Lc, n, (clptr)rhs, 1, (lptr)sol, 1);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PN; ++k) dp[k] = k < n ? sol[k] : 0.0;
        __syncthreads();
        dlam = act ? d * (adot(dp) + rp) - rc / s : 0.0;
        ds = act ? -(rc + s * dlam) / lam : 0.0;
        return true;
    };
    auto max_step = [&](double v, double dv) { return (act && dv < 0.0) ? -v / dv : INFINITY; };
    double last_rpn = INFINITY;
    for (; it < a.max_iter; ++it) {
        // residuals
        double rdn = 0.0;
#pragma unroll
        for (int k = 0; k < PN; ++k) {
            rd[k] = k < n ? p[k] - x[k] + wg::wave_sum(ar[k] * lam) : 0.0;
            rdn = fmax(rdn, fabs(rd[k]));
        }
        const double rp = act ? adot(p) + s - bi : 0.0;
        const double rpn = wg::wave_max(fabs(rp));
        last_rpn = rpn;
        const double mu = wg::wave_sum(act ? s * lam : 0.0) / mc;
        if (mu <= a.tol * scale && rpn <= 1e-10 * scale && rdn <= 1e-10 * scale) break;
        double dla, dsa, dpa[PN], dl, dsv, dp[PN];
        if (!newton(rp, act ? s * lam : 0.0, dla, dsa, dpa)) { st = 2; break; }
        double aa = fmin(1.0, wg::wave_min(fmin(max_step(s, dsa), max_step(lam, dla))));
        const double mua = wg::wave_sum(act ? (s + aa * dsa) * (lam + aa * dla) : 0.0) / mc;
        const double sig = (mua / mu) * (mua / mu) * (mua / mu);
        if (!newton(rp, act ? s * lam + dsa * dla - sig * mu : 0.0, dl, dsv, dp)) { st = 2; break; }
        const double al = fmin(1.0, 0.99 * wg::wave_min(fmin(max_step(s, dsv), max_step(lam, dl))));
#pragma unroll
        for (int k = 0; k < PN; ++k) p[k] += al * dp[k];
        if (act) { s += al * dsv; lam += al * dl; }
    }
    if (st == 0 && it >= a.max_iter) st = 1;
    // a polyhedron without strict interior (rows with lb == ub) has no central path to follow to a 1e-13 gap: the last
    // iterate is returned when it is feasible to 1e-6 (the reference's OSQP returns such an approximate point too)
    {
        const double vend = wg::wave_max(act ? adot(p) - bi : -INFINITY);     // the actual violation of the last iterate
        bool finite = true;                 // (fmax / fmin drop NaNs: a diverged iterate must not pass as feasible)
#pragma unroll
        for (int k = 0; k < PN; ++k) finite = finite && (p[k] == p[k]) && fabs(p[k]) < 1e300;
        if (st != 0 && finite && vend > -1e300 && vend <= 1e-6 * scale && last_rpn <= 1e-6 * scale) st = 0;
    }
    if (lane < n) a.out[(size_t)blockIdx.x * n + lane] = p[lane < PN ? lane : 0];
    if (lane == 0) a.status[blockIdx.x] = st;
}

}  // namespace

extern "C" {

int spoly_project(const double *A, const double *b, int n_rows, int n, const double *X, int64_t batch, double *out) {
    SRH_REQUIRE(A && b && X && out, "spoly_project: null argument");
    SRH_REQUIRE(n > 0 && n <= PN && n_rows > 0 && n_rows <= PM, "spoly_project: need 0 < n <= 16 and 0 < rows <= 64 (got %d, %d)",
                n, n_rows);
    if (batch == 0) return SRH_OK;
    srh::DevBuf dA, db, dX, dO, dS;
    int rc;
    if ((rc = dA.upload(A, sizeof(double) * n_rows * n)) || (rc = db.upload(b, sizeof(double) * n_rows)) ||
        (rc = dX.upload(X, sizeof(double) * batch * n)) || (rc = dO.alloc(sizeof(double) * batch * n)) ||
        (rc = dS.alloc(sizeof(int) * batch)))
        return rc;
    PolyArgs a{dA.as<double>(), db.as<double>(), dX.as<double>(), dO.as<double>(), dS.as<int>(), n_rows, n, 1e-13, 60};
    poly_project_kernel<<<(unsigned)batch, 64>>>(a);
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    std::vector<int> st((size_t)batch);
    if ((rc = dS.download(st.data(), sizeof(int) * batch))) return rc;
    for (int64_t i = 0; i < batch; ++i)
        if (st[i] != 0) {
            srh::set_error("spoly_project: point %lld: %s", (long long)i,
                           st[i] == 1 ? "no convergence (empty polyhedron?)" : "normal matrix not positive definite");
            return SRH_ENUMERIC;
        }
    return dO.download(out, sizeof(double) * batch * n);
}

}  // extern "C"
