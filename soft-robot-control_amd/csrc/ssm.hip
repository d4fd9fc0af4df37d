// SSM polynomial reduced model on the device: batched maps, Jacobian linearisation (+ discretisation),
// observer linearisation, observed -> reduced projection, rollout.
// Reference: sofacontrol/SSM/ssm.py (SSM, SSMDynamics).  The reference differentiates the lambdified
// polynomial maps with jax; here the monomial derivatives are analytic.
#include "ssm_host.h"

#include <algorithm>
#include <functional>
#include <map>

namespace {

constexpr int SSM_NT = 128;

__global__ __launch_bounds__(SSM_NT) void ssm_lin_kernel(SsmDev S, const double *__restrict__ X,
                                                         const double *__restrict__ U, int mode, double dt,
                                                         double *__restrict__ A, double *__restrict__ Bm,
                                                         double *__restrict__ d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = S.n, m = S.m, tid = threadIdx.x, nt = blockDim.x;
    const size_t b = blockIdx.x;
    ssm::Work w;
    ssm::carve(w, (lptr)smem, S);
    lptr xs = (lptr)smem + ssm::work_doubles(n, m, S.no, S.nr, S.ns);
    lptr us = xs + n, Al = us + m, Bl = Al + (size_t)n * n, dl = Bl + (size_t)n * m;
    for (int e = tid; e < n; e += nt) xs[e] = X[b * n + e];
    for (int e = tid; e < m; e += nt) us[e] = U[b * m + e];
    __syncthreads();
    ssm::linearize(S, mode, dt, xs, us, w, Al, n, Bl, dl);
    for (int e = tid; e < n * n; e += nt) A[b * n * n + e] = Al[e];
    for (int e = tid; e < n * m; e += nt) Bm[b * n * m + e] = Bl[e];
    for (int e = tid; e < n; e += nt) d[b * n + e] = dl[e];
}

// f = R phi(x) + B u (continuous) or Rd phi(x) + Bd u (discrete map)      ssm.py:167-178
__global__ __launch_bounds__(SSM_NT) void ssm_dyn_kernel(SsmDev S, const double *__restrict__ X,
                                                         const double *__restrict__ U, int discrete,
                                                         double *__restrict__ F) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = S.n, m = S.m, tid = threadIdx.x, nt = blockDim.x;
    const size_t b = blockIdx.x;
    ssm::Work w;
    ssm::carve(w, (lptr)smem, S);
    lptr xs = (lptr)smem + ssm::work_doubles(n, m, S.no, S.nr, S.ns);
    lptr us = xs + n;
    for (int e = tid; e < n; e += nt) xs[e] = X[b * n + e];
    for (int e = tid; e < m; e += nt) us[e] = U[b * m + e];
    __syncthreads();
    ssm::basis(S.er, S.pr, S.vr, S.dmr, S.lvr, S.order_r, S.nr, n, xs, w.phi, (lptr) nullptr);
    cgptr Rc = discrete ? S.Rd : S.R, Bg = discrete ? S.Bd : S.Bc;
    for (int i = tid; i < n; i += nt) {
        double s = 0.0;
        for (int k = 0; k < S.nr; ++k) s = fma(Rc[(size_t)i * S.nr + k], w.phi[k], s);
        double t = 0.0;
        for (int k = 0; k < m; ++k) t = fma(Bg[i * m + k], us[k], t);
        F[b * n + i] = s + t;
    }
}

__global__ __launch_bounds__(SSM_NT) void ssm_obs_kernel(SsmDev S, const double *__restrict__ X,
                                                         double *__restrict__ Z, double *__restrict__ Hout,
                                                         double *__restrict__ cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = S.n, no = S.no, tid = threadIdx.x, nt = blockDim.x;
    const size_t b = blockIdx.x;
    ssm::Work w;
    ssm::carve(w, (lptr)smem, S);
    lptr xs = (lptr)smem + ssm::work_doubles(n, S.m, no, S.nr, S.ns);
    lptr zs = xs + n, cs = zs + no, Hl = cs + no;
    for (int e = tid; e < n; e += nt) xs[e] = X[b * n + e];
    __syncthreads();
    ssm::observe(S, xs, w, zs, Hout ? Hl : (lptr) nullptr, cs);
    if (Z) for (int e = tid; e < no; e += nt) Z[b * no + e] = zs[e];
    if (Hout) {
        for (int e = tid; e < no * n; e += nt) Hout[b * no * n + e] = Hl[e];
        if (cout) for (int e = tid; e < no; e += nt) cout[b * no + e] = cs[e];
    }
}

// x = W_map(z - z_ref) = V phi_s(z - z_ref)                               ssm.py:176-178, 338-344
__global__ __launch_bounds__(SSM_NT) void ssm_reduce_kernel(SsmDev S, const double *__restrict__ Z,
                                                            double *__restrict__ X) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = S.n, no = S.no, tid = threadIdx.x, nt = blockDim.x;
    const size_t b = blockIdx.x;
    ssm::Work w;
    ssm::carve(w, (lptr)smem, S);
    lptr zs = (lptr)smem + ssm::work_doubles(n, S.m, no, S.nr, S.ns);
    for (int e = tid; e < no; e += nt) zs[e] = Z[b * no + e] - S.z_ref[e];
    __syncthreads();
    ssm::basis(S.es, S.ps, S.vs, S.dms, S.lvs, S.order_s, S.ns, no, zs, w.phi, (lptr) nullptr);
    for (int i = tid; i < n; i += nt) {
        double s = 0.0;
        for (int k = 0; k < S.ns; ++k) s = fma(S.Vc[(size_t)i * S.ns + k], w.phi[k], s);
        X[b * n + i] = s;
    }
}

// rollout (ssm.py:134-156): x_{i+1} = A_d x_i + B_d u_i + d_d with (A_d, B_d, d_d) re-linearised at (x_i, u_i);
// z = C_map(x) + z_ref for all N+1 states.
__global__ __launch_bounds__(SSM_NT) void ssm_rollout_kernel(SsmDev S, const double *__restrict__ x0,
                                                             const double *__restrict__ U, int N, int mode, double dt,
                                                             double *__restrict__ X, double *__restrict__ Z) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = S.n, m = S.m, no = S.no, tid = threadIdx.x, nt = blockDim.x;
    const size_t b = blockIdx.x;
    ssm::Work w;
    ssm::carve(w, (lptr)smem, S);
    lptr xs = (lptr)smem + ssm::work_doubles(n, m, no, S.nr, S.ns);
    lptr us = xs + n, Al = us + m, Bl = Al + (size_t)n * n, dl = Bl + (size_t)n * m, xn = dl + n, zs = xn + n;
    double *Xb = X + b * (size_t)(N + 1) * n;
    for (int e = tid; e < n; e += nt) { xs[e] = x0[b * n + e]; Xb[e] = xs[e]; }
    __syncthreads();
    for (int k = 0; k <= N; ++k) {
        if (Z != nullptr) {
            ssm::observe(S, xs, w, zs, (lptr) nullptr, (lptr) nullptr);
            for (int e = tid; e < no; e += nt) Z[(b * (N + 1) + k) * no + e] = zs[e] + S.z_ref[e];
        }
        if (k == N) break;
        for (int e = tid; e < m; e += nt) us[e] = U[(b * N + k) * m + e];
        __syncthreads();
        ssm::linearize(S, mode, dt, xs, us, w, Al, n, Bl, dl);
        for (int i = tid; i < n; i += nt) {
            double ax = 0.0, bu = 0.0;
            for (int j = 0; j < n; ++j) ax = fma(Al[i * n + j], xs[j], ax);
            for (int j = 0; j < m; ++j) bu = fma(Bl[i * m + j], us[j], bu);
            xn[i] = ax + bu + dl[i];
        }
        __syncthreads();
        for (int e = tid; e < n; e += nt) { xs[e] = xn[e]; Xb[(size_t)(k + 1) * n + e] = xn[e]; }
        __syncthreads();
    }
}

size_t lds_bytes(const sssm *h) {
    const size_t n = h->n, m = h->m, no = h->no;
    return sizeof(double) * (ssm::work_doubles(h->n, h->m, h->no, h->nr, h->ns) + 3 * n + m + n * n + n * m + 2 * no +
                             no * n + 8);
}

int set_lds(const void *fn, size_t bytes) {
    SRH_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return SRH_OK;
}

}  // namespace

// graded order, within a degree lexicographic with x1 first (sympy: sorted(itermonomials(zeta, order),
// key=monomial_key('grevlex', reversed(zeta)))[1:], ssm.py:158-164)
std::vector<int> ssm_exponents(int dim, int order) {
    std::vector<int> out;
    std::vector<int> cur(dim, 0);
    std::function<void(int, int)> rec = [&](int pos, int left) {
        if (pos == dim - 1) {
            cur[pos] = left;
            out.insert(out.end(), cur.begin(), cur.end());
            return;
        }
        for (int e = left; e >= 0; --e) {
            cur[pos] = e;
            rec(pos + 1, left - e);
        }
    };
    for (int deg = 1; deg <= order; ++deg) rec(0, deg);
    return out;
}

// evaluation tables of a graded monomial basis (see ssm::basis): parent, variable, derivative index, level offsets
static int ssm_upload_tables(const std::vector<int> &E, int dim, int order, srh::DevBuf (&t)[4]) {
    const int nm = (int)(E.size() / dim);
    std::map<std::vector<int>, int> index;
    for (int j = 0; j < nm; ++j) index[std::vector<int>(E.begin() + (size_t)j * dim, E.begin() + (size_t)(j + 1) * dim)] = j;
    std::vector<int> par(nm), var(nm), dm((size_t)nm * dim), lv(order + 1, nm);
    int prev_deg = 0;
    for (int j = 0; j < nm; ++j) {
        std::vector<int> e(E.begin() + (size_t)j * dim, E.begin() + (size_t)(j + 1) * dim);
        int deg = 0, last = 0;
        for (int i = 0; i < dim; ++i) { deg += e[i]; if (e[i] > 0) last = i; }
        if (deg != prev_deg) { for (int d = prev_deg; d < deg; ++d) lv[d] = j; prev_deg = deg; }
        var[j] = last;
        if (deg == 1) {
            par[j] = -1;
        } else {
            std::vector<int> p = e; p[last] -= 1;
            par[j] = index.at(p);
        }
        for (int i = 0; i < dim; ++i) {
            if (e[i] == 0) { dm[(size_t)j * dim + i] = -1; continue; }
            if (deg == 1) { dm[(size_t)j * dim + i] = -2; continue; }
            std::vector<int> p = e; p[i] -= 1;
            dm[(size_t)j * dim + i] = index.at(p);
        }
    }
    lv[order] = nm;
    // lv[d] = first index of degree d + 1, d = 0 .. order - 1
    int rc;
    if ((rc = t[0].upload(par.data(), sizeof(int) * nm)) || (rc = t[1].upload(var.data(), sizeof(int) * nm)) ||
        (rc = t[2].upload(dm.data(), sizeof(int) * dm.size())) || (rc = t[3].upload(lv.data(), sizeof(int) * lv.size())))
        return rc;
    return SRH_OK;
}

SsmDev sssm::view() const {
    SsmDev S{};
    S.n = n; S.m = m; S.no = no; S.nr = nr; S.ns = ns;
    S.er = er.as<int>(); S.es = es.as<int>();
    S.pr = tr[0].as<int>(); S.vr = tr[1].as<int>(); S.dmr = tr[2].as<int>(); S.lvr = tr[3].as<int>();
    S.ps = ts[0].as<int>(); S.vs = ts[1].as<int>(); S.dms = ts[2].as<int>(); S.lvs = ts[3].as<int>();
    S.order_r = order_r; S.order_s = order_s;
    auto g = [](const srh::DevBuf &b) { return (cgptr)b.as<double>(); };
    S.R = g(R); S.Bc = g(Bc); S.Rd = g(Rd); S.Bd = g(Bd); S.Wc = g(Wc); S.Vc = g(Vc); S.z_ref = g(z_ref); S.H = g(H);
    return S;
}

extern "C" {

int sssm_num_monomials(int dim, int order) {
    if (dim <= 0 || order <= 0) return 0;
    return (int)(ssm_exponents(dim, order).size() / dim);
}

int sssm_exponents(int dim, int order, int32_t *exps) {
    SRH_REQUIRE(dim > 0 && order > 0 && exps, "sssm_exponents: bad argument");
    auto e = ssm_exponents(dim, order);
    std::copy(e.begin(), e.end(), exps);
    return SRH_OK;
}

int sssm_create(sssm_t **out, int n_x, int n_u, int n_o, int rom_order, int ssm_order, const double *r_coeff,
                const double *B, const double *rd_coeff, const double *Bd, const double *w_coeff,
                const double *v_coeff, const double *z_ref) {
    SRH_REQUIRE(out && r_coeff && B && w_coeff && v_coeff && z_ref, "sssm_create: null argument");
    SRH_REQUIRE(n_x > 0 && n_x <= 32 && n_u > 0 && n_u <= n_x + 16 && n_o > 0 && rom_order > 0 && ssm_order > 0,
                "sssm_create: need 0 < n_x <= 32, n_u > 0, n_o > 0, orders > 0");
    SRH_REQUIRE(n_u <= (n_x | 1), "sssm_create: n_u > n_x is not supported by the discretisation scratch");
    auto *h = new sssm();
    h->n = n_x; h->m = n_u; h->no = n_o;
    auto er = ssm_exponents(n_x, rom_order), es = ssm_exponents(n_o, ssm_order);
    h->nr = (int)(er.size() / n_x); h->ns = (int)(es.size() / n_o);
    h->order_r = rom_order; h->order_s = ssm_order;
    int rc;
    if ((rc = ssm_upload_tables(er, n_x, rom_order, h->tr)) || (rc = ssm_upload_tables(es, n_o, ssm_order, h->ts))) {
        delete h;
        return rc;
    }
    std::vector<double> Hz((size_t)n_o * n_x, 0.0);
    if ((rc = h->er.upload(er.data(), sizeof(int) * er.size())) || (rc = h->es.upload(es.data(), sizeof(int) * es.size())) ||
        (rc = h->R.upload(r_coeff, sizeof(double) * n_x * h->nr)) || (rc = h->Bc.upload(B, sizeof(double) * n_x * n_u)) ||
        (rc = h->Wc.upload(w_coeff, sizeof(double) * n_o * h->ns)) ||
        (rc = h->Vc.upload(v_coeff, sizeof(double) * n_x * h->ns)) || (rc = h->z_ref.upload(z_ref, sizeof(double) * n_o)) ||
        (rc = h->H.upload(Hz.data(), sizeof(double) * Hz.size()))) {
        delete h;
        return rc;
    }
    if (rd_coeff && Bd) {
        if ((rc = h->Rd.upload(rd_coeff, sizeof(double) * n_x * h->nr)) || (rc = h->Bd.upload(Bd, sizeof(double) * n_x * n_u))) {
            delete h;
            return rc;
        }
        h->has_discrete = true;
    }
    h->lds = srh::lds_request(lds_bytes(h));
    if (h->lds > 160 * 1024) {
        delete h;
        srh::set_error("sssm_create: the polynomial basis does not fit the 160 KB LDS");
        return SRH_EINVAL;
    }
    if ((rc = set_lds((const void *)ssm_lin_kernel, h->lds)) || (rc = set_lds((const void *)ssm_dyn_kernel, h->lds)) ||
        (rc = set_lds((const void *)ssm_obs_kernel, h->lds)) || (rc = set_lds((const void *)ssm_reduce_kernel, h->lds)) ||
        (rc = set_lds((const void *)ssm_rollout_kernel, h->lds))) {
        delete h;
        return rc;
    }
    *out = h;
    return SRH_OK;
}

int sssm_destroy(sssm_t *h) {
    delete h;
    return SRH_OK;
}

int sssm_set_output(sssm_t *h, const double *H) {
    SRH_REQUIRE(h && H, "sssm_set_output: null argument");
    return h->H.upload(H, sizeof(double) * h->no * h->n);
}

static int check_mode(const sssm *h, int mode) {
    SRH_REQUIRE(mode >= SSM_CONT && mode <= SSM_DISCRETE_MAP, "self.discr_method must be in [fe, be, bil, zoh]");
    SRH_REQUIRE(mode != SSM_DISCRETE_MAP || h->has_discrete, "sssm: model has no discrete map (rd_coeff, Bd)");
    return SRH_OK;
}

int sssm_linearize(sssm_t *h, const double *X, const double *U, int64_t B, int mode, double dt, double *A, double *Bm,
                   double *d) {
    SRH_REQUIRE(h && X && U && A && Bm && d, "sssm_linearize: null argument");
    int rc;
    if ((rc = check_mode(h, mode))) return rc;
    if (B == 0) return SRH_OK;
    const size_t n = h->n, m = h->m;
    srh::DevBuf dX, dU, dA, dB, dd;
    if ((rc = dX.upload(X, sizeof(double) * B * n)) || (rc = dU.upload(U, sizeof(double) * B * m)) ||
        (rc = dA.alloc(sizeof(double) * B * n * n)) || (rc = dB.alloc(sizeof(double) * B * n * m)) ||
        (rc = dd.alloc(sizeof(double) * B * n)))
        return rc;
    ssm_lin_kernel<<<(unsigned)B, SSM_NT, h->lds>>>(h->view(), dX.as<double>(), dU.as<double>(), mode, dt,
                                                     dA.as<double>(), dB.as<double>(), dd.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if ((rc = dA.download(A, sizeof(double) * B * n * n)) || (rc = dB.download(Bm, sizeof(double) * B * n * m)) ||
        (rc = dd.download(d, sizeof(double) * B * n)))
        return rc;
    return SRH_OK;
}

int sssm_dynamics(sssm_t *h, const double *X, const double *U, int64_t B, int discrete, double *F) {
    SRH_REQUIRE(h && X && U && F, "sssm_dynamics: null argument");
    SRH_REQUIRE(!discrete || h->has_discrete, "sssm_dynamics: model has no discrete map (rd_coeff, Bd)");
    if (B == 0) return SRH_OK;
    const size_t n = h->n, m = h->m;
    srh::DevBuf dX, dU, dF;
    int rc;
    if ((rc = dX.upload(X, sizeof(double) * B * n)) || (rc = dU.upload(U, sizeof(double) * B * m)) ||
        (rc = dF.alloc(sizeof(double) * B * n)))
        return rc;
    ssm_dyn_kernel<<<(unsigned)B, SSM_NT, h->lds>>>(h->view(), dX.as<double>(), dU.as<double>(), discrete, dF.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    return dF.download(F, sizeof(double) * B * n);
}

int sssm_observe(sssm_t *h, const double *X, int64_t B, double *Z, double *H, double *c) {
    SRH_REQUIRE(h && X && (Z || H), "sssm_observe: null argument");
    SRH_REQUIRE(h->n == h->no, "sssm_observe: the reduced -> observed map takes n_o arguments (needs n_x == n_o)");
    if (B == 0) return SRH_OK;
    const size_t n = h->n, no = h->no;
    srh::DevBuf dX, dZ, dH, dc;
    int rc;
    if ((rc = dX.upload(X, sizeof(double) * B * n)) || (rc = dZ.alloc(sizeof(double) * B * no))) return rc;
    if (H && ((rc = dH.alloc(sizeof(double) * B * no * n)) || (rc = dc.alloc(sizeof(double) * B * no)))) return rc;
    ssm_obs_kernel<<<(unsigned)B, SSM_NT, h->lds>>>(h->view(), dX.as<double>(), dZ.as<double>(),
                                                     H ? dH.as<double>() : nullptr, H ? dc.as<double>() : nullptr);
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if (Z && (rc = dZ.download(Z, sizeof(double) * B * no))) return rc;
    if (H && (rc = dH.download(H, sizeof(double) * B * no * n))) return rc;
    if (H && c && (rc = dc.download(c, sizeof(double) * B * no))) return rc;
    return SRH_OK;
}

int sssm_reduce(sssm_t *h, const double *Z, int64_t B, double *X) {
    SRH_REQUIRE(h && Z && X, "sssm_reduce: null argument");
    if (B == 0) return SRH_OK;
    srh::DevBuf dZ, dX;
    int rc;
    if ((rc = dZ.upload(Z, sizeof(double) * B * h->no)) || (rc = dX.alloc(sizeof(double) * B * h->n))) return rc;
    ssm_reduce_kernel<<<(unsigned)B, SSM_NT, h->lds>>>(h->view(), dZ.as<double>(), dX.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    return dX.download(X, sizeof(double) * B * h->n);
}

int sssm_rollout(sssm_t *h, const double *x0, const double *U, int N, int64_t batch, int mode, double dt, double *X,
                 double *Z) {
    SRH_REQUIRE(h && x0 && U && X, "sssm_rollout: null argument");
    SRH_REQUIRE(N >= 0 && batch >= 0, "sssm_rollout: negative size");
    SRH_REQUIRE(mode != SSM_CONT, "sssm_rollout: need a discretisation mode");
    SRH_REQUIRE(Z == nullptr || h->n == h->no, "sssm_rollout: the reduced -> observed map needs n_x == n_o");
    int rc;
    if ((rc = check_mode(h, mode))) return rc;
    if (batch == 0) return SRH_OK;
    const size_t n = h->n, m = h->m, no = h->no;
    srh::DevBuf d0, dU, dX, dZ;
    if ((rc = d0.upload(x0, sizeof(double) * batch * n)) || (rc = dU.upload(U, sizeof(double) * batch * N * m)) ||
        (rc = dX.alloc(sizeof(double) * batch * (N + 1) * n)))
        return rc;
    if (Z && (rc = dZ.alloc(sizeof(double) * batch * (N + 1) * no))) return rc;
    ssm_rollout_kernel<<<(unsigned)batch, SSM_NT, h->lds>>>(h->view(), d0.as<double>(), dU.as<double>(), N, mode, dt,
                                                            dX.as<double>(), Z ? dZ.as<double>() : nullptr);
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if ((rc = dX.download(X, sizeof(double) * batch * (N + 1) * n))) return rc;
    if (Z && (rc = dZ.download(Z, sizeof(double) * batch * (N + 1) * no))) return rc;
    return SRH_OK;
}

}  // extern "C"
