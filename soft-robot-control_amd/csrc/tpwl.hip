// TPWL model on the device: nearest-point linearisation, table gather, batched rollout.
// Reference: sofacontrol/tpwl/tpwl.py:160-168 (calc_nearest_point), 236-270 (get_jacobians, nn),
// 193-216 (rollout), 336-339 (update_dynamics); sofacontrol/scp/models/tpwl.py:66-84.
#include "tpwl_host.h"

namespace {

__global__ void nearest_kernel(TpwlDev T, const double *__restrict__ X, int64_t B, int32_t *__restrict__ idx) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * nw + wave;
    if (k >= B) return;
    const int i = tpwl::nearest_wave(T, X + k * T.n);
    if ((threadIdx.x & 63) == 0) idx[k] = i;
}

__global__ void gather_kernel(TpwlDev T, const int32_t *__restrict__ idx, int64_t B, int discrete,
                              double *__restrict__ A, double *__restrict__ Bm, double *__restrict__ d) {
    const int64_t b = blockIdx.x;
    const int i = idx[b];
    cgptr As = (discrete ? T.Ad : T.Ac) + (size_t)i * T.n * T.n;
    cgptr Bs = (discrete ? T.Bd : T.Bc) + (size_t)i * T.n * T.m;
    cgptr ds = (discrete ? T.dd : T.dc) + (size_t)i * T.n;
    for (int e = threadIdx.x; e < T.n * T.n; e += blockDim.x) A[b * T.n * T.n + e] = As[e];
    for (int e = threadIdx.x; e < T.n * T.m; e += blockDim.x) Bm[b * T.n * T.m + e] = Bs[e];
    for (int e = threadIdx.x; e < T.n; e += blockDim.x) d[b * T.n + e] = ds[e];
}


// softmin weights over the stored points (tpwl.py:170-191): w_i = exp(-beta d_i / d_min) / sum_j(...),
// one-hot at the (first) minimum when d_min == 0.  One workgroup per query state; W (B x P).
__global__ __launch_bounds__(256) void weights_kernel(TpwlDev T, const double *__restrict__ X, int64_t B,
                                                      double beta, double *__restrict__ W) {
    __shared__ double rv[256];
    __shared__ int ri[256];
    const int64_t b = blockIdx.x;
    const double *x = X + b * T.n;
    double *w = W + b * T.P;
    double best = INFINITY;
    int besti = 0x7fffffff;
    for (int i = threadIdx.x; i < T.P; i += blockDim.x) {
        double sq = 0.0, sv = 0.0;
        for (int j = 0; j < T.r; ++j) {
            const double e = T.qT[j * T.P + i] - x[T.r + j];
            sq = fma(e, e, sq);
            const double f = T.vT[j * T.P + i] - x[j];
            sv = fma(f, f, sv);
        }
        const double d = T.w_q * sqrt(sq) + T.w_v * sqrt(sv);
        w[i] = d;
        if (d < best) { best = d; besti = i; }
    }
    rv[threadIdx.x] = best; ri[threadIdx.x] = besti;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double ob = rv[threadIdx.x + o];
            const int oi = ri[threadIdx.x + o];
            if (ob < rv[threadIdx.x] || (ob == rv[threadIdx.x] && oi < ri[threadIdx.x])) {
                rv[threadIdx.x] = ob; ri[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const double dmin = rv[0];
    const int imin = ri[0];
    __syncthreads();
    if (dmin == 0.0) {
        for (int i = threadIdx.x; i < T.P; i += blockDim.x) w[i] = (i == imin) ? 1.0 : 0.0;
        return;
    }
    double part = 0.0;
    for (int i = threadIdx.x; i < T.P; i += blockDim.x) {
        const double e = exp(-beta * w[i] / dmin);
        w[i] = e;
        part += e;
    }
    rv[threadIdx.x] = part;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) rv[threadIdx.x] += rv[threadIdx.x + o];
        __syncthreads();
    }
    const double tot = rv[0];
    for (int i = threadIdx.x; i < T.P; i += blockDim.x) w[i] = w[i] / tot;
}

// weighted tables (tpwl.py:244-248): A = sum_i w_i A_c[i], B = sum_i w_i B_c[i], d = sum_i w_i d_c[i]
__global__ __launch_bounds__(256) void blend_kernel(TpwlDev T, const double *__restrict__ W, int64_t B,
                                                    double *__restrict__ A, double *__restrict__ Bm,
                                                    double *__restrict__ d) {
    const int64_t b = blockIdx.x;
    const double *w = W + b * T.P;
    const int nn = T.n * T.n, nm = T.n * T.m, tot = nn + nm + T.n;
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < tot; e += gridDim.y * blockDim.x) {
        double s = 0.0;
        if (e < nn) {
            for (int i = 0; i < T.P; ++i) s = fma(w[i], T.Ac[(size_t)i * nn + e], s);
            A[b * nn + e] = s;
        } else if (e < nn + nm) {
            const int f = e - nn;
            for (int i = 0; i < T.P; ++i) s = fma(w[i], T.Bc[(size_t)i * nm + f], s);
            Bm[b * nm + f] = s;
        } else {
            const int f = e - nn - nm;
            for (int i = 0; i < T.P; ++i) s = fma(w[i], T.dc[(size_t)i * T.n + f], s);
            d[b * T.n + f] = s;
        }
    }
}

// one workgroup per rollout: x_{k+1} = A_d[i_k] x_k + B_d[i_k] u_k + d_d[i_k], i_k = nearest(x_k)
__global__ __launch_bounds__(256) void rollout_kernel(TpwlDev T, const double *__restrict__ x0,
                                                      const double *__restrict__ U, int N,
                                                      double *__restrict__ X, double *__restrict__ Z) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lptr xc = (lptr)smem;                                // n
    lptr xn = xc + T.n;                                  // n
    lptr uc = xn + T.n;                                  // m
    lptr part = uc + ((T.m + 3) & ~3);                   // blockDim
    liptr ip = (liptr)(part + blockDim.x);
    const int64_t b = blockIdx.x;
    const int n = T.n, m = T.m;
    double *Xb = X + b * (size_t)(N + 1) * n;
    for (int e = threadIdx.x; e < n; e += blockDim.x) { xc[e] = x0[b * n + e]; Xb[e] = xc[e]; }
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        if (threadIdx.x < 64) {
            const int i = tpwl::nearest_wave(T, xc);
            if (threadIdx.x == 0) *ip = i;
        }
        for (int e = threadIdx.x; e < m; e += blockDim.x) uc[e] = U[(b * N + k) * m + e];
        __syncthreads();
        const int i = *ip;
        // xn = A x + d  (via the transposed table: coalesced), then += B u
        wg::matTvec(xn, T.AdT + (size_t)i * n * n, n, n, n, xc, T.dd + (size_t)i * n, part);
        wg::matTvec(xn, T.BdT + (size_t)i * m * n, n, m, n, uc, (clptr)xn, part);
        for (int e = threadIdx.x; e < n; e += blockDim.x) { xc[e] = xn[e]; Xb[(size_t)(k + 1) * n + e] = xn[e]; }
        __syncthreads();
    }
    if (Z != nullptr && T.H != nullptr) {
        double *Zb = Z + b * (size_t)(N + 1) * T.nz;
        for (int e = threadIdx.x; e < (N + 1) * T.nz; e += blockDim.x) {
            const int k = e / T.nz, a = e % T.nz;
            double s = 0.0;
            for (int j = 0; j < n; ++j) s = fma(T.H[a * n + j], Xb[(size_t)k * n + j], s);
            Zb[e] = s + T.z_ref[a];
        }
    }
}

// f_i = A_c[j] x_i + B_c[j] u_i + d_c[j] at the stored points, j = nearest(x_i)  (models/tpwl.py:77-82)
__global__ __launch_bounds__(64) void char_kernel(TpwlDev T, double *__restrict__ xabs, double *__restrict__ fabs_) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lptr x = (lptr)smem;
    const int i = blockIdx.x, n = T.n, r = T.r;
    for (int e = threadIdx.x; e < n; e += 64) x[e] = e < r ? T.vT[e * T.P + i] : T.qT[(e - r) * T.P + i];
    __syncthreads();
    const int j = tpwl::nearest_wave(T, x);
    for (int e = threadIdx.x; e < n; e += 64) {
        double s = T.dc[(size_t)j * n + e];
        for (int c = 0; c < n; ++c) s = fma(T.Ac[((size_t)j * n + e) * n + c], x[c], s);
        for (int c = 0; c < T.m; ++c) s = fma(T.Bc[((size_t)j * n + e) * T.m + c], T.u[i * T.m + c], s);
        xabs[(size_t)i * n + e] = fabs(x[e]);
        fabs_[(size_t)i * n + e] = fabs(s);
    }
}

__global__ void colmax_kernel(const double *__restrict__ M, int rows, int cols, double *__restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double v = 0.0;
    for (int i = 0; i < rows; ++i) v = fmax(v, M[(size_t)i * cols + c]);
    out[c] = v;
}

std::vector<double> transpose_batch(const double *src, int P, int rows, int cols) {
    std::vector<double> out((size_t)P * rows * cols);
    for (int p = 0; p < P; ++p)
        for (int i = 0; i < rows; ++i)
            for (int j = 0; j < cols; ++j)
                out[((size_t)p * cols + j) * rows + i] = src[((size_t)p * rows + i) * cols + j];
    return out;
}

}  // namespace

TpwlDev stpwl::view() const {
    TpwlDev T{};
    T.P = P; T.r = r; T.n = n; T.m = m; T.nz = nz;
    T.w_q = w_q; T.w_v = w_v;
    auto g = [](const srh::DevBuf &b) { return (cgptr)b.as<double>(); };
    T.qT = g(qT); T.vT = g(vT); T.u = g(u);
    T.Ac = g(Ac); T.Bc = g(Bc); T.dc = g(dc);
    T.AcT = g(AcT); T.BcT = g(BcT);
    T.Ad = g(Ad); T.Bd = g(Bd); T.dd = g(dd);
    T.AdT = g(AdT); T.BdT = g(BdT);
    T.H = g(H); T.z_ref = g(z_ref);
    return T;
}

extern "C" {

int stpwl_set_discrete(stpwl_t *h, const double *A_d, const double *B_d, const double *d_d) {
    SRH_REQUIRE(h && A_d && B_d && d_d, "stpwl_set_discrete: null argument");
    int rc;
    const size_t nn = (size_t)h->P * h->n * h->n, nm = (size_t)h->P * h->n * h->m;
    auto AT = transpose_batch(A_d, h->P, h->n, h->n);
    auto BT = transpose_batch(B_d, h->P, h->n, h->m);
    if ((rc = h->Ad.upload(A_d, sizeof(double) * nn)) || (rc = h->Bd.upload(B_d, sizeof(double) * nm)) ||
        (rc = h->dd.upload(d_d, sizeof(double) * h->P * h->n)) ||
        (rc = h->AdT.upload(AT.data(), sizeof(double) * nn)) || (rc = h->BdT.upload(BT.data(), sizeof(double) * nm)))
        return rc;
    h->has_discrete = true;
    return SRH_OK;
}

int stpwl_create(stpwl_t **out, int P, int r, int n_u, const double *q, const double *v, const double *u,
                 const double *A_c, const double *B_c, const double *d_c, const double *A_d,
                 const double *B_d, const double *d_d, double w_q, double w_v) {
    SRH_REQUIRE(out && q && v && u && A_c && B_c && d_c, "stpwl_create: null argument");
    SRH_REQUIRE(P > 0 && r > 0 && n_u > 0 && n_u <= 16, "stpwl_create: need P, r > 0 and 0 < n_u <= 16");
    stpwl *h = new stpwl();
    h->P = P; h->r = r; h->n = 2 * r; h->m = n_u; h->nz = 0;
    h->w_q = w_q; h->w_v = w_v;
    const int n = h->n, m = h->m;
    auto qT = transpose_batch(q, 1, P, r);
    auto vT = transpose_batch(v, 1, P, r);
    auto AT = transpose_batch(A_c, P, n, n);
    auto BT = transpose_batch(B_c, P, n, m);
    int rc;
    if ((rc = h->qT.upload(qT.data(), sizeof(double) * P * r)) || (rc = h->vT.upload(vT.data(), sizeof(double) * P * r)) ||
        (rc = h->u.upload(u, sizeof(double) * P * m)) ||
        (rc = h->Ac.upload(A_c, sizeof(double) * P * n * n)) || (rc = h->Bc.upload(B_c, sizeof(double) * P * n * m)) ||
        (rc = h->dc.upload(d_c, sizeof(double) * P * n)) ||
        (rc = h->AcT.upload(AT.data(), sizeof(double) * P * n * n)) || (rc = h->BcT.upload(BT.data(), sizeof(double) * P * n * m))) {
        delete h;
        return rc;
    }
    if (A_d && B_d && d_d) {
        if ((rc = stpwl_set_discrete(h, A_d, B_d, d_d))) { delete h; return rc; }
    }
    *out = h;
    return SRH_OK;
}

int stpwl_destroy(stpwl_t *h) {
    delete h;
    return SRH_OK;
}

int stpwl_set_output(stpwl_t *h, const double *H, const double *z_ref, int n_z) {
    SRH_REQUIRE(h && H && n_z > 0, "stpwl_set_output: null argument");
    std::vector<double> zr(n_z, 0.0);
    if (z_ref) zr.assign(z_ref, z_ref + n_z);
    int rc;
    if ((rc = h->H.upload(H, sizeof(double) * n_z * h->n)) || (rc = h->z_ref.upload(zr.data(), sizeof(double) * n_z)))
        return rc;
    h->nz = n_z;
    h->H_host.assign(H, H + (size_t)n_z * h->n);
    h->zref_host = zr;
    return SRH_OK;
}

int stpwl_nearest_dev(stpwl_t *h, const double *X_dev, int64_t B, int32_t *idx_dev, void *stream) {
    SRH_REQUIRE(h && X_dev && idx_dev, "stpwl_nearest_dev: null argument");
    if (B == 0) return SRH_OK;
    nearest_kernel<<<(unsigned)srh::cdiv(B, 4), 256, 0, (hipStream_t)stream>>>(h->view(), X_dev, B, idx_dev);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

int stpwl_nearest(stpwl_t *h, const double *X, int64_t B, int32_t *idx) {
    SRH_REQUIRE(h && X && idx, "stpwl_nearest: null argument");
    if (B == 0) return SRH_OK;
    srh::DevBuf dX, dI;
    int rc;
    if ((rc = dX.upload(X, sizeof(double) * B * h->n)) || (rc = dI.alloc(sizeof(int32_t) * B))) return rc;
    if ((rc = stpwl_nearest_dev(h, dX.as<double>(), B, dI.as<int32_t>(), nullptr))) return rc;
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    return dI.download(idx, sizeof(int32_t) * B);
}

int stpwl_linearize(stpwl_t *h, const double *X, int64_t B, int discrete, double *A, double *Bm, double *d,
                    int32_t *idx) {
    SRH_REQUIRE(h && X && A && Bm && d, "stpwl_linearize: null argument");
    SRH_REQUIRE(!discrete || h->has_discrete, "stpwl_linearize: model has not been pre-discretised");
    if (B == 0) return SRH_OK;
    const int n = h->n, m = h->m;
    srh::DevBuf dX, dI, dA, dB, dd;
    int rc;
    if ((rc = dX.upload(X, sizeof(double) * B * n)) || (rc = dI.alloc(sizeof(int32_t) * B)) ||
        (rc = dA.alloc(sizeof(double) * B * n * n)) || (rc = dB.alloc(sizeof(double) * B * n * m)) ||
        (rc = dd.alloc(sizeof(double) * B * n)))
        return rc;
    if ((rc = stpwl_nearest_dev(h, dX.as<double>(), B, dI.as<int32_t>(), nullptr))) return rc;
    gather_kernel<<<(unsigned)B, 256>>>(h->view(), dI.as<int32_t>(), B, discrete, dA.as<double>(), dB.as<double>(), dd.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if ((rc = dA.download(A, sizeof(double) * B * n * n)) || (rc = dB.download(Bm, sizeof(double) * B * n * m)) ||
        (rc = dd.download(d, sizeof(double) * B * n)))
        return rc;
    if (idx) return dI.download(idx, sizeof(int32_t) * B);
    return SRH_OK;
}

int stpwl_weights(stpwl_t *h, const double *X, int64_t B, double beta, double *W) {
    SRH_REQUIRE(h && X && W, "stpwl_weights: null argument");
    if (B == 0) return SRH_OK;
    srh::DevBuf dX, dW;
    int rc;
    if ((rc = dX.upload(X, sizeof(double) * B * h->n)) || (rc = dW.alloc(sizeof(double) * B * h->P))) return rc;
    weights_kernel<<<(unsigned)B, 256>>>(h->view(), dX.as<double>(), B, beta, dW.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    return dW.download(W, sizeof(double) * B * h->P);
}

int stpwl_linearize_weighted(stpwl_t *h, const double *X, int64_t B, double beta, double *A, double *Bm, double *d,
                             double *W) {
    SRH_REQUIRE(h && X && A && Bm && d, "stpwl_linearize_weighted: null argument");
    if (B == 0) return SRH_OK;
    const int n = h->n, m = h->m;
    srh::DevBuf dX, dW, dA, dB, dd;
    int rc;
    if ((rc = dX.upload(X, sizeof(double) * B * n)) || (rc = dW.alloc(sizeof(double) * B * h->P)) ||
        (rc = dA.alloc(sizeof(double) * B * n * n)) || (rc = dB.alloc(sizeof(double) * B * n * m)) ||
        (rc = dd.alloc(sizeof(double) * B * n)))
        return rc;
    weights_kernel<<<(unsigned)B, 256>>>(h->view(), dX.as<double>(), B, beta, dW.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    const int tot = n * n + n * m + n;
    blend_kernel<<<dim3((unsigned)B, (unsigned)srh::cdiv(tot, 256)), 256>>>(h->view(), dW.as<double>(), B,
                                                                            dA.as<double>(), dB.as<double>(),
                                                                            dd.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if ((rc = dA.download(A, sizeof(double) * B * n * n)) || (rc = dB.download(Bm, sizeof(double) * B * n * m)) ||
        (rc = dd.download(d, sizeof(double) * B * n)))
        return rc;
    if (W) return dW.download(W, sizeof(double) * B * h->P);
    return SRH_OK;
}

int stpwl_rollout(stpwl_t *h, const double *x0, const double *U, int N, int64_t batch, double *X, double *Z) {
    SRH_REQUIRE(h && x0 && U && X, "stpwl_rollout: null argument");
    SRH_REQUIRE(h->has_discrete, "stpwl_rollout: model has not been pre-discretised");
    SRH_REQUIRE(N >= 0 && batch >= 0, "stpwl_rollout: negative size");
    SRH_REQUIRE(Z == nullptr || h->nz > 0, "stpwl_rollout: Need to set output or meas. model");
    if (batch == 0) return SRH_OK;
    const int n = h->n, m = h->m;
    srh::DevBuf d0, dU, dX, dZ;
    int rc;
    if ((rc = d0.upload(x0, sizeof(double) * batch * n)) || (rc = dU.upload(U, sizeof(double) * batch * N * m)) ||
        (rc = dX.alloc(sizeof(double) * batch * (N + 1) * n)))
        return rc;
    if (Z && (rc = dZ.alloc(sizeof(double) * batch * (N + 1) * h->nz))) return rc;
    size_t lds = sizeof(double) * (2 * n + ((m + 3) & ~3) + 256) + 16;
    rollout_kernel<<<(unsigned)batch, 256, lds>>>(h->view(), d0.as<double>(), dU.as<double>(), N, dX.as<double>(),
                                                 Z ? dZ.as<double>() : nullptr);
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if ((rc = dX.download(X, sizeof(double) * batch * (N + 1) * n))) return rc;
    if (Z) return dZ.download(Z, sizeof(double) * batch * (N + 1) * h->nz);
    return SRH_OK;
}

int stpwl_rollout_dev(stpwl_t *h, const double *x0_dev, const double *U_dev, int N, int64_t batch, double *X_dev, double *Z_dev,
                      void *stream) {
    SRH_REQUIRE(h && x0_dev && U_dev && X_dev, "stpwl_rollout_dev: null argument");
    SRH_REQUIRE(h->has_discrete, "stpwl_rollout_dev: model has not been pre-discretised");
    SRH_REQUIRE(N >= 0 && batch >= 0, "stpwl_rollout_dev: negative size");
    SRH_REQUIRE(Z_dev == nullptr || h->nz > 0, "stpwl_rollout_dev: Need to set output or meas. model");
    if (batch == 0) return SRH_OK;
    const size_t lds = sizeof(double) * (2 * h->n + ((h->m + 3) & ~3) + 256) + 16;
    rollout_kernel<<<(unsigned)batch, 256, lds, (hipStream_t)stream>>>(h->view(), x0_dev, U_dev, N, X_dev, Z_dev);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

int stpwl_characteristic(stpwl_t *h, double *x_char, double *f_char) {
    SRH_REQUIRE(h && x_char && f_char, "stpwl_characteristic: null argument");
    const int n = h->n;
    srh::DevBuf xa, fa, xo, fo;
    int rc;
    if ((rc = xa.alloc(sizeof(double) * h->P * n)) || (rc = fa.alloc(sizeof(double) * h->P * n)) ||
        (rc = xo.alloc(sizeof(double) * n)) || (rc = fo.alloc(sizeof(double) * n)))
        return rc;
    char_kernel<<<h->P, 64, sizeof(double) * n>>>(h->view(), xa.as<double>(), fa.as<double>());
    colmax_kernel<<<(unsigned)srh::cdiv(n, 64), 64>>>(xa.as<double>(), h->P, n, xo.as<double>());
    colmax_kernel<<<(unsigned)srh::cdiv(n, 64), 64>>>(fa.as<double>(), h->P, n, fo.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    if ((rc = xo.download(x_char, sizeof(double) * n))) return rc;
    return fo.download(f_char, sizeof(double) * n);
}

}  // extern "C"
