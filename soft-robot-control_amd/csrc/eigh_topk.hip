// Leading eigenpairs of the snapshot Gramian (method-of-snapshots POD, sofacontrol/mor/pod.py:181-200 keeps the first k
// singular triplets of the snapshot matrix: sigma_i^2 = the k LARGEST eigenvalues of G = S S^T) by blocked subspace
// iteration with a final Rayleigh-Ritz step -- instead of the full spectrum of a 10 000 x 10 000 matrix (rocSOLVER dsyevd:
// 1.08 s, 98 % of the C4 build in round 2) when only rom_dim = 64 modes are kept.  The tail energy the truncation rule
// needs is trace(G) - sum of the kept eigenvalues: no further eigenvalue is required.
//
//   Q (b x n, rows = vectors, b = k + oversampling <= 128)   <- b rows of G, orthonormalised
//   repeat:  Z = Q G                      (f64 MFMA, C = A B^T with B = G symmetric: 2 n^2 b flop, split along K)
//            M = Z Z^T = V D V^T          (b x b; one-workgroup LDS Jacobi of eigh.hip)
//            Q = D^-1/2 V^T Z             (the left singular vectors of Z: an orthonormal basis of span(G Q), ordered)
//            until the estimates sqrt(D_i), i < k, stop moving
//   Rayleigh-Ritz:  Z = Q G,  T = Q Z^T = W Theta W^T,  eigenvectors = W^T Q,  eigenvalues = Theta.
// Deterministic: fixed split-K partial sums reduced in order, fixed start block.
#include <algorithm>
#include <cmath>
#include <vector>

#include "abt_dev.h"

extern "C" int srom_eigh_dev(double *G_dev, int64_t n, double *w_dev, void *stream);

namespace {

// Qn (b x n) = Wm (b x b) Zt (b x n): a workgroup per 64 columns, 4 groups of 32 output rows; the coefficients are
// wave-uniform (scalar loads), the column block of Zt is staged in LDS
__global__ __launch_bounds__(256) void wz_kernel(const double *__restrict__ Wm, int b, const double *__restrict__ Zt, int64_t n,
                                                 double *__restrict__ Qn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lptr Zs = (lptr)smem;                                   // [b][64]
    const int tid = threadIdx.x, jc = tid & 63, ig = tid >> 6;
    const int64_t j = (int64_t)blockIdx.x * 64 + jc;
    for (int e = tid; e < b * 64; e += 256) {
        const int l = e >> 6;
        const int64_t jj = (int64_t)blockIdx.x * 64 + (e & 63);
        Zs[e] = jj < n ? Zt[(int64_t)l * n + jj] : 0.0;
    }
    __syncthreads();
    for (int i0 = ig * 32; i0 < b; i0 += 128) {
        double acc[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) acc[q] = 0.0;
        for (int l = 0; l < b; ++l) {
            const double z = Zs[l * 64 + jc];
#pragma unroll
            for (int q = 0; q < 32; ++q) acc[q] = fma(Wm[(size_t)min(i0 + q, b - 1) * b + l], z, acc[q]);
        }
        if (j < n) {
#pragma unroll
            for (int q = 0; q < 32; ++q) if (i0 + q < b) Qn[(int64_t)(i0 + q) * n + j] = acc[q];
        }
    }
}

// from the eigen-decomposition of a b x b matrix (rows of V = eigenvectors, w ascending): Wm[i][:] = s_i V[b-1-i][:], i-th
// LARGEST first; s_i = 1 / sqrt(w) (scale = 1: orthonormalisation through M = Z Z^T) or 1 (scale = 0: Ritz rotation).
// est[i] = sqrt(w) (scale = 1) or w (scale = 0).  Directions whose eigenvalue has cancelled to nothing get a zero row.
__global__ void build_w_kernel(const double *__restrict__ V, const double *__restrict__ w, int b, int scale, double *__restrict__ Wm,
                               double *__restrict__ est) {
    const int i = blockIdx.x, src = b - 1 - i;
    const double wi = w[src], wmax = w[b - 1];
    double s = 1.0;
    if (scale) s = wi > 1e-28 * wmax && wi > 0.0 ? 1.0 / sqrt(wi) : 0.0;
    for (int l = threadIdx.x; l < b; l += blockDim.x) Wm[(size_t)i * b + l] = s * V[(size_t)src * b + l];
    if (threadIdx.x == 0) est[i] = scale ? sqrt(fmax(wi, 0.0)) : wi;
}

// start block: b rows of G spread over the matrix (they lie in its range)
__global__ void start_rows_kernel(const double *__restrict__ G, int64_t n, int b, double *__restrict__ Qt) {
    const int i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t row = (int64_t)i * n / b;
    Qt[(int64_t)i * n + j] = G[row * n + j];
}

__global__ void symmetrise_kernel(double *__restrict__ T, int b) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b * b) return;
    const int i = e / b, j = e - i * b;
    if (i < j) { const double v = 0.5 * (T[i * b + j] + T[j * b + i]); T[i * b + j] = v; T[j * b + i] = v; }
}

__global__ void trace_kernel(const double *__restrict__ G, int64_t n, double *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += G[i * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

// Wk[j][i] = Vt[i][j] / sqrt(w_i)   (n x k, the layout srom_modes_dev takes)
__global__ void scale_cols_kernel(const double *__restrict__ Vt, const double *__restrict__ w, int64_t n, int k, double *__restrict__ Wk) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n) return;
    Wk[j * k + i] = Vt[(int64_t)i * n + j] / sqrt(fmax(w[i], 0.0));
}


}  // namespace

extern "C" {

int srom_eigh_topk_dev(const double *G_dev, int64_t n, int k, int oversample, double *w_dev, double *Wk_dev, double *Vt_dev,
                       double *trace_out, int *iters_out, void *stream) {
    SRH_REQUIRE(G_dev && w_dev && n > 0 && k > 0, "srom_eigh_topk_dev: bad argument");
    if (oversample < 0) oversample = std::max(16, k / 2);
    int b = (k + oversample + 15) & ~15;
    if (b > n) b = (int)n;
    SRH_REQUIRE(k <= b && b <= 128, "srom_eigh_topk_dev: k + oversampling must be <= 128 (got k = %d, block %d); use srom_eigh_dev for more modes", k, b);
    hipStream_t st = (hipStream_t)stream;
    Abt abt;                             // split-K scratch of this call (two calls on different streams must not share it)
    srh::DevBuf Qt, Zt, Mm, Wm, wv, est;
    int rc;
    if ((rc = Qt.alloc(sizeof(double) * (size_t)b * n)) || (rc = Zt.alloc(sizeof(double) * (size_t)b * n)) || (rc = Mm.alloc(sizeof(double) * b * b)) ||
        (rc = Wm.alloc(sizeof(double) * b * b)) || (rc = wv.alloc(sizeof(double) * b)) || (rc = est.alloc(sizeof(double) * (b + 1))))
        return rc;
    double *dQ = Qt.as<double>(), *dZ = Zt.as<double>(), *dM = Mm.as<double>(), *dW = Wm.as<double>(), *dw = wv.as<double>(), *de = est.as<double>();
    const size_t wz_lds = sizeof(double) * (size_t)b * 64;
    auto orthonormalise = [&](double *src, double *dst) -> int {          // dst = left singular vectors of src (rows), est = sigma
        int r2;
        if ((r2 = abt.run(src, n, b, src, n, b, n, dM, b, st))) return r2;
        if ((r2 = srom_eigh_dev(dM, b, dw, stream))) return r2;
        build_w_kernel<<<b, 128, 0, st>>>(dM, dw, b, 1, dW, de);
        wz_kernel<<<(unsigned)srh::cdiv(n, 64), 256, wz_lds, st>>>(dW, b, src, n, dst);
        SRH_CHECK_HIP(hipGetLastError());
        return SRH_OK;
    };
    trace_kernel<<<1, 256, 0, st>>>(G_dev, n, de + b);
    start_rows_kernel<<<dim3((unsigned)srh::cdiv(n, 256), (unsigned)b), 256, 0, st>>>(G_dev, n, b, dZ);
    SRH_CHECK_HIP(hipGetLastError());
    if ((rc = orthonormalise(dZ, dQ))) return rc;
    std::vector<double> prev(b, 0.0), cur(b + 1, 0.0);
    int it = 0;
    bool settled = false;
    const int max_it = 30;               // a block that has not settled by then sits in a flat part of the spectrum (the caller falls back)
    for (; it < max_it && !settled; ++it) {
        if ((rc = abt.run(dQ, n, b, G_dev, n, n, n, dZ, n, st))) return rc;          // Z = Q G   (G symmetric: Q G^T)
        if ((rc = orthonormalise(dZ, dQ))) return rc;
        SRH_CHECK_HIP(hipMemcpyAsync(cur.data(), de, sizeof(double) * (b + 1), hipMemcpyDeviceToHost, st));
        SRH_CHECK_HIP(hipStreamSynchronize(st));
        double change = 0.0;
        for (int i = 0; i < k; ++i) change = std::max(change, std::fabs(cur[i] - prev[i]));
        std::copy(cur.begin(), cur.begin() + b, prev.begin());
        settled = it >= 1 && change <= 1e-13 * cur[0];
    }
    // Rayleigh-Ritz on the converged block
    if ((rc = abt.run(dQ, n, b, G_dev, n, n, n, dZ, n, st))) return rc;
    if ((rc = abt.run(dQ, n, b, dZ, n, b, n, dM, b, st))) return rc;                 // T = Q Z^T
    symmetrise_kernel<<<(unsigned)srh::cdiv(b * b, 256), 256, 0, st>>>(dM, b);
    if ((rc = srom_eigh_dev(dM, b, dw, stream))) return rc;
    build_w_kernel<<<b, 128, 0, st>>>(dM, dw, b, 0, dW, de);
    wz_kernel<<<(unsigned)srh::cdiv(n, 64), 256, wz_lds, st>>>(dW, b, dQ, n, dZ);   // eigenvectors (rows, largest first)
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipMemcpyAsync(w_dev, de, sizeof(double) * k, hipMemcpyDeviceToDevice, st));
    if (Vt_dev) SRH_CHECK_HIP(hipMemcpyAsync(Vt_dev, dZ, sizeof(double) * (size_t)k * n, hipMemcpyDeviceToDevice, st));
    if (Wk_dev) scale_cols_kernel<<<dim3((unsigned)srh::cdiv(n, 256), (unsigned)k), 256, 0, st>>>(dZ, de, n, k, Wk_dev);
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(st));
    if (trace_out) *trace_out = cur[b];
    if (iters_out) *iters_out = settled ? it : max_it + 1;      // <= 30: settled after that many; 31: not settled
    return SRH_OK;
}

}  // extern "C"
