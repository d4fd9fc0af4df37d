// Symmetric eigendecomposition of the snapshot Gramian on the device (method-of-snapshots POD,
// sofacontrol/mor/pod.py:181-200 takes a thin SVD instead) and selection of the kept modes.
// n_s <= 128: one-workgroup Jacobi in LDS; n_s <= 2048: the same Jacobi over HBM, two launches per step; larger: a
// plain library call, rocSOLVER dsyevd, resolved at run time with dlopen so that the rest of the library does not
// depend on it (its first load in a process takes minutes on a cold box).
#include "common.h"
#include "dev_la.h"

#include <dlfcn.h>
#include <cmath>
#include <vector>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

namespace {

struct Solver {
    void *lib_blas = nullptr, *lib_solver = nullptr;
    decltype(&rocblas_create_handle) create = nullptr;
    decltype(&rocblas_destroy_handle) destroy = nullptr;
    decltype(&rocblas_set_stream) set_stream = nullptr;
    decltype(&rocsolver_dsyevd) dsyevd = nullptr;
    rocblas_handle handle = nullptr;
    bool ok = false;
};

Solver &solver() {
    static Solver s;
    if (s.ok) return s;
    s.lib_blas = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    if (!s.lib_blas) s.lib_blas = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_GLOBAL);
    s.lib_solver = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
    if (!s.lib_solver) s.lib_solver = dlopen("/opt/rocm/lib/librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
    if (!s.lib_blas || !s.lib_solver) return s;
    s.create = (decltype(s.create))dlsym(s.lib_blas, "rocblas_create_handle");
    s.destroy = (decltype(s.destroy))dlsym(s.lib_blas, "rocblas_destroy_handle");
    s.set_stream = (decltype(s.set_stream))dlsym(s.lib_blas, "rocblas_set_stream");
    s.dsyevd = (decltype(s.dsyevd))dlsym(s.lib_solver, "rocsolver_dsyevd");
    if (!s.create || !s.destroy || !s.set_stream || !s.dsyevd) return s;
    if (s.create(&s.handle) != rocblas_status_success) return s;
    s.ok = true;
    return s;
}

// Wk[j][i] = V_{n-1-i}[j] / sqrt(w_{n-1-i})  (i-th largest eigenpair; eigenvector i' is row i' of V)
__global__ void select_modes_kernel(const double *__restrict__ V, const double *__restrict__ w, int64_t n, int k,
                                    double *__restrict__ Wk) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= n) return;
    const int64_t src = n - 1 - i;
    const double sig = sqrt(fmax(w[src], 0.0));
    Wk[j * k + i] = V[src * n + j] / sig;
}


// ---- small Gramians (n <= 128, e.g. a few dozen snapshots): cyclic two-sided Jacobi in one workgroup.
// A lives in LDS (odd leading dimension), the eigenvectors as ROWS of Vt in HBM/L2 (row rotations: coalesced).
// Round-robin ordering: n/2 disjoint rotations per step, n - 1 steps per sweep; each step = compute (c, s) of every
// pair, rotate the rows of A and Vt, barrier, rotate the columns of A.  No library, no 200 s rocBLAS cold load.
constexpr int JAC_NT = 512;
constexpr int JAC_MAX = 128;

__global__ __launch_bounds__(JAC_NT) void jacobi_eigh_kernel(double *__restrict__ G, int n, double *__restrict__ Vt,
                                                             double *__restrict__ w, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ne = (n + 1) & ~1, h = ne >> 1, ld = ne | 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    lptr A = (lptr)smem;                       // ne x ld
    lptr cs = A + (size_t)ne * ld;             // (c, s) per pair: 2 h
    lptr red = cs + 2 * h;                     // 64
    liptr top = (liptr)(red + 64), bot = top + h, top2 = bot + h, bot2 = top2 + h;
    for (int e = tid; e < ne * ne; e += nt) {
        const int i = e / ne, j = e % ne;
        A[i * ld + j] = (i < n && j < n) ? G[(size_t)i * n + j] : 0.0;
        Vt[e] = (i == j) ? 1.0 : 0.0;
    }
    for (int i = tid; i < h; i += nt) { top[i] = 2 * i; bot[i] = 2 * i + 1; }
    __syncthreads();
    double diag2 = 0.0;
    for (int i = tid; i < n; i += nt) diag2 = fma(A[i * ld + i], A[i * ld + i], diag2);
    diag2 = wg::reduce(diag2, 0, red);
    int sweep = 0;
    bool done = false;
    for (; sweep < 40 && !done; ++sweep) {
        for (int step = 0; step < ne - 1; ++step) {
            for (int i = tid; i < h; i += nt) {
                const int p = top[i], q = bot[i];
                const double apq = A[p * ld + q], app = A[p * ld + p], aqq = A[q * ld + q];
                double c = 1.0, sn = 0.0;
                if (fabs(apq) > 1e-300 && fabs(apq) > 1e-30 * (fabs(app) + fabs(aqq))) {
                    const double th = (aqq - app) / (2.0 * apq);
                    const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0);
                    sn = t * c;
                }
                cs[2 * i] = c; cs[2 * i + 1] = sn;
            }
            __syncthreads();
            // rows of A and of Vt:  r_p' = c r_p - s r_q,  r_q' = s r_p + c r_q
            for (int e = tid; e < h * ne; e += nt) {
                const int i = e / ne, k = e % ne;
                const int p = top[i], q = bot[i];
                const double c = cs[2 * i], sn = cs[2 * i + 1];
                const double ap = A[p * ld + k], aq = A[q * ld + k];
                A[p * ld + k] = c * ap - sn * aq;
                A[q * ld + k] = sn * ap + c * aq;
                const double vp = Vt[(size_t)p * ne + k], vq = Vt[(size_t)q * ne + k];
                Vt[(size_t)p * ne + k] = c * vp - sn * vq;
                Vt[(size_t)q * ne + k] = sn * vp + c * vq;
            }
            __syncthreads();
            // columns of A
            for (int e = tid; e < h * ne; e += nt) {
                const int i = e / ne, k = e % ne;
                const int p = top[i], q = bot[i];
                const double c = cs[2 * i], sn = cs[2 * i + 1];
                const double ap = A[k * ld + p], aq = A[k * ld + q];
                A[k * ld + p] = c * ap - sn * aq;
                A[k * ld + q] = sn * ap + c * aq;
            }
            // next pairing (chess-tournament rotation, player top[0] fixed)
            for (int i = tid; i < h; i += nt) {
                top2[i] = i == 0 ? top[0] : (i == 1 ? bot[0] : top[i - 1]);
                bot2[i] = i == h - 1 ? top[h - 1] : bot[i + 1];
            }
            __syncthreads();
            for (int i = tid; i < h; i += nt) { top[i] = top2[i]; bot[i] = bot2[i]; }
            __syncthreads();
        }
        double off2 = 0.0;
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e % n;
            if (i != j) off2 = fma(A[i * ld + j], A[i * ld + j], off2);
        }
        off2 = wg::reduce(off2, 0, red);
        done = off2 <= 1e-30 * diag2 || off2 == 0.0;
        // the diagonal carries the norm after the first sweep
        double d2 = 0.0;
        for (int i = tid; i < n; i += nt) d2 = fma(A[i * ld + i], A[i * ld + i], d2);
        diag2 = wg::reduce(d2, 0, red);
    }
    // ascending order: rank of eigenvalue p (ties by index); row rank_p of G <- eigenvector p (first n entries)
    for (int p = tid; p < n; p += nt) {
        const double lp = A[p * ld + p];
        int rk = 0;
        for (int q = 0; q < n; ++q) {
            const double lq = A[q * ld + q];
            rk += (lq < lp || (lq == lp && q < p)) ? 1 : 0;
        }
        w[rk] = lp;
        // the dummy row / column of an odd n never mixes (its off-diagonals are exactly zero)
        for (int k = 0; k < n; ++k) G[(size_t)rk * n + k] = Vt[(size_t)p * ne + k];
    }
    if (tid == 0) *info = done ? 0 : 1;
}

// ---- medium Gramians (128 < n <= 2048: a typical snapshot set): the same cyclic Jacobi with A and Vt in HBM/L2, one
// pair of launches per round-robin step (n/2 disjoint rotations): `jac_angles_kernel` takes (c, s) of every pair from
// the current diagonal blocks, `jac_apply_kernel` applies J^T A J on disjoint 2 x 2 blocks (pair i x pair j: one
// thread each, in place) and rotates the rows of Vt.  Pairing by the circle method, computed from the step number:
// pair 0 = (ne-1, s), pair i = ((s+i) mod (ne-1), (s-i) mod (ne-1)).  ~10 sweeps x (n-1) steps x 2 launches: tens of
// milliseconds at n = 1000 -- against minutes for the first rocSOLVER / rocBLAS load of a process on a cold box.
constexpr int JAC_GRID_MAX = 2048;

__device__ __forceinline__ void jac_pair(int i, int step, int ne, int &p, int &q) {
    const int m = ne - 1;
    if (i == 0) { p = m; q = step; }
    else { p = (step + i) % m; q = (step - i + m) % m; }
    if (p > q) { const int t = p; p = q; q = t; }
}

__global__ void jac_init_kernel(const double *__restrict__ G, int n, int ne, double *__restrict__ A, double *__restrict__ Vt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)ne * ne) return;
    const int i = (int)(e / ne), j = (int)(e % ne);
    A[e] = (i < n && j < n) ? G[(size_t)i * n + j] : 0.0;
    Vt[e] = (i == j) ? 1.0 : 0.0;
}

__global__ void jac_angles_kernel(const double *__restrict__ A, int ne, int step, double *__restrict__ cs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ne / 2) return;
    int p, q;
    jac_pair(i, step, ne, p, q);
    const double apq = A[(size_t)p * ne + q], app = A[(size_t)p * ne + p], aqq = A[(size_t)q * ne + q];
    double c = 1.0, sn = 0.0;
    if (fabs(apq) > 1e-300 && fabs(apq) > 1e-30 * (fabs(app) + fabs(aqq))) {
        const double th = (aqq - app) / (2.0 * apq);
        const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
        c = 1.0 / sqrt(t * t + 1.0);
        sn = t * c;
    }
    cs[2 * i] = c; cs[2 * i + 1] = sn;
}

// blockIdx.y < h: 2 x 2 blocks of A for row pair blockIdx.y; blockIdx.y >= h: rows of Vt for pair blockIdx.y - h
__global__ __launch_bounds__(256) void jac_apply_kernel(double *__restrict__ A, double *__restrict__ Vt, int ne, int step,
                                                        const double *__restrict__ cs) {
    const int h = ne >> 1;
    const bool isV = (int)blockIdx.y >= h;
    const int i = isV ? blockIdx.y - h : blockIdx.y;
    int p, q;
    jac_pair(i, step, ne, p, q);
    const double c = cs[2 * i], sn = cs[2 * i + 1];
    if (isV) {
        for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < ne; k += gridDim.x * blockDim.x) {
            const double vp = Vt[(size_t)p * ne + k], vq = Vt[(size_t)q * ne + k];
            Vt[(size_t)p * ne + k] = c * vp - sn * vq;
            Vt[(size_t)q * ne + k] = sn * vp + c * vq;
        }
        return;
    }
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < h; j += gridDim.x * blockDim.x) {
        int pj, qj;
        jac_pair(j, step, ne, pj, qj);
        const double cj = cs[2 * j], sj = cs[2 * j + 1];
        const double a00 = A[(size_t)p * ne + pj], a01 = A[(size_t)p * ne + qj];
        const double a10 = A[(size_t)q * ne + pj], a11 = A[(size_t)q * ne + qj];
        // rows:  r_p' = c r_p - s r_q,  r_q' = s r_p + c r_q ; then the same on the columns with (cj, sj)
        const double b00 = c * a00 - sn * a10, b01 = c * a01 - sn * a11;
        const double b10 = sn * a00 + c * a10, b11 = sn * a01 + c * a11;
        A[(size_t)p * ne + pj] = cj * b00 - sj * b01;
        A[(size_t)p * ne + qj] = sj * b00 + cj * b01;
        A[(size_t)q * ne + pj] = cj * b10 - sj * b11;
        A[(size_t)q * ne + qj] = sj * b10 + cj * b11;
    }
}

// out[0] = sum of squares of the off-diagonal, out[1] = of the diagonal (fixed-order block partials, then one block)
__global__ __launch_bounds__(256) void jac_norms_kernel(const double *__restrict__ A, int ne, double *__restrict__ part) {
    __shared__ double red[2][4];
    double off = 0.0, dg = 0.0;
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < ne; j += blockDim.x) {
        const double v = A[(size_t)i * ne + j];
        if (i == j) dg = v * v; else off = fma(v, v, off);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dg += __shfl_xor(dg, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = off; red[1][threadIdx.x >> 6] = dg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * i] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        part[2 * i + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ void jac_rank_kernel(const double *__restrict__ A, int n, int ne, double *__restrict__ w, int *__restrict__ rank) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double lp = A[(size_t)p * ne + p];
    int rk = 0;
    for (int q = 0; q < n; ++q) {
        const double lq = A[(size_t)q * ne + q];
        rk += (lq < lp || (lq == lp && q < p)) ? 1 : 0;
    }
    w[rk] = lp;
    rank[p] = rk;
}

__global__ void jac_gather_kernel(const double *__restrict__ Vt, const int *__restrict__ rank, int n, int ne,
                                  double *__restrict__ G) {
    const int p = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) G[(size_t)rank[p] * n + k] = Vt[(size_t)p * ne + k];
}

int jacobi_grid(double *G_dev, int n, double *w_dev, hipStream_t st) {
    const int ne = (n + 1) & ~1, h = ne >> 1;
    srh::DevBuf A, Vt, cs, part, rank;
    int rc;
    if ((rc = A.alloc(sizeof(double) * ne * ne)) || (rc = Vt.alloc(sizeof(double) * ne * ne)) ||
        (rc = cs.alloc(sizeof(double) * 2 * h)) || (rc = part.alloc(sizeof(double) * 2 * ne)) ||
        (rc = rank.alloc(sizeof(int) * n)))
        return rc;
    jac_init_kernel<<<(unsigned)srh::cdiv((int64_t)ne * ne, 256), 256, 0, st>>>(G_dev, n, ne, A.as<double>(), Vt.as<double>());
    std::vector<double> hp(2 * (size_t)ne);
    const dim3 grid_apply((unsigned)srh::cdiv(ne, 256), (unsigned)(2 * h));
    double prev = INFINITY;
    bool done = false;
    for (int sweep = 0; sweep < 40 && !done; ++sweep) {
        for (int step = 0; step < ne - 1; ++step) {
            jac_angles_kernel<<<(unsigned)srh::cdiv(h, 128), 128, 0, st>>>(A.as<double>(), ne, step, cs.as<double>());
            jac_apply_kernel<<<grid_apply, 256, 0, st>>>(A.as<double>(), Vt.as<double>(), ne, step, cs.as<double>());
        }
        jac_norms_kernel<<<(unsigned)ne, 256, 0, st>>>(A.as<double>(), ne, part.as<double>());
        SRH_CHECK_HIP(hipGetLastError());
        SRH_CHECK_HIP(hipMemcpyAsync(hp.data(), part.p, sizeof(double) * 2 * ne, hipMemcpyDeviceToHost, st));
        SRH_CHECK_HIP(hipStreamSynchronize(st));
        double off2 = 0.0, diag2 = 0.0;
        for (int i = 0; i < ne; ++i) { off2 += hp[2 * i]; diag2 += hp[2 * i + 1]; }
        // converged, or stagnating at the rounding floor of a large matrix (n eps^2 relative)
        done = off2 <= 1e-30 * diag2 || (off2 <= 1e-26 * diag2 && off2 > 0.25 * prev);
        prev = off2;
    }
    if (!done) {
        srh::set_error("srom_eigh_dev: Jacobi sweeps did not converge");
        return SRH_ENUMERIC;
    }
    jac_rank_kernel<<<(unsigned)srh::cdiv(n, 128), 128, 0, st>>>(A.as<double>(), n, ne, w_dev, rank.as<int>());
    jac_gather_kernel<<<dim3((unsigned)srh::cdiv(n, 256), (unsigned)n), 256, 0, st>>>(Vt.as<double>(), rank.as<int>(), n, ne, G_dev);
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(st));
    return SRH_OK;
}

}  // namespace

extern "C" {

int srom_eigh_dev(double *G_dev, int64_t n, double *w_dev, void *stream) {
    SRH_REQUIRE(G_dev && w_dev && n > 0 && n < (1LL << 31), "srom_eigh_dev: bad argument");
    if (n <= JAC_MAX && !getenv("SRH_EIGH_ROCSOLVER")) {
        const int ne = ((int)n + 1) & ~1, ld = ne | 1;
        const size_t lds = srh::lds_request(sizeof(double) * ((size_t)ne * ld + ne + 64) + sizeof(int) * 2 * ne + 64);
        srh::DevBuf Vt, info;
        int rc;
        if ((rc = Vt.alloc(sizeof(double) * ne * ne)) || (rc = info.alloc(sizeof(int)))) return rc;
        SRH_CHECK_HIP(hipFuncSetAttribute((const void *)jacobi_eigh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
        jacobi_eigh_kernel<<<1, JAC_NT, lds, (hipStream_t)stream>>>(G_dev, (int)n, Vt.as<double>(), w_dev, info.as<int>());
        SRH_CHECK_HIP(hipGetLastError());
        SRH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
        int hinfo = 0;
        SRH_CHECK_HIP(hipMemcpy(&hinfo, info.p, sizeof(int), hipMemcpyDeviceToHost));
        if (hinfo != 0) {
            srh::set_error("srom_eigh_dev: Jacobi sweeps did not converge");
            return SRH_ENUMERIC;
        }
        return SRH_OK;
    }
    if (n <= JAC_GRID_MAX && !getenv("SRH_EIGH_ROCSOLVER")) return jacobi_grid(G_dev, (int)n, w_dev, (hipStream_t)stream);
    Solver &s = solver();
    if (!s.ok) {
        srh::set_error("srom_eigh_dev: rocSOLVER / rocBLAS could not be loaded (%s)", dlerror() ? dlerror() : "symbol missing");
        return SRH_EHIP;
    }
    srh::DevBuf E, info;
    int rc;
    if ((rc = E.alloc(sizeof(double) * n)) || (rc = info.alloc(sizeof(rocblas_int)))) return rc;
    if (s.set_stream(s.handle, (hipStream_t)stream) != rocblas_status_success) {
        srh::set_error("srom_eigh_dev: rocblas_set_stream failed");
        return SRH_EHIP;
    }
    // G is symmetric, so its row-major storage is also its column-major storage; on return column j
    // (= row j of the row-major view) holds the eigenvector of the j-th smallest eigenvalue.
    const rocblas_status st = s.dsyevd(s.handle, rocblas_evect_original, rocblas_fill_upper, (rocblas_int)n, G_dev,
                                       (rocblas_int)n, w_dev, E.as<double>(), info.as<rocblas_int>());
    if (st != rocblas_status_success) {
        srh::set_error("srom_eigh_dev: rocsolver_dsyevd returned status %d", (int)st);
        return SRH_EHIP;
    }
    SRH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    rocblas_int hinfo = 0;
    SRH_CHECK_HIP(hipMemcpy(&hinfo, info.p, sizeof(hinfo), hipMemcpyDeviceToHost));
    if (hinfo != 0) {
        srh::set_error("srom_eigh_dev: eigensolver did not converge (info = %d)", (int)hinfo);
        return SRH_ENUMERIC;
    }
    return SRH_OK;
}

int srom_select_modes_dev(const double *V_dev, const double *w_dev, int64_t n, int k, double *Wk_dev, void *stream) {
    SRH_REQUIRE(V_dev && w_dev && Wk_dev && n > 0 && k > 0 && k <= n, "srom_select_modes_dev: bad argument");
    select_modes_kernel<<<dim3((unsigned)srh::cdiv(n, 256), (unsigned)k), 256, 0, (hipStream_t)stream>>>(V_dev, w_dev, n, k,
                                                                                                        Wk_dev);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

}  // extern "C"
