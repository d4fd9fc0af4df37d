// Symmetric eigendecomposition of the snapshot Gramian on the device (method-of-snapshots POD,
// sofacontrol/mor/pod.py:181-200 takes a thin SVD instead) and selection of the kept modes.
// The dense n_s x n_s eigenproblem is a plain library call: rocSOLVER dsyevd, resolved at run time with
// dlopen so that the rest of the library does not depend on it.
#include "common.h"

#include <dlfcn.h>
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

}  // namespace

extern "C" {

int srom_eigh_dev(double *G_dev, int64_t n, double *w_dev, void *stream) {
    SRH_REQUIRE(G_dev && w_dev && n > 0 && n < (1LL << 31), "srom_eigh_dev: bad argument");
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
