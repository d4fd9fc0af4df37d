// Symmetric eigendecomposition of the snapshot Gramian on the device (method-of-snapshots POD,
// sofacontrol/mor/pod.py:181-200 takes a thin SVD instead) and selection of the kept modes.
// n_s <= 128: one-workgroup Jacobi in LDS; n_s <= 256: the same Jacobi over HBM, two launches per step; n_s <= 4096: the
// block Jacobi below; larger: rocSOLVER's dsyevd, resolved at run time with dlopen so that the library does not depend on it (its first load in a process
// takes minutes on a cold box) -- and where it cannot be loaded, or under SRH_EIGH_BLOCK=1, a two-sided block Jacobi on the
// MFMA pipe (round 6; no library).
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
constexpr int JAC_BLOCK_MIN = 256;       // above: the block Jacobi further down (SRH_EIGH_BLOCK=0: the scalar form up to 2048)
constexpr int JAC_BLOCK_MAX = 4096;      // up to here the block Jacobi is the default: 0.28 s at 2048, 0.53 s at 3000 -- rocSOLVER is faster once
                                         // resident (0.06 / 0.1 s), but its first load in a process costs up to two minutes

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
        if (getenv("SRH_EIGH_TRACE")) fprintf(stderr, "scalar Jacobi n %d sweep %d: off^2 / diag^2 = %.3e\n", n, sweep, off2 / diag2);
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

// ---- large Gramians (n > 2048, e.g. BASELINE C4's 10 000 snapshots): two-sided BLOCK Jacobi, no library.  The matrix is cut into
// blocks of 64; a round-robin step pairs the blocks (the circle method of `jac_pair` on block numbers), every pair is a 128 x 128
// symmetric sub-problem [A_PP A_PQ; A_QP A_QQ] that one workgroup takes through one cyclic Jacobi sweep in LDS (`bj_sub_kernel`: the
// small-Gramian kernel's loop, its accumulated rotations Qt = J_last' ... J_1' in L2), and the step applies all of them at once:
//   A[I, J] <- Qt_I A[I, J] Qt_J'   per pair of pairs (I <= J; the mirror tile is its transpose: A stays exactly symmetric),
//   Vt[I, :] <- Qt_I Vt[I, :],
// two chained 128^3 products per tile on the f64 MFMA pipe with the tile in LDS (`bj_update_a_kernel`, `bj_update_v_kernel`).
// Per step 2 n^2 x 128 x 2 flops instead of the n^2 x 8 bytes of traffic PER ROTATION of the scalar form: the scalar form at n = 10 000
// would move 32 TB per sweep.  Pairs whose off-diagonal block is already at the rounding floor are skipped (identity, flag 0).
// Pair size P (template): 128 (64-wide blocks; Qt of the pair problem in L2) or 64 (32-wide blocks: S AND Qt of the pair problem in LDS,
// an inner sweep of 63 steps at LDS latency -- 7 x faster than the 127 steps of the larger pair with Qt behind L2, for twice as many
// outer steps of half the flops each).  The update kernels run P / 16 waves (one 16-row strip each), tile row stride P + 16 doubles:
// (4 s + kk) ld + l16 hits 32 different 8-byte banks per half wave.
typedef double bj_d4 __attribute__((ext_vector_type(4)));
constexpr int BJ_SUB_NT = 512;

template <int P>
__device__ __forceinline__ int bj_index(int blk_p, int blk_q, int k) { return (k < P / 2 ? blk_p : blk_q) * (P / 2) + (k & (P / 2 - 1)); }

template <int P>
__global__ __launch_bounds__(BJ_SUB_NT) void bj_sub_kernel(const double *__restrict__ A, int ne, int nb, int step, double *__restrict__ Qt_all,
                                                       int *__restrict__ flags, int *__restrict__ perm_all, int order, int inner_max) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int n = P, h = P / 2, ld = P + 1;
    constexpr bool QLDS = P <= 64;             // Qt beside S in LDS
    const int tid = threadIdx.x, nt = blockDim.x;
    int Pb, Qb;
    jac_pair(blockIdx.x, step, nb, Pb, Qb);
    lptr S = (lptr)smem;                       // n x ld
    lptr cs = S + (size_t)n * ld;              // (c, s) per pair
    lptr red = cs + 2 * h;                     // 64
    liptr pq = (liptr)(red + 64);              // (p, q) per pair
    double *Qg = Qt_all + (size_t)blockIdx.x * n * n;
    int *perm = perm_all + (size_t)blockIdx.x * n;
    double off2 = 0.0, dg2 = 0.0;
    for (int i = tid; i < n; i += nt) perm[i] = i;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e % n;
        const double v = A[(size_t)bj_index<P>(Pb, Qb, i) * ne + bj_index<P>(Pb, Qb, j)];
        S[i * ld + j] = v;
        Qg[e] = (i == j) ? 1.0 : 0.0;
        if (i == j) dg2 = fma(v, v, dg2); else off2 = fma(v, v, off2);
    }
    off2 = wg::reduce(off2, 0, red);
    dg2 = wg::reduce(dg2, 0, red);
    if (!(off2 > 1e-32 * dg2)) {               // nothing left to rotate in this pair (or an all-zero padding pair)
        if (tid == 0) flags[blockIdx.x] = 0;
        return;
    }
    if (tid == 0) flags[blockIdx.x] = 1;
    __syncthreads();
    lptr dsave = red + 64 + h;                 // (behind pq)
    lptr Ql = dsave + n;                       // P <= 64: Qt in LDS (n x ld)
    for (int i = tid; i < n; i += nt) dsave[i] = S[i * ld + i];
    if (QLDS)
        for (int e = tid; e < n * n; e += nt) Ql[(e / n) * ld + e % n] = (e / n == e % n) ? 1.0 : 0.0;
    __syncthreads();
    // inner sweeps (inner_max, default ONE).  Measured at n = 3000 (graded random Gramian): solving the sub-problem to the rounding
    // floor (up to 10 inner sweeps) does not buy outer sweeps -- 19 against 21 -- and costs 0.6 ms per inner sweep: 4.6 s against
    // 1.0 s.  What did buy sweeps is the ORDER in which the pair's eigenpairs are handed back (below): 28 / 26 -> 19-21.
    for (int isw = 0; isw < inner_max; ++isw) {
        for (int st = 0; st < n - 1; ++st) {
            for (int i = tid; i < h; i += nt) {
                int p, q;
                jac_pair(i, st, n, p, q);
                const double apq = S[p * ld + q], app = S[p * ld + p], aqq = S[q * ld + q];
                double c = 1.0, sn = 0.0;
                if (fabs(apq) > 1e-300 && fabs(apq) > 1e-30 * (fabs(app) + fabs(aqq))) {
                    const double th = (aqq - app) / (2.0 * apq);
                    const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0);
                    sn = t * c;
                }
                cs[2 * i] = c; cs[2 * i + 1] = sn;
                pq[2 * i] = p; pq[2 * i + 1] = q;
            }
            __syncthreads();
            // J' S J on the disjoint 2 x 2 blocks (pair i x pair j): rows with (c_i, s_i), then columns with (c_j, s_j), in one pass
            for (int e = tid; e < h * h; e += nt) {
                const int i = e / h, j = e % h;
                const int p = pq[2 * i], q = pq[2 * i + 1], pj = pq[2 * j], qj = pq[2 * j + 1];
                const double c = cs[2 * i], sn = cs[2 * i + 1], cj = cs[2 * j], sj = cs[2 * j + 1];
                const double a00 = S[p * ld + pj], a01 = S[p * ld + qj], a10 = S[q * ld + pj], a11 = S[q * ld + qj];
                const double b00 = c * a00 - sn * a10, b01 = c * a01 - sn * a11;
                const double b10 = sn * a00 + c * a10, b11 = sn * a01 + c * a11;
                S[p * ld + pj] = cj * b00 - sj * b01;
                S[p * ld + qj] = sj * b00 + cj * b01;
                S[q * ld + pj] = cj * b10 - sj * b11;
                S[q * ld + qj] = sj * b10 + cj * b11;
            }
            // rows of Qt (LDS, or L2: independent iterations, loads issued four pairs deep)
#pragma unroll 4
            for (int e = tid; e < h * n; e += nt) {
                const int i = e / n, k = e % n;
                const int p = pq[2 * i], q = pq[2 * i + 1];
                const double c = cs[2 * i], sn = cs[2 * i + 1];
                if (QLDS) {
                    const double vp = Ql[p * ld + k], vq = Ql[q * ld + k];
                    Ql[p * ld + k] = c * vp - sn * vq;
                    Ql[q * ld + k] = sn * vp + c * vq;
                } else {
                    const double vp = Qg[p * n + k], vq = Qg[q * n + k];
                    Qg[p * n + k] = c * vp - sn * vq;
                    Qg[q * n + k] = sn * vp + c * vq;
                }
            }
            __syncthreads();
        }
        double o2 = 0.0, d2 = 0.0;
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e % n;
            const double v = S[i * ld + j];
            if (i == j) d2 = fma(v, v, d2); else o2 = fma(v, v, o2);
        }
        o2 = wg::reduce(o2, 0, red);
        d2 = wg::reduce(d2, 0, red);
        if (o2 <= 1e-30 * d2) break;
    }
    if (QLDS) {
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) Qg[e] = Ql[(e / n) * ld + e % n];
    }
    // which eigenpair goes to which index of the pair.  order 1: position i, whose diagonal entry was the r-th largest before the
    // rotations, receives the r-th largest eigenvalue -- the block analogue of the scalar method's small angle (no exchange): Qt stays
    // as close to the identity as the spectrum allows, and what an annihilated block gets back from later steps is of second order.
    // order 2: descending over the whole pair (block P takes the larger half).  order 0: as the rotations left them.
    if (order != 0) {
        __syncthreads();
        for (int p = tid; p < n; p += nt) {
            const double lp = S[p * ld + p], dp = dsave[p];
            int rk = 0, rd = 0;
            for (int q = 0; q < n; ++q) {
                const double lq = S[q * ld + q], dq = dsave[q];
                rk += (lq > lp || (lq == lp && q < p)) ? 1 : 0;        // rank of eigenvalue p (descending)
                rd += (dq > dp || (dq == dp && q < p)) ? 1 : 0;        // rank of the old diagonal entry p
            }
            ((liptr)red)[p] = rk;
            ((liptr)red)[n + p] = rd;
        }
        __syncthreads();
        // perm[position] = row of Qt that holds its eigenvector
        for (int p = tid; p < n; p += nt) {
            const int rk = ((liptr)red)[p];
            if (order == 2) perm[rk] = p;
            else
                for (int pos = 0; pos < n; ++pos)
                    if (((liptr)red)[n + pos] == rk) perm[pos] = p;
        }
    }
}

// T[16 w + kk + 4 q][16 ct + l16] = sum_k Qt[16 w + l16'][k] X[k][16 ct + l16]: wave w's 16-row strip of Qt X for the tile X in LDS
// (row stride P + 16); the strip of Qt sits in registers (the MFMA A operand: lane (l16, kk) holds Qt[16 w + l16][4 s + kk], s < 32)
template <int P>
__device__ __forceinline__ void bj_strip_product(const double *__restrict__ Qt, const int *__restrict__ perm, lptr X, int wave, int l16, int kk,
                                                 bj_d4 (&acc)[P / 16]) {
    constexpr int LD = P + 16;
    double qa[P / 4];
    const int row = perm[16 * wave + l16];
#pragma unroll
    for (int s = 0; s < P / 4; ++s) qa[s] = Qt[(size_t)row * P + 4 * s + kk];
#pragma unroll
    for (int ct = 0; ct < P / 16; ++ct) acc[ct] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int s = 0; s < P / 4; ++s) {
#pragma unroll
        for (int ct = 0; ct < P / 16; ++ct)
            acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[s], X[(4 * s + kk) * LD + 16 * ct + l16], acc[ct], 0, 0, 0);
    }
}

// one tile (pair i, pair j), i <= j, of A <- Qt_i A Qt_j' and its mirror
template <int P>
__global__ __launch_bounds__(P * 4) void bj_update_a_kernel(double *__restrict__ A, int ne, int nb, int step, const double *__restrict__ Qt_all,
                                                            const int *__restrict__ flags, const int *__restrict__ perm_all) {
    const int pi = blockIdx.y, pj = blockIdx.x;
    if (pi > pj || (flags[pi] == 0 && flags[pj] == 0)) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LD = P + 16;
    lptr X = (lptr)smem;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15, kk = lane >> 4;
    int Pi, Qi, Pj, Qj;
    jac_pair(pi, step, nb, Pi, Qi);
    jac_pair(pj, step, nb, Pj, Qj);
    for (int e = tid; e < P * P; e += P * 4) {
        const int k = e / P, t = e % P;
        X[k * LD + t] = A[(size_t)bj_index<P>(Pi, Qi, k) * ne + bj_index<P>(Pj, Qj, t)];
    }
    __syncthreads();
    bj_d4 acc[P / 16];
    bj_strip_product<P>(Qt_all + (size_t)pi * P * P, perm_all + (size_t)pi * P, X, wave, l16, kk, acc);       // T1 = Qt_i X
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < P / 16; ++ct)                                                     // Y = T1' over X
#pragma unroll
        for (int q = 0; q < 4; ++q) X[(16 * ct + l16) * LD + 16 * wave + kk + 4 * q] = acc[ct][q];
    __syncthreads();
    bj_strip_product<P>(Qt_all + (size_t)pj * P * P, perm_all + (size_t)pj * P, X, wave, l16, kk, acc);       // out' = Qt_j T1'
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < P / 16; ++rt)                                                     // out[r][c], r = 16 rt + l16, c = 16 w + kk + 4 q
#pragma unroll
        for (int q = 0; q < 4; ++q) X[(16 * rt + l16) * LD + 16 * wave + kk + 4 * q] = acc[rt][q];
    __syncthreads();
    if (pi == pj) {
        for (int e = tid; e < P * P; e += P * 4) {
            const int r = e / P, c = e % P;
            A[(size_t)bj_index<P>(Pi, Qi, r) * ne + bj_index<P>(Pi, Qi, c)] = 0.5 * (X[r * LD + c] + X[c * LD + r]);
        }
        return;
    }
    for (int e = tid; e < P * P; e += P * 4) {
        const int r = e / P, c = e % P;
        A[(size_t)bj_index<P>(Pi, Qi, r) * ne + bj_index<P>(Pj, Qj, c)] = X[r * LD + c];
    }
    for (int e = tid; e < P * P; e += P * 4) {                                   // the mirror tile (rows of pair j)
        const int c = e / P, r = e % P;
        A[(size_t)bj_index<P>(Pj, Qj, c) * ne + bj_index<P>(Pi, Qi, r)] = X[r * LD + c];
    }
}

// Vt[rows of pair i, 128 columns] <- Qt_i Vt[...]
template <int P>
__global__ __launch_bounds__(P * 4) void bj_update_v_kernel(double *__restrict__ Vt, int ne, int nb, int step, const double *__restrict__ Qt_all,
                                                            const int *__restrict__ flags, const int *__restrict__ perm_all) {
    const int pi = blockIdx.y, c0 = blockIdx.x * P;
    if (flags[pi] == 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LD = P + 16;
    lptr X = (lptr)smem;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15, kk = lane >> 4;
    int Pi, Qi;
    jac_pair(pi, step, nb, Pi, Qi);
    for (int e = tid; e < P * P; e += P * 4) {
        const int k = e / P, t = e % P;
        X[k * LD + t] = Vt[(size_t)bj_index<P>(Pi, Qi, k) * ne + c0 + t];
    }
    __syncthreads();
    bj_d4 acc[P / 16];
    bj_strip_product<P>(Qt_all + (size_t)pi * P * P, perm_all + (size_t)pi * P, X, wave, l16, kk, acc);
#pragma unroll
    for (int ct = 0; ct < P / 16; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) Vt[(size_t)bj_index<P>(Pi, Qi, 16 * wave + kk + 4 * q) * ne + c0 + 16 * ct + l16] = acc[ct][q];
}

// The ordering of the pair problems moves eigenpairs between positions, the padding's (eigenvalue 0, a unit vector in a padding
// column -- the padding never mixes with the rest) included: a position belongs to the padding iff its row of Vt is non-zero there.
__global__ void bj_pad_kernel(const double *__restrict__ Vt, int n, int ne, int *__restrict__ rank) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ne) return;
    int pad = 0;
    for (int k = n; k < ne; ++k) pad |= (Vt[(size_t)p * ne + k] != 0.0) ? 1 : 0;
    rank[p] = pad ? -1 : 0;
}

// rank[] on entry: -1 at the padding's positions, >= 0 elsewhere (and it stays that way while the ranks are written)
__global__ void bj_rank_kernel(const double *__restrict__ A, int ne, double *__restrict__ w, int *__restrict__ rank) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ne || rank[p] < 0) return;
    const double lp = A[(size_t)p * ne + p];
    int rk = 0;
    for (int q = 0; q < ne; ++q) {
        const double lq = A[(size_t)q * ne + q];
        if ((lq < lp || (lq == lp && q < p)) && rank[q] >= 0) ++rk;
    }
    w[rk] = lp;
    rank[p] = rk;
}

__global__ void bj_gather_kernel(const double *__restrict__ Vt, const int *__restrict__ rank, int n, int ne, double *__restrict__ G) {
    const int p = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && rank[p] >= 0) G[(size_t)rank[p] * n + k] = Vt[(size_t)p * ne + k];
}

template <int P>
int jacobi_block_p(double *G_dev, int n, double *w_dev, hipStream_t st) {
    const int ne = (n + P - 1) / P * P, nb = ne / (P / 2), np = nb / 2;
    srh::DevBuf A, Vt, Qt, flags, perm, part, rank;
    const int order = getenv("SRH_EIGH_BLOCK_ORDER") ? atoi(getenv("SRH_EIGH_BLOCK_ORDER")) : 2;
    const int inner_max = getenv("SRH_EIGH_BLOCK_INNER") ? atoi(getenv("SRH_EIGH_BLOCK_INNER")) : 1;
    int rc;
    if ((rc = A.alloc(sizeof(double) * (size_t)ne * ne)) || (rc = Vt.alloc(sizeof(double) * (size_t)ne * ne)) ||
        (rc = Qt.alloc(sizeof(double) * (size_t)np * P * P)) || (rc = flags.alloc(sizeof(int) * np)) || (rc = perm.alloc(sizeof(int) * (size_t)np * P)) ||
        (rc = part.alloc(sizeof(double) * 2 * ne)) || (rc = rank.alloc(sizeof(int) * ne)))
        return rc;
    const size_t lds_sub = srh::lds_request(sizeof(double) * ((size_t)P * (P + 1) * (P <= 64 ? 2 : 1) + 2 * P + 64 + P) + sizeof(int) * P + 64);
    const size_t lds_upd = srh::lds_request(sizeof(double) * (size_t)P * (P + 16));
    SRH_CHECK_HIP(hipFuncSetAttribute((const void *)bj_sub_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sub));
    SRH_CHECK_HIP(hipFuncSetAttribute((const void *)bj_update_a_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_upd));
    SRH_CHECK_HIP(hipFuncSetAttribute((const void *)bj_update_v_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_upd));
    jac_init_kernel<<<(unsigned)srh::cdiv((int64_t)ne * ne, 256), 256, 0, st>>>(G_dev, n, ne, A.as<double>(), Vt.as<double>());
    std::vector<double> hp(2 * (size_t)ne);
    double prev = INFINITY;
    bool done = false;
    for (int sweep = 0; sweep < 100 && !done; ++sweep) {      // (rank-deficient Gramians of several thousand snapshots: 40+ sweeps of the 64-pairs)
        for (int step = 0; step < nb - 1; ++step) {
            bj_sub_kernel<P><<<(unsigned)np, BJ_SUB_NT, lds_sub, st>>>(A.as<double>(), ne, nb, step, Qt.as<double>(), flags.as<int>(), perm.as<int>(), order, inner_max);
            bj_update_a_kernel<P><<<dim3((unsigned)np, (unsigned)np), P * 4, lds_upd, st>>>(A.as<double>(), ne, nb, step, Qt.as<double>(), flags.as<int>(), perm.as<int>());
            bj_update_v_kernel<P><<<dim3((unsigned)(ne / P), (unsigned)np), P * 4, lds_upd, st>>>(Vt.as<double>(), ne, nb, step, Qt.as<double>(), flags.as<int>(), perm.as<int>());
        }
        jac_norms_kernel<<<(unsigned)ne, 256, 0, st>>>(A.as<double>(), ne, part.as<double>());
        SRH_CHECK_HIP(hipGetLastError());
        SRH_CHECK_HIP(hipMemcpyAsync(hp.data(), part.p, sizeof(double) * 2 * ne, hipMemcpyDeviceToHost, st));
        SRH_CHECK_HIP(hipStreamSynchronize(st));
        double off2 = 0.0, diag2 = 0.0;
        for (int i = 0; i < ne; ++i) { off2 += hp[2 * i]; diag2 += hp[2 * i + 1]; }
        if (getenv("SRH_EIGH_TRACE")) fprintf(stderr, "block Jacobi n %d sweep %d: off^2 / diag^2 = %.3e\n", n, sweep, off2 / diag2);
        done = off2 <= 1e-30 * diag2 || (off2 <= 1e-26 * diag2 && off2 > 0.25 * prev);
        prev = off2;
    }
    if (!done) {
        srh::set_error("srom_eigh_dev: block Jacobi sweeps did not converge");
        return SRH_ENUMERIC;
    }
    bj_pad_kernel<<<(unsigned)srh::cdiv(ne, 128), 128, 0, st>>>(Vt.as<double>(), n, ne, rank.as<int>());
    bj_rank_kernel<<<(unsigned)srh::cdiv(ne, 128), 128, 0, st>>>(A.as<double>(), ne, w_dev, rank.as<int>());
    bj_gather_kernel<<<dim3((unsigned)srh::cdiv(n, 256), (unsigned)ne), 256, 0, st>>>(Vt.as<double>(), rank.as<int>(), n, ne, G_dev);
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(st));
    return SRH_OK;
}


// pair size: 64 up to SRH_EIGH_BLOCK_SWITCH (default 8192) snapshots, 128 above (SRH_EIGH_BLOCK_PAIR=64 / 128 forces one).  Measured
// 64 / 128: 0.11 / 0.28 s at 1200, 0.28 / 0.54 s at 2048, 0.53 / 1.01 s at 3000, 2.2 / 3.5 s at 5000, 17.9 / 18.2 s at 10 000.
int jacobi_block(double *G_dev, int n, double *w_dev, hipStream_t st) {
    const char *force = getenv("SRH_EIGH_BLOCK_PAIR");
    const int sw = getenv("SRH_EIGH_BLOCK_SWITCH") ? atoi(getenv("SRH_EIGH_BLOCK_SWITCH")) : 8192;
    const int P = force ? atoi(force) : (n <= sw ? 64 : 128);
    return P == 64 ? jacobi_block_p<64>(G_dev, n, w_dev, st) : jacobi_block_p<128>(G_dev, n, w_dev, st);
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
    // SRH_EIGH_BLOCK=1: the library-free block Jacobi for every n > 128 (tests, boxes without rocSOLVER)
    const bool want_block = getenv("SRH_EIGH_BLOCK") && atoi(getenv("SRH_EIGH_BLOCK")) != 0 && !getenv("SRH_EIGH_ROCSOLVER");
    if (want_block && n > JAC_MAX) return jacobi_block(G_dev, (int)n, w_dev, (hipStream_t)stream);
    // 256 < n <= 2048: the block form (pairs of 64) against the scalar one: 17 / 38 ms at 300, 42 / 105 ms at 600, 0.11 / 0.36 s at 1200, 0.28 / 1.37 s at 2048
    if (n > JAC_BLOCK_MIN && n <= JAC_BLOCK_MAX && !getenv("SRH_EIGH_ROCSOLVER") && !(getenv("SRH_EIGH_BLOCK") && atoi(getenv("SRH_EIGH_BLOCK")) == 0))
        return jacobi_block(G_dev, (int)n, w_dev, (hipStream_t)stream);
    if (n <= JAC_GRID_MAX && !getenv("SRH_EIGH_ROCSOLVER")) return jacobi_grid(G_dev, (int)n, w_dev, (hipStream_t)stream);
    // above 4096 (SRH_EIGH_BLOCK=0: above 2048): rocSOLVER's dsyevd where it loads (tridiagonalisation + divide and conquer: ~30 x fewer flops than any Jacobi
    // method -- 0.22 s against 3.0 s at n = 5000), the block Jacobi where it does not: the library is an accelerator, not a dependency
    Solver &s = solver();
    if (!s.ok) {
        if (getenv("SRH_EIGH_ROCSOLVER")) {
            srh::set_error("srom_eigh_dev: rocSOLVER / rocBLAS could not be loaded (%s)", dlerror() ? dlerror() : "symbol missing");
            return SRH_EHIP;
        }
        return jacobi_block(G_dev, (int)n, w_dev, (hipStream_t)stream);
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
