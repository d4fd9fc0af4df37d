// Discrete extended Kalman filter over the TPWL model, state and covariance resident in HBM.
// Reference: sofacontrol/tpwl/observer.py:33-126 (DiscreteEKFObserver): predict_state 97-106,
// update_state 108-126.  One workgroup per filter step; every matrix of the step lives in LDS.
#include "tpwl_host.h"

// pod.hip: two-phase staged projection (enqueue on a stream; read the pinned mirror once that stream has drained)
int srom_stage_project(srom *h, int which, const double *X, int64_t B, hipStream_t stream);
int srom_stage_collect(srom *h, double *out, int64_t B, int which);

struct sekf {
    stpwl *model = nullptr;
    int n = 0, m = 0, ny = 0;
    srh::DevBuf C, y_ref, W, V, x, Sigma, scratch, ext;
    size_t lds = 0;
    bool mfma = false, wide = false;      // wide: ekf_wide_kernel (64 < n_x <= 80)
    // pinned host mirrors of the per-step input (u, y) and output (x, status): one copy each way per step
    double *pin_in = nullptr, *pin_out = nullptr;
    hipStream_t side = nullptr;          // sekf_step_projected: the projection runs beside the filter kernel

    hipEvent_t side_gate = nullptr;    // orders the side stream behind earlier work of stream 0 on the same rom
    ~sekf() {
        if (side) (void)hipStreamDestroy(side);
        if (side_gate) (void)hipEventDestroy(side_gate);
        if (pin_in) (void)hipHostFree(pin_in);
        if (pin_out) (void)hipHostFree(pin_out);
    }
};

namespace {

struct EkfArgs {
    TpwlDev T;
    int n, m, ny, ld, ldy;
    const double *C, *y_ref, *W, *V;
    double *x, *Sigma;
    const double *u, *y;            // inputs of the step (device)
    const double *Aext, *Bext, *dext;   // explicit (A_d, B_d, d_d) instead of the nearest-point tables
    int do_predict, do_update;
    int *status;
};

constexpr int EKF_NT = 512;

// sum_k a[k * sa] * b[k * sb]: eight operand pairs in flight per trip (with two waves per SIMD a rolled
// load -> fma chain pays the full LDS latency for every k)
__device__ __forceinline__ double dotk(clptr a, int sa, clptr b, int sb, int K) {
    double acc = 0.0;
    int k = 0;
    for (; k + 8 <= K; k += 8) {
        double av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { av[q] = a[(k + q) * sa]; bv[q] = b[(k + q) * sb]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fma(av[q], bv[q], acc);
    }
    if (k < K) {                 // remainder as one predicated batch (a rolled tail pays the LDS latency per element)
        double av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool in = k + q < K;
            av[q] = in ? a[(k + q) * sa] : 0.0;
            bv[q] = in ? b[(k + q) * sb] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fma(av[q], bv[q], acc);
    }
    return acc;
}

__global__ __launch_bounds__(EKF_NT) void ekf_kernel(EkfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = a.n, m = a.m, ny = a.ny, ld = a.ld, ldy = a.ldy;
    const int tid = threadIdx.x, nt = blockDim.x;
    lptr Sg = (lptr)smem;                 // Sigma            (n x ld)
    lptr Am = Sg + (size_t)n * ld;        // A_d, later M1 = Sigma^- C^T then K   (n x ld)
    lptr Tm = Am + (size_t)n * ld;        // A Sigma, later C Sigma^-            (n x ld) / (ny x ld)
    lptr Cm = Tm + (size_t)n * ld;        // C                (ny x ld)
    lptr Sm = Cm + (size_t)ny * ld;       // S and its Cholesky factor (ny x ldy)
    const int nv = n > ny ? n : ny;
    lptr xv = Sm + (size_t)ny * ldy;      // state
    lptr xn = xv + nv;                    // predicted state
    lptr iv = xn + nv;                    // innovation
    lptr uv = iv + nv;                    // input
    liptr ip = (liptr)(uv + nv);

    for (int e = tid; e < n * n; e += nt) Sg[(e / n) * ld + e % n] = a.Sigma[e];
    for (int e = tid; e < n; e += nt) xv[e] = a.x[e];
    if (a.do_predict)
        for (int e = tid; e < m; e += nt) uv[e] = a.u[e];
    if (tid == 0) ip[1] = 0;
    __syncthreads();

    if (a.do_predict) {
        // ---- x^- = A x + B u + d, Sigma^- = A Sigma A^T + W                     (observer.py:104-106)
        const double *Ag, *Bg, *dg;
        if (a.Aext != nullptr) {
            Ag = a.Aext; Bg = a.Bext; dg = a.dext;
        } else {
            if (tid < 64) {
                const int i = tpwl::nearest_wave(a.T, xv);
                if (tid == 0) ip[0] = i;
            }
            __syncthreads();
            const int i = ip[0];
            Ag = (const double *)a.T.Ad + (size_t)i * n * n;
            Bg = (const double *)a.T.Bd + (size_t)i * n * m;
            dg = (const double *)a.T.dd + (size_t)i * n;
        }
        for (int e = tid; e < n * n; e += nt) Am[(e / n) * ld + e % n] = Ag[e];
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s = fma(Am[i * ld + k], xv[k], s);
            double t = 0.0;
            for (int k = 0; k < m; ++k) t = fma(Bg[i * m + k], uv[k], t);
            xn[i] = s + t + dg[i];
        }
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e % n;
            Tm[i * ld + j] = dotk(Am + i * ld, 1, Sg + j, ld, n);
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e % n;
            Sg[i * ld + j] = dotk(Tm + i * ld, 1, Am + j * ld, 1, n) + a.W[e];
        }
        for (int e = tid; e < n; e += nt) xv[e] = xn[e];
        __syncthreads();
    }

    if (a.do_update) {
        // ---- S = C Sigma C^T + V, K = Sigma C^T S^-1, x += K (y - y_ref - C x), Sigma = (I - K C) Sigma
        for (int e = tid; e < ny * n; e += nt) Cm[(e / n) * ld + e % n] = a.C[e];
        __syncthreads();
        for (int e = tid; e < n * ny; e += nt) {       // M1 = Sigma C^T  (n x ny)
            const int i = e / ny, j = e % ny;
            Am[i * ld + j] = dotk(Sg + i * ld, 1, Cm + j * ld, 1, n);
        }
        for (int e = tid; e < ny * n; e += nt) {       // CS = C Sigma    (ny x n)
            const int i = e / n, j = e % n;
            Tm[i * ld + j] = dotk(Cm + i * ld, 1, Sg + j, ld, n);
        }
        for (int i = tid; i < ny; i += nt) {           // innovation
            double s = 0.0;
            for (int k = 0; k < n; ++k) s = fma(Cm[i * ld + k], xv[k], s);
            iv[i] = a.y[i] - (a.y_ref ? a.y_ref[i] : 0.0) - s;
        }
        __syncthreads();
        for (int e = tid; e < ny * ny; e += nt) {      // S = C M1 + V
            const int i = e / ny, j = e % ny;
            Sm[i * ldy + j] = dotk(Cm + i * ld, 1, Am + j, ld, n) + a.V[e];
        }
        __syncthreads();
        // Cholesky of S (lower part), right-looking, by ONE wave: LDS operations of a wave execute in program
        // order, so the ny column steps need no workgroup barrier (90 barriers at n_y = 30 otherwise)
        if (tid < 64) {
            for (int j = 0; j < ny; ++j) {
                const double dj = Sm[j * ldy + j];
                if (!(dj > 0.0)) { if (tid == 0) ip[1] = 1; break; }       // uniform within the wave
                const double rj = sqrt(dj);
                for (int i = j + 1 + tid; i < ny; i += 64) Sm[i * ldy + j] = Sm[i * ldy + j] / rj;
                if (tid == 0) Sm[j * ldy + j] = rj;
                __builtin_amdgcn_wave_barrier();
                const int w = ny - j - 1;
                for (int e = tid; e < w * w; e += 64) {
                    const int i = j + 1 + e / w, k = j + 1 + e % w;
                    if (k <= i) Sm[i * ldy + k] = fma(-Sm[i * ldy + j], Sm[k * ldy + j], Sm[i * ldy + k]);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        if (ip[1] != 0) {
            if (tid == 0) *a.status = 1;
            return;
        }
        // K row i: solve k S = M1_i  <=>  S k^T = M1_i^T  (S = L L^T), in place; the inner products run with
        // several operand pairs in flight (dotk), only the ny substitution steps are sequential
        for (int i = tid; i < n; i += nt) {
            lptr row = Am + (size_t)i * ld;
            for (int c = 0; c < ny; ++c) row[c] = (row[c] - dotk(Sm + c * ldy, 1, row, 1, c)) / Sm[c * ldy + c];
            for (int c = ny - 1; c >= 0; --c)
                row[c] = (row[c] - dotk(Sm + (c + 1) * ldy + c, ldy, row + c + 1, 1, ny - 1 - c)) / Sm[c * ldy + c];
        }
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            double s = 0.0;
            for (int k = 0; k < ny; ++k) s = fma(Am[i * ld + k], iv[k], s);
            xn[i] = xv[i] + s;
        }
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e % n;
            a.Sigma[e] = Sg[i * ld + j] - dotk(Am + i * ld, 1, Tm + j, ld, ny);
        }
        __syncthreads();
        for (int e = tid; e < n; e += nt) a.x[e] = xn[e];
    } else {
        for (int e = tid; e < n * n; e += nt) a.Sigma[e] = Sg[(e / n) * ld + e % n];
        for (int e = tid; e < n; e += nt) a.x[e] = xv[e];
    }
    if (tid == 0) *a.status = 0;
}

// ---- the same filter step on f64 MFMA products (used when the padded panels fit LDS: n_x <= 64)
// All operands are kept k-major so that every product is C = Lm^T Rm (wg::mfma_atb):
//   U = Sigma A^T (Lm = Sigma, symmetric; Rm = A^T from the transposed table), Sigma^- = A U (Lm = A^T, Rm = U),
//   M1 = Sigma^- C^T (Lm = Sigma^-, Rm = C^T), CS = C Sigma^- (Lm = C^T, Rm = Sigma^-), S = C M1 (Lm = C^T, Rm = M1),
//   Sigma = Sigma^- - K CS with K^T = S^-1 CS (Lm = K^T, Rm = CS).
// S^-1 through the Cholesky factor and its explicit triangular inverse, both by one wave (n_y <= 64).
struct EkfMfmaDims {
    int n16, ny16, ld, ldy, NK, NKy;
};

__host__ __device__ inline EkfMfmaDims ekf_mfma_dims(int n, int ny) {
    EkfMfmaDims d;
    d.n16 = (n + 15) & ~15; d.ny16 = (ny + 15) & ~15; d.ld = d.n16 + 1; d.ldy = d.ny16 + 1;
    d.NK = (n + 3) & ~3; d.NKy = (ny + 3) & ~3;
    return d;
}

__host__ __device__ inline size_t ekf_mfma_doubles(int n, int ny) {
    const EkfMfmaDims d = ekf_mfma_dims(n, ny);
    const size_t nv = (size_t)(d.n16 > d.ny16 ? d.n16 : d.ny16);
    return 3 * (size_t)d.n16 * d.ld + (size_t)d.ny16 * d.ld + 3 * (size_t)d.ny16 * d.ldy + 4 * nv + 8;
}

__device__ __forceinline__ double ekf_readlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1 / p to working precision without the ~30-instruction IEEE division sequence (it sits on the critical path of
// every elimination step): v_rcp_f64 and two Newton steps.  p is a pivot of a positive definite matrix: normal range.
__device__ __forceinline__ double ekf_rcp(double p) {
    double r = __builtin_amdgcn_rcp(p);
    r = fma(fma(-p, r, 1.0), r, r);
    r = fma(fma(-p, r, 1.0), r, r);
    return r;
}

// K^T = S^-1 CS by Gauss-Jordan elimination of the tableau [S | CS] -> [I | S^-1 CS] on the whole workgroup.
// Wave w owns rows w, w + nw, ... (RW of them), lane = column (NC chunks of 64), every entry in a register with a
// compile-time index; the row index of an entry is wave-uniform, so "is this the pivot row" is a scalar branch and
// the multiplier one broadcast read.  Per pivot step the pivot row and the pivot column travel through LDS (double
// buffered: one barrier per step); their owners publish the row / column of the NEXT step right after updating them.
// No pivoting: S is symmetric positive definite, its elimination pivots are the squared diagonal of the Cholesky
// factor, so "pivot <= 0" reports exactly what a failed factorisation would.
// Measured alternatives at n_y = 30, n = 60 (tools/probes/ekf_prof.py): one wave, Cholesky + triangular inverse with
// LDS operands 80 k clocks (+ 12 k for the two triangular products); one wave, factor in registers with v_readlane
// operands 45 k; the same with broadcast LDS reads 42 k; this elimination with one thread per entry 60 k (the index
// arithmetic and the IEEE division dominate).
// buf: 2 x (n_y + n + RW nw) doubles inside a cleared panel of 128 doubles more; *bad cleared by the caller.
// Returns false when S is not positive definite (uniform over the workgroup).
template <int RW, int NC>
__device__ __forceinline__ bool ekf_gain_gj(clptr Sm, int ldy, int ny, clptr CS, int ld, int n, lptr KT, lptr buf, liptr bad) {
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = ny + n, stride = W + RW * nw;
    double av[RW][NC];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int i = wave + r * nw;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const int c = lane + 64 * ch;
            av[r][ch] = (i < ny && c < W) ? (c < ny ? Sm[i * ldy + c] : CS[i * ld + c - ny]) : 0.0;
        }
    }
    // The owner of row jn divides it by its pivot BEFORE publishing it (so only one wave pays the reciprocal and the
    // others eliminate with a -= a[i][jn] * row'); every wave publishes its entries of column jn, with a zero in
    // place of the pivot row's own entry (that row is final: its multiplier is zero).  Rows >= n_y of a wave write
    // into the padding behind the column (stride counts RW * nw rows) -- zeros, since their entries stay zero.
    auto publish = [&](int jn, lptr nr, lptr nc) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            if (wave + r * nw == jn) {                                       // wave-uniform
                const double piv = ekf_readlane(av[r][0], jn);               // column jn < n_y <= 64: chunk 0
                if (!(piv > 0.0) && lane == 0) *bad = 1;
                const double rp = ekf_rcp(piv);
#pragma unroll
                for (int ch = 0; ch < NC; ++ch) {
                    av[r][ch] *= rp;
                    if (lane + 64 * ch < W) nr[lane + 64 * ch] = av[r][ch];
                }
            }
        }
        if (lane == jn) {
#pragma unroll
            for (int r = 0; r < RW; ++r) nc[wave + r * nw] = (wave + r * nw == jn) ? 0.0 : av[r][0];
        }
    };
    publish(0, buf, buf + W);
    __syncthreads();
    for (int j = 0; j < ny; ++j) {
        clptr pr = buf + (j & 1) * stride, pc = pr + W;
        lptr nr = buf + ((j + 1) & 1) * stride;
        double prc[NC], mv[RW];
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) prc[ch] = pr[lane + 64 * ch];        // entries past W: never stored
#pragma unroll
        for (int r = 0; r < RW; ++r) mv[r] = pc[wave + r * nw];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int ch = 0; ch < NC; ++ch) av[r][ch] = fma(-mv[r], prc[ch], av[r][ch]);
        if (j + 1 < ny) publish(j + 1, nr, nr + W);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int i = wave + r * nw;
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const int c = lane + 64 * ch;
            if (i < ny && c >= ny && c < W) KT[i * ld + c - ny] = av[r][ch];
        }
    }
    return *bad == 0;
}

#ifdef SRH_PROFILE
#define EKF_LAP(i) do { __syncthreads(); const long long now_ = clock64(); ekp[i] += now_ - ekl; ekl = now_; } while (0)
#else
#define EKF_LAP(i) ((void)0)
#endif

template <int NSEL>      // n_x fixed at compile time (the Diamond models at r = 30 / 36), or 0: any size
__global__ __launch_bounds__(EKF_NT) void ekf_mfma_kernel(EkfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = NSEL > 0 ? NSEL : a.n, m = a.m, ny = a.ny;
    const EkfMfmaDims D = ekf_mfma_dims(n, ny);
    const int n16 = D.n16, ny16 = D.ny16, ld = D.ld, ldy = D.ldy, NK = D.NK, NKy = D.NKy;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
    lptr SG = (lptr)smem;                       // Sigma, then Sigma^-                         (n16 x ld)
    lptr AT = SG + (size_t)n16 * ld;            // A^T, then C^T, then K CS                    (n16 x ld)
    lptr UU = AT + (size_t)n16 * ld;            // U = Sigma A^T, then M1, then Y | K^T        (n16 x ld)
    lptr CS = UU + (size_t)n16 * ld;            // C Sigma^-                                   (ny16 x ld)
    lptr Sm = CS + (size_t)ny16 * ld;           // S, then its Cholesky factor L (lower)       (ny16 x ldy)
    lptr Li = Sm + (size_t)ny16 * ldy;          // L^-1 (lower)
    lptr LiT = Li + (size_t)ny16 * ldy;         // L^-T
    const int nv = n16 > ny16 ? n16 : ny16;
    lptr xv = LiT + (size_t)ny16 * ldy, xn = xv + nv, iv = xn + nv, uv = iv + nv;
    liptr ip = (liptr)(uv + nv);

#ifdef SRH_PROFILE
    long long ekp[16] = {0}, ekl = clock64();
#endif
    // Every global operand that does not depend on the nearest-point index is requested now, into registers, so that
    // the HBM / L2 latency (~2-4 k clocks each when exposed) is paid once: W, C, V, y - y_ref.  n <= 64, nt = 512:
    // at most 8 entries per thread each.
    constexpr int PQ = 8;
    double wreg[PQ], creg[PQ], vreg[PQ], yreg = 0.0;
    if (a.do_predict) {
#pragma unroll
        for (int k = 0; k < PQ; ++k) wreg[k] = tid + k * nt < n * n ? a.W[tid + k * nt] : 0.0;
    }
    if (a.do_update) {
#pragma unroll
        for (int k = 0; k < PQ; ++k) creg[k] = tid + k * nt < ny * n ? a.C[tid + k * nt] : 0.0;
#pragma unroll
        for (int k = 0; k < PQ; ++k) vreg[k] = tid + k * nt < ny * ny ? a.V[tid + k * nt] : 0.0;
        if (tid < ny) yreg = a.y[tid] - (a.y_ref ? a.y_ref[tid] : 0.0);
    }
    // wave 0: state, input and the nearest-point search; the other waves: Sigma into its zero-padded panel and the
    // clearing of every other panel (the two never touch the same LDS words, so one barrier ends both)
    const bool table = a.do_predict && a.Aext == nullptr;
    if (tid < 64) {
        for (int e = tid; e < n; e += 64) xv[e] = a.x[e];
        if (a.do_predict)
            for (int e = tid; e < m; e += 64) uv[e] = a.u[e];
        if (tid == 0) ip[1] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (table) {
            const int i = tpwl::nearest_wave(a.T, xv);
            if (tid == 0) ip[0] = i;
        }
    } else {
        const int t2 = tid - 64, nt2 = nt - 64;
        for (int e = t2; e < n16 * ld; e += nt2) {
            const int i = e / ld, j = e - i * ld;
            SG[e] = (i < n && j < n) ? a.Sigma[i * n + j] : 0.0;
        }
        for (int e = t2; e < 2 * n16 * ld + ny16 * ld + 3 * ny16 * ldy; e += nt2) AT[e] = 0.0;
    }
    __syncthreads();
    EKF_LAP(0);

    if (a.do_predict) {
        double tb = 0.0, dgv = 0.0;                 // row tid of B u and of d (n <= nt: one row per thread)
        if (a.Aext != nullptr) {
            if (tid < n) {
                for (int k = 0; k < m; ++k) tb = fma(a.Bext[tid * m + k], uv[k], tb);
                dgv = a.dext[tid];
            }
            for (int e = tid; e < n * n; e += nt) AT[(e % n) * ld + e / n] = a.Aext[e];
        } else {
            const size_t i = (size_t)ip[0];
            cgptr At = a.T.AdT + i * n * n;          // transposed table: At[k * n + r] = A[r][k]
            const double *Bg = (const double *)a.T.Bd + i * n * m;
            if (tid < n) {
                for (int k = 0; k < m; ++k) tb = fma(Bg[tid * m + k], uv[k], tb);
                dgv = ((const double *)a.T.dd + i * n)[tid];
            }
            for (int e = tid; e < n * n; e += nt) AT[(e / n) * ld + e % n] = At[e];
        }
        __syncthreads();
        EKF_LAP(2);
        if (tid < n) xn[tid] = dotk(AT + tid, ld, xv, 1, n) + tb + dgv;
        EKF_LAP(3);
        wg::mfma_atb(UU, ld, SG, AT, NK, n16 >> 4, n16 >> 4, ld, n);            // U = Sigma A^T
        wg::mfma_atb(SG, ld, AT, UU, NK, n16 >> 4, n16 >> 4, ld, n);            // Sigma^- = A U
        EKF_LAP(4);
#pragma unroll
        for (int k = 0; k < PQ; ++k) {
            const int e = tid + k * nt;
            if (e < n * n) SG[(e / n) * ld + e % n] += wreg[k];
        }
        for (int e = tid; e < n; e += nt) xv[e] = xn[e];
        __syncthreads();
    }

    EKF_LAP(5);
    if (a.do_update) {
        for (int e = tid; e < n16 * ld; e += nt) AT[e] = 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PQ; ++k) {
            const int e = tid + k * nt;
            if (e < ny * n) AT[(e % n) * ld + e / n] = creg[k];                          // C^T
        }
        __syncthreads();
        if (tid < ny) iv[tid] = yreg - dotk(AT + tid, ld, xv, 1, n);
        EKF_LAP(6);
        wg::mfma_atb(UU, ld, SG, AT, NK, n16 >> 4, ny16 >> 4, ld, n);             // M1 = Sigma^- C^T   (n x ny)
        wg::mfma_atb(Sm, ldy, AT, UU, NK, ny16 >> 4, ny16 >> 4, ld, ny);          // C M1               (ny x ny)
        // CS = C Sigma^- = M1^T: Sigma^- = A (Sigma A^T) + W is symmetric up to the rounding of the two products
        for (int e = tid; e < ny * n; e += nt) CS[(e / n) * ld + e % n] = UU[(e % n) * ld + e / n];
#pragma unroll
        for (int k = 0; k < PQ; ++k) {
            const int e = tid + k * nt;
            if (e < ny * ny) Sm[(e / ny) * ldy + e % ny] += vreg[k];
        }
        __syncthreads();
        EKF_LAP(7);
        lptr Y = UU, KT = UU + (size_t)ny16 * ld;
        if (ny <= 4 * (nt >> 6) && ny <= 64 && ny + n <= 128 && 2 * (ny + n + 4 * (nt >> 6)) + 128 <= 2 * ny16 * ldy) {
            for (int e = tid; e < ny16 * ld; e += nt) KT[e] = 0.0;          // M1 is dead: S and CS are built
            __syncthreads();
            (void)ekf_gain_gj<4, 2>(Sm, ldy, ny, CS, ld, n, KT, Li, ip + 1);     // Li, LiT: one cleared 2-panel buffer
            __syncthreads();
            EKF_LAP(8);
            if (ip[1] != 0) {
                if (tid == 0) *a.status = 1;
                return;
            }
        } else {
            if (tid < 64) {
                // left-looking Cholesky, lane = row: column j of L from the finished columns < j (no trailing update)
                bool ok = true;
                for (int j = 0; j < ny; ++j) {
                    double sj = 0.0;
                    if (lane >= j && lane < ny) sj = Sm[lane * ldy + j] - dotk(Sm + lane * ldy, 1, Sm + j * ldy, 1, j);
                    const double djj = __shfl(sj, j, 64);
                    if (!(djj > 0.0)) { ok = false; break; }                              // uniform
                    const double rj = sqrt(djj);
                    if (lane >= j && lane < ny) Sm[lane * ldy + j] = (lane == j) ? rj : sj / rj;
                    __builtin_amdgcn_wave_barrier();
                }
                if (!ok && tid == 0) ip[1] = 1;
                if (ok) {
                    // L^-1, lane = column c: row i from rows < i (entries above the diagonal stay zero)
                    for (int i = 0; i < ny; ++i) {
                        if (lane <= i && lane < ny) {
                            const double sdot = dotk(Sm + i * ldy, 1, Li + lane, ldy, i);
                            const double v = ((lane == i) ? 1.0 : -sdot) / Sm[i * ldy + i];
                            Li[i * ldy + lane] = v;
                            LiT[lane * ldy + i] = v;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            __syncthreads();
            EKF_LAP(8);
            if (ip[1] != 0) {
                if (tid == 0) *a.status = 1;
                return;
            }
            // Y = L^-1 CS (rows 0.. of UU), K^T = L^-T Y (rows ny16.. of UU)
            for (int e = tid; e < ny * n; e += nt) {
                const int i = e / n, j = e % n;
                Y[i * ld + j] = dotk(Li + i * ldy, 1, CS + j, ld, i + 1);
            }
            for (int e = tid; e < (ny16 - ny) * ld; e += nt) Y[ny * ld + e] = 0.0;
            __syncthreads();
            for (int e = tid; e < ny16 * ld; e += nt) {
                const int i = e / ld, j = e % ld;
                double v = 0.0;
                if (i < ny && j < n) v = dotk(LiT + i * ldy + i, 1, Y + (size_t)i * ld + j, ld, ny - i);
                KT[e] = v;
            }
            __syncthreads();
            EKF_LAP(9);
        }
        for (int i = tid; i < n; i += nt) xn[i] = xv[i] + dotk(KT + i, ld, iv, 1, ny);
        wg::mfma_atb(AT, ld, KT, CS, NKy, n16 >> 4, n16 >> 4, ld, n);              // K CS
        EKF_LAP(10);
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e % n;
            a.Sigma[e] = SG[i * ld + j] - AT[i * ld + j];
        }
        for (int e = tid; e < n; e += nt) a.x[e] = xn[e];
    } else {
        for (int e = tid; e < n * n; e += nt) a.Sigma[e] = SG[(e / n) * ld + e % n];
        for (int e = tid; e < n; e += nt) a.x[e] = xv[e];
    }
    if (tid == 0) *a.status = 0;
#ifdef SRH_PROFILE
    EKF_LAP(11);
    if (tid == 0)
        printf("ekf clocks: load %lld nearest %lld Aload %lld xn %lld pred-mfma %lld W %lld Cload+innov %lld upd-mfma %lld chol+inv %lld Y+KT %lld KCS %lld store %lld\n",
               ekp[0], ekp[1], ekp[2], ekp[3], ekp[4], ekp[5], ekp[6], ekp[7], ekp[8], ekp[9], ekp[10], ekp[11]);
#endif
}

// ---- the MFMA filter step for 64 < n_x <= 80 (the shipped Diamond basis: r = 36, n_x = 72), where three padded
// n16 x ld panels no longer fit LDS.  Two things make room: panels that only ever serve as k-major operands or receive
// n valid rows are allocated with NK = n rows instead of n16, and A^T is never staged -- both products that need it
// read it as an MFMA operand straight from the (L2-resident) transposed table.  K C Sigma^- is subtracted from Sigma^-
// on its way to HBM instead of going through a panel.  LDS at n = 72, n_y = 30: 161 KB.
//   C[i][j] = sum_k Lm[k][i] Rm[k][j], i < 16 MT, j < 16 NTl; Lm / Rm k-major with their own pitches, either one in LDS
//   or global (LG / RG), columns >= nl / nr of a global operand read as zero; rows >= crows of C are not stored,
//   columns >= ccols are stored as zeros; `epi(i, j, acc)` replaces the store when given.
template <bool LG, bool RG, typename LP, typename RP, typename EPI>
__device__ __forceinline__ void ekf_mm(lptr C, int ldc, int crows, int ccols, LP Lm, int ldl, int nl, RP Rm, int ldr, int nr,
                                       int K, int MT, int NTl, EPI epi, bool use_epi) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    for (int t = wave; t < MT * NTl; t += nw) {
        const int ti = t / NTl, tj = t - ti * NTl;
        const int ci = 16 * ti + l16, cj = 16 * tj + l16;
        const bool li = !LG || ci < nl, rj = !RG || cj < nr;
        wg::qp_d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < K; k0 += 24) {              // six k-steps of operands in flight
            double av[6], bv[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int k = k0 + 4 * q + kk;
                const bool in = k0 + 4 * q < K;
                av[q] = (in && li) ? Lm[(size_t)(in ? k : 0) * ldl + (li ? ci : 0)] : 0.0;
                bv[q] = (in && rj) ? Rm[(size_t)(in ? k : 0) * ldr + (rj ? cj : 0)] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 6; ++q)
                if (k0 + 4 * q < K) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 16 * ti + kk + 4 * q;
            if (use_epi) epi(i, cj, acc[q]);
            else if (i < crows) C[(size_t)i * ldc + cj] = cj < ccols ? acc[q] : 0.0;
        }
    }
    __syncthreads();
}

__host__ __device__ inline size_t ekf_wide_doubles(int n, int ny) {
    const int n16 = (n + 15) & ~15, ny16 = (ny + 15) & ~15, NK = (n + 3) & ~3, ld = n16 + 1, ldy = ny16 + 1;
    return 2 * (size_t)NK * ld + (size_t)ny16 * ld + (size_t)NK * ldy + 3 * (size_t)ny16 * ldy + 4 * (size_t)n16 + 8;
}

__global__ __launch_bounds__(EKF_NT) void ekf_wide_kernel(EkfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = a.n, m = a.m, ny = a.ny;
    const int n16 = (n + 15) & ~15, ny16 = (ny + 15) & ~15, NK = (n + 3) & ~3, NKy = (ny + 3) & ~3, ld = n16 + 1, ldy = ny16 + 1;
    const int tid = threadIdx.x, nt = blockDim.x;
    lptr SG = (lptr)smem;                       // Sigma, then Sigma^-                   (NK x ld)
    lptr UU = SG + (size_t)NK * ld;             // U = Sigma A^T, then M1, then K^T      (NK x ld)
    lptr CS = UU + (size_t)NK * ld;             // C Sigma^-                             (ny16 x ld)
    lptr CT = CS + (size_t)ny16 * ld;           // C^T                                   (NK x ldy)
    lptr Sm = CT + (size_t)NK * ldy;            // S                                     (ny16 x ldy)
    lptr GB = Sm + (size_t)ny16 * ldy;          // elimination buffer                    (2 ny16 x ldy)
    lptr xv = GB + 2 * (size_t)ny16 * ldy, xn = xv + n16, iv = xn + n16, uv = iv + n16;
    liptr ip = (liptr)(uv + n16);
    auto none = [](int, int, double) {};

    constexpr int PQ = 16;                      // n <= 80: at most 13 entries of an n x n matrix per thread
    double wreg[PQ], creg[PQ / 2], vreg[2], yreg = 0.0;
    if (a.do_predict) {
#pragma unroll
        for (int k = 0; k < PQ; ++k) wreg[k] = tid + k * nt < n * n ? a.W[tid + k * nt] : 0.0;
    }
    if (a.do_update) {
#pragma unroll
        for (int k = 0; k < PQ / 2; ++k) creg[k] = tid + k * nt < ny * n ? a.C[tid + k * nt] : 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) vreg[k] = tid + k * nt < ny * ny ? a.V[tid + k * nt] : 0.0;
        if (tid < ny) yreg = a.y[tid] - (a.y_ref ? a.y_ref[tid] : 0.0);
    }
    const bool table = a.do_predict && a.Aext == nullptr;
    if (tid < 64) {
        for (int e = tid; e < n; e += 64) xv[e] = a.x[e];
        if (a.do_predict)
            for (int e = tid; e < m; e += 64) uv[e] = a.u[e];
        if (tid == 0) ip[1] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (table) {
            const int i = tpwl::nearest_wave(a.T, xv);
            if (tid == 0) ip[0] = i;
        }
    } else {
        const int t2 = tid - 64, nt2 = nt - 64;
        for (int e = t2; e < NK * ld; e += nt2) {
            const int i = e / ld, j = e - i * ld;
            SG[e] = (i < n && j < n) ? a.Sigma[i * n + j] : 0.0;
        }
        for (int e = t2; e < NK * ld + ny16 * ld + NK * ldy + 3 * ny16 * ldy; e += nt2) UU[e] = 0.0;
    }
    __syncthreads();

    if (a.do_predict) {
        // A^T as an operand: the transposed table of the nearest point, or the caller's A transposed on the fly
        const double *Bg, *dg;
        cgptr At;
        int ats, att;                                          // At[k * ats + i * att] = A[i][k]
        if (a.Aext != nullptr) { Bg = a.Bext; dg = a.dext; At = (cgptr)a.Aext; ats = 1; att = n; }
        else {
            const size_t i = (size_t)ip[0];
            At = a.T.AdT + i * n * n; ats = n; att = 1;
            Bg = (const double *)a.T.Bd + i * n * m; dg = (const double *)a.T.dd + i * n;
        }
        if (tid < n) {
            double t = 0.0;
            for (int k = 0; k < m; ++k) t = fma(Bg[tid * m + k], uv[k], t);
            double acc = 0.0;
            for (int k0 = 0; k0 < n; k0 += 8) {                 // eight table entries in flight, sums in order
                double av[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) av[q] = At[(size_t)(k0 + q < n ? k0 + q : n - 1) * ats + (size_t)tid * att];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (k0 + q < n) acc = fma(av[q], xv[k0 + q], acc);
            }
            xn[tid] = acc + t + dg[tid];
        }
        if (att == 1) {
            ekf_mm<false, true>(UU, ld, NK, n, SG, ld, n, At, n, n, NK, n16 >> 4, n16 >> 4, none, false);     // U = Sigma A^T
            ekf_mm<true, false>(SG, ld, NK, n, At, n, n, UU, ld, n, NK, n16 >> 4, n16 >> 4, none, false);     // Sigma^- = A U
        } else {
            // explicit (A_d, B_d, d_d) (weighting-mode models): stage A^T through the free C Sigma^- / C^T panels? they
            // are too small -- transpose into UU's rows is impossible while U is being formed; use the VALU products
            for (int e = tid; e < n * n; e += nt) {
                const int i = e / n, j = e - i * n;            // U[i][j] = sum_k Sigma[i][k] A[j][k]
                double acc = 0.0;
                for (int k = 0; k < n; ++k) acc = fma(SG[i * ld + k], a.Aext[(size_t)j * n + k], acc);
                UU[i * ld + j] = acc;
            }
            __syncthreads();
            for (int e = tid; e < n * n; e += nt) {
                const int i = e / n, j = e - i * n;            // Sigma^-[i][j] = sum_k A[i][k] U[k][j]
                double acc = 0.0;
                for (int k = 0; k < n; ++k) acc = fma(a.Aext[(size_t)i * n + k], UU[k * ld + j], acc);
                SG[i * ld + j] = acc;                          // rows of SG are read above only: safe after the barrier
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < PQ; ++k) {
            const int e = tid + k * nt;
            if (e < n * n) SG[(e / n) * ld + e % n] += wreg[k];
        }
        if (tid < n) xv[tid] = xn[tid];
        __syncthreads();
    }

    if (a.do_update) {
#pragma unroll
        for (int k = 0; k < PQ / 2; ++k) {
            const int e = tid + k * nt;
            if (e < ny * n) CT[(e % n) * ldy + e / n] = creg[k];                                  // C^T (n x ny)
        }
        for (int e = tid; e < NK * ld; e += nt) UU[e] = 0.0;
        __syncthreads();
        if (tid < ny) {
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc = fma(CT[k * ldy + tid], xv[k], acc);
            iv[tid] = yreg - acc;
        }
        ekf_mm<false, false>(UU, ld, NK, ny, SG, ld, n, CT, ldy, ny, NK, n16 >> 4, ny16 >> 4, none, false);   // M1 = Sigma^- C^T
        ekf_mm<false, false>(Sm, ldy, ny16, ny, CT, ldy, ny, UU, ld, ny, NK, ny16 >> 4, ny16 >> 4, none, false);   // C M1
        for (int e = tid; e < ny * n; e += nt) CS[(e / n) * ld + e % n] = UU[(e % n) * ld + e / n];             // C Sigma^- = M1^T
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = tid + k * nt;
            if (e < ny * ny) Sm[(e / ny) * ldy + e % ny] += vreg[k];
        }
        __syncthreads();
        lptr KT = UU;                                                                              // K^T (ny16 x ld), M1 is dead
        for (int e = tid; e < ny16 * ld; e += nt) KT[e] = 0.0;
        __syncthreads();
        (void)ekf_gain_gj<4, 2>(Sm, ldy, ny, CS, ld, n, KT, GB, ip + 1);
        __syncthreads();
        if (ip[1] != 0) {
            if (tid == 0) *a.status = 1;
            return;
        }
        if (tid < n) {
            double acc = 0.0;
            for (int k = 0; k < ny; ++k) acc = fma(KT[k * ld + tid], iv[k], acc);
            xn[tid] = xv[tid] + acc;
        }
        double *Sout = a.Sigma;
        auto sub = [&](int i, int j, double v) { if (i < n && j < n) Sout[(size_t)i * n + j] = SG[i * ld + j] - v; };
        ekf_mm<false, false>(SG, ld, 0, 0, KT, ld, n, CS, ld, n, NKy, n16 >> 4, n16 >> 4, sub, true);            // Sigma^- - K C Sigma^-
        if (tid < n) a.x[tid] = xn[tid];
    } else {
        for (int e = tid; e < n * n; e += nt) a.Sigma[e] = SG[(e / n) * ld + e % n];
        if (tid < n) a.x[tid] = xv[tid];
    }
    if (tid == 0) *a.status = 0;
}

size_t lds_bytes(int n, int ny) {
    const int ld = n | 1, ldy = ny | 1;
    const int nv = n > ny ? n : ny;
    return sizeof(double) * ((size_t)3 * n * ld + (size_t)ny * ld + (size_t)ny * ldy + 4 * (size_t)nv + 8);
}

}  // namespace

extern "C" {

int sekf_create(sekf_t **out, stpwl_t *model, const double *C, const double *y_ref, int n_y, const double *Sigma0,
                const double *W, const double *V) {
    SRH_REQUIRE(out && model && C && Sigma0 && W && V, "sekf_create: null argument");
    SRH_REQUIRE(n_y > 0 && n_y <= model->n, "sekf_create: need 0 < n_y <= n_x");
    auto *h = new sekf();
    h->model = model; h->n = model->n; h->m = model->m; h->ny = n_y;
    h->lds = lds_bytes(h->n, n_y);
    if (n_y <= 64 && 2 * ((n_y + 15) & ~15) <= ((h->n + 15) & ~15) && sizeof(double) * ekf_mfma_doubles(h->n, n_y) <= 160 * 1024 &&
        !getenv("SRH_EKF_NO_MFMA")) {
        h->mfma = true;
        h->lds = sizeof(double) * ekf_mfma_doubles(h->n, n_y);
    }
    if (!h->mfma && h->n > 64 && h->n <= 80 && n_y <= 32 && sizeof(double) * ekf_wide_doubles(h->n, n_y) <= 160 * 1024 &&
        !getenv("SRH_EKF_NO_MFMA")) {
        h->wide = true;
        h->lds = sizeof(double) * ekf_wide_doubles(h->n, n_y);
    }
    h->lds = srh::lds_request(h->lds);
    if (h->lds > 160 * 1024) {
        delete h;
        srh::set_error("sekf_create: the filter step does not fit the 160 KB LDS (n_x too large)");
        return SRH_EINVAL;
    }
    const size_t n = h->n;
    int rc;
    if ((rc = h->C.upload(C, sizeof(double) * n_y * n)) || (rc = h->W.upload(W, sizeof(double) * n * n)) ||
        (rc = h->V.upload(V, sizeof(double) * n_y * n_y)) || (rc = h->Sigma.upload(Sigma0, sizeof(double) * n * n)) ||
        (rc = h->x.alloc(sizeof(double) * (n + 1))) || (rc = h->scratch.alloc(sizeof(double) * (h->m + n_y) + 64)) ||
        (rc = h->ext.alloc(sizeof(double) * (n * n + n * h->m + n)))) {
        delete h;
        return rc;
    }
    if (y_ref && (rc = h->y_ref.upload(y_ref, sizeof(double) * n_y))) { delete h; return rc; }
    SRH_CHECK_HIP(hipMemset(h->x.p, 0, sizeof(double) * (n + 1)));
    SRH_CHECK_HIP(hipHostMalloc((void **)&h->pin_in, sizeof(double) * (h->m + n_y) + 64, hipHostMallocDefault));
    SRH_CHECK_HIP(hipHostMalloc((void **)&h->pin_out, sizeof(double) * (n + 2), hipHostMallocDefault));
    SRH_CHECK_HIP(hipFuncSetAttribute(h->mfma ? (n == 60 ? (const void *)ekf_mfma_kernel<60> : (const void *)ekf_mfma_kernel<0>)
                                              : (h->wide ? (const void *)ekf_wide_kernel : (const void *)ekf_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds));
    *out = h;
    return SRH_OK;
}

int sekf_destroy(sekf_t *h) {
    delete h;
    return SRH_OK;
}

int sekf_set_state(sekf_t *h, const double *x, const double *Sigma) {
    SRH_REQUIRE(h && (x || Sigma), "sekf_set_state: null argument");
    int rc;
    if (x && (rc = h->x.upload(x, sizeof(double) * h->n))) return rc;
    if (Sigma && (rc = h->Sigma.upload(Sigma, sizeof(double) * h->n * h->n))) return rc;
    return SRH_OK;
}

int sekf_get_state(sekf_t *h, double *x, double *Sigma) {
    SRH_REQUIRE(h, "sekf_get_state: null argument");
    int rc;
    if (x && (rc = h->x.download(x, sizeof(double) * h->n))) return rc;
    if (Sigma && (rc = h->Sigma.download(Sigma, sizeof(double) * h->n * h->n))) return rc;
    return SRH_OK;
}

// enqueue one predictor/update on stream 0 (inputs through the pinned mirror, state and status copied back to it)
static int ekf_enqueue(sekf *h, const double *u, const double *y, const double *A_d, const double *B_d,
                       const double *d_d) {
    const bool ext = A_d != nullptr;
    const int n = h->n, m = h->m, ny = h->ny;
    // scratch layout (device): [u (m) | y (ny)]; h->x holds x (n) and, behind it, the status word: one copy back
    double *su = h->scratch.as<double>();
    double *sy = su + m;
    int *st = (int *)(h->x.as<double>() + n);
    if (u) memcpy(h->pin_in, u, sizeof(double) * m);
    if (y) memcpy(h->pin_in + m, y, sizeof(double) * ny);
    SRH_CHECK_HIP(hipMemcpyAsync(su, h->pin_in, sizeof(double) * (m + ny), hipMemcpyHostToDevice, nullptr));
    double *e = h->ext.as<double>();
    if (ext && u) {
        SRH_CHECK_HIP(hipMemcpy(e, A_d, sizeof(double) * n * n, hipMemcpyHostToDevice));
        SRH_CHECK_HIP(hipMemcpy(e + (size_t)n * n, B_d, sizeof(double) * n * m, hipMemcpyHostToDevice));
        SRH_CHECK_HIP(hipMemcpy(e + (size_t)n * n + (size_t)n * m, d_d, sizeof(double) * n, hipMemcpyHostToDevice));
    }
    EkfArgs a{};
    a.T = h->model->view();
    a.n = n; a.m = m; a.ny = ny; a.ld = n | 1; a.ldy = ny | 1;
    a.C = h->C.as<double>(); a.y_ref = h->y_ref.p ? h->y_ref.as<double>() : nullptr;
    a.W = h->W.as<double>(); a.V = h->V.as<double>();
    a.x = h->x.as<double>(); a.Sigma = h->Sigma.as<double>();
    a.u = su; a.y = sy;
    if (ext && u) { a.Aext = e; a.Bext = e + (size_t)n * n; a.dext = e + (size_t)n * n + (size_t)n * m; }
    a.do_predict = u != nullptr; a.do_update = y != nullptr;
    a.status = st;
    if (h->mfma) {
        if (n == 60) ekf_mfma_kernel<60><<<1, EKF_NT, h->lds>>>(a);
        else ekf_mfma_kernel<0><<<1, EKF_NT, h->lds>>>(a);
    } else if (h->wide) {
        ekf_wide_kernel<<<1, EKF_NT, h->lds>>>(a);
    } else {
        ekf_kernel<<<1, EKF_NT, h->lds>>>(a);
    }
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipMemcpyAsync(h->pin_out, h->x.p, sizeof(double) * (n + 1), hipMemcpyDeviceToHost, nullptr));
    return SRH_OK;
}

// after stream 0 has drained: status check and the state estimate out of the pinned mirror
static int ekf_collect(sekf *h, double *x_out, const char *who) {
    int status = 0;
    memcpy(&status, h->pin_out + h->n, sizeof(int));
    if (status != 0) {
        srh::set_error("%s: innovation covariance S is not positive definite", who);
        return SRH_ENUMERIC;
    }
    if (x_out) memcpy(x_out, h->pin_out, sizeof(double) * h->n);
    return SRH_OK;
}

int sekf_step(sekf_t *h, const double *u, const double *y, const double *A_d, const double *B_d, const double *d_d,
              double *x_out) {
    SRH_REQUIRE(h, "sekf_step: null argument");
    SRH_REQUIRE(u || y, "sekf_step: need an input (predict) and/or a measurement (update)");
    SRH_REQUIRE(!A_d || (B_d && d_d), "sekf_step: A_d given without B_d, d_d");
    SRH_REQUIRE(!u || A_d || h->model->has_discrete, "sekf_step: model has not been pre-discretised");
    int rc = ekf_enqueue(h, u, y, A_d, B_d, d_d);
    if (rc) return rc;
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    return ekf_collect(h, x_out, "sekf_step");
}

int sekf_step_projected(sekf_t *h, srom_t *rom, const double *x_full, const double *u, const double *y,
                        double *x_reduced_out, double *x_hat_out) {
    SRH_REQUIRE(h && rom && x_full && x_reduced_out, "sekf_step_projected: null argument");
    SRH_REQUIRE(u || y, "sekf_step_projected: need an input (predict) and/or a measurement (update)");
    SRH_REQUIRE(!u || h->model->has_discrete, "sekf_step_projected: model has not been pre-discretised");
    int rc;
    if (!h->side) SRH_CHECK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    if (!h->side_gate) SRH_CHECK_HIP(hipEventCreateWithFlags(&h->side_gate, hipEventDisableTiming));
    // the side stream uses the rom's shared staging / split-K workspace: it starts behind whatever stream 0 has been
    // given before this call (un-synchronised srom_*_dev(stream = NULL) launches on the same rom)
    SRH_CHECK_HIP(hipEventRecord(h->side_gate, nullptr));
    SRH_CHECK_HIP(hipStreamWaitEvent(h->side, h->side_gate, 0));
    // the two halves are independent (the filter never reads the projected state): filter on stream 0 -- enqueued
    // first, its one-workgroup kernel is the long pole -- projection on the side stream, one wait for each
    auto drain = [&](int code) { (void)hipStreamSynchronize(h->side); (void)hipStreamSynchronize(nullptr); return code; };
    if ((rc = ekf_enqueue(h, u, y, nullptr, nullptr, nullptr))) return drain(rc);
    if ((rc = srom_stage_project(rom, SROM_X, x_full, 1, h->side))) return drain(rc);
    if (hipStreamSynchronize(h->side) != hipSuccess) { srh::set_error("sekf_step_projected: side stream failed"); return drain(SRH_EHIP); }
    if ((rc = srom_stage_collect(rom, x_reduced_out, 1, SROM_X))) return drain(rc);
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    return ekf_collect(h, x_hat_out, "sekf_step_projected");
}

}  // extern "C"
