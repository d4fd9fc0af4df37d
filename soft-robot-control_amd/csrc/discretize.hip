// Time discretisation of continuous affine models  xdot = A x + B u + d  on the device, batched: TPWL.discretize_dynamics
// (sofacontrol/tpwl/tpwl.py:272-297) for the stored points of pre_discretize (tpwl.py:299-322) and for the blended model that
// weighting-mode TPWL re-discretises at every step (tpwl.py:244-250).  One workgroup per model.
//   fe :  A_d = I + dt A,                      B_d = dt B,                 d_d = dt d
//   be :  A_d = (I - dt A)^-1,                 [B_d d_d] = A^-1 (A_d - I) [B d]  =  dt A_d [B d]
//   bil:  A_d = (I + h A)(I - h A)^-1, h = dt/2, [B_d d_d] = A^-1 (A_d - I) [B d]  =  dt (I - h A)^-1 [B d]
//         (the reference forms A^-1 (A_d - I) with a second inverse; the right-hand forms are the same matrices without inverting A:
//          one Gauss-Jordan elimination with partial pivoting on [I - c A | right-hand sides] in LDS)
//   zoh:  expm([[A, B, d], [0, 0, 0]] dt)      (sofacontrol/utils.py:302-335; scipy.linalg.expm there)
// expm: scaling and squaring with the [13/13] Pade approximant (Higham 2005, "The scaling and squaring method for the matrix
// exponential revisited", Alg. 2.3 with m = 13 always: theta_13 = 5.37; scipy picks lower orders for small norms -- the same
// function to rounding).  Six 16 x 16-tiled f64 MFMA products + one elimination + s squarings per matrix, operands in an L2 workspace.
#include "common.h"
#include "dev_la.h"

#include <cmath>
#include <vector>

namespace {

constexpr int DZ_NT = 512;
typedef double dz_d4 __attribute__((ext_vector_type(4)));

// C = X Y for NP x NP row-major matrices in global memory (NP a multiple of 16): tiles dealt to the waves round-robin
__device__ __forceinline__ void dz_product(const double *__restrict__ X, const double *__restrict__ Y, double *__restrict__ Cm, int NP) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l16 = lane & 15, kk = lane >> 4;
    const int T = NP >> 4, nw = blockDim.x >> 6;
    for (int t = wave; t < T * T; t += nw) {
        const int ti = t / T, tj = t - ti * T;
        dz_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int s = 0; s < NP / 4; ++s)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(size_t)(16 * ti + l16) * NP + 4 * s + kk], Y[(size_t)(4 * s + kk) * NP + 16 * tj + l16], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) Cm[(size_t)(16 * ti + kk + 4 * q) * NP + 16 * tj + l16] = acc[q];
    }
    __threadfence_block();
    __syncthreads();
}

// Gauss-Jordan with partial pivoting on the LDS tableau T (rows x ld): the first `rows` columns hold the matrix, columns rows..cols-1
// the right-hand sides; on return those columns hold the solutions.  Returns false on a zero pivot.
__device__ inline bool dz_gauss_jordan(lptr T, int rows, int cols, int ld, lptr red, liptr ired) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int p = 0; p < rows; ++p) {
        // pivot: the largest |T[i][p]|, i >= p (ties: the smallest i)
        if (tid < 64) {
            double best = -1.0; int bi = p;
            for (int i = p + tid; i < rows; i += 64) {
                const double v = fabs(T[i * ld + p]);
                if (v > best) { best = v; bi = i; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) { ired[0] = bi; red[0] = best; }
        }
        __syncthreads();
        const int piv = ired[0];
        if (!(red[0] > 0.0)) return false;
        if (piv != p) {
            for (int j = tid; j < cols; j += nt) { const double a = T[p * ld + j]; T[p * ld + j] = T[piv * ld + j]; T[piv * ld + j] = a; }
            __syncthreads();
        }
        const double inv = 1.0 / T[p * ld + p];
        __syncthreads();
        for (int j = tid; j < cols; j += nt) T[p * ld + j] *= inv;
        // the multipliers, before the column is overwritten
        for (int i = tid; i < rows; i += nt) red[8 + i] = (i == p) ? 0.0 : T[i * ld + p];
        __syncthreads();
        for (int e = tid; e < rows * (cols - p); e += nt) {
            const int i = e / (cols - p), j = p + e - i * (cols - p);
            T[i * ld + j] = fma(-red[8 + i], T[p * ld + j], T[i * ld + j]);
        }
        __syncthreads();
    }
    return true;
}

struct DiscArgs {
    const double *A, *B, *d;       // batch x (n x n), (n x m), (n)
    double *Ad, *Bd, *dd;
    double *ws;                    // zoh: batch x 7 NP^2 doubles
    int *info;                     // batch
    double dt;
    int n, m, NP, method;          // method: 0 fe, 1 be, 2 bil, 3 zoh
};

__global__ __launch_bounds__(DZ_NT) void discretize_kernel(DiscArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, n = a.n, m = a.m;
    const size_t b = blockIdx.x;
    const double *A = a.A + b * n * n, *B = a.B + b * n * m, *d = a.d + b * n;
    double *Ad = a.Ad + b * n * n, *Bd = a.Bd + b * n * m, *dd = a.dd + b * n;
    const double dt = a.dt;
    if (tid == 0) a.info[b] = 0;
    if (a.method == 0) {
        for (int e = tid; e < n * n; e += nt) Ad[e] = ((e / n == e % n) ? 1.0 : 0.0) + dt * A[e];
        for (int e = tid; e < n * m; e += nt) Bd[e] = dt * B[e];
        for (int e = tid; e < n; e += nt) dd[e] = dt * d[e];
        return;
    }
    if (a.method == 1 || a.method == 2) {
        // tableau [I - c A | R | dt B | dt d],  R = I (be) or I + c A (bil)
        const double c = a.method == 1 ? dt : 0.5 * dt;
        const int cols = 2 * n + m + 1, ld = cols | 1;
        lptr T = (lptr)smem, red = T + (size_t)n * ld;
        liptr ired = (liptr)(red + 8 + n);
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e - i * n;
            const double id = (i == j) ? 1.0 : 0.0, ca = c * A[e];
            T[i * ld + j] = id - ca;
            T[i * ld + n + j] = a.method == 1 ? id : id + ca;
        }
        for (int e = tid; e < n * m; e += nt) { const int i = e / m, j = e - i * m; T[i * ld + 2 * n + j] = dt * B[e]; }
        for (int i = tid; i < n; i += nt) T[i * ld + 2 * n + m] = dt * d[i];
        __syncthreads();
        const bool ok = dz_gauss_jordan(T, n, cols, ld, red, ired);
        if (!ok) { if (tid == 0) a.info[b] = 1; return; }
        for (int e = tid; e < n * n; e += nt) { const int i = e / n, j = e - i * n; Ad[e] = T[i * ld + n + j]; }
        for (int e = tid; e < n * m; e += nt) { const int i = e / m, j = e - i * m; Bd[e] = T[i * ld + 2 * n + j]; }
        for (int i = tid; i < n; i += nt) dd[i] = T[i * ld + 2 * n + m];
        return;
    }
    // ---- zoh
    const int NP = a.NP, ne = n + m + 1;
    const size_t N2 = (size_t)NP * NP;
    double *M = a.ws + b * 7 * N2, *M2 = M + N2, *M4 = M2 + N2, *M6 = M4 + N2, *W = M6 + N2, *U = W + N2, *V = U + N2;
    lptr red = (lptr)smem;                               // reductions; the tableau of the solve behind it
    // M = [[A, B, d], [0, 0, 0]] dt, its 1-norm (largest column sum)
    for (int e = tid; e < (int)N2; e += nt) {
        const int i = e / NP, j = e - i * NP;
        double v = 0.0;
        if (i < n) v = j < n ? A[i * n + j] : (j < n + m ? B[i * m + (j - n)] : (j == n + m ? d[i] : 0.0));
        M[e] = v * dt;
    }
    __threadfence_block();
    __syncthreads();
    double cs = 0.0;
    for (int j = tid; j < ne; j += nt) {
        double sum = 0.0;
        for (int i = 0; i < n; ++i) sum += fabs(M[(size_t)i * NP + j]);
        cs = fmax(cs, sum);
    }
    const double nrm = wg::reduce(cs, 1, red);
    int s = 0;
    if (!(nrm == nrm) || nrm > 1e300) { if (tid == 0) a.info[b] = 2; return; }
    if (nrm > 5.371920351148152) s = (int)ceil(log2(nrm / 5.371920351148152));
    if (s > 0) {
        const double sc = ldexp(1.0, -s);
        for (int e = tid; e < (int)N2; e += nt) M[e] *= sc;
        __threadfence_block();
        __syncthreads();
    }
    const double b0 = 64764752532480000.0, b1 = 32382376266240000.0, b2 = 7771770303897600.0, b3 = 1187353796428800.0,
                 b4 = 129060195264000.0, b5 = 10559470521600.0, b6 = 670442572800.0, b7 = 33522128640.0, b8 = 1323241920.0,
                 b9 = 40840800.0, b10 = 960960.0, b11 = 16380.0, b12 = 182.0, b13 = 1.0;
    dz_product(M, M, M2, NP);
    dz_product(M2, M2, M4, NP);
    dz_product(M4, M2, M6, NP);
    // U = M (M6 (b13 M6 + b11 M4 + b9 M2) + b7 M6 + b5 M4 + b3 M2 + b1 I)
    for (int e = tid; e < (int)N2; e += nt) W[e] = b13 * M6[e] + b11 * M4[e] + b9 * M2[e];
    __threadfence_block();
    __syncthreads();
    dz_product(M6, W, V, NP);
    for (int e = tid; e < (int)N2; e += nt) V[e] += b7 * M6[e] + b5 * M4[e] + b3 * M2[e] + ((e / NP == e % NP) ? b1 : 0.0);
    __threadfence_block();
    __syncthreads();
    dz_product(M, V, U, NP);
    // V = M6 (b12 M6 + b10 M4 + b8 M2) + b6 M6 + b4 M4 + b2 M2 + b0 I
    for (int e = tid; e < (int)N2; e += nt) W[e] = b12 * M6[e] + b10 * M4[e] + b8 * M2[e];
    __threadfence_block();
    __syncthreads();
    dz_product(M6, W, V, NP);
    for (int e = tid; e < (int)N2; e += nt) V[e] += b6 * M6[e] + b4 * M4[e] + b2 * M2[e] + ((e / NP == e % NP) ? b0 : 0.0);
    __threadfence_block();
    __syncthreads();
    // (V - U) R = V + U on the leading ne x ne block (the padding is the identity)
    {
        const int cols = 2 * ne, ld = cols | 1;
        lptr T = red + 8 + NP + 8;
        liptr ired = (liptr)(red + 8 + NP);
        for (int e = tid; e < ne * ne; e += nt) {
            const int i = e / ne, j = e - i * ne;
            const double u = U[(size_t)i * NP + j], v = V[(size_t)i * NP + j];
            T[i * ld + j] = v - u;
            T[i * ld + ne + j] = v + u;
        }
        __syncthreads();
        const bool ok = dz_gauss_jordan(T, ne, cols, ld, red, ired);
        if (!ok) { if (tid == 0) a.info[b] = 1; return; }
        for (int e = tid; e < (int)N2; e += nt) {
            const int i = e / NP, j = e - i * NP;
            M2[e] = (i < ne && j < ne) ? T[i * ld + ne + j] : ((i == j) ? 1.0 : 0.0);
        }
        __threadfence_block();
        __syncthreads();
    }
    double *R = M2, *R2 = M4;
    for (int q = 0; q < s; ++q) {
        dz_product(R, R, R2, NP);
        double *t = R; R = R2; R2 = t;
    }
    for (int e = tid; e < n * n; e += nt) { const int i = e / n, j = e - i * n; Ad[e] = R[(size_t)i * NP + j]; }
    for (int e = tid; e < n * m; e += nt) { const int i = e / m, j = e - i * m; Bd[e] = R[(size_t)i * NP + n + j]; }
    for (int i = tid; i < n; i += nt) dd[i] = R[(size_t)i * NP + n + m];
}

size_t discretize_lds(int n, int m, int method) {
    if (method == 1 || method == 2) {
        const int cols = 2 * n + m + 1, ld = cols | 1;
        return sizeof(double) * ((size_t)n * ld + 8 + n) + 64;
    }
    if (method == 3) {
        const int ne = n + m + 1, NP = (ne + 15) & ~15, ld = (2 * ne) | 1;
        return sizeof(double) * ((size_t)ne * ld + 8 + NP + 8) + 64;
    }
    return 64;
}

}  // namespace

extern "C" {

int stpwl_discretize_dev(int method, int n, int m, int64_t batch, const double *A_dev, const double *B_dev, const double *d_dev, double dt,
                         double *Ad_dev, double *Bd_dev, double *dd_dev, void *stream) {
    SRH_REQUIRE(method >= 0 && method <= 3, "self.discr_method must be in [fe, be, bil, zoh]");
    SRH_REQUIRE(n > 0 && m > 0 && batch >= 0 && A_dev && B_dev && d_dev && Ad_dev && Bd_dev && dd_dev, "stpwl_discretize_dev: bad argument");
    if (batch == 0) return SRH_OK;
    const size_t lds = srh::lds_request(discretize_lds(n, m, method));
    SRH_REQUIRE(lds <= 160 * 1024, "stpwl_discretize_dev: model too large for the LDS tableau (n_x %d, n_u %d: %zu bytes)", n, m, lds);
    DiscArgs a{};
    a.A = A_dev; a.B = B_dev; a.d = d_dev; a.Ad = Ad_dev; a.Bd = Bd_dev; a.dd = dd_dev; a.dt = dt; a.n = n; a.m = m; a.method = method;
    a.NP = (n + m + 1 + 15) & ~15;
    srh::DevBuf ws, info;
    int rc;
    if ((rc = info.alloc(sizeof(int) * batch))) return rc;
    if (method == 3 && (rc = ws.alloc(sizeof(double) * (size_t)batch * 7 * a.NP * a.NP))) return rc;
    a.ws = ws.as<double>(); a.info = info.as<int>();
    SRH_CHECK_HIP(hipFuncSetAttribute((const void *)discretize_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    discretize_kernel<<<(unsigned)batch, DZ_NT, lds, (hipStream_t)stream>>>(a);
    SRH_CHECK_HIP(hipGetLastError());
    std::vector<int> hinfo((size_t)batch);
    SRH_CHECK_HIP(hipMemcpyAsync(hinfo.data(), info.p, sizeof(int) * batch, hipMemcpyDeviceToHost, (hipStream_t)stream));
    SRH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    for (int64_t i = 0; i < batch; ++i)
        if (hinfo[(size_t)i] != 0) {
            srh::set_error(hinfo[(size_t)i] == 1 ? "stpwl_discretize: singular matrix (model %lld)" : "stpwl_discretize: non-finite model (model %lld)", (long long)i);
            return SRH_ENUMERIC;
        }
    return SRH_OK;
}

int stpwl_discretize(int method, int n, int m, int64_t batch, const double *A, const double *B, const double *d, double dt, double *Ad,
                     double *Bd, double *dd) {
    SRH_REQUIRE(A && B && d && Ad && Bd && dd && n > 0 && m > 0 && batch >= 0, "stpwl_discretize: bad argument");
    if (batch == 0) return SRH_OK;
    srh::DevBuf dA, dB, dD, oA, oB, oD;
    int rc;
    const size_t sA = sizeof(double) * batch * n * n, sB = sizeof(double) * batch * n * m, sD = sizeof(double) * batch * n;
    if ((rc = dA.upload(A, sA)) || (rc = dB.upload(B, sB)) || (rc = dD.upload(d, sD)) || (rc = oA.alloc(sA)) || (rc = oB.alloc(sB)) ||
        (rc = oD.alloc(sD)))
        return rc;
    if ((rc = stpwl_discretize_dev(method, n, m, batch, dA.as<double>(), dB.as<double>(), dD.as<double>(), dt, oA.as<double>(),
                                   oB.as<double>(), oD.as<double>(), nullptr)))
        return rc;
    if ((rc = oA.download(Ad, sA)) || (rc = oB.download(Bd, sB)) || (rc = oD.download(dd, sD))) return rc;
    return SRH_OK;
}

}  // extern "C"
