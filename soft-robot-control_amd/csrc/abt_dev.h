// C = A B^T on f64 MFMA (A: M x K, B: N x K, row-major), 128 x 128 tiles, split along K with the parts summed in a fixed
// order (deterministic).  Shared by the leading-eigenpair solver (eigh_topk.hip: Z = Q G) and the snapshot preprocessing
// (snapshots.hip: the X C^T term of squared distances).  Everything lives in an anonymous namespace of the including unit.
#pragma once
#include <algorithm>

#include "common.h"
#include "dev_la.h"

namespace {

typedef double t_d4 __attribute__((ext_vector_type(4)));
constexpr int TB = 128, KC = 16, LDT = TB + 1;

// One K-part of one 128 x 128 tile of C = A B^T (A: M x K, B: N x K, row-major).  grid = tilesM * tilesN * ksplit; the
// parts go to scratch[(tile * ksplit + part)][128][128] and are summed in part order by abt_reduce_kernel.
__global__ __launch_bounds__(256) void abt_kernel(const double *__restrict__ A, int64_t lda, int64_t M, const double *__restrict__ B,
                                                  int64_t ldb, int64_t N, int64_t K, int tilesN, int ksplit, double *__restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lptr Xi = (lptr)smem;                  // [2][KC][LDT]
    lptr Xj = Xi + 2 * KC * LDT;
    const int tile = blockIdx.x / ksplit, part = blockIdx.x - tile * ksplit;
    const int ti = tile / tilesN, tj = tile - ti * tilesN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    const int lc = tid & 15, r4 = tid >> 4;
    const int64_t gi0 = (int64_t)ti * TB + r4, gj0 = (int64_t)tj * TB + r4;
    t_d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = t_d4{0.0, 0.0, 0.0, 0.0};
    double ri[8], rj[8];
    const double *pi[8], *pj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {                       // rows past M / N are clamped: they feed entries that are never used
        pi[q] = A + std::min<int64_t>(gi0 + 16 * q, M - 1) * lda + lc;
        pj[q] = B + std::min<int64_t>(gj0 + 16 * q, N - 1) * ldb + lc;
    }
    auto gload = [&](int64_t k0) {
        if (k0 + KC <= K) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { ri[q] = pi[q][k0]; rj[q] = pj[q][k0]; }
        } else {
            const bool vk = k0 + lc < K;
#pragma unroll
            for (int q = 0; q < 8; ++q) { ri[q] = vk ? pi[q][k0] : 0.0; rj[q] = vk ? pj[q][k0] : 0.0; }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            Xi[(buf * KC + lc) * LDT + r4 + 16 * q] = ri[q];
            Xj[(buf * KC + lc) * LDT + r4 + 16 * q] = rj[q];
        }
    };
    const int64_t nchunk_all = (K + KC - 1) / KC;
    const int64_t cbeg = nchunk_all * part / ksplit, cend = nchunk_all * (part + 1) / ksplit;
    if (cbeg < cend) {
        gload(cbeg * KC);
        lstore(cbeg & 1);
    }
    __syncthreads();
    for (int64_t c = cbeg; c < cend; ++c) {
        const int buf = c & 1;
        if (c + 1 < cend) gload((c + 1) * KC);
#pragma unroll
        for (int ks = 0; ks < KC; ks += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) af[a] = Xi[(buf * KC + ks + kk) * LDT + wr + 16 * a + l16];
#pragma unroll
            for (int a = 0; a < 4; ++a) bf[a] = Xj[(buf * KC + ks + kk) * LDT + wc + 16 * a + l16];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    acc[a][cc] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[cc], acc[a][cc], 0, 0, 0);
        }
        if (c + 1 < cend) lstore(buf ^ 1);
        __syncthreads();
    }
    double *dst = scratch + (size_t)blockIdx.x * TB * TB;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dst[(wr + 16 * a + kk + 4 * q) * TB + wc + 16 * cc + l16] = acc[a][cc][q];
}

__global__ __launch_bounds__(256) void abt_reduce_kernel(const double *__restrict__ scratch, int tilesN, int ksplit, double *__restrict__ C,
                                                         int64_t ldc, int64_t M, int64_t N) {
    const int tile = blockIdx.x, ti = tile / tilesN, tj = tile - ti * tilesN;
    const double *src = scratch + (size_t)tile * ksplit * TB * TB;
    for (int e = threadIdx.x; e < TB * TB; e += blockDim.x) {
        const int64_t r = (int64_t)ti * TB + e / TB, c = (int64_t)tj * TB + e % TB;
        if (r >= M || c >= N) continue;
        double v = 0.0;
        for (int p = 0; p < ksplit; ++p) v += src[(size_t)p * TB * TB + e];
        C[r * ldc + c] = v;
    }
}

struct Abt {
    srh::DevBuf scratch;
    size_t bytes = 0;
    int cus = 0;
    // C (M x N) = A (M x K) B^T
    int run(const double *A, int64_t lda, int64_t M, const double *B, int64_t ldb, int64_t N, int64_t K, double *C, int64_t ldc,
            hipStream_t st) {
        if (cus == 0) {
            int dev = 0;
            SRH_CHECK_HIP(hipGetDevice(&dev));
            SRH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            if (cus <= 0) cus = 256;
        }
        const int tilesM = (int)srh::cdiv(M, TB), tilesN = (int)srh::cdiv(N, TB), tiles = tilesM * tilesN;
        const int64_t nchunk = srh::cdiv(K, KC);
        int ksplit = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(2 * cus + tiles - 1) / tiles, nchunk / 8 > 0 ? nchunk / 8 : 1, 64}));
        const size_t need = sizeof(double) * (size_t)tiles * ksplit * TB * TB;
        if (need > bytes) {
            SRH_CHECK_HIP(hipStreamSynchronize(st));
            int rc = scratch.alloc(need);
            if (rc) return rc;
            bytes = need;
        }
        abt_kernel<<<(unsigned)(tiles * ksplit), 256, sizeof(double) * 4 * KC * LDT, st>>>(A, lda, M, B, ldb, N, K, tilesN, ksplit,
                                                                                             scratch.as<double>());
        abt_reduce_kernel<<<(unsigned)tiles, 256, 0, st>>>(scratch.as<double>(), tilesN, ksplit, C, ldc, M, N);
        SRH_CHECK_HIP(hipGetLastError());
        return SRH_OK;
    }
};

}  // namespace
