// Snapshot Gramian G = S S^T (method of snapshots for compute_POD, sofacontrol/mor/pod.py:181-200) and
// mode recovery U_k = S^T W_k, f64 MFMA.  S is (n_s x n_f) row-major (one snapshot per row, the layout of
// np.asarray(data['q']), pod.py:149).  Compute-bound (n_s/8 flop per byte): 128 x 128 output tiles per
// workgroup, 64 x 64 per wave (16 accumulator tiles), K streamed in 16-column chunks through a
// double-buffered k-major LDS panel; only tiles on or above the diagonal are computed and mirrored.
#include <algorithm>

#include "common.h"
#include "dev_la.h"

namespace {

typedef double g_d4 __attribute__((ext_vector_type(4)));
constexpr int TB = 128;      // tile edge
constexpr int KC = 16;       // K chunk
constexpr int LDT = TB + 1;  // LDS row stride of the k-major panels

// Blocks [0, nfull) compute whole tiles.  The remaining `ntiles - nfull` tiles -- the partial last round of the
// launch on a 2-workgroups-per-CU machine -- are split `ksplit` ways along K, each part written to `scratch`
// (TB x TB per part) and summed in a fixed order by gramian_tail_kernel: the tail costs 1/ksplit of a round.
__global__ __launch_bounds__(256) void gramian_kernel(const double *__restrict__ S, int64_t n_s, int64_t n_f, int64_t lds,
                                                      double *__restrict__ G, int ntile, int nfull, int ksplit,
                                                      double *__restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lptr Xi = (lptr)smem;                  // [2][KC][LDT]
    lptr Xj = Xi + 2 * KC * LDT;           // [2][KC][LDT]
    // linear block index -> (ti <= tj)
    int b = blockIdx.x, ti = 0, part = -1;
    if (b >= nfull) { part = (b - nfull) % ksplit; b = nfull + (b - nfull) / ksplit; }
    const int tail_slot = b - nfull;
    while (b >= ntile - ti) { b -= ntile - ti; ++ti; }
    const int tj = ti + b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kk = lane >> 4;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;        // wave sub-tile origin
    // loader: 16 consecutive lanes read the 16 columns (128 B) of one row of the chunk -- a wave instruction
    // covers 4 rows x 128 B, fully coalesced (one 8-byte element per lane; per-lane rows r4 + 16 q, q < 8)
    const int lc = tid & 15, r4 = tid >> 4;
    const int64_t gi0 = (int64_t)ti * TB + r4, gj0 = (int64_t)tj * TB + r4;

    g_d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = g_d4{0.0, 0.0, 0.0, 0.0};

    // rows beyond n_s are clamped, not zeroed: they only feed rows / columns of the tile that are never stored, and
    // the loads of a whole chunk (k0 + KC <= n_f, uniform) then carry no per-element branch
    double ri[8], rj[8];
    const double *pi[8], *pj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        pi[q] = S + min(gi0 + 16 * q, n_s - 1) * lds + lc;
        pj[q] = S + min(gj0 + 16 * q, n_s - 1) * lds + lc;
    }
    auto gload = [&](int64_t k0) {
        if (k0 + KC <= n_f) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { ri[q] = pi[q][k0]; rj[q] = pj[q][k0]; }
        } else {
            const bool vk = k0 + lc < n_f;
#pragma unroll
            for (int q = 0; q < 8; ++q) { ri[q] = vk ? pi[q][k0] : 0.0; rj[q] = vk ? pj[q][k0] : 0.0; }
        }
    };
    // k-major panel: element (row, k) at [k][row]; the 16 lanes of a store group differ in k: LDT odd -> all banks
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            Xi[(buf * KC + lc) * LDT + r4 + 16 * q] = ri[q];
            Xj[(buf * KC + lc) * LDT + r4 + 16 * q] = rj[q];
        }
    };
    const int64_t nchunk_all = (n_f + KC - 1) / KC;
    const int64_t cbeg = part < 0 ? 0 : nchunk_all * part / ksplit;
    const int64_t nchunk = part < 0 ? nchunk_all : nchunk_all * (part + 1) / ksplit;
    gload(cbeg * KC);
    lstore(cbeg & 1);
    __syncthreads();
    for (int64_t c = cbeg; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload((c + 1) * KC);
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
        if (c + 1 < nchunk) lstore(buf ^ 1);
        __syncthreads();
    }
    // D: col = lane&15, row = (lane>>4) + 4*reg
    if (part >= 0) {
        double *dst = scratch + ((size_t)tail_slot * ksplit + part) * TB * TB;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dst[(wr + 16 * a + kk + 4 * q) * TB + wc + 16 * cc + l16] = acc[a][cc][q];
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t r = (int64_t)ti * TB + wr + 16 * a + kk + 4 * q;
                const int64_t cidx = (int64_t)tj * TB + wc + 16 * cc + l16;
                if (r < n_s && cidx < n_s) {
                    const double v = acc[a][cc][q];
                    G[r * n_s + cidx] = v;
                    if (ti != tj) G[cidx * n_s + r] = v;
                }
            }
}

// sum of the K-parts of the split tail tiles in part order, written to G and mirrored
__global__ __launch_bounds__(256) void gramian_tail_kernel(const double *__restrict__ scratch, int64_t n_s, int ntile,
                                                           int nfull, int ksplit, double *__restrict__ G) {
    int b = nfull + blockIdx.x, ti = 0;
    while (b >= ntile - ti) { b -= ntile - ti; ++ti; }
    const int tj = ti + b;
    const double *src = scratch + (size_t)blockIdx.x * ksplit * TB * TB;
    for (int e = threadIdx.x; e < TB * TB; e += blockDim.x) {
        const int64_t r = (int64_t)ti * TB + e / TB, c = (int64_t)tj * TB + e % TB;
        if (r >= n_s || c >= n_s) continue;
        double v = 0.0;
        for (int p = 0; p < ksplit; ++p) v += src[(size_t)p * TB * TB + e];
        G[r * n_s + c] = v;
        if (ti != tj) G[c * n_s + r] = v;
    }
}

// U (n_f x k) = S^T W, W (n_s x k), k <= 64
__global__ __launch_bounds__(256) void modes_kernel(const double *__restrict__ S, int64_t n_s, int64_t n_f, int64_t lds,
                                                    const double *__restrict__ W, int k, double *__restrict__ U) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lptr Ws = (lptr)smem;                 // [64 rows of s][k]
    const int tid = threadIdx.x;
    const int ic = tid & 63, jg = tid >> 6;            // column of S (i), group of 16 output columns
    const int64_t i = (int64_t)blockIdx.x * 64 + ic;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int64_t s0 = 0; s0 < n_s; s0 += 64) {
        const int ns = (int)min((int64_t)64, n_s - s0);
        for (int e = tid; e < ns * k; e += 256) Ws[e] = W[(s0 + e / k) * k + (e % k)];
        __syncthreads();
        if (i < n_f) {
            for (int s = 0; s < ns; ++s) {
                const double v = S[(s0 + s) * lds + i];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int j = jg * 16 + q;
                    if (j < k) acc[q] = fma(v, Ws[s * k + j], acc[q]);
                }
            }
        }
        __syncthreads();
    }
    if (i < n_f) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int j = jg * 16 + q;
            if (j < k) U[i * k + j] = acc[q];
        }
    }
}

}  // namespace

extern "C" {

int srom_gramian_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, double *G_dev, void *stream) {
    SRH_REQUIRE(S_dev && G_dev, "srom_gramian_dev: null argument");
    SRH_REQUIRE(n_s > 0 && n_f > 0 && lds >= n_f, "srom_gramian_dev: bad dimensions");
    const int ntile = (int)srh::cdiv(n_s, TB);
    const int64_t nblk = (int64_t)ntile * (ntile + 1) / 2;
    const size_t lbytes = sizeof(double) * 4 * KC * LDT;
    // two workgroups of this kernel are resident per CU: when the last round of the launch is less than half
    // full, its tiles are split along K so that the tail costs a fraction of a round
    static int slots = 0;
    static srh::DevBuf scratch;
    static size_t scratch_bytes = 0;
    if (slots == 0) {
        int dev = 0, cus = 0;
        SRH_CHECK_HIP(hipGetDevice(&dev));
        SRH_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        slots = 2 * (cus > 0 ? cus : 256);
    }
    const int64_t nchunk = srh::cdiv(n_f, KC);
    const int rem = (int)(nblk % slots);
    int ksplit = 1, ntail = 0;
    if (nblk > slots && rem > 0 && 2 * rem <= slots && nchunk >= 64) {
        ksplit = std::min(8, slots / rem);
        ntail = rem;
    }
    const int nfull = (int)nblk - ntail;
    if (ntail > 0) {
        const size_t need = sizeof(double) * (size_t)ntail * ksplit * TB * TB;
        if (need > scratch_bytes) {
            SRH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
            int rc = scratch.alloc(need);
            if (rc) return rc;
            scratch_bytes = need;
        }
    }
    gramian_kernel<<<(unsigned)(nfull + ntail * ksplit), 256, lbytes, (hipStream_t)stream>>>(
        S_dev, n_s, n_f, lds, G_dev, ntile, nfull, ksplit, scratch.as<double>());
    SRH_CHECK_HIP(hipGetLastError());
    if (ntail > 0) {
        gramian_tail_kernel<<<(unsigned)ntail, 256, 0, (hipStream_t)stream>>>(scratch.as<double>(), n_s, ntile, nfull, ksplit,
                                                                             G_dev);
        SRH_CHECK_HIP(hipGetLastError());
    }
    return SRH_OK;
}

int srom_gramian(const double *S, int64_t n_s, int64_t n_f, double *G) {
    SRH_REQUIRE(S && G, "srom_gramian: null argument");
    srh::DevBuf dS, dG;
    int rc;
    if ((rc = dS.upload(S, sizeof(double) * n_s * n_f)) || (rc = dG.alloc(sizeof(double) * n_s * n_s))) return rc;
    if ((rc = srom_gramian_dev(dS.as<double>(), n_s, n_f, n_f, dG.as<double>(), nullptr))) return rc;
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    return dG.download(G, sizeof(double) * n_s * n_s);
}

int srom_modes_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, const double *W_dev, int k, double *U_dev,
                   void *stream) {
    SRH_REQUIRE(S_dev && W_dev && U_dev, "srom_modes_dev: null argument");
    SRH_REQUIRE(k > 0 && k <= 64, "srom_modes_dev: need 0 < k <= 64");
    modes_kernel<<<(unsigned)srh::cdiv(n_f, 64), 256, sizeof(double) * 64 * k, (hipStream_t)stream>>>(S_dev, n_s, n_f, lds, W_dev, k, U_dev);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

}  // extern "C"
