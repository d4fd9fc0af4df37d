// Cycle counts of the building blocks of the condensed interior point (locp_cond.h) in isolation: one workgroup.
#include "../../soft-robot-control_amd/csrc/common.h"
#include "../../soft-robot-control_amd/csrc/tpwl_dev.h"
#include "../../soft-robot-control_amd/csrc/locp_dev.h"
#include <cstdio>
#include <vector>
#include <cmath>

__global__ __launch_bounds__(512) void probe(QPDims d, double *Kmat, double *vec, long long *out, double *chk, double *GTbuf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    qpc::Lds L;
    qpc::lds_carve(L, (lptr)smem, d, 512);
    const int tid = threadIdx.x, KT = d.KT, wave = tid >> 6, lane = tid & 63;
    auto load_tiles = [&]() {
        for (int e = tid; e < KT * (KT + 1) / 2 * 256; e += 512) {
            int t = e / 256, rc = e % 256, r = rc / 16, cc = rc % 16, I = 0, tt = t;
            while (tt >= KT - I) { tt -= KT - I; ++I; }
            const int J = I + tt;
            L.B[(size_t)t * qpc::TSZ + r * qpc::TS + cc] = Kmat[(size_t)(16 * I + r) * 16 * KT + 16 * J + cc];
        }
        __syncthreads();
    };
    long long t0, t1;
    // 1. chol16 alone (wave 0), on a copy of tile 0
    load_tiles();
    t0 = clock64();
    if (wave == 0) qpc::chol16(L.B, L.Rinv);
    __syncthreads();
    t1 = clock64();
    if (tid == 0) out[0] = t1 - t0;
    // 2. full tile Cholesky
    load_tiles();
    t0 = clock64();
    const bool ok = qpc::tile_cholesky(d, L);
    t1 = clock64();
    if (tid == 0) { out[1] = t1 - t0; out[7] = ok; }
    // 3. k_solve
    for (int e = tid; e < 16 * KT; e += 512) L.yd[e] = vec[e];
    __syncthreads();
    t0 = clock64();
    qpc::k_solve(d, L, L.yd);
    t1 = clock64();
    if (tid == 0) out[2] = t1 - t0;
    for (int e = tid; e < 16 * KT; e += 512) chk[e] = L.yd[e];
    // 4. r_times + rT_times
    t0 = clock64();
    qpc::r_times(d, L, L.yd, L.ya);
    qpc::rT_times(d, L, L.ya, L.yb);
    t1 = clock64();
    if (tid == 0) out[3] = t1 - t0;
    for (int e = tid; e < 16 * KT; e += 512) chk[16 * KT + e] = L.yb[e];
    // 4b. products with G (L2 resident: written by this workgroup first)
    {
        QCWork qw; qw.GT = (gptr)GTbuf;
        const int nm = d.N * d.m, ldG = 16 * d.KT;
        for (int e = tid; e < nm * ldG; e += 512) { const int r = e / ldG, i = e % ldG; GTbuf[e] = (i >= (r / d.m) * d.po && i < d.N * d.po) ? 1e-3 * ((e * 7) % 13 - 6) : 0.0; }
        for (int e = tid; e < nm; e += 512) { L.u[e] = 0.01 * (e % 11); L.Ldi[e] = 1.0 + 0.001 * e; }
        for (int e = tid; e < ldG; e += 512) L.y[e] = 0.02 * (e % 7);
        for (int e = tid; e < d.N * 4; e += 512) L.Ls[e] = (e % 4 == 1) ? 0.0 : 1.0 + 0.01 * e;
        __syncthreads();
        for (int rep = 0; rep < 2; ++rep) {
            t0 = clock64();
            qpc::g_times(d, qw, L, L.u, L.ya);
            t1 = clock64();
            if (tid == 0) out[8] = t1 - t0;
            t0 = clock64();
            qpc::gT_times(d, qw, L, L.y, (clptr) nullptr, L.du, (lptr) nullptr);
            t1 = clock64();
            if (tid == 0) out[9] = t1 - t0;
            t0 = clock64();
            qpc::gT_times(d, qw, L, L.y, L.ya, L.du, L.ta);
            t1 = clock64();
            if (tid == 0) out[10] = t1 - t0;
        }
        t0 = clock64();
        qpc::gram<8>(d, qw, L);
        t1 = clock64();
        if (tid == 0) out[11] = t1 - t0;
    }
    // 5. a bare barrier and a reduce
    t0 = clock64();
    for (int i = 0; i < 10; ++i) __syncthreads();
    t1 = clock64();
    if (tid == 0) out[4] = (t1 - t0) / 10;
    t0 = clock64();
    double r = wg::reduce((double)tid, 1, L.red);
    t1 = clock64();
    if (tid == 0) { out[5] = t1 - t0; chk[32 * KT] = r; }
}

int main() {
    QPDims d{};
    d.N = 50; d.n = 60; d.m = 8; d.nz = 6; d.nU = 16; d.nX = 0; d.nXf = 0; d.po = 2; d.KT = 7; d.NK = 60; d.NPa = 80; d.ld = 81; d.diagD = 1; d.cond = 1;
    const int n = 16 * d.KT;
    std::vector<double> G((size_t)n * 3 * n), K((size_t)n * n, 0.0), v(n), x(n);
    srand(1);
    for (auto &e : G) e = (rand() / (double)RAND_MAX - 0.5);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { double s = i == j ? 1.0 : 0.0; for (int k = 0; k < 3 * n; ++k) s += G[(size_t)i * 3 * n + k] * G[(size_t)j * 3 * n + k]; K[(size_t)i * n + j] = s; }
    for (auto &e : v) e = rand() / (double)RAND_MAX;
    double *dK, *dv, *dchk; long long *dout;
    hipMalloc(&dK, K.size() * 8); hipMalloc(&dv, n * 8); hipMalloc(&dchk, (2 * n + 8) * 8); hipMalloc(&dout, 128); double *dG; hipMalloc(&dG, (size_t)d.N * d.m * 16 * d.KT * 8);
    hipMemcpy(dK, K.data(), K.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice);
    const size_t lds = qpc::lds_doubles(d, 512) * 8;
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) probe<<<1, 512, lds>>>(d, dK, dv, dout, dchk, dG);
    hipDeviceSynchronize();
    long long out[16]; std::vector<double> chk(2 * n + 8);
    hipMemcpy(out, dout, 128, hipMemcpyDeviceToHost); hipMemcpy(chk.data(), dchk, chk.size() * 8, hipMemcpyDeviceToHost);
    // check: K * chk[0..n) == v ; chk[n..2n) == K chk[0..n)
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += K[(size_t)i * n + j] * chk[j]; e1 = fmax(e1, fabs(s - v[i])); e2 = fmax(e2, fabs(s - chk[n + i])); }
    printf("lds %zu B; cycles: chol16 %lld, tile_cholesky(KT=%d) %lld (ok %lld), k_solve %lld, R^T R x %lld, barrier %lld, reduce %lld\n", lds, out[0], d.KT, out[1], out[7], out[2], out[3], out[4], out[5]);
    printf("G passes (N m = %d, ldG = %d, %zu KB): g_times %lld, gT_times %lld, gT_times(2 vectors) %lld; gram<8> %lld\n", d.N * d.m, 16 * d.KT, (size_t)d.N * d.m * 16 * d.KT * 8 / 1024, out[8], out[9], out[10], out[11]);
    printf("residuals: |K x - v| %.2e, |R^T R x - K x| %.2e\n", e1, e2);
    return 0;
}
