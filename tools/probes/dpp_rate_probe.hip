// Issue cost on ONE wave (gfx950) of the instructions a one-wave factorisation is made of: f64 FMA (independent / dependent),
// the same with a DPP row_newbcast operand, v_rsq_f64, v_mov_b64_dpp, a v_readlane pair feeding an FMA.  Shader clocks per instruction.
// Build: hipcc -O3 --offload-arch=gfx950 dpp_rate_probe.hip -o bin/dpp_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
__global__ __launch_bounds__(64) void probe(double *out, long long *t, int iters) {
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double s = a0 * 1e-9, o = 1e-9;
    long long c0, c1;
    int k = 0;
#define TIME(body, n) c0 = clock64(); for (int i = 0; i < iters; ++i) { body } c1 = clock64(); if (threadIdx.x == 0) t[k] = (c1 - c0) * 100 / ((long long)iters * (n)); ++k;
    // 0: 8 independent v_fma_f64
    TIME(asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o));, 8)
    // 1: 8 independent v_fmac_f64_dpp
    TIME(asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s), "v"(o));, 8)
    // 2: 8 dependent v_fma_f64
    TIME(asm volatile(REP8("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a0) : "v"(o), "v"(s));, 8)
    // 3: 8 dependent v_fmac_f64_dpp (accumulator chain)
    TIME(asm volatile(REP8("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a1) : "v"(s), "v"(o));, 8)
    // 4: 8 independent v_rsq_f64
    TIME(asm volatile("v_rsq_f64 %0, %8\n v_rsq_f64 %1, %8\n v_rsq_f64 %2, %8\n v_rsq_f64 %3, %8\n v_rsq_f64 %4, %8\n v_rsq_f64 %5, %8\n v_rsq_f64 %6, %8\n v_rsq_f64 %7, %8"
        : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(s));, 8)
    // 5: dependent v_rsq_f64 -> v_mul (latency of the pair)
    TIME(asm volatile(REP8("v_rsq_f64 %0, %0\n s_nop 0\n v_mul_f64 %0, %0, %1\n") : "+v"(a2) : "v"(o));, 8)
    // 6: 8 v_mov_b64_dpp independent
    TIME(asm volatile("v_mov_b64_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %2, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
        "v_mov_b64_dpp %4, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %5, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %6, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %7, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf"
        : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(s));, 8)
    // 7: readlane pair + fma (compiler form), 8 independent
    TIME(a0 = fma(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 1), __builtin_amdgcn_readlane(__double2loint(s), 1)), o, a0);
         a1 = fma(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 2), __builtin_amdgcn_readlane(__double2loint(s), 2)), o, a1);
         a2 = fma(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 3), __builtin_amdgcn_readlane(__double2loint(s), 3)), o, a2);
         a3 = fma(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 4), __builtin_amdgcn_readlane(__double2loint(s), 4)), o, a3);
         a4 = fma(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 5), __builtin_amdgcn_readlane(__double2loint(s), 5)), o, a4);
         a5 = fma(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 6), __builtin_amdgcn_readlane(__double2loint(s), 6)), o, a5);
         a6 = fma(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 7), __builtin_amdgcn_readlane(__double2loint(s), 7)), o, a6);
         a7 = fma(__hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 8), __builtin_amdgcn_readlane(__double2loint(s), 8)), o, a7);
         asm volatile("" : "+v"(s));, 8)
    // 8: dependent v_mul_f64 (no fma)
    TIME(asm volatile(REP8("v_mul_f64 %0, %0, %1\n") : "+v"(a3) : "v"(o));, 8)
    // 9: 8 independent v_mul_f64 with a 32-bit DPP-free VOP2 (v_add_f64)
    TIME(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));, 8)
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    {   // v_permlane16_swap_b32 as "lane ^ 16" (wg::xor16 of csrc/dev_la.h)
        const double v = 1000.0 + threadIdx.x;
        const unsigned lo = __double2loint(v), hi = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const bool odd = (threadIdx.x >> 4) & 1;
        out[64 + threadIdx.x] = __hiloint2double(odd ? b[0] : b[1], odd ? a[0] : a[1]);
    }
}

int main() {
    double *d; long long *t; hipMalloc(&d, 128 * 8); hipMalloc(&t, 16 * 8); hipMemset(d, 0, 64 * 8);
    long long h[16];
    const char *name[] = {"8 independent v_fma_f64", "8 independent v_fmac_f64_dpp", "dependent v_fma_f64", "dependent v_fmac_f64_dpp", "8 independent v_rsq_f64",
                          "dependent v_rsq_f64 + s_nop 0 + v_mul_f64 (per pair)", "8 independent v_mov_b64_dpp", "readlane pair + fma, 8 independent (per fma)", "dependent v_mul_f64", "8 independent v_add_f64"};
    for (int pass = 0; pass < 2; ++pass) {
        probe<<<1, 64>>>(d, t, 2000); hipDeviceSynchronize();
        hipMemcpy(h, t, 16 * 8, hipMemcpyDeviceToHost);
    }
    for (int k = 0; k < 10; ++k) printf("%-60s %6.2f clocks\n", name[k], h[k] / 100.0);
    double hx[64]; hipMemcpy(hx, d + 64, 64 * 8, hipMemcpyDeviceToHost);
    int okx = 1;
    for (int l = 0; l < 64; ++l) okx = okx && (hx[l] == 1000.0 + (l ^ 16));
    printf("v_permlane16_swap_b32 as lane ^ 16: %s (lane 0 <- %.0f, lane 16 <- %.0f, lane 40 <- %.0f)\n", okx ? "ok" : "WRONG", hx[0] - 1000, hx[16] - 1000, hx[40] - 1000);
    return !okx;
}
