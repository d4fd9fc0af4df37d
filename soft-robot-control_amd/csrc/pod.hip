// POD projection / lift / U^T M U on gfx950 (f64 MFMA 16x16x4, HBM-streaming).
//
// Reference arithmetic: sofacontrol/mor/pod.py:22-72 (numpy float64).  The snapshot matrix X is
// (B x n_f) row-major; the projection out = (X - 1 ref^T) U is a tall-skinny GEMM whose cost is the
// single read of X from HBM: 8*B*n_f bytes against 2*B*n_f*r flops (7.5 flop/B at r = 30), i.e.
// HBM-bound as long as the f64 matrix pipe stays >~65 % busy.  Layout decisions:
//   * X is consumed straight from HBM into MFMA A-operands: lane (row = l&15, kgrp = l>>4) loads 8
//     16-byte pieces of its row per 64-column chunk, the four k-groups of a row reading adjacent pieces
//     (64 contiguous bytes per row per load instruction, 512 B per row per chunk); the k-slot -> column
//     assignment (col = 64c + 8*(t>>1) + 2*kgrp + (t&1) for step t) is a permutation of the dot product
//     and needs no shuffle;
//   * U is pre-packed ONCE (srom_create) into the matching B-operand fragment order
//     Ufrag[chunk][t][ntile][lane], zero padded, so a fragment is one conflict-free 512 B LDS row;
//     chunks are staged through LDS and shared by the 4 waves of a workgroup (128 rows);
//   * the reference subtraction (x - x_ref) is done on the A operand before the MFMA (same op order as
//     numpy: subtract, then multiply-accumulate), x_ref chunk staged next to the U chunk;
//   * small B (closed-loop single vector, pod.py:51-52 via tpwl/controllers.py:96) uses split-K over
//     workgroups with a fixed-order second-stage reduction (deterministic, no atomics).
#include "common.h"
#include <vector>
#include <type_traits>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

#ifndef SRH_PROJ_KC
#define SRH_PROJ_KC 64
#endif
constexpr int KC = SRH_PROJ_KC;   // columns of X per chunk (a row is read in runs of 8 KC bytes)
constexpr int TSTEP = KC / 4;     // MFMA k-steps per chunk
#ifndef SRH_PROJ_MT
#define SRH_PROJ_MT 2
#endif
constexpr int MTP = SRH_PROJ_MT;      // M-tiles (16 rows) per wave in the projection
constexpr int ROWS_WG = 64 * MTP;     // rows of X per workgroup (4 waves x MTP M-tiles x 16)

struct ProjArgs {
    const double *X;      // (B x ldx)
    int64_t ldx;
    int64_t x_blk_off;    // column offset between the v and q blocks (SROM_X), else 0
    const double *ref0, *ref1;  // reference per block (may be null)
    const double *ufrag;  // packed basis
    double *out;          // (B x ldo)
    int64_t ldo;
    int64_t o_blk_off;
    double *partial;      // split-K workspace [ksplit][nblk][B][NT*16] or null
    int64_t B;
    int64_t n_f;
    int r;
    int nchunks, chunks_per_split;
    const double *Urows;  // UTMU only: the basis, row-major (n_f x r)
    const double *Xm[4];  // UTMU only: the matrices of one launch (blockIdx.z selects; srom_reduce_matrices_dev), Xm[0] == X
};

__global__ void pack_u_kernel(const double *__restrict__ U, int64_t n_f, int r, int NTF, int NQ, int nchunks,
                              double *__restrict__ ufrag) {
    // Ufrag[c][t][f][lane], row i = 64c + 8*(t>>1) + 2*(lane>>4) + (t&1) of U:
    //   f <  NTF : full 16-column tile f, column 16 f + (lane&15)            (B operand of v_mfma_f64_16x16x4)
    //   f >= NTF : 4-column tile q = f - NTF, column 16 NTF + 4 q + (lane&3), the same 4 x 4 block for each of the
    //              four row blocks (lane>>2)&3                                (B operand of v_mfma_f64_4x4x4, 4 blocks)
    const int NF = NTF + NQ;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)nchunks * TSTEP * NF * 64;
    if (idx >= total) return;
    int lane = idx & 63;
    int64_t rest = idx >> 6;
    int f = rest % NF;
    rest /= NF;
    int t = rest % TSTEP;
    int64_t c = rest / TSTEP;
    int64_t i = c * KC + 8 * (t >> 1) + 2 * (lane >> 4) + (t & 1);
    int j = f < NTF ? 16 * f + (lane & 15) : 16 * NTF + 4 * (f - NTF) + (lane & 3);
    ufrag[idx] = (i < n_f && j < r) ? U[i * r + j] : 0.0;
}

__global__ void pack_ut_kernel(const double *__restrict__ U, int64_t n_f, int r, int64_t ldu, int krows,
                               double *__restrict__ ut) {
    // Ut[k][16 + i] = U[i][k]: the basis transposed, 16 zero columns in front of and >= 16 behind every row and zero
    // rows up to the next multiple of four, so that a shifted 16-column window of the lift never needs a guard
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ldu * krows) return;
    const int k = (int)(idx / ldu);
    const int64_t i = idx % ldu - 16;
    ut[idx] = (k < r && i >= 0 && i < n_f) ? U[i * r + k] : 0.0;
}

// ------------------------------------------------------------------------------------ projection
// NTF full 16-column tiles + NQ 4-column tiles (r mod 16 <= 12: the remainder runs on v_mfma_f64_4x4x4, whose four
// 4 x 4 blocks take the SAME A operand layout -- lane = (row & 15, k-group) -- at a quarter of the issue time of a
// padded 16-column tile: measured 7.1 ns against 26.9 ns per instruction, tools/probes/mfma4_probe.hip; r = 36 costs
// 2.25 tiles instead of 3)
// UTMU: X is a square matrix M (B = n_f rows); instead of storing its tile of T = M U the workgroup multiplies it by
// the matching rows of the basis, P = U[rows]^T T (r x r), and stores that: U^T M U in ONE pass over M, the n_f x r
// intermediate never reaches HBM (mor/pod.py:62-64).  The r x r partials of all workgroups are summed by
// utmu_reduce_kernel in a fixed order.
template <int NTF, int NQ, bool HAS_REF, bool VEC2, bool UTMU = false>
__global__ __launch_bounds__(256) void proj_kernel(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NF = NTF + NQ;
    constexpr int NT = NTF > 0 ? NTF : 1, NQA = NQ > 0 ? NQ : 1;   // array extents
    constexpr int UCH = TSTEP * NF * 64;      // doubles of U fragments per chunk
    constexpr int BUF = UCH + KC;             // + reference chunk
    double *lds = reinterpret_cast<double *>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = blockIdx.z;
    const double *X = UTMU ? a.Xm[blk] : a.X + (int64_t)blk * a.x_blk_off;
    const double *ref = blk ? a.ref1 : a.ref0;
    const int c0 = blockIdx.y * a.chunks_per_split;
    const int c1 = min(c0 + a.chunks_per_split, a.nchunks);
    const int64_t rowbase = (int64_t)blockIdx.x * ROWS_WG + wave * (16 * MTP);
    const int lrow = lane & 15, kgrp = lane >> 4;

    d4 acc[MTP][NT];
    double accq[MTP][NQA];
#pragma unroll
    for (int mt = 0; mt < MTP; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < NQA; ++q) accq[mt][q] = 0.0;
    }

    const double *xrow[MTP];
#pragma unroll
    for (int mt = 0; mt < MTP; ++mt) {
        int64_t row = rowbase + mt * 16 + lrow;
        if (row >= a.B) row = a.B - 1;  // clamp: result rows beyond B are never stored
        xrow[mt] = X + row * a.ldx;
    }

    double xr[MTP][TSTEP];
    // register t of a lane holds column 64c + 8*(t>>1) + 2*kgrp + (t&1): the four k-groups of a row read
    // adjacent 16-byte pieces, so one load instruction covers 64 contiguous bytes of each of its 16 rows
    auto load_x = [&](int c, double (&dst)[MTP][TSTEP]) {
        const int64_t col0 = (int64_t)c * KC + 2 * kgrp;
        if ((int64_t)c * KC + KC <= a.n_f) {
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) {
                const double *p = xrow[mt] + col0;
                if (VEC2) {
#pragma unroll
                    for (int q = 0; q < TSTEP / 2; ++q) {
                        d2 v = *reinterpret_cast<const d2 *>(p + 8 * q);
                        dst[mt][2 * q] = v.x;
                        dst[mt][2 * q + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < TSTEP / 2; ++q) { dst[mt][2 * q] = p[8 * q]; dst[mt][2 * q + 1] = p[8 * q + 1]; }
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt)
#pragma unroll
                for (int q = 0; q < TSTEP / 2; ++q) {
                    const int64_t cc = col0 + 8 * q;
                    dst[mt][2 * q] = (cc < a.n_f) ? xrow[mt][cc] : 0.0;
                    dst[mt][2 * q + 1] = (cc + 1 < a.n_f) ? xrow[mt][cc + 1] : 0.0;
                }
        }
    };
    // stage chunk c of the packed basis (+ reference) into LDS buffer b: the global loads are issued BEFORE
    // the X loads of the same chunk and written to LDS only after the MFMAs of the current chunk, so the
    // wait for them (in-order vmcnt) never includes the X loads still in flight
    constexpr int UQ = UCH / 2 / 256;
    d2 ureg[UQ];
    double refreg = 0.0;
    auto stage_load = [&](int c) {
        const d2 *src = reinterpret_cast<const d2 *>(a.ufrag + (int64_t)c * UCH);
#pragma unroll
        for (int q = 0; q < UQ; ++q) ureg[q] = src[q * 256 + tid];
        if (HAS_REF && tid < KC) {
            int64_t i = (int64_t)c * KC + tid;
            refreg = (i < a.n_f) ? ref[i] : 0.0;
        }
    };
    auto stage_write = [&](int b) {
        d2 *dst = reinterpret_cast<d2 *>(lds + b * BUF);
#pragma unroll
        for (int q = 0; q < UQ; ++q) dst[q * 256 + tid] = ureg[q];
        if (HAS_REF && tid < KC) lds[b * BUF + UCH + tid] = refreg;
    };

    // UTMU: A operand of the epilogue product P = U[rows]^T T for this wave's first output tile, U[row0 + k][16 ti + i]
    // with lane = (i, k mod 4); all ROWS_WG / 4 loads are requested right after the main loop, before the T tile goes
    // to LDS (requesting them during the last chunk costs registers the streaming loop needs for two waves per SIMD)
    constexpr int LDPU = 16 * (NTF + (NQ ? 1 : 0)), NTTU = LDPU / 16;
    double ua[UTMU ? ROWS_WG / 4 : 1];
    auto load_ua = [&](int t) {
        const int ucol = 16 * (t / NTTU) + lrow;
        const int64_t row0 = (int64_t)blockIdx.x * ROWS_WG + kgrp;
#pragma unroll
        for (int q = 0; q < ROWS_WG / 4; ++q) {
            const int64_t row = row0 + 4 * q;
            ua[UTMU ? q : 0] = (row < a.B && ucol < a.r) ? a.Urows[row * a.r + ucol] : 0.0;
        }
    };

    if (c0 < c1) {
        stage_load(c0);
        load_x(c0, xr);
        stage_write(0);
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
        const int b = (c - c0) & 1;
        double xn[MTP][TSTEP];
        const bool more = (c + 1 < c1);
        if (more) {
            stage_load(c + 1);
            load_x(c + 1, xn);
        }
        const double *ub = lds + b * BUF;
        if (HAS_REF) {
            const double *rb = ub + UCH + 2 * kgrp;
#pragma unroll
            for (int q = 0; q < TSTEP; ++q) {
                double rv = rb[8 * (q >> 1) + (q & 1)];
#pragma unroll
                for (int mt = 0; mt < MTP; ++mt) xr[mt][q] -= rv;
            }
        }
#pragma unroll
        for (int t = 0; t < TSTEP; ++t) {
#pragma unroll
            for (int nt = 0; nt < NTF; ++nt) {
                double bv = ub[(t * NF + nt) * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < MTP; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[mt][t], bv, acc[mt][nt], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                double bv = ub[(t * NF + NTF + q) * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < MTP; ++mt)
                    accq[mt][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(xr[mt][t], bv, accq[mt][q], 0, 0, 0);
            }
        }
        if (more) stage_write(b ^ 1);
        __syncthreads();
        if (more) {
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt)
#pragma unroll
                for (int q = 0; q < TSTEP; ++q) xr[mt][q] = xn[mt][q];
        }
    }

    // D layout of v_mfma_f64_16x16x4: col = lane&15, row = (lane>>4) + 4*reg; of v_mfma_f64_4x4x4 (one value per
    // lane): col = lane&3, row = 4*((lane>>2)&3) + (lane>>4)
    const int col = lane & 15;
    constexpr int LDP = 16 * (NTF + (NQ ? 1 : 0));
    if (UTMU) {
        // T tile (ROWS_WG x LDP) into LDS -- the staging buffers are free after the last barrier of the loop -- with a
        // row pitch = 16 (mod 32) doubles: the four k-groups of a B-operand read then fall on disjoint banks
        constexpr int LDT = (LDP % 32 == 0) ? LDP + 16 : LDP;
        constexpr int NTT = LDP / 16;
        load_ua(wave);
#pragma unroll
        for (int mt = 0; mt < MTP; ++mt) {
            const int lr = wave * (16 * MTP) + mt * 16;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int nt = 0; nt < NTF; ++nt) lds[(lr + kgrp + 4 * reg) * LDT + 16 * nt + col] = acc[mt][nt][reg];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                lds[(lr + 4 * ((lane >> 2) & 3) + (lane >> 4)) * LDT + 16 * NTF + 4 * q + (lane & 3)] = accq[mt][q];
            if (NQ > 0 && NQ < 4) {              // columns of the last 16-wide tile that no 4-column tile covers
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (col >= 4 * NQ) lds[(lr + kgrp + 4 * reg) * LDT + 16 * NTF + col] = 0.0;
            }
        }
        __syncthreads();
        double *P = a.partial + (((int64_t)blk * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (LDP * LDP);
        for (int t = wave; t < NTT * NTT; t += 4) {
            const int ti = t / NTT, tj = t - ti * NTT;
            d4 pa = d4{0.0, 0.0, 0.0, 0.0};
            if (t != wave) load_ua(t);
#pragma unroll
            for (int q = 0; q < ROWS_WG / 4; ++q) {
                const double bv = lds[(4 * q + kgrp) * LDT + 16 * tj + lrow];                     // T[row][16 tj + j]
                pa = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[q], bv, pa, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) P[(16 * ti + kgrp + 4 * reg) * LDP + 16 * tj + col] = pa[reg];
        }
        return;
    }
    const bool direct = a.partial == nullptr;
    double *dst = direct ? a.out + (int64_t)blk * a.o_blk_off
                         : a.partial + (((int64_t)blockIdx.y * gridDim.z + blk) * a.B) * LDP;
    const int64_t ldd = direct ? a.ldo : LDP;
    const int jmax = direct ? a.r : LDP;
#pragma unroll
    for (int mt = 0; mt < MTP; ++mt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t row = rowbase + mt * 16 + kgrp + 4 * reg;
            if (row < a.B) {
#pragma unroll
                for (int nt = 0; nt < NTF; ++nt) {
                    const int j = 16 * nt + col;
                    if (j < jmax) dst[row * ldd + j] = acc[mt][nt][reg];
                }
            }
        }
        const int64_t rowq = rowbase + mt * 16 + 4 * ((lane >> 2) & 3) + (lane >> 4);
        if (rowq < a.B) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int j = 16 * NTF + 4 * q + (lane & 3);
                if (j < jmax) dst[rowq * ldd + j] = accq[mt][q];
            }
        }
    }
}

__global__ void splitk_reduce_kernel(const double *__restrict__ partial, int ksplit, int nblk, int64_t B,
                                     int ldp, int r, double *__restrict__ out, int64_t ldo,
                                     int64_t o_blk_off) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)nblk * B * r;
    if (idx >= total) return;
    int j = idx % r;
    int64_t row = (idx / r) % B;
    int blk = idx / ((int64_t)r * B);
    double s = 0.0;
    for (int k = 0; k < ksplit; ++k) s += partial[(((int64_t)k * nblk + blk) * B + row) * ldp + j];
    out[row * ldo + (int64_t)blk * o_blk_off + j] = s;
}

// Few outputs, many partials (one full state per simulation step: 2r outputs, ~77 K-slices): one wave per output,
// lanes over the K-slices, fixed butterfly order (deterministic).
__global__ void splitk_reduce_wave_kernel(const double *__restrict__ partial, int ksplit, int nblk, int64_t B,
                                          int ldp, int r, double *__restrict__ out, int64_t ldo,
                                          int64_t o_blk_off) {
    const int lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t total = (int64_t)nblk * B * r;
    if (idx >= total) return;
    const int j = idx % r;
    const int64_t row = (idx / r) % B;
    const int blk = idx / ((int64_t)r * B);
    double s = 0.0;
    for (int k = lane; k < ksplit; k += 64) s += partial[(((int64_t)k * nblk + blk) * B + row) * ldp + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[row * ldo + (int64_t)blk * o_blk_off + j] = s;
}

// U^T M U: out (r x r) = sum over the nwg workgroup partials (ldp x ldp each); one wave per output entry, lanes over
// the partials, fixed butterfly order (deterministic).  5.2 us for the 4 MB of 507 partials.  A coalesced two-level form in
// one launch (blocks over entry chunks x partial groups, the last-arriving block of a chunk adds the group sums) was
// measured at 14 us: the device-scope release / acquire around the arrival counter write back and invalidate L2; a
// sector-wise form (one wave per eight consecutive entries, whole 64-byte sectors per lane) measured the same 5.2 us:
// the time is the kernel boundary itself (write-back of the partials, launch, first misses), not the access pattern.
struct UtmuOuts { double *out[4]; };
__global__ void utmu_reduce_kernel(const double *__restrict__ partial_all, int nwg, int ldp, int r, UtmuOuts outs) {
    const double *__restrict__ partial = partial_all + (int64_t)blockIdx.y * nwg * ldp * ldp;      // blockIdx.y: matrix of the launch
    double *__restrict__ out = outs.out[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= r * r) return;
    const int i = idx / r, j = idx - i * r;
    double s = 0.0;
    for (int k0 = lane; k0 < nwg; k0 += 512) {          // eight loads in flight per lane (a rolled loop pays the latency per load)
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = k0 + 64 * q < nwg ? partial[((int64_t)(k0 + 64 * q) * ldp + i) * ldp + j] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += v[q];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[i * r + j] = s;
}

// ------------------------------------------------------------------------------------------ lift
// out = Xr U^T + 1 ref^T  (B x n_f, row pitch ldo): a pure write stream of 8 B n_f bytes.  What bounds it on gfx950 is
// the shape of the stores, not their number (tools/probes/lift_probe.hip): a 16-lane segment that straddles two
// 128-byte lines halves the rate of the memory pipe once loads share it (2.5 TB/s against 4.6 TB/s).  The pitch of
// the full-order arrays is 8 n_f bytes (n_f = 3 x nodes: never a multiple of 128), so the rows of one MFMA tile are
// chosen with equal alignment instead of consecutively:
//   * rows are split in c = 16 / gcd(ldo mod 16, 16) classes, class j = rows R with R = j (mod c): all rows of a
//     class start at the same offset a (in doubles) inside a 128-byte line;
//   * one wave owns MT tiles of one class -- rows R0 + c (16 mt + i) -- and slides over the columns in windows
//     [16 it - a, 16 it - a + 16): every store instruction writes four whole lines;
//   * the B operand of window `it` is read from the transposed, zero-padded basis Ut[k][16 + i] (four 128-byte
//     segments per load, any shift, no second packed copy); the first and last window are guarded, the interior
//     loop has no divergent branch, and the fragment of window it+1 is requested before the stores of window it
//     (loads and stores share the in-order vmcnt counter: a load issued after the stores would wait for their
//     write acknowledgements).
struct LiftArgs {
    const double *Xr;    // (B x ldr)
    int64_t ldr;
    int64_t r_blk_off;
    const double *ut;    // transposed padded basis, row pitch ldu
    int64_t ldu;
    const double *ref0, *ref1;   // null: no reference
    double *out;
    int64_t ldo;
    int64_t o_blk_off;
    int64_t B, n_f;
    int r;
    int classes;         // c
    int tiles_per_wg;
};

template <int KS, int MT>
__global__ __launch_bounds__(256) void lift_kernel(LiftArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = blockIdx.z;
    const double *Xr = a.Xr + (int64_t)blk * a.r_blk_off;
    const double *ref = blk ? a.ref1 : a.ref0;
    double *out = a.out + (int64_t)blk * a.o_blk_off;
    const int c = a.classes;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    const int64_t R0 = (item / c) * ((int64_t)c * 16 * MT) + item % c;       // rows R0 + c (16 mt + i)
    if (R0 >= a.B) return;
    const int lrow = lane & 15, kgrp = lane >> 4, col = lane & 15;
    const int al = (int)((((uint64_t)out >> 3) + (uint64_t)R0 * (uint64_t)a.ldo) & 15);   // doubles into the line

    double af[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int64_t row = R0 + (int64_t)c * (16 * mt + lrow);
        if (row >= a.B) row = a.B - 1;  // clamp: those rows are never stored
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const int j = 4 * t + kgrp;
            af[mt][t] = (j < a.r) ? Xr[row * a.ldr + j] : 0.0;
        }
    }
    const int64_t ntl = (a.n_f + al + 15) / 16;
    const int64_t t0 = (int64_t)blockIdx.y * a.tiles_per_wg;
    const int64_t t1 = min(t0 + (int64_t)a.tiles_per_wg, ntl);
    double bf[KS], rv = 0.0;
    auto fetch = [&](int64_t it, double (&dst)[KS], double &rdst) {
        const double *u = a.ut + 16 + 16 * it + col - al + (int64_t)kgrp * a.ldu;
#pragma unroll
        for (int t = 0; t < KS; ++t) dst[t] = u[(int64_t)(4 * t) * a.ldu];
        if (ref) {
            int64_t i = 16 * it + col - al;
            i = i < 0 ? 0 : (i >= a.n_f ? a.n_f - 1 : i);
            rdst = ref[i];
        }
    };
    auto tile = [&](int64_t it, const double (&b)[KS], double r, auto guard) {
        d4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[mt][t], b[t], acc[mt], 0, 0, 0);
        const int64_t i = 16 * it + col - al;
        // the lifted states are written once and not read back by this kernel: non-temporal stores (no L2 write-allocate)
        // took the kernel from 4.56 to 4.9 - 5.5 TB/s at r = 30, 4.34 -> 5.1 at r = 36 (tools/bench_lift.py, round 3; the
        // spread is between GPU boxes, the gain was measured on the same box)
        double *o = out + (R0 + (int64_t)c * kgrp) * a.ldo + i;
        if (!decltype(guard)::value) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) __builtin_nontemporal_store(acc[mt][reg] + r, &o[(int64_t)c * (4 * reg + 16 * mt) * a.ldo]);
        } else if (i >= 0 && i < a.n_f) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int64_t row = R0 + (int64_t)c * (kgrp + 4 * reg + 16 * mt);
                    if (row < a.B) o[(int64_t)c * (4 * reg + 16 * mt) * a.ldo] = acc[mt][reg] + r;
                }
        }
    };
    const bool rows_full = R0 + (int64_t)c * (16 * MT - 1) < a.B;                        // uniform per wave
    const int64_t tlast = rows_full ? min(t1, (a.n_f + al) / 16) : t0;                   // windows < tlast end inside the row
    int64_t it = t0;
    if (it < t1 && (it == 0 && al)) {            // first window starts in front of the row
        fetch(it, bf, rv);
        tile(it, bf, rv, std::true_type{});
        ++it;
    }
    if (it < tlast) {
        fetch(it, bf, rv);
        for (; it < tlast; ++it) {
            double bn[KS], rn = 0.0;
            fetch(it + 1 < tlast ? it + 1 : it, bn, rn);
            tile(it, bf, rv, std::false_type{});
#pragma unroll
            for (int t = 0; t < KS; ++t) bf[t] = bn[t];
            rv = rn;
        }
    }
    for (; it < t1; ++it) {
        fetch(it, bf, rv);
        tile(it, bf, rv, std::true_type{});
    }
}

// ------------------------------------------------------------------- small  out = U^T T  (r x c)
__global__ __launch_bounds__(256) void atb_partial_kernel(const double *__restrict__ U, int r,
                                                          const double *__restrict__ T, int64_t ldt,
                                                          int c, int64_t n_f, int rows_per_wg,
                                                          double *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *su = reinterpret_cast<double *>(smem);   // [64][r]
    double *st = su + 64 * r;                        // [64][c]
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t i1 = min(i0 + (int64_t)rows_per_wg, n_f);
    const int nout = r * c;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int64_t ib = i0; ib < i1; ib += 64) {
        int nr = (int)min((int64_t)64, i1 - ib);
        for (int e = tid; e < nr * r; e += 256) su[e] = U[ib * r + e];
        for (int e = tid; e < nr * c; e += 256) st[e] = T[(ib + e / c) * ldt + (e % c)];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int o = tid + q * 256;
            if (o < nout) {
                int aa = o / c, bb = o % c;
                double s = acc[q];
                // eight operand pairs in flight per trip (a rolled loop pays the LDS latency per row)
                for (int i = 0; i < nr; i += 8) {
                    double av[8], bv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool in = i + k < nr;
                        av[k] = in ? su[(i + k) * r + aa] : 0.0;
                        bv[k] = in ? st[(i + k) * c + bb] : 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) s = fma(av[k], bv[k], s);
                }
                acc[q] = s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        int o = tid + q * 256;
        if (o < nout) partial[(int64_t)blockIdx.x * nout + o] = acc[q];
    }
}

// one wave per output: lanes over the row-block partials, fixed butterfly order (deterministic)
__global__ void atb_reduce_kernel(const double *__restrict__ partial, int nblk, int nout,
                                  double *__restrict__ out, int c, int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (o >= nout) return;
    double s = 0.0;
    for (int k = lane; k < nblk; k += 64) s += partial[(int64_t)k * nout + o];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) out[(int64_t)(o / c) * ldo + (o % c)] = s;
}

}  // namespace

// ---------------------------------------------------------------------------------------- handle
struct srom {
    int64_t n_f = 0;
    int r = 0, NT = 0, NTF = 0, NQ = 0, nchunks = 0;   // NT = 16-column slots (partials), NTF full + NQ 4-column MFMA tiles
    int64_t ntiles = 0;
    srh::DevBuf U, q_ref, v_ref, ufrag, ut, work;
    int64_t ldu = 0;
    size_t work_bytes = 0;
    // per-simulation-step calls (one state in, one reduced state out) reuse these instead of paying
    // hipMalloc / hipFree and a pageable-memory copy per call: device staging + pinned host mirrors
    static constexpr size_t STAGE_BYTES = 1 << 20;
    void *st_dev_in = nullptr, *st_dev_out = nullptr, *st_host_in = nullptr, *st_host_out = nullptr;
    ~srom() {
        if (st_dev_in) (void)hipFree(st_dev_in);
        if (st_dev_out) (void)hipFree(st_dev_out);
        if (st_host_in) (void)hipHostFree(st_host_in);
        if (st_host_out) (void)hipHostFree(st_host_out);
    }
};

static int ensure_staging(srom *h) {
    if (h->st_dev_in) return SRH_OK;
    SRH_CHECK_HIP(hipMalloc(&h->st_dev_in, srom::STAGE_BYTES));
    SRH_CHECK_HIP(hipMalloc(&h->st_dev_out, srom::STAGE_BYTES));
    SRH_CHECK_HIP(hipHostMalloc(&h->st_host_in, srom::STAGE_BYTES, hipHostMallocDefault));
    SRH_CHECK_HIP(hipHostMalloc(&h->st_host_out, srom::STAGE_BYTES, hipHostMallocDefault));
    return SRH_OK;
}

// small host-pointer call through the staging buffers: fn(dev_in, dev_out) enqueues the kernels on stream 0
template <typename F>
static int staged_call(srom *h, const double *in, size_t in_bytes, double *out, size_t out_bytes, F fn) {
    int rc = ensure_staging(h);
    if (rc) return rc;
    memcpy(h->st_host_in, in, in_bytes);
    SRH_CHECK_HIP(hipMemcpyAsync(h->st_dev_in, h->st_host_in, in_bytes, hipMemcpyHostToDevice, nullptr));
    if ((rc = fn((const double *)h->st_dev_in, (double *)h->st_dev_out))) return rc;
    SRH_CHECK_HIP(hipMemcpyAsync(h->st_host_out, h->st_dev_out, out_bytes, hipMemcpyDeviceToHost, nullptr));
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    memcpy(out, h->st_host_out, out_bytes);
    return SRH_OK;
}

// Two-phase form of the staged projection for callers that overlap it with other work (sekf_step_projected):
// enqueue on `stream` (copy in, kernels, copy out), and after that stream has drained read the pinned mirror.
int srom_stage_project(srom *h, int which, const double *X, int64_t B, hipStream_t stream) {
    const int nblk = (which == SROM_X) ? 2 : 1;
    const size_t inb = sizeof(double) * B * nblk * h->n_f, outb = sizeof(double) * B * nblk * h->r;
    SRH_REQUIRE(inb <= srom::STAGE_BYTES && outb <= srom::STAGE_BYTES, "srom_stage_project: batch exceeds the staging buffer");
    int rc = ensure_staging(h);
    if (rc) return rc;
    memcpy(h->st_host_in, X, inb);
    SRH_CHECK_HIP(hipMemcpyAsync(h->st_dev_in, h->st_host_in, inb, hipMemcpyHostToDevice, stream));
    if ((rc = srom_project_dev(h, which, (const double *)h->st_dev_in, B, nblk * h->n_f, (double *)h->st_dev_out,
                               nblk * h->r, stream)))
        return rc;
    SRH_CHECK_HIP(hipMemcpyAsync(h->st_host_out, h->st_dev_out, outb, hipMemcpyDeviceToHost, stream));
    return SRH_OK;
}

int srom_stage_collect(srom *h, double *out, int64_t B, int which) {
    const int nblk = (which == SROM_X) ? 2 : 1;
    memcpy(out, h->st_host_out, sizeof(double) * B * nblk * h->r);
    return SRH_OK;
}

static int ensure_work(srom *h, size_t bytes) {
    if (bytes <= h->work_bytes) return SRH_OK;
    int rc = h->work.alloc(bytes);
    if (rc) return rc;
    h->work_bytes = bytes;
    return SRH_OK;
}

// split K across workgroups when there are too few row tiles to fill 256 CUs x 2
static int proj_ksplit(const srom *h, int64_t B, int nblk, int *chunks_per_split) {
    const int64_t rowtiles = srh::cdiv(B, ROWS_WG);
    int ksplit = 1;
    if (rowtiles * nblk < 512) ksplit = (int)std::min<int64_t>(h->nchunks, srh::cdiv(512, rowtiles * nblk));
    *chunks_per_split = (int)srh::cdiv(h->nchunks, ksplit);
    return (int)srh::cdiv(h->nchunks, *chunks_per_split);
}

// one launch of a projection kernel instantiation; dynamic LDS above the 64 KB default is requested once per instantiation
template <void (*KERN)(ProjArgs)>
static int launch_proj_kernel(const ProjArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    static bool raised = false;
    lds = srh::lds_request(lds);
    if (!raised && lds > 64 * 1024) {
        SRH_CHECK_HIP(hipFuncSetAttribute((const void *)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    KERN<<<grid, 256, lds, s>>>(a);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

template <int NTF, int NQ>
static int launch_utmu(const ProjArgs &a, bool vec2, dim3 grid, hipStream_t s) {
    constexpr int LDP = 16 * (NTF + (NQ ? 1 : 0)), LDT = (LDP % 32 == 0) ? LDP + 16 : LDP;
    size_t lds = std::max<size_t>(2 * (size_t)(TSTEP * (NTF + NQ) * 64 + KC), (size_t)ROWS_WG * LDT) * sizeof(double);
    return vec2 ? launch_proj_kernel<proj_kernel<NTF, NQ, false, true, true>>(a, grid, lds, s)
                : launch_proj_kernel<proj_kernel<NTF, NQ, false, false, true>>(a, grid, lds, s);
}

template <int NTF, int NQ>
static int launch_proj(const ProjArgs &a, bool has_ref, bool vec2, dim3 grid, hipStream_t s) {
    size_t lds = 2 * (size_t)(TSTEP * (NTF + NQ) * 64 + KC) * sizeof(double);
    if (has_ref)
        return vec2 ? launch_proj_kernel<proj_kernel<NTF, NQ, true, true>>(a, grid, lds, s)
                    : launch_proj_kernel<proj_kernel<NTF, NQ, true, false>>(a, grid, lds, s);
    return vec2 ? launch_proj_kernel<proj_kernel<NTF, NQ, false, true>>(a, grid, lds, s)
                : launch_proj_kernel<proj_kernel<NTF, NQ, false, false>>(a, grid, lds, s);
}

template <int KS>
static int launch_lift(const LiftArgs &a, int mt, dim3 grid, hipStream_t s) {
    if (mt == 4) lift_kernel<KS, 4><<<grid, 256, 0, s>>>(a);
    else lift_kernel<KS, 1><<<grid, 256, 0, s>>>(a);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

namespace {
// x = [v ; q] rows from reduced velocities / positions (utils.qv2x): a strided copy, one row per workgroup column
__global__ void qv2x_kernel(const double *__restrict__ q, int64_t ldq, const double *__restrict__ v, int64_t ldv, int64_t B, int r,
                            double *__restrict__ x, int64_t ldx) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * 2 * r) return;
    const int64_t b = e / (2 * r);
    const int j = (int)(e - b * 2 * r);
    x[b * ldx + j] = j < r ? (v ? v[b * ldv + j] : 0.0) : q[b * ldq + (j - r)];
}
}  // namespace

extern "C" {

int srom_qv2x_dev(const double *q_dev, int64_t ldq, const double *v_dev, int64_t ldv, int64_t B, int r, double *x_dev, int64_t ldx,
                  void *stream) {
    SRH_REQUIRE(q_dev && x_dev && B >= 0 && r >= 1 && ldq >= r && ldx >= 2 * r && (v_dev == nullptr || ldv >= r),
                "srom_qv2x_dev: bad argument");
    if (B == 0) return SRH_OK;
    qv2x_kernel<<<(unsigned)srh::cdiv(B * 2 * r, 256), 256, 0, (hipStream_t)stream>>>(q_dev, ldq, v_dev, ldv, B, r, x_dev, ldx);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}


int srom_create(srom_t **out, const double *U, int64_t n_f, int r, const double *q_ref,
                const double *v_ref) {
    SRH_REQUIRE(out && U, "srom_create: null argument");
    SRH_REQUIRE(n_f > 0 && r > 0 && r <= 64, "srom_create: need n_f > 0 and 0 < r <= 64 (got %lld, %d)",
                (long long)n_f, r);
    srom *h = new srom();
    h->n_f = n_f;
    h->r = r;
    h->NT = (r + 15) / 16;
    h->NTF = r / 16;
    h->NQ = (r % 16 + 3) / 4;
    if (h->NQ == 4) { h->NTF += 1; h->NQ = 0; }
    h->nchunks = (int)srh::cdiv(n_f, KC);
    h->ntiles = srh::cdiv(n_f, 16);
    int rc;
    std::vector<double> zeros;
    if (!q_ref || !v_ref) zeros.assign(n_f, 0.0);
    if ((rc = h->U.upload(U, sizeof(double) * n_f * r)) ||
        (rc = h->q_ref.upload(q_ref ? q_ref : zeros.data(), sizeof(double) * n_f)) ||
        (rc = h->v_ref.upload(v_ref ? v_ref : zeros.data(), sizeof(double) * n_f)) ||
        (rc = h->ufrag.alloc(sizeof(double) * (size_t)h->nchunks * TSTEP * (h->NTF + h->NQ) * 64)) ||
        (rc = h->ut.alloc(sizeof(double) * (size_t)(16 * h->ntiles + 32) * 4 * ((r + 3) / 4)))) {
        delete h;
        return rc;
    }
    int64_t tot = (int64_t)h->nchunks * TSTEP * (h->NTF + h->NQ) * 64;
    pack_u_kernel<<<(unsigned)srh::cdiv(tot, 256), 256>>>(h->U.as<double>(), n_f, r, h->NTF, h->NQ, h->nchunks,
                                                         h->ufrag.as<double>());
    h->ldu = 16 * h->ntiles + 32;
    const int krows = 4 * ((r + 3) / 4);
    pack_ut_kernel<<<(unsigned)srh::cdiv(h->ldu * krows, 256), 256>>>(h->U.as<double>(), n_f, r, h->ldu, krows,
                                                                      h->ut.as<double>());
    hipError_t e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        srh::set_error("srom_create: packing kernels failed: %s", hipGetErrorString(e));
        delete h;
        return SRH_EHIP;
    }
    *out = h;
    return SRH_OK;
}

int srom_destroy(srom_t *h) {
    delete h;
    return SRH_OK;
}

int srom_dims(const srom_t *h, int64_t *n_f, int *r) {
    SRH_REQUIRE(h, "srom_dims: null handle");
    if (n_f) *n_f = h->n_f;
    if (r) *r = h->r;
    return SRH_OK;
}

int srom_project_dev(srom_t *h, int which, const double *X, int64_t B, int64_t ldx, double *out,
                     int64_t ldo, void *stream) {
    SRH_REQUIRE(h && X && out, "srom_project_dev: null argument");
    SRH_REQUIRE(which >= SROM_Q && which <= SROM_RAW, "srom_project_dev: Must specify vector type");
    SRH_REQUIRE(B >= 0, "srom_project_dev: negative batch");
    if (B == 0) return SRH_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (which == SROM_X) ? 2 : 1;
    SRH_REQUIRE(ldx >= nblk * h->n_f && ldo >= nblk * h->r, "srom_project_dev: leading dimension too small");
    ProjArgs a{};
    a.X = X;
    a.ldx = ldx;
    a.x_blk_off = h->n_f;
    a.ufrag = h->ufrag.as<double>();
    a.out = out;
    a.ldo = ldo;
    a.o_blk_off = h->r;
    a.B = B;
    a.n_f = h->n_f;
    a.r = h->r;
    a.nchunks = h->nchunks;
    bool has_ref = which != SROM_RAW;
    if (which == SROM_Q) a.ref0 = h->q_ref.as<double>();
    if (which == SROM_V) a.ref0 = h->v_ref.as<double>();
    if (which == SROM_X) { a.ref0 = h->v_ref.as<double>(); a.ref1 = h->q_ref.as<double>(); }
    const bool vec2 = ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && (ldx % 2 == 0) && (h->n_f % 2 == 0);
    const int64_t rowtiles = srh::cdiv(B, ROWS_WG);
    int ksplit = proj_ksplit(h, B, nblk, &a.chunks_per_split);
    const int ldp = h->NT * 16;
    if (ksplit > 1) {
        int rc = ensure_work(h, sizeof(double) * (size_t)ksplit * nblk * B * ldp);
        if (rc) return rc;
        a.partial = h->work.as<double>();
    }
    dim3 grid((unsigned)rowtiles, (unsigned)ksplit, (unsigned)nblk);
    int rc;
    switch (4 * h->NTF + h->NQ) {
#define SRH_PROJ_CASE(F, Q) case 4 * F + Q: rc = launch_proj<F, Q>(a, has_ref, vec2, grid, s); break;
        SRH_PROJ_CASE(0, 1) SRH_PROJ_CASE(0, 2) SRH_PROJ_CASE(0, 3)
        SRH_PROJ_CASE(1, 0) SRH_PROJ_CASE(1, 1) SRH_PROJ_CASE(1, 2) SRH_PROJ_CASE(1, 3)
        SRH_PROJ_CASE(2, 0) SRH_PROJ_CASE(2, 1) SRH_PROJ_CASE(2, 2) SRH_PROJ_CASE(2, 3)
        SRH_PROJ_CASE(3, 0) SRH_PROJ_CASE(3, 1) SRH_PROJ_CASE(3, 2) SRH_PROJ_CASE(3, 3)
        SRH_PROJ_CASE(4, 0)
#undef SRH_PROJ_CASE
        default:
            srh::set_error("srom_project_dev: no kernel for r = %d", h->r);
            return SRH_EINVAL;
    }
    if (rc) return rc;
    if (ksplit > 1) {
        int64_t total = (int64_t)nblk * B * h->r;
        if (total <= 4096 && ksplit >= 16)
            splitk_reduce_wave_kernel<<<(unsigned)srh::cdiv(total, 4), 256, 0, s>>>(a.partial, ksplit, nblk, B, ldp,
                                                                                    h->r, out, ldo, h->r);
        else
            splitk_reduce_kernel<<<(unsigned)srh::cdiv(total, 256), 256, 0, s>>>(
                a.partial, ksplit, nblk, B, ldp, h->r, out, ldo, h->r);
        SRH_CHECK_HIP(hipGetLastError());
    }
    return SRH_OK;
}

int srom_project(srom_t *h, int which, const double *X, int64_t B, double *out) {
    SRH_REQUIRE(h && X && out, "srom_project: null argument");
    SRH_REQUIRE(which >= SROM_Q && which <= SROM_RAW, "srom_project: Must specify vector type");
    if (B == 0) return SRH_OK;
    const int nblk = (which == SROM_X) ? 2 : 1;
    const size_t inb = sizeof(double) * B * nblk * h->n_f, outb = sizeof(double) * B * nblk * h->r;
    if (inb <= srom::STAGE_BYTES && outb <= srom::STAGE_BYTES)
        return staged_call(h, X, inb, out, outb, [&](const double *di, double *dout) {
            return srom_project_dev(h, which, di, B, nblk * h->n_f, dout, nblk * h->r, nullptr);
        });
    srh::DevBuf dX, dO;
    int rc;
    if ((rc = dX.upload(X, sizeof(double) * B * nblk * h->n_f))) return rc;
    if ((rc = dO.alloc(sizeof(double) * B * nblk * h->r))) return rc;
    if ((rc = srom_project_dev(h, which, dX.as<double>(), B, nblk * h->n_f, dO.as<double>(), nblk * h->r, nullptr)))
        return rc;
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    return dO.download(out, sizeof(double) * B * nblk * h->r);
}

int srom_lift_dev(srom_t *h, int which, const double *Xr, int64_t B, int64_t ldr, double *out,
                  int64_t ldo, void *stream) {
    SRH_REQUIRE(h && Xr && out, "srom_lift_dev: null argument");
    SRH_REQUIRE(which >= SROM_Q && which <= SROM_RAW, "srom_lift_dev: Must specify vector type");
    if (B == 0) return SRH_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (which == SROM_X) ? 2 : 1;
    SRH_REQUIRE(ldr >= nblk * h->r && ldo >= nblk * h->n_f, "srom_lift_dev: leading dimension too small");
    LiftArgs a{};
    a.Xr = Xr;
    a.ldr = ldr;
    a.r_blk_off = h->r;
    a.ut = h->ut.as<double>();
    a.ldu = h->ldu;
    a.out = out;
    a.ldo = ldo;
    a.o_blk_off = h->n_f;
    a.B = B;
    a.n_f = h->n_f;
    a.r = h->r;
    if (which == SROM_Q) a.ref0 = h->q_ref.as<double>();
    if (which == SROM_V) a.ref0 = h->v_ref.as<double>();
    if (which == SROM_X) { a.ref0 = h->v_ref.as<double>(); a.ref1 = h->q_ref.as<double>(); }
    // alignment classes of the row starts (the block offset n_f of SROM_X shifts every row alike)
    const int m16 = (int)(ldo & 15);
    int g = 16;
    while (m16 % g) g >>= 1;
    a.classes = m16 ? 16 / g : 1;
    // four tiles per wave share one basis fragment; a handful of rows (closed loop: one state) takes one tile
    const int mt = (B >= (int64_t)a.classes * 64) ? 4 : 1;
    const int64_t rows_sb = (int64_t)a.classes * 16 * mt;               // rows of one super-block = `classes` items
    const int64_t items = srh::cdiv(B, rows_sb) * a.classes;
    const int64_t wgs = srh::cdiv(items, 4);
    const int64_t ntl = h->ntiles + 1;                                   // windows incl. the shifted first one
    int64_t ysplit = std::max<int64_t>(1, std::min<int64_t>(ntl, srh::cdiv(512, wgs * nblk)));
    a.tiles_per_wg = (int)srh::cdiv(ntl, ysplit);
    ysplit = srh::cdiv(ntl, a.tiles_per_wg);
    dim3 grid((unsigned)wgs, (unsigned)ysplit, (unsigned)nblk);
    switch ((h->r + 3) / 4) {
#define SRH_LIFT_CASE(K) case K: return launch_lift<K>(a, mt, grid, s);
        SRH_LIFT_CASE(1) SRH_LIFT_CASE(2) SRH_LIFT_CASE(3) SRH_LIFT_CASE(4) SRH_LIFT_CASE(5) SRH_LIFT_CASE(6)
        SRH_LIFT_CASE(7) SRH_LIFT_CASE(8) SRH_LIFT_CASE(9) SRH_LIFT_CASE(10) SRH_LIFT_CASE(11) SRH_LIFT_CASE(12)
        SRH_LIFT_CASE(13) SRH_LIFT_CASE(14) SRH_LIFT_CASE(15)
#undef SRH_LIFT_CASE
        default: return launch_lift<16>(a, mt, grid, s);
    }
}

int srom_lift(srom_t *h, int which, const double *Xr, int64_t B, double *out) {
    SRH_REQUIRE(h && Xr && out, "srom_lift: null argument");
    SRH_REQUIRE(which >= SROM_Q && which <= SROM_RAW, "srom_lift: Must specify vector type");
    if (B == 0) return SRH_OK;
    const int nblk = (which == SROM_X) ? 2 : 1;
    const size_t inb = sizeof(double) * B * nblk * h->r, outb = sizeof(double) * B * nblk * h->n_f;
    if (inb <= srom::STAGE_BYTES && outb <= srom::STAGE_BYTES)
        return staged_call(h, Xr, inb, out, outb, [&](const double *di, double *dout) {
            return srom_lift_dev(h, which, di, B, nblk * h->r, dout, nblk * h->n_f, nullptr);
        });
    srh::DevBuf dX, dO;
    int rc;
    if ((rc = dX.upload(Xr, sizeof(double) * B * nblk * h->r))) return rc;
    if ((rc = dO.alloc(sizeof(double) * B * nblk * h->n_f))) return rc;
    if ((rc = srom_lift_dev(h, which, dX.as<double>(), B, nblk * h->r, dO.as<double>(), nblk * h->n_f, nullptr)))
        return rc;
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    return dO.download(out, sizeof(double) * B * nblk * h->n_f);
}

// out (r x c) = U^T T for T (n_f x c), c <= 64 per pass
static int atb_dev(srom *h, const double *T, int64_t ldt, int c, double *out, int64_t ldo,
                   double *partial, hipStream_t s) {
    const int rows_per_wg = 64;
    const int nblk = (int)srh::cdiv(h->n_f, rows_per_wg);
    const int nout = h->r * c;
    size_t lds = sizeof(double) * 64 * (size_t)(h->r + c);
    atb_partial_kernel<<<nblk, 256, lds, s>>>(h->U.as<double>(), h->r, T, ldt, c, h->n_f, rows_per_wg, partial);
    SRH_CHECK_HIP(hipGetLastError());
    atb_reduce_kernel<<<(unsigned)srh::cdiv(nout, 4), 256, 0, s>>>(partial, nblk, nout, out, c, ldo);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

// U^T M_i U for up to four n_f x n_f matrices in ONE launch pair: every workgroup turns its tile of T = M_i U into an r x r partial
// U[rows]^T T (proj_kernel<..., UTMU>, blockIdx.z = i), then one reduction launch sums the partials of every matrix in a fixed order.
// The caller of the reference reduces K, D, M and S of the same linearisation point back to back (tpwl/tpwl_utils.py:96-103): four
// matrices per launch pay the launch ramp, the epilogue every workgroup reaches at the same moment and the second launch once.
static int utmu_one_pass(srom *h, const double *const *Ms, int count, double *const *outs, hipStream_t s) {
    SRH_REQUIRE(count >= 1 && count <= 4, "utmu_one_pass: 1..4 matrices per launch");
    ProjArgs a{};
    a.X = Ms[0]; a.ldx = h->n_f; a.ufrag = h->ufrag.as<double>(); a.B = h->n_f; a.n_f = h->n_f; a.r = h->r;
    a.nchunks = h->nchunks; a.Urows = h->U.as<double>();
    bool vec2 = h->n_f % 2 == 0;
    for (int i = 0; i < 4; ++i) {
        a.Xm[i] = Ms[i < count ? i : 0];
        vec2 = vec2 && ((reinterpret_cast<uintptr_t>(a.Xm[i]) & 15) == 0);
    }
    const int64_t rowtiles = srh::cdiv(h->n_f, ROWS_WG);
    int ksplit = proj_ksplit(h, h->n_f, 1, &a.chunks_per_split);
    if (count > 1) {
        // several matrices per launch: as many K-slices as keep ALL workgroups of the launch resident at once (two per CU: 512) --
        // measured at n_f = 4884, r = 30, four matrices: 3 slices (468 workgroups) 183 us, 4 (624) 236 us, 6 (936) 196 us, 13 (2028) 222 us
        const int want = (int)std::max<int64_t>(1, std::min<int64_t>(h->nchunks, 512 / (rowtiles * count)));
        a.chunks_per_split = (int)srh::cdiv(h->nchunks, want);
        ksplit = (int)srh::cdiv(h->nchunks, a.chunks_per_split);
    }
    if (const char *e = getenv("SRH_UTMU_KSPLIT")) {       // A/B: K-slices per row tile (default: ~512 workgroups)
        const int want = std::max(1, std::min(atoi(e), h->nchunks));
        a.chunks_per_split = (int)srh::cdiv(h->nchunks, want);
        ksplit = (int)srh::cdiv(h->nchunks, a.chunks_per_split);
    }
    const int ldp = 16 * (h->NTF + (h->NQ ? 1 : 0));
    const int nwg = (int)(rowtiles * ksplit);
    int rc = ensure_work(h, sizeof(double) * (size_t)count * nwg * ldp * ldp);
    if (rc) return rc;
    a.partial = h->work.as<double>();
    dim3 grid((unsigned)rowtiles, (unsigned)ksplit, (unsigned)count);
    switch (4 * h->NTF + h->NQ) {
#define SRH_UTMU_CASE(F, Q) case 4 * F + Q: rc = launch_utmu<F, Q>(a, vec2, grid, s); break;
        SRH_UTMU_CASE(0, 1) SRH_UTMU_CASE(0, 2) SRH_UTMU_CASE(0, 3)
        SRH_UTMU_CASE(1, 0) SRH_UTMU_CASE(1, 1) SRH_UTMU_CASE(1, 2) SRH_UTMU_CASE(1, 3)
        SRH_UTMU_CASE(2, 0) SRH_UTMU_CASE(2, 1) SRH_UTMU_CASE(2, 2) SRH_UTMU_CASE(2, 3)
        SRH_UTMU_CASE(3, 0) SRH_UTMU_CASE(3, 1) SRH_UTMU_CASE(3, 2) SRH_UTMU_CASE(3, 3)
        SRH_UTMU_CASE(4, 0)
#undef SRH_UTMU_CASE
        default:
            srh::set_error("srom_reduce_matrix_dev: no kernel for r = %d", h->r);
            return SRH_EINVAL;
    }
    if (rc) return rc;
    UtmuOuts uo{};
    for (int i = 0; i < 4; ++i) uo.out[i] = outs[i < count ? i : 0];
    utmu_reduce_kernel<<<dim3((unsigned)srh::cdiv(h->r * h->r, 4), (unsigned)count), 256, 0, s>>>(a.partial, nwg, ldp, h->r, uo);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

int srom_reduce_matrix_dev(srom_t *h, const double *M, int64_t ncols, int left, int right, double *out,
                           void *stream) {
    SRH_REQUIRE(h && M && out, "srom_reduce_matrix_dev: null argument");
    SRH_REQUIRE(ncols > 0, "srom_reduce_matrix_dev: ncols must be positive");
    hipStream_t s = (hipStream_t)stream;
    const bool both = (left && right) || (!left && !right);
    const int nblk_atb = (int)srh::cdiv(h->n_f, 64);
    if (both || right) SRH_REQUIRE(ncols == h->n_f, "srom_reduce_matrix_dev: M must be n_f x n_f");
    if (right && !both) {
        // M U : rows of M projected without a reference
        return srom_project_dev(h, SROM_RAW, M, h->n_f, ncols, out, h->r, stream);
    }
    if (both) {
        // one pass over M: every workgroup turns its tile of T = M U into an r x r partial U[rows]^T T; then one small
        // reduction over the partials (SRH_UTMU_TWO_PASS: the older T-through-HBM path, kept for A/B timing)
        if (!getenv("SRH_UTMU_TWO_PASS")) {
            const double *Ms[1] = {M};
            double *outs[1] = {out};
            return utmu_one_pass(h, Ms, 1, outs, s);
        }
        // T = M U (n_f x r) streamed once from HBM and stored, then U^T T (r x r)
        size_t tbytes = sizeof(double) * (size_t)h->n_f * h->r;
        size_t pbytes = sizeof(double) * (size_t)nblk_atb * h->r * h->r;
        // workspace: [split-K partials of the projection | T | atb partials]
        int cps;
        const int ks = proj_ksplit(h, h->n_f, 1, &cps);
        size_t proj_ws = ks > 1 ? sizeof(double) * (size_t)ks * h->n_f * h->NT * 16 : 0;
        int rc = ensure_work(h, proj_ws + tbytes + pbytes);
        if (rc) return rc;
        // the projection may use the front of the workspace for its split-K partials
        double *T = reinterpret_cast<double *>(h->work.as<char>() + proj_ws);
        double *part = reinterpret_cast<double *>(h->work.as<char>() + proj_ws + tbytes);
        if ((rc = srom_project_dev(h, SROM_RAW, M, h->n_f, ncols, T, h->r, stream))) return rc;
        return atb_dev(h, T, h->r, h->r, out, h->r, part, s);
    }
    // left only: U^T M (r x ncols), column panels of <= 64
    size_t pbytes = sizeof(double) * (size_t)nblk_atb * h->r * 64;
    int rc = ensure_work(h, pbytes);
    if (rc) return rc;
    for (int64_t c0 = 0; c0 < ncols; c0 += 64) {
        int c = (int)std::min<int64_t>(64, ncols - c0);
        if ((rc = atb_dev(h, M + c0, ncols, c, out + c0, ncols, h->work.as<double>(), s))) return rc;
    }
    return SRH_OK;
}

int srom_reduce_matrix(srom_t *h, const double *M, int64_t ncols, int left, int right, double *out) {
    SRH_REQUIRE(h && M && out, "srom_reduce_matrix: null argument");
    const bool both = (left && right) || (!left && !right);
    srh::DevBuf dM, dO;
    int rc;
    if ((rc = dM.upload(M, sizeof(double) * h->n_f * ncols))) return rc;
    size_t obytes = both ? sizeof(double) * h->r * h->r
                         : (left ? sizeof(double) * h->r * ncols : sizeof(double) * h->n_f * h->r);
    if ((rc = dO.alloc(obytes))) return rc;
    if ((rc = srom_reduce_matrix_dev(h, dM.as<double>(), ncols, left, right, dO.as<double>(), nullptr))) return rc;
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    return dO.download(out, obytes);
}

/* POD.compute_RO_matrix(left = right = True) for `count` n_f x n_f matrices at once: the K, D, M, S of one TPWL point
 * (tpwl/tpwl_utils.py:96-103).  Groups of four share a launch pair (utmu_one_pass). */
int srom_reduce_matrices_dev(srom_t *h, const double *const *M_dev, int count, double *const *out_dev, void *stream) {
    SRH_REQUIRE(h && M_dev && out_dev && count >= 1, "srom_reduce_matrices_dev: null argument / no matrix");
    for (int i = 0; i < count; ++i) SRH_REQUIRE(M_dev[i] && out_dev[i], "srom_reduce_matrices_dev: null matrix %d", i);
    if (getenv("SRH_UTMU_TWO_PASS")) {
        for (int i = 0; i < count; ++i) { int rc = srom_reduce_matrix_dev(h, M_dev[i], h->n_f, 1, 1, out_dev[i], stream); if (rc) return rc; }
        return SRH_OK;
    }
    for (int i0 = 0; i0 < count; i0 += 4) {
        int rc = utmu_one_pass(h, M_dev + i0, std::min(4, count - i0), out_dev + i0, (hipStream_t)stream);
        if (rc) return rc;
    }
    return SRH_OK;
}

int srom_reduce_matrices(srom_t *h, const double *const *M, int count, double *const *out) {
    SRH_REQUIRE(h && M && out && count >= 1, "srom_reduce_matrices: null argument / no matrix");
    const size_t mb = sizeof(double) * (size_t)h->n_f * h->n_f, ob = sizeof(double) * (size_t)h->r * h->r;
    std::vector<srh::DevBuf> dM((size_t)count), dO((size_t)count);
    std::vector<const double *> mp((size_t)count);
    std::vector<double *> op((size_t)count);
    int rc;
    for (int i = 0; i < count; ++i) {
        SRH_REQUIRE(M[i] && out[i], "srom_reduce_matrices: null matrix %d", i);
        if ((rc = dM[i].upload(M[i], mb)) || (rc = dO[i].alloc(ob))) return rc;
        mp[i] = dM[i].as<double>(); op[i] = dO[i].as<double>();
    }
    if ((rc = srom_reduce_matrices_dev(h, mp.data(), count, op.data(), nullptr))) return rc;
    SRH_CHECK_HIP(hipStreamSynchronize(nullptr));
    for (int i = 0; i < count; ++i) if ((rc = dO[i].download(out[i], ob))) return rc;
    return SRH_OK;
}

}  // extern "C"
