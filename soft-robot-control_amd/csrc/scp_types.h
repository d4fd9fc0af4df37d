// Kernel argument blocks shared by the fused kernels (scp.hip, gusto.hip) and the lean condensed kernels (lean.hip), and
// the launch entry points of the latter (lean.hip is its own translation unit: the variants compile in parallel).
#pragma once
#include "scp_host.h"

struct GustoPar {
    double delta0, omega0, rho, beta_fail, gamma_fail, epsilon, omega_max, convg_thresh, dt;
    int max_iters, max_trace;
    int warm_across;                    // 1: the first QP of a solve starts from the minimiser / multipliers the rollout's PREVIOUS solve left
                                        // (the reference's `warm_start=True`: its cvxpy problem keeps the solver state between solves, locp.py:181)
    int poison_warm;                    // test knobs of the lean kernel, read when the plan is created.  bit 0 (SRH_LEAN_POISON_WARM=1): every warm-started
                                        // QP fails and is repeated cold; bits 4.. (SRH_LEAN_FORCE_HANDOVER=k): k + 1, SCP iteration k is handed to the fused kernel
                                        // bit 1 (SRH_GUSTO_TRACE_QIT=1): trace slot 3 = interior-point iterations; bit 2 (SRH_LEAN_SERIAL_WAVE=1): the half-size
                                        // lean workgroup picks its serial wave by its wave slot (ql::serial_wave_pick; measured: see DESIGN.md section 15)
    int warm_full;                      // fused kernel, OFF unless SRH_GUSTO_WARM_FULL=1 at plan creation: a full (trust-region-active) QP behind a rejected
                                        // step starts from the previous one's (u, s, lambda).  Built and measured in round 6: the
                                        // uncapped tail's full QPs keep their 18-40 interior-point iterations -- the active set of the slack rows moves with
                                        // every omega x 5 -- so the default stays the cold start
};

struct GustoBatch {
    const double *x0, *u_init, *x_init, *z, *zf, *ud;
    const double *fs;                   // 1/|f_char| (n)
    double *xopt, *uopt, *zopt;
    int32_t *iters, *status;
    double *trace;
    double *work;                       // per problem: [qp work | xk | uk | acc | ints | resume record | condensed block]
    size_t work_stride;
    const int32_t *order;               // workgroup -> rollout (longest expected solve first), or null
    int32_t *last_iters;                // SCP iterations of this solve per rollout: the next solve's dispatch key
    int mode;                           // fused kernel: 0 = solve every rollout, 2 = only those a lean launch handed over
    int32_t *handed_over;               // lean kernel: counts the rollouts it hands to the fused kernel (or null)
    double *Jopt;                       // per rollout: LOCP optimal value of the solution returned (the last accepted step), or null
    int host_args;                      // x0, z, zf, ud point into pinned host memory (zero-copy solve): the kernels keep copies in the work block
};

struct LocpBatch {
    const double *Ad, *AdT, *Bd, *BdT, *dd;     // (batch x N x ...)
    const double *x0, *xk, *delta, *omega, *z, *zf, *ud;
    double *x, *u, *s, *J;
    int32_t *status, *iters;
    double *work;
    size_t work_stride;
    double *dbg;
    int only_pending;                   // fused kernel: 1 = only the problems a lean launch left with status LEAN_PENDING
    int32_t *handed_over;               // lean kernel: counts the QPs it hands to the fused kernel (or null)
};

namespace {

constexpr int LEAN_PENDING = -77;       // status of a QP / rollout the lean kernel hands to the fused kernel
constexpr int GUSTO_REC = 10;           // doubles of the resume record behind the SCP loop's index arrays ([8]: 1.0 = the work block holds a
                                        // converged QP of the previous solve: GustoPar::warm_across)

// offsets (doubles) of the SCP loop's own arrays inside a rollout's work block
struct GustoWork { size_t xk, uk, acc, idx, rec, x0c, zc, zfc, udc, end; };
__host__ __device__ inline GustoWork gusto_work(const QPDims &d) {
    GustoWork g;
    const size_t N = d.N, n = d.n, m = d.m;
    g.xk = qp_work_doubles(d);
    g.uk = g.xk + (N + 1) * n;
    g.acc = g.uk + N * m;
    g.idx = g.acc + 2 * N;
    g.rec = g.idx + (2 * N + 1) / 2;
    g.x0c = g.rec + GUSTO_REC;                  // copies of x0, the targets and the desired inputs: the arguments of a zero-copy solve sit in
    g.zc = g.x0c + n;                           // pinned HOST memory (gusto.hip) and every QP / interior-point iteration reads them
    g.zfc = g.zc + (N + 1) * d.nz;              // (GustoBatch::host_args; the short-horizon lean kernels copy x0 and z always)
    g.udc = g.zfc + d.nz;
    g.end = g.udc + N * m;
    return g;
}

}  // namespace

// lean.hip
int lean_select(const QPDims &d, int args[6]);          // index into the instantiation list (-1: none) + its template arguments
int lean_prepare(int variant, size_t lds);
int lean_launch_gusto(int variant, const QPDims &d, const QPConst &c, const TpwlDev &T, const GustoPar &par, const GustoBatch &b, unsigned grid,
                      size_t lds, hipStream_t stream);
int lean_launch_locp(int variant, const QPDims &d, const QPConst &c, const LocpBatch &b, unsigned grid, size_t lds, hipStream_t stream);
