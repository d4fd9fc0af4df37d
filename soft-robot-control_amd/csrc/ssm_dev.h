// Device view of an SSM (spectral submanifold) polynomial reduced model and workgroup-cooperative helpers.
// Reference: sofacontrol/SSM/ssm.py -- get_poly_basis 158-164, maps 167-178, Jacobians 198-235,
// discretize_dynamics 279-301, update_dynamics 331-333.
#pragma once
#include "dev_la.h"
#include <type_traits>

struct SsmDev {
    int n, m, no;              // reduced state, input, observed dimension
    int nr, ns;                // monomial counts: rom basis (n variables), ssm basis (no variables)
    const int *er, *es;        // exponent tables (nr x n), (ns x no)
    // evaluation tables per basis (host-built, ssm.hip): parent monomial / multiplied variable of every monomial
    // (phi_j = phi_parent * x_var; parent < 0: phi_j = x_var), index of the monomial e_j - 1_i for the derivatives
    // (-1: zero, -2: the constant 1), first index of every degree (order + 1 offsets)
    const int *pr, *vr, *dmr, *lvr, *ps, *vs, *dms, *lvs;
    int order_r, order_s;
    cgptr R, Bc;               // r_coeff (n x nr), B (n x m)              continuous reduced dynamics
    cgptr Rd, Bd;              // rd_coeff, Bd or null                      discrete reduced dynamics
    cgptr Wc;                  // w_coeff (no x ns)  reduced -> observed (needs n == no)
    cgptr Vc;                  // v_coeff (n x ns)   observed -> reduced
    cgptr z_ref;               // (no)
    cgptr H;                   // bookkeeping performance matrix (no x n) (zeros unless the user sets it)
};

// discretisation modes of sssm_linearize / rollout (ssm.py:279-301 and the `discrete` flag 212-218)
enum { SSM_CONT = 0, SSM_FE = 1, SSM_BE = 2, SSM_BIL = 3, SSM_DISCRETE_MAP = 4 };

namespace ssm {

// phi_j(x) = prod_i x_i^e_ji and, if D != null, D[j][i] = d phi_j / d x_i    (all threads; ends with a sync)
// Degree by degree: phi_j = phi_parent(j) * x_var(j) (one multiply per monomial); the derivative of a monomial is
// e_ji times the monomial with exponents e_j - 1_i, looked up in the table -- O(n_mon * dim) instead of
// O(n_mon * dim^2 * order) for the direct products.
__device__ inline void basis(const int *__restrict__ ex, const int *__restrict__ par, const int *__restrict__ var,
                             const int *__restrict__ dm, const int *__restrict__ lv, int order, int nmon, int dim, clptr x,
                             lptr phi, lptr D) {
    for (int d = 0; d < order; ++d) {
        const int j0 = lv[d], j1 = lv[d + 1];
        for (int j = j0 + SRH_TID; j < j1; j += blockDim.x) {
            const int pj = par[j];
            phi[j] = (pj < 0 ? 1.0 : phi[pj]) * x[var[j]];
        }
        __syncthreads();
    }
    if (D != nullptr) {
        for (int e = SRH_TID; e < nmon * dim; e += blockDim.x) {
            const int q = dm[e];
            D[e] = q == -1 ? 0.0 : (double)ex[e] * (q == -2 ? 1.0 : phi[q]);
        }
        __syncthreads();
    }
}

// sum_k a[k * sa] * b[k * sb] with eight operand pairs in flight per trip (a rolled load -> fma chain pays the
// L2 / LDS latency of every element); a, b in any address space
template <typename AP, typename BP>
__device__ __forceinline__ double dot8(AP a, int sa, BP b, int sb, int K) {
    double acc = 0.0;
    for (int k = 0; k < K; k += 8) {
        double av[8], bv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool in = k + q < K;
            av[q] = in ? a[(size_t)(k + q) * sa] : 0.0;
            bv[q] = in ? b[(size_t)(k + q) * sb] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fma(av[q], bv[q], acc);
    }
    return acc;
}

// In-place Gauss-Jordan inverse with partial pivoting of the n x n matrix M (LDS, leading dimension ld);
// Minv (LDS, ld) receives the inverse.  piv: LDS int scratch (2).  All threads; ends with a sync.
__device__ inline void inverse(lptr M, lptr Minv, int n, int ld, liptr piv) {
    const int tid = SRH_TID, nt = blockDim.x;
    for (int e = tid; e < n * n; e += nt) Minv[(e / n) * ld + e % n] = (e / n == e % n) ? 1.0 : 0.0;
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        if (tid == 0) {
            int p = k;
            double best = fabs(M[k * ld + k]);
            for (int i = k + 1; i < n; ++i) {
                const double v = fabs(M[i * ld + k]);
                if (v > best) { best = v; p = i; }
            }
            piv[0] = p;
        }
        __syncthreads();
        const int p = piv[0];
        if (p != k) {
            for (int j = tid; j < n; j += nt) {
                double t = M[k * ld + j]; M[k * ld + j] = M[p * ld + j]; M[p * ld + j] = t;
                t = Minv[k * ld + j]; Minv[k * ld + j] = Minv[p * ld + j]; Minv[p * ld + j] = t;
            }
            __syncthreads();
        }
        const double d = M[k * ld + k];
        __syncthreads();
        for (int j = tid; j < n; j += nt) { M[k * ld + j] = M[k * ld + j] / d; Minv[k * ld + j] = Minv[k * ld + j] / d; }
        __syncthreads();
        // eliminate column k from the other rows: thread per (row, col); the multiplier is read before any write
        // of its column (column k of M is only written by the j == k threads after the barrier below)
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e % n;
            if (i == k) continue;
            const double f = M[i * ld + k];
            Minv[i * ld + j] = fma(-f, Minv[k * ld + j], Minv[i * ld + j]);
            if (j != k) M[i * ld + j] = fma(-f, M[k * ld + j], M[i * ld + j]);
        }
        __syncthreads();
        for (int i = tid; i < n; i += nt) if (i != k) M[i * ld + k] = 0.0;
        __syncthreads();
    }
}

// The same elimination executed by ONE wave (n <= 64): the steps of `inverse` are separated by workgroup barriers -- six per
// pivot, ~450 clocks each with eight waves, 54 k clocks per backward-Euler discretisation of a 10 x 10 model (two inverses), 70 %
// of an iLQR forward step on the C3 shape -- while a 10 x 10 matrix has no work for more than one wave anyway.  Here the phases
// are ordered by the wave's own LDS queue (in order per wave) plus a fence for the compiler; every element goes through the same
// operations in the same order as in `inverse` (bit-identical results), and two matrices can be inverted side by side on two
// waves.  Pivot: first maximum of |M[i][k]|, i >= k, as the sequential scan of `inverse`.
__device__ inline void inverse_wave(lptr M, lptr Minv, int n, int ld) {
    const int lane = SRH_TID & 63;
    auto wsync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (int e = lane; e < n * n; e += 64) Minv[(e / n) * ld + e % n] = (e / n == e % n) ? 1.0 : 0.0;
    wsync();
    if (n * n <= 256) {
        // Round 5: ONE read phase and ONE write phase per pivot.  The three steps of `inverse` -- exchange rows k and p, divide
        // the pivot row by d, eliminate column k from the other rows -- read only OLD values when every lane gathers, for each of
        // its entries (i, j): the pivot row's M[p][j] and Minv[p][j], the multiplier and its own entries from the row that the
        // exchange puts in place i (row p for i = k, row k for i = p), and d.  The pivot row's new entries M[p][j] / d are formed
        // by every lane that needs them (the same IEEE division `inverse` performs once per column: same bits), the other rows
        // by the same FMAs.  ~0.9 k clocks per pivot instead of ~2 k (three LDS round trips and three fences less).
        auto run = [&](auto EP_) {
            constexpr int EP = decltype(EP_)::value;
            for (int k = 0; k < n; ++k) {
                const bool cand = lane >= k && lane < n;
                const double v = cand ? fabs(M[lane * ld + k]) : -1.0;
                const double mx = wg::wave_max(v);
                const unsigned long long hit = __ballot(cand && v == mx);
                const int p = hit ? __builtin_amdgcn_readfirstlane(__ffsll((long long)hit) - 1) : k;
                const double d = M[p * ld + k];
                double mp[EP], ip[EP], om[EP], oi[EP], f[EP];
#pragma unroll
                for (int q = 0; q < EP; ++q) {
                    const int e = lane + 64 * q, ec = e < n * n ? e : 0, i2 = ec / n, j2 = ec - i2 * n;
                    const int is = i2 == k ? p : (i2 == p ? k : i2);          // the row that stands in place i2 after the exchange
                    mp[q] = M[p * ld + j2]; ip[q] = Minv[p * ld + j2];
                    om[q] = M[is * ld + j2]; oi[q] = Minv[is * ld + j2]; f[q] = M[is * ld + k];
                }
                wsync();
#pragma unroll
                for (int q = 0; q < EP; ++q) {
                    const int e = lane + 64 * q, i2 = e / n, j2 = e - i2 * n;
                    if (e < n * n) {
                        const double rk = mp[q] / d, ri = ip[q] / d;
                        if (i2 == k) {
                            M[i2 * ld + j2] = rk; Minv[i2 * ld + j2] = ri;
                        } else {
                            Minv[i2 * ld + j2] = fma(-f[q], ri, oi[q]);
                            M[i2 * ld + j2] = j2 != k ? fma(-f[q], rk, om[q]) : 0.0;
                        }
                    }
                }
                wsync();
            }
        };
        if (n * n <= 64) run(std::integral_constant<int, 1>{});
        else if (n * n <= 128) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 4>{});
        return;
    }
    for (int k = 0; k < n; ++k) {
        const bool cand = lane >= k && lane < n;
        const double v = cand ? fabs(M[lane * ld + k]) : -1.0;
        const double mx = wg::wave_max(v);
        const unsigned long long hit = __ballot(cand && v == mx);
        const int p = hit ? __builtin_amdgcn_readfirstlane(__ffsll((long long)hit) - 1) : k;
        const double d = M[p * ld + k];
        for (int jc = lane; jc < n; jc += 64) {                       // n <= 64: one trip
            const double mp = M[p * ld + jc], ip = Minv[p * ld + jc], mk = M[k * ld + jc], ik = Minv[k * ld + jc];
            wsync();
            M[k * ld + jc] = mp / d; Minv[k * ld + jc] = ip / d;
            if (p != k) { M[p * ld + jc] = mk; Minv[p * ld + jc] = ik; }
        }
        wsync();
        for (int e = lane; e < n * n; e += 64) {
            const int i2 = e / n, j2 = e % n;
            if (i2 == k) continue;
            const double f = M[i2 * ld + k];
            Minv[i2 * ld + j2] = fma(-f, Minv[k * ld + j2], Minv[i2 * ld + j2]);
            if (j2 != k) M[i2 * ld + j2] = fma(-f, M[k * ld + j2], M[i2 * ld + j2]);
        }
        wsync();
        for (int i2 = lane; i2 < n; i2 += 64) if (i2 != k) M[i2 * ld + k] = 0.0;
        wsync();
    }
}

struct Work {
    lptr phi;      // max(nr, ns)
    lptr D;        // max(nr * n, ns * no)
    lptr M1, M2, M3, M4;   // n x ld each (discretisation scratch)
    lptr f;        // n
    liptr piv;
#ifdef SRH_PROFILE
    long long *prof;   // discretize(): 6 lap counters of the calling thread (or null)
#endif
};

__host__ __device__ inline size_t work_doubles(int n, int m, int no, int nr, int ns) {
    const int ld = n | 1;
    const size_t nb = (size_t)(nr > ns ? nr : ns);
    const size_t nd = (size_t)nr * n > (size_t)ns * no ? (size_t)nr * n : (size_t)ns * no;
    return nb + nd + 4 * (size_t)n + 4 * (size_t)n * ld + n + 4;      // (+ 4 n: the compact derivative table of jacobians_l may exceed nr n by 3 n)
}

__device__ inline void carve(Work &w, lptr base, const SsmDev &S) {
    const int ld = S.n | 1;
    const size_t nb = (size_t)(S.nr > S.ns ? S.nr : S.ns);
    const size_t nd = (size_t)S.nr * S.n > (size_t)S.ns * S.no ? (size_t)S.nr * S.n : (size_t)S.ns * S.no;
    w.phi = base; w.D = w.phi + nb; w.M1 = w.D + nd + 4 * (size_t)S.n; w.M2 = w.M1 + (size_t)S.n * ld; w.M3 = w.M2 + (size_t)S.n * ld;
    w.M4 = w.M3 + (size_t)S.n * ld;
    w.f = w.M4 + (size_t)S.n * ld;
    w.piv = (liptr)(w.f + S.n + 2);
#ifdef SRH_PROFILE
    w.prof = nullptr;
#endif
}

__device__ inline void discretize(const SsmDev &S, int mode, double dt, Work &w, lptr A, int lda, lptr Bm, lptr d);

// (A, B, d) of ssm.py:198-218 at (x, u) [LDS]: continuous Jacobians of f = R phi(x) + B u, affine remainder
// d = f - A x - B u, then discretised per `mode`.  A (n x lda), Bm (n x m), d (n) in LDS.  Ends with a sync.
__device__ inline void linearize(const SsmDev &S, int mode, double dt, clptr x, clptr u, Work &w, lptr A, int lda,
                                 lptr Bm, lptr d) {
    const int n = S.n, m = S.m, tid = SRH_TID, nt = blockDim.x;
    const bool dm = mode == SSM_DISCRETE_MAP;
    cgptr Rc = dm ? S.Rd : S.R, Bg = dm ? S.Bd : S.Bc;
    basis(S.er, S.pr, S.vr, S.dmr, S.lvr, S.order_r, S.nr, n, x, w.phi, w.D);
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e % n;
        A[i * lda + j] = dot8(Rc + (size_t)i * S.nr, 1, w.D + j, n, S.nr);
    }
    for (int e = tid; e < n * m; e += nt) Bm[e] = Bg[e];
    for (int i = tid; i < n; i += nt) {
        double t = 0.0;
        for (int k = 0; k < m; ++k) t = fma(Bg[i * m + k], u[k], t);
        w.f[i] = dot8(Rc + (size_t)i * S.nr, 1, w.phi, 1, S.nr) + t;
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        double ax = 0.0, bu = 0.0;
        for (int k = 0; k < n; ++k) ax = fma(A[i * lda + k], x[k], ax);
        for (int k = 0; k < m; ++k) bu = fma(Bm[i * m + k], u[k], bu);
        d[i] = w.f[i] - ax - bu;
    }
    __syncthreads();
    discretize(S, mode, dt, w, A, lda, Bm, d);
}

// (A, B, d) continuous -> discrete per `mode` (ssm.py:279-301), in place.  Ends with a sync.
__device__ inline void discretize(const SsmDev &S, int mode, double dt, Work &w, lptr A, int lda, lptr Bm, lptr d) {
    const int n = S.n, m = S.m, tid = SRH_TID, nt = blockDim.x;
    if (mode == SSM_CONT || mode == SSM_DISCRETE_MAP) return;
    if (mode == SSM_FE) {                                       // I + dt A, dt B, dt d
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e % n;
            A[i * lda + j] = (i == j ? 1.0 : 0.0) + dt * A[i * lda + j];
        }
        for (int e = tid; e < n * m; e += nt) Bm[e] = dt * Bm[e];
        for (int e = tid; e < n; e += nt) d[e] = dt * d[e];
        __syncthreads();
        return;
    }
    // be:  A_d = inv(I - dt A);  bil: A_d = (I + dt/2 A) inv(I - dt/2 A);  sep = inv(A) (A_d - I)
#ifdef SRH_PROFILE
    long long dl_ = clock64();
#define SSM_DLAP(i) do { if (w.prof) { const long long now_ = clock64(); w.prof[i] += now_ - dl_; dl_ = now_; } } while (0)
#else
#define SSM_DLAP(i) ((void)0)
#endif
    const int ld = n | 1;
    const double h = mode == SSM_BE ? dt : 0.5 * dt;
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e % n;
        w.M1[i * ld + j] = (i == j ? 1.0 : 0.0) - h * A[i * lda + j];
        w.M3[i * ld + j] = A[i * lda + j];
    }
    __syncthreads();
    SSM_DLAP(0);
    // M2 = inv(I - h A) and M4 = inv(A_c): independent, one wave each (a single-wave workgroup does them in turn)
    if (n <= 64) {
        const int wave = __builtin_amdgcn_readfirstlane(SRH_TID >> 6), nw = (blockDim.x + 63) >> 6;
        if (wave == 0) inverse_wave(w.M1, w.M2, n, ld);
        if (wave == (nw > 1 ? 1 : 0)) inverse_wave(w.M3, w.M4, n, ld);
        __syncthreads();
    } else {
        inverse(w.M1, w.M2, n, ld, w.piv);
        inverse(w.M3, w.M4, n, ld, w.piv);
    }
    SSM_DLAP(1);
    if (mode == SSM_BIL) {
        for (int e = tid; e < n * n; e += nt) {
            const int i = e / n, j = e % n;
            double s = 0.0;
            for (int k = 0; k < n; ++k) s = fma((i == k ? 1.0 : 0.0) + h * A[i * lda + k], w.M2[k * ld + j], s);
            w.M1[i * ld + j] = s;
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += nt) w.M2[(e / n) * ld + e % n] = w.M1[(e / n) * ld + e % n];
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += nt) {                      // M3 = sep = inv(A_c) (A_d - I)
        const int i = e / n, j = e % n;
        double s = 0.0;
        for (int k = 0; k < n; ++k) s = fma(w.M4[i * ld + k], w.M2[k * ld + j] - (k == j ? 1.0 : 0.0), s);
        w.M3[i * ld + j] = s;
    }
    __syncthreads();
    SSM_DLAP(2);
    for (int e = tid; e < n * n; e += nt) A[(e / n) * lda + e % n] = w.M2[(e / n) * ld + e % n];
    for (int i = tid; i < n; i += nt) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s = fma(w.M3[i * ld + k], d[k], s);
        w.f[i] = s;
    }
    for (int e = tid; e < n * m; e += nt) {
        const int i = e / m, j = e % m;
        double s = 0.0;
        for (int k = 0; k < n; ++k) s = fma(w.M3[i * ld + k], Bm[k * m + j], s);
        w.M1[i * ld + j] = s;                                    // m <= n assumed for the scratch (checked on host)
    }
    __syncthreads();
    SSM_DLAP(3);
    for (int e = tid; e < n * m; e += nt) Bm[e] = w.M1[(e / m) * ld + e % m];
    for (int e = tid; e < n; e += nt) d[e] = w.f[e];
    __syncthreads();
    SSM_DLAP(4);
}

// z = C_map(x) = W phi_s(x) (no z_ref); optional observer Jacobian Hj = W Dphi_s (no x n) and c = z - Hj x
// (ssm.py:220-235).  Ends with a sync.
__device__ inline void observe(const SsmDev &S, clptr x, Work &w, lptr z, lptr Hj, lptr c) {
    const int n = S.n, no = S.no, tid = SRH_TID, nt = blockDim.x;
    basis(S.es, S.ps, S.vs, S.dms, S.lvs, S.order_s, S.ns, no, x, w.phi, Hj != nullptr ? w.D : (lptr) nullptr);
    for (int i = tid; i < no; i += nt) {
        z[i] = dot8(S.Wc + (size_t)i * S.ns, 1, w.phi, 1, S.ns);
    }
    if (Hj != nullptr) {
        for (int e = tid; e < no * n; e += nt) {
            const int i = e / n, j = e % n;
            Hj[e] = dot8(S.Wc + (size_t)i * S.ns, 1, w.D + j, no, S.ns);
        }
    }
    __syncthreads();
    if (Hj != nullptr && c != nullptr) {
        for (int i = tid; i < no; i += nt) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s = fma(Hj[i * n + k], x[k], s);
            c[i] = z[i] - s;
        }
        __syncthreads();
    }
}

}  // namespace ssm

// ---------------------------------------------------------------------------------------------------------------
// LDS-resident form for kernels that evaluate the model at every step of a loop (iLQR forward pass on an SSM model:
// N steps per pass).  Measured on the C3 shape (n = 10, 285 monomials): with the coefficient rows and exponent tables
// read from L2 inside 100 latency-bound dot products the linearisation took ~80 k clocks per step; staged once per
// kernel in LDS and spread over all threads it is a few thousand.
struct SsmLds {
    lptr R;                    // (n x nr) coefficients of the map in use (r_coeff, or rd_coeff for the discrete map)
    lptr W;                    // (no x ns) w_coeff
    lptr Bg;                   // (n x m) input matrix of the map in use
    liptr er, dmr, pr, vr;     // rom basis tables: exponents (nr x n), derivative monomial (nr x n), parent, variable (nr)
    liptr es, dms, ps, vs;     // ssm basis tables
    int lvr[8], lvs[8];        // first monomial of every degree (order <= 6)
    // compact derivative lists of the rom basis (jacobians_l): column j of D = d phi / d x has a structural non-zero only where
    // x_j divides the monomial (cubic basis in 10 variables: 66 of 285).  List (j, g), g = k mod 4, holds those monomials k in
    // increasing order, jcap slots each (padded): jk = k, jq = index of the monomial e_k - 1_j (-2: the constant 1, -1: padding),
    // je = the exponent e_kj.
    liptr jk, jq, je;
    int jcap;
};

namespace ssm {

__host__ __device__ inline size_t lds_tab_doubles(int n, int no, int nr, int ns, int jcap = 0) {
    const size_t ints = 2 * (size_t)nr * n + 2 * (size_t)nr + 2 * (size_t)ns * no + 2 * (size_t)ns + 3 * (size_t)4 * n * jcap;
    return (size_t)n * nr + (size_t)no * ns + (size_t)n * 16 + (ints + 1) / 2 + 8;       // m <= 16
}
// slots per compact derivative list: the longest list (j, k mod 4) of the basis with exponent table E (nr x n); host side
inline int jacobian_list_cap(const int *E, int nr, int n) {
    int cap = 0;
    for (int j = 0; j < n; ++j)
        for (int g = 0; g < 4; ++g) {
            int c = 0;
            for (int k = g; k < nr; k += 4) c += E[(size_t)k * n + j] > 0;
            cap = c > cap ? c : cap;
        }
    return cap;
}

// copy the tables (all threads; ends with a sync)
__device__ inline void stage(SsmLds &T, lptr base, const SsmDev &S, bool discrete_map, int jcap = 0) {
    const int n = S.n, no = S.no, nr = S.nr, ns = S.ns, tid = SRH_TID, nt = blockDim.x;
    T.jcap = jcap;
    T.R = base; T.W = T.R + (size_t)n * nr; T.Bg = T.W + (size_t)no * ns;
    liptr ip = (liptr)(T.Bg + (size_t)n * 16);
    T.er = ip; T.dmr = T.er + nr * n; T.pr = T.dmr + nr * n; T.vr = T.pr + nr;
    T.es = T.vr + nr; T.dms = T.es + ns * no; T.ps = T.dms + ns * no; T.vs = T.ps + ns;
    T.jk = T.vs + ns; T.jq = T.jk + 4 * n * jcap; T.je = T.jq + 4 * n * jcap;
    cgptr Rc = discrete_map ? S.Rd : S.R;
    for (int e = tid; e < n * nr; e += nt) T.R[e] = Rc[e];
    for (int e = tid; e < no * ns; e += nt) T.W[e] = S.Wc[e];
    for (int e = tid; e < n * S.m; e += nt) T.Bg[e] = (discrete_map ? S.Bd : S.Bc)[e];
    for (int e = tid; e < nr * n; e += nt) { T.er[e] = S.er[e]; T.dmr[e] = S.dmr[e]; }
    for (int e = tid; e < nr; e += nt) { T.pr[e] = S.pr[e]; T.vr[e] = S.vr[e]; }
    for (int e = tid; e < ns * no; e += nt) { T.es[e] = S.es[e]; T.dms[e] = S.dms[e]; }
    for (int e = tid; e < ns; e += nt) { T.ps[e] = S.ps[e]; T.vs[e] = S.vs[e]; }
    for (int q = 0; q < 8; ++q) { T.lvr[q] = q <= S.order_r ? S.lvr[q] : 0; T.lvs[q] = q <= S.order_s ? S.lvs[q] : 0; }
    __syncthreads();
    if (jcap > 0) {                                              // one thread per list (j, g): scan its monomials in increasing order
        for (int l = tid; l < 4 * n; l += nt) {
            const int j = l >> 2, gq = l & 3;
            int c = 0;
            for (int k = gq; k < nr; k += 4) {
                const int e = T.er[k * n + j];
                if (e > 0 && c < jcap) { T.jk[l * jcap + c] = k; T.jq[l * jcap + c] = T.dmr[k * n + j]; T.je[l * jcap + c] = e; ++c; }
            }
            for (; c < jcap; ++c) { T.jk[l * jcap + c] = 0; T.jq[l * jcap + c] = -1; T.je[l * jcap + c] = 0; }
        }
        __syncthreads();
    }
}

// monomials (and derivative table D) from LDS tables; as `basis`
__device__ inline void basis_l(const liptr ex, const liptr par, const liptr var, const liptr dm, const int *lv, int order, int nmon,
                               int dim, clptr x, lptr phi, lptr D) {
    for (int d = 0; d < order; ++d) {
        const int j0 = lv[d], j1 = lv[d + 1];
        for (int j = j0 + SRH_TID; j < j1; j += blockDim.x) {
            const int pj = par[j];
            phi[j] = (pj < 0 ? 1.0 : phi[pj]) * x[var[j]];
        }
        __syncthreads();
    }
    if (D != nullptr) {
        const int tot = nmon * dim, nt = blockDim.x;
        for (int e0 = SRH_TID; e0 < tot; e0 += 4 * nt) {          // four entries per trip: their table look-ups overlap
            int q[4], ee[4];
            double pv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int e = e0 + c * nt; const bool in = e < tot; q[c] = in ? dm[e] : -1; ee[c] = in ? ex[e] : 0; }
#pragma unroll
            for (int c = 0; c < 4; ++c) pv[c] = q[c] >= 0 ? phi[q[c]] : 1.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int e = e0 + c * nt; if (e < tot) D[e] = q[c] == -1 ? 0.0 : (double)ee[c] * pv[c]; }
        }
        __syncthreads();
    }
}

// continuous Jacobians + affine remainder from the LDS tables (the front part of `linearize`); ends with a sync
__device__ inline void jacobians_l(const SsmDev &S, const SsmLds &T, bool dm, clptr x, clptr u, Work &w, lptr A, int lda, lptr Bm,
                                   lptr d) {
    const int n = S.n, m = S.m, nr = S.nr, tid = SRH_TID, nt = blockDim.x;
    clptr Bg = T.Bg;
    if (T.jcap > 0) {
        // Round 5: only the structural non-zeros of d phi / d x.  A[i][j] = sum_k R[i][k] D[k][j] runs over the monomials that
        // contain x_j (66 of 285 for the cubic basis in 10 variables), in the SAME order on the same four lanes as the dense sum
        // below (k = g mod 4, increasing): the skipped terms are exact zeros, the result is bit-identical.  f[i] = sum_k R[i][k]
        // phi[k] stays dense (285 terms) and takes 16 lanes per row instead of 4 so that it does not outlast the A sums; its
        // summation order differs from the dense form's (rounding-level difference in f).
        const int cap = T.jcap, tot = 4 * n * cap;
        basis_l(T.er, T.pr, T.vr, T.dmr, T.lvr, S.order_r, nr, n, x, w.phi, (lptr) nullptr);
        lptr DC = w.D;                                           // compact derivative values, list (j, g) x slot
        for (int e0 = tid; e0 < tot; e0 += 2 * nt) {             // two entries per trip: the look-ups overlap
            int qa[2], ea[2];
            double pv[2];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) { const int e = e0 + c2 * nt; const bool in = e < tot; qa[c2] = in ? T.jq[e] : -1; ea[c2] = in ? T.je[e] : 0; }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) pv[c2] = qa[c2] >= 0 ? w.phi[qa[c2]] : 1.0;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) { const int e = e0 + c2 * nt; if (e < tot) DC[e] = qa[c2] == -1 ? 0.0 : (double)ea[c2] * pv[c2]; }
        }
        {   // f[i]: 16 lanes per row
            const int g16 = tid & 15;
            for (int i0 = 0; i0 < n; i0 += nt / 16) {
                const int i = i0 + (tid >> 4);
                double acc = 0.0;
                if (i < n) {
                    clptr r = T.R + (size_t)i * nr;
                    for (int k0 = g16; k0 < nr; k0 += 128) {
                        double av[8], bv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) { const int k = k0 + 16 * q; const bool in = k < nr; av[q] = in ? r[k] : 0.0; bv[q] = in ? w.phi[k] : 0.0; }
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc = fma(av[q], bv[q], acc);
                    }
                }
                acc = wg::group_sum<16>(acc);
                if (g16 == 0 && i < n) w.f[i] = acc;
            }
        }
        __syncthreads();
        const int g4 = tid & 3;
        for (int o0 = 0; o0 < n * n; o0 += nt / 4) {             // uniform trip count
            const int o = o0 + (tid >> 2);
            const bool live = o < n * n;
            const int i = live ? o / n : 0, j = live ? o - i * n : 0;
            double acc = 0.0;
            if (live) {
                clptr r = T.R + (size_t)i * nr;
                const int lb = (4 * j + g4) * cap;
                for (int t0 = 0; t0 < cap; t0 += 8) {
                    double av[8], bv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const int t = t0 + q; const bool in = t < cap; av[q] = in ? r[T.jk[lb + (in ? t : 0)]] : 0.0; bv[q] = in ? DC[lb + t] : 0.0; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc = fma(av[q], bv[q], acc);
                }
            }
            acc = wg::group_sum<4>(acc);
            if (g4 == 0 && live) A[i * lda + j] = acc;
        }
        for (int e = tid; e < n * m; e += nt) Bm[e] = Bg[e];
        __syncthreads();
    } else {
    basis_l(T.er, T.pr, T.vr, T.dmr, T.lvr, S.order_r, nr, n, x, w.phi, w.D);
    // A[i][j] = sum_k R[i][k] D[k][j] and f[i] = sum_k R[i][k] phi[k]: n (n + 1) dot products of length nr, four lanes
    // each (the k range split four ways, DPP sum)
    const int g4 = tid & 3;
    for (int o0 = 0; o0 < n * (n + 1); o0 += nt / 4) {           // uniform trip count
        const int o = o0 + (tid >> 2);
        const bool live = o < n * (n + 1);
        const int i = live ? o / (n + 1) : 0, j = live ? o - i * (n + 1) : 0;
        double acc = 0.0;
        if (live) {
            // eight operand pairs in flight per trip (a rolled load -> fma chain pays the LDS latency of every element)
            clptr r = T.R + (size_t)i * nr;
            clptr b = j < n ? w.D + j : w.phi;
            const int bs = j < n ? n : 1;
            for (int k0 = g4; k0 < nr; k0 += 32) {
                double av[8], bv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int k = k0 + 4 * q; const bool in = k < nr; av[q] = in ? r[k] : 0.0; bv[q] = in ? b[(size_t)k * bs] : 0.0; }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = fma(av[q], bv[q], acc);
            }
        }
        acc = wg::group_sum<4>(acc);
        if (g4 == 0 && live) { if (j < n) A[i * lda + j] = acc; else w.f[i] = acc; }
    }
    for (int e = tid; e < n * m; e += nt) Bm[e] = Bg[e];
    __syncthreads();
    }
    {   // f = R phi + B u ;  d = f - A x - B u : eight lanes per row
        const int g8 = tid & 7;
        for (int i0 = 0; i0 < n; i0 += nt / 8) {
            const int i = i0 + (tid >> 3);
            double ax = 0.0, bu = 0.0;
            if (i < n) {
                for (int k = g8; k < n; k += 8) ax = fma(A[i * lda + k], x[k], ax);
                for (int k = g8; k < m; k += 8) bu = fma(Bg[i * m + k], u[k], bu);
            }
            ax = wg::group_sum<8>(ax);
            bu = wg::group_sum<8>(bu);
            if (g8 == 0 && i < n) {
                const double f = w.f[i] + bu;
                d[i] = f - ax - bu;
                w.f[i] = f;
            }
        }
    }
    __syncthreads();
}

// z = W phi_s(x) from the LDS tables (no Jacobian); ends with a sync
__device__ inline void observe_l(const SsmDev &S, const SsmLds &T, clptr x, Work &w, lptr z) {
    const int no = S.no, ns = S.ns, tid = SRH_TID, nt = blockDim.x;
    basis_l(T.es, T.ps, T.vs, T.dms, T.lvs, S.order_s, ns, no, x, w.phi, (lptr) nullptr);
    const int g8 = tid & 7;
    for (int o0 = 0; o0 < no; o0 += nt / 8) {
        const int o = o0 + (tid >> 3);
        double acc = 0.0;
        if (o < no) for (int k = g8; k < ns; k += 8) acc = fma(T.W[(size_t)o * ns + k], w.phi[k], acc);
        acc = wg::group_sum<8>(acc);
        if (g8 == 0 && o < no) z[o] = acc;
    }
    __syncthreads();
}

}  // namespace ssm
