// Snapshot preprocessing of the POD build on the device (sofacontrol/mor/pod.py:157-178, 207-216): column statistics,
// `normalize`, `substract_mean`, and the k-means step of `clustering` -- the snapshot matrix S (n_s x n_f, one snapshot per
// row, 4 GB at the C4 size) stays in HBM between get_snapshots and the Gramian.
//
// k-means: the reference calls sklearn's KMeans(k, n_init = 100, max_iter = 1000, random_state = 0) (pod.py:214).  What is
// restated here is that estimator's published algorithm (scikit-learn, `sklearn/cluster/_kmeans.py`, dense Lloyd):
//   * labels: argmin_j ( |c_j|^2 - 2 x_i . c_j ), first minimum wins; the x . c products are one f64-MFMA GEMM (abt_dev.h);
//   * centres: means of the members, summed in snapshot order (deterministic); an empty cluster takes the snapshot that is
//     farthest from its centre, which leaves its old cluster (`_relocate_empty_clusters_dense`);
//   * stop when the labels repeat, or when the summed squared centre shift is <= tol; labels are recomputed for the final
//     centres unless the labels repeated; inertia = sum_i |x_i - c_label(i)|^2.
// The random draws of the k-means++ seeding stay on the host (numpy's RandomState, like sklearn); the distance rows they
// need come from srom_sqdist_rows_dev.
#include <cmath>
#include <vector>

#include "abt_dev.h"

namespace {

// ---------------------------------------------------------------- column statistics: two deterministic stages
// stage 1: grid (column blocks of 256, row slabs), thread = column (coalesced across the block), rows of the slab in order
__global__ __launch_bounds__(256) void colstats_part_kernel(const double *__restrict__ S, int64_t n_s, int64_t n_f, int64_t lds, int rows_per,
                                                            double *__restrict__ pmin, double *__restrict__ pmax, double *__restrict__ psum) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= n_f) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = min(r0 + rows_per, n_s);
    double mn = INFINITY, mx = -INFINITY, sm = 0.0;
    for (int64_t i = r0; i < r1; ++i) {
        const double v = S[i * lds + col];
        mn = fmin(mn, v); mx = fmax(mx, v); sm += v;
    }
    const int64_t o = (int64_t)blockIdx.y * n_f + col;
    pmin[o] = mn; pmax[o] = mx; psum[o] = sm;
}
__global__ __launch_bounds__(256) void colstats_final_kernel(const double *__restrict__ pmin, const double *__restrict__ pmax, const double *__restrict__ psum,
                                                             int slabs, int64_t n_f, int64_t n_s, double *__restrict__ mn_out,
                                                             double *__restrict__ mx_out, double *__restrict__ mean_out) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= n_f) return;
    double mn = INFINITY, mx = -INFINITY, sm = 0.0;
    for (int s = 0; s < slabs; ++s) { mn = fmin(mn, pmin[(int64_t)s * n_f + col]); mx = fmax(mx, pmax[(int64_t)s * n_f + col]); sm += psum[(int64_t)s * n_f + col]; }
    if (mn_out) mn_out[col] = mn;
    if (mx_out) mx_out[col] = mx;
    if (mean_out) mean_out[col] = sm / (double)n_s;
}

// S <- (S - a) / (b + 1e-15 - a)  (pod.py:165) or S <- S - a  (pod.py:168)
template <bool SCALE>
__global__ __launch_bounds__(256) void affine_kernel(double *__restrict__ S, int64_t n_s, int64_t n_f, int64_t lds, const double *__restrict__ a,
                                                     const double *__restrict__ b) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= n_f) return;
    const double lo = a[col], den = SCALE ? (b[col] + 1e-15 - lo) : 1.0;
    for (int64_t i = blockIdx.y; i < n_s; i += gridDim.y) {
        const double v = S[i * lds + col] - lo;
        S[i * lds + col] = SCALE ? v / den : v;
    }
}

// |x_i|^2, one wave per row, fixed tree
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const double *__restrict__ S, int64_t n_s, int64_t n_f, int64_t lds, double *__restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_s) return;
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int64_t c = lane; c < n_f; c += 64) { const double v = S[row * lds + c]; acc = fma(v, v, acc); }
    acc = wg::wave_sum(acc);
    if (lane == 0) out[row] = acc;
}

// D[c][i] = max(0, |y_c|^2 - 2 P[i][c] + |x_i|^2)   (sklearn's euclidean_distances(squared=True))
__global__ void sqdist_kernel(const double *__restrict__ P, int64_t n_s, int nc, const double *__restrict__ xn, const double *__restrict__ yn,
                              double *__restrict__ D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_s) return;
    for (int c = 0; c < nc; ++c) D[(int64_t)c * n_s + i] = fmax(0.0, yn[c] - 2.0 * P[i * nc + c] + xn[i]);
}

// labels: first minimum of |c_j|^2 - 2 x_i . c_j; changed[0] counts the labels that differ from labels_old
__global__ void assign_kernel(const double *__restrict__ P, int64_t n_s, int k, const double *__restrict__ cn, int32_t *__restrict__ labels,
                              const int32_t *__restrict__ labels_old, int *__restrict__ changed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_s) return;
    double best = cn[0] - 2.0 * P[i * k];
    int bj = 0;
    for (int j = 1; j < k; ++j) {
        const double d = cn[j] - 2.0 * P[i * k + j];
        if (d < best) { best = d; bj = j; }
    }
    labels[i] = bj;
    if (labels_old && labels_old[i] != bj) atomicAdd(changed, 1);          // a count only: no floating-point order involved
}

// sums[j][col] = sum of the members' entries in snapshot order; counts[j]
__global__ __launch_bounds__(256) void cluster_sums_kernel(const double *__restrict__ S, int64_t n_s, int64_t n_f, int64_t lds, const int32_t *__restrict__ labels,
                                                           double *__restrict__ sums, int *__restrict__ counts) {
    const int j = blockIdx.y;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    int cnt = 0;
    for (int64_t i = 0; i < n_s; ++i) {
        if (labels[i] == j) {                                  // uniform across the block
            ++cnt;
            if (col < n_f) acc += S[i * lds + col];
        }
    }
    if (col < n_f) sums[(int64_t)j * n_f + col] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[j] = cnt;
}

// d_i = |x_i - c_label(i)|^2 (one wave per snapshot)
__global__ __launch_bounds__(256) void member_dist_kernel(const double *__restrict__ S, int64_t n_s, int64_t n_f, int64_t lds, const double *__restrict__ Cn,
                                                          const int32_t *__restrict__ labels, double *__restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_s) return;
    const int lane = threadIdx.x & 63;
    const double *c = Cn + (int64_t)labels[row] * n_f;
    double acc = 0.0;
    for (int64_t q = lane; q < n_f; q += 64) { const double v = S[row * lds + q] - c[q]; acc = fma(v, v, acc); }
    acc = wg::wave_sum(acc);
    if (lane == 0) out[row] = acc;
}

// relocation of one empty cluster (sums, before the division): the far snapshot leaves `from` and founds `to`
__global__ void relocate_kernel(const double *__restrict__ S, int64_t lds, int64_t n_f, int64_t far, int from, int to, double *__restrict__ sums) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n_f) return;
    const double v = S[far * lds + col];
    sums[(int64_t)from * n_f + col] -= v;
    sums[(int64_t)to * n_f + col] = v;
}

// C_new = sums / count (count == 0: keep the old centre, like sklearn's average_centers); shift2[j] = |C_new_j - C_old_j|^2
__global__ __launch_bounds__(256) void finish_centers_kernel(const double *__restrict__ sums, const int *__restrict__ counts, const double *__restrict__ Cold,
                                                             int64_t n_f, double *__restrict__ Cnew, double *__restrict__ shift2) {
    __shared__ double red[16];
    const int j = blockIdx.x;
    const int cnt = counts[j];
    double acc = 0.0;
    for (int64_t c = threadIdx.x; c < n_f; c += 256) {
        const double o = Cold[(int64_t)j * n_f + c];
        const double v = cnt > 0 ? sums[(int64_t)j * n_f + c] / (double)cnt : o;
        Cnew[(int64_t)j * n_f + c] = v;
        acc = fma(v - o, v - o, acc);
    }
    acc = wg::reduce(acc, 0, (lptr)red);
    if (threadIdx.x == 0) shift2[j] = acc;
}

// out[0] = sum of v (fixed order: one workgroup)
__global__ __launch_bounds__(256) void sum_kernel(const double *__restrict__ v, int64_t n, double *__restrict__ out) {
    __shared__ double red[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc += v[i];
    acc = wg::reduce(acc, 0, (lptr)red);
    if (threadIdx.x == 0) out[0] = acc;
}

struct Stats {
    srh::DevBuf part;
    int run(const double *S, int64_t n_s, int64_t n_f, int64_t lds, double *mn, double *mx, double *mean, hipStream_t st) {
        const int cb = (int)srh::cdiv(n_f, 256);
        int slabs = (int)std::min<int64_t>(std::max<int64_t>(1, 2048 / cb), srh::cdiv(n_s, 64));
        const int rows_per = (int)srh::cdiv(n_s, slabs);
        slabs = (int)srh::cdiv(n_s, rows_per);
        int rc = part.alloc(sizeof(double) * 3 * (size_t)slabs * n_f);
        if (rc) return rc;
        double *pmin = part.as<double>(), *pmax = pmin + (size_t)slabs * n_f, *psum = pmax + (size_t)slabs * n_f;
        colstats_part_kernel<<<dim3(cb, slabs), 256, 0, st>>>(S, n_s, n_f, lds, rows_per, pmin, pmax, psum);
        colstats_final_kernel<<<cb, 256, 0, st>>>(pmin, pmax, psum, slabs, n_f, n_s, mn, mx, mean);
        SRH_CHECK_HIP(hipGetLastError());
        return SRH_OK;
    }
};

}  // namespace

extern "C" {

int srom_snapshot_stats_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, double *min_dev, double *max_dev, double *mean_dev,
                            void *stream) {
    SRH_REQUIRE(S_dev && n_s > 0 && n_f > 0 && lds >= n_f, "srom_snapshot_stats_dev: bad arguments");
    Stats st;
    int rc = st.run(S_dev, n_s, n_f, lds, min_dev, max_dev, mean_dev, (hipStream_t)stream);
    if (rc) return rc;
    SRH_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));            // the partials go back to the pool with `st`
    return SRH_OK;
}

int srom_snapshot_normalize_dev(double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, const double *min_dev, const double *max_dev, void *stream) {
    SRH_REQUIRE(S_dev && min_dev && max_dev && n_s > 0 && n_f > 0 && lds >= n_f, "srom_snapshot_normalize_dev: bad arguments");
    const int cb = (int)srh::cdiv(n_f, 256);
    affine_kernel<true><<<dim3(cb, (unsigned)std::min<int64_t>(n_s, std::max<int64_t>(1, 4096 / cb))), 256, 0, (hipStream_t)stream>>>(S_dev, n_s, n_f, lds, min_dev, max_dev);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

int srom_snapshot_center_dev(double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, const double *mean_dev, void *stream) {
    SRH_REQUIRE(S_dev && mean_dev && n_s > 0 && n_f > 0 && lds >= n_f, "srom_snapshot_center_dev: bad arguments");
    const int cb = (int)srh::cdiv(n_f, 256);
    affine_kernel<false><<<dim3(cb, (unsigned)std::min<int64_t>(n_s, std::max<int64_t>(1, 4096 / cb))), 256, 0, (hipStream_t)stream>>>(S_dev, n_s, n_f, lds, mean_dev, nullptr);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

int srom_row_sqnorms_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, double *out_dev, void *stream) {
    SRH_REQUIRE(S_dev && out_dev && n_s > 0 && n_f > 0 && lds >= n_f, "srom_row_sqnorms_dev: bad arguments");
    row_sqnorm_kernel<<<(unsigned)srh::cdiv(n_s, 4), 256, 0, (hipStream_t)stream>>>(S_dev, n_s, n_f, lds, out_dev);
    SRH_CHECK_HIP(hipGetLastError());
    return SRH_OK;
}

int srom_sqdist_rows_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, const double *Y_dev, int nc, const double *xnorm_dev,
                         double *out_dev, void *stream) {
    SRH_REQUIRE(S_dev && Y_dev && xnorm_dev && out_dev && n_s > 0 && n_f > 0 && nc > 0 && lds >= n_f, "srom_sqdist_rows_dev: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    srh::DevBuf P, yn;
    int rc;
    if ((rc = P.alloc(sizeof(double) * (size_t)n_s * nc)) || (rc = yn.alloc(sizeof(double) * nc))) return rc;
    Abt abt;
    if ((rc = abt.run(S_dev, lds, n_s, Y_dev, n_f, nc, n_f, P.as<double>(), nc, st))) return rc;
    row_sqnorm_kernel<<<(unsigned)srh::cdiv(nc, 4), 256, 0, st>>>(Y_dev, nc, n_f, n_f, yn.as<double>());
    sqdist_kernel<<<(unsigned)srh::cdiv(n_s, 256), 256, 0, st>>>(P.as<double>(), n_s, nc, xnorm_dev, yn.as<double>(), out_dev);
    SRH_CHECK_HIP(hipGetLastError());
    SRH_CHECK_HIP(hipStreamSynchronize(st));
    return SRH_OK;
}

int srom_kmeans_lloyd_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, int k, double *C_dev, int max_iter, double tol,
                          int32_t *labels_dev, double *inertia_out, int *iters_out, void *stream) {
    SRH_REQUIRE(S_dev && C_dev && labels_dev && n_s > 0 && n_f > 0 && k > 0 && k <= n_s && lds >= n_f && max_iter > 0,
                "srom_kmeans_lloyd_dev: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    srh::DevBuf P, cn, Cn, sums, counts, lold, flag, shift2, dist, scal;
    int rc;
    if ((rc = P.alloc(sizeof(double) * (size_t)n_s * k)) || (rc = cn.alloc(sizeof(double) * k)) || (rc = Cn.alloc(sizeof(double) * (size_t)k * n_f)) ||
        (rc = sums.alloc(sizeof(double) * (size_t)k * n_f)) || (rc = counts.alloc(sizeof(int) * k)) || (rc = lold.alloc(sizeof(int32_t) * n_s)) ||
        (rc = flag.alloc(sizeof(int))) || (rc = shift2.alloc(sizeof(double) * k)) || (rc = dist.alloc(sizeof(double) * n_s)) ||
        (rc = scal.alloc(sizeof(double) * 2)))
        return rc;
    Abt abt;
    double *Ccur = C_dev, *Cnew = Cn.as<double>();
    const unsigned gb = (unsigned)srh::cdiv(n_s, 256), cb = (unsigned)srh::cdiv(n_f, 256);
    auto label_step = [&](const double *Cc, bool compare) -> int {
        int r2;
        if ((r2 = abt.run(S_dev, lds, n_s, Cc, n_f, k, n_f, P.as<double>(), k, st))) return r2;
        row_sqnorm_kernel<<<(unsigned)srh::cdiv(k, 4), 256, 0, st>>>(Cc, k, n_f, n_f, cn.as<double>());
        SRH_CHECK_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
        assign_kernel<<<gb, 256, 0, st>>>(P.as<double>(), n_s, k, cn.as<double>(), labels_dev, compare ? lold.as<int32_t>() : nullptr, flag.as<int>());
        SRH_CHECK_HIP(hipGetLastError());
        return SRH_OK;
    };
    bool strict = false;
    int it = 0;
    std::vector<int> hc(k);
    for (it = 0; it < max_iter; ++it) {
        if ((rc = label_step(Ccur, it > 0))) return rc;
        cluster_sums_kernel<<<dim3(cb, k), 256, 0, st>>>(S_dev, n_s, n_f, lds, labels_dev, sums.as<double>(), counts.as<int>());
        SRH_CHECK_HIP(hipGetLastError());
        SRH_CHECK_HIP(hipMemcpyAsync(hc.data(), counts.p, sizeof(int) * k, hipMemcpyDeviceToHost, st));
        int changed = 1;
        if (it > 0) SRH_CHECK_HIP(hipMemcpyAsync(&changed, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
        SRH_CHECK_HIP(hipStreamSynchronize(st));
        std::vector<int> empty;
        for (int j = 0; j < k; ++j) if (hc[j] == 0) empty.push_back(j);
        if (!empty.empty()) {
            // sklearn's _relocate_empty_clusters_dense: the snapshots farthest from their (old) centres, farthest first
            member_dist_kernel<<<(unsigned)srh::cdiv(n_s, 4), 256, 0, st>>>(S_dev, n_s, n_f, lds, Ccur, labels_dev, dist.as<double>());
            std::vector<double> hd(n_s);
            std::vector<int32_t> hl(n_s);
            SRH_CHECK_HIP(hipMemcpyAsync(hd.data(), dist.p, sizeof(double) * n_s, hipMemcpyDeviceToHost, st));
            SRH_CHECK_HIP(hipMemcpyAsync(hl.data(), labels_dev, sizeof(int32_t) * n_s, hipMemcpyDeviceToHost, st));
            SRH_CHECK_HIP(hipStreamSynchronize(st));
            std::vector<int64_t> order(n_s);
            for (int64_t i = 0; i < n_s; ++i) order[i] = i;
            std::partial_sort(order.begin(), order.begin() + empty.size(), order.end(),
                              [&](int64_t a, int64_t b) { return hd[a] > hd[b] || (hd[a] == hd[b] && a > b); });
            for (size_t e = 0; e < empty.size(); ++e) {
                const int64_t far = order[e];
                const int from = hl[far];
                relocate_kernel<<<cb, 256, 0, st>>>(S_dev, lds, n_f, far, from, empty[e], sums.as<double>());
                hc[empty[e]] = 1;
                hc[from] -= 1;
            }
            SRH_CHECK_HIP(hipMemcpyAsync(counts.p, hc.data(), sizeof(int) * k, hipMemcpyHostToDevice, st));
        }
        finish_centers_kernel<<<k, 256, 0, st>>>(sums.as<double>(), counts.as<int>(), Ccur, n_f, Cnew, shift2.as<double>());
        sum_kernel<<<1, 256, 0, st>>>(shift2.as<double>(), k, scal.as<double>());
        SRH_CHECK_HIP(hipGetLastError());
        double shift_tot = 0.0;
        SRH_CHECK_HIP(hipMemcpyAsync(&shift_tot, scal.p, sizeof(double), hipMemcpyDeviceToHost, st));
        SRH_CHECK_HIP(hipStreamSynchronize(st));
        std::swap(Ccur, Cnew);                                             // centers, centers_new = centers_new, centers
        if (it > 0 && changed == 0) { strict = true; ++it; break; }
        if (shift_tot <= tol) { ++it; break; }
        SRH_CHECK_HIP(hipMemcpyAsync(lold.p, labels_dev, sizeof(int32_t) * n_s, hipMemcpyDeviceToDevice, st));
    }
    if (!strict && (rc = label_step(Ccur, false))) return rc;              // labels of the final centres
    member_dist_kernel<<<(unsigned)srh::cdiv(n_s, 4), 256, 0, st>>>(S_dev, n_s, n_f, lds, Ccur, labels_dev, dist.as<double>());
    sum_kernel<<<1, 256, 0, st>>>(dist.as<double>(), n_s, scal.as<double>() + 1);
    SRH_CHECK_HIP(hipGetLastError());
    double inertia = 0.0;
    SRH_CHECK_HIP(hipMemcpyAsync(&inertia, scal.as<double>() + 1, sizeof(double), hipMemcpyDeviceToHost, st));
    if (Ccur != C_dev) SRH_CHECK_HIP(hipMemcpyAsync(C_dev, Ccur, sizeof(double) * (size_t)k * n_f, hipMemcpyDeviceToDevice, st));
    SRH_CHECK_HIP(hipStreamSynchronize(st));
    if (inertia_out) *inertia_out = inertia;
    if (iters_out) *iters_out = std::min(it, max_iter);
    return SRH_OK;
}

}  // extern "C"
