/*
 * sofacontrol_hip.h -- C ABI of libsofacontrol_hip.so (MI355X / gfx950).
 *
 * The reference (StanfordASL/soft-robot-control) is pure Python and has no FFI of its own: the seam
 * is its duck-typed Python protocol.  Each entry point below names the reference
 * method it serves (file:line relative to the reference root); the Python host package
 * (soft-robot-control_amd/sofacontrol_amd) binds these with ctypes and re-exposes the reference's
 * class/method surface.  INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (SRH_E*); never throws; srh_last_error() gives
 *    the message of the last failure on the calling thread;
 *  - all arithmetic is IEEE float64 (the reference is numpy float64 throughout);
 *  - matrices are dense row-major; "ld" arguments are leading dimensions in elements;
 *  - `*_dev` entry points take DEVICE pointers (HBM) and a hipStream_t passed as void* (NULL = the
 *    null stream) and are asynchronous on that stream; entry points without the suffix take HOST
 *    pointers, stage through HBM and are synchronous on return;
 *  - handles are owned by the caller and are not thread-safe (one host thread per handle, as the
 *    reference's single SOFA/solver thread).
 */
#ifndef SOFACONTROL_HIP_H
#define SOFACONTROL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRH_OK        0
#define SRH_EINVAL   -1   /* bad argument (the reference raises RuntimeError / AssertionError)        */
#define SRH_EHIP     -2   /* HIP runtime error (no device, launch failure, ...)                      */
#define SRH_ENOMEM   -3
#define SRH_ENUMERIC -4   /* numerical failure (non-PD matrix, QP not solved: reference returns flags) */

const char *srh_last_error(void);
int srh_version(void);

/* ---- device plumbing (lets a host without torch stage buffers; not part of the reference) ---- */
int srh_device_count(int *count);
int srh_set_device(int device);
int srh_malloc(void **dptr, size_t bytes);
int srh_free(void *dptr);
int srh_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int srh_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int srh_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes);
int srh_memset(void *dptr, int value, size_t bytes);
int srh_sync(void);
/* The host-pointer entry points keep the device blocks they release in a size-class cache (<= 2 GiB per process)
 * instead of paying hipMalloc / hipFree per call -- hipFree waits for EVERY stream of the device, which would stall the
 * caller behind an asynchronous solver request; this returns the cached blocks to the driver. */
int srh_release_cached(void);
/* average duration in milliseconds of `iters` back-to-back launches are measured by the caller with
 * these (hipEvent on the given stream): */
int srh_event_create(void **ev);
int srh_event_destroy(void *ev);
int srh_event_record(void *ev, void *stream);
int srh_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms);   /* synchronises ev_stop */

/* =====================================================================================================
 * POD reduced-order map.          reference: sofacontrol/mor/pod.py
 * ===================================================================================================== */
typedef struct srom srom_t;

/* POD.__init__ (pod.py:14-20): U (n_f x r) row-major, q_ref / v_ref (n_f,) or NULL (= zeros).
 * Copies the basis to HBM and packs it into MFMA fragment order. */
int srom_create(srom_t **h, const double *U, int64_t n_f, int r, const double *q_ref, const double *v_ref);
int srom_destroy(srom_t *h);
int srom_dims(const srom_t *h, int64_t *n_f, int *r);

#define SROM_Q 0   /* U^T (qf - q_ref)      pod.py:46-47 */
#define SROM_V 1   /* U^T (vf - v_ref)      pod.py:48-49 */
#define SROM_X 2   /* V^T (xf - x_ref), x = [v; q], V = kron(I2, U)   pod.py:51-52 */
#define SROM_RAW 3 /* U^T m (no reference subtracted): rows of a matrix, pod.py:68-72 */

/* POD.compute_RO_state (pod.py:39-54), batched: X is (B x n_f) [SROM_Q/V/RAW] or (B x 2 n_f) [SROM_X],
 * one snapshot per row (the layout of np.asarray(data['q']), pod.py:149); out is (B x r) or (B x 2r). */
int srom_project(srom_t *h, int which, const double *X, int64_t B, double *out);
int srom_project_dev(srom_t *h, int which, const double *X_dev, int64_t B, int64_t ldx,
                     double *out_dev, int64_t ldo, void *stream);
/* utils.qv2x (sofacontrol/utils.py:129-130) on resident reduced coordinates: x (B x 2r, row pitch ldx) = [v ; q] from
 * q_dev (B x r, pitch ldq) and v_dev (B x r, pitch ldv; NULL: zero velocities) -- what the controllers do with the two
 * halves of compute_RO_state before the model / solver sees a reduced state (tpwl/controllers.py:96). */
int srom_qv2x_dev(const double *q_dev, int64_t ldq, const double *v_dev, int64_t ldv, int64_t B, int r,
                  double *x_dev, int64_t ldx, void *stream);

/* POD.compute_FO_state (pod.py:22-37), batched: Xr (B x r | B x 2r) -> out (B x n_f | B x 2 n_f). */
int srom_lift(srom_t *h, int which, const double *Xr, int64_t B, double *out);
int srom_lift_dev(srom_t *h, int which, const double *Xr_dev, int64_t B, int64_t ldr,
                  double *out_dev, int64_t ldo, void *stream);

/* POD.compute_RO_matrix (pod.py:56-72) for a dense row-major M (n_f x ncols):
 *   left && right (or neither): U^T M U (r x r), needs ncols == n_f
 *   left only : U^T M (r x ncols)           right only: M U (n_f x r), needs ncols == n_f */
int srom_reduce_matrix(srom_t *h, const double *M, int64_t ncols, int left, int right, double *out);
int srom_reduce_matrix_dev(srom_t *h, const double *M_dev, int64_t ncols, int left, int right,
                           double *out_dev, void *stream);
/* The same U^T M U for `count` dense n_f x n_f matrices in one call: TPWLSnapshotData.add_point reduces K, D, M and S of one
 * linearisation point back to back (tpwl/tpwl_utils.py:96-103 -> pod.py:56-72); groups of four share ONE launch pair.  M / out:
 * arrays of `count` pointers (row-major n_f x n_f in, r x r out). */
int srom_reduce_matrices(srom_t *h, const double *const *M, int count, double *const *out);
int srom_reduce_matrices_dev(srom_t *h, const double *const *M_dev, int count, double *const *out_dev, void *stream);

/* Snapshot Gramian G = S S^T, S (n_s x n_f) row-major, G (n_s x n_s): the method-of-snapshots route
 * to compute_POD (pod.py:181-200: sigma_i = sqrt(eig_i(G)), U = S^T W Sigma^-1).  With S sharded by
 * columns over ranks the partial Gramians are summed by an RCCL all-reduce in the host layer. */
int srom_gramian_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, double *G_dev,
                     void *stream);
int srom_gramian(const double *S, int64_t n_s, int64_t n_f, double *G);
/* process_snapshots (pod.py:157-178) on a resident snapshot matrix S (n_s x n_f, one snapshot per row, row pitch lds >= n_f).
 * Column statistics over the snapshots (each output n_f doubles on the device, any of them may be NULL):
 * min / max for 'normalize' (pod.py:165), mean for 'substract_mean' (pod.py:168). */
int srom_snapshot_stats_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, double *min_dev, double *max_dev,
                            double *mean_dev, void *stream);
/* 'normalize' in place: S <- (S - min) / (max + 1e-15 - min)   (pod.py:165) */
int srom_snapshot_normalize_dev(double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, const double *min_dev,
                                const double *max_dev, void *stream);
/* 'substract_mean' in place: S <- S - mean   (pod.py:168) */
int srom_snapshot_center_dev(double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, const double *mean_dev, void *stream);
/* 'clustering' (pod.py:170-174 -> compute_kmeans_centroids, pod.py:207-216: sklearn KMeans(k, n_init=100, max_iter=1000,
 * random_state=0)); the estimator's dense Lloyd algorithm restated on the device, one run from given centres:
 * C_dev (k x n_f) holds the initial centres and receives the final ones; labels_dev (n_s int32); tol = the absolute
 * threshold on the summed squared centre shift (sklearn: 1e-4 * mean column variance); *iters_out = Lloyd iterations.
 * srom_row_sqnorms_dev / srom_sqdist_rows_dev provide what the k-means++ seeding needs: |x_i|^2 and the squared
 * distances max(0, |y_c|^2 - 2 y_c . x_i + |x_i|^2) of every snapshot to nc given rows Y (nc x n_f) -> out (nc x n_s);
 * the seeding's random draws stay with the caller (numpy's RandomState in the reference's stack). */
int srom_row_sqnorms_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, double *out_dev, void *stream);
int srom_sqdist_rows_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, const double *Y_dev, int nc,
                         const double *xnorm_dev, double *out_dev, void *stream);
int srom_kmeans_lloyd_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, int k, double *C_dev, int max_iter,
                          double tol, int32_t *labels_dev, double *inertia_out, int *iters_out, void *stream);
/* eigh of the (symmetric) Gramian in place on the device (cyclic Jacobi kernels up to n = 2048, rocSOLVER dsyevd
 * above): on return row j of G_dev is
 * the eigenvector of the j-th smallest eigenvalue, w_dev (n) ascending.  Replaces the SVD of pod.py:190. */
int srom_eigh_dev(double *G_dev, int64_t n, double *w_dev, void *stream);
/* The k LARGEST eigenpairs of the symmetric positive semidefinite G (n x n, left untouched) by blocked subspace iteration +
 * Rayleigh-Ritz (block = k + oversample <= 128; oversample < 0: max(16, k / 2)) -- what compute_POD keeps of the SVD
 * (pod.py:190-200) without the other n - k singular values: w_dev (k) descending; Wk_dev (n x k) = eigenvector columns
 * scaled by 1 / sqrt(w_i) (the input of srom_modes_dev) or NULL; Vt_dev (k x n) = eigenvector rows or NULL; *trace_out
 * = trace(G) (the truncation rule's denominator: tail energy = trace - sum w); *iters_out = subspace iterations until the
 * leading k eigenvalues settled to 1e-13 (<= 30), or 31 when they did not (no gap behind the block: take srom_eigh_dev). */
int srom_eigh_topk_dev(const double *G_dev, int64_t n, int k, int oversample, double *w_dev, double *Wk_dev,
                       double *Vt_dev, double *trace_out, int *iters_out, void *stream);
/* W_k (n x k, row-major) = the k leading eigenvectors as columns, scaled by 1/sigma_i = 1/sqrt(w_i) */
int srom_select_modes_dev(const double *V_dev, const double *w_dev, int64_t n, int k, double *Wk_dev,
                          void *stream);
/* U_k = S^T W_k  (n_f x k) for eigenvector columns W_k (n_s x k, already scaled by 1/sigma). */
int srom_modes_dev(const double *S_dev, int64_t n_s, int64_t n_f, int64_t lds, const double *W_dev,
                   int k, double *U_dev, void *stream);

/* =====================================================================================================
 * TPWL piecewise-affine model.    reference: sofacontrol/tpwl/tpwl.py, sofacontrol/scp/models/tpwl.py
 * ===================================================================================================== */
typedef struct stpwl stpwl_t;

/* Tables of the P linearisation points (tpwl.py:20-41; keys 'q','v','u','A_c','B_c','d_c' of the
 * tpwl_dict), reduced state x = [v; q], n_x = 2 r.  A_d/B_d/d_d are the pre-discretised tables
 * (TPWLATV.pre_discretize, tpwl.py:299-322) or NULL.  w_q, w_v = dist_weights (tpwl.py:165-166). */
int stpwl_create(stpwl_t **h, int P, int r, int n_u,
                 const double *q, const double *v, const double *u,
                 const double *A_c, const double *B_c, const double *d_c,
                 const double *A_d, const double *B_d, const double *d_d,
                 double w_q, double w_v);
int stpwl_destroy(stpwl_t *h);
/* install / replace the discrete tables (pre_discretize with another dt) */
int stpwl_set_discrete(stpwl_t *h, const double *A_d, const double *B_d, const double *d_d);
/* output model z = H x + z_ref (TPWL.set_output_model, tpwl.py:86-89); H (n_z x n_x) */
int stpwl_set_output(stpwl_t *h, const double *H, const double *z_ref, int n_z);

/* TPWL.calc_nearest_point (tpwl.py:160-168) for B states X (B x n_x): idx (B,) int32 */
int stpwl_nearest(stpwl_t *h, const double *X, int64_t B, int32_t *idx);
int stpwl_nearest_dev(stpwl_t *h, const double *X_dev, int64_t B, int32_t *idx_dev, void *stream);
/* TPWLATV.get_jacobians, nn branch (tpwl.py:251-265): gathers (A,B,d)[idx] of the discrete
 * (discrete=1) or continuous tables: A (B x n_x x n_x), Bm (B x n_x x n_u), d (B x n_x) */
int stpwl_linearize(stpwl_t *h, const double *X, int64_t B, int discrete, double *A, double *Bm,
                    double *d, int32_t *idx);
/* TPWL.calc_weighting_factors (tpwl.py:170-191): W (B x P) softmin weights exp(-beta d_i / d_min),
 * normalised; one-hot at the first minimum when d_min == 0 */
int stpwl_weights(stpwl_t *h, const double *X, int64_t B, double beta, double *W);
/* TPWLATV.get_jacobians, weighting branch (tpwl.py:244-248): continuous A = sum_i w_i A_c[i] etc.;
 * W (B x P) optional output of the weights */
int stpwl_linearize_weighted(stpwl_t *h, const double *X, int64_t B, double beta, double *A, double *Bm,
                             double *d, double *W);
/* TPWL.discretize_dynamics (tpwl.py:272-297; zoh: sofacontrol/utils.py:302-335 zoh_affine) for `batch` continuous affine models
 * xdot = A x + B u + d at once: the stored points of pre_discretize (tpwl.py:299-322), the blended model of weighting-mode TPWL
 * (tpwl.py:244-250).  method: 0 fe, 1 be, 2 bil, 3 zoh.  A (batch x n x n), B (batch x n x m), d (batch x n) -> same shapes.
 * SRH_ENUMERIC if a model's I - c A is singular. */
int stpwl_discretize(int method, int n, int m, int64_t batch, const double *A, const double *B, const double *d, double dt,
                     double *Ad, double *Bd, double *dd);
int stpwl_discretize_dev(int method, int n, int m, int64_t batch, const double *A_dev, const double *B_dev, const double *d_dev,
                         double dt, double *Ad_dev, double *Bd_dev, double *dd_dev, void *stream);
/* TPWL.rollout (tpwl.py:193-216) for `batch` independent rollouts:
 * x0 (batch x n_x), U (batch x N x n_u) -> X (batch x (N+1) x n_x), Z (batch x (N+1) x n_z) or NULL */
int stpwl_rollout(stpwl_t *h, const double *x0, const double *U, int N, int64_t batch, double *X,
                  double *Z);
/* the same on resident buffers, asynchronous on `stream` (Z_dev may be NULL): the zero-input initial guess of a
 * receding-horizon solve (scp/ros.py:78-79) without leaving the device */
int stpwl_rollout_dev(stpwl_t *h, const double *x0_dev, const double *U_dev, int N, int64_t batch,
                      double *X_dev, double *Z_dev, void *stream);
/* TPWLGuSTO.get_characteristic_vals (scp/models/tpwl.py:66-84): x_char, f_char (n_x,) */
int stpwl_characteristic(stpwl_t *h, double *x_char, double *f_char);

/* =====================================================================================================
 * Discrete EKF over the TPWL model.                        reference: sofacontrol/tpwl/observer.py:33-126
 * State estimate x (n_x) and covariance Sigma (n_x x n_x) stay resident in HBM between steps.
 * ===================================================================================================== */
typedef struct sekf sekf_t;
/* DiscreteEKFObserver.__init__ (observer.py:52-67): C (n_y x n_x) reduced measurement matrix, y_ref (n_y)
 * or NULL, Sigma0 / W (n_x x n_x), V (n_y x n_y).  The model handle must outlive the filter. */
int sekf_create(sekf_t **out, stpwl_t *model, const double *C, const double *y_ref, int n_y,
                const double *Sigma0, const double *W, const double *V);
int sekf_destroy(sekf_t *h);
/* initialize (observer.py:76-86): overwrite the estimate and/or the covariance (either may be NULL) */
int sekf_set_state(sekf_t *h, const double *x, const double *Sigma);
int sekf_get_state(sekf_t *h, double *x, double *Sigma);
/* update = predict_state + update_state (observer.py:88-126).  u (n_u) != NULL runs the predictor with the
 * nearest-point discrete tables of the model, or with (A_d, B_d, d_d) when given (weighting-mode models);
 * y (n_y, full-order measurement, y_ref is subtracted) != NULL runs the filter update.  x_out (n_x) optional.
 * Returns SRH_ENUMERIC when the innovation covariance is not positive definite. */
int sekf_step(sekf_t *h, const double *u, const double *y, const double *A_d, const double *B_d,
              const double *d_d, double *x_out);
/* One call for the per-simulation-step path of a closed-loop controller (closed_loop_controller.py:205-233,
 * tpwl/controllers.py:128-157: compute_RO_state(xf=x) followed by observer.update(u_prev, y, dt)): the full-order
 * state x_full = [v_f; q_f] (2 n_f) is projected to x_reduced_out (2 r) on a side stream while the filter step
 * (sekf_step with the model's own discrete tables) runs on stream 0; both go through their pinned mirrors and the
 * host waits once for each.  x_hat_out (n_x) optional.  Errors as sekf_step / srom_project. */
int sekf_step_projected(sekf_t *h, srom_t *rom, const double *x_full, const double *u, const double *y,
                        double *x_reduced_out, double *x_hat_out);

/* Polyhedron(with_reproject=True).project_to_polyhedron (utils.py:364-407; the measurement re-projection of
 * SSM/controllers.py:96-97): Euclidean projection of `batch` points X (batch x n) onto {p : A p <= b}, A (n_rows x n)
 * row-major, n <= 16, n_rows <= 64; exact (interior point to a 1e-13 gap; the reference hands the same QP to OSQP).
 * Points already inside are returned unchanged.  SRH_ENUMERIC when the iteration does not converge (empty set). */
int spoly_project(const double *A, const double *b, int n_rows, int n, const double *X, int64_t batch, double *out);

/* =====================================================================================================
 * SSM polynomial reduced model.                                  reference: sofacontrol/SSM/ssm.py
 * f(x,u) = r_coeff phi_rom(x) + B u;  z = w_coeff phi_ssm(x) + z_ref;  x = v_coeff phi_ssm(z - z_ref),
 * phi = monomials of degree 1..order, graded, lexicographic with x1 first (get_poly_basis, ssm.py:158-164).
 * ===================================================================================================== */
typedef struct sssm sssm_t;
#define SSSM_CONT         0   /* continuous Jacobians (get_continuous_jacobians, ssm.py:198-204)        */
#define SSSM_FE           1   /* discretize_dynamics 'fe'  (ssm.py:279-283)                              */
#define SSSM_BE           2   /* 'be'  (ssm.py:285-289)                                                  */
#define SSSM_BIL          3   /* 'bil' (ssm.py:291-296)                                                  */
#define SSSM_DISCRETE_MAP 4   /* self.discrete: Jacobians of rd_coeff phi + Bd u (ssm.py:206-218)        */
int sssm_num_monomials(int dim, int order);
int sssm_exponents(int dim, int order, int32_t *exps);            /* (n_mon x dim) exponent table */
/* SSM.__init__ (ssm.py:24-71): r_coeff (n_x x n_rom), B (n_x x n_u), rd_coeff / Bd or NULL, w_coeff
 * (n_o x n_ssm), v_coeff (n_x x n_ssm), z_ref (n_o); n_rom / n_ssm = sssm_num_monomials(n_x / n_o, order) */
int sssm_create(sssm_t **out, int n_x, int n_u, int n_o, int rom_order, int ssm_order,
                const double *r_coeff, const double *B, const double *rd_coeff, const double *Bd,
                const double *w_coeff, const double *v_coeff, const double *z_ref);
int sssm_destroy(sssm_t *h);
/* bookkeeping performance matrix H (n_o x n_x) used by iLQR cost Hessians (ssm.py:69-70; zeros by default) */
int sssm_set_output(sssm_t *h, const double *H);
/* SSMDynamics.get_jacobians / get_continuous_jacobians / get_discrete_jacobians (ssm.py:198-218) for B
 * points: X (B x n_x), U (B x n_u) -> A (B x n_x x n_x), Bm (B x n_x x n_u), d (B x n_x) */
int sssm_linearize(sssm_t *h, const double *X, const double *U, int64_t B, int mode, double dt,
                   double *A, double *Bm, double *d);
/* reduced_dynamics / reduced_dynamics_discrete (ssm.py:167-178): F (B x n_x) */
int sssm_dynamics(sssm_t *h, const double *X, const double *U, int64_t B, int discrete, double *F);
/* reduced_to_observed C_map (ssm.py:170-171) -> Z (B x n_o, WITHOUT z_ref) and/or get_observer_jacobians
 * (ssm.py:220-227): H (B x n_o x n_x), c = C(x) - H x (B x n_o).  Z or H may be NULL. */
int sssm_observe(sssm_t *h, const double *X, int64_t B, double *Z, double *H, double *c);
/* compute_RO_state (ssm.py:338-344): X = v_coeff phi(Z - z_ref) */
int sssm_reduce(sssm_t *h, const double *Z, int64_t B, double *X);
/* SSM.rollout (ssm.py:134-156) for `batch` rollouts: x0 (batch x n_x), U (batch x N x n_u) ->
 * X (batch x (N+1) x n_x), Z (batch x (N+1) x n_o, includes z_ref) or NULL */
int sssm_rollout(sssm_t *h, const double *x0, const double *U, int N, int64_t batch, int mode, double dt,
                 double *X, double *Z);

/* =====================================================================================================
 * Riccati recursions.             reference: sofacontrol/lqr/lqr.py, sofacontrol/lqr/traj_tracking_lqr.py
 * ===================================================================================================== */
/* TrajTrackingLQR.perform_dlqr_recursion (traj_tracking_lqr.py:18-48) for per-step (A_i, B_i),
 * i = 0..n-1 in forward time order, terminal P = Q: K (n x n_u x n_x), P (n+1 x n_x x n_x) or NULL. */
int sric_tvlqr(const double *A, const double *B, int n_steps, int n_x, int n_u, const double *Q,
               const double *R, double *K, double *P);
/* the same with the linearisation taken from a TPWL handle at the nominal states xbar (n x n_x) */
int sric_tvlqr_tpwl(stpwl_t *h, const double *xbar, int n_steps, const double *Q, const double *R,
                    double *K, double *P);
/* solve_riccati (lqr.py:6-21): fixed point until ||L - L_old||_F <= tol (reference: 1e-4);
 * `batch` independent (A,B) pairs share Q, R.  L (batch x n_u x n_x), P (batch x n_x x n_x). */
int sric_dare_fixed_point(const double *A, const double *B, int64_t batch, int n_x, int n_u,
                          const double *Q, const double *R, double tol, int max_iter, double *L,
                          double *P, int32_t *iters);
/* dare (lqr.py:24-31, scipy.linalg.solve_discrete_are in the reference; per TPWL point at the start-up of the scp
 * controller, tpwl/controllers.py:238-246): the stabilising solution by the structure-preserving doubling algorithm,
 * quadratic convergence (about a dozen steps).  Stops when max|H_k+1 - H_k| <= tol max|H_k+1|.  Same shapes as above;
 * iters = doubling steps.  SRH_ENUMERIC when R or R + B^T P B is not positive definite, I + G H is singular, or
 * max_iter steps do not converge. */
int sric_dare(const double *A, const double *B, int64_t batch, int n_x, int n_u, const double *Q, const double *R,
              double tol, int max_iter, double *L, double *P, int32_t *iters);

/* =====================================================================================================
 * iLQR.                           reference: sofacontrol/lqr/ilqr.py, sofacontrol/lqr/config.py
 * ===================================================================================================== */
typedef struct silqr_params {
    int    max_iter;          /* config.py:3   50  */
    double epsilon;           /* config.py:4   0.1 */
    double alpha0, alpha_scaling, improv_lb, improv_ub, alpha_min;   /* config.py:13-17 */
    int    counter_limit;     /* config.py:19  5   */
    double rho0, drho0, rho_scaling, rho_increase_fp, rho_max, rho_min;  /* config.py:25-30 */
    /* the four switches of config.py:6-9, 31 (all 1 in the reference's configuration):
     *   include_input_var_constraint  1: the input term of the cost is (u_t - u_{t-1})' R (u_t - u_{t-1}) (u_{-1} = u_last),
     *                                 0: u_t' R u_t                                          (ilqr.py:145-152, 250-256)
     *   do_linesearch                 0: the first forward pass (alpha0) is always accepted     (ilqr.py:75-88)
     *   regularize                    0: Q~_uu = Q_uu, Q~_ux = Q_ux; a Q_uu that is not positive definite ends the solve with
     *                                    iters = -1 (the reference prints a warning and goes on with the inverse of an
     *                                    indefinite matrix, ilqr.py:276-287)
     *   state_regularization          0: Q~_uu = Q_uu + rho I, Q~_ux = Q_ux                    (ilqr.py:264-270) */
    int    include_input_var_constraint, do_linesearch, regularize, state_regularization;
} silqr_params;
void silqr_default_params(silqr_params *p);

/* iLQR.ilqr_computation (ilqr.py:27-107) for `batch` independent problems on one TPWL model:
 *   x0 (batch x n_x), z_target (batch x (N+1) x n_z, absolute: includes z_ref), u_warm (batch x N x n_u)
 *   or NULL, u_last (batch x n_u) or NULL; Q, Qf (n_z x n_z), R (n_u x n_u).
 * Outputs x (batch x (N+1) x n_x), u (batch x N x n_u), K (batch x N x n_u x n_x), cost (batch),
 * iters (batch). */
int silqr_solve(stpwl_t *h, int N, int64_t batch, const double *x0, const double *z_target,
                const double *u_warm, const double *u_last, const double *Q, const double *R,
                const double *Qf, const silqr_params *p, double *x, double *u, double *K,
                double *cost, int32_t *iters);
/* The same on an SSM model: (A_t, B_t, d_t) = SSMDynamics.get_jacobians(x_t, u_t, dt) at every step of every
 * forward pass (ilqr.py:155), z = C_map(x) + z_ref in the costs, the model's constant H in the cost
 * Jacobians (ilqr.py:176-184).  mode = SSSM_FE / BE / BIL / DISCRETE_MAP. */
int silqr_solve_ssm(sssm_t *h, int mode, double dt, int N, int64_t batch, const double *x0,
                    const double *z_target, const double *u_warm, const double *u_last, const double *Q,
                    const double *R, const double *Qf, const silqr_params *p, double *x, double *u,
                    double *K, double *cost, int32_t *iters);

/* =====================================================================================================
 * LOCP (the horizon QP) and GuSTO. reference: sofacontrol/scp/locp.py, sofacontrol/scp/gusto.py
 * ===================================================================================================== */
typedef struct slocp_problem {
    int N, n_x, n_u, n_z;
    const double *H;        /* (n_z x n_x)                          locp.py:29          */
    const double *Qz, *R;   /* (n_z x n_z), (n_u x n_u)             locp.py:30-31       */
    const double *Qzf;      /* (n_z x n_z) or NULL                  locp.py:32, 251-252 */
    const double *x_scale;  /* (n_x,) = 1/|x_char| or NULL (ones)   locp.py:47-51       */
    int nU;  const double *UA, *Ub;    /* U.A (nU x n_u), U.b       locp.py:300-303     */
    int nX;  const double *XA, *Xb;    /* X.A (nX x n_x), X.b       locp.py:330-333     */
    int nXf; const double *XfA, *Xfb;  /* terminal set              locp.py:336-337     */
    int ndU; const double *dUA, *dUb;  /* dU.A (ndU x n_u), dU.b    locp.py:305-308: must be 0 here -- the rate rows
                                        * couple consecutive stages; the host layer passes them as state rows of the
                                        * augmented state [x; u_prev; du] (sofacontrol_amd/scp/locp.py) */
    int tr_active;                     /* is_tr_active              locp.py:57          */
} slocp_problem;

/* LOCP.update + solve + get_solution (locp.py:98-203) for `batch` independent QPs of one shape:
 *   Ad (batch x N x n_x x n_x), Bd (batch x N x n_x x n_u), dd (batch x N x n_x), x0 (batch x n_x),
 *   xk (batch x (N+1) x n_x) trust-region centre, delta/omega (batch), z (batch x (N+1) x n_z) or NULL,
 *   zf (batch x n_z) or NULL, u_des (batch x N x n_u) or NULL.
 * Outputs x (batch x (N+1) x n_x), u (batch x N x n_u), s (batch x (N+1)), J (batch; objective value
 * without the 1/2, as cvxpy reports), status (batch; 0 = optimal, else the reference's `success=False`). */
int slocp_solve(const slocp_problem *prob, int64_t batch, const double *Ad, const double *Bd,
                const double *dd, const double *x0, const double *xk, const double *delta,
                const double *omega, const double *z, const double *zf, const double *u_des,
                double *x, double *u, double *s, double *J, int32_t *status, int32_t *iters);

/* The same QP with everything resident (round 3): constants, horizon buffers, work blocks and result buffers are created once
 * for a fixed batch and reused by every solve -- what the host loops around the device QP call once per SCP iteration
 * (GuSTO over SSM / weighting-mode models: scp/gusto.py:371-402; linear MPC: the LOCP of locp.py with is_tr_active=False).
 * slocp_plan_solve: host pointers as slocp_solve; Ad = Bd = dd = NULL keeps the horizon of the previous call resident
 * (LOCP.update(full=False), locp.py:139-141: only delta / omega / x0 change); xk, z, zf, u_des NULL keep theirs.
 * slocp_plan_solve_dev: every array already in HBM (the linearisation kernels' outputs can be passed straight in), results
 * written to device pointers, asynchronous on `stream`; s_dev / iters_dev may be NULL. */
typedef struct slocp_plan slocp_plan_t;
int slocp_plan_create(slocp_plan_t **out, const slocp_problem *prob, int64_t batch);
void slocp_plan_destroy(slocp_plan_t *plan);
int slocp_plan_solve(slocp_plan_t *plan, const double *Ad, const double *Bd, const double *dd, const double *x0,
                     const double *xk, const double *delta, const double *omega, const double *z, const double *zf,
                     const double *u_des, double *x, double *u, double *s, double *J, int32_t *status, int32_t *iters);
int slocp_plan_solve_dev(slocp_plan_t *plan, const double *Ad_dev, const double *Bd_dev, const double *dd_dev,
                         const double *x0_dev, const double *xk_dev, const double *delta_dev, const double *omega_dev,
                         const double *z_dev, const double *zf_dev, const double *ud_dev, double *x_dev, double *u_dev,
                         double *s_dev, double *J_dev, int32_t *status_dev, int32_t *iters_dev, void *stream);

/* Which kernels a plan launches -- for tests and bench records (parity is claimed per instantiation; what ran is otherwise
 * visible only in a rocprof trace).  family: 1 = the lean condensed kernel first (csrc/lean.hip) with the fused kernel
 * (csrc/gusto.hip / scp.hip) taking what it hands over, 0 = the fused kernel alone.  lean_args = the template arguments of
 * the lean instantiation <n_u, n_x (0: run time), lanes per stage for the state rows (0: general row handling), horizon
 * (0: run time), first LDS-resident stage, state rows> (zeros when family = 0); fused_args = <split panel, n_u, n_x> of the
 * fused instantiation (0: that extent is a run-time value).  handed_over = problems (LOCP) / rollouts (GuSTO) of the LAST
 * solve that the lean kernel passed to the fused one (-1: no solve yet; 0 when family = 0).  The call waits for the plan's
 * last solve.  The instantiation is chosen when the plan is created. */
typedef struct srh_kernel_info {
    int32_t family;
    int32_t lean_args[6];
    int32_t fused_args[3];
    int32_t handed_over;
    int32_t lds_bytes_lean, lds_bytes_fused;
    int32_t threads;
} srh_kernel_info;
int slocp_plan_info(slocp_plan_t *plan, srh_kernel_info *info);

/* Whether QPs of this shape take the condensed (output-space) interior point for their trust-region-free pass
 * (csrc/locp_cond.h): enabled, the number of output directions found (rows of C_o spanning Cq, X.A, Xf.A) and whether
 * the input Hessian blocks 2R + U.A^T D U.A are diagonal for every D.  For tests and records. */
int slocp_condensed_info(const slocp_problem *prob, int *enabled, int *n_outputs, int *diag_input_hessian);

typedef struct sgusto_params {
    double delta0, omega0, rho, beta_fail, gamma_fail, epsilon, omega_max, convg_thresh; /* gusto.py:12-22 */
    int    max_gusto_iters;
} sgusto_params;
void sgusto_default_params(sgusto_params *p);

/* GuSTO.solve (gusto.py:283-487) on a pre-discretised nn-TPWL model, `batch` independent rollouts:
 *   x0 (batch x n_x), u_init (batch x N x n_u), x_init (batch x (N+1) x n_x), z (batch x (N+1) x n_z)
 *   or NULL, zf, u_des as slocp_solve; x_char / f_char (n_x,) or NULL (ones); dt = horizon step.
 * Outputs xopt (batch x (N+1) x n_x), uopt (batch x N x n_u), zopt (batch x (N+1) x n_z) = H xopt
 * (gusto.py:486), iters (batch) = number of SCP iterations (LOCP solves) performed,
 * status (batch; 0 ok, 1 = a QP could not be solved (gusto.py:357-365), 2 = omega > omega_max,
 * 3 = max iterations), trace (batch x max_trace x 4: J, delta, omega, rho_k per iteration) or NULL. */
int sgusto_solve(stpwl_t *h, const slocp_problem *prob, const sgusto_params *par, double dt,
                 int64_t batch, const double *x0, const double *u_init, const double *x_init,
                 const double *z, const double *zf, const double *u_des, const double *x_char,
                 const double *f_char, double *xopt, double *uopt, double *zopt, int32_t *iters,
                 int32_t *status, double *trace, int max_trace);

/* Resident form of sgusto_solve: the plan owns the device copies of the problem constants and the
 * per-rollout workspace for a fixed batch size, so that repeated receding-horizon solves (the
 * gusto_callback of scp/ros.py:94-127) launch ONE kernel and touch no host memory.  The `_dev` entry
 * point takes device pointers laid out as in sgusto_solve and is asynchronous on `stream`. */
typedef struct sgusto_plan sgusto_plan_t;
int sgusto_plan_create(sgusto_plan_t **plan, stpwl_t *h, const slocp_problem *prob, const sgusto_params *par,
                       double dt, int64_t batch, const double *x_char, const double *f_char, int max_trace);
int sgusto_plan_destroy(sgusto_plan_t *plan);
int sgusto_plan_set_max_iters(sgusto_plan_t *plan, int max_gusto_iters);   /* gusto.py:142-147 */
/* Solver state across solves (sofacontrol/scp/locp.py:181 `self.prob.solve(warm_start=self.warm_start, ...)`: the reference's cvxpy
 * problem object lives as long as the GuSTO object, so with warm_start=True its solver starts EVERY QP -- also the first one of the next
 * GuSTO.solve -- from the previous solution).  on = 1: the first QP of a solve starts from the minimiser and multipliers the same rollout's
 * previous solve ended with (lean kernels; later QPs of a solve always do).  Default 0: every solve starts cold (what bench.py times). */
int sgusto_plan_set_warm_across(sgusto_plan_t *plan, int on);
/* *active = 1 iff the flag above is set AND this plan's kernels honour it (lean kernels with the box-row interior point); fused-only
 * plans and general-row lean variants start every solve cold whatever was requested. */
int sgusto_plan_warm_across_active(const sgusto_plan_t *plan, int *active);
/* GuSTO on an SSM polynomial model (sofacontrol/scp/models/ssm.py + sofacontrol/SSM/ssm.py:198-235), the whole solve -- analytic
 * linearisation of the dynamics and of the output map along the trajectory, the LOCP QP with per-stage output maps (locp.py:231-245,
 * 312-329), the tests and the acceptance rules of gusto.py:371-473 -- inside ONE kernel launch per call: the loop the reference's
 * hardware driver runs every control period (examples/hardware/diamond_SSM.py:353-361).  `prob`: the QP in the augmented state
 * [x ; zeta] (n_x = n + n_o, H = [0 I], x_scale = 0 on zeta) when the model's output map is nonlinear, else the plain QP with the
 * constant H; `mode`: discretisation as sssm_linearize; Hm (n_z x n): model.H of zopt = H xopt (gusto.py:486); XA / Xb (nX x n): the state
 * polyhedron as GuSTO.state_constraints_violated applies it to the states (gusto.py:185-201), or NULL.  Arrays as sgusto_solve with
 * n_x = the model's n.  status: 0 ok, 1 a QP could not be solved, 2 omega > omega_max, 3 max iterations. */
typedef struct sgusto_ssm_plan sgusto_ssm_plan_t;
int sgusto_ssm_plan_create(sgusto_ssm_plan_t **plan, sssm_t *model, const slocp_problem *prob, const sgusto_params *par, double dt,
                           int mode, int64_t batch, const double *f_char, const double *Hm, int nX, const double *XA, const double *Xb,
                           int max_trace);
int sgusto_ssm_plan_destroy(sgusto_ssm_plan_t *plan);
int sgusto_ssm_plan_set_max_iters(sgusto_ssm_plan_t *plan, int max_gusto_iters);
int sgusto_ssm_plan_set_warm_across(sgusto_ssm_plan_t *plan, int on);   /* as sgusto_plan_set_warm_across (locp.py:181) */
int sgusto_ssm_plan_solve(sgusto_ssm_plan_t *plan, const double *x0, const double *u_init, const double *x_init, const double *z,
                          const double *u_des, double *xopt, double *uopt, double *zopt, int32_t *iters, int32_t *status, double *trace);
int sgusto_ssm_plan_solve_dev(sgusto_ssm_plan_t *plan, const double *x0_dev, const double *u_init_dev, const double *x_init_dev,
                              const double *z_dev, const double *u_des_dev, double *xopt_dev, double *uopt_dev, double *zopt_dev,
                              int32_t *iters_dev, int32_t *status_dev, double *trace_dev, void *stream);

/* Which kernel instantiation a solve of this plan launches: split (1: split W panel, n_x > 64), n_u_fixed / n_x_fixed
 * = the compile-time n_u / n_x of the instantiation (0: that extent is a run-time value; 0, 0 = the all-sizes
 * kernel).  For tests and bench records: parity is claimed per instantiation. */
int sgusto_plan_variant(const sgusto_plan_t *plan, int *split, int *n_u_fixed, int *n_x_fixed);
/* Optimal LOCP value of the solution each rollout of the plan's LAST solve returned (the cost `Jstar` of the last accepted
 * SCP step, gusto.py:340-370; +inf for a rollout that never accepted a step): J (batch).  What a sharded batch gathers to
 * pick its best rollout (distributed.gather_rollout_costs).  Waits for the solve. */
int sgusto_plan_costs(sgusto_plan_t *plan, double *J);
/* srh_kernel_info of a GuSTO plan (see slocp_plan_info). */
int sgusto_plan_info(sgusto_plan_t *plan, srh_kernel_info *info);
int sgusto_plan_solve(sgusto_plan_t *plan, const double *x0, const double *u_init, const double *x_init,
                      const double *z, const double *zf, const double *u_des, double *xopt, double *uopt,
                      double *zopt, int32_t *iters, int32_t *status, double *trace);
int sgusto_plan_solve_dev(sgusto_plan_t *plan, const double *x0, const double *u_init, const double *x_init,
                          const double *z, const double *zf, const double *u_des, double *xopt, double *uopt,
                          double *zopt, int32_t *iters, int32_t *status, double *trace, void *stream);

/* Asynchronous form of sgusto_plan_solve -- the `send_request(wait=False)` / `check_if_done` / `force_wait` protocol
 * of the reference's solver client (scp/ros.py:183-223; used by tpwl/controllers.py:276-292,327): `_begin` copies
 * the host inputs into the plan's pinned staging block, enqueues H2D copies, the solve and the D2H copies of the
 * results on the plan's own (non-blocking) stream and returns; `_done` polls the completion event without blocking;
 * `_end` waits for it and hands the results over.  One request per plan at a time. */
int sgusto_plan_prepare_async(sgusto_plan_t *plan);   /* optional: create the stream / pinned block now (~10 ms) */
int sgusto_plan_solve_begin(sgusto_plan_t *plan, const double *x0, const double *u_init, const double *x_init,
                            const double *z, const double *zf, const double *u_des, int want_trace);
int sgusto_plan_solve_done(sgusto_plan_t *plan, int *done);
int sgusto_plan_solve_end(sgusto_plan_t *plan, double *xopt, double *uopt, double *zopt, int32_t *iters,
                          int32_t *status, double *trace);

/* The device-side duration (ms) of the last request collected by sgusto_plan_solve_end: from the first host-to-device copy
 * to the last device-to-host copy on the plan's stream (HIP events) -- the solver's own time, what the reference reports
 * as its solve time (scp/ros.py:116-124), independent of when the caller came back for the result; -1 before the first. */
int sgusto_plan_last_async_ms(sgusto_plan_t *plan, double *ms);

#ifdef __cplusplus
}
#endif
#endif /* SOFACONTROL_HIP_H */
