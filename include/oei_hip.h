/*
 * oei_hip.h -- C ABI of the MI355X (gfx950) implementation of the Optimistic Expected
 * Improvement evaluation path.  All arithmetic is fp64.
 *
 * The reference has no FFI layer: its boundary is a set of Python callables
 * (methods/oei.py, methods/sdp.py) whose arithmetic runs inside third-party native code
 * (gpflow/TensorFlow kernels, SCS, MKL-Pardiso).  Each entry point below names the reference
 * callable (file:line under the reference root) whose arithmetic it replaces; the Python shim in
 * bayesian_optimization_amd/methods presents B=1 slices of them under the reference's names.
 *
 * Conventions
 *   - return value: 0 = success, negative = error (see OEI_ERR_*); nothing throws.
 *   - "host" pointers are ordinary process memory, "device" pointers are HIP device memory on the
 *     device that was current when the model was created (or is current, for the model-free calls).
 *   - the caller owns every input and output buffer; the library keeps internal scratch per model
 *     handle (or per device for model-free calls) and grows it on demand.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     stream-ordered; they return once the work is enqueued unless stated otherwise.
 *   - matrices are dense row-major.  k_ = q + 1.
 *   - per-candidate `status`: 0 = converged, 1 = Newton iteration limit reached (the result is the
 *     last iterate), 2 = numerical breakdown (Omega not positive definite, NaN; outputs are NaN).
 *     Like the reference (sdp.py:53-56: warn and continue) a non-zero status never aborts the
 *     batch.  The eigen-iteration (Jacobi) has a cap of 30 sweeps per evaluation: a solve whose
 *     final evaluation stopped at that cap also reports status 1; the sweep total is in the high
 *     half of `iters`.
 */
#ifndef OEI_HIP_H
#define OEI_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OEI_OK            0
#define OEI_ERR_ARG      -1   /* bad argument (null pointer, size out of range) */
#define OEI_ERR_HIP      -2   /* a HIP runtime call failed; see oei_last_error() */
#define OEI_ERR_NOT_PD   -3   /* K + noise*I is not positive definite */
#define OEI_ERR_NO_DEVICE -4  /* no HIP device present */

#define OEI_KERNEL_RBF       0   /* variance * exp(-r2/2)                  run.py:80 */
#define OEI_KERNEL_MATERN32  1   /* variance * (1+sqrt3 r) exp(-sqrt3 r)   run.py:19-23,81 */

#define OEI_FLOOR_NONE   0   /* r = sqrt(r2) */
#define OEI_FLOOR_SAFE   1   /* r = sqrt(max(r2, 1e-40))   run.py:19-23 (SafeMatern32) */
#define OEI_FLOOR_STOCK  2   /* r = sqrt(r2 + 1e-12)       tests/create_model.py:28 (stock Matern32) */

#define OEI_MAX_Q   64       /* largest batch size q the kernels accept */
#define OEI_NOALLOC_SCRATCH_DOUBLES (4u << 20)   /* fixed scratch of oei_value_grad_batched_noalloc: 32 MiB per GPU */
#define OEI_MAX_DIM 32       /* largest input dimension d */

typedef struct oei_model oei_model;   /* opaque GP model handle */

/* Human-readable description of the last error on this thread (never NULL). */
const char* oei_last_error(void);

/* Library/ABI version, for the loader to check. */
int oei_abi_version(void);

/* ---- model lifecycle ---------------------------------------------------------------------------
 * Replaces what gpflow's GPR recomputes inside every Session.run of the reference
 * (oei.py:27,33,34,49,58: K = k(X,X)+noise*I, its Cholesky factor, L^-1 y) plus the fmin of
 * oei.py:27,49 (minimum posterior MEAN at the training inputs).  Done once per model here.
 * X_train [n,d] and y [n] are host pointers; lengthscales has `n_lengthscales` entries (1 or d,
 * run.py:94 ARD flag).  Synchronous. */
int oei_model_create(oei_model** out, int n, int d,
                     const double* X_train, const double* y,
                     int kernel_kind, int floor_mode, double variance,
                     const double* lengthscales, int n_lengthscales, double noise);
int oei_model_destroy(oei_model* m);
/* Same data, new hyper-parameters: recomputes K, its factor, L^-1, L^-1 y and fmin in the handle's
 * existing buffers (no device allocation or release).  This is one objective evaluation of the
 * hyper-parameter fit (bo.py:194-221: gpflow's optimize() re-runs the Cholesky of the same X, y for
 * every trial parameter vector).  Synchronous, and ordered after everything already enqueued on
 * the model's device: it first waits for the device (hipDeviceSynchronize), because evaluations
 * of this handle that are still in flight on any stream read the factors it rewrites.  Evaluations
 * enqueued after it returns see the new factors.  After OEI_ERR_NOT_PD the handle stays valid for
 * another oei_model_refactor, but not for evaluations. */
int oei_model_refactor(oei_model* m, int kernel_kind, int floor_mode, double variance,
                       const double* lengthscales, int n_lengthscales, double noise);
/* fmin, n, d of a model (host outputs). */
int oei_model_info(const oei_model* m, double* fmin, int* n, int* d);
/* Copies the n x n lower Cholesky factor L of K + noise*I to a host buffer (diagnostics/tests). */
int oei_model_get_factor(const oei_model* m, double* L_host);

/* ---- GP posterior ------------------------------------------------------------------------------
 * predict_f (gpflow GPR.predict_f, called at oei.py:27,49 and tests/test_derivatives.py:45):
 * X device [m,d] -> mean device [m], var device [m] (marginal variances, noise NOT added). */
int oei_predict_f(oei_model* m, int npts, const double* X, double* mean, double* var, void* stream);

/* omega (oei.py:79-103): X device [B,q,d] -> omega device [B,k_,k_]. */
int oei_omega_batched(oei_model* m, int B, int q, const double* X, double* omega, void* stream);

/* ---- the SDP -----------------------------------------------------------------------------------
 * sdp (sdp.py:11-67): omega device [B,k_,k_], fmin device [B] ->
 *   value device [B] (= <Omega,M>, <= 0), M device [B,k_,k_],
 *   Y device [B,k_,k_,k_] or NULL (the k_ dual matrices Y_i, sdp.py:236-247),
 *   status device int[B], iters device int[B] (Newton iterations; sweeps in the high 16 bits).
 * Independent of any GP model.  Requires omega[b] positive definite. */
int oei_sdp_batched(int B, int k_, const double* omega, const double* fmin,
                    double* value, double* M, double* Y, int* status, int* iters, void* stream);

/* solution_derivative (sdp.py:159-206): the sensitivity of the SDP solution to K perturbations of
 * Omega.  M [B,k_,k_], Y [B,k_,k_,k_], fmin [B], domega [B,k_,k_,K] -> dM [B,k_,k_,K]
 * (and dY [B,k_,k_,k_] when K == 1 and dY != NULL, sdp.py:195-204).  All device pointers. */
int oei_solution_derivative_batched(int B, int k_, int K, const double* M, const double* Y,
                                    const double* fmin, const double* domega,
                                    double* dM, double* dY, int* status, void* stream);

/* ---- fused acquisition -------------------------------------------------------------------------
 * acquisition (oei.py:19-39 without `project`, which stays on the host): X device [B,q,d] ->
 * value device [B], grad device [B,q,d], status device int[B].  `flags` device int[B] or NULL
 * receives 1 where `project` (oei.py:157-199) would remove a point of candidate b (a point within
 * 1e-2 lengthscales of a training input in every coordinate, or two points equal after rounding
 * X/l to two decimals); the host shim re-evaluates those candidates on the reduced point set. */
int oei_value_grad_batched(oei_model* m, int B, int q, const double* X,
                           double* value, double* grad, int* status, int* flags, void* stream);

/* The same evaluation started from the solver state of a nearby evaluation: the role of the
 * reference's SDP warm start (sdp.py:41-48, 63-65, 77-121: history of past solutions, the nearest
 * one extrapolated; bo.py:146 resets it per multistart).  Here the caller keeps the history:
 * start_in device [B,q] or NULL holds, per candidate, the state a previous call wrote for a nearby X
 * (opaque: the ratio of the optimal to the closed-form start weights; a NaN, or a product outside
 * the simplex, means "no start" and the closed-form start is used); start_out device [B,q] or NULL
 * receives this solve's state (may equal start_in: a candidate's state is read before it is
 * written and the kernels do not assume the two are distinct); iters device int[B] or NULL the Newton
 * iteration count (low 16 bits) and Jacobi sweeps (high bits).  The converged result does not
 * depend on the start beyond the solver's tolerance (about 1e-10 relative on value and gradient). */
int oei_value_grad_warm_batched(oei_model* m, int B, int q, const double* X,
                                double* value, double* grad, int* status, int* flags,
                                const double* start_in, double* start_out, int* iters, void* stream);

/* The same evaluation for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Identical arguments and results (bitwise) to oei_value_grad_batched, but the intermediates
 * (kernel rows, A = Ks L^-T, Z, per-candidate state) live in a fixed 32 MiB array that every GPU
 * the library runs on holds until the process ends; the call performs no hipMalloc/hipFree.
 * Limit: 3 * roundup(B*q, 64) * roundup(n, 64) + B*q*(1 + q + (d+1) * roundup(n,64)/64) doubles must
 * fit in OEI_NOALLOC_SCRATCH_DOUBLES (about 200 candidates at q = 20, n = 310); larger calls
 * return OEI_ERR_ARG with both numbers in oei_last_error().  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (different streams or threads) race.  The model's
 * device must be the current device (OEI_ERR_ARG otherwise).  The per-device address table is
 * filled without a lock by oei_model_create, or by the first call on a device: create the first
 * model of each device (or make that first call) from one thread. */
int oei_value_grad_batched_noalloc(oei_model* m, int B, int q, const double* X,
                                   double* value, double* grad, int* status, int* flags, void* stream);

/* acquisition_hessian (oei.py:41-64, 105-135): X device [B,q,d] -> H device [B,q*d,q*d],
 * status device int[B] or NULL (that of the candidate's SDP solve).
 * q <= 32: the per-candidate tables live in LDS, which serves d up to about 12 (q = 20) or 7
 * (q = 32).  q > 32: the tables live in global memory and every q <= OEI_MAX_Q is served for
 * d <= 12 (q = 50, d = 10 included).  Shapes beyond that return OEI_ERR_ARG with the byte count in
 * oei_last_error(). */
int oei_hessian_batched(oei_model* m, int B, int q, const double* X,
                        double* H, int* status, void* stream);

/* The duplicate tests of `project` (oei.py:181-192) without an evaluation: flags device int[B]
 * receives 1 where project would remove a point of candidate b (same two tests as the `flags` of
 * oei_value_grad_batched), else 0.  acquisition_hessian returns zeros for such a candidate
 * (oei.py:53-55); callers of oei_hessian_batched that want that behaviour mask its output with
 * these flags. */
int oei_project_flags(oei_model* m, int B, int q, const double* X, int* flags, void* stream);

/* domega (oei.py:125-135): X device [B,q,d] -> domega device [B,k_,k_,q*d]. */
int oei_domega_batched(oei_model* m, int B, int q, const double* X, double* domega, void* stream);

/* ---- measurement ---------------------------------------------------------------------------------
 * No counterpart in the reference (its only timing hook is a commented-out print, sdp.py:52).
 * With profiling on, oei_value_grad_batched records HIP events on its stream around each of its
 * five kernels; oei_get_stage_times waits for the last call and returns their durations in
 * milliseconds: [0] gram, [1] GEMM Ks L^-T, [2] moments+SDP, [3] GEMM Z L^-1, [4] gradient. */
int oei_set_profiling(oei_model* m, int on);
int oei_get_stage_times(oei_model* m, float* ms5);
/* Duration (ms) of the moments+SDP kernel of the last profiled call, from start and stop events
 * attached to its dispatch (hipExtLaunchKernel): the kernel alone, as a kernel trace reports it,
 * where interval [2] above also holds the launch gap that an event between two kernels opens. */
int oei_get_solver_kernel_time(oei_model* m, float* ms);
/* No counterpart in the reference.  oei_value_grad_batched and oei_value_grad_warm_batched run
 * batches of at least two full rounds of the solver kernel (B >= 16 x compute units, q <= 32) as two
 * halves on two internal streams, so that one half's GEMM and kernel-row launches fill the compute
 * units the other half's solver waves have already left; the caller's stream waits for both, and
 * the results are bitwise those of the single-stream call.  On by default; off keeps every launch
 * on the caller's stream (per-kernel profiles of whole-batch launches; calls under stream
 * capture, since the first pipelined call creates the streams).  `on` = 1 selects the rule above;
 * a value n >= 2 is a diagnostic: n pieces, alternating between the two streams, at any batch size. */
int oei_set_pipelining(oei_model* m, int on);
/* No counterpart in the reference.  With `on` != 0 the value + gradient entry points skip the
 * per-candidate product Z = M11 A in the solver kernel: the second matrix product runs on the A rows
 * (A L^-1 does not depend on the SDP solution) and M11 is applied to every 64 x 64 tile before the
 * gradient contraction, so Z is neither written nor read (420 MB less HBM traffic at q = 20,
 * n = 310, B = 4096 for 15 % more matrix instructions in that product).  Same results up to the
 * order of the sums.  Off by default (measured no faster at that shape); d <= 11 only, otherwise
 * ignored. */
int oei_set_z_free(oei_model* m, int on);

/* ---- hyper-parameter objective ------------------------------------------------------------------
 * Negative log marginal likelihood of the GP behind the handle and its gradient, the objective of
 * the reference's hyper-parameter fit (gpflow's optimize() called from BO.optimize_restarts,
 * bo.py:85,194-221).  value: host double[1].  grad (host, may be NULL):
 * [d/d variance, d/d lengthscale (n_lengthscales entries: 1 shared or d ARD), d/d noise]. */
int oei_model_nlml(oei_model* m, double* value, double* grad, int n_lengthscales);

/* ---- multistart selection ----------------------------------------------------------------------
 * Local half of "keep the best start" (bo.py:158-161): arg-min over value[0..B) with ties to the
 * lowest index.  Writes best_value (device double[1]) and best_index (device long long[1], offset
 * by index_base).  The cross-rank exchange is a torch.distributed all_gather of the
 * (value, index, X[q,d]) record on the host side. */
int oei_argmin(int B, const double* value, long long index_base,
               double* best_value, long long* best_index, void* stream);

/* The same selection with the winner's whole record in one launch: record device double[2 + xlen]
 * receives value, the 64-bit global index (bit pattern in record[1]; -1 if no finite value) and
 * X[best] (xlen = q*d doubles) -- the unit of the cross-rank all-gather. */
int oei_best_record(int B, const double* value, const double* X, int xlen, long long index_base,
                    double* record, void* stream);

/* Selection among the gathered records of all ranks (the step after the all-gather; the reference
 * keeps the best start of its sequential loop at bo.py:157-161): records device double[nrec, reclen]
 * laid out as oei_best_record writes them -> out device double[reclen] = the record with the
 * smallest value, NaN counting as +inf and ties going to the smallest index (so every rank count
 * selects the same start).  One launch in place of a dozen small tensor operations. */
int oei_select_record(int nrec, int reclen, const double* records, double* out, void* stream);

/* ---- batched objective of the hyper-parameter fit ------------------------------------------------
 * bo.py:194-221 restarts gpflow's optimize() `model_restarts` times on the same data; the restarts
 * are independent, so R trial parameter vectors are evaluated in one pass (batch item z owns one
 * set of npad x npad factors; same kernels as oei_model_create).  theta [R, nt] (device) holds
 * log(variance), log(lengthscale) x n_lengthscales (1 or d) and, when fixed_noise < 0,
 * log(noise); with fixed_noise >= 0 the noise variance is that constant (run.py:99).  value [R]
 * and grad [R, nt] (device) receive the negative log marginal likelihood, plus the negative
 * log-priors of run.py:95-97 when `priors` != 0, and its gradient with respect to theta; +inf and
 * a zero gradient where K + noise I is not positive definite.  Asynchronous on `stream`.
 * X_train [n,d], y [n] are host pointers; max_batch bounds R (device memory: 4 max_batch npad^2
 * doubles). */
typedef struct oei_fit oei_fit;
int oei_fit_create(oei_fit** out, int n, int d, const double* X_train, const double* y,
                   int kernel_kind, int floor_mode, int max_batch);
int oei_fit_destroy(oei_fit* f);
int oei_fit_nlml_batched(oei_fit* f, int R, const double* theta, int n_lengthscales,
                         double fixed_noise, int priors, double* value, double* grad, void* stream);

/* ---- multistart descent on the device ------------------------------------------------------------
 * The loop over random starts of bo.py:129-192 (scipy L-BFGS-B per start, solvers.py:10-21) as B
 * bounded L-BFGS runs that advance together: each round the caller evaluates the objective at a
 * dense list of trial points and hands value and gradient back; one wave per run then applies the
 * Armijo test, stores the curvature pair, tests for convergence (projected gradient <= pgtol,
 * relative decrease <= ftol: the rules of L-BFGS-B), forms the next two-loop direction and the next
 * trial point.  The objective is whatever the caller evaluates (the acquisition through
 * oei_value_grad_batched, or the hyper-parameter likelihood): n variables per run, box bounds.
 *
 * State lives in two caller-allocated device workspaces (doubles and ints); oei_lbfgs_layout gives
 * their sizes and the element offsets of the arrays in them:
 *   doubles  X [B,n] iterate | F [B] value | G [B,n] gradient | DIR [B,n] | STEP [B] | XT [B,n]
 *            trial point per run | S, Y [B,history,n] pairs | RHO [B,history] | GAMMA [B] |
 *            XC [B,n] dense list of trial points (rows 0..count) | LO, HI [n]
 *   ints     STATUS [B] (OEI_LBFGS_*) | SLOT | NIT [B] accepted steps | NFEV [B] evaluations | LS |
 *            FIRST | ROWOF [B] row of run r in the list or -1 | MAP [B] run of list row c |
 *            COUNT [1] | NEED [B] */
#define OEI_LBFGS_D_X 0
#define OEI_LBFGS_D_F 1
#define OEI_LBFGS_D_G 2
#define OEI_LBFGS_D_DIR 3
#define OEI_LBFGS_D_STEP 4
#define OEI_LBFGS_D_XT 5
#define OEI_LBFGS_D_S 6
#define OEI_LBFGS_D_Y 7
#define OEI_LBFGS_D_RHO 8
#define OEI_LBFGS_D_GAMMA 9
#define OEI_LBFGS_D_XC 10
#define OEI_LBFGS_D_LO 11
#define OEI_LBFGS_D_HI 12
#define OEI_LBFGS_ND 13
#define OEI_LBFGS_I_STATUS 0
#define OEI_LBFGS_I_SLOT 1
#define OEI_LBFGS_I_NIT 2
#define OEI_LBFGS_I_NFEV 3
#define OEI_LBFGS_I_LS 4
#define OEI_LBFGS_I_FIRST 5
#define OEI_LBFGS_I_ROWOF 6
#define OEI_LBFGS_I_MAP 7
#define OEI_LBFGS_I_COUNT 8
#define OEI_LBFGS_I_NEED 9
#define OEI_LBFGS_NI 10

#define OEI_LBFGS_RUNNING      0
#define OEI_LBFGS_CONV_PG      1   /* projected gradient below pgtol */
#define OEI_LBFGS_CONV_F       2   /* relative decrease below ftol */
#define OEI_LBFGS_LS_FAILED    3   /* max_ls trial points rejected */
#define OEI_LBFGS_STALLED      4   /* zero search direction */
#define OEI_LBFGS_BAD_START    5   /* objective not finite at the start */
#define OEI_LBFGS_ITER_LIMIT   6

/* Host-only: sizes (elements) of the two workspaces and, optionally, the offsets of the
 * OEI_LBFGS_ND / OEI_LBFGS_NI arrays inside them. */
int oei_lbfgs_layout(int B, int n, int history, size_t* n_doubles, size_t* n_ints,
                     size_t* off_doubles, size_t* off_ints);
/* x = X0 clipped to the box (X0 [B,n], lower/upper [n], device); every run asks for its start:
 * XC = x, COUNT = B. */
int oei_lbfgs_init(int B, int n, int history, const double* X0, const double* lower,
                   const double* upper, double* dws, int* iws, void* stream);
/* ft [count], gt [count,n] (device): objective and gradient at the rows of XC written by the
 * previous call (+inf / NaN values are rejected trial points).  Advances every running run and
 * writes the next list.  With count_host != NULL the call waits for the stream and returns the
 * new COUNT there (0: all runs have stopped). */
int oei_lbfgs_advance(int B, int n, int history, const double* ft, const double* gt,
                      double pgtol, double ftol, double c1, int max_ls, int max_iter,
                      double* dws, int* iws, int* count_host, void* stream);

/* The same for the projected Newton iteration that consumes the exact Hessian (the role of the
 * reference's SQP solver with `--hessian 1`, solvers.py:56-69, 71-141): a running run waits either
 * for value and gradient at its trial point (list XC, COUNTS[0] rows) or for the Hessian at its
 * iterate (list XH, COUNTS[1] rows).  Direction: (H + lam I) d = -g on the variables not blocked at
 * a bound, lam from the ladder 0, 1e-4 s, 1e-3 s, ... (s = max |H_ij|, 12 rungs) until the matrix is
 * positive definite, else steepest descent; the climb starts two rungs below the rung the run's
 * previous iteration ended on (RUNG), so that a run far from an optimum does not repeat the failed
 * factorisations of the low rungs every iteration; projected backtracking (step / 4), stopping rules as
 * above.  The factor lives in LDS: n <= OEI_NEWTON_MAX_N variables per run.
 *   doubles  X | F | G | DIR | STEP | XT | XC [B,n] trial list | XH [B,n] Hessian list | LO | HI
 *   ints     STATUS (OEI_LBFGS_* codes) | NIT | NFEV | NHEV Hessian evaluations | LS | PHASE |
 *            ROWOF row of run r in its list | MAP_T, MAP_H run of a list row | COUNTS [2] | NEED | RUNG */
#define OEI_NEWTON_MAX_N 141
#define OEI_NEWTON_D_X 0
#define OEI_NEWTON_D_F 1
#define OEI_NEWTON_D_G 2
#define OEI_NEWTON_D_DIR 3
#define OEI_NEWTON_D_STEP 4
#define OEI_NEWTON_D_XT 5
#define OEI_NEWTON_D_XC 6
#define OEI_NEWTON_D_XH 7
#define OEI_NEWTON_D_LO 8
#define OEI_NEWTON_D_HI 9
#define OEI_NEWTON_ND 10
#define OEI_NEWTON_I_STATUS 0
#define OEI_NEWTON_I_NIT 1
#define OEI_NEWTON_I_NFEV 2
#define OEI_NEWTON_I_NHEV 3
#define OEI_NEWTON_I_LS 4
#define OEI_NEWTON_I_PHASE 5
#define OEI_NEWTON_I_ROWOF 6
#define OEI_NEWTON_I_MAP_T 7
#define OEI_NEWTON_I_MAP_H 8
#define OEI_NEWTON_I_COUNTS 9
#define OEI_NEWTON_I_NEED 10
#define OEI_NEWTON_I_RUNG 11
#define OEI_NEWTON_NI 12
int oei_newton_layout(int B, int n, size_t* n_doubles, size_t* n_ints,
                      size_t* off_doubles, size_t* off_ints);
int oei_newton_init(int B, int n, const double* X0, const double* lower, const double* upper,
                    double* dws, int* iws, void* stream);
/* ft [COUNTS[0]], gt [COUNTS[0], n]: objective and gradient at the rows of XC; Ht [COUNTS[1], n, n]:
 * Hessians at the rows of XH (non-finite entries count as zero; either list may be empty, its
 * pointers are then not read; a NULL ft/gt while runs wait for a trial value rejects those trial
 * points, a NULL Ht while runs wait for a Hessian gives them a zero matrix -- no fault).  counts_host (int[2], optional): waits for the stream and returns
 * the new list lengths; both zero: all runs have stopped. */
int oei_newton_advance(int B, int n, const double* ft, const double* gt, const double* Ht,
                       double pgtol, double ftol, double c1, int max_ls, int max_iter,
                       double* dws, int* iws, int* counts_host, void* stream);

/* ---- rounds of the multistart driven inside the library -------------------------------------------
 * The loop over rounds of bo.py:129-192's multistart for the acquisition of model m: per round
 * the library evaluates the dense list(s) of the running runs itself (oei_value_grad_batched,
 * oei_hessian_batched with the Hessian zeroed where `project` would shrink the candidate,
 * oei.py:53-55), advances every run (a solve that failed is a rejected trial point) and reads the
 * list length(s) back; the caller's Python loop pays per round what the device work costs at 20
 * starts.  State: the workspaces of oei_lbfgs_init / oei_newton_init (n = q d variables per run).
 * work: device scratch of oei_multistart_workspace(B, n, with_hessian) doubles.
 * use_flags bit 0 (for noise <= 1e-4, oei.py:174): a trial point that `project` would shrink stops
 * the loop before its run is advanced -- the run and its trial point stay in the list; bit 1: the
 * Hessian of such a candidate is zeroed.  The caller
 * then evaluates the current list itself (the reduced problem draws random directions on the
 * host, oei.py:186-187), advances once with oei_lbfgs_advance / oei_newton_advance and calls again.
 * info (host): L-BFGS in [count] out [count, flagged runs, rounds done]; Newton in [trial count,
 * Hessian count] out [trial count, Hessian count, flagged runs, rounds done].  Returns after
 * max_rounds rounds, when no run is left, or at a flagged trial point.  Synchronous. */
int oei_multistart_workspace(int B, int n, int with_hessian, size_t* n_doubles);
int oei_lbfgs_minimize_rounds(oei_model* m, int B, int q, int history,
                              double pgtol, double ftol, double c1, int max_ls, int max_iter,
                              double* dws, int* iws, double* work, int use_flags,
                              int max_rounds, int* info, void* stream);
int oei_newton_minimize_rounds(oei_model* m, int B, int q,
                               double pgtol, double ftol, double c1, int max_ls, int max_iter,
                               double* dws, int* iws, double* work, int use_flags,
                               int max_rounds, int* info, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OEI_HIP_H */
