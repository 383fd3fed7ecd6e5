// Shared declarations for the gfx950 OEI library: error plumbing, the model handle, kernel
// parameter block and small device helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include "../../include/oei_hip.h"

#define OEI_ABI_VERSION 1
#define OEI_NPAD 64            // training size is padded to a multiple of this (GEMM tile)

void oei_set_error(const char* fmt, ...);

#define OEI_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            oei_set_error("%s failed at %s:%d: %s", #expr, __FILE__, __LINE__,           \
                          hipGetErrorString(_e));                                        \
            return OEI_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

// Kernel hyper-parameters, passed by value to kernels.
struct KernParams {
    int kind;                       // OEI_KERNEL_*
    int floor_mode;                 // OEI_FLOOR_*
    int d;
    double variance;
    double inv_ls2[OEI_MAX_DIM];    // 1 / lengthscale^2 per dimension
    double inv_ls[OEI_MAX_DIM];     // 1 / lengthscale
    double ls[OEI_MAX_DIM];         // lengthscale
};

struct oei_model {
    int device;
    int n, d, npad;
    double noise;
    double fmin;
    KernParams kp;
    double* X;        // [npad, d]   training inputs (padding rows are zero)
    double* L;        // [npad, npad] lower Cholesky factor of K + noise I (identity in the padding)
    double* Linv;     // [npad, npad] L^-1 (lower)
    double* LinvT;    // [npad, npad] L^-T (upper), row-major
    double* v;        // [npad] L^-1 y
    double* alpha;    // [npad] K^-1 y
    double* y;        // [npad]
    // scratch, grown on demand (row counts are in candidate points)
    size_t ws_rows;   // rows currently allocated in each of ws0/ws1
    double* ws0;      // [rows, npad]
    double* ws1;      // [rows, npad]
    size_t ws2_rows;  // rows allocated in ws2
    double* ws2;      // [rows, npad]: kernel-derivative factors kept for the gradient
    size_t aux_bytes;
    double* aux;      // per-candidate state (p, M11, ...)
    size_t hs_bytes;  // Hessian-path scratch (Gram blocks, Z/W rows)
    double* hs;
    // factorisation temporaries, kept so that oei_model_refactor allocates nothing
    double* Dinv;     // [npad, 64] inverses of the diagonal blocks
    double* ftmp;     // [npad, 64] per-column-block tile of the triangular inverse; training means
    double* dscal;    // [2] fmin | nlml value
    int* dflag;       // not-positive-definite flag
    double* Kinv;     // [npad, npad] K^-1 (hyper-parameter gradient), allocated on first use
    double* nlml_part;   // [n, 2 + d] per-row gradient terms, allocated on first use
    int profile;      // record stage events in value_grad
    hipEvent_t ev[6];
    hipEvent_t kev[2];   // start / stop of the solver kernel itself (taken from its dispatch, no launch gaps)
    int ev_valid;
    // side streams of the pipelined value+gradient call and the events that tie them to the
    // caller's stream, created on first use
    hipStream_t side[2];
    hipEvent_t ev_fork, ev_join[2];
    int pipeline;     // large value+gradient batches run as two halves on the side streams
    int zfree;        // second GEMM on the A rows with M11 applied in its epilogue: no Z buffer (oei_set_z_free)
    int n_cu;         // compute units of the device
};

// Diagnostic build (make poison, -DOEI_POISON_LDS): every kernel with dynamic LDS first fills it
// with NaN, so that a result that depends on what an earlier kernel left in LDS shows up as NaN
// (or as a status) in the test suite instead of once in a long sweep.  The size is the
// hidden_dynamic_lds_size kernel argument (code object v5: byte 120 of the implicit arguments).
#ifdef OEI_POISON_LDS
#define OEI_LDS_POISON(sm) do { \
        const unsigned n_ = ((const unsigned*)__builtin_amdgcn_implicitarg_ptr())[30] / 8u; \
        for (unsigned i_ = threadIdx.x; i_ < n_; i_ += blockDim.x) (sm)[i_] = __longlong_as_double(0x7ff8000000000000LL); \
        __syncthreads(); } while (0)
// the same build fills the handle's scratch buffers with NaN bytes before a call uses them
#define OEI_SCRATCH_POISON(m, st) do { \
        if ((m)->ws0) hipMemsetAsync((m)->ws0, 0xFF, (m)->ws_rows * (size_t)(m)->npad * sizeof(double), st); \
        if ((m)->ws1) hipMemsetAsync((m)->ws1, 0xFF, (m)->ws_rows * (size_t)(m)->npad * sizeof(double), st); \
        if ((m)->ws2) hipMemsetAsync((m)->ws2, 0xFF, (m)->ws2_rows * (size_t)(m)->npad * sizeof(double), st); \
        if ((m)->aux) hipMemsetAsync((m)->aux, 0xFF, (m)->aux_bytes, st); \
        if ((m)->hs) hipMemsetAsync((m)->hs, 0xFF, (m)->hs_bytes, st); } while (0)
#else
#define OEI_LDS_POISON(sm)
#define OEI_SCRATCH_POISON(m, st)
#endif

// ---- device helpers ---------------------------------------------------------------------------

// kappa(t) and (optionally) its first and second derivatives with respect to t = r2.
// k(x,x') = variance * kappa(r2).
template <int ORDER>
__device__ __forceinline__ void kappa_eval(int kind, int floor_mode, double t,
                                           double& k0, double& k1, double& k2) {
    if (kind == OEI_KERNEL_RBF) {
        double e = exp(-0.5 * t);
        k0 = e;
        if (ORDER >= 1) k1 = -0.5 * e;
        if (ORDER >= 2) k2 = 0.25 * e;
    } else {
        double r;
        bool live;
        if (floor_mode == OEI_FLOOR_STOCK) { r = sqrt(t + 1e-12); live = true; }
        else if (floor_mode == OEI_FLOOR_SAFE) { r = sqrt(fmax(t, 1e-40)); live = t > 1e-40; }
        else { r = sqrt(t); live = t > 0.0; }
        const double s3 = 1.7320508075688772;
        double e = exp(-s3 * r);
        k0 = (1.0 + s3 * r) * e;
        if (ORDER >= 1) k1 = live ? -1.5 * e : 0.0;
        if (ORDER >= 2) k2 = live ? 0.75 * s3 * e / r : 0.0;
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

// ---- internal launch helpers (defined across the .hip files) -----------------------------------
int oei_ws_reserve(oei_model* m, size_t rows, size_t aux_bytes);

// Ks[r, k] = k(x_r, X_k) for r < rows (x_r = row r of Xq [rows, d]); zero for r >= rows or k >= n.
// MODE 0: kernel value; MODE 1..d: derivative of k(x_r, X_k) with respect to x_r[MODE-1].
// K1 (optional): 2 var kappa'(t) of the same pairs, the factor the gradient needs; with it, `q`
// and `flags`: flags[r / q] is raised when x_r duplicates a training point (oei.py:165-167).
int oei_launch_gram(const oei_model* m, const double* Xq, size_t rows, size_t rows_pad,
                    double* Ks, double* K1, int q, int* flags, hipStream_t st);
int oei_ws2_reserve(oei_model* m, size_t rows);
// Looks up the current device's copy of the fixed scratch array of oei_value_grad_batched_noalloc.
int oei_noalloc_setup_device(int device);
// Extended rows of the Hessian path: per candidate RP rows (value rows, derivative rows, y, zeros).
int oei_launch_gram_ext(const oei_model* m, const double* Xq, int B, int q, int RP, double* out,
                        hipStream_t st);
int oei_launch_stage2(const oei_model* m, int B, int q, const double* X, const double* A,
                      size_t a_cand_stride, size_t a_row_stride, double* value, int* status,
                      double* p_buf, double* M11_buf, double* Z, hipStream_t st,
                      const double* start_in = nullptr, double* start_out = nullptr, int* iters = nullptr,
                      hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// G_b = E_b E_b^T (lower block-triangle of 64x64 tiles), E [B*RP, npad], G [B, RP, RP].
// `used`: rows of a block that hold data (the others are zero rows whose products nothing reads; 0 = all).
int oei_launch_gram_blk(const double* E, double* G, int B, int RP, int npad, hipStream_t st, int used = 0);
// C[r, j] = sum_k A[r, k] * Bm[j, k], all [*, npad] row-major; `lower` != 0 means Bm is lower
// triangular (k <= j), else upper triangular (k >= j).
// Fused W = Z L^-1 and gradient contraction; Gpart [npad / 64][nrows][d+1] partial sums.
#define OEI_GG_MAX_DP 12
int oei_launch_gemm_grad(const oei_model* m, const double* Z, const double* K1, const double* pvec,
                         double* Gpart, size_t nrows, size_t rows_pad, hipStream_t st);
int oei_launch_gemm_grad_zfree(const oei_model* m, const double* A, const double* K1, const double* pvec,
                               const double* M11, double* Gpart, size_t nrows, size_t rows_pad, int q,
                               hipStream_t st);
int oei_launch_gemm_nt(const double* A, const double* Bm, double* C, size_t rows_pad, int npad,
                       int lower, hipStream_t st);

// ---- multistart state machines: launches of one advance and addresses inside their workspaces
// (oei_lbfgs.hip, oei_newton.hip), for the round drivers of oei_multistart.hip ----------------------
int oei_lbfgs_advance_launch(int B, int n, int history, const double* ft, const double* gt,
                             const int* tstatus, const int* tflags, int* flagged,
                             double pgtol, double ftol, double c1, int max_ls, int max_iter,
                             double* dws, int* iws, hipStream_t st);
int* oei_lbfgs_count_ptr(int B, int n, int history, double* dws, int* iws);
double* oei_lbfgs_list_ptr(int B, int n, int history, double* dws, int* iws);
int oei_newton_advance_launch(int B, int n, const double* ft, const double* gt, const double* Ht,
                              const int* tstatus, const int* tflags, int* flagged,
                              double pgtol, double ftol, double c1, int max_ls, int max_iter,
                              double* dws, int* iws, hipStream_t st);
int* oei_newton_counts_ptr(int B, int n, double* dws, int* iws);
double* oei_newton_trial_list_ptr(int B, int n, double* dws, int* iws);
double* oei_newton_hess_list_ptr(int B, int n, double* dws, int* iws);
int oei_newton_check_shape(const char* who, int B, int n);
