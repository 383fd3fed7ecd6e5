// GP model handle: one-time set-up of everything that is shared by all candidate batches.
//
//   K = k(X,X) + noise I  ->  blocked right-looking Cholesky (64x64 blocks)  ->  L
//   L^-1 by blocked triangular inversion, L^-T, v = L^-1 y, alpha = L^-T v,
//   fmin = min_i (k(X_i, X) alpha)            (posterior mean at the training inputs, oei.py:27)
//
// gpflow recomputes the Cholesky factor inside every Session.run of the reference (three times per
// acquisition call); here it is computed once per model and kept in HBM (8 MB at n = 1000).
#include "oei_common.h"
#include <cstdarg>
#include <cstring>
#include <vector>

static thread_local char g_err[512] = "no error";

void oei_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* oei_last_error(void) { return g_err; }
extern "C" int oei_abi_version(void) { return OEI_ABI_VERSION; }

#include "oei_factor.cuh"

// mean_i = sum_j k(X_i, X_j) alpha_j for i < n (one workgroup per training point), then the minimum
__global__ __launch_bounds__(256) void k_train_mean(KernParams kp, const double* __restrict__ X, int n,
                                                    const double* __restrict__ alpha,
                                                    double* __restrict__ mean) {
    __shared__ double red[4];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        double t = 0.0;
        for (int l = 0; l < kp.d; ++l) {
            double df = X[(size_t)i * kp.d + l] - X[(size_t)j * kp.d + l];
            t += df * df * kp.inv_ls2[l];
        }
        double k0, k1, k2;
        kappa_eval<0>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        s += kp.variance * k0 * alpha[j];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) mean[i] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_fmin(const double* __restrict__ mean, int n,
                                              double* __restrict__ fmin_out) {
    __shared__ double red[4];
    double best = INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) best = fmin(best, mean[i]);
    best = wave_min(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) *fmin_out = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
}

// ------------------------------------------------------------------------------------------------
static void model_free(oei_model* m) {
    if (!m) return;
    hipFree(m->X); hipFree(m->L); hipFree(m->Linv); hipFree(m->LinvT);
    hipFree(m->v); hipFree(m->alpha); hipFree(m->y);
    hipFree(m->ws0); hipFree(m->ws1); hipFree(m->ws2); hipFree(m->aux); hipFree(m->hs);
    hipFree(m->Dinv); hipFree(m->ftmp); hipFree(m->dscal); hipFree(m->dflag); hipFree(m->Kinv); hipFree(m->nlml_part);
    if (m->ev[0]) for (int i = 0; i < 6; ++i) hipEventDestroy(m->ev[i]);
    for (int i = 0; i < 2; ++i) if (m->kev[i]) hipEventDestroy(m->kev[i]);
    for (int i = 0; i < 2; ++i) {
        if (m->side[i]) hipStreamDestroy(m->side[i]);
        if (m->ev_join[i]) hipEventDestroy(m->ev_join[i]);
    }
    if (m->ev_fork) hipEventDestroy(m->ev_fork);
    delete m;
}

static int set_hyper(oei_model* m, int kernel_kind, int floor_mode, double variance,
                     const double* lengthscales, int n_lengthscales, double noise, const char* who) {
    const int d = m->d;
    if (kernel_kind != OEI_KERNEL_RBF && kernel_kind != OEI_KERNEL_MATERN32) { oei_set_error("%s: unknown kernel %d", who, kernel_kind); return OEI_ERR_ARG; }
    if (floor_mode < 0 || floor_mode > 2) { oei_set_error("%s: unknown floor mode %d", who, floor_mode); return OEI_ERR_ARG; }
    if (!lengthscales || (n_lengthscales != 1 && n_lengthscales != d)) { oei_set_error("%s: need 1 or d lengthscales", who); return OEI_ERR_ARG; }
    if (!(variance > 0.0) || !(noise >= 0.0)) { oei_set_error("%s: variance/noise out of range", who); return OEI_ERR_ARG; }
    for (int l = 0; l < d; ++l)
        if (!(lengthscales[n_lengthscales == 1 ? 0 : l] > 0.0)) { oei_set_error("%s: lengthscale <= 0", who); return OEI_ERR_ARG; }
    m->noise = noise;
    m->kp.kind = kernel_kind; m->kp.floor_mode = floor_mode; m->kp.d = d; m->kp.variance = variance;
    for (int l = 0; l < d; ++l) {
        const double ls = lengthscales[n_lengthscales == 1 ? 0 : l];
        m->kp.inv_ls[l] = 1.0 / ls;
        m->kp.ls[l] = ls;
        m->kp.inv_ls2[l] = 1.0 / (ls * ls);
    }
    return OEI_OK;
}

// K + noise I -> L, L^-1, L^-T, v, alpha, fmin from the data and hyper-parameters in the handle.
// Allocates nothing; synchronous.
static int factorise(oei_model* m) {
    const int n = m->n, npad = m->npad;
    hipStream_t st = 0;
    OEI_HIP_CHECK(hipMemsetAsync(m->dflag, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_gram_train, dim3((npad + 255) / 256, npad), dim3(256), 0, st, m->kp, m->X, n, npad, m->noise, m->L);
    const int nb = npad / TB;
    for (int kb = 0; kb < nb; ++kb) {
        hipLaunchKernelGGL(k_potf2, dim3(1), dim3(64), 0, st, m->L, npad, kb, m->Dinv, m->dflag);
        const int rem = nb - kb - 1;
        if (rem > 0) {
            hipLaunchKernelGGL(k_trsm_panel, dim3(rem), dim3(256), 0, st, m->L, npad, kb, m->Dinv);
            hipLaunchKernelGGL(k_syrk_trail, dim3(rem, rem), dim3(256), 0, st, m->L, npad, kb);
        }
    }
    hipLaunchKernelGGL(k_zero_upper, dim3((npad + 255) / 256, npad), dim3(256), 0, st, m->L, npad);
    hipLaunchKernelGGL(k_trinv, dim3(nb), dim3(256), 0, st, m->L, npad, m->Dinv, m->Linv, m->ftmp);
    hipLaunchKernelGGL(k_transpose, dim3(npad / 32, npad / 32), dim3(32, 32), 0, st, m->Linv, m->LinvT, npad);
    hipLaunchKernelGGL(k_matvec, dim3(npad), dim3(64), 0, st, m->Linv, m->y, m->v, npad, 0);
    hipLaunchKernelGGL(k_matvec, dim3(npad), dim3(64), 0, st, m->LinvT, m->v, m->alpha, npad, 0);
    // ftmp is free again after the triangular inverse: the n posterior means go there
    hipLaunchKernelGGL(k_train_mean, dim3(n), dim3(256), 0, st, m->kp, m->X, n, m->alpha, m->ftmp);
    hipLaunchKernelGGL(k_fmin, dim3(1), dim3(256), 0, st, m->ftmp, n, m->dscal);
    OEI_HIP_CHECK(hipGetLastError());
    int flag = 0;
    OEI_HIP_CHECK(hipMemcpy(&flag, m->dflag, sizeof(int), hipMemcpyDeviceToHost));      // waits for the stream
    OEI_HIP_CHECK(hipMemcpy(&m->fmin, m->dscal, sizeof(double), hipMemcpyDeviceToHost));
    if (flag) { oei_set_error("K + noise*I is not positive definite"); return OEI_ERR_NOT_PD; }
    return OEI_OK;
}

extern "C" int oei_model_create(oei_model** out, int n, int d, const double* X_train,
                                const double* y, int kernel_kind, int floor_mode, double variance,
                                const double* lengthscales, int n_lengthscales, double noise) {
    if (!out || !X_train || !y || !lengthscales) { oei_set_error("model_create: null pointer"); return OEI_ERR_ARG; }
    if (n < 1 || d < 1 || d > OEI_MAX_DIM) { oei_set_error("model_create: n=%d d=%d out of range (d <= %d)", n, d, OEI_MAX_DIM); return OEI_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        // argument errors are reported even without a device
        oei_model probe;
        memset(&probe, 0, sizeof(probe));
        probe.d = d;
        const int prc = set_hyper(&probe, kernel_kind, floor_mode, variance, lengthscales, n_lengthscales, noise, "model_create");
        if (prc) return prc;
        oei_set_error("no HIP device");
        return OEI_ERR_NO_DEVICE;
    }
    oei_model* m = new oei_model();
    memset(m, 0, sizeof(*m));
    m->n = n; m->d = d;
    m->npad = ((n + OEI_NPAD - 1) / OEI_NPAD) * OEI_NPAD;
    int rc = set_hyper(m, kernel_kind, floor_mode, variance, lengthscales, n_lengthscales, noise, "model_create");
    if (rc) { delete m; return rc; }
    if (hipGetDevice(&m->device) != hipSuccess) { delete m; oei_set_error("model_create: hipGetDevice failed"); return OEI_ERR_HIP; }
    m->pipeline = 1;
    m->zfree = 0;
    if (hipDeviceGetAttribute(&m->n_cu, hipDeviceAttributeMultiprocessorCount, m->device) != hipSuccess) m->n_cu = 0;
    if (m->device < 64) {       // set-up of the allocation-free entry point for this device
        const int src = oei_noalloc_setup_device(m->device);
        if (src != OEI_OK) { delete m; return src; }
    }
    const int npad = m->npad;
    const size_t nn = (size_t)npad * npad * sizeof(double);
    auto body = [&]() -> int {
        OEI_HIP_CHECK(hipMalloc(&m->X, (size_t)npad * d * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&m->L, nn));
        OEI_HIP_CHECK(hipMalloc(&m->Linv, nn));
        OEI_HIP_CHECK(hipMalloc(&m->LinvT, nn));
        OEI_HIP_CHECK(hipMalloc(&m->v, npad * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&m->alpha, npad * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&m->y, npad * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&m->Dinv, (size_t)npad * TB * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&m->ftmp, (size_t)npad * TB * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&m->dscal, 2 * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&m->dflag, sizeof(int)));
#ifdef OEI_POISON_LDS      // test variant: what the factorisation does not write is NaN, not the zeros of a fresh allocation
        hipMemset(m->L, 0xFF, nn); hipMemset(m->Linv, 0xFF, nn); hipMemset(m->LinvT, 0xFF, nn);
        hipMemset(m->v, 0xFF, npad * sizeof(double)); hipMemset(m->alpha, 0xFF, npad * sizeof(double));
        hipMemset(m->Dinv, 0xFF, (size_t)npad * TB * sizeof(double)); hipMemset(m->ftmp, 0xFF, (size_t)npad * TB * sizeof(double));
#endif
        OEI_HIP_CHECK(hipMemset(m->X, 0, (size_t)npad * d * sizeof(double)));
        OEI_HIP_CHECK(hipMemset(m->y, 0, npad * sizeof(double)));
        OEI_HIP_CHECK(hipMemcpy(m->X, X_train, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice));
        OEI_HIP_CHECK(hipMemcpy(m->y, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        return factorise(m);
    };
    rc = body();
    if (rc != OEI_OK) { model_free(m); return rc; }
    *out = m;
    return OEI_OK;
}

extern "C" int oei_model_refactor(oei_model* m, int kernel_kind, int floor_mode, double variance,
                                  const double* lengthscales, int n_lengthscales, double noise) {
    if (!m) { oei_set_error("model_refactor: null model"); return OEI_ERR_ARG; }
    int rc = set_hyper(m, kernel_kind, floor_mode, variance, lengthscales, n_lengthscales, noise, "model_refactor");
    if (rc) return rc;
    int cur = -1;
    OEI_HIP_CHECK(hipGetDevice(&cur));
    if (cur != m->device) OEI_HIP_CHECK(hipSetDevice(m->device));
    // Evaluations of this handle may still be in flight on the caller's streams or on the handle's
    // side streams (non-blocking streams are not ordered against the null stream the
    // factorisation runs on); they read the factors that are rewritten below.
    const hipError_t se = hipDeviceSynchronize();
    if (se != hipSuccess) {
        if (cur != m->device) hipSetDevice(cur);
        oei_set_error("model_refactor: hipDeviceSynchronize failed: %s", hipGetErrorString(se));
        return OEI_ERR_HIP;
    }
    rc = factorise(m);
    if (cur != m->device) hipSetDevice(cur);
    return rc;
}

extern "C" int oei_model_destroy(oei_model* m) {
    if (!m) return OEI_OK;
    int cur = -1;
    const int dev = m->device;                            // m is gone after model_free
    const bool have = hipGetDevice(&cur) == hipSuccess;
    if (!have || cur != dev) hipSetDevice(dev);
    model_free(m);
    if (have && cur != dev) hipSetDevice(cur);            // the caller's device stays current
    return OEI_OK;
}

extern "C" int oei_model_info(const oei_model* m, double* fmin, int* n, int* d) {
    if (!m) { oei_set_error("model_info: null model"); return OEI_ERR_ARG; }
    if (fmin) *fmin = m->fmin;
    if (n) *n = m->n;
    if (d) *d = m->d;
    return OEI_OK;
}

extern "C" int oei_model_get_factor(const oei_model* m, double* L_host) {
    if (!m || !L_host) { oei_set_error("model_get_factor: null pointer"); return OEI_ERR_ARG; }
    OEI_HIP_CHECK(hipMemcpy2D(L_host, (size_t)m->n * sizeof(double), m->L, (size_t)m->npad * sizeof(double),
                              (size_t)m->n * sizeof(double), m->n, hipMemcpyDeviceToHost));
    return OEI_OK;
}

int oei_ws2_reserve(oei_model* m, size_t rows) {
    if (rows > m->ws2_rows) {
        if (m->ws2) OEI_HIP_CHECK(hipFree(m->ws2));
        m->ws2 = nullptr; m->ws2_rows = 0;
        OEI_HIP_CHECK(hipMalloc(&m->ws2, rows * (size_t)m->npad * sizeof(double)));
        m->ws2_rows = rows;
    }
    return OEI_OK;
}

int oei_ws_reserve(oei_model* m, size_t rows, size_t aux_bytes) {
    if (rows > m->ws_rows) {
        if (m->ws0) OEI_HIP_CHECK(hipFree(m->ws0));
        if (m->ws1) OEI_HIP_CHECK(hipFree(m->ws1));
        m->ws0 = m->ws1 = nullptr; m->ws_rows = 0;
        OEI_HIP_CHECK(hipMalloc(&m->ws0, rows * (size_t)m->npad * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&m->ws1, rows * (size_t)m->npad * sizeof(double)));
        m->ws_rows = rows;
    }
    if (aux_bytes > m->aux_bytes) {
        if (m->aux) OEI_HIP_CHECK(hipFree(m->aux));
        m->aux = nullptr; m->aux_bytes = 0;
        OEI_HIP_CHECK(hipMalloc(&m->aux, aux_bytes));
        m->aux_bytes = aux_bytes;
    }
    return OEI_OK;
}
