// Negative log marginal likelihood of the GP and its gradient with respect to the hyper-parameters,
// from the factors a model handle already holds.  This is the quantity the reference minimises
// with gpflow's optimize() before every batch (bo.py:85, 194-221):
//
//   NLML = 1/2 y^T alpha + sum_i log L_ii + n/2 log 2 pi,        alpha = K^-1 y,  K = L L^T
//   dNLML/dtheta = 1/2 tr((K^-1 - alpha alpha^T) dK/dtheta)
//
// K^-1 = L^-T L^-1 is one triangular GEMM on the MFMA path; the traces are reductions over the
// n^2 pairs with the kernel derivative evaluated on the fly.
#include "oei_common.h"
#include "oei_factor.cuh"
#include <cstring>

// out[0] = 1/2 y.alpha + sum log L_ii + n/2 log(2 pi)
__global__ __launch_bounds__(256) void k_nlml_value(const double* __restrict__ L, const double* __restrict__ y,
                                                    const double* __restrict__ alpha, int n, int npad,
                                                    double* __restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += 0.5 * y[i] * alpha[i] + log(L[(size_t)i * npad + i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3] + 0.5 * n * 1.8378770664093453;
}

// One block per row i: partial[i][0] = 1/2 sum_j Q_ij k0_ij, partial[i][1 + l] = 1/2 sum_j Q_ij dK_ij/dls_l
// (l < nls; nls == 1: the shared lengthscale), partial[i][1 + nls] = 1/2 Q_ii; Q = K^-1 - alpha alpha^T.
__global__ __launch_bounds__(256) void k_nlml_grad(KernParams kp, const double* __restrict__ X,
                                                   const double* __restrict__ Kinv,
                                                   const double* __restrict__ alpha, int n, int npad,
                                                   int nls, double* __restrict__ partial) {
    __shared__ double red[4];
    const int i = blockIdx.x, d = kp.d, tid = threadIdx.x;
    const int nq = 2 + nls;
    for (int c = 0; c < nq; ++c) {
        double s = 0.0;
        for (int j = tid; j < n; j += 256) {
            const double Q = Kinv[(size_t)i * npad + j] - alpha[i] * alpha[j];
            double t = 0.0, dsel = 0.0;
            for (int l = 0; l < d; ++l) {
                const double df = X[(size_t)i * d + l] - X[(size_t)j * d + l];
                const double tl = df * df * kp.inv_ls2[l];
                t += tl;
                if (nls == 1 || l == c - 1) dsel += tl * kp.inv_ls[l];       // (df/ls)^2 / ls
            }
            double k0, k1, k2;
            kappa_eval<1>(kp.kind, kp.floor_mode, t, k0, k1, k2);
            double v;
            if (c == 0) v = k0;                                              // dK/dvariance
            else if (c <= nls) v = -2.0 * kp.variance * k1 * dsel;           // dK/dls: k1 dt/dls
            else v = (i == j) ? 1.0 : 0.0;                                   // dK/dnoise
            s += 0.5 * Q * v;
        }
        s = wave_sum(s);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) partial[(size_t)i * nq + c] = red[0] + red[1] + red[2] + red[3];
    }
}

extern "C" int oei_model_nlml(oei_model* m, double* value, double* grad, int n_lengthscales) {
    if (!m || !value) { oei_set_error("model_nlml: null pointer"); return OEI_ERR_ARG; }
    const int nls = n_lengthscales;
    if (nls != 1 && nls != m->d) { oei_set_error("model_nlml: need 1 or d lengthscale derivatives"); return OEI_ERR_ARG; }
    if (nls == 1) {
        for (int l = 1; l < m->d; ++l)
            if (m->kp.ls[l] != m->kp.ls[0]) { oei_set_error("model_nlml: shared-lengthscale gradient of an ARD model"); return OEI_ERR_ARG; }
    }
    const int n = m->n, npad = m->npad, nq = 2 + nls;
    hipStream_t st = 0;
    double* dval = m->dscal + 1;
    hipLaunchKernelGGL(k_nlml_value, dim3(1), dim3(256), 0, st, m->L, m->y, m->alpha, n, npad, dval);
    int rc = OEI_OK;
    if (grad) {
        // K^-1 and the per-row terms live in the handle (allocated once: an optimiser calls this
        // hundreds of times on one handle, see oei_model_refactor)
        if (!m->Kinv && hipMalloc(&m->Kinv, (size_t)npad * npad * sizeof(double)) != hipSuccess) { m->Kinv = nullptr; oei_set_error("model_nlml: out of device memory"); return OEI_ERR_HIP; }
        if (!m->nlml_part && hipMalloc(&m->nlml_part, (size_t)n * (2 + m->d) * sizeof(double)) != hipSuccess) { m->nlml_part = nullptr; oei_set_error("model_nlml: out of device memory"); return OEI_ERR_HIP; }
        rc = oei_launch_gemm_nt(m->LinvT, m->LinvT, m->Kinv, (size_t)npad, npad, 0, st);   // L^-T L^-1
        if (rc == OEI_OK)
            hipLaunchKernelGGL(k_nlml_grad, dim3(n), dim3(256), 0, st, m->kp, m->X, m->Kinv, m->alpha, n, npad, nls, m->nlml_part);
    }
    double* Kinv = m->Kinv;
    double* partial = m->nlml_part;
    if (rc == OEI_OK && hipGetLastError() != hipSuccess) { oei_set_error("model_nlml: launch failed"); rc = OEI_ERR_HIP; }
    if (rc == OEI_OK && hipMemcpy(value, dval, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { oei_set_error("model_nlml: copy failed"); rc = OEI_ERR_HIP; }
    if (rc == OEI_OK && grad) {
        double* host = new double[(size_t)n * nq];
        if (hipMemcpy(host, partial, (size_t)n * nq * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
            oei_set_error("model_nlml: copy failed"); rc = OEI_ERR_HIP;
        } else {
            for (int c = 0; c < nq; ++c) {
                double s = 0.0;
                for (int i = 0; i < n; ++i) s += host[(size_t)i * nq + c];
                grad[c] = s;
            }
        }
        delete[] host;
    }
    (void)Kinv;
    return rc;
}

// ================================================================================================
// Batched objective of the hyper-parameter fit: R trial parameter vectors on the same data in one
// pass.  bo.py:194-221 runs gpflow's optimize() from `model_restarts` random starts one after the
// other; the restarts are independent, so their objective evaluations are batched here the way
// the candidate batches of the acquisition are: batch item z owns one npad x npad set of factors.
// theta = log(variance, lengthscale(s), [noise]) as in hyperfit.py; the gradient is with respect
// to theta and includes the optional log-priors (run.py:95-97).
// ================================================================================================
struct oei_fit {
    int device, n, d, npad, kind, floor_mode, cap;
    double *X, *y;                 // [npad, d], [npad]
    KernParams* kps;               // [cap]
    double* noise;                 // [cap]
    double *L, *Linv, *LinvT, *Kinv;   // [cap, npad, npad]
    double *Dinv, *tmp;            // [cap, npad, 64]
    double *v, *alpha;             // [cap, npad]
    double* part;                  // [cap, n, 2 + d]
    double* val;                   // [cap]
    int* flag;                     // [cap]
};

// theta -> kernel parameters and noise of batch item z (one thread each)
__global__ void k_fit_params(const double* __restrict__ theta, int R, int nt, int nls, int d, int kind,
                             int floor_mode, double fixed_noise, KernParams* __restrict__ kps,
                             double* __restrict__ noise, int* __restrict__ flag) {
    const int z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= R) return;
    const double* th = theta + (size_t)z * nt;
    KernParams& P = kps[z];
    P.kind = kind; P.floor_mode = floor_mode; P.d = d;
    P.variance = exp(th[0]);
    for (int l = 0; l < d; ++l) {
        const double ls = exp(th[1 + (nls == 1 ? 0 : l)]);
        P.ls[l] = ls; P.inv_ls[l] = 1.0 / ls; P.inv_ls2[l] = 1.0 / (ls * ls);
    }
    noise[z] = fixed_noise >= 0.0 ? fixed_noise : exp(th[1 + nls]);
    flag[z] = 0;
}

__global__ void k_gram_train_batched(const KernParams* __restrict__ kps, const double* __restrict__ noise,
                                     const double* __restrict__ X, int n, int npad, double* __restrict__ K) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y, z = blockIdx.z;
    if (j >= npad) return;
    const KernParams* P = kps + z;
    const int d = P->d;
    double val;
    if (i < n && j < n) {
        double t = 0.0;
        for (int l = 0; l < d; ++l) {
            const double df = X[(size_t)i * d + l] - X[(size_t)j * d + l];
            t += df * df * P->inv_ls2[l];
        }
        double k0, k1, k2;
        kappa_eval<0>(P->kind, P->floor_mode, t, k0, k1, k2);
        val = P->variance * k0 + (i == j ? noise[z] : 0.0);
    } else {
        val = (i == j) ? 1.0 : 0.0;
    }
    K[(size_t)z * npad * npad + (size_t)i * npad + j] = val;
}

// K^-1 = L^-T L^-1, tile (ib, jb) of batch item z: sum over kb >= max(ib, jb) of
// LinvT[ib, kb] LinvT[jb, kb]^T (LinvT is upper triangular).
__global__ __launch_bounds__(256) void k_kinv_batched(const double* __restrict__ LinvT, int npad,
                                                      double* __restrict__ Kinv) {
    __shared__ double As[TB * TLD];
    __shared__ double Bs[TB * TLD];
    const int ib = blockIdx.y, jb = blockIdx.x, nb = npad / TB;
    LinvT += (size_t)blockIdx.z * npad * npad;
    Kinv += (size_t)blockIdx.z * npad * npad;
    double acc[4][4] = {};
    for (int kb = (ib > jb ? ib : jb); kb < nb; ++kb)
        tile_mm<true>(acc, LinvT + ((size_t)ib * TB) * npad + (size_t)kb * TB, npad,
                      LinvT + ((size_t)jb * TB) * npad + (size_t)kb * TB, npad, As, Bs);
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    for (int u = 0; u < 4; ++u)
        for (int w = 0; w < 4; ++w)
            Kinv[((size_t)ib * TB + 4 * ty + u) * npad + (size_t)jb * TB + 4 * tx + w] = acc[u][w];
}

__global__ __launch_bounds__(256) void k_nlml_value_batched(const double* __restrict__ L, const double* __restrict__ y,
                                                            const double* __restrict__ alpha, int n, int npad,
                                                            double* __restrict__ out) {
    __shared__ double red[4];
    const int z = blockIdx.x;
    L += (size_t)z * npad * npad;
    alpha += (size_t)z * npad;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += 0.5 * y[i] * alpha[i] + log(L[(size_t)i * npad + i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[z] = red[0] + red[1] + red[2] + red[3] + 0.5 * n * 1.8378770664093453;
}

// The per-row terms of k_nlml_grad for batch item z (same sums, parameters read from kps[z]).
__global__ __launch_bounds__(256) void k_nlml_grad_batched(const KernParams* __restrict__ kps, const double* __restrict__ X,
                                                           const double* __restrict__ Kinv,
                                                           const double* __restrict__ alpha, int n, int npad,
                                                           int nls, double* __restrict__ partial) {
    __shared__ double red[4];
    const int i = blockIdx.x, z = blockIdx.z, tid = threadIdx.x;
    const KernParams* P = kps + z;
    const int d = P->d, nq = 2 + nls;
    Kinv += (size_t)z * npad * npad;
    alpha += (size_t)z * npad;
    partial += (size_t)z * n * nq;
    const double var = P->variance;
    for (int c = 0; c < nq; ++c) {
        double s = 0.0;
        for (int j = tid; j < n; j += 256) {
            const double Q = Kinv[(size_t)i * npad + j] - alpha[i] * alpha[j];
            double t = 0.0, dsel = 0.0;
            for (int l = 0; l < d; ++l) {
                const double df = X[(size_t)i * d + l] - X[(size_t)j * d + l];
                const double tl = df * df * P->inv_ls2[l];
                t += tl;
                if (nls == 1 || l == c - 1) dsel += tl * P->inv_ls[l];
            }
            double k0, k1, k2;
            kappa_eval<1>(P->kind, P->floor_mode, t, k0, k1, k2);
            double v;
            if (c == 0) v = k0;
            else if (c <= nls) v = -2.0 * var * k1 * dsel;
            else v = (i == j) ? 1.0 : 0.0;
            s += 0.5 * Q * v;
        }
        s = wave_sum(s);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) partial[(size_t)i * nq + c] = red[0] + red[1] + red[2] + red[3];
    }
}

// One wave per batch item: column sums of the per-row terms, chain rule to theta = log(parameter),
// log-priors (Gamma(2, 0.5) on the lengthscales, N(1, 2) on the variance: run.py:95-97), and the
// value (+inf, zero gradient, when the covariance was not positive definite or the value is not finite).
__global__ __launch_bounds__(64) void k_nlml_finish(const double* __restrict__ theta, const double* __restrict__ partial,
                                                    const double* __restrict__ val, const int* __restrict__ flag,
                                                    int n, int nt, int nls, int priors, double fixed_noise,
                                                    double* __restrict__ value, double* __restrict__ grad) {
    const int z = blockIdx.x, lane = threadIdx.x, nq = 2 + nls;
    const double* th = theta + (size_t)z * nt;
    const double* part = partial + (size_t)z * n * nq;
    double v = val[z];
    const double variance = exp(th[0]);
    double prior = 0.0;
    if (priors) {
        prior = 0.5 * (variance - 1.0) * (variance - 1.0) / 2.0;
        for (int l = 0; l < nls; ++l) { const double ls = exp(th[1 + l]); prior -= log(ls) - ls / 0.5; }
    }
    v += prior;
    const bool bad = flag[z] != 0 || !isfinite(v);
    for (int c = 0; c < nt; ++c) {
        double s = 0.0;
        for (int i = lane; i < n; i += 64) s += part[(size_t)i * nq + c];
        s = wave_sum(s);
        const double par = exp(th[c]);
        double dprior = 0.0;
        if (priors) {
            if (c == 0) dprior = (par - 1.0) / 2.0;
            else if (c <= nls) dprior = -(1.0 / par - 1.0 / 0.5);
        }
        if (lane == 0) grad[(size_t)z * nt + c] = bad ? 0.0 : (s + dprior) * par;
    }
    if (lane == 0) value[z] = bad ? INFINITY : v;
}

static void fit_free(oei_fit* f) {
    if (!f) return;
    hipFree(f->X); hipFree(f->y); hipFree(f->kps); hipFree(f->noise); hipFree(f->L); hipFree(f->Linv);
    hipFree(f->LinvT); hipFree(f->Kinv); hipFree(f->Dinv); hipFree(f->tmp); hipFree(f->v); hipFree(f->alpha);
    hipFree(f->part); hipFree(f->val); hipFree(f->flag);
    delete f;
}

extern "C" int oei_fit_create(oei_fit** out, int n, int d, const double* X_train, const double* y,
                              int kernel_kind, int floor_mode, int max_batch) {
    if (!out || !X_train || !y) { oei_set_error("fit_create: null pointer"); return OEI_ERR_ARG; }
    if (n < 1 || d < 1 || d > OEI_MAX_DIM || max_batch < 1) { oei_set_error("fit_create: n=%d d=%d max_batch=%d out of range", n, d, max_batch); return OEI_ERR_ARG; }
    if (kernel_kind != OEI_KERNEL_RBF && kernel_kind != OEI_KERNEL_MATERN32) { oei_set_error("fit_create: unknown kernel %d", kernel_kind); return OEI_ERR_ARG; }
    if (floor_mode < 0 || floor_mode > 2) { oei_set_error("fit_create: unknown floor mode %d", floor_mode); return OEI_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { oei_set_error("no HIP device"); return OEI_ERR_NO_DEVICE; }
    oei_fit* f = new oei_fit();
    memset(f, 0, sizeof(*f));
    f->n = n; f->d = d; f->npad = ((n + OEI_NPAD - 1) / OEI_NPAD) * OEI_NPAD;
    f->kind = kernel_kind; f->floor_mode = floor_mode; f->cap = max_batch;
    const size_t npad = f->npad, R = max_batch, nn = npad * npad * sizeof(double);
    auto body = [&]() -> int {
        OEI_HIP_CHECK(hipGetDevice(&f->device));
        OEI_HIP_CHECK(hipMalloc(&f->X, npad * d * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&f->y, npad * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&f->kps, R * sizeof(KernParams)));
        OEI_HIP_CHECK(hipMalloc(&f->noise, R * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&f->L, R * nn));
        OEI_HIP_CHECK(hipMalloc(&f->Linv, R * nn));
        OEI_HIP_CHECK(hipMalloc(&f->LinvT, R * nn));
        OEI_HIP_CHECK(hipMalloc(&f->Kinv, R * nn));
        OEI_HIP_CHECK(hipMalloc(&f->Dinv, R * npad * TB * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&f->tmp, R * npad * TB * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&f->v, R * npad * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&f->alpha, R * npad * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&f->part, R * n * (2 + d) * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&f->val, R * sizeof(double)));
        OEI_HIP_CHECK(hipMalloc(&f->flag, R * sizeof(int)));
#ifdef OEI_POISON_LDS      // test variant: see oei_model_create
        hipMemset(f->L, 0xFF, R * nn); hipMemset(f->Linv, 0xFF, R * nn); hipMemset(f->LinvT, 0xFF, R * nn);
        hipMemset(f->Kinv, 0xFF, R * nn); hipMemset(f->Dinv, 0xFF, R * npad * TB * sizeof(double));
        hipMemset(f->tmp, 0xFF, R * npad * TB * sizeof(double)); hipMemset(f->v, 0xFF, R * npad * sizeof(double));
        hipMemset(f->alpha, 0xFF, R * npad * sizeof(double)); hipMemset(f->part, 0xFF, R * n * (2 + d) * sizeof(double));
#endif
        OEI_HIP_CHECK(hipMemset(f->X, 0, npad * d * sizeof(double)));
        OEI_HIP_CHECK(hipMemset(f->y, 0, npad * sizeof(double)));
        OEI_HIP_CHECK(hipMemcpy(f->X, X_train, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice));
        OEI_HIP_CHECK(hipMemcpy(f->y, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        return OEI_OK;
    };
    const int rc = body();
    if (rc != OEI_OK) { fit_free(f); return rc; }
    *out = f;
    return OEI_OK;
}

extern "C" int oei_fit_destroy(oei_fit* f) {
    if (!f) return OEI_OK;
    int cur = -1;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    if (have && cur != f->device) hipSetDevice(f->device);
    fit_free(f);
    if (have && cur != f->device) hipSetDevice(cur);
    return OEI_OK;
}

extern "C" int oei_fit_nlml_batched(oei_fit* f, int R, const double* theta, int n_lengthscales,
                                    double fixed_noise, int priors, double* value, double* grad,
                                    void* stream) {
    if (!f || !theta || !value || !grad) { oei_set_error("fit_nlml_batched: null pointer"); return OEI_ERR_ARG; }
    const int nls = n_lengthscales;
    if (nls != 1 && nls != f->d) { oei_set_error("fit_nlml_batched: need 1 or d lengthscales"); return OEI_ERR_ARG; }
    if (R < 1 || R > f->cap) { oei_set_error("fit_nlml_batched: batch %d outside 1..%d", R, f->cap); return OEI_ERR_ARG; }
    const int n = f->n, npad = f->npad, nb = npad / TB, d = f->d;
    const int nt = 1 + nls + (fixed_noise >= 0.0 ? 0 : 1);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fit_params, dim3((R + 63) / 64), dim3(64), 0, st, theta, R, nt, nls, d, f->kind, f->floor_mode,
                       fixed_noise, f->kps, f->noise, f->flag);
    hipLaunchKernelGGL(k_gram_train_batched, dim3((npad + 255) / 256, npad, R), dim3(256), 0, st, f->kps, f->noise, f->X, n, npad, f->L);
    for (int kb = 0; kb < nb; ++kb) {
        hipLaunchKernelGGL(k_potf2, dim3(1, 1, R), dim3(64), 0, st, f->L, npad, kb, f->Dinv, f->flag);
        const int rem = nb - kb - 1;
        if (rem > 0) {
            hipLaunchKernelGGL(k_trsm_panel, dim3(rem, 1, R), dim3(256), 0, st, f->L, npad, kb, f->Dinv);
            hipLaunchKernelGGL(k_syrk_trail, dim3(rem, rem, R), dim3(256), 0, st, f->L, npad, kb);
        }
    }
    hipLaunchKernelGGL(k_zero_upper, dim3((npad + 255) / 256, npad, R), dim3(256), 0, st, f->L, npad);
    hipLaunchKernelGGL(k_trinv, dim3(nb, 1, R), dim3(256), 0, st, f->L, npad, f->Dinv, f->Linv, f->tmp);
    hipLaunchKernelGGL(k_transpose, dim3(npad / 32, npad / 32, R), dim3(32, 32), 0, st, f->Linv, f->LinvT, npad);
    hipLaunchKernelGGL(k_matvec, dim3(npad, 1, R), dim3(64), 0, st, f->Linv, f->y, f->v, npad, 0);
    hipLaunchKernelGGL(k_matvec, dim3(npad, 1, R), dim3(64), 0, st, f->LinvT, f->v, f->alpha, npad, npad);
    hipLaunchKernelGGL(k_nlml_value_batched, dim3(R), dim3(256), 0, st, f->L, f->y, f->alpha, n, npad, f->val);
    hipLaunchKernelGGL(k_kinv_batched, dim3(nb, nb, R), dim3(256), 0, st, f->LinvT, npad, f->Kinv);
    hipLaunchKernelGGL(k_nlml_grad_batched, dim3(n, 1, R), dim3(256), 0, st, f->kps, f->X, f->Kinv, f->alpha, n, npad, nls, f->part);
    hipLaunchKernelGGL(k_nlml_finish, dim3(R), dim3(64), 0, st, theta, f->part, f->val, f->flag, n, nt, nls, priors,
                       fixed_noise, value, grad);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}
