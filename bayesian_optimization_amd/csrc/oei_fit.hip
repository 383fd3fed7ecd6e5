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
