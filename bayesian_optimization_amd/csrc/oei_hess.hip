// Hessian path: stand-alone SDP sensitivity (sdp.solution_derivative), dOmega/dx (OEI.domega) and
// the acquisition Hessian (OEI.acquisition_hessian).
//
// Hessian pipeline for B candidates (reference: oei.py:41-64, 105-135):
//   k_gram_ext   rows per candidate: k(x_a,.), d k(x_a,.)/d x_al (a < q, l < d), then y
//   GEMM         E = rows L^-T            (so the y row becomes v = L^-1 y)
//   k_gram_blk   G_b = E_b E_b^T          every n-length reduction of the path, on MFMA:
//                A A^T (-> Sigma), A v (-> mu), T A^T (-> dSigma/dx), T v (-> dmu/dx), T T^T
//   k_hess_sdp   mu, Sigma from G -> SDP -> p, M11, Z = M11 A
//   GEMM         W = Z L^-1
//   k_hess       state at p, one sensitivity solve per x-direction fused with the contraction
//                <dM_(a,l), dOmega_(b,m)>, plus d2<Omega,M>/dx2 at fixed M in closed form
#include "oei_sens.cuh"

typedef double double4_t __attribute__((ext_vector_type(4)));

// ================================================================================================
// stand-alone sensitivity: sdp.solution_derivative (sdp.py:159-206)
// ================================================================================================
template <int T>
__global__ __launch_bounds__(T) void k_soldiff(int q, int K, const double* __restrict__ Yin,
                                               const double* __restrict__ fmin,
                                               const double* __restrict__ domega,
                                               double* __restrict__ dM, double* __restrict__ dY,
                                               int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    SdpWork w;
    sdp_carve(w, sm, q);
    const int VL = sdp_vlen(q), LD = w.LD, k_ = q + 1;
    double* mu = sm + sdp_work_doubles(q);
    SensVec sv;
    sens_carve(sv, mu + VL, q);
    double* Sc = mu + VL + SENS_NVEC * VL;            // copy of S (+ mu mu^T) for dw0 != 0
    const int tid = threadIdx.x, b = blockIdx.x;
    const double* Yb = Yin + (size_t)b * k_ * k_ * k_;
    // Omega = sum_i Y_i
    double w0 = 0.0;
    for (int blk = 0; blk < k_; ++blk) w0 += Yb[(size_t)blk * k_ * k_ + q * k_ + q];
    if (tid < q) {
        double acc = 0.0;
        for (int blk = 0; blk < k_; ++blk)
            acc += 0.5 * (Yb[(size_t)blk * k_ * k_ + tid * k_ + q] + Yb[(size_t)blk * k_ * k_ + q * k_ + tid]);
        mu[tid] = acc / w0;
        w.dl[tid] = mu[tid] - fmin[b];
        w.p[tid] = Yb[(size_t)(tid + 1) * k_ * k_ + q * k_ + q] / w0;
    }
    __syncthreads();
    for (int e = tid; e < q * q; e += T) {
        const int i = e / q, j = e - i * q;
        double acc = 0.0;
        for (int blk = 0; blk < k_; ++blk)
            acc += 0.5 * (Yb[(size_t)blk * k_ * k_ + i * k_ + j] + Yb[(size_t)blk * k_ * k_ + j * k_ + i]);
        Sc[i * LD + j] = acc / w0;                     // Omega11 / w0
        w.S[i * LD + j] = acc / w0 - mu[i] * mu[j];
    }
    __syncthreads();
    int sweeps = 0;
    const double g = sdp_eval<T>(w, w.p, sweeps);
    int st = (g > -INFINITY) ? 0 : 2;
    if (!st && sens_prepare<T>(w, mu, sv)) st = 2;
    if (st) {
        for (size_t e = tid; e < (size_t)k_ * k_ * K; e += T) dM[(size_t)b * k_ * k_ * K + e] = NAN;
        if (dY) for (int e = tid; e < k_ * k_ * k_; e += T) dY[(size_t)b * k_ * k_ * k_ + e] = NAN;
        if (tid == 0 && status) status[b] = 2;
        return;
    }
    const double* dOb = domega + (size_t)b * k_ * k_ * K;
    double* dMb = dM + (size_t)b * k_ * k_ * K;
    for (int j = 0; j < K; ++j) {
        const double dw0 = dOb[((size_t)q * k_ + q) * K + j];
        __syncthreads();
        if (tid < q)
            sv.dmu[tid] = (0.5 * (dOb[((size_t)tid * k_ + q) * K + j] + dOb[((size_t)q * k_ + tid) * K + j]) - mu[tid] * dw0) / w0;
        __syncthreads();
        for (int e = tid; e < q * q; e += T) {
            const int i = e / q, l = e - i * q;
            const double dO = 0.5 * (dOb[((size_t)i * k_ + l) * K + j] + dOb[((size_t)l * k_ + i) * K + j]);
            w.X[i * LD + l] = (dO - Sc[i * LD + l] * dw0) / w0 - sv.dmu[i] * mu[l] - mu[i] * sv.dmu[l];
        }
        const double dnu = sens_direction<T>(w, sv);
        for (int e = tid; e < k_ * k_; e += T) {
            const int i = e / k_, l = e - i * k_;
            double val;
            if (i < q && l < q) val = -w.X[i * LD + l];
            else if (i < q) val = sv.w1[i] + sv.w2[i];
            else if (l < q) val = sv.w1[l] + sv.w2[l];
            else val = -dnu;
            dMb[(size_t)e * K + j] = val;
        }
        if (dY && K == 1) {
            // y_blk = sqrt(w0 p_blk) [x_blk; 1], x_0 = c, x_i = c - (B/2) e_i;  dY = y dy^T + dy y^T
            __syncthreads();
            double* Bh = w.V;       // B/2  = Rs R^T
            double* dBh = Sc;       // dB/2 = -2 Rs Z Rs^T
            for (int e = tid; e < q * q; e += T) {
                const int i = e / q, l = e - i * q;
                double bh = 0.0, db = 0.0;
                for (int a = 0; a < q; ++a) {
                    const double ria = w.R[i * LD + a] * w.sq[a];
                    bh += ria * w.R[l * LD + a];
                    double inner = 0.0;
                    for (int c2 = 0; c2 < q; ++c2) inner += w.Kt[a * LD + c2] * w.R[l * LD + c2] * w.sq[c2];
                    db += ria * inner;
                }
                Bh[i * LD + l] = bh;
                dBh[i * LD + l] = -2.0 * db;
            }
            __syncthreads();
            double psum = 0.0, dpsum = 0.0;
            for (int i = 0; i < q; ++i) { psum += w.p[i]; dpsum += sv.dp[i]; }
            double* dYb = dY + (size_t)b * k_ * k_ * k_;
            for (int e = tid; e < k_ * k_ * k_; e += T) {
                const int blk = e / (k_ * k_), rem = e - blk * k_ * k_;
                const int i = rem / k_, l = rem - i * k_;
                const double pb = blk ? w.p[blk - 1] : 1.0 - psum;
                const double dpb = blk ? sv.dp[blk - 1] : -dpsum;
                const double amp = sqrt(w0 * pb);
                const double damp = 0.5 * (dw0 * pb + w0 * dpb) / amp;
                const double xi = (i < q) ? sv.c[i] - (blk ? Bh[i * LD + blk - 1] : 0.0) : 1.0;
                const double xl = (l < q) ? sv.c[l] - (blk ? Bh[l * LD + blk - 1] : 0.0) : 1.0;
                const double dxi = (i < q) ? sv.dc[i] - (blk ? dBh[i * LD + blk - 1] : 0.0) : 0.0;
                const double dxl = (l < q) ? sv.dc[l] - (blk ? dBh[l * LD + blk - 1] : 0.0) : 0.0;
                const double yi = amp * xi, yl = amp * xl;
                const double dyi = damp * xi + amp * dxi, dyl = damp * xl + amp * dxl;
                dYb[e] = yi * dyl + dyi * yl;
            }
        }
    }
    if (tid == 0 && status) status[b] = 0;
}

template <typename Kf>
static int set_lds_attr(Kf kernel, size_t lds) {
    if (lds > 160 * 1024) { oei_set_error("LDS request %zu exceeds 160 KiB (q too large for this entry point)", lds); return OEI_ERR_ARG; }
    if (lds > 48 * 1024)
        OEI_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return OEI_OK;
}

extern "C" int oei_solution_derivative_batched(int B, int k_, int K, const double* M,
                                               const double* Y, const double* fmin,
                                               const double* domega, double* dM, double* dY,
                                               int* status, void* stream) {
    (void)M;   // the primal solution is implied by Y; kept in the signature to mirror sdp.py:159
    if (B < 0 || k_ < 2 || k_ - 1 > OEI_MAX_Q || K < 1) { oei_set_error("solution_derivative_batched: B=%d k_=%d K=%d out of range", B, k_, K); return OEI_ERR_ARG; }
    if (B == 0) return OEI_OK;
    if (!Y || !fmin || !domega || !dM) { oei_set_error("solution_derivative_batched: null pointer"); return OEI_ERR_ARG; }
    if (dY && K != 1) { oei_set_error("solution_derivative_batched: dY needs K == 1 (sdp.py:196)"); return OEI_ERR_ARG; }
    const int q = k_ - 1;
    const size_t lds = (sdp_work_doubles(q) + (size_t)(1 + SENS_NVEC) * sdp_vlen(q) + (size_t)q * sdp_ld(q)) * sizeof(double);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (q <= 32) {
        if ((rc = set_lds_attr(k_soldiff<64>, lds))) return rc;
        hipLaunchKernelGGL(k_soldiff<64>, dim3(B), dim3(64), lds, st, q, K, Y, fmin, domega, dM, dY, status);
    } else {
        if ((rc = set_lds_attr(k_soldiff<256>, lds))) return rc;
        hipLaunchKernelGGL(k_soldiff<256>, dim3(B), dim3(256), lds, st, q, K, Y, fmin, domega, dM, dY, status);
    }
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

extern "C" int oei_hessian_batched(oei_model* m, int B, int q, const double* X, double* H,
                                   int* status, void* stream) {
    oei_set_error("hessian_batched: not built yet");
    return OEI_ERR_ARG;
}

extern "C" int oei_domega_batched(oei_model* m, int B, int q, const double* X, double* domega,
                                  void* stream) {
    oei_set_error("domega_batched: not built yet");
    return OEI_ERR_ARG;
}
