// Stand-alone batched SDP entry point: Omega[B,k_,k_], fmin[B] -> value, M, Y, status, iters.
// Replaces sdp.sdp of the reference (sdp.py:11-67) for B independent problems at once.
#include "oei_sdp.cuh"

template <int T, int Q>
__global__ __launch_bounds__(T, SDP_MIN_WAVES(Q)) void k_sdp_omega(int q, const double* __restrict__ omega,
                                                 const double* __restrict__ fmin,
                                                 double* __restrict__ value, double* __restrict__ M,
                                                 double* __restrict__ Y, int* __restrict__ status,
                                                 int* __restrict__ iters) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    SdpWork w;
    sdp_carve(w, sm, q, false, sdp_mm<Q>());
    lds_f64* mu = (lds_f64*)sm + sdp_work_doubles(q, false, sdp_mm<Q>());  // [q]
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int k_ = q + 1, LD = w.LD;
    const double* om = omega + (size_t)b * k_ * k_;
    const double w0 = om[q * k_ + q];
    const double fm = fmin[b];
    int bad = !(w0 > 0.0);
    if (tid < q) {
        mu[tid] = 0.5 * (om[tid * k_ + q] + om[q * k_ + tid]) / w0;      // sdp.py:34 symmetrises
        w.dl[tid] = mu[tid] - fm;
    }
    __syncthreads();
    for (int e = tid; e < q * q; e += T) {
        const int i = e / q, j = e - i * q;
        const double v = 0.5 * (om[i * k_ + j] + om[j * k_ + i]) / w0 - mu[i] * mu[j];
        if (!(v == v)) bad = 1;
        w.S[i * LD + j] = v;
    }
    bad = __syncthreads_or(bad);
    double g = 0.0;
    int it = 0, st = 2;
    if (!bad) st = sdp_solve<T, Q>(w, &g, &it);
    double* Mo = M + (size_t)b * k_ * k_;
    if (st == 2) {        // no usable state: report NaN like a failed solve, never stale memory
        for (int e = tid; e < k_ * k_; e += T) Mo[e] = NAN;
        if (Y) for (int e = tid; e < k_ * k_ * k_; e += T) Y[(size_t)b * k_ * k_ * k_ + e] = NAN;
        if (tid == 0) { value[b] = NAN; status[b] = 2; if (iters) iters[b] = it; }
        return;
    }
    const double nu = sdp_primal<T>(w, mu);
    for (int e = tid; e < k_ * k_; e += T) {
        const int i = e / k_, j = e - i * k_;
        double v;
        if (i < q && j < q) v = -w.H[i * LD + j];
        else if (i < q) v = w.tv[i];
        else if (j < q) v = w.tv[j];
        else v = -nu;
        Mo[e] = v;
    }
    if (Y) {
        // y_0 = sqrt(w0 p0) [c;1];  y_i = sqrt(w0 p_i) [c - B e_i/2; 1],  B/2 = R sq R^T into w.S
        // (the factor of S is no longer needed)
        __syncthreads();
        lds_f64* Bh = w.S;
        for (int e = tid; e < q * q; e += T) {
            const int i = e / q, j = e - i * q;
            double acc = 0.0;
            for (int a = 0; a < q; ++a) acc += sdp_R(w, i, a) * sdp_R(w, j, a) * w.sq[a];
            Bh[i * LD + j] = acc;
        }
        __syncthreads();
        double psum = 0.0;
        for (int i = 0; i < q; ++i) psum += w.p[i];
        double* Yo = Y + (size_t)b * k_ * k_ * k_;
        for (int e = tid; e < k_ * k_ * k_; e += T) {
            const int blk = e / (k_ * k_), rem = e - blk * k_ * k_;
            const int i = rem / k_, j = rem - i * k_;
            const double wgt = w0 * (blk == 0 ? (1.0 - psum) : w.p[blk - 1]);
            double yi = (i < q) ? w.ci[i] - (blk ? Bh[i * LD + (blk - 1)] : 0.0) : 1.0;
            double yj = (j < q) ? w.ci[j] - (blk ? Bh[j * LD + (blk - 1)] : 0.0) : 1.0;
            Yo[e] = wgt * yi * yj;
        }
    }
    if (tid == 0) { value[b] = -w0 * g; status[b] = st; if (iters) iters[b] = it; }
#ifdef OEI_SDP_TIMING
    __syncthreads();
    if (tid < 8 && tid < k_ * k_) M[(size_t)b * k_ * k_ + tid] = w.scal[tid];     // diagnostic build: phase cycles
#endif
}

extern "C" int oei_sdp_batched(int B, int k_, const double* omega, const double* fmin,
                               double* value, double* M, double* Y, int* status, int* iters,
                               void* stream) {
    if (B < 0 || k_ < 2 || k_ - 1 > OEI_MAX_Q) { oei_set_error("sdp_batched: B=%d k_=%d out of range (q <= %d)", B, k_, OEI_MAX_Q); return OEI_ERR_ARG; }
    if (B == 0) return OEI_OK;
    if (!omega || !fmin || !value || !M || !status) { oei_set_error("sdp_batched: null pointer"); return OEI_ERR_ARG; }
    const int q = k_ - 1;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_SDP(T_, Q_) do { \
        const size_t lds = (sdp_work_doubles(q, false, sdp_mm<Q_>()) + q + 2) * sizeof(double); \
        if (lds > 48 * 1024) OEI_HIP_CHECK(hipFuncSetAttribute((const void*)k_sdp_omega<T_, Q_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_sdp_omega<T_, Q_>), dim3(B), dim3(T_), lds, st, q, omega, fmin, value, M, Y, status, iters); } while (0)
    SDP_DISPATCH_Q(q, LAUNCH_SDP);
#undef LAUNCH_SDP
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}
