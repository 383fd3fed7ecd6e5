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
#include <type_traits>
#include "oei_sens.cuh"

typedef double double4_t __attribute__((ext_vector_type(4)));

// ================================================================================================
// stand-alone sensitivity: sdp.solution_derivative (sdp.py:159-206)
// ================================================================================================
template <int T, int Q>
__global__ __launch_bounds__(T) void k_soldiff(int q, int K, const double* __restrict__ Yin,
                                               const double* __restrict__ fmin,
                                               const double* __restrict__ domega,
                                               double* __restrict__ dM, double* __restrict__ dY,
                                               int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    SdpWork w;
    sdp_carve(w, sm, q, true);
    const int VL = sdp_vlen(q), LD = w.LD, k_ = q + 1;
    lds_f64* mu = (lds_f64*)sm + sdp_work_doubles(q, true);
    SensVec sv;
    sens_carve(sv, mu + VL, q);
    lds_f64* Sc = mu + VL + SENS_NVEC * VL;            // copy of S (+ mu mu^T) for dw0 != 0
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
    int st = sdp_factor<T, Q>(w) ? 2 : 0;
    const double g = st ? -INFINITY : sdp_eval<T, Q>(w, w.p, sweeps, false);
    if (!(g > -INFINITY)) st = 2;
    if (!st && sens_prepare<T, Q>(w, mu, sv)) st = 2;
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
#ifdef OEI_HESS_TIMING
        long long sens_cyc[12] = {};
        const double dnu = sens_direction<T, T == 64, false>(w, sv, SensDir{0, nullptr, nullptr, nullptr, nullptr}, sens_cyc);
#else
        const double dnu = sens_direction<T, T == 64, false>(w, sv, SensDir{0, nullptr, nullptr, nullptr, nullptr});
#endif
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
            lds_f64* Bh = w.V;       // B/2  = Rs R^T
            lds_f64* dBh = Sc;       // dB/2 = -2 Rs Z Rs^T
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
    const size_t lds = (sdp_work_doubles(q, true) + (size_t)(1 + SENS_NVEC) * sdp_vlen(q) + (size_t)q * sdp_ld(q)) * sizeof(double);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (q <= 32) {
        if ((rc = set_lds_attr((k_soldiff<64, 32>), lds))) return rc;
        hipLaunchKernelGGL((k_soldiff<64, 32>), dim3(B), dim3(64), lds, st, q, K, Y, fmin, domega, dM, dY, status);
    } else {
        if ((rc = set_lds_attr((k_soldiff<256, 0>), lds))) return rc;
        hipLaunchKernelGGL((k_soldiff<256, 0>), dim3(B), dim3(256), lds, st, q, K, Y, fmin, domega, dM, dY, status);
    }
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// ================================================================================================
// dOmega/dx tables from the Gram block G_b = E_b E_b^T
// ================================================================================================
// Row indices inside a candidate's block: A_a = a(d+1), T_(a,l) = a(d+1)+1+l, v = q(d+1).
// G is stored as lower block-triangle of full 64x64 tiles, so G(r1, r2) is read at (max, min).
__device__ __forceinline__ double gsym(const double* __restrict__ Gb, int RP, int r1, int r2) {
    return (r1 >= r2) ? Gb[(size_t)r1 * RP + r2] : Gb[(size_t)r2 * RP + r1];
}

// mu[a], DS[a][l][b] = d Sigma_ab / d x_al (b == a: the full derivative of Sigma_aa),
// dmuT[a][l] = d mu_a / d x_al, k1s[a][b] = 2 var kappa'(t_ab), k2s[a][b] = 4 var kappa''(t_ab).
// TP: where the q^2-sized tables live -- lds_f64* (q <= 32) or double* (global memory, q > 32).
template <int T, typename TP>
__device__ __forceinline__ void build_tables(const KernParams& kp, int q, const lds_f64* xq,
                             const double* __restrict__ Gb, int RP, lds_f64* mu, TP DS,
                             lds_f64* dmuT, TP k1s, TP k2s) {
    const int tid = threadIdx.x, d = kp.d, dp = d + 1, vrow = q * dp;
    for (int e = tid; e < q * q; e += T) {
        const int a = e / q, b = e - a * q;
        double t = 0.0;
        for (int l = 0; l < d; ++l) {
            const double df = xq[a * d + l] - xq[b * d + l];
            t += df * df * kp.inv_ls2[l];
        }
        double k0, k1, k2;
        kappa_eval<2>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        k1s[e] = (a == b) ? 0.0 : 2.0 * kp.variance * k1;
        k2s[e] = (a == b) ? 0.0 : 4.0 * kp.variance * k2;
    }
    for (int a = tid; a < q; a += T) mu[a] = gsym(Gb, RP, vrow, a * dp);
    for (int e = tid; e < q * d; e += T) {
        const int a = e / d, l = e - a * d;
        dmuT[e] = gsym(Gb, RP, vrow, a * dp + 1 + l);
    }
    __syncthreads();
    for (int e = tid; e < q * d * q; e += T) {
        const int a = e / (d * q), rem = e - a * d * q, l = rem / q, b = rem - l * q;
        const double ta = gsym(Gb, RP, a * dp + 1 + l, b * dp);       // T_(a,l) . A_b
        DS[e] = (a == b) ? -2.0 * ta
                         : k1s[a * q + b] * (xq[a * d + l] - xq[b * d + l]) * kp.inv_ls2[l] - ta;
    }
    __syncthreads();
}

// ---- domega (oei.py:125-135): dense [k_, k_, q*d] -----------------------------------------------
// Every non-zero entry needs one entry of dSigma/dx, which is formed on the spot from the Gram
// block: no q*d*q table, so the kernel fits every q <= 64.
template <int T>
__global__ __launch_bounds__(T) void k_domega(KernParams kp, int q, int RP,
                                              const double* __restrict__ Xq,
                                              const double* __restrict__ G,
                                              double* __restrict__ domega) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    const int d = kp.d, dp = d + 1, qd = q * d, k_ = q + 1, vrow = q * dp;
    lds_f64* xq = (lds_f64*)sm;          // [q*d]
    lds_f64* mu = xq + qd;               // [q]
    lds_f64* dmuT = mu + q;              // [q*d]
    lds_f64* k1s = dmuT + qd;            // [q*q]
    const int tid = threadIdx.x;
    const size_t b = blockIdx.x;
    const double* Gb = G + b * (size_t)RP * RP;
    for (int e = tid; e < qd; e += T) xq[e] = Xq[b * qd + e];
    __syncthreads();
    for (int e = tid; e < q * q; e += T) {
        const int a = e / q, c = e - a * q;
        double t = 0.0;
        for (int l = 0; l < d; ++l) {
            const double df = xq[a * d + l] - xq[c * d + l];
            t += df * df * kp.inv_ls2[l];
        }
        double k0, k1, k2;
        kappa_eval<1>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        k1s[e] = (a == c) ? 0.0 : 2.0 * kp.variance * k1;
    }
    for (int a = tid; a < q; a += T) mu[a] = gsym(Gb, RP, vrow, a * dp);
    for (int e = tid; e < qd; e += T) {
        const int a = e / d, l = e - a * d;
        dmuT[e] = gsym(Gb, RP, vrow, a * dp + 1 + l);
    }
    __syncthreads();
    // d Sigma_ac / d x_al (c == a: the full derivative of Sigma_aa)
    auto ds = [&](int a, int l, int c) -> double {
        const double ta = gsym(Gb, RP, a * dp + 1 + l, c * dp);       // T_(a,l) . A_c
        return (a == c) ? -2.0 * ta
                        : k1s[a * q + c] * (xq[a * d + l] - xq[c * d + l]) * kp.inv_ls2[l] - ta;
    };
    double* out = domega + b * (size_t)k_ * k_ * qd;
    for (size_t e = tid; e < (size_t)k_ * k_ * qd; e += T) {
        const int ij = (int)(e / qd), al = (int)(e - (size_t)ij * qd);
        const int i = ij / k_, j = ij - i * k_, a = al / d, l = al - a * d;
        double val = 0.0;
        if (i < q && j < q) {
            if (i == j) { if (a == i) val = ds(a, l, a) + 2.0 * mu[a] * dmuT[al]; }
            else {
                if (a == i) val = ds(a, l, j) + dmuT[al] * mu[j];
                else if (a == j) val = ds(a, l, i) + dmuT[al] * mu[i];
            }
        } else if (i < q) { if (a == i) val = dmuT[al]; }
        else if (j < q) { if (a == j) val = dmuT[al]; }
        out[e] = val;
    }
}

// ================================================================================================
// acquisition Hessian (oei.py:41-64, 105-123)
// ================================================================================================
// H[(a,l),(b,m)] = d2<Omega,M>/dx2 |_M  +  <dM_(a,l), dOmega_(b,m)>, with
//   d2<Omega,M>/dx2 = 2 M11_ab dmu_al dmu_bm - 2 M11_ab T_al.T_bm - [a!=b] 2 M11_ab Hs_ab[l][m]
//                     + [a==b] Dg_a[l][m],
//   Dg_a = sum_k (p_a alpha_k - 2 W[a,k]) d2k(x_a,X_k)/dx_a^2 + 2 sum_{c!=a} M11_ac Hs_ac,
//   Hs_ab[l][m] = d2k(x_a,x_b)/dx_al dx_am.
#define HCH_MIN 16

__host__ __device__ inline size_t hess_tab_doubles(int q, int d) {
    const int dl = (d + 1) | 1;
    size_t ds = (size_t)q * d * q, chunk = (size_t)2 * q * (HCH_MIN + 1) + (size_t)HCH_MIN * dl;
    return ds > chunk ? ds : chunk;
}


// Dg pre-pass: Dg_a[l][m] = sum_k (p_a alpha_k - 2 W[a,k]) d2k(x_a,X_k)/dx_al dx_am, training points
// staged in `tab` in chunks.  T threads; leaves a barrier behind.
template <int T, typename DP>
__device__ __forceinline__ void hess_dg_prepass(const KernParams& kp, int n, int npad, int q,
                                                const lds_f64* xq, const lds_f64* pv, lds_f64* tab,
                                                size_t tabsz, DP Dg, const double* __restrict__ Xtrain,
                                                const double* __restrict__ alpha,
                                                const double* __restrict__ Wrows) {
    const int tid = threadIdx.x, d = kp.d, dl = (d + 1) | 1;
    int ch = (int)((tabsz - 2 * q) / (2 * q + dl));
    ch = (ch / 16) * 16;
    if (ch > 64) ch = 64;
    lds_f64* c1 = tab;                        // [q][ch+1]  coef * 2 var kappa'
    lds_f64* c2 = c1 + q * (ch + 1);          // [q][ch+1]  coef * 4 var kappa''
    lds_f64* Xc = c2 + q * (ch + 1);          // [ch][dl]
    const int nsym = d * (d + 1) / 2;
    const int nitems = q * nsym;             // per a: one item per (l <= m)
    for (int e = tid; e < q * d * d; e += T) Dg[e] = 0.0;
    for (int c0 = 0; c0 < n; c0 += ch) {
        __syncthreads();
        for (int e = tid; e < ch * d; e += T) {
            const int k = e / d, l = e - k * d, kg = c0 + k;
            Xc[k * dl + l] = (kg < n) ? Xtrain[(size_t)kg * d + l] : 0.0;
        }
        __syncthreads();
        for (int e = tid; e < q * ch; e += T) {
            const int a = e / ch, k = e - a * ch, kg = c0 + k;
            double v1 = 0.0, v2 = 0.0;
            if (kg < n) {
                double t = 0.0;
                for (int l = 0; l < d; ++l) {
                    const double df = xq[a * d + l] - Xc[k * dl + l];
                    t += df * df * kp.inv_ls2[l];
                }
                double k0, k1, k2;
                kappa_eval<2>(kp.kind, kp.floor_mode, t, k0, k1, k2);
                const double coef = pv[a] * alpha[kg] - 2.0 * Wrows[(size_t)a * npad + kg];
                v1 = coef * 2.0 * kp.variance * k1;
                v2 = coef * 4.0 * kp.variance * k2;
            }
            c1[a * (ch + 1) + k] = v1;
            c2[a * (ch + 1) + k] = v2;
        }
        __syncthreads();
        for (int e = tid; e < nitems; e += T) {
            // entry (a, l, m2 >= l) and its mirror belong to this thread alone
            const int a = e / nsym, it = e - a * nsym;
            int l = 0, rem = it;
            while (rem >= d - l) { rem -= d - l; ++l; }
            const int m2 = l + rem;
            double s = 0.0, s1 = 0.0;
            const double xl = xq[a * d + l], xm = xq[a * d + m2];
            for (int k = 0; k < ch; ++k) {
                s += c2[a * (ch + 1) + k] * (xl - Xc[k * dl + l]) * (xm - Xc[k * dl + m2]);
                s1 += c1[a * (ch + 1) + k];
            }
            s *= kp.inv_ls2[l] * kp.inv_ls2[m2];
            if (m2 == l) s += s1 * kp.inv_ls2[l];       // 2 var kappa' delta_lm / ls_l^2
            Dg[(a * d + l) * d + m2] += s;
            if (m2 != l) Dg[(a * d + m2) * d + l] += s;
        }
    }
    __syncthreads();
}

// The same sums for a workgroup of NW = T / 64 waves with per-wave scratch (k_hess_dirs, d <= 15):
// the second-moment part sum_k c2[a,k] (x_al - X_kl)(x_am - X_km) runs on the matrix pipe.  A wave
// owns `na` = 16 / (d + 1) points a; row i = ia (d + 1) + r of its 16 x 16 tile is coordinate r of
// point ia (r < d), or the row that sums c1 (r = d: left operand c1, right operand 1).  Per four
// training points lane (i, kk) supplies c2[a,k] diff and diff (k = 4 s + kk), the same index
// pattern on both sides, so one LDS read pair feeds both operands; the diagonal blocks of the tile
// are the wanted sums, the off-diagonal blocks (products across points) are ignored.  The tile
// accumulates over all training points in registers.  The training inputs are staged in `tab` once
// per workgroup when they fit (n ((d + 1) | 1) <= tabsz; else per wave and chunk); after that every
// wave works alone on its points, in chunks of up to 128 training points whose coefficients go
// through its own scratch (`wsz` doubles): kernel derivatives of the chunk's (point, training
// point) pairs by the 64 lanes, four pairs per lane at a time so that their global loads are in
// flight together, then the matrix instructions.  No workgroup barrier inside the pass.  (With
// chunks of 36 staged per workgroup the pass was three barriers and five serial global-load
// latencies per chunk: 86 k of a candidate's 530 k cycles.)  Leaves a barrier behind.
template <int T, typename DP, bool ACC = false>
__device__ __forceinline__ void hess_dg_prepass_mfma(const KernParams& kp, int n, int npad, int q,
                                                     const lds_f64* xq, const lds_f64* pv, lds_f64* tab,
                                                     size_t tabsz, DP Dg, lds_f64* wscratch, size_t wsz,
                                                     const double* __restrict__ Xtrain,
                                                     const double* __restrict__ alpha,
                                                     const double* __restrict__ Wrows
#ifdef OEI_HESS_TIMING
                                                     , long long (&dt_)[4]
#endif
                                                     ) {
    constexpr int NW = T / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = kp.d, dl = (d + 1) | 1, dp = d + 1;
    const int na = 16 / dp;                              // points per tile (d <= 15: >= 1)
    const int li = lane & 15, kk = lane >> 4;
    const int ia = li / dp, r = li - ia * dp;
    const bool rowlive = ia < na;
    const float rd = 1.0f / (float)d;
#ifdef OEI_HESS_TIMING
    long long tt_ = clock64();
#define PRE_STAMP(i) do { const long long t2_ = clock64(); dt_[i] += t2_ - tt_; tt_ = t2_; } while (0)
#else
#define PRE_STAMP(i) do { } while (0)
#endif
    const bool xall = (size_t)n * dl <= tabsz;           // uniform
    if (xall) {
        for (int e = tid; e < n * d; e += T) {
            const int k = (int)(((float)e + 0.5f) * rd), l = e - k * d;      // exact for e < 2^22 / d
            tab[k * dl + l] = Xtrain[e];
        }
    }
    __syncthreads();
    PRE_STAMP(0);
    int chs = 7;                                         // chunk length 2^chs <= 128 that fits the scratch
    while (chs > 2 && (size_t)(2 * na * ((1 << chs) + 1) + (xall ? 0 : (1 << chs) * dl)) > wsz) --chs;
    const int ch = 1 << chs, ch1 = ch + 1;
    lds_f64* c1 = wscratch;                   // [na][ch+1]  coef * 2 var kappa'
    lds_f64* c2 = c1 + na * ch1;              // [na][ch+1]  coef * 4 var kappa''
    lds_f64* Xw = c2 + na * ch1;              // [ch][dl] (only without the shared copy)
    for (int a_lo = wave * na; a_lo < q; a_lo += NW * na) {
        const int a_cnt = min(q - a_lo, na);
        const bool live = rowlive && ia < a_cnt;
        const int ac = live ? ia : 0;
        const double xa = (live && r < d) ? xq[(a_lo + ac) * d + r] : 0.0;
        sens_double4 acc = (sens_double4){0.0, 0.0, 0.0, 0.0};
        for (int c0 = 0; c0 < n; c0 += ch) {
            sens_sync<true>();
            if (!xall) {
                for (int e = lane; e < ch * d; e += 64) {
                    const int k = (int)(((float)e + 0.5f) * rd), l = e - k * d, kg = c0 + k;
                    Xw[k * dl + l] = (kg < n) ? Xtrain[(size_t)kg * d + l] : 0.0;
                }
                sens_sync<true>();
            }
            const lds_f64* Xc = xall ? tab + (size_t)c0 * dl : Xw;      // row k of the chunk at Xc + k dl
#pragma unroll 4
            for (int e = lane; e < a_cnt * ch; e += 64) {
                const int aa = e >> chs, k = e & (ch - 1), kg = c0 + k, a = a_lo + aa;
                const bool in = kg < n;
                const int kgc = in ? kg : 0;
                const double al = alpha[kgc], wv = Wrows[(size_t)a * npad + kgc];
                double t = 0.0;
                for (int l = 0; l < d; ++l) {
                    const double df = xq[a * d + l] - Xc[(in ? k : 0) * dl + l];
                    t += df * df * kp.inv_ls2[l];
                }
                double k0, k1, k2;
                kappa_eval<2>(kp.kind, kp.floor_mode, t, k0, k1, k2);
                const double coef = pv[a] * al - 2.0 * wv;
                c1[aa * ch1 + k] = in ? coef * 2.0 * kp.variance * k1 : 0.0;
                c2[aa * ch1 + k] = in ? coef * 4.0 * kp.variance * k2 : 0.0;
            }
            sens_sync<true>();
            PRE_STAMP(1);
            const int kend = min(ch, (n - c0 + 3) & ~3);            // whole groups of four; entries past n are zeros
            // four steps at a time: their eight LDS reads first, then the matrix instructions
            // (one step at a time every instruction waited for its own reads: 350 cycles a step)
            const lds_f64* crow = (r < d ? c2 : c1) + ac * ch1 + kk;
            const lds_f64* xcol = Xc + (r < d ? r : 0);
            for (int s4 = 0; s4 < kend; s4 += 16) {
                double xv[4], cv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = s4 + 4 * u + kk;
                    const bool in = c0 + k < n && s4 + 4 * u < kend;
                    xv[u] = xcol[(in ? k : 0) * dl];
                    cv[u] = crow[in ? s4 + 4 * u : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool in = c0 + s4 + 4 * u + kk < n && s4 + 4 * u < kend;
                    const double diff = xa - xv[u];
                    const double av = (live && in) ? (r < d ? cv[u] * diff : cv[u]) : 0.0;
                    const double bv = (live && in) ? (r < d ? diff : 1.0) : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
            }
            PRE_STAMP(2);
        }
        // tile -> the wave's scratch (row-major 16 x 16), then the diagonal blocks -> Dg
        sens_sync<true>();
#pragma unroll
        for (int g = 0; g < 4; ++g) wscratch[(kk + 4 * g) * 16 + li] = acc[g];
        sens_sync<true>();
        for (int e = lane; e < a_cnt * d * d; e += 64) {
            const int ja = e / (d * d), rem = e - ja * d * d, l = rem / d, m2 = rem - l * d;
            const int a = a_lo + ja;
            const int lo = min(l, m2), hi = max(l, m2);          // one triangle: exactly symmetric
            double v = wscratch[(ja * dp + lo) * 16 + ja * dp + hi] * kp.inv_ls2[l] * kp.inv_ls2[m2];
            if (l == m2) v += wscratch[(ja * dp + d) * 16 + ja * dp + d] * kp.inv_ls2[l];
            if (ACC) Dg[(a * d + l) * d + m2] += v;      // a later piece of the training set (same wave, same lane)
            else Dg[(a * d + l) * d + m2] = v;
        }
        sens_sync<true>();
    }
    PRE_STAMP(3);
    __syncthreads();
}

template <int T, typename MP, typename TP, typename DP>
__device__ __forceinline__ void hess_dg_selfcov(const KernParams& kp, int q, const lds_f64* xq,
                                                MP M11, TP k1s, TP k2s, DP Dg) {
    const int d = kp.d;
    for (int e = threadIdx.x; e < q * d * d; e += T) {
        const int a = e / (d * d), rem = e - a * d * d, l = rem / d, m2 = rem - l * d;
        double s = 0.0;
        for (int c = 0; c < q; ++c) {
            if (c == a) continue;
            const double sdl = (xq[a * d + l] - xq[c * d + l]) * kp.inv_ls2[l];
            const double sdm = (xq[a * d + m2] - xq[c * d + m2]) * kp.inv_ls2[m2];
            s += M11[a * q + c] * (k2s[a * q + c] * sdl * sdm + (l == m2 ? k1s[a * q + c] * kp.inv_ls2[l] : 0.0));
        }
        Dg[e] += 2.0 * s;
    }
}

// The same sums with the tables in global memory (the wide kernel): the weights M11 k2s and
// M11 k1s go through two q x LD LDS matrices first (zero at c = a), so that the q d^2 sums of q
// terms read LDS only.  Contains a barrier; the caller synchronises before the matrices are reused.
template <int T>
__device__ __forceinline__ void hess_dg_selfcov_staged(const KernParams& kp, int q, int LD, const lds_f64* xq,
                                                       const double* M11, const double* k1s, const double* k2s,
                                                       lds_f64* W2, lds_f64* W1, double* Dg) {
    const int d = kp.d;
    for (int e = threadIdx.x; e < q * q; e += T) {
        const int a = e / q, c = e - a * q;
        const double m = (a == c) ? 0.0 : M11[e];
        W2[a * LD + c] = m * k2s[e];
        W1[a * LD + c] = m * k1s[e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < q * d * d; e += T) {
        const int a = e / (d * d), rem = e - a * d * d, l = rem / d, m2 = rem - l * d;
        const double xl = xq[a * d + l], xm = xq[a * d + m2];
        double s0 = 0.0, s1 = 0.0, t1 = 0.0;
        int c = 0;
        for (; c + 2 <= q; c += 2) {
            const double w0 = W2[a * LD + c], w1 = W2[a * LD + c + 1];
            const double l0 = xl - xq[c * d + l], l1 = xl - xq[(c + 1) * d + l];
            const double m0 = xm - xq[c * d + m2], m1 = xm - xq[(c + 1) * d + m2];
            s0 = fma(w0 * l0, m0, s0);
            s1 = fma(w1 * l1, m1, s1);
            if (l == m2) t1 += W1[a * LD + c] + W1[a * LD + c + 1];
        }
        for (; c < q; ++c) {
            s0 = fma(W2[a * LD + c] * (xl - xq[c * d + l]), xm - xq[c * d + m2], s0);
            if (l == m2) t1 += W1[a * LD + c];
        }
        Dg[e] += 2.0 * ((s0 + s1) * kp.inv_ls2[l] * kp.inv_ls2[m2] + t1 * kp.inv_ls2[l]);
    }
}

// Row (a,l) of the Hessian from the direction's dN (q x LD), w1, w2, dN mu: the contraction with
// every dOmega_(b,m) plus d2<Omega,M>/dx2 at fixed M.  `nth` threads numbered `tid` share the row.
// WVFORM: `dN` holds Y = U Z and `DS` the table WV[b,m] = U^T v'_(b,m) in place of dSigma/dx, so
// that <dN, dS_(b,m)> = 2 (dN v')_b = 2 Y[b,:] . WV[b,m]  (in LDS: row-major; in global memory: transposed).
template <bool WVFORM = false, typename MP, typename TP>
__device__ __forceinline__ void hess_contract(int tid, int nth, const KernParams& kp, int q, int RP,
                                              int a, int l, const lds_f64* dN, int LD,
                                              TP DS, const lds_f64* dmuT,
                                              const lds_f64* w1, const lds_f64* w2,
                                              const lds_f64* dNmu, const double* __restrict__ Gb,
                                              MP M11, const lds_f64* xq,
                                              TP k1s, TP k2s,
                                              TP Dg, double* __restrict__ Hb,
                                              const double* ttpre = nullptr) {
    const int d = kp.d, dp = d + 1, qd = q * d, al = a * d + l;
    // the part of entry (al, bm) that does not depend on <dN, dS_(b,m)> (`acc`), and the store
    auto finish = [&](int bm, int pass, double acc) {
        const int bb = bm / d, m2 = bm - bb * d;
        // <dM, dOmega_(b,m)>, dM11 = -dN, dM[:q,q] = w1 + w2
        double P = -acc + 2.0 * dmuT[bm] * (w1[bb] + w2[bb] - dNmu[bb]);
        // d2<Omega,M>/dx2 at fixed M
        // T_al . T_bm: the first two passes may have been fetched ahead by the caller
        const double tt = (ttpre && pass < 2) ? (pass ? ttpre[1] : ttpre[0])
                                              : gsym(Gb, RP, a * dp + 1 + l, bb * dp + 1 + m2);
        double inner = 2.0 * M11[a * q + bb] * (dmuT[al] * dmuT[bm] - tt);
        if (a != bb) {
            const double sdl = (xq[a * d + l] - xq[bb * d + l]) * kp.inv_ls2[l];
            const double sdm = (xq[a * d + m2] - xq[bb * d + m2]) * kp.inv_ls2[m2];
            inner -= 2.0 * M11[a * q + bb] *
                     (k2s[a * q + bb] * sdl * sdm + (l == m2 ? k1s[a * q + bb] * kp.inv_ls2[l] : 0.0));
        } else {
            inner += Dg[(a * d + l) * d + m2];
        }
        Hb[(size_t)al * qd + bm] = inner + P;
    };
    if constexpr (WVFORM && std::is_same<TP, const lds_f64*>::value) {
        // two rows per thread and walk (bm and bm + nth): their dot products are independent
        // chains, so one's LDS latency passes behind the other's.  Rows of the table are q doubles
        // apart, so rows 8 apart start on the same LDS bank (q = 20): each row is walked from a
        // different element.
        int pass = 0;
        for (int bm = tid; bm < qd; bm += 2 * nth, pass += 2) {
            const bool two = bm + nth < qd;
            const int bm2 = two ? bm + nth : bm;
            const int bb = bm / d, bb2 = bm2 / d;
            const auto row1 = DS + (size_t)bm * q, row2 = DS + (size_t)bm2 * q;
            const int rot1 = q > 4 ? (bm >> 3) & 3 : 0, rot2 = q > 4 ? (bm2 >> 3) & 3 : 0;
            double acc1 = 0.0, acc2 = 0.0;
#pragma unroll 4
            for (int c = 0; c < q; ++c) {
                const int c1 = c + rot1 < q ? c + rot1 : c + rot1 - q;
                const int c2 = c + rot2 < q ? c + rot2 : c + rot2 - q;
                acc1 += dN[bb * LD + c1] * row1[c1];
                acc2 += dN[bb2 * LD + c2] * row2[c2];
            }
            finish(bm, pass, 2.0 * acc1);
            if (two) finish(bm2, pass + 1, 2.0 * acc2);
        }
    } else {
        int pass = 0;
        for (int bm = tid; bm < qd; bm += nth, ++pass) {
            const int bb = bm / d;
            double acc = 0.0;
            const auto dsrow = DS + (size_t)bm * q;
            if constexpr (WVFORM) {          // table in global memory, transposed (q > 32): ten loads in flight
                double acc1 = 0.0;
                const auto col = DS + bm;                    // entry c of row bm at col[c * qd]
                int c = 0;
                for (; c + 10 <= q; c += 10) {
                    double tv[10];
#pragma unroll
                    for (int u = 0; u < 10; ++u) tv[u] = col[(size_t)(c + u) * qd];
#pragma unroll
                    for (int u = 0; u < 10; u += 2) {
                        acc = fma(dN[bb * LD + c + u], tv[u], acc);
                        acc1 = fma(dN[bb * LD + c + u + 1], tv[u + 1], acc1);
                    }
                }
                for (; c < q; ++c) acc = fma(dN[bb * LD + c], col[(size_t)c * qd], acc);
                acc = 2.0 * (acc + acc1);
            } else {
#pragma unroll 4
                for (int c = 0; c < q; ++c) acc += dN[bb * LD + c] * dsrow[c] * (c == bb ? 1.0 : 2.0);
            }
            finish(bm, pass, acc);
        }
    }
}

// ---- q > 32: one 256-thread workgroup per candidate ------------------------------------------------
// LDS holds three q x q matrices (the solver's; for the directions U and two scratch matrices: R is
// expressed through U and the inverse of the Newton factor sits in registers, see
// sens_direction_lean), the solver's vectors, the candidate's points and d mu/dx: 80 KB at q = 50,
// d = 10, so that two workgroups share a compute unit and fill each other's barrier waits.  The
// q^2-sized tables (dSigma/dx: q*d*q doubles, the kernel-derivative factors, the diagonal blocks
// Dg) live in a per-candidate block of global memory `tabs`; they are written once, then read by
// the contraction of every direction (L2-resident).  The table the contraction walks, WV = rows
// U^T v'_(b,m), is stored transposed (entry c of row (b,m) at WVT[c q d + (b,m)]): a thread owns a
// row, so the 64 rows a wave reads at a time are 512 contiguous bytes per entry; row-major, every
// load instruction touched 64 cache lines and the contraction was half of a direction's time.
// Fits every q <= 64 for d <= 12.
__host__ __device__ inline size_t hess_wide_table_doubles(int q, int d) {
    return 2 * (size_t)q * d * q + 2 * (size_t)q * q + (size_t)q * d * d;  // DS | k1s | k2s | Dg | WVT
}
__host__ __device__ inline size_t hess_wide_lds_doubles(int q, int d) {
    // H S X | 15 solver vectors + 8 scalars | mu, 9 sensitivity vectors | xq | dmuT
    // (dNmu, utc, utmu, wv, scl of the directions take the place of solver vectors that are dead by then)
    return (size_t)3 * q * sdp_ld(q) + (size_t)(SDP_NVEC + 1 + SENS_NVEC) * sdp_vlen(q) + 8 + 2 * (size_t)q * d;
}

// Rows (b,m) of the q d x q table DS in global memory (entry c = b halved) times U (q x q, LDS) on
// the matrix pipe, 16 rows per wave at a time; the product goes to WVT transposed (q x q d).
// KS = number of k steps (4 KS >= q), a template parameter for straight-line code.
template <int T, int KS>
__device__ __forceinline__ void hess_table_transform(int q, int LD, int d, const lds_f64* U, const double* DS,
                                                     double* WVT) {
    constexpr int NCT = (KS + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kg = lane >> 4, qd = q * d;
    for (int I = wave; 16 * I < qd; I += T / 64) {
        const int bm = 16 * I + li;
        const bool rin = bm < qd;
        const int bmc = rin ? bm : 0, bb = bmc / d;
        const double* row = DS + (size_t)bmc * q;
        double av[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 4 * s + kg;
            const double x = row[c < q ? c : 0];
            av[s] = (rin && c < q) ? (c == bb ? 0.5 * x : x) : 0.0;
        }
        sens_double4 acc[NCT];
#pragma unroll
        for (int J = 0; J < NCT; ++J) acc[J] = (sens_double4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 4 * s + kg, cc = c < q ? c : 0;
#pragma unroll
            for (int J = 0; J < NCT; ++J) {
                const int j = 16 * J + li;
                const double u = U[cc * LD + (j < q ? j : 0)];
                acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], (c < q && j < q) ? u : 0.0, acc[J], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * I + kg + 4 * g;
#pragma unroll
            for (int J = 0; J < NCT; ++J) {
                const int j = 16 * J + li;
                if (r < qd && j < q) WVT[(size_t)j * qd + r] = acc[J][g];
            }
        }
    }
}

template <int T>
__global__ __launch_bounds__(T, 2) void k_hess_wide(KernParams kp, double noise, double fmin, int n,
                                                 int npad, int q, int RP,
                                                 const double* __restrict__ Xtrain,
                                                 const double* __restrict__ alpha,
                                                 const double* __restrict__ Xq,
                                                 const double* __restrict__ G,
                                                 const double* __restrict__ Wm,
                                                 const double* __restrict__ p_in,
                                                 const double* __restrict__ M11_in,
                                                 const int* __restrict__ sdp_status,
                                                 double* tabs,
                                                 double* __restrict__ Hout, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
#ifdef OEI_HESS_TIMING
    const long long t_kernel0 = clock64();
#endif
    const int VL = sdp_vlen(q), LD = sdp_ld(q), ML = q * LD, d = kp.d, dp = d + 1, qd = q * d;
    SdpWork w;
    w.q = q; w.LD = LD;
    lds_f64* ptr = (lds_f64*)sm;
    w.H = ptr; ptr += ML;  w.S = ptr; ptr += ML;  w.X = ptr; ptr += ML;
    w.V = w.R = w.Kt = nullptr;
    w.p = ptr; ptr += VL;  w.pn = ptr; ptr += VL;  w.s = ptr; ptr += VL;  w.dl = ptr; ptr += VL;
    w.sq = ptr; ptr += VL; w.grad = ptr; ptr += VL; w.step = ptr; ptr += VL; w.tv = ptr; ptr += VL;
    w.sv = ptr; ptr += VL; w.rp = ptr; ptr += VL;  w.ci = ptr; ptr += VL;  w.ip = ptr; ptr += VL;
    w.wp = ptr; ptr += VL; w.rc = ptr; ptr += VL;  w.rs = ptr; ptr += VL;
    w.scal = ptr; ptr += 8;
    lds_f64* mu = ptr; ptr += VL;
    SensVec sv;
    sens_carve(sv, ptr, q); ptr += SENS_NVEC * VL;
    // written only after the solve, in the place of solver vectors nothing reads any more
    lds_f64* dNmu = w.pn;
    lds_f64* utc = w.grad;
    lds_f64* utmu = w.tv;
    lds_f64* wv = w.wp;
    lds_f64* scl = w.rs;
    lds_f64* xq = ptr; ptr += qd;
    lds_f64* dmuT = ptr;
    const int tid = threadIdx.x;
    const size_t b = blockIdx.x;
    const double* Gb = G + b * (size_t)RP * RP;
    const double* M11 = M11_in + b * (size_t)q * q;
    double* Hb = Hout + b * (size_t)qd * qd;
    // gridDim.y workgroups share a candidate's directions (small batches): slice s takes the
    // directions s, s + S, ... and works on its own copy of the tables
    const int slice = blockIdx.y, nslice = gridDim.y;
    double* DS = tabs + (b * nslice + slice) * hess_wide_table_doubles(q, d);
    double* k1s = DS + (size_t)qd * q;
    double* k2s = k1s + (size_t)q * q;
    double* Dg = k2s + (size_t)q * q;
    double* WVT = Dg + (size_t)qd * d;
    for (int e = tid; e < qd; e += T) xq[e] = Xq[b * qd + e];
    if (tid < q) w.p[tid] = p_in[b * q + tid];
    __syncthreads();

    // the three matrices (adjacent, nothing in them yet) are the scratch of the Dg pre-pass: a quarter
    // per wave for the matrix-pipe form (a point's d + 1 rows fit the 16-row tile), else one staging area
    if (d < 16) {
#ifdef OEI_HESS_TIMING
        long long pre_cyc[4] = {0, 0, 0, 0};
#define PRE_CYC_ARG , pre_cyc
#else
#define PRE_CYC_ARG
#endif
        // The training inputs are staged once per workgroup, in as many pieces as the space behind
        // the waves' coefficient scratch needs (two at n = 1000, d = 10), and every wave runs its
        // points over the staged piece; staged per wave and chunk, each of a wave's 13 points
        // fetched the whole training set again.
        const int dl = (d + 1) | 1, na = 16 / (d + 1);
        const size_t wsz = (size_t)max(256, 2 * na * 129);
        const size_t cap = (size_t)3 * ML > (T / 64) * wsz ? (size_t)3 * ML - (T / 64) * wsz : 0;
        lds_f64* wsc = w.H + (tid >> 6) * wsz;
        lds_f64* stage = w.H + (T / 64) * wsz;
        if (cap >= (size_t)64 * dl) {
            const int npc = (int)(((size_t)n * dl + cap - 1) / cap);
            int nh = (n + npc - 1) / npc;
            nh = (nh + 3) & ~3;
            if ((size_t)nh * dl > cap) nh = (int)(cap / dl) & ~3;
            const double* Wr = Wm + b * (size_t)q * npad;
            hess_dg_prepass_mfma<T, double*, false>(kp, min(nh, n), npad, q, xq, w.p, stage, cap, Dg, wsc, wsz, Xtrain, alpha, Wr PRE_CYC_ARG);
            for (int k0 = nh; k0 < n; k0 += nh)
                hess_dg_prepass_mfma<T, double*, true>(kp, min(nh, n - k0), npad, q, xq, w.p, stage, cap, Dg, wsc, wsz,
                                                       Xtrain + (size_t)k0 * d, alpha + k0, Wr + k0 PRE_CYC_ARG);
        } else {
            const size_t wsz1 = (size_t)3 * ML / (T / 64);
            hess_dg_prepass_mfma<T>(kp, n, npad, q, xq, w.p, (lds_f64*)nullptr, 0, Dg, w.H + (tid >> 6) * wsz1, wsz1, Xtrain, alpha,
                                    Wm + b * (size_t)q * npad PRE_CYC_ARG);
        }
#undef PRE_CYC_ARG
    } else {
        hess_dg_prepass<T>(kp, n, npad, q, xq, w.p, w.H, (size_t)3 * ML, Dg, Xtrain, alpha, Wm + b * (size_t)q * npad);
    }
#ifdef OEI_HESS_TIMING
    const long long t_pre1 = clock64();
#endif
    build_tables<T>(kp, q, xq, Gb, RP, mu, DS, dmuT, k1s, k2s);
#ifdef OEI_HESS_TIMING
    const long long t_pre2 = clock64();
#endif
    for (int e = tid; e < q * q; e += T) {
        const int a = e / q, c = e - a * q;
        double t = 0.0;
        for (int l = 0; l < d; ++l) {
            const double df = xq[a * d + l] - xq[c * d + l];
            t += df * df * kp.inv_ls2[l];
        }
        double k0, k1, k2;
        kappa_eval<0>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        w.S[a * LD + c] = kp.variance * k0 - gsym(Gb, RP, a * dp, c * dp) + (a == c ? noise : 0.0);
    }
    if (tid < q) w.dl[tid] = mu[tid] - fmin;
    hess_dg_selfcov_staged<T>(kp, q, LD, xq, M11, (const double*)k1s, (const double*)k2s, w.H, w.X, Dg);
    __syncthreads();

    // ---- state at the optimum: U, the factor of -hess g, c, t1 ----------------------------------
#ifdef OEI_HESS_TIMING
    const long long t_state0 = clock64();
#endif
    int sweeps = 0;
    int st = sdp_factor<T, 0>(w) ? 2 : 0;
    const double g = st ? -INFINITY : sdp_eval<T, 0>(w, w.p, sweeps, false);
    if (!(g > -INFINITY)) st = 2;
    if (!st) {
        if (sdp_newton_dir<T, 0>(w)) st = 2;          // reads U from w.X; w.S (Lc) is free afterwards
    }
#ifdef OEI_HESS_TIMING
    const long long t_state1 = clock64();
#endif
    if (st) {
        // every slice answers for its own rows: the slices repeat the same solve, but one that
        // breaks down alone must not leave its rows unwritten behind a status of 0
        for (int al = slice; al < qd; al += nslice)
            for (int e = tid; e < qd; e += T) Hb[(size_t)al * qd + e] = NAN;
        if (tid == 0 && status) atomicMax(status + b, 2);
        return;
    }
    __syncthreads();
    for (int e = tid; e < q * q; e += T) {
        const int i = e / q, a = e - i * q;
        w.S[i * LD + a] = w.X[i * LD + a];
    }
    if (tid < q) {
        double acc = 0.0;                             // c = mu + R (sq o rp)
        for (int a = 0; a < q; ++a) acc += sdp_R(w, tid, a) * w.sq[a] * w.rp[a];
        sv.c[tid] = mu[tid] + acc;
        sv.t1[tid] = w.sq[tid] * w.rp[tid];
        w.rc[tid] = 1.0 / w.H[tid * LD + tid];
    }
    __syncthreads();
    // Row (b,m) of the dSigma/dx table becomes U^T v'_(b,m) (half weight at j = b), in place: the
    // direction's own input and what the contraction of every other direction meets (see
    // k_hess_dirs).
    if (tid < q) {
        utc[tid] = lds_dot(w.S + tid, LD, sv.c, 1, q);
        utmu[tid] = lds_dot(w.S + tid, LD, mu, 1, q);
    }
#ifdef OEI_HESS_TIMING
    long long t_linv1 = 0;
#endif
    {
        const int lane = tid & 63, wave = tid >> 6;
        // L^-1 for the directions' solves, column j by lane j of the first wave, into w.X (free
        // until the first direction); every thread takes its share into registers behind the barrier
        if (wave == 0) {
            for (int e = lane; e < q * LD; e += 64) w.X[e] = 0.0;
            sdp_wave_sync();
            if (lane < q) {
                const int j = lane;
                w.X[j * LD + j] = w.rc[j];
                for (int i = j + 1; i < q; ++i) {
                    double acc = 0.0;
                    for (int k = j; k < i; ++k) acc = fma(w.H[i * LD + k], w.X[k * LD + j], acc);
                    w.X[i * LD + j] = -acc * w.rc[i];
                }
            }
        }
#ifdef OEI_HESS_TIMING
        t_linv1 = clock64();
#endif
        if (q <= 40) hess_table_transform<T, 10>(q, LD, d, w.S, DS, WVT);
        else if (q <= 52) hess_table_transform<T, 13>(q, LD, d, w.S, DS, WVT);
        else hess_table_transform<T, 16>(q, LD, d, w.S, DS, WVT);
    }
    __syncthreads();
    double lm[16];
    sens_linv_regs(w.X, q, LD, lm);
#ifdef OEI_HESS_TIMING
    const long long extra_cyc[4] = {t_pre1 - t_kernel0, t_pre2 - t_pre1, t_linv1 - t_state1, 0};
#endif

#ifdef OEI_HESS_TIMING
    long long sens_cyc[12] = {};
    const long long t_dirs0 = clock64();
    sens_cyc[11] = t_dirs0 - t_kernel0;
    sens_cyc[8] = t_state0 - t_kernel0;
    sens_cyc[9] = t_state1 - t_state0;
#endif
    // ---- one sensitivity solve per direction, fused with the contraction ------------------------
    for (int al = slice; al < qd; al += nslice) {
        const int a = al / d, l = al - a * d;
        __syncthreads();
        // the row's T_al . T_bm entries come from the candidate's Gram block in global memory (too
        // large to stay cached across a batch): ask for them before the solve
        double ttpre[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int bm = min(tid + T * u, qd - 1), bb = bm / d, m2 = bm - bb * d;
            ttpre[u] = gsym(Gb, RP, a * dp + 1 + l, bb * dp + 1 + m2);
        }
        if (tid < q) {
            sv.dmu[tid] = (tid == a) ? dmuT[al] : 0.0;
            wv[tid] = WVT[(size_t)tid * qd + al];
        }
#ifdef OEI_HESS_TIMING
        sens_direction_lean<T>(w, sv, a, wv, scl, utc, lm, sens_cyc);
        long long sens_t0 = clock64();
#else
        sens_direction_lean<T>(w, sv, a, wv, scl, utc, lm);
#endif
        {
            const int ri = tid >> 2, row = ri < q ? ri : 0;
            const double v = quad_row_sum(q, [&](int k) { return w.H[row * LD + k] * utmu[k]; });
            if (ri < q && (tid & 3) == 0) dNmu[ri] = v;                        // dN mu = Y (U^T mu)
        }
        __syncthreads();
        hess_contract<true>(tid, T, kp, q, RP, a, l, w.H, LD, (const double*)WVT, dmuT, sv.w1, sv.w2, dNmu, Gb, M11, xq,
                            (const double*)k1s, (const double*)k2s, (const double*)Dg, Hb, ttpre);
        SENS_STAMP(7);
    }
#ifdef OEI_HESS_TIMING
    __syncthreads();
    if (b == 0 && slice == 0 && tid == 0) {
        sens_cyc[10] = clock64() - t_dirs0;
        for (int i = 0; i < 12; ++i) Hb[i] = (double)sens_cyc[i];
        for (int i = 0; i < 4; ++i) Hb[12 + i] = (double)extra_cyc[i];
    }
#endif
    if (slice == 0 && tid == 0 && status && sdp_status) atomicMax(status + b, sdp_status[b]);   // status[] starts at 0
}

// ---- register-path sizes (q <= 32): the same work in two kernels ---------------------------------
// k_hess_prep (one wave per candidate) recomputes the solver state at the optimum and leaves what
// the directions share in global memory; k_hess_dirs gives every candidate a workgroup of several
// waves, each of which takes every NW-th direction through its own scratch matrices.  The q*d
// directions of a candidate are independent, and one wave per candidate leaves most of the chip
// idle for the batch sizes a BO loop produces.
//
// state per candidate: U, R, L^-1 (q x LD each), then rc, sq, rp, c, t1 (VL each)
__host__ __device__ inline size_t hess_state_doubles(int q) {
    return (size_t)3 * q * sdp_ld(q) + (size_t)5 * sdp_vlen(q);
}

template <int T, int Q>
__global__ __launch_bounds__(T, SDP_MIN_WAVES(Q)) void k_hess_prep(KernParams kp, double noise, double fmin, int q,
                                                 int RP, const double* __restrict__ Xq,
                                                 const double* __restrict__ G,
                                                 const double* __restrict__ p_in,
                                                 double* __restrict__ state,
                                                 int* __restrict__ stflag) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    SdpWork w;
    sdp_carve(w, sm, q, true, sdp_mm<Q>());
    const int VL = sdp_vlen(q), LD = w.LD, d = kp.d, dp = d + 1, qd = q * d;
    lds_f64* mu = (lds_f64*)sm + sdp_work_doubles(q, true, sdp_mm<Q>());
    SensVec sv;
    sens_carve(sv, mu + VL, q);
    lds_f64* xq = mu + VL + SENS_NVEC * VL;      // [q*d]
    const int tid = threadIdx.x;
    const size_t b = blockIdx.x;
    const double* Gb = G + b * (size_t)RP * RP;
    for (int e = tid; e < qd; e += T) xq[e] = Xq[b * qd + e];
    if (tid < q) {
        w.p[tid] = p_in[b * q + tid];
        mu[tid] = gsym(Gb, RP, q * dp, tid * dp);
        w.dl[tid] = mu[tid] - fmin;
    }
    __syncthreads();
    for (int e = tid; e < q * q; e += T) {
        const int a = e / q, c = e - a * q;
        double t = 0.0;
        for (int l = 0; l < d; ++l) {
            const double df = xq[a * d + l] - xq[c * d + l];
            t += df * df * kp.inv_ls2[l];
        }
        double k0, k1, k2;
        kappa_eval<0>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        w.S[a * LD + c] = kp.variance * k0 - gsym(Gb, RP, a * dp, c * dp) + (a == c ? noise : 0.0);
    }
    __syncthreads();
    int sweeps = 0;
    int st = sdp_factor<T, Q>(w) ? 2 : 0;
    const double g = st ? -INFINITY : sdp_eval<T, Q>(w, w.p, sweeps, false);
    if (!(g > -INFINITY)) st = 2;
    if (!st && sens_prepare<T, Q>(w, mu, sv)) st = 2;
    if (tid == 0) stflag[b] = st;
    if (st) return;
    // L -> L^-1 in place, as a full matrix (zeros above the diagonal): the directions solve with two
    // matrix-vector products.  Lane j forms column j by forward substitution, x_i = -(sum_{k<i}
    // L_ik x_k) / L_ii with x_k = 0 for k < j and x_j = 1 / L_jj; the entries of L are broadcast reads.
    {
        double x[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            x[i] = 0.0;
            if (i < q) {
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) {
                    const double l = w.H[i * LD + k];
                    if (k & 1) acc1 = fma(l, x[k], acc1); else acc0 = fma(l, x[k], acc0);
                }
                const double rci = w.rc[i];
                x[i] = (i == tid) ? rci : -(acc0 + acc1) * rci;
            }
        }
        __syncthreads();
        if (tid < q) {
#pragma unroll
            for (int i = 0; i < Q; ++i) if (i < q) w.H[i * LD + tid] = (i >= tid) ? x[i] : 0.0;
        }
        __syncthreads();
    }
    double* sb = state + b * hess_state_doubles(q);
    const int ML = q * LD;
    for (int e = tid; e < ML; e += T) {
        sb[e] = w.S[e];
        sb[ML + e] = w.R[e];
        sb[2 * ML + e] = w.H[e];
    }
    if (tid < q) {
        double* vb = sb + 3 * ML;
        vb[tid] = w.rc[tid];
        vb[VL + tid] = w.sq[tid];
        vb[2 * VL + tid] = w.rp[tid];
        vb[3 * VL + tid] = sv.c[tid];
        vb[4 * VL + tid] = sv.t1[tid];
    }
}

#define HESS_WAVE_NVEC 8      // dmu dp rd t2 dc w1 w2 dNmu
__host__ __device__ inline size_t hess_dirs_shared_doubles(int q, int d) {
    // U R L phiS phiD | rc sq rp c t1 mu p utc utmu | xq | tab | dmuT | k1s k2s M11 | Dg
    return (size_t)5 * q * sdp_ld(q) + (size_t)9 * sdp_vlen(q) + (size_t)q * d + hess_tab_doubles(q, d) +
           (size_t)q * d + 3 * (size_t)q * q + (size_t)q * d * d;
}
__host__ __device__ inline size_t hess_dirs_wave_doubles(int q) {
    // at least the 16 x 16 tile of the Dg pre-pass
    const size_t need = (size_t)2 * q * sdp_ld(q) + (size_t)HESS_WAVE_NVEC * sdp_vlen(q);
    return need > 256 ? need : 256;
}

template <int T, int QE>
__global__ __launch_bounds__(T) void k_hess_dirs(KernParams kp, int n, int npad, int q_arg, int RP,
                                                 const double* __restrict__ Xtrain,
                                                 const double* __restrict__ alpha,
                                                 const double* __restrict__ Xq,
                                                 const double* __restrict__ G,
                                                 const double* __restrict__ Wm,
                                                 const double* __restrict__ p_in,
                                                 const double* __restrict__ M11_in,
                                                 const double* __restrict__ state,
                                                 const int* __restrict__ stflag,
                                                 const int* __restrict__ sdp_status,
                                                 double* __restrict__ Hout, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
#ifdef OEI_HESS_TIMING
    const long long t_kernel0 = clock64();
#endif
    constexpr int NW = T / 64;
    const int q = QE > 0 ? QE : q_arg;           // QE > 0: instance for exactly this q
    const int VL = sdp_vlen(q), LD = sdp_ld(q), ML = q * LD, d = kp.d, qd = q * d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t b = blockIdx.x;
    double* Hb = Hout + b * (size_t)qd * qd;
    const int slice = blockIdx.y, nslice = gridDim.y;     // workgroups sharing this candidate's directions
    if (stflag[b]) {
        if (slice == 0) {
            for (size_t e = tid; e < (size_t)qd * qd; e += T) Hb[e] = NAN;
            if (tid == 0 && status) status[b] = stflag[b];
        }
        return;
    }
    lds_f64* base = (lds_f64*)sm;
    lds_f64* U = base;            lds_f64* R = U + ML;        lds_f64* L = R + ML;
    lds_f64* phiS = L + ML;       lds_f64* phiD = phiS + ML;
    lds_f64* rc = phiD + ML;      lds_f64* sq = rc + VL;      lds_f64* rp = sq + VL;
    lds_f64* cv = rp + VL;        lds_f64* t1 = cv + VL;      lds_f64* mu = t1 + VL;
    lds_f64* pv = mu + VL;
    lds_f64* utc = pv + VL;                      // U^T c
    lds_f64* utmu = utc + VL;                    // U^T mu
    lds_f64* xq = utmu + VL;                     // [q*d]
    lds_f64* tab = xq + qd;                      // DS [q*d*q] (then U^T v' of every direction), first used as chunk staging
    lds_f64* dmuT = tab + hess_tab_doubles(q, d);
    lds_f64* k1s = dmuT + qd;
    lds_f64* k2s = k1s + q * q;
    lds_f64* M11 = k2s + q * q;
    lds_f64* Dg = M11 + q * q;                   // [q][d][d]
    lds_f64* wbase = Dg + q * d * d + (size_t)wave * hess_dirs_wave_doubles(q);
    const double* Gb = G + b * (size_t)RP * RP;
    for (int e = tid; e < qd; e += T) xq[e] = Xq[b * qd + e];
    for (int e = tid; e < q * q; e += T) M11[e] = M11_in[b * q * q + e];
    if (tid < q) pv[tid] = p_in[b * q + tid];
    {
        const double* sb = state + b * hess_state_doubles(q);
        for (int e = tid; e < 3 * ML; e += T) U[e] = sb[e];               // U, R, L are adjacent
        for (int e = tid; e < 5 * VL; e += T) rc[e] = sb[3 * ML + e];     // rc sq rp c t1 likewise
    }
    __syncthreads();
#ifdef OEI_HESS_TIMING
    const long long t_pre0 = clock64();
    long long pre_cyc[4] = {0, 0, 0, 0};
#endif
    if (d < 16)       // a point's d + 1 rows fit the 16-row tile
        hess_dg_prepass_mfma<T>(kp, n, npad, q, xq, pv, tab, hess_tab_doubles(q, d), Dg, wbase, hess_dirs_wave_doubles(q), Xtrain, alpha, Wm + b * (size_t)q * npad
#ifdef OEI_HESS_TIMING
                                , pre_cyc
#endif
                                );
    else
        hess_dg_prepass<T>(kp, n, npad, q, xq, pv, tab, hess_tab_doubles(q, d), Dg, Xtrain, alpha, Wm + b * (size_t)q * npad);
#ifdef OEI_HESS_TIMING
    const long long t_pre1 = clock64();
#endif
    build_tables<T>(kp, q, xq, Gb, RP, mu, tab, dmuT, k1s, k2s);
    const lds_f64* DS = tab;
    hess_dg_selfcov<T>(kp, q, xq, (const lds_f64*)M11, k1s, k2s, Dg);
    for (int e = tid; e < q * q; e += T) {
        const int a = e / q, c = e - a * q;
        const double sa = sq[a], sc = sq[c];
        phiS[a * LD + c] = -0.5 * sdp_rcp(sa * sc * (sa + sc));
        phiD[a * LD + c] = 0.5 * sdp_rcp(sa + sc);
    }
    // (the state's third matrix is L^-1, full with zeros above the diagonal: the preparation kernel
    // inverts the factor in registers, and the directions solve with two matrix-vector products)
#ifdef OEI_HESS_TIMING
    const long long t_pre2 = clock64();
#endif
    // Row (b,m) of the table becomes U^T v'_(b,m) (v'_j = dSigma_bj/dx_bm, j != b; half of it at
    // j = b), in place: it is both the direction's own input and what the contraction of every
    // other direction meets, so no direction forms dN = Y U^T.  On the matrix pipe: a wave takes 16
    // rows at a time (left operand: the rows, halved at c = b; right operand: U), reads them whole
    // and only then overwrites them.
    {
        const int li = lane & 15, kg = lane >> 4, ks = (q + 3) >> 2;
        for (int I = wave; 16 * I < qd; I += NW) {
            sens_double4 acc0 = (sens_double4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
            const int bm = 16 * I + li;
            const bool rin = bm < qd;
            const int bmc = rin ? bm : 0, bb = bmc / d;
            const bool j0 = li < q, j1 = 16 + li < q;
            for (int s4 = 0; s4 < ks; ++s4) {
                const int c = 4 * s4 + kg;
                const bool kin = c < q;
                const int cc = kin ? c : 0;
                const double x = tab[(size_t)bmc * q + cc];
                const double av = (rin && kin) ? (cc == bb ? 0.5 * x : x) : 0.0;
                const double u0 = U[cc * LD + (j0 ? li : 0)], u1 = U[cc * LD + (j1 ? 16 + li : 0)];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, (kin && j0) ? u0 : 0.0, acc0, 0, 0, 0);
                if (q > 16) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, (kin && j1) ? u1 : 0.0, acc1, 0, 0, 0);
            }
            sens_sync<true>();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = 16 * I + kg + 4 * g;
                if (row < qd) {
                    if (j0) tab[(size_t)row * q + li] = acc0[g];
                    if (j1) tab[(size_t)row * q + 16 + li] = acc1[g];
                }
            }
        }
    }
    if (tid < q) {
        utc[tid] = lds_dot(U + tid, LD, cv, 1, q);
        utmu[tid] = lds_dot(U + tid, LD, mu, 1, q);
    }
    __syncthreads();

    // ---- directions: wave `wave` of slice s takes al = wave + NW s, then every NW * nslice-th ----
    SdpWork w;
    w.q = q; w.LD = LD;
    w.S = U; w.R = R; w.H = L; w.rc = rc; w.sq = sq; w.rp = rp;
    w.X = w.Kt = wbase; w.V = wbase + ML;     // dN replaces Z: two scratch matrices per wave
    lds_f64* wv0 = wbase + 2 * ML;
    SensVec sv;
    sv.dmu = wv0;            sv.dp = wv0 + VL;      sv.rd = wv0 + 2 * VL;  sv.t2 = wv0 + 3 * VL;
    sv.dc = wv0 + 4 * VL;    sv.w1 = wv0 + 5 * VL;  sv.w2 = wv0 + 6 * VL;
    lds_f64* dNmu = wv0 + 7 * VL;
    sv.c = cv; sv.t1 = t1;
#ifdef OEI_HESS_TIMING
    long long sens_cyc[12] = {};
    const long long t_dirs0 = clock64();
    sens_cyc[11] = t_dirs0 - t_kernel0;
    sens_cyc[8] = t_pre1 - t_kernel0;
    sens_cyc[9] = t_pre2 - t_pre1;
#endif
    for (int al = wave + NW * slice; al < qd; al += NW * nslice) {
        const int a = al / d, l = al - a * d;
        sens_sync<true>();
        // the row's T_al . T_bm entries come from global memory: ask for them before the solve
        double ttpre[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int bm = min(lane + 64 * u, qd - 1), bb = bm / d, m2 = bm - bb * d;
            ttpre[u] = gsym(Gb, RP, a * (d + 1) + 1 + l, bb * (d + 1) + 1 + m2);
        }
        if (lane < q) sv.dmu[lane] = (lane == a) ? dmuT[al] : 0.0;
#ifdef OEI_HESS_TIMING
        sens_direction<64, true, true, QE, true, true>(w, sv, SensDir{a, nullptr, tab + (size_t)al * q, phiS, phiD, utc}, sens_cyc);
        long long sens_t0 = clock64();
#else
        sens_direction<64, true, true, QE, true, true>(w, sv, SensDir{a, nullptr, tab + (size_t)al * q, phiS, phiD, utc});
#endif
        if (lane < q) dNmu[lane] = lds_dot(w.V + lane * LD, 1, utmu, 1, q);      // dN mu = Y (U^T mu)
        sens_sync<true>();
        hess_contract<true>(lane, 64, kp, q, RP, a, l, w.V, LD, DS, dmuT, sv.w1, sv.w2, dNmu, Gb, (const lds_f64*)M11, xq,
                      (const lds_f64*)k1s, (const lds_f64*)k2s, (const lds_f64*)Dg, Hb, ttpre);
        SENS_STAMP(7);
    }
#ifdef OEI_HESS_TIMING
    // diagnostic build: the cycle counts of this wave overwrite the first entries of the output
    sens_cyc[10] = clock64() - t_dirs0;
    __syncthreads();
    if (b == 0 && slice == 0 && tid == 0)
    {
        for (int i = 0; i < 12; ++i) Hb[i] = (double)sens_cyc[i];
        for (int i = 0; i < 4; ++i) Hb[16 + i] = (double)pre_cyc[i];
        Hb[20] = (double)(t_pre0 - t_kernel0);
    }
#endif
    // a solve that stopped at the Newton iteration limit (status 1) still gets its Hessian
    if (slice == 0 && tid == 0 && status) status[b] = sdp_status ? sdp_status[b] : 0;
}

// ------------------------------------------------------------------------------------------------
static inline size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

static int hs_reserve(oei_model* m, size_t bytes) {
    if (bytes > m->hs_bytes) {
        if (m->hs) OEI_HIP_CHECK(hipFree(m->hs));
        m->hs = nullptr; m->hs_bytes = 0;
        OEI_HIP_CHECK(hipMalloc(&m->hs, bytes));
        m->hs_bytes = bytes;
    }
    return OEI_OK;
}

static int check_call(const oei_model* m, int B, int q, const void* X, const void* out, const char* who) {
    if (!m) { oei_set_error("%s: null model", who); return OEI_ERR_ARG; }
    if (B < 0 || q < 1 || q > OEI_MAX_Q) { oei_set_error("%s: B=%d q=%d out of range (1 <= q <= %d)", who, B, q, OEI_MAX_Q); return OEI_ERR_ARG; }
    if (B > 0 && (!X || !out)) { oei_set_error("%s: null pointer", who); return OEI_ERR_ARG; }
    return OEI_OK;
}

// gram_ext -> GEMM -> Gram blocks for candidates [0, Bc): E in ws1, G at m->hs.
static int hess_front(oei_model* m, int Bc, int q, const double* X, int RP, size_t extra_bytes,
                      hipStream_t st) {
    // the blocks are packed densely (RP is a multiple of 16); the GEMM runs on whole 128-row tiles
    // and the last block's partial Gram tiles read up to 64 rows past it: slack rows behind the data
    const size_t rows = round_up_sz((size_t)Bc * RP, 128);
    int rc = oei_ws_reserve(m, rows + 64, ((size_t)Bc * q + (size_t)Bc * q * q) * sizeof(double));
    if (rc) return rc;
    const size_t gbytes = (size_t)Bc * RP * RP * sizeof(double);
    if ((rc = hs_reserve(m, gbytes + extra_bytes))) return rc;
    OEI_SCRATCH_POISON(m, st);
    if ((rc = oei_launch_gram_ext(m, X, Bc, q, RP, m->ws0, st))) return rc;
    if ((rc = oei_launch_gemm_nt(m->ws0, m->Linv, m->ws1, rows, m->npad, 1, st))) return rc;
    return oei_launch_gram_blk(m->ws1, m->hs, Bc, RP, m->npad, st, q * (m->d + 1) + 1);
}

extern "C" int oei_domega_batched(oei_model* m, int B, int q, const double* X, double* domega,
                                  void* stream) {
    int rc = check_call(m, B, q, X, domega, "domega_batched");
    if (rc) return rc;
    if (B == 0) return OEI_OK;
    hipStream_t st = (hipStream_t)stream;
    const int d = m->d, qd = q * d, k_ = q + 1;
    const int RP = (int)round_up_sz((size_t)q * (d + 1) + 1, 16);
    const size_t lds = ((size_t)2 * qd + q + (size_t)q * q) * sizeof(double);
    const int chunk = 512;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int Bc = (B - b0 < chunk) ? B - b0 : chunk;
        const double* Xc = X + (size_t)b0 * qd;
        if ((rc = hess_front(m, Bc, q, Xc, RP, 0, st))) return rc;
        double* out = domega + (size_t)b0 * k_ * k_ * qd;
        if (q <= 32) {
            if ((rc = set_lds_attr(k_domega<64>, lds))) return rc;
            hipLaunchKernelGGL(k_domega<64>, dim3(Bc), dim3(64), lds, st, m->kp, q, RP, Xc, m->hs, out);
        } else {
            if ((rc = set_lds_attr(k_domega<256>, lds))) return rc;
            hipLaunchKernelGGL(k_domega<256>, dim3(Bc), dim3(256), lds, st, m->kp, q, RP, Xc, m->hs, out);
        }
        OEI_HIP_CHECK(hipGetLastError());
    }
    return OEI_OK;
}

extern "C" int oei_hessian_batched(oei_model* m, int B, int q, const double* X, double* H,
                                   int* status, void* stream) {
    int rc = check_call(m, B, q, X, H, "hessian_batched");
    if (rc) return rc;
    if (B == 0) return OEI_OK;
    hipStream_t st = (hipStream_t)stream;
    const int d = m->d, qd = q * d, npad = m->npad;
    const int RP = (int)round_up_sz((size_t)q * (d + 1) + 1, 16);
    const size_t lds = hess_wide_lds_doubles(q, d) * sizeof(double);     // wide kernel only
    // q <= 32 runs on the register-path kernels while a candidate's tables fit in LDS beside one
    // wave's scratch matrices; past that (q near 32 with d >= 9) the wide kernel, which keeps the
    // tables in global memory, takes over
    const bool wide = q > 32 ||
        (hess_dirs_shared_doubles(q, d) + hess_dirs_wave_doubles(q)) * sizeof(double) > (size_t)160 * 1024;
    // candidates per pass: bounded by the per-candidate workspace (E rows and Gram block, about
    // 2 * RP * (npad + RP / 2) doubles), 2048 when that stays below ~4 GB
    int chunk = 2048;
    const double per_cand = (double)RP * (2.0 * npad + RP) + (wide ? (double)hess_wide_table_doubles(q, d) : 0.0);
    while (chunk > 64 && (double)chunk * per_cand * sizeof(double) > 4e9) chunk /= 2;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int Bc = (B - b0 < chunk) ? B - b0 : chunk;
        const double* Xc = X + (size_t)b0 * qd;
        const size_t zrows = round_up_sz((size_t)Bc * q, 64);
        const size_t zbytes = zrows * npad * sizeof(double);
        const size_t gbytes = (size_t)Bc * RP * RP * sizeof(double);
        // small batches: several workgroups per candidate share its q d directions (the chip has
        // 256 CUs and these kernels run one workgroup per CU)
        // (the wide kernel fits two per CU when its LDS allows)
        const int slots = (wide && 2 * lds <= (size_t)160 * 1024) ? 512 : 256;
        int nslice = (2 * Bc <= slots) ? slots / Bc : 1;
        if (nslice > 8) nslice = 8;
        const size_t sbytes = !wide ? (size_t)Bc * hess_state_doubles(q) * sizeof(double) + (size_t)Bc * sizeof(int) + 16
                                    : (size_t)Bc * nslice * hess_wide_table_doubles(q, d) * sizeof(double) + 16;
        if ((rc = hess_front(m, Bc, q, Xc, RP, 2 * zbytes + Bc * sizeof(double) + ((Bc + 1) & ~1) * sizeof(int) + sbytes, st))) return rc;
        double* Zb = m->hs + gbytes / sizeof(double);
        double* Wb = Zb + zrows * npad;
        double* valb = Wb + zrows * npad;
        int* stb = reinterpret_cast<int*>(valb + Bc);
        double* p_buf = m->aux;
        double* M11_buf = m->aux + (size_t)Bc * q;
        OEI_HIP_CHECK(hipMemsetAsync(Zb, 0, zbytes, st));
        // A rows of candidate b: E + b*RP*npad + a*(d+1)*npad
        if ((rc = oei_launch_stage2(m, Bc, q, Xc, m->ws1, (size_t)RP * npad, (size_t)(d + 1) * npad,
                                    valb, stb, p_buf, M11_buf, Zb, st))) return rc;
        if ((rc = oei_launch_gemm_nt(Zb, m->LinvT, Wb, zrows, npad, 0, st))) return rc;
        double* Hc = H + (size_t)b0 * qd * qd;
        int* sc = status ? status + b0 : nullptr;
        if (!wide) {
            // state + flags behind the buffers already carved from m->hs
            double* state = reinterpret_cast<double*>(stb + ((Bc + 1) & ~1));
            int* pflag = reinterpret_cast<int*>(state + (size_t)Bc * hess_state_doubles(q));
            // the solver instance of the candidate's size, as in the value + gradient path
#define LAUNCH_PREP(T_, Q_) do { \
                const size_t lds_prep = (sdp_work_doubles(q, true, sdp_mm<Q_>()) + (size_t)(1 + SENS_NVEC) * sdp_vlen(q) + (size_t)qd) * sizeof(double); \
                if ((rc = set_lds_attr((k_hess_prep<64, (Q_) ? (Q_) : 32>), lds_prep))) return rc; \
                hipLaunchKernelGGL((k_hess_prep<64, (Q_) ? (Q_) : 32>), dim3(Bc), dim3(64), lds_prep, st, m->kp, m->noise, m->fmin, q, RP, Xc, m->hs, p_buf, state, pflag); } while (0)
            SDP_DISPATCH_Q(q, LAUNCH_PREP);
#undef LAUNCH_PREP
            OEI_HIP_CHECK(hipGetLastError());
            const size_t shared = hess_dirs_shared_doubles(q, d) * sizeof(double), per_wave = hess_dirs_wave_doubles(q) * sizeof(double);
            const size_t budget = 160 * 1024;
            if (shared + per_wave > budget) { oei_set_error("hessian_batched: q=%d d=%d needs %zu bytes of LDS (limit %zu)", q, d, shared + per_wave, budget); return OEI_ERR_ARG; }
            int nw = (int)((budget - shared) / per_wave);
            if (nw > qd) nw = qd;
#define LAUNCH_DIRS(TT, QE) do { \
                const size_t lds_d = shared + (size_t)((TT) / 64) * per_wave; \
                if ((rc = set_lds_attr((k_hess_dirs<TT, QE>), lds_d))) return rc; \
                int ns = nslice, rounds = (qd + (TT) / 64 - 1) / ((TT) / 64); \
                if (ns > rounds) ns = rounds; \
                hipLaunchKernelGGL((k_hess_dirs<TT, QE>), dim3(Bc, ns), dim3(TT), lds_d, st, m->kp, m->n, npad, q, RP, m->X, m->alpha, Xc, m->hs, Wb, p_buf, M11_buf, state, pflag, stb, Hc, sc); } while (0)
            // (an instance specialised for q = 20 was measured slower: its straight-line code
            // spills 160 registers per lane, and the spill traffic costs more than the constants save)
            if (nw >= 12 && qd >= 96) LAUNCH_DIRS(768, 0);
            else if (nw >= 8) LAUNCH_DIRS(512, 0); else if (nw >= 4) LAUNCH_DIRS(256, 0);
            else if (nw >= 2) LAUNCH_DIRS(128, 0); else LAUNCH_DIRS(64, 0);
#undef LAUNCH_DIRS
        } else {
            double* tabs = reinterpret_cast<double*>(stb + ((Bc + 1) & ~1));
            if (sc) OEI_HIP_CHECK(hipMemsetAsync(sc, 0, (size_t)Bc * sizeof(int), st));     // the slices raise it
            if ((rc = set_lds_attr((k_hess_wide<256>), lds))) return rc;
            hipLaunchKernelGGL((k_hess_wide<256>), dim3(Bc, nslice < qd ? nslice : qd), dim3(256), lds, st, m->kp, m->noise, m->fmin, m->n, npad, q, RP, m->X, m->alpha, Xc, m->hs, Wb, p_buf, M11_buf, stb, tabs, Hc, sc);
        }
        OEI_HIP_CHECK(hipGetLastError());
    }
    return OEI_OK;
}
