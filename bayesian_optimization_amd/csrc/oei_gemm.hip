// Cross-covariance (kernel Gram) generation and the fp64 MFMA GEMM that applies the shared
// triangular factor inverse to many candidate batches at once.
//
//   Ks[r, k] = k(x_r, X_k)                      gram kernel (VALU: sqrt/exp per entry)
//   C[r, j]  = sum_k A[r, k] * Bm[j, k]         "NT" GEMM on v_mfma_f64_16x16x4_f64
//
// With Bm = L^-1 (lower) the GEMM gives A = Ks L^-T, i.e. the rows of (L^-1 k(X, x_r))^T that
// gpflow obtains with a triangular solve per call (oei.py:86 -> GPR.build_predict).  With
// Bm = L^-T (upper, stored row-major) it gives Z L^-1, the second solve of K^-1.  The K range of
// each 64-column tile stops (starts) at the diagonal, so only the non-zero half is multiplied.
#include "oei_common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

#define GM_BM 64
#define GM_BN 64
#define GM_BK 16
#define GM_LDS (GM_BK + 1)

// ---------------------------------------------------------------------------------------------
// Gram: block = 256 threads = 64 training columns x 4 row lanes; each block covers 64 columns and
// GR_ROWS query rows.  Training tile and query tile are staged in LDS so every global word is read
// once per block; stores are coalesced along k.
#define GR_ROWS 32

template <int DERIV>
__global__ __launch_bounds__(256) void k_gram(KernParams kp, const double* __restrict__ Xtrain,
                                              int n, int npad, const double* __restrict__ Xq,
                                              size_t rows, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int d = kp.d;
    double* Xs = sm;                 // [64][d]
    double* Qs = sm + 64 * d;        // [GR_ROWS][d]
    const int tid = threadIdx.x;
    const int k0 = blockIdx.y * 64;
    const size_t r0 = (size_t)blockIdx.x * GR_ROWS;
    for (int i = tid; i < 64 * d; i += 256) {
        int k = k0 + i / d;
        Xs[i] = (k < n) ? Xtrain[(size_t)k * d + (i % d)] : 0.0;
    }
    for (int i = tid; i < GR_ROWS * d; i += 256) {
        size_t r = r0 + i / d;
        Qs[i] = (r < rows) ? Xq[r * d + (i % d)] : 0.0;
    }
    __syncthreads();
    const int kk = tid & 63;
    const int k = k0 + kk;
    for (int rr = tid >> 6; rr < GR_ROWS; rr += 4) {
        const size_t r = r0 + rr;
        double t = 0.0;
        for (int l = 0; l < d; ++l) {
            double df = Qs[rr * d + l] - Xs[kk * d + l];
            t += df * df * kp.inv_ls2[l];
        }
        double k0v, k1v, k2v;
        const bool live = (r < rows) && (k < n);
        if (DERIV == 0) {
            kappa_eval<0>(kp.kind, kp.floor_mode, t, k0v, k1v, k2v);
            out[r * npad + k] = live ? kp.variance * k0v : 0.0;
        } else {
            kappa_eval<1>(kp.kind, kp.floor_mode, t, k0v, k1v, k2v);
            const double c = live ? 2.0 * kp.variance * k1v : 0.0;
            for (int l = 0; l < d; ++l) {
                double df = Qs[rr * d + l] - Xs[kk * d + l];
                out[(r * d + l) * npad + k] = c * df * kp.inv_ls2[l];
            }
        }
    }
}

int oei_launch_gram(const oei_model* m, const double* Xq, size_t rows, size_t rows_pad,
                    double* Ks, hipStream_t st) {
    dim3 grid((unsigned)(rows_pad / GR_ROWS), m->npad / 64);
    size_t lds = (size_t)(64 + GR_ROWS) * m->d * sizeof(double);
    hipLaunchKernelGGL(k_gram<0>, grid, dim3(256), lds, st, m->kp, m->X, m->n, m->npad, Xq, rows, Ks);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

int oei_launch_gram_deriv(const oei_model* m, const double* Xq, size_t rows, size_t rows_pad,
                          double* Kd, hipStream_t st) {
    dim3 grid((unsigned)(rows_pad / GR_ROWS), m->npad / 64);
    size_t lds = (size_t)(64 + GR_ROWS) * m->d * sizeof(double);
    hipLaunchKernelGGL(k_gram<1>, grid, dim3(256), lds, st, m->kp, m->X, m->n, m->npad, Xq, rows, Kd);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// ---------------------------------------------------------------------------------------------
// GEMM.  Block tile 64 x 64, K step 16, 4 waves; wave w owns rows [16w, 16w+16) x 64 columns as
// four 16x16 accumulators.  Operand lane maps of v_mfma_f64_16x16x4_f64: lane l supplies
// A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; accumulator register g of lane l is
// D[row = (l >> 4) + 4 g][col = l & 15].
__global__ __launch_bounds__(256) void k_gemm_nt(const double* __restrict__ A,
                                                 const double* __restrict__ Bm,
                                                 double* __restrict__ C, int npad, int lower) {
    __shared__ __attribute__((aligned(16))) double As[2][GM_BM * GM_LDS];
    __shared__ __attribute__((aligned(16))) double Bs[2][GM_BN * GM_LDS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j0 = blockIdx.y * GM_BN;
    const size_t r0 = (size_t)blockIdx.x * GM_BM;
    const int kbeg = lower ? 0 : j0;
    const int kend = lower ? (j0 + GM_BN) : npad;

    // staging map: thread t moves 4 consecutive doubles of row t/4
    const int srow = tid >> 2;
    const int scol = (tid & 3) * 4;
    const double* Ag = A + (r0 + srow) * (size_t)npad + scol;
    const double* Bg = Bm + (size_t)(j0 + srow) * npad + scol;

    double4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};

    double2 a0, a1, b0, b1;
    a0 = *reinterpret_cast<const double2*>(Ag + kbeg);
    a1 = *reinterpret_cast<const double2*>(Ag + kbeg + 2);
    b0 = *reinterpret_cast<const double2*>(Bg + kbeg);
    b1 = *reinterpret_cast<const double2*>(Bg + kbeg + 2);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += GM_BK) {
        double* as = As[buf];
        double* bs = Bs[buf];
        as[srow * GM_LDS + scol + 0] = a0.x; as[srow * GM_LDS + scol + 1] = a0.y;
        as[srow * GM_LDS + scol + 2] = a1.x; as[srow * GM_LDS + scol + 3] = a1.y;
        bs[srow * GM_LDS + scol + 0] = b0.x; bs[srow * GM_LDS + scol + 1] = b0.y;
        bs[srow * GM_LDS + scol + 2] = b1.x; bs[srow * GM_LDS + scol + 3] = b1.y;
        __syncthreads();
        if (k0 + GM_BK < kend) {   // prefetch the next K step while this one is multiplied
            a0 = *reinterpret_cast<const double2*>(Ag + k0 + GM_BK);
            a1 = *reinterpret_cast<const double2*>(Ag + k0 + GM_BK + 2);
            b0 = *reinterpret_cast<const double2*>(Bg + k0 + GM_BK);
            b1 = *reinterpret_cast<const double2*>(Bg + k0 + GM_BK + 2);
        }
        const int li = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < GM_BK / 4; ++ks) {
            const double av = as[(wave * 16 + li) * GM_LDS + ks * 4 + lk];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double bv = bs[(t * 16 + li) * GM_LDS + ks * 4 + lk];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
            }
        }
        buf ^= 1;   // two LDS buffers: the next step's stores cannot race this step's reads
    }
    const int col = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const size_t row = r0 + wave * 16 + rq + 4 * g;
            C[row * npad + j0 + t * 16 + col] = acc[t][g];
        }
    }
}

int oei_launch_gemm_nt(const double* A, const double* Bm, double* C, size_t rows_pad, int npad,
                       int lower, hipStream_t st) {
    if (rows_pad % GM_BM != 0 || npad % GM_BN != 0) {
        oei_set_error("gemm_nt: rows_pad=%zu / npad=%d not tile multiples", rows_pad, npad);
        return OEI_ERR_ARG;
    }
    dim3 grid((unsigned)(rows_pad / GM_BM), npad / GM_BN);
    hipLaunchKernelGGL(k_gemm_nt, grid, dim3(256), 0, st, A, Bm, C, npad, lower);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}
