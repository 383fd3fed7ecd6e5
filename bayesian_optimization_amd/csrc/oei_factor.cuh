// Blocked Cholesky factorisation, triangular inverse and the matrix-vector products of the GP
// model, shared by the model handle (oei_model.hip: one matrix) and the batched hyper-parameter
// objective (oei_fit.hip: one matrix per trial parameter vector).  Every kernel takes the batch
// item from blockIdx.z: matrices are npad x npad apart, the diagonal-block inverses and the
// scratch tiles npad x 64, vectors npad, flags 1; a single matrix is a launch with gridDim.z = 1.
#pragma once
#include "oei_common.h"

#define TB 64          // block size of the blocked factorisation
#define TLD 65         // LDS leading dimension (odd: conflict-free column walks)

// ---- K = k(X,X) + noise I, identity in the padding ---------------------------------------------
static __global__ void k_gram_train(KernParams kp, const double* __restrict__ X, int n, int npad,
                             double noise, double* __restrict__ K) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= npad) return;
    K += (size_t)blockIdx.z * npad * npad;
    double val;
    if (i < n && j < n) {
        double t = 0.0;
        for (int l = 0; l < kp.d; ++l) {
            double df = X[(size_t)i * kp.d + l] - X[(size_t)j * kp.d + l];
            t += df * df * kp.inv_ls2[l];
        }
        double k0, k1, k2;
        kappa_eval<0>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        val = kp.variance * k0 + (i == j ? noise : 0.0);
    } else {
        val = (i == j) ? 1.0 : 0.0;
    }
    K[(size_t)i * npad + j] = val;
}

// ---- 64x64 tile product helper: acc(4x4 per thread) += A(64x64) * op(B)(64x64) -----------------
// thread (ty, tx) = (tid / 16, tid % 16) owns rows 4*ty.., cols 4*tx..
template <bool TRANSB>
static __device__ __forceinline__ void tile_mm(double acc[4][4], const double* __restrict__ A, int lda,
                                        const double* __restrict__ Bt, int ldb, double* As,
                                        double* Bs) {
    const int tid = threadIdx.x;
    __syncthreads();
    for (int e = tid; e < TB * TB; e += 256) {
        int r = e / TB, c = e % TB;
        As[r * TLD + c] = A[(size_t)r * lda + c];
        // Bs[k][j] = op(B)[k][j]
        Bs[r * TLD + c] = TRANSB ? Bt[(size_t)c * ldb + r] : Bt[(size_t)r * ldb + c];
    }
    __syncthreads();
    const int ty = tid / 16, tx = tid % 16;
    for (int k = 0; k < TB; ++k) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = As[(4 * ty + u) * TLD + k];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = Bs[k * TLD + 4 * tx + u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) acc[u][w] += a[u] * b[w];
    }
}

// ---- diagonal block: Cholesky of A_kk and the inverse of its factor ------------------------------
// One wave.  Lane i keeps row i of the block in registers; pivots and multipliers travel through
// the scalar unit (v_readlane), so the 64 elimination steps need no barrier (a 256-thread version
// with the block in LDS spent its time in 192 barriers: 165 us per block against about 40 us).
// The inverse is built column by column from the factor in LDS: lane j, forward substitution.
static __device__ __forceinline__ double potf2_bcast(double v, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

static __global__ __launch_bounds__(64) void k_potf2(double* __restrict__ Lm, int npad, int kb,
                                              double* __restrict__ Dinv, int* __restrict__ flag) {
    __shared__ double T[TB * TLD];
    __shared__ double Ti[TB * TLD];
    const int lane = threadIdx.x;
    Lm += (size_t)blockIdx.z * npad * npad;
    Dinv += (size_t)blockIdx.z * npad * TB;
    flag += blockIdx.z;
    double* Ablk = Lm + ((size_t)kb * TB) * npad + (size_t)kb * TB;
    for (int e = lane; e < TB * TB; e += 64) T[(e / TB) * TLD + e % TB] = Ablk[(size_t)(e / TB) * npad + e % TB];
    __syncthreads();
    double a[TB];
#pragma unroll
    for (int k = 0; k < TB; ++k) a[k] = (k <= lane) ? T[lane * TLD + k] : 0.0;
    int bad = 0;
#pragma unroll
    for (int k = 0; k < TB; ++k) {
        double dkk = potf2_bcast(a[k], k);
        if (!(dkk > 0.0)) { bad = 1; dkk = 1.0; }              // identical in every lane
        const double piv = sqrt(dkk);
        const double lik = (lane > k) ? a[k] / piv : (lane == k ? piv : 0.0);
        a[k] = lik;
#pragma unroll
        for (int jj = k + 1; jj < TB; ++jj) {
            const double ljk = potf2_bcast(lik, jj);           // L[jj][k]
            if (lane >= jj) a[jj] -= lik * ljk;
        }
    }
    if (bad && lane == 0) *flag = 1;
#pragma unroll
    for (int k = 0; k < TB; ++k) T[lane * TLD + k] = a[k];     // zeros above the diagonal
    __syncthreads();
    {   // inverse of the lower-triangular block: lane j builds column j by forward substitution
        const int j = lane;
        for (int i = 0; i < TB; ++i) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = j; k < i; ++k) s -= T[i * TLD + k] * Ti[k * TLD + j];
            Ti[i * TLD + j] = (i < j) ? 0.0 : s / T[i * TLD + i];
        }
    }
    __syncthreads();
    for (int e = lane; e < TB * TB; e += 64) {
        int r = e / TB, c = e % TB;
        Ablk[(size_t)r * npad + c] = T[r * TLD + c];
        Dinv[(size_t)kb * TB * TB + e] = Ti[r * TLD + c];
    }
}

// ---- panel: L_ik = A_ik * inv(L_kk)^T for i > k -------------------------------------------------
static __global__ __launch_bounds__(256) void k_trsm_panel(double* __restrict__ Lm, int npad, int kb,
                                                    const double* __restrict__ Dinv) {
    __shared__ double As[TB * TLD];
    __shared__ double Bs[TB * TLD];
    const int ib = kb + 1 + blockIdx.x;
    Lm += (size_t)blockIdx.z * npad * npad;
    Dinv += (size_t)blockIdx.z * npad * TB;
    double* Ablk = Lm + ((size_t)ib * TB) * npad + (size_t)kb * TB;
    double acc[4][4] = {};
    tile_mm<true>(acc, Ablk, npad, Dinv + (size_t)kb * TB * TB, TB, As, Bs);
    __syncthreads();
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    for (int u = 0; u < 4; ++u)
        for (int w = 0; w < 4; ++w) Ablk[(size_t)(4 * ty + u) * npad + 4 * tx + w] = acc[u][w];
}

// ---- trailing update: A_ij -= L_ik L_jk^T for i >= j > k ---------------------------------------
static __global__ __launch_bounds__(256) void k_syrk_trail(double* __restrict__ Lm, int npad, int kb) {
    __shared__ double As[TB * TLD];
    __shared__ double Bs[TB * TLD];
    const int ib = kb + 1 + blockIdx.y;
    const int jb = kb + 1 + blockIdx.x;
    if (jb > ib) return;
    Lm += (size_t)blockIdx.z * npad * npad;
    const double* Lik = Lm + ((size_t)ib * TB) * npad + (size_t)kb * TB;
    const double* Ljk = Lm + ((size_t)jb * TB) * npad + (size_t)kb * TB;
    double acc[4][4] = {};
    tile_mm<true>(acc, Lik, npad, Ljk, npad, As, Bs);
    double* Aij = Lm + ((size_t)ib * TB) * npad + (size_t)jb * TB;
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    for (int u = 0; u < 4; ++u)
        for (int w = 0; w < 4; ++w) Aij[(size_t)(4 * ty + u) * npad + 4 * tx + w] -= acc[u][w];
}

// ---- zero the strict upper triangle of block row 0.. (blocks above the diagonal) ---------------
static __global__ void k_zero_upper(double* __restrict__ Lm, int npad) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    Lm += (size_t)blockIdx.z * npad * npad;
    if (j < npad && j > i) Lm[(size_t)i * npad + j] = 0.0;
}

// ---- L^-1: block column jb is built top-down by one workgroup ----------------------------------
// Linv_jj = inv(L_jj);  Linv_ij = -inv(L_ii) * sum_{k=j}^{i-1} L_ik Linv_kj   (i > j)
static __global__ __launch_bounds__(256) void k_trinv(const double* __restrict__ Lm, int npad,
                                               const double* __restrict__ Dinv,
                                               double* __restrict__ Linv, double* __restrict__ tmp) {
    __shared__ double As[TB * TLD];
    __shared__ double Bs[TB * TLD];
    const int nb = npad / TB;
    const int jb = blockIdx.x;
    const int tid = threadIdx.x;
    const int ty = tid / 16, tx = tid % 16;
    Lm += (size_t)blockIdx.z * npad * npad;
    Linv += (size_t)blockIdx.z * npad * npad;
    Dinv += (size_t)blockIdx.z * npad * TB;
    tmp += (size_t)blockIdx.z * npad * TB;
    // zero blocks above the diagonal, copy the diagonal inverse
    for (int ib = 0; ib < jb; ++ib)
        for (int e = tid; e < TB * TB; e += 256)
            Linv[((size_t)ib * TB + e / TB) * npad + (size_t)jb * TB + e % TB] = 0.0;
    for (int e = tid; e < TB * TB; e += 256)
        Linv[((size_t)jb * TB + e / TB) * npad + (size_t)jb * TB + e % TB] = Dinv[(size_t)jb * TB * TB + e];
    __threadfence_block();
    __syncthreads();
    double* T = tmp + (size_t)jb * TB * TB;     // per-workgroup scratch tile
    for (int ib = jb + 1; ib < nb; ++ib) {
        double acc[4][4] = {};
        for (int kb = jb; kb < ib; ++kb)
            tile_mm<false>(acc, Lm + ((size_t)ib * TB) * npad + (size_t)kb * TB, npad,
                           Linv + ((size_t)kb * TB) * npad + (size_t)jb * TB, npad, As, Bs);
        __syncthreads();
        for (int u = 0; u < 4; ++u)
            for (int w = 0; w < 4; ++w) T[(4 * ty + u) * TB + 4 * tx + w] = acc[u][w];
        __threadfence_block();
        __syncthreads();
        double acc2[4][4] = {};
        tile_mm<false>(acc2, Dinv + (size_t)ib * TB * TB, TB, T, TB, As, Bs);
        for (int u = 0; u < 4; ++u)
            for (int w = 0; w < 4; ++w)
                Linv[((size_t)ib * TB + 4 * ty + u) * npad + (size_t)jb * TB + 4 * tx + w] = -acc2[u][w];
        __threadfence_block();
        __syncthreads();
    }
}

static __global__ void k_transpose(const double* __restrict__ A, double* __restrict__ At, int npad) {
    __shared__ double t[32][33];
    A += (size_t)blockIdx.z * npad * npad;
    At += (size_t)blockIdx.z * npad * npad;
    int x = blockIdx.x * 32 + threadIdx.x, y = blockIdx.y * 32 + threadIdx.y;
    t[threadIdx.y][threadIdx.x] = A[(size_t)y * npad + x];
    __syncthreads();
    x = blockIdx.y * 32 + threadIdx.x; y = blockIdx.x * 32 + threadIdx.y;
    At[(size_t)y * npad + x] = t[threadIdx.x][threadIdx.y];
}

// out[i] = sum_j Mat[i][j] * vec[j]   (one wave per row)
// (batch item z: matrix z, out + z npad, vec + z vec_stride)
static __global__ void k_matvec(const double* __restrict__ Mat, const double* __restrict__ vec,
                         double* __restrict__ out, int npad, int vec_stride) {
    const int i = blockIdx.x;
    Mat += (size_t)blockIdx.z * npad * npad;
    vec += (size_t)blockIdx.z * vec_stride;
    out += (size_t)blockIdx.z * npad;
    double s = 0.0;
    for (int j = threadIdx.x; j < npad; j += 64) s += Mat[(size_t)i * npad + j] * vec[j];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[i] = s;
}

