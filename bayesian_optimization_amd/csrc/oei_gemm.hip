// Cross-covariance (kernel Gram) generation and the fp64 MFMA GEMMs that apply the shared
// triangular factor inverse to many candidate batches at once.
//
//   Ks[r, k] = k(x_r, X_k)                      gram kernels (VALU: sqrt/exp per entry)
//   C[r, j]  = sum_k A[r, k] * Bm[j, k]         "NT" GEMM on v_mfma_f64_16x16x4_f64
//
// With Bm = L^-1 (lower) the GEMM gives A = Ks L^-T, i.e. the rows of (L^-1 k(X, x_r))^T that
// gpflow obtains with a triangular solve per call (oei.py:86 -> GPR.build_predict).  With
// Bm = L^-T (upper, stored row-major) it gives Z L^-1, the second solve of K^-1.  The K range of
// each 64-column tile stops (starts) at the diagonal, so only the non-zero half is multiplied.
// The same tile routine, batched over candidates with both operands taken from one row block,
// forms the Gram matrices E_b E_b^T of the Hessian path.
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

// DM > 0: instance for d <= DM.  The thread's training point sits in registers for its GR_ROWS / 4
// rows, the query rows are staged with a stride of DM (zero-padded, with zero weights behind d),
// the loop over the coordinates is unrolled, and the per-coordinate closeness test of the
// duplicate flag only runs behind its necessary condition t < d 1e-4.  The kernel is bound by
// these vector instructions (with its stores removed it runs 0.110 instead of 0.124 ms).
template <bool WITH_K1, int DM>
__global__ __launch_bounds__(256) void k_gram(KernParams kp, const double* __restrict__ Xtrain,
                                              int n, int npad, const double* __restrict__ Xq,
                                              size_t rows, double* __restrict__ out,
                                              double* __restrict__ out1, int q,
                                              int* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    const int d = kp.d;
    double* Xs = sm;                 // [64][d]
    double* Qs = sm + 64 * d;        // [GR_ROWS][d]  (DM > 0: [GR_ROWS][DM])
    const int tid = threadIdx.x;
    const int k0 = blockIdx.y * 64;
    const size_t r0 = (size_t)blockIdx.x * GR_ROWS;
    if constexpr (DM > 0) {
        for (int i = tid; i < 64 * d; i += 256) {
            int k = k0 + i / d;
            Xs[i] = (k < n) ? Xtrain[(size_t)k * d + (i % d)] : 0.0;
        }
        for (int i = tid; i < GR_ROWS * DM; i += 256) {
            const size_t r = r0 + i / DM;
            const int l = i % DM;
            Qs[i] = (r < rows && l < d) ? Xq[r * d + l] : 0.0;
        }
        __syncthreads();
        const int kk = tid & 63;
        const int k = k0 + kk;
        double xk[DM], w2[DM];
#pragma unroll
        for (int l = 0; l < DM; ++l) {
            xk[l] = (l < d) ? Xs[kk * d + l] : 0.0;
            w2[l] = (l < d) ? kp.inv_ls2[l] : 0.0;
        }
        const double tclose = 1e-4 * d;
        for (int rr = tid >> 6; rr < GR_ROWS; rr += 4) {
            const size_t r = r0 + rr;
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int l = 0; l < DM; l += 2) {
                const double d0 = Qs[rr * DM + l] - xk[l];
                t0 = fma(d0 * d0, w2[l], t0);
                if (l + 1 < DM) {
                    const double d1 = Qs[rr * DM + l + 1] - xk[l + 1];
                    t1 = fma(d1 * d1, w2[l + 1], t1);
                }
            }
            const double t = t0 + t1;
            double k0v, k1v, k2v;
            kappa_eval<WITH_K1 ? 1 : 0>(kp.kind, kp.floor_mode, t, k0v, k1v, k2v);
            const bool in = (r < rows) && (k < n);
            out[r * npad + k] = in ? kp.variance * k0v : 0.0;
            if (WITH_K1) {
                out1[r * npad + k] = in ? 2.0 * kp.variance * k1v : 0.0;
                if (in && flags && t < tclose) {             // every |df_l| / ls_l < 1e-2 implies t < d 1e-4
                    bool close = true;
                    for (int l = 0; l < d; ++l)
                        close = close && (fabs((Qs[rr * DM + l] - xk[l]) * kp.inv_ls[l]) < 1e-2);
                    if (close) flags[r / q] = 1;             // every writer stores the same value
                }
            }
        }
        return;
    }
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
        bool close = true;
        for (int l = 0; l < d; ++l) {
            double df = Qs[rr * d + l] - Xs[kk * d + l];
            t += df * df * kp.inv_ls2[l];
            if (WITH_K1) close = close && (fabs(df * kp.inv_ls[l]) < 1e-2);
        }
        double k0v, k1v, k2v;
        kappa_eval<WITH_K1 ? 1 : 0>(kp.kind, kp.floor_mode, t, k0v, k1v, k2v);
        const bool in = (r < rows) && (k < n);
        out[r * npad + k] = in ? kp.variance * k0v : 0.0;
        if (WITH_K1) {
            out1[r * npad + k] = in ? 2.0 * kp.variance * k1v : 0.0;
            if (in && close && flags) flags[r / q] = 1;      // every writer stores the same value
        }
    }
}

int oei_launch_gram(const oei_model* m, const double* Xq, size_t rows, size_t rows_pad,
                    double* Ks, double* K1, int q, int* flags, hipStream_t st) {
    dim3 grid((unsigned)(rows_pad / GR_ROWS), m->npad / 64);
    if (m->d <= 8) {
        const size_t lds = ((size_t)64 * m->d + (size_t)GR_ROWS * 8) * sizeof(double);
        if (K1)
            hipLaunchKernelGGL((k_gram<true, 8>), grid, dim3(256), lds, st, m->kp, m->X, m->n, m->npad, Xq, rows, Ks, K1, q, flags);
        else
            hipLaunchKernelGGL((k_gram<false, 8>), grid, dim3(256), lds, st, m->kp, m->X, m->n, m->npad, Xq, rows, Ks, (double*)nullptr, 1, (int*)nullptr);
    } else {
        const size_t lds = (size_t)(64 + GR_ROWS) * m->d * sizeof(double);
        if (K1)
            hipLaunchKernelGGL((k_gram<true, 0>), grid, dim3(256), lds, st, m->kp, m->X, m->n, m->npad, Xq, rows, Ks, K1, q, flags);
        else
            hipLaunchKernelGGL((k_gram<false, 0>), grid, dim3(256), lds, st, m->kp, m->X, m->n, m->npad, Xq, rows, Ks, (double*)nullptr, 1, (int*)nullptr);
    }
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// Extended rows for the Hessian path.  Candidate b owns rows [b*RP, (b+1)*RP) (RP a multiple of 16;
// the blocks are packed densely, so a workgroup takes a group of up to GR_ROWS rows of ONE block):
//   r = a*(d+1)       -> k(x_a, X_k)
//   r = a*(d+1)+1+l   -> d k(x_a, X_k) / d x_a[l]
//   r = q*(d+1)       -> y_k            (becomes v = L^-1 y after the GEMM)
//   larger r          -> 0
__global__ __launch_bounds__(256) void k_gram_ext(KernParams kp, const double* __restrict__ Xtrain,
                                                  const double* __restrict__ y, int n, int npad,
                                                  const double* __restrict__ Xq, int q, int RP,
                                                  double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    const int d = kp.d, dp = d + 1;
    double* Xs = sm;                 // [64][d]
    double* Qs = sm + 64 * d;        // [GR_ROWS][d]
    const int tid = threadIdx.x;
    const int k0 = blockIdx.y * 64;
    const int ngrp = (RP + GR_ROWS - 1) / GR_ROWS;
    const size_t b = blockIdx.x / ngrp;
    const int rl0 = (int)(blockIdx.x - b * ngrp) * GR_ROWS;
    const int rlim = min(rl0 + GR_ROWS, RP);              // rows [rl0, rlim) of block b
    for (int i = tid; i < 64 * d; i += 256) {
        int k = k0 + i / d;
        Xs[i] = (k < n) ? Xtrain[(size_t)k * d + (i % d)] : 0.0;
    }
    for (int i = tid; i < GR_ROWS * d; i += 256) {
        const int rl = rl0 + i / d, a = rl / dp;
        Qs[i] = (a < q) ? Xq[(b * q + a) * d + (i % d)] : 0.0;
    }
    __syncthreads();
    const int kk = tid & 63;
    const int k = k0 + kk;
    // One kernel evaluation per (point, training point) serves the point's value row and its d
    // derivative rows (they share t and the exponential); a point whose rows straddle two row
    // groups is evaluated by both.  Wave w takes every fourth point of the group.
    const int a_lo = rl0 / dp, a_hi = (rlim - 1) / dp;
    for (int a = a_lo + (tid >> 6); a <= a_hi; a += 4) {
        const int rbeg = max(a * dp, rl0), rend = min(a * dp + dp, rlim);     // rows [rbeg, rend)
        if (a < q && k < n) {
            const double* xa = Qs + (rbeg - rl0) * d;    // every row of the group carries its point
            double t = 0.0;
            for (int l = 0; l < d; ++l) {
                const double df = xa[l] - Xs[kk * d + l];
                t += df * df * kp.inv_ls2[l];
            }
            double k0v, k1v, k2v;
            kappa_eval<1>(kp.kind, kp.floor_mode, t, k0v, k1v, k2v);
            for (int rl = rbeg; rl < rend; ++rl) {
                const int lp = rl - a * dp;
                const double val = (lp == 0) ? kp.variance * k0v
                    : 2.0 * kp.variance * k1v * (xa[lp - 1] - Xs[kk * d + lp - 1]) * kp.inv_ls2[lp - 1];
                out[(b * RP + rl) * npad + k] = val;
            }
        } else {
            for (int rl = rbeg; rl < rend; ++rl)
                out[(b * RP + rl) * npad + k] = (rl == q * dp && k < n) ? y[k] : 0.0;
        }
    }
}

int oei_launch_gram_ext(const oei_model* m, const double* Xq, int B, int q, int RP, double* out,
                        hipStream_t st) {
    dim3 grid((unsigned)((size_t)B * ((RP + GR_ROWS - 1) / GR_ROWS)), m->npad / 64);
    size_t lds = (size_t)(64 + GR_ROWS) * m->d * sizeof(double);
    hipLaunchKernelGGL(k_gram_ext, grid, dim3(256), lds, st, m->kp, m->X, m->y, m->n, m->npad, Xq, q, RP, out);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// ---------------------------------------------------------------------------------------------
// GEMM tile.  Block tile 64 x 64, K step 16, 4 waves; wave w owns rows [16w, 16w+16) x 64 columns
// as four 16x16 accumulators.  Operand lane maps of v_mfma_f64_16x16x4_f64: lane l supplies
// A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; accumulator register g of lane l is
// D[row = (l >> 4) + 4 g][col = l & 15].
// Ag/Bg point at the first row of the two 64-row operand panels (leading dimension ld, K along
// the row); Cg at the output tile (leading dimension ldc).
// One K step (GM_BK columns) of a wave's share of the tile from the staged panels: NH groups of 16
// rows against the 16-column groups TLO..THI.  `al` / `bl` point at the lane's entry of the first
// row group / column group in the A / B panel.  All fragments of the step are read first, then
// the matrix instructions run back to back (straight-line code: the group bounds are template
// arguments, so that the skipped groups of a diagonal block cost no branch inside the step).
// SWAP: the operands change roles (the B fragment goes in as the left operand), which leaves the
// transposed 16 x 16 blocks in the accumulators.
template <int NH, int TLO, int THI, bool SWAP = false>
__device__ __forceinline__ void gemm_mma_step(const double* __restrict__ al, const double* __restrict__ bl,
                                              double4_t (&acc)[NH][4]) {
    double av[NH][GM_BK / 4], bv[4][GM_BK / 4];
#pragma unroll
    for (int ks = 0; ks < GM_BK / 4; ++ks) {
#pragma unroll
        for (int h = 0; h < NH; ++h) av[h][ks] = al[h * 16 * GM_LDS + ks * 4];
#pragma unroll
        for (int t = TLO; t <= THI; ++t) bv[t][ks] = bl[t * 16 * GM_LDS + ks * 4];
    }
#pragma unroll
    for (int ks = 0; ks < GM_BK / 4; ++ks)
#pragma unroll
        for (int t = TLO; t <= THI; ++t)
#pragma unroll
            for (int h = 0; h < NH; ++h)
                acc[h][t] = SWAP ? __builtin_amdgcn_mfma_f64_16x16x4f64(bv[t][ks], av[h][ks], acc[h][t], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f64_16x16x4f64(av[h][ks], bv[t][ks], acc[h][t], 0, 0, 0);
}
// One tile of C = A B^T from the panels Ag0 (NH * 64 rows) and Bg0 (64 rows), K range [kbeg, kend),
// by 256 threads: wave w owns rows [16 NH w, 16 NH (w + 1)) x 64 columns as NH x 4 accumulators of
// 16 x 16.  Operand lane maps of v_mfma_f64_16x16x4_f64: lane l supplies A[i = l & 15][k = l >> 4]
// and B[k = l >> 4][j = l & 15]; accumulator register g of lane l is D[row = (l >> 4) + 4 g][col = l & 15].
// K steps of GM_BK columns go through two LDS buffers (the next step's stores cannot race this
// step's reads), the global loads of the next step are issued before this step's arithmetic.
// TRI: B is triangular and its 64 x 64 diagonal block lies at the end (1) or at the start (2) of
// the K range; inside that block a 16-column group of the output only meets non-zero entries in
// the K steps on its side of the diagonal, so those four steps are peeled with fewer column
// groups each (compile-time bounds: the accumulators then flow through straight-line code --
// with run-time bounds the compiler merged the variants through register copies, 48 moves per
// step, and with branches inside a step it waited for every fragment read).  TRI = 0: dense.
// SWAP: accumulator g of lane (li, lk) = (l & 15, l >> 4) then holds C[row li][column 4 lk + g] of
// its 16 x 16 block instead of C[row lk + 4 g][column li] -- the lane reads row 4 (li & 3) + (li >> 2)
// of a 16-row group of the B panel as the left operand, so that the block's row lk + 4 g is column
// 4 lk + g of C: a lane owns one row of C and four consecutive columns per column group.
template <int NH, int TRI, bool SWAP = false, bool DBUF = true>
__device__ __forceinline__ void gemm_tile_acc(const double* __restrict__ Ag0,
                                              const double* __restrict__ Bg0, int ld, int kbeg,
                                              int kend, double* As, double* Bs,
                                              double4_t (&acc)[NH][4], int arows = 1 << 30) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int srow = tid >> 2, scol = (tid & 3) * 4;        // staging: thread t moves 4 consecutive doubles of row t / 4
    // arows: rows of the A panel that exist (NH = 1 callers whose panel may run past the buffer:
    // the rows behind it repeat the last one and their results are never stored)
    const double* Ag = Ag0 + (size_t)(srow < arows ? srow : arows - 1) * ld + scol;
    const double* Bg = Bg0 + (size_t)srow * ld + scol;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[h][t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double2 a0[NH], a1[NH], b0, b1;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        a0[h] = *reinterpret_cast<const double2*>(Ag + (size_t)h * 64 * ld + kbeg);
        a1[h] = *reinterpret_cast<const double2*>(Ag + (size_t)h * 64 * ld + kbeg + 2);
    }
    b0 = *reinterpret_cast<const double2*>(Bg + kbeg);
    b1 = *reinterpret_cast<const double2*>(Bg + kbeg + 2);
    int buf = 0;
    const int li = lane & 15, lk = lane >> 4;
    const double* al = nullptr;
    const double* bl = nullptr;
    auto stage = [&](int k0) {
        if constexpr (!DBUF) __syncthreads();        // one buffer: the previous step's fragment reads are done
        double* as = As + buf * (NH * 64 * GM_LDS);
        double* bs = Bs + buf * (GM_BN * GM_LDS);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            as[(srow + 64 * h) * GM_LDS + scol + 0] = a0[h].x; as[(srow + 64 * h) * GM_LDS + scol + 1] = a0[h].y;
            as[(srow + 64 * h) * GM_LDS + scol + 2] = a1[h].x; as[(srow + 64 * h) * GM_LDS + scol + 3] = a1[h].y;
        }
        bs[srow * GM_LDS + scol + 0] = b0.x; bs[srow * GM_LDS + scol + 1] = b0.y;
        bs[srow * GM_LDS + scol + 2] = b1.x; bs[srow * GM_LDS + scol + 3] = b1.y;
        __syncthreads();
        if (k0 + GM_BK < kend) {
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                a0[h] = *reinterpret_cast<const double2*>(Ag + (size_t)h * 64 * ld + k0 + GM_BK);
                a1[h] = *reinterpret_cast<const double2*>(Ag + (size_t)h * 64 * ld + k0 + GM_BK + 2);
            }
            b0 = *reinterpret_cast<const double2*>(Bg + k0 + GM_BK);
            b1 = *reinterpret_cast<const double2*>(Bg + k0 + GM_BK + 2);
        }
        al = as + (wave * 16 * NH + li) * GM_LDS + lk;
        bl = bs + (SWAP ? 4 * (li & 3) + (li >> 2) : li) * GM_LDS + lk;
        if constexpr (DBUF) buf ^= 1;
    };
    int k0 = kbeg;
    if constexpr (TRI == 2) {
        stage(k0); gemm_mma_step<NH, 0, 0, SWAP>(al, bl, acc); k0 += GM_BK;
        stage(k0); gemm_mma_step<NH, 0, 1, SWAP>(al, bl, acc); k0 += GM_BK;
        stage(k0); gemm_mma_step<NH, 0, 2, SWAP>(al, bl, acc); k0 += GM_BK;
    }
    const int kfull = (TRI == 1) ? kend - GM_BN + GM_BK : kend;
    for (; k0 < kfull; k0 += GM_BK) { stage(k0); gemm_mma_step<NH, 0, 3, SWAP>(al, bl, acc); }
    if constexpr (TRI == 1) {
        stage(k0); gemm_mma_step<NH, 1, 3, SWAP>(al, bl, acc); k0 += GM_BK;
        stage(k0); gemm_mma_step<NH, 2, 3, SWAP>(al, bl, acc); k0 += GM_BK;
        stage(k0); gemm_mma_step<NH, 3, 3, SWAP>(al, bl, acc);
    }
}

// stores a wave's accumulators to the output tile (leading dimension ldc); rows / columns of the
// tile at and past rlim / clim are not written
template <int NH>
__device__ __forceinline__ void gemm_tile_store(double* __restrict__ Cg, int ldc, const double4_t (&acc)[NH][4],
                                                int rlim = 64 * NH, int clim = 64) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const size_t row = wave * 16 * NH + h * 16 + rq + 4 * g;
                if ((int)row < rlim && t * 16 + col < clim) Cg[row * ldc + t * 16 + col] = acc[h][t][g];
            }
}

// Ag/Bg point at the first row of the two operand panels (leading dimension ld, K along the row);
// Cg at the output tile (leading dimension ldc).  64 x 64 tile.
template <int TRI>
__device__ __forceinline__ void gemm_tile_nt(const double* __restrict__ Ag0,
                                             const double* __restrict__ Bg0,
                                             double* __restrict__ Cg, int ld, int ldc, int kbeg,
                                             int kend, double* As, double* Bs, int rlim = 64, int clim = 64) {
    double4_t acc[1][4];
    gemm_tile_acc<1, TRI>(Ag0, Bg0, ld, kbeg, kend, As, Bs, acc);
    gemm_tile_store<1>(Cg, ldc, acc, rlim, clim);
}

// The same with a 128-row A panel whose wave layout is 32 rows per wave: every B fragment read from
// LDS feeds two MFMAs and a workgroup does twice the work per prologue, barrier and epilogue.
#define GM_BM2 128
template <int TRI>
__device__ __forceinline__ void gemm_tile_nt_tall(const double* __restrict__ Ag0,
                                                  const double* __restrict__ Bg0,
                                                  double* __restrict__ Cg, int ld, int ldc, int kbeg,
                                                  int kend, double* As, double* Bs) {
    double4_t acc[2][4];
    gemm_tile_acc<2, TRI>(Ag0, Bg0, ld, kbeg, kend, As, Bs, acc);
    gemm_tile_store<2>(Cg, ldc, acc);
}

// Column tiles of a triangular product differ in their K range (tile j of nt: j + 1 units for a
// lower operand, nt - j for an upper one).  A workgroup takes a chain of one or two tiles of its row
// block whose units add up to nt -- the longest tile alone, then the second longest with the
// shortest, and so on -- so that every workgroup does the same work and there are (nt + 1) / 2
// starts per row block instead of nt.
__device__ __forceinline__ int gemm_chain_tiles(int chain, int nt, int lower, int (&tiles)[2]) {
    // ranks by length: rank 0 is the longest tile
    auto tile_of_rank = [&](int r) { return lower ? nt - 1 - r : r; };
    if (chain == 0) { tiles[0] = tile_of_rank(0); return 1; }
    const int ra = chain, rb = nt - chain;           // ranks ra (long) and rb (short), ra <= rb
    tiles[0] = tile_of_rank(ra);
    if (rb == ra) return 1;
    tiles[1] = tile_of_rank(rb);
    return 2;
}

__global__ __launch_bounds__(256) void k_gemm_nt_tall(const double* __restrict__ A,
                                                      const double* __restrict__ Bm,
                                                      double* __restrict__ C, int npad, int lower,
                                                      unsigned nrb) {
    __shared__ __attribute__((aligned(16))) double As[2 * GM_BM2 * GM_LDS];
    __shared__ __attribute__((aligned(16))) double Bs[2 * GM_BN * GM_LDS];
    // same workgroup order as k_gemm_nt: the chains of one row panel share an XCD (one L2)
    const int nt = npad / GM_BN, nch = nt / 2 + 1;
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const size_t rb = (size_t)(slot / nch) * 8 + xcd;
    if (rb >= nrb) return;
    int tiles[2];
    const int ntile = gemm_chain_tiles((int)(slot % nch), nt, lower, tiles);
    const size_t r0 = rb * GM_BM2;
    for (int i = 0; i < ntile; ++i) {
        const int j0 = tiles[i] * GM_BN;
        const int kbeg = lower ? 0 : j0;
        const int kend = lower ? (j0 + GM_BN) : npad;
        if (i) __syncthreads();                      // the previous tile's fragment reads are done
        if (lower) gemm_tile_nt_tall<1>(A + r0 * npad, Bm + (size_t)j0 * npad, C + r0 * npad + j0, npad, npad, kbeg, kend, As, Bs);
        else gemm_tile_nt_tall<2>(A + r0 * npad, Bm + (size_t)j0 * npad, C + r0 * npad + j0, npad, npad, kbeg, kend, As, Bs);
    }
}

__global__ __launch_bounds__(256) void k_gemm_nt(const double* __restrict__ A,
                                                 const double* __restrict__ Bm,
                                                 double* __restrict__ C, int npad, int lower,
                                                 unsigned nrb) {
    __shared__ __attribute__((aligned(16))) double As[2 * GM_BM * GM_LDS];
    __shared__ __attribute__((aligned(16))) double Bs[2 * GM_BN * GM_LDS];
    // Workgroups are dealt to the 8 XCDs round-robin by their linear id, and each XCD has its own
    // L2.  The nt column tiles of one 64-row panel of A are therefore given consecutive slots of
    // the SAME XCD, so that the panel is fetched from HBM once and re-read from that L2.
    const int nt = npad / GM_BN;
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const size_t rb = (size_t)(slot / nt) * 8 + xcd;
    if (rb >= nrb) return;
    const int j0 = (int)(slot % nt) * GM_BN;
    const size_t r0 = rb * GM_BM;
    const int kbeg = lower ? 0 : j0;
    const int kend = lower ? (j0 + GM_BN) : npad;
    if (lower) gemm_tile_nt<1>(A + r0 * npad, Bm + (size_t)j0 * npad, C + r0 * npad + j0, npad, npad, kbeg, kend, As, Bs);
    else gemm_tile_nt<2>(A + r0 * npad, Bm + (size_t)j0 * npad, C + r0 * npad + j0, npad, npad, kbeg, kend, As, Bs);
}

int oei_launch_gemm_nt(const double* A, const double* Bm, double* C, size_t rows_pad, int npad,
                       int lower, hipStream_t st) {
    if (rows_pad % GM_BM != 0 || npad % GM_BN != 0) {
        oei_set_error("gemm_nt: rows_pad=%zu / npad=%d not tile multiples", rows_pad, npad);
        return OEI_ERR_ARG;
    }
    const unsigned nt = (unsigned)(npad / GM_BN);
    if (rows_pad % GM_BM2 == 0) {
        const unsigned nrb = (unsigned)(rows_pad / GM_BM2);
        dim3 grid(((nrb + 7) / 8) * 8 * (nt / 2 + 1));
        hipLaunchKernelGGL(k_gemm_nt_tall, grid, dim3(256), 0, st, A, Bm, C, npad, lower, nrb);
    } else {
        const unsigned nrb = (unsigned)(rows_pad / GM_BM);
        dim3 grid(((nrb + 7) / 8) * 8 * nt);
        hipLaunchKernelGGL(k_gemm_nt, grid, dim3(256), 0, st, A, Bm, C, npad, lower, nrb);
    }
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// ---------------------------------------------------------------------------------------------
// Second GEMM fused with the n-length contraction of the gradient.  W = Z L^-1 is needed only in
//   Gacc[r][l] = sum_k (p_r alpha_k - 2 W[r,k]) K1[r,k] X[k,l]   (l < d),   Gacc[r][d] = sum_k (...)
// (r = candidate point, k = training point), so every 64 x 64 tile of W is contracted with its
// K1 tile and 64 rows of X while it is still in the MFMA accumulators, and W never travels to
// memory.  A workgroup owns one 64 x 64 tile (the K range of column tile j starts at its diagonal)
// and writes that tile's partial sums; the finishing kernel adds the npad / 64 partials of a row.
// The contraction is itself a small matrix product, Gacc = CK X with CK = (p alpha^T - 2 W) o K1
// (64 x 64) and X (64 x (d + 1), padded to 16 columns), and runs on the matrix pipe: the tile is
// accumulated with swapped operands, so that accumulator g of lane (li, kg) holds
// W[row li][column 4 kg + g] of its 16-column group -- exactly the left-operand layout of
// v_mfma_f64_16x16x4 (row li, k position kg) for a step that covers the columns 4 kg + g, with
// X[column 4 kg + g][l = li] from LDS as the right operand.  Sixteen matrix instructions per wave
// replace 32 sixteen-lane DPP reductions (about 400 vector instructions), and a lane reads 32
// contiguous bytes of its K1 row per column group.
// NH: 64-row groups of Z per workgroup (2 when the padded row count allows: every fragment of
// L^-T read from LDS then feeds two matrix instructions).
template <int DPM, int NH>
__global__ __launch_bounds__(256, NH == 1 ? 3 : 2) void k_gemm_grad(const double* __restrict__ Z,
                                                   const double* __restrict__ LinvT,
                                                   const double* __restrict__ K1,
                                                   const double* __restrict__ alpha,
                                                   const double* __restrict__ Xtrain,
                                                   const double* __restrict__ pvec,
                                                   double* __restrict__ Gpart, int npad, int d,
                                                   unsigned nrb, size_t nrows) {
    // (one LDS buffer with a second barrier per step would fit five workgroups per compute unit
    // instead of four: 0.207 against 0.215 ms, inside the run-to-run spread; not kept)
    constexpr bool DBUF = true;
    __shared__ __attribute__((aligned(16))) double As[(DBUF ? 2 : 1) * NH * GM_BM * GM_LDS];
    __shared__ __attribute__((aligned(16))) double Bs[(DBUF ? 2 : 1) * GM_BN * GM_LDS];
    __shared__ double Xs[64 * (DPM + 1)];
    // a balanced chain of one or two column tiles per workgroup (gemm_chain_tiles); the chains of
    // one row block of Z get consecutive slots of the same XCD (as in k_gemm_nt), so they run
    // together and share the block through that L2
    const int nt = npad / GM_BN, nch = nt / 2 + 1, dp = d + 1;
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const size_t rb = (size_t)(slot / nch) * 8 + xcd;
    if (rb >= nrb) return;
    int tiles[2];
    const int ntile = gemm_chain_tiles((int)(slot % nch), nt, 0, tiles);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const size_t r0 = (size_t)rb * (NH * GM_BM);
  for (int it = 0; it < ntile; ++it) {
    const unsigned split = (unsigned)tiles[it];
    const int j0 = (int)split * GM_BN;
    if (it) __syncthreads();                         // the previous tile's readers of Xs and the panels are done
    for (int e = tid; e < 64 * dp; e += 256) {
        const int k = e / dp, l = e - k * dp;
        Xs[k * (DPM + 1) + l] = (l < d) ? Xtrain[(size_t)(j0 + k) * d + l] : 1.0;
    }
    // L^-T is upper triangular: in the diagonal block (the first four K steps) column group t of
    // the tile only meets non-zero entries from K step t on
    double4_t acc[NH][4];
    gemm_tile_acc<NH, 2, true, DBUF>(Z + r0 * npad, LinvT + (size_t)j0 * npad, npad, j0, npad, As, Bs, acc);
    // (Xs was published by the barriers of the K loop)
    const double* alp = alpha + j0 + 4 * kg;
    const bool lin = li < dp;
    const double* xl = Xs + (size_t)(4 * kg) * (DPM + 1) + (lin ? li : 0);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const size_t rg = r0 + wave * 16 * NH + h * 16;       // first row of this 16-row group
        const size_t r = rg + li;
        const size_t rc = r < nrows ? r : nrows - 1;          // padding rows read a valid row, never stored
        const double pr = pvec[rc];
        const double* k1row = K1 + rc * npad + j0 + 4 * kg;
        double4_t cacc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double2 ka = *reinterpret_cast<const double2*>(k1row + 16 * t);
            const double2 kb = *reinterpret_cast<const double2*>(k1row + 16 * t + 2);
            const double k1v[4] = {ka.x, ka.y, kb.x, kb.y};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double ck = fma(pr, alp[16 * t + g], -2.0 * acc[h][t][g]) * k1v[g];
                const double xv = xl[(16 * t + g) * (DPM + 1)];
                cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(ck, lin ? xv : 0.0, cacc, 0, 0, 0);
            }
        }
        // cacc[g] = Gacc[row rg + kg + 4 g][l = li]
        if (lin) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const size_t ro = rg + kg + 4 * g;
                if (ro < nrows) Gpart[((size_t)split * nrows + ro) * dp + li] = cacc[g];
            }
        }
    }
  }
}

// The same kernel without Z = M11 A: the product runs on the A rows themselves (V = A L^-1 does not
// depend on the SDP solution) and M11 is applied to the tile before the contraction, W = M11 V per
// candidate, so that Z is neither written by the solver kernel nor read here.  A tile holds whole
// candidates (64 / q of them, q <= 64; the rows behind them are ignored).  After the K loop the
// 64 x 64 tile of V goes through LDS (in place of the operand panels); wave w then forms the rows
// 16 w .. 16 w + 15 of W on the matrix pipe as W^T = V_c^T M11^T over the rows of the candidates its
// 16 rows belong to: the left operand is V read from LDS with the same column permutation as the
// swapped K loop (lane li supplies column 4 (li & 3) + (li >> 2) of a 16-column group), the right
// operand the lane's row of the block-diagonal M11 (LDS, zero outside the row's own candidate), and
// the accumulators come out in the layout the contraction expects: accumulator g of lane (li, kg)
// = W[row li][column 4 kg + g].
// Dynamic LDS: the tile's M11 rows, 64 x (q | 1) doubles.
template <int DPM>
__global__ __launch_bounds__(256, 3) void k_gemm_grad_zfree(const double* __restrict__ A,
                                                   const double* __restrict__ LinvT,
                                                   const double* __restrict__ K1,
                                                   const double* __restrict__ alpha,
                                                   const double* __restrict__ Xtrain,
                                                   const double* __restrict__ pvec,
                                                   const double* __restrict__ M11,
                                                   double* __restrict__ Gpart, int npad, int d, int q,
                                                   unsigned nrb, size_t nrows, size_t arows) {
    constexpr int VLD = 66;
    static_assert(2 * GM_BM * GM_LDS + 2 * GM_BN * GM_LDS >= 64 * VLD, "the V tile reuses the operand panels");
    __shared__ __attribute__((aligned(16))) double Pan[2 * GM_BM * GM_LDS + 2 * GM_BN * GM_LDS];
    __shared__ double Xs[64 * (DPM + 1)];
    extern __shared__ __attribute__((aligned(16))) double Ms[];       // [64][q | 1]
    double* As = Pan;
    double* Bs = Pan + 2 * GM_BM * GM_LDS;
    const int nt = npad / GM_BN, nch = nt / 2 + 1, dp = d + 1;
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const size_t rb = (size_t)(slot / nch) * 8 + xcd;
    if (rb >= nrb) return;
    int tiles[2];
    const int ntile = gemm_chain_tiles((int)(slot % nch), nt, 0, tiles);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int cpt = 64 / q, live = cpt * q, mld = q | 1;
    const size_t r0 = rb * (size_t)live;
    // the tile's rows of M11 (block diagonal: row r of the tile meets the q columns of its own candidate)
    for (int e = tid; e < live * q; e += 256) {
        const int r = e / q, c = e - r * q;
        const size_t rg = r0 + r;
        Ms[r * mld + c] = rg < nrows ? M11[rg * q + c] : 0.0;       // M11 is [B][q][q]: row rg = b q + a
    }
    // this wave's rows 16 w .. 16 w + 15 and the candidates they belong to
    const int row = 16 * wave + li;
    const bool rowin = row < live && r0 + row < nrows;
    const int cand = (rowin ? row : 0) / q, cbase = cand * q;
    const int c_lo = (16 * wave) / q, c_hi = min(cpt - 1, (16 * wave + 15) / q);
    const int klo = c_lo * q, khi = (c_hi + 1) * q;                   // uniform; empty for a wave past the live rows
    const int ks = khi > klo ? (khi - klo + 3) >> 2 : 0;
    const int pcol = 4 * (li & 3) + (li >> 2);
  for (int it = 0; it < ntile; ++it) {
    const unsigned split = (unsigned)tiles[it];
    const int j0 = (int)split * GM_BN;
    __syncthreads();                                 // the previous tile's readers of Xs and V are done; Ms is published
    for (int e = tid; e < 64 * dp; e += 256) {
        const int k = e / dp, l = e - k * dp;
        Xs[k * (DPM + 1) + l] = (l < d) ? Xtrain[(size_t)(j0 + k) * d + l] : 1.0;
    }
    double4_t acc[1][4];
    const size_t left = arows - r0;
    gemm_tile_acc<1, 2, true, true>(A + r0 * npad, LinvT + (size_t)j0 * npad, npad, j0, npad, As, Bs, acc,
                                    left < 64 ? (int)left : 64);
    __syncthreads();                                 // every wave has read its last fragments of the panels
    double* Vs = Pan;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) Vs[row * VLD + 16 * t + 4 * kg + g] = acc[0][t][g];
    __syncthreads();
    double4_t wacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) wacc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int s = 0; s < ks; ++s) {
        const int rp = klo + 4 * s + kg;             // row of the tile = k position of this step
        const bool on = rowin && rp >= cbase && rp < cbase + q;
        const double mv = Ms[row * mld + (on ? rp - cbase : 0)];
        const double* vrow = Vs + (rp < 64 ? rp : 63) * VLD + pcol;
        double vv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) vv[t] = vrow[16 * t];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            wacc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(rp < khi ? vv[t] : 0.0, on ? mv : 0.0, wacc[t], 0, 0, 0);
    }
    // contraction with K1 and X as in k_gemm_grad (Xs was published by the barriers above)
    const double* alp = alpha + j0 + 4 * kg;
    const bool lin = li < dp;
    const double* xl = Xs + (size_t)(4 * kg) * (DPM + 1) + (lin ? li : 0);
    {
        const size_t rg = r0 + wave * 16;                     // first row of this 16-row group
        const size_t r = rg + li;
        const size_t rc = rowin ? r : (r0 < nrows ? r0 : nrows - 1);   // rows past the live ones read a valid row, never stored
        const double pr = pvec[rc];
        const double* k1row = K1 + rc * npad + j0 + 4 * kg;
        double4_t cacc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double2 ka = *reinterpret_cast<const double2*>(k1row + 16 * t);
            const double2 kb = *reinterpret_cast<const double2*>(k1row + 16 * t + 2);
            const double k1v[4] = {ka.x, ka.y, kb.x, kb.y};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const double ck = fma(pr, alp[16 * t + g], -2.0 * wacc[t][g]) * k1v[g];
                const double xv = xl[(16 * t + g) * (DPM + 1)];
                cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(ck, lin ? xv : 0.0, cacc, 0, 0, 0);
            }
        }
        // cacc[g] = Gacc[row rg + kg + 4 g][l = li]
        if (lin) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rt = wave * 16 + kg + 4 * g;
                const size_t ro = r0 + rt;
                if (rt < live && ro < nrows) Gpart[((size_t)split * nrows + ro) * dp + li] = cacc[g];
            }
        }
    }
  }
}

int oei_launch_gemm_grad_zfree(const oei_model* m, const double* A, const double* K1, const double* pvec,
                               const double* M11, double* Gpart, size_t nrows, size_t rows_pad, int q,
                               hipStream_t st) {
    const unsigned nt = (unsigned)(m->npad / GM_BN);
    if (m->d + 1 > 12 || q > 64) return OEI_ERR_ARG;
    const int live = (64 / q) * q;
    const unsigned nrb = (unsigned)((nrows + live - 1) / live);
    const size_t lds = (size_t)64 * (q | 1) * sizeof(double);
    dim3 grid(((nrb + 7) / 8) * 8 * (nt / 2 + 1));
    if (m->d + 1 <= 8)
        hipLaunchKernelGGL((k_gemm_grad_zfree<8>), grid, dim3(256), lds, st, A, m->LinvT, K1, m->alpha, m->X, pvec, M11, Gpart, m->npad, m->d, q, nrb, nrows, rows_pad);
    else
        hipLaunchKernelGGL((k_gemm_grad_zfree<12>), grid, dim3(256), lds, st, A, m->LinvT, K1, m->alpha, m->X, pvec, M11, Gpart, m->npad, m->d, q, nrb, nrows, rows_pad);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// Gpart [npad / 64][nrows][d+1]; returns OEI_ERR_ARG when d + 1 exceeds the register budget of the
// fused kernel (the caller then takes the two-kernel route).
int oei_launch_gemm_grad(const oei_model* m, const double* Z, const double* K1, const double* pvec,
                         double* Gpart, size_t nrows, size_t rows_pad, hipStream_t st) {
    const unsigned nt = (unsigned)(m->npad / GM_BN);
    if (m->d + 1 > 12) return OEI_ERR_ARG;
    // (128-row workgroups, NH = 2, were measured slower here: 0.235 against 0.210 ms at 81 920 rows.
    // The instance for d + 1 <= 8 keeps X in a smaller LDS array: four workgroups per compute unit
    // instead of three, 0.210 against 0.226 ms.)
    const unsigned nrb = (unsigned)(rows_pad / GM_BM);
    dim3 grid(((nrb + 7) / 8) * 8 * (nt / 2 + 1));
    if (m->d + 1 <= 8)
        hipLaunchKernelGGL((k_gemm_grad<8, 1>), grid, dim3(256), 0, st, Z, m->LinvT, K1, m->alpha, m->X, pvec, Gpart, m->npad, m->d, nrb, nrows);
    else
        hipLaunchKernelGGL((k_gemm_grad<12, 1>), grid, dim3(256), 0, st, Z, m->LinvT, K1, m->alpha, m->X, pvec, Gpart, m->npad, m->d, nrb, nrows);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// The 64 x 64 tile when only the first 16 rows of the A panel hold data (the last row tile of a
// block whose row count is just past a multiple of 64): the four waves share those 16 rows and
// take one 16-column group each, a quarter of the matrix instructions per wave.  Rows 16..63 of the
// output tile are not written (they are not rows of this block), nor are columns at and past clim.
__device__ __forceinline__ void gemm_tile_nt_strip16(const double* __restrict__ Ag0,
                                                     const double* __restrict__ Bg0,
                                                     double* __restrict__ Cg, int ld, int ldc,
                                                     int kend, double* As, double* Bs, int clim = 64) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int srow = tid >> 2, scol = (tid & 3) * 4;
    const double* Ag = Ag0 + (size_t)(srow & 15) * ld + scol;      // only 16 A rows are staged (four times over)
    const double* Bg = Bg0 + (size_t)srow * ld + scol;
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    double2 a0, a1, b0, b1;
    a0 = *reinterpret_cast<const double2*>(Ag);
    a1 = *reinterpret_cast<const double2*>(Ag + 2);
    b0 = *reinterpret_cast<const double2*>(Bg);
    b1 = *reinterpret_cast<const double2*>(Bg + 2);
    int buf = 0;
    for (int k0 = 0; k0 < kend; k0 += GM_BK) {
        double* as = As + buf * (GM_BM * GM_LDS);
        double* bs = Bs + buf * (GM_BN * GM_LDS);
        if (srow < 16) {
            as[srow * GM_LDS + scol + 0] = a0.x; as[srow * GM_LDS + scol + 1] = a0.y;
            as[srow * GM_LDS + scol + 2] = a1.x; as[srow * GM_LDS + scol + 3] = a1.y;
        }
        bs[srow * GM_LDS + scol + 0] = b0.x; bs[srow * GM_LDS + scol + 1] = b0.y;
        bs[srow * GM_LDS + scol + 2] = b1.x; bs[srow * GM_LDS + scol + 3] = b1.y;
        __syncthreads();
        if (k0 + GM_BK < kend) {
            a0 = *reinterpret_cast<const double2*>(Ag + k0 + GM_BK);
            a1 = *reinterpret_cast<const double2*>(Ag + k0 + GM_BK + 2);
            b0 = *reinterpret_cast<const double2*>(Bg + k0 + GM_BK);
            b1 = *reinterpret_cast<const double2*>(Bg + k0 + GM_BK + 2);
        }
        const int li = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < GM_BK / 4; ++ks) {
            const double av = as[li * GM_LDS + ks * 4 + lk];
            const double bv = bs[(wave * 16 + li) * GM_LDS + ks * 4 + lk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
        buf ^= 1;
    }
    const int col = lane & 15, rq = lane >> 4;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (wave * 16 + col < clim) Cg[(size_t)(rq + 4 * g) * ldc + wave * 16 + col] = acc[g];
}

// G_b = E_b E_b^T for every candidate: lower block-triangle of 64x64 tiles, tile index t of
// blockIdx.y enumerates (I >= J).  E [B*RP, npad] with the blocks packed densely (RP a multiple of
// 16, not of 64: the last tile of a block is partial, and what a panel reads past the block's rows
// -- the next block, or the slack behind the last one -- lands in rows and columns that are not
// stored), G [B, RP, RP].  `used`: rows of a block that hold data (the rest are zero rows).
__global__ __launch_bounds__(256) void k_gram_blk(const double* __restrict__ E,
                                                  double* __restrict__ G, int RP, int npad, int used) {
    __shared__ __attribute__((aligned(16))) double As[2 * GM_BM * GM_LDS];
    __shared__ __attribute__((aligned(16))) double Bs[2 * GM_BN * GM_LDS];
    const size_t b = blockIdx.x;
    int t = blockIdx.y, I = 0;
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    const double* Eb = E + b * (size_t)RP * npad;
    double* Gt = G + b * (size_t)RP * RP + (size_t)I * 64 * RP + J * 64;
    const int rlim = min(64, RP - 64 * I), clim = min(64, RP - 64 * J);
    if (used - I * 64 <= 16)
        gemm_tile_nt_strip16(Eb + (size_t)I * 64 * npad, Eb + (size_t)J * 64 * npad, Gt, npad, RP, npad, As, Bs, clim);
    else
        gemm_tile_nt<0>(Eb + (size_t)I * 64 * npad, Eb + (size_t)J * 64 * npad, Gt, npad, RP, 0, npad, As, Bs, rlim, clim);
}

int oei_launch_gram_blk(const double* E, double* G, int B, int RP, int npad, hipStream_t st, int used) {
    const int nt = (RP + 63) / 64;
    if (used <= 0 || used > RP) used = RP;
    dim3 grid((unsigned)B, nt * (nt + 1) / 2);
    hipLaunchKernelGGL(k_gram_blk, grid, dim3(256), 0, st, E, G, RP, npad, used);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}
