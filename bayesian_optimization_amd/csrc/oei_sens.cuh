// Sensitivity of the SDP solution to perturbations of Omega, one workgroup per candidate.
//
// The reference differentiates the KKT conditions (M - C_i) y_i = 0, sum y_i y_i^T = Omega and
// solves the resulting (k_^2 + n_tri)-dimensional sparse system with Pardiso for every direction
// (sdp.py:159-234).  In the reduced variables of oei_sdp.cuh the same linearisation is
//
//   N S N = D(p)/4,   diag(B) - 2 B p = 4 (mu - fmin),   B = N^-1,
//
// whose differential, in the eigenbasis X = W S W^T = V lam V^T (U = W^T V, R = W^-1 V = U^-T,
// sq = sqrt(lam), Phi_ab = 2/(sq_a + sq_b), Rs = R diag(sq)), is
//
//   dN = U Z U^T,   Z = Phi o (R^T dD R / 4  -  (U^T dS U) / (4 sq sq^T)),   dB = -4 Rs Z Rs^T,
//   (hess g) dp = dmu - (diag(dB_S) - 2 dB_S p)/4        (dB_S: the dS part of dB; q x q solve)
//   dc = dmu + (dB p + B dp)/2,  dn = -dN c - N dc,  dnu = 2 c^T N dc + c^T dN c,
//   dM = -[[dN, dn], [dn^T, dnu]].
//
// Cost per direction: five q^3 products in LDS instead of a solve with a 672 x 672 (q = 20)
// factor.  The result equals the reference's dM at the exact optimum (checked against a dense
// solve of the reference's system in the tests).
#pragma once
#include "oei_sdp.cuh"

struct SensVec {
    lds_f64 *dmu, *dp, *rd, *t1, *t2, *dc, *w1, *w2, *c;   // each sdp_vlen(q) long
};
#define SENS_NVEC 9

__device__ inline void sens_carve(SensVec& sv, lds_f64* base, int q) {
    const int VL = sdp_vlen(q);
    sv.dmu = base;           sv.dp = base + VL;      sv.rd = base + 2 * VL;
    sv.t1 = base + 3 * VL;   sv.t2 = base + 4 * VL;  sv.dc = base + 5 * VL;
    sv.w1 = base + 6 * VL;   sv.w2 = base + 7 * VL;  sv.c = base + 8 * VL;
}

// After sdp_eval at the optimum w.p (full LDS layout): store U in w.S, R in w.R, factor -hess g in
// w.H (lower Cholesky), c in sv.c, t1 = sq o rp.  Leaves w.X, w.V, w.Kt free.  Returns non-zero
// on breakdown.
template <int T, int Q>
__device__ int sens_prepare(const SdpWork w, const lds_f64* mu, const SensVec sv) {
    const int tid = threadIdx.x;
    const int q = w.q, LD = w.LD;
    __syncthreads();
    for (int e = tid; e < q * q; e += T) {         // the evaluation left U in w.X
        const int i = e / q, a = e - i * q;
        w.S[i * LD + a] = w.X[i * LD + a];
        w.R[i * LD + a] = sdp_R(w, i, a);
    }
    __syncthreads();
    if (sdp_newton_dir<T, Q>(w)) return 1;
    __syncthreads();
    if (tid < q) {
        double acc = 0.0;
        for (int a = 0; a < q; ++a) acc += w.R[tid * LD + a] * w.sq[a] * w.rp[a];
        sv.c[tid] = mu[tid] + acc;
        sv.t1[tid] = w.sq[tid] * w.rp[tid];
        w.rc[tid] = 1.0 / w.H[tid * LD + tid];           // for the per-direction triangular solves
    }
    __syncthreads();
    return 0;
}

// C[i][j] = sum_k A[i][k] * B[k][j] * sc[k]  (TA: use A[k][i]; TB: use B[j][k]; SC: with the
// scale vector, else without).  Eight terms at a time: the LDS reads of a group are issued before
// its multiply-adds so that the read latency is paid once per group, not once per term.
template <int T, bool TA, bool TB, bool SC = false>
__device__ __forceinline__ void lds_mm(int q, int LD, const lds_f64* A, const lds_f64* B, lds_f64* C,
                                       const lds_f64* sc = nullptr) {
    for (int e = threadIdx.x; e < q * q; e += T) {
        const int i = e / q, j = e - i * q;
        const lds_f64* ap = TA ? A + i : A + i * LD;
        const lds_f64* bp = TB ? B + j * LD : B + j;
        const int sa = TA ? LD : 1, sb = TB ? 1 : LD;
        double acc0 = 0.0, acc1 = 0.0;
        int k = 0;
        for (; k + 8 <= q; k += 8) {
            double av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = ap[(k + u) * sa];
                bv[u] = bp[(k + u) * sb];
                if (SC) av[u] *= sc[k + u];
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) { acc0 += av[u] * bv[u]; acc1 += av[u + 1] * bv[u + 1]; }
            SDP_READS_FIRST(24, 16);
        }
        for (; k < q; ++k) acc0 += ap[k * sa] * bp[k * sb] * (SC ? sc[k] : 1.0);
        C[i * LD + j] = acc0 + acc1;
    }
}

// sum_k a[k sa] b[k sb] over k < q, and sum_k r[k] s[k] (x[k sx] - 2 y[k]); the reads of ten (eight)
// terms are issued together, so a 20-term sum pays the LDS latency twice instead of per term.
__device__ __forceinline__ double lds_dot(const lds_f64* a, int sa, const lds_f64* b, int sb, int q) {
    double acc0 = 0.0, acc1 = 0.0;
    int k = 0;
    for (; k + 10 <= q; k += 10) {
        double av[10], bv[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) { av[u] = a[(k + u) * sa]; bv[u] = b[(k + u) * sb]; }
#pragma unroll
        for (int u = 0; u < 10; u += 2) { acc0 = fma(av[u], bv[u], acc0); acc1 = fma(av[u + 1], bv[u + 1], acc1); }
        SDP_READS_FIRST(20, 10);
    }
    for (; k < q; ++k) acc0 = fma(a[k * sa], b[k * sb], acc0);
    return acc0 + acc1;
}
__device__ __forceinline__ double lds_dot_rs(const lds_f64* r, const lds_f64* s, const lds_f64* x, int sx,
                                             const lds_f64* y, int q) {
    double acc0 = 0.0, acc1 = 0.0;
    int k = 0;
    for (; k + 8 <= q; k += 8) {
        double rv[8], sv[8], xv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { rv[u] = r[k + u]; sv[u] = s[k + u]; xv[u] = x[(k + u) * sx]; yv[u] = y[k + u]; }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            acc0 = fma(rv[u] * sv[u], fma(-2.0, yv[u], xv[u]), acc0);
            acc1 = fma(rv[u + 1] * sv[u + 1], fma(-2.0, yv[u + 1], xv[u + 1]), acc1);
        }
        SDP_READS_FIRST(32, 24);
    }
    for (; k < q; ++k) acc0 = fma(r[k] * s[k], fma(-2.0, y[k], x[k * sx]), acc0);
    return acc0 + acc1;
}

// Synchronisation between the phases of a direction: workgroup-wide, or -- when every wave works
// on a direction of its own in private scratch (WAVE) -- nothing but a compiler fence: the LDS
// serves the accesses of one wave in program order.
template <bool WAVE>
__device__ __forceinline__ void sens_sync() {
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// The same product for one wave on the matrix pipe (q <= 32): v_mfma_f64_16x16x4 takes one entry of
// each operand per lane -- lane l supplies A'[16 I + (l & 15)][4 s + (l >> 4)] and
// B'[4 s + (l >> 4)][16 J + (l & 15)] -- so a q = 20 product is 20 LDS reads, 20 MFMAs and 16
// stores per lane instead of about 100 reads and 125 multiply-adds, and the vector unit stays free
// for the wave that shares the SIMD.  Entries past q are supplied as zeros.
typedef double sens_double4 __attribute__((ext_vector_type(4)));
// I0, J0: first row and column tile of the NT x NT tiles this wave computes (a workgroup of four
// waves covers q <= 64 with NT = 2 and (I0, J0) = (0,0), (0,2), (2,0), (2,2)).
template <int NT, bool TA, bool TB, bool SC>
__device__ __forceinline__ void wave_mm_mfma(int lane, int q, int LD, const lds_f64* A, const lds_f64* B,
                                             lds_f64* C, const lds_f64* sc, int I0 = 0, int J0 = 0) {
    const int li = lane & 15, kg = lane >> 4;
    sens_double4 acc[NT][NT];
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J) acc[I][J] = (sens_double4){0.0, 0.0, 0.0, 0.0};
    const int ks = (q + 3) >> 2;
    for (int s = 0; s < ks; ++s) {
        const int k = 4 * s + kg;
        const bool kin = k < q;
        const int kc = kin ? k : 0;
        double av[NT], bv[NT];
        const double scale = SC ? sc[kc] : 1.0;
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            const int i = 16 * (I0 + I) + li;
            const bool in = kin && i < q;
            const int ic = in ? i : 0;
            const double x = A[TA ? kc * LD + ic : ic * LD + kc];
            av[I] = in ? (SC ? x * scale : x) : 0.0;
        }
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            const int j = 16 * (J0 + J) + li;
            const bool in = kin && j < q;
            const int jc = in ? j : 0;
            const double x = B[TB ? jc * LD + kc : kc * LD + jc];
            bv[J] = in ? x : 0.0;
        }
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = 0; J < NT; ++J)
                acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I], bv[J], acc[I][J], 0, 0, 0);
    }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = 16 * (I0 + I) + kg + 4 * g, j = 16 * (J0 + J) + li;
                if (i < q && j < q) C[i * LD + j] = acc[I][J][g];
            }
}

// The 2 x 2 tiles of a wave that has its SIMD to itself or shares it with one other wave (the
// 256-thread kernels): a wave issues in order, so the index arithmetic and LDS reads of a step (about
// 40 instructions in wave_mm_mfma) and its four matrix instructions take turns -- a q = 50 product
// took 7.1 k cycles for 3.3 k cycles of matrix instructions (tools/micro/mm_lds.hip).  Here the
// lane's operand pointers advance by a constant per step, rows and columns past q carry no masks
// (they read row / column 0 and only reach entries of C that are not stored), the next step's reads
// are issued before this step's matrix instructions, and the partial last step (q not a multiple of
// four) is peeled: 5.0 k cycles.
template <bool TA, bool TB, bool SC>
__device__ __forceinline__ void wave_mm_mfma_lean(int lane, int q, int LD, const lds_f64* A, const lds_f64* B,
                                                  lds_f64* C, const lds_f64* sc, int I0, int J0) {
    const int li = lane & 15, kg = lane >> 4;
    sens_double4 acc[2][2];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) acc[I][J] = (sens_double4){0.0, 0.0, 0.0, 0.0};
    const lds_f64* pa[2];
    const lds_f64* pb[2];
#pragma unroll
    for (int I = 0; I < 2; ++I) {
        const int i = 16 * (I0 + I) + li, j = 16 * (J0 + I) + li;
        const int ic = i < q ? i : 0, jc = j < q ? j : 0;
        pa[I] = A + (TA ? kg * LD + ic : ic * LD + kg);
        pb[I] = B + (TB ? jc * LD + kg : kg * LD + jc);
    }
    const lds_f64* ps = SC ? sc + kg : nullptr;
    const int sa = TA ? 4 * LD : 4, sb = TB ? 4 : 4 * LD;
    const int ksf = q >> 2;                                  // steps whose four k positions all lie inside q
    double av[2], bv[2], an[2], bn[2];
    {
        const double s0 = SC ? ps[0] : 1.0;
#pragma unroll
        for (int I = 0; I < 2; ++I) { const double x = pa[I][0]; av[I] = SC ? x * s0 : x; bv[I] = pb[I][0]; }
    }
    for (int s = 0; s < ksf; ++s) {
        const bool more = s + 1 < ksf;
        if (SC) ps += more ? 4 : 0;
        const double sn = SC ? ps[0] : 1.0;
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            pa[I] += more ? sa : 0; pb[I] += more ? sb : 0;
            const double x = pa[I][0];
            an[I] = SC ? x * sn : x; bn[I] = pb[I][0];
        }
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J)
                acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I], bv[J], acc[I][J], 0, 0, 0);
#pragma unroll
        for (int I = 0; I < 2; ++I) { av[I] = an[I]; bv[I] = bn[I]; }
    }
    if (q & 3) {
        const int k = 4 * ksf + kg;
        const bool kin = k < q;
        const int kc = kin ? k : 0;
        const double s0 = SC ? sc[kc] : 1.0;
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            const int i = 16 * (I0 + I) + li, j = 16 * (J0 + I) + li;
            const int ic = i < q ? i : 0, jc = j < q ? j : 0;
            const double x = A[TA ? kc * LD + ic : ic * LD + kc], y = B[TB ? jc * LD + kc : kc * LD + jc];
            av[I] = kin ? (SC ? x * s0 : x) : 0.0;
            bv[I] = kin ? y : 0.0;
        }
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J)
                acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I], bv[J], acc[I][J], 0, 0, 0);
    }
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = 16 * (I0 + I) + kg + 4 * g, j = 16 * (J0 + J) + li;
                if (i < q && j < q) C[i * LD + j] = acc[I][J][g];
            }
}

// The product for a 256-thread workgroup and q <= 64 on the matrix pipe: wave w computes the
// 32 x 32 block (w >> 1, w & 1) of C; a block that lies past q altogether is skipped.
template <bool TA, bool TB, bool SC>
__device__ __forceinline__ void wg4_mm_mfma(int q, int LD, const lds_f64* A, const lds_f64* B, lds_f64* C,
                                            const lds_f64* sc = nullptr) {
    const int wave = threadIdx.x >> 6, I0 = 2 * (wave >> 1), J0 = 2 * (wave & 1);
    if (16 * I0 < q && 16 * J0 < q) wave_mm_mfma_lean<TA, TB, SC>(threadIdx.x & 63, q, LD, A, B, C, sc, I0, J0);
}

template <int T, bool WAVE, bool TA, bool TB, bool SC>
__device__ __forceinline__ void sens_mm(int tid, int q, int LD, const lds_f64* A, const lds_f64* B,
                                        lds_f64* C, const lds_f64* sc = nullptr) {
    if constexpr (WAVE) {
        if (q <= 16) wave_mm_mfma<1, TA, TB, SC>(tid, q, LD, A, B, C, sc);
        else wave_mm_mfma<2, TA, TB, SC>(tid, q, LD, A, B, C, sc);
    } else if constexpr (T == 256) {
        wg4_mm_mfma<TA, TB, SC>(q, LD, A, B, C, sc);
    } else {
        lds_mm<T, TA, TB, SC>(q, LD, A, B, C, sc);
    }
}

// What a direction reads besides the solver state: the perturbation itself and, optionally,
// the two coefficient matrices that do not depend on the direction.
struct SensDir {
    // RANK2 form: dS = e_a v'^T + v' e_a^T with v'_j = vrow[j] (j != a), v'_a = vrow[a] / 2
    int a;
    const lds_f64* vrow;
    lds_f64* wv;                 // [q] scratch for U^T v' (YFORM: the caller's U^T v', read only)
    // phiS[a][b] = -1 / (2 sq_a sq_b (sq_a + sq_b)),  phiD[a][b] = 1 / (2 (sq_a + sq_b)); or null
    const lds_f64 *phiS, *phiD;
    const lds_f64* utc = nullptr;   // YFORM: U^T c
};

// One direction.  In: sv.dmu and dS -- a symmetric matrix in w.X, or in the RANK2 form of
// SensDir.  Out: dN in w.X, sv.w1 = N dc, sv.w2 = dN c, sv.dp, Z in w.Kt; returns dnu.
// dM11 = -dN, dM[:q,q] = w1 + w2, dM[q][q] = -dnu.
// Needs w.rc = 1 / diag(L) of the factor in w.H (sens_prepare).  With WAVE the calling wave works
// alone: w.X, w.V, w.Kt and the sv vectors other than c, t1 are its own, `T` is ignored.  In the
// RANK2 form w.X is not an input and may be the same buffer as w.Kt (Z is then overwritten by dN).
// QE > 0: q is known at compile time (q == QE), which turns the index arithmetic of the passes and
// the strides of the products into constants.
// YFORM (with RANK2): the caller needs dN only in products dN x with vectors x whose U^T x it
// already holds (the Hessian contraction: x = v' of every direction, c, mu), so the direction
// stops at Y = U Z (in w.V; dN x = Y (U^T x)) and leaves out the product dN = Y U^T; its own
// U^T v' comes in through sd.wv.  Out: Y in w.V, sv.w1 = N dc, sv.w2 = dN c = Y (U^T c), sv.dp.
// LINV (one wave): w.H holds L^-1 as a full matrix (zeros above the diagonal) instead of L, and
// the two triangular solves become two matrix-vector products -- 2 q multiply-adds per lane with
// no chain through the scalar unit, against 2 q dependent broadcast steps.
#ifdef OEI_HESS_TIMING
#define SENS_STAMP(i) do { const long long t_ = clock64(); sens_cyc[i] += t_ - sens_t0; sens_t0 = t_; } while (0)
#else
#define SENS_STAMP(i) do { } while (0)
#endif
template <int T, bool WAVE, bool RANK2, int QE = 0, bool YFORM = false, bool LINV = false>
__device__ __forceinline__ double sens_direction(const SdpWork w, const SensVec sv, const SensDir sd
#ifdef OEI_HESS_TIMING
                                                 , long long (&sens_cyc)[12]
#endif
                                                 ) {
#ifdef OEI_HESS_TIMING
    long long sens_t0 = clock64();
#endif
    constexpr int TT = WAVE ? 64 : T;
    const int tid = WAVE ? (threadIdx.x & 63) : threadIdx.x;
    const int q = QE > 0 ? QE : w.q, LD = QE > 0 ? (QE | 1) : w.LD;
    const lds_f64* U = w.S;
    lds_f64* Ta = w.X;
    lds_f64* Tb = w.V;
    lds_f64* Tc = w.Kt;
    // row index of entry e of a q x q pass without an integer division (q is a run-time value
    // here: the division is some forty instructions per entry): exact for e < 4096, q <= 64
    const float rq = 1.0f / (float)q;
    auto row_of = [&](int e) { return QE > 0 ? e / QE : (int)(((float)e + 0.5f) * rq); };
    sens_sync<WAVE>();
    if constexpr (RANK2) {
        // U^T dS U = u wv^T + wv u^T with u = row a of U, wv = U^T v'
        if constexpr (!YFORM) {
            if (tid < q) {
                const double acc = lds_dot(U + tid, LD, sd.vrow, 1, q);
                sd.wv[tid] = acc - 0.5 * U[sd.a * LD + tid] * sd.vrow[sd.a];
            }
            sens_sync<WAVE>();
        }
        const lds_f64* ua = U + sd.a * LD;
        for (int e = tid; e < q * q; e += TT) {            // ZS
            const int a = row_of(e), b = e - a * q;
            Tc[a * LD + b] = (ua[a] * sd.wv[b] + sd.wv[a] * ua[b]) * sd.phiS[a * LD + b];
        }
    } else {
        sens_mm<T, WAVE, false, false, false>(tid, q, LD, Ta, U, Tb);            // Tb = dS U
        sens_sync<WAVE>();
        sens_mm<T, WAVE, true, false, false>(tid, q, LD, U, Tb, Tc);             // Tc = U^T dS U
        sens_sync<WAVE>();
        for (int e = tid; e < q * q; e += TT) {            // ZS
            const int a = row_of(e), b = e - a * q;
            const double sa = w.sq[a], sb = w.sq[b];
            Tc[a * LD + b] *= sd.phiS ? sd.phiS[a * LD + b] : -0.5 * sdp_rcp(sa * sb * (sa + sb));
        }
    }
    sens_sync<WAVE>();
    SENS_STAMP(0);
    // rhs = dmu - (dBS_ii - 2 (dBS p)_i)/4 with dBS = -4 Rs ZS Rs^T, Rs = R diag(sq):
    // M1 = ZS Rs^T (into Tb), t2 = ZS t1, rhs_i = dmu_i + sum_a Rs[i][a] (M1[a][i] - 2 t2[a])
    sens_mm<T, WAVE, false, true, true>(tid, q, LD, Tc, w.R, Tb, w.sq);
    if (tid < q) sv.t2[tid] = lds_dot(Tc + tid * LD, 1, sv.t1, 1, q);
    sens_sync<WAVE>();
    SENS_STAMP(1);
    double rhs = 0.0;
    if (tid < q) rhs = sv.dmu[tid] + lds_dot_rs(w.R + tid * LD, w.sq, Tb + tid, LD, sv.t2, q);
    // A x = rhs with A = L L^T = -hess g
    if constexpr (LINV) {
        static_assert(!LINV || TT == 64, "LINV is a one-wave form");
        if (tid < q) sv.dp[tid] = rhs;
        sens_sync<WAVE>();
        const double y = (tid < q) ? lds_dot(w.H + tid * LD, 1, sv.dp, 1, q) : 0.0;      // L^-1 rhs
        sens_sync<WAVE>();
        if (tid < q) sv.rd[tid] = y;
        sens_sync<WAVE>();
        if (tid < q) sv.dp[tid] = -lds_dot(w.H + tid, LD, sv.rd, 1, q);                  // -L^-T y
    } else if constexpr (TT == 64) {
        // one wave: the running vector stays in a register, solved components are broadcast
        // through the scalar unit; L is read from LDS (reads do not depend on the chain)
        const int row = tid < q ? tid : 0;
        const double dinv = w.rc[row];
        double bv = rhs;
#pragma unroll 4
        for (int k = 0; k < q; ++k) {
            const double lk = w.H[row * LD + k];
            const double yk = sdp_lane_bcast(bv * dinv, k);
            bv = (tid == k) ? yk : (tid > k ? bv - lk * yk : bv);
        }
#pragma unroll 4
        for (int k = q - 1; k >= 0; --k) {
            const double lk = w.H[k * LD + row];
            const double xk = sdp_lane_bcast(bv * dinv, k);
            bv = (tid == k) ? xk : (tid < k ? bv - lk * xk : bv);
        }
        if (tid < q) sv.dp[tid] = -bv;
    } else {
        if (tid < q) w.step[tid] = rhs;
        __syncthreads();
        for (int k = 0; k < q; ++k) {
            const double yk = w.step[k] * w.rc[k];
            __syncthreads();
            if (tid == k) w.step[k] = yk;
            if (tid > k && tid < q) w.step[tid] -= w.H[tid * LD + k] * yk;
            __syncthreads();
        }
        for (int k = q - 1; k >= 0; --k) {
            const double xk = w.step[k] * w.rc[k];
            __syncthreads();
            if (tid == k) w.step[k] = xk;
            if (tid < k) w.step[tid] -= w.H[k * LD + tid] * xk;
            __syncthreads();
        }
        if (tid < q) sv.dp[tid] = -w.step[tid];
    }
    sens_sync<WAVE>();
    SENS_STAMP(2);
    if (tid < q) sv.rd[tid] = lds_dot(w.R + tid, LD, sv.dp, 1, q);
    sens_mm<T, WAVE, true, false, true>(tid, q, LD, w.R, w.R, Tb, sv.dp);   // Tb = R^T diag(dp) R
    sens_sync<WAVE>();
    SENS_STAMP(3);
    for (int e = tid; e < q * q; e += TT) {               // Z = ZS + Phi o (R^T dD R)/4
        const int a = row_of(e), b = e - a * q;
        const double acc = Tb[a * LD + b] - sv.rd[a] * w.rp[b] - w.rp[a] * sv.rd[b];
        Tc[a * LD + b] += acc * (sd.phiD ? sd.phiD[a * LD + b] : 0.5 * sdp_rcp(w.sq[a] + w.sq[b]));
    }
    sens_sync<WAVE>();
    SENS_STAMP(4);
    if constexpr (YFORM) {
        sens_mm<T, WAVE, false, false, false>(tid, q, LD, U, Tc, Tb);        // Tb = Y = U Z
        if (tid < q) sv.t2[tid] = lds_dot(Tc + tid * LD, 1, sv.t1, 1, q);     // Z t1
        sens_sync<WAVE>();
    } else {
        sens_mm<T, WAVE, false, true, false>(tid, q, LD, Tc, U, Tb);         // Tb = Z U^T
        if (tid < q) sv.t2[tid] = lds_dot(Tc + tid * LD, 1, sv.t1, 1, q);     // Z t1
        sens_sync<WAVE>();
        sens_mm<T, WAVE, false, false, false>(tid, q, LD, U, Tb, Ta);        // Ta = dN = U Z U^T
    }
    SENS_STAMP(5);
    // dc = dmu + (dB p + B dp)/2 = dmu - 2 Rs (Z t1) + R (sq o rd)
    if (tid < q) sv.dc[tid] = sv.dmu[tid] + lds_dot_rs(w.R + tid * LD, w.sq, sv.rd, 1, sv.t2, q);
    sens_sync<WAVE>();
    if (tid < q) sv.t2[tid] = lds_dot(U + tid, LD, sv.dc, 1, q) * sdp_rcp(w.sq[tid]);   // (U^T dc)/sq
    sens_sync<WAVE>();
    if (tid < q) {
        sv.w1[tid] = 0.5 * lds_dot(U + tid * LD, 1, sv.t2, 1, q);      // N dc
        sv.w2[tid] = YFORM ? lds_dot(Tb + tid * LD, 1, sd.utc, 1, q)    // dN c = Y (U^T c)
                           : lds_dot(Ta + tid * LD, 1, sv.c, 1, q);    // dN c
    }
    sens_sync<WAVE>();
    SENS_STAMP(6);
    double dnu = 0.0;
    for (int i = 0; i < q; ++i) dnu += sv.c[i] * (2.0 * sv.w1[i] + sv.w2[i]);
    return dnu;
}

// A q-term sum per row for a 256-thread workgroup with q <= 64: threads 4 i .. 4 i + 3 share row i
// (terms k = part, part + 4, ...) and add their parts with two quad exchanges; every thread of the
// workgroup calls it, rows past q compute on row 0 and are discarded by the caller.
template <class F>
__device__ __forceinline__ double quad_row_sum(int q, F term) {
    const int part = threadIdx.x & 3;
    double acc = 0.0;
#pragma unroll 4
    for (int k = part; k < q; k += 4) acc += term(k);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    return acc;
}

// The same direction for a whole workgroup when LDS is short (q > 32 in the Hessian kernel): R is
// not stored but expressed through U, R = diag(1/p) U + 1 rp^T, so that four q x q matrices
// suffice: U (w.S), the factor L of -hess g (w.H) and two scratch matrices (w.X, w.V).  With
// t1 = sq o rp and M1' = ZS diag(sq) U^T:
//   ZS diag(sq) R^T = M1' diag(1/p) + (ZS t1) 1^T,
//   R^T diag(dp) R  = U^T diag(dp / p^2) U + h rp^T + rp h^T + (sum dp) rp rp^T,  h = U^T (dp / p),
//   rd = R^T dp = h + (sum dp) rp,   so  R^T dD R = U^T diag(dp / p^2) U - (sum dp) rp rp^T.
// In: sv.dmu and the perturbation in rank-2 form dS = e_a v'^T + v' e_a^T through wv = U^T v'
// (in LDS; overwritten).  Like the YFORM of sens_direction it stops at Y = U Z: out Y in w.H,
// sv.w1 = N dc, sv.w2 = dN c = Y utc (utc = U^T c), sv.dp.
// Needs w.ip = 1 / p, w.sq, w.rp, sv.c, sv.t1 and, in registers, the thread's share of L^-1 (L the
// factor of -hess g, L^-1 lower triangular): thread (row, part) = (tid >> 2, tid & 3) holds, for
// k = part + 4 j, lm[j] = L^-1[row][k] where k <= row and L^-1[k][row] where k > row -- its pieces
// of row `row` and of column `row` in one array, the diagonal entry serving both -- zeros past q
// (sens_linv_regs).  scl is [q] scratch.
// Three q x q matrices of LDS: U (w.S) and the two scratch matrices w.X and w.H.
__device__ __forceinline__ void sens_linv_regs(const lds_f64* Li, int q, int LD, double (&lm)[16]) {
    const int ri = threadIdx.x >> 2, part = threadIdx.x & 3, row = ri < q ? ri : 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = part + 4 * j, kc = k < q ? k : 0;
        const double a = Li[row * LD + kc], b = Li[kc * LD + row];
        lm[j] = k >= q ? 0.0 : (k <= row ? a : b);
    }
}
// sum_k m[j(k)] x[k] over the thread's k = part, part + 4, ... on one side of the diagonal (LOWER:
// k <= row, the product with row `row` of L^-1; else k >= row, the product with its column) and
// over the four threads of the row
template <bool LOWER>
__device__ __forceinline__ double quad_reg_sum(int q, const double (&m)[16], const lds_f64* x) {
    const int part = threadIdx.x & 3, ri = threadIdx.x >> 2, row = ri < q ? ri : 0;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        if (4 * j < q) {
            const int k0 = part + 4 * j, k1 = k0 + 4;
            const double x0 = x[k0 < q ? k0 : 0], x1 = x[k1 < q ? k1 : 0];
            const bool on0 = LOWER ? k0 <= row : k0 >= row, on1 = LOWER ? k1 <= row : k1 >= row;
            acc0 = fma(on0 ? m[j] : 0.0, x0, acc0);              // m is zero where k >= q
            acc1 = fma(on1 ? m[j + 1] : 0.0, x1, acc1);
        }
    }
    double acc = acc0 + acc1;
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    return acc;
}
template <int T>
__device__ __forceinline__ void sens_direction_lean(const SdpWork w, const SensVec sv, int a_dir,
                                                    lds_f64* wv, lds_f64* scl, const lds_f64* utc,
                                                    const double (&lm)[16]
#ifdef OEI_HESS_TIMING
                                                    , long long (&sens_cyc)[12]
#endif
                                                    ) {
#ifdef OEI_HESS_TIMING
    long long sens_t0 = clock64();
#endif
    const int tid = threadIdx.x;
    const int q = w.q, LD = w.LD;
    const lds_f64* U = w.S;
    lds_f64* Tc = w.X;
    lds_f64* Tb = w.H;
    // the q-term sums: four threads per row (T = 256 >= 4 q)
    static_assert(T == 256, "four threads per row need 4 q <= T for every q <= 64");
    const int ri = tid >> 2, row = ri < q ? ri : 0;
    const bool rown = ri < q && (tid & 3) == 0;              // the thread that stores row ri's result
    __syncthreads();
    {
        const lds_f64* ua = U + a_dir * LD;
        for (int e = tid; e < q * q; e += T) {                 // ZS
            const int a = e / q, b = e - a * q;
            const double sa = w.sq[a], sb = w.sq[b];
            Tc[a * LD + b] = (ua[a] * wv[b] + wv[a] * ua[b]) * (-0.5 * sdp_rcp(sa * sb * (sa + sb)));
        }
    }
    __syncthreads();
    SENS_STAMP(0);
    wg4_mm_mfma<false, true, true>(q, LD, Tc, U, Tb, w.sq);    // Tb = M1' = ZS diag(sq) U^T
    {
        const double v = quad_row_sum(q, [&](int k) { return Tc[row * LD + k] * sv.t1[k]; });
        if (rown) sv.t2[ri] = v;
    }
    __syncthreads();
    SENS_STAMP(1);
    {
        // rhs_i = dmu_i + sum_a R[i][a] sq[a] (M1[a][i] - 2 t2[a]),  M1[a][i] = M1'[a][i] / p_i + t2[a]
        const double ipi = w.ip[row];
        const double acc = quad_row_sum(q, [&](int a) {
            const double r = fma(U[row * LD + a], ipi, w.rp[a]);
            return r * w.sq[a] * fma(ipi, Tb[a * LD + row], -sv.t2[a]);
        });
        if (rown) w.step[ri] = sv.dmu[ri] + acc;
    }
    __syncthreads();
    // (-hess g) x = rhs through L^-1: two matrix-vector products, four threads per row, the
    // matrix entries in registers.  (A substitution with L costs 2 q dependent steps in one wave,
    // each a broadcast through the scalar unit, or two barriers per component across the workgroup.)
    {
        const double y = quad_reg_sum<true>(q, lm, w.step);                                  // y = L^-1 rhs
        if (rown) sv.rd[ri] = y;
    }
    __syncthreads();
    {
        const double x = quad_reg_sum<false>(q, lm, sv.rd);                                   // L^-T y
        if (rown) {
            const double dp = -x, ipi = w.ip[ri];
            sv.dp[ri] = dp;
            scl[ri] = dp * ipi * ipi;
            wv[ri] = dp * ipi;                             // wv is free again
        }
    }
    __syncthreads();
    SENS_STAMP(2);
    // sum of dp: every quad adds all q entries (a thread walking them alone: q dependent reads)
    const double sdp = quad_row_sum(q, [&](int k) { return sv.dp[k]; });
    {
        const double h = quad_row_sum(q, [&](int k) { return U[k * LD + row] * wv[k]; });
        if (rown) sv.rd[ri] = h + w.rp[ri] * sdp;
    }
    wg4_mm_mfma<true, false, true>(q, LD, U, U, Tb, scl);      // Tb = U^T diag(dp / p^2) U
    __syncthreads();
    SENS_STAMP(3);
    for (int e = tid; e < q * q; e += T) {                     // Z = ZS + Phi o (R^T dD R) / 4
        const int a = e / q, b = e - a * q;
        const double acc = Tb[a * LD + b] - sdp * w.rp[a] * w.rp[b];
        Tc[a * LD + b] += acc * (0.5 * sdp_rcp(w.sq[a] + w.sq[b]));
    }
    __syncthreads();
    SENS_STAMP(4);
    wg4_mm_mfma<false, false, false>(q, LD, U, Tc, Tb);        // Tb = Y = U Z
    {
        const double v = quad_row_sum(q, [&](int k) { return Tc[row * LD + k] * sv.t1[k]; });      // Z t1
        if (rown) sv.t2[ri] = v;
    }
    __syncthreads();
    SENS_STAMP(5);
    {
        // dc = dmu + sum_a R[i][a] sq[a] (rd[a] - 2 t2[a])
        const double ipi = w.ip[row];
        const double acc = quad_row_sum(q, [&](int a) {
            const double r = fma(U[row * LD + a], ipi, w.rp[a]);
            return r * w.sq[a] * fma(-2.0, sv.t2[a], sv.rd[a]);
        });
        if (rown) sv.dc[ri] = sv.dmu[ri] + acc;
    }
    __syncthreads();
    {
        const double v = quad_row_sum(q, [&](int k) { return U[k * LD + row] * sv.dc[k]; });
        if (rown) wv[ri] = v * sdp_rcp(w.sq[ri]);            // (U^T dc) / sq
    }
    __syncthreads();
    {
        const double v1 = quad_row_sum(q, [&](int k) { return U[row * LD + k] * wv[k]; });
        const double v2 = quad_row_sum(q, [&](int k) { return Tb[row * LD + k] * utc[k]; });
        if (rown) {
            sv.w1[ri] = 0.5 * v1;                             // N dc
            sv.w2[ri] = v2;                                   // dN c = Y (U^T c)
        }
    }
    __syncthreads();
    SENS_STAMP(6);
}
