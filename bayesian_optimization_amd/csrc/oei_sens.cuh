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

// One direction.  In: sv.dmu, dS (symmetric) in w.X.  Out: dN in w.X, sv.w1 = N dc, sv.w2 = dN c,
// sv.dp, Z in w.Kt; returns dnu.  dM11 = -dN, dM[:q,q] = w1 + w2, dM[q][q] = -dnu.
// Needs w.rc = 1 / diag(L) of the factor in w.H (sens_prepare).
template <int T>
__device__ double sens_direction(const SdpWork w, const SensVec sv) {
    const int tid = threadIdx.x;
    const int q = w.q, LD = w.LD;
    const lds_f64* U = w.S;
    lds_f64* Ta = w.X;
    lds_f64* Tb = w.V;
    lds_f64* Tc = w.Kt;
    __syncthreads();
    lds_mm<T, false, false>(q, LD, Ta, U, Tb);            // Tb = dS U
    __syncthreads();
    lds_mm<T, true, false>(q, LD, U, Tb, Tc);             // Tc = U^T dS U
    __syncthreads();
    for (int e = tid; e < q * q; e += T) {                // ZS
        const int a = e / q, b = e - a * q;
        const double sa = w.sq[a], sb = w.sq[b];
        Tc[a * LD + b] *= -0.5 * sdp_rcp(sa * sb * (sa + sb));
    }
    __syncthreads();
    // rhs = dmu - (dBS_ii - 2 (dBS p)_i)/4 with dBS = -4 Rs ZS Rs^T, Rs = R diag(sq):
    // M1 = ZS Rs^T (into Tb), t2 = ZS t1, rhs_i = dmu_i + sum_a Rs[i][a] (M1[a][i] - 2 t2[a])
    lds_mm<T, false, true, true>(q, LD, Tc, w.R, Tb, w.sq);
    if (tid < q) {
        double acc = 0.0;
        for (int b = 0; b < q; ++b) acc += Tc[tid * LD + b] * sv.t1[b];
        sv.t2[tid] = acc;
    }
    __syncthreads();
    double rhs = 0.0;
    if (tid < q) {
        double acc = 0.0;
#pragma unroll 4
        for (int a = 0; a < q; ++a) acc += w.R[tid * LD + a] * w.sq[a] * (Tb[a * LD + tid] - 2.0 * sv.t2[a]);
        rhs = sv.dmu[tid] + acc;
    }
    // A x = rhs with A = L L^T = -hess g
    if constexpr (T == 64) {
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
    __syncthreads();
    if (tid < q) {
        double acc = 0.0;
        for (int i = 0; i < q; ++i) acc += w.R[i * LD + tid] * sv.dp[i];
        sv.rd[tid] = acc;
    }
    lds_mm<T, true, false, true>(q, LD, w.R, w.R, Tb, sv.dp);   // Tb = R^T diag(dp) R
    __syncthreads();
    for (int e = tid; e < q * q; e += T) {                // Z = ZS + Phi o (R^T dD R)/4
        const int a = e / q, b = e - a * q;
        const double acc = Tb[a * LD + b] - sv.rd[a] * w.rp[b] - w.rp[a] * sv.rd[b];
        Tc[a * LD + b] += 0.5 * acc * sdp_rcp(w.sq[a] + w.sq[b]);
    }
    __syncthreads();
    lds_mm<T, false, true>(q, LD, Tc, U, Tb);             // Tb = Z U^T
    if (tid < q) {
        double acc = 0.0;
        for (int b = 0; b < q; ++b) acc += Tc[tid * LD + b] * sv.t1[b];
        sv.t2[tid] = acc;                                 // Z t1
    }
    __syncthreads();
    lds_mm<T, false, false>(q, LD, U, Tb, Ta);            // Ta = dN = U Z U^T
    if (tid < q) {        // dc = dmu + (dB p + B dp)/2 = dmu - 2 Rs (Z t1) + R (sq o rd)
        double acc = 0.0;
        for (int a = 0; a < q; ++a) acc += w.R[tid * LD + a] * w.sq[a] * (sv.rd[a] - 2.0 * sv.t2[a]);
        sv.dc[tid] = sv.dmu[tid] + acc;
    }
    __syncthreads();
    if (tid < q) {        // t2 = (U^T dc)/sq
        double acc = 0.0;
        for (int i = 0; i < q; ++i) acc += U[i * LD + tid] * sv.dc[i];
        sv.t2[tid] = acc / w.sq[tid];
    }
    __syncthreads();
    if (tid < q) {
        double a1 = 0.0, a2 = 0.0;
        for (int k = 0; k < q; ++k) {
            a1 += U[tid * LD + k] * sv.t2[k];
            a2 += Ta[tid * LD + k] * sv.c[k];
        }
        sv.w1[tid] = 0.5 * a1;                            // N dc
        sv.w2[tid] = a2;                                  // dN c
    }
    __syncthreads();
    double dnu = 0.0;
    for (int i = 0; i < q; ++i) dnu += sv.c[i] * (2.0 * sv.w1[i] + sv.w2[i]);
    return dnu;
}
