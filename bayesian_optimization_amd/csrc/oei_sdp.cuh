// Device-side solver for the OEI semidefinite program, one workgroup per candidate.
//
//   value = max_M <Omega, M>  s.t.  C_i - M >= 0,  i = 0..q        (reference: sdp.py:11-67)
//
// The reference hands this cone program (k_ PSD blocks of side k_ = q+1) to a first-order solver.
// Here the k_ Schur complements are eliminated analytically.  With
// Omega = w0 [[S + mu mu^T, mu], [mu^T, 1]], delta = mu - fmin, the problem is equivalent to the
// smooth concave maximisation over the open simplex {p > 0, sum p < 1}
//
//   g(p) = tr (S^1/2 D(p) S^1/2)^1/2 - p^T delta,     D(p) = diag(p) - p p^T,     value = -w0 max g,
//
// where p_i is the weight of the i-th rank-one dual block Y_i (p_0 = 1 - sum p that of Y_0).
// g is maximised by a damped Newton iteration.  Every evaluation needs the singular values and
// right singular vectors of a q x q matrix G with G^T G = X = W S W^T (D = W^T W):
//
//   W = diag(s) (I - gamma 1 p^T),  s = sqrt(p),  gamma = 1/(1 + sqrt(p0)),  S = Lc Lc^T
//   G = Lc^T W^T,  G V = (orthogonal columns of norms sq),  U = W^T V = Lc^-T (G V),
//   R = W^-1 V = D^-1 U = U / p + 1 rp^T,  rp = R^T p = (1^T U) / p0
//   B = 2 R diag(sq) R^T,   N = B^-1 = 1/2 U diag(1/sq) U^T
//   grad g = diag(B)/4 - B p/2 - delta  =  (sum_a sq_a U_ia^2) / (2 p_i^2) - (sum_a sq_a rp_a^2)/2 - delta_i
//   hess g = -1/2 sum_ab K_ab E_i,ab E_j,ab - B/2,   K_ab = 1/(sq_a sq_b (sq_a + sq_b)),
//            E_i = w_i w_i^T - m m^T,  w_i = sq o (R[i,:] - rp) = sq o U[i,:] / p_i,  m = sq o rp
//   M = -[[N, -N c], [-(N c)^T, c^T N c]],  c = mu + B p / 2
//   y_0 = sqrt(w0 p0) [c; 1],  y_i = sqrt(w0 p_i) [c - B e_i / 2; 1],  Y_i = y_i y_i^T
//
// Two implementations:
//   * register path (q <= Q <= 32, one 64-lane wave, template parameter Q > 0): one-sided
//     (Hestenes) Jacobi with the columns of G in registers (a pair of columns per lane, odd-even
//     transposition ordering, columns move between lanes with DPP shifts); no LDS traffic and
//     no eigenvector accumulation (U follows from a triangular solve with Lc).  The previous U
//     seeds the next evaluation (G <- G V_prev), which saves sweeps as the Newton steps shrink.
//     Everything the Newton step
//     needs is expressed through U, so three q x q matrices of LDS suffice (Lc, U and one that
//     holds the Hessian in its lower and the Loewner kernel in its upper triangle): the small
//     footprint is what lets ten and more candidates share a compute unit.
//   * wide path (larger q, 256 threads, Q = 0): the same iteration with the columns of G in LDS;
//     groups of 8 threads own a pair of columns per round and reduce with DPP shuffles.  Same
//     state and the same three matrices.
// All matrices are q x q with an odd leading dimension LD so that row and column walks are free
// of LDS bank conflicts.  Every thread of the workgroup executes every call (barriers inside).
#pragma once
#include <climits>
#include "oei_common.h"

#ifndef SDP_OE_UNROLL
#define SDP_OE_UNROLL 2
#endif
#define SDP_MAX_NEWTON 60
#define SDP_MAX_SWEEPS 30
#ifndef SDP_JACOBI_TOL
#define SDP_JACOBI_TOL 1e-15
#endif
// Stop after a full Newton step of relative size below this.  The iteration converges
// quadratically with a constant near one (relative steps 4e-2, 1e-3, 1e-6, 5e-13 on the
// benchmark inputs), so the point reached is within about 1e-11 of the optimum; a tolerance of
// 1e-7 costs one more direction and evaluation whose step is at rounding level.
#define SDP_STEP_TOL 3e-6
#define SDP_DECREMENT_TOL 1e-19
#ifndef SDP_JACOBI_FINAL
#define SDP_JACOBI_FINAL 1e-7
#endif
// evaluations that only steer the Newton iteration (every one but the last) stop earlier
#ifndef SDP_JACOBI_TOL_LOOSE
#define SDP_JACOBI_TOL_LOOSE 1e-11
#endif
#ifndef SDP_JACOBI_FINAL_LOOSE
#define SDP_JACOBI_FINAL_LOOSE 1e-5
#endif
#define SDP_EVAL_WARM 1
#define SDP_EVAL_LOOSE 2
// the very first evaluation feeds a step of several per cent: it may stop a sweep earlier
#define SDP_EVAL_FIRST 4
#ifndef SDP_JACOBI_FINAL_FIRST
#define SDP_JACOBI_FINAL_FIRST 1e-3
#endif

// Inside one scheduling region: issue the LDS reads (up to nr) back to back, then the arithmetic,
// and close the region.  Without this the register-pressure heuristic of the scheduler tends to
// emit read / wait / use triples, which exposes the full LDS latency on every element.
#define SDP_READS_FIRST(nr, nv) do { \
        __builtin_amdgcn_sched_group_barrier(0x100, (nr), 0); \
        __builtin_amdgcn_sched_group_barrier(0x002, (nv), 0); \
        __builtin_amdgcn_sched_barrier(0); } while (0)

// Pointers into LDS carry their address space in the type: pointers that pass through a struct
// would otherwise degrade to generic (flat) accesses, which are several times slower than ds_*.
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;

// Diagnostic build only (-DOEI_SDP_TIMING): cycle counters per solver phase in w.scal[0..6]
// (0 set-up of an evaluation, 1 Jacobi, 2 rest of the evaluation, 3 Newton direction, of which
// 4 Hessian assembly, 5 Cholesky and solves).
#ifdef OEI_SDP_TIMING
#define SDP_TIC(var) const long long var = clock64()
#define SDP_TOC(slot, var) do { if (threadIdx.x == 0) w.scal[slot] += (double)(clock64() - (var)); } while (0)
#else
#define SDP_TIC(var)
#define SDP_TOC(slot, var)
#endif

// The struct is passed BY VALUE everywhere: a reference would live in private memory and every
// LDS store would force the pointers to be reloaded from there.
struct SdpWork {
    int q, LD;
    lds_f64 *H, *S, *X;                   // q x LD matrices, both layouts
    lds_f64 *V, *R, *Kt;                  // full layout only (sensitivity kernels)
    lds_f64 *Li;                          // Lc^-T as a full matrix (instances with sdp_mm<Q>() only)
    lds_f64 *p, *pn, *s, *dl, *sq, *grad, *step, *tv, *sv, *rp, *ci, *ip, *wp;   // vectors
    lds_f64 *rc, *rs;                     // 1/diag(Lc) | 1/s
    lds_f64 *scal;                        // [8] scalars
};
#define SDP_NVEC 15

__host__ __device__ inline int sdp_ld(int q) { return q | 1; }
__host__ __device__ inline int sdp_vlen(int q) { return (q + 3) & ~1; }   // even, >= q + 2

// Instances of the register path that run their q x q products on the matrix pipe (template
// parameter Q of the kernels; the host picks such an instance for every q <= 20): they keep
// Lc^-T in a fourth q x q matrix.
template <int Q> __host__ __device__ constexpr bool sdp_mm() { return Q > 0 && Q <= 20; }
__host__ __device__ inline bool sdp_mm_q(int q) { return q <= 20; }

__host__ __device__ inline size_t sdp_work_doubles(int q, bool full, bool mm = false) {
    const int LD = sdp_ld(q), VL = sdp_vlen(q);
    return (size_t)((full ? 6 : 3) + (mm ? 1 : 0)) * q * LD + (size_t)SDP_NVEC * VL + 8;
}

// Callers may use everything behind w.H (starting at w.S) as a staging buffer of `stage` doubles
// while no solver state other than w.H is live.  Size of the region that must be reserved:
__host__ __device__ inline size_t sdp_region_doubles(int q, size_t stage, bool full, bool mm = false) {
    const size_t h = (size_t)q * sdp_ld(q), rest = sdp_work_doubles(q, full, mm) - h;
    return h + (rest > stage ? rest : stage);
}

__device__ inline void sdp_carve(SdpWork& w, double* base, int q, bool full, bool mm = false) {
    const int LD = sdp_ld(q), VL = sdp_vlen(q);
    w.q = q; w.LD = LD;
    lds_f64* ptr = (lds_f64*)base;
    w.H = ptr; ptr += q * LD;  w.S = ptr; ptr += q * LD;  w.X = ptr; ptr += q * LD;
    w.V = w.R = w.Kt = nullptr;
    if (full) { w.V = ptr; ptr += q * LD;  w.R = ptr; ptr += q * LD;  w.Kt = ptr; ptr += q * LD; }
    w.Li = nullptr;
    if (mm) { w.Li = ptr; ptr += q * LD; }
    w.p = ptr; ptr += VL;  w.pn = ptr; ptr += VL;  w.s = ptr; ptr += VL;  w.dl = ptr; ptr += VL;
    w.sq = ptr; ptr += VL; w.grad = ptr; ptr += VL; w.step = ptr; ptr += VL; w.tv = ptr; ptr += VL;
    w.sv = ptr; ptr += VL; w.rp = ptr; ptr += VL;  w.ci = ptr; ptr += VL;  w.ip = ptr; ptr += VL;
    w.wp = ptr; ptr += VL; w.rc = ptr; ptr += VL;  w.rs = ptr; ptr += VL;
    w.scal = ptr;
}

// R[i][a] from the state both paths leave behind (U in w.X, 1/p in w.ip, rp)
__device__ __forceinline__ double sdp_R(const SdpWork w, int i, int a) {
    return fma(w.X[i * w.LD + a], w.ip[i], w.rp[a]);
}

// Reciprocal and reciprocal square root from the hardware seeds (about 2^-23 relative) plus two
// Newton steps: within a few ulp, a third of the instructions of the correctly rounded forms.
// Arguments here are positive and far from the ends of the exponent range.
__device__ __forceinline__ double sdp_rcp(double a) {
    double x = __builtin_amdgcn_rcp(a);
    double e = fma(-a, x, 1.0);
    x = fma(x, e, x);
    e = fma(-a, x, 1.0);
    return fma(x, e, x);
}
__device__ __forceinline__ double sdp_rsq(double a) {
    double y = __builtin_amdgcn_rsq(a);
    double e = fma(-(a * y), y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-(a * y), y, 1.0);
    return fma(0.5 * y, e, y);
}
template <int NR>
__device__ __forceinline__ double sdp_rsq_steps(double a) {     // hardware seed (2^-23) plus NR Newton steps
    double y = __builtin_amdgcn_rsq(a);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const double e = fma(-(a * y), y, 1.0);
        y = fma(0.5 * y, e, y);
    }
    return y;
}
// value of lane k (k uniform) in every lane, through the scalar unit instead of the LDS crossbar
__device__ __forceinline__ double sdp_lane_bcast(double v, int k) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}
// index pair (i >= j) of the e-th entry of a row-major lower triangle
__device__ __forceinline__ void sdp_tri_index(int e, int& i, int& j) {
    i = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
    i += ((i + 1) * (i + 2) / 2 <= e) ? 1 : 0;
    i -= (i * (i + 1) / 2 > e) ? 1 : 0;
    j = e - i * (i + 1) / 2;
}

// ================================================================================================
// Wide path (several waves per candidate, any q <= 64): the same algorithm as the register path
// with the columns of G in LDS.  State and conventions are those of the register path (Lc in w.S,
// U in w.X, w.H = exchange buffer / Newton matrix), so three q x q matrices suffice here too.
// ================================================================================================
// value of the lane given by a DPP control word, for a double
template <int CTRL>
__device__ __forceinline__ double sdp_dpp(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// Sum over the 8 lanes of an aligned group; every lane of the group gets the same bits (each
// butterfly step adds the same two numbers in both partners).  All 64 lanes must be active.
__device__ __forceinline__ double sdp_sum8(double v) {
    v += sdp_dpp<0xB1>(v);        // quad_perm [1,0,3,2]
    v += sdp_dpp<0x4E>(v);        // quad_perm [2,3,0,1]
    v += sdp_dpp<0x141>(v);       // row_half_mirror
    return v;
}

// Sum / maximum over lanes 0..31 of a wave, identical bits in every lane (butterfly inside the two
// 16-lane rows, then the two row results through the scalar unit).  All 64 lanes must be active.
__device__ __forceinline__ double sdp_sum32(double v) {
    v = sdp_sum8(v);
    v += sdp_dpp<0x140>(v);       // row_mirror
    return sdp_lane_bcast(v, 0) + sdp_lane_bcast(v, 16);
}
__device__ __forceinline__ double sdp_max32(double v) {
    v = fmax(v, sdp_dpp<0xB1>(v));
    v = fmax(v, sdp_dpp<0x4E>(v));
    v = fmax(v, sdp_dpp<0x141>(v));
    v = fmax(v, sdp_dpp<0x140>(v));
    return fmax(sdp_lane_bcast(v, 0), sdp_lane_bcast(v, 16));
}

// In-place Cholesky of the lower triangle of A (q x q, leading dimension LD) by T threads.
// Returns 1 at the first non-positive pivot (identical in every thread).
template <int T>
__device__ int sdp_chol_lds(lds_f64* A, int q, int LD) {
    const int tid = threadIdx.x;
    for (int k = 0; k < q; ++k) {
        const double akk = A[k * LD + k];
        if (!(akk > 0.0)) return 1;
        const double rinv = sdp_rsq(akk);
        __syncthreads();
        if (tid == 0) A[k * LD + k] = akk * rinv;
        for (int i = k + 1 + tid; i < q; i += T) A[i * LD + k] *= rinv;
        __syncthreads();
        const int rem = q - k - 1;
        for (int e = tid; e < rem * rem; e += T) {
            const int i = k + 1 + e / rem, j = k + 1 + e % rem;
            if (j <= i) A[i * LD + j] -= A[i * LD + k] * A[j * LD + k];
        }
        __syncthreads();
    }
    return 0;
}

// S = Lc Lc^T in place: Lc in the lower triangle of w.S, the strict upper triangle receives
// Lc^-T (row i: entries r > i) and w.rc the reciprocal diagonal, so that the back substitution
// U = Lc^-T (G V) of every evaluation becomes a product that all threads share.
template <int T>
__device__ int sdp_factor_wide(const SdpWork w) {
    const int tid = threadIdx.x;
    const int q = w.q, LD = w.LD;
    __syncthreads();
    if (sdp_chol_lds<T>(w.S, q, LD)) return 1;
    // thread j: column j of Lc^-1 by forward substitution, kept as row j of the upper triangle;
    // a thread reads the factor and its own row only
    if (tid < q) {
        const int j = tid;
        const double xj = sdp_rcp(w.S[j * LD + j]);
        w.rc[j] = xj;
        for (int i = j + 1; i < q; ++i) {
            double acc = w.S[i * LD + j] * xj;
            for (int k = j + 1; k < i; ++k) acc += w.S[i * LD + k] * w.S[j * LD + k];
            w.S[j * LD + i] = -acc * sdp_rcp(w.S[i * LD + i]);
        }
    }
    __syncthreads();
    return 0;
}

// One-sided (Hestenes) Jacobi on the q columns of G, column j at Gc[j * LD + r].  Groups of 8
// threads own one pair of columns per round (round-robin pairing: in round r, slot 0 pairs
// (qe-1, r), slot k pairs (r+k, r-k) mod (qe-1); qe = q rounded up to even, a pair that touches
// the padding index idles); thread s of a group holds the rows s, s+8, ... of both columns.
template <int T>
__device__ int sdp_jacobi_cols(lds_f64* Gc, int q, int LD, double tol2, double fin2) {
    constexpr int NG = T / 8, MR = 8;           // rows per thread: q <= 64
    const int tid = threadIdx.x, grp = tid >> 3, s = tid & 7;
    const int qe = q + (q & 1), np = qe >> 1, nm1 = qe - 1;
    int sweep = 0;
    for (; sweep < SDP_MAX_SWEEPS; ++sweep) {
        int rotated = 0;
        for (int r = 0; r < nm1; ++r) {
            for (int k0 = 0; k0 < np; k0 += NG) {
                const int k = k0 + grp;
                int a = nm1, b = r;
                if (k > 0) { a = r + k; if (a >= nm1) a -= nm1; b = r - k; if (b < 0) b += nm1; }
                const int i = a < b ? a : b, j = a < b ? b : a;
                const bool active = (k < np) && (j < q);
                lds_f64* ci = Gc + (active ? i : 0) * LD;
                lds_f64* cj = Gc + (active ? j : 0) * LD;
                double gi[MR], gj[MR];
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const int row = s + 8 * m;
                    const bool in = row < q;
                    const double x = ci[in ? row : 0], y = cj[in ? row : 0];
                    gi[m] = in ? x : 0.0;
                    gj[m] = in ? y : 0.0;
                }
                double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    al = fma(gi[m], gi[m], al);
                    be = fma(gj[m], gj[m], be);
                    ga = fma(gi[m], gj[m], ga);
                }
                al = sdp_sum8(al); be = sdp_sum8(be); ga = sdp_sum8(ga);
                const double g2 = ga * ga, aa = al * be;
                const bool rot = active && (g2 > tol2 * aa);
                if (rot) {
                    // with d = beta - alpha, h = sqrt(d^2 + 4 gamma^2):  c^2 = (h + |d|) / (2 h),
                    // s = sign(d) gamma / (h c)
                    const double d = be - al;
                    const double hinv = sdp_rsq(fma(d, d, 4.0 * g2));
                    const double v = fma(0.5 * fabs(d), hinv, 0.5);
                    const double rv = sdp_rsq(v);
                    const double c = v * rv;
                    double sg = ga * hinv * rv;
                    if (d < 0.0) sg = -sg;
                    rotated |= (g2 > fin2 * aa) ? 1 : 0;
#pragma unroll
                    for (int m = 0; m < MR; ++m) {
                        const int row = s + 8 * m;
                        if (row < q) {
                            ci[row] = c * gi[m] - sg * gj[m];
                            cj[row] = sg * gi[m] + c * gj[m];
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) { ++sweep; break; }
    }
    return sweep;
}

template <int NR>
__device__ int sdp_jacobi_rows4(lds_f64* Gc, int q, int LD, double tol2, double fin2, lds_f64* red);

// The two q x q x q products of an evaluation for the 256-thread path on the matrix pipe:
//   P = 1:  G[r][a] = sum_{j >= r} Lc[j][r] Vp[j][a] - gam tv[r] grad[a]      (Vp in w.X; stored as column a of Gc)
//   P = 2:  U[i][a] = rc_i G[i][a] + sum_{r > i} LinvT[i][r] G[r][a]          (into w.X)
// Both are (upper triangular) x (full): wave w computes the 32 x 32 block (w >> 1, w & 1) as 2 x 2
// tiles of v_mfma_f64_16x16x4, from its block row's diagonal on; the second row tile skips the four
// steps that lie left of its own diagonal.  Lane pointers advance by a constant per step and the
// next step's reads are issued before this step's matrix instructions (a wave issues in order: with
// index arithmetic per step the reads, the arithmetic and the matrix instructions took turns).  The
// triangle and P = 2's diagonal are applied by selects on the left operand; rows and columns past q
// read row / column 0 and only reach entries that are not stored.  Out of line: inlined into the
// solver, whose registers are all taken, its 70 registers were spilled around every call.
template <int P>
static __device__ __noinline__ void sdp_wg4_tri_mm(const SdpWork w, lds_f64* Gc, double gam) {
    typedef double sdp_double4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int q = w.q, LD = w.LD;
    const int I0 = 2 * (wave >> 1), J0 = 2 * (wave & 1);
    if (16 * I0 >= q || 16 * J0 >= q) return;
    sdp_double4 acc[2][2];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) acc[I][J] = (sdp_double4){0.0, 0.0, 0.0, 0.0};
    const int kb = 16 * I0;                                  // first k position of the block row
    int ir[2];
    const lds_f64* pa[2];
    const lds_f64* pb[2];
    double dg[2];
#pragma unroll
    for (int I = 0; I < 2; ++I) {
        const int i = 16 * (I0 + I) + li, j = 16 * (J0 + I) + li;
        const int ic = i < q ? i : 0, jc = j < q ? j : 0;
        ir[I] = i;
        pa[I] = (P == 1) ? w.S + (kb + kg) * LD + ic : w.S + ic * LD + kb + kg;
        pb[I] = (P == 1) ? w.X + (kb + kg) * LD + jc : Gc + jc * LD + kb + kg;
        dg[I] = (P == 2) ? w.rc[ic] : 0.0;
    }
    const int sa = (P == 1) ? 4 * LD : 4, sb = sa;
    const int nfull = (q - kb) >> 2;                         // steps whose four k positions all lie inside q
    auto left = [&](int I, int k, double x) {                // the triangle (and P = 2's diagonal) of the left operand
        if (P == 1) return k >= ir[I] ? x : 0.0;
        return k > ir[I] ? x : (k == ir[I] ? dg[I] : 0.0);
    };
    double av[2], bv[2], an[2], bn[2];
#pragma unroll
    for (int I = 0; I < 2; ++I) { av[I] = left(I, kb + kg, pa[I][0]); bv[I] = pb[I][0]; }
    for (int s = 0; s < nfull; ++s) {
        const bool more = s + 1 < nfull;
        const int kn = kb + 4 * (s + 1) + kg;
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            pa[I] += more ? sa : 0; pb[I] += more ? sb : 0;
            an[I] = left(I, kn, pa[I][0]); bn[I] = pb[I][0];
        }
        // the second row tile starts at its own diagonal, four steps in
#pragma unroll
        for (int J = 0; J < 2; ++J) acc[0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[J], acc[0][J], 0, 0, 0);
        if (s >= 4) {
#pragma unroll
            for (int J = 0; J < 2; ++J) acc[1][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[J], acc[1][J], 0, 0, 0);
        }
#pragma unroll
        for (int I = 0; I < 2; ++I) { av[I] = an[I]; bv[I] = bn[I]; }
    }
    if ((q - kb) & 3) {
        const int k = kb + 4 * nfull + kg;
        const bool kin = k < q;
        const int kc = kin ? k : 0;
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            const int i = ir[I], j = 16 * (J0 + I) + li;
            const int ic = i < q ? i : 0, jc = j < q ? j : 0;
            const double x = (P == 1) ? w.S[kc * LD + ic] : w.S[ic * LD + kc];
            const double y = (P == 1) ? w.X[kc * LD + jc] : Gc[jc * LD + kc];
            av[I] = kin ? left(I, k, x) : 0.0;
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
                if (i < q && j < q) {
                    if (P == 1) Gc[j * LD + i] = acc[I][J][g] - gam * w.tv[i] * w.grad[j];
                    else w.X[i * LD + j] = acc[I][J][g];
                }
            }
}

// Evaluate g, grad g and the eigen-state at pt.  Needs w.S, w.rc from sdp_factor_wide.  With
// SDP_EVAL_WARM the previous state (U in w.X, 1/s in w.rs, gamma' p in w.wp, column sums in
// w.sv) seeds this one: G <- G V_prev.  Leaves U in w.X and the vectors sq, sv, rp, s, rs, ip,
// wp, grad.  Returns g, or -inf when pt leaves the simplex or the matrix loses rank numerically.
template <int T>
__device__ double sdp_eval_wide(const SdpWork w, const lds_f64* pt, int& sweeps, int mode) {
    const int tid = threadIdx.x;
    const int q = w.q, LD = w.LD;
    const int qb = (q + 1) >> 1;                  // 2 x 2 blocks per dimension
    SDP_TIC(t_setup);
    double sum = 0.0, pmin = INFINITY;
    for (int i = 0; i < q; ++i) { const double v = pt[i]; sum += v; pmin = fmin(pmin, v); }
    const double p0 = 1.0 - sum;
    if (!(p0 > 0.0) || !(pmin > 0.0)) return -INFINITY;
    const double rsp0 = sdp_rsq(p0), sp0 = p0 * rsp0, ip0 = rsp0 * rsp0;
    const double gam = sdp_rcp(1.0 + sp0), gamp = gam * rsp0;
    lds_f64* Gc = w.H;                            // free during an evaluation
    __syncthreads();                              // earlier readers of the vectors are done
    if (tid < q) {
        const double pj = pt[tid];
        const double sj = pj * sdp_rsq(pj);
        w.s[tid] = sj;
        double acc = 0.0;                         // m = Lc^T p
        for (int i = tid; i < q; ++i) acc += w.S[i * LD + tid] * pt[i];
        w.tv[tid] = acc;
        w.ci[tid] = sj * w.rs[tid];               // s_new / s_prev (warm start only)
    }
    __syncthreads();
    if (!(mode & SDP_EVAL_WARM)) {
        // column j of G = s_j (Lc^T e_j - gam m)
        for (int e = tid; e < q * q; e += T) {
            const int j = e / q, r = e - j * q;
            const double l = w.S[j * LD + r];
            Gc[j * LD + r] = w.s[j] * ((r <= j ? l : 0.0) - gam * w.tv[r]);
        }
        __syncthreads();
    } else {
        // Vp = diag(s_new / s_prev) (U_prev + wp sv^T) in place of U_prev, then
        // G V_prev = Lc^T Vp - gam m (1^T Vp)
        for (int e = tid; e < q * q; e += T) {
            const int j = e / q, a = e - j * q;
            w.X[j * LD + a] = w.ci[j] * fma(w.wp[j], w.sv[a], w.X[j * LD + a]);
        }
        __syncthreads();
        if (tid < q) {
            double acc = 0.0;
            for (int j = 0; j < q; ++j) acc += w.X[j * LD + tid];
            w.grad[tid] = acc;                    // scratch: the gradient is written at the end
        }
        __syncthreads();
        if constexpr (T == 256) sdp_wg4_tri_mm<1>(w, Gc, gam);
        else
        for (int blk = tid; blk < qb * qb; blk += T) {
            const int rb = blk / qb, ab = blk - rb * qb;
            const int r0 = 2 * rb, a0 = 2 * ab;
            const int r1 = min(r0 + 1, q - 1), a1 = min(a0 + 1, q - 1);
            // j = r0: only row r0 of the block sees Lc[r0][r0]
            const double l00 = w.S[r0 * LD + r0];
            double c00 = l00 * w.X[r0 * LD + a0], c01 = l00 * w.X[r0 * LD + a1], c10 = 0.0, c11 = 0.0;
#pragma unroll 4
            for (int j = r0 + 1; j < q; ++j) {
                const double l0 = w.S[j * LD + r0], l1 = w.S[j * LD + r1];
                const double v0 = w.X[j * LD + a0], v1 = w.X[j * LD + a1];
                c00 = fma(l0, v0, c00); c01 = fma(l0, v1, c01);
                c10 = fma(l1, v0, c10); c11 = fma(l1, v1, c11);
            }
            const double m0 = gam * w.tv[r0], m1 = gam * w.tv[r1];
            const double s0 = w.grad[a0], s1 = w.grad[a1];
            Gc[a0 * LD + r0] = c00 - m0 * s0;
            if (a0 + 1 < q) Gc[a1 * LD + r0] = c01 - m0 * s1;
            if (r0 + 1 < q) {
                Gc[a0 * LD + r1] = c10 - m1 * s0;
                if (a0 + 1 < q) Gc[a1 * LD + r1] = c11 - m1 * s1;
            }
        }
        __syncthreads();
    }
    SDP_TOC(0, t_setup);
    SDP_TIC(t_jac);
    {
        const bool loose = (mode & SDP_EVAL_LOOSE) != 0;
        const double tol = loose ? SDP_JACOBI_TOL_LOOSE : SDP_JACOBI_TOL;
        const double fin = (mode & SDP_EVAL_FIRST) ? SDP_JACOBI_FINAL_FIRST : loose ? SDP_JACOBI_FINAL_LOOSE : SDP_JACOBI_FINAL;
        if constexpr (T == 256) {
            // columns in registers, w.X (free until U is written below) as the reduction buffer
            sweeps += loose ? sdp_jacobi_rows4<1>(Gc, q, LD, tol * tol, fin * fin, w.X)
                            : sdp_jacobi_rows4<2>(Gc, q, LD, tol * tol, fin * fin, w.X);
        } else {
            sweeps += sdp_jacobi_cols<T>(Gc, q, LD, tol * tol, fin * fin);
        }
    }
    SDP_TOC(1, t_jac);
    SDP_TIC(t_rest);
    int bad = 0;
    if (tid < q) {
        double n2 = 0.0;
        for (int r = 0; r < q; ++r) { const double x = Gc[tid * LD + r]; n2 = fma(x, x, n2); }
        if (!(n2 > 0.0)) bad = 1;
        w.sq[tid] = sqrt(fmax(n2, 1e-300));
        const double pj = pt[tid], rs = sdp_rsq(pj);
        w.rs[tid] = rs; w.ip[tid] = rs * rs; w.wp[tid] = gamp * pj;
    }
    // U = Lc^-T (G V): U[i][a] = rc_i G[i][a] + sum_{r > i} LinvT[i][r] G[r][a]
    if constexpr (T == 256) sdp_wg4_tri_mm<2>(w, Gc, 0.0);
    else
    for (int blk = tid; blk < qb * qb; blk += T) {
        const int ib = blk / qb, ab = blk - ib * qb;
        const int i0 = 2 * ib, a0 = 2 * ab;
        const int i1 = min(i0 + 1, q - 1), a1 = min(a0 + 1, q - 1);
        const lds_f64* g0 = Gc + a0 * LD;
        const lds_f64* g1 = Gc + a1 * LD;
        double c00 = w.rc[i0] * g0[i0], c01 = w.rc[i0] * g1[i0];
        double c10 = w.rc[i1] * g0[i1], c11 = w.rc[i1] * g1[i1];
        if (i0 + 1 < q) {
            const double l = w.S[i0 * LD + i1];
            c00 = fma(l, g0[i1], c00); c01 = fma(l, g1[i1], c01);
        }
#pragma unroll 4
        for (int r = i0 + 2; r < q; ++r) {
            const double l0 = w.S[i0 * LD + r], l1 = w.S[i1 * LD + r];
            const double v0 = g0[r], v1 = g1[r];
            c00 = fma(l0, v0, c00); c01 = fma(l0, v1, c01);
            c10 = fma(l1, v0, c10); c11 = fma(l1, v1, c11);
        }
        w.X[i0 * LD + a0] = c00;
        if (a0 + 1 < q) w.X[i0 * LD + a1] = c01;
        if (i0 + 1 < q) {
            w.X[i1 * LD + a0] = c10;
            if (a0 + 1 < q) w.X[i1 * LD + a1] = c11;
        }
    }
    bad = __syncthreads_or(bad);
    if (bad) return -INFINITY;
    if (tid < q) {
        double acc = 0.0;
        for (int i = 0; i < q; ++i) acc += w.X[i * LD + tid];
        w.sv[tid] = acc;
        w.rp[tid] = acc * ip0;
    }
    __syncthreads();
    double t0 = 0.0, gval = 0.0;
    for (int a = 0; a < q; ++a) {
        const double sa = w.sq[a], ra = w.rp[a];
        t0 = fma(sa * ra, ra, t0);
        gval += sa - pt[a] * w.dl[a];
    }
    if (tid < q) {     // grad_i = (sum_a sq_a U_ia^2) / (2 p_i^2) - (sum_a sq_a rp_a^2)/2 - delta_i
        double d = 0.0;
#pragma unroll 4
        for (int a = 0; a < q; ++a) { const double x = w.X[tid * LD + a]; d = fma(x * x, w.sq[a], d); }
        const double ipi = w.ip[tid];
        w.grad[tid] = 0.5 * (d * ipi * ipi - t0) - w.dl[tid];
    }
    __syncthreads();
    SDP_TOC(2, t_rest);
    return gval;
}

// ================================================================================================
// Register path (one 64-lane wave, q <= Q <= 32).  EXACT: q == Q, the unrolled loops carry no
// bounds tests, LD is a constant and the code is straight-line.
// ================================================================================================
// Cholesky A = L L^T inside one wave: lane i keeps row i (a[k], k <= i, zeros behind) in
// registers, pivots and multipliers travel through the scalar unit.  On return a[] holds row i of
// L and dinv = 1/L_ii.  Returns 1 if a pivot is not positive (the arithmetic carries on).
// (Handing the finished column to the other lanes through an LDS vector instead -- one write and
// ten two-entry reads per step in place of up to 38 v_readlane -- halves the instruction count of
// the factorisation and was measured slower, 0.66 against 0.61 ms for the whole kernel: every
// step then waits for an LDS round trip inside the serial chain of pivots.)
template <int Q, bool EXACT, int NRS = 2>
__device__ __forceinline__ int sdp_chol_rows(double (&a)[Q], double& dinv, int q, int lane) {
    int fail = 0;
    dinv = 0.0;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        if (EXACT || k < q) {
            const double akk = sdp_lane_bcast(a[k], k);
            if (!(akk > 0.0)) fail = 1;              // identical in every lane
            const double rinv = sdp_rsq_steps<NRS>(akk);
            const double lik = (lane >= k) ? a[k] * rinv : 0.0;
            a[k] = lik;
            if (lane == k) dinv = rinv;
#pragma unroll
            for (int j = k + 1; j < Q; ++j) {
                if (EXACT || j < q) { const double ljk = sdp_lane_bcast(lik, j); a[j] -= lik * ljk; }
            }
#ifndef SDP_CHOL_NO_BARRIER
            if (EXACT) __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }
    return fail;
}

// In-place Cholesky S = Lc Lc^T of w.S (lower; the strict upper part is zeroed), reciprocal
// diagonal in w.rc.  Returns non-zero if S is not PD.
template <int Q, bool EXACT>
__device__ __forceinline__ int sdp_factor_reg(const SdpWork w) {
    const int lane = threadIdx.x;
    const int q = w.q, LD = EXACT ? (Q | 1) : w.LD;
    const bool act = lane < q;
    const int row = act ? lane : 0;
    __syncthreads();
    double a[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        if (EXACT || k < q) { const double h = w.S[row * LD + k]; a[k] = (act && k <= lane) ? h : 0.0; }
        else a[k] = 0.0;
    }
    if (EXACT) SDP_READS_FIRST(Q, 4 * Q);
    double dinv;
    const int fail = sdp_chol_rows<Q, EXACT>(a, dinv, q, lane);
    if (act) {
        w.rc[lane] = dinv;
#pragma unroll
        for (int k = 0; k < Q; ++k) if (EXACT || k < q) w.S[row * LD + k] = (k <= lane) ? a[k] : 0.0;
    }
    __syncthreads();
    if constexpr (sdp_mm<Q>()) {
        // Lc^-T as a full matrix (zeros below the diagonal) for the matrix-pipe product
        // U = Lc^-T (G V) of every evaluation: lane j forms column j of Lc^-1 by forward
        // substitution (x_k = 0 for k < j, so the sums need no bounds) and stores it as row j.
        double x[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            if (EXACT || i < q) {
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) {
                    const double l = w.S[i * LD + k];
                    if (k & 1) acc1 = fma(l, x[k], acc1); else acc0 = fma(l, x[k], acc0);
                }
                const double rci = w.rc[i];
                x[i] = (i == lane) ? rci : -(acc0 + acc1) * rci;
                if (EXACT) { asm volatile("" : "+v"(x[i])); SDP_READS_FIRST(Q + 1, 2 * Q); }
            } else {
                x[i] = 0.0;
            }
        }
        if (act) {
#pragma unroll
            for (int i = 0; i < Q; ++i) if (EXACT || i < q) w.Li[row * LD + i] = (i >= lane) ? x[i] : 0.0;
        }
        __syncthreads();
    }
    return fail;
}

// Split layout of the Jacobi iteration: column j of G is spread over the P = 64/Q lanes j, j+Q, ...,
// each holding RS = ceil(Q/P) consecutive rows, so that (nearly) all 64 lanes work on the
// rotations instead of Q of them.  Sums over a column are formed in one fixed order in every lane
// of the column, which keeps the rotation decisions and parameters bit-identical across its parts.
// ---- register path (q <= 32): pair-per-lane layout, no LDS traffic inside the Jacobi iteration ----
// Lane 16 part + k (k < Q / 2) holds rows (Q / 4) part .. of the columns at positions 2k ("top")
// and 2k + 1 ("bottom") of a linear array of Q columns (Q = 20: k < 10, five rows, four parts).  Odd-even transposition ordering: even
// rounds rotate the pairs (2k, 2k + 1), which sit in one lane; odd rounds the pairs (2k + 1, 2k + 2),
// for which the top of lane k + 1 visits lane k and goes back (two one-lane DPP shifts inside a
// 16-lane row).  Every rotation also exchanges its two columns, so the columns travel through the
// array, are reflected at its ends, and all Q (Q - 1) / 2 pairs meet once in the Q rounds of a sweep.  Where
// a column ends up does not matter: singular values and vectors are used as an unordered set.
// The sum of a pair's dot product over the four row groups is two v_permlane*_swap exchanges.
// Vector-unit instructions only: a layout with a column spread over 64 / Q lanes and arbitrary
// partners (ds_bpermute, 22 per round at Q = 20) ran a quarter slower -- the LDS pipe is what the
// waves of a CU share.
template <int CTRL, bool ZERO>
__device__ __forceinline__ double sdp_dpp_move(double keep, double v) {    // lanes without a source: `keep`, or zero
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(v), CTRL, 0xf, 0xf, ZERO);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(v), CTRL, 0xf, 0xf, ZERO);
    return __hiloint2double(hi, lo);
}
// sum over the four 16-lane rows; every lane gets the same bits
__device__ __forceinline__ double sdp_rowsum4(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    const unsigned xl = __double2loint(x), xh = __double2hiint(x);
    const auto c = __builtin_amdgcn_permlane32_swap(xl, xl, false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(xh, xh, false, false);
    return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}

// Small-angle rotation of the pair (left, right) that annihilates their dot product gamma: with
// d = |r|^2 - |l|^2 and h = sqrt(d^2 + 4 gamma^2):  c^2 = (h + |d|) / (2 h),  s = sign(d) gamma / (h c).
// Two reciprocal square roots, no division.  Returns c, sn of  l' = c l - sn r,  r' = sn l + c r
// and updates the squared norms.
// NR: Newton steps of the two reciprocal square roots; one step (1e-13) is enough for evaluations
// that only steer the Newton iteration.
template <int NR>
__device__ __forceinline__ bool sdp_pair_rotation(double gamma, double& nl, double& nr, double tol2, double fin2,
                                                  double& c, double& sn) {
    c = 1.0; sn = 0.0;
    const double g2 = gamma * gamma, aa = nl * nr;
    const bool rot = g2 > tol2 * aa;
    if (rot) {
        const double d = nr - nl;
        const double hinv = sdp_rsq_steps<NR>(fma(d, d, 4.0 * g2));
        const double v = fma(0.5 * fabs(d), hinv, 0.5);
        const double rv = sdp_rsq_steps<NR>(v);          // 1 / c
        c = v * rv;
        sn = gamma * hinv * rv;
        if (d < 0.0) sn = -sn;
        const double tg = sn * rv * gamma;
        nl -= tg;
        nr += tg;
    }
    return rot && (g2 > fin2 * aa);
}

// Q columns (even, Q / 2 <= 16 positions per row), RP = Q / 4 rows per lane.  Columns and rows beyond
// the problem size are zero: a zero column meets no rotation and just travels.  After an odd
// number of sweeps the array is in reverse order (position i holds column Q - 1 - i).
template <int Q, int NR>
__device__ int sdp_jacobi_oe(double (&gt)[Q / 4], double (&gb)[Q / 4], double tol2, double fin2) {
    constexpr int RP = Q / 4, K = Q / 2;
    const int lane = threadIdx.x & 63, k = lane & 15;
    const bool on = k < K, last = k == K - 1;
    int sweep = 0;
    for (; sweep < SDP_MAX_SWEEPS; ++sweep) {
        bool rotated = false;
        double a2 = 0.0, b2 = 0.0;
#pragma unroll
        for (int i = 0; i < RP; ++i) { a2 += gt[i] * gt[i]; b2 += gb[i] * gb[i]; }
        double nt = sdp_rowsum4(a2), nb = sdp_rowsum4(b2);       // squared norms of top and bottom
#pragma unroll SDP_OE_UNROLL
        for (int r = 0; r < K; ++r) {
            {   // even round: (top, bottom) of this lane; the rotated columns change places
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < RP; ++i) dot += gt[i] * gb[i];
                double c, sn;
                rotated |= sdp_pair_rotation<NR>(sdp_rowsum4(dot), nt, nb, tol2, fin2, c, sn) && on;
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const double l = gt[i], rr = gb[i];
                    gt[i] = sn * l + c * rr;
                    gb[i] = c * l - sn * rr;
                }
                const double t = nt; nt = nb; nb = t;
            }
            {   // odd round: (bottom of this lane, top of the next lane); the last position rests
                double vis[RP];
#pragma unroll
                for (int i = 0; i < RP; ++i) vis[i] = sdp_dpp_move<0x101, true>(0.0, gt[i]);    // row_shl:1
                double nv = sdp_dpp_move<0x101, true>(0.0, nt);
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < RP; ++i) dot += gb[i] * vis[i];
                double c, sn;
                rotated |= sdp_pair_rotation<NR>(sdp_rowsum4(dot), nb, nv, tol2, fin2, c, sn) && on && !last;
                // the last lane has no visitor (a zero column): a quarter turn plus the exchange
                // leaves its bottom column where it is and sends the zero column on
                const double nb_rot = nb, nv_rot = nv;
                if (last) { c = 0.0; sn = 1.0; }
                nb = last ? nb_rot : nv_rot;
                const double nsend = last ? nv_rot : nb_rot;
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const double l = gb[i], rr = vis[i];
                    gb[i] = sn * l + c * rr;
                    vis[i] = c * l - sn * rr;
                }
#pragma unroll
                for (int i = 0; i < RP; ++i) gt[i] = sdp_dpp_move<0x111, false>(gt[i], vis[i]); // row_shr:1, lane 0 keeps its top
                nt = sdp_dpp_move<0x111, false>(nt, nsend);
            }
        }
        if (!__any(rotated)) { ++sweep; break; }
    }
    return sweep;
}

// The same iteration for the 256-thread path (32 < q <= 64): the columns stay in registers here too.
// Wave w holds the rows 16 w .. 16 w + 15 of every column: lane (part, kpos) = (lane >> 5, lane & 31)
// the rows 16 w + 8 part .. + 7 of the columns at positions 2 kpos and 2 kpos + 1 of a linear array
// of 2 K columns (K = ceil(q / 2) positions; columns and rows past q are zeros).  Odd-even
// transposition as in sdp_jacobi_oe, the travelling column moving one lane with whole-wave DPP
// shifts (wave_shl:1 / wave_shr:1; lane 32 is position 0 of the second row group and keeps its
// own column).  A pair's dot product is summed over a lane's 8 rows, over the two row groups of
// the wave (v_permlane32_swap) and over the four waves through LDS (`red`, 512 doubles: two
// buffers in turn, one barrier per round) in one fixed order, so every wave takes the same
// rotation decision with the same parameters.  Against the version with the columns in LDS
// (sdp_jacobi_cols: every round reads and writes the whole matrix, 64 KB per workgroup) the LDS
// traffic of a round is 1.5 KB.  In: column j at Gc[j * LD + r]; out: the same layout, columns in
// the order the array ends in (an unordered set to the caller).  Returns the sweeps.
template <int NR>
__device__ int sdp_jacobi_rows4(lds_f64* Gc, int q, int LD, double tol2, double fin2, lds_f64* red) {
    constexpr int RP = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kpos = lane & 31, part = lane >> 5;
    const int K = (q + 1) >> 1;
    const bool last = kpos == K - 1, first = kpos == 0;
    const int row0 = 16 * wave + 8 * part;
    double gt[RP], gb[RP];
    {
        const int ct = 2 * kpos, cb = 2 * kpos + 1;
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int r = row0 + i;
            const bool rin = r < q;
            const double xt = Gc[(ct < q ? ct : 0) * LD + (rin ? r : 0)], xb = Gc[(cb < q ? cb : 0) * LD + (rin ? r : 0)];
            gt[i] = (rin && ct < q) ? xt : 0.0;
            gb[i] = (rin && cb < q) ? xb : 0.0;
        }
    }
    int flip = 0;
    // sums of two per-lane partial values over all 64 rows of the lane's position, same bits everywhere
    auto colsum2 = [&](double& v0, double& v1) {
        const unsigned l0 = __double2loint(v0), h0 = __double2hiint(v0), l1 = __double2loint(v1), h1 = __double2hiint(v1);
        const auto a = __builtin_amdgcn_permlane32_swap(l0, l0, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(h0, h0, false, false);
        const auto c = __builtin_amdgcn_permlane32_swap(l1, l1, false, false);
        const auto d = __builtin_amdgcn_permlane32_swap(h1, h1, false, false);
        const double x0 = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
        const double x1 = __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
        lds_f64* rb = red + flip * 256;
        flip ^= 1;
        if (part == 0) { rb[wave * 32 + kpos] = x0; rb[128 + wave * 32 + kpos] = x1; }
        __syncthreads();
        v0 = (rb[kpos] + rb[32 + kpos]) + (rb[64 + kpos] + rb[96 + kpos]);
        v1 = (rb[128 + kpos] + rb[160 + kpos]) + (rb[192 + kpos] + rb[224 + kpos]);
    };
    auto colsum = [&](double v) -> double {
        const unsigned l0 = __double2loint(v), h0 = __double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(l0, l0, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(h0, h0, false, false);
        const double x0 = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
        lds_f64* rb = red + flip * 256;
        flip ^= 1;
        if (part == 0) rb[wave * 32 + kpos] = x0;
        __syncthreads();
        return (rb[kpos] + rb[32 + kpos]) + (rb[64 + kpos] + rb[96 + kpos]);
    };
    int sweep = 0;
    for (; sweep < SDP_MAX_SWEEPS; ++sweep) {
        bool rotated = false;
        double nt = 0.0, nb = 0.0;
#pragma unroll
        for (int i = 0; i < RP; ++i) { nt = fma(gt[i], gt[i], nt); nb = fma(gb[i], gb[i], nb); }
        colsum2(nt, nb);                                  // squared norms of top and bottom
        for (int r = 0; r < K; ++r) {
            {   // even round: (top, bottom) of this lane; the rotated columns change places
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < RP; ++i) dot = fma(gt[i], gb[i], dot);
                double c, sn;
                rotated |= sdp_pair_rotation<NR>(colsum(dot), nt, nb, tol2, fin2, c, sn);
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const double l = gt[i], rr = gb[i];
                    gt[i] = sn * l + c * rr;
                    gb[i] = c * l - sn * rr;
                }
                const double t = nt; nt = nb; nb = t;
            }
            {   // odd round: (bottom of this lane, top of the next lane); the last position rests
                double vis[RP];
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const double v = sdp_dpp_move<0x130, true>(0.0, gt[i]);          // wave_shl:1
                    vis[i] = last ? 0.0 : v;
                }
                double nv = sdp_dpp_move<0x130, true>(0.0, nt);
                nv = last ? 0.0 : nv;
                double dot = 0.0;
#pragma unroll
                for (int i = 0; i < RP; ++i) dot = fma(gb[i], vis[i], dot);
                double c, sn;
                rotated |= sdp_pair_rotation<NR>(colsum(dot), nb, nv, tol2, fin2, c, sn) && !last;
                // the last position has no visitor (a zero column): a quarter turn plus the exchange
                // leaves its bottom column where it is and sends the zero column on
                const double nb_rot = nb, nv_rot = nv;
                if (last) { c = 0.0; sn = 1.0; }
                nb = last ? nb_rot : nv_rot;
                const double nsend = last ? nv_rot : nb_rot;
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const double l = gb[i], rr = vis[i];
                    gb[i] = sn * l + c * rr;
                    vis[i] = c * l - sn * rr;
                }
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const double m = sdp_dpp_move<0x138, false>(gt[i], vis[i]);      // wave_shr:1
                    gt[i] = first ? gt[i] : m;           // position 0 (lanes 0 and 32) keeps its top
                }
                const double mn = sdp_dpp_move<0x138, false>(nt, nsend);
                nt = first ? nt : mn;
            }
        }
        if (!__any(rotated)) { ++sweep; break; }
    }
    // where the columns are now: the array is reversed after an odd number of sweeps
    const bool rev = (sweep & 1) != 0;
    const int ot = rev ? 2 * K - 1 - 2 * kpos : 2 * kpos, ob = rev ? 2 * K - 2 - 2 * kpos : 2 * kpos + 1;
    __syncthreads();
    if (kpos < K) {
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const int r = row0 + i;
            if (r < q) {
                if (ot < q) Gc[ot * LD + r] = gt[i];
                if (ob < q) Gc[ob * LD + r] = gb[i];
            }
        }
    }
    __syncthreads();
    return sweep;
}

// The evaluation with the pair-per-lane layout (every register-path size).  With `warm`, the state of the previous
// evaluation (U in w.X, 1/s in w.rs, gamma' p in w.wp, column sums in w.sv) seeds this one:
// G <- G V_prev with V_prev = diag(1/s) (U + (gamma' p) sv^T).
// Instances with sdp_mm<Q>() run the two q x q x q products of an evaluation -- the warm start
// G V_prev and U = Lc^-T (G V) -- on the matrix pipe with the Jacobi registers as operands and
// results: lane (part, kpos) holds the rows part, part + 4, part + 8, ... of its two columns, which
// is the k position (l >> 4) + 4 s that v_mfma_f64_16x16x4 expects from lane l in its s-th step
// and the row (l >> 4) + 4 g of its g-th accumulator, and tile J = 0 / 1 of the right operand is
// the lane's top / bottom column.  Nothing of an evaluation goes through the exchange buffer then
// (w.H keeps the Newton factor).  The other instances (Q = 32) keep rows RP part .. RP part + RP - 1
// per lane, stage G through w.H and substitute with Lc, one column per lane.
template <int Q, bool EXACT>
__device__ __forceinline__ double sdp_eval_oe(const SdpWork w, const lds_f64* pt, int& sweeps, int mode) {
    constexpr int RP = Q / 4, K = Q / 2;
    constexpr bool MM = sdp_mm<Q>();
    constexpr int NI = (Q + 15) / 16;                         // 16-row tiles of a q x q product
    typedef double sdp_double4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int q = w.q, LD = EXACT ? (Q | 1) : w.LD;
    SDP_TIC(t_setup);
    // lane tid serves column `me` (its piece of it during the Jacobi iteration); the lanes
    // tid < q ("act") own the column for everything else
    const bool act = tid < q;
    const int me = (EXACT || tid % Q < q) ? tid % Q : 0;      // column served outside the Jacobi iteration
    const int kpos = tid & 15, part = tid >> 4, r0 = RP * part;
    const bool lane_on = kpos < K;
    // the two columns of this lane (column indices beyond q: zero columns, reads clamped)
    const bool ton = lane_on && (EXACT || 2 * kpos < q), bon = lane_on && (EXACT || 2 * kpos + 1 < q);
    const int ct = ton ? 2 * kpos : 0, cb = bon ? 2 * kpos + 1 : 0;
    const double pme = pt[me];
    double sum, pmin;
    if constexpr (EXACT) {
        sum = 0.0; pmin = INFINITY;
#pragma unroll
        for (int k = 0; k < Q; ++k) { const double v = pt[k]; sum += v; pmin = fmin(pmin, v); }
        SDP_READS_FIRST(Q + 1, 4 * Q);
    } else {
        sum = wave_sum(act ? pme : 0.0);
        pmin = wave_min(act ? pme : INFINITY);
    }
    const double p0 = 1.0 - sum;
    if (!(p0 > 0.0) || !(pmin > 0.0)) return -INFINITY;
    const double rsp0 = sdp_rsq(p0), sp0 = p0 * rsp0, ip0 = rsp0 * rsp0;
    const double gam = sdp_rcp(1.0 + sp0), gamp = gam * rsp0;
    const double rsme = sdp_rsq(pme), sme = pme * rsme, ipme = rsme * rsme;
    const double rs_t = w.rs[ct], sv_t = w.sv[ct], rs_b = w.rs[cb], sv_b = w.sv[cb];   // read before this call overwrites them
    const double rs_me = w.rs[me];
    const double pt_t = pt[ct], pt_b = pt[cb];
    __syncthreads();                    // earlier readers of the vectors are done
    {
        double acc = 0.0;               // m = Lc^T p (the strict upper part of Lc is zero)
        if constexpr (EXACT) {
#pragma unroll
            for (int i = 0; i < Q; ++i) acc += w.S[i * LD + me] * pt[i];
            SDP_READS_FIRST(2 * Q, 2 * Q);
        } else {
#pragma unroll 4
            for (int i = 0; i < q; ++i) acc += w.S[i * LD + me] * pt[i];
        }
        if (act) {
            w.tv[tid] = acc;
            if constexpr (MM) w.ci[tid] = sme * rs_me;        // s_new / s_prev (warm start only)
        }
    }
    __syncthreads();
    lds_f64* GT = w.H;                  // q x q exchange buffer of the instances without MM
    double gt[RP], gb[RP];
    int rr[RP];                         // this lane's rows, clamped; rows beyond q hold zeros
    bool rin[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int r = MM ? 4 * i + part : r0 + i;
        rin[i] = EXACT || r < q;
        rr[i] = rin[i] ? r : 0;
    }
    if (MM && (mode & SDP_EVAL_WARM)) {
        // G V_prev = A B on the matrix pipe: A[r][j] = (s_j / s_prev_j) (Lc[j][r] - gam m_r),
        // B[j][c] = U_prev[j][c] + wp_j sv_c.  Rows of A past q only reach rows of the product that
        // nothing reads; columns of B past q and positions past K must be exact zeros (the Jacobi
        // iteration relies on zero columns there).
        sdp_double4 acc[NI][2];
        double gm[NI];
        int ra[NI];
#pragma unroll
        for (int I = 0; I < NI; ++I) {
            acc[I][0] = (sdp_double4){0.0, 0.0, 0.0, 0.0};
            acc[I][1] = (sdp_double4){0.0, 0.0, 0.0, 0.0};
            const int r = 16 * I + kpos;
            ra[I] = (r < q) ? r : 0;
            gm[I] = gam * w.tv[ra[I]];
        }
#pragma unroll
        for (int s = 0; s < RP; ++s) {
            const int j = 4 * s + part;
            const bool jin = EXACT || j < q;
            const int jc = jin ? j : 0;
            const double cj = w.ci[jc], wpj = w.wp[jc];
            const double xt = w.X[jc * LD + ct], xb = w.X[jc * LD + cb];
            double av[NI];
#pragma unroll
            for (int I = 0; I < NI; ++I) av[I] = cj * (w.S[jc * LD + ra[I]] - gm[I]);
            const double bt = (ton && jin) ? fma(sv_t, wpj, xt) : 0.0;
            const double bb = (bon && jin) ? fma(sv_b, wpj, xb) : 0.0;
#pragma unroll
            for (int I = 0; I < NI; ++I) {
                acc[I][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I], bt, acc[I][0], 0, 0, 0);
                acc[I][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[I], bb, acc[I][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            gt[i] = rin[i] ? acc[i / 4][0][i % 4] : 0.0;
            gb[i] = rin[i] ? acc[i / 4][1][i % 4] : 0.0;
        }
    } else {
        // rows of columns ct, cb of G = s_j (Lc^T e_j - gam m); idle lanes: zeros
        const double st = ton ? pt_t * sdp_rsq(pt_t) : 0.0, sb = bon ? pt_b * sdp_rsq(pt_b) : 0.0;
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            const double m = gam * w.tv[rr[i]];
            gt[i] = rin[i] ? st * (w.S[ct * LD + rr[i]] - m) : 0.0;
            gb[i] = rin[i] ? sb * (w.S[cb * LD + rr[i]] - m) : 0.0;
        }
    }
    if (!MM && (mode & SDP_EVAL_WARM)) {
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            if (ton && rin[i]) GT[rr[i] * LD + ct] = gt[i] * rs_t;
            if (bon && rin[i]) GT[rr[i] * LD + cb] = gb[i] * rs_b;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RP; ++i) { gt[i] = 0.0; gb[i] = 0.0; }
        if constexpr (EXACT) {
#pragma unroll 2
            for (int j = 0; j < Q; ++j) {
                const double wpj = w.wp[j];
                const double vt = ton ? fma(sv_t, wpj, w.X[j * LD + ct]) : 0.0;     // s_j V_prev[j][ct]
                const double vb = bon ? fma(sv_b, wpj, w.X[j * LD + cb]) : 0.0;
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const double g = GT[rr[i] * LD + j];
                    gt[i] = fma(g, vt, gt[i]);
                    gb[i] = fma(g, vb, gb[i]);
                }
                SDP_READS_FIRST(RP + 3, 2 * RP + 4);
            }
        } else {
#pragma unroll 2
            for (int j = 0; j < q; ++j) {
                const double wpj = w.wp[j];
                const double vt = ton ? fma(sv_t, wpj, w.X[j * LD + ct]) : 0.0;
                const double vb = bon ? fma(sv_b, wpj, w.X[j * LD + cb]) : 0.0;
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const double g = rin[i] ? GT[rr[i] * LD + j] : 0.0;
                    gt[i] = fma(g, vt, gt[i]);
                    gb[i] = fma(g, vb, gb[i]);
                }
            }
        }
        __syncthreads();
    }
    SDP_TOC(0, t_setup);
    SDP_TIC(t_jac);
    int nsw;
    {
        const bool loose = (mode & SDP_EVAL_LOOSE) != 0;
        const double tol = loose ? SDP_JACOBI_TOL_LOOSE : SDP_JACOBI_TOL;
        const double fin = (mode & SDP_EVAL_FIRST) ? SDP_JACOBI_FINAL_FIRST : loose ? SDP_JACOBI_FINAL_LOOSE : SDP_JACOBI_FINAL;
        nsw = loose ? sdp_jacobi_oe<Q, 1>(gt, gb, tol * tol, fin * fin) : sdp_jacobi_oe<Q, 2>(gt, gb, tol * tol, fin * fin);
        sweeps += nsw;
    }
    SDP_TOC(1, t_jac);
    SDP_TIC(t_rest);
    // singular values = column norms; U = Lc^-T (G V)
    int bad = 0;
    double sqv = 0.0, su = 0.0;
    {
        double n2t = 0.0, n2b = 0.0;
#pragma unroll
        for (int i = 0; i < RP; ++i) { n2t += gt[i] * gt[i]; n2b += gb[i] * gb[i]; }
        n2t = sdp_rowsum4(n2t); n2b = sdp_rowsum4(n2b);
        // where the columns are now: the array is reversed after an odd number of sweeps
        const bool rev = (nsw & 1) != 0;
        const int ot = rev ? Q - 1 - 2 * kpos : 2 * kpos, ob = rev ? Q - 2 - 2 * kpos : 2 * kpos + 1;
        const bool wt = lane_on && (EXACT || ot < q), wb = lane_on && (EXACT || ob < q);
        if ((wt && !(n2t > 0.0)) || (wb && !(n2b > 0.0))) bad = 1;
        if constexpr (MM) {
            // U[k][c] = sum_r Lc^-T[k][r] (G V)[r][c]: the lane's s-th row is the k position of
            // step s; Lc^-T[k][r] = 0 for r < k, so row tile I only needs the steps with 4 s + 3 >= 16 I
            sdp_double4 u[NI][2];
            int ka[NI];
#pragma unroll
            for (int I = 0; I < NI; ++I) {
                u[I][0] = (sdp_double4){0.0, 0.0, 0.0, 0.0};
                u[I][1] = (sdp_double4){0.0, 0.0, 0.0, 0.0};
                const int k = 16 * I + kpos;
                ka[I] = (k < q) ? k : 0;
            }
#pragma unroll
            for (int s = 0; s < RP; ++s) {
#pragma unroll
                for (int I = 0; I < NI; ++I) {
                    if (4 * s + 3 < 16 * I) continue;
                    const double a = w.Li[ka[I] * LD + rr[s]];          // rows of G V past q are zeros
                    u[I][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, gt[s], u[I][0], 0, 0, 0);
                    u[I][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, gb[s], u[I][1], 0, 0, 0);
                }
            }
            double st_ = 0.0, sb_ = 0.0;
#pragma unroll
            for (int I = 0; I < NI; ++I)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (16 * I + 4 * g >= Q) continue;
                    const int k = 16 * I + part + 4 * g;
                    if (EXACT || k < q) {
                        st_ += u[I][0][g]; sb_ += u[I][1][g];
                        if (wt) w.X[k * LD + ot] = u[I][0][g];
                        if (wb) w.X[k * LD + ob] = u[I][1][g];
                    }
                }
            st_ = sdp_rowsum4(st_); sb_ = sdp_rowsum4(sb_);
            if (part == 0) {
                if (wt) { w.sq[ot] = sqrt(fmax(n2t, 1e-300)); w.sv[ot] = st_; w.rp[ot] = st_ * ip0; }
                if (wb) { w.sq[ob] = sqrt(fmax(n2b, 1e-300)); w.sv[ob] = sb_; w.rp[ob] = sb_ * ip0; }
            }
            if (act) { w.s[tid] = sme; w.rs[tid] = rsme; w.ip[tid] = ipme; w.wp[tid] = gamp * pme; }
        } else {
            // the columns go to LDS by column index, their singular values with them
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                if (wt && rin[i]) GT[rr[i] * LD + ot] = gt[i];
                if (wb && rin[i]) GT[rr[i] * LD + ob] = gb[i];
            }
            if (part == 0) {
                if (wt) w.sq[ot] = sqrt(fmax(n2t, 1e-300));
                if (wb) w.sq[ob] = sqrt(fmax(n2b, 1e-300));
            }
            __syncthreads();
            double g[Q], u[Q];
#pragma unroll
            for (int k = 0; k < Q; ++k) {
                if (EXACT || k < q) g[k] = GT[k * LD + me];
                else g[k] = 0.0;
            }
            if (EXACT) SDP_READS_FIRST(Q, Q);
#pragma unroll
            for (int k = Q - 1; k >= 0; --k) {         // back substitution, one column per lane
                double acc = 0.0;
                if (EXACT || k < q) {
                    acc = g[k];
#pragma unroll
                    for (int r = k + 1; r < Q; ++r) if (EXACT || r < q) acc -= w.S[r * LD + k] * u[r];
                    acc *= w.rc[k];
                }
                asm volatile("" : "+v"(acc));      // keeps the arithmetic here (not sunk behind the loads)
                u[k] = acc;
                su += acc;
                if (EXACT) SDP_READS_FIRST(Q, 2 * Q);
            }
            if (act) {
                w.sv[tid] = su; w.rp[tid] = su * ip0;
                w.s[tid] = sme; w.rs[tid] = rsme; w.ip[tid] = ipme; w.wp[tid] = gamp * pme;
#pragma unroll
                for (int k = 0; k < Q; ++k) if (EXACT || k < q) w.X[k * LD + tid] = u[k];
            }
        }
    }
    bad = __syncthreads_or(bad);
    if (bad) return -INFINITY;
    sqv = w.sq[me];
    if constexpr (MM) su = w.sv[me];
    {   // grad_i = (sum_a sq_a U_ia^2) / (2 p_i^2) - (sum_a sq_a rp_a^2)/2 - delta_i: lane i walks row i of U
        double d = 0.0;
        if constexpr (EXACT) {
#pragma unroll
            for (int a = 0; a < Q; ++a) {
                const double x = w.X[me * LD + a];
                d += x * x * w.sq[a];
                if ((a & 3) == 3) SDP_READS_FIRST(8, 16);
            }
        } else {
#pragma unroll 4
            for (int a = 0; a < q; ++a) {
                const double x = w.X[me * LD + a];
                d += x * x * w.sq[a];
            }
        }
        const double rpa = su * ip0;
        const double t0 = wave_sum(act ? sqv * rpa * rpa : 0.0);
        if (act) w.grad[tid] = 0.5 * (d * ipme * ipme - t0) - w.dl[tid];
    }
    const double gval = wave_sum(act ? sqv - pme * w.dl[me] : 0.0);
    __syncthreads();
    SDP_TOC(2, t_rest);
    return gval;
}

// ------------------------------------------------------------------------------------------------
// Newton direction, register path.  With z = U_i o U_j (pair items), U_i o rp, rp o rp and the
// scaled kernel K'_ab = sq_a sq_b / (sq_a + sq_b) (upper triangle of w.H; K'_aa = sq_a / 2):
//   qf = z^T K' z,  lf = sq^T z,
//   A_ij = -hess g_ij = qf/(2 p_i^2 p_j^2) + lf/(p_i p_j) + e_i + e_j + e_0,
//   e_i = -qf_i/(2 p_i^2) + lf_i/p_i  (z = U_i o rp),   e_0 = qf_0/2 + lf_0  (z = rp o rp).
// Every quadratic form is one work item; an item keeps its z vector in registers and walks K'
// once, two items per walk (each broadcast LDS read feeds two multiply-adds).
// ------------------------------------------------------------------------------------------------
template <int Q, bool EXACT>
__device__ __forceinline__ void sdp_quadform2(const double (&z0)[Q], const double (&z1)[Q],
                                              const SdpWork w, int LD, int q, double& r0, double& r1) {
    double t0 = 0.0, t1 = 0.0;
    const lds_f64* Kt = w.H;           // K'(a,b), b < a, at Kt[b * LD + a]
#pragma unroll
    for (int a = 0; a < Q; ++a) {
        if (EXACT || a < q) {
            const double kaa = 0.25 * w.sq[a];
            double i0 = kaa * z0[a], i1 = kaa * z1[a];
#pragma unroll
            for (int b = 0; b < a; ++b) {
                const double kab = Kt[b * LD + a];
                i0 += kab * z0[b];
                i1 += kab * z1[b];
            }
            t0 += z0[a] * i0;
            t1 += z1[a] * i1;
            if (EXACT) {
                asm volatile("" : "+v"(t0), "+v"(t1));      // keeps the arithmetic with its reads
                SDP_READS_FIRST(Q, 4 * Q);
            }
        }
    }
    r0 = 2.0 * t0;
    r1 = 2.0 * t1;
}

template <int Q, bool EXACT>
__device__ __forceinline__ double sdp_quadform1(const double (&z)[Q], const SdpWork w, int LD, int q) {
    double tot = 0.0;
    const lds_f64* Kt = w.H;
#pragma unroll
    for (int a = 0; a < Q; ++a) {
        if (EXACT || a < q) {
            double inner = 0.25 * w.sq[a] * z[a];
#pragma unroll
            for (int b = 0; b < a; ++b) inner += Kt[b * LD + a] * z[b];
            tot += z[a] * inner;
            if (EXACT) { asm volatile("" : "+v"(tot)); SDP_READS_FIRST(Q, 4 * Q); }
        }
    }
    return 2.0 * tot;
}

// z vector of work item e, lf = sq^T z, the scale 1/(p_i p_j) and where the result goes:
// slot >= 0 -> w.H[slot] (pair (i,j)), slot = -1-i -> w.ci[i] (i == q: e_0), INT_MIN: idle.
// Branch-free; all LDS reads unconditional (idle items read valid rows).
template <int Q, bool EXACT>
__device__ __forceinline__ void sdp_item_z(const SdpWork w, int LD, int e, int npair, double (&z)[Q],
                                           double& lf, double& scale, int& slot) {
    const int q = w.q;
    const bool k0 = e < npair, k1 = !k0 && e < npair + q, k2 = e == npair + q;
    int i, j;
    sdp_tri_index(k0 ? e : 0, i, j);
    const int ia = k1 ? e - npair : i;
    const lds_f64* A = (k0 || k1) ? w.X + ia * LD : w.rp;
    const lds_f64* B = k0 ? w.X + j * LD : w.rp;
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < Q; ++a) {
        if (EXACT || a < q) {
            z[a] = A[a] * B[a];
            acc += z[a] * w.sq[a];
        } else {
            z[a] = 0.0;
        }
        if (EXACT && (a & 3) == 3) {
            asm volatile("" : "+v"(acc));         // keeps the sum here (not sunk behind the reads)
            SDP_READS_FIRST(12, 16);
        }
    }
    const double ipi = w.ip[ia], ipj = w.ip[j];
    scale = ((k0 || k1) ? ipi : 1.0) * (k0 ? ipj : 1.0);
    lf = acc;
    slot = k0 ? i * LD + j : k1 ? -1 - ia : k2 ? -1 - q : INT_MIN;
}

__device__ __forceinline__ void sdp_item_store(const SdpWork w, int slot, double qf, double lf, double scale) {
    const double vq = 0.5 * qf * scale * scale, vl = lf * scale;
    if (slot >= 0) w.H[slot] = vq + vl;
    else if (slot != INT_MIN) w.ci[-1 - slot] = (slot == -1 - w.q) ? vq + vl : vl - vq;
}

template <int T, int Q, bool EXACT, bool PAIR>
__device__ __forceinline__ void sdp_hess_items(const SdpWork w) {
    const int tid = threadIdx.x;
    const int q = w.q, LD = EXACT ? (Q | 1) : w.LD;
    const int npair = q * (q + 1) / 2, nitems = npair + q + 1;
    if constexpr (PAIR) {
        for (int e0 = tid; e0 < nitems; e0 += 2 * T) {
            double z0[Q], z1[Q];
            int s0, s1;
            double l0, l1, c0, c1, r0, r1;
            sdp_item_z<Q, EXACT>(w, LD, e0, npair, z0, l0, c0, s0);
            sdp_item_z<Q, EXACT>(w, LD, e0 + T, npair, z1, l1, c1, s1);
            sdp_quadform2<Q, EXACT>(z0, z1, w, LD, q, r0, r1);
            sdp_item_store(w, s0, r0, l0, c0);
            sdp_item_store(w, s1, r1, l1, c1);
        }
    } else {
        for (int e0 = tid; e0 < nitems; e0 += T) {
            double z0[Q];
            int s0;
            double l0, c0;
            sdp_item_z<Q, EXACT>(w, LD, e0, npair, z0, l0, c0, s0);
            sdp_item_store(w, s0, sdp_quadform1<Q, EXACT>(z0, w, LD, q), l0, c0);
        }
    }
}

// Cholesky of A (lower triangle of w.H) and the solve A step = grad inside one wave.  L is
// written back to w.H (lower) for later solves.  Returns 1 if A is not numerically PD.
template <int Q, bool EXACT>
__device__ __forceinline__ int sdp_chol_solve_reg(const SdpWork w) {
    const int lane = threadIdx.x;
    const int q = w.q, LD = EXACT ? (Q | 1) : w.LD;
    const bool act = lane < q;
    const int row = act ? lane : 0;
    double a[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        if (EXACT || k < q) { const double h = w.H[row * LD + k]; a[k] = (act && k <= lane) ? h : 0.0; }
        else a[k] = 0.0;
    }
    if (EXACT) SDP_READS_FIRST(Q, 4 * Q);
    double dinv;
    // the factor of the Newton matrix gives a search direction: one Newton step (1e-13) in the
    // reciprocal square roots of its pivots is ample
    if (sdp_chol_rows<Q, EXACT, 1>(a, dinv, q, lane)) return 1;
    if (act) {
#pragma unroll
        for (int k = 0; k < Q; ++k) if ((EXACT || k < q) && k <= lane) w.H[row * LD + k] = a[k];
    }
    double bv = act ? w.grad[row] : 0.0;
#pragma unroll
    for (int k = 0; k < Q; ++k) {                    // forward: L y = grad, column oriented
        if (EXACT || k < q) {
            const double yk = sdp_lane_bcast(bv * dinv, k);
            bv = (lane == k) ? yk : bv - a[k] * yk;  // a[k] = 0 in lanes < k
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < Q; ++k) {                    // column `lane` of L = row of L^T
        if (EXACT || k < q) { const double h = w.H[k * LD + row]; a[k] = (act && k >= lane) ? h : 0.0; }
        else a[k] = 0.0;
    }
    if (EXACT) SDP_READS_FIRST(Q, 4 * Q);
#pragma unroll
    for (int k = Q - 1; k >= 0; --k) {               // backward: L^T x = y
        if (EXACT || k < q) {
            const double xk = sdp_lane_bcast(bv * dinv, k);
            bv = (lane == k) ? xk : bv - a[k] * xk;  // a[k] = 0 in lanes > k
        }
    }
    if (act) w.step[lane] = bv;
    __syncthreads();
    return 0;
}

// Register-path instances above this size run the quadratic forms of the Newton matrix on the
// matrix pipe (sdp_hess_items_mfma); at Q = 20 the 16 x 4 tiles cover 384 entries of K' for 210
// that are needed and the vector-unit form, which shares each read of K' between two items, is
// faster (k_stage2 0.464 against 0.492 ms at q = 20, B = 2048).
#ifndef SDP_ITEMS_MFMA_MIN_Q
#define SDP_ITEMS_MFMA_MIN_Q 20
#endif
template <int T, int KS>
__device__ __forceinline__ void sdp_hess_items_mfma(const SdpWork w);

// w.step = (-hess g)^-1 grad g at the state left by sdp_eval_oe (which stays intact: U in w.X).
// Overwrites w.H (K' above, the factor of A in and below the diagonal), w.ci, w.step.
template <int Q, bool EXACT>
__device__ __forceinline__ int sdp_newton_dir_reg(const SdpWork w) {
    constexpr int T = 64;
    const int tid = threadIdx.x;
    const int q = w.q, LD = EXACT ? (Q | 1) : w.LD;
    __syncthreads();
    SDP_TIC(t_items);
    if constexpr (Q > SDP_ITEMS_MFMA_MIN_Q) {
        sdp_hess_items_mfma<T, (Q + 3) / 4>(w);
    } else {
        for (int e = tid; e < q * q; e += T) {
            const int a = EXACT ? e / Q : e / q, b = e - a * q;
            if (b < a) {
                const double sa = w.sq[a], sb = w.sq[b];
                w.H[b * LD + a] = sa * sb * sdp_rcp(sa + sb);
            }
        }
        __syncthreads();
        sdp_hess_items<T, Q, EXACT, (Q <= 20)>(w);
    }
    __syncthreads();
    const double e0 = w.ci[q];
    const int npair = q * (q + 1) / 2;
    for (int e = tid; e < npair; e += T) {
        int i, j;
        sdp_tri_index(e, i, j);
        w.H[i * LD + j] += e0 + w.ci[i] + w.ci[j];
    }
    __syncthreads();
    SDP_TOC(4, t_items);
    SDP_TIC(t_chol);
    const int f = sdp_chol_solve_reg<Q, EXACT>(w);
    SDP_TOC(5, t_chol);
    return f;
}

// ---- Newton direction, wide path ------------------------------------------------------------------
// Order of the LDS accesses of one wave without a workgroup barrier (the LDS serves one wave's
// accesses in program order; this only stops the compiler from moving them).
__device__ __forceinline__ void sdp_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Block column c0 .. c0 + 15 of the factorisation by the first wave (see sdp_chol_solve_blocked).
// Out of line: its broadcasts and lane masks live in scalar registers, and inlined into the solver
// they pushed the scalar register file into spilling through vector lanes.
static __device__ __noinline__ int sdp_chol_block_wave0(const SdpWork w, int c0) {
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(w.q), LD = __builtin_amdgcn_readfirstlane(w.LD);
    const bool act = lane < q;
    const int row = act ? lane : 0;
    const int lanem = act ? lane : -1;           // k <= lanem: entry (lane, k) of the lower triangle exists
    int fail = 0;
    double a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int k = c0 + u;
        const double h = w.H[row * LD + (k < q ? k : 0)];
        a[u] = (k <= lanem) ? h : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (c0 + k < q) {
            const double akk = sdp_lane_bcast(a[k], c0 + k);
            if (!(akk > 0.0)) fail = 1;              // identical in every lane
            const double rinv = sdp_rsq_steps<1>(akk);   // a search direction: 1e-13 is ample
            const double lik = (lane >= c0 + k) ? a[k] * rinv : 0.0;
            a[k] = lik;
#pragma unroll
            for (int j = k + 1; j < 16; ++j) {
                const double ljk = sdp_lane_bcast(lik, c0 + j);     // c0 + j <= 63; zero past q
                a[j] -= lik * ljk;
            }
            // a step's broadcasts go through scalar registers: keep the next step's from being
            // hoisted over this one (hundreds of live scalars, spilled through vector lanes)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int k = c0 + u;
        if (k <= lanem) w.H[row * LD + k] = a[u];
    }
    return fail;
}

// The two triangular solves L y = grad, L^T step = y by the first wave, L read from w.H: lane i keeps
// row i (then column i) in registers, the components travel through the scalar unit.  Out of line
// for the same reason as sdp_chol_block_wave0.
template <int QM>
static __device__ __noinline__ void sdp_tri_solves_wave0(const SdpWork w) {
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(w.q), LD = __builtin_amdgcn_readfirstlane(w.LD);
    const bool act = lane < q;
    const int row = act ? lane : 0;
    const int lanem = act ? lane : -1, lanex = act ? lane : 64;
    sdp_wave_sync();
    double a[QM];
#pragma unroll
    for (int k0 = 0; k0 < QM; k0 += 16) {
        if (k0 < q) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + u;
                const double h = w.H[row * LD + k];          // past q: inside the allocation, masked
                a[k] = (k <= lanem) ? h : 0.0;               // steps k >= q never run: no test on q
            }
            SDP_READS_FIRST(16, 32);
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) a[k0 + u] = 0.0;
        }
    }
    const double dinv = act ? sdp_rcp(w.H[row * LD + row]) : 0.0;
    double bv = act ? w.grad[row] : 0.0;
#pragma unroll
    for (int k = 0; k < QM; ++k) {                   // forward: L y = grad, column oriented
        if (k < q) {
            const double yk = sdp_lane_bcast(bv * dinv, k);
            bv = (lane == k) ? yk : bv - a[k] * yk;  // a[k] = 0 in lanes < k
        }
    }
#pragma unroll
    for (int k0 = 0; k0 < QM; k0 += 16) {            // column `lane` of L = row of L^T
        if (k0 < q) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + u;
                const double h = w.H[(k < q ? k : 0) * LD + row];
                a[k] = (k >= lanex) ? h : 0.0;
            }
            SDP_READS_FIRST(16, 32);
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) a[k0 + u] = 0.0;
        }
    }
#pragma unroll
    for (int k = QM - 1; k >= 0; --k) {              // backward: L^T x = y
        if (k < q) {
            const double xk = sdp_lane_bcast(bv * dinv, k);
            bv = (lane == k) ? xk : bv - a[k] * xk;  // a[k] = 0 in lanes > k
        }
    }
    if (act) w.step[lane] = bv;
}

// Cholesky of A (lower triangle of w.H; L goes back there) and the solve A step = grad for the
// whole workgroup, blocked by 16 columns: the first
// wave factors a block column (lane i keeps the 16 entries of row i, pivots and multipliers through
// the scalar unit: the unblocked algorithm restricted to the block's columns, 120 updates instead
// of the q^2 / 2 of a full elimination step by step), then all waves subtract the panel's outer
// product from the trailing lower triangle on the matrix pipe (16 x 16 tiles of w.H, read, updated
// with four matrix instructions and written back), two barriers per block.  The two triangular
// solves stay in the first wave, rows re-read from LDS.  Called by every thread; returns 1 for
// all if A is not numerically positive definite.
template <int T, int QM>
__device__ __forceinline__ int sdp_chol_solve_blocked(const SdpWork w) {
    typedef double sdp_double4 __attribute__((ext_vector_type(4)));
    constexpr int NW = T / 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = __builtin_amdgcn_readfirstlane(w.q), LD = __builtin_amdgcn_readfirstlane(w.LD);
    const int nb = (q + 15) >> 4;
    const bool act = lane < q;
    const int row = act ? lane : 0;
    const int li = lane & 15, kg = lane >> 4;
    int fail = 0;
    for (int kb = 0; kb < nb; ++kb) {
        const int c0 = 16 * kb;
        if (wave == 0) fail |= sdp_chol_block_wave0(w, c0);
        if (c0 + 16 >= q) break;                             // uniform: last block, nothing behind it
        __syncthreads();
        const int nt = nb - 1 - kb, ntile = nt * (nt + 1) / 2;
        for (int t = wave; t < ntile; t += NW) {
            int ti, tj;
            sdp_tri_index(t, ti, tj);
            const int I = kb + 1 + ti, J = kb + 1 + tj;
            sdp_double4 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = 16 * I + kg + 4 * g, j = 16 * J + li;
                const bool in = i < q && j <= i;
                const double c = w.H[(in ? i : 0) * LD + (in ? j : 0)];
                acc[g] = in ? c : 0.0;
            }
            const int ra = 16 * I + li, rb = 16 * J + li;
            double av[4], bv[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int k = c0 + 4 * s4 + kg;              // < c0 + 16 < q
                const double x = w.H[(ra < q ? ra : 0) * LD + k], y = w.H[(rb < q ? rb : 0) * LD + k];
                av[s4] = ra < q ? -x : 0.0;
                bv[s4] = rb < q ? y : 0.0;
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = 16 * I + kg + 4 * g, j = 16 * J + li;
                if (i < q && j <= i) w.H[i * LD + j] = acc[g];
            }
        }
        __syncthreads();
    }
    fail = __builtin_amdgcn_readfirstlane(fail);             // the same in every lane: a scalar branch
    if (wave == 0 && !fail) sdp_tri_solves_wave0<QM>(w);
    return __syncthreads_or(fail);
}

// The quadratic forms of all work items on the matrix pipe, 16 items per wave at a time.  With
// K2 = the upper triangle of K' (entries above the diagonal doubled, diagonal sq / 2) and the z
// vectors of the 16 items as the columns of Z, z^T K' z = z^T (K2 z) and K2 Z is a product of a
// q x q triangle with a q x 16 matrix: row tile I needs the k steps from 4 I on.  Lane l supplies
// Z at the k position (l >> 4) + 4 s of step s and receives row (l >> 4) + 4 (4 I + g) of the
// product in accumulator g of row tile I -- the same 16 positions, so the z values a lane builds
// for the right operand are the ones it multiplies the result with.  The fragments of K2 are
// computed from sq in registers and stay there for all tiles of a call (K' never goes to LDS on
// this path); a broadcast read of K' per multiply-add (what a loop over threads does) would keep
// the LDS port busy four times as long as the matrix pipe is here.
// KS = number of k steps (4 KS >= q), a template parameter so that the products form straight-line
// code: with a bound known only at run time every product sits behind a branch and the compiler
// moves the accumulators between register sets at each of them.
template <int T, int KS>
__device__ __forceinline__ void sdp_hess_items_mfma(const SdpWork w) {
    typedef double sdp_double4 __attribute__((ext_vector_type(4)));
    constexpr int NW = T / 64, NRT = (KS + 3) / 4;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = __builtin_amdgcn_readfirstlane(w.q), LD = __builtin_amdgcn_readfirstlane(w.LD);
    const int li = lane & 15, kg = lane >> 4;
    const int npair = q * (q + 1) / 2, nitems = npair + q + 1, ntile = (nitems + 15) >> 4;
    double af[NRT][KS], sqv[KS], sqr[NRT];
#pragma unroll
    for (int I = 0; I < NRT; ++I) { const int row = 16 * I + li; sqr[I] = w.sq[row < q ? row : 0]; }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int a = 4 * s + kg;
        const double sa = w.sq[a < q ? a : 0];
        sqv[s] = a < q ? sa : 0.0;
#pragma unroll
        for (int I = 0; I < NRT; ++I) {
            if (s >= 4 * I) {                          // K'(b, a) = sq_a sq_b / (sq_a + sq_b)
                const int row = 16 * I + li;
                const bool in = row < q && a < q;
                const double sb = sqr[I], h2 = 2.0 * sa * sb * sdp_rcp(sa + sb);
                af[I][s] = !in ? 0.0 : (a > row) ? h2 : (a == row) ? 0.5 * sb : 0.0;
            } else {
                af[I][s] = 0.0;
            }
        }
    }
    for (int t = wv; t < ntile; t += NW) {
        const int e = 16 * t + li;
        const bool k0 = e < npair, k1 = !k0 && e < npair + q, k2 = e == npair + q;
        int i, j;
        sdp_tri_index(k0 ? e : 0, i, j);
        const int ia = k1 ? e - npair : i;
        const lds_f64* A = (k0 || k1) ? w.X + ia * LD : w.rp;
        const lds_f64* B = k0 ? w.X + j * LD : w.rp;
        double z[4 * NRT];
        double lf = 0.0;
#pragma unroll
        for (int s = 0; s < 4 * NRT; ++s) {
            z[s] = 0.0;
            if (s < KS) {
                const int a = 4 * s + kg, ac = a < q ? a : 0;
                const double av = A[ac], bv = B[ac];
                z[s] = a < q ? av * bv : 0.0;
                lf = fma(z[s], sqv[s], lf);
            }
        }
        sdp_double4 acc[NRT];
#pragma unroll
        for (int I = 0; I < NRT; ++I) acc[I] = (sdp_double4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int I = 0; I < NRT; ++I)
                if (s >= 4 * I) acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[I][s], z[s], acc[I], 0, 0, 0);
        }
        double qf = 0.0;
#pragma unroll
        for (int I = 0; I < NRT; ++I) {
#pragma unroll
            for (int g = 0; g < 4; ++g) qf = fma(z[4 * I + g], acc[I][g], qf);
        }
        qf += __shfl_xor(qf, 16); qf += __shfl_xor(qf, 32);
        lf += __shfl_xor(lf, 16); lf += __shfl_xor(lf, 32);
        const double ipi = w.ip[ia], ipj = w.ip[j];
        const double scale = ((k0 || k1) ? ipi : 1.0) * (k0 ? ipj : 1.0);
        const int slot = k0 ? i * LD + j : k1 ? -1 - ia : k2 ? -1 - q : INT_MIN;
        if (kg == 0) sdp_item_store(w, slot, qf, lf, scale);
    }
}

// The quadratic forms on the matrix pipe, 16 work items per wave at a time, then the blocked
// factorisation and the solves.  Same outputs as sdp_newton_dir_reg.
template <int T>
__device__ int sdp_newton_dir_wide(const SdpWork w) {
    const int tid = threadIdx.x;
    const int q = __builtin_amdgcn_readfirstlane(w.q), LD = __builtin_amdgcn_readfirstlane(w.LD);
    __syncthreads();
    SDP_TIC(t_items);
    static_assert(T % 64 == 0, "whole waves");
    if (q <= 40) sdp_hess_items_mfma<T, 10>(w);
    else if (q <= 52) sdp_hess_items_mfma<T, 13>(w);
    else sdp_hess_items_mfma<T, 16>(w);
    __syncthreads();
    const double e0 = w.ci[q];
    const int npair = q * (q + 1) / 2;
    for (int e = tid; e < npair; e += T) {
        int i, j;
        sdp_tri_index(e, i, j);
        w.H[i * LD + j] += e0 + w.ci[i] + w.ci[j];
    }
    __syncthreads();
    SDP_TOC(4, t_items);
    SDP_TIC(t_chol);
    const int fail = sdp_chol_solve_blocked<T, OEI_MAX_Q>(w);
    SDP_TOC(5, t_chol);
    return fail;
}

// ---- dispatch between the two paths: Q > 0 register path (T == 64), Q == 0 wide path -----------
// Inside the solver the register-path pieces are inlined (one call site each): a real call would
// save and restore the callee-saved half of a 170-register frame through scratch memory every
// time.  The kernels that use single pieces (sensitivities) call them out of line instead.
template <int T, int Q, bool EXACT>
__device__ __forceinline__ int sdp_factor_x(const SdpWork w) {
    if constexpr (Q > 0) return sdp_factor_reg<Q, EXACT>(w);
    else return sdp_factor_wide<T>(w);
}
template <int T, int Q, bool EXACT>
__device__ __forceinline__ double sdp_eval_x(const SdpWork w, const lds_f64* pt, int& sweeps, int mode) {
    if constexpr (Q > 0) return sdp_eval_oe<Q, EXACT>(w, pt, sweeps, mode);
    else return sdp_eval_wide<T>(w, pt, sweeps, mode);
}
template <int T, int Q, bool EXACT>
__device__ __forceinline__ int sdp_newton_dir_x(const SdpWork w) {
    if constexpr (Q > 0) return sdp_newton_dir_reg<Q, EXACT>(w);
    else return sdp_newton_dir_wide<T>(w);
}
template <int T, int Q, bool EXACT>
__device__ __noinline__ int sdp_factor_call(const SdpWork w) { return sdp_factor_x<T, Q, EXACT>(w); }
template <int T, int Q, bool EXACT>
__device__ __noinline__ double sdp_eval_call(const SdpWork w, const lds_f64* pt, int& sweeps, int mode) {
    return sdp_eval_x<T, Q, EXACT>(w, pt, sweeps, mode);
}
template <int T, int Q, bool EXACT>
__device__ __noinline__ int sdp_newton_dir_call(const SdpWork w) { return sdp_newton_dir_x<T, Q, EXACT>(w); }

template <int T, int Q>
__device__ __forceinline__ int sdp_factor(const SdpWork w) {
    if constexpr (Q > 0) return (w.q == Q) ? sdp_factor_call<T, Q, true>(w) : sdp_factor_call<T, Q, false>(w);
    else return sdp_factor_call<T, 0, false>(w);
}
template <int T, int Q>
__device__ __forceinline__ double sdp_eval(const SdpWork w, const lds_f64* pt, int& sweeps, int mode) {
    if constexpr (Q > 0)
        return (w.q == Q) ? sdp_eval_call<T, Q, true>(w, pt, sweeps, mode) : sdp_eval_call<T, Q, false>(w, pt, sweeps, mode);
    else return sdp_eval_call<T, 0, false>(w, pt, sweeps, mode);
}
template <int T, int Q>
__device__ __forceinline__ int sdp_newton_dir(const SdpWork w) {
    if constexpr (Q > 0) return (w.q == Q) ? sdp_newton_dir_call<T, Q, true>(w) : sdp_newton_dir_call<T, Q, false>(w);
    else return sdp_newton_dir_call<T, 0, false>(w);
}

// ================================================================================================
// Full solve.  On entry w.S (symmetric) and w.dl are set.  On exit w.p holds the optimum and the
// state (U in w.X, sv, sq, rp, ip, s) belongs to it.  Returns status; *g_out = max g.
// ================================================================================================
// The loop is written as a state machine so that the evaluation has a single (inlined) call site:
// first the start point, then one or more trial points per Newton iteration.
// Pieces as sdp_solve_impl uses them: inlined, except in the Q = 32 instance.  Inlined there, the
// padded (q < 32) solver needs about 1000 scalar registers spilled into lanes of vector registers
// and, at 256 vector registers, the compiler spills those carrier registers to scratch memory in
// turn; code built that way returned wrong iterates for q = 21..31 (with the Jacobi round loop
// unrolled: 61 Newton iterations, values off by 5e-3), and the same source built with
// -mllvm -amdgpu-spill-sgpr-to-vgpr=false was correct, so the defect is in that spill path, not
// in the arithmetic.  Out of line, every piece has a register allocation of its own and no
// carrier register is spilled (tests/test_cpu_host.py checks the built code for that pattern).
template <int T, int Q, bool EXACT>
__device__ __forceinline__ int sdp_factor_s(const SdpWork w) {
    if constexpr (Q == 32) return sdp_factor_call<T, Q, EXACT>(w);
    else return sdp_factor_x<T, Q, EXACT>(w);
}
template <int T, int Q, bool EXACT>
__device__ __forceinline__ double sdp_eval_s(const SdpWork w, const lds_f64* pt, int& sweeps, int mode) {
    if constexpr (Q == 32) return sdp_eval_call<T, Q, EXACT>(w, pt, sweeps, mode);
    else return sdp_eval_x<T, Q, EXACT>(w, pt, sweeps, mode);
}
template <int T, int Q, bool EXACT>
__device__ __forceinline__ int sdp_newton_dir_s(const SdpWork w) {
    if constexpr (Q == 32) return sdp_newton_dir_call<T, Q, EXACT>(w);
    else return sdp_newton_dir_x<T, Q, EXACT>(w);
}

// `start_in` / `start_out` (global memory, q entries each, or null): solver start state carried from
// a nearby problem to this one -- the role of the reference's warm start (sdp.py:77-121: nearest
// past solution).  The state is the ratio of the optimal weights to the closed-form start weights
// of the same problem; multiplied with this problem's closed-form weights it gives a start that
// follows the first-order change of every point's own mean and variance, so it stays useful over
// larger moves than the old weights themselves (which beat the closed form only for moves below
// 1e-4 of the box).  Used when the product lies inside the simplex.
template <int T, int Q, bool EXACT>
__device__ __forceinline__ int sdp_solve_impl(const SdpWork w, double* g_out, int* iters_out,
                                              const double* start_in = nullptr,
                                              double* start_out = nullptr) {
    const int tid = threadIdx.x;
    const int q = w.q, LD = w.LD;
    // closed-form q = 1 weights, shrunk into the simplex
    if (tid < q) {
        const double d = w.dl[tid], s2 = w.S[tid * LD + tid];
        const double r = sqrt(s2 + d * d);
        w.pn[tid] = (d >= 0.0) ? 0.5 * s2 / (r * (r + d)) : 0.5 * (r - d) / r;
    }
    __syncthreads();
    double tot = 0.0;
    for (int i = 0; i < q; ++i) tot += w.pn[i];
    __syncthreads();
    const double pcf = (tid < q) ? fmax(w.pn[tid] / (1.0 + tot), 1e-300) : 0.0;
    bool warm = false;
    if (start_in) {                                      // uniform
        const double v = (tid < q) ? start_in[tid] * pcf : 0.0;
        const int inside = (tid < q) ? (v > 1e-300 && v < 1.0) : 1;      // false for NaN
        if (tid < q) w.pn[tid] = inside ? v : 0.0;
        warm = __syncthreads_and(inside) != 0;
        double ws = 0.0;
        for (int i = 0; i < q; ++i) ws += w.pn[i];
        warm = warm && (ws < 1.0 - 1e-9);
        __syncthreads();
        if (tid < q) w.p[tid] = warm ? v : pcf;
    } else {
        if (tid < q) w.p[tid] = pcf;
    }
    __syncthreads();
    int sweeps = 0, it = 0, status = 1;
#ifdef OEI_SDP_TIMING
    if (tid == 0) { for (int i = 0; i < 8; ++i) w.scal[i] = 0.0; }
#endif
    if (sdp_factor_s<T, Q, EXACT>(w)) { *g_out = -INFINITY; *iters_out = 0; return 2; }
    double g = -INFINITY, t = 1.0, dec = 0.0, slack = 0.0, rel = 1.0;
    bool first = true, full = true, finish = false;
    int ls = 0;
    for (;;) {
        // evaluations that only steer the iteration run with the loose Jacobi tolerance; the one
        // after a step below the stopping tolerance is the final state and runs tight
        // (from a given starting point the first step may already be tiny: no early Jacobi exit)
        const int mode = first ? (warm ? SDP_EVAL_LOOSE : (SDP_EVAL_LOOSE | SDP_EVAL_FIRST))
                               : (SDP_EVAL_WARM | (rel < SDP_STEP_TOL ? 0 : SDP_EVAL_LOOSE));
        const int sweeps_before = sweeps;
        const double gv = sdp_eval_s<T, Q, EXACT>(w, first ? w.p : w.pn, sweeps, mode);
        // an eigen-iteration that ran into its sweep limit has not converged: the state is then
        // reported as "iteration limit", not as solved
        const bool capped = sweeps - sweeps_before >= SDP_MAX_SWEEPS;
        if (first) {
            first = false;
            g = gv;
            if (!(g > -INFINITY)) { status = 2; break; }
        } else {
            if (!(gv >= g + 1e-4 * t * dec - slack) && ++ls < 40) {     // backtrack
                t *= 0.5; full = false;
                __syncthreads();
                if (tid < q) w.pn[tid] = w.p[tid] + t * w.step[tid];
                __syncthreads();
                continue;
            }
            __syncthreads();
            if (tid < q) w.p[tid] = w.pn[tid];
            __syncthreads();
            g = gv;
            if (!(g > -INFINITY)) { status = 2; break; }
            if (finish || (full && rel < SDP_STEP_TOL)) { status = capped ? 1 : 0; break; }
        }
        if (++it > SDP_MAX_NEWTON) break;
        SDP_TIC(t_dir);
        const int dir_fail = sdp_newton_dir_s<T, Q, EXACT>(w);
        SDP_TOC(3, t_dir);
        if (dir_fail) { status = 2; break; }
        double ssum = 0.0, psum = 0.0;
        rel = 0.0; dec = 0.0;
        if constexpr (T == 64) {                         // one entry per lane (q <= 32), wave reductions
            const bool in = tid < q;
            const double st = in ? w.step[tid] : 0.0, pi = in ? w.p[tid] : 1.0, gi = in ? w.grad[tid] : 0.0;
            rel = sdp_max32(fabs(st) / pi);
            ssum = sdp_sum32(st);
            psum = sdp_sum32(in ? pi : 0.0);
            dec = sdp_sum32(gi * st);
        } else {
            for (int i = 0; i < q; ++i) {
                const double st = w.step[i];
                rel = fmax(rel, fabs(st) / w.p[i]);
                ssum += st; psum += w.p[i];
                dec += w.grad[i] * st;
            }
        }
        rel = fmax(rel, fabs(ssum) / (1.0 - psum));
        if (!(rel == rel)) { status = 2; break; }
        t = rel < 0.9 ? 1.0 : 0.9 / rel;
        slack = 1e-14 * (1.0 + fabs(g));
        full = (t == 1.0);
        // second stopping rule: a Newton decrement below rounding level.  Weights that vanish at
        // the optimum keep a large relative step for ever; the objective cannot improve any more.
        // One evaluation at the current point with the tight tolerance leaves the final state.
        if (dec <= SDP_DECREMENT_TOL * (1.0 + fabs(g))) { finish = true; rel = 0.0; t = 0.0; }
        ls = 0;
        __syncthreads();
        if (tid < q) w.pn[tid] = w.p[tid] + t * w.step[tid];
        __syncthreads();
    }
    *g_out = g;
    *iters_out = (it & 0xffff) | (sweeps << 16);
    if (start_out && tid < q) start_out[tid] = (status == 2) ? NAN : w.p[tid] / pcf;
    return status;
}

template <int T, int Q>
__device__ __forceinline__ int sdp_solve(const SdpWork w, double* g_out, int* iters_out,
                                         const double* start_in = nullptr,
                                         double* start_out = nullptr) {
    if constexpr (Q > 0) {
        return (w.q == Q) ? sdp_solve_impl<T, Q, true>(w, g_out, iters_out, start_in, start_out)
                          : sdp_solve_impl<T, Q, false>(w, g_out, iters_out, start_in, start_out);
    } else {
        return sdp_solve_impl<T, 0, false>(w, g_out, iters_out, start_in, start_out);
    }
}

// ---- outputs from the final state ---------------------------------------------------------------
// Writes N = B^-1 into w.H and c into w.ci, (N c) into w.tv (w.step is scratch); returns c^T N c.
// Needs w.X = U, w.sq, w.ip, w.rp of the optimum; `mu` in LDS.
template <int T>
__device__ double sdp_primal(const SdpWork w, const lds_f64* mu) {
    const int tid = threadIdx.x;
    const int q = w.q, LD = w.LD;
    const lds_f64* U = w.X;
    __syncthreads();
    if (tid < q) {     // c = mu + B p / 2 = mu + R (sq * rp)
        double acc = 0.0;
        for (int a = 0; a < q; ++a) acc += sdp_R(w, tid, a) * w.sq[a] * w.rp[a];
        w.ci[tid] = mu[tid] + acc;
        w.step[tid] = 0.5 / w.sq[tid];
    }
    __syncthreads();
    if (T == 64 && q <= 32) {
        // one wave: N = U diag(1/(2 sq)) U^T on the matrix pipe, lower tiles (0,0), (1,0), (1,1),
        // mirrored on the way out.  Lane l supplies row 16 I + (l & 15) at k position 4 s + (l >> 4).
        typedef double sdp_double4 __attribute__((ext_vector_type(4)));
        const int li = tid & 15, kg = tid >> 4;
        const bool two = q > 16;                           // uniform
        sdp_double4 acc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] = (sdp_double4){0.0, 0.0, 0.0, 0.0};
        const int i0 = li, i1 = 16 + li;
        const bool v0 = i0 < q, v1 = i1 < q;
        const lds_f64* u0 = U + (v0 ? i0 : 0) * LD;
        const lds_f64* u1 = U + (v1 ? i1 : 0) * LD;
        const int ks = (q + 3) >> 2;
        for (int st = 0; st < ks; ++st) {
            const int a = 4 * st + kg;
            const bool ain = a < q;
            const int ac = ain ? a : 0;
            const double wa = w.step[ac];
            const double x0 = u0[ac], x1 = u1[ac];
            const double b0 = (v0 && ain) ? x0 : 0.0, b1 = (v1 && ain) ? x1 : 0.0;
            const double a0 = b0 * wa, a1 = b1 * wa;
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
            if (two) {
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[2], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r0 = kg + 4 * g, r1 = 16 + kg + 4 * g, c0 = li, c1 = 16 + li;
            if (r0 < q && c0 <= r0) { w.H[r0 * LD + c0] = acc[0][g]; w.H[c0 * LD + r0] = acc[0][g]; }
            if (two) {
                if (r1 < q && c0 < q) { w.H[r1 * LD + c0] = acc[1][g]; w.H[c0 * LD + r1] = acc[1][g]; }
                if (r1 < q && c1 <= r1) { w.H[r1 * LD + c1] = acc[2][g]; w.H[c1 * LD + r1] = acc[2][g]; }
            }
        }
    } else {
        for (int e = tid; e < q * q; e += T) {
            const int i = e / q, j = e - i * q;
            if (j > i) continue;
            double acc = 0.0;
#pragma unroll 4
            for (int a = 0; a < q; ++a) acc += U[i * LD + a] * U[j * LD + a] * w.step[a];
            w.H[i * LD + j] = acc;
            w.H[j * LD + i] = acc;
        }
    }
    __syncthreads();
    if (tid < q) {
        double acc = 0.0;
        for (int j = 0; j < q; ++j) acc += w.H[tid * LD + j] * w.ci[j];
        w.tv[tid] = acc;
    }
    __syncthreads();
    double nu = 0.0;
    for (int i = 0; i < q; ++i) nu += w.ci[i] * w.tv[i];
    return nu;
}

// Host-side choice of the kernel instance for a given q: CALL(T, Q) with Q = 0 for the wide path.
#define SDP_DISPATCH_Q(q, CALL) do { \
        if ((q) <= 8) { CALL(64, 8); } else if ((q) <= 16) { CALL(64, 16); } \
        else if ((q) <= 20) { CALL(64, 20); } else if ((q) <= 32) { CALL(64, 32); } \
        else { CALL(256, 0); } } while (0)
// waves per SIMD the register budget is tuned for (launch bound of the solver kernels)
#define SDP_MIN_WAVES(Q) 2
