// Multistart projected Newton on the device: the batched consumer of the exact acquisition
// Hessian (the role KNITRO's SQP has in the reference, solvers.py:56-69, 71-141, for every start
// of bo.py:129-192 at once).  Like the L-BFGS driver (oei_lbfgs.hip) every run is a state machine
// in global memory; a running run waits either for the value and gradient at its trial point or
// for the Hessian at its iterate.  One round = the caller evaluates the two dense lists, then
// k_newton_step (one 256-thread workgroup per run) consumes the results:
//   trial result   Armijo test; accepted: stopping tests, then the run asks for the Hessian at the
//                  new iterate; rejected: step length / 4 and a new trial point;
//   Hessian        variables blocked at a bound are frozen; (H + lam I) d = -g on the others with
//                  the smallest lam of the ladder 0, 1e-4 s, 1e-3 s, ... (s = max |H_ij|) that
//                  makes the matrix positive definite (the acquisition is not convex); Cholesky
//                  in LDS, column by column; steepest descent if no lam of the ladder works or d
//                  is not a descent direction; first trial point x + t d, t = min(1, |box| / |d|).
// Same method and constants as lockstep.minimize_lockstep_newton.
#include "oei_common.h"

namespace {

enum { ST_RUNNING = 0, ST_CONV_PG = 1, ST_CONV_F = 2, ST_LS_FAILED = 3, ST_STALLED = 4,
       ST_BAD_START = 5, ST_ITER_LIMIT = 6 };
enum { PH_TRIAL = 0, PH_HESS = 1 };
constexpr int NT = 256;
constexpr int LADDER = 12;

struct NewtonView {
    int B, n;
    double *x, *f, *g, *d, *t, *xt, *Xc, *Xh, *lo, *hi;
    int *status, *nit, *nfev, *nhev, *ls, *phase, *rowof, *map_t, *map_h, *counts, *need, *rung;
};

inline size_t up32(size_t v) { return (v + 31) / 32 * 32; }

void layout(int B, int n, size_t* off_d, size_t* off_i, size_t* nd, size_t* ni) {
    const size_t Bn = (size_t)B * n;
    const size_t dsz[OEI_NEWTON_ND] = {Bn, (size_t)B, Bn, Bn, (size_t)B, Bn, Bn, Bn, (size_t)n, (size_t)n};
    size_t o = 0;
    for (int k = 0; k < OEI_NEWTON_ND; ++k) { off_d[k] = o; o += up32(dsz[k]); }
    *nd = o;
    o = 0;
    for (int k = 0; k < OEI_NEWTON_NI; ++k) { off_i[k] = o; o += up32(k == OEI_NEWTON_I_COUNTS ? 2 : (size_t)B); }
    *ni = o;
}

NewtonView make_view(int B, int n, double* dws, int* iws) {
    size_t od[OEI_NEWTON_ND], oi[OEI_NEWTON_NI], nd, ni;
    layout(B, n, od, oi, &nd, &ni);
    NewtonView v;
    v.B = B; v.n = n;
    v.x = dws + od[OEI_NEWTON_D_X]; v.f = dws + od[OEI_NEWTON_D_F]; v.g = dws + od[OEI_NEWTON_D_G];
    v.d = dws + od[OEI_NEWTON_D_DIR]; v.t = dws + od[OEI_NEWTON_D_STEP]; v.xt = dws + od[OEI_NEWTON_D_XT];
    v.Xc = dws + od[OEI_NEWTON_D_XC]; v.Xh = dws + od[OEI_NEWTON_D_XH];
    v.lo = dws + od[OEI_NEWTON_D_LO]; v.hi = dws + od[OEI_NEWTON_D_HI];
    v.status = iws + oi[OEI_NEWTON_I_STATUS]; v.nit = iws + oi[OEI_NEWTON_I_NIT]; v.nfev = iws + oi[OEI_NEWTON_I_NFEV];
    v.nhev = iws + oi[OEI_NEWTON_I_NHEV]; v.ls = iws + oi[OEI_NEWTON_I_LS]; v.phase = iws + oi[OEI_NEWTON_I_PHASE];
    v.rowof = iws + oi[OEI_NEWTON_I_ROWOF]; v.map_t = iws + oi[OEI_NEWTON_I_MAP_T]; v.map_h = iws + oi[OEI_NEWTON_I_MAP_H];
    v.counts = iws + oi[OEI_NEWTON_I_COUNTS]; v.need = iws + oi[OEI_NEWTON_I_NEED]; v.rung = iws + oi[OEI_NEWTON_I_RUNG];
    return v;
}

// projection on the box; NaN stays NaN (fmin / fmax alone would turn it into a bound)
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v != v ? v : fmin(fmax(v, lo), hi); }

// Sum / maximum over the workgroup; every thread gets the result.  `red` holds 4 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block_max(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

__global__ __launch_bounds__(64) void k_newton_init(NewtonView V, const double* __restrict__ X0,
                                                    const double* __restrict__ lower, const double* __restrict__ upper) {
    const int r = blockIdx.x, lane = threadIdx.x, n = V.n;
    if (r == 0) for (int i = lane; i < n; i += 64) { V.lo[i] = lower[i]; V.hi[i] = upper[i]; }
    for (int i = lane; i < n; i += 64) {
        const double v = clampd(X0[(size_t)r * n + i], lower[i], upper[i]);
        V.x[(size_t)r * n + i] = v;
        V.xt[(size_t)r * n + i] = v;
        V.Xc[(size_t)r * n + i] = v;
    }
    if (lane == 0) {
        V.status[r] = ST_RUNNING; V.nit[r] = 0; V.nfev[r] = 0; V.nhev[r] = 0; V.ls[r] = 0; V.phase[r] = PH_TRIAL;
        V.rowof[r] = r; V.map_t[r] = r; V.need[r] = 1; V.rung[r] = 0;
        V.t[r] = 1.0; V.f[r] = INFINITY;
        if (r == 0) { V.counts[0] = V.B; V.counts[1] = 0; }
    }
}

// Entry (i, j) of the matrix the direction is computed from: the symmetrised, sanitised Hessian on
// the free variables, the identity on the blocked ones, plus lam on the diagonal.
__device__ __forceinline__ double masked_h(const double* __restrict__ H, int n, int i, int j,
                                           const double* fm, double lam) {
    double a = H ? H[(size_t)i * n + j] : 0.0, b = H ? H[(size_t)j * n + i] : 0.0;   // no list: a zero matrix
    if (!isfinite(a)) a = 0.0;
    if (!isfinite(b)) b = 0.0;
    double v = 0.5 * (a + b) * fm[i] * fm[j];
    if (i == j) v += (1.0 - fm[i]) + lam;
    return v;
}

// tstatus / tflags / flagged: as in k_lbfgs_step, for the rows of the trial list.
__global__ __launch_bounds__(NT) void k_newton_step(NewtonView V, const double* __restrict__ ft,
                                                    const double* __restrict__ gt, const double* __restrict__ Ht,
                                                    double pgtol, double ftol, double c1, int max_ls, int max_iter,
                                                    const int* __restrict__ tstatus, const int* __restrict__ tflags,
                                                    int* __restrict__ flagged) {
    extern __shared__ double lds[];
    const int r = blockIdx.x, tid = threadIdx.x, n = V.n;
    const int LD = n | 1;
    double* Lm = lds;                         // n x LD Cholesky factor
    double* fm = Lm + (size_t)n * LD;         // n: 1 on free variables
    double* bv = fm + n;                      // n: right-hand side, then the forward solution's workspace
    double* zv = bv + n;                      // n: solution
    double* red = zv + n;                     // 4
    if (V.status[r] != ST_RUNNING) { if (tid == 0) V.need[r] = 0; return; }
    const size_t o = (size_t)r * n;
    double* x = V.x + o; double* g = V.g + o; double* d = V.d + o; double* xt = V.xt + o;
    const double* lo = V.lo; const double* hi = V.hi;
    const int c = V.rowof[r];
    if (V.phase[r] == PH_TRIAL && tflags && tflags[c]) {       // left as it is: the caller evaluates this point another way
        if (tid == 0) { V.need[r] = 1; atomicAdd(flagged, 1); }
        return;
    }
    __syncthreads();                       // every thread has read need / phase before they change
    if (tid == 0) V.need[r] = 0;
    double fcur = V.f[r];
    double t = V.t[r];
    int nit = V.nit[r];
    if (V.phase[r] == PH_TRIAL) {
        // a run that waits for a trial value while the caller passed no list sees a rejected point
        const double fv = (ft && gt && !(tstatus && tstatus[c] != 0)) ? ft[c] : INFINITY;
        const double* gn = (ft && gt) ? gt + (size_t)c * n : g;
        const int nfev = V.nfev[r] + 1;
        __syncthreads();
        if (tid == 0) V.nfev[r] = nfev;
        if (nfev == 1) {
            for (int i = tid; i < n; i += NT) g[i] = gn[i];
            if (tid == 0) V.f[r] = fv;
            if (!isfinite(fv)) { if (tid == 0) V.status[r] = ST_BAD_START; return; }
            fcur = fv;
        } else {
            double dot = 0.0;
            for (int i = tid; i < n; i += NT) dot += g[i] * (xt[i] - x[i]);
            dot = block_sum(dot, red);
            const bool ok = fv <= fcur + c1 * dot;
            if (!ok) {
                t *= 0.25;
                const int ls = V.ls[r] + 1;
                __syncthreads();
                if (tid == 0) { V.ls[r] = ls; V.t[r] = t; }
                if (ls >= max_ls) { if (tid == 0) V.status[r] = ST_LS_FAILED; return; }
                for (int i = tid; i < n; i += NT) xt[i] = clampd(x[i] + t * d[i], lo[i], hi[i]);
                if (tid == 0) V.need[r] = 1;
                return;
            }
            const bool small = (fcur - fv) <= ftol * fmax(fmax(fabs(fcur), fabs(fv)), 1.0);
            for (int i = tid; i < n; i += NT) { x[i] = xt[i]; g[i] = gn[i]; }
            nit += 1;
            if (tid == 0) { V.f[r] = fv; V.nit[r] = nit; if (small) V.status[r] = ST_CONV_F; }
            if (small) return;
            fcur = fv;
        }
        if (nit >= max_iter) { if (tid == 0) V.status[r] = ST_ITER_LIMIT; return; }
        double pg = 0.0;
        for (int i = tid; i < n; i += NT) pg = fmax(pg, fabs(x[i] - clampd(x[i] - g[i], lo[i], hi[i])));
        pg = block_max(pg, red);
        if (pg <= pgtol) { if (tid == 0) V.status[r] = ST_CONV_PG; return; }
        if (tid == 0) { V.phase[r] = PH_HESS; V.need[r] = 2; }
        return;
    }
    // ---- Hessian at the iterate: direction and first trial point --------------------------------
    const double* H = Ht ? Ht + (size_t)c * n * n : nullptr;
    if (tid == 0) V.nhev[r] += 1;
    for (int i = tid; i < n; i += NT) {
        const bool blocked = (x[i] <= lo[i] && g[i] > 0.0) || (x[i] >= hi[i] && g[i] < 0.0);
        fm[i] = blocked ? 0.0 : 1.0;
    }
    __syncthreads();
    double amax = 0.0;
    for (int e = tid; e < n * n; e += NT) amax = fmax(amax, fabs(masked_h(H, n, e / n, e % n, fm, 0.0)));
    const double scale = fmax(block_max(amax, red), 1e-12);
    // rung k of the ladder: lam = 0 for k = 0, 1e-4 scale 10^(k-1) above; the climb starts two rungs
    // below the rung this run's previous iteration ended on
    const int first_rung = max(0, V.rung[r] - 2);
    double lam = 0.0;
    for (int k = 1; k <= first_rung; ++k) lam = (k == 1) ? 1e-4 * scale : 10.0 * lam;
    bool solved = false;
    int attempt = first_rung;
    const int ty = tid >> 4, tx = tid & 15;
    __syncthreads();                         // every thread has read the rung before thread 0 stores the new one
    for (; attempt < LADDER && !solved; ++attempt) {
        // lower triangle of the masked matrix + lam I into LDS, then a right-looking factorisation
        // in place: pivot, scaled column, rank-one update of the trailing triangle spread over the
        // 16 x 16 thread grid (three barriers per column, no long dependent chains)
        for (int i = ty; i < n; i += 16)
            for (int j = tx; j <= i; j += 16) Lm[i * LD + j] = masked_h(H, n, i, j, fm, lam);
        __syncthreads();
        bool fail = false;
        for (int j = 0; j < n; ++j) {
            const double piv = Lm[j * LD + j];           // the same word for every thread
            if (!(piv > 0.0) || !isfinite(piv)) { fail = true; break; }
            const double root = sqrt(piv), rinv = 1.0 / root;
            __syncthreads();
            for (int i = j + 1 + tid; i < n; i += NT) Lm[i * LD + j] *= rinv;
            if (tid == 0) Lm[j * LD + j] = root;
            __syncthreads();
            for (int i = j + 1 + ty; i < n; i += 16) {
                const double lij = Lm[i * LD + j];
                for (int k = j + 1 + tx; k <= i; k += 16) Lm[i * LD + k] -= lij * Lm[k * LD + j];
            }
            __syncthreads();
        }
        __syncthreads();
        if (!fail) { solved = true; break; }
        lam = (lam > 0.0) ? 10.0 * lam : 1e-4 * scale;
    }
    if (tid == 0) V.rung[r] = solved ? attempt : 0;
    double gd = 0.0;
    if (solved) {
        for (int i = tid; i < n; i += NT) bv[i] = g[i] * fm[i];
        __syncthreads();
        for (int j = 0; j < n; ++j) {                    // L z = b
            const double zj = bv[j] / Lm[j * LD + j];
            for (int i = j + 1 + tid; i < n; i += NT) bv[i] -= Lm[i * LD + j] * zj;
            if (tid == 0) zv[j] = zj;
            __syncthreads();
        }
        for (int j = n - 1; j >= 0; --j) {               // L^T w = z
            const double wj = zv[j] / Lm[j * LD + j];
            for (int i = tid; i < j; i += NT) zv[i] -= Lm[j * LD + i] * wj;
            __syncthreads();
            if (tid == 0) zv[j] = wj;
            __syncthreads();
        }
        for (int i = tid; i < n; i += NT) { const double v = -zv[i] * fm[i]; d[i] = v; gd += g[i] * v; }
        gd = block_sum(gd, red);
    }
    if (!solved || !(gd < 0.0))
        for (int i = tid; i < n; i += NT) d[i] = -g[i] * fm[i];
    double dn = 0.0, box = 0.0;
    for (int i = tid; i < n; i += NT) { dn += d[i] * d[i]; const double w = hi[i] - lo[i]; box += w * w; }
    dn = sqrt(block_sum(dn, red));
    box = sqrt(block_sum(box, red));
    if (!(dn > 0.0)) { if (tid == 0) V.status[r] = ST_STALLED; return; }
    t = fmin(box / fmax(dn, 1e-300), 1.0);
    for (int i = tid; i < n; i += NT) xt[i] = clampd(x[i] + t * d[i], lo[i], hi[i]);
    if (tid == 0) { V.t[r] = t; V.ls[r] = 0; V.phase[r] = PH_TRIAL; V.need[r] = 1; }
}

// Two dense lists in run order: runs that need a trial evaluation (need == 1) and runs that need
// a Hessian (need == 2).
__global__ __launch_bounds__(1024) void k_newton_compact(NewtonView V) {
    __shared__ int wsum[2][16];
    __shared__ int base_s[2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid < 2) base_s[tid] = 0;
    __syncthreads();
    for (int r0 = 0; r0 < V.B; r0 += 1024) {
        const int r = r0 + tid;
        const int need = (r < V.B) ? V.need[r] : 0;
        const unsigned long long m1 = __ballot(need == 1), m2 = __ballot(need == 2);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (lane == 0) { wsum[0][wid] = __popcll(m1); wsum[1][wid] = __popcll(m2); }
        __syncthreads();
        int woff[2] = {0, 0}, total[2] = {0, 0};
        for (int k = 0; k < 16; ++k)
            for (int w = 0; w < 2; ++w) { const int v = wsum[w][k]; if (k < wid) woff[w] += v; total[w] += v; }
        const int b0 = base_s[0], b1 = base_s[1];
        if (r < V.B) {
            if (need == 1) { const int cc = b0 + woff[0] + __popcll(m1 & below); V.rowof[r] = cc; V.map_t[cc] = r; }
            else if (need == 2) { const int cc = b1 + woff[1] + __popcll(m2 & below); V.rowof[r] = cc; V.map_h[cc] = r; }
            else V.rowof[r] = -1;
        }
        __syncthreads();
        if (tid == 0) { base_s[0] = b0 + total[0]; base_s[1] = b1 + total[1]; }
        __syncthreads();
    }
    if (tid < 2) V.counts[tid] = base_s[tid];
}

__global__ __launch_bounds__(64) void k_newton_gather(NewtonView V) {
    const int c = blockIdx.x, n = V.n;
    if (c < V.counts[0]) {
        const int r = V.map_t[c];
        for (int i = threadIdx.x; i < n; i += 64) V.Xc[(size_t)c * n + i] = V.xt[(size_t)r * n + i];
    }
    if (c < V.counts[1]) {
        const int r = V.map_h[c];
        for (int i = threadIdx.x; i < n; i += 64) V.Xh[(size_t)c * n + i] = V.x[(size_t)r * n + i];
    }
}

size_t step_lds_bytes(int n) { return ((size_t)n * (n | 1) + 3 * (size_t)n + 4) * sizeof(double); }

int check_shape(const char* who, int B, int n) {
    if (B < 1 || n < 1) { oei_set_error("%s: need B >= 1, n >= 1 (got %d, %d)", who, B, n); return OEI_ERR_ARG; }
    if (step_lds_bytes(n) > 160 * 1024) {
        oei_set_error("%s: n = %d variables per run exceed the LDS-resident factor (n <= %d)", who, n, OEI_NEWTON_MAX_N);
        return OEI_ERR_ARG;
    }
    return OEI_OK;
}

}  // namespace

extern "C" int oei_newton_layout(int B, int n, size_t* n_doubles, size_t* n_ints,
                                 size_t* off_doubles, size_t* off_ints) {
    const int rc = check_shape("newton_layout", B, n);
    if (rc) return rc;
    size_t od[OEI_NEWTON_ND], oi[OEI_NEWTON_NI], nd, ni;
    layout(B, n, od, oi, &nd, &ni);
    if (n_doubles) *n_doubles = nd;
    if (n_ints) *n_ints = ni;
    if (off_doubles) for (int k = 0; k < OEI_NEWTON_ND; ++k) off_doubles[k] = od[k];
    if (off_ints) for (int k = 0; k < OEI_NEWTON_NI; ++k) off_ints[k] = oi[k];
    return OEI_OK;
}

extern "C" int oei_newton_init(int B, int n, const double* X0, const double* lower, const double* upper,
                               double* dws, int* iws, void* stream) {
    const int rc = check_shape("newton_init", B, n);
    if (rc) return rc;
    if (!X0 || !lower || !upper || !dws || !iws) { oei_set_error("newton_init: null pointer"); return OEI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    NewtonView V = make_view(B, n, dws, iws);
    hipLaunchKernelGGL(k_newton_init, dim3(B), dim3(64), 0, st, V, X0, lower, upper);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// The launches of one advance; tstatus / tflags / flagged as in k_newton_step (all optional).
int oei_newton_advance_launch(int B, int n, const double* ft, const double* gt, const double* Ht,
                              const int* tstatus, const int* tflags, int* flagged,
                              double pgtol, double ftol, double c1, int max_ls, int max_iter,
                              double* dws, int* iws, hipStream_t st) {
    NewtonView V = make_view(B, n, dws, iws);
    const size_t lds = step_lds_bytes(n);
    if (lds > 48 * 1024)
        OEI_HIP_CHECK(hipFuncSetAttribute((const void*)k_newton_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_newton_step, dim3(B), dim3(NT), lds, st, V, ft, gt, Ht, pgtol, ftol, c1, max_ls, max_iter,
                       tstatus, tflags, flagged);
    hipLaunchKernelGGL(k_newton_compact, dim3(1), dim3(1024), 0, st, V);
    hipLaunchKernelGGL(k_newton_gather, dim3(B), dim3(64), 0, st, V);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

int* oei_newton_counts_ptr(int B, int n, double* dws, int* iws) { return make_view(B, n, dws, iws).counts; }
double* oei_newton_trial_list_ptr(int B, int n, double* dws, int* iws) { return make_view(B, n, dws, iws).Xc; }
double* oei_newton_hess_list_ptr(int B, int n, double* dws, int* iws) { return make_view(B, n, dws, iws).Xh; }
int oei_newton_check_shape(const char* who, int B, int n) { return check_shape(who, B, n); }

extern "C" int oei_newton_advance(int B, int n, const double* ft, const double* gt, const double* Ht,
                                  double pgtol, double ftol, double c1, int max_ls, int max_iter,
                                  double* dws, int* iws, int* counts_host, void* stream) {
    const int rc = check_shape("newton_advance", B, n);
    if (rc) return rc;
    if (!dws || !iws) { oei_set_error("newton_advance: null workspace"); return OEI_ERR_ARG; }
    if (max_ls < 1 || max_iter < 0) { oei_set_error("newton_advance: max_ls >= 1 and max_iter >= 0 required"); return OEI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int lrc = oei_newton_advance_launch(B, n, ft, gt, Ht, nullptr, nullptr, nullptr, pgtol, ftol, c1, max_ls,
                                              max_iter, dws, iws, st);
    if (lrc) return lrc;
    if (counts_host) {
        OEI_HIP_CHECK(hipMemcpyAsync(counts_host, oei_newton_counts_ptr(B, n, dws, iws), 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        OEI_HIP_CHECK(hipStreamSynchronize(st));
    }
    return OEI_OK;
}
