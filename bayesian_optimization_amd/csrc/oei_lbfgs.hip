// Multistart quasi-Newton driver on the device: B bounded L-BFGS runs, one wave per run.
//
// The reference optimises the acquisition function from every random start with scipy's
// L-BFGS-B, one start after the other (bo.py:129-192, solvers.py:10-21).  Here every run is a
// small state machine in global memory.  One round = one batched evaluation of the point each
// running run asked for, then k_lbfgs_step consumes the results: Armijo test of the trial point,
// on acceptance the curvature pair, the stopping tests, the next two-loop direction and the first
// trial point of the next iteration; on rejection a halved step.  Runs do not wait for each other:
// a run in its line search and a run starting a new iteration share the same evaluation call.
// k_lbfgs_compact + k_lbfgs_gather turn the runs that need an evaluation into a dense list (in
// run order, so the list is the same from call to call) and the host reads back its length.
//
// Per run the method is the projected L-BFGS of lockstep.py (same formulae, same constants):
// projected-gradient and relative-decrease stopping rules of L-BFGS-B, two-loop recursion over
// the last `history` pairs on the variables not blocked at a bound, projected backtracking.
#include "oei_common.h"

namespace {

enum { ST_RUNNING = 0, ST_CONV_PG = 1, ST_CONV_F = 2, ST_LS_FAILED = 3, ST_STALLED = 4,
       ST_BAD_START = 5, ST_ITER_LIMIT = 6 };

struct LbfgsView {
    int B, n, m;
    double *x, *f, *g, *d, *t, *xt, *S, *Y, *rho, *gamma, *Xc, *lo, *hi;
    int *status, *slot, *nit, *nfev, *ls, *first, *rowof, *map, *count, *need;
};

__host__ __device__ inline size_t up32(size_t v) { return (v + 31) / 32 * 32; }

// Offsets (in elements) of the arrays inside the two workspaces; the order is the order of the
// OEI_LBFGS_D_* / OEI_LBFGS_I_* indices of the header.
void layout(int B, int n, int m, size_t* off_d, size_t* off_i, size_t* nd, size_t* ni) {
    const size_t Bn = (size_t)B * n;
    const size_t dsz[OEI_LBFGS_ND] = {Bn, (size_t)B, Bn, Bn, (size_t)B, Bn, Bn * m, Bn * m, (size_t)B * m,
                                      (size_t)B, Bn, (size_t)n, (size_t)n};
    size_t o = 0;
    for (int k = 0; k < OEI_LBFGS_ND; ++k) { off_d[k] = o; o += up32(dsz[k]); }
    *nd = o;
    o = 0;
    for (int k = 0; k < OEI_LBFGS_NI; ++k) { off_i[k] = o; o += up32(k == OEI_LBFGS_I_COUNT ? 1 : (size_t)B); }
    *ni = o;
}

LbfgsView make_view(int B, int n, int m, double* dws, int* iws) {
    size_t od[OEI_LBFGS_ND], oi[OEI_LBFGS_NI], nd, ni;
    layout(B, n, m, od, oi, &nd, &ni);
    LbfgsView v;
    v.B = B; v.n = n; v.m = m;
    v.x = dws + od[OEI_LBFGS_D_X]; v.f = dws + od[OEI_LBFGS_D_F]; v.g = dws + od[OEI_LBFGS_D_G];
    v.d = dws + od[OEI_LBFGS_D_DIR]; v.t = dws + od[OEI_LBFGS_D_STEP]; v.xt = dws + od[OEI_LBFGS_D_XT];
    v.S = dws + od[OEI_LBFGS_D_S]; v.Y = dws + od[OEI_LBFGS_D_Y]; v.rho = dws + od[OEI_LBFGS_D_RHO];
    v.gamma = dws + od[OEI_LBFGS_D_GAMMA]; v.Xc = dws + od[OEI_LBFGS_D_XC];
    v.lo = dws + od[OEI_LBFGS_D_LO]; v.hi = dws + od[OEI_LBFGS_D_HI];
    v.status = iws + oi[OEI_LBFGS_I_STATUS]; v.slot = iws + oi[OEI_LBFGS_I_SLOT]; v.nit = iws + oi[OEI_LBFGS_I_NIT];
    v.nfev = iws + oi[OEI_LBFGS_I_NFEV]; v.ls = iws + oi[OEI_LBFGS_I_LS]; v.first = iws + oi[OEI_LBFGS_I_FIRST];
    v.rowof = iws + oi[OEI_LBFGS_I_ROWOF]; v.map = iws + oi[OEI_LBFGS_I_MAP]; v.count = iws + oi[OEI_LBFGS_I_COUNT];
    v.need = iws + oi[OEI_LBFGS_I_NEED];
    return v;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// projection on the box; NaN stays NaN (fmin / fmax alone would turn it into a bound)
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v != v ? v : fmin(fmax(v, lo), hi); }

// Start: x = P(X0), every run asks for the evaluation of its start (list = identity).
__global__ __launch_bounds__(64) void k_lbfgs_init(LbfgsView V, const double* __restrict__ X0,
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
        V.status[r] = ST_RUNNING; V.slot[r] = 0; V.nit[r] = 0; V.nfev[r] = 0; V.ls[r] = 0; V.first[r] = 1;
        V.rowof[r] = r; V.map[r] = r; V.need[r] = 1;
        V.gamma[r] = 1.0; V.t[r] = 1.0; V.f[r] = INFINITY;
        for (int k = 0; k < V.m; ++k) V.rho[(size_t)r * V.m + k] = 0.0;
        if (r == 0) *V.count = V.B;
    }
}

// One wave per run.  Every running run has exactly one evaluation outstanding (row rowof[r] of
// ft / gt): the value and gradient at xt.  Lane l owns the vector elements l, l + 64, ..., so a
// vector written by this wave is read back by the lanes that wrote it.
// tstatus (optional, per list row): a non-zero entry makes the row's value +inf (an evaluation that
// did not solve is a rejected trial point).  tflags (optional, per list row) with `flagged`: a run
// whose row is flagged is left exactly as it is -- its trial point stays in the next list -- and
// counted, so that the caller can evaluate that point another way and advance again.
__global__ __launch_bounds__(64) void k_lbfgs_step(LbfgsView V, const double* __restrict__ ft,
                                                   const double* __restrict__ gt, double pgtol, double ftol,
                                                   double c1, int max_ls, int max_iter,
                                                   const int* __restrict__ tstatus, const int* __restrict__ tflags,
                                                   int* __restrict__ flagged) {
    const int r = blockIdx.x, lane = threadIdx.x, n = V.n, m = V.m;
    if (V.status[r] != ST_RUNNING) { if (lane == 0) V.need[r] = 0; return; }
    const size_t o = (size_t)r * n;
    double* x = V.x + o; double* g = V.g + o; double* d = V.d + o; double* xt = V.xt + o;
    const double* lo = V.lo; const double* hi = V.hi;
    const int c = V.rowof[r];
    if (tflags && tflags[c]) {
        if (lane == 0) { V.need[r] = 1; atomicAdd(flagged, 1); }
        return;
    }
    if (lane == 0) V.need[r] = 0;
    const double fv = (tstatus && tstatus[c] != 0) ? INFINITY : ft[c];
    const double* gn = gt + (size_t)c * n;
    const int nfev = V.nfev[r] + 1;
    if (lane == 0) V.nfev[r] = nfev;
    // the run's scalars live in registers of every lane; lane 0 writes the changes back
    double fcur = V.f[r];
    double t = V.t[r];
    double gam = V.gamma[r];
    int nit = V.nit[r], slot = V.slot[r], first = V.first[r];
    bool new_iteration = false;
    int fresh = -1;                                    // pair written in this call: its rho is taken
    double rho_fresh = 0.0;                            // from the register, not read back
    if (nfev == 1) {                                   // the start itself
        for (int i = lane; i < n; i += 64) g[i] = gn[i];
        if (lane == 0) V.f[r] = fv;
        if (!isfinite(fv)) { if (lane == 0) V.status[r] = ST_BAD_START; return; }
        fcur = fv;
        new_iteration = true;
    } else {
        double dot = 0.0;
        for (int i = lane; i < n; i += 64) dot += g[i] * (xt[i] - x[i]);
        dot = wave_sum(dot);
        const bool ok = fv <= fcur + c1 * dot;        // false for NaN and +inf
        if (ok) {
            double sy = 0.0, ss = 0.0, yy = 0.0;
            for (int i = lane; i < n; i += 64) {
                const double s = xt[i] - x[i], y = gn[i] - g[i];
                sy += s * y; ss += s * s; yy += y * y;
            }
            sy = wave_sum(sy); ss = wave_sum(ss); yy = wave_sum(yy);
            const bool valid = sy > 1e-10 * sqrt(ss) * sqrt(yy);
            double* Sr = V.S + ((size_t)r * m + slot) * n;
            double* Yr = V.Y + ((size_t)r * m + slot) * n;
            for (int i = lane; i < n; i += 64) {
                const double s = xt[i] - x[i], y = gn[i] - g[i];
                Sr[i] = valid ? s : 0.0;
                Yr[i] = valid ? y : 0.0;
                x[i] = xt[i];
                g[i] = gn[i];
            }
            const bool small = (fcur - fv) <= ftol * fmax(fmax(fabs(fcur), fabs(fv)), 1.0);
            const double rho_new = valid ? 1.0 / sy : 0.0;
            if (valid) gam = sy / fmax(yy, 1e-300);
            if (lane == 0) {
                V.rho[(size_t)r * m + slot] = rho_new;
                V.gamma[r] = gam;
                V.slot[r] = (slot + 1) % m;
                V.nit[r] = nit + 1;
                V.first[r] = 0;
                V.f[r] = fv;
                if (small) V.status[r] = ST_CONV_F;
            }
            if (small) return;
            fresh = slot; rho_fresh = rho_new;
            slot = (slot + 1) % m;
            nit += 1;
            first = 0;
            fcur = fv;
            new_iteration = true;
        } else {
            t *= 0.5;
            const int ls = V.ls[r] + 1;
            if (lane == 0) { V.ls[r] = ls; V.t[r] = t; }
            if (ls >= max_ls) { if (lane == 0) V.status[r] = ST_LS_FAILED; return; }
        }
    }
    if (new_iteration) {
        if (nit >= max_iter) { if (lane == 0) V.status[r] = ST_ITER_LIMIT; return; }
        double pg = 0.0;
        for (int i = lane; i < n; i += 64) pg = fmax(pg, fabs(x[i] - clampd(x[i] - g[i], lo[i], hi[i])));
        pg = wave_max(pg);
        if (pg <= pgtol) { if (lane == 0) V.status[r] = ST_CONV_PG; return; }
        // two-loop recursion on the free variables, newest pair first; d is the work vector
        for (int i = lane; i < n; i += 64) {
            const bool blocked = (x[i] <= lo[i] && g[i] > 0.0) || (x[i] >= hi[i] && g[i] < 0.0);
            d[i] = blocked ? 0.0 : g[i];
        }
        const double* rho = V.rho + (size_t)r * m;
        double al = 0.0;                               // lane k keeps alpha of the k-th newest pair
        for (int k = 0; k < m; ++k) {
            const int j = (slot - 1 - k + 2 * m) % m;
            const double rj = (j == fresh) ? rho_fresh : rho[j];
            if (rj == 0.0) continue;                   // empty or rejected pair: no contribution
            const double* Sj = V.S + ((size_t)r * m + j) * n;
            const double* Yj = V.Y + ((size_t)r * m + j) * n;
            double a = 0.0;
            for (int i = lane; i < n; i += 64) a += Sj[i] * d[i];
            a = rj * wave_sum(a);
            if (lane == (k & 63)) al = a;
            for (int i = lane; i < n; i += 64) d[i] -= a * Yj[i];
        }
        for (int i = lane; i < n; i += 64) d[i] *= gam;
        for (int k = m - 1; k >= 0; --k) {
            const int j = (slot - 1 - k + 2 * m) % m;
            const double rj = (j == fresh) ? rho_fresh : rho[j];
            if (rj == 0.0) continue;
            const double* Sj = V.S + ((size_t)r * m + j) * n;
            const double* Yj = V.Y + ((size_t)r * m + j) * n;
            double b = 0.0;
            for (int i = lane; i < n; i += 64) b += Yj[i] * d[i];
            b = rj * wave_sum(b);
            const double a = __shfl(al, k & 63, 64);
            for (int i = lane; i < n; i += 64) d[i] += (a - b) * Sj[i];
        }
        double gd = 0.0;
        for (int i = lane; i < n; i += 64) {
            const bool blocked = (x[i] <= lo[i] && g[i] > 0.0) || (x[i] >= hi[i] && g[i] < 0.0);
            const double v = blocked ? 0.0 : -d[i];
            d[i] = v;
            gd += g[i] * v;
        }
        gd = wave_sum(gd);
        if (!(gd < 0.0)) {                             // not a descent direction: projected steepest descent
            for (int i = lane; i < n; i += 64) {
                const bool blocked = (x[i] <= lo[i] && g[i] > 0.0) || (x[i] >= hi[i] && g[i] < 0.0);
                d[i] = blocked ? 0.0 : -g[i];
            }
        }
        double dn = 0.0;
        for (int i = lane; i < n; i += 64) dn += d[i] * d[i];
        dn = sqrt(wave_sum(dn));
        if (!(dn > 0.0)) { if (lane == 0) V.status[r] = ST_STALLED; return; }
        t = first ? fmin(1.0 / fmax(dn, 1e-300), 1.0) : 1.0;
        if (lane == 0) { V.t[r] = t; V.ls[r] = 0; }
    }
    for (int i = lane; i < n; i += 64) xt[i] = clampd(x[i] + t * d[i], lo[i], hi[i]);
    if (lane == 0) V.need[r] = 1;
}

// Dense list of the runs that need an evaluation, in run order: rowof[r], map[c], count.
__global__ __launch_bounds__(1024) void k_lbfgs_compact(LbfgsView V) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int r0 = 0; r0 < V.B; r0 += 1024) {
        const int r = r0 + tid;
        const int need = (r < V.B) ? V.need[r] : 0;
        const unsigned long long mask = __ballot(need);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wid] = __popcll(mask);
        __syncthreads();
        int woff = 0, total = 0;
        for (int k = 0; k < 16; ++k) { const int v = wsum[k]; if (k < wid) woff += v; total += v; }
        const int base = base_s;
        if (r < V.B) {
            if (need) { const int c = base + woff + before; V.rowof[r] = c; V.map[c] = r; }
            else V.rowof[r] = -1;
        }
        __syncthreads();
        if (tid == 0) base_s = base + total;
        __syncthreads();
    }
    if (tid == 0) *V.count = base_s;
}

__global__ __launch_bounds__(64) void k_lbfgs_gather(LbfgsView V) {
    const int c = blockIdx.x;
    if (c >= *V.count) return;
    const int r = V.map[c], n = V.n;
    for (int i = threadIdx.x; i < n; i += 64) V.Xc[(size_t)c * n + i] = V.xt[(size_t)r * n + i];
}

int check_shape(const char* who, int B, int n, int history) {
    if (B < 1 || n < 1 || history < 1 || history > 64) {
        oei_set_error("%s: need B >= 1, n >= 1, 1 <= history <= 64 (got %d, %d, %d)", who, B, n, history);
        return OEI_ERR_ARG;
    }
    return OEI_OK;
}

}  // namespace

extern "C" int oei_lbfgs_layout(int B, int n, int history, size_t* n_doubles, size_t* n_ints,
                                size_t* off_doubles, size_t* off_ints) {
    const int rc = check_shape("lbfgs_layout", B, n, history);
    if (rc) return rc;
    size_t od[OEI_LBFGS_ND], oi[OEI_LBFGS_NI], nd, ni;
    layout(B, n, history, od, oi, &nd, &ni);
    if (n_doubles) *n_doubles = nd;
    if (n_ints) *n_ints = ni;
    if (off_doubles) for (int k = 0; k < OEI_LBFGS_ND; ++k) off_doubles[k] = od[k];
    if (off_ints) for (int k = 0; k < OEI_LBFGS_NI; ++k) off_ints[k] = oi[k];
    return OEI_OK;
}

extern "C" int oei_lbfgs_init(int B, int n, int history, const double* X0, const double* lower,
                              const double* upper, double* dws, int* iws, void* stream) {
    const int rc = check_shape("lbfgs_init", B, n, history);
    if (rc) return rc;
    if (!X0 || !lower || !upper || !dws || !iws) { oei_set_error("lbfgs_init: null pointer"); return OEI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    LbfgsView V = make_view(B, n, history, dws, iws);
    hipLaunchKernelGGL(k_lbfgs_init, dim3(B), dim3(64), 0, st, V, X0, lower, upper);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// The launches of one advance; tstatus / tflags / flagged as in k_lbfgs_step (all optional).
int oei_lbfgs_advance_launch(int B, int n, int history, const double* ft, const double* gt,
                             const int* tstatus, const int* tflags, int* flagged,
                             double pgtol, double ftol, double c1, int max_ls, int max_iter,
                             double* dws, int* iws, hipStream_t st) {
    LbfgsView V = make_view(B, n, history, dws, iws);
    hipLaunchKernelGGL(k_lbfgs_step, dim3(B), dim3(64), 0, st, V, ft, gt, pgtol, ftol, c1, max_ls, max_iter,
                       tstatus, tflags, flagged);
    hipLaunchKernelGGL(k_lbfgs_compact, dim3(1), dim3(1024), 0, st, V);
    hipLaunchKernelGGL(k_lbfgs_gather, dim3(B), dim3(64), 0, st, V);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// device address of COUNT and of the list XC inside the workspaces
int* oei_lbfgs_count_ptr(int B, int n, int history, double* dws, int* iws) { return make_view(B, n, history, dws, iws).count; }
double* oei_lbfgs_list_ptr(int B, int n, int history, double* dws, int* iws) { return make_view(B, n, history, dws, iws).Xc; }

extern "C" int oei_lbfgs_advance(int B, int n, int history, const double* ft, const double* gt,
                                 double pgtol, double ftol, double c1, int max_ls, int max_iter,
                                 double* dws, int* iws, int* count_host, void* stream) {
    const int rc = check_shape("lbfgs_advance", B, n, history);
    if (rc) return rc;
    if (!ft || !gt || !dws || !iws) { oei_set_error("lbfgs_advance: null pointer"); return OEI_ERR_ARG; }
    if (max_ls < 1 || max_iter < 0) { oei_set_error("lbfgs_advance: max_ls >= 1 and max_iter >= 0 required"); return OEI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int lrc = oei_lbfgs_advance_launch(B, n, history, ft, gt, nullptr, nullptr, nullptr, pgtol, ftol, c1,
                                             max_ls, max_iter, dws, iws, st);
    if (lrc) return lrc;
    if (count_host) {
        OEI_HIP_CHECK(hipMemcpyAsync(count_host, oei_lbfgs_count_ptr(B, n, history, dws, iws), sizeof(int), hipMemcpyDeviceToHost, st));
        OEI_HIP_CHECK(hipStreamSynchronize(st));
    }
    return OEI_OK;
}
