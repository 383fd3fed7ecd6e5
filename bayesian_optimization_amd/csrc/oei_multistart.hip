// Rounds of the multistart descent on the acquisition function, driven from inside the library.
//
// One round of the lock-step optimisers (oei_lbfgs.hip, oei_newton.hip) is: evaluate the dense
// list(s) of points the running runs asked for, advance every run, read the new list length(s).
// Driven from Python a round costs more in host work (tensor bookkeeping, two synchronisations)
// than in device work at the batch sizes of a BO loop (bo.py:129-192: 20 starts).  Here the loop
// is host code of the library: the evaluations go straight to oei_value_grad_batched /
// oei_hessian_batched, a failed solve becomes a rejected trial point inside the advance kernel, and
// one 12-byte read-back per round decides whether to go on.
//
// Candidates that the reference's `project` would shrink (oei.py:157-199: a point on a training
// input, two equal points) cannot be evaluated here -- the reduced problem draws random directions
// from numpy's generator on the host.  Such a trial point stops the loop: its run is left
// untouched by the advance (still in the list), the call returns with the count of such runs, and
// the caller evaluates the current list its own way, advances once with oei_*_advance, and
// calls again.
#include "oei_common.h"

namespace {

__global__ void k_mask_hessians(int nh, size_t nn, const int* __restrict__ hstatus, const int* __restrict__ hflags,
                                double* __restrict__ Ht) {
    const int c = blockIdx.y;
    if (c >= nh) return;
    if (hstatus[c] == 0 && !(hflags && hflags[c])) return;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nn; e += (size_t)gridDim.x * blockDim.x)
        Ht[(size_t)c * nn + e] = 0.0;
}

inline size_t up32(size_t v) { return (v + 31) / 32 * 32; }

}  // namespace

extern "C" int oei_multistart_workspace(int B, int n, int with_hessian, size_t* n_doubles) {
    if (B < 1 || n < 1 || !n_doubles) { oei_set_error("multistart_workspace: bad argument"); return OEI_ERR_ARG; }
    // ft [B] | gt [B n] | Ht [B n n] (Newton) | ints: status [B], flags [B], hstatus [B], hflags [B], flagged [1]
    size_t nd = up32((size_t)B) + up32((size_t)B * n);
    if (with_hessian) nd += up32((size_t)B * n * n);
    nd += up32(((size_t)4 * B + 32) / 2 + 1);
    *n_doubles = nd;
    return OEI_OK;
}

extern "C" int oei_lbfgs_minimize_rounds(oei_model* m, int B, int q, int history,
                                         double pgtol, double ftol, double c1, int max_ls, int max_iter,
                                         double* dws, int* iws, double* work, int use_flags,
                                         int max_rounds, int* info, void* stream) {
    if (!m || !dws || !iws || !work || !info || B < 1 || q < 1 || q > OEI_MAX_Q || history < 1 || history > 64 ||
        max_ls < 1 || max_iter < 0) {
        oei_set_error("lbfgs_minimize_rounds: bad argument");
        return OEI_ERR_ARG;
    }
    const int n = q * m->d;
    hipStream_t st = (hipStream_t)stream;
    double* ft = work;
    double* gt = ft + up32((size_t)B);
    int* ib = reinterpret_cast<int*>(gt + up32((size_t)B * n));
    int* status = ib;
    int* flags = ib + B;
    int* flagged = ib + 4 * (size_t)B;
    const double* Xc = oei_lbfgs_list_ptr(B, n, history, dws, iws);
    int* count_dev = oei_lbfgs_count_ptr(B, n, history, dws, iws);
    int count = info[0], nflag = 0, rounds = 0;
    if (count < 0 || count > B) { oei_set_error("lbfgs_minimize_rounds: list length %d out of range", count); return OEI_ERR_ARG; }
    while (count > 0 && rounds < max_rounds) {
        int rc = oei_value_grad_batched(m, count, q, Xc, ft, gt, status, (use_flags & 1) ? flags : nullptr, st);
        if (rc) return rc;
        OEI_HIP_CHECK(hipMemsetAsync(flagged, 0, sizeof(int), st));
        rc = oei_lbfgs_advance_launch(B, n, history, ft, gt, status, (use_flags & 1) ? flags : nullptr, flagged, pgtol, ftol,
                                      c1, max_ls, max_iter, dws, iws, st);
        if (rc) return rc;
        OEI_HIP_CHECK(hipMemcpyAsync(&count, count_dev, sizeof(int), hipMemcpyDeviceToHost, st));
        OEI_HIP_CHECK(hipMemcpyAsync(&nflag, flagged, sizeof(int), hipMemcpyDeviceToHost, st));
        OEI_HIP_CHECK(hipStreamSynchronize(st));
        ++rounds;
        if (nflag > 0) break;
    }
    info[0] = count; info[1] = nflag; info[2] = rounds;
    return OEI_OK;
}

extern "C" int oei_newton_minimize_rounds(oei_model* m, int B, int q,
                                          double pgtol, double ftol, double c1, int max_ls, int max_iter,
                                          double* dws, int* iws, double* work, int use_flags,
                                          int max_rounds, int* info, void* stream) {
    if (!m || !dws || !iws || !work || !info || B < 1 || q < 1 || q > OEI_MAX_Q || max_ls < 1 || max_iter < 0) {
        oei_set_error("newton_minimize_rounds: bad argument");
        return OEI_ERR_ARG;
    }
    const int n = q * m->d;
    int rc = oei_newton_check_shape("newton_minimize_rounds", B, n);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t nn = (size_t)n * n;
    double* ft = work;
    double* gt = ft + up32((size_t)B);
    double* Ht = gt + up32((size_t)B * n);
    int* ib = reinterpret_cast<int*>(Ht + up32((size_t)B * nn));
    int* status = ib;
    int* flags = ib + B;
    int* hstatus = ib + 2 * (size_t)B;
    int* hflags = ib + 3 * (size_t)B;
    int* flagged = ib + 4 * (size_t)B;
    const double* Xc = oei_newton_trial_list_ptr(B, n, dws, iws);
    const double* Xh = oei_newton_hess_list_ptr(B, n, dws, iws);
    int* counts_dev = oei_newton_counts_ptr(B, n, dws, iws);
    int counts[2] = {info[0], info[1]}, nflag = 0, rounds = 0;
    if (counts[0] < 0 || counts[0] > B || counts[1] < 0 || counts[1] > B) { oei_set_error("newton_minimize_rounds: list lengths out of range"); return OEI_ERR_ARG; }
    while ((counts[0] > 0 || counts[1] > 0) && rounds < max_rounds) {
        if (counts[0] > 0) {
            rc = oei_value_grad_batched(m, counts[0], q, Xc, ft, gt, status, (use_flags & 1) ? flags : nullptr, st);
            if (rc) return rc;
        }
        if (counts[1] > 0) {
            rc = oei_hessian_batched(m, counts[1], q, Xh, Ht, hstatus, st);
            if (rc) return rc;
            if ((use_flags & 2) && (rc = oei_project_flags(m, counts[1], q, Xh, hflags, st))) return rc;
            // acquisition_hessian is zero where project would shrink the candidate (oei.py:53-55); a
            // failed solve gives a zero matrix too (the step then falls back to the gradient)
            hipLaunchKernelGGL(k_mask_hessians, dim3(8, counts[1]), dim3(256), 0, st, counts[1], nn, hstatus,
                               (use_flags & 2) ? hflags : (const int*)nullptr, Ht);
            OEI_HIP_CHECK(hipGetLastError());
        }
        OEI_HIP_CHECK(hipMemsetAsync(flagged, 0, sizeof(int), st));
        rc = oei_newton_advance_launch(B, n, ft, gt, Ht, status, (use_flags & 1) ? flags : nullptr, flagged, pgtol, ftol, c1,
                                       max_ls, max_iter, dws, iws, st);
        if (rc) return rc;
        OEI_HIP_CHECK(hipMemcpyAsync(counts, counts_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        OEI_HIP_CHECK(hipMemcpyAsync(&nflag, flagged, sizeof(int), hipMemcpyDeviceToHost, st));
        OEI_HIP_CHECK(hipStreamSynchronize(st));
        ++rounds;
        if (nflag > 0) break;
    }
    info[0] = counts[0]; info[1] = counts[1]; info[2] = nflag; info[3] = rounds;
    return OEI_OK;
}
