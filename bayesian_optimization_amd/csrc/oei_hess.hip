// Hessian path entry points (sensitivity of the SDP, dOmega/dx, acquisition Hessian).
#include "oei_sdp.cuh"

extern "C" int oei_solution_derivative_batched(int B, int k_, int K, const double* M,
                                               const double* Y, const double* fmin,
                                               const double* domega, double* dM, double* dY,
                                               int* status, void* stream) {
    oei_set_error("solution_derivative_batched: not built yet");
    return OEI_ERR_ARG;
}

extern "C" int oei_hessian_batched(oei_model* m, int B, int q, const double* X, double* H,
                                   int* status, void* stream) {
    oei_set_error("hessian_batched: not built yet");
    return OEI_ERR_ARG;
}

extern "C" int oei_domega_batched(oei_model* m, int B, int q, const double* X, double* domega,
                                  void* stream) {
    oei_set_error("domega_batched: not built yet");
    return OEI_ERR_ARG;
}
