// Microbenchmark: throughput of v_mfma_f64_16x16x4 as a function of the number of waves per SIMD
// that issue it (independent accumulator chains, no memory traffic).  One workgroup per compute
// unit with 4 * W waves; W = 1, 2, 3, 4, 6, 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef LONG
#define LONG 1
#endif
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(1024) void k_mfma(double* out, int iters) {
    double4_t acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0.0;
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(double* out, int wps) {
    const int blocks = 256, threads = 256 * wps > 1024 ? 1024 : 256 * wps;
    const int nblk = 256 * wps > 1024 ? blocks * (256 * wps / 1024) : blocks;      // 8 waves per SIMD: two workgroups per CU
    const int iters = 80000 / NACC / wps * 2 * LONG;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma<NACC>, dim3(nblk), dim3(threads), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)nblk * (threads / 64) * iters * NACC * 2048.0;
    printf("%d waves per SIMD, %d accumulators per wave: %.2f ms  %.1f TFLOP/s\n", wps, NACC, best, flop / best / 1e9);
}

int main() {
    double* out;
    (void)hipMalloc(&out, 512 * 1024 * sizeof(double));
    for (int wps : {1, 2, 3, 4, 8}) run<8>(out, wps);
    for (int wps : {1, 2, 3, 4, 8}) run<4>(out, wps);
    for (int wps : {1, 2, 4}) run<2>(out, wps);
    (void)hipFree(out);
    return 0;
}
