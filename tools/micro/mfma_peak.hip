// Microbenchmark: sustained rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 with all CUs busy and
// no memory traffic (what a kernel on this chip can reach at most, clocks as they settle under load).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ long long g_ticks[4];     // core-clock cycles and 100 MHz ticks of block 0, per kernel

__global__ __launch_bounds__(256) void k_mfma(double* out, int iters) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double4_t acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0.0;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_ticks[0] = clock64() - c0; g_ticks[1] = wall_clock64() - w0; }
}

__global__ __launch_bounds__(256) void k_fma(double* out, int iters) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = t;
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = fma(acc[t], a, b);
    }
    double s = 0.0;
    for (int t = 0; t < 16; ++t) s += acc[t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_ticks[2] = clock64() - c0; g_ticks[3] = wall_clock64() - w0; }
}

int main() {
    double* out;
    const int blocks = 256 * 8;
    (void)hipMalloc(&out, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        const int iters = 20000;
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)blocks * 4 * iters * 8 * 2048.0;
        printf("mfma_f64_16x16x4: %.2f ms  %.1f TFLOP/s\n", ms, flop / ms / 1e9);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        flop = (double)blocks * 256 * iters * 16 * 2.0;
        printf("v_fma_f64:        %.2f ms  %.1f TFLOP/s\n", ms, flop / ms / 1e9);
    }
    long long ticks[4];
    (void)hipMemcpyFromSymbol(ticks, HIP_SYMBOL(g_ticks), sizeof(ticks));
    printf("block 0: mfma kernel %lld clock cycles in %lld ticks of 100 MHz -> %.0f MHz; fma kernel %lld in %lld -> %.0f MHz\n",
           ticks[0], ticks[1], 100.0 * ticks[0] / ticks[1], ticks[2], ticks[3], 100.0 * ticks[2] / ticks[3]);
    printf("block 0: %.1f cycles per MFMA per wave slot, %.2f cycles per v_fma_f64 per wave\n",
           (double)ticks[0] / (20000.0 * 8), (double)ticks[2] / (20000.0 * 16));
    (void)hipFree(out);
    return 0;
}
