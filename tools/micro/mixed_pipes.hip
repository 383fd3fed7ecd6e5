// Microbenchmark: do fp64 MFMA work of one wave and fp64 vector work of another wave on the same
// SIMD overlap?  One 512-thread workgroup per compute unit = two waves per SIMD (wave w and wave
// w + 4 share SIMD w).  Modes: both waves MFMA, both waves FMA, one of each.  Prints the cycles
// each role needs per instruction in each mode: with independent pipes the mixed mode costs each
// role what it costs alone on the SIMD; with a shared pipe it costs the sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ long long g_cyc[2][8];

// role: 0 = MFMA chain (8 independent accumulators), 1 = v_fma_f64 chains (16 accumulators)
__global__ __launch_bounds__(512) void k_mix(double* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const int role = mode == 0 ? 0 : mode == 1 ? 1 : (wave < 4 ? 0 : 1);
    __syncthreads();
    const long long c0 = clock64();
    double s = 0.0;
    if (role == 0) {
        double4_t acc[8];
        for (int t = 0; t < 8; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
        double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    } else {
        double acc[16];
        for (int t = 0; t < 16; ++t) acc[t] = t;
        double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-6;
        for (int i = 0; i < 16 * iters; ++i) {       // 16 x the instruction count of the MFMA role per iteration
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = fma(acc[t], a, b);
        }
        for (int t = 0; t < 16; ++t) s += acc[t];
    }
    const long long c1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_cyc[role][wave] = c1 - c0;
}

int main() {
    double* out;
    const int blocks = 256;
    (void)hipMalloc(&out, blocks * 512 * sizeof(double));
    const int iters = 4000;
    const char* names[3] = {"all waves MFMA", "all waves FMA", "waves 0-3 MFMA, waves 4-7 FMA"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            long long zero[2][8] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cyc), zero, sizeof(zero));
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(512), 0, 0, out, iters, mode);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            long long cyc[2][8];
            (void)hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
            double m = 0, f = 0; int nm = 0, nf = 0;
            for (int w = 0; w < 8; ++w) { if (cyc[0][w]) { m += cyc[0][w]; ++nm; } if (cyc[1][w]) { f += cyc[1][w]; ++nf; } }
            printf("%-32s %.2f ms:", names[mode], ms);
            if (nm) printf("  %.1f cycles per MFMA per wave", m / nm / (iters * 8.0));
            if (nf) printf("  %.2f cycles per v_fma_f64 per wave", f / nf / (iters * 16.0 * 16.0));
            printf("\n");
        }
    (void)hipFree(out);
    return 0;
}
