// Microbenchmark: cycles of one q x q x q product of LDS matrices by a 256-thread workgroup on the
// matrix pipe (wg4_mm_mfma of oei_sens.cuh: wave w computes a 32 x 32 block as 2 x 2 tiles), one
// workgroup per compute unit, against variants of the step loop.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I bayesian_optimization_amd/csrc -I include -o /tmp/mm_lds tools/micro/mm_lds.hip && /tmp/mm_lds [q]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "oei_sens.cuh"

// variant 1: the step loop unrolled by two, both steps' reads issued before the first matrix instruction
template <bool TA, bool TB>
__device__ __forceinline__ void mm_unroll2(int lane, int q, int LD, const lds_f64* A, const lds_f64* B, lds_f64* C, int I0, int J0) {
    const int li = lane & 15, kg = lane >> 4;
    sens_double4 acc[2][2];
    for (int I = 0; I < 2; ++I) for (int J = 0; J < 2; ++J) acc[I][J] = (sens_double4){0.0, 0.0, 0.0, 0.0};
    int ic[2], jc[2]; bool iin[2], jin[2];
    for (int I = 0; I < 2; ++I) {
        const int i = 16 * (I0 + I) + li, j = 16 * (J0 + I) + li;
        iin[I] = i < q; ic[I] = iin[I] ? i : 0; jin[I] = j < q; jc[I] = jin[I] ? j : 0;
    }
    const int ks = (q + 3) >> 2;
    for (int s = 0; s < ks; s += 2) {
        double av[2][2], bv[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = 4 * (s + u) + kg;
            const bool kin = k < q && s + u < ks;
            const int kc = kin ? k : 0;
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const double x = A[TA ? kc * LD + ic[I] : ic[I] * LD + kc];
                const double y = B[TB ? jc[I] * LD + kc : kc * LD + jc[I]];
                av[u][I] = (kin && iin[I]) ? x : 0.0;
                bv[u][I] = (kin && jin[I]) ? y : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J)
                    acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][I], bv[u][J], acc[I][J], 0, 0, 0);
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

// variant 2: lane pointers advanced per step instead of index arithmetic, no row / column masks (rows
// and columns past q only reach entries of C that are not stored), the next step's reads issued
// before this step's matrix instructions, the partial last step peeled
template <bool TA, bool TB>
__device__ __forceinline__ void mm_fast(int lane, int q, int LD, const lds_f64* A, const lds_f64* B, lds_f64* C, int I0, int J0) {
    const int li = lane & 15, kg = lane >> 4;
    sens_double4 acc[2][2];
    for (int I = 0; I < 2; ++I) for (int J = 0; J < 2; ++J) acc[I][J] = (sens_double4){0.0, 0.0, 0.0, 0.0};
    const lds_f64* pa[2];
    const lds_f64* pb[2];
#pragma unroll
    for (int I = 0; I < 2; ++I) {
        const int i = 16 * (I0 + I) + li, j = 16 * (J0 + I) + li;
        const int ic = i < q ? i : 0, jc = j < q ? j : 0;
        pa[I] = A + (TA ? kg * LD + ic : ic * LD + kg);
        pb[I] = B + (TB ? jc * LD + kg : kg * LD + jc);
    }
    const int sa = TA ? 4 * LD : 4, sb = TB ? 4 : 4 * LD;
    const int ksf = q >> 2;                                  // steps whose four k positions all lie inside q
    double av[2], bv[2], an[2], bn[2];
#pragma unroll
    for (int I = 0; I < 2; ++I) { av[I] = pa[I][0]; bv[I] = pb[I][0]; }
    for (int s = 0; s < ksf; ++s) {
        const bool more = s + 1 < ksf;
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            pa[I] += more ? sa : 0; pb[I] += more ? sb : 0;
            an[I] = pa[I][0]; bn[I] = pb[I][0];
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
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            const int i = 16 * (I0 + I) + li, j = 16 * (J0 + I) + li;
            const int ic = i < q ? i : 0, jc = j < q ? j : 0;
            const double x = A[TA ? kc * LD + ic : ic * LD + kc], y = B[TB ? jc * LD + kc : kc * LD + jc];
            av[I] = kin ? x : 0.0; bv[I] = kin ? y : 0.0;
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

template <int VAR, bool TA, bool TB>
__global__ __launch_bounds__(256) void k_mm(int q, int reps, long long* out, double* sink) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int LD = q | 1, tid = threadIdx.x;
    lds_f64* A = (lds_f64*)sm; lds_f64* B = A + q * LD; lds_f64* C = B + q * LD;
    for (int e = tid; e < q * LD; e += 256) { A[e] = 1e-3 * (e % 17); B[e] = 1e-3 * (e % 13); C[e] = 0.0; }
    __syncthreads();
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        const int wave = tid >> 6, I0 = 2 * (wave >> 1), J0 = 2 * (wave & 1);
        if (VAR == 0) wg4_mm_mfma<TA, TB, false>(q, LD, A, B, C);
        else if (VAR == 1) { if (16 * I0 < q && 16 * J0 < q) mm_unroll2<TA, TB>(tid & 63, q, LD, A, B, C, I0, J0); }
        else if (16 * I0 < q && 16 * J0 < q) mm_fast<TA, TB>(tid & 63, q, LD, A, B, C, I0, J0);
        __syncthreads();
    }
    const long long t1 = clock64();
    if (tid == 0) out[blockIdx.x] = (t1 - t0) / reps;
    if (tid == 0) sink[blockIdx.x] = C[(q - 1) * LD + q - 1];
}

int main(int argc, char** argv) {
    const int q = argc > 1 ? atoi(argv[1]) : 50, reps = 200, nb = 256;
    long long* out; double* sink;
    hipMalloc(&out, nb * sizeof(long long)); hipMalloc(&sink, nb * sizeof(double));
    const size_t lds = (size_t)3 * q * (q | 1) * sizeof(double);
    std::vector<long long> h(nb);
#define RUN(VAR, TA, TB, name) do { \
        hipFuncSetAttribute((const void*)k_mm<VAR, TA, TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_mm<VAR, TA, TB>), dim3(nb), dim3(256), lds, 0, q, reps, out, sink); \
        hipDeviceSynchronize(); hipMemcpy(h.data(), out, nb * sizeof(long long), hipMemcpyDeviceToHost); \
        long long s = 0; for (auto v : h) s += v; \
        printf("%-44s q=%d  %lld cycles per product (+ barrier)\n", name, q, s / nb); } while (0)
    RUN(0, false, false, "wg4_mm_mfma  C = A B");
    RUN(0, false, true, "wg4_mm_mfma  C = A B^T");
    RUN(0, true, false, "wg4_mm_mfma  C = A^T B");
    RUN(1, false, false, "two steps per iteration  C = A B");
    RUN(1, false, true, "two steps per iteration  C = A B^T");
    RUN(1, true, false, "two steps per iteration  C = A^T B");
    RUN(2, false, false, "pointers, reads one step ahead  C = A B");
    RUN(2, false, true, "pointers, reads one step ahead  C = A B^T");
    RUN(2, true, false, "pointers, reads one step ahead  C = A^T B");
    return 0;
}
