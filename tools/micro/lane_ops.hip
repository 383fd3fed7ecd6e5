// Checks the lane-exchange primitives the tournament Jacobi relies on: sum over the four 16-lane rows
// with v_permlane16_swap / v_permlane32_swap, and one-lane shifts inside a row with DPP.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rowsum4(double v) {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    const unsigned xl = __double2loint(x), xh = __double2hiint(x);
    auto c = __builtin_amdgcn_permlane32_swap(xl, xl, false, false);
    auto d = __builtin_amdgcn_permlane32_swap(xh, xh, false, false);
    return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}
template <int CTRL>
__device__ __forceinline__ double dppd(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
    const int lane = threadIdx.x;
    const double v = 1.0 + lane * 1.0009765625;
    out[lane] = rowsum4(v);
    out[64 + lane] = dppd<0x111>(-1.0, v);      // row_shr:1: lane k takes lane k-1
    out[128 + lane] = dppd<0x101>(-2.0, v);     // row_shl:1: lane k takes lane k+1
}
int main() {
    double* d; double h[192];
    (void)hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int k = l & 15;
        double s = 0; for (int p = 0; p < 4; ++p) s += 1.0 + (16 * p + k) * 1.0009765625;
        if (h[l] != h[k] || fabs(h[l] - s) > 1e-12) { ++bad; printf("rowsum lane %d: %.17g expected %.17g\n", l, h[l], s); }
        const double shr = k == 0 ? -1.0 : 1.0 + (l - 1) * 1.0009765625, shl = k == 15 ? -2.0 : 1.0 + (l + 1) * 1.0009765625;
        if (h[64 + l] != shr) { ++bad; printf("shr lane %d: %g expected %g\n", l, h[64 + l], shr); }
        if (h[128 + l] != shl) { ++bad; printf("shl lane %d: %g expected %g\n", l, h[128 + l], shl); }
    }
    printf("lane ops: %d mismatches\n", bad);
    return bad != 0;
}
