// Per-candidate kernels of the acquisition path and the C entry points that chain them:
//
//   gram -> GEMM (A = Ks L^-T) -> k_stage2 (mu, Sigma, SDP, Z = M11 A) -> GEMM (W = Z L^-1)
//        -> k_grad (closed-form d<Omega,M>/dX)
//
// Replaces OEI.acquisition / OEI.omega / predict_f of the reference (oei.py:19-39, 79-103) for B
// candidate batches at once.  The gradient is the closed form of what the reference gets from
// tf.gradients with M held constant (oei.py:66-77):
//   d value/d x_a = J_a^T (p_a alpha - 2 W[:,a]) + 2 sum_{b != a} M11_ab dk(x_a,x_b)/dx_a,
//   J_a = dk(X, x_a)/dx_a,  W = K^-1 k(X,X*) M11,  p = d value/d mu.
#include "oei_sdp.cuh"

#define CH 64            // training-point chunk staged in LDS
#define ITEMS_MAX 10     // per-thread accumulators in the chunked reductions
#ifndef OEI_PF
#define OEI_PF 4
#endif
#ifndef OEI_PFZ
#define OEI_PFZ 4
#endif

// Posterior moments of candidate b into LDS: w.S = Sigma + noise I (full, symmetric), mu[].
// xq [q][d] must already be in LDS.  Ac is a [q][CH+1] staging buffer, vc a [CH] one.
template <int T>
__device__ __forceinline__ void posterior_moments(const SdpWork w, const KernParams& kp, double noise, int n,
                                  int npad, size_t astride, const double* __restrict__ Arows,
                                  const double* __restrict__ v, const lds_f64* xq, lds_f64* mu,
                                  lds_f64* Ac, lds_f64* vc) {
    const int tid = threadIdx.x;
    const int q = w.q, LD = w.LD, d = kp.d;
    const int npair = q * (q + 1) / 2, nitems = npair + q;
    double acc[ITEMS_MAX];
    int ii[ITEMS_MAX], jj[ITEMS_MAX];
#pragma unroll
    for (int u = 0; u < ITEMS_MAX; ++u) {
        acc[u] = 0.0;
        const int e = tid + u * T;
        int i = 0, j = -1;
        if (e < npair) {
            i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while ((i + 1) * (i + 2) / 2 <= e) ++i;
            while (i * (i + 1) / 2 > e) --i;
            j = e - i * (i + 1) / 2;
        } else if (e < nitems) {
            i = e - npair; j = -1;
        }
        ii[u] = i; jj[u] = j;
    }
    const int nchunks = (n + CH - 1) / CH;
    // The A rows of the next chunk travel from HBM into registers while the current chunk is
    // being reduced: NE elements per thread cover q * CH (q <= 32 with 64 threads, <= 64 with 256).
    constexpr int NE = (T == 64) ? 32 : 16;
    double nxt[NE];
    double vnxt = 0.0;
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            if (tid + u * T < q * CH) {                  // the last group is partial when q * CH % T != 0
                const int e = tid + u * T, a = e / CH, k = e - a * CH;
                nxt[u] = Arows[(size_t)a * astride + c * CH + k];
            }
        }
        if (tid < CH) vnxt = v[c * CH + tid];
    };
    fetch(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            if (tid + u * T < q * CH) {
                const int e = tid + u * T, a = e / CH, k = e - a * CH;
                Ac[a * (CH + 1) + k] = nxt[u];
            }
        }
        if (tid < CH) vc[tid] = vnxt;
        __syncthreads();
        if (c + 1 < nchunks) fetch(c + 1);
#pragma unroll
        for (int u = 0; u < ITEMS_MAX; ++u) {
            if (u * T >= nitems) break;                  // uniform; idle lanes redo item 0
            const lds_f64* ra = Ac + ii[u] * (CH + 1);
            const lds_f64* rb = (jj[u] >= 0) ? Ac + jj[u] * (CH + 1) : vc;
            double s = 0.0;
#pragma unroll 4
            for (int k = 0; k < CH; ++k) s += ra[k] * rb[k];
            acc[u] += s;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ITEMS_MAX; ++u) {
        const int e = tid + u * T;
        if (e >= nitems) break;
        const int i = ii[u], j = jj[u];
        if (j < 0) { mu[i] = acc[u]; continue; }
        double t = 0.0;
        for (int l = 0; l < d; ++l) {
            const double df = xq[i * d + l] - xq[j * d + l];
            t += df * df * kp.inv_ls2[l];
        }
        double k0, k1, k2;
        kappa_eval<0>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        const double val = kp.variance * k0 - acc[u] + (i == j ? noise : 0.0);
        w.S[i * LD + j] = val;
        w.S[j * LD + i] = val;
    }
    __syncthreads();
}

typedef double double4_t __attribute__((ext_vector_type(4)));

// The same moments for the one-wave kernels (q <= Q <= 32) on the matrix pipe: with E = [A rows;
// v] ((q+1) x npad), E E^T holds A A^T and A v.  v_mfma_f64_16x16x4 takes, per lane, one entry
// of each operand: lane l supplies row (l & 15) of a 16-row tile at one of four k positions
// (l >> 4).  A sum over k may visit k in any order, so lane l loads four consecutive doubles of
// its row per 16 columns (16 rows x 128 B per load instruction, every fetched line used whole)
// and feeds them to four successive MFMAs.  No LDS staging; the co-resident wave's vector
// arithmetic runs while the matrix pipe works.
template <int Q>
__device__ __forceinline__ void posterior_moments_mfma(const SdpWork w, const KernParams& kp, double noise,
                                                       int npad, size_t astride,
                                                       const double* __restrict__ Arows,
                                                       const double* __restrict__ v,
                                                       const lds_f64* xq, lds_f64* mu) {
    constexpr int NR = (Q + 1 + 15) / 16;              // row tiles of E
    constexpr int NTILE = NR * (NR + 1) / 2;           // lower block triangle of E E^T
    const int lane = threadIdx.x;
    const int q = w.q, LD = w.LD, d = kp.d;
    const int li = lane & 15, kg = lane >> 4;
    const double* src[NR];
    bool valid[NR];
#pragma unroll
    for (int I = 0; I < NR; ++I) {
        const int r = 16 * I + li;
        valid[I] = r <= q;
        // rows past q read v as well: their products only reach entries of E E^T that nothing reads
        src[I] = (r < q ? Arows + (size_t)r * astride : v) + 4 * kg;
    }
    double4_t acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // A ring of PF column groups in flight: the rows come from HBM / L2 with a latency of a few
    // microseconds under load, against well under a microsecond of matrix work per group, so one
    // group ahead (as in the first version) left the wave waiting for most of this phase.
    // The loads are unconditional (groups past the end re-read the last one): with a branch around
    // them the compiler can no longer count the loads in flight and waits for all of them.
    constexpr int PF = OEI_PF;                    // npad is a multiple of 64, so ns is one of PF = 4
    double4_t ring[PF][NR];
    const int ns = npad / 16;
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int I = 0; I < NR; ++I) {
            ring[u][I] = *reinterpret_cast<const double4_t*>(src[I] + 16 * u);
        }
    for (int s0 = 0; s0 < ns; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            double4_t cur[NR];
#pragma unroll
            for (int I = 0; I < NR; ++I) cur[I] = ring[u][I];
            const int sn = min(s0 + u + PF, ns - 1);
#pragma unroll
            for (int I = 0; I < NR; ++I)
                ring[u][I] = *reinterpret_cast<const double4_t*>(src[I] + 16 * sn);
#pragma unroll
            for (int v4 = 0; v4 < 4; ++v4) {
                int t = 0;
#pragma unroll
                for (int I = 0; I < NR; ++I)
#pragma unroll
                    for (int J = 0; J <= I; ++J, ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][v4], cur[J][v4], acc[t], 0, 0, 0);
            }
        }
    }
#ifdef OEI_SDP_TIMING
    const long long tm_loop = clock64();
#endif
    // accumulator g of lane l is entry (row (l >> 4) + 4 g, column l & 15) of its tile
    {
        int t = 0;
#pragma unroll
        for (int I = 0; I < NR; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J, ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int i = 16 * I + kg + 4 * g, j = 16 * J + li;
                    if (i < q && j <= i) w.S[i * LD + j] = acc[t][g];
                    else if (i == q && j < q) mu[j] = acc[t][g];
                }
    }
    __syncthreads();
    for (int e = lane; e < q * (q + 1) / 2; e += 64) {      // lower triangle only: a kernel evaluation per entry
        int i, j;
        sdp_tri_index(e, i, j);
        double t = 0.0;
        for (int l = 0; l < d; ++l) {
            const double df = xq[i * d + l] - xq[j * d + l];
            t += df * df * kp.inv_ls2[l];
        }
        double k0, k1, k2;
        kappa_eval<0>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        const double val = kp.variance * k0 - w.S[i * LD + j] + (i == j ? noise : 0.0);
        w.S[i * LD + j] = val;
        w.S[j * LD + i] = val;
    }
    __syncthreads();
#ifdef OEI_SDP_TIMING
    if (lane == 0) mu[q + 1] = (double)(clock64() - tm_loop);     // cycles after the row loop (diagnostic)
#endif
}

// The same for the 256-thread kernels (32 < q <= 64, NR = row tiles of E: 3, 4 or 5): the waves
// share the columns (wave w takes the 16-column groups w, w + 4, ...: no row is loaded twice and
// the matrix work is even), every wave keeps a partial E E^T in its accumulators, and the four
// partials are added through w.S / mu one wave after the other (a fixed order, so the sums carry
// the same bits in every run).  It replaced a loop over LDS-staged chunks of the A rows on the
// vector unit (2.6 M multiply-adds per candidate at q = 50, LDS-bound: a fifth of the kernel).
template <int T, int NR>
__device__ __forceinline__ void posterior_moments_mfma_wide(const SdpWork w, const KernParams& kp, double noise,
                                                            int npad, size_t astride,
                                                            const double* __restrict__ Arows,
                                                            const double* __restrict__ v,
                                                            const lds_f64* xq, lds_f64* mu) {
    constexpr int NTILE = NR * (NR + 1) / 2, NW = T / 64, PF = 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = w.q, LD = w.LD, d = kp.d;
    const int li = lane & 15, kg = lane >> 4;
    const double* src[NR];
#pragma unroll
    for (int I = 0; I < NR; ++I) {
        const int r = 16 * I + li;
        // rows past q read v as well: their products only reach entries of E E^T that nothing reads
        src[I] = (r < q ? Arows + (size_t)r * astride : v) + 4 * kg;
    }
    double4_t acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int ns = npad / 16, nsw = ns / NW;          // npad is a multiple of 64: every wave has nsw groups
    double4_t ring[PF][NR];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int I = 0; I < NR; ++I)
            ring[u][I] = *reinterpret_cast<const double4_t*>(src[I] + 16 * (wave + NW * min(u, nsw - 1)));
    for (int s0 = 0; s0 < nsw; s0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            double4_t cur[NR];
#pragma unroll
            for (int I = 0; I < NR; ++I) cur[I] = ring[u][I];
            const int sn = min(s0 + u + PF, nsw - 1);
#pragma unroll
            for (int I = 0; I < NR; ++I)
                ring[u][I] = *reinterpret_cast<const double4_t*>(src[I] + 16 * (wave + NW * sn));
            if (s0 + u < nsw) {
#pragma unroll
                for (int v4 = 0; v4 < 4; ++v4) {
                    int t = 0;
#pragma unroll
                    for (int I = 0; I < NR; ++I)
#pragma unroll
                        for (int J = 0; J <= I; ++J, ++t)
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[I][v4], cur[J][v4], acc[t], 0, 0, 0);
                }
            }
        }
    }
    // accumulator g of lane l is entry (row (l >> 4) + 4 g, column l & 15) of its tile
    for (int r = 0; r < NW; ++r) {
        if (wave == r) {
            int t = 0;
#pragma unroll
            for (int I = 0; I < NR; ++I)
#pragma unroll
                for (int J = 0; J <= I; ++J, ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int i = 16 * I + kg + 4 * g, j = 16 * J + li;
                        if (i < q && j <= i) w.S[i * LD + j] = (r ? w.S[i * LD + j] : 0.0) + acc[t][g];
                        else if (i == q && j < q) mu[j] = (r ? mu[j] : 0.0) + acc[t][g];
                    }
        }
        __syncthreads();
    }
    for (int e = tid; e < q * (q + 1) / 2; e += T) {        // lower triangle only: a kernel evaluation per entry
        int i, j;
        sdp_tri_index(e, i, j);
        double t = 0.0;
        for (int l = 0; l < d; ++l) {
            const double df = xq[i * d + l] - xq[j * d + l];
            t += df * df * kp.inv_ls2[l];
        }
        double k0, k1, k2;
        kappa_eval<0>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        const double val = kp.variance * k0 - w.S[i * LD + j] + (i == j ? noise : 0.0);
        w.S[i * LD + j] = val;
        w.S[j * LD + i] = val;
    }
    __syncthreads();
}
template <int T>
__device__ __forceinline__ void posterior_moments_wide(const SdpWork w, const KernParams& kp, double noise,
                                                       int npad, size_t astride, const double* __restrict__ Arows,
                                                       const double* __restrict__ v, const lds_f64* xq, lds_f64* mu) {
    if (w.q < 48) posterior_moments_mfma_wide<T, 3>(w, kp, noise, npad, astride, Arows, v, xq, mu);
    else if (w.q < 64) posterior_moments_mfma_wide<T, 4>(w, kp, noise, npad, astride, Arows, v, xq, mu);
    else posterior_moments_mfma_wide<T, 5>(w, kp, noise, npad, astride, Arows, v, xq, mu);
}

// Z[a, k] = -sum_b N[a][b] A[b][k] for the one-wave kernels, on the matrix pipe: N (w.H) is the
// left operand, held in registers for the whole pass; each 16-column tile of the candidate's A
// rows is loaded once (128 B per row and load instruction) and its Z tile stored the same way.
template <int Q>
__device__ __forceinline__ void zmat_mfma(const SdpWork w, int npad, size_t astride,
                                          const double* __restrict__ Arows, double* __restrict__ Zrows) {
    constexpr int NRZ = (Q + 15) / 16, NKZ = (Q + 3) / 4;
    const int lane = threadIdx.x;
    const int q = w.q, LD = w.LD;
    const int li = lane & 15, kg = lane >> 4;
    double nop[NRZ][NKZ];
#pragma unroll
    for (int I = 0; I < NRZ; ++I)
#pragma unroll
        for (int s = 0; s < NKZ; ++s) {
            const int i = 16 * I + li, b = 4 * s + kg;
            const bool in = i < q && b < q;
            const double h = w.H[(in ? i : 0) * LD + (in ? b : 0)];
            nop[I][s] = in ? h : 0.0;
        }
    const double* brow[NKZ];
    bool bval[NKZ];
#pragma unroll
    for (int s = 0; s < NKZ; ++s) {
        const int b = 4 * s + kg;
        bval[s] = b < q;
        brow[s] = Arows + (size_t)(bval[s] ? b : 0) * astride + li;
    }
    // PFZ column groups of the A rows in flight (see posterior_moments_mfma)
    constexpr int PFZ = OEI_PFZ;                  // npad / 16 is a multiple of PFZ = 4
    double ring[PFZ][NKZ];
    const int ng = npad / 16;
#pragma unroll
    for (int u = 0; u < PFZ; ++u)
#pragma unroll
        for (int s = 0; s < NKZ; ++s) ring[u][s] = brow[s][16 * u];
    for (int g0 = 0; g0 < ng; g0 += PFZ) {
#pragma unroll
        for (int u = 0; u < PFZ; ++u) {
            const int c0 = 16 * (g0 + u);
            double bop[NKZ];
#pragma unroll
            for (int s = 0; s < NKZ; ++s) bop[s] = bval[s] ? ring[u][s] : 0.0;
            const int cn = 16 * min(g0 + u + PFZ, ng - 1);
#pragma unroll
            for (int s = 0; s < NKZ; ++s) ring[u][s] = brow[s][cn];
            double4_t acc[NRZ];
#pragma unroll
            for (int I = 0; I < NRZ; ++I) acc[I] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < NKZ; ++s)
#pragma unroll
                for (int I = 0; I < NRZ; ++I)
                    acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(nop[I][s], bop[s], acc[I], 0, 0, 0);
#pragma unroll
            for (int I = 0; I < NRZ; ++I)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = 16 * I + kg + 4 * g;
                    if (row < q) Zrows[(size_t)row * npad + c0 + li] = -acc[I][g];
                }
        }
    }
}

// LDS of the per-candidate kernels: the solver region (with the A-row staging buffers Ac [q][CH+1]
// and vc [CH] overlaid on everything behind w.H, which is idle while rows are staged), then
// mu [q+2] and xq [q*d].  `full`: six-matrix layout (sensitivity kernels), else three.
__host__ __device__ inline size_t acq_stage_doubles(int q) { return (size_t)q * (CH + 1) + CH; }
// `staged`: the 256-thread kernels stage A rows; the one-wave kernels read them through the MFMAs.
// `mm`: the instance keeps Lc^-T in a fourth matrix (sdp_mm<Q>()).
__host__ __device__ inline size_t acq_lds_doubles(int q, int d, bool staged, bool mm = false) {
    return sdp_region_doubles(q, staged ? acq_stage_doubles(q) : 0, false, mm) + (size_t)(q + 2) + (size_t)q * d;
}

// ---- omega (oei.py:79-96) ------------------------------------------------------------------------
template <int T, int Q>
__global__ __launch_bounds__(T) void k_omega(KernParams kp, double noise, int n, int npad, int q,
                                             const double* __restrict__ Xq,
                                             const double* __restrict__ A,
                                             const double* __restrict__ v,
                                             double* __restrict__ omega) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    SdpWork w;
    sdp_carve(w, sm, q, false);
    lds_f64* mu = (lds_f64*)sm + sdp_region_doubles(q, Q == 0 ? acq_stage_doubles(q) : 0, false);
    lds_f64* xq = mu + q + 2;
    lds_f64* Ac = w.S;
    lds_f64* vc = Ac + q * (CH + 1);
    const int tid = threadIdx.x, b = blockIdx.x, k_ = q + 1, LD = w.LD;
    for (int e = tid; e < q * kp.d; e += T) xq[e] = Xq[(size_t)b * q * kp.d + e];
    __syncthreads();
    if constexpr (Q > 0) posterior_moments_mfma<Q>(w, kp, noise, npad, (size_t)npad, A + (size_t)b * q * npad, v, xq, mu);
    else posterior_moments_wide<T>(w, kp, noise, npad, (size_t)npad, A + (size_t)b * q * npad, v, xq, mu);
    double* om = omega + (size_t)b * k_ * k_;
    for (int e = tid; e < k_ * k_; e += T) {
        const int i = e / k_, j = e - i * k_;
        double val;
        if (i < q && j < q) val = w.S[i * LD + j] + mu[i] * mu[j];
        else if (i < q) val = mu[i];
        else if (j < q) val = mu[j];
        else val = 1.0;
        om[e] = val;
    }
}

// The same product for the 256-thread kernels (q <= 64) on the matrix pipe: a wave takes 16 columns
// of the A rows at a time (lane (li, kg) loads A[4 s + kg][16 c + li], 128 contiguous bytes per row
// and instruction, the next column group's loads issued before this one's matrix instructions),
// the left operand -N comes from LDS, and the 16-column tile of Z is stored the way it was loaded.
// KS = number of k steps (4 KS >= q), a template parameter for straight-line code.
template <int T, int KS>
__device__ __forceinline__ void zmat_wide_mfma(const SdpWork w, int npad, size_t a_row_stride,
                                               const double* __restrict__ Arows, double* __restrict__ Zrows) {
    constexpr int NRT = (KS + 3) / 4, NW = T / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int q = w.q, LD = w.LD, ng = npad / 16;
    const double* brow[KS];
    bool bin[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int b = 4 * s + kg;
        bin[s] = b < q;
        brow[s] = Arows + (size_t)(bin[s] ? b : 0) * a_row_stride + li;
    }
    double cur[KS], nxt[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) cur[s] = brow[s][16 * min(wave, ng - 1)];
    for (int c = wave; c < ng; c += NW) {
        const int cn = 16 * min(c + NW, ng - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) nxt[s] = brow[s][cn];
        double4_t acc[NRT];
#pragma unroll
        for (int I = 0; I < NRT; ++I) acc[I] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int b = 4 * s + kg;
            const double bv = bin[s] ? cur[s] : 0.0;
#pragma unroll
            for (int I = 0; I < NRT; ++I) {
                const int a = 16 * I + li;
                const bool in = a < q && bin[s];
                const double h = w.H[(in ? a : 0) * LD + (in ? b : 0)];
                acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(in ? -h : 0.0, bv, acc[I], 0, 0, 0);
            }
        }
#pragma unroll
        for (int I = 0; I < NRT; ++I)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int a = 16 * I + kg + 4 * g;
                if (a < q) Zrows[(size_t)a * npad + 16 * c + li] = acc[I][g];
            }
#pragma unroll
        for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
    }
}

// ---- stage 2: moments -> SDP -> value, p, M11, Z = M11 A ------------------------------------------
template <int T, int Q>
__global__ __launch_bounds__(T, SDP_MIN_WAVES(Q)) void k_stage2(KernParams kp, double noise, double fmin, int n,
                                              int npad, int q, const double* __restrict__ Xq,
                                              const double* __restrict__ A,
                                              size_t a_cand_stride, size_t a_row_stride,
                                              const double* __restrict__ v,
                                              double* __restrict__ value, int* __restrict__ status,
                                              double* __restrict__ p_out,
                                              double* __restrict__ M11_out,
                                              double* __restrict__ Z,
                                              const double* start_in,
                                              double* start_out,
                                              int* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    SdpWork w;
    sdp_carve(w, sm, q, false, sdp_mm<Q>());
    lds_f64* mu = (lds_f64*)sm + sdp_region_doubles(q, Q == 0 ? acq_stage_doubles(q) : 0, false, sdp_mm<Q>());
    lds_f64* xq = mu + q + 2;
    lds_f64* Ac = w.S;
    lds_f64* vc = Ac + q * (CH + 1);
    const int tid = threadIdx.x, b = blockIdx.x, LD = w.LD;
    for (int e = tid; e < q * kp.d; e += T) xq[e] = Xq[(size_t)b * q * kp.d + e];
    __syncthreads();
    const double* Arows = A + (size_t)b * a_cand_stride;
#ifdef OEI_SDP_TIMING
    const long long tk0 = clock64();
#endif
    if constexpr (Q > 0) posterior_moments_mfma<Q>(w, kp, noise, npad, a_row_stride, Arows, v, xq, mu);
    else posterior_moments_wide<T>(w, kp, noise, npad, a_row_stride, Arows, v, xq, mu);
#ifdef OEI_SDP_TIMING
    const long long tk1 = clock64();
#endif
    if (tid < q) w.dl[tid] = mu[tid] - fmin;
    __syncthreads();
    double g = 0.0;
    int it = 0;
    const int st = sdp_solve<T, Q>(w, &g, &it, start_in ? start_in + (size_t)b * q : nullptr,
                                   start_out ? start_out + (size_t)b * q : nullptr);
    if (iters_out && tid == 0) iters_out[b] = it;
    double* Zrows = Z + (size_t)b * q * npad;
    if (st == 2) {
        if (tid == 0) { value[b] = NAN; status[b] = 2; }
        for (int e = tid; e < q; e += T) p_out[(size_t)b * q + e] = NAN;
        for (int e = tid; e < q * q; e += T) M11_out[(size_t)b * q * q + e] = NAN;
        if (Z) for (int e = tid; e < q * npad; e += T) Zrows[e] = NAN;
        return;
    }
#ifdef OEI_SDP_TIMING
    const long long tk2 = clock64();
#endif
    sdp_primal<T>(w, mu);       // N = B^-1 in w.H
#ifdef OEI_SDP_TIMING
    const long long tk3 = clock64();
#endif
    if (tid == 0) { value[b] = -g; status[b] = st; }
    for (int e = tid; e < q; e += T) p_out[(size_t)b * q + e] = w.p[e];
    for (int e = tid; e < q * q; e += T) {
        const int i = e / q, j = e - i * q;
        M11_out[(size_t)b * q * q + e] = -w.H[i * LD + j];
    }
    // Z[a, k] = sum_b M11[a, b] A[b, k] with M11 = -N (w.H); without a Z buffer the second GEMM applies M11 itself
    if (!Z) {
    } else if constexpr (Q > 0) {
        zmat_mfma<Q>(w, npad, a_row_stride, Arows, Zrows);
    } else {
        __syncthreads();                               // N (w.H) is complete
        if (q <= 40) zmat_wide_mfma<T, 10>(w, npad, a_row_stride, Arows, Zrows);
        else if (q <= 52) zmat_wide_mfma<T, 13>(w, npad, a_row_stride, Arows, Zrows);
        else zmat_wide_mfma<T, 16>(w, npad, a_row_stride, Arows, Zrows);
    }
#ifdef OEI_SDP_TIMING
    // diagnostic build: cycles of [moments, solve, primal, outputs + Z] and the solver's own
    // phase counters in place of the start state
    if (start_out) {
        __syncthreads();
        const long long tk4 = clock64();
        double* so = start_out + (size_t)b * q;
        if (tid == 0 && q >= 12) {
            so[0] = (double)(tk1 - tk0); so[1] = (double)(tk2 - tk1); so[2] = (double)(tk3 - tk2); so[3] = (double)(tk4 - tk3);
            for (int i = 0; i < 8; ++i) so[4 + i] = w.scal[i];
            so[12] = mu[q + 1];
        }
    }
#endif
}

// ---- gradient -------------------------------------------------------------------------------------
// For each chunk of CH training points: ck[a][k] = (p_a alpha_k - 2 W[a,k]) * K1[a,k] with
// K1 = 2 var kappa'(t_ak) kept from the Gram kernel, then G[a][l] += sum_k ck[a][k] X[k][l] and
// G[a][d] += sum_k ck[a][k]; finally grad[a][l] = (x_al G[a][d] - G[a][l]) / ls_l^2 + self terms.
// Thread t owns training point k = t mod CH of the chunk and walks the candidate's points
// a = t / CH, t / CH + T / CH, ...; W and K1 rows are read coalesced along k.  Raises the `project`
// flag for mutual duplicates (the Gram kernel raises it for duplicates of training points).
template <int T>
__global__ __launch_bounds__(T) void k_grad(KernParams kp, int n, int npad, int q,
                                            const double* __restrict__ Xtrain,
                                            const double* __restrict__ alpha,
                                            const double* __restrict__ Xq,
                                            const double* __restrict__ Wm,
                                            const double* __restrict__ K1m,
                                            const double* __restrict__ p_in,
                                            const double* __restrict__ M11_in,
                                            double* __restrict__ grad, int* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    const int d = kp.d, dp = d + 1, dl = (d + 1) | 1;
    lds_f64* xq = (lds_f64*)sm;               // [q][d]
    lds_f64* M11 = xq + q * d;                // [q][q]
    lds_f64* pv = M11 + q * q;                // [q]
    lds_f64* ck = pv + q;                     // [q][CH+1]
    lds_f64* Xc = ck + q * (CH + 1);          // [CH][dl]   (column d holds 1)
    lds_f64* Gs = Xc + CH * dl;               // [q][q]     2 var kappa'(t_ab)
    lds_f64* Gl = Gs + q * q;                 // [q][d+1]   accumulators, one owner thread each
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int e = tid; e < q * d; e += T) xq[e] = Xq[(size_t)b * q * d + e];
    for (int e = tid; e < q * q; e += T) M11[e] = M11_in[(size_t)b * q * q + e];
    for (int e = tid; e < q; e += T) pv[e] = p_in[(size_t)b * q + e];
    __syncthreads();
    const int nitems = q * dp;
    for (int e = tid; e < nitems; e += T) Gl[e] = 0.0;
    int flag = 0;
    const double* Wrows = Wm + (size_t)b * q * npad;
    const double* Krows = K1m + (size_t)b * q * npad;
    const int nchunks = (n + CH - 1) / CH;
    const int k = tid & (CH - 1), ag = tid / CH;
    constexpr int NA = T / CH;
    for (int c = 0; c < nchunks; ++c) {
        const int kg = c * CH + k;                   // < npad: inside the padded rows of W and K1
        const double al = (kg < n) ? alpha[kg] : 0.0;
        __syncthreads();                     // the previous chunk's contraction is done
        if (ag == 0) {
            for (int l = 0; l < d; ++l) Xc[k * dl + l] = (kg < n) ? Xtrain[(size_t)kg * d + l] : 0.0;
            Xc[k * dl + d] = 1.0;
        }
#pragma unroll 4
        for (int a = ag; a < q; a += NA) {
            const double wv = Wrows[(size_t)a * npad + kg], kv = Krows[(size_t)a * npad + kg];
            ck[a * (CH + 1) + k] = (kg < n) ? (pv[a] * al - 2.0 * wv) * kv : 0.0;
        }
        __syncthreads();
        for (int e = tid; e < nitems; e += T) {
            const int a = e / dp, l = e - a * dp;
            const lds_f64* ca = ck + a * (CH + 1);
            const lds_f64* xl = Xc + l;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll 2
            for (int k0 = 0; k0 < CH; k0 += 8) {         // 16 LDS reads in flight, then 8 multiply-adds
                double av[8], bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { av[j] = ca[k0 + j]; bv[j] = xl[(k0 + j) * dl]; }
#pragma unroll
                for (int j = 0; j < 8; j += 2) { s0 += av[j] * bv[j]; s1 += av[j + 1] * bv[j + 1]; }
                SDP_READS_FIRST(16, 8);
            }
            Gl[e] += s0 + s1;                            // this thread's own accumulator
        }
    }
    // self-covariance derivative factors and mutual-duplicate test (oei.py:190)
    for (int e = tid; e < q * q; e += T) {
        const int a = e / q, bb = e - a * q;
        double t = 0.0;
        bool same = true;
        for (int l = 0; l < d; ++l) {
            const double df = xq[a * d + l] - xq[bb * d + l];
            t += df * df * kp.inv_ls2[l];
            same = same && (rint(xq[a * d + l] / kp.ls[l] * 100.0) == rint(xq[bb * d + l] / kp.ls[l] * 100.0));
        }
        double k0, k1, k2;
        kappa_eval<1>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        Gs[e] = (a == bb) ? 0.0 : 2.0 * kp.variance * k1;
        if (a != bb && same) flag = 1;
    }
    __syncthreads();
    const lds_f64* Gacc = Gl;
    for (int e = tid; e < q * d; e += T) {
        const int a = e / d, l = e - a * d;
        double gval = (xq[a * d + l] * Gacc[a * dp + d] - Gacc[a * dp + l]) * kp.inv_ls2[l];
        double s = 0.0;
        for (int bb = 0; bb < q; ++bb)
            s += M11[a * q + bb] * Gs[a * q + bb] * (xq[a * d + l] - xq[bb * d + l]);
        gval += 2.0 * s * kp.inv_ls2[l];
        grad[(size_t)b * q * d + e] = gval;
    }
    if (flags) {
        flag = __syncthreads_or(flag);
        if (tid == 0 && flag) flags[b] = 1;          // zeroed by the caller; the Gram kernel may have set it
    }
}

// ---- gradient, finishing step after the fused GEMM + contraction -----------------------------------
// Gpart [nsplit][nrows][d+1] holds the n-length sums of k_grad's chunk loop; what is left is
// per candidate: grad[a][l] = (x_al G[a][d] - G[a][l]) / ls_l^2 + the self-covariance terms, and
// the mutual-duplicate flag (oei.py:190).
__global__ __launch_bounds__(64) void k_grad_fin(KernParams kp, int q, size_t nrows, int nsplit,
                                                 const double* __restrict__ Xq,
                                                 const double* __restrict__ Gpart,
                                                 const double* __restrict__ M11_in,
                                                 double* __restrict__ grad, int* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    constexpr int T = 64;
    const int d = kp.d, dp = d + 1;
    lds_f64* xq = (lds_f64*)sm;               // [q][d]
    lds_f64* M11 = xq + q * d;                // [q][q]
    lds_f64* Gs = M11 + q * q;                // [q][q]     2 var kappa'(t_ab)
    lds_f64* Gl = Gs + q * q;                 // [q][d+1]
    lds_f64* xr = Gl + q * dp;                // [q][d]     coordinates on the grid of the duplicate test
    const int tid = threadIdx.x, b = blockIdx.x;
    for (int e = tid; e < q * d; e += T) {
        const double x = Xq[(size_t)b * q * d + e];
        xq[e] = x;
        xr[e] = rint(x / kp.ls[e % d] * 100.0);          // once per coordinate, not twice per pair
    }
    for (int e = tid; e < q * q; e += T) M11[e] = M11_in[(size_t)b * q * q + e];
    for (int e = tid; e < q * dp; e += T) {
        double acc = 0.0;
        for (int sp = 0; sp < nsplit; ++sp) acc += Gpart[((size_t)sp * nrows + (size_t)b * q) * dp + e];
        Gl[e] = acc;
    }
    __syncthreads();
    int flag = 0;
    for (int e = tid; e < q * (q + 1) / 2; e += T) {         // pairs a >= bb: the factor is symmetric
        int a, bb;
        sdp_tri_index(e, a, bb);
        double t = 0.0;
        bool same = true;
        for (int l = 0; l < d; ++l) {
            const double df = xq[a * d + l] - xq[bb * d + l];
            t += df * df * kp.inv_ls2[l];
            same = same && (xr[a * d + l] == xr[bb * d + l]);
        }
        double k0, k1, k2;
        kappa_eval<1>(kp.kind, kp.floor_mode, t, k0, k1, k2);
        const double gv = (a == bb) ? 0.0 : 2.0 * kp.variance * k1;
        Gs[a * q + bb] = gv;
        Gs[bb * q + a] = gv;
        if (a != bb && same) flag = 1;
    }
    __syncthreads();
    for (int e = tid; e < q * d; e += T) {
        const int a = e / d, l = e - a * d;
        double gval = (xq[a * d + l] * Gl[a * dp + d] - Gl[a * dp + l]) * kp.inv_ls2[l];
        double s = 0.0;
        for (int bb = 0; bb < q; ++bb)
            s += M11[a * q + bb] * Gs[a * q + bb] * (xq[a * d + l] - xq[bb * d + l]);
        gval += 2.0 * s * kp.inv_ls2[l];
        grad[(size_t)b * q * d + e] = gval;
    }
    if (flags) {
        flag = __syncthreads_or(flag);
        if (tid == 0 && flag) flags[b] = 1;          // zeroed by the caller; the Gram kernel may have set it
    }
}

// ---- predict_f: mean_i = A_i . v, var_i = k(x_i,x_i) - |A_i|^2 (one wave per point) ---------------
__global__ void k_predict(KernParams kp, int npad, const double* __restrict__ A,
                          const double* __restrict__ v, double* __restrict__ mean,
                          double* __restrict__ var) {
    const size_t r = blockIdx.x;
    double m = 0.0, s = 0.0;
    for (int k = threadIdx.x; k < npad; k += 64) {
        const double a = A[r * npad + k];
        m += a * v[k];
        s += a * a;
    }
    m = wave_sum(m); s = wave_sum(s);
    if (threadIdx.x == 0) {
        double k0, k1, k2;
        kappa_eval<0>(kp.kind, kp.floor_mode, 0.0, k0, k1, k2);
        mean[r] = m;
        var[r] = kp.variance * k0 - s;
    }
}

// ---- arg-min with ties to the lowest index --------------------------------------------------------
__global__ __launch_bounds__(1024) void k_argmin(int B, const double* __restrict__ value,
                                                 long long base, double* __restrict__ best_value,
                                                 long long* __restrict__ best_index) {
    __shared__ double sv[1024];
    __shared__ int si[1024];
    double bv = INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < B; i += 1024) {
        const double x = value[i];
        if (x < bv) { bv = x; bi = i; }      // strided scan keeps the lowest index per thread
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const double ov = sv[threadIdx.x + off];
            const int oi = si[threadIdx.x + off];
            if (ov < sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = ov; si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *best_value = sv[0];
        *best_index = (si[0] == 0x7fffffff) ? -1 : base + si[0];
    }
}

// The same arg-min, leaving the whole (value, index, X) record of the winner in one buffer:
// record[0] = value, record[1] = the 64-bit index (bit pattern), record[2 ..] = X[best] (xlen doubles).
__global__ __launch_bounds__(1024) void k_best_record(int B, const double* __restrict__ value,
                                                      const double* __restrict__ X, int xlen,
                                                      long long base, double* __restrict__ record) {
    __shared__ double sv[1024];
    __shared__ int si[1024];
    double bv = INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < B; i += 1024) {
        const double x = value[i];
        if (x < bv) { bv = x; bi = i; }      // NaN never wins; strided scan keeps the lowest index per thread
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const double ov = sv[threadIdx.x + off];
            const int oi = si[threadIdx.x + off];
            if (ov < sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) {
                sv[threadIdx.x] = ov; si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    const int win = si[0];
    const bool none = win == 0x7fffffff;                  // every value NaN or +inf (or B == 0)
    if (threadIdx.x == 0) {
        record[0] = none ? NAN : sv[0];
        reinterpret_cast<long long*>(record)[1] = none ? -1 : base + win;
    }
    for (int e = threadIdx.x; e < xlen; e += 1024) record[2 + e] = none ? 0.0 : X[(size_t)win * xlen + e];
}

// Winner among `nrec` records of `reclen` doubles (value, index bits, payload): smallest value with
// NaN counted as +inf, ties to the smallest index.
__global__ __launch_bounds__(256) void k_select_record(int nrec, int reclen, const double* __restrict__ rec,
                                                       double* __restrict__ out) {
    __shared__ double sv[256];
    __shared__ long long si[256];
    __shared__ int sr[256];
    double bv = INFINITY;
    long long bi = LLONG_MAX;
    int br = -1;
    for (int r = threadIdx.x; r < nrec; r += 256) {
        const double x = rec[(size_t)r * reclen];
        const double v = (x != x) ? INFINITY : x;
        const long long idx = reinterpret_cast<const long long*>(rec + (size_t)r * reclen)[1];
        if (br < 0 || v < bv || (v == bv && idx < bi)) { bv = v; bi = idx; br = r; }
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi; sr[threadIdx.x] = br;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const double ov = sv[threadIdx.x + off];
            const long long oi = si[threadIdx.x + off];
            const int orr = sr[threadIdx.x + off];
            const bool mine = sr[threadIdx.x] >= 0;
            if (orr >= 0 && (!mine || ov < sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))) {
                sv[threadIdx.x] = ov; si[threadIdx.x] = oi; sr[threadIdx.x] = orr;
            }
        }
        __syncthreads();
    }
    const int win = sr[0];
    for (int e = threadIdx.x; e < reclen; e += 256) out[e] = win >= 0 ? rec[(size_t)win * reclen + e] : NAN;
}

// ------------------------------------------------------------------------------------------------
static inline size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

static int check_model_call(const oei_model* m, int B, int q, const void* X, const char* who) {
    if (!m) { oei_set_error("%s: null model", who); return OEI_ERR_ARG; }
    if (B < 0 || q < 1 || q > OEI_MAX_Q) { oei_set_error("%s: B=%d q=%d out of range (1 <= q <= %d)", who, B, q, OEI_MAX_Q); return OEI_ERR_ARG; }
    if (B > 0 && !X) { oei_set_error("%s: null X", who); return OEI_ERR_ARG; }
    return OEI_OK;
}

// gram + first GEMM for `rows` query points: leaves A = Ks L^-T in m->ws1 (Ks in m->ws0).
static int cross_factor(oei_model* m, const double* Xq, size_t rows, size_t aux_bytes, hipStream_t st) {
    const size_t rows_pad = round_up(rows, 64);
    int rc = oei_ws_reserve(m, rows_pad, aux_bytes);
    if (rc) return rc;
    rc = oei_launch_gram(m, Xq, rows, rows_pad, m->ws0, nullptr, 1, nullptr, st);
    if (rc) return rc;
    return oei_launch_gemm_nt(m->ws0, m->Linv, m->ws1, rows_pad, m->npad, 1, st);
}

template <typename K>
static int set_lds(K kernel, size_t lds) {
    if (lds > 160 * 1024) { oei_set_error("LDS request %zu exceeds 160 KiB", lds); return OEI_ERR_ARG; }
    if (lds > 48 * 1024)
        OEI_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return OEI_OK;
}

// moments + SDP + Z for B candidates whose A rows sit at A + b*a_cand_stride + a*a_row_stride.
int oei_launch_stage2(const oei_model* m, int B, int q, const double* X, const double* A,
                      size_t a_cand_stride, size_t a_row_stride, double* value, int* status,
                      double* p_buf, double* M11_buf, double* Z, hipStream_t st,
                      const double* start_in, double* start_out, int* iters,
                      hipEvent_t ev_start, hipEvent_t ev_stop) {
    int rc;
    if (q * (q + 1) / 2 + q > ITEMS_MAX * (q <= 32 ? 64 : 256)) {
        oei_set_error("stage2: q=%d exceeds the per-thread accumulator budget", q);
        return OEI_ERR_ARG;
    }
#define LAUNCH_STAGE2(T_, Q_) do { \
        const size_t lds2 = acq_lds_doubles(q, m->d, (Q_) == 0, sdp_mm<Q_>()) * sizeof(double); \
        if ((rc = set_lds(k_stage2<T_, Q_>, lds2))) return rc; \
        if (ev_start) /* the events take the kernel's own start and end from its dispatch */ \
            hipExtLaunchKernelGGL((k_stage2<T_, Q_>), dim3(B), dim3(T_), (uint32_t)lds2, st, ev_start, ev_stop, 0, m->kp, m->noise, m->fmin, m->n, m->npad, q, X, A, a_cand_stride, a_row_stride, (const double*)m->v, value, status, p_buf, M11_buf, Z, start_in, start_out, iters); \
        else \
            hipLaunchKernelGGL((k_stage2<T_, Q_>), dim3(B), dim3(T_), lds2, st, m->kp, m->noise, m->fmin, m->n, m->npad, q, X, A, a_cand_stride, a_row_stride, m->v, value, status, p_buf, M11_buf, Z, start_in, start_out, iters); } while (0)
    SDP_DISPATCH_Q(q, LAUNCH_STAGE2);
#undef LAUNCH_STAGE2
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

extern "C" int oei_predict_f(oei_model* m, int npts, const double* X, double* mean, double* var,
                             void* stream) {
    if (!m || npts < 0 || (npts > 0 && (!X || !mean || !var))) { oei_set_error("predict_f: bad argument"); return OEI_ERR_ARG; }
    if (npts == 0) return OEI_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc = cross_factor(m, X, (size_t)npts, 0, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_predict, dim3(npts), dim3(64), 0, st, m->kp, m->npad, m->ws1, m->v, mean, var);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

extern "C" int oei_omega_batched(oei_model* m, int B, int q, const double* X, double* omega,
                                 void* stream) {
    int rc = check_model_call(m, B, q, X, "omega_batched");
    if (rc) return rc;
    if (B == 0) return OEI_OK;
    if (!omega) { oei_set_error("omega_batched: null output"); return OEI_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    rc = cross_factor(m, X, (size_t)B * q, 0, st);
    if (rc) return rc;
    if (q <= 32) {
        const size_t lds = acq_lds_doubles(q, m->d, false) * sizeof(double);
        if ((rc = set_lds(k_omega<64, 32>, lds))) return rc;
        hipLaunchKernelGGL((k_omega<64, 32>), dim3(B), dim3(64), lds, st, m->kp, m->noise, m->n, m->npad, q, X, m->ws1, m->v, omega);
    } else {
        const size_t lds = acq_lds_doubles(q, m->d, true) * sizeof(double);
        if ((rc = set_lds(k_omega<256, 0>, lds))) return rc;
        hipLaunchKernelGGL((k_omega<256, 0>), dim3(B), dim3(256), lds, st, m->kp, m->noise, m->n, m->npad, q, X, m->ws1, m->v, omega);
    }
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

// Per-candidate state behind the three row buffers: p [B q], M11 [B q q], partial gradient sums
// [npad/64][B q][d+1].
static inline size_t value_grad_aux_doubles(const oei_model* m, int B, int q) {
    return (size_t)B * q + (size_t)B * q * q + (size_t)(m->npad / 64) * B * q * (m->d + 1);
}

// The five launches of the value+gradient path on caller-chosen scratch: ws0, ws1, ws2 hold
// rows_pad x npad doubles each (Ks then Z | A | K1), aux holds value_grad_aux_doubles().  Nothing
// in here allocates.
static int value_grad_run(oei_model* m, int B, int q, const double* X, double* value, double* grad,
                          int* status, int* flags, hipStream_t st, double* ws0, double* ws1,
                          double* ws2, double* aux, const double* start_in = nullptr,
                          double* start_out = nullptr, int* iters = nullptr) {
    int rc;
    const int d = m->d;
    const size_t rows = (size_t)B * q, rows_pad = round_up(rows, 64);
    const bool prof = m->profile != 0;
    if (prof) OEI_HIP_CHECK(hipEventRecord(m->ev[0], st));
    if (flags) OEI_HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)B * sizeof(int), st));
    rc = oei_launch_gram(m, X, rows, rows_pad, ws0, ws2, q, flags, st);   // ws0 = Ks, ws2 = K1
    if (rc) return rc;
    if (prof) OEI_HIP_CHECK(hipEventRecord(m->ev[1], st));
    rc = oei_launch_gemm_nt(ws0, m->Linv, ws1, rows_pad, m->npad, 1, st);   // ws1 = A
    if (rc) return rc;
    if (prof) OEI_HIP_CHECK(hipEventRecord(m->ev[2], st));
    double* p_buf = aux;
    double* M11_buf = aux + (size_t)B * q;
    const size_t ldsg = ((size_t)q * d + 2 * (size_t)q * q + q + (size_t)q * (CH + 1) + (size_t)CH * ((d + 1) | 1) + (size_t)q * (d + 1)) * sizeof(double);
    if (q * (q + 1) / 2 + q > ITEMS_MAX * (q <= 32 ? 64 : 256)) {
        oei_set_error("value_grad_batched: q=%d d=%d exceeds the per-thread accumulator budget", q, d);
        return OEI_ERR_ARG;
    }
    const bool zfree = m->zfree && d + 1 <= OEI_GG_MAX_DP;
    rc = oei_launch_stage2(m, B, q, X, ws1, (size_t)q * m->npad, (size_t)m->npad, value, status, p_buf, M11_buf, zfree ? nullptr : ws0, st, start_in, start_out, iters,
                           prof ? m->kev[0] : nullptr, prof ? m->kev[1] : nullptr);
    if (rc) return rc;
    if (rows_pad > rows && !zfree)   // rows of Z beyond the batch feed the GEMM: keep them finite
        OEI_HIP_CHECK(hipMemsetAsync(ws0 + rows * (size_t)m->npad, 0, (rows_pad - rows) * (size_t)m->npad * sizeof(double), st));
    if (prof) OEI_HIP_CHECK(hipEventRecord(m->ev[3], st));
    if (d + 1 <= OEI_GG_MAX_DP) {
        // W = Z L^-1 contracted with K1 and X inside the GEMM; a small per-candidate kernel finishes
        double* Gpart = M11_buf + (size_t)B * q * q;
        rc = zfree ? oei_launch_gemm_grad_zfree(m, ws1, ws2, p_buf, M11_buf, Gpart, rows, rows_pad, q, st)
                   : oei_launch_gemm_grad(m, ws0, ws2, p_buf, Gpart, rows, rows_pad, st);
        if (rc) return rc;
        if (prof) OEI_HIP_CHECK(hipEventRecord(m->ev[4], st));
        const size_t ldsf = ((size_t)2 * q * d + 2 * (size_t)q * q + (size_t)q * (d + 1)) * sizeof(double);
        if ((rc = set_lds(k_grad_fin, ldsf))) return rc;
        hipLaunchKernelGGL(k_grad_fin, dim3(B), dim3(64), ldsf, st, m->kp, q, rows, m->npad / 64, X, Gpart, M11_buf, grad, flags);
    } else {
        rc = oei_launch_gemm_nt(ws0, m->LinvT, ws1, rows_pad, m->npad, 0, st);   // ws1 = W = Z L^-1
        if (rc) return rc;
        if (prof) OEI_HIP_CHECK(hipEventRecord(m->ev[4], st));
        // four waves per candidate split its points four ways and overlap their memory latencies
        // (measured at q = 20: 0.29 ms with one wave, 0.19 ms with four, 0.26 ms with eight)
        if (q <= 4) {
            if ((rc = set_lds(k_grad<64>, ldsg))) return rc;
            hipLaunchKernelGGL(k_grad<64>, dim3(B), dim3(64), ldsg, st, m->kp, m->n, m->npad, q, m->X, m->alpha, X, ws1, ws2, p_buf, M11_buf, grad, flags);
        } else {
            if ((rc = set_lds(k_grad<256>, ldsg))) return rc;
            hipLaunchKernelGGL(k_grad<256>, dim3(B), dim3(256), ldsg, st, m->kp, m->n, m->npad, q, m->X, m->alpha, X, ws1, ws2, p_buf, M11_buf, grad, flags);
        }
    }
    OEI_HIP_CHECK(hipGetLastError());
    if (prof) { OEI_HIP_CHECK(hipEventRecord(m->ev[5], st)); m->ev_valid = 1; }
    return OEI_OK;
}

// The same work as value_grad_run in `nchunks` pieces that alternate between `nstreams` side
// streams: the tail of one piece's solver kernel (its waves finish at different times, and a
// compute unit that has run out of candidates idles) is filled by the GEMM and Gram kernels of
// the next piece.  Pieces hold a whole number of 64-row tiles, so each works in its own slice of
// the row buffers; its per-candidate state sits at the slice of `aux` that a batch of the
// preceding candidates would occupy.  The caller's stream waits for every piece.
static int value_grad_pipelined(oei_model* m, int B, int q, const double* X, double* value, double* grad,
                                int* status, int* flags, hipStream_t st, double* ws0, double* ws1,
                                double* ws2, double* aux, int nchunks, int nstreams,
                                const double* start_in = nullptr, double* start_out = nullptr,
                                int* iters = nullptr) {
    int rc;
    int g = q, h = 64;                                   // candidates per 64-row tile boundary: 64 / gcd(q, 64)
    while (h) { const int t = g % h; g = h; h = t; }
    const int unit = 64 / g;
    int per = (B + nchunks - 1) / nchunks;
    per = (per + unit - 1) / unit * unit;
    if (nstreams > 2) nstreams = 2;
    if (per >= B || nstreams < 1)
        return value_grad_run(m, B, q, X, value, grad, status, flags, st, ws0, ws1, ws2, aux, start_in, start_out, iters);
    for (int i = 0; i < nstreams; ++i) {
        if (!m->side[i]) OEI_HIP_CHECK(hipStreamCreateWithFlags(&m->side[i], hipStreamNonBlocking));
        if (!m->ev_join[i]) OEI_HIP_CHECK(hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming));
    }
    if (!m->ev_fork) OEI_HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    OEI_HIP_CHECK(hipEventRecord(m->ev_fork, st));      // nothing is on the side streams yet: plain returns up to here
    rc = OEI_OK;
    for (int i = 0; i < nstreams && rc == OEI_OK; ++i) {
        const hipError_t e = hipStreamWaitEvent(m->side[i], m->ev_fork, 0);
        if (e != hipSuccess) { oei_set_error("value_grad: fork to side stream %d failed: %s", i, hipGetErrorString(e)); rc = OEI_ERR_HIP; }
    }
    const int d = m->d;
    int c = 0;
    for (int b0 = 0; b0 < B && rc == OEI_OK; b0 += per, ++c) {
        const int bc = (B - b0 < per) ? B - b0 : per;
        const size_t row0 = (size_t)b0 * q * m->npad;
        rc = value_grad_run(m, bc, q, X + (size_t)b0 * q * d, value + b0, grad + (size_t)b0 * q * d, status + b0,
                            flags ? flags + b0 : nullptr, m->side[c % nstreams], ws0 + row0, ws1 + row0, ws2 + row0,
                            aux + value_grad_aux_doubles(m, b0, q),
                            start_in ? start_in + (size_t)b0 * q : nullptr,
                            start_out ? start_out + (size_t)b0 * q : nullptr, iters ? iters + b0 : nullptr);
        if (rc) break;                                   // the caller's stream still joins what was enqueued
    }
    // Always join: whatever was enqueued on the side streams must be ordered before the caller's
    // stream goes on, also after an error (the caller may release X, the outputs or the handle).
    // If the event route fails the side stream is drained on the host instead.  After an error
    // return the contents of the outputs are undefined.
    for (int i = 0; i < nstreams; ++i) {
        hipError_t e = hipEventRecord(m->ev_join[i], m->side[i]);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, m->ev_join[i], 0);
        if (e != hipSuccess) {
            hipStreamSynchronize(m->side[i]);
            if (rc == OEI_OK) {
                oei_set_error("value_grad: joining side stream %d failed: %s", i, hipGetErrorString(e));
                rc = OEI_ERR_HIP;
            }
        }
    }
    return rc;
}

// Batches of at least two full rounds of the solver kernel (eight one-wave candidates per compute
// unit and round) run as two halves on two side streams: measured 1.15 -> 1.11 ms at B = 4096 and
// 2.16 -> 2.09 ms at B = 8192 (q = 20, n = 310); smaller pieces leave compute units without a
// candidate and are slower (four pieces of 1024: 1.29 ms).  Results do not depend on the split.
// Profiled calls stay on the caller's stream (the stage events bracket whole-batch launches).
static int value_grad_auto(oei_model* m, int B, int q, const double* X, double* value, double* grad,
                           int* status, int* flags, hipStream_t st, const double* start_in = nullptr,
                           double* start_out = nullptr, int* iters = nullptr) {
    // m->pipeline: 0 off, 1 the rule above, n >= 2 (diagnostic) n pieces whatever the batch size
    if (m->pipeline && !m->profile && q <= 32 && m->n_cu > 0 && (m->pipeline > 1 || B >= 2 * 8 * m->n_cu))
        return value_grad_pipelined(m, B, q, X, value, grad, status, flags, st, m->ws0, m->ws1, m->ws2, m->aux,
                                    m->pipeline > 1 ? m->pipeline : 2, 2, start_in, start_out, iters);
    return value_grad_run(m, B, q, X, value, grad, status, flags, st, m->ws0, m->ws1, m->ws2, m->aux, start_in, start_out, iters);
}

extern "C" int oei_set_z_free(oei_model* m, int on) {
    if (!m) { oei_set_error("set_z_free: null model"); return OEI_ERR_ARG; }
    m->zfree = on ? 1 : 0;
    return OEI_OK;
}

extern "C" int oei_set_pipelining(oei_model* m, int on) {
    if (!m) { oei_set_error("set_pipelining: null model"); return OEI_ERR_ARG; }
    m->pipeline = on < 0 ? 0 : on;
    return OEI_OK;
}

// Scratch of the entry point that allocates nothing: one array per GPU, in the code object's own
// global memory.  In host code the array's name is only a stand-in: the device address comes from
// hipGetSymbolAddress, once per device with that device current, and is what kernels, memsets and
// copies get.  32 MiB: about 200 candidates at q = 20, n = 310.
static __device__ __align__(256) double scratch[OEI_NOALLOC_SCRATCH_DOUBLES];
#define OEI_MAX_DEVICES 64
static double* scratch_ptr[OEI_MAX_DEVICES];          // per-device address of `scratch`

// Looks up this device's copy of the scratch array (no allocation).  Called by oei_model_create
// for the model's device and, if a device was never set up, by the entry point itself.
int oei_noalloc_setup_device(int device) {
    if (device < 0 || device >= OEI_MAX_DEVICES) { oei_set_error("noalloc: device %d out of range", device); return OEI_ERR_ARG; }
    if (scratch_ptr[device]) return OEI_OK;
    int cur = -1;
    OEI_HIP_CHECK(hipGetDevice(&cur));
    if (cur != device) { oei_set_error("noalloc: device %d is not current (current: %d)", device, cur); return OEI_ERR_ARG; }
    void* ptr = nullptr;
    OEI_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) { oei_set_error("noalloc: no address for the scratch array on device %d", device); return OEI_ERR_HIP; }
    scratch_ptr[device] = (double*)ptr;
    return OEI_OK;
}

extern "C" int oei_value_grad_batched(oei_model* m, int B, int q, const double* X, double* value,
                                      double* grad, int* status, int* flags, void* stream) {
    int rc = check_model_call(m, B, q, X, "value_grad_batched");
    if (rc) return rc;
    if (B == 0) return OEI_OK;
    if (!value || !grad || !status) { oei_set_error("value_grad_batched: null output"); return OEI_ERR_ARG; }
    const size_t rows_pad = round_up((size_t)B * q, 64);
    if ((rc = oei_ws_reserve(m, rows_pad, value_grad_aux_doubles(m, B, q) * sizeof(double)))) return rc;
    if ((rc = oei_ws2_reserve(m, rows_pad))) return rc;
    OEI_SCRATCH_POISON(m, (hipStream_t)stream);
    return value_grad_auto(m, B, q, X, value, grad, status, flags, (hipStream_t)stream);
}

extern "C" int oei_value_grad_warm_batched(oei_model* m, int B, int q, const double* X, double* value,
                                           double* grad, int* status, int* flags,
                                           const double* start_in, double* start_out, int* iters, void* stream) {
    int rc = check_model_call(m, B, q, X, "value_grad_warm_batched");
    if (rc) return rc;
    if (B == 0) return OEI_OK;
    if (!value || !grad || !status) { oei_set_error("value_grad_warm_batched: null output"); return OEI_ERR_ARG; }
    const size_t rows_pad = round_up((size_t)B * q, 64);
    if ((rc = oei_ws_reserve(m, rows_pad, value_grad_aux_doubles(m, B, q) * sizeof(double)))) return rc;
    if ((rc = oei_ws2_reserve(m, rows_pad))) return rc;
    return value_grad_auto(m, B, q, X, value, grad, status, flags, (hipStream_t)stream, start_in, start_out, iters);
}

extern "C" int oei_value_grad_batched_noalloc(oei_model* m, int B, int q, const double* X, double* value,
                                              double* grad, int* status, int* flags, void* stream) {
    int rc = check_model_call(m, B, q, X, "value_grad_batched_noalloc");
    if (rc) return rc;
    if (B == 0) return OEI_OK;
    if (!value || !grad || !status) { oei_set_error("value_grad_batched_noalloc: null output"); return OEI_ERR_ARG; }
    const size_t rows_pad = round_up((size_t)B * q, 64);
    const size_t buf = rows_pad * (size_t)m->npad;                       // a multiple of 64 * 64 doubles
    const size_t need = 3 * buf + round_up(value_grad_aux_doubles(m, B, q), 32);
    if (need > (size_t)OEI_NOALLOC_SCRATCH_DOUBLES) {
        oei_set_error("value_grad_batched_noalloc: B=%d q=%d n=%d needs %zu doubles of scratch, the fixed "
                      "array holds %zu (use oei_value_grad_batched)", B, q, m->n, need, (size_t)OEI_NOALLOC_SCRATCH_DOUBLES);
        return OEI_ERR_ARG;
    }
    if (m->device < 0 || m->device >= OEI_MAX_DEVICES) { oei_set_error("noalloc: device %d out of range", m->device); return OEI_ERR_ARG; }
    int cur = -1;
    OEI_HIP_CHECK(hipGetDevice(&cur));
    if (cur != m->device) {      // the launches below go to the current device; the scratch address belongs to the model's
        oei_set_error("value_grad_batched_noalloc: the model lives on device %d, the current device is %d", m->device, cur);
        return OEI_ERR_ARG;
    }
    if (!scratch_ptr[m->device] && (rc = oei_noalloc_setup_device(m->device))) return rc;
    double* base = scratch_ptr[m->device];
    return value_grad_run(m, B, q, X, value, grad, status, flags, (hipStream_t)stream,
                          base, base + buf, base + 2 * buf, base + 3 * buf);
}

// ---- the duplicate tests of `project` alone (oei.py:181-192) ---------------------------------------
// flags[b] = 1 when a point of candidate b lies within 1e-2 lengthscales of a training input in
// every coordinate, or two of its points are equal after rounding X/l to two decimals: the same
// two tests, in the same arithmetic, that the Gram kernel and the finishing kernel of the
// value+gradient path apply.
__global__ __launch_bounds__(256) void k_project_flags(KernParams kp, int n, int q,
                                                       const double* __restrict__ Xtrain,
                                                       const double* __restrict__ Xq,
                                                       int* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    OEI_LDS_POISON(sm);
    const int d = kp.d, tid = threadIdx.x, b = blockIdx.x;
    double* xq = sm;                              // [q][d]
    for (int e = tid; e < q * d; e += 256) xq[e] = Xq[(size_t)b * q * d + e];
    __syncthreads();
    int flag = 0;
    for (int e = tid; e < q * n; e += 256) {
        const int a = e / n, k = e - a * n;
        bool close = true;
        for (int l = 0; l < d; ++l) {
            const double df = xq[a * d + l] - Xtrain[(size_t)k * d + l];
            close = close && (fabs(df * kp.inv_ls[l]) < 1e-2);
        }
        if (close) flag = 1;
    }
    for (int e = tid; e < q * q; e += 256) {
        const int a = e / q, bb = e - a * q;
        bool same = true;
        for (int l = 0; l < d; ++l)
            same = same && (rint(xq[a * d + l] / kp.ls[l] * 100.0) == rint(xq[bb * d + l] / kp.ls[l] * 100.0));
        if (a != bb && same) flag = 1;
    }
    flag = __syncthreads_or(flag);
    if (tid == 0) flags[b] = flag ? 1 : 0;
}

extern "C" int oei_project_flags(oei_model* m, int B, int q, const double* X, int* flags, void* stream) {
    int rc = check_model_call(m, B, q, X, "project_flags");
    if (rc) return rc;
    if (B == 0) return OEI_OK;
    if (!flags) { oei_set_error("project_flags: null output"); return OEI_ERR_ARG; }
    hipLaunchKernelGGL(k_project_flags, dim3(B), dim3(256), (size_t)q * m->d * sizeof(double), (hipStream_t)stream,
                       m->kp, m->n, q, m->X, X, flags);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

extern "C" int oei_set_profiling(oei_model* m, int on) {
    if (!m) { oei_set_error("set_profiling: null model"); return OEI_ERR_ARG; }
    if (on && !m->ev[0]) {
        hipEvent_t tmp[8] = {};
        for (int i = 0; i < 8; ++i) {
            const hipError_t e = hipEventCreate(&tmp[i]);
            if (e != hipSuccess) {           // all eight or none
                for (int j = 0; j < i; ++j) hipEventDestroy(tmp[j]);
                oei_set_error("set_profiling: hipEventCreate failed: %s", hipGetErrorString(e));
                return OEI_ERR_HIP;
            }
        }
        for (int i = 0; i < 6; ++i) m->ev[i] = tmp[i];
        m->kev[0] = tmp[6]; m->kev[1] = tmp[7];
    }
    m->profile = on;
    m->ev_valid = 0;
    return OEI_OK;
}

extern "C" int oei_get_stage_times(oei_model* m, float* ms5) {
    if (!m || !ms5) { oei_set_error("get_stage_times: null pointer"); return OEI_ERR_ARG; }
    if (!m->ev_valid) { oei_set_error("get_stage_times: no profiled call recorded"); return OEI_ERR_ARG; }
    OEI_HIP_CHECK(hipEventSynchronize(m->ev[5]));
    for (int i = 0; i < 5; ++i) OEI_HIP_CHECK(hipEventElapsedTime(&ms5[i], m->ev[i], m->ev[i + 1]));
    return OEI_OK;
}

extern "C" int oei_get_solver_kernel_time(oei_model* m, float* ms) {
    if (!m || !ms) { oei_set_error("get_solver_kernel_time: null pointer"); return OEI_ERR_ARG; }
    if (!m->ev_valid) { oei_set_error("get_solver_kernel_time: no profiled call recorded"); return OEI_ERR_ARG; }
    OEI_HIP_CHECK(hipEventSynchronize(m->kev[1]));
    OEI_HIP_CHECK(hipEventElapsedTime(ms, m->kev[0], m->kev[1]));
    return OEI_OK;
}

extern "C" int oei_argmin(int B, const double* value, long long index_base, double* best_value,
                          long long* best_index, void* stream) {
    if (B < 0 || !best_value || !best_index || (B > 0 && !value)) { oei_set_error("argmin: bad argument"); return OEI_ERR_ARG; }
    hipLaunchKernelGGL(k_argmin, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, value, index_base, best_value, best_index);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

extern "C" int oei_select_record(int nrec, int reclen, const double* records, double* out, void* stream) {
    if (nrec < 1 || reclen < 2 || !records || !out) { oei_set_error("select_record: bad argument"); return OEI_ERR_ARG; }
    hipLaunchKernelGGL(k_select_record, dim3(1), dim3(256), 0, (hipStream_t)stream, nrec, reclen, records, out);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}

extern "C" int oei_best_record(int B, const double* value, const double* X, int xlen, long long index_base,
                               double* record, void* stream) {
    if (B < 0 || xlen < 0 || !record || (B > 0 && (!value || (xlen > 0 && !X)))) { oei_set_error("best_record: bad argument"); return OEI_ERR_ARG; }
    hipLaunchKernelGGL(k_best_record, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, value, X, xlen, index_base, record);
    OEI_HIP_CHECK(hipGetLastError());
    return OEI_OK;
}
