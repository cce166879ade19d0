// Prototype multipliers for VERDICT r1 item 4 (i): carry-free reduced-radix Montgomery products against the
// shipped 32-bit product-scanning one, in registers, one product chain per lane.  Prints Gmul/s per variant and
// the limbs of one product so that the caller can check them against Python integers (fqmul_radix_check.py).
//   V0  shipped: 12 x 32-bit limbs, v_mad_u64_u32 + v_addc_co_u32 per multiply-accumulate (fq_mul_gfx950.h)
//   V1  14 x 28-bit limbs, R = 2^392: column sums of up to 28 products of 56 bits fit 64 bits: no carries,
//       one interleaved pass (392 multiply-accumulates)
//   V2  13 x 30-bit limbs, R = 2^390: a column of 26 products of 60 bits does NOT fit, so the product and
//       the reduction are separate passes (2 x 169 multiply-accumulates, 26 + 26 column extractions)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../python-bls_amd/csrc/fq32.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// q in radix 2^28 / 2^30 and -q^-1 mod 2^28 / 2^30 (filled in by main from the 32-bit limbs)
__constant__ uint32_t N28[14], N30[13];
__constant__ uint32_t NINV28, NINV30;

template <int W, int L>
__device__ __forceinline__ void mont_interleaved(uint32_t* __restrict__ r, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                 const uint32_t* __restrict__ N, uint32_t ninv) {
    constexpr uint32_t MASK = (1u << W) - 1u;
    uint32_t m[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * N[k - i];
        m[k] = ((uint32_t)acc * ninv) & MASK;
        acc += (uint64_t)m[k] * N[0];
        acc >>= W;
    }
#pragma unroll
    for (int k = L; k < 2 * L; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * N[k - i];
        r[k - L] = (uint32_t)acc & MASK;
        acc >>= W;
    }
}

template <int W, int L>
__device__ __forceinline__ void mont_two_pass(uint32_t* __restrict__ r, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                              const uint32_t* __restrict__ N, uint32_t ninv) {
    constexpr uint32_t MASK = (1u << W) - 1u;
    uint32_t t[2 * L], m[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
        for (int i = (k < L ? 0 : k - L + 1); i <= (k < L ? k : L - 1); i++) acc += (uint64_t)a[i] * b[k - i];
        t[k] = (uint32_t)acc & MASK;
        acc >>= W;
    }
    t[2 * L - 1] = (uint32_t)acc;
    acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
        acc += t[k];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * N[k - i];
        m[k] = ((uint32_t)acc * ninv) & MASK;
        acc += (uint64_t)m[k] * N[0];
        acc >>= W;
    }
#pragma unroll
    for (int k = L; k < 2 * L; k++) {
        acc += t[k];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * N[k - i];
        r[k - L] = (uint32_t)acc & MASK;
        acc >>= W;
    }
}

template <int V> struct Cfg;
template <> struct Cfg<0> { static constexpr int L = 12; };
template <> struct Cfg<1> { static constexpr int L = 14; };
template <> struct Cfg<2> { static constexpr int L = 13; };

template <int V>
__global__ void k_mul(uint32_t* out, int iters, uint32_t s, int dump) {
    constexpr int L = Cfg<V>::L;
    uint32_t a[L], b[L];
    const uint32_t mask = V == 0 ? 0xFFFFFFFFu : (V == 1 ? (1u << 28) - 1u : (1u << 30) - 1u);
#pragma unroll
    for (int j = 0; j < L; j++) { a[j] = (threadIdx.x * 977u + j * s + 1) & mask; b[j] = (blockIdx.x * 31u + j * 2654435761u + s) & mask; }
    a[L - 1] &= 0x000fffffu; b[L - 1] &= 0x000fffffu;            // values below 2^380
    if (dump) {
#pragma unroll
        for (int j = 0; j < L; j++) { a[j] = (0x12345678u * (j + 1) + 0x9abcdefu) & mask; b[j] = (0x0fedcba9u * (j + 3) + 0x7654321u) & mask; }
        a[L - 1] &= 0x000fffffu; b[L - 1] &= 0x000fffffu;
    }
    for (int i = 0; i < iters; i++) {
        uint32_t r[L];
        if (V == 0) bls::fq_mul_relaxed(r, a, b);
        else if (V == 1) mont_interleaved<28, 14>(r, a, b, N28, NINV28);
        else mont_two_pass<30, 13>(r, a, b, N30, NINV30);
#pragma unroll
        for (int j = 0; j < L; j++) { b[j] = a[j]; a[j] = r[j]; }
    }
    if (dump) {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int j = 0; j < L; j++) { out[j] = a[j]; out[L + j] = b[j]; }
        return;
    }
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < L; j++) x ^= a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}

template <typename F> double timeit(F f, int reps = 3) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; r++) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    return best;
}

static void split(const uint32_t* q32, int W, int L, uint32_t* out) {
    for (int j = 0; j < L; j++) {
        uint64_t v = 0;
        for (int bit = 0; bit < W; bit++) {
            int p = j * W + bit;
            if (p < 384 && ((q32[p / 32] >> (p % 32)) & 1u)) v |= 1ull << bit;
        }
        out[j] = (uint32_t)v;
    }
}
static uint32_t neg_inv(uint32_t n0, int W) {           // -n0^-1 mod 2^W
    uint32_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - n0 * x;
    return (0u - x) & ((W == 32) ? 0xFFFFFFFFu : ((1u << W) - 1u));
}

int main() {
    const uint32_t q32[12] = BLS_Q_LIMBS;
    uint32_t n28[14], n30[13];
    split(q32, 28, 14, n28); split(q32, 30, 13, n30);
    uint32_t i28 = neg_inv(n28[0], 28), i30 = neg_inv(n30[0], 30);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(N28), n28, sizeof n28)); CK(hipMemcpyToSymbol(HIP_SYMBOL(N30), n30, sizeof n30));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(NINV28), &i28, 4)); CK(hipMemcpyToSymbol(HIP_SYMBOL(NINV30), &i30, 4));
    uint32_t* d; CK(hipMalloc(&d, 1 << 26));
    uint32_t h[64];
    // one product each, for the Python check: prints a_in, b_in implicitly (fixed pattern) and the result limbs
    hipLaunchKernelGGL(k_mul<0>, dim3(1), dim3(64), 0, 0, d, 1, 3u, 1); CK(hipMemcpy(h, d, 256, hipMemcpyDeviceToHost));
    printf("dump V0 W=32 L=12:"); for (int j = 0; j < 24; j++) printf(" %08x", h[j]); printf("\n");
    hipLaunchKernelGGL(k_mul<1>, dim3(1), dim3(64), 0, 0, d, 1, 3u, 1); CK(hipMemcpy(h, d, 256, hipMemcpyDeviceToHost));
    printf("dump V1 W=28 L=14:"); for (int j = 0; j < 28; j++) printf(" %08x", h[j]); printf("\n");
    hipLaunchKernelGGL(k_mul<2>, dim3(1), dim3(64), 0, 0, d, 1, 3u, 1); CK(hipMemcpy(h, d, 256, hipMemcpyDeviceToHost));
    printf("dump V2 W=30 L=13:"); for (int j = 0; j < 26; j++) printf(" %08x", h[j]); printf("\n");
    const int cus = 256;
    for (int wps : {1, 2, 4}) {
        int blocks = cus * wps, it = 2000;
        double ms;
        ms = timeit([&] { hipLaunchKernelGGL(k_mul<0>, dim3(blocks), dim3(256), 0, 0, d, it, 3u, 0); });
        printf("V0 shipped 12x32 (mad + addc)      wps=%d  %.3f ms  %.2f Gmul/s\n", wps, ms, (double)blocks * 256 * it / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL(k_mul<1>, dim3(blocks), dim3(256), 0, 0, d, it, 3u, 0); });
        printf("V1 14x28 carry-free, interleaved   wps=%d  %.3f ms  %.2f Gmul/s\n", wps, ms, (double)blocks * 256 * it / ms * 1e-6);
        ms = timeit([&] { hipLaunchKernelGGL(k_mul<2>, dim3(blocks), dim3(256), 0, 0, d, it, 3u, 0); });
        printf("V2 13x30 carry-free, two passes    wps=%d  %.3f ms  %.2f Gmul/s\n", wps, ms, (double)blocks * 256 * it / ms * 1e-6);
    }
    return 0;
}
