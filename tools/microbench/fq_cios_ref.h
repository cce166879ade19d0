#pragma once
#include <stdint.h>
#define Q0 0xffffaaabu
__device__ __constant__ uint32_t QL[12] = {0xffffaaabu,0xb9feffffu,0xb153ffffu,0x1eabfffeu,0xf6b0f624u,0x6730d2a0u,0xf38512bfu,0x64774b84u,0x434bacd7u,0x4b1ba7b6u,0x397fe69au,0x1a0111eau};
#define QINV32 0xfffcfffdu
static __device__ __forceinline__ void fq_mul_cios(uint32_t* __restrict__ r, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b) {
    const uint32_t q[12] = {0xffffaaabu,0xb9feffffu,0xb153ffffu,0x1eabfffeu,0xf6b0f624u,0x6730d2a0u,0xf38512bfu,0x64774b84u,0x434bacd7u,0x4b1ba7b6u,0x397fe69au,0x1a0111eau};
    uint32_t t[13];
#pragma unroll
    for (int j = 0; j < 13; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            c = (uint64_t)a[j] * b[i] + t[j] + c;
            t[j] = (uint32_t)c; c >>= 32;
        }
        uint32_t t12 = t[12] + (uint32_t)c;   // q < 2^381 keeps this below 2^32
        uint32_t m = t[0] * QINV32;
        c = (uint64_t)m * q[0] + t[0]; c >>= 32;
#pragma unroll
        for (int j = 1; j < 12; j++) {
            c = (uint64_t)m * q[j] + t[j] + c;
            t[j-1] = (uint32_t)c; c >>= 32;
        }
        c += t12;
        t[11] = (uint32_t)c; t[12] = (uint32_t)(c >> 32);
    }
    // conditional subtract
    uint32_t d[12]; uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        uint64_t x = (uint64_t)t[j] - q[j] - br;
        d[j] = (uint32_t)x; br = (uint32_t)(x >> 63);
    }
    bool ge = (t[12] != 0) || (br == 0);
#pragma unroll
    for (int j = 0; j < 12; j++) r[j] = ge ? d[j] : t[j];
}
