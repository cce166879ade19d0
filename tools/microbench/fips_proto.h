#pragma once
#include <stdint.h>
// (t2 : lo64) += a * b     -- one v_mad_u64_u32 + one v_addc_co_u32
#define MAC3(lo64, t2, a, b) \
    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" \
                 : "+v"(lo64), "+v"(t2) : "v"(a), "v"(b) : "vcc")
#define MAC3S(lo64, t2, a, sb) \
    asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" \
                 : "+v"(lo64), "+v"(t2) : "v"(a), "s"(sb) : "vcc")
__device__ __forceinline__ void fq_mul_fips(uint32_t* __restrict__ r, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b) {
    const uint32_t q[12] = {0xffffaaabu,0xb9feffffu,0xb153ffffu,0x1eabfffeu,0xf6b0f624u,0x6730d2a0u,0xf38512bfu,0x64774b84u,0x434bacd7u,0x4b1ba7b6u,0x397fe69au,0x1a0111eau};
    const uint32_t qinv = 0xfffcfffdu;
    uint32_t m[12];
    uint64_t lo = 0; uint32_t t2 = 0;
    uint32_t t[13];
#pragma unroll
    for (int i = 0; i < 12; i++) {
#pragma unroll
        for (int j = 0; j < i; j++) { MAC3(lo, t2, a[j], b[i - j]); MAC3S(lo, t2, m[j], q[i - j]); }
        MAC3(lo, t2, a[i], b[0]);
        m[i] = (uint32_t)lo * qinv;
        MAC3S(lo, t2, m[i], q[0]);
        lo = (lo >> 32) | ((uint64_t)t2 << 32); t2 = 0;
    }
#pragma unroll
    for (int i = 12; i < 24; i++) {
#pragma unroll
        for (int j = i - 11; j < 12; j++) { MAC3(lo, t2, a[j], b[i - j]); MAC3S(lo, t2, m[j], q[i - j]); }
        t[i - 12] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)t2 << 32); t2 = 0;
    }
    t[12] = (uint32_t)lo;
    uint32_t d[12]; unsigned br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) d[j] = __builtin_subc(t[j], q[j], br, &br);
    bool ge = (t[12] != 0) || (br == 0);
#pragma unroll
    for (int j = 0; j < 12; j++) r[j] = ge ? d[j] : t[j];
}
