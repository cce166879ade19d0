// fq32.h -- BLS12-381 base-field arithmetic on 12 x 32-bit limbs (little-endian
// limb order, Montgomery form with R = 2^384).  Written for gfx950: the inner
// product step is one v_mad_u64_u32 per limb pair (full-rate on CDNA4, see
// profiles/r01_intrate_microbench.txt).  The same source compiles for the host
// (BLS_HD empty) so tests/test_fq32_host.py can check it against Python ints.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BLS_HD __host__ __device__ __forceinline__
#else
#define BLS_HD static inline
#endif

namespace bls {

#define BLS_Q_LIMBS                                                                      \
    {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,     \
     0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau}
// R^3 mod q (R = 2^384): content c^-1 times this, Montgomery-multiplied, is c^-1 R^2
#define BLS_R3_LIMBS                                                                     \
    {0xd94ca1e0u, 0xed48ac6bu, 0x03a7adf8u, 0x315f831eu, 0x615e29ddu, 0x9a53352au,    \
     0x921e1761u, 0x34c04e5eu, 0x65724728u, 0x2512d435u, 0x91755d4du, 0x0aa63460u}
constexpr uint32_t QINV32 = 0xfffcfffdu;   // -q^-1 mod 2^32

struct Fq { uint32_t l[12]; };

// r = a * b * R^-1 mod q, canonical (CIOS Montgomery)
BLS_HD void fq_mul(uint32_t* __restrict__ r, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b) {
    const uint32_t q[12] = BLS_Q_LIMBS;
    uint32_t t[13];
#pragma unroll
    for (int j = 0; j < 13; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            c = (uint64_t)a[j] * b[i] + t[j] + c;
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        uint32_t t12 = t[12] + (uint32_t)c;   // stays below 2^32 because q < 2^381
        uint32_t m = t[0] * QINV32;
        c = (uint64_t)m * q[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 12; j++) {
            c = (uint64_t)m * q[j] + t[j] + c;
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t12;
        t[11] = (uint32_t)c;
        t[12] = (uint32_t)(c >> 32);
    }
    uint32_t d[12];
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        uint64_t x = (uint64_t)t[j] - q[j] - br;
        d[j] = (uint32_t)x;
        br = (uint32_t)(x >> 63);
    }
    bool ge = (t[12] != 0) || (br == 0);
#pragma unroll
    for (int j = 0; j < 12; j++) r[j] = ge ? d[j] : t[j];
}

// acc = (acc + s) mod q for acc < q, s <= q
BLS_HD void fq_add_mod(uint32_t* __restrict__ acc, const uint32_t* __restrict__ s) {
    const uint32_t q[12] = BLS_Q_LIMBS;
    uint32_t t[12], d[12];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        c += (uint64_t)acc[j] + s[j];
        t[j] = (uint32_t)c;
        c >>= 32;
    }
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        uint64_t x = (uint64_t)t[j] - q[j] - br;
        d[j] = (uint32_t)x;
        br = (uint32_t)(x >> 63);
    }
    bool ge = (c != 0) || (br == 0);
#pragma unroll
    for (int j = 0; j < 12; j++) acc[j] = ge ? d[j] : t[j];
}

// s = q - s (so that acc + s == acc - s_old mod q); maps 0 -> q, which
// fq_add_mod accepts
BLS_HD void fq_neg_raw(uint32_t* __restrict__ s) {
    const uint32_t q[12] = BLS_Q_LIMBS;
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        uint64_t x = (uint64_t)q[j] - s[j] - br;
        s[j] = (uint32_t)x;
        br = (uint32_t)(x >> 63);
    }
}

BLS_HD bool fq_is_zero(const uint32_t* a) {
    uint32_t t = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) t |= a[j];
    return t == 0;
}

// ---- helpers for the inversion ------------------------------------------
BLS_HD bool big_is_one(const uint32_t* a) {
    uint32_t t = a[0] ^ 1u;
#pragma unroll
    for (int j = 1; j < 12; j++) t |= a[j];
    return t == 0;
}
BLS_HD bool big_geq(const uint32_t* a, const uint32_t* b) {   // a >= b
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        uint64_t x = (uint64_t)a[j] - b[j] - br;
        br = (uint32_t)(x >> 63);
    }
    return br == 0;
}
BLS_HD void big_sub(uint32_t* a, const uint32_t* b) {         // a -= b (a >= b)
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        uint64_t x = (uint64_t)a[j] - b[j] - br;
        a[j] = (uint32_t)x;
        br = (uint32_t)(x >> 63);
    }
}
BLS_HD void big_shr1(uint32_t* a, uint32_t top) {             // a = (top:a) >> 1
#pragma unroll
    for (int j = 0; j < 11; j++) a[j] = (a[j] >> 1) | (a[j + 1] << 31);
    a[11] = (a[11] >> 1) | (top << 31);
}
// x = x / 2 mod q  (x < q)
BLS_HD void fq_half(uint32_t* x) {
    const uint32_t q[12] = BLS_Q_LIMBS;
    uint32_t carry = 0;
    if (x[0] & 1u) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            c += (uint64_t)x[j] + q[j];
            x[j] = (uint32_t)c;
            c >>= 32;
        }
        carry = (uint32_t)c;
    }
    big_shr1(x, carry);
}
// x = (x - y) mod q  (x, y < q)
BLS_HD void fq_sub_mod(uint32_t* x, const uint32_t* y) {
    const uint32_t q[12] = BLS_Q_LIMBS;
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        uint64_t t = (uint64_t)x[j] - y[j] - br;
        x[j] = (uint32_t)t;
        br = (uint32_t)(t >> 63);
    }
    if (br) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            c += (uint64_t)x[j] + q[j];
            x[j] = (uint32_t)c;
            c >>= 32;
        }
    }
}

// Montgomery inverse: content a = x R  ->  content x^-1 R ; 0 -> 0 (the
// reference's fq_invert returns 0 for 0, fields_t.py:47-55).
// Binary extended Euclid on the content (plain inverse c^-1), then one
// Montgomery product by R^3 to land back in the domain.
BLS_HD void fq_inv(uint32_t* __restrict__ r, const uint32_t* __restrict__ a) {
    const uint32_t q[12] = BLS_Q_LIMBS;
    const uint32_t r3[12] = BLS_R3_LIMBS;
    uint32_t u[12], v[12], x1[12], x2[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { u[j] = a[j]; v[j] = q[j]; x1[j] = 0; x2[j] = 0; }
    x1[0] = 1;
    if (fq_is_zero(u)) {
#pragma unroll
        for (int j = 0; j < 12; j++) r[j] = 0;
        return;
    }
    // invariant: x1 * a == u, x2 * a == v (mod q)
    while (!big_is_one(u) && !big_is_one(v)) {
        while (!(u[0] & 1u)) { big_shr1(u, 0); fq_half(x1); }
        while (!(v[0] & 1u)) { big_shr1(v, 0); fq_half(x2); }
        if (big_geq(u, v)) { big_sub(u, v); fq_sub_mod(x1, x2); }
        else { big_sub(v, u); fq_sub_mod(x2, x1); }
    }
    const bool from_u = big_is_one(u);
    uint32_t tmp[12];
#pragma unroll
    for (int j = 0; j < 12; j++) tmp[j] = from_u ? x1[j] : x2[j];
    fq_mul(r, tmp, r3);
}

}  // namespace bls
