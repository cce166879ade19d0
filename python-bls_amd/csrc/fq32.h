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

#if defined(__HIP_DEVICE_COMPILE__)
#include "fq_mul_gfx950.h"
#endif

namespace bls {

// add / subtract with carry: clang lowers the builtins to v_addc_co_u32 /
// v_subb_co_u32 chains; plain C++ for other host compilers
#if defined(__clang__)
BLS_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t& c) { unsigned co; uint32_t r = __builtin_addc(a, b, c, &co); c = co; return r; }
BLS_HD uint32_t subc(uint32_t a, uint32_t b, uint32_t& c) { unsigned co; uint32_t r = __builtin_subc(a, b, c, &co); c = co; return r; }
#else
BLS_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t& c) { uint64_t t = (uint64_t)a + b + c; c = (uint32_t)(t >> 32); return (uint32_t)t; }
BLS_HD uint32_t subc(uint32_t a, uint32_t b, uint32_t& c) { uint64_t t = (uint64_t)a - b - c; c = (uint32_t)(t >> 63); return (uint32_t)t; }
#endif

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
#if defined(__HIP_DEVICE_COMPILE__)
    fq_mul_dev<true>(r, a, b);       // v_mad_u64_u32 / v_addc_co_u32 columns (fq_mul_gfx950.h)
    return;
#endif
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

// ---- relaxed residues ------------------------------------------------------
// Inside the VM every stored value is only kept below 2q ("relaxed"): the
// Montgomery product of two values below 3q is below 2q without a final
// subtraction, and a linear combination is brought back below 2q with one
// quotient estimate.  Canonical form is restored where it is observable
// (outputs, the inversion's zero test).

// r = a * b * R^-1 mod q with r < 2q for a * b < 9 q^2 (host model of fq_mul_dev<false>)
BLS_HD void fq_mul_relaxed(uint32_t* __restrict__ r, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b) {
#if defined(__HIP_DEVICE_COMPILE__)
    fq_mul_dev<false>(r, a, b);
#else
    const uint32_t q[12] = BLS_Q_LIMBS;
    uint32_t t[14];
    for (int j = 0; j < 14; j++) t[j] = 0;
    for (int i = 0; i < 12; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 12; j++) { c += (uint64_t)a[j] * b[i] + t[j]; t[j] = (uint32_t)c; c >>= 32; }
        c += t[12]; t[12] = (uint32_t)c; t[13] = (uint32_t)(c >> 32);
        uint32_t m = t[0] * QINV32;
        c = (uint64_t)m * q[0] + t[0]; c >>= 32;
        for (int j = 1; j < 12; j++) { c += (uint64_t)m * q[j] + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
        c += t[12]; t[11] = (uint32_t)c; c >>= 32;
        t[12] = t[13] + (uint32_t)c; t[13] = 0;
    }
    for (int j = 0; j < 12; j++) r[j] = t[j];   // t[12] == 0 whenever the result is below 2^384
#endif
}
// r = a^2 * R^-1 mod q with r < 2q for a < 3q: the dedicated squaring columns on the GPU
// (fq_mul_gfx950.h: 78 + 144 multiply-accumulates instead of 288), the product on the host
BLS_HD void fq_sqr_relaxed(uint32_t* __restrict__ r, const uint32_t* __restrict__ a) {
#if defined(__HIP_DEVICE_COMPILE__)
    fq_sqr_dev(r, a);
#else
    uint32_t b[12];
    for (int j = 0; j < 12; j++) b[j] = a[j];
    fq_mul_relaxed(r, a, b);
#endif
}
// x (< 2q) -> canonical
BLS_HD void fq_canon(uint32_t* __restrict__ x) {
    const uint32_t q[12] = BLS_Q_LIMBS;
    uint32_t d[12];
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) d[j] = subc(x[j], q[j], br);
    const bool ge = (br == 0);
#pragma unroll
    for (int j = 0; j < 12; j++) x[j] = ge ? d[j] : x[j];
}
// ---- carry-free linear combinations (the VM's LIN rounds) --------------------
// gfx950 needs wait states between a VALU write of VCC and a dependent VALU read,
// so long v_addc chains are slow.  A linear combination is therefore accumulated
// limb by limb into 64-bit "fat" limbs with independent v_mad_u64_u32 (no carries,
// no VCC): first the negative terms as plain sums, one sign flip (fat_flip), then the
// positive terms (vmgen/emit.py plan_lin_round); carries are resolved once in fat_reduce.
#define BLS_QC_LIMBS /* 2^384 - q */ \
    {0x00005555u, 0x46010000u, 0x4eac0000u, 0xe1540001u, 0x094f09dbu, 0x98cf2d5fu, 0x0c7aed40u, 0x9b88b47bu, 0xbcb45328u, 0xb4e45849u, 0xc6801965u, 0xe5feee15u}

#define BLS_BIAS1_FAT /* BIAS + 1 per 64-bit limb; BIAS = 0 mod q, every limb 2^40 + (32-bit digit) */ \
    {0x00000100fcb7adf4ull, 0x00000100a026ff00ull, 0x000001004433fc4full, 0x000001000bcbf221ull, 0x0000010054a7e8b3ull, 0x000001002fca321eull, 0x0000010019759de0ull, 0x000001005ac6af43ull, 0x00000100b439141dull, 0x00000100a35690ddull, 0x000001003c85e46eull, 0x0000010014896a91ull}
// acc <- BIAS - acc limb by limb (every acc[j] < 2^40): the sum of the negative terms changes sign
BLS_HD void fat_flip(uint64_t* __restrict__ acc) {
    const uint64_t b1[12] = BLS_BIAS1_FAT;
#pragma unroll
    for (int j = 0; j < 12; j++) acc[j] = b1[j] + ~acc[j];
}
// acc[j] += cf * s[j]
BLS_HD void fat_mac_plain(uint64_t* __restrict__ acc, const uint32_t* __restrict__ s, uint32_t cf) {
#pragma unroll
    for (int j = 0; j < 12; j++) acc[j] += (uint64_t)s[j] * cf;
}
// out = V mod q with 0 <= out < 2q, V = sum acc[j] 2^(32 j) < 2^396 (every acc[j] < 2^44)
BLS_HD void fat_reduce(uint32_t* __restrict__ out, const uint64_t* __restrict__ acc) {
    const uint32_t qc[12] = BLS_QC_LIMBS;
    // V / q from the top fat limb in double precision (exact conversion; what is ignored
    // -- the lower limbs and the low bits of q -- is worth < 2e-4): k = floor(. - 1e-3) is
    // floor(V / q) or one less
    double hf = (double)(uint32_t)(acc[11] >> 32) * 4294967296.0 + (double)(uint32_t)acc[11];
    double e = hf * (1.0 / 436277738.0) - 0.001;        // q >> 352 = 0x1a0111ea
    uint32_t k = (e > 0.0) ? (uint32_t)e : 0u;
    // (V + k (2^384 - q)) mod 2^384 = V - k q
#if defined(__HIP_DEVICE_COMPILE__)
    // twelve independent multiply-adds, then ONE add-with-carry per limb (out_j = lo_j + hi_(j-1) + carry)
    // instead of a 64-bit add of the running carry per limb (mad + move + 64-bit add)
    uint64_t t[12];
#pragma unroll
    for (int j = 0; j < 12; j++) t[j] = acc[j] + (uint64_t)k * qc[j];
    out[0] = (uint32_t)t[0];
    uint32_t cy = 0;
#pragma unroll
    for (int j = 1; j < 12; j++) out[j] = addc((uint32_t)t[j], (uint32_t)(t[j - 1] >> 32), cy);
#else
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        uint64_t t = acc[j] + (uint64_t)k * qc[j] + c;
        out[j] = (uint32_t)t;
        c = t >> 32;
    }
#endif
}

// ---- sign of a field element (the VM's SGN rounds) ----------------------------
#define BLS_HALF_LIMBS /* (q - 1) / 2 */ \
    {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u, 0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u}
#define BLS_R2_LIMBS /* R^2 mod q: x * R2 (Montgomery product) = x R */ \
    {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u, 0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u}
#define BLS_ONE_MONT_LIMBS /* R mod q */ \
    {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u, 0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u}
// out = Montgomery 1 if the canonical value of a (Montgomery, relaxed) exceeds (q-1)/2, else 0:
// the "lexicographically larger than its negation" test of ec.py:94-100
BLS_HD void fq_sgn(uint32_t* __restrict__ out, const uint32_t* __restrict__ a) {
    const uint32_t half[12] = BLS_HALF_LIMBS;
    const uint32_t onem[12] = BLS_ONE_MONT_LIMBS;
    const uint32_t raw1[12] = {1u, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t x[12];
    fq_mul_relaxed(x, a, raw1);       // leaves the Montgomery domain; result <= q
    fq_canon(x);
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) (void)subc(half[j], x[j], br);   // borrow <=> x > half
    const bool gt = br != 0;
#pragma unroll
    for (int j = 0; j < 12; j++) out[j] = gt ? onem[j] : 0u;
}

// ---- modular subtraction (register arithmetic of blsgpu_reg.hip) ------------
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
//
// Plain inverse of the content by Bernstein-Yang "safegcd" division steps (eprint 2019/266),
// branch-free: 37 batches of 30 divsteps on the low words, each batch applied to the full
// (f, g) and (d, e) as a 2x2 integer matrix.  1110 >= floor((49 * 381 + 57) / 17) = 1101
// divsteps suffice for 381-bit inputs (theorem 11.2 of the paper, delta = 1).  Every lane
// runs the same instruction sequence -- what a lock-step wavefront needs; the binary
// Euclid it replaces diverged at every step.  Then one Montgomery product by R^3.
// Numbers are 13 signed limbs of 30 bits.
#define BLS_Q30_LIMBS {0x3fffaaab, 0x27fbffff, 0x153ffffb, 0x2affffac, 0x30f6241e, 0x034a83da, 0x112bf673, 0x12e13ce1, 0x2cd76477, 0x1ed90d2e, 0x29a4b1ba, 0x3a8e5ff9, 0x001a0111}
#define BLS_Q_INV30 0x00030003u   /* q^-1 mod 2^30 */

struct inv_trans { int32_t u, v, q, r; };

// 30 division steps on the low words; returns the new eta = -delta; t maps (f, g) to 2^30 (f', g')
BLS_HD int32_t inv_divsteps30(int32_t eta, uint32_t f0, uint32_t g0, inv_trans& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
    for (int i = 0; i < 30; i++) {
        uint32_t c1 = (uint32_t)(eta >> 31);          // delta > 0
        const uint32_t c2 = 0u - (g & 1u);            // g odd
        const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
        g += x & c2; q += y & c2; r += z & c2;
        c1 &= c2;
        eta = (int32_t)(((uint32_t)eta ^ c1) - (c1 + 1u));
        f += g & c1; u += q & c1; v += r & c1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return eta;
}
// (f, g) <- t (f, g) / 2^30 (exact)
BLS_HD void inv_update_fg(int32_t* f, int32_t* g, const inv_trans& t) {
    const int32_t M30 = 0x3FFFFFFF;
    int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
    int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
    cf >>= 30; cg >>= 30;
#pragma unroll
    for (int i = 1; i < 13; i++) {
        cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
        cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
        f[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g[i - 1] = (int32_t)cg & M30; cg >>= 30;
    }
    f[12] = (int32_t)cf; g[12] = (int32_t)cg;
}
// (d, e) <- t (d, e) / 2^30 mod q, both kept in (-2q, q)
BLS_HD void inv_update_de(int32_t* d, int32_t* e, const inv_trans& t) {
    const int32_t M30 = 0x3FFFFFFF;
    const int32_t m[13] = BLS_Q30_LIMBS;
    const int32_t sd = d[12] >> 31, se = e[12] >> 31;
    int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
    int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
    md -= (int32_t)((BLS_Q_INV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((BLS_Q_INV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)m[0] * md; ce += (int64_t)m[0] * me;
    cd >>= 30; ce >>= 30;
#pragma unroll
    for (int i = 1; i < 13; i++) {
        cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)m[i] * md;
        ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)m[i] * me;
        d[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e[i - 1] = (int32_t)ce & M30; ce >>= 30;
    }
    d[12] = (int32_t)cd; e[12] = (int32_t)ce;
}
// r in (-2q, q), sign < 0 means negate: -> canonical [0, q)
BLS_HD void inv_normalize(int32_t* r, int32_t sign) {
    const int32_t M30 = 0x3FFFFFFF;
    const int32_t m[13] = BLS_Q30_LIMBS;
    int32_t cond_add = r[12] >> 31;
    const int32_t cond_neg = sign >> 31;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        int32_t x = r[i] + (m[i] & cond_add);
        x = (x ^ cond_neg) - cond_neg;
        x += c;
        c = x >> 30;
        r[i] = (i < 12) ? (x & M30) : x;
    }
    cond_add = r[12] >> 31;
    c = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        int32_t x = r[i] + (m[i] & cond_add) + c;
        c = x >> 30;
        r[i] = (i < 12) ? (x & M30) : x;
    }
}

BLS_HD void fq_inv(uint32_t* __restrict__ r, const uint32_t* __restrict__ a) {
    const uint32_t r3[12] = BLS_R3_LIMBS;
    const int32_t m[13] = BLS_Q30_LIMBS;
    uint32_t x[12];
#pragma unroll
    for (int j = 0; j < 12; j++) x[j] = a[j];
    fq_canon(x);                      // the VM hands over a relaxed value (< 2q)
    // 12 x 32 bits -> 13 x 30 bits
    int32_t f[13], g[13], d[13], e[13];
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        uint32_t lo = x[w] >> s;
        if (s > 2 && w + 1 < 12) lo |= x[w + 1] << (32 - s);
        g[i] = (int32_t)(lo & 0x3FFFFFFFu);
        f[i] = m[i]; d[i] = 0; e[i] = 0;
    }
    e[0] = 1;
    int32_t eta = -1;
#pragma unroll 1
    for (int it = 0; it < 37; it++) {
        inv_trans t;
        eta = inv_divsteps30(eta, (uint32_t)f[0], (uint32_t)g[0], t);
        inv_update_de(d, e, t);
        inv_update_fg(f, g, t);
    }
    // now g = 0 and f = +-gcd = +-1 (or f = +-q when the input was 0: then d = 0)
    inv_normalize(d, f[12]);
    // 13 x 30 bits -> 12 x 32 bits
    uint32_t y[12];
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 32 * w, i = bit / 30, s = bit - 30 * i;
        uint32_t v = (uint32_t)d[i] >> s;
        v |= (uint32_t)d[i + 1] << (30 - s);
        if (60 - s < 32 && i + 2 < 13) v |= (uint32_t)d[i + 2] << (60 - s);
        y[w] = v;
    }
    fq_mul(r, y, r3);
}

// ---- the same inverse for a value that is THE SAME IN EVERY LANE of the wavefront (round 4) ----------------------------
// k_fexp_wide (blsgpu_fexpw.hip) inverts one Fq value per result while the whole wavefront waits: every lane runs the
// routine on the same number, so data-dependent control flow costs nothing (the branches are wave-uniform) and the
// constant-time schedule above -- 37 x 30 single division steps, ~990 instructions per batch -- is not needed.  Same
// algorithm and matrices (Bernstein-Yang division steps, eprint 2019/266), in the variable-time form: runs of zero bits
// of g are shifted out at once (count of trailing zeros) and up to six low bits of g are cancelled per iteration with
// w = -g/f mod 2^6 (f (f^2 - 2) = -1/f mod 64 for odd f), so a batch of 30 steps takes ~6 iterations, and the loop ends as
// soon as g = 0 (~27 batches on average for 381-bit inputs instead of the proven bound 37).  Results are identical to
// fq_inv's for every input, 0 -> 0 included (tests/test_abi_and_host.py runs both on the host).
// low 32 bits of a b + c: on the GPU one full-rate v_mad_u64_u32 (v_mul_lo_u32 is a quarter-rate instruction)
BLS_HD uint32_t inv_mad32(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t t;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=&v"(t) : "v"(a), "v"(b), "v"((uint64_t)c) : "vcc");
    return (uint32_t)t;
#else
    return a * b + c;
#endif
}
BLS_HD int32_t inv_divsteps30_var(int32_t eta, uint32_t f0, uint32_t g0, inv_trans& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    int i = 30;
    for (;;) {
        const uint32_t zeros = (uint32_t)__builtin_ctz(g | (0xFFFFFFFFu << i));     // at most i
        g >>= zeros; u <<= zeros; v <<= zeros;
        eta -= (int32_t)zeros; i -= (int)zeros;
        if (i == 0) break;
        // f and g are odd
        if (eta < 0) {                                   // delta > 0: (f, g) <- (g, -f), the matrix rows likewise
            eta = -eta;
            uint32_t tmp = f; f = g; g = 0u - tmp;
            tmp = u; u = q; q = 0u - tmp;
            tmp = v; v = r; r = 0u - tmp;
        }
        // cancel min(eta + 1, i, 6) low bits of g:  g <- g + w f  with  w = -g / f mod 2^limit
        const int lim = (eta + 1) > i ? i : (eta + 1);
        const uint32_t m = (0xFFFFFFFFu >> (32 - lim)) & 63u;
        const uint32_t fi = inv_mad32(f, inv_mad32(f, f, 0u - 2u), 0u);          // f (f^2 - 2) = -1/f mod 2^6
        const uint32_t w = inv_mad32(g, fi, 0u) & m;
        g = inv_mad32(f, w, g); q = inv_mad32(u, w, q); r = inv_mad32(v, w, r);
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return eta;
}
BLS_HD void fq_inv_var(uint32_t* __restrict__ r, const uint32_t* __restrict__ a) {
    const uint32_t r3[12] = BLS_R3_LIMBS;
    const int32_t m[13] = BLS_Q30_LIMBS;
    uint32_t x[12];
#pragma unroll
    for (int j = 0; j < 12; j++) x[j] = a[j];
    fq_canon(x);
    int32_t f[13], g[13], d[13], e[13];
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        uint32_t lo = x[w] >> s;
        if (s > 2 && w + 1 < 12) lo |= x[w + 1] << (32 - s);
        g[i] = (int32_t)(lo & 0x3FFFFFFFu);
        f[i] = m[i]; d[i] = 0; e[i] = 0;
    }
    e[0] = 1;
    int32_t eta = -1;
#pragma unroll 1
    for (int it = 0; it < 37; it++) {                    // never more than the constant-time bound
        int32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 13; i++) nz |= g[i];
        if (nz == 0) break;
        inv_trans t;
        eta = inv_divsteps30_var(eta, (uint32_t)f[0], (uint32_t)g[0], t);
        inv_update_de(d, e, t);
        inv_update_fg(f, g, t);
    }
    inv_normalize(d, f[12]);
    uint32_t y[12];
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 32 * w, i = bit / 30, s = bit - 30 * i;
        uint32_t v = (uint32_t)d[i] >> s;
        v |= (uint32_t)d[i + 1] << (30 - s);
        if (60 - s < 32 && i + 2 < 13) v |= (uint32_t)d[i + 2] << (60 - s);
        y[w] = v;
    }
    fq_mul(r, y, r3);
}

// ---- the quadratic character (a / q) from the same division steps (round 4) ------------------------------------------------
// Hash to G2 decides three quadratic characters per encoding (csrc/blsgpu_h2c.hip: which candidate x has a square norm,
// which of delta+- is a square; ec.py:489-500, fields.py:463-482).  Round 3 did it with the binary algorithm on the
// full numbers: 768 iterations of compare / swap / subtract / halve on 12 words, ~77 k instructions.  Here the Jacobi
// symbol rides on division steps that keep f and g NON-NEGATIVE (so that the reciprocity and the (2 / f) rules apply as
// they stand): batches of 30 steps on the low words -- runs of zero bits of g shifted out at once (the sign flips when
// the run is odd and f is 3 or 5 mod 8), f and g swapped when delta > 0 (flip when both are 3 mod 4), up to six low
// bits of g cancelled by adding a multiple of f -- each applied to the full numbers as one 2x2 matrix with
// non-negative entries.  gcd(a, q) = 1 for 0 < a < q, so f reaches 1 (38 batches on average, 42 at most over 30 000
// random values: tests/test_jacobi_model.py holds the integer model and the histogram); the symbol is latched there.
// The loops are data dependent: lanes of a wavefront wait for the slowest (~12 inner iterations, ~42 batches), still a
// third of the binary routine.  A value that has not converged after BLS_JACOBI_BATCHES batches reports 2 and the caller
// falls back to the binary routine (never observed).
#define BLS_JACOBI_BATCHES 56
BLS_HD int32_t jac_posdivsteps30_var(int32_t eta, uint32_t f0, uint32_t g0, inv_trans& t, uint32_t& jac) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    int i = 30;
    for (;;) {
        const uint32_t zeros = (uint32_t)__builtin_ctz(g | (0xFFFFFFFFu << i));
        g >>= zeros; u <<= zeros; v <<= zeros;
        eta -= (int32_t)zeros; i -= (int)zeros;
        jac ^= zeros & ((f >> 1) ^ (f >> 2));            // (2 / f)^zeros: -1 iff zeros is odd and f is 3 or 5 mod 8
        if (i == 0) break;
        if (eta < 0) {                                   // (f, g) <- (g, f): reciprocity, -1 iff both are 3 mod 4
            eta = -eta;
            jac ^= (f & g) >> 1;
            uint32_t tmp = f; f = g; g = tmp;
            tmp = u; u = q; q = tmp;
            tmp = v; v = r; r = tmp;
        }
        const int lim = (eta + 1) > i ? i : (eta + 1);
        const uint32_t m = (0xFFFFFFFFu >> (32 - lim)) & 63u;
        const uint32_t fi = inv_mad32(f, inv_mad32(f, f, 0u - 2u), 0u);          // f (f^2 - 2) = -1/f mod 2^6
        const uint32_t w = inv_mad32(g, fi, 0u) & m;
        g = inv_mad32(f, w, g); q = inv_mad32(u, w, q); r = inv_mad32(v, w, r);
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return eta;
}
// a: canonical residue (12 words).  Returns 1 / -1 (a is a non-zero square / non-square mod q), 0 (a = 0), 2 (not converged).
BLS_HD int fq_jacobi_var(const uint32_t* __restrict__ a) {
    const int32_t m[13] = BLS_Q30_LIMBS;
    int32_t f[13], g[13];
    int32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;
        uint32_t lo = a[w] >> s;
        if (s > 2 && w + 1 < 12) lo |= a[w + 1] << (32 - s);
        g[i] = (int32_t)(lo & 0x3FFFFFFFu);
        f[i] = m[i];
        nz |= g[i];
    }
    int32_t eta = -1;
    uint32_t jac = 0;
    int res = nz == 0 ? 0 : 2;                           // 0: a = 0; 2: still running
#pragma unroll 1
    for (int it = 0; it < BLS_JACOBI_BATCHES; it++) {
        if (res != 2) break;
        inv_trans t;
        // 32 low bits of f and g: 30 steps and the two bits above them that the mod-8 / mod-4 tests read
        const uint32_t f0 = (uint32_t)f[0] | ((uint32_t)f[1] << 30), g0 = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
        eta = jac_posdivsteps30_var(eta, f0, g0, t, jac);
        inv_update_fg(f, g, t);
        int32_t rest = f[0] ^ 1;
#pragma unroll
        for (int i = 1; i < 13; i++) rest |= f[i];
        if (rest == 0) res = (jac & 1u) ? -1 : 1;        // f = 1: the symbol is what has been collected
    }
    return res;
}

}  // namespace bls
