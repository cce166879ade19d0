// fp28.h -- BLS12-381 field and curve arithmetic with ONE ITEM PER LANE, everything in registers, on 14 SIGNED
// limbs of 28 bits in Montgomery form with R = 2^392 (round 3; replaces the 12 x 32-bit register arithmetic of
// rounds 1-2).  gfx950 device code only.
//
// Why: a 32-bit-limb multiply-accumulate is v_mad_u64_u32 + v_addc_co_u32; with 28-bit limbs a whole column of a
// product fits one 64-bit accumulator, so it is ONE v_mad_i64_i32 (vmgen/gen_fp28.py -> fp28_mul_gfx950.h:
// 462 instructions per product against 637-676; measured 78.6 G against 61.8 G products/s,
// profiles/r03_fp28_microbench.txt).  The headroom is used three times over:
//   * additions and subtractions are limb-wise (14 v_add / v_sub): no carries, no modular correction;
//   * operands of a product may be such unreduced sums and differences;
//   * a product is a SUM of products with ONE Montgomery reduction (dot2 / dot3 / dot4): X3 = t3 t1 - t4 y3 of a
//     point addition costs 588 multiply-accumulates instead of 784, an Fq2 product 2 x 588 with no additions.
//
// What keeps this safe is in the TYPES: F<LO, HI> is a field element whose limbs lie in [-LO 2^28, HI 2^28).  Every
// operation computes the range of its result at compile time, and a product refuses to compile (static_assert) if a
// column of its schoolbook sum could leave the signed 64-bit accumulator:
//     14 * sum_t max(|a_t limb| |b_t limb|) * 2^56 + 14 * 2^56 (the m q part) < 2^63   i.e.  14 * S + 14 <= 126.
// Values: a product returns digits 0..12 in [0, 2^28) and a small signed digit 13, value in (-q, 2q) -- type fe =
// F<0, 1>; that is also the form kept in memory ("L28": 14 int32 per element, no conversion at load or store).  By
// induction a value of type F<LO, HI> lies in (-(2 LO + HI) q, (2 HI + LO) q), so the column bound above also bounds
// |sum a_t b_t| by a few hundred q^2, far inside the 2520 q^2 = R q that the reduction tolerates.  The signed top
// digit (|.| < 2^23 for such values) is not counted in LO / HI: its products are below 2^52 per column, inside the
// two units (2^57) of slack the column bound leaves.
//
// Boundaries: the wavefront VM keeps x 2^384 in 12 x 32-bit words; from_vm / to_vm convert with one product by a
// constant (2^400 resp. 2^384 mod q).  canon() gives the canonical residue where it is observable (bytes, zero tests).
#pragma once
#include <stdint.h>
#include "fp28_mul_gfx950.h"

namespace blsgpu {
namespace r28 {
constexpr int NL = 14;
constexpr int LW = 28;
constexpr int32_t LMASK = 0x0FFFFFFF;
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int LO, int HI> struct F {
    static_assert(LO >= 0 && HI >= 0 && LO <= 8 && HI <= 8, "limb range leaves int32");
    int32_t v[NL];
};
typedef F<0, 1> fe;

// ---- linear operations: limb-wise, ranges add up --------------------------------------------------------------
template <int A, int B, int C, int D> __device__ __forceinline__ F<A + C, B + D> add(const F<A, B>& x, const F<C, D>& y) {
    F<A + C, B + D> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = x.v[j] + y.v[j];
    return r;
}
template <int A, int B, int C, int D> __device__ __forceinline__ F<A + D, B + C> sub(const F<A, B>& x, const F<C, D>& y) {
    F<A + D, B + C> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = x.v[j] - y.v[j];
    return r;
}
template <int A, int B> __device__ __forceinline__ F<B, A> neg(const F<A, B>& x) {
    F<B, A> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = -x.v[j];
    return r;
}
template <int C, int A, int B> __device__ __forceinline__ F<C * A, C * B> mulc(const F<A, B>& x) {
    static_assert(C > 0, "positive constants only");
    F<C * A, C * B> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = x.v[j] * C;
    return r;
}
// carry pass: any limb range -> digits 0..12 in [0, 2^28), the sign moves to digit 13 (3 instructions per limb)
template <int A, int B> __device__ __forceinline__ fe norm(const F<A, B>& x) {
    fe r;
    int32_t c = 0;
#pragma unroll
    for (int j = 0; j < NL - 1; j++) {
        const int32_t t = x.v[j] + c;
        r.v[j] = t & LMASK;
        c = t >> LW;
    }
    r.v[NL - 1] = x.v[NL - 1] + c;
    return r;
}
// C x with the carry pass fused (the product may leave 32 bits: 64-bit multiply-add per limb); C up to 2^20
template <int C, int A, int B> __device__ __forceinline__ fe mulc_norm(const F<A, B>& x) {
    static_assert(C > 0 && C < (1 << 20), "constant too large");
    fe r;
    int64_t c = 0;
#pragma unroll
    for (int j = 0; j < NL - 1; j++) {
        c += (int64_t)x.v[j] * C;
        r.v[j] = (int32_t)((uint32_t)c & (uint32_t)LMASK);
        c >>= LW;
    }
    r.v[NL - 1] = (int32_t)(c + (int64_t)x.v[NL - 1] * C);
    return r;
}
__device__ __forceinline__ fe fe_zero() { fe r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = 0;
    return r; }
__device__ __forceinline__ fe fe_const(const int32_t (&c)[NL]) { fe r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = c[j];
    return r; }
__device__ __forceinline__ fe fe_one() { const int32_t c[NL] = BLS28_ONE; return fe_const(c); }

// ---- products ---------------------------------------------------------------------------------------------------
// share of one term a b in a column, in units of 2^56: positive and negative side
template <int A, int B, int C, int D> struct Term {
    static constexpr int pos = cmax(B * D, A * C), neg = cmax(A * D, B * C);
};
template <int POS, int NEG> struct ColumnsFit { static_assert(14 * POS + 14 <= 126 && 14 * NEG <= 126, "a column of this sum of products may overflow 64 bits: normalise an operand (norm / mulc_norm)"); };

template <int A, int B, int C, int D> __device__ __forceinline__ fe mul(const F<A, B>& x, const F<C, D>& y) {
    typedef Term<A, B, C, D> T0;
    (void)sizeof(ColumnsFit<T0::pos, T0::neg>);
    fe r;
    bls28::fp28_dot1(r.v, x.v, y.v);
    return r;
}
template <int A, int B> __device__ __forceinline__ fe sqr(const F<A, B>& x) {
    typedef Term<A, B, A, B> T0;
    (void)sizeof(ColumnsFit<T0::pos, T0::neg>);
    fe r;
    bls28::fp28_sqr1(r.v, x.v);
    return r;
}
// x0 y0 + x1 y1
template <int A0, int B0, int C0, int D0, int A1, int B1, int C1, int D1>
__device__ __forceinline__ fe dot2(const F<A0, B0>& x0, const F<C0, D0>& y0, const F<A1, B1>& x1, const F<C1, D1>& y1) {
    typedef Term<A0, B0, C0, D0> T0;
    typedef Term<A1, B1, C1, D1> T1;
    (void)sizeof(ColumnsFit<T0::pos + T1::pos, T0::neg + T1::neg>);
    fe r;
    bls28::fp28_dot2(r.v, x0.v, y0.v, x1.v, y1.v);
    return r;
}
template <int A0, int B0, int C0, int D0, int A1, int B1, int C1, int D1, int A2, int B2, int C2, int D2, int A3, int B3, int C3, int D3>
__device__ __forceinline__ fe dot4(const F<A0, B0>& x0, const F<C0, D0>& y0, const F<A1, B1>& x1, const F<C1, D1>& y1,
                                   const F<A2, B2>& x2, const F<C2, D2>& y2, const F<A3, B3>& x3, const F<C3, D3>& y3) {
    typedef Term<A0, B0, C0, D0> T0;
    typedef Term<A1, B1, C1, D1> T1;
    typedef Term<A2, B2, C2, D2> T2;
    typedef Term<A3, B3, C3, D3> T3;
    (void)sizeof(ColumnsFit<T0::pos + T1::pos + T2::pos + T3::pos, T0::neg + T1::neg + T2::neg + T3::neg>);
    fe r;
    bls28::fp28_dot4(r.v, x0.v, y0.v, x1.v, y1.v, x2.v, y2.v, x3.v, y3.v);
    return r;
}

// ---- canonical form, zero test, memory ------------------------------------------------------------------------
// value in (-q, 2q) (what a product returns) -> [0, q), digits canonical
__device__ __forceinline__ fe canon(const fe& x) {
    const int32_t n[NL] = BLS28_Q;
    F<0, 2> t;
    const int32_t neg_mask = x.v[NL - 1] >> 31;
#pragma unroll
    for (int j = 0; j < NL; j++) t.v[j] = x.v[j] + (n[j] & neg_mask);
    const fe a = norm(t);                                   // [0, 2q)
    F<1, 1> u;
#pragma unroll
    for (int j = 0; j < NL; j++) u.v[j] = a.v[j] - n[j];
    const fe b = norm(u);                                   // [-q, q)
    const bool keep = b.v[NL - 1] < 0;
    fe r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = keep ? a.v[j] : b.v[j];
    return r;
}
// x = 0 mod q for a value in (-q, 2q): the digits are those of 0 or of q
__device__ __forceinline__ bool is_zero(const fe& x) {
    const int32_t n[NL] = BLS28_Q;
    int32_t z = 0, e = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) { z |= x.v[j]; e |= x.v[j] ^ n[j]; }
    return z == 0 || e == 0;
}
// L28 memory form: the 14 limbs as they are
__device__ __forceinline__ fe ld(const uint32_t* __restrict__ p) { fe r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = (int32_t)p[j];
    return r; }
__device__ __forceinline__ void st(const fe& x, uint32_t* __restrict__ p) {
#pragma unroll
    for (int j = 0; j < NL; j++) p[j] = (uint32_t)x.v[j];
}
// 12 x 32-bit words of a non-negative integer below 2^384 -> digits
__device__ __forceinline__ fe unpack32(const uint32_t* x) {
    fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = LW * i, w = bit >> 5, s = bit & 31;
        uint32_t lo = x[w] >> s;
        if (s > 32 - LW && w + 1 < 12) lo |= x[w + 1] << (32 - s);
        r.v[i] = (int32_t)(lo & (uint32_t)LMASK);
    }
    return r;
}
// canonical digits (value < 2^384) -> 12 x 32-bit words
__device__ __forceinline__ void pack32(uint32_t* y, const fe& d) {
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 32 * w, i = bit / LW, s = bit - LW * i;
        uint32_t v = (uint32_t)d.v[i] >> s;
        v |= (uint32_t)d.v[i + 1] << (LW - s);
        if (2 * LW - s < 32 && i + 2 < NL) v |= (uint32_t)d.v[i + 2] << (2 * LW - s);
        y[w] = v;
    }
}
// The VM's own Montgomery product  D = A B / 2^384 mod q  (12 x 32-bit words in and out, relaxed: A B < 9 q^2 gives
// D < 2q, as fq_mul_relaxed) computed on 28-bit limbs: A is unpacked SHIFTED LEFT BY 8 BITS, so that the R = 2^392
// reduction divides by 2^384 in effect -- (A 2^8 B + m q) / 2^392 -- and the result is the same residue class and range
// as the 32-bit product's.  56 unpack + 462 + ~30 pack instructions against 623 (v_mad_u64_u32 + v_addc_co_u32 pairs).
__device__ __forceinline__ void vm_mul28(uint32_t* __restrict__ D, const uint32_t* __restrict__ A, const uint32_t* __restrict__ B) {
    int32_t a[NL], b[NL], r[NL];
    a[0] = (int32_t)((A[0] << 8) & (uint32_t)LMASK);
#pragma unroll
    for (int i = 1; i < NL; i++) {
        const int bit = LW * i - 8, w = bit >> 5, sh = bit & 31;
        const uint32_t hi = (w + 1 < 12) ? A[w + 1] : 0u;
        a[i] = (int32_t)(__builtin_amdgcn_alignbit(hi, A[w], sh) & (uint32_t)LMASK);
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = LW * i, w = bit >> 5, sh = bit & 31;
        const uint32_t hi = (w + 1 < 12) ? B[w + 1] : 0u;
        b[i] = (int32_t)(__builtin_amdgcn_alignbit(hi, B[w], sh) & (uint32_t)LMASK);
    }
    bls28::fp28_dot1(r, a, b);
#pragma unroll
    for (int w = 0; w < 12; w++) {
        const int bit = 32 * w, i = bit / LW, sh = bit - LW * i;
        uint32_t v = (uint32_t)r[i] >> sh;
        v |= (uint32_t)r[i + 1] << (LW - sh);
        if (2 * LW - sh < 32 && i + 2 < NL) v |= (uint32_t)r[i + 2] << (2 * LW - sh);
        D[w] = v;
    }
}
// the VM's form (x 2^384 as 12 words, any value below 2^384) <-> x R
__device__ __forceinline__ fe from_vm(const uint32_t* x) { const int32_t c[NL] = BLS28_FROM_VM; return mul(unpack32(x), fe_const(c)); }
__device__ __forceinline__ void to_vm(uint32_t* y, const fe& x) { const int32_t c[NL] = BLS28_TO_VM; pack32(y, canon(mul(x, fe_const(c)))); }
// content (a plain integer below 2^384, e.g. a coordinate read from bytes) <-> x R
__device__ __forceinline__ fe from_raw(const uint32_t* x) { const int32_t c[NL] = BLS28_R2; return mul(unpack32(x), fe_const(c)); }
__device__ __forceinline__ void to_raw(uint32_t* y, const fe& x) {
    fe one = fe_zero();
    one.v[0] = 1;
    pack32(y, canon(mul(x, one)));
}

// ---- Fq2 = Fq[u] / (u^2 + 1) ------------------------------------------------------------------------------------
template <int LO, int HI> struct F2 { F<LO, HI> a, b; };
typedef F2<0, 1> fe2;
struct raw2 { int32_t a[NL], b[NL]; };                     // the untyped form the out-of-line products take
template <int A, int B> __device__ __forceinline__ raw2 to_raw2(const F2<A, B>& x) { raw2 r;
#pragma unroll
    for (int j = 0; j < NL; j++) { r.a[j] = x.a.v[j]; r.b[j] = x.b.v[j]; }
    return r; }
__device__ __forceinline__ fe2 from_raw2(const raw2& x) { fe2 r;
#pragma unroll
    for (int j = 0; j < NL; j++) { r.a.v[j] = x.a[j]; r.b.v[j] = x.b[j]; }
    return r; }
// A point addition on the twist has 11 Fq2 products.  Expanded in place (the default) they are ~110 KB of code per
// addition, more than the instruction cache, but nothing else moves; as real function calls (BLS28_F2_CALLS) the code
// is one copy each, but the ABI passes only 32 argument registers and the other 24 - 80 operand words go through
// scratch memory (measured: profiles/r03_g2_lane_variants.txt).
#if defined(BLS28_F2_CALLS)
#define BLS28_F2_LINKAGE __attribute__((noinline))
#else
#define BLS28_F2_LINKAGE __forceinline__
#endif
__device__ BLS28_F2_LINKAGE raw2 f2_mul_call(raw2 x, raw2 y) {           // x y: two sums of two products
    int32_t nb[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) nb[j] = -x.b[j];
    raw2 r;
    bls28::fp28_dot2(r.a, x.a, y.a, nb, y.b);
    bls28::fp28_dot2(r.b, x.a, y.b, x.b, y.a);
    return r;
}
__device__ BLS28_F2_LINKAGE raw2 f2_dot2_call(raw2 x, raw2 y, raw2 z, raw2 w) {   // x y + z w: two sums of four products
    int32_t nxb[NL], nzb[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) { nxb[j] = -x.b[j]; nzb[j] = -z.b[j]; }
    raw2 r;
    bls28::fp28_dot4(r.a, x.a, y.a, nxb, y.b, z.a, w.a, nzb, w.b);
    bls28::fp28_dot4(r.b, x.a, y.b, x.b, y.a, z.a, w.b, z.b, w.a);
    return r;
}
__device__ BLS28_F2_LINKAGE raw2 f2_sqr_call(raw2 x) {                   // (a + b)(a - b), 2 a b
    int32_t s[NL], d[NL], a2[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) { s[j] = x.a[j] + x.b[j]; d[j] = x.a[j] - x.b[j]; a2[j] = x.a[j] << 1; }
    raw2 r;
    bls28::fp28_dot1(r.a, s, d);
    bls28::fp28_dot1(r.b, a2, x.b);
    return r;
}
template <int A, int B, int C, int D> __device__ __forceinline__ F2<A + C, B + D> add(const F2<A, B>& x, const F2<C, D>& y) { return {add(x.a, y.a), add(x.b, y.b)}; }
template <int A, int B, int C, int D> __device__ __forceinline__ F2<A + D, B + C> sub(const F2<A, B>& x, const F2<C, D>& y) { return {sub(x.a, y.a), sub(x.b, y.b)}; }
template <int A, int B> __device__ __forceinline__ F2<B, A> neg(const F2<A, B>& x) { return {neg(x.a), neg(x.b)}; }
template <int C, int A, int B> __device__ __forceinline__ F2<C * A, C * B> mulc(const F2<A, B>& x) { return {mulc<C>(x.a), mulc<C>(x.b)}; }
template <int A, int B> __device__ __forceinline__ fe2 norm(const F2<A, B>& x) { return {norm(x.a), norm(x.b)}; }
template <int C, int A, int B> __device__ __forceinline__ fe2 mulc_norm(const F2<A, B>& x) { return {mulc_norm<C>(x.a), mulc_norm<C>(x.b)}; }
template <int A, int B> __device__ __forceinline__ F2<cmax(A, B), cmax(A, B)> conj(const F2<A, B>& x) {
    F2<cmax(A, B), cmax(A, B)> r;
#pragma unroll
    for (int j = 0; j < NL; j++) { r.a.v[j] = x.a.v[j]; r.b.v[j] = -x.b.v[j]; }
    return r;
}
// (1 + u) x = (a - b) + (a + b) u
template <int A, int B> __device__ __forceinline__ F2<A + cmax(A, B), B + cmax(A, B)> mul_xi(const F2<A, B>& x) {
    F2<A + cmax(A, B), B + cmax(A, B)> r;
#pragma unroll
    for (int j = 0; j < NL; j++) { r.a.v[j] = x.a.v[j] - x.b.v[j]; r.b.v[j] = x.a.v[j] + x.b.v[j]; }
    return r;
}
template <int A, int B, int C, int D> __device__ __forceinline__ fe2 mul(const F2<A, B>& x, const F2<C, D>& y) {
    typedef Term<A, B, C, D> T0;
    typedef Term<B, A, C, D> T1;                            // the negated imaginary part
    (void)sizeof(ColumnsFit<T0::pos + cmax(T0::pos, T1::pos), T0::neg + cmax(T0::neg, T1::neg)>);
    return from_raw2(f2_mul_call(to_raw2(x), to_raw2(y)));
}
template <int A, int B> __device__ __forceinline__ fe2 sqr(const F2<A, B>& x) {
    typedef Term<A + B, A + B, A + B, A + B> T0;             // (a + b)(a - b) and (2 a) b
    (void)sizeof(ColumnsFit<T0::pos, T0::neg>);
    return from_raw2(f2_sqr_call(to_raw2(x)));
}
template <int A0, int B0, int C0, int D0, int A1, int B1, int C1, int D1>
__device__ __forceinline__ fe2 dot2(const F2<A0, B0>& x0, const F2<C0, D0>& y0, const F2<A1, B1>& x1, const F2<C1, D1>& y1) {
    typedef Term<A0, B0, C0, D0> T0;
    typedef Term<B0, A0, C0, D0> T0n;
    typedef Term<A1, B1, C1, D1> T1;
    typedef Term<B1, A1, C1, D1> T1n;
    (void)sizeof(ColumnsFit<T0::pos + cmax(T0::pos, T0n::pos) + T1::pos + cmax(T1::pos, T1n::pos),
                            T0::neg + cmax(T0::neg, T0n::neg) + T1::neg + cmax(T1::neg, T1n::neg)>);
    return from_raw2(f2_dot2_call(to_raw2(x0), to_raw2(y0), to_raw2(x1), to_raw2(y1)));
}
__device__ __forceinline__ fe2 canon(const fe2& x) { return {canon(x.a), canon(x.b)}; }
__device__ __forceinline__ bool is_zero(const fe2& x) { return is_zero(x.a) && is_zero(x.b); }
__device__ __forceinline__ fe2 ld2(const uint32_t* __restrict__ p) { return {ld(p), ld(p + NL)}; }
__device__ __forceinline__ void st(const fe2& x, uint32_t* __restrict__ p) { st(x.a, p); st(x.b, p + NL); }
__device__ __forceinline__ fe2 fe2_zero() { return {fe_zero(), fe_zero()}; }
__device__ __forceinline__ fe2 fe2_one() { return {fe_one(), fe_zero()}; }

// ---- curve arithmetic, generic over the coordinate field: G1 (fe, b = 4) and the twist (fe2, b' = 4 (1 + u)) ------
// Complete formulas of Renes-Costello-Batina 2015 for a = 0 (the same as vmgen/msm_programs.py): infinity (0 : 1 : 0),
// doubling inside an addition and P + (-P) need no branch.  Coordinates at rest are `fe` / `fe2`.
template <class E> struct Elem;
template <> struct Elem<fe> {
    static constexpr int DW = NL;
    static __device__ __forceinline__ fe zero() { return fe_zero(); }
    static __device__ __forceinline__ fe one() { return fe_one(); }
    static __device__ __forceinline__ fe load(const uint32_t* __restrict__ p) { return ld(p); }
};
template <> struct Elem<fe2> {
    static constexpr int DW = 2 * NL;
    static __device__ __forceinline__ fe2 zero() { return fe2_zero(); }
    static __device__ __forceinline__ fe2 one() { return fe2_one(); }
    static __device__ __forceinline__ fe2 load(const uint32_t* __restrict__ p) { return ld2(p); }
};
// 3 b x, normalised: 12 x on G1, 12 (1 + u) x on the twist
template <int A, int B> __device__ __forceinline__ fe b3(const F<A, B>& x) { return mulc_norm<12>(x); }
template <int A, int B> __device__ __forceinline__ fe2 b3(const F2<A, B>& x) { return mulc_norm<12>(mul_xi(x)); }

// An Fq2 product sums two (four) products per part, so its operands have half the room: tight() normalises an
// Fq2 operand and leaves an Fq operand as it is.  times3: 3 x, normalised.
template <int A, int B> __device__ __forceinline__ F<A, B> tight(const F<A, B>& x) { return x; }
template <int A, int B> __device__ __forceinline__ fe2 tight(const F2<A, B>& x) { return norm(x); }

template <class E> struct ptT { E X, Y, Z; };
template <class E> __device__ __forceinline__ ptT<E> pt_inf() { return {Elem<E>::zero(), Elem<E>::one(), Elem<E>::zero()}; }
template <class E> __device__ __forceinline__ ptT<E> pt_ld(const uint32_t* __restrict__ p) {
    return {Elem<E>::load(p), Elem<E>::load(p + Elem<E>::DW), Elem<E>::load(p + 2 * Elem<E>::DW)};
}
template <class E> __device__ __forceinline__ void pt_st(const ptT<E>& P, uint32_t* __restrict__ p) {
    st(P.X, p); st(P.Y, p + Elem<E>::DW); st(P.Z, p + 2 * Elem<E>::DW);
}
// complete addition, RCB algorithm 7: 6 products + 3 sums of two products
template <class E>
__device__ __forceinline__ ptT<E> padd(const ptT<E>& P, const ptT<E>& Q) {
    const E t0 = mul(P.X, Q.X), t1 = mul(P.Y, Q.Y), t2 = mul(P.Z, Q.Z);
    const auto t3 = sub(sub(mul(add(P.X, P.Y), add(Q.X, Q.Y)), t0), t1);
    const auto t4 = sub(sub(mul(add(P.Y, P.Z), add(Q.Y, Q.Z)), t1), t2);
    const auto t5 = sub(sub(mul(add(P.X, P.Z), add(Q.X, Q.Z)), t0), t2);
    const E x3 = mulc_norm<3>(t0);
    const E bz = b3(t2);
    const auto z3 = tight(add(t1, bz));
    const auto t1m = sub(t1, bz);
    const E y3 = b3(t5);
    ptT<E> R;
    R.X = dot2(t3, t1m, neg(t4), y3);
    R.Y = dot2(t1m, z3, y3, x3);
    R.Z = dot2(z3, tight(t4), x3, tight(t3));
    return R;
}
// complete mixed addition, RCB algorithm 8: P += (x2 : y2 : 1) for an affine (x2, y2) and any P: 5 products + 3 sums
template <class E>
__device__ __forceinline__ void pmadd(ptT<E>& P, const E& x2, const E& y2) {
    const E t0 = mul(P.X, x2), t1 = mul(P.Y, y2);
    const auto t3 = sub(sub(mul(add(x2, y2), add(P.X, P.Y)), t0), t1);
    const auto t4 = add(mul(y2, P.Z), P.Y);
    const E y3 = b3(add(mul(x2, P.Z), P.X));
    const E x3 = mulc_norm<3>(t0);
    const E bz = b3(P.Z);
    const auto z3 = tight(add(t1, bz));
    const auto t1m = sub(t1, bz);
    P.X = dot2(t3, t1m, neg(t4), y3);
    P.Y = dot2(y3, x3, t1m, z3);
    P.Z = dot2(z3, t4, x3, t3);
}
// complete doubling, RCB algorithm 9
template <class E>
__device__ __forceinline__ ptT<E> pdbl(const ptT<E>& P) {
    const E t0 = sqr(P.Y), t1 = mul(P.Y, P.Z), t2 = b3(sqr(P.Z)), txy = mul(P.X, P.Y);
    const E z8 = mulc_norm<8>(t0);
    const auto d = tight(sub(t0, mulc<3>(t2)));
    ptT<E> R;
    R.X = mul(d, add(txy, txy));
    R.Y = dot2(t2, z8, d, add(t0, t2));
    R.Z = mul(t1, z8);
    return R;
}
// out-of-line forms for kernels that add or double at several places (a twist addition is ~140 KB of code): one copy
// per code object; the two points travel through the ABI's 32 argument registers and scratch, < 1 % of the addition
template <class E> __device__ __attribute__((noinline)) ptT<E> padd_fn(ptT<E> P, ptT<E> Q) { return padd(P, Q); }
template <class E> __device__ __attribute__((noinline)) ptT<E> pdbl_fn(ptT<E> P) { return pdbl(P); }
template <class E> __device__ __forceinline__ ptT<E> pneg(const ptT<E>& P) { return {P.X, norm(neg(P.Y)), P.Z}; }
}  // namespace r28
}  // namespace blsgpu
