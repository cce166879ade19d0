// blsgpu_reg.hip -- field and curve arithmetic with ONE ITEM PER LANE, everything in
// registers (included by blsgpu_api.hip before the kernels that use it).
//
// The wavefront VM packs the few dozen independent field operations of ONE item into 64
// lanes; where a batch offers thousands of independent items and an item's state fits the
// register file, a lane per item executes nothing but products and modular additions: no
// tables, no scratchpad, every lane busy.  Used for cofactor clearing of large hash batches
// (blsgpu_h2c.hip) and for the bucket sums (blsgpu_msm.hip).  Same complete formulas as
// vmgen/msm_programs.py (Renes-Costello-Batina 2015, a = 0).  The field product is a real
// function call (s_swappc, operands by value in VGPRs): a point addition has 12 (G1) or
// 36 (G2) of them, inlined they would not fit the instruction cache.
#pragma once

namespace blsgpu {
namespace reg {
struct fe { uint32_t v[12]; };                              // Montgomery, canonical (< q)
struct fe2 { fe a, b; };

__device__ __attribute__((noinline)) fe fe_mul(fe x, fe y) {
    fe r;
    bls::fq_mul(r.v, x.v, y.v);
    return r;
}
// the same product expanded in place: for a loop whose body is ONE G1 addition (11 products = 64 KB of code, still
// cached) the call's operand moves are worth 11 % (k_srt_accum); G2 additions (36 products) are faster with the call
__device__ __forceinline__ fe fe_mul_inl(const fe& x, const fe& y) {
    fe r;
    bls::fq_mul(r.v, x.v, y.v);
    return r;
}
// modular addition / subtraction of canonical values as three carry chains of 12 (sum, trial subtraction, select)
__device__ __forceinline__ fe fe_add(const fe& x, const fe& y) {
    const uint32_t q[12] = BLS_Q_LIMBS;
    uint32_t t[12], d[12], c = 0, br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) t[j] = bls::addc(x.v[j], y.v[j], c);
#pragma unroll
    for (int j = 0; j < 12; j++) d[j] = bls::subc(t[j], q[j], br);
    const bool ge = (c != 0) || (br == 0);
    fe r;
#pragma unroll
    for (int j = 0; j < 12; j++) r.v[j] = ge ? d[j] : t[j];
    return r;
}
__device__ __forceinline__ fe fe_sub(const fe& x, const fe& y) {
    const uint32_t q[12] = BLS_Q_LIMBS;
    uint32_t d[12], br = 0, c = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) d[j] = bls::subc(x.v[j], y.v[j], br);
    const uint32_t mask = 0u - br;
    fe r;
#pragma unroll
    for (int j = 0; j < 12; j++) r.v[j] = bls::addc(d[j], q[j] & mask, c);
    return r;
}
__device__ __forceinline__ fe fe_zero() { fe r; for (int j = 0; j < 12; j++) r.v[j] = 0; return r; }
__device__ __forceinline__ fe fe_neg(const fe& x) { return fe_sub(fe_zero(), x); }
__device__ __forceinline__ fe2 f2_add(const fe2& x, const fe2& y) { return {fe_add(x.a, y.a), fe_add(x.b, y.b)}; }
__device__ __forceinline__ fe2 f2_sub(const fe2& x, const fe2& y) { return {fe_sub(x.a, y.a), fe_sub(x.b, y.b)}; }
__device__ __forceinline__ fe2 f2_neg(const fe2& x) { return {fe_neg(x.a), fe_neg(x.b)}; }
__device__ __forceinline__ fe2 f2_conj(const fe2& x) { return {x.a, fe_neg(x.b)}; }
__device__ __forceinline__ fe2 f2_mul_xi(const fe2& x) { return {fe_sub(x.a, x.b), fe_add(x.a, x.b)}; }   // (1 + u) x
__device__ fe2 f2_mul(const fe2& x, const fe2& y) {
    fe t0 = fe_mul(x.a, y.a), t1 = fe_mul(x.b, y.b), t2 = fe_mul(fe_add(x.a, x.b), fe_add(y.a, y.b));
    return {fe_sub(t0, t1), fe_sub(fe_sub(t2, t0), t1)};
}
__device__ fe2 f2_sqr(const fe2& x) {
    fe m = fe_mul(x.a, x.b);
    return {fe_mul(fe_add(x.a, x.b), fe_sub(x.a, x.b)), fe_add(m, m)};
}
// element operations by overload: E = fe (G1 coordinates) or fe2 (G2 coordinates)
__device__ __forceinline__ fe eadd(const fe& x, const fe& y) { return fe_add(x, y); }
__device__ __forceinline__ fe2 eadd(const fe2& x, const fe2& y) { return f2_add(x, y); }
__device__ __forceinline__ fe esub(const fe& x, const fe& y) { return fe_sub(x, y); }
__device__ __forceinline__ fe2 esub(const fe2& x, const fe2& y) { return f2_sub(x, y); }
__device__ __forceinline__ fe eneg(const fe& x) { return fe_neg(x); }
__device__ __forceinline__ fe2 eneg(const fe2& x) { return f2_neg(x); }
__device__ __forceinline__ fe emul(const fe& x, const fe& y) { return fe_mul(x, y); }
__device__ __forceinline__ fe2 emul(const fe2& x, const fe2& y) { return f2_mul(x, y); }
__device__ __forceinline__ fe esqr(const fe& x) { return fe_mul(x, x); }
__device__ __forceinline__ fe2 esqr(const fe2& x) { return f2_sqr(x); }
__device__ __forceinline__ fe exi(const fe& x) { return x; }                   // b = 4 on G1
__device__ __forceinline__ fe2 exi(const fe2& x) { return f2_mul_xi(x); }      // b' = 4 (1 + u) on the twist
template <class E> __device__ __forceinline__ E edbl(const E& x) { return eadd(x, x); }
template <class E> __device__ __forceinline__ E ex3(const E& x) { return eadd(edbl(x), x); }
template <class E> __device__ __forceinline__ E ex8(const E& x) { return edbl(edbl(edbl(x))); }
template <class E> __device__ __forceinline__ E eb3(const E& x) {               // 3 b x
    E t = edbl(edbl(exi(x)));
    return eadd(edbl(t), t);
}
template <class E> struct ptT { E X, Y, Z; };
template <bool INL> __device__ __forceinline__ fe emul_sel(const fe& x, const fe& y) {
    if constexpr (INL) return fe_mul_inl(x, y);
    else return fe_mul(x, y);
}
template <bool INL> __device__ __forceinline__ fe2 emul_sel(const fe2& x, const fe2& y) { return f2_mul(x, y); }
// complete addition, RCB algorithm 7 (a = 0) -- msm_programs.padd.  INL: G1 products expanded in place (fe_mul_inl)
template <class E, bool INL = false>
__device__ ptT<E> padd(const ptT<E>& P, const ptT<E>& Q) {
    auto emul = [](const E& a, const E& b) { return emul_sel<INL>(a, b); };
    E t0 = emul(P.X, Q.X), t1 = emul(P.Y, Q.Y), t2 = emul(P.Z, Q.Z);
    E t3 = esub(esub(emul(eadd(P.X, P.Y), eadd(Q.X, Q.Y)), t0), t1);
    E t4 = esub(esub(emul(eadd(P.Y, P.Z), eadd(Q.Y, Q.Z)), t1), t2);
    E t5 = esub(esub(emul(eadd(P.X, P.Z), eadd(Q.X, Q.Z)), t0), t2);
    E x3 = ex3(t0), bz = eb3(t2);
    E z3 = eadd(t1, bz), t1m = esub(t1, bz), y3 = eb3(t5);
    ptT<E> R;
    R.X = esub(emul(t3, t1m), emul(t4, y3));
    R.Y = eadd(emul(t1m, z3), emul(y3, x3));
    R.Z = eadd(emul(z3, t4), emul(x3, t3));
    return R;
}
// complete doubling, RCB algorithm 9 (a = 0) -- msm_programs.pdbl
template <class E>
__device__ ptT<E> pdbl(const ptT<E>& P) {
    E t0 = esqr(P.Y), t1 = emul(P.Y, P.Z), t2 = eb3(esqr(P.Z)), txy = emul(P.X, P.Y);
    E z8 = ex8(t0), d = esub(t0, ex3(t2));
    ptT<E> R;
    R.X = edbl(emul(d, txy));
    R.Y = eadd(emul(t2, z8), emul(d, eadd(t0, t2)));
    R.Z = emul(t1, z8);
    return R;
}
// complete mixed addition, RCB algorithm 8 (a = 0): P += (x2 : y2 : 1); (x2, y2) is an affine point, P any
// point including (0 : 1 : 0).  11 products; the operands die early, so a G1 addition fits 256 registers.
template <class E, bool INL = false>
__device__ __forceinline__ void pmadd(ptT<E>& P, const E& x2, const E& y2) {
    auto emul = [](const E& a, const E& b) { return emul_sel<INL>(a, b); };
    E t0 = emul(P.X, x2), t1 = emul(P.Y, y2);
    E t3 = esub(esub(emul(eadd(x2, y2), eadd(P.X, P.Y)), t0), t1);
    E t4 = eadd(emul(y2, P.Z), P.Y);
    E y3 = eadd(emul(x2, P.Z), P.X);
    t0 = ex3(t0);
    const E t2 = eb3(P.Z);
    const E z3 = eadd(t1, t2);
    t1 = esub(t1, t2);
    y3 = eb3(y3);
    P.X = esub(emul(t3, t1), emul(t4, y3));
    P.Y = eadd(emul(y3, t0), emul(t1, z3));
    P.Z = eadd(emul(z3, t4), emul(t0, t3));
}
template <class E> __device__ __forceinline__ ptT<E> pneg(const ptT<E>& P) { return {P.X, eneg(P.Y), P.Z}; }
typedef ptT<fe2> pt;                                         // a point of the twist
__device__ pt mul_x(const pt& P) {                           // [|x|] P, |x| = 0xd201000000010000
    pt A = P;
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        A = pdbl(A);
        if ((0xd201000000010000ull >> bit) & 1ull) A = padd(A, P);
    }
    return A;
}
}  // namespace reg
}  // namespace blsgpu
