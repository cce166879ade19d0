// blsgpu_ml.hip -- the LINE-STREAM multi-pairing (round 3): the Miller loops of a large batch as two data-parallel
// register kernels joined through HBM instead of one accumulator per wavefront (included by blsgpu_api.hip).
//
// Replaces the loop of fq_ate_pairing_multi / fq_miller_loop (fields_t.py:1091-1121; lines :1035-1078) for batches
// that fill the chip.  vmgen/linestream_model.py is the integer model of exactly this data flow (CPU-tested against
// the oracle and the reference's vectors):
//
//   k_ml_lines2  one PAIR PER LANE PAIR, 28-bit-limb register arithmetic (fp28.h) with every Fq2 value split over two
//                adjacent lanes (namespace sp).  The twist-point chain T <- 2T (+ Q) does not depend on the Miller
//                accumulator, so it runs alone; every step stores its line l = l0 + l2 w^2 + l3 w^3 (three Fq2 values,
//                P already multiplied in) as an 84-dword record: lines[(L * n + pair) * 84], L = 0 .. 67 in execution
//                order (63 tangents, 5 chords).  22.8 KB per pair: the 8 TB/s of HBM3E are what makes cutting the
//                loop here affordable.  (k_ml_lines4: the same on lane quads for calls that leave SIMDs empty on lane pairs.)
//   k_ml_lines_exact  the pairs the fast formulas are not valid for: the reference's own line values into the same records.
//   k_ml_accum   SIX LANES PER ACCUMULATOR (ten accumulators per wavefront): lane k holds the coefficient f_k of
//                f = sum f_k w^k (Fq12 = Fq2[w]/(w^6 - xi)).  A team multiplies the line L of the pairs of its chunk
//                into its accumulator: c_k = f_k l0 + F_{k-2} l2 + F_{k-3} l3 with F_i = f_i (i >= 0), xi f_{i+6}
//                (i < 0) fetched from the team's lanes with ds_bpermute; the three Fq2 terms are summed lazily and taken
//                Karatsuba-wise (round 5: three sums of three products, fp28_dot3, instead of two sums of six).  No
//                squarings, no dependency between line indices.
//   k_ml_small   groups of a few pairs: one team per group runs the whole loop f <- f^2 prod l on the same lines.
//   k_ml_merge   dense products of the chunks' partial products (same lane layout).
//   k_ml_horner_fexp (blsgpu_fexpw.hip)  f <- f^2 (before a tangent) ; f <- f M_L over the 68 per-line products of a group with
//                a product per lane, then the final exponentiation in place or the wavefront VM's form of the partial.  (The
//                forms of rounds 3 - 4 that lost at every size -- one pair per lane, ten groups per Horner wavefront, a dense
//                product over 36 lanes -- were removed in round 5.)
//
// The value differs from the reference's Miller product by the line scalings (Fq2 factors and w^3 per line) that
// the final exponentiation removes iff they are non-zero: a pair whose Q is off the twist, flagged,
// or whose chain ends with Z = 0 is marked in bad[] and listed; k_ml_lines_exact rewrites its line records with the
// reference's own line values (affine formulas, 0^-1 := 0, the chord step's branches), which k_ml_accum multiplies in
// like any other line -- their product is the reference's Miller value of that pair itself.
#pragma once
#include "mlx_consts_gfx950.h"

namespace blsgpu {
namespace ml {
using r28::fe;
using r28::fe2;
using r28::NL;

constexpr int LINES = 68;                  // 63 tangent steps + 5 chord steps of |x| = 0xd201000000010000
constexpr int LINE_DW = 6 * NL;            // l0, l2, l3 (re, im each)
constexpr int DENSE_DW = 12 * NL;          // f_0 .. f_5 (re, im each), w-power order
constexpr int TEAMS = 10;                  // accumulators per wavefront (lanes 60 .. 63 idle)
constexpr uint64_t ML_NX = 0xd201000000010000ull;

// line index -> 1 iff it is a tangent (a new step of the loop: the accumulator is squared first)
__device__ __forceinline__ bool line_is_tangent(uint32_t L) {
    // chords follow the tangents of bits 62, 60, 57, 48, 16: line indices 1, 4, 8, 18, 51
    return !(L == 1u || L == 4u || L == 8u || L == 18u || L == 51u);
}

// ---- stage A ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ fe load_coord(const uint32_t* __restrict__ src) {       // 48 big-endian bytes -> x R
    uint32_t c[12];
#pragma unroll
    for (int w = 0; w < 12; w++) c[11 - w] = bswap32(src[w]);
    return r28::from_raw(c);
}

// ---- stage A on LANE PAIRS: an Fq2 value split over two adjacent lanes ---------------------------------------------
// The even lane holds the real part, the odd lane the imaginary part; the partner's part is one DPP move per limb away
// (quad_perm 1,0,3,2).  Half the registers and half the code per lane (two wavefronts per SIMD, the step stays in the
// instruction cache) for about 12 % more instructions than one pair per lane.
namespace sp {
template <int M> struct S { int32_t v[NL]; };          // the lane's part; limbs in (-M 2^28, M 2^28)
typedef S<1> h;
__device__ __forceinline__ bool odd() { return (threadIdx.x & 1u) != 0; }
template <int M> __device__ __forceinline__ S<M> swp(const S<M>& x) {
    S<M> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = __builtin_amdgcn_update_dpp(0, x.v[j], 0xB1, 0xF, 0xF, true);
    return r;
}
template <int A, int B> __device__ __forceinline__ S<A + B> add(const S<A>& x, const S<B>& y) { S<A + B> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = x.v[j] + y.v[j];
    return r; }
template <int A, int B> __device__ __forceinline__ S<A + B> sub(const S<A>& x, const S<B>& y) { S<A + B> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = x.v[j] - y.v[j];
    return r; }
template <int A> __device__ __forceinline__ S<A> neg(const S<A>& x) { S<A> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = -x.v[j];
    return r; }
template <int C, int A> __device__ __forceinline__ S<C * A> mulc(const S<A>& x) { S<C * A> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = x.v[j] * C;
    return r; }
template <int A> __device__ __forceinline__ h norm(const S<A>& x) {
    static_assert(A <= 8, "limb range leaves int32");
    h r;
    int32_t c = 0;
#pragma unroll
    for (int j = 0; j < NL - 1; j++) { const int32_t t = x.v[j] + c; r.v[j] = t & r28::LMASK; c = t >> r28::LW; }
    r.v[NL - 1] = x.v[NL - 1] + c;
    return r;
}
template <int C, int A> __device__ __forceinline__ h mulc_norm(const S<A>& x) {
    static_assert(C > 0 && C < (1 << 20) && A <= 8, "constant too large");
    h r;
    int64_t c = 0;
#pragma unroll
    for (int j = 0; j < NL - 1; j++) { c += (int64_t)x.v[j] * C; r.v[j] = (int32_t)((uint32_t)c & (uint32_t)r28::LMASK); c >>= r28::LW; }
    r.v[NL - 1] = (int32_t)(c + (int64_t)x.v[NL - 1] * C);
    return r;
}
// (1 + u) x: re = a - b, im = a + b
template <int A> __device__ __forceinline__ S<2 * A> mul_xi(const S<A>& x) {
    const S<A> p = swp(x);
    S<2 * A> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = odd() ? x.v[j] + p.v[j] : x.v[j] - p.v[j];
    return r;
}
template <int A> __device__ __forceinline__ h b3(const S<A>& x) { return mulc_norm<12>(mul_xi(x)); }      // 12 (1 + u) x
// operands of a product x y: re = x.re y.re - x.im y.im, im = x.re y.im + x.im y.re.  With o = own part, p = partner's:
// even lane  o_x o_y + (-p_x) p_y,  odd lane  p_x o_y + o_x p_y  -- both lanes  a o_y + b p_y.
template <int M> struct Lop { int32_t a[NL], b[NL]; };
template <int M> struct Rop { int32_t o[NL], p[NL]; };
template <int M> __device__ __forceinline__ Lop<M> left(const S<M>& x) {
    const S<M> p = swp(x);
    Lop<M> r;
#pragma unroll
    for (int j = 0; j < NL; j++) { r.a[j] = odd() ? p.v[j] : x.v[j]; r.b[j] = odd() ? x.v[j] : -p.v[j]; }
    return r;
}
template <int M> __device__ __forceinline__ Rop<M> right(const S<M>& x) {
    const S<M> p = swp(x);
    Rop<M> r;
#pragma unroll
    for (int j = 0; j < NL; j++) { r.o[j] = x.v[j]; r.p[j] = p.v[j]; }
    return r;
}
// column bound of fp28.h: 14 * (sum of |a||b| in units of 2^56) + 14 <= 126
template <int MA, int MB> __device__ __forceinline__ h mul(const Lop<MA>& x, const Rop<MB>& y) {
    static_assert(2 * MA * MB <= 8, "a column of this product may overflow 64 bits");
    h r;
    bls28::fp28_dot2(r.v, x.a, y.o, x.b, y.p);
    return r;
}
template <int MA, int MB, int MC, int MD>
__device__ __forceinline__ h dot2(const Lop<MA>& x0, const Rop<MB>& y0, const Lop<MC>& x1, const Rop<MD>& y1) {
    static_assert(2 * MA * MB + 2 * MC * MD <= 8, "a column of this sum of products may overflow 64 bits");
    h r;
    bls28::fp28_dot4(r.v, x0.a, y0.o, x0.b, y0.p, x1.a, y1.o, x1.b, y1.p);
    return r;
}
// x^2: re = (a + b)(a - b), im = (2 b) a
__device__ __forceinline__ h sqr(const h& x) {
    const h p = swp(x);
    int32_t u[NL], w[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) { u[j] = x.v[j] + (odd() ? x.v[j] : p.v[j]); w[j] = odd() ? p.v[j] : x.v[j] - p.v[j]; }
    h r;
    bls28::fp28_dot1(r.v, u, w);                                       // |u|, |w| < 2^29: 4 units
    return r;
}
// g^2 - 12 e^2 (round 5): the lane's part of a square is ONE product -- (a + b)(a - b) on the even lane, (2 b) a on the odd one --,
// so the difference of the two squares is a sum of two products with one reduction (588 multiply-adds; two squares, a
// scaling and a carry pass before: 784 + 84).  Column bound: |u|, |w| < 2^29 for g (4 units), 12 u normalised x |w| < 2^29 for e (2).
__device__ __forceinline__ h sqr_m12sqr(const h& g, const h& e) {
    const h pg = swp(g), pe = swp(e);
    int32_t ug[NL], wg[NL], nwe[NL];
    S<2> ue;
#pragma unroll
    for (int j = 0; j < NL; j++) {
        ug[j] = g.v[j] + (odd() ? g.v[j] : pg.v[j]); wg[j] = odd() ? pg.v[j] : g.v[j] - pg.v[j];
        ue.v[j] = e.v[j] + (odd() ? e.v[j] : pe.v[j]); nwe[j] = odd() ? -pe.v[j] : pe.v[j] - e.v[j];
    }
    const h ue12 = mulc_norm<12>(ue);
    h r;
    bls28::fp28_dot2(r.v, ug, wg, ue12.v, nwe);
    return r;
}
template <int M> __device__ __forceinline__ h mulf(const S<M>& x, const fe& k) {     // by an Fq value (both lanes hold it)
    static_assert(M <= 8, "");
    h r;
    bls28::fp28_dot1(r.v, x.v, k.v);
    return r;
}
__device__ __forceinline__ h load_part(const uint32_t* __restrict__ src) {       // the lane's 48 bytes -> x R
    const fe t = load_coord(src);
    h r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = t.v[j];
    return r;
}
// zero mod q in BOTH lanes of the pair (value in (-q, 2q), normalised digits)
__device__ __forceinline__ bool is_zero2(const h& x) {
    fe t;
#pragma unroll
    for (int j = 0; j < NL; j++) t.v[j] = x.v[j];
    const int z = r28::is_zero(t) ? 1 : 0;
    return (z & __builtin_amdgcn_update_dpp(0, z, 0xB1, 0xF, 0xF, true)) != 0;
}
template <int M> __device__ __forceinline__ void store_part(int32_t* __restrict__ rec, int coef, const S<M>& x) {
    int32_t* o = rec + (coef * 2 + (odd() ? 1 : 0)) * NL;
#pragma unroll
    for (int j = 0; j < NL; j++) o[j] = x.v[j];
}

__device__ __forceinline__ void tangent_step(h& X, h& Y, h& Z, const fe& px3n, const fe& py, int32_t* __restrict__ rec) {
    // (the statements are asm volatile underneath, so this order IS the execution order: every value is consumed as
    // early as the data flow allows, which keeps the step inside 256 registers)
    // Round 4: the two mixed products as squares -- 2YZ = (Y + Z)^2 - Y^2 - Z^2 and 2XY = (X + Y)^2 - X^2 - Y^2 with the
    // squares the step computes anyway: a complex squaring is ONE product per lane (392 multiply-adds), a product a sum
    // of two (588); the sums are normalised first (a square of 2^29-limbs would not fit the columns).  Same field
    // elements, same line records.
    const h XX = sqr(X);
    store_part(rec, 1, mulf(XX, px3n));                               // X^2 (-3 px)
    const h C = sqr(Z);
    const h B = sqr(Y);
    const S<3> H = sub(sub(sqr(norm(add(Y, Z))), B), C);              // 2YZ
    store_part(rec, 2, mulf(H, py));                                  // 2YZ py
    const h z8 = mulc_norm<4>(H);                                     // 8YZ
    const h E = b3(C);
    const S<3> A2 = sub(sub(sqr(norm(add(X, Y))), XX), B);            // 2XY
    store_part(rec, 0, sub(B, E));
    const S<3> F3 = mulc<3>(E);
    const h BmF = norm(sub(B, F3)), G = norm(add(B, F3));
    X = mul(left(A2), right(BmF));
    Y = sqr_m12sqr(G, E);                                             // G^2 - 12 E^2: one sum of two products (round 4: two squares, 2 x 392 + the scaling)
    Z = mul(left(B), right(z8));
}
__device__ __forceinline__ void chord_step(h& X, h& Y, h& Z, const h& xq, const h& yq, const fe& px3n, const fe& py3,
                                           int32_t* __restrict__ rec) {
    const Rop<1> rz = right(Z);
    const h th = norm(sub(Y, mul(left(yq), rz))), la = norm(sub(X, mul(left(xq), rz)));
    const Lop<1> lth = left(th), lla = left(la);
    const h C = sqr(th), D = sqr(la);
    const Rop<1> rd = right(D);
    const h E = mul(lla, rd), Fz = mul(left(Z), right(C)), Gg = mul(left(X), rd);
    const h H = norm(sub(add(E, Fz), add(Gg, Gg)));
    const h GH = norm(sub(Gg, H));
    const h xq3 = mulc_norm<3>(xq), nyq3 = mulc_norm<3>(neg(yq));
    store_part(rec, 0, dot2(lth, right(xq3), lla, right(nyq3)));
    store_part(rec, 1, mulf(th, px3n));
    store_part(rec, 2, mulf(la, py3));
    const h nE = norm(neg(E));
    const h Y3 = dot2(lth, right(GH), left(nE), right(Y));
    X = mul(lla, right(H));
    Z = mul(left(Z), right(E));
    Y = Y3;
}
}  // namespace sp

// ---- lane QUADS: both lane pairs of a quad hold the same Fq2 values and share a level of independent products ------------
// (round 4; also the G2 point operations of blsgpu_msm.hip sp4).  Both pairs execute the same instructions on operands
// picked by the pair index and swap results with DPP quad_perm [2,3,0,1].
namespace sq {
using namespace sp;
__device__ __forceinline__ bool hi() { return (threadIdx.x & 2u) != 0; }               // the second pair of the quad
template <int M> __device__ __forceinline__ S<M> oth(const S<M>& x) {                    // the other pair's value
    S<M> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = __builtin_amdgcn_update_dpp(0, x.v[j], 0x4E, 0xF, 0xF, true);
    return r;
}
template <int M> __device__ __forceinline__ S<M> pick(const S<M>& a, const S<M>& b) {    // pair 0: a, pair 1: b
    S<M> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = hi() ? b.v[j] : a.v[j];
    return r;
}
template <int W, int M> __device__ __forceinline__ S<W> widen(const S<M>& x) {           // the same limbs under a looser bound
    static_assert(M <= W, "");
    S<W> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = x.v[j];
    return r;
}
// The tangent step of sp::tangent_step dealt out over the two pairs: level 1 the five squares (X^2, Y^2, (X + Y)^2 | Z^2,
// (Y + Z)^2), level 2 {2XY (B - F), G^2, X^2 (-3 px) | B 8YZ, E^2, 2YZ py}: 3 squares + product + square + product by an Fq
// value deep (2548 multiply-adds per lane) instead of 4718.  Pair 0 stores the line's coefficients 0 and 1, pair 1 the third.
__device__ __forceinline__ void tangent_step(h& X, h& Y, h& Z, const fe& px3n, const fe& py, int32_t* __restrict__ rec) {
    const h sa = sqr(pick(X, Z)), sb = sqr(pick(Y, norm(add(Y, Z)))), sc = sqr(norm(add(X, Y)));
    const h oa = oth(sa), ob = oth(sb);
    const h XX = pick(sa, oa), C = pick(oa, sa), B = pick(sb, ob), T1 = pick(ob, sb);
    const h& T2 = sc;                                                 // (both pairs computed it)
    const S<3> H = sub(sub(T1, B), C);                                // 2YZ
    const h z8 = mulc_norm<4>(H);
    const h E = b3(C);
    const S<3> A2 = sub(sub(T2, XX), B);                              // 2XY
    if (!hi()) store_part(rec, 0, sub(B, E));
    const S<3> F3 = mulc<3>(E);
    const h BmF = norm(sub(B, F3)), G = norm(add(B, F3));
    fe kf;
#pragma unroll
    for (int j = 0; j < NL; j++) kf.v[j] = hi() ? py.v[j] : px3n.v[j];
    const h lc = mulf(pick(widen<3>(XX), H), kf);                      // X^2 (-3 px) | 2YZ py
    {
        int32_t* o = rec + ((hi() ? 2 : 1) * 2 + (odd() ? 1 : 0)) * NL;
#pragma unroll
        for (int j = 0; j < NL; j++) o[j] = lc.v[j];
    }
    const h m = mul(left(pick(A2, widen<3>(B))), right(pick(BmF, z8)));   // X' | Z'
    const h q = sqr(pick(G, E));                                       // G^2 | E^2
    const h om = oth(m), oq = oth(q);
    X = pick(m, om);
    Z = pick(om, m);
    Y = norm(sub(pick(q, oq), mulc_norm<12>(pick(oq, q))));            // G^2 - 12 E^2
}
}  // namespace sq

// Two lanes per pair (lane 2p: real parts, lane 2p + 1: imaginary parts).  Same lines, flags and work list as k_ml_lines.
#ifndef BLSGPU_ML_LINES2_WAVES
#define BLSGPU_ML_LINES2_WAVES 2
#endif
__global__ void __launch_bounds__(256, BLSGPU_ML_LINES2_WAVES) k_ml_lines2(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2, uint32_t n,
                                                                           int32_t* __restrict__ lines, uint8_t* __restrict__ bad, DegenList dg)
#if BLSGPU_EMIT(BLSGPU_TU_ML)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pr = t >> 1;
    const uint32_t p = pr < n ? pr : n - 1u;                      // the last wavefront's spare lanes repeat the last pair
    const uint32_t part = t & 1u;
    const uint32_t* s1 = g1 + (size_t)p * 24;
    const uint32_t* s2 = g2 + (size_t)p * 48 + part * 12;
    fe px3n, pyv;
    sp::h X, Y, Z;
    bool ok = !q_flagged(dg, p);
    {
        const fe px = load_coord(s1), py = load_coord(s1 + 12);
        px3n = r28::mulc_norm<3>(r28::neg(px));
        pyv = py;
        X = sp::load_part(s2);
        Y = sp::load_part(s2 + 24);
        const int32_t one[NL] = BLS28_ONE;
#pragma unroll
        for (int j = 0; j < NL; j++) Z.v[j] = part ? 0 : one[j];
        // Q on the twist: y^2 - x^3 - 4 (1 + u) = 0
        const sp::h yy = sp::sqr(Y), xx = sp::sqr(X);
        const sp::h xxx = sp::mul(sp::left(xx), sp::right(X));
        sp::S<4> four;
#pragma unroll
        for (int j = 0; j < NL; j++) four.v[j] = 4 * one[j];
        const auto d = sp::sub(sp::sub(yy, xxx), four);              // S<6>
        ok = ok && sp::is_zero2(sp::mulf(d, r28::fe_one()));
    }
    int32_t* rec = lines + (size_t)p * LINE_DW;
    const size_t lstride = (size_t)n * LINE_DW;
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        sp::tangent_step(X, Y, Z, px3n, pyv, rec);
        rec += lstride;
        if ((ML_NX >> bit) & 1ull) {
            const sp::h xq = sp::load_part(s2), yq = sp::load_part(s2 + 24);
            const fe py3 = r28::mulc_norm<3>(load_coord(s1 + 12));
            sp::chord_step(X, Y, Z, xq, yq, px3n, py3, rec);
            rec += lstride;
        }
    }
    ok = ok && !sp::is_zero2(Z);
    if (part == 0 && pr < n) {
        bad[p] = ok ? 0 : 1;
        if (!ok) {
            const uint32_t at = atomicAdd(dg.count, 1u);
            dg.blocks[at] = p;
        }
    }
}
#else
;
#endif

// Four lanes per pair (round 4): for calls that leave SIMDs empty on lane pairs -- 8192 pairs are 256 wavefronts there, each
// with the whole chain of 68 steps to itself (1.12 ms whatever the count) -- the two pairs of a quad share the tangent
// step's levels (sq::tangent_step): about 0.6 of the depth.  Same records, flags and work list.
__global__ void __launch_bounds__(256, BLSGPU_ML_LINES2_WAVES) k_ml_lines4(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2, uint32_t n,
                                                                           int32_t* __restrict__ lines, uint8_t* __restrict__ bad, DegenList dg)
#if BLSGPU_EMIT(BLSGPU_TU_ML)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pr = t >> 2;
    const uint32_t p = pr < n ? pr : n - 1u;                      // the last wavefront's spare lanes repeat the last pair
    const uint32_t part = t & 1u;
    const uint32_t* s1 = g1 + (size_t)p * 24;
    const uint32_t* s2 = g2 + (size_t)p * 48 + part * 12;
    fe px3n, pyv;
    sp::h X, Y, Z;
    bool ok = !q_flagged(dg, p);
    {
        const fe px = load_coord(s1), py = load_coord(s1 + 12);
        px3n = r28::mulc_norm<3>(r28::neg(px));
        pyv = py;
        X = sp::load_part(s2);
        Y = sp::load_part(s2 + 24);
        const int32_t one[NL] = BLS28_ONE;
#pragma unroll
        for (int j = 0; j < NL; j++) Z.v[j] = part ? 0 : one[j];
        // Q on the twist: y^2 - x^3 - 4 (1 + u) = 0
        const sp::h yy = sp::sqr(Y), xx = sp::sqr(X);
        const sp::h xxx = sp::mul(sp::left(xx), sp::right(X));
        sp::S<4> four;
#pragma unroll
        for (int j = 0; j < NL; j++) four.v[j] = 4 * one[j];
        const auto d = sp::sub(sp::sub(yy, xxx), four);              // S<6>
        ok = ok && sp::is_zero2(sp::mulf(d, r28::fe_one()));
    }
    int32_t* rec = lines + (size_t)p * LINE_DW;
    const size_t lstride = (size_t)n * LINE_DW;
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        sq::tangent_step(X, Y, Z, px3n, pyv, rec);
        rec += lstride;
        if ((ML_NX >> bit) & 1ull) {
            const sp::h xq = sp::load_part(s2), yq = sp::load_part(s2 + 24);
            const fe py3 = r28::mulc_norm<3>(load_coord(s1 + 12));
            sp::chord_step(X, Y, Z, xq, yq, px3n, py3, rec);      // (five of 68 steps: both pairs run it whole and store the same record)
            rec += lstride;
        }
    }
    ok = ok && !sp::is_zero2(Z);
    if ((t & 3u) == 0u && pr < n) {
        bad[p] = ok ? 0 : 1;
        if (!ok) {
            const uint32_t at = atomicAdd(dg.count, 1u);
            dg.blocks[at] = p;
        }
    }
}
#else
;
#endif

// ---- the reference's own lines for the listed pairs ---------------------------------------------------------------
// fq_miller_loop is a total function of the coordinates (fields_t.py:1035-1078, 641-686): affine formulas with
// 0^-1 := 0 and three branches in the chord step; vmgen/slow_programs.py spells them out and
// vmgen/linestream_model.exact_pair_lines is this kernel's integer model (it reproduces the reference's Miller value
// of every pair of tests/golden/pairing_degenerate.json).  The line values are sparse in the w-power basis as well --
// tangent  py + c3 w^3 + c5 w^5,  chord the same or, on the "vertical" branch,  px + c4 w^4  -- so they are written
// into the pair's ordinary line records ([c0 | zeros | cA | cB]) and k_ml_accum multiplies them in with the positions
// (3, 5) / (4, 5) instead of (2, 3): the group's product is then the reference's Miller value of these pairs times
// the scaled lines of the others.  One pair per lane pair; every inversion is the safegcd routine of fq32.h run by
// all lanes at once, so a batch made of nothing but such pairs runs at a third of the ordinary rate, not a fiftieth.
namespace sp {
__device__ __attribute__((noinline)) fe fq_inv_lane(fe n) {           // 1/n, 0 -> 0 (fields_t.py:47-55)
    uint32_t w[12], v[12];
    r28::to_vm(w, n);
    bls::fq_inv(v, w);
    return r28::from_vm(v);
}
template <int M> __device__ __forceinline__ bool zero2(const S<M>& x) { return is_zero2(mulf(x, r28::fe_one())); }
// 1/z for z = re + im u: conj(z) / (re^2 + im^2); z with digits below 2^29
template <int M> __device__ __forceinline__ h inv2(const S<M>& z) {
    static_assert(2 * M * M <= 8, "");
    const S<M> p = swp(z);
    fe n;
    bls28::fp28_dot2(n.v, z.v, z.v, p.v, p.v);
    const fe ni = fq_inv_lane(n);
    S<M> zc;
#pragma unroll
    for (int j = 0; j < NL; j++) zc.v[j] = odd() ? -z.v[j] : z.v[j];
    return mulf(zc, ni);
}
// (a0 + a1, a1 - a0): xi^-1 a = this / 2
template <int M> __device__ __forceinline__ S<2 * M> xisum(const S<M>& a) {
    const S<M> p = swp(a);
    S<2 * M> r;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = odd() ? a.v[j] - p.v[j] : a.v[j] + p.v[j];
    return r;
}
struct Aff { h x, y; };
// fq2_double_point (fields_t.py:641-646) with the slope; i2y = 1 / (2 ry)
__device__ __forceinline__ Aff exact_double(const h& rx, const h& ry, const h& i2y, h& lam) {
    lam = mul(left(mulc_norm<3>(sqr(rx))), right(i2y));
    // (the differences go through a product by one: x' = lam^2 - 2x is a linear recurrence, digit normalisation alone
    // would let the VALUE double with every step of the chain)
    Aff r;
    r.x = mulf(sub(sqr(lam), add(rx, rx)), r28::fe_one());
    r.y = mulf(sub(mul(left(lam), right(norm(sub(rx, r.x)))), ry), r28::fe_one());
    return r;
}
}  // namespace sp

// Block mode (per_block > 0; the wavefront-VM kernels' work list): an entry is a BLOCK of up to per_block consecutive
// pairs of one group (block b: group b / bpg, pairs from (b - group * bpg) * per_block on); item v = entry * per_block + j
// is a "virtual pair" whose records go to lines[(L * n + v) * 84] and whose flag to bad[v] (255: past the end of the
// block's group).  per_block == 0: an entry is a pair and the records are that pair's own.
__global__ void __launch_bounds__(64) k_ml_lines_exact(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2, uint32_t n,
                                                       int32_t* __restrict__ lines, uint8_t* __restrict__ bad, DegenList dg,
                                                       uint32_t gsz, uint32_t bpg, uint32_t per_block)
#if BLSGPU_EMIT(BLSGPU_TU_ML)
{
    using namespace sp;
    const uint32_t entries = __builtin_amdgcn_readfirstlane(*(volatile const uint32_t*)dg.count);
    const uint32_t total = per_block ? entries * per_block : entries;
    if (blockIdx.x * 32u >= total) return;
    const uint32_t part = threadIdx.x & 1u;
    const int32_t halfc[NL] = BLS28_HALF;
    fe half;
#pragma unroll
    for (int j = 0; j < NL; j++) half.v[j] = halfc[j];
#pragma unroll 1
    for (uint32_t e0 = blockIdx.x * 32u; e0 < total; e0 += gridDim.x * 32u) {
        const uint32_t er = e0 + (threadIdx.x >> 1);
        const uint32_t e = er < total ? er : total - 1u;              // spare lane pairs repeat the last item
        uint32_t p = dg.blocks[per_block ? e / per_block : e], v = p;
        if (per_block) {
            const uint32_t b = p, grp = b / bpg, in_grp = (b - grp * bpg) * per_block + (e % per_block);
            v = e;
            if (in_grp >= gsz) {                                      // past the end of the group: nothing to multiply
                if (part == 0u && er < total) bad[v] = 255;
                continue;
            }
            p = grp * gsz + in_grp;
        }
        const uint32_t* s1 = g1 + (size_t)p * 24;
        const uint32_t* s2 = g2 + (size_t)p * 48 + part * 12;
        const bool qinf = dg.inf != nullptr && (dg.inf[2 * (size_t)p + 1] & 1u) != 0;   // (bit 1: listed without being flagged -- blsgpu_miller_loop_batch's pairs with py = 0)
        const fe px = load_coord(s1), py = load_coord(s1 + 12);
        const fe hnpx = r28::mul(r28::neg(px), half);                 // -px / 2
        const h qx = load_part(s2), qy = load_part(s2 + 24);
        h rx = qx, ry = qy;
        int32_t* rec = lines + (size_t)v * LINE_DW;
        const size_t lstride = (size_t)n * LINE_DW;
        uint32_t flag = 1u, chord = 0u;
        h c0;
#pragma unroll 1
        for (int bit = 62; bit >= 0; bit--) {
            {   // tangent: line py - lam px / w - (ry - lam rx) / w^3  (fq2_double_line_eval, fields_t.py:1035-1049)
                h lam;
                const Aff d = exact_double(rx, ry, inv2(add(ry, ry)), lam);
                const h l0 = norm(sub(mul(left(lam), right(rx)), ry));
#pragma unroll
                for (int j = 0; j < NL; j++) c0.v[j] = part ? 0 : py.v[j];
                store_part(rec, 0, c0);
                store_part(rec, 1, mulf(xisum(l0), half));            // w^3:  xi^-1 (lam rx - ry)
                store_part(rec, 2, mulf(xisum(lam), hnpx));           // w^5:  xi^-1 (-lam px)
                rx = d.x; ry = d.y;
                rec += lstride;
            }
            if ((ML_NX >> bit) & 1ull) {
                // chord: fq2_add_line_eval (fields_t.py:1052-1078) and fq2_add_points (:673-686)
                const h dd = norm(sub(qx, rx)), u = norm(sub(ry, qy));
                const bool n1 = !zero2(dd), same = !n1 && zero2(u);
                const bool vert = zero2(add(rx, qx)) && zero2(add(ry, qy));
                const h D = inv2(dd);
                const h mu = mul(left(norm(neg(u))), right(D));
                const h tt = dot2(left(qy), right(rx), left(norm(neg(ry))), right(qx));
                const h nu = mul(left(norm(neg(tt))), right(D));
                const h av = mulf(xisum(neg(rx)), half);              // vertical: px - rx / w^2 -> w^4: xi^-1 (-rx)
                const h an = mulf(xisum(neg(nu)), half);              // else w^3: xi^-1 (-nu)
                const h bn = mulf(xisum(mu), hnpx);                   //      w^5: xi^-1 (-mu px)
                h a, b;
#pragma unroll
                for (int j = 0; j < NL; j++) {
                    c0.v[j] = part ? 0 : (vert ? px.v[j] : py.v[j]);
                    a.v[j] = vert ? av.v[j] : an.v[j];
                    b.v[j] = vert ? 0 : bn.v[j];
                }
                store_part(rec, 0, c0);
                store_part(rec, 1, a);
                store_part(rec, 2, b);
                if (vert) flag |= 2u << chord;
                chord++;
                rec += lstride;
                // R + Q: the chord point unless rx = qx; then the double if also ry = qy, else (0, 0); a flagged Q leaves R
                const h xc = mulf(sub(sub(sqr(mu), rx), qx), r28::fe_one());
                const h yc = mulf(sub(mul(left(mu), right(norm(sub(rx, xc)))), ry), r28::fe_one());
                h lam;
                const Aff d = exact_double(rx, ry, inv2(add(ry, ry)), lam);
                if (!qinf) {
#pragma unroll
                    for (int j = 0; j < NL; j++) {
                        rx.v[j] = n1 ? xc.v[j] : (same ? d.x.v[j] : 0);
                        ry.v[j] = n1 ? yc.v[j] : (same ? d.y.v[j] : 0);
                    }
                }
            }
        }
        if (part == 0u && er < total) bad[v] = (uint8_t)flag;
    }
}
#else
;
#endif

// ---- blsgpu_miller_loop_batch on the lane kernels (round 5): every pair listed, one accumulator per pair ------------------------------
// The work list of ALL n pairs (count, then 0 .. n - 1): k_ml_lines_exact then writes every pair's line records with the reference's
// own formulas and k_ml_small (groups of one) multiplies them up -- the reference's fq_miller_loop value itself.
__global__ void __launch_bounds__(256) k_ml_list_all(uint32_t n, uint32_t* __restrict__ count, uint32_t* __restrict__ blocks)
#if BLSGPU_EMIT(BLSGPU_TU_ML)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0u) *count = n;
    if (i < n) blocks[i] = i;
}
#else
;
#endif
// flags for blsgpu_miller_loop_batch's fast form: the caller's (P, Q) flags with bit 1 of Q's set where py = 0 mod q -- the lane
// kernels send any non-zero flag through the reference's own lines (k_ml_exact_fixup divides by py)
__global__ void __launch_bounds__(256) k_ml_exact_flags(const uint32_t* __restrict__ g1, const uint8_t* __restrict__ inf, uint32_t n, uint8_t* __restrict__ flags)
#if BLSGPU_EMIT(BLSGPU_TU_ML)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const fe py = load_coord(g1 + (size_t)p * 24 + 12);
    flags[2 * (size_t)p] = inf ? inf[2 * (size_t)p] : 0;
    flags[2 * (size_t)p + 1] = (uint8_t)((inf ? (inf[2 * (size_t)p + 1] & 1u) : 0u) | (r28::is_zero(py) ? 2u : 0u));
}
#else
;
#endif
// partials in the wavefront VM's form (12 x 12 words x 2^384 per Fq12, the reference's flat order) -> n x 576 canonical big-endian bytes
__global__ void __launch_bounds__(256) k_ml_partials_to_bytes(const uint32_t* __restrict__ partials, uint32_t nvalues, uint32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_ML)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvalues) return;
    uint32_t x[12], y[12];
#pragma unroll
    for (int j = 0; j < 12; j++) x[j] = partials[(size_t)i * 12 + j];
    r28::to_raw(y, r28::from_vm(x));
#pragma unroll
    for (int j = 0; j < 12; j++) out[(size_t)i * 12 + j] = bswap32(y[11 - j]);
}
#else
;
#endif

// blsgpu_miller_loop_batch from the FAST lines (round 5): one Fq2 factor per pair instead of 73 inversions.  The fast lines are the
// exact ones times an Fq2 factor and w^3 --  l_fast = c w^3 l_exact  with  c = 2YZ (tangent), 3 (X - xq Z) (chord), and the third
// coefficient of l_fast is  c py  -- so with S = sum over the lines of 2^(squarings after the line) (odd) and
// L3 = prod (third coefficient)^(2^(squarings after)):   f_exact = w^-3 D f_fast,   D = py^S / (L3 xi^((S - 1) / 2))
// (vmgen/gen_mlx.py: derivation, constants, the identity on the integer model -- tests/test_mlx_model.py).  One pair per lane pair:
// reads the pair's 68 line records (third coefficients), its partial (the wavefront VM's form, flat order) and py; writes the
// reference's 576 bytes.  A pair whose flag byte is set went through the reference's own lines: its partial IS the value.
__global__ void __launch_bounds__(256, 2) k_ml_exact_fixup(const uint32_t* __restrict__ g1, const int32_t* __restrict__ lines, const uint8_t* __restrict__ bad,
                                                          const uint32_t* __restrict__ partials, uint32_t n, uint32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_ML)
{
    using namespace sp;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pr = t >> 1, part = t & 1u;
    const uint32_t p = pr < n ? pr : n - 1u;                          // the last wavefront's spare lane pairs repeat the last pair
    h f[6];                                                           // flat Fq2 index i = power (0, 2, 4, 1, 3, 5)[i] of w: this lane's parts
#pragma unroll
    for (int i = 0; i < 6; i++) {
        uint32_t x[12];
#pragma unroll
        for (int j = 0; j < 12; j++) x[j] = partials[(size_t)p * 144 + (2 * i + part) * 12 + j];
        const fe v = r28::from_vm(x);
#pragma unroll
        for (int j = 0; j < NL; j++) f[i].v[j] = v.v[j];
    }
    if (bad[p] == 0) {
        const size_t lstride = (size_t)n * LINE_DW;
        const int32_t* rec = lines + (size_t)p * LINE_DW + (2 * 2 + part) * NL;
        h L3;
#pragma unroll
        for (int j = 0; j < NL; j++) L3.v[j] = rec[j];
#pragma unroll 1
        for (uint32_t L = 1; L < (uint32_t)LINES; L++) {
            rec += lstride;
            h l;
#pragma unroll
            for (int j = 0; j < NL; j++) l.v[j] = rec[j];
            const bool chord = L < 64u && ((MLX_CHORD_MASK_LO >> L) & 1ull) != 0ull;        // (every chord is among the first 64 lines)
            if (!chord) L3 = sqr(L3);
            L3 = mul(left(L3), right(l));
        }
        const fe py = load_coord(g1 + (size_t)p * 24 + 12);
        fe ps = py;                                                   // py^S, from the top bit of S down
#pragma unroll 1
        for (int b = 62; b >= 0; b--) {
            ps = r28::sqr(ps);
            if ((MLX_S >> b) & 1ull) ps = r28::mul(ps, py);
        }
        const int32_t kr[NL] = MLX_KAPPA_RE, ki[NL] = MLX_KAPPA_IM, halfc[NL] = BLS28_HALF;
        h kap;
        fe half;
#pragma unroll
        for (int j = 0; j < NL; j++) { kap.v[j] = part ? ki[j] : kr[j]; half.v[j] = halfc[j]; }
        const h D = mulf(inv2(mul(left(L3), right(kap))), ps);
        const Rop<1> rD = right(D);
        // power k of w: k < 3 takes D f[k + 3], k >= 3 takes xi^-1 D f[k - 3]; flat index of power k: (0, 3, 1, 4, 2, 5)[k]
        const h g0 = mul(left(f[4]), rD), g1v = mul(left(f[2]), rD), g2v = mul(left(f[5]), rD);                 // powers 0, 1, 2 <- 3, 4, 5
        const h g3 = mulf(xisum(mul(left(f[0]), rD)), half), g4 = mulf(xisum(mul(left(f[3]), rD)), half),      // powers 3, 4, 5 <- xi^-1 x 0, 1, 2
                g5 = mulf(xisum(mul(left(f[1]), rD)), half);
        f[0] = g0; f[3] = g1v; f[1] = g2v; f[4] = g3; f[2] = g4; f[5] = g5;
    }
    if (pr < n) {
#pragma unroll 1
        for (int i = 0; i < 6; i++) {
            fe v;
#pragma unroll
            for (int j = 0; j < NL; j++) v.v[j] = f[i].v[j];
            uint32_t y[12];
            r28::to_raw(y, v);
#pragma unroll
            for (int j = 0; j < 12; j++) out[(size_t)p * 144 + (2 * i + part) * 12 + j] = bswap32(y[11 - j]);
        }
    }
}
#else
;
#endif

// ---- stages B / merge / Horner: six lanes per accumulator -----------------------------------------------------------
struct Team {
    uint32_t c;            // the lane's coefficient: power of w
    uint32_t slot;         // team within the wavefront (10 = the idle lanes)
    uint32_t base4;        // byte address (lane * 4) of the team's first lane, for ds_bpermute
};
__device__ __forceinline__ Team team_of_lane() {
    const uint32_t lane = threadIdx.x & 63u;
    Team t;
    t.slot = lane / 6u;
    t.c = lane - t.slot * 6u;
    t.base4 = (lane - t.c) * 4u;
    return t;
}
__device__ __forceinline__ void bperm14(int32_t* __restrict__ d, const int32_t* __restrict__ s, uint32_t addr) {
#pragma unroll
    for (int j = 0; j < NL; j++) d[j] = __builtin_amdgcn_ds_bpermute((int)addr, s[j]);
}
// What a lane offers its team for one product: f_k and xi f_k, with the negated imaginary parts (a sum of products has
// no subtraction).  NORM: the xi forms with normalised digits (dense products sum three wrapped terms per reduction).
struct Pub { int32_t re[NL], im[NL], nim[NL], xre[NL], xim[NL], nxim[NL]; };
template <bool NORM>
__device__ __forceinline__ void publish(Pub& P, const int32_t* __restrict__ fre, const int32_t* __restrict__ fim) {
    if (NORM) {
        r28::F<1, 1> d; r28::F<0, 2> s;
#pragma unroll
        for (int j = 0; j < NL; j++) { d.v[j] = fre[j] - fim[j]; s.v[j] = fre[j] + fim[j]; }
        const fe dn = r28::norm(d), sn = r28::norm(s);
#pragma unroll
        for (int j = 0; j < NL; j++) { P.xre[j] = dn.v[j]; P.xim[j] = sn.v[j]; P.nxim[j] = -sn.v[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < NL; j++) { P.xre[j] = fre[j] - fim[j]; P.xim[j] = fre[j] + fim[j]; P.nxim[j] = -(fre[j] + fim[j]); }
    }
#pragma unroll
    for (int j = 0; j < NL; j++) { P.re[j] = fre[j]; P.im[j] = fim[j]; P.nim[j] = -fim[j]; }
}
// the operand F_{c - J} of the lane: from lane (c - J) mod 6 of the team, the xi form iff J > c
struct Xop { int32_t re[NL], im[NL], nim[NL]; };
template <int J>
__device__ __forceinline__ void fetch(Xop& X, const Pub& P, const Team& t) {
    if (J == 0) {
#pragma unroll
        for (int j = 0; j < NL; j++) { X.re[j] = P.re[j]; X.im[j] = P.im[j]; X.nim[j] = P.nim[j]; }
        return;
    }
    const uint32_t src = t.c >= (uint32_t)J ? t.c - J : t.c + 6u - J;
    const uint32_t addr = t.base4 + src * 4u;
    // the SOURCE chooses the form (round 4): lane s is read by lane (s + J) mod 6 alone, which wants the xi form iff it wraps,
    // i.e. iff s + J >= 6 -- one select per limb here and ONE fetch instead of two fetches and a select at the reader
    const bool offer_xi = t.c + (uint32_t)J >= 6u;
    int32_t o[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) o[j] = offer_xi ? P.xre[j] : P.re[j];
    bperm14(X.re, o, addr);
#pragma unroll
    for (int j = 0; j < NL; j++) o[j] = offer_xi ? P.xim[j] : P.im[j];
    bperm14(X.im, o, addr);
#pragma unroll
    for (int j = 0; j < NL; j++) o[j] = offer_xi ? P.nxim[j] : P.nim[j];
    bperm14(X.nim, o, addr);
}
// (re, im) = sum over three terms of F_{c - J} y_J; y given as (re, im) arrays.  Column bound of fp28_dot6 (units of
// 2^56): an unwrapped term 1 + 1, a wrapped one 1 + 2 (|xi f| limbs below 2^29 when not normalised) -- at most
// 2 + 3 + 3 = 8 for the line positions (0, 2, 3), 3 x 2 = 6 with normalised xi forms: inside the 8 that 64 bits hold.
template <int J0, int J1, int J2>
__device__ __forceinline__ void mul3(int32_t* __restrict__ re, int32_t* __restrict__ im, const Pub& P, const Team& t,
                                     const int32_t* y0r, const int32_t* y0i, const int32_t* y1r, const int32_t* y1i,
                                     const int32_t* y2r, const int32_t* y2i) {
    Xop X0, X1, X2;
    fetch<J0>(X0, P, t);
    fetch<J1>(X1, P, t);
    fetch<J2>(X2, P, t);
    bls28::fp28_dot6(re, X0.re, y0r, X0.nim, y0i, X1.re, y1r, X1.nim, y1i, X2.re, y2r, X2.nim, y2i);
    bls28::fp28_dot6(im, X0.re, y0i, X0.im, y0r, X1.re, y1i, X1.im, y1r, X2.re, y2i, X2.im, y2r);
}
// A LINE is f_0-term + w^jA + w^jB with (jA, jB) = (2, 3) for the scaled lines of k_ml_lines2 and (3, 5) / (4, 5) for the
// reference's own line values (k_ml_lines_exact: coefficient 0 in Fq, its imaginary part stored as zeros); the sparse product
// takes the positions at run time (mul_line_k3 below).
// positions of line L of a pair whose flag byte is `flag` (0: scaled lines; else bit 0 set and bit 1 + c = chord c took the
// reference's "vertical" branch)
__device__ __forceinline__ void line_positions(uint32_t L, uint32_t flag, uint32_t& jA, uint32_t& jB) {
    const int ci = L == 1u ? 0 : (L == 4u ? 1 : (L == 8u ? 2 : (L == 18u ? 3 : (L == 51u ? 4 : -1))));
    const bool vert = ci >= 0 && ((flag >> (1 + ci)) & 1u) != 0u;
    jA = flag ? (vert ? 4u : 3u) : 2u;
    jB = flag ? 5u : 3u;
}
// f <- f g for a dense g given as 12 x 14 dwords (w-power order) behind `ld` (global or LDS): two half sums, added
// and normalised (value in (-2q, 4q), digits below 2^28: a valid operand)
template <class LD>
__device__ __forceinline__ void mul_dense(int32_t* __restrict__ fre, int32_t* __restrict__ fim, const Team& t, const LD& ld) {
    Pub P;
    publish<true>(P, fre, fim);
    int32_t y[6][NL], r0[NL], i0[NL], r1[NL], i1[NL];
#pragma unroll
    for (int k = 0; k < 6; k++) ld(y[k], k);
    mul3<0, 1, 2>(r0, i0, P, t, y[0], y[1], y[2], y[3], y[4], y[5]);
#pragma unroll
    for (int k = 0; k < 6; k++) ld(y[k], 6 + k);
    mul3<3, 4, 5>(r1, i1, P, t, y[0], y[1], y[2], y[3], y[4], y[5]);
    r28::F<0, 2> sr, si;
#pragma unroll
    for (int j = 0; j < NL; j++) { sr.v[j] = r0[j] + r1[j]; si.v[j] = i0[j] + i1[j]; }
    const fe nr = r28::norm(sr), ni = r28::norm(si);
#pragma unroll
    for (int j = 0; j < NL; j++) { fre[j] = nr.v[j]; fim[j] = ni.v[j]; }
}
struct GlobalRec {
    const int32_t* __restrict__ p;
    __device__ __forceinline__ void operator()(int32_t* __restrict__ d, int k) const {
#pragma unroll
        for (int j = 0; j < NL; j++) d[j] = p[k * NL + j];
    }
};
struct LdsRec {
    const int32_t* p;
    __device__ __forceinline__ void operator()(int32_t* __restrict__ d, int k) const {
#pragma unroll
        for (int j = 0; j < NL; j++) d[j] = p[k * NL + j];
    }
};
__device__ __forceinline__ void set_one(int32_t* __restrict__ fre, int32_t* __restrict__ fim, const Team& t) {
    const int32_t one[NL] = BLS28_ONE;
#pragma unroll
    for (int j = 0; j < NL; j++) { fre[j] = t.c == 0u ? one[j] : 0; fim[j] = 0; }
}

// ---- the sparse product f <- f l: Karatsuba over the lazily summed terms (round 5) -------------------------------------------
// c = sum_j F_j y_j (three Fq2 terms) as  P1 = sum Xre yr,  P2 = sum Xim yi,  P3 = sum (Xre + Xim)(yr + yi):  re = P1 - P2,
// im = P3 - P1 - P2 -- nine Fq products and THREE reductions (3 x fp28_dot3 = 2352 multiply-adds) against the twelve and two
// (2 x fp28_dot6 = 2744) of rounds 3 - 4; the unreduced columns of P1 and P2 are never needed (reducing each sum by itself costs
// one more reduction, 196, but no 64-bit column bookkeeping, and fewer operands are live than in fp28_dot6: 229 registers
// against 238).  What it costs: the sums of the operand parts, two carry passes on the line's (yr + yi) for the column bound
// of P3, and a carry pass on each part of the result (differences of reduced values).  Measured on the bench's step
// (profiles/r05_accum_variants_ab.txt): k_ml_accum 22.3 -> 20.2 ms per 524 800 pairs -- the ratio of the two instruction
// streams (3805 -> 3450), which DESIGN.md had argued away on a register estimate for two rounds.
// the line record of the constant 1 (coefficient 0 = R mod q, every other part zero)
static __device__ const int32_t __attribute__((aligned(16))) ML_LINE_ONE[LINE_DW] = BLS28_ONE;
struct Pub2 { int32_t re[NL], im[NL], xre[NL], xim[NL]; };
struct Xop2 { int32_t re[NL], im[NL]; };
__device__ __forceinline__ void fetch2_rt(Xop2& X, const Pub2& P, const Team& t, uint32_t J) {
    const uint32_t src = t.c >= J ? t.c - J : t.c + 6u - J;
    const uint32_t addr = t.base4 + src * 4u;
    const bool offer_xi = t.c + J >= 6u;                  // (the source chooses the form: see fetch<J>)
    int32_t o[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) o[j] = offer_xi ? P.xre[j] : P.re[j];
    bperm14(X.re, o, addr);
#pragma unroll
    for (int j = 0; j < NL; j++) o[j] = offer_xi ? P.xim[j] : P.im[j];
    bperm14(X.im, o, addr);
}
// fre, fim: normalised digits (value in (-2q, 3q)); y: the line record's six parts
__device__ __forceinline__ void mul_line_k3(int32_t* __restrict__ fre, int32_t* __restrict__ fim, const Team& t, uint32_t jA, uint32_t jB,
                                            const int32_t (&y)[6][NL]) {
    Pub2 P;
#pragma unroll
    for (int j = 0; j < NL; j++) { P.re[j] = fre[j]; P.im[j] = fim[j]; P.xre[j] = fre[j] - fim[j]; P.xim[j] = fre[j] + fim[j]; }
    Xop2 X1, X2;
    fetch2_rt(X1, P, t, jA);
    fetch2_rt(X2, P, t, jB);
    int32_t p1[NL], p2[NL], p3[NL];
    // column bounds (units of 2^56; fp28_dot3 holds 8): P1 1 + 1 + 1, P2 1 + 2 + 2 (|xim| limbs below 2^29), P3 (2 x 2) + (2 x 1) + (2 x 1)
    bls28::fp28_dot3(p1, P.re, y[0], X1.re, y[2], X2.re, y[4]);
    bls28::fp28_dot3(p2, P.im, y[1], X1.im, y[3], X2.im, y[5]);
    int32_t s0[NL], s1[NL], s2[NL], ys0[NL];
    r28::F<0, 2> t1, t2;
#pragma unroll
    for (int j = 0; j < NL; j++) {
        s0[j] = P.re[j] + P.im[j]; s1[j] = X1.re[j] + X1.im[j]; s2[j] = X2.re[j] + X2.im[j];
        ys0[j] = y[0][j] + y[1][j]; t1.v[j] = y[2][j] + y[3][j]; t2.v[j] = y[4][j] + y[5][j];
    }
    const fe ys1 = r28::norm(t1), ys2 = r28::norm(t2);
    bls28::fp28_dot3(p3, s0, ys0, s1, ys1.v, s2, ys2.v);
    r28::F<1, 1> dr; r28::F<2, 1> di;
#pragma unroll
    for (int j = 0; j < NL; j++) { dr.v[j] = p1[j] - p2[j]; di.v[j] = p3[j] - p1[j] - p2[j]; }
    const fe nr = r28::norm(dr), ni = r28::norm(di);
#pragma unroll
    for (int j = 0; j < NL; j++) { fre[j] = nr.v[j]; fim[j] = ni.v[j]; }
}
// Team (group g, chunk j, line L) = index ((g * cpg + j) * 68 + L): the product of line L over the pairs
// [j * chunk, min(gsz, (j + 1) * chunk)) of group g that are not marked bad -> out[index * 168].
#ifndef BLSGPU_ML_ACCUM_WAVES
#define BLSGPU_ML_ACCUM_WAVES 2
#endif
__global__ void __launch_bounds__(256, BLSGPU_ML_ACCUM_WAVES) k_ml_accum(const int32_t* __restrict__ lines, const uint8_t* __restrict__ bad, uint32_t n,
                                                                            uint32_t gsz, uint32_t chunk, uint32_t cpg, uint32_t nteams,
                                                                            int32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_ML)
{
#if BLSGPU_ML_ACCUM_WAVES == 1
    asm volatile("" ::: "a63");           // measurement build: an allocation past 256 registers (229 + 64 accumulation registers) holds the SIMD at one wavefront
#endif
    const Team t = team_of_lane();
    const uint32_t id = wave_index() * TEAMS + t.slot;
    const bool valid = t.slot < (uint32_t)TEAMS && id < nteams;
    const uint32_t idc = valid ? id : 0u;
    const uint32_t L = idc % LINES, gj = idc / LINES, j = gj % cpg, g = gj / cpg;
    const uint32_t lo = j * chunk, hi = min(gsz, lo + chunk);
    const uint32_t cnt = valid ? hi - lo : 0u;
    const size_t first = (size_t)g * gsz + lo;
    int32_t fre[NL], fim[NL];
    set_one(fre, fim, t);
    const int32_t* base = lines + ((size_t)L * n + first) * LINE_DW;
#pragma unroll 1
    for (uint32_t i = 0; __any(i < cnt); i++) {
        const bool use = i < cnt;
        const uint32_t ii = use ? i : 0u;
        uint32_t jA, jB;
        line_positions(L, bad[first + ii], jA, jB);
        // a team past its last pair multiplies by the line "1" (same residues, no 28 selects on the way out)
        const int4* rec = reinterpret_cast<const int4*>(use ? base + (size_t)ii * LINE_DW : ML_LINE_ONE);
        int32_t y[6][NL];
        {
            int4 q[LINE_DW / 4];
#pragma unroll
            for (int k = 0; k < LINE_DW / 4; k++) q[k] = rec[k];
#pragma unroll
            for (int k = 0; k < LINE_DW / 4; k++) {
                const int e = 4 * k;
                y[e / NL][e % NL] = q[k].x; y[(e + 1) / NL][(e + 1) % NL] = q[k].y;
                y[(e + 2) / NL][(e + 2) % NL] = q[k].z; y[(e + 3) / NL][(e + 3) % NL] = q[k].w;
            }
        }
        mul_line_k3(fre, fim, t, jA, jB, y);
    }
    if (valid) {
        int32_t* o = out + (size_t)id * DENSE_DW + t.c * 2 * NL;
#pragma unroll
        for (int k = 0; k < NL; k++) { o[k] = fre[k]; o[NL + k] = fim[k]; }
    }
}
#else
;
#endif

// One level of the product tree over the chunks: team (g, j', L) multiplies the records (g, j, L), j in
// [j' * fan, min(cpg_in, (j' + 1) * fan)), of `in` (indexed as k_ml_accum's output with cpg_in) -> out (cpg_out).
__global__ void __launch_bounds__(256, 2) k_ml_merge(const int32_t* __restrict__ in, uint32_t cpg_in, uint32_t fan, uint32_t cpg_out,
                                                    uint32_t nteams, int32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_ML)
{
    const Team t = team_of_lane();
    const uint32_t id = wave_index() * TEAMS + t.slot;
    const bool valid = t.slot < (uint32_t)TEAMS && id < nteams;
    const uint32_t idc = valid ? id : 0u;
    const uint32_t L = idc % LINES, gj = idc / LINES, jo = gj % cpg_out, g = gj / cpg_out;
    const uint32_t lo = jo * fan, hi = min(cpg_in, lo + fan);
    const uint32_t cnt = valid ? hi - lo : 0u;
    int32_t fre[NL], fim[NL];
    {
        const int32_t* r = in + ((size_t)(g * cpg_in + lo) * LINES + L) * DENSE_DW + t.c * 2 * NL;
#pragma unroll
        for (int k = 0; k < NL; k++) { fre[k] = r[k]; fim[k] = r[NL + k]; }
    }
#pragma unroll 1
    for (uint32_t i = 1; __any(i < cnt); i++) {
        const bool act = i < cnt;
        const uint32_t ii = act ? i : 0u;
        int32_t nre[NL], nim[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) { nre[k] = fre[k]; nim[k] = fim[k]; }
        mul_dense(nre, nim, t, GlobalRec{in + ((size_t)(g * cpg_in + lo + ii) * LINES + L) * DENSE_DW});
#pragma unroll
        for (int k = 0; k < NL; k++) { fre[k] = act ? nre[k] : fre[k]; fim[k] = act ? nim[k] : fim[k]; }
    }
    if (valid) {
        int32_t* o = out + (size_t)id * DENSE_DW + t.c * 2 * NL;
#pragma unroll
        for (int k = 0; k < NL; k++) { o[k] = fre[k]; o[NL + k] = fim[k]; }
    }
}
#else
;
#endif

// Small groups (a batch of verifications of a few pairs each: threshold verifies, single signatures): the classic loop
// f <- f^2 prod_i l_{i,L} with ONE GROUP PER TEAM of six lanes, lines from k_ml_lines2 -- the per-line products of a
// group of two are not worth an accumulator each, and the group count already fills the chip.
struct TeamRec {                                   // the team's own f as the dense operand (squaring)
    const int32_t* re; const int32_t* im; uint32_t base4;
    __device__ __forceinline__ void operator()(int32_t* __restrict__ d, int k) const {
        bperm14(d, (k & 1) ? im : re, base4 + (uint32_t)(k >> 1) * 4u);
    }
};
// List mode (count != nullptr; the wavefront-VM kernels' degenerate blocks): group g is entry g of the work list, its
// pairs the virtual pairs [g * gsz, (g + 1) * gsz) of k_ml_lines_exact's block mode (flag 255: no pair), and its partial
// goes to partials[index[g] * pstride].
__global__ void __launch_bounds__(256, 2) k_ml_small(const int32_t* __restrict__ lines, const uint8_t* __restrict__ bad, uint32_t n, uint32_t gsz,
                                                    uint32_t groups, uint32_t* __restrict__ partials, uint32_t pstride,
                                                    const uint32_t* __restrict__ count, const uint32_t* __restrict__ index)
#if BLSGPU_EMIT(BLSGPU_TU_ML)
{
    const Team t = team_of_lane();
    if (count != nullptr) {
        groups = min(groups, __builtin_amdgcn_readfirstlane(*(volatile const uint32_t*)count));
        if (wave_index() * TEAMS >= groups) return;
    }
    const uint32_t g = wave_index() * TEAMS + t.slot;
    const bool valid = t.slot < (uint32_t)TEAMS && g < groups;
    const size_t first = (size_t)(valid ? g : 0u) * gsz;
    int32_t fre[NL], fim[NL];
    set_one(fre, fim, t);
#pragma unroll 1
    for (uint32_t L = 0; L < (uint32_t)LINES; L++) {
        if (L > 0 && line_is_tangent(L)) {
            int32_t cre[NL], cim[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) { cre[k] = fre[k]; cim[k] = fim[k]; }
            mul_dense(fre, fim, t, TeamRec{cre, cim, t.base4});
        }
        const int32_t* base = lines + ((size_t)L * n + first) * LINE_DW;
#pragma unroll 1
        for (uint32_t i = 0; i < gsz; i++) {
            uint32_t jA, jB;
            const uint32_t flag = bad[first + i];
            const bool use = flag != 255u;
            line_positions(L, flag, jA, jB);
            const int4* rec = reinterpret_cast<const int4*>(base + (size_t)i * LINE_DW);
            int32_t y[6][NL];
            {
                int4 q[LINE_DW / 4];
#pragma unroll
                for (int k = 0; k < LINE_DW / 4; k++) q[k] = rec[k];
#pragma unroll
                for (int k = 0; k < LINE_DW / 4; k++) {
                    const int e = 4 * k;
                    y[e / NL][e % NL] = q[k].x; y[(e + 1) / NL][(e + 1) % NL] = q[k].y;
                    y[(e + 2) / NL][(e + 2) % NL] = q[k].z; y[(e + 3) / NL][(e + 3) % NL] = q[k].w;
                }
            }
            int32_t re[NL], im[NL];                         // (f: normalised digits, from a dense product's sum or a sparse product)
#pragma unroll
            for (int k = 0; k < NL; k++) { re[k] = fre[k]; im[k] = fim[k]; }
            mul_line_k3(re, im, t, jA, jB, y);
#pragma unroll
            for (int k = 0; k < NL; k++) { fre[k] = use ? re[k] : fre[k]; fim[k] = use ? im[k] : fim[k]; }
        }
    }
    if (valid) {
        const uint32_t flat = (t.c & 1u) ? 3u + (t.c >> 1) : (t.c >> 1);
        uint32_t* o = partials + (size_t)(index != nullptr ? index[g] : g) * pstride + flat * 24u;
        fe a, b;
#pragma unroll
        for (int k = 0; k < NL; k++) { a.v[k] = fre[k]; b.v[k] = fim[k]; }
        uint32_t w[12];
        r28::to_vm(w, a);
#pragma unroll
        for (int k = 0; k < 12; k++) o[k] = w[k];
        r28::to_vm(w, b);
#pragma unroll
        for (int k = 0; k < 12; k++) o[12 + k] = w[k];
    }
}
#else
;
#endif

}  // namespace ml

}  // namespace blsgpu
