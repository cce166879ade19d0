// blsgpu_fexp.hip -- final exponentiations in BATCHES: six lanes per result, ten results per wavefront, on the
// 28-bit-limb register arithmetic of fp28.h (round 3; included by blsgpu_api.hip after blsgpu_ml.hip).
//
// fq12_final_exp (fields_t.py:44, 1124-1128): f -> f^((q^12 - 1)/n).  The wavefront VM spends one wavefront and
// ~575 k instructions on a result (k_final_groups / k_reduce, 1.25 ms of latency each); a batch of thousands -- the
// 10 000 verifies of BASELINE configs[3], any stream of single-signature verifications -- is bound by that
// instruction count.  Here a result costs ~130 k wave-level instructions: the same exponent chain
// (vmgen/programs.final_exp_script; easy part (q^6 - 1)(q^2 + 1), hard part ((x-1)^2/3)(x+q)(x^2+q^2-1) + 1) as a
// SCRIPT (fexp_tables_gfx950.h, generated from vmgen/fexp_model.py, which is pinned to the reference's vectors on the
// CPU) over one accumulator in registers -- lane k of a team holds the coefficient f_k of sum f_k w^k -- and five
// slots in HBM:
//   MUL   acc *= slot          dense product of blsgpu_ml.hip (ds_bpermute fetches, sums of six products)
//   CSQ   acc <- acc^(2^n)     Granger-Scott squaring of the cyclotomic subgroup: every lane ONE pair of sums of three
//                              products (operands chosen by the lane's role: x^2 + xi y^2, 2xy, xi 2xy)
//   CONJ / FROB                f^(q^6) (odd coefficients change sign) / f^(q^i), i = 1, 2, 4: lane-local products by constants
//   TINV                       the one inversion: of an Fq2 value (coefficient 0) through its norm and the VM's safegcd
//                              inversion (0 -> 0 as fields_t.py:47-55); the Fq12 inverse itself is products and Frobenius maps:
//                              N = f conj(f) in Fq6, t = N N^(q^2) N^(q^4) in Fq2, f^-1 = conj(f) N^(q^2) N^(q^4) t^-1.
// Used for batches of at least fexp_team_threshold results (blsgpu_api.hip); a single result keeps the VM's latency.
#pragma once
#include "fexp_tables_gfx950.h"

namespace blsgpu {
namespace fx {
using r28::fe;
using r28::NL;
using ml::Team;
using ml::bperm14;

// x - k q with k = the multiple of q nearest to x, read off the top limb (q = 106513.12 x 2^364; the lower limbs of an
// unnormalised x shift the estimate by less than 1e-4): digits normalised, value in (-2q, 2q) for any |x| < 2^12 q
__device__ __forceinline__ void reduce_small(int32_t* __restrict__ r, const int32_t* __restrict__ x) {
    const int32_t n[NL] = BLS28_Q;
    const int32_t k = (int32_t)__builtin_rintf((float)x[NL - 1] * (1.0f / 106513.12f));
    int64_t c = 0;
#pragma unroll
    for (int j = 0; j < NL - 1; j++) {
        c += (int64_t)x[j] - (int64_t)k * n[j];
        r[j] = (int32_t)((uint32_t)c & (uint32_t)r28::LMASK);
        c >>= r28::LW;
    }
    r[NL - 1] = (int32_t)(c + (int64_t)x[NL - 1] - (int64_t)k * n[NL - 1]);
}
// the cyclotomic squaring of the team's accumulator (vmgen/fexp_model.cyc_sqr_lane_forms)
__device__ __forceinline__ void cyc_sqr(int32_t* __restrict__ fre, int32_t* __restrict__ fim, const Team& t) {
    const uint32_t pair = (t.c == 0u || t.c == 3u) ? 0u : ((t.c == 2u || t.c == 5u) ? 1u : 2u);
    const uint32_t ax = t.base4 + pair * 4u, ay = ax + 12u;
    int32_t xr[NL], xi[NL], yr[NL], yi[NL];
    bperm14(xr, fre, ax); bperm14(xi, fim, ax); bperm14(yr, fre, ay); bperm14(yi, fim, ay);
    const bool ev = (t.c & 1u) == 0u, l1 = t.c == 1u;
    int32_t a1r[NL], b1r[NL], a2r[NL], b2r[NL], a3[NL], b1i[NL], a2i[NL], b2i[NL], x2r[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int32_t xs = xr[j] + xi[j], xd = xr[j] - xi[j], ys = yr[j] + yi[j], yd = yr[j] - yi[j];
        const int32_t x2i = xi[j] + xi[j];
        x2r[j] = xr[j] + xr[j];
        // real part:  even (xs)(xd) + (yd)(yd) + (-2yi)(yi) | lane 1 (2xr)(yd) + (-2xi)(ys) | lanes 3, 5 (2xr)(yr) + (-2xi)(yi)
        a1r[j] = ev ? xs : x2r[j];
        b1r[j] = ev ? xd : (l1 ? yd : yr[j]);
        a2r[j] = ev ? yd : -x2i;
        b2r[j] = ev ? yd : (l1 ? ys : yi[j]);
        a3[j] = ev ? -(yi[j] + yi[j]) : 0;
        // imaginary part:  even (2xr)(xi) + (ys)(ys) + (-2yi)(yi) | lane 1 (2xr)(ys) + (2xi)(yd) | lanes 3, 5 (2xr)(yi) + (2xi)(yr)
        b1i[j] = ev ? xi[j] : (l1 ? ys : yi[j]);
        a2i[j] = ev ? ys : x2i;
        b2i[j] = ev ? ys : (l1 ? yd : yr[j]);
    }
    // column bound (units of 2^56, digits of f normalised): real 2 + 1 + 2, imaginary 2 + 4 + 2 = 8 on even lanes; 2 + 4 on lane 1
    int32_t tr[NL], ti[NL];
    bls28::fp28_dot3(tr, a1r, b1r, a2r, b2r, a3, yi);
    bls28::fp28_dot3(ti, x2r, b1i, a2i, b2i, a3, yi);
    int32_t sr[NL], si[NL];                                    // 3 t -+ 2 f_k: a value in (-10 q, 14 q) ...
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int32_t dr = fre[j] + fre[j], di = fim[j] + fim[j];
        sr[j] = 3 * tr[j] + (ev ? -dr : dr);
        si[j] = 3 * ti[j] + (ev ? -di : di);
    }
    reduce_small(fre, sr);                                     // ... brought back to (-2q, 2q): the squarings are chained
    reduce_small(fim, si);                                     // by the dozen and the linear term would double the range each time
}
// f^(q^i): conj^i(f_k) gamma_{i,k}
__device__ __forceinline__ void frob(int32_t* __restrict__ fre, int32_t* __restrict__ fim, const Team& t, uint32_t j3) {
    const bool cj = j3 == 0u;                                  // q^1 conjugates; q^2 and q^4 do not
    int32_t gr[NL], gi[NL], xim[NL], nxim[NL], re[NL], im[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        gr[j] = BLS28_GAMMA[j3][t.c][0][j]; gi[j] = BLS28_GAMMA[j3][t.c][1][j];
        xim[j] = cj ? -fim[j] : fim[j]; nxim[j] = cj ? fim[j] : -fim[j];
    }
    bls28::fp28_dot2(re, fre, gr, nxim, gi);
    bls28::fp28_dot2(im, fre, gi, xim, gr);
#pragma unroll
    for (int j = 0; j < NL; j++) { fre[j] = re[j]; fim[j] = im[j]; }
}
// a^(q-2) for every lane's a (sliding windows, the schedule of fexp_tables_gfx950.h)
__device__ __forceinline__ fe fq_inverse(const fe& b) {
    fe tbl[4];
    tbl[0] = b;
    const fe b2 = r28::sqr(b);
#pragma unroll
    for (int i = 1; i < 4; i++) tbl[i] = r28::mul(tbl[i - 1], b2);
    auto pick = [&](uint32_t k) {
        fe m;
#pragma unroll
        for (int j = 0; j < NL; j++) {
            m.v[j] = tbl[0].v[j];
#pragma unroll
            for (int i = 1; i < 4; i++) m.v[j] = (k == (uint32_t)i) ? tbl[i].v[j] : m.v[j];
        }
        return m;
    };
    fe a = pick(BLS28_INV_WIN[0][1]);
#pragma unroll 1
    for (int s = 1; s < BLS28_INV_STEPS; s++) {
        const uint32_t nsq = BLS28_INV_WIN[s][0], k = BLS28_INV_WIN[s][1];
#pragma unroll 1
        for (uint32_t i = 0; i < nsq; i++) a = r28::sqr(a);
        if (k != 255u) a = r28::mul(a, pick(k));
    }
    return a;
}
// the accumulator holds an Fq2 value t in coefficient 0 (the other lanes' contents do not matter): acc <- 1/t
__device__ __forceinline__ void tinv(int32_t* __restrict__ fre, int32_t* __restrict__ fim, const Team& t) {
    fe re, im, nim, n;
#pragma unroll
    for (int j = 0; j < NL; j++) { re.v[j] = fre[j]; im.v[j] = fim[j]; nim.v[j] = -fim[j]; }
    bls28::fp28_dot2(n.v, re.v, re.v, im.v, im.v);
    // 1/n by the VM's branch-free safegcd inversion (fq32.h fq_inv, 0 -> 0): ~20 k instructions against ~220 k for
    // the fixed power n^(q-2) (fq_inverse above, kept for BLSGPU_FEXP_FERMAT builds); the forms differ by 2^8
#ifdef BLSGPU_FEXP_FERMAT
    const fe ni = fq_inverse(n);
#else
    uint32_t w[12], v[12];
    r28::to_vm(w, n);
    bls::fq_inv(v, w);
    const fe ni = r28::from_vm(v);
#endif
    const fe a = r28::mul(re, ni), b = r28::mul(nim, ni);
#pragma unroll
    for (int j = 0; j < NL; j++) { fre[j] = t.c == 0u ? a.v[j] : 0; fim[j] = t.c == 0u ? b.v[j] : 0; }
}

// Team g: product of the partials in[(i * istride + g * gstride) * 144], i < m (the wavefront VM's form), then the final
// exponentiation; 576 canonical big-endian bytes to out_bytes[g].  ws: NSLOTS x 168 dwords per team.
__global__ void __launch_bounds__(256, 2) k_fexp_team(const uint32_t* __restrict__ in, uint32_t m, uint32_t istride, uint32_t gstride,
                                                     uint32_t groups, int32_t* __restrict__ ws, uint32_t* __restrict__ out_bytes,
                                                     uint32_t* __restrict__ dbg)
#if BLSGPU_EMIT(BLSGPU_TU_FX)
{
    const Team t = ml::team_of_lane();
    const uint32_t g = wave_index() * ml::TEAMS + t.slot;
    const bool valid = t.slot < (uint32_t)ml::TEAMS && g < groups;
    const uint32_t gc = valid ? g : 0u;
    const uint32_t flat = (t.c & 1u) ? 3u + (t.c >> 1) : (t.c >> 1);           // w-powers 0,2,4,1,3,5 in the flat order
    int32_t fre[NL], fim[NL];
    ml::set_one(fre, fim, t);
#pragma unroll 1
    for (uint32_t i = 0; i < m; i++) {
        const uint32_t* p = in + ((size_t)i * istride + (size_t)gc * gstride) * 144 + flat * 24u;
        uint32_t w[12];
        int32_t yre[NL], yim[NL];
#pragma unroll
        for (int j = 0; j < 12; j++) w[j] = p[j];
        const fe a = r28::from_vm(w);
#pragma unroll
        for (int j = 0; j < 12; j++) w[j] = p[12 + j];
        const fe b = r28::from_vm(w);
#pragma unroll
        for (int j = 0; j < NL; j++) { yre[j] = a.v[j]; yim[j] = b.v[j]; }
        if (i == 0) {
#pragma unroll
            for (int j = 0; j < NL; j++) { fre[j] = yre[j]; fim[j] = yim[j]; }
        } else {
            ml::mul_dense(fre, fim, t, ml::TeamRec{yre, yim, t.base4});
        }
    }
    int32_t* slots = ws + (size_t)(wave_index() * (ml::TEAMS + 1) + t.slot) * BLS28_FEXP_NSLOTS * ml::DENSE_DW;
#pragma unroll 1
    for (uint32_t pc = 0; pc < (uint32_t)BLS28_FEXP_NOPS; pc++) {
        const uint32_t op = BLS28_FEXP_OPS[pc][0], arg = BLS28_FEXP_OPS[pc][1];
        if (op == 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            ml::mul_dense(fre, fim, t, ml::GlobalRec{slots + arg * ml::DENSE_DW});
        } else if (op == 2u) {
#pragma unroll 1
            for (uint32_t i = 0; i < arg; i++) cyc_sqr(fre, fim, t);
        } else if (op == 3u) {
            int32_t* o = slots + arg * ml::DENSE_DW + t.c * 2 * NL;
#pragma unroll
            for (int j = 0; j < NL; j++) { o[j] = fre[j]; o[NL + j] = fim[j]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        } else if (op == 4u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int32_t* o = slots + arg * ml::DENSE_DW + t.c * 2 * NL;
#pragma unroll
            for (int j = 0; j < NL; j++) { fre[j] = o[j]; fim[j] = o[NL + j]; }
        } else if (op == 5u) {
            r28::F<1, 1> a, b;
#pragma unroll
            for (int j = 0; j < NL; j++) { a.v[j] = (t.c & 1u) ? -fre[j] : fre[j]; b.v[j] = (t.c & 1u) ? -fim[j] : fim[j]; }
            const fe na = r28::norm(a), nb = r28::norm(b);
#pragma unroll
            for (int j = 0; j < NL; j++) { fre[j] = na.v[j]; fim[j] = nb.v[j]; }
        } else if (op == 6u) {
            frob(fre, fim, t, arg);
        } else if (op == 7u) {
            tinv(fre, fim, t);
        } else {
            break;
        }
        if (dbg != nullptr && g == 0u) {                       // trace of result 0 after every operation (tools/fexp_trace.py)
            uint32_t* o = dbg + (size_t)pc * 144 + flat * 24u;
            fe a, b;
#pragma unroll
            for (int j = 0; j < NL; j++) { a.v[j] = fre[j]; b.v[j] = fim[j]; }
            const fe one = r28::fe_one();
            uint32_t w[12];
            r28::to_raw(w, r28::mul(a, one));
#pragma unroll
            for (int j = 0; j < 12; j++) o[j] = bswap32(w[11 - j]);
            r28::to_raw(w, r28::mul(b, one));
#pragma unroll
            for (int j = 0; j < 12; j++) o[12 + j] = bswap32(w[11 - j]);
        }
    }
    if (valid) {
        uint32_t* o = out_bytes + (size_t)g * 144 + flat * 24u;
        fe a, b;
#pragma unroll
        for (int j = 0; j < NL; j++) { a.v[j] = fre[j]; b.v[j] = fim[j]; }
        uint32_t w[12];
        r28::to_raw(w, a);
#pragma unroll
        for (int j = 0; j < 12; j++) o[j] = bswap32(w[11 - j]);
        r28::to_raw(w, b);
#pragma unroll
        for (int j = 0; j < 12; j++) o[12 + j] = bswap32(w[11 - j]);
    }
}
#else
;
#endif
}  // namespace fx
}  // namespace blsgpu
