// blsgpu_fexpw.hip -- the final exponentiation of ONE result on ONE wavefront with every Fq product of a step on its
// own lane (round 4; included by blsgpu_api.hip after blsgpu_fexp.hip; model + table generator: vmgen/fexpw_model.py).
//
// fq12_final_exp (fields_t.py:44, 1124-1128) of a single result is a dependent chain of 314 cyclotomic squarings and 59
// dense products; what a call waits for is the number of instructions ONE wavefront has to issue for it (a lone
// wavefront issues a vector instruction every ~5.5 cycles whatever it is).  The wavefront VM needs ~550 k (a field
// product per lane, but every linear combination is a round of its own through LDS): 1.25 ms, the floor under every
// call of rounds 1 - 3.  Here the squaring is ~620 instructions:
//
//   quad o = lane / 4, o < 12: the HOME of the Fq value number o = 2 k + part of f = sum_k f_k w^k; its four lanes hold
//   the value times 1, -1, 2, -2 (14 signed 28-bit limbs each, fp28.h).  Quad 12 holds the constant 1/3, quad 15 zero.
//   A step: every lane fetches its operands from other lanes' registers (ds_bpermute) and ADDS two of them limb-wise --
//   (X + X')(Y + Y'): the additions of the Granger-Scott formulas and the xi-wrap of the dense product cost no carries,
//   and their small coefficients are in the choice of the source lane (the -1 / 2 / -2 variant, or the zero quad) --,
//   multiplies (ONE Montgomery reduction: fp28_dot1, or fp28_dot3 for three products), the four lanes of a quad add
//   their results (DPP), every lane scales the sum by the step's constant times its own variant's factor and normalises:
//   the quad's new value.  CSQ puts the three products of fexp_model.cyc_sqr_lane_forms on lanes 0 .. 2 and the -+2 f_k
//   term as (-+2 f_k)(1/3) on lane 3 (the sum is scaled by 3); MUL puts the twelve products of an output part three per
//   lane.  Values stay below 16 q with no modular correction at all (vmgen/fexpw_model.py runs the tables digit by digit
//   with the multiplier's 64-bit column bounds asserted).  Slots of the script (fexp_tables_gfx950.h, the same as
//   k_fexp_team's) are LDS rows.
//
// Used for calls that end in fewer results than the six-lanes-per-result form wants (blsgpu_api.hip launch_fexp_wide);
// both forms and the VM's return the reference's bytes (tests/test_gpu_fexp_wide.py).
#pragma once
#include "fexpw_tables_gfx950.h"

namespace blsgpu {
namespace fxw {
using r28::fe;
using r28::NL;

constexpr int ROW = 16;                                    // dwords per lane and slot in LDS (14 used)

__device__ __forceinline__ int32_t bperm(uint32_t addr, int32_t v) { return __builtin_amdgcn_ds_bpermute((int)addr, v); }
// X[s1] + X[s2], limb by limb
__device__ __forceinline__ void fetch2(int32_t* __restrict__ out, const int32_t* __restrict__ X, uint32_t s1, uint32_t s2) {
#pragma unroll
    for (int j = 0; j < NL; j++) out[j] = bperm(s1, X[j]) + bperm(s2, X[j]);
}
__device__ __forceinline__ void fetch1(int32_t* __restrict__ out, const int32_t* __restrict__ X, uint32_t s) {
#pragma unroll
    for (int j = 0; j < NL; j++) out[j] = bperm(s, X[j]);
}
// limbs of any size times the lane's factor, carry pass: limbs 0 .. 12 in [0, 2^28), the sign in limb 13
__device__ __forceinline__ void scale_norm(int32_t* __restrict__ V, const int32_t* __restrict__ t, int32_t scale) {
    int64_t c = 0;
#pragma unroll
    for (int j = 0; j < NL - 1; j++) {
        c += (int64_t)t[j] * scale;
        V[j] = (int32_t)((uint32_t)c & (uint32_t)r28::LMASK);
        c >>= r28::LW;
    }
    V[NL - 1] = (int32_t)(c + (int64_t)t[NL - 1] * scale);
}
// the four lanes of a quad add their limbs; every lane keeps the sum times `scale` (the step's constant times the lane's
// variant factor), limbs normalised; written to V on the home lanes only
__device__ __forceinline__ void quad_sum_store(int32_t* __restrict__ V, const int32_t* __restrict__ p, int32_t scale, bool home_lane) {
    int32_t t[NL], s[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int32_t u = p[j] + __builtin_amdgcn_update_dpp(0, p[j], 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
        t[j] = u + __builtin_amdgcn_update_dpp(0, u, 0x4E, 0xF, 0xF, true);                       // quad_perm [2,3,0,1]
    }
    scale_norm(s, t, scale);
    if (home_lane) {
#pragma unroll
        for (int j = 0; j < NL; j++) V[j] = s[j];
    }
}
struct P1 { uint32_t s1, s2, s3, s4; };
struct P3 { uint32_t s1[3], s2[3], s3[3]; };
__device__ __forceinline__ P1 load_p1(const int32_t (&T)[4][64], uint32_t lane) {
    return {(uint32_t)T[0][lane], (uint32_t)T[1][lane], (uint32_t)T[2][lane], (uint32_t)T[3][lane]};
}
__device__ __forceinline__ P3 load_p3(const int32_t (&T)[3][3][64], uint32_t lane) {
    P3 p;
#pragma unroll
    for (int i = 0; i < 3; i++) { p.s1[i] = (uint32_t)T[i][0][lane]; p.s2[i] = (uint32_t)T[i][1][lane]; p.s3[i] = (uint32_t)T[i][2][lane]; }
    return p;
}
// V <- 3 x variant x quad sum of (V[s1] + V[s2]) (V[s3] + V[s4]).  Column bound of fp28_dot1 (units of 2^56): the limbs of
// every variant are normalised, an operand is the sum of two: 2 x 2 = 4 of the 8 that 64 bits hold.
__device__ __forceinline__ void step_vv(int32_t* __restrict__ V, const P1& p, int32_t variant, bool home_lane) {
    int32_t A[NL], B[NL], r[NL];
    fetch2(A, V, p.s1, p.s2);
    fetch2(B, V, p.s3, p.s4);
    bls28::fp28_dot1(r, A, B);
    quad_sum_store(V, r, 3 * variant, home_lane);
}
// V <- variant x quad sum of sum_i (V[s1] + V[s2]) G[s3]: 3 x 2 x 1 = 6 units
__device__ __forceinline__ void step_vg(int32_t* __restrict__ V, const int32_t* __restrict__ G, const P3& p, int32_t variant, bool home_lane) {
    int32_t A0[NL], A1[NL], A2[NL], B0[NL], B1[NL], B2[NL], r[NL];
    fetch2(A0, V, p.s1[0], p.s2[0]);
    fetch2(A1, V, p.s1[1], p.s2[1]);
    fetch2(A2, V, p.s1[2], p.s2[2]);
    fetch1(B0, G, p.s3[0]);
    fetch1(B1, G, p.s3[1]);
    fetch1(B2, G, p.s3[2]);
    bls28::fp28_dot3(r, A0, B0, A1, B1, A2, B2);
    quad_sum_store(V, r, variant, home_lane);
}
// the accumulator holds t in Fq2 in coefficient 0: V <- conj(t) / (tr^2 + ti^2), 0 -> 0 (fields_t.py:47-55); every lane runs the
// same inversion (the VM's branch-free safegcd routine, fq32.h) on the same value
__device__ __forceinline__ void tinv(int32_t* __restrict__ V, uint32_t quad, int32_t variant, bool home_lane) {
    fe re, im, nim, n;
    fetch1(re.v, V, 0u);                                   // lane 0: +t.re, lane 4: +t.im
    fetch1(im.v, V, 16u);
#pragma unroll
    for (int j = 0; j < NL; j++) nim.v[j] = -im.v[j];
    bls28::fp28_dot2(n.v, re.v, re.v, im.v, im.v);
    uint32_t w[12], v[12];
    r28::to_vm(w, n);
    bls::fq_inv_var(v, w);                                 // the same value in every lane: the data-dependent form (fq32.h)
    const fe ni = r28::from_vm(v);
    const fe a = r28::mul(re, ni), b = r28::mul(nim, ni);
    int32_t t[NL], s[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) t[j] = quad == 0u ? a.v[j] : (quad == 1u ? b.v[j] : 0);
    scale_norm(s, t, variant);
    if (home_lane) {
#pragma unroll
        for (int j = 0; j < NL; j++) V[j] = s[j];
    }
}

// the script of fexp_tables_gfx950.h (vmgen/fexp_model.script) on the accumulator V; G and the LDS rows are scratch
__device__ __forceinline__ void final_exp_script(int32_t* __restrict__ V, int32_t* __restrict__ G, int32_t (*slots)[64][ROW], uint32_t lane,
                                                 uint32_t quad, int32_t variant, bool home_lane, const P3& p_mul,
                                                 unsigned long long* __restrict__ stamps) {
    const P1 p_csq = load_p1(BLS28W_CSQ, lane);
    uint32_t g_slot = 255u;                                // the slot G holds (255: none)
    const bool stamp = stamps != nullptr && lane == 0u;
    if (stamp) stamps[0] = __builtin_readcyclecounter();
#pragma unroll 1
    for (uint32_t pc = 0; pc < (uint32_t)BLS28_FEXP_NOPS; pc++) {
        const uint32_t op = BLS28_FEXP_OPS[pc][0], arg = BLS28_FEXP_OPS[pc][1];
        if (stamp && pc) stamps[pc] = __builtin_readcyclecounter();
        if (op == 2u || op == 5u) {                        // CSQ n | CONJ: the "vv" kinds
            const P1 p = op == 2u ? p_csq : load_p1(BLS28W_CONJ, lane);
            const uint32_t cnt = op == 2u ? arg : 1u;
#pragma unroll 1
            for (uint32_t i = 0; i < cnt; i++) step_vv(V, p, variant, home_lane);
        } else if (op == 1u || op == 6u) {                 // MUL slot | FROB j: the "vg" kinds
            P3 p = p_mul;
            if (op == 1u) {
                if (g_slot != arg) {
#pragma unroll
                    for (int j = 0; j < NL; j++) G[j] = slots[arg][lane][j];
                    g_slot = arg;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NL; j++) G[j] = BLS28W_GAMMA[arg][lane][j];
                g_slot = 255u;
                p = arg == 0u ? load_p3(BLS28W_FROBC, lane) : load_p3(BLS28W_FROB, lane);
            }
            step_vg(V, G, p, variant, home_lane);
        } else if (op == 3u) {                             // ST slot
#pragma unroll
            for (int j = 0; j < NL; j++) slots[arg][lane][j] = V[j];
            if (g_slot == arg) g_slot = 255u;
        } else if (op == 4u) {                             // LD slot (the constant lanes get their own constant back)
#pragma unroll
            for (int j = 0; j < NL; j++) V[j] = slots[arg][lane][j];
        } else if (op == 7u) {
            tinv(V, quad, variant, home_lane);
        } else {
            break;
        }
    }
    if (stamp) stamps[BLS28_FEXP_NOPS] = __builtin_readcyclecounter();
}
// the value itself (lane 0 of a home quad), canonical, as 48 big-endian bytes
__device__ __forceinline__ void store_bytes(const int32_t* __restrict__ V, uint32_t* __restrict__ o, bool active) {
    if (!active) return;
    fe a;
#pragma unroll
    for (int j = 0; j < NL; j++) a.v[j] = V[j];
    uint32_t w[12];
    r28::to_raw(w, a);
#pragma unroll
    for (int j = 0; j < 12; j++) o[j] = bswap32(w[11 - j]);
}

// Result g = blockIdx.x: product of the partials in[(i * istride + g * gstride) * 144], i < m (the wavefront VM's form:
// 12 x 12 words x 2^384, flat order), then the final exponentiation; 576 canonical big-endian bytes to out_bytes[g].
// stamps (diagnostic, may be null): result 0 writes the cycle counter before the script and after every operation of it
// (tools/fexpw_stamps.py).
__global__ void __launch_bounds__(64) k_fexp_wide(const uint32_t* __restrict__ in, uint32_t m, uint32_t istride, uint32_t gstride,
                                                  uint32_t* __restrict__ out_bytes, unsigned long long* __restrict__ stamps)
#if BLSGPU_EMIT(BLSGPU_TU_FXW)
{
    __shared__ int32_t slots[BLS28_FEXP_NSLOTS][64][ROW];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t quad = lane >> 2;
    const bool home_lane = quad < 12u;
    const uint32_t k = (quad >> 1) % 6u, part = quad & 1u;
    const uint32_t flat = (k & 1u) ? 3u + (k >> 1) : (k >> 1);                 // w-powers 0,2,4,1,3,5 in the flat order
    const size_t g = blockIdx.x;
    const uint32_t vr = lane & 3u;
    const int32_t variant = vr == 0u ? 1 : (vr == 1u ? -1 : (vr == 2u ? 2 : -2));     // this lane holds variant x the quad's value
    int32_t V[NL], G[NL];
    {
        const int32_t third[NL] = BLS28W_THIRD;
        int32_t t[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) { t[j] = quad == 12u ? third[j] : 0; G[j] = 0; }
        scale_norm(V, t, variant);                         // quad 12: 1/3 in its variants; quads 13 .. 15: zero
    }
    const P3 p_mul = load_p3(BLS28W_MUL, lane);
#pragma unroll 1
    for (uint32_t i = 0; i < m; i++) {
        const uint32_t* p = in + ((size_t)i * istride + g * gstride) * 144 + flat * 24u + part * 12u;
        uint32_t w[12];
#pragma unroll
        for (int j = 0; j < 12; j++) w[j] = p[j];
        const fe a = r28::from_vm(w);
        if (i == 0) {
            int32_t s[NL];
            scale_norm(s, a.v, variant);
            if (home_lane) {
#pragma unroll
                for (int j = 0; j < NL; j++) V[j] = s[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NL; j++) G[j] = a.v[j];    // the products read lane 0 of a quad only: the value itself
            step_vg(V, G, p_mul, variant, home_lane);
        }
    }
    final_exp_script(V, G, slots, lane, quad, variant, home_lane, p_mul, g == 0 ? stamps : nullptr);
    store_bytes(V, out_bytes + g * 144 + flat * 24u + part * 12u, home_lane && (lane & 3u) == 0u);
}
#else
;
#endif

// ---- the same lane layout for the serial tails of the line-stream Miller stage (blsgpu_ml.hip) -------------------------
// A dense record is 12 x 14 limbs in w-power order: value (k, part) at [(2 k + part) * 14].  In the six-lanes-per-value
// layout of k_ml_merge / k_ml_horner_wide a dense product is 6.8 k (1.6 k) instructions deep; with a product per lane it is
// 1.2 k: what a call waits for once few teams are left.
__device__ __forceinline__ void load_dense(int32_t* __restrict__ X, const int32_t* __restrict__ rec, uint32_t quad) {
    const int32_t* p = rec + (quad < 12u ? quad : 0u) * NL;
#pragma unroll
    for (int j = 0; j < NL; j++) X[j] = p[j];
}
// One wavefront per output record: team (g, jo, L) multiplies the records (g, j, L), j in [jo * fan, min(cpg_in, (jo + 1) * fan)),
// of `in` (indexed as k_ml_accum's output with cpg_in) -> out (cpg_out): the tree levels of k_ml_merge with few outputs.
__global__ void __launch_bounds__(64) k_ml_merge_wide(const int32_t* __restrict__ in, uint32_t cpg_in, uint32_t fan, uint32_t cpg_out,
                                                      int32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_FXW)
{
    const uint32_t lane = threadIdx.x & 63u, quad = lane >> 2, vr = lane & 3u;
    const bool home_lane = quad < 12u;
    const int32_t variant = vr == 0u ? 1 : (vr == 1u ? -1 : (vr == 2u ? 2 : -2));
    const uint32_t id = blockIdx.x;
    const uint32_t L = id % (uint32_t)ml::LINES, gj = id / (uint32_t)ml::LINES, jo = gj % cpg_out, g = gj / cpg_out;
    const uint32_t lo = jo * fan, hi = min(cpg_in, lo + fan);
    const P3 p_mul = load_p3(BLS28W_MUL, lane);
    const int32_t* rec = in + ((size_t)(g * cpg_in + lo) * ml::LINES + L) * ml::DENSE_DW;
    int32_t V[NL], G[NL], t[NL];
    load_dense(t, rec, quad);
    scale_norm(V, t, home_lane ? variant : 0);             // quads 12 .. 15: zero
#pragma unroll 1
    for (uint32_t i = lo + 1u; i < hi; i++) {
        rec += (size_t)ml::LINES * ml::DENSE_DW;
        load_dense(G, rec, quad);
        step_vg(V, G, p_mul, variant, home_lane);
    }
    if (home_lane && vr == 0u) {
        int32_t* o = out + (size_t)id * ml::DENSE_DW + quad * NL;
#pragma unroll
        for (int j = 0; j < NL; j++) o[j] = V[j];
    }
}
#else
;
#endif

// Group g = blockIdx.x: f = M_0; for L = 1 .. 67: (tangent: f <- f^2;) f <- f M_L with M_L = prods[(g * 68 + L) * 168] (the serial
// chain of the line-stream stage, k_ml_horner_wide's job) -- then either the wavefront VM's form of f to
// partials[g * pstride] (144 words; the sharded entries and k_reduce take it from there), or, with out_bytes, the final
// exponentiation right here (no partial, no second launch): 576 canonical big-endian bytes to out_bytes[g].
__global__ void __launch_bounds__(64) k_ml_horner_fexp(const int32_t* __restrict__ prods, uint32_t* __restrict__ partials, uint32_t pstride,
                                                       uint32_t* __restrict__ out_bytes)
#if BLSGPU_EMIT(BLSGPU_TU_FXW)
{
    __shared__ int32_t slots[BLS28_FEXP_NSLOTS][64][ROW];
    const uint32_t lane = threadIdx.x & 63u, quad = lane >> 2, vr = lane & 3u;
    const bool home_lane = quad < 12u;
    const int32_t variant = vr == 0u ? 1 : (vr == 1u ? -1 : (vr == 2u ? 2 : -2));
    const uint32_t k = (quad >> 1) % 6u, part = quad & 1u;
    const uint32_t flat = (k & 1u) ? 3u + (k >> 1) : (k >> 1);
    const size_t g = blockIdx.x;
    const P3 p_mul = load_p3(BLS28W_MUL, lane);
    const int32_t* rec = prods + g * ml::LINES * ml::DENSE_DW;
    int32_t V[NL], G[NL], t[NL];
    load_dense(t, rec, quad);
    {
        const int32_t third[NL] = BLS28W_THIRD;
#pragma unroll
        for (int j = 0; j < NL; j++) t[j] = home_lane ? t[j] : (quad == 12u ? third[j] : 0);
    }
    scale_norm(V, t, variant);
#pragma unroll 1
    for (uint32_t L = 1; L < (uint32_t)ml::LINES; L++) {
        if (ml::line_is_tangent(L)) {
#pragma unroll
            for (int j = 0; j < NL; j++) G[j] = V[j];      // lane 0 of a quad holds the value itself: f <- f f
            step_vg(V, G, p_mul, variant, home_lane);
        }
        load_dense(G, rec + (size_t)L * ml::DENSE_DW, quad);
        step_vg(V, G, p_mul, variant, home_lane);
    }
    if (out_bytes == nullptr) {
        if (home_lane && vr == 0u) {
            uint32_t* o = partials + g * pstride + flat * 24u + part * 12u;
            fe a;
#pragma unroll
            for (int j = 0; j < NL; j++) a.v[j] = V[j];
            uint32_t w[12];
            r28::to_vm(w, a);                              // a product by 2^384 mod q, canonical (limbs normalised, |value| < 16 q)
#pragma unroll
            for (int j = 0; j < 12; j++) o[j] = w[j];
        }
        return;
    }
    final_exp_script(V, G, slots, lane, quad, variant, home_lane, p_mul, nullptr);
    store_bytes(V, out_bytes + g * 144 + flat * 24u + part * 12u, home_lane && vr == 0u);
}
#else
;
#endif
}  // namespace fxw
}  // namespace blsgpu
