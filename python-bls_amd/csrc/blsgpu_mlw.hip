// blsgpu_mlw.hip -- the WIDE Miller loop (round 5): fq_miller_loop (fields_t.py:1091-1111; lines :1035-1078, twist point steps
// :641-686) of ONE pair on a workgroup of TWO wavefronts with a field product per lane -- the latency form for calls of a few
// pairs (BLS.verify of one signature is two pairs, bls.py:197-201).  Included by blsgpu_api.hip; model, formulas and the table
// generator: vmgen/mlw_model.py, vmgen/gen_mlw.py (tests/test_mlw_model.py pins the tables to the reference's vectors).
//
// What a small call waits for is the number of instructions ONE wavefront issues (a lone wavefront issues an instruction every
// ~5 cycles whatever it is).  The wavefront VM's k_miller needs ~280 k per pair (a field product per lane, but every linear
// combination a round of its own).  Here:
//
//   * wave 1 runs the twist-point chain T <- 2T (+ Q) and the line coefficients, one loop iteration AHEAD of wave 0, which runs
//     f <- f^2, f <- f l (sparse products); they meet at one barrier per iteration, the lines go through two buffers;
//   * every Fq value lives in LDS (the "value file": limb j of slot s at dword s + 64 j, so any lane reads any value with seven
//     ds_read2st64_b32 and 64 lanes reading 64 different slots never meet in a bank), in the multiples 1, -1, 2, -2 written by
//     the four lanes of the quad that computed it -- the small coefficients of the formulas are in the choice of the slot;
//   * a step (the only routine there is: wstep): every lane reads two pairs of operands, each the SUM of two slots, forms the sum of
//     two products with one Montgomery reduction (fp28_dot2), the four lanes of a quad add their results (DPP), every lane
//     multiplies the sum by its own scale, takes a multiple of q off (read from the top limb) while normalising the limbs, and
//     stores its slot.  All stored values lie in (-q/64, q + q/64): no range bookkeeping, no modular corrections.
//   * what the lanes read and write is DATA: the per-lane records and the two step programs of mlw_tables_gfx950.h.
//
// A pair the fast formulas are not valid for (Q flagged, Q off the twist, final Z = 0: DESIGN.md 2f) puts its block on the
// work list; k_ml_lines_exact (block mode) + k_ml_small (list mode) rewrite the block's partial like those of the wavefront-VM
// kernels (blsgpu_api.hip launch_miller).
#pragma once
#include "mlw_tables_gfx950.h"

namespace blsgpu {
namespace mlw {
using r28::fe;
using r28::NL;

constexpr int VF_DW = MLW_PAGES * MLW_PAGE_BYTES / 4;

struct Rec { uint32_t w[5]; };
__device__ __forceinline__ Rec load_rec(uint32_t kind, uint32_t lane) {
    Rec r;
    const uint32_t k = kind < (uint32_t)MLW_KINDS ? kind : 0u;          // (MLW_NOP: loaded, not used)
#pragma unroll
    for (int i = 0; i < 5; i++) r.w[i] = MLW_REC[k][i][lane];
    return r;
}
// V[a] + V[b], limb by limb (byte addresses of limb 0; limb j at + 256 j)
__device__ __forceinline__ void rd2(int32_t* __restrict__ out, const char* vf, uint32_t ab) {
    const char* pa = vf + (ab & 0xFFFFu);
    const char* pb = vf + (ab >> 16);
#pragma unroll
    for (int j = 0; j < NL; j++) out[j] = *reinterpret_cast<const int32_t*>(pa + 256 * j) + *reinterpret_cast<const int32_t*>(pb + 256 * j);
}
// limbs of s t - k q, normalised (limbs 0 .. 12 in [0, 2^28), the sign in limb 13), k = floor(s t[13] M47 / 2^47) ~ s t / q:
// the result lies in (-q/64, q + q/64) whatever s (vmgen/mlw_model.scale_reduce_norm is this routine)
__device__ __forceinline__ void srn(int32_t* __restrict__ V, const int32_t* __restrict__ t, int32_t s) {
    const int32_t qd[NL] = BLS28_Q;
    const int32_t top = t[NL - 1] * s;
    const int32_t nk = -(int32_t)(((int64_t)top * MLW_M47) >> 47);
    int64_t c = 0;
#pragma unroll
    for (int j = 0; j < NL - 1; j++) {
        c += (int64_t)t[j] * s + (int64_t)nk * qd[j];
        V[j] = (int32_t)((uint32_t)c & (uint32_t)r28::LMASK);
        c >>= r28::LW;
    }
    V[NL - 1] = (int32_t)(c + (int64_t)t[NL - 1] * s + (int64_t)nk * qd[NL - 1]);
}
__device__ __forceinline__ void st14(char* vf, uint32_t a, const int32_t* __restrict__ V) {
    char* p = vf + a;
#pragma unroll
    for (int j = 0; j < NL; j++) *reinterpret_cast<int32_t*>(p + 256 * j) = V[j];
}
// V[a], limb by limb
__device__ __forceinline__ void rd1(int32_t* __restrict__ out, const char* vf, uint32_t a) {
    const char* pa = vf + a;
#pragma unroll
    for (int j = 0; j < NL; j++) out[j] = *reinterpret_cast<const int32_t*>(pa + 256 * j);
}
// One step of a wavefront, in the four shapes the tables use (bits 8 - 9 of a program word): K products per lane; BS: every second
// operand is one slot (its second address is ignored); OCT: eight lanes per output (a third level of the lane sum).  Column bound of fp28_dot2 (units of 2^56): every stored limb is below 2^28,
// an operand the sum of two: 2 x (2 x 2) = 8, what 64 bits hold.
template <int K, bool BS, bool OCT = false>
__device__ __forceinline__ void wstep(char* vf, const Rec& r) {
    int32_t A0[NL], B0[NL], p[NL], t[NL], V[NL];
    rd2(A0, vf, r.w[0]);
    if (BS) rd1(B0, vf, r.w[1] & 0xFFFFu); else rd2(B0, vf, r.w[1]);
    if (K == 2) {
        int32_t A1[NL], B1[NL];
        rd2(A1, vf, r.w[2]);
        if (BS) rd1(B1, vf, r.w[3] & 0xFFFFu); else rd2(B1, vf, r.w[3]);
        bls28::fp28_dot2(p, A0, B0, A1, B1);
    } else {
        bls28::fp28_dot1(p, A0, B0);
    }
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int32_t u = p[j] + __builtin_amdgcn_update_dpp(0, p[j], 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
        t[j] = u + __builtin_amdgcn_update_dpp(0, u, 0x4E, 0xF, 0xF, true);                       // quad_perm [2,3,0,1]
        if (OCT) t[j] += __builtin_amdgcn_update_dpp(0, t[j], 0x141, 0xF, 0xF, true);             // row_half_mirror: the other quad of the eight lanes
    }
    srn(V, t, (int32_t)r.w[4] >> 16);
    st14(vf, r.w[4] & 0xFFFFu, V);
}
// a stored value is 0 mod q: its digits are those of 0 or of q
__device__ __forceinline__ bool stored_zero(const char* vf, uint32_t a) {
    fe x;
#pragma unroll
    for (int j = 0; j < NL; j++) x.v[j] = *reinterpret_cast<const int32_t*>(vf + a + 256 * j);
    return r28::is_zero(x);
}

// Block b = pair b of the call (group b / gsz): its Miller value (up to the factors the final exponentiation removes) as
// ONE partial in the wavefront VM's form: partials[b * 144 ...] (12 x 12 words x 2^384, the reference's flat order).
// W = 2: wave 0 the accumulator, wave 1 the chain.  W = 3 (calls of so few pairs that three SIMDs per pair are free): the accumulator's
// steps split over waves 0 and 1 -- an output's products on eight lanes, ONE per lane, so an accumulator step costs what a chain step
// costs (780 instead of ~1000 instructions) and a barrier follows each; wave 2 the chain.
// The two-wavefront form carries TWO pairs per workgroup (256 threads: four wavefronts, one per SIMD of a CU -- 128-thread workgroups
// were stacked on two SIMDs of a CU while the other two stayed empty: 342 - 512 pairs took 0.44 ms instead of 0.32); the pairs share
// nothing but the barriers.
template <int W>
__global__ void __launch_bounds__(W == 2 ? 256 : 192) k_miller_wide(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2, uint32_t n,
                                                                    uint32_t* __restrict__ partials, DegenList dg)
#if BLSGPU_EMIT(BLSGPU_TU_FXW)
{
    constexpr uint32_t PAIRS = W == 2 ? 2u : 1u;
    __shared__ int32_t vfiles[PAIRS][VF_DW];
    __shared__ int32_t bad_flags[PAIRS];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t slot = wv / (uint32_t)W, wave = wv - slot * (uint32_t)W;
    const uint32_t pair_raw = blockIdx.x * PAIRS + slot;
    const bool active = pair_raw < n;
    const uint32_t pair = active ? pair_raw : n - 1u;              // (an odd call's spare half repeats the last pair and writes nothing)
    int32_t* vfile = vfiles[slot];
    int32_t& bad_flag = bad_flags[slot];
    char* vf = reinterpret_cast<char*>(vfile);
    for (uint32_t i = wave * 64u + lane; i < (uint32_t)VF_DW; i += 64u * W) vfile[i] = 0;
    if (wave == 0u && lane == 0u) bad_flag = 0;
    __syncthreads();
    constexpr uint32_t CHAIN = W - 1;
    if (wave == CHAIN) {
        // the inputs in their four multiples: quad i of this wavefront stores value i of the list below
        const uint32_t qd = lane >> 2, vr = lane & 3u;
        const int32_t variant = vr == 0u ? 1 : (vr == 1u ? -1 : (vr == 2u ? 2 : -2));
        const uint32_t* s1 = g1 + (size_t)pair * 24;
        const uint32_t* s2 = g2 + (size_t)pair * 48;
        // value i: source words, scale, destination
        const uint32_t* src = qd == 0u ? s1 : (qd <= 2u ? s1 + 12 : s2 + 12u * ((qd - 3u) & 3u));
        const fe x = ml::load_coord(src);
        const int32_t one[NL] = BLS28_ONE;
        int32_t t[NL], V[NL];
        const bool is_one = qd >= 11u;
#pragma unroll
        for (int j = 0; j < NL; j++) t[j] = is_one ? one[j] : x.v[j];
        const int32_t sc = qd == 0u ? -3 : (qd == 2u ? 3 : 1);
        uint32_t dst = (uint32_t)MLW_AT_TRASH;
        switch (qd) {
            case 0: dst = MLW_AT_PX3N; break;   case 1: dst = MLW_AT_PY; break;     case 2: dst = MLW_AT_PY3; break;
            case 3: dst = MLW_AT_XQ0; break;    case 4: dst = MLW_AT_XQ1; break;    case 5: dst = MLW_AT_YQ0; break;
            case 6: dst = MLW_AT_YQ1; break;    case 7: dst = MLW_AT_X0; break;     case 8: dst = MLW_AT_X1; break;
            case 9: dst = MLW_AT_Y0; break;     case 10: dst = MLW_AT_Y1; break;    case 11: dst = MLW_AT_Z0; break;
            case 12: dst = MLW_AT_ONE; break;   case 13: dst = MLW_AT_F00; break;   default: break;
        }
        srn(V, t, sc * variant);
        if (qd < 14u) st14(vf, dst + 4u * vr, V);
    }
    __syncthreads();
    // the wavefront's program: word pc + 2 is fetched (a scalar load) while step pc runs, the lanes' records of step pc + 1 as well
    const uint32_t* prog = W == 2 ? (wave ? MLW_PROG_CHAIN : MLW_PROG_ACC) : (wave == 0u ? MLW3_PROG_A : (wave == 1u ? MLW3_PROG_B : MLW3_PROG_C));
    uint32_t pc = 0;
    uint32_t k1 = prog[0], k2 = prog[1];
    Rec r1 = load_rec(k1 & 0x3Fu, lane);
#pragma unroll 1
    for (uint32_t ph = 0; ph < (uint32_t)(W == 2 ? MLW_PHASES : MLW3_PHASES); ph++) {
#pragma unroll 1
        while (true) {
            const uint32_t k = k1;
            const Rec r = r1;
            k1 = k2;
            k2 = prog[pc + 2];
            pc++;
            r1 = load_rec(k1 & 0x3Fu, lane);
            if ((k & 0x3Fu) != (uint32_t)MLW_NOP) {
                const uint32_t shape = (k >> 8) & 3u;
                if (W == 3 && shape == 3u) wstep<1, false, true>(vf, r);
                else if (shape == 2u || W == 3) wstep<1, false>(vf, r);          // (the three-wavefront programs have no two-product steps)
                else if (shape == 1u) wstep<2, true>(vf, r);
                else wstep<2, false>(vf, r);
            }
            if (k & (uint32_t)MLW_LAST) break;
        }
        __syncthreads();
    }
    if (wave == CHAIN) {
        // the fast formulas are the reference's value iff Q is on the twist, the chain did not end at Z = 0 and Q is not flagged
        const bool on_twist = stored_zero(vf, MLW_AT_D0) && stored_zero(vf, MLW_AT_D1);
        const bool z_zero = stored_zero(vf, MLW_AT_Z0) && stored_zero(vf, MLW_AT_Z1);
        if (lane == 0u && (!on_twist || z_zero || q_flagged(dg, pair))) bad_flag = 1;
    }
    __syncthreads();
    if (wave == 0u && active) {
        const uint32_t quad = lane >> 2;
        if (quad < 12u && (lane & 3u) == 0u) {
            const uint32_t k = quad >> 1, part = quad & 1u;
            const uint32_t flat = (k & 1u) ? 3u + (k >> 1) : (k >> 1);             // w-powers 0,2,4,1,3,5 in the flat order
            fe a;
#pragma unroll
            for (int j = 0; j < NL; j++) a.v[j] = *reinterpret_cast<const int32_t*>(vf + (W == 2 ? MLW_AT_F00 : MLW3_AT_F_FINAL) + 4u * lane + 256 * j);
            uint32_t w[12];
            r28::to_vm(w, a);
            uint32_t* o = partials + (size_t)pair * 144 + flat * 24u + part * 12u;
#pragma unroll
            for (int j = 0; j < 12; j++) o[j] = w[j];
        }
        if (lane == 0u && bad_flag) {
            const uint32_t at = atomicAdd(dg.count, 1u);
            dg.blocks[at] = pair;
        }
    }
}
#else
;
#endif
// every instantiation the host side launches: this translation unit is the one that emits them (blsgpu_tu.h)
#if BLSGPU_TU == BLSGPU_TU_FXW
__attribute__((used)) static const void* const blsgpu_instances_mlw[] = {(const void*)&k_miller_wide<2>, (const void*)&k_miller_wide<3>};
#endif
}  // namespace mlw
}  // namespace blsgpu
