// blsgpu_lsw.hip -- the WIDE point chains of the line-stream stage (round 5): T <- 2T (+ Q) and the line coefficients of
// fq_miller_loop (fields_t.py:1091-1111; lines :1035-1078, point steps :641-686) with SIXTEEN LANES PER PAIR, four pairs per
// wavefront -- for calls of a few thousand pairs, where one pair per lane quad (k_ml_lines4) leaves half the SIMDs empty and the
// call waits for one quad's chain of 68 steps (0.9 ms).  Included by blsgpu_api.hip after blsgpu_mlw.hip; model, formulas and
// table generator: vmgen/lsw_model.py, vmgen/gen_lsw.py (tests/test_lsw_model.py).
//
// The machine of blsgpu_mlw.hip without lane sums: every Fq value of a pair lives in LDS (limb j of slot s of pair i at dword
// 896 (s / 16) + 4 (s % 16) + i + 64 j: 64 lanes reading the slots of their own pairs never meet in a bank), a step gives every lane
// ONE output -- the sum of up to K products of sums of two slots with one Montgomery reduction, scaled, a multiple of q taken off
// inside the carry pass (mlw::srn) -- stored as itself and, where the formulas read it so, as its negative (limb-wise, no carry
// pass) and its double (a second scale-and-reduce); a coefficient 2 on a lone operand is the same slot read twice.  Twelve outputs
// per level of the tangent step (A, B, E, F, X^2, YZ; then X3, Y3, Z3 and the line), so twelve of a pair's sixteen lanes work and a
// loop iteration is two steps for FOUR pairs.  The line coefficients go straight into the pair's line record in HBM -- the records
// of k_ml_lines2 / k_ml_lines4, same field elements -- and the same flags (bad[], work list) come out, so k_ml_lines_exact,
// k_ml_accum, k_ml_small and everything behind them are unchanged.
#pragma once
#include "lsw_tables_gfx950.h"

namespace blsgpu {
namespace lsw {
using r28::fe;
using r28::NL;

constexpr int VF_DW = LSW_ROWS * 896;

template <int NW> struct RecN { uint32_t w[NW]; };
template <int K> __device__ __forceinline__ RecN<2 * K + 3> load_rec(uint32_t kind, uint32_t r) {
    RecN<2 * K + 3> o;
#pragma unroll
    for (int i = 0; i < 2 * K; i++) o.w[i] = LSW_REC[kind][i][r];
#pragma unroll
    for (int i = 0; i < 3; i++) o.w[2 * K + i] = LSW_REC[kind][8 + i][r];
    return o;
}
__device__ __forceinline__ void st14n(char* vf, uint32_t a, const int32_t* __restrict__ V) {      // the negative, limb by limb
    char* p = vf + a;
#pragma unroll
    for (int j = 0; j < NL; j++) *reinterpret_cast<int32_t*>(p + 256 * j) = -V[j];
}
// One step of a wavefront: `vf` already points at the lane's pair (value file base + 4 x pair index).  line: the pair's record of
// this line index in HBM, or nullptr.  Column bounds (units of 2^56, 8 fit): asserted per kind by the model's digit-level run; the
// formulas keep every output at <= 8 (vmgen/lsw_model.py).
template <int K, bool HAS2>
__device__ __forceinline__ void lstep(char* vf, const RecN<2 * K + 3>& r, int32_t* __restrict__ line) {
    int32_t A[K][NL], B[K][NL], p[NL], V[NL];
#pragma unroll
    for (int k = 0; k < K; k++) {
        mlw::rd2(A[k], vf, r.w[2 * k]);
        mlw::rd2(B[k], vf, r.w[2 * k + 1]);
    }
    if (K == 2) bls28::fp28_dot2(p, A[0], B[0], A[1], B[1]);
    else if (K == 3) bls28::fp28_dot3(p, A[0], B[0], A[1], B[1], A[2], B[2]);
    else bls28::fp28_dot4(p, A[0], B[0], A[1], B[1], A[2], B[2], A[3], B[3]);
    const uint32_t d1 = r.w[2 * K], d2 = r.w[2 * K + 1], sl = r.w[2 * K + 2];
    const int32_t scale = (int32_t)(sl << 16) >> 16;
    mlw::srn(V, p, scale);
    mlw::st14(vf, d1 & 0xFFFFu, V);
    st14n(vf, d1 >> 16, V);
    const uint32_t lo = (sl >> 16) & 0xFFu;
    if (line != nullptr && lo != 0xFFu) {
        int32_t* o = line + lo * NL;
#pragma unroll
        for (int j = 0; j < NL; j++) o[j] = V[j];
    }
    if (HAS2) {
        int32_t V2[NL];
        mlw::srn(V2, p, 2 * scale);
        mlw::st14(vf, d2 & 0xFFFFu, V2);
        st14n(vf, d2 >> 16, V2);
    }
}

// Pair p = 4 x (global wavefront index) + lane / 16.  Same outputs as k_ml_lines2: lines[(L * n + p) * 84], bad[p], the work list.
__global__ void __launch_bounds__(256, 2) k_ml_lines_wide(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2, uint32_t n,
                                                          int32_t* __restrict__ lines, uint8_t* __restrict__ bad, DegenList dg)
#if BLSGPU_EMIT(BLSGPU_TU_FXW)
{
    __shared__ int32_t vfiles[4][VF_DW];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t r = lane & 15u, slot = lane >> 4;
    const uint32_t pr = (blockIdx.x * 4u + wv) * 4u + slot;
    const bool active = pr < n;
    const uint32_t p = active ? pr : n - 1u;                       // the last wavefront's spare pairs repeat the last pair and write nothing
    int32_t* vfile = vfiles[wv];
    char* vf = reinterpret_cast<char*>(vfile) + 4u * slot;
    for (uint32_t i = lane; i < (uint32_t)VF_DW; i += 64u) vfile[i] = 0;
    {
        // inputs: lane r of a pair reads one source and stores it (times a scale, in the multiples the formulas read) at up to two values
        const uint32_t src = LSW_IN[0][r];
        const uint32_t* s1 = g1 + (size_t)p * 24;
        const uint32_t* s2 = g2 + (size_t)p * 48;
        const uint32_t* sp = src == 0u ? s1 : (src == 1u ? s1 + 12 : s2 + 12u * ((src - 2u) & 3u));
        const fe x = ml::load_coord(sp);
        const int32_t one[NL] = BLS28_ONE;
        int32_t t[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) t[j] = src == 6u ? one[j] : (src == 7u ? 0 : x.v[j]);
        __syncthreads();                                           // the zeroing is done before anything is stored
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int32_t scale = (int32_t)(LSW_IN[1 + 3 * e][r] << 16) >> 16;
            const uint32_t d1 = LSW_IN[2 + 3 * e][r], d2 = LSW_IN[3 + 3 * e][r];
            int32_t V[NL];
            mlw::srn(V, t, scale);
            mlw::st14(vf, d1 & 0xFFFFu, V);
            st14n(vf, d1 >> 16, V);
            mlw::srn(V, t, 2 * scale);
            mlw::st14(vf, d2 & 0xFFFFu, V);
            st14n(vf, d2 >> 16, V);
        }
    }
    // Q on the twist: D = yq^2 - xq^3 - 4 (1 + u) = 0
    lstep<2, false>(vf, load_rec<2>(LSW_K_CK1, r), nullptr);
    lstep<4, false>(vf, load_rec<4>(LSW_K_CK2, r), nullptr);
    bool ok = !q_flagged(dg, p) && mlw::stored_zero(vf, LSW_AT_D0) && mlw::stored_zero(vf, LSW_AT_D1);
    const RecN<7> rl1 = load_rec<2>(LSW_K_L1, r);
    const RecN<9> rl2 = load_rec<3>(LSW_K_L2, r);
    int32_t* rec = active ? lines + (size_t)p * ml::LINE_DW : nullptr;
    const size_t lstride = (size_t)n * ml::LINE_DW;
#pragma unroll 1
    for (int bit = 62; bit >= 0; bit--) {
        lstep<2, true>(vf, rl1, nullptr);
        lstep<3, true>(vf, rl2, rec);
        if (rec) rec += lstride;
        if ((ml::ML_NX >> bit) & 1ull) {
            lstep<3, false>(vf, load_rec<3>(LSW_K_C1, r), nullptr);
            lstep<4, false>(vf, load_rec<4>(LSW_K_C2, r), rec);
            lstep<4, false>(vf, load_rec<4>(LSW_K_C3, r), nullptr);
            lstep<4, true>(vf, load_rec<4>(LSW_K_C4, r), nullptr);
            if (rec) rec += lstride;
        }
    }
    ok = ok && !(mlw::stored_zero(vf, LSW_AT_Z0) && mlw::stored_zero(vf, LSW_AT_Z1));
    if (r == 0u && active) {
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
}  // namespace lsw
}  // namespace blsgpu
