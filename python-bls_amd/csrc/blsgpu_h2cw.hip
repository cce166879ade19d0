// blsgpu_h2cw.hip -- the WIDE cofactor clearing of hash-to-G2 (round 5): the last stage of hash_to_point_prehashed_Fq2
// (ec.py:528-550: S0 + S1, then [x^2 - x - 1] P + [x - 1] psi(P) + psi^2(2 P), ec.py:536-550) for ONE message on ONE wavefront
// with a field product per lane -- the latency form for BLS.verify of a single signature (bls.py:194-195), where the
// wavefront VM's k_h2c_clear took 1.1 ms whatever the count.  Included by blsgpu_api.hip after blsgpu_mlw.hip and blsgpu_h2c.hip.
//
// The machine is k_miller_wide's (blsgpu_mlw.hip: the LDS value file, mlw::wstep) and the program is data: the op script of the
// register kernels (BLS28_H2C_OPS) compiled by vmgen/h2cw_model.py into steps -- a complete homogeneous doubling is two steps (the
// Miller loop's tangent step without its line), a complete addition (Renes-Costello-Batina algorithm 7) two, psi one -- and copies
// between the accumulator's and a slot point's 24 value slots (h2cw_tables_gfx950.h).  tests/test_h2cw_model.py runs the tables
// digit by digit against the host's integer code (pinned to the reference's vectors); tests/test_gpu_h2c_forced.py runs this
// kernel over every reference-generated hash-to-G2 fixture.
#pragma once
#include "h2cw_tables_gfx950.h"

namespace blsgpu {
namespace h2cw {
using r28::fe;
using r28::NL;

constexpr int VF_DW = H2CW_PAGES * MLW_PAGE_BYTES / 4;

__device__ __forceinline__ mlw::Rec load_rec(uint32_t kind, uint32_t lane) {
    mlw::Rec r;
    const uint32_t k = kind < (uint32_t)H2CW_KINDS ? kind : 0u;
#pragma unroll
    for (int i = 0; i < 5; i++) r.w[i] = H2CW_REC[k][i][lane];
    return r;
}
__device__ __forceinline__ fe ld_fe(const char* vf, uint32_t a) {
    fe x;
#pragma unroll
    for (int j = 0; j < NL; j++) x.v[j] = *reinterpret_cast<const int32_t*>(vf + a + 256 * j);
    return x;
}

// enc = the stage image (see k_h2c_stage): encoding e sits in team e / NE, slots S + 5 (e % NE) .. + 5 (X.re, X.im, Y.re, Y.im, Z.re
// in the wavefront VM's form, 12 words x 2^384 each).  out: n_msg x 192 bytes canonical affine, (0, 0) for infinity.
__global__ void __launch_bounds__(64) k_h2c_clear_wide(VmTables T, const uint32_t* __restrict__ enc, uint32_t n_msg, uint32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_FXW)
{
    __shared__ int32_t vfile[VF_DW];
    char* vf = reinterpret_cast<char*>(vfile);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t m = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < (uint32_t)VF_DW; i += 64u) vfile[i] = 0;
    {
        // the inputs in their four multiples: quad i stores value i of  S0 (X.re X.im Y.re Y.im Z.re) | S1 (the same) | psi_x, psi_y | 1
        const uint32_t qd = lane >> 2, vr = lane & 3u;
        const int32_t variant = vr == 0u ? 1 : (vr == 1u ? -1 : (vr == 2u ? 2 : -2));
        constexpr uint32_t PSIX = BLSVM_HC_PSIX - BLSVM_HC_SLOT0 + BLSVM_HC_TBL0, PSIY = BLSVM_HC_PSIY - BLSVM_HC_SLOT0 + BLSVM_HC_TBL0;
        const uint32_t s = qd < 5u ? 0u : 1u, c = qd < 5u ? qd : (qd < 10u ? qd - 5u : 0u);
        const uint32_t e = 2u * m + s;
        const uint32_t* src = enc + ((size_t)(e / BLSVM_H1_NE) * H1_IMG + (BLSVM_H1_S - BLSVM_H1_STATE0) + 5 * (e % BLSVM_H1_NE)) * 12 + 12u * c;
        if (qd >= 10u && qd < 12u) src = T.consts + (PSIX * 12u) + 12u * (qd - 10u);
        if (qd >= 12u && qd < 14u) src = T.consts + (PSIY * 12u) + 12u * (qd - 12u);
        uint32_t w[12];
#pragma unroll
        for (int j = 0; j < 12; j++) w[j] = src[j];
        const fe x = r28::from_vm(w);
        const int32_t one[NL] = BLS28_ONE;
        int32_t t[NL], V[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) t[j] = qd == 14u ? one[j] : x.v[j];
        uint32_t dst = (uint32_t)H2CW_AT_TRASH;
        if (qd < 10u) dst = H2CW_POINT[s] + 16u * c;
        else if (qd < 12u) dst = (uint32_t)H2CW_AT_PSIX0 + 16u * (qd - 10u);
        else if (qd < 14u) dst = (uint32_t)H2CW_AT_PSIY0 + 16u * (qd - 12u);
        else if (qd == 14u) dst = (uint32_t)H2CW_AT_ONE;
        mlw::srn(V, t, variant);
        __syncthreads();                                          // (one wavefront: the zeroing above is done before anything is stored)
        if (qd < 15u) mlw::st14(vf, dst + 4u * vr, V);
    }
    uint32_t pc = 0;
    uint32_t w1 = H2CW_PROG[0], w2 = H2CW_PROG[1];
    mlw::Rec r1 = load_rec(w1 & 0x3Fu, lane);
#pragma unroll 1
    while (true) {
        const uint32_t w = w1;
        const mlw::Rec r = r1;
        w1 = w2;
        w2 = H2CW_PROG[pc + 2];
        pc++;
        r1 = load_rec((w1 & (uint32_t)H2CW_COPY) ? 0u : (w1 & 0x3Fu), lane);
        if (w == (uint32_t)H2CW_END) break;
        if (w & (uint32_t)H2CW_COPY) {
            const uint32_t src = H2CW_POINT[(w >> 4) & 0xFu], dst = H2CW_POINT[w & 0xFu];
            if (lane < 24u) {
                int32_t V[NL];
                mlw::rd1(V, vf, src + 4u * lane);
                mlw::st14(vf, dst + 4u * lane, V);
            }
        } else if (((w >> 8) & 3u) == 2u) {
            mlw::wstep<1, false>(vf, r);
        } else if (((w >> 8) & 3u) == 1u) {
            mlw::wstep<2, true>(vf, r);
        } else {
            mlw::wstep<2, false>(vf, r);
        }
    }
    // affine: (X, Y) / Z with 1 / Z = conj(Z) / N(Z); Z = 0 gives (0, 0).  Every lane holds the same norm: the variable-time
    // division steps of fq32.h (data-dependent control flow is free when the data is wave-uniform).
    const uint32_t A0 = H2CW_POINT[0];
    const fe z0 = ld_fe(vf, A0 + 64u), z1 = ld_fe(vf, A0 + 80u);
    const fe n = r28::dot2(z0, z0, z1, z1);
    uint32_t nv[12], niv[12];
    r28::to_vm(nv, n);
    bls::fq_inv_var(niv, nv);
    const fe ninv = r28::from_vm(niv);
    const fe zi0 = r28::mul(z0, ninv), zi1 = r28::mul(r28::neg(z1), ninv);
    if (lane < 4u) {
        // lane k: part k of (x.re, x.im, y.re, y.im):  re = a0 zi0 - a1 zi1,  im = a0 zi1 + a1 zi0
        const uint32_t base = A0 + ((lane & 2u) ? 32u : 0u);
        const fe a0 = ld_fe(vf, base), a1 = ld_fe(vf, base + 16u);
        const bool im = (lane & 1u) != 0u;
        fe p, q;
        const fe na1 = r28::norm(r28::neg(a1));
#pragma unroll
        for (int j = 0; j < NL; j++) { p.v[j] = im ? zi1.v[j] : zi0.v[j]; q.v[j] = im ? zi0.v[j] : zi1.v[j]; }
        fe b1;
#pragma unroll
        for (int j = 0; j < NL; j++) b1.v[j] = im ? a1.v[j] : na1.v[j];
        const fe o = r28::dot2(a0, p, b1, q);
        uint32_t y[12];
        r28::to_raw(y, o);
        if (m < n_msg) {
#pragma unroll
            for (int wd = 0; wd < 12; wd++) out[(size_t)m * 48 + lane * 12u + wd] = bswap32(y[11 - wd]);
        }
    }
}
#else
;
#endif

// ---- the window Horner of a G2 sum on the same tables (round 5) -------------------------------------------------------------------
// result = sum_i 2^(c i) P_i over a short list of projective G2 points, Horner from the top: the tail of the sorted-bucket G2 sum
// (BLS.aggregate_sigs(secure), bls.py:225-261, as one multi-scalar sum) -- the last folds (c = 0), the window sums (c = 1, one
// wavefront per window) and the sum over the windows (c = 13) -- what blsgpu_g1w.hip's k_msm_horner_wide is for G1.  The
// accumulator is point 0 of the clearing's value file, the addend slot point 0, the steps its DBL1 / DBL2 / ADD1_0p / ADD2 (complete
// formulas: infinity anywhere, a doubling inside an addition, P + (-P) need no branch); vmgen/h2cw_model.horner is this loop on the
// tables (tests/test_h2cw_model.py).
// in: gridDim.x lists of npts projective points in the L28 form (X.re X.im Y.re Y.im Z.re Z.im, 84 dwords; index 0 the lowest term).
// AFFINE = 0: out = the sum in the same form; AFFINE = 1: out = 192 bytes canonical affine (x.c0, x.c1, y.c0, y.c1 big-endian), (0, 0)
// and out_inf[g] = 1 for infinity.
template <int AFFINE>
__global__ void __launch_bounds__(64) k_msm_horner_wide2(const uint32_t* __restrict__ in, uint32_t npts, uint32_t cbits, uint32_t* __restrict__ out,
                                                         uint8_t* __restrict__ out_inf)
#if BLSGPU_EMIT(BLSGPU_TU_FXW)
{
    constexpr uint32_t PJ_DW = 6 * NL;
    __shared__ int32_t vfile[VF_DW];
    char* vf = reinterpret_cast<char*>(vfile);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = blockIdx.x;
    for (uint32_t i = lane; i < (uint32_t)VF_DW; i += 64u) vfile[i] = 0;
    // quad c < 6 stores value c of a point (X.re X.im Y.re Y.im Z.re Z.im) in its four multiples
    const uint32_t qd = lane >> 2, vr = lane & 3u;
    const int32_t variant = vr == 0u ? 1 : (vr == 1u ? -1 : (vr == 2u ? 2 : -2));
    const uint32_t* P = in + (size_t)g * npts * PJ_DW + (qd < 6u ? qd : 0u) * NL;
    int32_t t[NL], V[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) t[j] = (int32_t)P[(size_t)(npts - 1u) * PJ_DW + j];
    mlw::srn(V, t, variant);
    __syncthreads();                                              // (one wavefront: the zeroing above is done before anything is stored)
    const uint32_t A0 = H2CW_POINT[0], S0 = H2CW_POINT[1];
    if (qd < 6u) mlw::st14(vf, A0 + 16u * qd + 4u * vr, V);
    const mlw::Rec d1 = load_rec(H2CW_KIND_DBL1, lane), d2 = load_rec(H2CW_KIND_DBL2, lane), a1 = load_rec(H2CW_KIND_ADD1_0p, lane),
                   a2 = load_rec(H2CW_KIND_ADD2, lane);
#pragma unroll 1
    for (int i = (int)npts - 2; i >= 0; i--) {
#pragma unroll
        for (int j = 0; j < NL; j++) t[j] = (int32_t)P[(size_t)i * PJ_DW + j];            // (in flight behind the doublings)
#pragma unroll 1
        for (uint32_t s = 0; s < cbits; s++) {
            mlw::wstep<1, false>(vf, d1);
            mlw::wstep<1, false>(vf, d2);
        }
        mlw::srn(V, t, variant);
        if (qd < 6u) mlw::st14(vf, S0 + 16u * qd + 4u * vr, V);
        mlw::wstep<1, false>(vf, a1);
        mlw::wstep<1, false>(vf, a2);
    }
    if (!AFFINE) {
        if (qd < 6u && vr == 0u) {
            const fe c = ld_fe(vf, A0 + 16u * qd);
#pragma unroll
            for (int j = 0; j < NL; j++) out[(size_t)g * PJ_DW + qd * NL + j] = (uint32_t)c.v[j];
        }
        return;
    }
    // affine: (X, Y) / Z with 1 / Z = conj(Z) / N(Z); Z = 0 gives (0, 0) (the tail of k_h2c_clear_wide)
    const fe z0 = ld_fe(vf, A0 + 64u), z1 = ld_fe(vf, A0 + 80u);
    const fe n = r28::dot2(z0, z0, z1, z1);
    uint32_t nv[12], niv[12];
    r28::to_vm(nv, n);
    bls::fq_inv_var(niv, nv);
    const fe ninv = r28::from_vm(niv);
    const fe zi0 = r28::mul(z0, ninv), zi1 = r28::mul(r28::neg(z1), ninv);
    const uint32_t lk = lane & 3u;                                // lane k < 4: part k of (x.re, x.im, y.re, y.im)
    const uint32_t base = A0 + ((lk & 2u) ? 32u : 0u);
    const fe c0 = ld_fe(vf, base), c1 = ld_fe(vf, base + 16u);
    const bool im = (lk & 1u) != 0u;
    const fe nc1 = r28::norm(r28::neg(c1));
    fe p, q, b1;
#pragma unroll
    for (int j = 0; j < NL; j++) { p.v[j] = im ? zi1.v[j] : zi0.v[j]; q.v[j] = im ? zi0.v[j] : zi1.v[j]; b1.v[j] = im ? c1.v[j] : nc1.v[j]; }
    uint32_t y[12];
    r28::to_raw(y, r28::dot2(c0, p, b1, q));
    uint32_t any = 0;
#pragma unroll
    for (int wd = 0; wd < 12; wd++) {
        any |= y[wd];
        if (lane < 4u) out[(size_t)g * 48 + lane * 12u + wd] = bswap32(y[11 - wd]);
    }
    const uint64_t nz = __ballot(any != 0u && lane < 4u);
    if (out_inf && lane == 0u) out_inf[g] = nz == 0 ? 1 : 0;
}
#else
;
#endif
#if BLSGPU_TU == BLSGPU_TU_FXW
__attribute__((used)) static const void* const blsgpu_instances_h2cw[] = {(const void*)&k_msm_horner_wide2<0>, (const void*)&k_msm_horner_wide2<1>};
#endif
}  // namespace h2cw
}  // namespace blsgpu
