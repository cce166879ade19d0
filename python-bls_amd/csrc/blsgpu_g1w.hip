// blsgpu_g1w.hip -- the WIDE tail of the sorted-bucket G1 sum (round 5): result = sum_i 2^(c i) P_i over a short list of
// projective G1 points, Horner from the top -- the window sums W_w = sum_b 2^b S_(w,b) (c = 1, one wavefront per window) and the
// sum over the windows sum_w 2^(13 w) W_w (c = 13, one wavefront) of BLS.aggregate_pub_keys(secure) at scale (bls.py:203-223; the
// reference's double-and-add summed over the points, fields_t.py:705-740).  247 doublings and 19 additions of ONE point are a
// dependent chain; the wavefront VM ran it on one team with every linear combination a round of its own (k_msm_pip_horner<1>:
// 1.31 ms of a 6.2 ms sum, k_srt_windows another 0.14).  Included by blsgpu_api.hip after blsgpu_mlw.hip.
//
// The machine is k_miller_wide's (blsgpu_mlw.hip: the LDS value file, mlw::wstep<1>: one product per lane, quad sum, scale, a
// multiple of q taken off in the carry pass) and the formulas are DATA (g1w_tables_gfx950.h from vmgen/g1w_model.py): the complete
// doubling and the complete addition of Renes-Costello-Batina for a = 0, b = 4 in two steps each -- infinity (0 : 1 : 0) anywhere
// in the list, a doubling inside an addition and P + (-P) need no branch, as in csrc/fp28.h pdbl / padd, whose results these are
// as projective points.  tests/test_g1w_model.py runs the tables digit by digit against the host's integer curve arithmetic;
// tests/test_gpu_msm.py runs the kernel behind every large G1 sum against the reference's sums.
#pragma once
#include "g1w_tables_gfx950.h"

namespace blsgpu {
namespace g1w {
using r28::fe;
using r28::NL;

constexpr int VF_DW = G1W_PAGES * MLW_PAGE_BYTES / 4;
constexpr uint32_t PJ_DW = 3 * NL;                          // a projective point in the L28 form (blsgpu_msm.hip L28_PJ)

__device__ __forceinline__ mlw::Rec load_rec(uint32_t kind, uint32_t lane) {
    mlw::Rec r;
#pragma unroll
    for (int i = 0; i < 5; i++) r.w[i] = G1W_REC[kind][i][lane];
    return r;
}
__device__ __forceinline__ fe ld_fe(const char* vf, uint32_t a) {
    fe x;
#pragma unroll
    for (int j = 0; j < NL; j++) x.v[j] = *reinterpret_cast<const int32_t*>(vf + a + 256 * j);
    return x;
}

// in: gridDim.x lists of npts projective points in the L28 form (index 0 the lowest term); list g -> sum_i 2^(cbits i) in[g][i].
// AFFINE = 0: out = the sum as a projective L28 point (42 dwords per list); AFFINE = 1: out = 96 bytes canonical affine (x, y)
// big-endian per list, (0, 0) and out_inf[g] = 1 for infinity (what k_msm_pip_horner<1> writes).
template <int AFFINE>
__global__ void __launch_bounds__(64) k_msm_horner_wide(const uint32_t* __restrict__ in, uint32_t npts, uint32_t cbits, uint32_t* __restrict__ out,
                                                        uint8_t* __restrict__ out_inf)
#if BLSGPU_EMIT(BLSGPU_TU_FXW)
{
    __shared__ int32_t vfile[VF_DW];
    char* vf = reinterpret_cast<char*>(vfile);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t g = blockIdx.x;
    for (uint32_t i = lane; i < (uint32_t)VF_DW; i += 64u) vfile[i] = 0;
    // quad c < 3 stores coordinate c of a point in its four multiples
    const uint32_t qd = lane >> 2, vr = lane & 3u;
    const int32_t variant = vr == 0u ? 1 : (vr == 1u ? -1 : (vr == 2u ? 2 : -2));
    const uint32_t* P = in + (size_t)g * npts * PJ_DW + (qd < 3u ? qd : 0u) * NL;
    int32_t t[NL], V[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) t[j] = (int32_t)P[(size_t)(npts - 1u) * PJ_DW + j];
    mlw::srn(V, t, variant);
    __syncthreads();                                              // (one wavefront: the zeroing above is done before anything is stored)
    if (qd < 3u) mlw::st14(vf, (uint32_t)G1W_AT_AX + 16u * qd + 4u * vr, V);
    const mlw::Rec d1 = load_rec(G1W_DBL1, lane), d2 = load_rec(G1W_DBL2, lane), a1 = load_rec(G1W_ADD1, lane), a2 = load_rec(G1W_ADD2, lane);
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
        if (qd < 3u) mlw::st14(vf, (uint32_t)G1W_AT_SX + 16u * qd + 4u * vr, V);
        mlw::wstep<1, false>(vf, a1);
        mlw::wstep<1, false>(vf, a2);
    }
    if (!AFFINE) {
        if (qd < 3u && vr == 0u) {
            const fe c = ld_fe(vf, (uint32_t)G1W_AT_AX + 16u * qd);
#pragma unroll
            for (int j = 0; j < NL; j++) out[(size_t)g * PJ_DW + qd * NL + j] = (uint32_t)c.v[j];
        }
        return;
    }
    // affine: (X, Y) / Z, Z = 0 gives (0, 0).  Every lane holds the same Z: the variable-time division steps of fq32.h
    // (data-dependent control flow is free when the data is wave-uniform)
    const fe z = ld_fe(vf, (uint32_t)G1W_AT_AZ);
    uint32_t zv[12], ziv[12];
    r28::to_vm(zv, z);
    bls::fq_inv_var(ziv, zv);
    const fe zi = r28::from_vm(ziv);
    const fe c = ld_fe(vf, (uint32_t)G1W_AT_AX + 16u * (lane & 1u));
    uint32_t y[12];
    r28::to_raw(y, r28::mul(c, zi));
    uint32_t any = 0;
#pragma unroll
    for (int w = 0; w < 12; w++) {
        any |= y[w];
        if (lane < 2u) out[(size_t)g * 24 + lane * 12u + w] = bswap32(y[11 - w]);
    }
    const uint64_t nz = __ballot(any != 0u && lane < 2u);
    if (out_inf && lane == 0u) out_inf[g] = nz == 0 ? 1 : 0;
}
#else
;
#endif

#if BLSGPU_TU == BLSGPU_TU_FXW
__attribute__((used)) static const void* const blsgpu_instances_g1w[] = {(const void*)&k_msm_horner_wide<0>, (const void*)&k_msm_horner_wide<1>};
#endif
}  // namespace g1w
}  // namespace blsgpu
