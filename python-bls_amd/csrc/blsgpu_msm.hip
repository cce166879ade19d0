// blsgpu_msm.hip -- G1 / G2 scalar multiplication and multi-scalar sums on the
// same wavefront-team VM as the pairing (included by blsgpu_api.hip).
//
// Replaces the reference's fq_scalar_mult_jacobian / fq2_scalar_mult_jacobian
// + *_add_points_jacobian loops (fields_t.py:705-740, 762-875; native
// fields_t_c.pyx:1368-1413) as used by BLS.aggregate_pub_keys (bls.py:203-223),
// aggregate_sigs* (bls.py:12-151) and Threshold.aggregate_unit_sigs
// (threshold.py:127-136).  Same LSB-first double-and-add; complete projective
// formulas (vmgen/msm_programs.py); parity on the affine output.
#pragma once

namespace blsgpu {

template <int DEG> struct MsmCfg;
template <> struct MsmCfg<1> {
    static constexpr int NP = BLSVM_MSM1_NP, IN = BLSVM_MSM1_IN, R = BLSVM_MSM1_R, A = BLSVM_MSM1_A,
                         S = BLSVM_MSM1_S, PR0 = BLSVM_MSM1_PR0, PR1 = BLSVM_MSM1_PR1, OUT = BLSVM_MSM1_OUT;
    static constexpr int LOAD_OFF = BLSVM_SEGF_G1_LOAD_OFF, LOAD_LEN = BLSVM_SEGF_G1_LOAD_LEN,
                         STEP_OFF = BLSVM_SEGF_G1_STEP_OFF, STEP_LEN = BLSVM_SEGF_G1_STEP_LEN,
                         FOLD_OFF = BLSVM_SEGF_G1_FOLD_OFF, FOLD_LEN = BLSVM_SEGF_G1_FOLD_LEN,
                         PADD_OFF = BLSVM_SEGF_G1_PADD_OFF, PADD_LEN = BLSVM_SEGF_G1_PADD_LEN,
                         AFF_OFF = BLSVM_SEGF_G1_AFFINE_OFF, AFF_LEN = BLSVM_SEGF_G1_AFFINE_LEN,
                         ACC_OFF = BLSVM_SEGF_G1_ACC_OFF, ACC_LEN = BLSVM_SEGF_G1_ACC_LEN,
                         DBL_OFF = BLSVM_SEGF_G1_DBL_OFF, DBL_LEN = BLSVM_SEGF_G1_DBL_LEN;
};
template <> struct MsmCfg<2> {
    static constexpr int NP = BLSVM_MSM2_NP, IN = BLSVM_MSM2_IN, R = BLSVM_MSM2_R, A = BLSVM_MSM2_A,
                         S = BLSVM_MSM2_S, PR0 = BLSVM_MSM2_PR0, PR1 = BLSVM_MSM2_PR1, OUT = BLSVM_MSM2_OUT;
    static constexpr int LOAD_OFF = BLSVM_SEGF_G2_LOAD_OFF, LOAD_LEN = BLSVM_SEGF_G2_LOAD_LEN,
                         STEP_OFF = BLSVM_SEGF_G2_STEP_OFF, STEP_LEN = BLSVM_SEGF_G2_STEP_LEN,
                         FOLD_OFF = BLSVM_SEGF_G2_FOLD_OFF, FOLD_LEN = BLSVM_SEGF_G2_FOLD_LEN,
                         PADD_OFF = BLSVM_SEGF_G2_PADD_OFF, PADD_LEN = BLSVM_SEGF_G2_PADD_LEN,
                         AFF_OFF = BLSVM_SEGF_G2_AFFINE_OFF, AFF_LEN = BLSVM_SEGF_G2_AFFINE_LEN,
                         ACC_OFF = BLSVM_SEGF_G2_ACC_OFF, ACC_LEN = BLSVM_SEGF_G2_ACC_LEN,
                         DBL_OFF = BLSVM_SEGF_G2_DBL_OFF, DBL_LEN = BLSVM_SEGF_G2_DBL_LEN;
};

// dword k (0 .. 36*DEG-1) of the projective point at infinity (0 : 1 : 0), Montgomery
template <int DEG>
__device__ __forceinline__ uint32_t inf_dword(const uint32_t* team, uint32_t k) {
    return (k >= 12u * DEG && k < 12u * DEG + 12u) ? team[BLSVM_SLOT_C_ONE * 12 + (k - 12u * DEG)] : 0u;
}

// Kernel A: block b handles points [lo, hi) of group g = b / bpg (chunk b % bpg of
// `chunk` points); its teams walk them NP at a time; output: one projective
// Montgomery partial (36*DEG u32) per block.
//   pts:     affine big-endian coordinates, 96*DEG bytes per point ((0,0) = infinity)
//   scalars: 32 bytes big-endian per point, or nullptr for all-ones (plain sums)
template <int DEG>
__global__ void __launch_bounds__(256, 3) k_msm(VmTables T, const uint32_t* __restrict__ pts, const uint32_t* __restrict__ scalars,
                                             uint32_t k, uint32_t chunk, uint32_t bpg, uint32_t* __restrict__ partials)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    using C = MsmCfg<DEG>;
    constexpr uint32_t PT_DW = 24 * DEG;            // dwords per affine input point
    constexpr uint32_t PJ_DW = 36 * DEG;            // dwords per projective point
    uint32_t* smem = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = blockDim.x >> 6;
    uint32_t* team = smem + wave * TEAM_DW;
    const uint32_t base16 = wave * (TEAM_BYTES / 16);
    const uint32_t g = blockIdx.x / bpg, cidx = blockIdx.x % bpg;
    const uint32_t lo = g * k + cidx * chunk;
    const uint32_t hi = min(g * k + k, lo + chunk);
    team_init_consts(T, team, lane);
    wave_fence();
    for (uint32_t d = lane; d < C::NP * PJ_DW; d += 64) team[C::R * 12 + d] = inf_dword<DEG>(team, d % PJ_DW);
    wave_fence();
    for (uint32_t first = lo + wave * C::NP; first < hi; first += nwaves * C::NP) {
        const uint32_t cnt = min((uint32_t)C::NP, hi - first);
        // raw coordinates -> IN slots (limb order reversed, bytes swapped)
        for (uint32_t d = lane; d < C::NP * PT_DW; d += 64) {
            uint32_t p = d / PT_DW, o = d % PT_DW, e = o / 12, w = o % 12;
            uint32_t v = (p < cnt) ? bswap32(pts[(size_t)(first + p) * PT_DW + o]) : 0u;
            team[(C::IN + p * 2 * DEG + e) * 12 + (11 - w)] = v;
        }
        // this lane's scalar (lanes 0..NP-1), little-endian words
        uint32_t sw[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t v = 0;
            if (lane < cnt) v = scalars ? bswap32(scalars[(size_t)(first + lane) * 8 + (7 - j)]) : (j == 0 ? 1u : 0u);
            sw[j] = v;
        }
        wave_fence();
        run_rounds<true>(T, T.segflat + C::LOAD_OFF, C::LOAD_LEN, base16, lane);
        // (0,0) inputs are the point at infinity
        uint32_t zero_in = 0;
        if (lane < C::NP) {
            uint32_t acc = 0;
            for (uint32_t i = 0; i < 2u * DEG * 12u; i++) acc |= team[(C::IN + lane * 2 * DEG) * 12 + i];
            zero_in = (acc == 0u);
        }
        const uint64_t zmask = __ballot(zero_in != 0);
        for (uint32_t d = lane; d < C::NP * PJ_DW; d += 64) {
            uint32_t p = d / PJ_DW;
            if ((zmask >> p) & 1ull) team[C::A * 12 + d] = inf_dword<DEG>(team, d % PJ_DW);
        }
        // number of scalar bits to walk (wave-uniform)
        uint32_t bl = 0;
#pragma unroll
        for (int j = 7; j >= 0; j--)
            if (bl == 0 && sw[j] != 0) bl = 32u * j + (32u - __builtin_clz(sw[j]));
        for (int off = 32; off > 0; off >>= 1) bl = max(bl, (uint32_t)__shfl_xor((int)bl, off));
        bl = __builtin_amdgcn_readfirstlane(bl);
        wave_fence();
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
            const uint32_t word = sw[j];
#pragma unroll 1
            for (uint32_t b = 0; b < 32; b++) {
                const uint32_t bit = 32u * j + b;
                if (bit >= bl) break;
                const uint64_t bits = __ballot(lane < (uint32_t)C::NP && ((word >> b) & 1u));
                // S_p = bit ? A_p : infinity
                for (uint32_t d = lane; d < C::NP * PJ_DW; d += 64) {
                    uint32_t p = d / PJ_DW;
                    team[C::S * 12 + d] = ((bits >> p) & 1ull) ? team[C::A * 12 + d] : inf_dword<DEG>(team, d % PJ_DW);
                }
                wave_fence();
                run_rounds<true>(T, T.segflat + C::STEP_OFF, C::STEP_LEN, base16, lane);
            }
            if (32u * (j + 1) >= bl) break;
        }
    }
    wave_fence();
    run_rounds<true>(T, T.segflat + C::FOLD_OFF, C::FOLD_LEN, base16, lane);
    // product tree over the teams of the workgroup
    for (uint32_t s = 1; s < nwaves; s <<= 1) {
        __syncthreads();
        if ((wave % (2 * s)) == 0 && wave + s < nwaves) {
            const uint32_t* other = smem + (wave + s) * TEAM_DW + C::PR0 * 12;
            for (uint32_t i = lane; i < PJ_DW; i += 64) team[C::PR1 * 12 + i] = other[i];
            wave_fence();
            run_rounds<true>(T, T.segflat + C::PADD_OFF, C::PADD_LEN, base16, lane);
        }
    }
    if (wave == 0) {
        wave_fence();
        for (uint32_t i = lane; i < PJ_DW; i += 64) partials[(size_t)blockIdx.x * PJ_DW + i] = team[C::PR0 * 12 + i];
    }
}
#else
;
#endif

// Kernel B: one team per group: sum of the group's bpg partials, conversion to
// affine canonical bytes (x || y, (0,0) for infinity) and an infinity flag.
template <int DEG>
__global__ void __launch_bounds__(256) k_msm_finish(VmTables T, const uint32_t* __restrict__ partials, uint32_t bpg,
                                                    uint32_t groups, uint32_t* __restrict__ out, uint8_t* __restrict__ out_inf)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    using C = MsmCfg<DEG>;
    constexpr uint32_t PJ_DW = 36 * DEG;
    constexpr uint32_t PT_DW = 24 * DEG;
    uint32_t* smem = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = blockDim.x >> 6;
    uint32_t* team = smem + wave * TEAM_DW;
    const uint32_t base16 = wave * (TEAM_BYTES / 16);
    const uint32_t g = blockIdx.x * nwaves + wave;
    if (g >= groups) return;                     // no workgroup barrier below
    team_init_consts(T, team, lane);
    wave_fence();
    for (uint32_t i = 0; i < bpg; i++) {
        const uint32_t* src = partials + ((size_t)g * bpg + i) * PJ_DW;
        const uint32_t dst = (i == 0) ? C::PR0 : C::PR1;
        for (uint32_t d = lane; d < PJ_DW; d += 64) team[dst * 12 + d] = src[d];
        wave_fence();
        if (i) run_rounds<true>(T, T.segflat + C::PADD_OFF, C::PADD_LEN, base16, lane);
    }
    run_rounds(T, T.segflat + C::AFF_OFF, C::AFF_LEN, base16, lane);
    if (lane < 2u * DEG) {
        uint32_t X[12];
        lds_load12(X, base16 + (C::OUT + lane) * 3);
        bls::fq_canon(X);
        lds_store12(X, base16 + (C::OUT + lane) * 3);
    }
    wave_fence();
    uint32_t any = 0;
    for (uint32_t d = lane; d < PT_DW; d += 64) {
        uint32_t e = d / 12, w = d % 12;
        uint32_t v = team[(C::OUT + e) * 12 + (11 - w)];
        any |= v;
        out[(size_t)g * PT_DW + d] = bswap32(v);
    }
    const uint64_t nz = __ballot(any != 0);
    if (out_inf && lane == 0) out_inf[g] = (nz == 0) ? 1 : 0;
}
#else
;
#endif

// ---------------------------------------------------------------------------
// Bucket method (Pippenger) for one large sum  sum_i s_i P_i  (BLS.aggregate_pub_keys at
// scale, BASELINE config "1 M-point G1 multi-scalar-mul, Pippenger buckets in LDS").
// Same value as the double-and-add of fields_t.py:705-740 summed over i (parity is on
// the affine result).  Scalars are cut into PIP_W windows of PIP_C bits.
//
// k_msm_pip: block (chunk, window) owns, per lane group p < NP, 2^PIP_C buckets in LDS
// (bucket 0 collects the zero digits and is dropped).  Every step adds NP points to
// the buckets their digits select: the kernel copies bucket -> R_p, point (converted once by
// k_msm_prep) -> S_p, runs
// the static program g*_acc (R_p += S_p, complete formulas) and copies R_p back.
// Then sum_j j B_j by the running-sum trick (2 (2^PIP_C - 1) more g*_acc runs), the NP
// group totals are folded and the window partial of this chunk is written.
// k_msm_pip_windows: one team per window adds the chunk partials.
// k_msm_pip_horner: result = sum_w 2^(PIP_C w) W_w, affine, canonical bytes.
constexpr int PIP_C = 4;
constexpr int PIP_W = 64;                                    // 256 / PIP_C
constexpr int PIP_NB = 1 << PIP_C;

template <int DEG> struct PipCfg {
    using C = MsmCfg<DEG>;
    static constexpr int BUCKET0 = BLSVM_TEAM_SLOTS;         // above every program's temporaries
    static constexpr int SLOTS = BUCKET0 + C::NP * PIP_NB * 3 * DEG;
};

// affine big-endian points -> projective Montgomery triples (36*DEG u32 each, (0:1:0) for
// the (0,0) encoding of infinity), once, so that the 64 window passes need no conversion
template <int DEG>
__global__ void __launch_bounds__(256, 3) k_msm_prep(VmTables T, const uint32_t* __restrict__ pts, uint32_t k, uint32_t* __restrict__ prep)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    using C = MsmCfg<DEG>;
    constexpr uint32_t PT_DW = 24 * DEG, PJ_DW = 36 * DEG;
    uint32_t* smem = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* team = smem + wave * TEAM_DW;
    const uint32_t base16 = wave * (TEAM_BYTES / 16);
    const uint32_t first = (blockIdx.x * (blockDim.x >> 6) + wave) * C::NP;
    if (first >= k) return;                                   // no workgroup barrier below
    const uint32_t cnt = min((uint32_t)C::NP, k - first);
    team_init_consts(T, team, lane);
    for (uint32_t d = lane; d < C::NP * PT_DW; d += 64) {
        uint32_t p = d / PT_DW, o = d % PT_DW, e = o / 12, w = o % 12;
        uint32_t v = (p < cnt) ? bswap32(pts[(size_t)(first + p) * PT_DW + o]) : 0u;
        team[(C::IN + p * 2 * DEG + e) * 12 + (11 - w)] = v;
    }
    wave_fence();
    run_rounds<true>(T, T.segflat + C::LOAD_OFF, C::LOAD_LEN, base16, lane);
    uint32_t zero_in = 0;
    if (lane < C::NP) {
        uint32_t acc = 0;
        for (uint32_t i = 0; i < 2u * DEG * 12u; i++) acc |= team[(C::IN + lane * 2 * DEG) * 12 + i];
        zero_in = (acc == 0u);
    }
    const uint64_t zmask = __ballot(zero_in != 0);
    for (uint32_t d = lane; d < cnt * PJ_DW; d += 64) {
        const uint32_t p = d / PJ_DW;
        prep[(size_t)first * PJ_DW + d] = ((zmask >> p) & 1ull) ? inf_dword<DEG>(team, d % PJ_DW) : team[C::A * 12 + d];
    }
}
#else
;
#endif

template <int DEG>
__global__ void __launch_bounds__(64, 2) k_msm_pip(VmTables T, const uint32_t* __restrict__ prep, const uint32_t* __restrict__ scalars,
                                                   uint32_t k, uint32_t chunk, uint32_t* __restrict__ partials)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    using C = MsmCfg<DEG>;
    using P = PipCfg<DEG>;
    constexpr uint32_t PJ_DW = 36 * DEG;
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t win = blockIdx.y;
    const uint32_t grp = blockIdx.z;                           // independent sums: group g owns points [g k, (g + 1) k)
    const uint32_t lo = grp * k + blockIdx.x * chunk, hi = min(grp * k + k, lo + chunk);
    team_init_consts(T, team, lane);
    wave_fence();
    for (uint32_t d = lane; d < C::NP * PIP_NB * PJ_DW; d += 64) team[P::BUCKET0 * 12 + d] = inf_dword<DEG>(team, d % PJ_DW);
    wave_fence();
    auto bucket = [&](uint32_t p, uint32_t j) { return (uint32_t)(P::BUCKET0 * 12) + (p * PIP_NB + j) * PJ_DW; };
    for (uint32_t first = lo; first < hi; first += C::NP) {
        const uint32_t cnt = min((uint32_t)C::NP, hi - first);
        // digit of this window: bits [PIP_C win, PIP_C win + PIP_C) of the big-endian scalar
        uint32_t dig = 0;
        if (lane < cnt) {
            if (scalars) {
                const uint32_t word = bswap32(scalars[(size_t)(first + lane) * 8 + (7 - win / 8)]);
                dig = (word >> (4 * (win % 8))) & 15u;
            } else {
                dig = (win == 0) ? 1u : 0u;
            }
        }
        // S_p = the point (infinity for (0,0) and past the end), R_p = its bucket
        uint32_t digs[C::NP];
#pragma unroll
        for (int p = 0; p < C::NP; p++) digs[p] = __builtin_amdgcn_readlane(dig, p);
        for (uint32_t d = lane; d < C::NP * PJ_DW; d += 64) {
            const uint32_t p = d / PJ_DW, o = d % PJ_DW;
            uint32_t dp = 0;
#pragma unroll
            for (int q = 0; q < C::NP; q++) dp = (p == (uint32_t)q) ? digs[q] : dp;
            team[C::S * 12 + d] = (p < cnt) ? prep[(size_t)first * PJ_DW + d] : inf_dword<DEG>(team, o);
            team[C::R * 12 + d] = team[bucket(p, dp) + o];
        }
        wave_fence();
        run_rounds<true>(T, T.segflat + C::ACC_OFF, C::ACC_LEN, 0, lane);
        for (uint32_t d = lane; d < C::NP * PJ_DW; d += 64) {
            const uint32_t p = d / PJ_DW, o = d % PJ_DW;
            uint32_t dp = 0;
#pragma unroll
            for (int q = 0; q < C::NP; q++) dp = (p == (uint32_t)q) ? digs[q] : dp;
            team[bucket(p, dp) + o] = team[C::R * 12 + d];
        }
        wave_fence();
    }
    // sum_j j B_j: acc (kept in A_p) runs down the buckets, tot (kept in bucket 0) adds acc every time
    for (uint32_t d = lane; d < C::NP * PJ_DW; d += 64) {
        const uint32_t p = d / PJ_DW, o = d % PJ_DW;
        team[C::A * 12 + d] = inf_dword<DEG>(team, o);
        team[bucket(p, 0) + o] = inf_dword<DEG>(team, o);
    }
    wave_fence();
    for (int j = PIP_NB - 1; j >= 1; j--) {
        for (uint32_t d = lane; d < C::NP * PJ_DW; d += 64) {
            const uint32_t p = d / PJ_DW, o = d % PJ_DW;
            team[C::R * 12 + d] = team[C::A * 12 + d];
            team[C::S * 12 + d] = team[bucket(p, (uint32_t)j) + o];
        }
        wave_fence();
        run_rounds<true>(T, T.segflat + C::ACC_OFF, C::ACC_LEN, 0, lane);
        for (uint32_t d = lane; d < C::NP * PJ_DW; d += 64) {
            const uint32_t p = d / PJ_DW, o = d % PJ_DW;
            const uint32_t r = team[C::R * 12 + d];
            team[C::A * 12 + d] = r;                          // acc
            team[C::S * 12 + d] = r;
            team[C::R * 12 + d] = team[bucket(p, 0) + o];     // tot
        }
        wave_fence();
        run_rounds<true>(T, T.segflat + C::ACC_OFF, C::ACC_LEN, 0, lane);
        for (uint32_t d = lane; d < C::NP * PJ_DW; d += 64) {
            const uint32_t p = d / PJ_DW, o = d % PJ_DW;
            team[bucket(p, 0) + o] = team[C::R * 12 + d];
        }
        wave_fence();
    }
    // R_p = tot_p already; fold the NP totals
    run_rounds<true>(T, T.segflat + C::FOLD_OFF, C::FOLD_LEN, 0, lane);
    uint32_t* dst = partials + (((size_t)grp * PIP_W + win) * gridDim.x + blockIdx.x) * PJ_DW;
    for (uint32_t i = lane; i < PJ_DW; i += 64) dst[i] = team[C::PR0 * 12 + i];
}
#else
;
#endif

// W_w = sum over chunks of partial[w][chunk]; one team per window
template <int DEG>
__global__ void __launch_bounds__(64) k_msm_pip_windows(VmTables T, const uint32_t* __restrict__ partials, uint32_t chunks,
                                                        uint32_t* __restrict__ winsums)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    using C = MsmCfg<DEG>;
    constexpr uint32_t PJ_DW = 36 * DEG;
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t win = blockIdx.x + blockIdx.y * PIP_W;      // blockIdx.y = group
    team_init_consts(T, team, lane);
    wave_fence();
    for (uint32_t i = 0; i < chunks; i++) {
        const uint32_t* src = partials + ((size_t)win * chunks + i) * PJ_DW;
        const uint32_t dst = (i == 0) ? C::PR0 : C::PR1;
        for (uint32_t d = lane; d < PJ_DW; d += 64) team[dst * 12 + d] = src[d];
        wave_fence();
        if (i) run_rounds<true>(T, T.segflat + C::PADD_OFF, C::PADD_LEN, 0, lane);
    }
    wave_fence();
    for (uint32_t d = lane; d < PJ_DW; d += 64) winsums[(size_t)win * PJ_DW + d] = team[C::PR0 * 12 + d];
}
#else
;
#endif

// result = sum_w 2^(cbits w) W_w over nwin windows (Horner from the top window), affine canonical bytes
template <int DEG>
__global__ void __launch_bounds__(64) k_msm_pip_horner(VmTables T, const uint32_t* __restrict__ winsums, uint32_t nwin, uint32_t cbits,
                                                       uint32_t* __restrict__ out, uint8_t* __restrict__ out_inf)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    using C = MsmCfg<DEG>;
    constexpr uint32_t PJ_DW = 36 * DEG, PT_DW = 24 * DEG;
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    team_init_consts(T, team, lane);
    wave_fence();
    winsums += (size_t)blockIdx.x * nwin * PJ_DW;               // blockIdx.x = group
    out += (size_t)blockIdx.x * PT_DW;
    for (uint32_t d = lane; d < PJ_DW; d += 64) team[C::PR0 * 12 + d] = winsums[(size_t)(nwin - 1u) * PJ_DW + d];
    wave_fence();
    for (int w = (int)nwin - 2; w >= 0; w--) {
        for (uint32_t s = 0; s < cbits; s++) run_rounds<true>(T, T.segflat + C::DBL_OFF, C::DBL_LEN, 0, lane);
        for (uint32_t d = lane; d < PJ_DW; d += 64) team[C::PR1 * 12 + d] = winsums[(size_t)w * PJ_DW + d];
        wave_fence();
        run_rounds<true>(T, T.segflat + C::PADD_OFF, C::PADD_LEN, 0, lane);
    }
    run_rounds(T, T.segflat + C::AFF_OFF, C::AFF_LEN, 0, lane);
    if (lane < 2u * DEG) {
        uint32_t X[12];
        lds_load12(X, (C::OUT + lane) * 3);
        bls::fq_canon(X);
        lds_store12(X, (C::OUT + lane) * 3);
    }
    wave_fence();
    uint32_t any = 0;
    for (uint32_t d = lane; d < PT_DW; d += 64) {
        uint32_t e = d / 12, w = d % 12;
        uint32_t v = team[(C::OUT + e) * 12 + (11 - w)];
        any |= v;
        out[d] = bswap32(v);
    }
    const uint64_t nz = __ballot(any != 0);
    if (out_inf && lane == 0) out_inf[blockIdx.x] = (nz == 0) ? 1 : 0;
}
#else
;
#endif

// The same Horner for a BATCH of G2 sums (the threshold combine of BASELINE config 4): BLSVM_HMSM2_NP sums per team in
// lock step (vmgen/msm_programs.build_horner) -- a G2 doubling is 10-12 lane operations, so one sum per team leaves
// five lanes in six idle.  winsums: groups x nwin projective points; out: groups x 192 bytes.
__global__ void __launch_bounds__(64) k_msm_horner_np(VmTables T, const uint32_t* __restrict__ winsums, uint32_t nwin, uint32_t cbits,
                                                      uint32_t groups, uint32_t* __restrict__ out, uint8_t* __restrict__ out_inf)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    constexpr uint32_t DEG = 2, PJ_DW = 36 * DEG, PT_DW = 24 * DEG, NP = BLSVM_HMSM2_NP;
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t first = blockIdx.x * NP;
    const uint32_t cnt = min(NP, groups - first);
    team_init_consts(T, team, lane);
    wave_fence();
    auto load = [&](uint32_t slot0, uint32_t w) {            // point w of every sum of the team (infinity past the end)
        for (uint32_t d = lane; d < NP * PJ_DW; d += 64) {
            const uint32_t p = d / PJ_DW, o = d % PJ_DW;
            team[slot0 * 12 + d] = (p < cnt) ? winsums[((size_t)(first + p) * nwin + w) * PJ_DW + o] : inf_dword<DEG>(team, o);
        }
        wave_fence();
    };
    load(BLSVM_HMSM2_R, nwin - 1u);
    for (int w = (int)nwin - 2; w >= 0; w--) {
        for (uint32_t s = 0; s < cbits; s++) run_rounds<true>(T, T.segflat + BLSVM_SEGF_G2H_DBL_OFF, BLSVM_SEGF_G2H_DBL_LEN, 0, lane);
        wave_fence();
        load(BLSVM_HMSM2_S, (uint32_t)w);
        run_rounds<true>(T, T.segflat + BLSVM_SEGF_G2H_ACC_OFF, BLSVM_SEGF_G2H_ACC_LEN, 0, lane);
    }
    run_rounds(T, T.segflat + BLSVM_SEGF_G2H_AFFINE_OFF, BLSVM_SEGF_G2H_AFFINE_LEN, 0, lane);
    if (lane < NP * 2u * DEG) {
        uint32_t X[12];
        lds_load12(X, (BLSVM_HMSM2_OUT + lane) * 3);
        bls::fq_canon(X);
        lds_store12(X, (BLSVM_HMSM2_OUT + lane) * 3);
    }
    wave_fence();
    for (uint32_t p = 0; p < cnt; p++) {
        uint32_t any = 0;
        for (uint32_t d = lane; d < PT_DW; d += 64) {
            const uint32_t e = d / 12, w = d % 12;
            const uint32_t v = team[(BLSVM_HMSM2_OUT + p * 2 * DEG + e) * 12 + (11 - w)];
            any |= v;
            out[(size_t)(first + p) * PT_DW + d] = bswap32(v);
        }
        const uint64_t nz = __ballot(any != 0);
        if (out_inf && lane == 0) out_inf[first + p] = (nz == 0) ? 1 : 0;
    }
}
#else
;
#endif

// ---------------------------------------------------------------------------
// Bucket method with ONE (group, chunk, window) PER LANE (fp28.h: 14 signed 28-bit limbs, sums of
// products): the lanes of a wavefront are the 64 windows of the same chunk of points, so they read
// the same point (one broadcast load) and add it to the bucket their own digit selects (signed
// nibbles: buckets 1 .. 8 take +-P; 15 buckets for a scalar too large to recode).  The buckets of
// a lane live in HBM in the L28 form (168 B G1 / 336 B G2 each, read and written once per addition:
// ~0.4 KB of traffic against ~4.5 k / 14 k instructions).  Used when a batch offers enough lanes to
// fill the chip (blsgpu_api.hip); the LDS-bucket kernels above serve the rest.
template <int DEG> struct LaneElem;
template <> struct LaneElem<1> { typedef r28::fe E; };
template <> struct LaneElem<2> { typedef r28::fe2 E; };
constexpr uint32_t L28_AFF = 2 * r28::NL;                     // dwords of an affine L28 point per degree
constexpr uint32_t L28_PJ = 3 * r28::NL;                      // dwords of a projective L28 point per degree

// the VM's projective form (36 * DEG dwords, x 2^384, canonical) of an L28 point
__device__ __forceinline__ void lane_st_vm(const r28::fe& x, uint32_t* p) { r28::to_vm(p, x); }
__device__ __forceinline__ void lane_st_vm(const r28::fe2& x, uint32_t* p) { r28::to_vm(p, x.a); r28::to_vm(p + 12, x.b); }
template <class E> __device__ __forceinline__ void lane_st_pt_vm(const r28::ptT<E>& r, uint32_t* p) {
    constexpr int W = 12 * (r28::Elem<E>::DW / r28::NL);
    lane_st_vm(r.X, p); lane_st_vm(r.Y, p + W); lane_st_vm(r.Z, p + 2 * W);
}

// affine big-endian points (96 * DEG bytes, (0, 0) = infinity) -> affine L28 points + live flags, one point per
// lane: a product by R^2 per coordinate
template <int DEG>
__global__ void __launch_bounds__(256) k_lane_prep(const uint32_t* __restrict__ pts, uint32_t n, uint32_t* __restrict__ prep,
                                                   uint8_t* __restrict__ live)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t any = 0;
#pragma unroll 1
    for (int k = 0; k < 2 * DEG; k++) {
        uint32_t c[12];
#pragma unroll
        for (int w = 0; w < 12; w++) { c[11 - w] = bswap32(pts[(size_t)i * 24 * DEG + k * 12 + w]); any |= c[11 - w]; }
        r28::st(r28::from_raw(c), prep + (size_t)i * L28_AFF * DEG + k * r28::NL);
    }
    live[i] = any ? 1 : 0;
}
#else
;
#endif

// prep / live: k_lane_prep's; partial (win, chunk) of group g is written at partials[((g * PIP_W + win) * chunks +
// chunk) * PJ] -- the layout k_msm_pip_windows reads -- in the VM's form (PJ = 36 DEG) when vm_out, else L28 (42 DEG).
template <int DEG>
__global__ void __launch_bounds__(64) k_msm_lane(const uint32_t* __restrict__ prep, const uint8_t* __restrict__ live,
                                                 const uint32_t* __restrict__ scalars, uint32_t k,
                                                 uint32_t chunk, uint32_t chunks, uint32_t total, uint32_t* __restrict__ buckets,
                                                 uint32_t* __restrict__ partials, uint32_t vm_out)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    typedef typename LaneElem<DEG>::E E;
    constexpr uint32_t PJ_DW = L28_PJ * DEG, AF_DW = L28_AFF * DEG;
    const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= total) return;
    const uint32_t win = L & 63u, cidx = (L >> 6) % chunks, grp = (L >> 6) / chunks;
    const uint32_t lo = grp * k + cidx * chunk, hi = min(grp * k + k, lo + chunk);
    uint32_t* B = buckets + (size_t)L * (PIP_NB - 1) * PJ_DW;                   // buckets 1 .. 15
    {
        const r28::ptT<E> inf = r28::pt_inf<E>();
        for (int j = 0; j < PIP_NB - 1; j++) r28::pt_st(inf, B + j * PJ_DW);
    }
    // Signed digits: the nibbles of s + 0x88..8 minus 8 are digits in [-8, 7] with the same value (no carry out of
    // 256 bits while s < 2^256 - 0x88..8, which covers every scalar below the group order), so 8 buckets take
    // +-P and the running sums below cover 8 buckets instead of 15.  A larger scalar keeps its plain nibbles
    // (buckets 1 .. 15); the lanes of a wavefront see the same scalars, so `wide` is uniform.
    bool wide = false;
#pragma unroll 1
    for (uint32_t i = lo; i < hi; i++) {
        if (!live[i]) continue;                                                 // the point at infinity
        uint32_t word;                                                          // the 32 bits that hold this window's nibble
        bool big = false;
        if (scalars) {
            const uint32_t* sc = scalars + (size_t)i * 8;
            big = bswap32(sc[0]) > 0x77777776u;
            if (!big) {
                uint32_t c = 0;
                word = 0;
                for (uint32_t j = 0; j <= win / 8; j++) word = bls::addc(bswap32(sc[7u - j]), 0x88888888u, c);
            } else {
                word = bswap32(sc[7 - win / 8]);
            }
        } else {
            word = (win < 8) ? 0x88888889u : 0x88888888u;
        }
        const int nib = (int)((word >> (4 * (win % 8))) & 15u);
        const int d = big ? nib : nib - 8;
        wide = wide || big;
        if (d != 0) {
            uint32_t* b = B + ((d < 0 ? -d : d) - 1) * PJ_DW;
            const uint32_t* pt = prep + (size_t)i * AF_DW;
            const E x2 = r28::Elem<E>::load(pt);
            E y2 = r28::Elem<E>::load(pt + r28::Elem<E>::DW);
            if (d < 0) y2 = r28::norm(r28::neg(y2));
            r28::ptT<E> r = r28::pt_ld<E>(b);
            r28::pmadd(r, x2, y2);
            r28::pt_st(r, b);
        }
    }
    r28::ptT<E> acc = r28::pt_inf<E>(), tot = r28::pt_inf<E>();                 // sum_j j B_j by running sums
#pragma unroll 1
    for (int j = wide ? PIP_NB - 2 : PIP_NB / 2 - 1; j >= 0; j--) {
        if constexpr (DEG == 1) {
            acc = r28::padd(acc, r28::pt_ld<E>(B + j * PJ_DW));
            tot = r28::padd(tot, acc);
        } else {                                                                // a twist addition is ~140 KB of code: one copy
            acc = r28::padd_fn(acc, r28::pt_ld<E>(B + j * PJ_DW));
            tot = r28::padd_fn(tot, acc);
        }
    }
    const size_t o = ((size_t)grp * PIP_W + win) * chunks + cidx;
    if (vm_out) lane_st_pt_vm(tot, partials + o * 36 * DEG);
    else r28::pt_st(tot, partials + o * PJ_DW);
}
#else
;
#endif

// ---- the same bucket sums for G2 on LANE PAIRS (round 3) ------------------------------------------------------------
// k_msm_lane<2> needs 511 registers and 1.7 KB of scratch per lane for a twist addition (one wavefront per SIMD, 55 % of
// the issue rate).  With every Fq2 value split over two adjacent lanes (blsgpu_ml.hip, namespace sp: even lane real
// part, odd lane imaginary part, partner's part one DPP move away) a lane holds half of every coordinate: no scratch,
// two wavefronts per SIMD, and the mixed addition (7 k instructions per lane) stays in the instruction cache.  Same
// complete formulas (RCB 2015 algorithms 7 and 8), same bucket and partial layouts as k_msm_lane<2>.
namespace sp2 {
using namespace ml::sp;
struct pt { h X, Y, Z; };
__device__ __forceinline__ pt pt_inf() {
    const int32_t one[r28::NL] = BLS28_ONE;
    pt r;
#pragma unroll
    for (int j = 0; j < r28::NL; j++) { r.X.v[j] = 0; r.Y.v[j] = odd() ? 0 : one[j]; r.Z.v[j] = 0; }
    return r;
}
__device__ __forceinline__ h ldh(const uint32_t* __restrict__ p) {                  // the lane's half of an fe2 at p
    h r;
    const uint32_t* q = p + (odd() ? r28::NL : 0);
#pragma unroll
    for (int j = 0; j < r28::NL; j++) r.v[j] = (int32_t)q[j];
    return r;
}
__device__ __forceinline__ void sth(const h& x, uint32_t* __restrict__ p) {
    uint32_t* q = p + (odd() ? r28::NL : 0);
#pragma unroll
    for (int j = 0; j < r28::NL; j++) q[j] = (uint32_t)x.v[j];
}
__device__ __forceinline__ pt pt_ld(const uint32_t* __restrict__ p) { return {ldh(p), ldh(p + 2 * r28::NL), ldh(p + 4 * r28::NL)}; }
__device__ __forceinline__ void pt_st(const pt& P, uint32_t* __restrict__ p) { sth(P.X, p); sth(P.Y, p + 2 * r28::NL); sth(P.Z, p + 4 * r28::NL); }
// P += (x2 : y2 : 1), RCB algorithm 8
__device__ __forceinline__ void pmadd(pt& P, const h& x2, const h& y2) {
    const Lop<1> lx2 = left(x2), ly2 = left(y2);
    const Rop<1> rX = right(P.X), rY = right(P.Y), rZ = right(P.Z);
    const h t0 = mul(lx2, rX), t1 = mul(ly2, rY);
    const h t3 = norm(sub(sub(mul(left(add(x2, y2)), right(add(P.X, P.Y))), t0), t1));
    const h t4 = norm(add(mul(ly2, rZ), P.Y));
    const h y3 = b3(add(mul(lx2, rZ), P.X));
    const h x3 = mulc_norm<3>(t0);
    const h bz = b3(P.Z);
    const h z3 = norm(add(t1, bz)), t1m = norm(sub(t1, bz));
    const Lop<1> lz3 = left(z3), lx3 = left(x3), ly3 = left(y3), lt1m = left(t1m);
    const Rop<1> rt3 = right(t3), rt4 = right(t4);
    P.X = dot2(left(t3), right(t1m), left(neg(t4)), right(y3));
    P.Y = dot2(ly3, right(x3), lt1m, right(z3));
    P.Z = dot2(lz3, rt4, lx3, rt3);
}
// P + Q, RCB algorithm 7
__device__ __forceinline__ pt padd(const pt& P, const pt& Q) {
    const Rop<1> rQX = right(Q.X), rQY = right(Q.Y), rQZ = right(Q.Z);
    const h t0 = mul(left(P.X), rQX), t1 = mul(left(P.Y), rQY), t2 = mul(left(P.Z), rQZ);
    const h t3 = norm(sub(sub(mul(left(add(P.X, P.Y)), right(add(Q.X, Q.Y))), t0), t1));
    const h t4 = norm(sub(sub(mul(left(add(P.Y, P.Z)), right(add(Q.Y, Q.Z))), t1), t2));
    const S<3> t5 = sub(sub(mul(left(add(P.X, P.Z)), right(add(Q.X, Q.Z))), t0), t2);
    const h x3 = mulc_norm<3>(t0);
    const h bz = b3(t2);
    const h z3 = norm(add(t1, bz)), t1m = norm(sub(t1, bz));
    const h y3 = b3(t5);
    pt R;
    R.X = dot2(left(t3), right(t1m), left(neg(t4)), right(y3));
    R.Y = dot2(left(t1m), right(z3), left(y3), right(x3));
    R.Z = dot2(left(z3), right(t4), left(x3), right(t3));
    return R;
}
// 2 P, RCB algorithm 9
__device__ __forceinline__ pt pdbl(const pt& P) {
    const Lop<1> ly = left(P.Y);
    const h t0 = sqr(P.Y), t1 = mul(ly, right(P.Z)), t2 = b3(sqr(P.Z)), txy = mul(left(P.X), right(P.Y));
    const h z8 = mulc_norm<8>(t0);
    const h d = norm(sub(t0, mulc<3>(t2)));
    const Lop<1> ld = left(d);
    pt R;
    R.X = mul(ld, right(add(txy, txy)));
    R.Y = dot2(left(t2), right(z8), ld, right(add(t0, t2)));
    R.Z = mul(left(t1), right(z8));
    return R;
}
__device__ __forceinline__ pt pneg(const pt& P) { return {P.X, norm(neg(P.Y)), P.Z}; }
// ---- Jacobian coordinates (X / Z^2, Y / Z^3) for long runs of doublings (round 4; the cofactor clearing of the hash) -----
// 2 P for a = 0 (EFD dbl-2009-l): A = X^2, B = Y^2, C = B^2, D = 2 ((X + B)^2 - A - C), E = 3 A, X3 = E^2 - 2 D,
// Y3 = E (D - X3) - 8 C, Z3 = 2 Y Z: five squares (one product per lane each) and two products -- 3136 multiply-adds per lane
// against 4116 of the complete homogeneous doubling.  Exact for every point of E'(Fq2): the group order is odd (no point
// with Y = 0 doubles to infinity), and infinity itself is carried as (1 : 1 : 0), which this map fixes.
__device__ __forceinline__ pt pdblj(const pt& P) {
    const h A = sqr(P.X), B = sqr(P.Y);
    const h C = sqr(B);
    const h D = mulc_norm<2>(sub(sub(sqr(norm(add(P.X, B))), A), C));
    const h E = mulc_norm<3>(A);
    pt R;
    R.X = norm(sub(sqr(E), mulc<2>(D)));
    R.Z = mul(left(add(P.Y, P.Y)), right(P.Z));
    R.Y = norm(sub(mul(left(E), right(norm(sub(D, R.X)))), mulc_norm<8>(C)));
    return R;
}
// homogeneous (X : Y : Z) -> Jacobian (X Z : Y Z^2 : Z); infinity (Z = 0) -> (1 : 1 : 0)
__device__ __forceinline__ pt to_jacobian(const pt& P) {
    const bool inf = zero2(P.Z);
    const int32_t one[r28::NL] = BLS28_ONE;
    const h zz = sqr(P.Z);
    const h x = mul(left(P.X), right(P.Z)), y = mul(left(P.Y), right(zz));
    pt R;
#pragma unroll
    for (int j = 0; j < r28::NL; j++) {
        const int32_t o = odd() ? 0 : one[j];
        R.X.v[j] = inf ? o : x.v[j];
        R.Y.v[j] = inf ? o : y.v[j];
        R.Z.v[j] = P.Z.v[j];
    }
    return R;
}
// Jacobian (X : Y : Z) -> homogeneous (X Z : Y : Z^3); (1 : 1 : 0) -> (0 : 1 : 0)
__device__ __forceinline__ pt to_homogeneous(const pt& P) {
    const h zz = sqr(P.Z);
    pt R;
    R.X = mul(left(P.X), right(P.Z));
    R.Y = P.Y;
    R.Z = mul(left(zz), right(P.Z));
    return R;
}
__device__ __forceinline__ void st_vm(const pt& r, uint32_t* __restrict__ p) {     // the VM's projective form: 72 dwords
    const h* c[3] = {&r.X, &r.Y, &r.Z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r28::fe t;
#pragma unroll
        for (int j = 0; j < r28::NL; j++) t.v[j] = c[k]->v[j];
        uint32_t w[12];
        r28::to_vm(w, t);
        uint32_t* o = p + k * 24 + (odd() ? 12 : 0);
#pragma unroll
        for (int j = 0; j < 12; j++) o[j] = w[j];
    }
}
}  // namespace sp2

// ---- the same curve arithmetic on LANE QUADS (round 4): one point per four lanes ----------------------------------------
// A batch of a few thousand points leaves most SIMDs empty on lane pairs (16 384 messages of the hash = 512 wavefronts) and
// what the call waits for is the length of one wavefront's instruction stream.  Here the two lane PAIRS of a quad hold
// the same point (each lane its real or imaginary halves, as in sp2) and every level of independent Fq2 products of a
// formula is dealt out over them -- both pairs execute the same instructions on operands picked by the pair index, then
// swap results (DPP quad_perm [2,3,0,1]) -- so a formula is about half as deep: Jacobian doubling 3 squares + 1 product
// deep instead of 5 + 2, complete addition 6 products instead of 12.  Throughput-wise it is the same work plus the
// swaps, so it is used only while the batch does not fill the chip on pairs (blsgpu_api.hip h2c_quad_max).
namespace sp4 {
using namespace sp2;
using ml::sq::hi;
using ml::sq::oth;
using ml::sq::pick;
template <int M> __device__ __forceinline__ S<2> wide(const S<M>& x) { return ml::sq::widen<2>(x); }
// 2 P, Jacobian, a = 0 (sp2::pdblj): levels {A = X^2 | B = Y^2}, {C = B^2 | (X + B)^2}, {E^2}, {E (D - X3) | 2Y Z}
__device__ __forceinline__ pt pdblj(const pt& P) {
    const h r1 = sqr(pick(P.X, P.Y)), o1 = oth(r1);
    const h A = pick(r1, o1), B = pick(o1, r1);
    const h r2 = sqr(pick(B, norm(add(P.X, B)))), o2 = oth(r2);
    const h C = pick(r2, o2), T = pick(o2, r2);
    const h D = mulc_norm<2>(sub(sub(T, A), C));
    const h E = mulc_norm<3>(A);
    pt R;
    R.X = norm(sub(sqr(E), mulc<2>(D)));
    const h r4 = mul(left(pick(wide(E), add(P.Y, P.Y))), right(pick(norm(sub(D, R.X)), P.Z))), o4 = oth(r4);
    R.Z = pick(o4, r4);
    R.Y = norm(sub(pick(r4, o4), mulc_norm<8>(C)));
    return R;
}
// 2 P, complete homogeneous (sp2::pdbl, RCB algorithm 9): levels {Y^2, Y Z | Z^2, X Y}, {d 2XY, YZ 8Y^2 | 3b'Z^2 8Y^2, d (Y^2 + 3b'Z^2)}
__device__ __forceinline__ pt pdbl(const pt& P) {
    const h s1 = sqr(pick(P.Y, P.Z)), m1 = mul(left(pick(P.Y, P.X)), right(pick(P.Z, P.Y)));
    const h os = oth(s1), om = oth(m1);
    const h t0 = pick(s1, os), t1 = pick(m1, om), t2 = b3(pick(os, s1)), txy = pick(om, m1);
    const h z8 = mulc_norm<8>(t0);
    const h d = norm(sub(t0, mulc<3>(t2)));
    const h ma = mul(left(pick(wide(d), wide(t2))), right(pick(norm(add(txy, txy)), z8)));
    const h mb = mul(left(pick(wide(t1), wide(d))), right(pick(z8, norm(add(t0, t2)))));
    const h oa = oth(ma), ob = oth(mb);
    pt R;
    R.X = pick(ma, oa);
    R.Z = pick(mb, ob);
    R.Y = norm(add(pick(oa, ma), pick(ob, mb)));
    return R;
}
// P + Q, complete homogeneous (sp2::padd, RCB algorithm 7): levels {X1X2, Y1Y2, Z1Z2 | the three products of sums},
// {t3 t1m, t1m z3, z3 t4 | (-t4) y3, y3 x3, x3 t3}
__device__ __forceinline__ pt padd(const pt& P, const pt& Q) {
    const S<2> l0 = pick(wide(P.X), add(P.X, P.Y)), l1 = pick(wide(P.Y), add(P.Y, P.Z)), l2 = pick(wide(P.Z), add(P.X, P.Z));
    const S<2> q0 = pick(wide(Q.X), add(Q.X, Q.Y)), q1 = pick(wide(Q.Y), add(Q.Y, Q.Z)), q2 = pick(wide(Q.Z), add(Q.X, Q.Z));
    const h a0 = mul(left(l0), right(q0)), a1 = mul(left(l1), right(q1)), a2 = mul(left(l2), right(q2));
    const h b0 = oth(a0), b1 = oth(a1), b2 = oth(a2);
    const h t0 = pick(a0, b0), t1 = pick(a1, b1), t2 = pick(a2, b2);
    const h u3 = pick(b0, a0), u4 = pick(b1, a1), u5 = pick(b2, a2);
    const h t3 = norm(sub(sub(u3, t0), t1));
    const h t4 = norm(sub(sub(u4, t1), t2));
    const S<3> t5 = sub(sub(u5, t0), t2);
    const h x3 = mulc_norm<3>(t0);
    const h bz = b3(t2);
    const h z3 = norm(add(t1, bz)), t1m = norm(sub(t1, bz));
    const h y3 = b3(t5);
    const h nt4 = norm(neg(t4));
    const h c0 = mul(left(pick(t3, nt4)), right(pick(t1m, y3)));
    const h c1 = mul(left(pick(t1m, y3)), right(pick(z3, x3)));
    const h c2 = mul(left(pick(z3, x3)), right(pick(t4, t3)));
    pt R;
    R.X = norm(add(c0, oth(c0)));
    R.Y = norm(add(c1, oth(c1)));
    R.Z = norm(add(c2, oth(c2)));
    return R;
}
// homogeneous -> Jacobian (sp2::to_jacobian): {Z^2}, {X Z | Y Z^2}
__device__ __forceinline__ pt to_jacobian(const pt& P) {
    const bool inf = zero2(P.Z);
    const int32_t one[r28::NL] = BLS28_ONE;
    const h zz = sqr(P.Z);
    const h m = mul(left(pick(P.X, P.Y)), right(pick(P.Z, zz))), o = oth(m);
    const h x = pick(m, o), y = pick(o, m);
    pt R;
#pragma unroll
    for (int j = 0; j < r28::NL; j++) {
        const int32_t u = odd() ? 0 : one[j];
        R.X.v[j] = inf ? u : x.v[j];
        R.Y.v[j] = inf ? u : y.v[j];
        R.Z.v[j] = P.Z.v[j];
    }
    return R;
}
// Jacobian -> homogeneous (sp2::to_homogeneous): {Z^2}, {X Z | Z^2 Z}
__device__ __forceinline__ pt to_homogeneous(const pt& P) {
    const h zz = sqr(P.Z);
    const h m = mul(left(pick(P.X, zz)), right(P.Z)), o = oth(m);
    pt R;
    R.X = pick(m, o);
    R.Y = P.Y;
    R.Z = pick(o, m);
    return R;
}
}  // namespace sp4

// The window Horner of a BATCH of G2 sums with ONE SUM PER LANE QUAD (round 4): result_g = sum_w 2^(cbits w) W_{g,w} from the
// top window down, 252 complete doublings and 63 complete additions deep for the threshold combine (64 windows of 4 bits).
// k_msm_horner_np runs five sums per wavefront on the wavefront VM (10 000 groups: 2000 wavefronts x 1 M instructions);
// here the point operations are sp4's (both pairs of a quad share each level of independent Fq2 products), the window
// sums are read in the L28 form k_msm_lane2x leaves them in (no copy through the VM's form), and the affine conversion
// is the hash's (one safegcd inversion of the norm of Z per lane).  out: groups x 192 bytes, (0, 0) + flag for infinity.
__global__ void __launch_bounds__(256, 2) k_msm_horner_quads(const uint32_t* __restrict__ winsums, uint32_t nwin, uint32_t cbits, uint32_t groups,
                                                            uint32_t* __restrict__ out, uint8_t* __restrict__ out_inf)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    using namespace sp2;
    constexpr uint32_t PJ_DW = L28_PJ * 2;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t g = min(tid >> 2, groups - 1u), part = tid & 1u;          // spare quads of the last wavefront repeat the last sum
    const bool store = (tid >> 2) < groups && (tid & 2u) == 0u;
    const uint32_t* W = winsums + (size_t)g * nwin * PJ_DW;
    pt acc = pt_ld(W + (size_t)(nwin - 1u) * PJ_DW);
#pragma unroll 1
    for (int w = (int)nwin - 2; w >= 0; w--) {
#pragma unroll 1
        for (uint32_t s = 0; s < cbits; s++) acc = sp4::pdbl(acc);
        const pt Q = pt_ld(W + (size_t)w * PJ_DW);
        acc = sp4::padd(acc, Q);
    }
    // affine: (X, Y) / Z with 1 / Z = conj(Z) / N(Z); Z = 0 gives (0, 0)
    const h zp = swp(acc.Z);
    r28::fe n;
    bls28::fp28_dot2(n.v, acc.Z.v, acc.Z.v, zp.v, zp.v);
    uint32_t nv[12], niv[12];
    r28::to_vm(nv, n);
    bls::fq_inv(niv, nv);
    const r28::fe ninv = r28::from_vm(niv);
    S<1> zc;
#pragma unroll
    for (int j = 0; j < r28::NL; j++) zc.v[j] = part ? -acc.Z.v[j] : acc.Z.v[j];
    const h zi = mulf(zc, ninv);
    const Rop<1> rzi = right(zi);
    const h xa = mul(left(acc.X), rzi), ya = mul(left(acc.Y), rzi);
    const h* o[2] = {&xa, &ya};
    uint32_t any = 0;
    for (int k = 0; k < 2; k++) {
        r28::fe t;
#pragma unroll
        for (int j = 0; j < r28::NL; j++) t.v[j] = o[k]->v[j];
        uint32_t y[12];
        r28::to_raw(y, t);
#pragma unroll
        for (int w = 0; w < 12; w++) {
            any |= y[w];
            if (store) out[(size_t)g * 48 + (2 * k + part) * 12 + w] = bswap32(y[11 - w]);
        }
    }
    any |= (uint32_t)__shfl_xor((int)any, 1);
    if (out_inf && store && part == 0u) out_inf[g] = any ? 0 : 1;
}
#else
;
#endif

#ifndef BLSGPU_MSM_LANE2X_WAVES
#define BLSGPU_MSM_LANE2X_WAVES 2
#endif
__global__ void __launch_bounds__(64, BLSGPU_MSM_LANE2X_WAVES) k_msm_lane2x(const uint32_t* __restrict__ prep, const uint8_t* __restrict__ live,
                                                  const uint32_t* __restrict__ scalars, uint32_t k,
                                                  uint32_t chunk, uint32_t chunks, uint32_t total, uint32_t* __restrict__ buckets,
                                                  uint32_t* __restrict__ partials, uint32_t vm_out)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    constexpr uint32_t PJ_DW = L28_PJ * 2, AF_DW = L28_AFF * 2;
    const uint32_t T = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t L = min(T >> 1, total - 1u);                                  // spare lanes of the last wavefront repeat the last item
    const bool store = (T >> 1) < total;
    const uint32_t win = L & 63u, cidx = (L >> 6) % chunks, grp = (L >> 6) / chunks;
    const uint32_t lo = grp * k + cidx * chunk, hi = min(grp * k + k, lo + chunk);
    uint32_t* B = buckets + (size_t)L * (PIP_NB - 1) * PJ_DW;                   // buckets 1 .. 15
    {
        const sp2::pt inf = sp2::pt_inf();
        for (int j = 0; j < PIP_NB - 1; j++) sp2::pt_st(inf, B + j * PJ_DW);
    }
    bool wide = false;                                                          // (digits as in k_msm_lane)
#pragma unroll 1
    for (uint32_t i = lo; i < hi; i++) {
        if (!live[i]) continue;
        uint32_t word;
        bool big = false;
        if (scalars) {
            const uint32_t* sc = scalars + (size_t)i * 8;
            big = bswap32(sc[0]) > 0x77777776u;
            if (!big) {
                uint32_t c = 0;
                word = 0;
                for (uint32_t j = 0; j <= win / 8; j++) word = bls::addc(bswap32(sc[7u - j]), 0x88888888u, c);
            } else {
                word = bswap32(sc[7 - win / 8]);
            }
        } else {
            word = (win < 8) ? 0x88888889u : 0x88888888u;
        }
        const int nib = (int)((word >> (4 * (win % 8))) & 15u);
        const int d = big ? nib : nib - 8;
        wide = wide || big;
        if (d != 0) {
            uint32_t* b = B + ((d < 0 ? -d : d) - 1) * PJ_DW;
            const uint32_t* pt = prep + (size_t)i * AF_DW;
            const sp2::h x2 = sp2::ldh(pt);
            sp2::h y2 = sp2::ldh(pt + 2 * r28::NL);
            if (d < 0) y2 = sp2::norm(sp2::neg(y2));
            sp2::pt r = sp2::pt_ld(b);
            sp2::pmadd(r, x2, y2);
            sp2::pt_st(r, b);
        }
    }
    // sum_j j B_j by running sums: acc += B_j, tot += acc -- ONE inlined addition, its operands chosen by the step's parity
    sp2::pt acc = sp2::pt_inf(), tot = sp2::pt_inf();
    const int nb = wide ? PIP_NB - 1 : PIP_NB / 2;
#pragma unroll 1
    for (int it = 0; it < 2 * nb; it++) {
        const bool first = (it & 1) == 0;
        const int j = nb - 1 - (it >> 1);
        const sp2::pt Bj = sp2::pt_ld(B + j * PJ_DW);
        sp2::pt P, Q;
#pragma unroll
        for (int e = 0; e < r28::NL; e++) {
            P.X.v[e] = first ? acc.X.v[e] : tot.X.v[e]; P.Y.v[e] = first ? acc.Y.v[e] : tot.Y.v[e]; P.Z.v[e] = first ? acc.Z.v[e] : tot.Z.v[e];
            Q.X.v[e] = first ? Bj.X.v[e] : acc.X.v[e]; Q.Y.v[e] = first ? Bj.Y.v[e] : acc.Y.v[e]; Q.Z.v[e] = first ? Bj.Z.v[e] : acc.Z.v[e];
        }
        const sp2::pt R = sp2::padd(P, Q);
        if (first) acc = R; else tot = R;
    }
    if (store) {
        const size_t o = ((size_t)grp * PIP_W + win) * chunks + cidx;
        if (vm_out) sp2::st_vm(tot, partials + o * 72);
        else sp2::pt_st(tot, partials + o * PJ_DW);
    }
}
#else
;
#endif

// out[w * nfold + f] = sum of partials[w * chunks + f * per .. + per): one run per lane; L28 in, L28 or the VM's form out
template <int DEG>
__global__ void __launch_bounds__(64) k_msm_lane_fold(const uint32_t* __restrict__ partials, uint32_t chunks, uint32_t per, uint32_t nfold,
                                                      uint32_t total, uint32_t* __restrict__ out, uint32_t vm_out)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    typedef typename LaneElem<DEG>::E E;
    constexpr uint32_t PJ_DW = L28_PJ * DEG;
    const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= total) return;
    const uint32_t w = L / nfold, f = L % nfold;
    const uint32_t lo = f * per, hi = min(chunks, lo + per);
    r28::ptT<E> acc = r28::pt_inf<E>();
#pragma unroll 1
    for (uint32_t i = lo; i < hi; i++) acc = r28::padd(acc, r28::pt_ld<E>(partials + ((size_t)w * chunks + i) * PJ_DW));
    if (vm_out) lane_st_pt_vm(acc, out + ((size_t)w * nfold + f) * 36 * DEG);
    else r28::pt_st(acc, out + ((size_t)w * nfold + f) * PJ_DW);
}
#else
;
#endif

// ---------------------------------------------------------------------------
// Sorted buckets: ONE large G1 sum with scalars (BLS.aggregate_pub_keys(secure) at scale, bls.py:203-223 --
// BASELINE config 5).  Windows of `cb` bits (20 windows instead of 64), SIGNED digits (below); key = window * 2^(cb-1) +
// |digit| - 1.  A counting sort (k_srt_count, k_srt_scan_*, k_srt_scatter) lists the point indices by key; the list is cut
// into EQUAL pieces, one per lane (k_srt_accum): a lane keeps its running bucket sum in registers and adds affine
// points to it (complete mixed addition, two wavefronts per SIMD), writing a sum out whenever the key changes.  A
// bucket whose run begins in the lane is written by that lane alone; the piece of a run that continues from the lane
// before goes to the lane's "head" slot and k_srt_fix / k_srt_fix_long add the heads to their buckets.  sum_d d B_d is taken bit by
// bit -- S_b = sum of the buckets whose digit has bit b (k_srt_bits + k_msm_lane_fold: plain sums, no dependent
// chain over 8191 buckets), W_w = sum_b 2^b S_b (k_srt_windows) -- then Horner over the windows.
// Same value as the reference's double-and-add summed over the points (fields_t.py:705-740); parity is on the
// affine result.  Points at infinity ((0,0)) and zero digits are left out of the list.  Points and sums are in the
// L28 form of fp28.h (k_srt_prep); the tail is blsgpu_g1w.hip's (or, BLSGPU_MSM_WIDE_TAIL=0, the VM's: the last fold hands it its own form).
constexpr uint32_t SRT_LANES = 2048u * 64u;                   // two wavefronts per SIMD

// SIGNED digits (round 5): s = sum_w d_w 2^(cb w) with d_w in [-2^(cb-1), 2^(cb-1)) -- the digits of s + C, C = sum_w 2^(cb-1) 2^(cb w),
// are d_w + 2^(cb-1) in [0, 2^cb) -- so a window has 2^(cb-1) buckets |d| = 1 .. 2^(cb-1) (key index |d| - 1) that take +-P (the list
// entry carries the sign, the prepared point its -y): half the bucket sums to reduce, half the histogram, and one more bit per
// window in the same LDS.  258 <= cb x windows keeps s + C below 2^(cb x windows) for every s < 2^256.
constexpr uint32_t SRT_SCW = 9;                               // words of s + C, least significant first
struct SrtBias { uint32_t w[SRT_SCW]; };                      // C
// The group a sorted-bucket sum runs in (round 5: G2 too -- BLS.aggregate_sigs(secure), bls.py:225-261, as one multi-scalar sum):
// G1 one piece / key / item per LANE on r28's point arithmetic, G2 one per LANE PAIR on sp2's (an Fq2 value split over two adjacent
// lanes: no scratch, two wavefronts per SIMD).  AFF: dwords of a prepared point (x, y, -y), PJ: of a projective sum, both L28.
template <int DEG> struct SrtG;
template <> struct SrtG<1> {
    typedef r28::ptT<r28::fe> P;
    static constexpr uint32_t LP = 1, AFF = 3 * r28::NL, PJ = L28_PJ;
    static __device__ __forceinline__ P inf() { return r28::pt_inf<r28::fe>(); }
    static __device__ __forceinline__ P ld(const uint32_t* __restrict__ p) { return r28::pt_ld<r28::fe>(p); }
    static __device__ __forceinline__ void st(const P& a, uint32_t* __restrict__ p) { r28::pt_st(a, p); }
    static __device__ __forceinline__ P add(const P& a, const P& b) { return r28::padd(a, b); }
    static __device__ __forceinline__ void madd(P& a, const uint32_t* __restrict__ pt, uint32_t neg) {        // a += (x, neg ? -y : y)
        const r28::fe x2 = r28::ld(pt), y2 = r28::ld(pt + r28::NL + neg * r28::NL);
        r28::pmadd(a, x2, y2);
    }
    static __device__ __forceinline__ void xor_lanes(P& o, const P& a, int off) {
#pragma unroll
        for (int q = 0; q < r28::NL; q++) { o.X.v[q] = __shfl_xor(a.X.v[q], off); o.Y.v[q] = __shfl_xor(a.Y.v[q], off); o.Z.v[q] = __shfl_xor(a.Z.v[q], off); }
    }
    static __device__ __forceinline__ P dbl(const P& a) { return r28::pdbl(a); }
    // (X, Y) / Z as 96 canonical big-endian bytes, (0, 0) and the flag for Z = 0 (one safegcd inversion per lane)
    static __device__ __forceinline__ void affine_out(const P& a, uint32_t* __restrict__ out, uint8_t* __restrict__ inf, bool store) {
        uint32_t zv[12], ziv[12];
        r28::to_vm(zv, a.Z);
        bls::fq_inv(ziv, zv);
        const r28::fe zi = r28::from_vm(ziv);
        const r28::fe c[2] = {r28::mul(a.X, zi), r28::mul(a.Y, zi)};
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            uint32_t y[12];
            r28::to_raw(y, c[k]);
#pragma unroll
            for (int w = 0; w < 12; w++) {
                any |= y[w];
                if (store) out[k * 12 + w] = bswap32(y[11 - w]);
            }
        }
        if (inf && store) *inf = any ? 0 : 1;
    }
};
template <> struct SrtG<2> {
    typedef sp2::pt P;
    static constexpr uint32_t LP = 2, AFF = 6 * r28::NL, PJ = 2 * L28_PJ;
    static __device__ __forceinline__ P inf() { return sp2::pt_inf(); }
    static __device__ __forceinline__ P ld(const uint32_t* __restrict__ p) { return sp2::pt_ld(p); }
    static __device__ __forceinline__ void st(const P& a, uint32_t* __restrict__ p) { sp2::pt_st(a, p); }
    static __device__ __forceinline__ P add(const P& a, const P& b) { return sp2::padd(a, b); }
    static __device__ __forceinline__ void madd(P& a, const uint32_t* __restrict__ pt, uint32_t neg) {
        const sp2::h x2 = sp2::ldh(pt), y2 = sp2::ldh(pt + 2 * r28::NL + neg * 2 * r28::NL);
        sp2::pmadd(a, x2, y2);
    }
    static __device__ __forceinline__ void xor_lanes(P& o, const P& a, int off) {              // the same part of the pair `off` pairs away
#pragma unroll
        for (int q = 0; q < r28::NL; q++) { o.X.v[q] = __shfl_xor(a.X.v[q], 2 * off); o.Y.v[q] = __shfl_xor(a.Y.v[q], 2 * off); o.Z.v[q] = __shfl_xor(a.Z.v[q], 2 * off); }
    }
    static __device__ __forceinline__ P dbl(const P& a) { return sp2::pdbl(a); }
    // (X, Y) / Z with 1 / Z = conj(Z) / N(Z) as 192 canonical big-endian bytes (x.c0, x.c1, y.c0, y.c1; each lane of the pair its
    // parts), (0, 0) and the flag for Z = 0 -- the tail of k_msm_horner_quads on a lane pair
    static __device__ __forceinline__ void affine_out(const P& a, uint32_t* __restrict__ out, uint8_t* __restrict__ inf, bool store) {
        using namespace sp2;
        const uint32_t part = odd() ? 1u : 0u;
        const h zp = swp(a.Z);
        r28::fe n;
        bls28::fp28_dot2(n.v, a.Z.v, a.Z.v, zp.v, zp.v);
        uint32_t nv[12], niv[12];
        r28::to_vm(nv, n);
        bls::fq_inv(niv, nv);
        const r28::fe ninv = r28::from_vm(niv);
        S<1> zc;
#pragma unroll
        for (int j = 0; j < r28::NL; j++) zc.v[j] = part ? -a.Z.v[j] : a.Z.v[j];
        const h zi = mulf(zc, ninv);
        const Rop<1> rzi = right(zi);
        const h xa = mul(left(a.X), rzi), ya = mul(left(a.Y), rzi);
        const h* o[2] = {&xa, &ya};
        uint32_t any = 0;
        for (int k = 0; k < 2; k++) {
            r28::fe t;
#pragma unroll
            for (int j = 0; j < r28::NL; j++) t.v[j] = o[k]->v[j];
            uint32_t y[12];
            r28::to_raw(y, t);
#pragma unroll
            for (int w = 0; w < 12; w++) {
                any |= y[w];
                if (store) out[(2 * k + part) * 12 + w] = bswap32(y[11 - w]);
            }
        }
        any |= (uint32_t)__shfl_xor((int)any, 1);
        if (inf && store && part == 0u) *inf = any ? 0 : 1;
    }
};

// affine big-endian points (96 DEG bytes, (0, 0) = infinity) -> x, y, -y in the L28 form + live flags; big-endian scalars -> s + C
template <int DEG>
__global__ void __launch_bounds__(256) k_srt_prep(const uint32_t* __restrict__ pts, const uint32_t* __restrict__ scalars, uint32_t n, SrtBias C,
                                                  uint32_t* __restrict__ prep, uint8_t* __restrict__ live, uint32_t* __restrict__ rec)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t any = 0;
    uint32_t* dst = prep + (size_t)i * SrtG<DEG>::AFF;
#pragma unroll 1
    for (int k = 0; k < 2 * DEG; k++) {
        uint32_t c[12];
#pragma unroll
        for (int w = 0; w < 12; w++) { c[11 - w] = bswap32(pts[(size_t)i * 24 * DEG + k * 12 + w]); any |= c[11 - w]; }
        const r28::fe v = r28::from_raw(c);
        r28::st(v, dst + k * r28::NL);
        if (k >= DEG) r28::st(r28::norm(r28::neg(v)), dst + (k + DEG) * r28::NL);          // a part of y: its negative behind y
    }
    live[i] = any ? 1 : 0;
    uint64_t t = 0;
#pragma unroll
    for (int j = 0; j < (int)SRT_SCW; j++) {
        t += (uint64_t)(j < 8 ? bswap32(scalars[(size_t)i * 8 + 7 - j]) : 0u) + C.w[j];
        rec[(size_t)i * SRT_SCW + j] = (uint32_t)t;
        t >>= 32;
    }
}
#else
;
#endif

// biased digit of window w of a recoded scalar (one or two of its nine words)
__device__ __forceinline__ uint32_t srt_digit_at(const uint32_t* __restrict__ rec, uint32_t w, uint32_t cb) {
    const uint32_t o = w * cb, j = o >> 5, sft = o & 31u;
    uint32_t v = rec[j] >> sft;
    if (sft + cb > 32u && j + 1u < SRT_SCW) v |= rec[j + 1u] << (32u - sft);
    return v & ((1u << cb) - 1u);
}
// biased digit -> key index inside the window (|d| - 1) and the sign; false for d = 0
__device__ __forceinline__ bool srt_key(uint32_t v, uint32_t cb, uint32_t& k, uint32_t& sign) {
    const uint32_t half = 1u << (cb - 1u);
    sign = v < half ? 1u : 0u;
    k = (sign ? half - v : v - half) - 1u;
    return v != half;
}

// Counting sort by key, histograms in LDS.  Workgroup (slice, window) owns SRT_SLICE consecutive points and the
// 2^cb digits of one window: k_srt_count adds its histogram to cnt[]; k_srt_scatter rebuilds the histogram, reserves a
// range of every non-empty bucket's run with ONE global atomic per bucket, and hands the positions out from LDS.
constexpr uint32_t SRT_SLICE = 65536;
constexpr uint32_t SRT_MAXBITS = 14;                          // 2^13 u32 of LDS per workgroup (2^(cb-1) keys per window)

__device__ __forceinline__ void srt_histogram(uint32_t* hist, const uint32_t* __restrict__ rec, const uint8_t* __restrict__ live,
                                              uint32_t lo, uint32_t hi, uint32_t w, uint32_t cb) {
    for (uint32_t k = threadIdx.x; k < (1u << (cb - 1u)); k += blockDim.x) hist[k] = 0;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        uint32_t k, sign;
        if (live[i] && srt_key(srt_digit_at(rec + (size_t)i * SRT_SCW, w, cb), cb, k, sign)) atomicAdd(&hist[k], 1u);
    }
    __syncthreads();
}
__global__ void __launch_bounds__(1024) k_srt_count(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ live, uint32_t n,
                                                    uint32_t cb, uint32_t* __restrict__ cnt)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    __shared__ uint32_t hist[1u << (SRT_MAXBITS - 1)];
    const uint32_t w = blockIdx.y, lo = blockIdx.x * SRT_SLICE, hi = min(n, lo + SRT_SLICE), kb = cb - 1u;
    srt_histogram(hist, scalars, live, lo, hi, w, cb);
    for (uint32_t k = threadIdx.x; k < (1u << kb); k += blockDim.x)
        if (hist[k]) atomicAdd(&cnt[(w << kb) + k], hist[k]);
}
#else
;
#endif

// start[key] = exclusive prefix sum of cnt (start[nkeys] = number of list entries), cursor = start.
// k_srt_scan_window: workgroup w scans the 2^cb counts of its window (positions inside the window) and notes the window's
// total and longest run; k_srt_scan_add: adds the totals of the windows before.
__global__ void __launch_bounds__(1024) k_srt_scan_window(const uint32_t* __restrict__ cnt, uint32_t cb, uint32_t* __restrict__ start,
                                                          uint32_t* __restrict__ wtot, uint32_t* __restrict__ wmax)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t mx[1024];
    const uint32_t t = threadIdx.x, nb = 1u << cb, per = (nb + 1023u) / 1024u;
    const uint32_t lo = min(nb, t * per), hi = min(nb, lo + per);
    const uint32_t* c = cnt + ((size_t)blockIdx.x << cb);
    uint32_t s = 0, m = 0;
    for (uint32_t k = lo; k < hi; k++) { s += c[k]; m = max(m, c[k]); }
    part[t] = s;
    mx[t] = m;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {           // Hillis-Steele inclusive scan
        const uint32_t v = (t >= off) ? part[t - off] : 0u;
        const uint32_t w = (t >= off) ? mx[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        mx[t] = max(mx[t], w);
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (uint32_t k = lo; k < hi; k++) {
        start[((size_t)blockIdx.x << cb) + k] = run;
        run += c[k];
    }
    if (t == 1023) { wtot[blockIdx.x] = part[t]; wmax[blockIdx.x] = mx[t]; }
}
#else
;
#endif
__global__ void __launch_bounds__(1024) k_srt_scan_add(uint32_t nwin, uint32_t cb, const uint32_t* __restrict__ wtot,
                                                       const uint32_t* __restrict__ wmax, uint32_t* __restrict__ start,
                                                       uint32_t* __restrict__ cursor, uint32_t* __restrict__ maxcnt)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    uint32_t base = 0, all = 0, m = 0;
    for (uint32_t w = 0; w < nwin; w++) {
        if (w < blockIdx.x) base += wtot[w];
        all += wtot[w];
        m = max(m, wmax[w]);
    }
    for (uint32_t k = threadIdx.x; k < (1u << cb); k += blockDim.x) {
        const size_t key = ((size_t)blockIdx.x << cb) + k;
        const uint32_t v = start[key] + base;
        start[key] = v;
        cursor[key] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { start[(size_t)nwin << cb] = all; maxcnt[0] = m; }
}
#else
;
#endif

__global__ void __launch_bounds__(1024) k_srt_scatter(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ live, uint32_t n,
                                                      uint32_t cb, uint32_t* __restrict__ cursor, uint32_t* __restrict__ idx)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    __shared__ uint32_t hist[1u << (SRT_MAXBITS - 1)];
    const uint32_t w = blockIdx.y, lo = blockIdx.x * SRT_SLICE, hi = min(n, lo + SRT_SLICE), kb = cb - 1u;
    srt_histogram(hist, scalars, live, lo, hi, w, cb);
    for (uint32_t k = threadIdx.x; k < (1u << kb); k += blockDim.x)
        if (hist[k]) hist[k] = atomicAdd(&cursor[(w << kb) + k], hist[k]);        // first list position of this slice's entries
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        uint32_t k, sign;
        if (live[i] && srt_key(srt_digit_at(scalars + (size_t)i * SRT_SCW, w, cb), cb, k, sign)) idx[atomicAdd(&hist[k], 1u)] = i | (sign << 31);
    }
}
#else
;
#endif

// unit l (a lane, or a lane pair for G2) sums list entries [l per, (l + 1) per): see the header of this section.  headkey[l] = key of
// the piece that continues a run begun before the unit (or ~0), headpart[l] its sum.
#ifndef BLSGPU_SRT_WAVES
#define BLSGPU_SRT_WAVES 2
#endif
constexpr uint32_t SRT_PJ = L28_PJ;                           // dwords of a G1 sum (L28)
template <int DEG>
__global__ void __launch_bounds__(64, BLSGPU_SRT_WAVES) k_srt_accum(const uint32_t* __restrict__ prep, const uint32_t* __restrict__ idx,
                                                     const uint32_t* __restrict__ start, uint32_t nkeys, uint32_t nunits,
                                                     uint32_t* __restrict__ bsum, uint32_t* __restrict__ headpart,
                                                     uint32_t* __restrict__ headkey)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    typedef SrtG<DEG> G;
    const uint32_t L = (blockIdx.x * blockDim.x + threadIdx.x) / G::LP;
    if (L >= nunits) return;
    const uint32_t total = start[nkeys], per = (total + nunits - 1u) / nunits;
    const uint32_t p0 = min(total, L * per), p1 = min(total, p0 + per);
    uint32_t hk = 0xFFFFFFFFu;
    if (p0 < p1) {
        uint32_t lo = 0, hi = nkeys;                          // last key with start[key] <= p0 (the one that owns p0)
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (start[mid] <= p0) lo = mid; else hi = mid;
        }
        uint32_t key = lo, nxt = start[key + 1];
        bool head = start[key] < p0;
        typename G::P acc = G::inf();
#pragma unroll 1
        for (uint32_t p = p0; p < p1; p++) {
            if (p == nxt) {                                   // the run of `key` ends here
                if (head) { G::st(acc, headpart + (size_t)L * G::PJ); hk = key; }
                else G::st(acc, bsum + (size_t)key * G::PJ);
                head = false;
                acc = G::inf();
                do { key++; nxt = start[key + 1]; } while (nxt <= p);
            }
            const uint32_t e = idx[p];                         // point index, bit 31: the digit is negative (add -P)
            G::madd(acc, prep + (size_t)(e & 0x7FFFFFFFu) * G::AFF, e >> 31);
        }
        if (head) { G::st(acc, headpart + (size_t)L * G::PJ); hk = key; }
        else G::st(acc, bsum + (size_t)key * G::PJ);
    }
    headkey[L] = hk;
}
#else
;
#endif

// one unit per key: empty buckets become infinity, the pieces of later units are added to the bucket.  A run of more
// than SRT_LONG pieces (the top window of 255-bit scalars has 2^8 digits for 2^20 points) goes to k_srt_fix_long.
constexpr uint32_t SRT_LONG = 3;
template <int DEG>
__global__ void __launch_bounds__(64) k_srt_fix(const uint32_t* __restrict__ start, uint32_t nkeys, uint32_t nunits,
                                                const uint32_t* __restrict__ headpart, const uint32_t* __restrict__ headkey,
                                                uint32_t* __restrict__ bsum, uint32_t* __restrict__ nlong, uint32_t* __restrict__ longkeys)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    typedef SrtG<DEG> G;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, key = tid / G::LP;
    if (key >= nkeys) return;
    const uint32_t total = start[nkeys], per = (total + nunits - 1u) / nunits;
    const uint32_t s = start[key], e = start[key + 1];
    if (s == e) { G::st(G::inf(), bsum + (size_t)key * G::PJ); return; }
    const uint32_t l0 = s / per, l1 = (e - 1u) / per;
    if (l0 == l1) return;
    if (l1 - l0 > SRT_LONG) {
        if (tid % G::LP == 0u) longkeys[atomicAdd(nlong, 1u)] = key;
        return;
    }
    typename G::P acc = G::ld(bsum + (size_t)key * G::PJ);
#pragma unroll 1
    for (uint32_t l = l0 + 1; l <= l1; l++)
        if (headkey[l] == key) acc = G::add(acc, G::ld(headpart + (size_t)l * G::PJ));
    G::st(acc, bsum + (size_t)key * G::PJ);
}
#else
;
#endif

// one wavefront per long run: unit j sums the pieces l0 + 1 + j, + 64 / LP, ..; butterfly over the units; unit 0 adds the bucket
template <int DEG>
__global__ void __launch_bounds__(64) k_srt_fix_long(const uint32_t* __restrict__ start, uint32_t nkeys, uint32_t nunits,
                                                     const uint32_t* __restrict__ headpart, uint32_t* __restrict__ bsum,
                                                     const uint32_t* __restrict__ nlong, const uint32_t* __restrict__ longkeys)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    typedef SrtG<DEG> G;
    constexpr uint32_t UNITS = 64u / G::LP;
    const uint32_t unit = (threadIdx.x & 63u) / G::LP;
    const uint32_t total = start[nkeys], per = (total + nunits - 1u) / nunits, cnt = *nlong;
    for (uint32_t k = blockIdx.x; k < cnt; k += gridDim.x) {
        const uint32_t key = longkeys[k];
        const uint32_t l0 = start[key] / per, l1 = (start[key + 1] - 1u) / per;
        typename G::P acc = G::inf();
#pragma unroll 1
        for (uint32_t l = l0 + 1u + unit; l <= l1; l += UNITS) acc = G::add(acc, G::ld(headpart + (size_t)l * G::PJ));
#pragma unroll 1
        for (int off = (int)UNITS / 2; off > 0; off >>= 1) {
            typename G::P o;
            G::xor_lanes(o, acc, off);
            acc = G::add(acc, o);
        }
        if (unit == 0) {
            acc = G::add(acc, G::ld(bsum + (size_t)key * G::PJ));
            G::st(acc, bsum + (size_t)key * G::PJ);
        }
    }
}
#else
;
#endif

// unit (window w, bit b, item j): the SRT_BITADDS buckets of window w whose values |d| are the numbers m = SRT_BITADDS j .. + SRT_BITADDS - 1
// with a 1 inserted at bit b (the m-th value below 2^(cb-1) that has bit b) -> out[(w * cb + b) * nitem + j]; every unit the same
// number of additions, 2^(cb-2) / SRT_BITADDS items per (window, bit).  Bit cb - 1 is the one bucket |d| = 2^(cb-1) (item 0).
constexpr uint32_t SRT_BITADDS = 8;
template <int DEG>
__global__ void __launch_bounds__(64) k_srt_bits(const uint32_t* __restrict__ bsum, uint32_t nwin, uint32_t cb, uint32_t total,
                                                 uint32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    typedef SrtG<DEG> G;
    const uint32_t L = (blockIdx.x * blockDim.x + threadIdx.x) / G::LP;
    if (L >= total) return;
    const uint32_t kb = cb - 1u, nitem = (1u << (cb - 2u)) / SRT_BITADDS;
    const uint32_t j = L % nitem, b = (L / nitem) % cb, w = L / (nitem * cb);
    typename G::P acc = G::inf();
    if (b == kb) {
        if (j == 0u) acc = G::ld(bsum + ((size_t)(w << kb) + (1u << kb) - 1u) * G::PJ);
    } else {
#pragma unroll 1
        for (uint32_t m = j * SRT_BITADDS; m < (j + 1u) * SRT_BITADDS; m++) {
            const uint32_t v = ((m >> b) << (b + 1u)) | (1u << b) | (m & ((1u << b) - 1u));
            acc = G::add(acc, G::ld(bsum + ((size_t)(w << kb) + v - 1u) * G::PJ));
        }
    }
    G::st(acc, out + (size_t)L * G::PJ);
}
#else
;
#endif

// out[w * nfold + f] = sum of partials[w * chunks + f * per .. + per): one run per unit (k_msm_lane_fold on SrtG: G2 on lane pairs)
template <int DEG>
__global__ void __launch_bounds__(64) k_srt_fold(const uint32_t* __restrict__ partials, uint32_t chunks, uint32_t per, uint32_t nfold,
                                                 uint32_t total, uint32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    typedef SrtG<DEG> G;
    const uint32_t L = (blockIdx.x * blockDim.x + threadIdx.x) / G::LP;
    if (L >= total) return;
    const uint32_t w = L / nfold, f = L % nfold;
    const uint32_t lo = f * per, hi = min(chunks, lo + per);
    typename G::P acc = G::inf();
#pragma unroll 1
    for (uint32_t i = lo; i < hi; i++) acc = G::add(acc, G::ld(partials + ((size_t)w * chunks + i) * G::PJ));
    G::st(acc, out + ((size_t)w * nfold + f) * G::PJ);
}
#else
;
#endif

// ---- plain sums of many points (no scalars): BLS.aggregate_pub_keys / aggregate_sigs without exponents (bls.py:203-261) --------
// unit u (a lane; a lane pair for G2) adds the live points [u chunk, (u + 1) chunk) of the prepared list (k_lane_prep: x, y in the
// L28 form) with complete mixed additions -> out[u]; the partial sums are folded by k_srt_fold / the wide machine (blsgpu_api.hip
// msm_plain).  The wavefront VM's k_msm served these sums until round 5 at 50 M (G1) / 28 M (G2) points/s.
template <int DEG>
__global__ void __launch_bounds__(64, 2) k_sum_chunks(const uint32_t* __restrict__ prep, const uint8_t* __restrict__ live, uint32_t n, uint32_t chunk,
                                                      uint32_t nunits, uint32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    typedef SrtG<DEG> G;
    const uint32_t u = (blockIdx.x * blockDim.x + threadIdx.x) / G::LP;
    if (u >= nunits) return;
    const uint32_t lo = u * chunk, hi = min(n, lo + chunk);
    typename G::P acc = G::inf();
#pragma unroll 1
    for (uint32_t i = lo; i < hi; i++)
        if (live[i]) G::madd(acc, prep + (size_t)i * L28_AFF * DEG, 0u);          // (x, y) adjacent: the prepared point of the sorted buckets without its -y
    G::st(acc, out + (size_t)u * G::PJ);
}
#else
;
#endif

// ---- batches of scalar multiplications and of SMALL sums with scalars (round 5) -------------------------------------------------------
// groups x k points with k of a few (key generation, sk H(m) for many messages, pk_i e_i per message in BLS.verify's
// secure aggregation, bls.py:177-192) ran the wavefront VM's double-and-add at 0.58 M (G1) / 0.48 M (G2) scalar multiplications a
// second whatever the count.  Here ONE GROUP PER UNIT (a lane; a lane pair for G2): fixed 4-bit windows from the top, the multiples
// 0 .. 15 of each of the group's points in a table of the unit's own in HBM (fifteen mixed additions; entry 0 is infinity, so the
// loop has no branch on a digit), 63 x 4 doublings and 64 k complete additions in registers, then the affine conversion with the
// lane's own inversion.  Same value as the reference's double-and-add (fields_t.py:705-740; scalars are taken as the 256-bit
// integers they are).  Spare units of the last wavefront repeat the last group and write nothing.
constexpr uint32_t SMUL_T = 16;
template <int DEG>
__global__ void __launch_bounds__(64, 2) k_smul(const uint32_t* __restrict__ prep, const uint8_t* __restrict__ live, const uint32_t* __restrict__ scalars,
                                                uint32_t k, uint32_t groups, uint32_t* __restrict__ table, uint32_t* __restrict__ out,
                                                uint8_t* __restrict__ out_inf)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    typedef SrtG<DEG> G;
    const uint32_t unit = (blockIdx.x * blockDim.x + threadIdx.x) / G::LP;
    const bool real = unit < groups;
    const uint32_t g = real ? unit : groups - 1u;
    uint32_t* T = table + (size_t)unit * k * SMUL_T * G::PJ;
#pragma unroll 1
    for (uint32_t i = 0; i < k; i++) {
        const size_t pi = (size_t)g * k + i;
        const bool lv = live[pi] != 0;
        typename G::P m = G::inf();
        G::st(m, T + (size_t)i * SMUL_T * G::PJ);
#pragma unroll 1
        for (uint32_t d = 1; d < SMUL_T; d++) {
            if (lv) G::madd(m, prep + pi * L28_AFF * DEG, 0u);
            G::st(m, T + ((size_t)i * SMUL_T + d) * G::PJ);
        }
    }
    typename G::P acc = G::inf();
#pragma unroll 1
    for (int w = 63; w >= 0; w--) {
        if (w != 63) {
#pragma unroll 1
            for (int t = 0; t < 4; t++) acc = G::dbl(acc);
        }
#pragma unroll 1
        for (uint32_t i = 0; i < k; i++) {
            const uint32_t word = bswap32(scalars[((size_t)g * k + i) * 8 + 7u - ((uint32_t)w >> 3)]);
            const uint32_t d = (word >> (((uint32_t)w & 7u) * 4u)) & 15u;
            acc = G::add(acc, G::ld(T + ((size_t)i * SMUL_T + d) * G::PJ));
        }
    }
    G::affine_out(acc, out + (size_t)g * 24 * DEG, out_inf ? out_inf + g : nullptr, real);
}
#else
;
#endif

// W_w = sum_b 2^b S_(w,b): one team per window, Horner over the bits
__global__ void __launch_bounds__(64) k_srt_windows(VmTables T, const uint32_t* __restrict__ bitsums, uint32_t cb, uint32_t* __restrict__ winsums)
#if BLSGPU_EMIT(BLSGPU_TU_MSM)
{
    using C = MsmCfg<1>;
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    team_init_consts(T, team, lane);
    wave_fence();
    const uint32_t* src = bitsums + (size_t)blockIdx.x * cb * 36;
    for (uint32_t d = lane; d < 36; d += 64) team[C::PR0 * 12 + d] = src[(size_t)(cb - 1u) * 36 + d];
    wave_fence();
    for (int b = (int)cb - 2; b >= 0; b--) {
        run_rounds<true>(T, T.segflat + C::DBL_OFF, C::DBL_LEN, 0, lane);
        for (uint32_t d = lane; d < 36; d += 64) team[C::PR1 * 12 + d] = src[(size_t)b * 36 + d];
        wave_fence();
        run_rounds<true>(T, T.segflat + C::PADD_OFF, C::PADD_LEN, 0, lane);
    }
    wave_fence();
    for (uint32_t d = lane; d < 36; d += 64) winsums[(size_t)blockIdx.x * 36 + d] = team[C::PR0 * 12 + d];
}
#else
;
#endif

// every instantiation the host side launches: this translation unit is the one that emits them (blsgpu_tu.h)
#if BLSGPU_TU == BLSGPU_TU_MSM
__attribute__((used)) static const void* const blsgpu_instances_msm[] = {
    (const void*)&k_msm<1>,
    (const void*)&k_msm<2>,
    (const void*)&k_msm_finish<1>,
    (const void*)&k_msm_finish<2>,
    (const void*)&k_msm_prep<1>,
    (const void*)&k_msm_prep<2>,
    (const void*)&k_msm_pip<1>,
    (const void*)&k_msm_pip<2>,
    (const void*)&k_msm_pip_windows<1>,
    (const void*)&k_msm_pip_windows<2>,
    (const void*)&k_msm_pip_horner<1>,
    (const void*)&k_msm_pip_horner<2>,
    (const void*)&k_lane_prep<1>,
    (const void*)&k_srt_prep<1>, (const void*)&k_srt_prep<2>,
    (const void*)&k_srt_accum<1>, (const void*)&k_srt_accum<2>, (const void*)&k_srt_fix<1>, (const void*)&k_srt_fix<2>,
    (const void*)&k_srt_fix_long<1>, (const void*)&k_srt_fix_long<2>, (const void*)&k_srt_bits<1>, (const void*)&k_srt_bits<2>,
    (const void*)&k_srt_fold<1>, (const void*)&k_srt_fold<2>, (const void*)&k_sum_chunks<1>, (const void*)&k_sum_chunks<2>, (const void*)&k_smul<1>, (const void*)&k_smul<2>,
    (const void*)&k_lane_prep<2>,
    (const void*)&k_msm_lane<1>,
    (const void*)&k_msm_lane<2>,
    (const void*)&k_msm_lane_fold<1>,
    (const void*)&k_msm_lane_fold<2>};
#endif
}  // namespace blsgpu
