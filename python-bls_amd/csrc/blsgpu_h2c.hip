// blsgpu_h2c.hip -- hash-to-G2 after the SHA-256 step (included by blsgpu_api.hip).
//
// Replaces, for batches, the reference's hash_to_point_prehashed_Fq2 (ec.py:528-550)
// from the field elements t0, t1 onwards: sw_encode (ec.py:449-507) twice, the sum,
// and the psi-based cofactor clearing.  Branch-free restatement in
// vmgen/h2c_programs.py; the SHA-256/hash512 chain stays on the host.
#pragma once

namespace blsgpu {

constexpr int H1_TEAM_DW = BLSVM_H1_SLOTS * 12;
constexpr int H2_TEAM_DW = BLSVM_H2_SLOTS * 12;

__device__ __forceinline__ void team_init_consts_h2c(const VmTables& T, uint32_t* team, uint32_t lane) {
    for (uint32_t i = lane; i < BLSVM_NCONST_H2C * 12; i += 64) team[i] = T.consts[i];
}

// Kernel H1: one team = BLSVM_H1_NE encodings.  t: n_enc x 96 bytes (c0 || c1,
// big-endian canonical); out: n_enc x 60 u32 = (x, y, z.c0) in Montgomery limbs, z = 0 for infinity.
__global__ void __launch_bounds__(64, 2) k_h2c_encode(VmTables T, const uint32_t* __restrict__ t, uint32_t n_enc,
                                                      uint32_t* __restrict__ out) {
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t first = blockIdx.x * BLSVM_H1_NE;
    team_init_consts_h2c(T, team, lane);
    for (uint32_t d = lane; d < BLSVM_H1_NE * 24; d += 64) {
        uint32_t e = d / 24, o = d % 24, c = o / 12, w = o % 12;
        // encodings past the end run on t = (1, 0); their results are dropped
        uint32_t v = (first + e < n_enc) ? bswap32(t[(size_t)(first + e) * 24 + o]) : ((c == 0 && w == 11) ? 1u : 0u);
        team[(BLSVM_H1_T + 2 * e + c) * 12 + (11 - w)] = v;
    }
    wave_fence();
    run_rounds(T, T.h1flat, BLSVM_H1_FLAT_LEN, 0, lane);
    for (uint32_t d = lane; d < BLSVM_H1_NE * 60; d += 64) {
        uint32_t e = d / 60;
        if (first + e < n_enc) out[(size_t)first * 60 + d] = team[BLSVM_H1_S * 12 + d];
    }
}

// Kernel H2: one team = BLSVM_H2_NM messages: P = S0 + S1, cofactor clearing,
// canonical affine bytes (x.c0 || x.c1 || y.c0 || y.c1, 192 B per message).
__global__ void __launch_bounds__(64, 3) k_h2c_clear(VmTables T, const uint32_t* __restrict__ enc, uint32_t n_msg,
                                                     uint32_t* __restrict__ out) {
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t first = blockIdx.x * BLSVM_H2_NM;
    team_init_consts_h2c(T, team, lane);
    for (uint32_t d = lane; d < BLSVM_H2_NM * 120; d += 64) {
        uint32_t m = d / 120;
        uint32_t src_m = (first + m < n_msg) ? first + m : first;       // pad with a valid message
        team[BLSVM_H2_S * 12 + d] = enc[(size_t)src_m * 120 + (d % 120)];
    }
    wave_fence();
    run_rounds(T, T.h2flat, BLSVM_H2_FLAT_LEN, 0, lane);
    if (lane < 4u * BLSVM_H2_NM) {
        uint32_t X[12];
        lds_load12(X, (BLSVM_H2_OUT + lane) * 3);
        bls::fq_canon(X);
        lds_store12(X, (BLSVM_H2_OUT + lane) * 3);
    }
    wave_fence();
    for (uint32_t d = lane; d < BLSVM_H2_NM * 48; d += 64) {
        uint32_t m = d / 48, o = d % 48, e = o / 12, w = o % 12;
        if (first + m < n_msg) out[(size_t)(first + m) * 48 + o] = bswap32(team[(BLSVM_H2_OUT + 4 * m + e) * 12 + (11 - w)]);
    }
}

// ---------------------------------------------------------------------------
// Point decompression: PublicKey.from_bytes (keys.py:28-40, DEG 1)
// and Signature.from_bytes (signature.py:21-38, DEG 2) for a batch.  in: n x 48*DEG
// bytes as serialised (bit 0x80 of byte 0 = "the larger y", top three bits masked as
// `& 0x1f`); out: n x 96*DEG bytes affine (x, y) canonical big-endian; ok[i] = 1 iff
// the reference would have accepted the encoding (it raises ValueError otherwise;
// the point bytes are then unspecified).  Programs: vmgen/decomp_programs.py.
template <int DEG> struct DecompCfg;
template <> struct DecompCfg<1> {
    static constexpr int NE = BLSVM_D1_NE, SLOTS = BLSVM_D1_SLOTS, X = BLSVM_D1_X, BIG = BLSVM_D1_BIG, OUT = BLSVM_D1_OUT,
                         LEN = BLSVM_D1_FLAT_LEN;
    static __device__ __forceinline__ const uint2* flat(const VmTables& T) { return T.d1flat; }
};
template <> struct DecompCfg<2> {
    static constexpr int NE = BLSVM_D2_NE, SLOTS = BLSVM_D2_SLOTS, X = BLSVM_D2_X, BIG = BLSVM_D2_BIG, OUT = BLSVM_D2_OUT,
                         LEN = BLSVM_D2_FLAT_LEN;
    static __device__ __forceinline__ const uint2* flat(const VmTables& T) { return T.d2flat; }
};

template <int DEG>
__global__ void __launch_bounds__(64, 2) k_decompress(VmTables T, const uint32_t* __restrict__ in, uint32_t n,
                                                      uint32_t* __restrict__ out, uint8_t* __restrict__ ok) {
    using C = DecompCfg<DEG>;
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t first = blockIdx.x * C::NE;
    team_init_consts_h2c(T, team, lane);
    wave_fence();
    constexpr uint32_t DW_IN = 12 * DEG;                       // dwords per encoded point
    for (uint32_t d = lane; d < C::NE * DW_IN; d += 64) {
        const uint32_t e = d / DW_IN, o = d % DW_IN, c = o / 12, w = o % 12;
        // points past the end decode x = 0 (any value would do; their results are dropped)
        uint32_t v = (first + e < n) ? bswap32(in[(size_t)(first + e) * DW_IN + o]) : 0u;
        if (o == 0) {                                          // byte 0 carries the flags
            team[(C::BIG + e) * 12] = (v >> 31) ? 1u : 0u;     // raw 1 / 0, made Montgomery below
            v &= 0x1FFFFFFFu;
        }
        team[(C::X + DEG * e + c) * 12 + (11 - w)] = v;
    }
    for (uint32_t d = lane; d < C::NE * 11; d += 64) team[(C::BIG + d / 11) * 12 + 1 + d % 11] = 0u;
    wave_fence();
    if (lane < (uint32_t)C::NE) {                              // flag -> Montgomery 0 / 1
        const uint32_t onem[12] = BLS_ONE_MONT_LIMBS;
        const bool big = team[(C::BIG + lane) * 12] != 0u;
        for (int j = 0; j < 12; j++) team[(C::BIG + lane) * 12 + j] = big ? onem[j] : 0u;
    }
    wave_fence();
    run_rounds(T, C::flat(T), C::LEN, 0, lane);
    constexpr uint32_t NOUT = 2 * DEG + 1;                     // x, y coordinates and the flag
    for (uint32_t d = lane; d < C::NE * NOUT; d += 64) {       // relaxed -> canonical residues
        uint32_t X[12];
        lds_load12(X, (C::OUT + d) * 3);
        bls::fq_canon(X);
        lds_store12(X, (C::OUT + d) * 3);
    }
    wave_fence();
    constexpr uint32_t DW_OUT = 24 * DEG;
    for (uint32_t d = lane; d < C::NE * DW_OUT; d += 64) {
        const uint32_t e = d / DW_OUT, o = d % DW_OUT, c = o / 12, w = o % 12;
        if (first + e < n) out[(size_t)(first + e) * DW_OUT + o] = bswap32(team[(C::OUT + NOUT * e + c) * 12 + (11 - w)]);
    }
    if (lane < (uint32_t)C::NE && first + lane < n) ok[first + lane] = (uint8_t)(team[(C::OUT + NOUT * lane + 2 * DEG) * 12] & 1u);
}

#ifdef BLSGPU_STAMPS
// diagnostic build only: see blsgpu_debug_run
__global__ void __launch_bounds__(64) k_debug_run(VmTables T, const uint2* seq, uint32_t nrounds, uint32_t nslots, uint32_t* image) {
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < nslots * 12; i += 64) team[i] = image[i];
    wave_fence();
    run_rounds(T, seq, nrounds, 0, lane);
    wave_fence();
    for (uint32_t i = lane; i < nslots * 12; i += 64) image[i] = team[i];
}
#endif

}  // namespace blsgpu
