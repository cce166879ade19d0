// blsgpu_kernels.hip -- gfx950 kernels of the BLS12-381 multi-pairing path.
//
// Replaces, behind the C ABI of include/blsgpu.h, the reference's native
// fq_ate_pairing_multi (extmod/bls_py/fields_t_c.pyx:2333-2391; semantic
// definition bls_py/fields_t.py:1114-1128).
//
// Execution model: ONE WAVEFRONT PER PAIRING.  A wavefront ("team") keeps every
// Fq value of its pairing in a private LDS scratchpad (BLSVM_TEAM_SLOTS slots of
// 48 bytes, Montgomery limbs) and interprets the statically scheduled rounds of
// vm_tables.h (generated and CPU-verified by python-bls_amd/vmgen):
//   MUL round: each active lane loads two slots, does one 12x12-limb Montgomery
//              product (v_mad_u64_u32 chains) and stores one slot;
//   LIN round: each active lane accumulates +/- slots (modular add chains);
//   INV round: field inversion (one lane, once per final exponentiation).
// Lanes never diverge on data: all control flow below is wavefront-uniform.
// Within a team LDS traffic is ordered by the hardware's in-order DS queue, so
// no barrier is needed between rounds; workgroup barriers only separate the
// cross-team product tree.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fq32.h"
#include "vm_tables.h"

namespace blsgpu {

struct VmTables {
    const uint2* segs;        // {first_round, nrounds}
    const uint2* rounds;      // {data_off, kind | K << 8 | nlanes << 16}
    const uint16_t* data;
    const uint16_t* mscript;
    const uint16_t* fscript;
    const uint32_t* consts;   // BLSVM_NCONST x 12 limbs
};

constexpr int TEAM_DW = BLSVM_TEAM_SLOTS * 12;           // dwords per team
constexpr int TEAM_BYTES = TEAM_DW * 4;
constexpr int F_DW = BLSVM_SLOT_REG0 * 12;               // register 0 (accumulator)
constexpr int R1_DW = (BLSVM_SLOT_REG0 + 12) * 12;       // register 1

extern __shared__ uint4 smem4[];

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void lds_load12(uint32_t* x, uint32_t idx16) {
    uint4 a = smem4[idx16], b = smem4[idx16 + 1], c = smem4[idx16 + 2];
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
    x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    x[8] = c.x; x[9] = c.y; x[10] = c.z; x[11] = c.w;
}
__device__ __forceinline__ void lds_store12(const uint32_t* x, uint32_t idx16) {
    smem4[idx16] = make_uint4(x[0], x[1], x[2], x[3]);
    smem4[idx16 + 1] = make_uint4(x[4], x[5], x[6], x[7]);
    smem4[idx16 + 2] = make_uint4(x[8], x[9], x[10], x[11]);
}

// Run one scheduled segment on the team whose scratchpad starts at base16
// (in 16-byte units).  seg must be wavefront-uniform.
__device__ __noinline__ void run_segment(const VmTables& T, uint32_t seg, uint32_t base16, uint32_t lane) {
    const uint32_t q[12] = BLS_Q_LIMBS;
    seg = __builtin_amdgcn_readfirstlane(seg);
    uint2 si = T.segs[seg];
    uint32_t r0 = __builtin_amdgcn_readfirstlane(si.x), r1 = r0 + __builtin_amdgcn_readfirstlane(si.y);
    for (uint32_t r = r0; r < r1; ++r) {
        uint2 ri = T.rounds[r];
        uint32_t off = __builtin_amdgcn_readfirstlane(ri.x);
        uint32_t meta = __builtin_amdgcn_readfirstlane(ri.y);
        uint32_t kind = meta & 0xffu, K = (meta >> 8) & 0xffu;
        const uint16_t* d = T.data + off;
        if (kind == 0u) {                                    // MUL
            ushort4 e = reinterpret_cast<const ushort4*>(d)[lane];
            uint32_t A[12], B[12], D[12];
            lds_load12(A, base16 + e.x);
            lds_load12(B, base16 + e.y);
            bls::fq_mul(D, A, B);
            if (e.z != 0xFFFFu) lds_store12(D, base16 + e.z);
        } else if (kind == 1u) {                             // LIN
            uint32_t acc[12];
#pragma unroll
            for (int j = 0; j < 12; j++) acc[j] = 0;
            for (uint32_t k = 0; k < K; ++k) {
                uint32_t u = d[k * 64 + lane];
                uint32_t op = u >> 14;
                uint32_t S[12];
                lds_load12(S, base16 + (u & 0x3FFFu));       // NOP / DBL reference slot 0 = ZERO
                bool is_dbl = (op == 2u), is_sub = (op == 1u);
                uint32_t N[12];
                uint32_t br = 0;
#pragma unroll
                for (int j = 0; j < 12; j++) {
                    uint32_t s = is_dbl ? acc[j] : S[j];
                    uint64_t x = (uint64_t)q[j] - s - br;
                    N[j] = (uint32_t)x;
                    br = (uint32_t)(x >> 63);
                    S[j] = s;
                }
#pragma unroll
                for (int j = 0; j < 12; j++) S[j] = is_sub ? N[j] : S[j];
                bls::fq_add_mod(acc, S);
            }
            uint32_t dst = d[K * 64 + lane];
            if (dst != 0xFFFFu) lds_store12(acc, base16 + dst);
        } else {                                             // INV
            ushort4 e = reinterpret_cast<const ushort4*>(d)[lane];
            if (e.z != 0xFFFFu) {
                uint32_t A[12], D[12];
                lds_load12(A, base16 + e.x);
                bls::fq_inv(D, A);
                lds_store12(D, base16 + e.z);
            }
        }
        wave_fence();
    }
}

__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }

// copy the constant slots into a team's scratchpad
__device__ __forceinline__ void team_init_consts(const VmTables& T, uint32_t* team, uint32_t lane) {
    for (uint32_t i = lane; i < BLSVM_NCONST * 12; i += 64) team[i] = T.consts[i];
}
// accumulator register 0 <- 1 (one != 0) or 0
__device__ __forceinline__ void team_set_acc(uint32_t* team, uint32_t lane, bool one) {
    for (uint32_t i = lane; i < 144; i += 64) {
        uint32_t v = 0;
        if (one && i < 12) v = team[BLSVM_SLOT_C_ONE * 12 + i];
        team[F_DW + i] = v;
    }
}

// In-workgroup product tree over the teams' accumulators; on return team 0's
// register 0 holds the product.  All waves of the workgroup must call this.
__device__ __forceinline__ void wg_product_tree(const VmTables& T, uint32_t* smem, uint32_t wave, uint32_t nwaves, uint32_t lane) {
    uint32_t* team = smem + wave * TEAM_DW;
    for (uint32_t s = 1; s < nwaves; s <<= 1) {
        __syncthreads();
        if ((wave % (2 * s)) == 0 && wave + s < nwaves) {
            const uint32_t* other = smem + (wave + s) * TEAM_DW + F_DW;
            for (uint32_t i = lane; i < 144; i += 64) team[R1_DW + i] = other[i];
            wave_fence();
            run_segment(T, BLSVM_SEG_MUL_0_1, wave * (TEAM_BYTES / 16), lane);
        }
    }
}

// ---------------------------------------------------------------------------
// Kernel 1: one wavefront per (P, Q) pair: Miller loop, then the product of the
// workgroup's Miller values -> one Montgomery Fq12 partial (144 u32) per block.
//   g1: n x 96 bytes (x || y), g2: n x 192 bytes (x.c0 || x.c1 || y.c0 || y.c1),
//   big-endian canonical coordinates (the reference's serialisation, fields.py:87-88).
// Infinity follows the reference's observable behaviour (tests/golden/pairing.json
// "edge"): coordinates (0,0) for Q contribute 1 (0 if P.y is 0 too), (0,0) for P
// contributes 1.
__global__ void __launch_bounds__(256) k_miller(VmTables T, const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                                uint32_t n, uint32_t* __restrict__ partials) {
    uint32_t* smem = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t pair = blockIdx.x * nwaves + wave;
    uint32_t* team = smem + wave * TEAM_DW;
    const uint32_t base16 = wave * (TEAM_BYTES / 16);
    team_init_consts(T, team, lane);
    if (pair < n) {
        // coalesced load of the 72 big-endian dwords of the pair
        uint32_t w0, w1 = 0;
        {
            uint32_t d = lane;
            w0 = (d < 24) ? g1[(size_t)pair * 24 + d] : g2[(size_t)pair * 48 + (d - 24)];
            if (lane < 8) w1 = g2[(size_t)pair * 48 + (lane + 40)];
        }
        uint64_t nz0 = __ballot(w0 != 0);
        uint64_t nz1 = __ballot(lane < 8 && w1 != 0);
        {
            uint32_t d = lane, e = d / 12, w = d % 12;
            team[(BLSVM_SLOT_PX + e) * 12 + (11 - w)] = bswap32(w0);
            if (lane < 8) {
                d = lane + 64; e = d / 12; w = d % 12;
                team[(BLSVM_SLOT_PX + e) * 12 + (11 - w)] = bswap32(w1);
            }
        }
        const bool p_zero = (nz0 & 0xFFFFFFull) == 0;
        const bool py_zero = (nz0 & 0xFFF000ull) == 0;
        const bool q_zero = (nz0 >> 24) == 0 && nz1 == 0;
        wave_fence();
        if (q_zero) {
            team_set_acc(team, lane, !py_zero);
        } else if (p_zero) {
            team_set_acc(team, lane, true);
        } else {
            for (uint32_t i = 0; i < BLSVM_MILLER_LEN; ++i) run_segment(T, T.mscript[i], base16, lane);
        }
    } else {
        wave_fence();
        team_set_acc(team, lane, true);
    }
    wave_fence();
    wg_product_tree(T, smem, wave, nwaves, lane);
    if (wave == 0) {
        wave_fence();
        for (uint32_t i = lane; i < 144; i += 64) partials[(size_t)blockIdx.x * 144 + i] = team[F_DW + i];
    }
}

// ---------------------------------------------------------------------------
// Kernel 2: product of Montgomery Fq12 partials.  Block b multiplies partials
// [b * per_block, min(m, (b+1) * per_block)) and writes one partial.  When
// do_final != 0 (single block) the product additionally goes through the final
// exponentiation and is written as 576 big-endian bytes (12 x 48, flat ZT
// order of fields.py:624-629) to out_bytes.
__global__ void __launch_bounds__(512) k_reduce(VmTables T, const uint32_t* __restrict__ in, uint32_t m, uint32_t per_block,
                                                 uint32_t* __restrict__ out_partials, uint32_t do_final,
                                                 uint32_t* __restrict__ out_bytes) {
    uint32_t* smem = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = blockDim.x >> 6;
    uint32_t* team = smem + wave * TEAM_DW;
    const uint32_t base16 = wave * (TEAM_BYTES / 16);
    team_init_consts(T, team, lane);
    wave_fence();
    const uint32_t lo = blockIdx.x * per_block;
    const uint32_t hi = min(m, lo + per_block);
    bool first = true;
    for (uint32_t i = lo + wave; i < hi; i += nwaves) {
        const uint32_t* src = in + (size_t)i * 144;
        uint32_t dst = first ? F_DW : R1_DW;
        for (uint32_t k = lane; k < 144; k += 64) team[dst + k] = src[k];
        wave_fence();
        if (!first) run_segment(T, BLSVM_SEG_MUL_0_1, base16, lane);
        first = false;
    }
    if (first) team_set_acc(team, lane, true);
    wave_fence();
    wg_product_tree(T, smem, wave, nwaves, lane);
    if (wave == 0) {
        wave_fence();
        if (do_final) {
            for (uint32_t i = 0; i < BLSVM_FEXP_LEN; ++i) run_segment(T, T.fscript[i], base16, lane);
            run_segment(T, BLSVM_SEG_FROM_MONT_1_0, base16, lane);
            for (uint32_t k = lane; k < 144; k += 64) {
                uint32_t c = k / 12, w = k % 12;
                out_bytes[k] = bswap32(team[R1_DW + c * 12 + (11 - w)]);
            }
        } else {
            for (uint32_t k = lane; k < 144; k += 64) out_partials[(size_t)blockIdx.x * 144 + k] = team[F_DW + k];
        }
    }
}

}  // namespace blsgpu
