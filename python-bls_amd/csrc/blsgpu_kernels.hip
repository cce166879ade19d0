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
//   LIN round: each active lane accumulates coef * slot (or coef * ~slot for a
//              negative term) limb by limb into 64-bit "fat" limbs -- independent
//              v_mad_u64_u32, no carry chain, no VCC hazard -- and reduces once;
//   values are kept "relaxed" (< 2q): products need no final subtraction and a
//   linear combination is reduced with a single quotient estimate (fq32.h);
//   INV round: field inversion (one lane, once per final exponentiation).
// Lanes never diverge on data: all control flow below is wavefront-uniform.
// Within a team LDS traffic is ordered by the hardware's in-order DS queue, so
// no barrier is needed between rounds; workgroup barriers only separate the
// cross-team product tree.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fq32.h"
#include "fp28.h"
#include "vm_tables.h"

namespace blsgpu {

struct VmTables {
    const uint2* mflat;       // Miller loop: flat round sequence {data_off, meta}
    const uint2* mpflat;      // Miller loop, BLSVM_MP_G pairs per team
    const uint2* mp2flat;     // Miller loop, two pairs per team
    const uint2* h2flat;      // hash to G2: sum + cofactor clearing
    const uint2* fflat;       // final exponentiation
    const uint2* sflat;       // reference-faithful Miller loop (vmgen/slow_programs.py)
    const uint2* segflat;     // directly called segments (BLSVM_SEGF_*)
    const uint16_t* data;
    const uint32_t* consts;   // BLSVM_NCONST x 12 limbs
    unsigned long long* stamps;   // diagnostic builds only (-DBLSGPU_STAMPS), else unused
};

constexpr int LIN_CHUNKS = 8;                           // a LIN record holds <= 30 micro-ops
static_assert(BLSVM_MAX_LIN_K <= 4 * LIN_CHUNKS - 2, "LIN record too long");
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

// per-lane record length of a round, in u16 units (wave-uniform)
__device__ __forceinline__ uint32_t rec_len(uint32_t meta) {
    if ((meta & 3u) != 1u) return 4u;
    return (((meta >> 8) & 0xFFu) + 2u + 3u) & ~3u;      // destination, merge flags, K micro-ops
}

// global-address-space views: a pointer taken out of the kernel-argument struct
// is generic, and generic (flat) loads tie up the LDS wait counter
typedef const __attribute__((address_space(1))) uint64_t* gptr_u2;     // 8-byte table entries
typedef const __attribute__((address_space(1))) uint16_t* gptr_u16;
__device__ __forceinline__ uint2 ld2(gptr_u2 p, uint32_t i) {
    uint64_t v = p[i];
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

// sequential reader of a lane's u16 record, 4 entries (8 bytes) per load,
// with the next chunk requested one chunk ahead
struct RecReader {
    gptr_u2 p;
    uint2 cur, nxt;
    uint32_t pos, nchunks;
    __device__ __forceinline__ void init(gptr_u2 rec, uint2 first, uint32_t len_u16) {
        p = rec;
        cur = first;
        pos = 0;
        nchunks = len_u16 >> 2;
        nxt = (nchunks > 1) ? ld2(p, 1) : first;
    }
    __device__ __forceinline__ uint32_t next() {
        uint32_t sub = pos & 3u;                 // wave-uniform
        uint32_t w = (sub & 2u) ? cur.y : cur.x;
        uint32_t v = (sub & 1u) ? (w >> 16) : (w & 0xFFFFu);
        ++pos;
        if ((pos & 3u) == 0u) {
            uint32_t c = pos >> 2;
            cur = nxt;
            if (c + 1 < nchunks) nxt = ld2(p, c + 1);
        }
        return v;
    }
};

template <int CTRL>
__device__ __forceinline__ void dpp_quad_absorb(uint64_t& acc, uint32_t mask);
#define BLSGPU_DPP_ABSORB(CTRL, PERM)                                                                            \
    template <>                                                                                                  \
    __device__ __forceinline__ void dpp_quad_absorb<CTRL>(uint64_t& acc, uint32_t mask) {                        \
        uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32), t0, t1;                                        \
        asm volatile("v_and_b32_dpp %2, %0, %4 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  \
                     "v_and_b32_dpp %3, %1, %4 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"  \
                     "v_add_co_u32 %0, vcc, %0, %2\n\t"                                                          \
                     "v_addc_co_u32 %1, vcc, %1, %3, vcc"                                                        \
                     : "+v"(lo), "+v"(hi), "=&v"(t0), "=&v"(t1)                                                  \
                     : "v"(mask)                                                                                 \
                     : "vcc");                                                                                   \
        acc = ((uint64_t)hi << 32) | lo;                                                                         \
    }
// acc += (the accumulator of another lane of the same quad) & mask, mask = 0 or ~0 of the
// READING lane (the value if the lane absorbs it, else 0): four instructions per limb
BLSGPU_DPP_ABSORB(0xF5, "[1,1,3,3]")       // lanes 0,2 of a quad += lanes 1,3
BLSGPU_DPP_ABSORB(0xAA, "[2,2,2,2]")       // lane 0 += lane 2
#undef BLSGPU_DPP_ABSORB

// Walk `n` rounds of a flat sequence on the team whose scratchpad starts at
// base16 (16-byte units).  Round headers are read three rounds ahead and every
// lane record one round ahead, so that table latency overlaps the arithmetic of
// the previous round.
// LIGHT = the program holds MUL and LIN rounds only (Miller loops, group sums):
// the inversion and sign code is left out of the kernel.
// (in the multi-pair Miller programs kinds 2 / 3 are SAVE / RESTORE: the 12-slot window at
// slot K of the round header <-> `stash`, three registers per lane -- programs.MPLayout.)
// CHK: a wave-uniform test made once, after the first `check_at` rounds (the Miller kernels look
// at the on-curve residual their first segment leaves); a false result ends the walk there and
// is returned.  One call site instead of two keeps a single copy of the interpreter per program.
struct NoCheck { __device__ __forceinline__ bool operator()() const { return true; } };
template <bool LIGHT = false, class CHK = NoCheck>
__device__ __forceinline__ bool run_rounds(const VmTables& T, const uint2* __restrict__ seq_, uint32_t n, uint32_t base16, uint32_t lane,
                                           uint32_t* stash = nullptr, uint32_t check_at = 0xFFFFFFFFu, CHK chk = CHK()) {
    if (n == 0) return true;
    gptr_u2 seq = (gptr_u2)seq_;
    gptr_u16 gdata = (gptr_u16)T.data;
    uint2 h0 = ld2(seq, 0);
    uint2 h1 = ld2(seq, n > 1 ? 1 : 0);
    // header of round i + 2 waits in VGPRs (hraw) for a whole round before it is made
    // wave-uniform: readfirstlane right after the load would stall on it every round
    uint2 hraw = ld2(seq, n > 2 ? 2 : 0);
    h0.x = __builtin_amdgcn_readfirstlane(h0.x); h0.y = __builtin_amdgcn_readfirstlane(h0.y);
    h1.x = __builtin_amdgcn_readfirstlane(h1.x); h1.y = __builtin_amdgcn_readfirstlane(h1.y);
#if defined(BLSGPU_EXP) && (BLSGPU_EXP & 32)
#define BLSGPU_DOFF(h) ((h).x & 0xFFFu)                      /* timing experiment only: records from one hot 8 KB window */
#else
#define BLSGPU_DOFF(h) ((h).x)
#endif
    // the lane's WHOLE record of the next round (<= LIN_CHUNKS x 4 u16) is fetched
    // during the current round; positions are compile-time so every value stays in
    // a fixed register
    uint2 nx[LIN_CHUNKS];
    {
        gptr_u2 rec = (gptr_u2)(gdata + BLSGPU_DOFF(h0) + lane * rec_len(h0.y));
        const uint32_t nch0 = rec_len(h0.y) >> 2;
#pragma unroll
        for (int c = 0; c < LIN_CHUNKS; c++) nx[c] = (c < (int)nch0) ? ld2(rec, c) : make_uint2(0u, 0u);
    }
#ifdef BLSGPU_STAMPS
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_cnt[4] = {0, 0, 0, 0}, st_lin[3] = {0, 0, 0};
#endif
    uint32_t i = 0;                                          // rounds started (advanced inside `round`)
    // one round: `ch` = this round's lane record (fetched during the previous round),
    // `nxt` receives the next round's
    auto round = [&](const uint2 (&ch)[LIN_CHUNKS], uint2 (&nxt)[LIN_CHUNKS]) __attribute__((always_inline)) {
#ifdef BLSGPU_STAMPS
        unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t meta = h0.y;
        const uint2 e_cur = ch[0];
        // ---- prefetch for the following rounds
        h0 = h1;
        h1.x = __builtin_amdgcn_readfirstlane(hraw.x); h1.y = __builtin_amdgcn_readfirstlane(hraw.y);
        if (i + 3 < n) hraw = ld2(seq, i + 3);
        ++i;
        {
            // (after the last round h0 still holds a valid, older header: the fetch is harmless)
            gptr_u2 rec = (gptr_u2)(gdata + BLSGPU_DOFF(h0) + lane * rec_len(h0.y));
            const uint32_t nchn = rec_len(h0.y) >> 2;
#pragma unroll
            for (int c = 0; c < LIN_CHUNKS; c++) {
                // chunks past the record are never read: leave their registers as they are
                // instead of carrying a value around the loop (a register copy per chunk and round)
                uint2 any;
                asm volatile("" : "=v"(any.x), "=v"(any.y));
                nxt[c] = (c < (int)nchn) ? ld2(rec, c) : any;
            }
        }
        const uint32_t kind = meta & 3u;
        if (kind == 0u) {                                    // MUL
            uint32_t ra = e_cur.x & 0xFFFFu, rb = e_cur.x >> 16, rd = e_cur.y & 0xFFFFu;
            uint32_t A[12], B[12], D[12];
            lds_load12(A, base16 + ra);
            lds_load12(B, base16 + rb);
#if defined(BLSGPU_EXP) && (BLSGPU_EXP & 1)
            for (int j = 0; j < 12; j++) D[j] = A[j] ^ B[j];  // timing experiment only
#else
#if defined(BLSGPU_MUL32)
            bls::fq_mul_relaxed(D, A, B);                     // rounds 1-2: v_mad_u64_u32 + v_addc_co_u32 columns
#else
            r28::vm_mul28(D, A, B);                           // carry-free 28-bit limbs inside the round (fp28.h)
#endif
#endif
            if (rd != 0xFFFFu) lds_store12(D, base16 + rd);
        } else if (kind == 1u) {                             // LIN
#if defined(BLSGPU_EXP) && (BLSGPU_EXP & 2)
            const uint32_t K = ((meta >> 8) & 0xFFu) ? 1u : 0u;   // timing experiment only
#else
            const uint32_t K = (meta >> 8) & 0xFFu;
#endif
#ifdef BLSGPU_STAMPS
            unsigned long long lt0 = __builtin_amdgcn_s_memtime();
#endif
            const uint32_t rd = ch[0].x & 0xFFFFu;
            uint64_t acc[12];
#pragma unroll
            for (int j = 0; j < 12; j++) acc[j] = 0;
            // record: destination, merge flags, micro-ops; micro-op p sits at
            // position p + 2; operands are fetched one micro-op ahead into two
            // alternating buffers
            uint32_t S[2][12];
#define BLSGPU_UOP(p) ((((p) + 2) & 1) ? ((((p) + 2) & 2) ? (ch[((p) + 2) >> 2].y >> 16) : (ch[((p) + 2) >> 2].x >> 16)) \
                                       : ((((p) + 2) & 2) ? (ch[((p) + 2) >> 2].y & 0xFFFFu) : (ch[((p) + 2) >> 2].x & 0xFFFFu)))
#if defined(BLSGPU_EXP) && (BLSGPU_EXP & 8)
#define BLSGPU_SLOTMASK 3u                                   /* timing experiment only: 4 hot slots */
#else
#define BLSGPU_SLOTMASK 1023u
#endif
            // K >= 1 (emit.py).  Micro-ops 0 .. MN-1 are the negative terms (summed as plain
            // products), then every lane flips the sign of its accumulators, then the positive
            // terms.  The loop leaves through `break`, so every use of an operand buffer is
            // dominated by its load; the look-ahead load past the last micro-op reads slot 0.
            const uint32_t MN = (meta >> 18) & 0xFFu;
            lds_load12(S[0], base16 + (BLSGPU_UOP(0) & BLSGPU_SLOTMASK) * 3u);
#pragma unroll
            for (int p = 0; p < 4 * LIN_CHUNKS - 2; p++) {
                if (p + 1 < 4 * LIN_CHUNKS - 2) {
                    const uint32_t live = ((uint32_t)(p + 1) < K) ? BLSGPU_SLOTMASK : 0u;
                    lds_load12(S[(p + 1) & 1], base16 + (BLSGPU_UOP(p + 1) & live) * 3u);
                }
                if (p > 0 && (uint32_t)p == MN) bls::fat_flip(acc);
                const uint32_t u = BLSGPU_UOP(p);
#if defined(BLSGPU_EXP) && (BLSGPU_EXP & 16)
                for (int j = 0; j < 12; j++) acc[j] ^= S[p & 1][j];          // timing experiment only
#else
                bls::fat_mac_plain(acc, S[p & 1], (u >> 10) & 31u);
#endif
                if ((uint32_t)(p + 1) >= K) break;
            }
            if (MN == K) bls::fat_flip(acc);
#undef BLSGPU_UOP
            const uint32_t w1 = ch[0].x >> 16;               // absorb1 << 14 | absorb2 << 15
            // a combination split over 2 or 4 adjacent lanes: the flagged lanes add their
            // neighbours' partial limb accumulators (exact 64-bit integer adds)
            const uint32_t levels = (meta >> 16) & 3u;
            if (levels >= 1u) {
                const uint32_t m1 = (uint32_t)((int32_t)(w1 << 17) >> 31);
                asm volatile("s_nop 4");      // the hazard recogniser does not see the DPP reads inside the asm blocks
#pragma unroll
                for (int j = 0; j < 12; j++) dpp_quad_absorb<0xF5>(acc[j], m1);
            }
            if (levels >= 2u) {
                const uint32_t m2 = (uint32_t)((int32_t)(w1 << 16) >> 31);
                asm volatile("s_nop 4");
#pragma unroll
                for (int j = 0; j < 12; j++) dpp_quad_absorb<0xAA>(acc[j], m2);
            }
#ifdef BLSGPU_STAMPS
            asm volatile("" :: "v"(acc[0]), "v"(acc[11]));
            unsigned long long lt1 = __builtin_amdgcn_s_memtime();
#endif
            uint32_t D[12];
#if defined(BLSGPU_EXP) && (BLSGPU_EXP & 4)
            for (int j = 0; j < 12; j++) D[j] = (uint32_t)acc[j];  // timing experiment only
#else
            bls::fat_reduce(D, acc);
#endif
#ifdef BLSGPU_STAMPS
            asm volatile("" :: "v"(D[0]), "v"(D[11]));
            unsigned long long lt2 = __builtin_amdgcn_s_memtime();
            st_lin[0] += lt1 - lt0; st_lin[1] += lt2 - lt1; st_lin[2] += K;
#endif
            if (rd != 0xFFFFu) lds_store12(D, base16 + rd);
        } else if (LIGHT) {
            if (stash) {                                     // SAVE / RESTORE of the window at slot K
                uint32_t* win = reinterpret_cast<uint32_t*>(smem4) + base16 * 4u + ((meta >> 8) & 0xFFu) * 12u;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const uint32_t idx = lane + 64u * j;
                    if (idx < 144u) {
                        if (kind == 2u) stash[j] = win[idx];
                        else win[idx] = stash[j];
                    }
                }
            }
        } else if (kind == 2u) {                             // INV
            uint32_t ra = e_cur.x & 0xFFFFu, rd = e_cur.y & 0xFFFFu;
            if (rd != 0xFFFFu) {
                uint32_t A[12], D[12];
                lds_load12(A, base16 + ra);
                bls::fq_inv(D, A);
                lds_store12(D, base16 + rd);
            }
        } else {                                             // SGN
            uint32_t ra = e_cur.x & 0xFFFFu, rd = e_cur.y & 0xFFFFu;
            if (rd != 0xFFFFu) {
                uint32_t A[12], D[12];
                lds_load12(A, base16 + ra);
                bls::fq_sgn(D, A);
                lds_store12(D, base16 + rd);
            }
        }
        wave_fence();
#ifdef BLSGPU_STAMPS
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the round's LDS stores are issued
        st_acc[kind] += __builtin_amdgcn_s_memtime() - st_t0;
        st_cnt[kind] += 1;
#endif
    };
    if (LIGHT) {
        // two copies of the round with the record buffers swapped: no register copies
        // from the prefetch buffer to the current one (8 x 64 bit per round)
        uint2 nb[LIN_CHUNKS];
        while (true) {
            if (i == check_at && !chk()) return false;
            round(nx, nb);
            if (i >= n) break;
            if (i == check_at && !chk()) return false;
            round(nb, nx);
            if (i >= n) break;
        }
    } else {
        while (i < n) {
            if (i == check_at && !chk()) return false;
            uint2 ch[LIN_CHUNKS];
#pragma unroll
            for (int c = 0; c < LIN_CHUNKS; c++) ch[c] = nx[c];
            round(ch, nx);
        }
    }
#ifdef BLSGPU_STAMPS
    if (T.stamps && blockIdx.x == 0 && threadIdx.x == 0)
        for (int k2 = 0; k2 < 3; k2++) { atomicAdd(&T.stamps[k2], st_acc[k2]); atomicAdd(&T.stamps[3 + k2], st_cnt[k2]); atomicAdd(&T.stamps[6 + k2], st_lin[k2]); }
#endif
    return true;
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

// ---------------------------------------------------------------------------
// Degenerate pairs.  The fast Miller programs equal the reference only while (a) Q is a point
// of the twist (the projective formulas use the curve equation, the reference's affine ones do
// not), (b) no step of the loop degenerates -- then the reference's 0^-1 := 0 and the branches
// of fq2_add_line_eval / fq2_add_points decide (fields_t.py:47-55, 1062-1065, 673-686) -- and
// (c) Q carries no infinity flag (:676-677).  (b) is exactly "the final projective Z is not
// zero": Z' = 8 Y^3 Z in a tangent step, Z' = Z (X - xq Z)^3 in a chord step, zero is
// absorbing.  A team that sees any of the three puts its BLOCK on a work list, and
// k_miller_slow recomputes that block's partial with the reference-faithful program.
struct DegenList {
    uint32_t* count;          // entries used (zeroed before every Miller launch)
    uint32_t* blocks;         // block indices
    const uint8_t* inf;       // n x 2 flags (P, Q) as in fq_ate_pairing_multi's tuples, or nullptr
};

// wave-uniform: bit i set iff slot (slot0 + i) holds a value != 0 mod q   (i < n <= 64; relaxed values)
__device__ __forceinline__ uint64_t slots_nonzero(uint32_t base16, uint32_t slot0, uint32_t n, uint32_t lane) {
    bool nz = false;
    if (lane < n) {
        uint32_t X[12];
        lds_load12(X, base16 + (slot0 + lane) * 3);
        bls::fq_canon(X);
        nz = !bls::fq_is_zero(X);
    }
    return __ballot(nz);
}
__device__ __forceinline__ bool q_flagged(const DegenList& dg, size_t pair) {
    return dg.inf != nullptr && dg.inf[2 * pair + 1] != 0;
}
__device__ __forceinline__ void degen_push(const DegenList& dg, uint32_t lane) {
    if (lane == 0) {
        uint32_t at = atomicAdd(dg.count, 1u);
        dg.blocks[at] = blockIdx.x;
    }
}

// One pair (raw values already in the PX.. slots of the single-pair layout) through the fast
// single-pair program.  Returns false -- accumulator undefined -- when the pair is degenerate.
struct OnCurveCheck {                       // the n residual slots from slot0 are all zero
    uint32_t base16, slot0, n, lane;
    __device__ __forceinline__ bool operator()() const { return slots_nonzero(base16, slot0, n, lane) == 0; }
};
__device__ __forceinline__ bool miller_single(const VmTables& T, uint32_t base16, uint32_t lane, bool qflag) {
    if (qflag) return false;
    const OnCurveCheck chk{base16, BLSVM_SLOT_REG0 + 1, 2, lane};
    if (!run_rounds<true>(T, T.mflat, BLSVM_MILLER_FLAT_LEN, base16, lane, nullptr, BLSVM_MILLER_INIT_LEN, chk)) return false;
    wave_fence();
    return slots_nonzero(base16, BLSVM_SLOT_TX + 4, 2, lane) != 0;                                // final Z != 0
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
            run_rounds<true>(T, T.segflat + BLSVM_SEGF_MUL_0_1_OFF, BLSVM_SEGF_MUL_0_1_LEN, wave * (TEAM_BYTES / 16), lane);
        }
    }
}

// ---------------------------------------------------------------------------
// Kernel 1: one wavefront per (P, Q) pair: Miller loop, then the product of the
// workgroup's Miller values -> one Montgomery Fq12 partial (144 u32) per block.
//   g1: n x 96 bytes (x || y), g2: n x 192 bytes (x.c0 || x.c1 || y.c0 || y.c1),
//   big-endian canonical coordinates (the reference's serialisation, fields.py:87-88).
// Every input gives the reference's bytes: a team whose pair fails the fast program's
// validity tests (Q off the twist, final Z = 0, or a flagged Q -- zero coordinates, low-order
// and off-curve points all end there) puts its block on the DegenList and k_miller_slow
// recomputes that block's partial with the reference's own affine formulas.
#ifndef BLSGPU_MILLER_WPS
#define BLSGPU_MILLER_WPS 4
#endif
#ifndef BLSGPU_MP_WPS
#define BLSGPU_MP_WPS 3
#endif
// Groups: the pairs form `groups` consecutive runs of gsz pairs; block b works on
// group b / bpg and never mixes groups (bpg = blocks per group), so partials
// [g * bpg, (g + 1) * bpg) belong to group g.  A single multi-pairing is one group.
__global__ void __launch_bounds__(256, BLSGPU_MILLER_WPS) k_miller(VmTables T, const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                                uint32_t gsz, uint32_t bpg, uint32_t* __restrict__ partials, DegenList dg)
#if BLSGPU_EMIT(BLSGPU_TU_VM)
{
    uint32_t* smem = reinterpret_cast<uint32_t*>(smem4);
    __shared__ uint32_t any_degen;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t grp = blockIdx.x / bpg;
    const uint32_t in_grp = (blockIdx.x - grp * bpg) * nwaves + wave;
    const size_t pair = (size_t)grp * gsz + in_grp;
    uint32_t* team = smem + wave * TEAM_DW;
    const uint32_t base16 = wave * (TEAM_BYTES / 16);
    if (threadIdx.x == 0) any_degen = 0;
    team_init_consts(T, team, lane);
    bool ok = true;
    if (in_grp < gsz) {
        // coalesced load of the 72 big-endian dwords of the pair
        uint32_t w0, w1 = 0;
        {
            uint32_t d = lane;
            w0 = (d < 24) ? g1[pair * 24 + d] : g2[pair * 48 + (d - 24)];
            if (lane < 8) w1 = g2[pair * 48 + (lane + 40)];
        }
        {
            uint32_t d = lane, e = d / 12, w = d % 12;
            team[(BLSVM_SLOT_PX + e) * 12 + (11 - w)] = bswap32(w0);
            if (lane < 8) {
                d = lane + 64; e = d / 12; w = d % 12;
                team[(BLSVM_SLOT_PX + e) * 12 + (11 - w)] = bswap32(w1);
            }
        }
        wave_fence();
        ok = miller_single(T, base16, lane, q_flagged(dg, pair));
    } else {
        wave_fence();
    }
    if (!ok || in_grp >= gsz) team_set_acc(team, lane, true);
    __syncthreads();
    if (!ok && lane == 0) any_degen = 1;
    wave_fence();
    wg_product_tree(T, smem, wave, nwaves, lane);
    if (wave == 0) {
        wave_fence();
        // (a listed block's partial is rewritten by k_miller_slow before anything reads it)
        for (uint32_t i = lane; i < 144; i += 64) partials[(size_t)blockIdx.x * 144 + i] = team[F_DW + i];
        if (any_degen) degen_push(dg, lane);
    }
}
#else
;
#endif

// ---------------------------------------------------------------------------
// Kernel 1b (large batches): BLSVM_MP_G pairs per wavefront sharing ONE Miller
// accumulator  f <- f^2 * prod_g l_g  -- one squaring per iteration for the whole
// team and much better lane packing (vmgen/programs.py build_multi): ~31 % fewer
// instructions per pairing.  One team per block; a team with an incomplete or a
// special (zero-coordinate) pair falls back to the single-pair program for each
// of its pairs, so the observable behaviour is that of k_miller.
constexpr int MP_TEAM_DW = BLSVM_MP_TEAM_SLOTS * 12;
constexpr int MP_TEAM_BYTES = MP_TEAM_DW * 4;
// The multi-pair programs exist for G = BLSVM_MP_G (3: fewest instructions per pairing) and for G = 2: a
// call of a few thousand pairs cannot fill the chip with teams of three (8192 pairs = 2731 teams for
// 4096 places) and a team of two finishes in 3/4 of the time.
template <int G> struct MpCfg;
template <> struct MpCfg<BLSVM_MP_G> {
    static constexpr uint32_t F = BLSVM_MP_F, CORE = BLSVM_MP_CORE, Q = BLSVM_MP_Q, INIT_LEN = BLSVM_MP_INIT_LEN, FLAT_LEN = BLSVM_MP_FLAT_LEN;
    static __device__ __forceinline__ const uint2* flat(const VmTables& T) { return T.mpflat; }
};
template <> struct MpCfg<2> {
    static constexpr uint32_t F = BLSVM_MP2_F, CORE = BLSVM_MP2_CORE, Q = BLSVM_MP2_Q, INIT_LEN = BLSVM_MP2_INIT_LEN, FLAT_LEN = BLSVM_MP2_FLAT_LEN;
    static __device__ __forceinline__ const uint2* flat(const VmTables& T) { return T.mp2flat; }
};
static_assert(BLSVM_MP2_TEAM_SLOTS <= BLSVM_MP_TEAM_SLOTS, "one scratchpad size for both multi-pair programs");

// raw big-endian pair -> limb slots: P (2 values) from slot `p_slot`, Q (4 values) from `q_slot`
__device__ __forceinline__ void load_pair_raw(uint32_t* team, uint32_t p_slot, uint32_t q_slot, const uint32_t* __restrict__ g1,
                                              const uint32_t* __restrict__ g2, size_t pair, uint32_t lane) {
    for (uint32_t d = lane; d < 72; d += 64) {
        uint32_t w = (d < 24) ? g1[pair * 24 + d] : g2[pair * 48 + (d - 24)];
        uint32_t e = d / 12, k = d % 12;
        uint32_t slot = (e < 2) ? p_slot + e : q_slot + (e - 2);
        team[slot * 12 + (11 - k)] = bswap32(w);
    }
}

template <int G>
__global__ void __launch_bounds__(64, BLSGPU_MP_WPS) k_miller_mp(VmTables T, const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                                     uint32_t gsz, uint32_t bpg, uint32_t* __restrict__ partials, DegenList dg)
#if BLSGPU_EMIT(BLSGPU_TU_VM)
{
    using C = MpCfg<G>;
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t grp = blockIdx.x / bpg;                       // groups as in k_miller
    const uint32_t in_grp = (blockIdx.x - grp * bpg) * G;
    const size_t first = (size_t)grp * gsz + in_grp;
    const uint32_t cnt = min((uint32_t)G, gsz - in_grp);
    team_init_consts(T, team, lane);
    wave_fence();
    bool ok = true;
    uint32_t f_dw = C::F * 12u;
    if (cnt == (uint32_t)G) {
        // the multi-pair programs have their own scratchpad layout (vmgen/programs.MPLayout)
        for (uint32_t g = 0; g < cnt; ++g) {
            load_pair_raw(team, C::CORE + 14u * g, C::Q + 4u * g, g1, g2, first + g, lane);
            ok = ok && !q_flagged(dg, first + g);
        }
        wave_fence();
        uint32_t stash[3] = {0u, 0u, 0u};                        // Q of the team's pairs between the chord steps
        const OnCurveCheck chk{0u, C::F + 1, 2 * G, lane};       // every Q on the twist
        if (ok) ok = run_rounds<true>(T, C::flat(T), C::FLAT_LEN, 0, lane, stash, C::INIT_LEN, chk);
        if (ok) {
            wave_fence();
            // final Z of pair g in slots CORE + 14 g + 6, + 7: zero iff both are
            bool nz = false;
            if (lane < 2u * G) {
                uint32_t X[12];
                lds_load12(X, (C::CORE + 14u * (lane >> 1) + 6u + (lane & 1u)) * 3u);
                bls::fq_canon(X);
                nz = !bls::fq_is_zero(X);
            }
            const uint64_t m = __ballot(nz);
            for (uint32_t g = 0; g < (uint32_t)G; ++g) ok = ok && ((m >> (2u * g)) & 3ull) != 0;
        }
    } else {
        // ragged last team of a group: its pairs one by one through the single-pair program
        f_dw = (uint32_t)F_DW;
        bool have = false;
        for (uint32_t g = 0; g < cnt && ok; ++g) {
            load_pair_raw(team, BLSVM_SLOT_PX, BLSVM_SLOT_QX0, g1, g2, first + g, lane);
            wave_fence();
            ok = miller_single(T, 0, lane, q_flagged(dg, first + g));
            wave_fence();
            if (have) run_rounds<true>(T, T.segflat + BLSVM_SEGF_MUL_0_1_OFF, BLSVM_SEGF_MUL_0_1_LEN, 0, lane);
            run_rounds<true>(T, T.segflat + BLSVM_SEGF_COPY_1_0_OFF, BLSVM_SEGF_COPY_1_0_LEN, 0, lane);
            have = true;
        }
        if (!have) team_set_acc(team, lane, true);
    }
    wave_fence();
    for (uint32_t i = lane; i < 144; i += 64) partials[(size_t)blockIdx.x * 144 + i] = team[f_dw + i];
    if (!ok) degen_push(dg, lane);                               // k_miller_slow rewrites this partial
}
#else
;
#endif

// ---------------------------------------------------------------------------
// The reference-faithful Miller loop for one pair (vmgen/slow_programs.py): affine twist point,
// one field inversion per step with 0^-1 := 0, the reference's branches as 0/1 selections.
// Leaves fq_miller_loop(P, Q) itself (fields_t.py:1091-1111; Montgomery form) in register 0.
// (the scratchpad also runs the general segments mul_0_1 / copy_1_0 / from_mont_1_0, whose temporaries
// BLSVM_TEAM_SLOTS covers)
constexpr int SLOW_TEAM_BYTES = (BLSVM_SLOW_SLOTS > BLSVM_TEAM_SLOTS ? BLSVM_SLOW_SLOTS : BLSVM_TEAM_SLOTS) * 48;
__device__ __forceinline__ void miller_exact_one(const VmTables& T, uint32_t* team, uint32_t lane, const uint32_t* __restrict__ g1,
                                                 const uint32_t* __restrict__ g2, const uint8_t* __restrict__ inf, size_t pair) {
    load_pair_raw(team, BLSVM_SLOT_PX, BLSVM_SLOT_QX0, g1, g2, pair, lane);
    const bool qf = inf != nullptr && inf[2 * pair + 1] != 0;
    if (lane < 12) team[BLSVM_SLOT_QINF * 12 + lane] = qf ? team[BLSVM_SLOT_C_ONE * 12 + lane] : 0u;
    wave_fence();
    run_rounds(T, T.sflat, BLSVM_SLOW_FLAT_LEN, 0, lane);
    wave_fence();
}

// Rewrites the partials of the blocks a Miller kernel listed: block b held the pairs
// [grp * gsz + (b - grp * bpg) * per_block, ... + per_block) clipped to its group.  One
// wavefront per listed block; every wavefront leaves the loop at the same bound.
__global__ void __launch_bounds__(64) k_miller_slow(VmTables T, const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                                    uint32_t gsz, uint32_t bpg, uint32_t per_block, uint32_t* __restrict__ partials,
                                                    DegenList dg)
#if BLSGPU_EMIT(BLSGPU_TU_VM)
{
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total = __builtin_amdgcn_readfirstlane(*(volatile const uint32_t*)dg.count);
    if (blockIdx.x >= total) return;
    team_init_consts(T, team, lane);
    for (uint32_t e = blockIdx.x; e < total; e += gridDim.x) {
        const uint32_t b = __builtin_amdgcn_readfirstlane(dg.blocks[e]);
        const uint32_t grp = b / bpg;
        const uint32_t in_grp = (b - grp * bpg) * per_block;
        const uint32_t cnt = min(per_block, gsz - in_grp);
        const size_t first = (size_t)grp * gsz + in_grp;
        for (uint32_t g = 0; g < cnt; ++g) {
            wave_fence();
            miller_exact_one(T, team, lane, g1, g2, dg.inf, first + g);
            if (g) run_rounds<true>(T, T.segflat + BLSVM_SEGF_MUL_0_1_OFF, BLSVM_SEGF_MUL_0_1_LEN, 0, lane);
            run_rounds<true>(T, T.segflat + BLSVM_SEGF_COPY_1_0_OFF, BLSVM_SEGF_COPY_1_0_LEN, 0, lane);
        }
        wave_fence();
        for (uint32_t i = lane; i < 144; i += 64) partials[(size_t)b * 144 + i] = team[F_DW + i];
    }
}
#else
;
#endif

// register 0 (Montgomery, relaxed) of the team at the start of LDS -> 576 canonical big-endian bytes
__device__ __forceinline__ void write_acc_bytes(const VmTables& T, uint32_t* team, uint32_t lane, uint32_t* __restrict__ dst) {
    run_rounds<true>(T, T.segflat + BLSVM_SEGF_FROM_MONT_1_0_OFF, BLSVM_SEGF_FROM_MONT_1_0_LEN, 0, lane);
    if (lane < 12) {                             // relaxed (< 2q) -> canonical residues
        uint32_t X[12];
        lds_load12(X, (BLSVM_SLOT_REG0 + 12 + lane) * 3);
        bls::fq_canon(X);
        lds_store12(X, (BLSVM_SLOT_REG0 + 12 + lane) * 3);
    }
    wave_fence();
    for (uint32_t k = lane; k < 144; k += 64) {
        uint32_t c = k / 12, w = k % 12;
        dst[k] = bswap32(team[R1_DW + c * 12 + (11 - w)]);
    }
}

// fq_miller_loop for every pair (blsgpu_miller_loop_batch): out[p] = 576 canonical big-endian
// bytes of the reference's own Miller value -- not a multiple of it.
__global__ void __launch_bounds__(64) k_miller_exact(VmTables T, const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2,
                                                     const uint8_t* __restrict__ inf, uint32_t n, uint32_t* __restrict__ out_bytes)
#if BLSGPU_EMIT(BLSGPU_TU_VM)
{
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    team_init_consts(T, team, lane);
    for (uint32_t p = blockIdx.x; p < n; p += gridDim.x) {
        wave_fence();
        miller_exact_one(T, team, lane, g1, g2, inf, p);
        write_acc_bytes(T, team, lane, out_bytes + (size_t)p * 144);
    }
}
#else
;
#endif

// Fq12 operations on byte inputs, one element (pair) per wavefront: op 0 add, 1 sub, 2 mul, 3 neg, 4 invert
// (fq12_add / fq12_sub / fq12_mul / fq12_neg / fq12_invert, fields_t.py:321-352, 503-554, 328-337; 0^-1 = 0),
// 5 = power by the exponent bits ebits[0 .. nbits) (most significant first; fq12_pow, fields_t.py:340-353).
// Fq, Fq2 and Fq6 elements are Fq12 elements with the other coefficients zero.
#define BLSGPU_SEG(NAME) T.segflat + BLSVM_SEGF_##NAME##_OFF, BLSVM_SEGF_##NAME##_LEN
__global__ void __launch_bounds__(64) k_fq12_op(VmTables T, uint32_t op, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                const uint8_t* __restrict__ ebits, uint32_t nbits, uint32_t n,
                                                uint32_t* __restrict__ out_bytes)
#if BLSGPU_EMIT(BLSGPU_TU_VM)
{
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    team_init_consts(T, team, lane);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        wave_fence();
        for (uint32_t k = lane; k < 144; k += 64) team[R1_DW + (k / 12) * 12 + (11 - k % 12)] = bswap32(a[(size_t)i * 144 + k]);
        wave_fence();
        run_rounds<true>(T, BLSGPU_SEG(TO_MONT_0_1), 0, lane);                   // register 0 = a
        if (op <= 2u) {
            wave_fence();
            for (uint32_t k = lane; k < 144; k += 64) team[R1_DW + (k / 12) * 12 + (11 - k % 12)] = bswap32(b[(size_t)i * 144 + k]);
            wave_fence();
            run_rounds<true>(T, BLSGPU_SEG(TO_MONT_1_1), 0, lane);               // register 1 = b
            if (op == 0u) run_rounds<true>(T, BLSGPU_SEG(ADD_0_1), 0, lane);
            else if (op == 1u) run_rounds<true>(T, BLSGPU_SEG(SUB_0_1), 0, lane);
            else run_rounds<true>(T, BLSGPU_SEG(MUL_0_1), 0, lane);
        } else if (op == 3u) {
            run_rounds<true>(T, BLSGPU_SEG(NEG_0_0), 0, lane);
        } else if (op == 4u) {
            run_rounds(T, BLSGPU_SEG(INV12_2_0), 0, lane);
            run_rounds<true>(T, BLSGPU_SEG(COPY_0_2), 0, lane);
        } else {
            run_rounds<true>(T, BLSGPU_SEG(COPY_1_0), 0, lane);                  // register 1 = base
            run_rounds<true>(T, BLSGPU_SEG(SET_ONE_0), 0, lane);
            for (uint32_t k = 0; k < nbits; ++k) {                               // the reference's loop squares and multiplies too
                run_rounds<true>(T, BLSGPU_SEG(MUL_0_0), 0, lane);
                if (ebits[k]) run_rounds<true>(T, BLSGPU_SEG(MUL_0_1), 0, lane);
            }
        }
        wave_fence();
        write_acc_bytes(T, team, lane, out_bytes + (size_t)i * 144);
    }
}
#else
;
#endif
#undef BLSGPU_SEG

// fq2_double_line_eval(R, P) (fields_t.py:1035-1049; q == nullptr) / fq2_add_line_eval(R, Q, P)
// (:1052-1078) for n triples: r, q n x 192 bytes, p n x 96 bytes -> n x 576 bytes.
__global__ void __launch_bounds__(64) k_line_eval(VmTables T, const uint32_t* __restrict__ r, const uint32_t* __restrict__ q,
                                                  const uint32_t* __restrict__ p, uint32_t n, uint32_t* __restrict__ out_bytes)
#if BLSGPU_EMIT(BLSGPU_TU_VM)
{
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    team_init_consts(T, team, lane);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        wave_fence();
        if (lane < 24) team[(BLSVM_SLOT_PX + lane / 12) * 12 + (11 - lane % 12)] = bswap32(p[(size_t)i * 24 + lane]);
        if (lane < 48) {
            team[(BLSVM_SLOT_TX + lane / 12) * 12 + (11 - lane % 12)] = bswap32(r[(size_t)i * 48 + lane]);
            if (q) team[(BLSVM_SLOT_QX0 + lane / 12) * 12 + (11 - lane % 12)] = bswap32(q[(size_t)i * 48 + lane]);
        }
        wave_fence();
        if (q) run_rounds(T, T.segflat + BLSVM_SEGF_LINE_ADD_OFF, BLSVM_SEGF_LINE_ADD_LEN, 0, lane);
        else run_rounds(T, T.segflat + BLSVM_SEGF_LINE_DBL_OFF, BLSVM_SEGF_LINE_DBL_LEN, 0, lane);
        wave_fence();
        write_acc_bytes(T, team, lane, out_bytes + (size_t)i * 144);
    }
}
#else
;
#endif

// ---------------------------------------------------------------------------
// Kernel 2: product of Montgomery Fq12 partials.  Block b multiplies partials
// [b * per_block, min(m, (b+1) * per_block)) and writes one partial.  When
// do_final != 0 (single block) the product additionally goes through the final
// exponentiation and is written as 576 big-endian bytes (12 x 48, flat ZT
// order of fields.py:624-629) to out_bytes.
// blockIdx.y selects a group: its partial i is in[(i * istride + group * gstride) * 144];
// outputs are group-major (out_partials[group * gridDim.x + block], out_bytes[group]).
__global__ void __launch_bounds__(512) k_reduce(VmTables T, const uint32_t* __restrict__ in, uint32_t m, uint32_t per_block,
                                                 uint32_t istride, uint32_t gstride,
                                                 uint32_t* __restrict__ out_partials, uint32_t do_final,
                                                 uint32_t* __restrict__ out_bytes)
#if BLSGPU_EMIT(BLSGPU_TU_VM)
{
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
        const uint32_t* src = in + ((size_t)i * istride + (size_t)blockIdx.y * gstride) * 144;
        uint32_t dst = first ? F_DW : R1_DW;
        for (uint32_t k = lane; k < 144; k += 64) team[dst + k] = src[k];
        wave_fence();
        if (!first) run_rounds<true>(T, T.segflat + BLSVM_SEGF_MUL_0_1_OFF, BLSVM_SEGF_MUL_0_1_LEN, base16, lane);
        first = false;
    }
    if (first) team_set_acc(team, lane, true);
    wave_fence();
    wg_product_tree(T, smem, wave, nwaves, lane);
    if (wave == 0) {
        wave_fence();
        if (do_final) {
            run_rounds(T, T.fflat, BLSVM_FEXP_HEAD_LEN, base16, lane);                     // the segment with the inversion
            run_rounds<true>(T, T.fflat + BLSVM_FEXP_HEAD_LEN, BLSVM_FEXP_FLAT_LEN - BLSVM_FEXP_HEAD_LEN, base16, lane);
            run_rounds<true>(T, T.segflat + BLSVM_SEGF_FROM_MONT_1_0_OFF, BLSVM_SEGF_FROM_MONT_1_0_LEN, base16, lane);
            if (lane < 12) {                         // relaxed (< 2q) -> canonical residues
                uint32_t X[12];
                lds_load12(X, base16 + (BLSVM_SLOT_REG0 + 12 + lane) * 3);
                bls::fq_canon(X);
                lds_store12(X, base16 + (BLSVM_SLOT_REG0 + 12 + lane) * 3);
            }
            wave_fence();
            for (uint32_t k = lane; k < 144; k += 64) {
                uint32_t c = k / 12, w = k % 12;
                out_bytes[(size_t)blockIdx.y * 144 + k] = bswap32(team[R1_DW + c * 12 + (11 - w)]);
            }
        } else {
            uint32_t* dstp = out_partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 144;
            for (uint32_t k = lane; k < 144; k += 64) dstp[k] = team[F_DW + k];
        }
    }
}
#else
;
#endif

// ---------------------------------------------------------------------------
// Kernel 3: many independent results.  Team g multiplies partials
// [g * gsz, (g + 1) * gsz) (Montgomery Fq12, 144 u32 each), applies the final
// exponentiation and writes 576 canonical bytes to out_bytes[g].  No workgroup
// barrier: every team is on its own.
__global__ void __launch_bounds__(512) k_final_groups(VmTables T, const uint32_t* __restrict__ in, uint32_t gsz, uint32_t groups,
                                                       uint32_t* __restrict__ out_bytes)
#if BLSGPU_EMIT(BLSGPU_TU_VM)
{
    uint32_t* smem = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = blockDim.x >> 6;
    const uint32_t g = blockIdx.x * nwaves + wave;
    if (g >= groups) return;
    uint32_t* team = smem + wave * TEAM_DW;
    const uint32_t base16 = wave * (TEAM_BYTES / 16);
    team_init_consts(T, team, lane);
    wave_fence();
    if (gsz == 0) team_set_acc(team, lane, true);
    for (uint32_t i = 0; i < gsz; ++i) {
        const uint32_t* src = in + ((size_t)g * gsz + i) * 144;
        const uint32_t dst = (i == 0) ? F_DW : R1_DW;
        for (uint32_t k = lane; k < 144; k += 64) team[dst + k] = src[k];
        wave_fence();
        if (i) run_rounds<true>(T, T.segflat + BLSVM_SEGF_MUL_0_1_OFF, BLSVM_SEGF_MUL_0_1_LEN, base16, lane);
    }
    wave_fence();
    run_rounds(T, T.fflat, BLSVM_FEXP_HEAD_LEN, base16, lane);                             // the segment with the inversion
    run_rounds<true>(T, T.fflat + BLSVM_FEXP_HEAD_LEN, BLSVM_FEXP_FLAT_LEN - BLSVM_FEXP_HEAD_LEN, base16, lane);
    run_rounds<true>(T, T.segflat + BLSVM_SEGF_FROM_MONT_1_0_OFF, BLSVM_SEGF_FROM_MONT_1_0_LEN, base16, lane);
    if (lane < 12) {
        uint32_t X[12];
        lds_load12(X, base16 + (BLSVM_SLOT_REG0 + 12 + lane) * 3);
        bls::fq_canon(X);
        lds_store12(X, base16 + (BLSVM_SLOT_REG0 + 12 + lane) * 3);
    }
    wave_fence();
    for (uint32_t k = lane; k < 144; k += 64) {
        uint32_t c = k / 12, w = k % 12;
        out_bytes[(size_t)g * 144 + k] = bswap32(team[R1_DW + c * 12 + (11 - w)]);
    }
}
#else
;
#endif

// bytes (m x 576, canonical big-endian) -> Montgomery partials (m x 144 u32), one team each
__global__ void __launch_bounds__(512) k_bytes_to_partials(VmTables T, const uint32_t* __restrict__ in_bytes, uint32_t m,
                                                           uint32_t* __restrict__ out_partials)
#if BLSGPU_EMIT(BLSGPU_TU_VM)
{
    uint32_t* smem = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t g = blockIdx.x * (blockDim.x >> 6) + wave;
    if (g >= m) return;
    uint32_t* team = smem + wave * TEAM_DW;
    team_init_consts(T, team, lane);
    for (uint32_t k = lane; k < 144; k += 64) {
        uint32_t cidx = k / 12, w = k % 12;
        team[R1_DW + cidx * 12 + (11 - w)] = bswap32(in_bytes[(size_t)g * 144 + k]);
    }
    wave_fence();
    run_rounds<true>(T, T.segflat + BLSVM_SEGF_TO_MONT_0_1_OFF, BLSVM_SEGF_TO_MONT_0_1_LEN, wave * (TEAM_BYTES / 16), lane);
    for (uint32_t k = lane; k < 144; k += 64) out_partials[(size_t)g * 144 + k] = team[F_DW + k];
}
#else
;
#endif

// every instantiation the host side launches: this translation unit is the one that emits them (blsgpu_tu.h)
#if BLSGPU_TU == BLSGPU_TU_VM
__attribute__((used)) static const void* const blsgpu_instances_vm[] = {
    (const void*)&k_miller_mp<2>,
    (const void*)&k_miller_mp<BLSVM_MP_G>};
#endif
}  // namespace blsgpu
