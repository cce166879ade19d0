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
    // shared constants [0, HC_SLOT0), then the extra ones (kept behind the final exponentiation's in the table)
    for (uint32_t i = lane; i < BLSVM_HC_SLOT0 * 12; i += 64) team[i] = T.consts[i];
    for (uint32_t i = lane; i < (BLSVM_NCONST_H2C - BLSVM_HC_SLOT0) * 12; i += 64)
        team[BLSVM_HC_SLOT0 * 12 + i] = T.consts[BLSVM_HC_TBL0 * 12 + i];
}

// SHA-256 (FIPS 180-4) of one 40-byte message = one block, for the hash512 chain of
// ec.py:531-534 / util.py:7-16:  hash512(m) = sha256(m || 00) || sha256(m || 01) with
// m = message hash (32 bytes) || "G2_j_ck" (7 bytes).
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
__device__ void sha256_block(const uint32_t w_in[16], uint32_t out[8]) {
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = w_in[i];
#pragma unroll
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = 0x6a09e667, b = 0xbb67ae85, c = 0x3c6ef372, d = 0xa54ff53a, e = 0x510e527f, f = 0x9b05688c, g = 0x1f83d9ab,
             hh = 0x5be0cd19;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K[i] + w[i];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    out[0] = a + 0x6a09e667; out[1] = b + 0xbb67ae85; out[2] = c + 0x3c6ef372; out[3] = d + 0xa54ff53a;
    out[4] = e + 0x510e527f; out[5] = f + 0x9b05688c; out[6] = g + 0x1f83d9ab; out[7] = hh + 0x5be0cd19;
}

// One thread per SHA-256: thread t of message i computes half (t & 1) of hash512 number
// (t >> 1) in {G2_0_c0, G2_0_c1, G2_1_c0, G2_1_c1}.  msg: n x 32 bytes; digests:
// n x 4 x 64 bytes, the big-endian 512-bit values of ec.py:531-534 before `% q`.
__global__ void __launch_bounds__(256) k_h2c_hash(const uint32_t* __restrict__ msg, uint32_t n, uint32_t* __restrict__ digests)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gid >> 3, t = gid & 7u;
    if (i >= n) return;
    const uint32_t j = t >> 2, c = (t >> 1) & 1u, half = t & 1u;
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = bswap32(msg[(size_t)i * 8 + k]);
    // "G2_" j "_c" c  then the hash512 selector byte, then the 0x80 padding byte
    w[8] = 0x47325F00u | (0x30u + j);                       // 'G' '2' '_' ('0' + j)
    w[9] = 0x5F630000u | ((0x30u + c) << 8) | half;         // '_' 'c' ('0' + c) selector
    w[10] = 0x80000000u;
    w[11] = 0; w[12] = 0; w[13] = 0; w[14] = 0;
    w[15] = 320;                                            // message length in bits
    uint32_t d[8];
    sha256_block(w, d);
#pragma unroll
    for (int k = 0; k < 8; k++) digests[((size_t)i * 4 + (t >> 1)) * 16 + half * 8 + k] = bswap32(d[k]);
}
#else
;
#endif

// ---------------------------------------------------------------------------
// Fixed-exponent powers x^((q-3)/4) -- the square-root / Legendre-symbol exponent of
// vmgen/h2c_programs.py and decomp_programs.py -- in registers, one value per lane: 379
// squarings + ~110 products (sliding window, odd powers in registers), no LDS.  The VM programs around it are cut into
// stages that hand their scratchpad state over through an image in HBM:
//   stage kernel (few rounds) -> k_pow on the image's BASE slots -> next stage kernel ...
// image: per team the slots [STATE0, STATE1) of its scratchpad, 12 u32 each.
// k_pow: value v = (team v / cnt, index v % cnt): ACC[index] <- BASE[index]^E (Montgomery, < 2q).
__global__ void __launch_bounds__(256) k_pow(uint32_t* __restrict__ img, uint32_t img_slots, uint32_t base_off, uint32_t acc_off,
                                             uint32_t cnt, uint32_t total)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= total) return;
    const uint32_t team = v / cnt, k = v % cnt;
    const uint32_t* src = img + ((size_t)team * img_slots + base_off + k) * 12;
    uint32_t* dst = img + ((size_t)team * img_slots + acc_off + k) * 12;
    // sliding window over the fixed exponent (vmgen/emit.pow_windows): odd powers b, b^3, ... in registers, on the
    // carry-free 28-bit limbs of fp28.h (a squaring: 105 + 196 multiply-accumulates, nothing else)
    constexpr int NT = 1 << (BLSVM_POW_WINDOW - 1);
    r28::fe tbl[NT], a;
    {
        uint32_t x[12];
#pragma unroll
        for (int j = 0; j < 12; j++) x[j] = src[j];
        tbl[0] = r28::from_vm(x);
    }
    const r28::fe t = r28::sqr(tbl[0]);
#pragma unroll
    for (int i = 1; i < NT; i++) tbl[i] = r28::mul(tbl[i - 1], t);
    auto pick = [&](uint32_t k) {
        r28::fe m;
#pragma unroll
        for (int j = 0; j < r28::NL; j++) {
            m.v[j] = tbl[0].v[j];
#pragma unroll
            for (int i = 1; i < NT; i++) m.v[j] = (k == (uint32_t)i) ? tbl[i].v[j] : m.v[j];
        }
        return m;
    };
    a = pick(BLSVM_POW_WIN[0][1]);
#pragma unroll 1
    for (int s = 1; s < BLSVM_POW_STEPS; s++) {
        const uint32_t nsq = BLSVM_POW_WIN[s][0], k = BLSVM_POW_WIN[s][1];
#pragma unroll 1
        for (uint32_t i = 0; i < nsq; i++) a = r28::sqr(a);
        if (k != 255u) a = r28::mul(a, pick(k));
    }
    {
        uint32_t y[12];
        r28::to_vm(y, a);
#pragma unroll
        for (int j = 0; j < 12; j++) dst[j] = y[j];
    }
}
#else
;
#endif

// The same powers for a batch that leaves SIMDs empty (round 5): TWO wavefronts per 64 values.  k_pow is one lane's chain of 379
// squarings + ~110 products (0.47 ms whether for 32 768 values or for one: hashing ONE message pays it twice).  Right to left in
// base 4 the squarings are a chain of their own -- p_j = b^(4^j), two squarings per digit -- and the products  B_d <- B_d p_j  (d = the
// exponent's digit j) only READ it: wave 0 squares and posts p_j in LDS, wave 1 multiplies the posted powers into three buckets one
// digit behind (the digits are the same for every lane: a uniform branch picks the bucket) and ends with b^e = B_1 B_2^2 B_3^3 in
// four products.  The chain is 378 squarings (105 + 196 multiply-adds each) and nothing else: 0.32 ms.
// (256-thread workgroups, two sets of 64 values each: four wavefronts, one per SIMD of a CU -- 128-thread workgroups get stacked on
// two SIMDs of a CU, profiles/r04_workgroup_shape.txt)
__global__ void __launch_bounds__(256) k_pow2(uint32_t* __restrict__ img, uint32_t img_slots, uint32_t base_off, uint32_t acc_off,
                                              uint32_t cnt, uint32_t total)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    __shared__ int32_t posts[2][2][r28::NL][64];                 // per set: p_j of its 64 values, two buffers
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t set = wv >> 1, wave = wv & 1u;
    int32_t (*post)[r28::NL][64] = posts[set];
    const uint32_t vr = (blockIdx.x * 2u + set) * 64u + lane;
    const uint32_t v = vr < total ? vr : total - 1u;               // (spare lanes repeat the last value and write nothing)
    const uint32_t team = v / cnt, k = v % cnt;
    const uint32_t* src = img + ((size_t)team * img_slots + base_off + k) * 12;
    uint32_t* dst = img + ((size_t)team * img_slots + acc_off + k) * 12;
    constexpr uint32_t DIGITS = (BLSVM_POW_E_BITS + 1) / 2;
    auto digit = [&](uint32_t j) { return (BLSVM_POW_E[(2u * j) >> 5] >> ((2u * j) & 31u)) & 3u; };      // (bit 2 j never straddles a word)
    r28::fe p;
    if (wave == 0u) {
        uint32_t x[12];
#pragma unroll
        for (int j = 0; j < 12; j++) x[j] = src[j];
        p = r28::from_vm(x);
    }
    r28::fe B1 = r28::fe_one(), B2 = r28::fe_one(), B3 = r28::fe_one();
#pragma unroll 1
    for (uint32_t j = 0; j < DIGITS; j++) {
        if (wave == 0u) {
#pragma unroll
            for (int l = 0; l < r28::NL; l++) post[j & 1u][l][lane] = p.v[l];
        }
        __syncthreads();
        if (wave == 0u) {
            if (j + 1u < DIGITS) p = r28::sqr(r28::sqr(p));
        } else {
            const uint32_t d = digit(j);
            if (d != 0u) {
                r28::fe q;
#pragma unroll
                for (int l = 0; l < r28::NL; l++) q.v[l] = post[j & 1u][l][lane];
                if (d == 1u) B1 = r28::mul(B1, q);
                else if (d == 2u) B2 = r28::mul(B2, q);
                else B3 = r28::mul(B3, q);
            }
        }
    }
    if (wave == 1u) {
        r28::fe run = r28::mul(B3, B2);
        r28::fe acc = r28::mul(B3, run);
        run = r28::mul(run, B1);
        acc = r28::mul(acc, run);
        if (vr < total) {
            uint32_t y[12];
            r28::to_vm(y, acc);
#pragma unroll
            for (int j = 0; j < 12; j++) dst[j] = y[j];
        }
    }
}
#else
;
#endif

__device__ __forceinline__ void img_load(uint32_t* team, const uint32_t* __restrict__ img, uint32_t state0, uint32_t nslots, uint32_t lane) {
    for (uint32_t d = lane; d < nslots * 12; d += 64) team[state0 * 12 + d] = img[d];
}
__device__ __forceinline__ void img_store(const uint32_t* team, uint32_t* __restrict__ img, uint32_t state0, uint32_t nslots, uint32_t lane) {
    for (uint32_t d = lane; d < nslots * 12; d += 64) img[d] = team[state0 * 12 + d];
}

constexpr uint32_t H1_IMG = BLSVM_H1_STATE1 - BLSVM_H1_STATE0;       // image slots per team

// Hash-to-G2 stage STAGE of the encodings (one team = BLSVM_H1_NE encodings):
//   0: inputs -> h1_a / h1w_a (candidates, norms)         -> image   [then k_pow on 3 NE values]
//   1: image -> h1_b (pick the candidate, delta+-)         -> image   [then k_pow on 2 NE values]
//   2: image -> h1_c (root, sign, result S)                -> image   (k_h2c_clear reads S from it)
// WIDE = 0: t is n_enc x 96 bytes (c0 || c1, big-endian, < 2^384); WIDE = 1: n_enc x 128
// bytes (two 512-bit big-endian hash values, reduced mod q in h1w_a).
template <int STAGE, int WIDE>
__global__ void __launch_bounds__(64, 2) k_h2c_stage(VmTables T, const uint32_t* __restrict__ t, uint32_t n_enc, uint32_t* __restrict__ img)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t first = blockIdx.x * BLSVM_H1_NE;
    uint32_t* my = img + (size_t)blockIdx.x * H1_IMG * 12;
    team_init_consts_h2c(T, team, lane);
    if (STAGE == 0) {
        if (WIDE) {
            for (uint32_t d = lane; d < BLSVM_H1_NE * 24; d += 64) team[BLSVM_H1_TH * 12 + d] = 0u;
            wave_fence();
            for (uint32_t d = lane; d < BLSVM_H1_NE * 32; d += 64) {
                const uint32_t e = d / 32, o = d % 32, c = o / 16, w = o % 16;     // dword w of the 64-byte value, MSB first
                // encodings past the end run on t = (1, 0); their results are dropped
                uint32_t v = (first + e < n_enc) ? bswap32(t[(size_t)(first + e) * 32 + o]) : ((c == 0 && w == 15) ? 1u : 0u);
                if (w < 4) team[(BLSVM_H1_TH + 2 * e + c) * 12 + (3 - w)] = v;     // bits 384..511
                else team[(BLSVM_H1_T + 2 * e + c) * 12 + (15 - w)] = v;           // bits 0..383
            }
        } else {
            for (uint32_t d = lane; d < BLSVM_H1_NE * 24; d += 64) {
                uint32_t e = d / 24, o = d % 24, c = o / 12, w = o % 12;
                uint32_t v = (first + e < n_enc) ? bswap32(t[(size_t)(first + e) * 24 + o]) : ((c == 0 && w == 11) ? 1u : 0u);
                team[(BLSVM_H1_T + 2 * e + c) * 12 + (11 - w)] = v;
            }
        }
    } else {
        img_load(team, my, BLSVM_H1_STATE0, H1_IMG, lane);
    }
    wave_fence();
    if (STAGE == 0) {
        if (WIDE) run_rounds(T, T.segflat + BLSVM_SEGF_H1W_A_OFF, BLSVM_SEGF_H1W_A_LEN, 0, lane);
        else run_rounds(T, T.segflat + BLSVM_SEGF_H1_A_OFF, BLSVM_SEGF_H1_A_LEN, 0, lane);
    } else if (STAGE == 1) {
        run_rounds<true>(T, T.segflat + BLSVM_SEGF_H1_B_OFF, BLSVM_SEGF_H1_B_LEN, 0, lane);
    } else {
        run_rounds(T, T.segflat + BLSVM_SEGF_H1_C_OFF, BLSVM_SEGF_H1_C_LEN, 0, lane);
    }
    wave_fence();
    if (STAGE == 0) {
        // A candidate whose u = x^3 + b' has zero imaginary part has no y in the reference (Fq2.modsqrt's a1 == 0
        // branch, fields.py:466-467; sw_encode's except, ec.py:489-498): n' = 0 keeps h1_b from choosing it
        // (vmgen.h2c_programs.real_u_step; tests/golden/g2_real_u.json).
        bool real_u = false;
        if (lane < 3 * BLSVM_H1_NE) {
            uint32_t A1[12];
            lds_load12(A1, (BLSVM_H1_U + 2 * lane + 1) * 3);
            bls::fq_canon(A1);
            real_u = bls::fq_is_zero(A1);
        }
        if (real_u)
            for (uint32_t w = 0; w < 12; ++w) team[(BLSVM_H1_N + lane) * 12 + w] = 0u;
        wave_fence();
    }
    img_store(team, my, BLSVM_H1_STATE0, H1_IMG, lane);
}
#else
;
#endif

// ---------------------------------------------------------------------------
// The three stages once more with ONE ENCODING PER LANE on the register arithmetic (round 3).  On the VM a team of 64
// lanes serves 12 encodings and an inversion round serves 12 of its 64 lanes; at 262 144 messages the three stages
// cost 10 ms.  Here every lane carries an encoding through the same formulas (vmgen/h2c_programs.py h1_a / h1_b /
// h1_c, i.e. sw_encode of ec.py:449-507 with its control flow as arithmetic selects: the selectors stay FIELD values
// (chi^2 + chi)/2, (1 + chi)/2 exactly as there, so every input gives the VM's values), reads and writes the SAME image
// in the VM's form -- k_pow between the stages and the clearing kernels after them are unchanged -- and the one
// inversion per stage is the safegcd routine run by all 64 lanes at once.
namespace swl {
using r28::fe;
using r28::fe2;
constexpr uint32_t NE = BLSVM_H1_NE;
constexpr uint32_t S_PAR = BLSVM_H1_TH + 2 * NE, S_X = S_PAR + NE, S_A1 = BLSVM_H1_BASE + 3 * NE, S_FT = S_A1 + NE;
__device__ __forceinline__ fe ldv(const uint32_t* __restrict__ team, uint32_t slot) {
    uint32_t x[12];
#pragma unroll
    for (int j = 0; j < 12; j++) x[j] = team[(slot - BLSVM_H1_STATE0) * 12 + j];
    return r28::from_vm(x);
}
template <int A, int B> __device__ __forceinline__ void stv(uint32_t* __restrict__ team, uint32_t slot, const r28::F<A, B>& v) {
    uint32_t y[12];
    r28::to_vm(y, r28::norm(v));
#pragma unroll
    for (int j = 0; j < 12; j++) team[(slot - BLSVM_H1_STATE0) * 12 + j] = y[j];
}
__device__ __forceinline__ void st_zero(uint32_t* __restrict__ team, uint32_t slot) {
#pragma unroll
    for (int j = 0; j < 12; j++) team[(slot - BLSVM_H1_STATE0) * 12 + j] = 0u;
}
__device__ __forceinline__ fe cst(const int32_t (&c)[r28::NL]) { return r28::fe_const(c); }
// canonical(v) > (q - 1) / 2: the "lexicographically larger than its negation" test of ec.py:94-100 (the VM's SGN round)
template <int A, int B> __device__ __forceinline__ bool sgn(const r28::F<A, B>& v) {
    const uint32_t half[12] = BLS_HALF_LIMBS;
    uint32_t x[12];
    r28::to_raw(x, r28::norm(v));
    uint32_t br = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) (void)bls::subc(half[j], x[j], br);
    return br != 0;
}
__device__ __forceinline__ fe inv(const fe& n) {                     // 1/n, 0 -> 0
    uint32_t w[12], v[12];
    r28::to_vm(w, n);
    bls::fq_inv(v, w);
    return r28::from_vm(v);
}
template <int A, int B> __device__ __forceinline__ fe red(const r28::F<A, B>& x) { return r28::mul(x, r28::fe_one()); }   // value into (-q, 2q)
__device__ __forceinline__ fe2 scale(const fe2& x, const fe& k) { return {r28::mul(x.a, k), r28::mul(x.b, k)}; }
}  // namespace swl

// ---- quadratic characters without a power -------------------------------------------------------------------------
// Of the five fixed-exponent powers per encoding (377 squarings + 110 products each) three only decide a quadratic
// character: which of x1, x2 has a square norm (the third candidate is taken unseen, ec.py:489-500), and which of
// delta+- is the square (fields.py:463-482).  swl::jacobi decides it with the binary algorithm on the canonical value:
// 768 fixed iterations of compare / conditional swap (reciprocity: the sign flips when both are 3 mod 4) / subtract /
// halve (the sign flips when n is 3 or 5 mod 8), ~100 instructions each and the same for every lane -- 0.4 of a power.
// The k_h2c_swj* kernels are the encoding stages built on it: TWO powers per encoding instead of five.  Results are the
// VM's and the k_h2c_sw* kernels' field elements in every case (chi in {1, -1, 0} drives the same selections).
namespace swl {
__device__ __forceinline__ int jacobi(uint32_t (&a)[12]) {
    uint32_t n[12] = BLS_Q_LIMBS;
    uint32_t sg = 0;
#pragma unroll 1
    for (int it = 0; it < 768; it++) {
        const bool odd = (a[0] & 1u) != 0u;
        uint32_t d[12], e[12], br = 0, br2 = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) d[j] = bls::subc(a[j], n[j], br);            // a - n, borrow <=> a < n
#pragma unroll
        for (int j = 0; j < 12; j++) e[j] = bls::subc(n[j], a[j], br2);           // n - a
        const bool sw = odd && br != 0u;                                           // a odd and a < n: the two change places
        sg ^= sw ? ((a[0] & n[0] & 2u) >> 1) : 0u;                                 // reciprocity: both 3 mod 4
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const uint32_t keep = odd ? d[j] : a[j];
            n[j] = sw ? a[j] : n[j];
            a[j] = sw ? e[j] : keep;
        }
        const uint32_t any = (a[0] | a[1] | a[2]) | (a[3] | a[4] | a[5]) | (a[6] | a[7] | a[8]) | (a[9] | a[10] | a[11]);
        sg ^= any ? (((n[0] >> 1) ^ (n[0] >> 2)) & 1u) : 0u;                       // a halving: n is 3 or 5 mod 8
#pragma unroll
        for (int j = 0; j < 11; j++) a[j] = __builtin_amdgcn_alignbit(a[j + 1], a[j], 1);     // (zero stays zero)
        a[11] >>= 1;
    }
    uint32_t rest = n[0] ^ 1u;
#pragma unroll
    for (int j = 1; j < 12; j++) rest |= n[j];
    return rest ? 0 : (sg ? -1 : 1);
}
// the quadratic character of v.  Round 4: from division steps (fq32.h fq_jacobi_var, a third of the binary routine's
// instructions even with the lanes of a wavefront waiting for the slowest); the binary routine above remains the
// answer for a value the batches did not finish (never observed; tests/test_jacobi_model.py) and, with
// BLSGPU_H2C_BINARY_JACOBI builds, the A/B reference.
template <int A, int B> __device__ __forceinline__ int chi(const r28::F<A, B>& v) {
    uint32_t x[12];
    r28::to_raw(x, r28::norm(v));
#ifdef BLSGPU_H2C_BINARY_JACOBI
    return jacobi(x);
#else
    int r = bls::fq_jacobi_var(x);
    if (r == 2) r = jacobi(x);
    return r;
#endif
}
}  // namespace swl

// stage 0 with the choice of the candidate inside: X[6i..] = x, U[6i] = u0, A1 = u1, N[3i] = n' (0 for a real u),
// BASE[i] = ACC[i] = N(u) of the chosen candidate -> ONE power per encoding
template <int WIDE>
__global__ void __launch_bounds__(256, 2) k_h2c_swj0(const uint32_t* __restrict__ t, uint32_t n_enc, uint32_t total, uint32_t* __restrict__ img)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    using namespace swl;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    uint32_t* team = img + (size_t)(e / NE) * H1_IMG * 12;
    const uint32_t i = e % NE;
    const int32_t onec[r28::NL] = BLS28_ONE, s3c[r28::NL] = BLS28_SW_S3, hhc[r28::NL] = BLS28_SW_HH, sic[r28::NL] = BLS28_SW_SINV;
    const fe one = cst(onec);
    fe2 tv;
    if (e < n_enc) {
        fe part[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            uint32_t lo[12];
            if (WIDE) {
                const uint32_t* src = t + (size_t)e * 32 + c * 16;
                uint32_t hi[12];
#pragma unroll
                for (int j = 0; j < 12; j++) { lo[j] = bswap32(src[15 - j]); hi[j] = j < 4 ? bswap32(src[3 - j]) : 0u; }
                const int32_t c2[r28::NL] = BLS28_WIDE_C2;
                part[c] = r28::norm(r28::add(r28::from_raw(lo), r28::mul(r28::unpack32(hi), cst(c2))));
            } else {
                const uint32_t* src = t + (size_t)e * 24 + c * 12;
#pragma unroll
                for (int j = 0; j < 12; j++) lo[j] = bswap32(src[11 - j]);
                part[c] = r28::from_raw(lo);
            }
        }
        tv = {part[0], part[1]};
    } else {
        tv = {one, r28::fe_zero()};
    }
    if (sgn(tv.b)) stv(team, S_PAR + i, one); else st_zero(team, S_PAR + i);
    const fe2 tt = r28::sqr(tv);
    const fe2 w = {r28::norm(r28::add(tt.a, r28::mulc<5>(one))), r28::norm(r28::add(tt.b, r28::mulc<4>(one)))};
    const fe2 wt = r28::mul(w, tv);
    const fe ni = inv(r28::dot2(wt.a, wt.a, wt.b, wt.b));
    const fe2 iwt = {r28::mul(wt.a, ni), r28::mul(r28::neg(wt.b), ni)};
    const fe2 wi = r28::mul(tv, iwt), ti = r28::mul(w, iwt);
    stv(team, S_FT + i, r28::mul(tv, ti).a);
    const fe2 wp = scale(r28::mul(wi, tv), cst(s3c));
    const fe2 wpt = r28::mul(wp, tv);
    const fe2 x1 = {r28::norm(r28::sub(cst(hhc), wpt.a)), r28::norm(r28::neg(wpt.b))};
    const fe2 x2 = {r28::norm(r28::sub(r28::neg(one), x1.a)), r28::norm(r28::neg(x1.b))};
    const fe2 wpi = scale(r28::mul(w, ti), cst(sic));
    const fe2 q3 = r28::sqr(wpi);
    const fe2 x3 = {r28::norm(r28::add(q3.a, one)), q3.b};
    const fe four = r28::norm(r28::mulc<4>(one));
    // candidates in the reference's order; the first of x1, x2 whose n' is a square, else x3 (ec.py:489-500)
    fe2 x = x3, u;
    fe n;
    bool realu, taken = false;
#pragma unroll 1
    for (int j = 0; j < 3; j++) {
        const fe2 xc = j == 0 ? x1 : (j == 1 ? x2 : x3);
        const fe2 x3p = r28::mul(r28::sqr(xc), xc);
        const fe2 uc = {red(r28::add(x3p.a, four)), red(r28::add(x3p.b, four))};
        const fe nc = r28::dot2(uc.a, uc.a, uc.b, uc.b);
        const bool ru = r28::is_zero(uc.b);                   // u with zero imaginary part: n' = 0, never a square here
        const bool sq = j == 2 || (!ru && chi(nc) == 1);
        const bool take = sq && !taken;
#pragma unroll
        for (int l = 0; l < r28::NL; l++) {
            x.a.v[l] = take ? xc.a.v[l] : x.a.v[l]; x.b.v[l] = take ? xc.b.v[l] : x.b.v[l];
            u.a.v[l] = take ? uc.a.v[l] : u.a.v[l]; u.b.v[l] = take ? uc.b.v[l] : u.b.v[l];
            n.v[l] = take ? nc.v[l] : n.v[l];
        }
        realu = take ? ru : realu;
        taken = taken || take;
    }
    stv(team, S_X + 6 * i, x.a); stv(team, S_X + 6 * i + 1, x.b);
    stv(team, BLSVM_H1_U + 6 * i, u.a);
    stv(team, S_A1 + i, u.b);
    if (realu) st_zero(team, BLSVM_H1_N + 3 * i); else stv(team, BLSVM_H1_N + 3 * i, n);
    stv(team, BLSVM_H1_ACC + i, n); stv(team, BLSVM_H1_BASE + i, n);
}
#else
;
#endif
// after z = n^E: r = z n', delta+- = (u0 +- r)/2, the square one of them -> BASE[i] = ACC[i]; its character -> N[3i + 1]
__global__ void __launch_bounds__(256) k_h2c_swj1(uint32_t total, uint32_t* __restrict__ img)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    using namespace swl;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    uint32_t* team = img + (size_t)(e / NE) * H1_IMG * 12;
    const uint32_t i = e % NE;
    const int32_t halfc[r28::NL] = BLS28_HALF;
    const fe inv2 = cst(halfc);
    const fe r = r28::mul(ldv(team, BLSVM_H1_ACC + i), ldv(team, BLSVM_H1_N + 3 * i));
    const fe u0 = ldv(team, BLSVM_H1_U + 6 * i);
    const fe dp = r28::mul(r28::add(u0, r), inv2), dm = r28::mul(r28::sub(u0, r), inv2);
    const int J = chi(dp);
    fe d;
#pragma unroll
    for (int l = 0; l < r28::NL; l++) d.v[l] = J == 1 ? dp.v[l] : dm.v[l];
    stv(team, BLSVM_H1_ACC + i, d); stv(team, BLSVM_H1_BASE + i, d);
    team[(BLSVM_H1_N + 3 * i + 1 - BLSVM_H1_STATE0) * 12] = (uint32_t)(J + 1);
}
#else
;
#endif
// after z = d^E: x0 = z d (halved when delta+ was zero: the VM's (1 + chi)/2 selection with chi = 0), then h1_c as k_h2c_sw2
__global__ void __launch_bounds__(256) k_h2c_swj2(uint32_t total, uint32_t* __restrict__ img)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    using namespace swl;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    uint32_t* team = img + (size_t)(e / NE) * H1_IMG * 12;
    const uint32_t i = e % NE;
    const int32_t halfc[r28::NL] = BLS28_HALF, onec[r28::NL] = BLS28_ONE;
    const fe inv2 = cst(halfc), one = cst(onec);
    const fe sq = r28::mul(ldv(team, BLSVM_H1_ACC + i), ldv(team, BLSVM_H1_BASE + i));
    const bool zeroplus = team[(BLSVM_H1_N + 3 * i + 1 - BLSVM_H1_STATE0) * 12] == 1u;
    const fe hsq = r28::mul(sq, inv2);
    fe x0;
#pragma unroll
    for (int l = 0; l < r28::NL; l++) x0.v[l] = zeroplus ? hsq.v[l] : sq.v[l];
    const fe a1 = ldv(team, S_A1 + i);
    const fe x1c = r28::mul(a1, inv(red(r28::add(x0, x0))));
    const bool g = sgn(x1c);
    uint32_t pw = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) pw |= team[(S_PAR + i - BLSVM_H1_STATE0) * 12 + j];
    const bool flip = g != (pw != 0u);
    const fe y0 = flip ? r28::norm(r28::neg(x0)) : x0, y1 = flip ? r28::norm(r28::neg(x1c)) : x1c;
    const fe ft = ldv(team, S_FT + i);
    stv(team, BLSVM_H1_S + 5 * i, r28::mul(ft, ldv(team, S_X + 6 * i)));
    stv(team, BLSVM_H1_S + 5 * i + 1, r28::mul(ft, ldv(team, S_X + 6 * i + 1)));
    stv(team, BLSVM_H1_S + 5 * i + 2, r28::add(one, r28::mul(ft, red(r28::sub(y0, one)))));
    stv(team, BLSVM_H1_S + 5 * i + 3, r28::mul(ft, y1));
    stv(team, BLSVM_H1_S + 5 * i + 4, ft);
}
#else
;
#endif

template <int WIDE>
__global__ void __launch_bounds__(64) k_h2c_sw0(const uint32_t* __restrict__ t, uint32_t n_enc, uint32_t total, uint32_t* __restrict__ img)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    using namespace swl;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    uint32_t* team = img + (size_t)(e / NE) * H1_IMG * 12;
    const uint32_t i = e % NE;
    const int32_t onec[r28::NL] = BLS28_ONE, s3c[r28::NL] = BLS28_SW_S3, hhc[r28::NL] = BLS28_SW_HH, sic[r28::NL] = BLS28_SW_SINV;
    const fe one = cst(onec);
    fe2 tv;
    if (e < n_enc) {
        fe part[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            uint32_t lo[12];
            if (WIDE) {
                const uint32_t* src = t + (size_t)e * 32 + c * 16;             // 64 bytes, most significant dword first
                uint32_t hi[12];
#pragma unroll
                for (int j = 0; j < 12; j++) { lo[j] = bswap32(src[15 - j]); hi[j] = j < 4 ? bswap32(src[3 - j]) : 0u; }
                const int32_t c2[r28::NL] = BLS28_WIDE_C2;
                part[c] = r28::norm(r28::add(r28::from_raw(lo), r28::mul(r28::unpack32(hi), cst(c2))));
            } else {
                const uint32_t* src = t + (size_t)e * 24 + c * 12;
#pragma unroll
                for (int j = 0; j < 12; j++) lo[j] = bswap32(src[11 - j]);
                part[c] = r28::from_raw(lo);
            }
        }
        tv = {part[0], part[1]};
    } else {
        tv = {one, r28::fe_zero()};                                            // encodings past the end run on t = (1, 0)
    }
    stv(team, BLSVM_H1_T + 2 * i, tv.a); stv(team, BLSVM_H1_T + 2 * i + 1, tv.b);
    if (sgn(tv.b)) stv(team, S_PAR + i, one); else st_zero(team, S_PAR + i);     // parity: t.c1 > (-t).c1
    const fe2 tt = r28::sqr(tv);
    const fe2 w = {r28::norm(r28::add(tt.a, r28::mulc<5>(one))), r28::norm(r28::add(tt.b, r28::mulc<4>(one)))};   // t^2 + b' + 1
    const fe2 wt = r28::mul(w, tv);
    const fe ni = inv(r28::dot2(wt.a, wt.a, wt.b, wt.b));
    const fe2 iwt = {r28::mul(wt.a, ni), r28::mul(r28::neg(wt.b), ni)};
    const fe2 wi = r28::mul(tv, iwt), ti = r28::mul(w, iwt);                     // 1/w, 1/t
    stv(team, S_FT + i, r28::mul(tv, ti).a);                                     // 1, or 0 when t = 0
    const fe2 wp = scale(r28::mul(wi, tv), cst(s3c));                            // w' = sqrt(-3) t / w
    const fe2 wpt = r28::mul(wp, tv);
    const fe2 x1 = {r28::norm(r28::sub(cst(hhc), wpt.a)), r28::norm(r28::neg(wpt.b))};
    const fe2 x2 = {r28::norm(r28::sub(r28::neg(one), x1.a)), r28::norm(r28::neg(x1.b))};
    const fe2 wpi = scale(r28::mul(w, ti), cst(sic));                            // 1/w'
    const fe2 q3 = r28::sqr(wpi);
    const fe2 x3 = {r28::norm(r28::add(q3.a, one)), q3.b};
    const fe2 xs[3] = {x1, x2, x3};
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
        const fe2 x = xs[j];
        const fe2 x3p = r28::mul(r28::sqr(x), x);
        const fe four = r28::norm(r28::mulc<4>(one));
        const fe2 u = {red(r28::add(x3p.a, four)), red(r28::add(x3p.b, four))};
        const fe n = r28::dot2(u.a, u.a, u.b, u.b);                             // N(u)
        const uint32_t k = 3 * i + j;
        stv(team, S_X + 2 * k, x.a); stv(team, S_X + 2 * k + 1, x.b);
        stv(team, BLSVM_H1_U + 2 * k, u.a); stv(team, BLSVM_H1_U + 2 * k + 1, u.b);
        // a candidate whose u has zero imaginary part has no y in the reference: n' = 0 keeps h1_b from choosing it
        if (r28::is_zero(u.b)) st_zero(team, BLSVM_H1_N + k); else stv(team, BLSVM_H1_N + k, n);
        stv(team, BLSVM_H1_ACC + k, n); stv(team, BLSVM_H1_BASE + k, n);
    }
}
#else
;
#endif

__global__ void __launch_bounds__(64) k_h2c_sw1(uint32_t total, uint32_t* __restrict__ img)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    using namespace swl;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    uint32_t* team = img + (size_t)(e / NE) * H1_IMG * 12;
    const uint32_t i = e % NE;
    const int32_t halfc[r28::NL] = BLS28_HALF;
    const fe inv2 = cst(halfc);
    fe c[2], r[3];
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) {
        const uint32_t k = 3 * i + j;
        const fe z = ldv(team, BLSVM_H1_ACC + k), n = ldv(team, BLSVM_H1_N + k);
        r[j] = r28::mul(z, n);
        const fe chi = r28::mul(r[j], z);
        if (j < 2) c[j] = r28::mul(r28::add(r28::sqr(chi), chi), inv2);           // 1 iff chi = 1 (0 for chi = -1 and for n' = 0)
    }
    // x1 if c1 else (x2 if c2 else x3) (index rule of ec.py:489-500), as arithmetic selects v = B + c (A - B)
    fe ch[5];
#pragma unroll
    for (uint32_t m = 0; m < 5; m++) {
        fe p[3];
#pragma unroll
        for (uint32_t j = 0; j < 3; j++) {
            const uint32_t k = 3 * i + j;
            p[j] = m < 2 ? ldv(team, S_X + 2 * k + m) : (m < 4 ? ldv(team, BLSVM_H1_U + 2 * k + (m - 2)) : r[j]);
        }
        const fe inner = red(r28::add(p[2], r28::mul(c[1], red(r28::sub(p[1], p[2])))));
        ch[m] = red(r28::add(inner, r28::mul(c[0], red(r28::sub(p[0], inner)))));
    }
    stv(team, S_X + 6 * i, ch[0]); stv(team, S_X + 6 * i + 1, ch[1]);
    stv(team, S_A1 + i, ch[3]);
    const fe dp = r28::mul(r28::add(ch[2], ch[4]), inv2), dm = r28::mul(r28::sub(ch[2], ch[4]), inv2);
    stv(team, BLSVM_H1_ACC + 2 * i, dp); stv(team, BLSVM_H1_BASE + 2 * i, dp);
    stv(team, BLSVM_H1_ACC + 2 * i + 1, dm); stv(team, BLSVM_H1_BASE + 2 * i + 1, dm);
}
#else
;
#endif

__global__ void __launch_bounds__(64) k_h2c_sw2(uint32_t total, uint32_t* __restrict__ img)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    using namespace swl;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    uint32_t* team = img + (size_t)(e / NE) * H1_IMG * 12;
    const uint32_t i = e % NE;
    const int32_t halfc[r28::NL] = BLS28_HALF, onec[r28::NL] = BLS28_ONE;
    const fe inv2 = cst(halfc), one = cst(onec);
    fe sq[2], chi0;
#pragma unroll
    for (uint32_t j = 0; j < 2; j++) {
        const fe z = ldv(team, BLSVM_H1_ACC + 2 * i + j), d = ldv(team, BLSVM_H1_BASE + 2 * i + j);
        sq[j] = r28::mul(z, d);
        if (j == 0) chi0 = r28::mul(sq[0], z);
    }
    const fe c = r28::mul(r28::add(one, chi0), inv2);
    const fe x0 = red(r28::add(sq[1], r28::mul(c, red(r28::sub(sq[0], sq[1])))));
    const fe a1 = ldv(team, S_A1 + i);
    const fe x1c = r28::mul(a1, inv(red(r28::add(x0, x0))));
    const bool g = sgn(x1c);
    uint32_t pw = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) pw |= team[(S_PAR + i - BLSVM_H1_STATE0) * 12 + j];
    const bool flip = g != (pw != 0u);                                          // g xor p
    const fe y0 = flip ? r28::norm(r28::neg(x0)) : x0, y1 = flip ? r28::norm(r28::neg(x1c)) : x1c;
    const fe ft = ldv(team, S_FT + i);
    // t = 0: the projective point at infinity (0, 1, 0)  (ec.py:450-452)
    stv(team, BLSVM_H1_S + 5 * i, r28::mul(ft, ldv(team, S_X + 6 * i)));
    stv(team, BLSVM_H1_S + 5 * i + 1, r28::mul(ft, ldv(team, S_X + 6 * i + 1)));
    stv(team, BLSVM_H1_S + 5 * i + 2, r28::add(one, r28::mul(ft, red(r28::sub(y0, one)))));
    stv(team, BLSVM_H1_S + 5 * i + 3, r28::mul(ft, y1));
    stv(team, BLSVM_H1_S + 5 * i + 4, ft);
}
#else
;
#endif

// ---------------------------------------------------------------------------
// Kernel H2: one team = BLSVM_H2_NM messages: P = S0 + S1, cofactor clearing,
// canonical affine bytes (x.c0 || x.c1 || y.c0 || y.c1, 192 B per message).
// enc = the stage image: encoding e sits in team e / NE, slots S + 5 (e % NE) .. + 5.
__global__ void __launch_bounds__(64, 3) k_h2c_clear(VmTables T, const uint32_t* __restrict__ enc, uint32_t n_msg,
                                                     uint32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t first = blockIdx.x * BLSVM_H2_NM;
    team_init_consts_h2c(T, team, lane);
    for (uint32_t d = lane; d < BLSVM_H2_NM * 120; d += 64) {
        uint32_t m = d / 120;
        uint32_t src_m = (first + m < n_msg) ? first + m : first;       // pad with a valid message
        const uint32_t e = 2 * src_m + (d % 120) / 60;                             // encoding index
        team[BLSVM_H2_S * 12 + d] = enc[((size_t)(e / BLSVM_H1_NE) * H1_IMG + (BLSVM_H1_S - BLSVM_H1_STATE0) + 5 * (e % BLSVM_H1_NE)) * 12 + d % 60];
    }
    wave_fence();
    // the two double-and-add chains hold products and sums only: the light interpreter
    run_rounds<true>(T, T.h2flat, BLSVM_H2_FLAT_LEN - BLSVM_H2_FINAL_LEN, 0, lane);
    run_rounds(T, T.h2flat + (BLSVM_H2_FLAT_LEN - BLSVM_H2_FINAL_LEN), BLSVM_H2_FINAL_LEN, 0, lane);
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
#else
;
#endif

// ---------------------------------------------------------------------------
// Cofactor clearing in registers (the same formulas as vmgen/h2c_programs.build_h2: complete projective addition / doubling of
// Renes-Costello-Batina for a = 0, the endomorphism psi, ec.py:536-550).  The point arithmetic of one message is strictly
// sequential (two double-and-add chains by |x|), so the 64-lane VM needs five messages per wavefront to keep its lanes busy;
// the register forms carry a message per lane pair / lane quad.  (Round 2's form with one message per LANE -- 460 registers and
// 6 KB of scratch -- lost to the lane-pair form at every size and was removed in round 5.)
// enc = the stage image (see k_h2c_stage): encoding e sits in team e / NE, slots S + 5 (e % NE) .. + 5.
// out: n_msg x 192 bytes canonical affine (x.c0 || x.c1 || y.c0 || y.c1), (0,0) for infinity.

// ONE MESSAGE PER LANE PAIR (round 3): the Fq2 split of blsgpu_ml.hip / sp2 (even lane real parts, odd
// lane imaginary parts).  One message per lane held 460 registers and 6 KB of scratch per lane; here a lane holds half of
// every coordinate (256 registers, two wavefronts per SIMD), the point operations exist ONCE in the code -- the
// clearing is a script (fexp_tables_gfx950.h BLS28_H2C_OPS, vmgen/gen_fexp.h2c_clear_script) over one point in
// registers and five lane-private slots in HBM -- and 16 384 messages already give 512 wavefronts.
__global__ void __launch_bounds__(256, 2) k_h2c_clear_pairs(VmTables T, const uint32_t* __restrict__ enc, uint32_t n_msg, uint32_t* __restrict__ ws,
                                                           uint32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    using namespace sp2;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t m = min(tid >> 1, n_msg - 1u), part = tid & 1u;
    const bool store = (tid >> 1) < n_msg;
    auto vmh = [&](const uint32_t* p) {                           // the lane's half of a VM value pair at p (12 words each)
        uint32_t x[12];
#pragma unroll
        for (int j = 0; j < 12; j++) x[j] = p[12 * part + j];
        const r28::fe t = r28::from_vm(x);
        h r;
#pragma unroll
        for (int j = 0; j < r28::NL; j++) r.v[j] = t.v[j];
        return r;
    };
    uint32_t* slots = ws + (size_t)tid * BLS28_H2C_NSLOTS * 3 * r28::NL;
    auto st_slot = [&](uint32_t sl, const pt& P) {
        uint32_t* o = slots + sl * 3 * r28::NL;
#pragma unroll
        for (int j = 0; j < r28::NL; j++) { o[j] = (uint32_t)P.X.v[j]; o[r28::NL + j] = (uint32_t)P.Y.v[j]; o[2 * r28::NL + j] = (uint32_t)P.Z.v[j]; }
    };
    auto ld_slot = [&](uint32_t sl) {
        const uint32_t* o = slots + sl * 3 * r28::NL;
        pt P;
#pragma unroll
        for (int j = 0; j < r28::NL; j++) { P.X.v[j] = (int32_t)o[j]; P.Y.v[j] = (int32_t)o[r28::NL + j]; P.Z.v[j] = (int32_t)o[2 * r28::NL + j]; }
        return P;
    };
    pt acc;
    for (int sidx = 1; sidx >= 0; sidx--) {                       // S1 -> slot 0, S0 -> the accumulator
        const uint32_t e = 2 * m + sidx;
        const uint32_t* src = enc + ((size_t)(e / BLSVM_H1_NE) * H1_IMG + (BLSVM_H1_S - BLSVM_H1_STATE0) + 5 * (e % BLSVM_H1_NE)) * 12;
        acc.X = vmh(src); acc.Y = vmh(src + 24);
        uint32_t x[12];
#pragma unroll
        for (int j = 0; j < 12; j++) x[j] = src[48 + j];
        const r28::fe z = r28::from_vm(x);
#pragma unroll
        for (int j = 0; j < r28::NL; j++) acc.Z.v[j] = part ? 0 : z.v[j];
        if (sidx == 1) st_slot(0, acc);
    }
    constexpr uint32_t PSIX = BLSVM_HC_PSIX - BLSVM_HC_SLOT0 + BLSVM_HC_TBL0, PSIY = BLSVM_HC_PSIY - BLSVM_HC_SLOT0 + BLSVM_HC_TBL0;
    const h psix = vmh(T.consts + PSIX * 12), psiy = vmh(T.consts + PSIY * 12);
#pragma unroll 1
    for (uint32_t pc = 0; pc < (uint32_t)BLS28_H2C_NOPS; pc++) {
        const uint32_t op = BLS28_H2C_OPS[pc][0], sl = BLS28_H2C_OPS[pc][1];
        if (op == 1u || op == 2u) {
            pt Q = ld_slot(sl);
            if (op == 2u) Q.Y = norm(neg(Q.Y));
            acc = padd(acc, Q);
        } else if (op == 3u) {
            st_slot(sl, acc);
        } else if (op == 4u) {
            acc = ld_slot(sl);
        } else if (op == 5u) {
            acc = pdbl(acc);
        } else if (op == 7u) {                                   // a long run of doublings: through Jacobian coordinates
            acc = to_jacobian(acc);
        } else if (op == 8u) {
            acc = pdblj(acc);
        } else if (op == 9u) {
            acc = to_homogeneous(acc);
        } else if (op == 6u) {                                   // psi: (conj X psix, conj Y psiy, conj Z)
            S<1> cx, cy, cz;
    #pragma unroll
        for (int j = 0; j < r28::NL; j++) { cx.v[j] = part ? -acc.X.v[j] : acc.X.v[j]; cy.v[j] = part ? -acc.Y.v[j] : acc.Y.v[j]; cz.v[j] = part ? -acc.Z.v[j] : acc.Z.v[j]; }
            acc.X = mul(left(cx), right(psix));
            acc.Y = mul(left(cy), right(psiy));
            acc.Z = norm(cz);
        } else {
            break;
        }
    }
    // affine: (X, Y) / Z with 1 / Z = conj(Z) / N(Z); Z = 0 gives (0, 0).  safegcd inversion of fq32.h on the VM's form of the norm.
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
    if (store) {
        const h* o[2] = {&xa, &ya};
        for (int k = 0; k < 2; k++) {
            r28::fe t;
    #pragma unroll
        for (int j = 0; j < r28::NL; j++) t.v[j] = o[k]->v[j];
            uint32_t y[12];
            r28::to_raw(y, t);
#pragma unroll
            for (int w = 0; w < 12; w++) out[(size_t)m * 48 + (2 * k + part) * 12 + w] = bswap32(y[11 - w]);
        }
    }
}
#else
;
#endif

// The same with ONE MESSAGE PER LANE QUAD (round 4; sp4 in blsgpu_msm.hip): the two pairs of a quad share the levels of
// independent Fq2 products of every point operation, so the script is about half as deep -- what counts while the batch
// leaves SIMDs empty on lane pairs (16 384 messages: 512 wavefronts for 1024 SIMDs).  Same script, same slots (written
// lane-private like the pair kernel's), same results.
__global__ void __launch_bounds__(256, 2) k_h2c_clear_quads(VmTables T, const uint32_t* __restrict__ enc, uint32_t n_msg, uint32_t* __restrict__ ws,
                                                           uint32_t* __restrict__ out)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    using namespace sp2;                                          // (the point operations below are sp4's, named so)
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t m = min(tid >> 2, n_msg - 1u), part = tid & 1u;
    const bool first_pair = (tid & 2u) == 0u;
    const bool store = (tid >> 2) < n_msg && first_pair;
    auto vmh = [&](const uint32_t* p) {                           // the lane's half of a VM value pair at p (12 words each)
        uint32_t x[12];
#pragma unroll
        for (int j = 0; j < 12; j++) x[j] = p[12 * part + j];
        const r28::fe t = r28::from_vm(x);
        h r;
#pragma unroll
        for (int j = 0; j < r28::NL; j++) r.v[j] = t.v[j];
        return r;
    };
    uint32_t* slots = ws + (size_t)tid * BLS28_H2C_NSLOTS * 3 * r28::NL;      // every lane its own rows (the two pairs of a quad hold the same values)
    auto st_slot = [&](uint32_t sl, const pt& P) {
        uint32_t* o = slots + sl * 3 * r28::NL;
#pragma unroll
        for (int j = 0; j < r28::NL; j++) { o[j] = (uint32_t)P.X.v[j]; o[r28::NL + j] = (uint32_t)P.Y.v[j]; o[2 * r28::NL + j] = (uint32_t)P.Z.v[j]; }
    };
    auto ld_slot = [&](uint32_t sl) {
        const uint32_t* o = slots + sl * 3 * r28::NL;
        pt P;
#pragma unroll
        for (int j = 0; j < r28::NL; j++) { P.X.v[j] = (int32_t)o[j]; P.Y.v[j] = (int32_t)o[r28::NL + j]; P.Z.v[j] = (int32_t)o[2 * r28::NL + j]; }
        return P;
    };
    pt acc;
    for (int sidx = 1; sidx >= 0; sidx--) {                       // S1 -> slot 0, S0 -> the accumulator
        const uint32_t e = 2 * m + sidx;
        const uint32_t* src = enc + ((size_t)(e / BLSVM_H1_NE) * H1_IMG + (BLSVM_H1_S - BLSVM_H1_STATE0) + 5 * (e % BLSVM_H1_NE)) * 12;
        acc.X = vmh(src); acc.Y = vmh(src + 24);
        uint32_t x[12];
#pragma unroll
        for (int j = 0; j < 12; j++) x[j] = src[48 + j];
        const r28::fe z = r28::from_vm(x);
#pragma unroll
        for (int j = 0; j < r28::NL; j++) acc.Z.v[j] = part ? 0 : z.v[j];
        if (sidx == 1) st_slot(0, acc);
    }
    constexpr uint32_t PSIX = BLSVM_HC_PSIX - BLSVM_HC_SLOT0 + BLSVM_HC_TBL0, PSIY = BLSVM_HC_PSIY - BLSVM_HC_SLOT0 + BLSVM_HC_TBL0;
    const h psix = vmh(T.consts + PSIX * 12), psiy = vmh(T.consts + PSIY * 12);
#pragma unroll 1
    for (uint32_t pc = 0; pc < (uint32_t)BLS28_H2C_NOPS; pc++) {
        const uint32_t op = BLS28_H2C_OPS[pc][0], sl = BLS28_H2C_OPS[pc][1];
        if (op == 1u || op == 2u) {
            pt Q = ld_slot(sl);
            if (op == 2u) Q.Y = norm(neg(Q.Y));
            acc = sp4::padd(acc, Q);
        } else if (op == 3u) {
            st_slot(sl, acc);
        } else if (op == 4u) {
            acc = ld_slot(sl);
        } else if (op == 5u) {
            acc = sp4::pdbl(acc);
        } else if (op == 7u) {                                   // a long run of doublings: through Jacobian coordinates
            acc = sp4::to_jacobian(acc);
        } else if (op == 8u) {
            acc = sp4::pdblj(acc);
        } else if (op == 9u) {
            acc = sp4::to_homogeneous(acc);
        } else if (op == 6u) {                                   // psi: (conj X psix, conj Y psiy, conj Z): one product per pair
            S<1> cx, cy, cz;
#pragma unroll
            for (int j = 0; j < r28::NL; j++) { cx.v[j] = part ? -acc.X.v[j] : acc.X.v[j]; cy.v[j] = part ? -acc.Y.v[j] : acc.Y.v[j]; cz.v[j] = part ? -acc.Z.v[j] : acc.Z.v[j]; }
            const h pm = mul(left(sp4::pick(cx, cy)), right(sp4::pick(psix, psiy))), po = sp4::oth(pm);
            acc.X = sp4::pick(pm, po);
            acc.Y = sp4::pick(po, pm);
            acc.Z = norm(cz);
        } else {
            break;
        }
    }
    // affine: (X, Y) / Z with 1 / Z = conj(Z) / N(Z); Z = 0 gives (0, 0).  safegcd inversion of fq32.h on the VM's form of the norm.
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
    if (store) {
        const h* o[2] = {&xa, &ya};
        for (int k = 0; k < 2; k++) {
            r28::fe t;
    #pragma unroll
        for (int j = 0; j < r28::NL; j++) t.v[j] = o[k]->v[j];
            uint32_t y[12];
            r28::to_raw(y, t);
#pragma unroll
            for (int w = 0; w < 12; w++) out[(size_t)m * 48 + (2 * k + part) * 12 + w] = bswap32(y[11 - w]);
        }
    }
}
#else
;
#endif

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
                         ACC = BLSVM_D1_ACC, BASE = BLSVM_D1_BASE, STATE0 = BLSVM_D1_STATE0, IMG = BLSVM_D1_STATE1 - BLSVM_D1_STATE0;
    static __host__ __device__ constexpr uint32_t seg_off(int stage) { return stage == 0 ? BLSVM_SEGF_D1_A_OFF : BLSVM_SEGF_D1_C_OFF; }
    static __host__ __device__ constexpr uint32_t seg_len(int stage) { return stage == 0 ? BLSVM_SEGF_D1_A_LEN : BLSVM_SEGF_D1_C_LEN; }
};
template <> struct DecompCfg<2> {
    static constexpr int NE = BLSVM_D2_NE, SLOTS = BLSVM_D2_SLOTS, X = BLSVM_D2_X, BIG = BLSVM_D2_BIG, OUT = BLSVM_D2_OUT,
                         ACC = BLSVM_D2_ACC, BASE = BLSVM_D2_BASE, STATE0 = BLSVM_D2_STATE0, IMG = BLSVM_D2_STATE1 - BLSVM_D2_STATE0;
    static __host__ __device__ constexpr uint32_t seg_off(int stage) {
        return stage == 0 ? BLSVM_SEGF_D2_A_OFF : (stage == 1 ? BLSVM_SEGF_D2_B_OFF : BLSVM_SEGF_D2_C_OFF);
    }
    static __host__ __device__ constexpr uint32_t seg_len(int stage) {
        return stage == 0 ? BLSVM_SEGF_D2_A_LEN : (stage == 1 ? BLSVM_SEGF_D2_B_LEN : BLSVM_SEGF_D2_C_LEN);
    }
};

// Stages: 0: bytes -> d*_a -> image [k_pow]; (DEG 2 only) 1: image -> d2_b -> image [k_pow];
// 2: image -> d*_c -> canonical bytes + accept flags.
template <int DEG, int STAGE>
__global__ void __launch_bounds__(64, 2) k_decompress(VmTables T, const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ img,
                                                      uint32_t* __restrict__ out, uint8_t* __restrict__ ok)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    using C = DecompCfg<DEG>;
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t first = blockIdx.x * C::NE;
    uint32_t* my = img + (size_t)blockIdx.x * C::IMG * 12;
    team_init_consts_h2c(T, team, lane);
    wave_fence();
    if (STAGE == 0) {
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
    } else {
        img_load(team, my, C::STATE0, C::IMG, lane);
    }
    wave_fence();
    run_rounds(T, T.segflat + C::seg_off(STAGE), C::seg_len(STAGE), 0, lane);
    wave_fence();
    if (STAGE != 2) {
        img_store(team, my, C::STATE0, C::IMG, lane);
        return;
    }
    constexpr uint32_t NOUT = 2 * DEG + 1;                         // x, y coordinates and the flag
    for (uint32_t d = lane; d < C::NE * NOUT; d += 64) {           // relaxed -> canonical residues
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
#else
;
#endif

#ifdef BLSGPU_STAMPS
// diagnostic build only: see blsgpu_debug_run
__global__ void __launch_bounds__(64) k_debug_run(VmTables T, const uint2* seq, uint32_t nrounds, uint32_t nslots, uint32_t* image, uint32_t light)
#if BLSGPU_EMIT(BLSGPU_TU_H2C)
{
    uint32_t* team = reinterpret_cast<uint32_t*>(smem4);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < nslots * 12; i += 64) team[i] = image[i];
    wave_fence();
    if (light) {                                             // the multi-pair Miller program: SAVE / RESTORE rounds
        uint32_t stash[3] = {0u, 0u, 0u};
        run_rounds<true>(T, seq, nrounds, 0, lane, stash);
    } else {
        run_rounds(T, seq, nrounds, 0, lane);
    }
    wave_fence();
    for (uint32_t i = lane; i < nslots * 12; i += 64) image[i] = team[i];
}
#else
;
#endif
#endif

// every instantiation the host side launches: this translation unit is the one that emits them (blsgpu_tu.h)
#if BLSGPU_TU == BLSGPU_TU_H2C
__attribute__((used)) static const void* const blsgpu_instances_h2c[] = {
    (const void*)&k_h2c_stage<0, 0>,
    (const void*)&k_h2c_stage<0, 1>,
    (const void*)&k_h2c_stage<1, 0>,
    (const void*)&k_h2c_stage<2, 0>,
    (const void*)&k_h2c_sw0<0>,
    (const void*)&k_h2c_sw0<1>,
    (const void*)&k_h2c_swj0<0>,
    (const void*)&k_h2c_swj0<1>,
    (const void*)&k_decompress<1, 0>,
    (const void*)&k_decompress<1, 1>,
    (const void*)&k_decompress<1, 2>,
    (const void*)&k_decompress<2, 0>,
    (const void*)&k_decompress<2, 1>,
    (const void*)&k_decompress<2, 2>};
#endif
}  // namespace blsgpu
