// blsgpu_api.hip -- host side of the C ABI declared in include/blsgpu.h.
// Pure HIP runtime: no torch types, no CPU fallback.  If no GPU is usable the
// context cannot be created and every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/blsgpu.h"
#include "blsgpu_tu.h"
#include "blsgpu_kernels.hip"
#include "fp28.h"
#include "blsgpu_ml.hip"
#include "blsgpu_fexp.hip"
#include "blsgpu_fexpw.hip"
#include "blsgpu_mlw.hip"
#include "blsgpu_lsw.hip"
#include "blsgpu_g1w.hip"
#include "blsgpu_msm.hip"
#include "blsgpu_h2c.hip"
#include "blsgpu_h2cw.hip"
#include "blsgpu_probe.hip"

#if BLSGPU_EMIT(BLSGPU_TU_HOST)
namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(-EIO, std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

constexpr int MILLER_WAVES = 4;        // teams (pairings) per workgroup in k_miller
constexpr int REDUCE_WAVES = 8;        // teams per workgroup in k_reduce
constexpr size_t BATCH_TREE_MIN_GROUP = 24;   // batches of groups at least this long use the per-group product tree
constexpr int REDUCE_PER_BLOCK = 64;   // partials folded by one k_reduce block
constexpr size_t LS_MAX_PAIRS = (size_t)1 << 20;       // pairs per line-stream launch sequence: 24 GB of line records

}  // namespace

struct blsgpu_ctx {
    int device = 0;
    blsgpu::VmTables tabs{};
    void* d_tables = nullptr;          // one allocation holding every table
    uint32_t* d_part[2] = {nullptr, nullptr};
    size_t part_cap = 0;               // capacity of each partial buffer, in partials
    void* d_io = nullptr;              // staging for the host-buffer entry points
    size_t io_cap = 0;
    uint32_t* d_out = nullptr;         // 576-byte result staging
    uint32_t* d_degen = nullptr;       // [0] count, [1 ..] block indices of degenerate pairs (k_miller_slow's work list)
    size_t degen_cap = 0;
    size_t miller_wide3_max = 256;     // ... with the accumulator split over two wavefronts (three per pair) up to this many pairs: three SIMDs per pair are free
    size_t miller_wide_max = 1536;     // calls of at most this many pairs run the wide Miller loop (blsgpu_mlw.hip: one pair per two-wavefront workgroup, a product per lane); 0: never
    size_t mp_threshold = 4096;        // pairs from which k_miller_mp is used
    size_t mp3_threshold = (size_t)-1; // ... with three pairs per wavefront from here on, two below; -1: the measured schedule
    size_t pip_threshold = 4096;       // points from which a single sum uses the bucket method
    size_t pip_group_threshold = 48;   // points per sum from which a batch of sums does
    size_t pow2_max = 32768;           // fixed-exponent powers (hash to G2, decompression): up to this many values per launch two wavefronts per 64 values (k_pow2: 0.32 ms against 0.47); 0: never
    bool msm_wide_tail = true;         // the sorted-bucket G1 sum: window sums and the Horner over the windows on the wide machine (k_msm_horner_wide: 0.9 ms against the wavefront VM's 1.45); false: k_srt_windows + k_msm_pip_horner<1>
    size_t h2c_wide_max = 2048;        // up to this many messages the cofactor clearing runs one message per WAVEFRONT with a product per lane (blsgpu_h2cw.hip: the latency form); 0: never
    size_t h2c_reg_threshold = 8192;   // messages from which cofactor clearing runs in registers (one message per lane PAIR; measured: DESIGN.md 2c)
    size_t h2c_lane_threshold = 2048;  // messages from which the three encoding stages run one encoding per lane (k_h2c_sw0/1/2)
    bool h2c_jacobi = true;            // ... with the quadratic characters decided by a Jacobi-symbol routine: two powers per encoding, not five
    size_t h2c_jacobi_threshold = 16384;   // ... from this many messages (below, five parallel powers finish sooner than three serial symbol loops)
    size_t h2c_quad_max = 16384;       // ... on lane QUADS up to this many messages (k_h2c_clear_quads: half the depth while the chip is not full)
    bool test_ls_nomem = false;        // test hook (BLSGPU_TEST_LS_NOMEM=1): the line-stream workspace "cannot be allocated"
    void* d_h2c_ws = nullptr;          // the lane-private point slots of k_h2c_clear_pairs
    size_t h2c_ws_cap = 0;
    size_t msm_sort_threshold = 1;      // points from which one G1 sum with scalars uses sorted buckets (k_srt_*): since the tail runs on the wide machine (round 5) they win at every size -- 1 point 1.24 ms against 1.63, 8192 points 1.45 against 2.59 (profiles/r05_c5_window_bits.txt)
    size_t msm_sort2_threshold = 1;     // the same for ONE G2 sum with scalars (round 5: BLS.aggregate_sigs(secure) as a multi-scalar sum)
    size_t msm_plain_threshold = 2;     // points from which ONE plain sum (no scalars) runs on the register kernels (k_sum_chunks + folds; round 5) instead of the wavefront VM's k_msm
    size_t smul_min_groups = 4096;      // sums per call from which a batch of SMALL sums with scalars (scalar multiplications: groups x 1 point) runs one group per lane / lane pair (k_smul, round 5) instead of the wavefront VM's k_msm
    size_t smul_max_k = 8;              // ... for sums of up to this many points
    static uint32_t msm_sort2_bits(size_t n) { return n >= 16384 ? 13 : (n >= 512 ? 11 : 9); }   // window bits of the G2 path by size (tools/g2_single_sum_probe.py, profiles/r05_g2_single_sum.txt)
    size_t horner_np_threshold = 1024; // G2 sums per call from which the window Horner runs several sums per team
    size_t horner_quads_threshold = 2; // G2 sums per call (lane-pair bucket kernel) from which the window Horner runs one sum per lane quad
    size_t wg256_max_waves = 4096;     // register kernels: launches of up to this many wavefronts go out as 256-thread workgroups (blsgpu_tu.h)
    size_t msm_lane_threshold = 65536; // points from which the bucket sums run one (group, chunk, window) per lane
    bool msm_lane_pairs = true;        // G2: every (group, chunk, window) on a lane PAIR (k_msm_lane2x) instead of one lane
    uint32_t* d_buckets = nullptr;     // their buckets (HBM)
    size_t bucket_cap = 0;
    uint32_t* d_msm_part = nullptr;    // MSM partials
    // line-stream multi-pairing (blsgpu_ml.hip): used from ls_threshold pairs per call when every group has at least
    // ls_min_group pairs
    size_t ls_threshold = 2304;        // measured crossover (tools/ls_wide_sweep.py, round 5 with the point chains sixteen lanes per pair): 2048 pairs 1.70 (VM) vs 1.70 ms, 3072 pairs 1.93 vs 1.79; 5120 in round 4, 16 384 in round 3
    size_t ls_min_group = 64;
    size_t ls_teams = 163840;          // accumulators k_ml_accum aims at (10 per wavefront: 8 wavefronts per place at two per SIMD)
    void* d_lines = nullptr;           // 68 x pairs line records
    size_t lines_cap = 0;              // bytes
    void* d_lsp[2] = {nullptr, nullptr};   // dense partial products (ping-pong over the merge levels)
    size_t lsp_cap[2] = {0, 0};        // bytes
    void* d_bad = nullptr;             // one byte per pair: left to the slow program
    void* d_exflags = nullptr;         // blsgpu_miller_loop_batch's fast form: the caller's flags with "py = 0" marked (2 bytes per pair)
    size_t exflags_cap = 0;
    size_t bad_cap = 0;
    bool vm_exact_lanes = true;        // degenerate blocks of the VM kernels through the lane kernels (k_ml_lines_exact / k_ml_small) instead of k_miller_slow
    bool miller_exact_lanes = true;    // blsgpu_miller_loop_batch (one exact Fq12 per pair) on the lane kernels (k_ml_lines_exact + k_ml_small, round 5) instead of the VM's k_miller_exact
    bool miller_exact_fast = true;     // ... from the FAST lines: the line-stream kernels + one Fq2 factor per pair (k_ml_exact_fixup) instead of the reference's 73 affine slopes per pair; false: k_ml_lines_exact for every pair
    size_t ls_merge_wide_max = 16384;  // merge levels with at most this many outputs run one wavefront per output
    size_t ls_wide_max = 5120;         // calls of at most this many pairs run the point chains sixteen lanes per pair with the values in LDS (k_ml_lines_wide, blsgpu_lsw.hip); 0: never
    size_t ls_quad_max = 20480;        // calls of at most this many pairs run the point chains on lane QUADS (k_ml_lines4: 0.6 of the depth while lane pairs leave SIMDs empty)
    size_t fexp_team_threshold = 5120; // results per call from which the final exponentiations run six lanes each (blsgpu_fexp.hip); below: one result per wavefront (measured crossover, tools/fexp_latency.py)
    bool fexp_wide = true;             // fewer results than that: one result per wavefront, a product per lane (blsgpu_fexpw.hip); false: the VM program
    size_t fexp_wide_max_partials = 8; // ... which also multiplies up to this many partials per result itself (a dense product is ~2.5 us)
    void* d_fexp_dbg = nullptr;        // tools/fexp_trace.py: the accumulator of result 0 after every operation of the script (k_fexp_team)
    void* d_fexpw_stamps = nullptr;    // tools/fexpw_stamps.py: cycle counter of result 0 around every operation of the script (k_fexp_wide)
    void* d_fexp_ws = nullptr;         // their slots
    size_t fexp_ws_cap = 0;
    hipEvent_t bulk_event = nullptr;   // caller's event, recorded after the last chip-filling kernel of a Miller stage
    size_t msm_part_cap = 0;           // in u32
    // optional per-kernel timing (blsgpu_timing_enable): HIP events recorded on
    // the launch stream around every kernel, ring of TIMING_SLOTS launches
    bool timing = false;
    static constexpr int TIMING_SLOTS = 1024;
    hipEvent_t* ev0 = nullptr;
    hipEvent_t* ev1 = nullptr;
    int* ev_kind = nullptr;            // 0 k_miller, 1 k_reduce, 2 k_reduce with final exponentiation
    size_t ev_count = 0;
    // Workspace buffers only ever GROW: the buffer a larger one replaces is kept until the
    // context is destroyed (or trimmed by blsgpu_ctx_reserve on an idle context), so work
    // already enqueued on it stays valid and no hipFree -- a device-wide synchronisation --
    // happens inside a pipeline.
    std::vector<void*> retired;
    // The workspace is shared by everything a context launches: a call on another stream than
    // the previous one first waits for that one's work (StreamGuard).
    hipStream_t last_stream = nullptr;
    hipEvent_t last_event = nullptr;
    bool used = false;
};

namespace {
struct KernelTimer {
    blsgpu_ctx* c; hipStream_t st; int slot;
    KernelTimer(blsgpu_ctx* c_, hipStream_t st_, int kind) : c(c_), st(st_), slot(-1) {
        if (c->timing && c->ev_count < (size_t)blsgpu_ctx::TIMING_SLOTS) {
            slot = (int)c->ev_count++;
            c->ev_kind[slot] = kind;
            (void)hipEventRecord(c->ev0[slot], st);
        }
    }
    ~KernelTimer() { if (slot >= 0) (void)hipEventRecord(c->ev1[slot], st); }
};
}  // namespace

// Batches of at least mp_threshold pairs use the multi-pair program (k_miller_mp:
// fewer instructions per pairing, longer per-batch latency); smaller ones the
// one-pair-per-wavefront program (k_miller).  Default 4096; per context via
// blsgpu_ctx_set_mp_threshold, or BLSGPU_MP_THRESHOLD in the environment.
static size_t default_mp_threshold() {
    const char* e = getenv("BLSGPU_MP_THRESHOLD");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)4096;
}
static bool use_mp(const blsgpu_ctx* c, size_t n) { return n >= c->mp_threshold; }
// Two or three pairs per wavefront?  Three costs fewest instructions per pairing (large batches), two fills the
// chip sooner: 4096 teams are one "round" of the chip, so teams of two win while the batch is a little under a
// multiple of 8192 pairs and teams of three where it is a little under a multiple of 12288 -- measured crossovers
// (tools/mp_threshold_sweep.py, profiles/r02_schedule_experiments.txt); from ~20 000 pairs on blocks flow
// continuously and three wins by 8 %.
static bool use_mp2(const blsgpu_ctx* c, size_t n) {
    if (c->mp3_threshold != (size_t)-1) return n < c->mp3_threshold;
    return n <= 8704 || (n > 9728 && n <= 11264) || (n > 14336 && n <= 18432);
}
// Grid and workgroup size for `waves` independent wavefronts of a register kernel (blsgpu_tu.h: wave_index()): four
// wavefronts per workgroup -- one per SIMD of a CU -- while the launch does not fill the chip several times over.
struct WaveShape { unsigned blocks, threads; };
static WaveShape wave_shape(const blsgpu_ctx* c, size_t waves) {
    const unsigned per = (waves <= c->wg256_max_waves) ? 4u : 1u;
    return {(unsigned)((waves + per - 1) / per), per * 64u};
}
// grow-only (see blsgpu_ctx::retired): *p gets at least `bytes`; contents are scratch, not copied
static int grow_buffer(blsgpu_ctx* c, void** p, size_t* cap_bytes, size_t bytes) {
    if (bytes <= *cap_bytes) return 0;
    size_t want = bytes + bytes / 4;                     // headroom: fewer regrowths
    void* n = nullptr;
    if (hipMalloc(&n, want) != hipSuccess) {
        (void)hipGetLastError();                         // the failed attempt must not show up in a later launch check
        want = bytes;
        HIP_TRY(hipMalloc(&n, want));
    }
    if (*p) c->retired.push_back(*p);
    *p = n;
    *cap_bytes = want;
    return 0;
}
template <class T>
static int grow_elems(blsgpu_ctx* c, T** p, size_t* cap_elems, size_t elems) {
    if (elems <= *cap_elems) return 0;
    size_t bytes = *cap_elems * sizeof(T);
    int rc = grow_buffer(c, (void**)p, &bytes, elems * sizeof(T));
    if (rc) return rc;
    *cap_elems = bytes / sizeof(T);
    return 0;
}
static int ensure_workspace(blsgpu_ctx* c, size_t max_pairs) {
    // one partial per Miller block: at most a team of two pairs (k_miller_mp<2>), plus the levels of the reduce chain
    size_t need = (max_pairs + 1) / 2 + (max_pairs + MILLER_WAVES - 1) / MILLER_WAVES + 1;
    if (need > c->part_cap) {
        size_t cap0 = c->part_cap * 144, cap1 = c->part_cap * 144;     // in u32
        int rc = grow_elems(c, &c->d_part[0], &cap0, need * 144);
        if (!rc) rc = grow_elems(c, &c->d_part[1], &cap1, need * 144);
        if (rc) return rc;
        c->part_cap = (cap0 < cap1 ? cap0 : cap1) / 144;
    }
    // one work-list entry per Miller block at most (+ the counter)
    return grow_elems(c, &c->d_degen, &c->degen_cap, c->part_cap + 2);
}
namespace {
// Serialises the use of the context's workspace across streams (blsgpu_ctx::last_stream).
struct StreamGuard {
    blsgpu_ctx* c; hipStream_t st;
    StreamGuard(blsgpu_ctx* c_, hipStream_t st_) : c(c_), st(st_) {
        if (c->used && c->last_stream != st && c->last_event) (void)hipStreamWaitEvent(st, c->last_event, 0);
    }
    ~StreamGuard() {
        if (c->last_event) (void)hipEventRecord(c->last_event, st);
        c->last_stream = st;
        c->used = true;
    }
};
}  // namespace

// fixed-exponent powers on a stage image (blsgpu_h2c.hip)
static int launch_pow(blsgpu_ctx* c, uint32_t* img, uint32_t img_slots, uint32_t base_off, uint32_t acc_off, size_t teams, uint32_t cnt,
                      hipStream_t st) {
    size_t total = teams * cnt;
    if (total <= c->pow2_max)          // a batch that leaves SIMDs empty: two wavefronts per 64 values (the squarings a chain of their own)
        hipLaunchKernelGGL(blsgpu::k_pow2, dim3((unsigned)((total + 127) / 128)), dim3(256), 0, st, img, img_slots, base_off, acc_off, cnt,
                           (uint32_t)total);
    else
        hipLaunchKernelGGL(blsgpu::k_pow, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, img, img_slots, base_off, acc_off, cnt,
                           (uint32_t)total);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------ decompression --
namespace {
template <int DEG>
int decompress_dev(blsgpu_ctx* c, const void* d_in, size_t n, void* d_out, void* d_ok, hipStream_t st) {
    using C = blsgpu::DecompCfg<DEG>;
    if (n == 0) return 0;
    StreamGuard sg(c, st);
    if (n > 0x0FFFFFF0ull) return fail(-EINVAL, "batch too large");
    const size_t teams = (n + C::NE - 1) / C::NE;
    size_t need = teams * C::IMG * 12;
    if (int rc_ = grow_elems(c, &c->d_msm_part, &c->msm_part_cap, need)) return rc_;
    uint32_t* img = c->d_msm_part;
    const size_t lds = (size_t)C::SLOTS * 48;
    constexpr uint32_t BASE = C::BASE - C::STATE0, ACC = C::ACC - C::STATE0;
    hipLaunchKernelGGL((blsgpu::k_decompress<DEG, 0>), dim3((unsigned)teams), dim3(64), lds, st, c->tabs, (const uint32_t*)d_in,
                       (uint32_t)n, img, (uint32_t*)d_out, (uint8_t*)d_ok);
    HIP_TRY(hipGetLastError());
    int rc = launch_pow(c, img, C::IMG, BASE, ACC, teams, C::NE, st);
    if (rc) return rc;
    if (DEG == 2) {
        hipLaunchKernelGGL((blsgpu::k_decompress<DEG, 1>), dim3((unsigned)teams), dim3(64), lds, st, c->tabs, (const uint32_t*)d_in,
                           (uint32_t)n, img, (uint32_t*)d_out, (uint8_t*)d_ok);
        HIP_TRY(hipGetLastError());
        rc = launch_pow(c, img, C::IMG, BASE, ACC, teams, 2 * C::NE, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((blsgpu::k_decompress<DEG, 2>), dim3((unsigned)teams), dim3(64), lds, st, c->tabs, (const uint32_t*)d_in,
                       (uint32_t)n, img, (uint32_t*)d_out, (uint8_t*)d_ok);
    HIP_TRY(hipGetLastError());
    return 0;
}
template <int DEG>
int decompress_host(blsgpu_ctx* c, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
    if (!c || (n && (!in || !out || !ok))) return fail(-EINVAL, "NULL argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    size_t need = n * (48 * DEG + 96 * DEG + 1) + 64;
    if (int rc_ = grow_buffer(c, &c->d_io, &c->io_cap, need)) return rc_;
    char* din = (char*)c->d_io;
    char* dout = din + n * 48 * DEG;
    char* dok = dout + n * 96 * DEG;
    HIP_TRY(hipMemcpy(din, in, n * 48 * DEG, hipMemcpyHostToDevice));
    int rc = decompress_dev<DEG>(c, din, n, dout, dok, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, dout, n * 96 * DEG, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ok, dok, n, hipMemcpyDeviceToHost));
    return 0;
}
}  // namespace

// ---------------------------------------------------------------- MSM -------
namespace {
constexpr int MSM_WAVES = 4;

// One sum with scalars by sorted buckets (blsgpu_msm.hip, k_srt_*; G1 a unit per lane, G2 per lane pair): enqueues on `st` and
// returns, like every _dev path (no synchronisation, usable under stream capture).  Returns 1 only when the key list would not fit
// 32 bits; the caller then takes the fixed-window path.
template <int DEG>
static int msm_sorted(blsgpu_ctx* c, const void* d_pts, const void* d_scalars, size_t n, void* d_out, void* d_out_inf, hipStream_t st) {
    typedef blsgpu::SrtG<DEG> G;
    // Window bits: with 131 072 equal pieces a run covers 131072 / (keys per window x windows) pieces whatever n is, and a run of
    // more than three pieces costs a whole wavefront in k_srt_fix_long.  With signed digits (2^(cb-1) keys per window) 13 bits are
    // the best width at every size the sorted path serves (tools/c5_probe.py under BLSGPU_MSM_SORT_BITS, profiles/r05_c5_window_bits.txt:
    // 2^20 points 5.71 / 5.42 / 5.46 ms for 12 / 13 / 14 bits, 16 384 points 1.88 / 1.52 / 1.68).
    uint32_t cb = DEG == 1 ? (n < 512 ? 7u : 13u) : c->msm_sort2_bits(n);   // (a few points: 37 windows of 64 keys, 1.06 ms for one point against 1.24)
    if (const char* e = getenv(DEG == 1 ? "BLSGPU_MSM_SORT_BITS" : "BLSGPU_MSM_SORT2_BITS")) cb = (uint32_t)strtoul(e, nullptr, 10);
    if (cb < 5 || cb > blsgpu::SRT_MAXBITS) return fail(-EINVAL, "BLSGPU_MSM_SORT_BITS out of range");
    // signed digits (blsgpu_msm.hip): 2^(cb-1) keys per window; 258 <= cb x windows keeps the recoded scalar inside the windows
    const uint32_t kb = cb - 1, nwin = (258 + cb - 1) / cb;
    const size_t nkeys = (size_t)nwin << kb;
    if ((size_t)nwin * n > 0xFFFFFFF0ull || n >= 0x80000000ull) return 1;
    const size_t nch = ((size_t)1 << (cb - 2)) / blsgpu::SRT_BITADDS, nsum = (size_t)nwin * cb;
    const size_t units = blsgpu::SRT_LANES / G::LP;
    const bool wide = DEG == 2 || c->msm_wide_tail;           // (G2 has no tail on the wavefront VM)
    // workspace: prep | recoded scalars | cnt | start (+1) | cursor | maxcnt, total | idx | bsum | headpart | headkey | bit sums (two buffers) | winsums
    size_t off = 0;
    auto take = [&](size_t words) { size_t o = off; off += (words + 3) & ~(size_t)3; return o; };
    const size_t PJ = G::PJ;
    const size_t o_prep = take(n * G::AFF), o_rec = take(n * blsgpu::SRT_SCW), o_cnt = take(nkeys), o_start = take(nkeys + 1), o_cur = take(nkeys),
                 o_max = take(4), o_idx = take((size_t)nwin * n), o_bsum = take(nkeys * PJ), o_hp = take(units * PJ), o_hk = take(units),
                 o_b0 = take(nsum * nch * PJ), o_b1 = take(nsum * ((nch + 7) / 8) * PJ + PJ), o_win = take((size_t)nwin * PJ), o_live = take((n + 3) / 4),
                 o_wtot = take(2 * (size_t)nwin), o_long = take(nkeys + 4);
    if (int rc_ = grow_elems(c, &c->d_buckets, &c->bucket_cap, off)) return rc_;
    uint32_t* W = c->d_buckets;
    HIP_TRY(hipMemsetAsync(W + o_cnt, 0, nkeys * 4, st));
    HIP_TRY(hipMemsetAsync(W + o_long, 0, 16, st));          // counter of the long runs (the key list follows it)
    uint8_t* live = (uint8_t*)(W + o_live);
    blsgpu::SrtBias bias;
    for (uint32_t j = 0; j < blsgpu::SRT_SCW; j++) bias.w[j] = 0;
    for (uint32_t w = 0; w < nwin; w++) {
        const uint32_t pos = cb * w + cb - 1;
        bias.w[pos >> 5] |= 1u << (pos & 31);
    }
    hipLaunchKernelGGL(blsgpu::k_srt_prep<DEG>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_pts, (const uint32_t*)d_scalars, (uint32_t)n,
                       bias, W + o_prep, live, W + o_rec);
    HIP_TRY(hipGetLastError());
    const uint32_t* sc = W + o_rec;
    const dim3 sgrid((unsigned)((n + blsgpu::SRT_SLICE - 1) / blsgpu::SRT_SLICE), nwin);
    hipLaunchKernelGGL(blsgpu::k_srt_count, sgrid, dim3(1024), 0, st, sc, live, (uint32_t)n, cb, W + o_cnt);
    hipLaunchKernelGGL(blsgpu::k_srt_scan_window, dim3(nwin), dim3(1024), 0, st, W + o_cnt, kb, W + o_start, W + o_wtot, W + o_wtot + nwin);
    hipLaunchKernelGGL(blsgpu::k_srt_scan_add, dim3(nwin), dim3(1024), 0, st, nwin, kb, W + o_wtot, W + o_wtot + nwin, W + o_start, W + o_cur,
                       W + o_max);
    HIP_TRY(hipGetLastError());
    // No read-back, no host decision (round 3): whatever the digit distribution, a long run is finished by a wavefront of
    // k_srt_fix_long -- all-equal scalars cost ~100 additions per lane there (a millisecond), and only a batch whose scalars
    // leave all windows but one empty is slow (still correct, and still faster than the fixed windows it used to fall back to).
    hipLaunchKernelGGL(blsgpu::k_srt_scatter, sgrid, dim3(1024), 0, st, sc, live, (uint32_t)n, cb, W + o_cur, W + o_idx);
    const auto blocks = [&](size_t nunits) { return dim3((unsigned)((nunits * G::LP + 63) / 64)); };
    hipLaunchKernelGGL(blsgpu::k_srt_accum<DEG>, blocks(units), dim3(64), 0, st, W + o_prep, W + o_idx, W + o_start, (uint32_t)nkeys, (uint32_t)units,
                       W + o_bsum, W + o_hp, W + o_hk);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(blsgpu::k_srt_fix<DEG>, blocks(nkeys), dim3(64), 0, st, W + o_start, (uint32_t)nkeys, (uint32_t)units, W + o_hp, W + o_hk, W + o_bsum,
                       W + o_long, W + o_long + 4);
    hipLaunchKernelGGL(blsgpu::k_srt_fix_long<DEG>, dim3(1024), dim3(64), 0, st, W + o_start, (uint32_t)nkeys, (uint32_t)units, W + o_hp, W + o_bsum, W + o_long,
                       W + o_long + 4);
    const size_t btotal = nsum * nch;
    hipLaunchKernelGGL(blsgpu::k_srt_bits<DEG>, blocks(btotal), dim3(64), 0, st, W + o_bsum, nwin, cb, (uint32_t)btotal, W + o_b0);
    HIP_TRY(hipGetLastError());
    uint32_t *src = W + o_b0, *dst = W + o_b1;
    // the tail on the wide machine: result = sum_i 2^(c i) P_i over a list, one wavefront per list (blsgpu_g1w.hip, blsgpu_h2cw.hip)
    const auto horner = [&](size_t lists, const uint32_t* in, uint32_t npts, uint32_t cbits, uint32_t* out) {
        if (DEG == 1) hipLaunchKernelGGL(blsgpu::g1w::k_msm_horner_wide<0>, dim3((unsigned)lists), dim3(64), 0, st, in, npts, cbits, out, (uint8_t*)nullptr);
        else hipLaunchKernelGGL(blsgpu::h2cw::k_msm_horner_wide2<0>, dim3((unsigned)lists), dim3(64), 0, st, in, npts, cbits, out, (uint8_t*)nullptr);
    };
    if (nch == 1 && !wide) {                                  // (5-bit windows: nothing to fold, but the VM's tail reads its own form)
        hipLaunchKernelGGL(blsgpu::k_msm_lane_fold<1>, dim3((unsigned)((nsum + 63) / 64)), dim3(64), 0, st, src, 1u, 1u, 1u, (uint32_t)nsum, dst, 1u);
        src = dst;
    }
    for (size_t cur = nch; cur > 1;) {                        // runs of 8 partial sums per unit until one is left per (window, bit)
        const size_t nfold = (cur + 7) / 8, ftotal = nsum * nfold;
        if (wide && ftotal <= 4096 && (cur <= 8 || cur % 8 == 0))
            // few runs left: one wavefront per run, an addition two steps of the wide machine (a lane's own addition is ~6000 instructions)
            horner(ftotal, src, (uint32_t)(cur < 8 ? cur : 8), 0u, dst);
        else if (wide)
            hipLaunchKernelGGL(blsgpu::k_srt_fold<DEG>, blocks(ftotal), dim3(64), 0, st, src, (uint32_t)cur, 8u, (uint32_t)nfold, (uint32_t)ftotal, dst);
        else
            hipLaunchKernelGGL(blsgpu::k_msm_lane_fold<1>, dim3((unsigned)((ftotal + 63) / 64)), dim3(64), 0, st, src, (uint32_t)cur, 8u,
                               (uint32_t)nfold, (uint32_t)ftotal, dst, nfold == 1 ? 1u : 0u);   // the last fold: the VM's form for the VM's tail
        HIP_TRY(hipGetLastError());
        uint32_t* t = src; src = dst; dst = t;
        cur = nfold;
    }
    if (wide) {
        // W_w = sum_b 2^b S_(w,b), one wavefront per window; then sum_w 2^(cb w) W_w on one wavefront
        horner(nwin, src, cb, 1u, W + o_win);
        if (DEG == 1) hipLaunchKernelGGL(blsgpu::g1w::k_msm_horner_wide<1>, dim3(1), dim3(64), 0, st, W + o_win, nwin, cb, (uint32_t*)d_out, (uint8_t*)d_out_inf);
        else hipLaunchKernelGGL(blsgpu::h2cw::k_msm_horner_wide2<1>, dim3(1), dim3(64), 0, st, W + o_win, nwin, cb, (uint32_t*)d_out, (uint8_t*)d_out_inf);
    } else {
        hipLaunchKernelGGL(blsgpu::k_srt_windows, dim3(nwin), dim3(64), (size_t)blsgpu::TEAM_BYTES, st, c->tabs, src, cb, W + o_win);
        hipLaunchKernelGGL(blsgpu::k_msm_pip_horner<1>, dim3(1), dim3(64), (size_t)blsgpu::TEAM_BYTES, st, c->tabs, W + o_win, nwin, cb,
                           (uint32_t*)d_out, (uint8_t*)d_out_inf);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// One plain sum of n points (no scalars) -- BLS.aggregate_pub_keys / aggregate_sigs without exponents (bls.py:203-261): a chunk of
// the list per unit (k_sum_chunks: complete mixed additions in registers), then runs of eight partial sums until eight are left
// (k_srt_fold, or one wavefront per run on the wide machine once at most 4096 runs are left), then the last run with the affine
// conversion.  Enqueues on `st` and returns.
template <int DEG>
static int msm_plain(blsgpu_ctx* c, const void* d_pts, size_t n, void* d_out, void* d_out_inf, hipStream_t st) {
    typedef blsgpu::SrtG<DEG> G;
    const size_t units_max = blsgpu::SRT_LANES / G::LP;
    size_t U = (n + 3) / 4;                                   // four points per unit while units are free (a small sum is a latency)
    if (U > units_max) U = units_max;
    const size_t chunk = (n + U - 1) / U;
    U = (n + chunk - 1) / chunk;
    size_t off = 0;
    auto take = [&](size_t words) { size_t o = off; off += (words + 3) & ~(size_t)3; return o; };
    const size_t PJ = G::PJ;
    const size_t o_prep = take(n * blsgpu::L28_AFF * DEG), o_live = take((n + 3) / 4), o_b0 = take(U * PJ), o_b1 = take(((U + 7) / 8) * PJ + PJ);
    if (int rc_ = grow_elems(c, &c->d_buckets, &c->bucket_cap, off)) return rc_;
    uint32_t* W = c->d_buckets;
    uint8_t* live = (uint8_t*)(W + o_live);
    hipLaunchKernelGGL(blsgpu::k_lane_prep<DEG>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_pts, (uint32_t)n, W + o_prep, live);
    const auto blocks = [&](size_t nunits) { return dim3((unsigned)((nunits * G::LP + 63) / 64)); };
    hipLaunchKernelGGL(blsgpu::k_sum_chunks<DEG>, blocks(U), dim3(64), 0, st, W + o_prep, live, (uint32_t)n, (uint32_t)chunk, (uint32_t)U, W + o_b0);
    HIP_TRY(hipGetLastError());
    uint32_t *src = W + o_b0, *dst = W + o_b1;
    size_t cur = U;
    while (cur > 8) {
        const size_t nfold = (cur + 7) / 8;
        if (nfold <= 4096 && cur % 8 == 0) {
            if (DEG == 1) hipLaunchKernelGGL(blsgpu::g1w::k_msm_horner_wide<0>, dim3((unsigned)nfold), dim3(64), 0, st, src, 8u, 0u, dst, (uint8_t*)nullptr);
            else hipLaunchKernelGGL(blsgpu::h2cw::k_msm_horner_wide2<0>, dim3((unsigned)nfold), dim3(64), 0, st, src, 8u, 0u, dst, (uint8_t*)nullptr);
        } else {
            hipLaunchKernelGGL(blsgpu::k_srt_fold<DEG>, blocks(nfold), dim3(64), 0, st, src, (uint32_t)cur, 8u, (uint32_t)nfold, (uint32_t)nfold, dst);
        }
        HIP_TRY(hipGetLastError());
        uint32_t* t = src; src = dst; dst = t;
        cur = nfold;
    }
    if (DEG == 1) hipLaunchKernelGGL(blsgpu::g1w::k_msm_horner_wide<1>, dim3(1), dim3(64), 0, st, src, (uint32_t)cur, 0u, (uint32_t*)d_out, (uint8_t*)d_out_inf);
    else hipLaunchKernelGGL(blsgpu::h2cw::k_msm_horner_wide2<1>, dim3(1), dim3(64), 0, st, src, (uint32_t)cur, 0u, (uint32_t*)d_out, (uint8_t*)d_out_inf);
    HIP_TRY(hipGetLastError());
    return 0;
}

// A batch of scalar multiplications / of small sums with scalars: one group per unit (k_smul).  Enqueues on `st` and returns.
template <int DEG>
static int msm_small_groups(blsgpu_ctx* c, const void* d_pts, const void* d_scalars, size_t k, size_t groups, void* d_out, void* d_out_inf, hipStream_t st) {
    typedef blsgpu::SrtG<DEG> G;
    // slices of groups whose tables (16 projective multiples per point) stay below 2 GB: 2^20 groups of eight G2 points would ask for 45 GB
    const size_t upw = 64 / G::LP, per_group = k * blsgpu::SMUL_T * G::PJ * 4;
    size_t slice = ((size_t)2 << 30) / per_group / upw * upw;
    if (slice < upw) slice = upw;
    if (slice > groups) slice = groups;
    const size_t units0 = (slice + upw - 1) / upw * upw, n0 = k * slice;
    size_t off = 0;
    auto take = [&](size_t words) { size_t o = off; off += (words + 3) & ~(size_t)3; return o; };
    const size_t o_prep = take(n0 * blsgpu::L28_AFF * DEG), o_live = take((n0 + 3) / 4), o_tab = take(units0 * k * blsgpu::SMUL_T * G::PJ);
    if (grow_elems(c, &c->d_buckets, &c->bucket_cap, off)) {
        (void)hipGetLastError();
        return 1;                                              // no room: the caller's other kernels
    }
    uint32_t* W = c->d_buckets;
    uint8_t* live = (uint8_t*)(W + o_live);
    for (size_t lo = 0; lo < groups; lo += slice) {
        const size_t m = groups - lo < slice ? groups - lo : slice, n = k * m, units = (m + upw - 1) / upw * upw;
        hipLaunchKernelGGL(blsgpu::k_lane_prep<DEG>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_pts + lo * k * 24 * DEG, (uint32_t)n,
                           W + o_prep, live);
        hipLaunchKernelGGL(blsgpu::k_smul<DEG>, dim3((unsigned)(units * G::LP / 64)), dim3(64), 0, st, W + o_prep, live, (const uint32_t*)d_scalars + lo * k * 8,
                           (uint32_t)k, (uint32_t)m, W + o_tab, (uint32_t*)d_out + lo * 24 * DEG, d_out_inf ? (uint8_t*)d_out_inf + lo : nullptr);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

template <int DEG>
int msm_dev(blsgpu_ctx* c, const void* d_pts, const void* d_scalars, size_t k, size_t groups, void* d_out,
            void* d_out_inf, hipStream_t st) {
    using C = blsgpu::MsmCfg<DEG>;
    if (groups == 0) return 0;
    StreamGuard sg(c, st);
    if (k == 0) {                                   // empty sums: infinity
        HIP_TRY(hipMemsetAsync(d_out, 0, groups * 96 * DEG, st));
        if (d_out_inf) HIP_TRY(hipMemsetAsync(d_out_inf, 1, groups, st));
        return 0;
    }
    if (k > 0x7FFFFFFFull || groups > 0x7FFFFFFFull || k * groups > 0xFFFFFFF0ull) return fail(-EINVAL, "msm too large");
    if (groups == 1 && !d_scalars && k >= c->msm_plain_threshold) return msm_plain<DEG>(c, d_pts, k, d_out, d_out_inf, st);
    if (d_scalars && groups >= c->smul_min_groups && k <= c->smul_max_k) {
        const int rc_ = msm_small_groups<DEG>(c, d_pts, d_scalars, k, groups, d_out, d_out_inf, st);
        if (rc_ != 1) return rc_;
    }
    if (groups == 1 && d_scalars && k >= (DEG == 1 ? c->msm_sort_threshold : c->msm_sort2_threshold)) {
        const int rc_ = msm_sorted<DEG>(c, d_pts, d_scalars, k, d_out, d_out_inf, st);
        if (rc_ != 1) return rc_;
    }
    if ((groups == 1 && k >= c->pip_threshold) || (groups > 1 && groups <= 65535 && k >= c->pip_group_threshold)) {
        // bucket method: one large sum is cut into about 256 chunks; a batch of sums uses one chunk per group
        using P = blsgpu::PipCfg<DEG>;
        size_t n = k * groups;
        // enough points for one (group, chunk, window) per lane to fill the chip?  (3 waves per SIMD = 3072 chunks)
        const bool lane_path = n >= c->msm_lane_threshold;
        size_t want = lane_path ? 3072 : 256;
        if (const char* e = getenv("BLSGPU_PIP_CHUNKS")) want = (size_t)strtoull(e, nullptr, 10);
        size_t chunk = (groups > 1) ? k : (k + want - 1) / want;
        if (!lane_path && chunk < 64 * (size_t)C::NP && groups == 1) chunk = 64 * (size_t)C::NP;
        if (!lane_path) chunk = ((chunk + C::NP - 1) / C::NP) * C::NP;
        if (chunk == 0) chunk = 1;
        size_t chunks = (k + chunk - 1) / chunk;
        const size_t fold_n = (lane_path && chunks > 96) ? (chunks + 63) / 64 : 0;
        // partials: the VM's form (36 DEG dwords) except between the lane kernel and its fold (L28: 42 DEG); prep: the VM's
        // projective triples (36 DEG) or the lane path's affine L28 points (28 DEG) + live flags
        constexpr size_t PJ28 = blsgpu::L28_PJ * DEG;
        size_t need = (chunks + 1) * groups * blsgpu::PIP_W * PJ28 + n * 36 * DEG + (n + 3) / 4 + 4;
        if (int rc_ = grow_elems(c, &c->d_msm_part, &c->msm_part_cap, need)) return rc_;
        uint32_t* d_win = c->d_msm_part + chunks * groups * blsgpu::PIP_W * PJ28;
        uint32_t* d_prep = d_win + groups * blsgpu::PIP_W * PJ28;
        if (lane_path) {
            // one (group, chunk, window) per lane, buckets in HBM
            uint8_t* d_live = (uint8_t*)(d_prep + n * blsgpu::L28_AFF * DEG);
            hipLaunchKernelGGL(blsgpu::k_lane_prep<DEG>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_pts, (uint32_t)n,
                               d_prep, d_live);
            HIP_TRY(hipGetLastError());
            const size_t lanes = groups * chunks * blsgpu::PIP_W;
            const size_t bneed = lanes * (blsgpu::PIP_NB - 1) * PJ28 + fold_n * groups * blsgpu::PIP_W * 36 * DEG;
            if (int rc_ = grow_elems(c, &c->d_buckets, &c->bucket_cap, bneed)) return rc_;
            // a batch of G2 sums (one chunk each): the window sums stay in the L28 form and the Horner runs one sum per lane quad
            const bool horner_quads = DEG == 2 && c->msm_lane_pairs && chunks == 1 && groups >= c->horner_quads_threshold;
            if (DEG == 2 && c->msm_lane_pairs)
                hipLaunchKernelGGL(blsgpu::k_msm_lane2x, dim3((unsigned)((2 * lanes + 63) / 64)), dim3(64), 0, st, d_prep, d_live,
                                   (const uint32_t*)d_scalars, (uint32_t)k, (uint32_t)chunk, (uint32_t)chunks, (uint32_t)lanes, c->d_buckets,
                                   c->d_msm_part, (fold_n || horner_quads) ? 0u : 1u);
            else
                hipLaunchKernelGGL(blsgpu::k_msm_lane<DEG>, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, d_prep, d_live,
                                   (const uint32_t*)d_scalars, (uint32_t)k, (uint32_t)chunk, (uint32_t)chunks, (uint32_t)lanes, c->d_buckets,
                                   c->d_msm_part, fold_n ? 0u : 1u);
            HIP_TRY(hipGetLastError());
            if (horner_quads) {
                const WaveShape ws = wave_shape(c, (4 * groups + 63) / 64);
                hipLaunchKernelGGL(blsgpu::k_msm_horner_quads, dim3(ws.blocks), dim3(ws.threads), 0, st, c->d_msm_part,
                                   (uint32_t)blsgpu::PIP_W, (uint32_t)blsgpu::PIP_C, (uint32_t)groups, (uint32_t*)d_out, (uint8_t*)d_out_inf);
                HIP_TRY(hipGetLastError());
                return 0;
            }
            const uint32_t* winsrc = c->d_msm_part;
            size_t wchunks = chunks;
            if (fold_n) {                                    // many chunks: fold runs of 64 partials per lane first
                uint32_t* d_fold = c->d_buckets + lanes * (blsgpu::PIP_NB - 1) * PJ28;
                const size_t ftotal = groups * blsgpu::PIP_W * fold_n;
                hipLaunchKernelGGL(blsgpu::k_msm_lane_fold<DEG>, dim3((unsigned)((ftotal + 63) / 64)), dim3(64), 0, st, c->d_msm_part,
                                   (uint32_t)chunks, 64u, (uint32_t)fold_n, (uint32_t)ftotal, d_fold, 1u);
                HIP_TRY(hipGetLastError());
                winsrc = d_fold;
                wchunks = fold_n;
            }
            hipLaunchKernelGGL(blsgpu::k_msm_pip_windows<DEG>, dim3(blsgpu::PIP_W, (unsigned)groups), dim3(64), (size_t)blsgpu::TEAM_BYTES, st,
                               c->tabs, winsrc, (uint32_t)wchunks, d_win);
            HIP_TRY(hipGetLastError());
        } else {
            size_t pblocks = (n + (size_t)MSM_WAVES * C::NP - 1) / ((size_t)MSM_WAVES * C::NP);
            hipLaunchKernelGGL(blsgpu::k_msm_prep<DEG>, dim3((unsigned)pblocks), dim3(MSM_WAVES * 64), (size_t)MSM_WAVES * blsgpu::TEAM_BYTES,
                               st, c->tabs, (const uint32_t*)d_pts, (uint32_t)n, d_prep);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(blsgpu::k_msm_pip<DEG>, dim3((unsigned)chunks, blsgpu::PIP_W, (unsigned)groups), dim3(64), (size_t)P::SLOTS * 48,
                               st, c->tabs, d_prep, (const uint32_t*)d_scalars, (uint32_t)k, (uint32_t)chunk, c->d_msm_part);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(blsgpu::k_msm_pip_windows<DEG>, dim3(blsgpu::PIP_W, (unsigned)groups), dim3(64), (size_t)blsgpu::TEAM_BYTES, st,
                               c->tabs, c->d_msm_part, (uint32_t)chunks, d_win);
            HIP_TRY(hipGetLastError());
        }
        if (DEG == 2 && groups >= c->horner_np_threshold) {  // a batch of G2 sums: BLSVM_HMSM2_NP sums per team
            hipLaunchKernelGGL(blsgpu::k_msm_horner_np, dim3((unsigned)((groups + BLSVM_HMSM2_NP - 1) / BLSVM_HMSM2_NP)), dim3(64),
                               (size_t)BLSVM_HMSM2_SLOTS * 48, st, c->tabs, d_win, (uint32_t)blsgpu::PIP_W, (uint32_t)blsgpu::PIP_C,
                               (uint32_t)groups, (uint32_t*)d_out, (uint8_t*)d_out_inf);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        hipLaunchKernelGGL(blsgpu::k_msm_pip_horner<DEG>, dim3((unsigned)groups), dim3(64), (size_t)blsgpu::TEAM_BYTES, st, c->tabs, d_win,
                           (uint32_t)blsgpu::PIP_W, (uint32_t)blsgpu::PIP_C, (uint32_t*)d_out, (uint8_t*)d_out_inf);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // points per block: whole group when small, else chunks that give >= ~2k blocks
    size_t per_pass = (size_t)MSM_WAVES * C::NP;
    size_t chunk = k;
    if (groups < 1024 && k > 4 * per_pass) {
        size_t want_blocks = 2048 / groups + 1;
        chunk = (k + want_blocks - 1) / want_blocks;
        chunk = ((chunk + per_pass - 1) / per_pass) * per_pass;
        if (chunk > k) chunk = k;
    }
    size_t bpg = (k + chunk - 1) / chunk;
    size_t blocks = bpg * groups;
    size_t need = blocks * 36 * DEG;
    if (int rc_ = grow_elems(c, &c->d_msm_part, &c->msm_part_cap, need)) return rc_;
    size_t lds = (size_t)MSM_WAVES * blsgpu::TEAM_BYTES;
    hipLaunchKernelGGL(blsgpu::k_msm<DEG>, dim3((unsigned)blocks), dim3(MSM_WAVES * 64), lds, st, c->tabs,
                       (const uint32_t*)d_pts, (const uint32_t*)d_scalars, (uint32_t)k, (uint32_t)chunk, (uint32_t)bpg,
                       c->d_msm_part);
    HIP_TRY(hipGetLastError());
    size_t fblocks = (groups + MSM_WAVES - 1) / MSM_WAVES;
    hipLaunchKernelGGL(blsgpu::k_msm_finish<DEG>, dim3((unsigned)fblocks), dim3(MSM_WAVES * 64), lds, st, c->tabs,
                       c->d_msm_part, (uint32_t)bpg, (uint32_t)groups, (uint32_t*)d_out, (uint8_t*)d_out_inf);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int DEG>
int msm_host(blsgpu_ctx* c, const uint8_t* pts, const uint8_t* scalars, size_t k, size_t groups, uint8_t* out,
             uint8_t* out_inf) {
    if (!c || !out) return fail(-EINVAL, "NULL argument");
    size_t n = k * groups;
    if (n && !pts) return fail(-EINVAL, "NULL point buffer");
    HIP_TRY(hipSetDevice(c->device));
    size_t pb = n * 96 * DEG, sb = scalars ? n * 32 : 0, ob = groups * 96 * DEG;
    size_t need = pb + sb + ob + groups + 64;
    if (int rc_ = grow_buffer(c, &c->d_io, &c->io_cap, need)) return rc_;
    char* dp = (char*)c->d_io;
    char* ds = dp + pb;
    char* dout = ds + ((sb + 15) & ~size_t(15));
    char* dinf = dout + ob;
    if (pb) HIP_TRY(hipMemcpyAsync(dp, pts, pb, hipMemcpyHostToDevice, 0));
    if (sb) HIP_TRY(hipMemcpyAsync(ds, scalars, sb, hipMemcpyHostToDevice, 0));
    int rc = msm_dev<DEG>(c, dp, scalars ? ds : nullptr, k, groups, dout, dinf, 0);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, dout, ob, hipMemcpyDeviceToHost));
    if (out_inf) HIP_TRY(hipMemcpy(out_inf, dinf, groups, hipMemcpyDeviceToHost));
    return 0;
}
}  // namespace


#define BLSGPU_EXPORT __attribute__((visibility("default")))

extern "C" {

BLSGPU_EXPORT const char* blsgpu_version(void) { return "blsgpu/1 gfx950 vm-tables " BLSVM_TABLE_HASH; }
BLSGPU_EXPORT const char* blsgpu_last_error(void) { return g_err.c_str(); }

BLSGPU_EXPORT int blsgpu_ctx_create(int device, blsgpu_ctx** out) {
    if (!out) return fail(-EINVAL, "out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(-ENODEV, std::string("no HIP device available (") + hipGetErrorString(e) +
                                 "); blsgpu has no CPU fallback");
    if (device < 0 || device >= count) return fail(-EINVAL, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    blsgpu_ctx* c = new blsgpu_ctx();
    c->device = device;
    c->mp_threshold = default_mp_threshold();
    if (const char* e = getenv("BLSGPU_MILLER_WIDE3_MAX")) c->miller_wide3_max = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_MILLER_WIDE_MAX")) c->miller_wide_max = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_MP3_THRESHOLD")) c->mp3_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_LS_THRESHOLD")) c->ls_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_LS_MIN_GROUP")) c->ls_min_group = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_LS_TEAMS")) c->ls_teams = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_FEXP_TEAM_THRESHOLD")) c->fexp_team_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_FEXP_WIDE")) c->fexp_wide = atoi(e) != 0;
    if (const char* e = getenv("BLSGPU_FEXP_WIDE_MAX_PARTIALS")) c->fexp_wide_max_partials = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_VM_EXACT_LANES")) c->vm_exact_lanes = atoi(e) != 0;
    if (const char* e = getenv("BLSGPU_LS_MERGE_WIDE_MAX")) c->ls_merge_wide_max = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_LS_WIDE_MAX")) c->ls_wide_max = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_LS_QUAD_MAX")) c->ls_quad_max = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_PIP_THRESHOLD")) c->pip_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_PIP_GROUP_THRESHOLD")) c->pip_group_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_POW2_MAX")) c->pow2_max = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_MSM_WIDE_TAIL")) c->msm_wide_tail = atoi(e) != 0;
    if (const char* e = getenv("BLSGPU_H2C_WIDE_MAX")) c->h2c_wide_max = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_H2C_REG_THRESHOLD")) c->h2c_reg_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_H2C_QUAD_MAX")) c->h2c_quad_max = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_TEST_LS_NOMEM")) c->test_ls_nomem = atoi(e) != 0;
    if (const char* e = getenv("BLSGPU_H2C_JACOBI")) c->h2c_jacobi = atoi(e) != 0;
    if (const char* e = getenv("BLSGPU_H2C_JACOBI_THRESHOLD")) c->h2c_jacobi_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_H2C_LANE_THRESHOLD")) c->h2c_lane_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_MSM_SORT_THRESHOLD")) c->msm_sort_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_MSM_SORT2_THRESHOLD")) c->msm_sort2_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_MILLER_EXACT_LANES")) c->miller_exact_lanes = atoi(e) != 0;
    if (const char* e = getenv("BLSGPU_MILLER_EXACT_FAST")) c->miller_exact_fast = atoi(e) != 0;
    if (const char* e = getenv("BLSGPU_MSM_PLAIN_THRESHOLD")) c->msm_plain_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_SMUL_MIN_GROUPS")) c->smul_min_groups = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_SMUL_MAX_K")) c->smul_max_k = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_HORNER_NP_THRESHOLD")) c->horner_np_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_HORNER_QUADS_THRESHOLD")) c->horner_quads_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_WG256_MAX_WAVES")) c->wg256_max_waves = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_MSM_LANE_THRESHOLD")) c->msm_lane_threshold = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("BLSGPU_MSM_LANE_PAIRS")) c->msm_lane_pairs = atoi(e) != 0;
    // pack all tables into one device allocation (16-byte aligned pieces)
    auto al = [](size_t x) { return (x + 15) & ~size_t(15); };
    size_t o_m = 0;
    size_t o_mp = o_m + al(sizeof(BLSVM_MILLER_FLAT));
    size_t o_mp2 = o_mp + al(sizeof(BLSVM_MP_FLAT));
    size_t o_h2 = o_mp2 + al(sizeof(BLSVM_MP2_FLAT));
    size_t o_f = o_h2 + al(sizeof(BLSVM_H2_FLAT));
    size_t o_sl = o_f + al(sizeof(BLSVM_FEXP_FLAT));
    size_t o_s = o_sl + al(sizeof(BLSVM_SLOW_FLAT));
    size_t o_data = o_s + al(sizeof(BLSVM_SEG_FLAT));
    size_t o_c = o_data + al(sizeof(BLSVM_DATA));
    size_t total = o_c + al(sizeof(BLSVM_CONSTS));
    if (hipMalloc(&c->d_tables, total) != hipSuccess) {
        delete c;
        return fail(-ENOMEM, "hipMalloc(tables) failed");
    }
    char* base = (char*)c->d_tables;
    struct { size_t off; const void* src; size_t len; } parts[] = {
        {o_m, BLSVM_MILLER_FLAT, sizeof(BLSVM_MILLER_FLAT)}, {o_mp, BLSVM_MP_FLAT, sizeof(BLSVM_MP_FLAT)},
        {o_mp2, BLSVM_MP2_FLAT, sizeof(BLSVM_MP2_FLAT)},
        {o_h2, BLSVM_H2_FLAT, sizeof(BLSVM_H2_FLAT)},
        {o_f, BLSVM_FEXP_FLAT, sizeof(BLSVM_FEXP_FLAT)},
        {o_sl, BLSVM_SLOW_FLAT, sizeof(BLSVM_SLOW_FLAT)},
        {o_s, BLSVM_SEG_FLAT, sizeof(BLSVM_SEG_FLAT)},       {o_data, BLSVM_DATA, sizeof(BLSVM_DATA)},
        {o_c, BLSVM_CONSTS, sizeof(BLSVM_CONSTS)}};
    for (auto& p : parts) {
        if (hipMemcpy(base + p.off, p.src, p.len, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(c->d_tables);
            delete c;
            return fail(-EIO, "hipMemcpy(tables) failed");
        }
    }
    c->tabs.mflat = (const uint2*)(base + o_m);
    c->tabs.mpflat = (const uint2*)(base + o_mp);
    c->tabs.mp2flat = (const uint2*)(base + o_mp2);
    c->tabs.h2flat = (const uint2*)(base + o_h2);
    c->tabs.fflat = (const uint2*)(base + o_f);
    c->tabs.sflat = (const uint2*)(base + o_sl);
    c->tabs.segflat = (const uint2*)(base + o_s);
    c->tabs.data = (const uint16_t*)(base + o_data);
    c->tabs.consts = (const uint32_t*)(base + o_c);
    c->tabs.stamps = nullptr;
#ifdef BLSGPU_STAMPS
    {
        void* p = nullptr;
        if (hipMalloc(&p, 128) == hipSuccess) { (void)hipMemset(p, 0, 128); c->tabs.stamps = (unsigned long long*)p; }
    }
#endif
    if (hipMalloc((void**)&c->d_out, BLSGPU_FQ12_BYTES) != hipSuccess) {
        (void)hipFree(c->d_tables);
        delete c;
        return fail(-ENOMEM, "hipMalloc(out) failed");
    }
    // the kernels need more than the default 64 KiB of dynamic LDS
    (void)hipFuncSetAttribute((const void*)blsgpu::k_miller, hipFuncAttributeMaxDynamicSharedMemorySize,
                              MILLER_WAVES * blsgpu::TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_reduce, hipFuncAttributeMaxDynamicSharedMemorySize,
                              REDUCE_WAVES * blsgpu::TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_h2c_stage<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, blsgpu::H1_TEAM_DW * 4);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_h2c_stage<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, blsgpu::H1_TEAM_DW * 4);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_h2c_stage<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, blsgpu::H1_TEAM_DW * 4);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_h2c_stage<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, blsgpu::H1_TEAM_DW * 4);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_h2c_clear, hipFuncAttributeMaxDynamicSharedMemorySize,
                              blsgpu::H2_TEAM_DW * 4);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_decompress<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, BLSVM_D1_SLOTS * 48);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_decompress<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, BLSVM_D1_SLOTS * 48);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_decompress<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, BLSVM_D2_SLOTS * 48);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_decompress<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, BLSVM_D2_SLOTS * 48);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_decompress<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, BLSVM_D2_SLOTS * 48);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_miller_mp<BLSVM_MP_G>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              blsgpu::MP_TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_miller_mp<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              blsgpu::MP_TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_final_groups, hipFuncAttributeMaxDynamicSharedMemorySize,
                              REDUCE_WAVES * blsgpu::TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_bytes_to_partials, hipFuncAttributeMaxDynamicSharedMemorySize,
                              REDUCE_WAVES * blsgpu::TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_msm_prep<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * blsgpu::TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_msm_prep<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * blsgpu::TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_msm_pip<1>, hipFuncAttributeMaxDynamicSharedMemorySize, blsgpu::PipCfg<1>::SLOTS * 48);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_msm_pip<2>, hipFuncAttributeMaxDynamicSharedMemorySize, blsgpu::PipCfg<2>::SLOTS * 48);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_msm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * blsgpu::TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_msm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * blsgpu::TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_msm_finish<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * blsgpu::TEAM_BYTES);
    (void)hipFuncSetAttribute((const void*)blsgpu::k_msm_finish<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * blsgpu::TEAM_BYTES);
    int rc = ensure_workspace(c, 4096);
    if (!rc && hipEventCreateWithFlags(&c->last_event, hipEventDisableTiming) != hipSuccess) rc = fail(-EIO, "hipEventCreate failed");
    if (rc) {
        blsgpu_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return 0;
}

BLSGPU_EXPORT void blsgpu_ctx_destroy(blsgpu_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_tables) (void)hipFree(c->d_tables);
    for (int i = 0; i < 2; i++)
        if (c->d_part[i]) (void)hipFree(c->d_part[i]);
    if (c->d_io) (void)hipFree(c->d_io);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_msm_part) (void)hipFree(c->d_msm_part);
    if (c->d_buckets) (void)hipFree(c->d_buckets);
    if (c->d_degen) (void)hipFree(c->d_degen);
    if (c->d_lines) (void)hipFree(c->d_lines);
    for (int i = 0; i < 2; i++)
        if (c->d_lsp[i]) (void)hipFree(c->d_lsp[i]);
    if (c->d_bad) (void)hipFree(c->d_bad);
    if (c->d_exflags) (void)hipFree(c->d_exflags);
    if (c->d_fexp_ws) (void)hipFree(c->d_fexp_ws);
    if (c->d_h2c_ws) (void)hipFree(c->d_h2c_ws);
    for (void* q : c->retired) (void)hipFree(q);
    if (c->last_event) (void)hipEventDestroy(c->last_event);
    if (c->ev0) {
        for (int i = 0; i < blsgpu_ctx::TIMING_SLOTS; i++) { (void)hipEventDestroy(c->ev0[i]); (void)hipEventDestroy(c->ev1[i]); }
        delete[] c->ev0; delete[] c->ev1; delete[] c->ev_kind;
    }
    delete c;
}

BLSGPU_EXPORT int blsgpu_timing_enable(blsgpu_ctx* c, int enable) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (enable && !c->ev0) {
        c->ev0 = new hipEvent_t[blsgpu_ctx::TIMING_SLOTS];
        c->ev1 = new hipEvent_t[blsgpu_ctx::TIMING_SLOTS];
        c->ev_kind = new int[blsgpu_ctx::TIMING_SLOTS];
        for (int i = 0; i < blsgpu_ctx::TIMING_SLOTS; i++) { HIP_TRY(hipEventCreate(&c->ev0[i])); HIP_TRY(hipEventCreate(&c->ev1[i])); }
    }
    c->timing = enable != 0;
    c->ev_count = 0;
    return 0;
}

BLSGPU_EXPORT int blsgpu_timing_read(blsgpu_ctx* c, float* ms, int* kind, size_t cap, size_t* count) {
    if (!c || !count) return fail(-EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    size_t n = c->ev_count < cap ? c->ev_count : cap;
    for (size_t i = 0; i < n; i++) {
        HIP_TRY(hipEventSynchronize(c->ev1[i]));
        HIP_TRY(hipEventElapsedTime(&ms[i], c->ev0[i], c->ev1[i]));
        kind[i] = c->ev_kind[i];
    }
    *count = n;
    c->ev_count = 0;
    return 0;
}

// The chip's v_mad_i64_i32 rate right now: a probe kernel of about `target_ms` milliseconds (2048 workgroups x 256 threads, eight
// independent multiply-add chains per lane), timed with HIP events on `stream`; *tmacs = 10^12 multiply-adds per second.
BLSGPU_EXPORT int blsgpu_timing_mad_probe(blsgpu_ctx* c, double target_ms, double* tmacs, void* stream) {
    if (!c || !tmacs) return fail(-EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    constexpr unsigned BLOCKS = 2048, THREADS = 256;
    if (int rc = grow_buffer(c, &c->d_io, &c->io_cap, (size_t)BLOCKS * THREADS * 4)) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    double rate = 0.0;
    uint32_t iters = 20000;                                   // ~2.4 ms at 34 T/s: calibrates the second launch
    for (int pass = 0; pass < 2; pass++) {
        HIP_TRY(hipEventRecord(e0, st));
        hipLaunchKernelGGL(blsgpu::probe::k_mad_probe, dim3(BLOCKS), dim3(THREADS), 0, st, (uint32_t*)c->d_io, iters, (uint32_t)pass);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e1, st));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        rate = (double)BLOCKS * THREADS * 8.0 * iters / (ms * 1e-3) / 1e12;
        if (pass == 0 && ms > 0.f) {
            double want = (double)iters * target_ms / ms;
            iters = want < 1000.0 ? 1000u : (want > 4e6 ? 4000000u : (uint32_t)want);
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *tmacs = rate;
    return 0;
}

// One dispatch of an empty kernel (blsgpu::probe::k_mark) on `stream`: a caller brackets its timed region with two of them so that
// a profile of the run can be cut to that region (the counters of rocprofv3 --pmc are per dispatch).
BLSGPU_EXPORT int blsgpu_timing_mark(blsgpu_ctx* c, unsigned tag, void* stream) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(blsgpu::probe::k_mark, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint32_t)tag);
    HIP_TRY(hipGetLastError());
    return 0;
}

BLSGPU_EXPORT int blsgpu_ctx_set_mp_threshold(blsgpu_ctx* c, size_t pairs) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    c->mp_threshold = pairs;
    return 0;
}
// Calls of at least `pairs` pairs whose groups all have at least `min_group` pairs run the line-stream kernels
// (blsgpu_ml.hip); (size_t)-1 for `pairs` keeps every call on the wavefront-VM kernels.
BLSGPU_EXPORT int blsgpu_ctx_set_ls_threshold(blsgpu_ctx* c, size_t pairs, size_t min_group) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    c->ls_threshold = pairs;
    c->ls_min_group = min_group ? min_group : 1;
    return 0;
}
// The caller's event (or NULL: none) is recorded on the call's stream right after the last kernel of a Miller stage that
// fills the chip; what follows (Horner, the product of the partials, the final exponentiation) occupies a few dozen
// wavefronts.  A server that pipelines calls over several contexts lets the next call's stream wait for this event
// instead of the end of the call.
// Calls with at least `results` final exponentiations run them six lanes per result on the register arithmetic
// (blsgpu_fexp.hip); fewer keep one wavefront each on the VM (lower latency).  (size_t)-1: never.
BLSGPU_EXPORT int blsgpu_ctx_set_fexp_team_threshold(blsgpu_ctx* c, size_t results) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    c->fexp_team_threshold = results;
    return 0;
}
// Diagnostic (tools/exact_trace.py): copies the first `bytes` bytes of the line records of the last line-stream call.
BLSGPU_EXPORT int blsgpu_debug_read_lines(blsgpu_ctx* c, void* host_buf, size_t bytes) {
    if (!c || !host_buf) return fail(-EINVAL, "NULL argument");
    if (bytes > c->lines_cap) return fail(-EINVAL, "more than the buffer holds");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_buf, c->d_lines, bytes, hipMemcpyDeviceToHost));
    return 0;
}
// Diagnostic (tools/fexp_trace.py): device buffer of BLS28_FEXP_NOPS x 576 bytes that receives the accumulator of
// result 0 after every operation of the batched final exponentiation's script, or NULL.
BLSGPU_EXPORT int blsgpu_ctx_set_fexp_trace(blsgpu_ctx* c, void* d_buf) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    c->d_fexp_dbg = d_buf;
    return 0;
}
// Diagnostic (tools/fexpw_stamps.py): device buffer of (BLS28_FEXP_NOPS + 1) x 8 bytes that receives the cycle counter of result 0
// before the one-result-per-wavefront final exponentiation's script and after every operation of it, or NULL.
BLSGPU_EXPORT int blsgpu_ctx_set_fexpw_stamps(blsgpu_ctx* c, void* d_buf) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    c->d_fexpw_stamps = d_buf;
    return 0;
}
BLSGPU_EXPORT int blsgpu_ctx_set_bulk_event(blsgpu_ctx* c, void* event) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    c->bulk_event = (hipEvent_t)event;
    return 0;
}
// accumulators (six lanes each) the line-stream product kernel aims at; decides the chunk of pairs per accumulator
BLSGPU_EXPORT int blsgpu_ctx_set_ls_teams(blsgpu_ctx* c, size_t teams) {
    if (!c || teams == 0) return fail(-EINVAL, "bad argument");
    c->ls_teams = teams;
    return 0;
}
// Calls of at most `pairs` pairs (that do not take the line-stream kernels) run the wide Miller loop; 0: never.
BLSGPU_EXPORT int blsgpu_ctx_set_miller_wide_max(blsgpu_ctx* c, size_t pairs) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    c->miller_wide_max = pairs;
    return 0;
}
BLSGPU_EXPORT int blsgpu_ctx_set_mp3_threshold(blsgpu_ctx* c, size_t pairs) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    c->mp3_threshold = pairs;
    return 0;
}

BLSGPU_EXPORT int blsgpu_ctx_reserve(blsgpu_ctx* c, size_t max_pairs) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = ensure_workspace(c, max_pairs)) return rc;
    if (max_pairs >= c->ls_threshold) {
        // the line-stream stage's buffers as well (a call of that size takes them): line records of one slice, the
        // flags and the work list, and dense partial products for the usual chunking (ls_teams accumulators plus 68 per
        // group of at least ls_min_group pairs); a call that needs more grows them itself
        const size_t n = max_pairs < LS_MAX_PAIRS ? max_pairs : LS_MAX_PAIRS;
        const size_t teams = c->ls_teams + blsgpu::ml::LINES * (n / (c->ls_min_group ? c->ls_min_group : 1) + 1);
        if (grow_buffer(c, &c->d_lines, &c->lines_cap, n * blsgpu::ml::LINES * blsgpu::ml::LINE_DW * 4) ||
            grow_buffer(c, &c->d_bad, &c->bad_cap, n) || grow_elems(c, &c->d_degen, &c->degen_cap, n + 2) ||
            grow_buffer(c, &c->d_lsp[0], &c->lsp_cap[0], teams * blsgpu::ml::DENSE_DW * 4) ||
            grow_buffer(c, &c->d_lsp[1], &c->lsp_cap[1], (teams / 8 + blsgpu::ml::LINES) * blsgpu::ml::DENSE_DW * 4)) {
            (void)hipGetLastError();
            return fail(-ENOMEM, "no memory for the line-stream workspace (calls of that size will use the wavefront-VM kernels)");
        }
    }
    return 0;
}

// Bytes of HBM the context holds, by purpose (grow-only buffers: the high-water mark of the calls made so far).
BLSGPU_EXPORT int blsgpu_ctx_workspace_bytes(blsgpu_ctx* c, size_t out[BLSGPU_WS_FIELDS]) {
    if (!c || !out) return fail(-EINVAL, "NULL argument");
    out[BLSGPU_WS_PARTIALS] = 2 * c->part_cap * 576;
    out[BLSGPU_WS_STAGING] = c->io_cap;
    out[BLSGPU_WS_LINES] = c->lines_cap;
    out[BLSGPU_WS_LINE_PRODUCTS] = c->lsp_cap[0] + c->lsp_cap[1];
    out[BLSGPU_WS_FLAGS_AND_LISTS] = c->bad_cap + c->degen_cap * 4;
    out[BLSGPU_WS_GROUP_SUMS] = c->msm_part_cap * 4 + c->bucket_cap * 4;
    out[BLSGPU_WS_SLOTS] = c->fexp_ws_cap + c->h2c_ws_cap;
    size_t total = 0;
    for (int i = 0; i < BLSGPU_WS_TOTAL; i++) total += out[i];
    out[BLSGPU_WS_TOTAL] = total;
    return 0;
}

// Waits for the context's enqueued work and releases the buffers that larger ones replaced.
BLSGPU_EXPORT int blsgpu_ctx_trim(blsgpu_ctx* c) {
    if (!c) return fail(-EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (c->used && c->last_event) HIP_TRY(hipEventSynchronize(c->last_event));
    for (void* q : c->retired) (void)hipFree(q);
    c->retired.clear();
    return 0;
}

// `groups` results at once: product of the m partials of each group (partial i of group g at
// d_in[(i * istride + g * gstride) * 144]) and its final exponentiation, six lanes per result (blsgpu_fexp.hip).
static bool use_fexp_team(const blsgpu_ctx* c, size_t m, size_t groups) { return groups >= c->fexp_team_threshold && m <= 64; }
static int launch_fexp_team(blsgpu_ctx* c, const uint32_t* d_in, size_t m, size_t istride, size_t gstride, size_t groups, void* d_out_bytes,
                            hipStream_t st) {
    using namespace blsgpu;
    const WaveShape ws = wave_shape(c, (groups + ml::TEAMS - 1) / ml::TEAMS);
    const size_t waves = (size_t)ws.blocks * (ws.threads / 64);                 // every launched wavefront owns rows of the workspace
    if (int rc = grow_buffer(c, &c->d_fexp_ws, &c->fexp_ws_cap, waves * (ml::TEAMS + 1) * BLS28_FEXP_NSLOTS * ml::DENSE_DW * 4)) return rc;
    KernelTimer kt(c, st, 2);
    hipLaunchKernelGGL(fx::k_fexp_team, dim3(ws.blocks), dim3(ws.threads), 0, st, d_in, (uint32_t)m, (uint32_t)istride, (uint32_t)gstride,
                       (uint32_t)groups, (int32_t*)c->d_fexp_ws, (uint32_t*)d_out_bytes, (uint32_t*)c->d_fexp_dbg);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Fewer results than the team form wants: one result per wavefront (blsgpu_fexpw.hip), the latency form.
static bool use_fexp_wide(const blsgpu_ctx* c, size_t m, size_t groups) {
    return c->fexp_wide && groups < c->fexp_team_threshold && m >= 1 && m <= c->fexp_wide_max_partials && groups <= 0x7FFFFFFFull;
}
static int launch_fexp_wide(blsgpu_ctx* c, const uint32_t* d_in, size_t m, size_t istride, size_t gstride, size_t groups, void* d_out_bytes,
                            hipStream_t st) {
    KernelTimer kt(c, st, 2);
    hipLaunchKernelGGL(blsgpu::fxw::k_fexp_wide, dim3((unsigned)groups), dim3(64), 0, st, d_in, (uint32_t)m, (uint32_t)istride,
                       (uint32_t)gstride, (uint32_t*)d_out_bytes, (unsigned long long*)c->d_fexpw_stamps);
    HIP_TRY(hipGetLastError());
    return 0;
}
// the register forms of the final exponentiation: batches six lanes per result, a few results one wavefront each
static bool use_fexp_reg(const blsgpu_ctx* c, size_t m, size_t groups) { return use_fexp_team(c, m, groups) || use_fexp_wide(c, m, groups); }
static int launch_fexp_reg(blsgpu_ctx* c, const uint32_t* d_in, size_t m, size_t istride, size_t gstride, size_t groups, void* d_out_bytes,
                           hipStream_t st) {
    return use_fexp_team(c, m, groups) ? launch_fexp_team(c, d_in, m, istride, gstride, groups, d_out_bytes, st)
                                       : launch_fexp_wide(c, d_in, m, istride, gstride, groups, d_out_bytes, st);
}

// For each of `groups` groups fold its m partials down to one; the last launch
// optionally applies the final exponentiation and writes 576 bytes per group to
// d_out_bytes, otherwise one partial per group to d_out_partial.  Partial i of
// group g is read from d_in[(i * istride + g * gstride) * 144].
static int reduce_chain(blsgpu_ctx* c, const uint32_t* d_in, size_t m, size_t groups, size_t istride, size_t gstride,
                        bool do_final, uint32_t* d_out_partial, void* d_out_bytes, hipStream_t st) {
    const uint32_t* src = d_in;
    int pp = (d_in == c->d_part[0]) ? 1 : 0;
    size_t lds = (size_t)REDUCE_WAVES * blsgpu::TEAM_BYTES;
    if (groups > 65535) return fail(-EINVAL, "too many groups");
    while (true) {
        // the final exponentiation in registers as soon as few enough partials per group are left
        if (do_final && use_fexp_reg(c, m, groups)) return launch_fexp_reg(c, src, m, istride, gstride, groups, d_out_bytes, st);
        size_t blocks = (m + REDUCE_PER_BLOCK - 1) / REDUCE_PER_BLOCK;
        if (blocks == 0) blocks = 1;
        bool last = blocks == 1;
        // the level that leaves one partial per group hands it to the register forms (one more launch, but the VM's
        // final exponentiation inside k_reduce is 1.25 ms of one wavefront)
        const bool hand_over = last && do_final && use_fexp_reg(c, 1, groups) && groups <= c->part_cap;
        uint32_t* dst = (last && !hand_over) ? d_out_partial : c->d_part[pp];
        if (!last && blocks * groups > c->part_cap) return fail(-ENOMEM, "workspace too small; call blsgpu_ctx_reserve");
        {
            KernelTimer kt(c, st, (last && do_final && !hand_over) ? 2 : 1);
            hipLaunchKernelGGL(blsgpu::k_reduce, dim3((unsigned)blocks, (unsigned)groups), dim3(REDUCE_WAVES * 64), lds, st, c->tabs,
                               src, (uint32_t)m, (uint32_t)REDUCE_PER_BLOCK, (uint32_t)istride, (uint32_t)gstride, dst,
                               (uint32_t)(last && do_final && !hand_over ? 1 : 0), (uint32_t*)d_out_bytes);
        }
        HIP_TRY(hipGetLastError());
        if (hand_over) return launch_fexp_reg(c, dst, 1, 1, 1, groups, d_out_bytes, st);
        if (last) break;
        src = dst;
        m = blocks;
        istride = 1;
        gstride = blocks;
        pp ^= 1;
    }
    return 0;
}

// Miller loops of `groups` runs of gsz pairs; returns the partials per group (bpg).
// Then k_miller_slow: it rewrites the partials of the blocks that met a degenerate pair with the
// reference-faithful program (normally none: every wavefront leaves at once).
constexpr unsigned SLOW_GRID = 3072;           // three wavefronts per SIMD (166 VGPRs, 10 KB of LDS each)
// team: 0 = choose by batch size; 2 / 3 = k_miller_mp with that many pairs per wavefront (groups of two or three pairs: one team
// per group, the group's product comes out of the Miller kernel)
static int launch_miller(blsgpu_ctx* c, const void* d_g1, const void* d_g2, const void* d_inf, size_t gsz, size_t groups, bool one_per_block,
                         uint32_t* d_partials, hipStream_t st, size_t* bpg_out, int team = 0) {
    // a few pairs: one pair per two-wavefront workgroup with a product per lane (blsgpu_mlw.hip), every pair its own partial
    // (where the one-pair-per-wavefront k_miller ran: calls below the throughput kernels' threshold)
    const bool wide = !team && gsz * groups <= c->miller_wide_max && !use_mp(c, gsz * groups);
    const bool mp = team ? true : (!wide && !one_per_block && use_mp(c, gsz * groups));
    const bool mp2 = team ? team == 2 : (mp && use_mp2(c, gsz * groups));   // a few thousand pairs: teams of two fill the chip
    const size_t per_block = (one_per_block || wide) ? 1 : (mp ? (mp2 ? (size_t)2 : (size_t)BLSVM_MP_G) : (size_t)MILLER_WAVES);
    size_t bpg = (gsz + per_block - 1) / per_block;
    *bpg_out = bpg;
    if (bpg * groups > 0x7FFFFFFFull) return fail(-EINVAL, "batch too large");
    if (bpg * groups + 2 > c->degen_cap) return fail(-ENOMEM, "work list too small");
    blsgpu::DegenList dg{c->d_degen, c->d_degen + 1, (const uint8_t*)d_inf};
    HIP_TRY(hipMemsetAsync(c->d_degen, 0, sizeof(uint32_t), st));
    if (wide) {
        KernelTimer kt(c, st, 0);
        if (gsz * groups <= c->miller_wide3_max)
            hipLaunchKernelGGL(blsgpu::mlw::k_miller_wide<3>, dim3((unsigned)(bpg * groups)), dim3(192), 0, st, (const uint32_t*)d_g1,
                               (const uint32_t*)d_g2, (uint32_t)(gsz * groups), d_partials, dg);
        else
            hipLaunchKernelGGL(blsgpu::mlw::k_miller_wide<2>, dim3((unsigned)((bpg * groups + 1) / 2)), dim3(256), 0, st, (const uint32_t*)d_g1,
                               (const uint32_t*)d_g2, (uint32_t)(gsz * groups), d_partials, dg);
    } else if (mp2) {
        KernelTimer kt(c, st, 0);
        hipLaunchKernelGGL(blsgpu::k_miller_mp<2>, dim3((unsigned)(bpg * groups)), dim3(64), (size_t)blsgpu::MP_TEAM_BYTES, st, c->tabs,
                           (const uint32_t*)d_g1, (const uint32_t*)d_g2, (uint32_t)gsz, (uint32_t)bpg, d_partials, dg);
    } else if (mp) {
        KernelTimer kt(c, st, 0);
        hipLaunchKernelGGL(blsgpu::k_miller_mp<BLSVM_MP_G>, dim3((unsigned)(bpg * groups)), dim3(64), (size_t)blsgpu::MP_TEAM_BYTES, st,
                           c->tabs, (const uint32_t*)d_g1, (const uint32_t*)d_g2, (uint32_t)gsz, (uint32_t)bpg, d_partials, dg);
    } else {
        size_t lds = per_block * blsgpu::TEAM_BYTES;
        KernelTimer kt(c, st, 0);
        hipLaunchKernelGGL(blsgpu::k_miller, dim3((unsigned)(bpg * groups)), dim3((unsigned)per_block * 64), lds, st, c->tabs,
                           (const uint32_t*)d_g1, (const uint32_t*)d_g2, (uint32_t)gsz, (uint32_t)bpg, d_partials, dg);
    }
    HIP_TRY(hipGetLastError());
    if (c->bulk_event) HIP_TRY(hipEventRecord(c->bulk_event, st));
    // The listed blocks once more, exactly: the reference's own line values of their pairs on lane pairs
    // (k_ml_lines_exact, block mode) and one six-lane accumulator per block over them (k_ml_small, list mode) -- 1.5 ms
    // where the VM's slow program (k_miller_slow: one pair at a time per wavefront, 68 lane-serial inversions each)
    // takes 6 ms per PAIR.  Both kernels leave at once when the list is empty.
    const size_t nv = bpg * groups * per_block;            // virtual pairs: every block could be listed
    if (c->vm_exact_lanes && nv <= ((size_t)1 << 16) &&
        !grow_buffer(c, &c->d_lines, &c->lines_cap, nv * blsgpu::ml::LINES * blsgpu::ml::LINE_DW * 4) && !grow_buffer(c, &c->d_bad, &c->bad_cap, nv)) {
        KernelTimer kt(c, st, 3);
        hipLaunchKernelGGL(blsgpu::ml::k_ml_lines_exact, dim3(1024), dim3(64), 0, st, (const uint32_t*)d_g1, (const uint32_t*)d_g2, (uint32_t)nv,
                           (int32_t*)c->d_lines, (uint8_t*)c->d_bad, dg, (uint32_t)gsz, (uint32_t)bpg, (uint32_t)per_block);
        hipLaunchKernelGGL(blsgpu::ml::k_ml_small, dim3((unsigned)((bpg * groups + blsgpu::ml::TEAMS - 1) / blsgpu::ml::TEAMS)), dim3(64), 0, st,
                           (const int32_t*)c->d_lines, (const uint8_t*)c->d_bad, (uint32_t)nv, (uint32_t)per_block, (uint32_t)(bpg * groups),
                           d_partials, 144u, (const uint32_t*)dg.count, (const uint32_t*)dg.blocks);
    } else {
        (void)hipGetLastError();
        KernelTimer kt(c, st, 3);
        hipLaunchKernelGGL(blsgpu::k_miller_slow, dim3(SLOW_GRID), dim3(64), (size_t)blsgpu::SLOW_TEAM_BYTES, st, c->tabs,
                           (const uint32_t*)d_g1, (const uint32_t*)d_g2, (uint32_t)gsz, (uint32_t)bpg, (uint32_t)per_block, d_partials, dg);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// The line-stream form of launch_miller (blsgpu_ml.hip): ONE partial per group comes out (bpg = 1).
static bool use_ls(const blsgpu_ctx* c, size_t gsz, size_t groups) {
    return gsz >= 1 && gsz * groups >= c->ls_threshold && gsz * groups <= 0x3FFFFFF0ull;
}
// d_fused_out (optional): the caller wants nothing but the final exponentiation of each group's product -- the kernel
// that ends the stage (k_ml_horner_fexp) then goes on to it in place: no partial, no further launch; *fused tells.
static int launch_miller_ls(blsgpu_ctx* c, const void* d_g1, const void* d_g2, const void* d_inf, size_t gsz, size_t groups,
                            uint32_t* d_partials, hipStream_t st, void* d_fused_out = nullptr, bool* fused = nullptr) {
    using namespace blsgpu;
    const size_t n = gsz * groups;
    // chunks: equal runs of a group's pairs, one accumulator each per line index, sized so that about ls_teams
    // accumulators exist (a wavefront of ten then runs a few hundred products: many short wavefronts per SIMD, so the
    // last round of the launch costs little) but never fewer than 16 pairs (the merge is a dense product per chunk)
    // ... and fewer, longer chunks for mid-size calls (at least 128 pairs each while 40 960 accumulators -- two wavefronts per
    // SIMD -- remain): the merge tree over the chunks is latency, 65 536 pairs 7.05 -> 6.8 ms (tools/c3_probe.py)
    size_t aim = n * ml::LINES / 128;
    if (aim < 40960) aim = 40960;
    if (aim > c->ls_teams) aim = c->ls_teams;
    size_t want = (n * ml::LINES + aim - 1) / aim;
    if (want < 16) want = 16;
    if (want > gsz) want = gsz;
    size_t cpg = (gsz + want - 1) / want;
    const size_t chunk = (gsz + cpg - 1) / cpg;
    cpg = (gsz + chunk - 1) / chunk;
    constexpr size_t FAN = 8;
    const bool small = gsz < c->ls_min_group;              // one accumulator per group runs the whole loop (k_ml_small)
    if (c->test_ls_nomem ||                                // tests/test_gpu_alternate_forms.py: the fallback below, without exhausting a GPU
        grow_buffer(c, &c->d_lines, &c->lines_cap, n * ml::LINES * ml::LINE_DW * 4) ||
        (!small && grow_buffer(c, &c->d_lsp[0], &c->lsp_cap[0], groups * cpg * ml::LINES * ml::DENSE_DW * 4)) ||
        (!small && grow_buffer(c, &c->d_lsp[1], &c->lsp_cap[1], groups * ((cpg + FAN - 1) / FAN) * ml::LINES * ml::DENSE_DW * 4)) ||
        grow_buffer(c, &c->d_bad, &c->bad_cap, n) ||
        grow_elems(c, &c->d_degen, &c->degen_cap, n + 2)) {
        (void)hipGetLastError();
        return -ENOMEM;                                    // the caller falls back to the wavefront-VM kernels
    }
    DegenList dg{c->d_degen, c->d_degen + 1, (const uint8_t*)d_inf};
    HIP_TRY(hipMemsetAsync(c->d_degen, 0, sizeof(uint32_t), st));
    {
        KernelTimer kt(c, st, 4);
        if (n <= c->ls_wide_max) {                       // a few thousand pairs: sixteen lanes each, the values in LDS, a product per lane
            hipLaunchKernelGGL(lsw::k_ml_lines_wide, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, (const uint32_t*)d_g1,
                               (const uint32_t*)d_g2, (uint32_t)n, (int32_t*)c->d_lines, (uint8_t*)c->d_bad, dg);
        } else if (n <= c->ls_quad_max) {                // few pairs: four lanes each, the tangent step's levels shared by the two pairs
            const WaveShape ws = wave_shape(c, (4 * n + 63) / 64);
            hipLaunchKernelGGL(ml::k_ml_lines4, dim3(ws.blocks), dim3(ws.threads), 0, st, (const uint32_t*)d_g1,
                               (const uint32_t*)d_g2, (uint32_t)n, (int32_t*)c->d_lines, (uint8_t*)c->d_bad, dg);
        } else {
            const WaveShape ws = wave_shape(c, (2 * n + 63) / 64);
            hipLaunchKernelGGL(ml::k_ml_lines2, dim3(ws.blocks), dim3(ws.threads), 0, st, (const uint32_t*)d_g1,
                               (const uint32_t*)d_g2, (uint32_t)n, (int32_t*)c->d_lines, (uint8_t*)c->d_bad, dg);
        }
    }
    HIP_TRY(hipGetLastError());
    {   // the listed pairs once more with the reference's own formulas: their line records are rewritten (leaves at once
        // when the list is empty)
        KernelTimer kt(c, st, 3);
        hipLaunchKernelGGL(ml::k_ml_lines_exact, dim3(2048), dim3(64), 0, st, (const uint32_t*)d_g1, (const uint32_t*)d_g2, (uint32_t)n,
                           (int32_t*)c->d_lines, (uint8_t*)c->d_bad, dg, 0u, 0u, 0u);
    }
    HIP_TRY(hipGetLastError());
    if (small) {
        {
            KernelTimer kt(c, st, 5);
            const WaveShape ws = wave_shape(c, (groups + ml::TEAMS - 1) / ml::TEAMS);
            hipLaunchKernelGGL(ml::k_ml_small, dim3(ws.blocks), dim3(ws.threads), 0, st, (const int32_t*)c->d_lines,
                               (const uint8_t*)c->d_bad, (uint32_t)n, (uint32_t)gsz, (uint32_t)groups, d_partials, 144u,
                               (const uint32_t*)nullptr, (const uint32_t*)nullptr);
        }
        HIP_TRY(hipGetLastError());
        if (c->bulk_event) HIP_TRY(hipEventRecord(c->bulk_event, st));
        return 0;
    }
    size_t teams = groups * cpg * ml::LINES;
    {
        KernelTimer kt(c, st, 5);
        const WaveShape ws = wave_shape(c, (teams + ml::TEAMS - 1) / ml::TEAMS);
        hipLaunchKernelGGL(ml::k_ml_accum, dim3(ws.blocks), dim3(ws.threads), 0, st, (const int32_t*)c->d_lines,
                           (const uint8_t*)c->d_bad, (uint32_t)n, (uint32_t)gsz, (uint32_t)chunk, (uint32_t)cpg, (uint32_t)teams,
                           (int32_t*)c->d_lsp[0]);
    }
    HIP_TRY(hipGetLastError());
    int cur = 0;
    while (cpg > 1) {
        const size_t cpo = (cpg + FAN - 1) / FAN;
        teams = groups * cpo * ml::LINES;
        KernelTimer kt(c, st, 6);
        if (teams <= c->ls_merge_wide_max)                                 // few outputs: one wavefront each, a product per lane
            hipLaunchKernelGGL(fxw::k_ml_merge_wide, dim3((unsigned)teams), dim3(64), 0, st, (const int32_t*)c->d_lsp[cur], (uint32_t)cpg,
                               (uint32_t)FAN, (uint32_t)cpo, (int32_t*)c->d_lsp[cur ^ 1]);
        else {
            const WaveShape ws = wave_shape(c, (teams + ml::TEAMS - 1) / ml::TEAMS);
            hipLaunchKernelGGL(ml::k_ml_merge, dim3(ws.blocks), dim3(ws.threads), 0, st,
                               (const int32_t*)c->d_lsp[cur], (uint32_t)cpg, (uint32_t)FAN, (uint32_t)cpo, (uint32_t)teams,
                               (int32_t*)c->d_lsp[cur ^ 1]);
        }
        HIP_TRY(hipGetLastError());
        cpg = cpo;
        cur ^= 1;
    }
    if (c->bulk_event) HIP_TRY(hipEventRecord(c->bulk_event, st));
    {
        const bool fuse = d_fused_out != nullptr && use_fexp_wide(c, 1, groups);
        KernelTimer kt(c, st, fuse ? 2 : 7);
        hipLaunchKernelGGL(fxw::k_ml_horner_fexp, dim3((unsigned)groups), dim3(64), 0, st, (const int32_t*)c->d_lsp[cur], d_partials, 144u,
                           (uint32_t*)(fuse ? d_fused_out : nullptr));
        if (fused) *fused = fuse;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// Miller loops + per-group product; final exponentiation iff d_out_bytes
static int grouped_pairing(blsgpu_ctx* c, const void* d_g1, const void* d_g2, const void* d_inf, size_t gsz, size_t groups,
                           uint32_t* d_out_partial, void* d_out_bytes, hipStream_t st) {
    constexpr size_t MAX_GROUPS = 32768;               // groups ride on gridDim.y: larger batches go in slices
    if (groups > MAX_GROUPS) {
        for (size_t g0 = 0; g0 < groups; g0 += MAX_GROUPS) {
            const size_t gn = groups - g0 < MAX_GROUPS ? groups - g0 : MAX_GROUPS;
            int rc = grouped_pairing(c, (const char*)d_g1 + g0 * gsz * BLSGPU_G1_BYTES, (const char*)d_g2 + g0 * gsz * BLSGPU_G2_BYTES,
                                     d_inf ? (const char*)d_inf + g0 * gsz * 2 : nullptr, gsz,
                                     gn, d_out_partial ? d_out_partial + g0 * 144 : nullptr,
                                     d_out_bytes ? (char*)d_out_bytes + g0 * BLSGPU_FQ12_BYTES : nullptr, st);
            if (rc) return rc;
        }
        return 0;
    }
    size_t need_pairs = (gsz + 3) * groups;            // every group rounds its team count up
    if ((need_pairs + 1) / 2 + (need_pairs + MILLER_WAVES - 1) / MILLER_WAVES + 1 > c->part_cap) {
        int rc = ensure_workspace(c, need_pairs);
        if (rc) return rc;
    }
    size_t bpg = 0;
    bool ls_done = false;
    if (gsz > 0 && use_ls(c, gsz, groups)) {
        // The line records are 22.8 KB per pair: a call is cut into slices of at most LS_MAX_PAIRS pairs -- whole groups,
        // or, for ONE long group, runs of its pairs that each leave a partial for the product below.
        int rc = 0;
        if (gsz * groups <= LS_MAX_PAIRS) {
            bool fused = false;
            // one partial per group comes out of the stage: straight into the caller's buffer when that is all it wants
            // (the sharded entries), no copying pass of k_reduce behind it
            uint32_t* target = (d_out_partial && !d_out_bytes) ? d_out_partial : c->d_part[0];
            rc = launch_miller_ls(c, d_g1, d_g2, d_inf, gsz, groups, target, st, d_out_bytes, &fused);
            if (rc == 0 && (fused || target == d_out_partial)) return 0;   // (fused: the stage's last kernel ran the final exponentiations too)
            bpg = 1;
        } else if (gsz <= LS_MAX_PAIRS) {
            const size_t per = LS_MAX_PAIRS / gsz;
            for (size_t g0 = 0; g0 < groups && !rc; g0 += per) {
                const size_t gn = groups - g0 < per ? groups - g0 : per;
                rc = launch_miller_ls(c, (const char*)d_g1 + g0 * gsz * BLSGPU_G1_BYTES, (const char*)d_g2 + g0 * gsz * BLSGPU_G2_BYTES,
                                      d_inf ? (const char*)d_inf + g0 * gsz * 2 : nullptr, gsz, gn, c->d_part[0] + g0 * 144, st);
            }
            bpg = 1;
        } else if (groups == 1) {
            size_t k = 0;
            for (size_t p0 = 0; p0 < gsz && !rc; p0 += LS_MAX_PAIRS, k++) {
                const size_t pn = gsz - p0 < LS_MAX_PAIRS ? gsz - p0 : LS_MAX_PAIRS;
                rc = launch_miller_ls(c, (const char*)d_g1 + p0 * BLSGPU_G1_BYTES, (const char*)d_g2 + p0 * BLSGPU_G2_BYTES,
                                      d_inf ? (const char*)d_inf + p0 * 2 : nullptr, pn, 1, c->d_part[0] + k * 144, st);
            }
            bpg = k;
        } else {
            rc = -ENOMEM;                                  // several groups of more than LS_MAX_PAIRS pairs: the VM kernels
        }
        if (rc == 0) ls_done = true;
        else if (rc != -ENOMEM) return rc;                 // no memory for the line records: the wavefront-VM kernels instead
    }
    if (ls_done) {
    } else if (gsz > 0) {
        int rc = launch_miller(c, d_g1, d_g2, d_inf, gsz, groups, false, c->d_part[0], st, &bpg);
        if (rc) return rc;
    }
    return reduce_chain(c, c->d_part[0], bpg, groups, 1, bpg, d_out_bytes != nullptr, d_out_partial, d_out_bytes, st);
}

BLSGPU_EXPORT int blsgpu_miller_product_dev(blsgpu_ctx* c, const void* d_g1, const void* d_g2, const void* d_inf, size_t n,
                              void* d_partial, void* stream) {
    return blsgpu_miller_product_batch_dev(c, d_g1, d_g2, d_inf, n, 1, d_partial, stream);
}

BLSGPU_EXPORT int blsgpu_miller_product_batch_dev(blsgpu_ctx* c, const void* d_g1, const void* d_g2, const void* d_inf, size_t gsz,
                                                  size_t groups, void* d_partials, void* stream) {
    if (!c || (groups && !d_partials)) return fail(-EINVAL, "NULL argument");
    if (groups == 0) return 0;
    if (gsz > 0 && (!d_g1 || !d_g2)) return fail(-EINVAL, "NULL point buffer");
    if (gsz > 0xFFFFFFF0ull || gsz * groups > 0xFFFFFFF0ull) return fail(-EINVAL, "batch too large");
    HIP_TRY(hipSetDevice(c->device));
    StreamGuard sg(c, (hipStream_t)stream);
    return grouped_pairing(c, d_g1, d_g2, d_inf, gsz, groups, (uint32_t*)d_partials, nullptr, (hipStream_t)stream);
}

BLSGPU_EXPORT int blsgpu_final_exp_product_dev(blsgpu_ctx* c, const void* d_partials, size_t m, void* d_out, void* stream) {
    return blsgpu_final_exp_product_batch_dev(c, d_partials, m, 1, d_out, stream);
}

// d_partials holds m x groups partials, partial (i, g) at index i * groups + g: the
// layout an all-gather of every rank's `groups` partials produces.
BLSGPU_EXPORT int blsgpu_final_exp_product_batch_dev(blsgpu_ctx* c, const void* d_partials, size_t m, size_t groups, void* d_out,
                                                     void* stream) {
    if (!c || (groups && !d_out)) return fail(-EINVAL, "NULL argument");
    if (groups == 0) return 0;
    if (m > 0 && !d_partials) return fail(-EINVAL, "NULL partials");
    HIP_TRY(hipSetDevice(c->device));
    StreamGuard sg(c, (hipStream_t)stream);
    if (((m + REDUCE_PER_BLOCK - 1) / REDUCE_PER_BLOCK) * groups + 1 > c->part_cap) {
        int rc = ensure_workspace(c, m * groups * MILLER_WAVES);
        if (rc) return rc;
    }
    return reduce_chain(c, (const uint32_t*)d_partials, m, groups, groups, 1, true, nullptr, d_out, (hipStream_t)stream);
}

BLSGPU_EXPORT int blsgpu_pairing_multi_dev(blsgpu_ctx* c, const void* d_g1, const void* d_g2, const void* d_inf, size_t n,
                             void* d_out, void* stream) {
    if (!c || !d_out) return fail(-EINVAL, "NULL argument");
    if (n > 0 && (!d_g1 || !d_g2)) return fail(-EINVAL, "NULL point buffer");
    if (n > 0xFFFFFFF0ull) return fail(-EINVAL, "n too large");
    HIP_TRY(hipSetDevice(c->device));
    StreamGuard sg(c, (hipStream_t)stream);
    return grouped_pairing(c, d_g1, d_g2, d_inf, n, 1, nullptr, d_out, (hipStream_t)stream);
}

BLSGPU_EXPORT int blsgpu_pairing_multi(blsgpu_ctx* c, const uint8_t* g1, const uint8_t* g2, const uint8_t* inf, size_t n,
                                       uint8_t out[576]) {
    if (!c || !out) return fail(-EINVAL, "NULL argument");
    if (n > 0 && (!g1 || !g2)) return fail(-EINVAL, "NULL point buffer");
    HIP_TRY(hipSetDevice(c->device));
    size_t need = n * (BLSGPU_G1_BYTES + BLSGPU_G2_BYTES + 2);
    if (int rc_ = grow_buffer(c, &c->d_io, &c->io_cap, need)) return rc_;
    char* d1 = (char*)c->d_io;
    char* d2 = d1 + n * BLSGPU_G1_BYTES;
    char* di = d2 + n * BLSGPU_G2_BYTES;
    if (n) {
        HIP_TRY(hipMemcpyAsync(d1, g1, n * BLSGPU_G1_BYTES, hipMemcpyHostToDevice, 0));
        HIP_TRY(hipMemcpyAsync(d2, g2, n * BLSGPU_G2_BYTES, hipMemcpyHostToDevice, 0));
        if (inf) HIP_TRY(hipMemcpyAsync(di, inf, n * 2, hipMemcpyHostToDevice, 0));
    }
    int rc = blsgpu_pairing_multi_dev(c, d1, d2, inf ? di : nullptr, n, c->d_out, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, c->d_out, BLSGPU_FQ12_BYTES, hipMemcpyDeviceToHost));
    return 0;
}

// fq_miller_loop (fields_t.py:1091-1111) of every pair: n x 576 bytes, the reference's own
// Miller values (not multiples of them).
BLSGPU_EXPORT int blsgpu_miller_loop_batch_dev(blsgpu_ctx* c, const void* d_g1, const void* d_g2, const void* d_inf, size_t n,
                                               void* d_out, void* stream) {
    if (!c || (n && (!d_g1 || !d_g2 || !d_out))) return fail(-EINVAL, "NULL argument");
    if (n == 0) return 0;
    if (n > 0x7FFFFFF0ull) return fail(-EINVAL, "n too large");
    HIP_TRY(hipSetDevice(c->device));
    StreamGuard sg(c, (hipStream_t)stream);
    if (c->miller_exact_lanes) {
        // Round 5: the lane kernels instead of the VM's reference-faithful program (one pair per wavefront, 68 lane-serial inversions:
        // 7 ms for one pair, 158 k pairs/s): every pair on the work list, k_ml_lines_exact writes its 68 lines with the reference's
        // own formulas (a pair per lane pair), k_ml_small multiplies them up with one six-lane accumulator per pair, in slices that
        // keep the line records below 6 GB.
        using namespace blsgpu;
        const hipStream_t st = (hipStream_t)stream;
        const size_t slice = 262144;
        const size_t m0 = n < slice ? n : slice;
        if (c->miller_exact_fast && grow_buffer(c, &c->d_exflags, &c->exflags_cap, 2 * m0) == 0 && ensure_workspace(c, 2 * m0) == 0) {
            // the ordinary line-stream kernels with one accumulator per pair, then one Fq2 factor per pair turns the fast value into the
            // reference's (k_ml_exact_fixup: 73 dependent inversions per pair become one); a pair the fast formulas are not valid for
            // -- or whose py is 0, which the factor divides by -- takes the reference's own lines inside the same launches
            bool ok = true;
            for (size_t lo = 0; lo < n && ok; lo += slice) {
                const size_t m = n - lo < slice ? n - lo : slice;
                const uint32_t* p1 = (const uint32_t*)d_g1 + lo * 24;
                const uint32_t* p2 = (const uint32_t*)d_g2 + lo * 48;
                hipLaunchKernelGGL(ml::k_ml_exact_flags, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, p1, d_inf ? (const uint8_t*)d_inf + 2 * lo : nullptr,
                                   (uint32_t)m, (uint8_t*)c->d_exflags);
                const int rc_ = launch_miller_ls(c, p1, p2, c->d_exflags, 1, m, c->d_part[0], st);
                if (rc_ == -ENOMEM && lo == 0) { ok = false; break; }          // no room for the line records: the forms below
                if (rc_) return rc_;
                hipLaunchKernelGGL(ml::k_ml_exact_fixup, dim3((unsigned)((2 * m + 255) / 256)), dim3(256), 0, st, p1, (const int32_t*)c->d_lines,
                                   (const uint8_t*)c->d_bad, c->d_part[0], (uint32_t)m, (uint32_t*)d_out + lo * 144);
                HIP_TRY(hipGetLastError());
            }
            if (ok) return 0;
        }
        if (grow_buffer(c, &c->d_lines, &c->lines_cap, m0 * ml::LINES * ml::LINE_DW * 4) == 0 && grow_buffer(c, &c->d_bad, &c->bad_cap, m0) == 0 &&
            grow_elems(c, &c->d_degen, &c->degen_cap, m0 + 2) == 0 && ensure_workspace(c, 2 * m0) == 0) {
            for (size_t lo = 0; lo < n; lo += slice) {
                const size_t m = n - lo < slice ? n - lo : slice;
                const uint32_t* p1 = (const uint32_t*)d_g1 + lo * 24;
                const uint32_t* p2 = (const uint32_t*)d_g2 + lo * 48;
                DegenList dg{c->d_degen, c->d_degen + 1, d_inf ? (const uint8_t*)d_inf + 2 * lo : nullptr};
                hipLaunchKernelGGL(ml::k_ml_list_all, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, (uint32_t)m, c->d_degen, c->d_degen + 1);
                hipLaunchKernelGGL(ml::k_ml_lines_exact, dim3(2048), dim3(64), 0, st, p1, p2, (uint32_t)m, (int32_t*)c->d_lines, (uint8_t*)c->d_bad, dg, 0u, 0u, 0u);
                const WaveShape ws = wave_shape(c, (m + ml::TEAMS - 1) / ml::TEAMS);
                hipLaunchKernelGGL(ml::k_ml_small, dim3(ws.blocks), dim3(ws.threads), 0, st, (const int32_t*)c->d_lines, (const uint8_t*)c->d_bad, (uint32_t)m, 1u,
                                   (uint32_t)m, c->d_part[0], 144u, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
                hipLaunchKernelGGL(ml::k_ml_partials_to_bytes, dim3((unsigned)((m * 12 + 255) / 256)), dim3(256), 0, st, c->d_part[0], (uint32_t)(m * 12),
                                   (uint32_t*)d_out + lo * 144);
                HIP_TRY(hipGetLastError());
            }
            return 0;
        }
        (void)hipGetLastError();                               // no room for the line records: the VM's program below
    }
    const unsigned grid = (unsigned)(n < 16384 ? n : 16384);
    hipLaunchKernelGGL(blsgpu::k_miller_exact, dim3(grid), dim3(64), (size_t)blsgpu::SLOW_TEAM_BYTES, (hipStream_t)stream, c->tabs,
                       (const uint32_t*)d_g1, (const uint32_t*)d_g2, (const uint8_t*)d_inf, (uint32_t)n, (uint32_t*)d_out);
    HIP_TRY(hipGetLastError());
    return 0;
}
BLSGPU_EXPORT int blsgpu_miller_loop_batch(blsgpu_ctx* c, const uint8_t* g1, const uint8_t* g2, const uint8_t* inf, size_t n,
                                           uint8_t* out) {
    if (!c || (n && (!g1 || !g2 || !out))) return fail(-EINVAL, "NULL argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    size_t need = n * (BLSGPU_G1_BYTES + BLSGPU_G2_BYTES + BLSGPU_FQ12_BYTES + 2) + 64;
    if (int rc_ = grow_buffer(c, &c->d_io, &c->io_cap, need)) return rc_;
    char* dout = (char*)c->d_io;
    char* d1 = dout + n * BLSGPU_FQ12_BYTES;
    char* d2 = d1 + n * BLSGPU_G1_BYTES;
    char* di = d2 + n * BLSGPU_G2_BYTES;
    HIP_TRY(hipMemcpyAsync(d1, g1, n * BLSGPU_G1_BYTES, hipMemcpyHostToDevice, 0));
    HIP_TRY(hipMemcpyAsync(d2, g2, n * BLSGPU_G2_BYTES, hipMemcpyHostToDevice, 0));
    if (inf) HIP_TRY(hipMemcpyAsync(di, inf, n * 2, hipMemcpyHostToDevice, 0));
    int rc = blsgpu_miller_loop_batch_dev(c, d1, d2, inf ? di : nullptr, n, dout, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, dout, n * BLSGPU_FQ12_BYTES, hipMemcpyDeviceToHost));
    return 0;
}

// fq2_double_line_eval (q == NULL) / fq2_add_line_eval on n (R, [Q,] P) triples
BLSGPU_EXPORT int blsgpu_line_eval_batch(blsgpu_ctx* c, const uint8_t* r, const uint8_t* q, const uint8_t* p, size_t n, uint8_t* out) {
    if (!c || (n && (!r || !p || !out))) return fail(-EINVAL, "NULL argument");
    if (n == 0) return 0;
    if (n > 0x00FFFFF0ull) return fail(-EINVAL, "n too large");
    HIP_TRY(hipSetDevice(c->device));
    size_t need = n * (2 * BLSGPU_G2_BYTES + BLSGPU_G1_BYTES + BLSGPU_FQ12_BYTES) + 64;
    if (int rc_ = grow_buffer(c, &c->d_io, &c->io_cap, need)) return rc_;
    StreamGuard sg(c, nullptr);
    char* dout = (char*)c->d_io;
    char* dr = dout + n * BLSGPU_FQ12_BYTES;
    char* dq = dr + n * BLSGPU_G2_BYTES;
    char* dp = dq + n * BLSGPU_G2_BYTES;
    HIP_TRY(hipMemcpyAsync(dr, r, n * BLSGPU_G2_BYTES, hipMemcpyHostToDevice, 0));
    if (q) HIP_TRY(hipMemcpyAsync(dq, q, n * BLSGPU_G2_BYTES, hipMemcpyHostToDevice, 0));
    HIP_TRY(hipMemcpyAsync(dp, p, n * BLSGPU_G1_BYTES, hipMemcpyHostToDevice, 0));
    const unsigned grid = (unsigned)(n < 16384 ? n : 16384);
    hipLaunchKernelGGL(blsgpu::k_line_eval, dim3(grid), dim3(64), (size_t)blsgpu::SLOW_TEAM_BYTES, 0, c->tabs, (const uint32_t*)dr,
                       q ? (const uint32_t*)dq : (const uint32_t*)nullptr, (const uint32_t*)dp, (uint32_t)n, (uint32_t*)dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * BLSGPU_FQ12_BYTES, hipMemcpyDeviceToHost));
    return 0;
}

// Fq12 field operations on n elements (op: 0 add, 1 sub, 2 mul, 3 neg, 4 invert), or a^e for one exponent
namespace {
int fq12_op_host(blsgpu_ctx* c, uint32_t op, const uint8_t* a, const uint8_t* b, const uint8_t* ebits, size_t nbits, size_t n, uint8_t* out) {
    if (!c || (n && (!a || !out)) || (n && op <= 2 && !b)) return fail(-EINVAL, "NULL argument");
    if (n == 0) return 0;
    if (n > 0x00FFFFF0ull || nbits > 0x10000) return fail(-EINVAL, "too large");
    HIP_TRY(hipSetDevice(c->device));
    size_t need = n * BLSGPU_FQ12_BYTES * 3 + nbits + 64;
    if (int rc_ = grow_buffer(c, &c->d_io, &c->io_cap, need)) return rc_;
    StreamGuard sg(c, nullptr);
    char* dout = (char*)c->d_io;
    char* da = dout + n * BLSGPU_FQ12_BYTES;
    char* db = da + n * BLSGPU_FQ12_BYTES;
    char* de = db + n * BLSGPU_FQ12_BYTES;
    HIP_TRY(hipMemcpyAsync(da, a, n * BLSGPU_FQ12_BYTES, hipMemcpyHostToDevice, 0));
    if (op <= 2) HIP_TRY(hipMemcpyAsync(db, b, n * BLSGPU_FQ12_BYTES, hipMemcpyHostToDevice, 0));
    if (nbits) HIP_TRY(hipMemcpyAsync(de, ebits, nbits, hipMemcpyHostToDevice, 0));
    const unsigned grid = (unsigned)(n < 16384 ? n : 16384);
    hipLaunchKernelGGL(blsgpu::k_fq12_op, dim3(grid), dim3(64), (size_t)blsgpu::SLOW_TEAM_BYTES, 0, c->tabs, op, (const uint32_t*)da,
                       (const uint32_t*)db, (const uint8_t*)de, (uint32_t)nbits, (uint32_t)n, (uint32_t*)dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, n * BLSGPU_FQ12_BYTES, hipMemcpyDeviceToHost));
    return 0;
}
}  // namespace
BLSGPU_EXPORT int blsgpu_fq12_op_batch(blsgpu_ctx* c, int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
    if (op < 0 || op > 4) return fail(-EINVAL, "bad op");
    return fq12_op_host(c, (uint32_t)op, a, b, nullptr, 0, n, out);
}
BLSGPU_EXPORT int blsgpu_fq12_pow_batch(blsgpu_ctx* c, const uint8_t* a, const uint8_t* e_be, size_t e_len, size_t n, uint8_t* out) {
    if (n && (!e_be && e_len)) return fail(-EINVAL, "NULL exponent");
    // exponent bytes (big-endian) -> bits, most significant first, leading zeros dropped
    std::vector<uint8_t> bits;
    bool started = false;
    for (size_t i = 0; i < e_len; i++)
        for (int k = 7; k >= 0; k--) {
            const uint8_t bit = (e_be[i] >> k) & 1;
            started = started || bit;
            if (started) bits.push_back(bit);
        }
    return fq12_op_host(c, 5u, a, nullptr, bits.data(), bits.size(), n, out);
}

// m independent final exponentiations: fq12_final_exp on each 576-byte element
BLSGPU_EXPORT int blsgpu_final_exp_batch(blsgpu_ctx* c, const uint8_t* in, size_t m, uint8_t* out) {
    if (!c || (m && (!in || !out))) return fail(-EINVAL, "NULL argument");
    if (m == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    size_t need = m * 576 * 2 + 64;
    if (int rc_ = grow_buffer(c, &c->d_io, &c->io_cap, need)) return rc_;
    int rc = ensure_workspace(c, m * MILLER_WAVES);
    if (rc) return rc;
    StreamGuard sg(c, nullptr);
    char* din = (char*)c->d_io;
    char* dout = din + m * 576;
    HIP_TRY(hipMemcpy(din, in, m * 576, hipMemcpyHostToDevice));
    size_t lds = (size_t)REDUCE_WAVES * blsgpu::TEAM_BYTES;
    unsigned blocks = (unsigned)((m + REDUCE_WAVES - 1) / REDUCE_WAVES);
    hipLaunchKernelGGL(blsgpu::k_bytes_to_partials, dim3(blocks), dim3(REDUCE_WAVES * 64), lds, 0, c->tabs,
                       (const uint32_t*)din, (uint32_t)m, c->d_part[1]);
    HIP_TRY(hipGetLastError());
    if (use_fexp_reg(c, 1, m)) {
        if (int rc2 = launch_fexp_reg(c, c->d_part[1], 1, 1, 1, m, dout, 0)) return rc2;
    } else {
        hipLaunchKernelGGL(blsgpu::k_final_groups, dim3(blocks), dim3(REDUCE_WAVES * 64), lds, 0, c->tabs, c->d_part[1],
                           1u, (uint32_t)m, (uint32_t*)dout);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpy(out, dout, m * 576, hipMemcpyDeviceToHost));
    return 0;
}

BLSGPU_EXPORT int blsgpu_final_exp(blsgpu_ctx* c, const uint8_t in[576], uint8_t out[576]) {
    return blsgpu_final_exp_batch(c, in, 1, out);
}

// `groups` independent multi-pairings of gsz pairs each (pairs stored group after
// group): out[g] = fq_ate_pairing_multi of group g.  One wavefront per pair for
// the Miller loops, one wavefront per group for product + final exponentiation.
BLSGPU_EXPORT int blsgpu_pairing_multi_batch_dev(blsgpu_ctx* c, const void* d_g1, const void* d_g2, const void* d_inf, size_t gsz,
                                                 size_t groups, void* d_out, void* stream) {
    if (!c || (groups && !d_out)) return fail(-EINVAL, "NULL argument");
    if (groups == 0) return 0;
    size_t n = gsz * groups;
    if (n && (!d_g1 || !d_g2)) return fail(-EINVAL, "NULL point buffer");
    if (n > 0x3FFFFFF0ull) return fail(-EINVAL, "batch too large");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    StreamGuard sg(c, st);
    if (gsz >= BATCH_TREE_MIN_GROUP) return grouped_pairing(c, d_g1, d_g2, d_inf, gsz, groups, nullptr, d_out, st);
    int rc = ensure_workspace(c, (n + 1) * MILLER_WAVES);      // one partial per PAIR here
    if (rc) return rc;
    if (gsz >= 1 && use_ls(c, gsz, groups) && n <= LS_MAX_PAIRS) {
        // a large batch of small groups: point chains on lane pairs, then one accumulator per group (blsgpu_ml.hip)
        rc = launch_miller_ls(c, d_g1, d_g2, d_inf, gsz, groups, c->d_part[0], st);
        if (rc == 0) {
            if (use_fexp_reg(c, 1, groups)) return launch_fexp_reg(c, c->d_part[0], 1, 1, 1, groups, d_out, st);
            size_t lds = (size_t)REDUCE_WAVES * blsgpu::TEAM_BYTES;
            unsigned blocks = (unsigned)((groups + REDUCE_WAVES - 1) / REDUCE_WAVES);
            KernelTimer kt(c, st, 2);
            hipLaunchKernelGGL(blsgpu::k_final_groups, dim3(blocks), dim3(REDUCE_WAVES * 64), lds, st, c->tabs, c->d_part[0], 1u,
                               (uint32_t)groups, (uint32_t*)d_out);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        if (rc != -ENOMEM) return rc;
    }
    // groups of two or three pairs in a batch large enough for the team kernels: the group IS the team (one
    // accumulator, its squarings shared), one partial per group; otherwise one partial per pair
    const bool team_groups = (gsz == 2 || gsz == (size_t)BLSVM_MP_G) && use_mp(c, n) && groups <= 0x7FFFFFFFull;
    if (n) {
        size_t bpg = 0;
        rc = team_groups ? launch_miller(c, d_g1, d_g2, d_inf, gsz, groups, false, c->d_part[0], st, &bpg, (int)gsz)
                         : launch_miller(c, d_g1, d_g2, d_inf, n, 1, true, c->d_part[0], st, &bpg);
        if (rc) return rc;
    }
    if (use_fexp_reg(c, team_groups ? 1 : gsz, groups))
        return launch_fexp_reg(c, c->d_part[0], team_groups ? 1 : gsz, 1, team_groups ? 1 : gsz, groups, d_out, st);
    size_t lds = (size_t)REDUCE_WAVES * blsgpu::TEAM_BYTES;
    unsigned blocks = (unsigned)((groups + REDUCE_WAVES - 1) / REDUCE_WAVES);
    {
        KernelTimer kt(c, st, 2);
        hipLaunchKernelGGL(blsgpu::k_final_groups, dim3(blocks), dim3(REDUCE_WAVES * 64), lds, st, c->tabs, c->d_part[0],
                           (uint32_t)(team_groups ? 1 : gsz), (uint32_t)groups, (uint32_t*)d_out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

BLSGPU_EXPORT int blsgpu_pairing_multi_batch(blsgpu_ctx* c, const uint8_t* g1, const uint8_t* g2, const uint8_t* inf, size_t gsz,
                                             size_t groups, uint8_t* out) {
    if (!c || (groups && !out)) return fail(-EINVAL, "NULL argument");
    if (groups == 0) return 0;
    size_t n = gsz * groups;
    if (n && (!g1 || !g2)) return fail(-EINVAL, "NULL point buffer");
    HIP_TRY(hipSetDevice(c->device));
    size_t need = n * (BLSGPU_G1_BYTES + BLSGPU_G2_BYTES + 2) + groups * 576 + 64;
    if (int rc_ = grow_buffer(c, &c->d_io, &c->io_cap, need)) return rc_;
    char* d1 = (char*)c->d_io;
    char* d2 = d1 + n * BLSGPU_G1_BYTES;
    char* dout = d2 + n * BLSGPU_G2_BYTES;
    char* di = dout + groups * 576;
    if (n) {
        HIP_TRY(hipMemcpyAsync(d1, g1, n * BLSGPU_G1_BYTES, hipMemcpyHostToDevice, 0));
        HIP_TRY(hipMemcpyAsync(d2, g2, n * BLSGPU_G2_BYTES, hipMemcpyHostToDevice, 0));
        if (inf) HIP_TRY(hipMemcpyAsync(di, inf, n * 2, hipMemcpyHostToDevice, 0));
    }
    int rc = blsgpu_pairing_multi_batch_dev(c, d1, d2, inf ? di : nullptr, gsz, groups, dout, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, dout, groups * 576, hipMemcpyDeviceToHost));
    return 0;
}

BLSGPU_EXPORT int blsgpu_g1_msm(blsgpu_ctx* c, const uint8_t* pts, const uint8_t* scalars, size_t k, size_t groups,
                                uint8_t* out, uint8_t* out_inf) {
    return msm_host<1>(c, pts, scalars, k, groups, out, out_inf);
}
BLSGPU_EXPORT int blsgpu_g2_msm(blsgpu_ctx* c, const uint8_t* pts, const uint8_t* scalars, size_t k, size_t groups,
                                uint8_t* out, uint8_t* out_inf) {
    return msm_host<2>(c, pts, scalars, k, groups, out, out_inf);
}
BLSGPU_EXPORT int blsgpu_g1_msm_dev(blsgpu_ctx* c, const void* d_pts, const void* d_scalars, size_t k, size_t groups,
                                    void* d_out, void* d_out_inf, void* stream) {
    if (!c || !d_out) return fail(-EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    return msm_dev<1>(c, d_pts, d_scalars, k, groups, d_out, d_out_inf, (hipStream_t)stream);
}
BLSGPU_EXPORT int blsgpu_g2_msm_dev(blsgpu_ctx* c, const void* d_pts, const void* d_scalars, size_t k, size_t groups,
                                    void* d_out, void* d_out_inf, void* stream) {
    if (!c || !d_out) return fail(-EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    return msm_dev<2>(c, d_pts, d_scalars, k, groups, d_out, d_out_inf, (hipStream_t)stream);
}

// ------------------------------------------------------------ hash to G2 -----
// t: n x 192 bytes = (t0.c0, t0.c1, t1.c0, t1.c1) canonical big-endian, the four
// hash512 values of ec.py:531-534 reduced mod q; out: n x 192 bytes affine G2.
// from_hashes: d_in = message hashes (n x 32 bytes), else t values (n x 192 bytes)
static int map_to_g2_impl(blsgpu_ctx* c, const void* d_in, size_t n, void* d_out, hipStream_t st, bool from_hashes) {
    if (n == 0) return 0;
    if (n > 0x03FFFFF0ull) return fail(-EINVAL, "batch too large");
    StreamGuard sg(c, st);
    const size_t teams = (2 * n + BLSVM_H1_NE - 1) / BLSVM_H1_NE;
    size_t need = teams * blsgpu::H1_IMG * 12 + (from_hashes ? n * 64 : 0);     // stage image (+ digests), u32
    if (int rc_ = grow_elems(c, &c->d_msm_part, &c->msm_part_cap, need)) return rc_;
    uint32_t* img = c->d_msm_part;
    const size_t lds = (size_t)blsgpu::H1_TEAM_DW * 4;
    constexpr uint32_t BASE = BLSVM_H1_BASE - BLSVM_H1_STATE0, ACC = BLSVM_H1_ACC - BLSVM_H1_STATE0;
    const bool lanes = n >= c->h2c_lane_threshold;         // the three encoding stages one encoding per lane (k_h2c_sw*)
    const uint32_t total = (uint32_t)(teams * BLSVM_H1_NE);
    const unsigned lgrid = (unsigned)((total + 63) / 64);
    const WaveShape lsh = wave_shape(c, lgrid);            // the division-step kernels (k_h2c_swj*): any workgroup size
    if (from_hashes) {
        uint32_t* d_dig = img + teams * blsgpu::H1_IMG * 12;
        hipLaunchKernelGGL(blsgpu::k_h2c_hash, dim3((unsigned)((8 * n + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_in,
                           (uint32_t)n, d_dig);
        HIP_TRY(hipGetLastError());
        if (lanes && c->h2c_jacobi && n >= c->h2c_jacobi_threshold)
            hipLaunchKernelGGL(blsgpu::k_h2c_swj0<1>, dim3(lsh.blocks), dim3(lsh.threads), 0, st, (const uint32_t*)d_dig, (uint32_t)(2 * n), total, img);
        else if (lanes)
            hipLaunchKernelGGL(blsgpu::k_h2c_sw0<1>, dim3(lgrid), dim3(64), 0, st, (const uint32_t*)d_dig, (uint32_t)(2 * n), total, img);
        else
            hipLaunchKernelGGL((blsgpu::k_h2c_stage<0, 1>), dim3((unsigned)teams), dim3(64), lds, st, c->tabs, (const uint32_t*)d_dig,
                               (uint32_t)(2 * n), img);
    } else if (lanes && c->h2c_jacobi && n >= c->h2c_jacobi_threshold) {
        hipLaunchKernelGGL(blsgpu::k_h2c_swj0<0>, dim3(lsh.blocks), dim3(lsh.threads), 0, st, (const uint32_t*)d_in, (uint32_t)(2 * n), total, img);
    } else if (lanes) {
        hipLaunchKernelGGL(blsgpu::k_h2c_sw0<0>, dim3(lgrid), dim3(64), 0, st, (const uint32_t*)d_in, (uint32_t)(2 * n), total, img);
    } else {
        hipLaunchKernelGGL((blsgpu::k_h2c_stage<0, 0>), dim3((unsigned)teams), dim3(64), lds, st, c->tabs, (const uint32_t*)d_in,
                           (uint32_t)(2 * n), img);
    }
    HIP_TRY(hipGetLastError());
    const bool jac = lanes && c->h2c_jacobi && n >= c->h2c_jacobi_threshold;   // two powers per encoding instead of five (swl::jacobi)
    int rc = launch_pow(c, img, blsgpu::H1_IMG, BASE, ACC, teams, (jac ? 1 : 3) * BLSVM_H1_NE, st);
    if (rc) return rc;
    if (jac)
        hipLaunchKernelGGL(blsgpu::k_h2c_swj1, dim3(lsh.blocks), dim3(lsh.threads), 0, st, total, img);
    else if (lanes)
        hipLaunchKernelGGL(blsgpu::k_h2c_sw1, dim3(lgrid), dim3(64), 0, st, total, img);
    else
        hipLaunchKernelGGL((blsgpu::k_h2c_stage<1, 0>), dim3((unsigned)teams), dim3(64), lds, st, c->tabs, (const uint32_t*)nullptr,
                           (uint32_t)(2 * n), img);
    HIP_TRY(hipGetLastError());
    rc = launch_pow(c, img, blsgpu::H1_IMG, BASE, ACC, teams, (jac ? 1 : 2) * BLSVM_H1_NE, st);
    if (rc) return rc;
    if (jac)
        hipLaunchKernelGGL(blsgpu::k_h2c_swj2, dim3(lsh.blocks), dim3(lsh.threads), 0, st, total, img);
    else if (lanes)
        hipLaunchKernelGGL(blsgpu::k_h2c_sw2, dim3(lgrid), dim3(64), 0, st, total, img);
    else
        hipLaunchKernelGGL((blsgpu::k_h2c_stage<2, 0>), dim3((unsigned)teams), dim3(64), lds, st, c->tabs, (const uint32_t*)nullptr,
                           (uint32_t)(2 * n), img);
    HIP_TRY(hipGetLastError());
    if (n <= c->h2c_wide_max && n < c->h2c_reg_threshold) {   // a few messages: one per wavefront, a product per lane (the latency form)
        hipLaunchKernelGGL(blsgpu::h2cw::k_h2c_clear_wide, dim3((unsigned)n), dim3(64), 0, st, c->tabs, img, (uint32_t)n, (uint32_t*)d_out);
    } else if (n < c->h2c_reg_threshold) { // the VM form (BLSVM_H2_NM messages per wavefront)
        unsigned b2 = (unsigned)((n + BLSVM_H2_NM - 1) / BLSVM_H2_NM);
        hipLaunchKernelGGL(blsgpu::k_h2c_clear, dim3(b2), dim3(64), (size_t)blsgpu::H2_TEAM_DW * 4, st, c->tabs, img, (uint32_t)n,
                           (uint32_t*)d_out);
    } else if (n <= c->h2c_quad_max) {   // a batch that leaves SIMDs empty on lane pairs: one message per lane QUAD
        const WaveShape ws = wave_shape(c, (4 * n + 63) / 64);                   // every launched lane owns rows of the workspace
        if (int rc2 = grow_buffer(c, &c->d_h2c_ws, &c->h2c_ws_cap, (size_t)ws.blocks * ws.threads * BLS28_H2C_NSLOTS * 3 * blsgpu::r28::NL * 4)) return rc2;
        hipLaunchKernelGGL(blsgpu::k_h2c_clear_quads, dim3(ws.blocks), dim3(ws.threads), 0, st, c->tabs, img, (uint32_t)n,
                           (uint32_t*)c->d_h2c_ws, (uint32_t*)d_out);
    } else {                               // one message per lane pair, the point operations as a script
        const WaveShape ws = wave_shape(c, (2 * n + 63) / 64);
        if (int rc2 = grow_buffer(c, &c->d_h2c_ws, &c->h2c_ws_cap, (size_t)ws.blocks * ws.threads * BLS28_H2C_NSLOTS * 3 * blsgpu::r28::NL * 4)) return rc2;
        hipLaunchKernelGGL(blsgpu::k_h2c_clear_pairs, dim3(ws.blocks), dim3(ws.threads), 0, st, c->tabs, img, (uint32_t)n,
                           (uint32_t*)c->d_h2c_ws, (uint32_t*)d_out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

BLSGPU_EXPORT int blsgpu_map_to_g2_dev(blsgpu_ctx* c, const void* d_t, size_t n, void* d_out, void* stream) {
    if (!c || (n && (!d_t || !d_out))) return fail(-EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    return map_to_g2_impl(c, d_t, n, d_out, (hipStream_t)stream, false);
}

// The whole hash_to_point_prehashed_Fq2 (ec.py:528-550) for 32-byte message hashes.
BLSGPU_EXPORT int blsgpu_hash_to_g2_dev(blsgpu_ctx* c, const void* d_msg_hashes, size_t n, void* d_out, void* stream) {
    if (!c || (n && (!d_msg_hashes || !d_out))) return fail(-EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    return map_to_g2_impl(c, d_msg_hashes, n, d_out, (hipStream_t)stream, true);
}

// ---- the device work of BLS.verify (bls.py:153-201) in one call ----------------------------------------------------------
// d_g1: (n + 1) x 96 bytes, slot 0 = -G1, slots 1 .. n = the per-message keys (given, or written here by the key sums);
// d_g2: (n + 1) x 192 bytes, slot 0 = the aggregate signature, slots 1 .. n written here by the hash to G2.
BLSGPU_EXPORT int blsgpu_verify_pipeline_dev(blsgpu_ctx* c, void* d_g1, void* d_g2, const void* d_msg_hashes, size_t n, const void* d_key_pts,
                                             const void* d_key_scalars, size_t k, void* d_out, void* stream) {
    if (!c || !d_g1 || !d_g2 || !d_out || (n && !d_msg_hashes)) return fail(-EINVAL, "NULL argument");
    if (k && (!d_key_pts || !d_key_scalars)) return fail(-EINVAL, "NULL key buffer");
    if (n > 0x03FFFFF0ull) return fail(-EINVAL, "batch too large");
    HIP_TRY(hipSetDevice(c->device));
    if (n) {
        if (int rc = map_to_g2_impl(c, d_msg_hashes, n, (char*)d_g2 + BLSGPU_G2_BYTES, (hipStream_t)stream, true)) return rc;
        if (k)
            if (int rc = msm_dev<1>(c, d_key_pts, d_key_scalars, k, n, (char*)d_g1 + BLSGPU_G1_BYTES, nullptr, (hipStream_t)stream)) return rc;
    }
    return blsgpu_pairing_multi_dev(c, d_g1, d_g2, nullptr, n + 1, d_out, stream);
}

// Host buffers: ONE upload (points, hashes, keys), the three stages on the device, 576 bytes back.
BLSGPU_EXPORT int blsgpu_verify_pipeline(blsgpu_ctx* c, const uint8_t neg_g1[96], const uint8_t sig[192], const uint8_t* msg_hashes, size_t n,
                                         const uint8_t* keys_affine, const uint8_t* key_pts, const uint8_t* key_scalars, size_t k,
                                         uint8_t out[576]) {
    if (!c || !neg_g1 || !sig || !out || (n && !msg_hashes)) return fail(-EINVAL, "NULL argument");
    if (n && !keys_affine && !(k && key_pts && key_scalars)) return fail(-EINVAL, "neither keys nor key sums given");
    if (n > 0x03FFFFF0ull) return fail(-EINVAL, "batch too large");
    HIP_TRY(hipSetDevice(c->device));
    const bool sums = n && !keys_affine;
    const size_t o_g1 = 0, o_g2 = (n + 1) * BLSGPU_G1_BYTES, o_h = o_g2 + (n + 1) * BLSGPU_G2_BYTES, o_kp = o_h + ((n * 32 + 63) & ~(size_t)63);
    const size_t o_ks = o_kp + (sums ? n * k * BLSGPU_G1_BYTES : 0), o_out = o_ks + (sums ? n * k * 32 : 0);
    // the VM kernels round their team counts up: reserve as a call of n + 1 pairs does
    if (int rc = grow_buffer(c, &c->d_io, &c->io_cap, o_out + 576 + 64)) return rc;
    if (int rc = ensure_workspace(c, (n + 4) * MILLER_WAVES)) return rc;
    char* d = (char*)c->d_io;
    {
        StreamGuard sg(c, nullptr);
        HIP_TRY(hipMemcpyAsync(d + o_g1, neg_g1, BLSGPU_G1_BYTES, hipMemcpyHostToDevice, nullptr));
        HIP_TRY(hipMemcpyAsync(d + o_g2, sig, BLSGPU_G2_BYTES, hipMemcpyHostToDevice, nullptr));
        if (n) {
            HIP_TRY(hipMemcpyAsync(d + o_h, msg_hashes, n * 32, hipMemcpyHostToDevice, nullptr));
            if (sums) {
                HIP_TRY(hipMemcpyAsync(d + o_kp, key_pts, n * k * BLSGPU_G1_BYTES, hipMemcpyHostToDevice, nullptr));
                HIP_TRY(hipMemcpyAsync(d + o_ks, key_scalars, n * k * 32, hipMemcpyHostToDevice, nullptr));
            } else {
                HIP_TRY(hipMemcpyAsync(d + o_g1 + BLSGPU_G1_BYTES, keys_affine, n * BLSGPU_G1_BYTES, hipMemcpyHostToDevice, nullptr));
            }
        }
    }
    if (int rc = blsgpu_verify_pipeline_dev(c, d + o_g1, d + o_g2, d + o_h, n, sums ? d + o_kp : nullptr, sums ? d + o_ks : nullptr,
                                            sums ? k : 0, d + o_out, nullptr))
        return rc;
    HIP_TRY(hipMemcpy(out, d + o_out, 576, hipMemcpyDeviceToHost));
    return 0;
}

BLSGPU_EXPORT int blsgpu_hash_to_g2(blsgpu_ctx* c, const uint8_t* msg_hashes, size_t n, uint8_t* out) {
    if (!c || (n && (!msg_hashes || !out))) return fail(-EINVAL, "NULL argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    size_t need = n * (32 + 192) + 64;
    if (int rc_ = grow_buffer(c, &c->d_io, &c->io_cap, need)) return rc_;
    char* din = (char*)c->d_io;
    char* dout = din + ((n * 32 + 63) & ~(size_t)63);
    HIP_TRY(hipMemcpy(din, msg_hashes, n * 32, hipMemcpyHostToDevice));
    int rc = map_to_g2_impl(c, din, n, dout, nullptr, true);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, dout, n * 192, hipMemcpyDeviceToHost));
    return 0;
}

BLSGPU_EXPORT int blsgpu_map_to_g2(blsgpu_ctx* c, const uint8_t* t, size_t n, uint8_t* out) {
    if (!c || (n && (!t || !out))) return fail(-EINVAL, "NULL argument");
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    size_t need = n * 192 * 2 + 64;
    if (int rc_ = grow_buffer(c, &c->d_io, &c->io_cap, need)) return rc_;
    char* din = (char*)c->d_io;
    char* dout = din + n * 192;
    HIP_TRY(hipMemcpy(din, t, n * 192, hipMemcpyHostToDevice));
    int rc = blsgpu_map_to_g2_dev(c, din, n, dout, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, dout, n * 192, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------ decompression --
BLSGPU_EXPORT int blsgpu_g1_decompress(blsgpu_ctx* c, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
    return decompress_host<1>(c, in, n, out, ok);
}
BLSGPU_EXPORT int blsgpu_g2_decompress(blsgpu_ctx* c, const uint8_t* in, size_t n, uint8_t* out, uint8_t* ok) {
    return decompress_host<2>(c, in, n, out, ok);
}
BLSGPU_EXPORT int blsgpu_g1_decompress_dev(blsgpu_ctx* c, const void* d_in, size_t n, void* d_out, void* d_ok, void* stream) {
    if (!c || (n && (!d_in || !d_out || !d_ok))) return fail(-EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    return decompress_dev<1>(c, d_in, n, d_out, d_ok, (hipStream_t)stream);
}
BLSGPU_EXPORT int blsgpu_g2_decompress_dev(blsgpu_ctx* c, const void* d_in, size_t n, void* d_out, void* d_ok, void* stream) {
    if (!c || (n && (!d_in || !d_out || !d_ok))) return fail(-EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    return decompress_dev<2>(c, d_in, n, d_out, d_ok, (hipStream_t)stream);
}

#ifdef BLSGPU_STAMPS
// diagnostic build: {cycles MUL, LIN, INV, rounds MUL, LIN, INV} of block 0 / wave 0, then reset
BLSGPU_EXPORT int blsgpu_debug_stamps(blsgpu_ctx* c, unsigned long long out[9]) {
    if (!c || !c->tabs.stamps) return fail(-EINVAL, "no stamps");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, c->tabs.stamps, 72, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(c->tabs.stamps, 0, 128));
    return 0;
}
// diagnostic build: load a whole scratchpad image (nslots x 12 u32), run the first `nrounds`
// rounds of flat program `which` (0 miller, 1 multi-pair, 2 final exp, 3 h2)
// on one team and copy the image back -- lets tools/trace_rounds.py bisect a wrong result
// against vmgen/tablesim.py round by round
BLSGPU_EXPORT int blsgpu_debug_run(blsgpu_ctx* c, int which, unsigned nrounds, unsigned nslots, uint32_t* image) {
    if (!c || !image || nslots == 0 || nslots > 1023) return fail(-EINVAL, "bad argument");
    const uint2* seqs[4] = {c->tabs.mflat, c->tabs.mpflat, c->tabs.fflat, c->tabs.h2flat};
    if (which < 0 || which > 3) return fail(-EINVAL, "bad program");
    uint32_t* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (size_t)nslots * 48));
    HIP_TRY(hipMemcpy(d, image, (size_t)nslots * 48, hipMemcpyHostToDevice));
    (void)hipFuncSetAttribute((const void*)blsgpu::k_debug_run, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nslots * 48);
    hipLaunchKernelGGL(blsgpu::k_debug_run, dim3(1), dim3(64), (size_t)nslots * 48, 0, c->tabs, seqs[which], nrounds, nslots, d, which == 1 ? 1u : 0u);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(image, d, (size_t)nslots * 48, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return 0;
}
#endif

}  // extern "C"
#endif  // BLSGPU_TU_HOST
