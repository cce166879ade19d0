/*
 * blsgpu.h -- C ABI of the MI355X (gfx950) BLS12-381 multi-pairing engine.
 *
 * Drop-in boundary: this library replaces the reference's only native
 * component for the aggregate-verify path, the Cython/GMP module
 * bls_py.fields_t_c (extmod/bls_py/fields_t_c.pyx), whose functions shadow the
 * pure-Python ones of bls_py/fields_t.py at import (fields_t.py:1218-1265).
 * Each entry point names the reference function it stands in for.
 *
 * Byte conventions = the reference's own serialisation:
 *   Fq    48-byte big-endian canonical residue           (fields.py:87-88)
 *   Fq12  12 x Fq in flat "ZT" order, 576 bytes          (fields.py:624-629, 273-278)
 *   G1    affine x || y, 96 bytes    -- the (x, y) of fq_ate_pairing_multi's Ps
 *   G2    affine x.c0 || x.c1 || y.c0 || y.c1, 192 bytes -- the ((x0,x1),(y0,y1)) of Qs
 * Infinity is the reference's coordinate encoding (0,0) (fields_t.py:609-622).
 *   inf   optional n x 2 bytes, (P flag, Q flag) per pair -- the third member of the
 *         (x, y, inf) tuples fq_ate_pairing_multi takes (fields_t_c.pyx:2333-2346); NULL = all
 *         False.  As in the reference only Q's flag has an effect (fields_t.py:676-677).
 * Coordinates must be canonical residues (< q), as the reference assumes (no check there
 * either).  For EVERY such input -- points of the wrong order, off the curve, zero
 * coordinates, flags -- the pairing entry points return the reference's bytes: pairs on
 * which the reference's special cases decide (0^-1 := 0, fields_t.py:47-55; the branches of
 * fq2_add_line_eval :1062-1065 and fq2_add_points :673-686) are detected on the GPU and
 * recomputed there by a reference-faithful program (k_miller_slow).
 *
 * All functions return 0 on success or a negative errno-style code;
 * blsgpu_last_error() describes the last failure on the calling thread.
 * Buffers are caller-allocated; nothing owned by the library crosses the ABI
 * except the opaque context.
 *
 * Threading and streams.  A context is used by ONE host thread at a time (the reference's
 * native module is not re-entrant either: module-global scratch pools,
 * fields_t_c.pyx:2401-2412); different contexts are independent.  Every `_dev` entry point
 * enqueues on the caller's stream and returns; all of a context's work shares one
 * workspace, so a call on another stream than the context's previous call first waits (on
 * the device) for that earlier work -- for concurrency use one context per stream.  The
 * workspace only grows: a buffer that a larger one replaces is kept until
 * blsgpu_ctx_trim / blsgpu_ctx_destroy, so growth never invalidates enqueued work and never
 * synchronises the device inside a pipeline; blsgpu_ctx_reserve sizes it up front.
 */
#ifndef BLSGPU_H
#define BLSGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLSGPU_FQ_BYTES 48
#define BLSGPU_G1_BYTES 96
#define BLSGPU_G2_BYTES 192
#define BLSGPU_FQ12_BYTES 576
#define BLSGPU_PARTIAL_WORDS 144   /* one Fq12 in Montgomery limb form, uint32 */

typedef struct blsgpu_ctx blsgpu_ctx;

/* Library / table identification string (static storage). */
const char *blsgpu_version(void);
const char *blsgpu_last_error(void);

/* Create a context on HIP device `device` (uploads the schedule tables).
 * Fails with -ENODEV (-19) when no gfx950-class GPU is usable: there is no CPU
 * fallback in this library. */
int blsgpu_ctx_create(int device, blsgpu_ctx **out);
void blsgpu_ctx_destroy(blsgpu_ctx *ctx);
/* Pre-size the per-context workspace for batches of up to max_pairs pairs. */
int blsgpu_ctx_reserve(blsgpu_ctx *ctx, size_t max_pairs);
/* Bytes of device memory the context's grow-only workspace holds right now, by purpose (the high-water mark of the
 * calls made so far; the line-stream stage keeps 68 x 336 bytes of line records per pair of its largest call, the
 * counterpart of the n-arrays of mpz_t the reference mallocs per call, fields_t_c.pyx:2348-2388).  out[BLSGPU_WS_TOTAL]
 * is the sum.  No device call is made. */
enum { BLSGPU_WS_PARTIALS = 0, BLSGPU_WS_STAGING, BLSGPU_WS_LINES, BLSGPU_WS_LINE_PRODUCTS, BLSGPU_WS_FLAGS_AND_LISTS,
       BLSGPU_WS_GROUP_SUMS, BLSGPU_WS_SLOTS, BLSGPU_WS_TOTAL, BLSGPU_WS_FIELDS };
int blsgpu_ctx_workspace_bytes(blsgpu_ctx *ctx, size_t out[BLSGPU_WS_FIELDS]);
/* Wait for the context's enqueued work and free the buffers that larger ones replaced. */
int blsgpu_ctx_trim(blsgpu_ctx *ctx);
/* Batches of at least `pairs` pairs run the throughput-oriented Miller kernel
 * (several pairs per wavefront sharing one accumulator); smaller batches the
 * latency-oriented one (one pair per wavefront).  Default 4096; 0 = always the
 * throughput kernel.  Results are identical either way. */
/* Measurement aid (bench.py): the chip's 32 x 32 + 64-bit multiply-add rate (v_mad_i64_i32) measured NOW by a probe kernel of
 * about `target_ms` milliseconds on `stream`; *tmacs = 10^12 multiply-adds per second.  The roofline of this path is that
 * instruction's issue rate (integer VALU, not HBM or MFMA), and the clock the package's power limit leaves differs
 * by a few per cent between boxes. */
int blsgpu_timing_mad_probe(blsgpu_ctx *ctx, double target_ms, double *tmacs, void *stream);
/* Measurement aid: one dispatch of an empty kernel (blsgpu::probe::k_mark) on `stream`.  bench.py brackets every timed region
 * with two of them; tools/collect_profiles.py cuts the per-dispatch counters of a rocprofv3 --pmc run to the dispatches between
 * the marks (the HBM traffic per step of roofline.traffic). */
int blsgpu_timing_mark(blsgpu_ctx *ctx, unsigned tag, void *stream);
int blsgpu_ctx_set_mp_threshold(blsgpu_ctx *ctx, size_t pairs);
/* Calls of at most `pairs` pairs (below the line-stream threshold) run the WIDE Miller loop (csrc/blsgpu_mlw.hip): one pair
 * per workgroup of two wavefronts with a field product per lane -- the loop of fq_miller_loop (fields_t.py:1091-1111) at
 * the depth of one wavefront's instruction stream, the latency form for BLS.verify of a few signatures
 * (bls.py:197-201).  Default 1536 (measured crossover against the wavefront-VM kernel, tools/miller_wide_probe.py); 0 = never (the wavefront-VM kernels).  Results are identical either way. */
int blsgpu_ctx_set_miller_wide_max(blsgpu_ctx *ctx, size_t pairs);
/* The throughput kernel runs three pairs per wavefront from `pairs` pairs per call on and two pairs per
 * wavefront below (a call of a few thousand pairs fills the chip with teams of two and each finishes sooner).
 * 0 = always three; (size_t)-1 = the measured schedule (default: two up to ~8.7 k pairs and in the pockets
 * where teams of two quantise better, three elsewhere).  Results are identical either way. */
int blsgpu_ctx_set_mp3_threshold(blsgpu_ctx *ctx, size_t pairs);
/* Calls of at least `pairs` pairs whose groups all have at least `min_group` pairs run the LINE-STREAM form of
 * the Miller loops (csrc/blsgpu_ml.hip: the twist-point chains alone, their 68 lines per pair through
 * HBM, the accumulator products six lanes each, Horner over the line indices) -- the same loop as
 * fq_miller_loop, fields_t.py:1091-1111, cut where the data dependency allows.  Default 2304 / 64 (the measured
 * crossover, tools/ls_wide_sweep.py; calls of up to 5120 pairs run the point chains sixteen lanes per pair with the values in
 * LDS -- csrc/blsgpu_lsw.hip --, up to 20 480 pairs four lanes per pair, above that two); (size_t)-1
 * keeps every call on the wavefront-VM kernels.  Results are identical either way. */
int blsgpu_ctx_set_ls_threshold(blsgpu_ctx *ctx, size_t pairs, size_t min_group);
/* Number of accumulators the line-stream product kernel aims at (default 163840): a group's pairs are cut into equal
 * chunks of at least 16 pairs, one accumulator per (chunk, line index); the chunks' products are merged by a tree of
 * dense products.  A tuning knob; results are identical for every value. */
int blsgpu_ctx_set_ls_teams(blsgpu_ctx *ctx, size_t teams);
/* `event` (a hipEvent_t, or NULL for none) is recorded on the call's stream right after the last kernel of a Miller
 * stage that fills the chip; what follows (Horner over the line products, the product of the partials, the final
 * exponentiation) occupies a few dozen wavefronts.  A caller that pipelines calls over several contexts lets the
 * next call's stream wait for this event instead of the end of the call (bench.py does). */
int blsgpu_ctx_set_bulk_event(blsgpu_ctx *ctx, void *event);
/* Calls that end in at least `results` final exponentiations (fq12_final_exp, fields_t.py:1124-1128) run them six
 * lanes per result, ten results per wavefront, on the register arithmetic (csrc/blsgpu_fexp.hip: the fewest
 * instructions per result, 2.6 ms of latency).  Fewer results run ONE PER WAVEFRONT with every Fq product of a step
 * on its own lane (csrc/blsgpu_fexpw.hip: 0.70 ms for one result against 1.25 ms of the wavefront-VM program of
 * rounds 1 - 3, which BLSGPU_FEXP_WIDE=0 in the environment brings back).  Default 5120 (the measured crossover: 4096 results 3.55 against 4.04 ms, 6144 results 5.09 against 4.77 ms);
 * (size_t)-1: never.  Results are identical either way. */
int blsgpu_ctx_set_fexp_team_threshold(blsgpu_ctx *ctx, size_t results);
/* Diagnostic: a device buffer of (script length) x 576 bytes that receives the accumulator of result 0 after every
 * operation of the batched final exponentiation's script -- the six-lanes-per-result form k_fexp_team only; force it with
 * blsgpu_ctx_set_fexp_team_threshold(ctx, 1) -- (tools/fexp_trace.py compares it with the integer model), or
 * NULL (default). */
int blsgpu_ctx_set_fexp_trace(blsgpu_ctx *ctx, void *d_buf);
/* Diagnostic: a device buffer of (script length + 1) x 8 bytes that receives the GPU cycle counter of result 0 before the
 * one-result-per-wavefront final exponentiation's script (csrc/blsgpu_fexpw.hip) and after every operation of it
 * (tools/fexpw_stamps.py), or NULL (default).  (A buffer of its own: blsgpu_ctx_set_fexp_trace's holds field elements and is
 * written by the six-lanes-per-result form only.) */
int blsgpu_ctx_set_fexpw_stamps(blsgpu_ctx *ctx, void *d_buf);
/* Diagnostic: the first `bytes` bytes of the line records the last line-stream call left in the workspace
 * (lines[(L * n + pair) * 84] int32, csrc/blsgpu_ml.hip); tools/exact_trace.py compares them with the integer model. */
int blsgpu_debug_read_lines(blsgpu_ctx *ctx, void *host_buf, size_t bytes);

/* The device work of BLS.verify (bls.py:153-201) in ONE call:  e(-G1, sig) * prod_i e(P_i, H(m_i))  for n distinct
 * message hashes (32 bytes each) -- hash_to_point_prehashed_Fq2 of every hash (bls.py:194-195, ec.py:528-550), the
 * per-message key P_i either given (keys_affine: n x 96 bytes) or folded here as sum_j t_ij pk_ij (bls.py:177-192:
 * key_pts n x k x 96 bytes, key_scalars n x k x 32 bytes big-endian), then the (n + 1)-pair multi-pairing in the
 * reference's order Ps[0] = -G1, Qs[0] = sig (bls.py:197-199).  One upload, nothing returns to the host between the
 * stages, 576 bytes back: the caller compares them with Fq12 one.  The scheme logic around it (grouping keys by
 * message, the False of a missing tree key, bls.py:181-190) stays with the caller. */
int blsgpu_verify_pipeline(blsgpu_ctx *ctx, const uint8_t neg_g1[BLSGPU_G1_BYTES], const uint8_t sig[BLSGPU_G2_BYTES],
                           const uint8_t *msg_hashes, size_t n, const uint8_t *keys_affine, const uint8_t *key_pts,
                           const uint8_t *key_scalars, size_t k, uint8_t out[BLSGPU_FQ12_BYTES]);
/* The same on device-resident buffers, enqueued on `stream`: d_g1 = (n + 1) x 96 bytes with slot 0 = -G1 and slots
 * 1 .. n the keys (k == 0) or left for the key sums to fill (k > 0); d_g2 = (n + 1) x 192 bytes with slot 0 = sig,
 * slots 1 .. n filled by the hash; d_out receives 576 bytes. */
int blsgpu_verify_pipeline_dev(blsgpu_ctx *ctx, void *d_g1, void *d_g2, const void *d_msg_hashes, size_t n,
                               const void *d_key_pts, const void *d_key_scalars, size_t k, void *d_out, void *stream);

/* fq_ate_pairing_multi(Ps, Qs) -- fields_t.py:1114-1121 / fields_t_c.pyx:2333-2391.
 * Host buffers in, 576 result bytes out; synchronous.  n == 0 returns one. */
int blsgpu_pairing_multi(blsgpu_ctx *ctx, const uint8_t *g1, const uint8_t *g2, const uint8_t *inf,
                         size_t n, uint8_t out[BLSGPU_FQ12_BYTES]);

/* Same computation on device-resident buffers, enqueued on `stream`
 * (a hipStream_t, or NULL for the default stream); asynchronous.
 * d_out receives 576 bytes. */
int blsgpu_pairing_multi_dev(blsgpu_ctx *ctx, const void *d_g1, const void *d_g2, const void *d_inf,
                             size_t n, void *d_out, void *stream);

/* fq_miller_loop(px, py, pinf, qx, qy, qinf) -- fields_t.py:1091-1111 / fields_t_c.pyx:2295-2319
 * (pairing.miller_loop, pairing.py:51-65) for n pairs at once: out receives n x 576 bytes,
 * the reference's own Miller values bit for bit (computed by the reference-faithful program:
 * affine twist point, one inversion per step; the throughput path is blsgpu_pairing_multi). */
int blsgpu_miller_loop_batch(blsgpu_ctx *ctx, const uint8_t *g1, const uint8_t *g2, const uint8_t *inf,
                             size_t n, uint8_t *out);
int blsgpu_miller_loop_batch_dev(blsgpu_ctx *ctx, const void *d_g1, const void *d_g2, const void *d_inf,
                                 size_t n, void *d_out, void *stream);

/* fq2_double_line_eval(R, P) -- fields_t.py:1035-1049 / fields_t_c.pyx:1416-1445 (q == NULL) and
 * fq2_add_line_eval(R, Q, P) -- fields_t.py:1052-1078 / fields_t_c.pyx:1448-1511 (pairing.double_line_eval,
 * pairing.add_line_eval, pairing.py:16-48), for n triples at once: r, q: n x 192 bytes (twist
 * points), p: n x 96 bytes; out: n x 576 bytes, the reference's Fq12 line values. */
int blsgpu_line_eval_batch(blsgpu_ctx *ctx, const uint8_t *r, const uint8_t *q, const uint8_t *p, size_t n,
                           uint8_t *out);

/* The Fq12 arithmetic of the reference's native module on n elements at once (576 bytes each; an Fq, Fq2 or
 * Fq6 element is an Fq12 element whose other coefficients are zero):
 *   op 0 fq12_add (fields_t.py:339-343), 1 fq12_sub (:346-350), 2 fq12_mul (:503-554 / fields_t_c.pyx:824-875),
 *      3 fq12_neg (:321-325), 4 fq12_invert (:328-337; 0 -> 0 as fq_invert :47-55);  b is ignored for 3 and 4.
 * blsgpu_fq12_pow_batch: fq12_pow(a_i, e) (:340-353 / fields_t_c.pyx:771-821) for ONE exponent e given as
 * e_len big-endian bytes. */
int blsgpu_fq12_op_batch(blsgpu_ctx *ctx, int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out);
int blsgpu_fq12_pow_batch(blsgpu_ctx *ctx, const uint8_t *a, const uint8_t *e_be, size_t e_len, size_t n,
                          uint8_t *out);

/* Sharded form (one rank per GPU).  Step 1: the product of the n Miller-loop
 * values of this shard (fq_miller_loop, fields_t.py:1091-1111, folded with
 * fq12_mul as in :1119-1120) as ONE partial of BLSGPU_PARTIAL_WORDS uint32.
 * Step 2, after the partials of all ranks were gathered: their product and the
 * final exponentiation (fq12_final_exp, fields_t.py:1124-1128) -> 576 bytes. */
int blsgpu_miller_product_dev(blsgpu_ctx *ctx, const void *d_g1, const void *d_g2, const void *d_inf,
                              size_t n, void *d_partial, void *stream);
int blsgpu_final_exp_product_dev(blsgpu_ctx *ctx, const void *d_partials, size_t m,
                                 void *d_out, void *stream);

/* fq12_final_exp(t) on a host element (fields_t.py:1124-1128). */
int blsgpu_final_exp(blsgpu_ctx *ctx, const uint8_t in[BLSGPU_FQ12_BYTES],
                     uint8_t out[BLSGPU_FQ12_BYTES]);

/* m independent fq12_final_exp (fields_t.py:1124-1128), host buffers m x 576 bytes. */
int blsgpu_final_exp_batch(blsgpu_ctx *ctx, const uint8_t *in, size_t m, uint8_t *out);

/* `groups` independent fq_ate_pairing_multi calls of gsz pairs each, in one
 * launch sequence (e.g. 10 000 threshold verifications of 2 pairs: BLS.verify,
 * bls.py:153-201, once per group).  Pairs are stored group after group;
 * out receives groups x 576 bytes. */
int blsgpu_pairing_multi_batch(blsgpu_ctx *ctx, const uint8_t *g1, const uint8_t *g2, const uint8_t *inf,
                               size_t gsz, size_t groups, uint8_t *out);
int blsgpu_pairing_multi_batch_dev(blsgpu_ctx *ctx, const void *d_g1, const void *d_g2, const void *d_inf,
                                   size_t gsz, size_t groups, void *d_out, void *stream);

/* Sharded form of the batch: every rank holds gsz pairs of each of the `groups`
 * multi-pairings.  Step 1 writes one partial per group (groups x
 * BLSGPU_PARTIAL_WORDS uint32).  Step 2 takes the all-gathered buffer of m ranks
 * (partial of rank i, group g at index i * groups + g), multiplies per group and
 * applies fq12_final_exp (fields_t.py:1116-1121 per group) -> groups x 576 bytes. */
int blsgpu_miller_product_batch_dev(blsgpu_ctx *ctx, const void *d_g1, const void *d_g2, const void *d_inf,
                                    size_t gsz, size_t groups, void *d_partials, void *stream);
int blsgpu_final_exp_product_batch_dev(blsgpu_ctx *ctx, const void *d_partials, size_t m,
                                       size_t groups, void *d_out, void *stream);

/* Multi-scalar sums  out[g] = sum_{i<k} scalars[g*k+i] * pts[g*k+i]  for `groups`
 * independent groups of k points: the loops of BLS.aggregate_pub_keys
 * (bls.py:203-223, G1), BLS.aggregate_sigs* (bls.py:12-151, G2) and
 * Threshold.aggregate_unit_sigs (threshold.py:127-136, G2), i.e. the reference's
 * fq_/fq2_scalar_mult_jacobian + *_add_points_jacobian (fields_t.py:705-875).
 * pts: affine big-endian coordinates (96 B per G1 point, 192 B per G2 point,
 * (0,0) = infinity); scalars: 32 bytes big-endian each, or NULL for plain sums;
 * out: affine bytes per group ((0,0) for infinity); out_inf (may be NULL): 1 if
 * the group's sum is the point at infinity. */
int blsgpu_g1_msm(blsgpu_ctx *ctx, const uint8_t *pts, const uint8_t *scalars, size_t k,
                  size_t groups, uint8_t *out, uint8_t *out_inf);
int blsgpu_g2_msm(blsgpu_ctx *ctx, const uint8_t *pts, const uint8_t *scalars, size_t k,
                  size_t groups, uint8_t *out, uint8_t *out_inf);
int blsgpu_g1_msm_dev(blsgpu_ctx *ctx, const void *d_pts, const void *d_scalars, size_t k,
                      size_t groups, void *d_out, void *d_out_inf, void *stream);
int blsgpu_g2_msm_dev(blsgpu_ctx *ctx, const void *d_pts, const void *d_scalars, size_t k,
                      size_t groups, void *d_out, void *d_out_inf, void *stream);

/* Hash to G2 after the SHA-256 step: hash_to_point_prehashed_Fq2 (ec.py:528-550)
 * from the two Fq2 elements t0, t1 (the four hash512 values of ec.py:531-534,
 * reduced mod q) onwards: sw_encode twice (ec.py:449-507), their sum, cofactor
 * clearing with psi.  t: n x 192 bytes (t0.c0, t0.c1, t1.c0, t1.c1 big-endian
 * canonical); out: n x 192 bytes affine G2, (0,0) for infinity.  t = 0 encodes
 * to infinity as in ec.py:450-452; a candidate x whose x^3 + b' has zero imaginary part is skipped
 * as the reference's bare `except` skips it (ec.py:489-498).  Where the reference's sw_encode itself
 * raises (its LAST candidate is of that kind -- never for hashed input) the output is unspecified. */
int blsgpu_map_to_g2(blsgpu_ctx *ctx, const uint8_t *t, size_t n, uint8_t *out);
int blsgpu_map_to_g2_dev(blsgpu_ctx *ctx, const void *d_t, size_t n, void *d_out, void *stream);

/* The whole hash_to_point_prehashed_Fq2(m) (ec.py:528-550) for n 32-byte messages m
 * (the message hashes of AggregationInfo, bls.py:194-195): the four hash512 values
 * (util.py:13-16, SHA-256 on the GPU) reduced mod q, then as blsgpu_map_to_g2.
 * msg_hashes: n x 32 bytes; out: n x 192 bytes affine G2. */
int blsgpu_hash_to_g2(blsgpu_ctx *ctx, const uint8_t *msg_hashes, size_t n, uint8_t *out);
int blsgpu_hash_to_g2_dev(blsgpu_ctx *ctx, const void *d_msg_hashes, size_t n, void *d_out, void *stream);

/* Batched point decompression: PublicKey.from_bytes (keys.py:28-40) and
 * Signature.from_bytes (signature.py:21-38) from the serialised bytes: mask the top
 * three bits (`& 0x1f`), y_for_x (ec.py:255-269; square roots fields.py:199-205 and
 * 463-482), and the reference's choice between y and -y by bit 0x80 (G2: on the
 * imaginary part only, signature.py:31-35).  in: n x 48 (G1) / n x 96 (G2) bytes;
 * out: n x 96 / n x 192 bytes affine; ok[i] = 1 iff the reference accepts encoding i (it raises
 * otherwise -- ValueError, or for a G2 x whose x^3 + 4(1+i) has zero imaginary part the AffinePoint
 * constructor's Exception, because Fq2.modsqrt returns an Fq there, fields.py:466-467; out bytes of
 * such an entry are unspecified). */
int blsgpu_g1_decompress(blsgpu_ctx *ctx, const uint8_t *in, size_t n, uint8_t *out, uint8_t *ok);
int blsgpu_g2_decompress(blsgpu_ctx *ctx, const uint8_t *in, size_t n, uint8_t *out, uint8_t *ok);
int blsgpu_g1_decompress_dev(blsgpu_ctx *ctx, const void *d_in, size_t n, void *d_out, void *d_ok, void *stream);
int blsgpu_g2_decompress_dev(blsgpu_ctx *ctx, const void *d_in, size_t n, void *d_out, void *d_ok, void *stream);

/* Measurement aid (bench.py): when enabled, HIP events are recorded on the
 * launch stream around every kernel this context launches (up to 1024 launches
 * between reads).  blsgpu_timing_read waits for them and returns, per launch,
 * the duration in ms and the kernel kind: 0 = k_miller (Miller loops + workgroup
 * product), 1 = k_reduce (partial products), 2 = k_reduce with the final
 * exponentiation, 3 = k_miller_slow (degenerate pairs; empty work list normally).
 * Reading resets the ring. */
int blsgpu_timing_enable(blsgpu_ctx *ctx, int enable);
int blsgpu_timing_read(blsgpu_ctx *ctx, float *ms, int *kind, size_t cap, size_t *count);

#ifdef __cplusplus
}
#endif
#endif
