"""Generate csrc/fp28_mul_gfx950.h: carry-free Montgomery products on 14 signed limbs of 28 bits (R = 2^392).

Why this radix (round 3).  The 12 x 32-bit product (gen_fqmul.py) pays one v_addc_co_u32 per
v_mad_u64_u32 -- 576 of its 623-676 instructions.  With 28-bit limbs a whole column of the schoolbook product,
14 terms of at most 56 bits, fits ONE 64-bit accumulator with room to spare, so a multiply-accumulate is exactly one
v_mad_i64_i32 and a column costs one mask + one shift: 392 + 14 (quotient digits) + 56 = 462 instructions for a
product.  13 x 30-bit limbs (338 multiply-accumulates) leave no room at all: a column of 13 products of 60 bits fills
the accumulator, so the reduction has to be a second pass (26 extra extractions and re-additions: ~507) and no
operand may be an unreduced sum.  Here the room pays a second time:

  * limbs are SIGNED (v_mad_i64_i32, arithmetic shift): a difference is 14 v_sub, a sum 14 v_add, neither needs a
    modular correction nor a carry -- operands of a product may be sums and differences of a few reduced values
    (csrc/fp28.h tracks the limb ranges in the types and refuses a product whose columns could overflow);
  * a product takes a SUM OF PRODUCTS  r = (a0 b0 + a1 b1 + ...) / R  with one reduction: the 196 reduction
    multiply-accumulates are shared, e.g. X3 = t3 t1 - t4 y3 of a point addition costs 588 instead of 784.

Value range: with T = sum a_t b_t and m in [0, R) the result (T + m q) / R lies in (T/R, T/R + q); R / q = 2520, so
|T| < 2520 q^2 gives a result in (-q, 2q).  Digits 0..12 of a result are in [0, 2^28), digit 13 carries the sign.

The statements are grouped like gen_fqmul.py's (hipcc limits an asm statement to 30 operands): one statement per
(column, term), one per column for the m q part.  There is no carry flag to protect, so the statement boundaries
cost nothing; the carry-out operand of v_mad_i64_i32 is vcc, never read.
"""
import os

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
W, L = 28, 14
MASK = (1 << W) - 1
NL = [(Q >> (W * i)) & MASK for i in range(L)]
NINV = (-pow(Q, -1, 1 << W)) % (1 << W)
R = 1 << (W * L)


def stmt(macs, zero_init=False):
    """macs: list of (xexpr, xkind, yexpr, ykind), kinds 'v' / 's'.  acc is operand 0."""
    ops, names, lines = [], {}, []
    for n, (xe, xk, ye, yk) in enumerate(macs):
        idx = []
        for e, k in ((xe, xk), (ye, yk)):
            if (e, k) not in names:
                names[(e, k)] = 1 + len(ops)
                ops.append((e, k))
            idx.append(names[(e, k)])
        add = "0" if (zero_init and n == 0) else "%0"
        lines.append("v_mad_i64_i32 %%0, vcc, %%%d, %%%d, %s" % (idx[0], idx[1], add))
    assert len(ops) + 1 <= 30, len(ops)
    ins = ", ".join('"%s"(%s)' % (k, e) for e, k in ops)
    acc = '"=&v"(acc)' if zero_init else '"+&v"(acc)'
    return '    asm volatile("%s" BLS28_XNOP\n                 : %s : %s : "vcc");\n' % ("\\n\\t".join(lines), acc, ins)


def chunks(macs, zero_init=False):
    """statements of at most 14 multiply-accumulates / 28 distinct operands"""
    out, cur, names = [], [], set()
    for mac in macs:
        ops = {(mac[0], mac[1]), (mac[2], mac[3])}
        if cur and (len(cur) == 14 or len(names | ops) > 28):
            out.append(cur)
            cur, names = [], set()
        cur.append(mac)
        names |= ops
    if cur:
        out.append(cur)
    return "".join(stmt(c, zero_init=(zero_init and n == 0)) for n, c in enumerate(out))


def body(term_macs):
    """term_macs(k) -> list of per-term mac lists for column k.  Emits the interleaved product + reduction."""
    out = []
    for j in range(L):
        out.append("    const int32_t n%d = 0x%07x;\n" % (j, NL[j]))
    out.append("    int32_t " + ", ".join("m%d" % j for j in range(L)) + ";\n")
    out.append("    int64_t acc;\n")
    first = True
    for k in range(L):
        for macs in term_macs(k):
            if macs:
                out.append(chunks(macs, zero_init=first))
                first = False
        if k:
            out.append(chunks([("m%d" % i, "v", "n%d" % (k - i), "s") for i in range(k)]))
        out.append("    m%d = (int32_t)(((uint32_t)acc * 0x%07xu) & 0x%07xu);\n" % (k, NINV, MASK))
        out.append(stmt([("m%d" % k, "v", "n0", "s")]))
        out.append("    acc >>= %d;\n" % W)
    for k in range(L, 2 * L - 1):
        for macs in term_macs(k):
            if macs:
                out.append(chunks(macs))
        out.append(chunks([("m%d" % i, "v", "n%d" % (k - i), "s") for i in range(k - L + 1, L)]))
        out.append("    r[%d] = (int32_t)((uint32_t)acc & 0x%07xu);\n    acc >>= %d;\n" % (k - L, MASK, W))
    out.append("    r[%d] = (int32_t)acc;\n" % (L - 1))
    return "".join(out)


def gen_dot(K):
    args = ", ".join("const int32_t* __restrict__ a%d, const int32_t* __restrict__ b%d" % (t, t) for t in range(K))
    head = ("// r = (%s) / R mod q: %d multiply-accumulates\n" % (" + ".join("a%d b%d" % (t, t) for t in range(K)), (K + 1) * L * L) +
            "__device__ __forceinline__ void fp28_dot%d(int32_t* __restrict__ r, %s) {\n" % (K, args))

    def term_macs(k):
        lo, hi = max(0, k - L + 1), min(k, L - 1)
        return [[("a%d[%d]" % (t, i), "v", "b%d[%d]" % (t, k - i), "v") for i in range(lo, hi + 1)] for t in range(K)]
    return head + body(term_macs) + "}\n"


def gen_sqr(K):
    """r = (a0^2 + a1 b1 + ...) / R: the first term is a square -- its 91 cross products are taken once against
    the doubled operand (d = 2 a: 14 shifts), plus the 14 squares: 105 multiply-accumulates instead of 196."""
    args = "const int32_t* __restrict__ a0" + "".join(", const int32_t* __restrict__ a%d, const int32_t* __restrict__ b%d" % (t, t) for t in range(1, K))
    head = ("// r = (a0^2%s) / R mod q\n" % "".join(" + a%d b%d" % (t, t) for t in range(1, K)) +
            "__device__ __forceinline__ void fp28_sqr%d(int32_t* __restrict__ r, %s) {\n" % (K, args) +
            "    int32_t d[%d];\n#pragma unroll\n    for (int j = 1; j < %d; j++) d[j] = a0[j] << 1;\n" % (L, L))

    def term_macs(k):
        lo, hi = max(0, k - L + 1), min(k, L - 1)
        sq = [("a0[%d]" % i, "v", "d[%d]" % (k - i), "v") for i in range(lo, hi + 1) if i < k - i]
        if k % 2 == 0:
            sq.append(("a0[%d]" % (k // 2), "v", "a0[%d]" % (k // 2), "v"))
        return [sq] + [[("a%d[%d]" % (t, i), "v", "b%d[%d]" % (t, k - i), "v") for i in range(lo, hi + 1)] for t in range(1, K)]
    return head + body(term_macs) + "}\n"


def generate(path=None):
    out = ["/* generated by python-bls_amd/vmgen/gen_fp28.py -- do not edit */\n#pragma once\n#include <stdint.h>\n",
           "// Carry-free Montgomery sums of products on 14 signed 28-bit limbs, R = 2^392 (gfx950 device code only).\n"
           "// Column bound (checked by the callers' types, csrc/fp28.h): 14 * sum_t |a_t limb| |b_t limb| + 14 * 2^56 < 2^63.\n",
           "// (timing experiment: -DBLS28_XNOP='\"\\n\\ts_nop 0\"' appends a wait state to every statement)\n#ifndef BLS28_XNOP\n#define BLS28_XNOP\n#endif\n",
           "namespace bls28 {\n"]
    for K in (1, 2, 3, 4, 6):
        out.append(gen_dot(K))
    for K in (1, 2):
        out.append(gen_sqr(K))
    def arr(name, x, doc):
        return "// %s\n#define %s {%s}\n" % (doc, name, ", ".join("0x%07x" % d for d in to_limbs(x)))
    out.append(arr("BLS28_Q", Q, "q"))
    out.append(arr("BLS28_ONE", R % Q, "R mod q: the Montgomery form of 1"))
    out.append(arr("BLS28_R2", R * R % Q, "R^2 mod q: content c (an integer < 2^391) times this is c R"))
    # the Shallue-van de Woestijne encoding's constants (vmgen/h2c_programs.py; ec.py:449-507), Montgomery form
    S3 = 1586958781458431025242759403266842894121773480562120986020912974854563298150952611241517463240701
    assert (S3 * S3 + 3) % Q == 0
    out.append(arr("BLS28_SW_S3", S3 * R % Q, "sqrt(-3) R"))
    out.append(arr("BLS28_SW_HH", (S3 - 1) * pow(2, -1, Q) % Q * R % Q, "(sqrt(-3) - 1) / 2 R"))
    out.append(arr("BLS28_SW_SINV", pow(S3, -1, Q) * R % Q, "1 / sqrt(-3) R"))
    out.append(arr("BLS28_WIDE_C2", (1 << 384) * R * R % Q, "2^384 R^2 mod q: the digits of hi (a plain integer) times this is hi 2^384 R"))
    out.append(arr("BLS28_HALF", (Q + 1) // 2 * R % Q, "R / 2 mod q: the Montgomery form of 1/2"))
    out.append(arr("BLS28_FROM_VM", (1 << 400) % Q, "2^400 mod q: a value of the wavefront VM (x 2^384, 12 x 32 bits) times this is x R"))
    out.append(arr("BLS28_TO_VM", (1 << 384) % Q, "2^384 mod q: x R times this is x 2^384, the VM's Montgomery form"))
    out.append("// 2^384 mod q as 12 x 32-bit words: the VM's Montgomery form of 1\n#define BLS28_VM_ONE_WORDS {%s}\n" % ", ".join("0x%08xu" % ((((1 << 384) % Q) >> (32 * i)) & 0xFFFFFFFF) for i in range(12)))
    out.append("}  // namespace bls28\n")
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc", "fp28_mul_gfx950.h")
    with open(path, "w") as f:
        f.write("".join(out))
    return path


# ---- exact integer model of the generated code (tests/test_fp28_model.py; also the constants of csrc/fp28.h) ----
def to_limbs(x):
    """signed integer |x| < 2^391 -> 14 digits, 0..12 in [0, 2^28), digit 13 signed"""
    d = []
    for _ in range(L - 1):
        d.append(x & MASK)
        x >>= W
    d.append(x)
    return d


def from_limbs(d):
    return sum(int(v) << (W * i) for i, v in enumerate(d))


def model_dot(terms):
    """terms: list of (a_limbs, b_limbs); the column arithmetic of fp28_dotK with the 64-bit range asserted"""
    m, r, acc = [], [0] * L, 0
    for k in range(2 * L - 1):
        lo, hi = max(0, k - L + 1), min(k, L - 1)
        for a, b in terms:
            for i in range(lo, hi + 1):
                acc += a[i] * b[k - i]
                assert -(1 << 63) <= acc < (1 << 63), "column overflow"
        for i in range(max(0, k - L + 1), min(k, len(m))):
            acc += m[i] * NL[k - i]
            assert -(1 << 63) <= acc < (1 << 63), "column overflow"
        if k < L:
            m.append(((acc & 0xFFFFFFFF) * NINV) & MASK)
            acc += m[k] * NL[0]
            assert -(1 << 63) <= acc < (1 << 63), "column overflow"
            assert acc & MASK == 0
        else:
            r[k - L] = acc & MASK
        acc >>= W
    r[L - 1] = acc
    assert -(1 << 31) <= acc < (1 << 31)
    return r


if __name__ == "__main__":
    print(generate())
