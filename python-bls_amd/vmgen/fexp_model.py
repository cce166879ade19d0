"""Integer model of the final exponentiation in the w-power basis (round 3; csrc/blsgpu_fexp.hip).

fq12_final_exp (fields_t.py:44, 1124-1128) raises to (q^12 - 1)/n.  The wavefront VM does it with one wavefront per
result (958 rounds, 1.25 ms each way it is sliced); a batch of thousands of results -- the 10 000 verifies of BASELINE
configs[3] -- is bound by its instruction count.  This is the same exponent chain (vmgen/programs.final_exp_script:
easy part (q^6 - 1)(q^2 + 1), hard part E = ((x-1)^2/3)(x+q)(x^2+q^2-1) + 1, identity checked in
tests/test_vm_schedule.py) written for SIX LANES PER RESULT in the basis Fq12 = Fq2[w]/(w^6 - xi) of the line-stream
kernels: lane k holds f_k.  What the lanes need:

  mul      dense product (linestream_model.mul_dense: the lane-wise wrap rule)
  conj     f^(q^6): w -> -w, the odd coefficients change sign
  frob_i   f^(q^i): f_k -> conj^i(f_k) * gamma_{i,k}, gamma_{i,k} = xi^(k (q^i - 1)/6) -- lane-local
  inverse  through norms, so that it is products and Frobenius maps only:
             N = f conj(f) in Fq6 = Fq2[w^2];  t = N N^(q^2) N^(q^4) in Fq2;  f^-1 = conj(f) N^(q^2) N^(q^4) t^-1,
             t^-1 = conj(t) / (t0^2 + t1^2),  the one Fq inversion as a fixed power a^(q-2)  (0 -> 0 as fields_t.py:47-55)
  cyc_sqr  Granger-Scott squaring in the cyclotomic subgroup on the pairs (f_0, f_3), (f_1, f_4), (f_2, f_5)

Plain integers mod q; tests/test_fexp_model.py pins it to the reference's vectors and the oracle.
"""
from .linestream_model import Q, NX, FLAT_POW, add2, sub2, mul2, scl2, xi2, mul_dense, one6


def conj2(x):
    return (x[0] % Q, (-x[1]) % Q)


def pow2(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = mul2(r, a)
        a = mul2(a, a)
        e >>= 1
    return r


def gamma(i, k):
    """xi^(k (q^i - 1)/6)"""
    return pow2((1, 1), k * (Q ** i - 1) // 6)


GAMMA = {i: [gamma(i, k) for k in range(6)] for i in (1, 2, 4)}


def conj6(f):
    return [c if k % 2 == 0 else ((-c[0]) % Q, (-c[1]) % Q) for k, c in enumerate(f)]


def frob(f, i):
    return [mul2(conj2(c) if i % 2 else c, GAMMA[i][k]) for k, c in enumerate(f)]


def fq_inv(a):
    return pow(a, Q - 2, Q)                       # 0 -> 0


def inverse(f):
    fb = conj6(f)
    N = mul_dense(f, fb)                          # in Fq6: odd coefficients vanish
    assert all(N[k] == (0, 0) for k in (1, 3, 5))
    N2, N4 = frob(N, 2), frob(N, 4)
    M = mul_dense(N2, N4)
    t = mul_dense(N, M)                           # in Fq2: only coefficient 0
    assert all(t[k] == (0, 0) for k in range(1, 6))
    n = (t[0][0] * t[0][0] + t[0][1] * t[0][1]) % Q
    ni = fq_inv(n)
    ti = (t[0][0] * ni % Q, (-t[0][1]) * ni % Q)
    return [mul2(c, ti) for c in mul_dense(fb, M)]


CYC_PAIR = (0, 2, 1, 0, 2, 1)                     # lane k squares the pair (f_p, f_{p+3}), p = CYC_PAIR[k]


def cyc_sqr(f):
    """tower.f12_cyclo_sqr in the w-power order: even lanes x^2 + xi y^2, result 3 t - 2 f_k; odd lanes 2 x y
    (times xi on lane 1), result 3 t + 2 f_k"""
    out = []
    for k in range(6):
        x, y = f[CYC_PAIR[k]], f[CYC_PAIR[k] + 3]
        if k % 2 == 0:
            t = add2(mul2(x, x), xi2(mul2(y, y)))
            out.append(sub2(scl2(t, 3), scl2(f[k], 2)))
        else:
            t = scl2(mul2(x, y), 2)
            if k == 1:
                t = xi2(t)
            out.append(add2(scl2(t, 3), scl2(f[k], 2)))
    return out


def cyc_sqr_lane_forms(f):
    """the same with every part written as the sum of three products the kernel evaluates (fp28_dot3): pins the
    operand table of csrc/blsgpu_fexp.hip"""
    out = []
    for k in range(6):
        (xr, xi_), (yr, yi) = f[CYC_PAIR[k]], f[CYC_PAIR[k] + 3]
        if k % 2 == 0:
            re = (xr + xi_) * (xr - xi_) + (yr - yi) * (yr - yi) + (-2 * yi) * yi
            im = (2 * xr) * xi_ + (yr + yi) * (yr + yi) + (-2 * yi) * yi
            sgn = -1
        elif k == 1:
            re = (2 * xr) * (yr - yi) + (-2 * xi_) * (yi + yr)
            im = (2 * xr) * (yr + yi) + (2 * xi_) * (yr - yi)
            sgn = 1
        else:
            re = (2 * xr) * yr + (-2 * xi_) * yi
            im = (2 * xr) * yi + (2 * xi_) * yr
            sgn = 1
        out.append(((3 * re + sgn * 2 * f[k][0]) % Q, (3 * im + sgn * 2 * f[k][1]) % Q))
    return out


def pow_cyc(f, e):
    """f^e by squarings in the cyclotomic subgroup (programs.final_exp_script pow_acc)"""
    base, acc = f, f
    for bit in range(e.bit_length() - 2, -1, -1):
        acc = cyc_sqr(acc)
        if (e >> bit) & 1:
            acc = mul_dense(acc, base)
    return acc


def final_exp(f):
    """f (w-power order) -> f^((q^12 - 1)/n)"""
    e1 = (NX + 1) // 3
    t = mul_dense(conj6(f), inverse(f))           # f^(q^6 - 1)
    t = mul_dense(frob(t, 2), t)                  # cyclotomic from here on
    s = pow_cyc(t, e1)
    a = mul_dense(pow_cyc(s, NX), s)
    b = mul_dense(conj6(pow_cyc(a, NX)), frob(a, 1))
    c = mul_dense(mul_dense(pow_cyc(pow_cyc(b, NX), NX), frob(b, 2)), conj6(b))
    return mul_dense(c, t)


def from_flat12(v):
    f = [None] * 6
    for i, p in enumerate(FLAT_POW):
        f[p] = (v[2 * i] % Q, v[2 * i + 1] % Q)
    return f


def to_flat12(f):
    out = []
    for p in FLAT_POW:
        out += [f[p][0], f[p][1]]
    return out


# ---- the kernel's script: one accumulator, a few memory slots ----------------------------------------------------------
# (op, arg): MUL s: acc *= M[s] | CSQ n: acc <- acc^(2^n) (cyclotomic) | ST s | LD s | CONJ | FROB j (q^1, q^2, q^4 for
# j = 0, 1, 2) | TINV: acc (an Fq2 value in coefficient 0) <- its inverse | END
END, MUL, CSQ, ST, LD, CONJ, FROB, TINV = range(8)
FROB_POW = (1, 2, 4)
NSLOTS = 5


def script():
    S = []

    def pow_acc(e):
        """acc <- acc^e, base kept in slot 1"""
        S.append((ST, 1))
        run = 0
        for bit in range(e.bit_length() - 2, -1, -1):
            run += 1
            if (e >> bit) & 1:
                S.append((CSQ, run))
                S.append((MUL, 1))
                run = 0
        if run:
            S.append((CSQ, run))

    def pow_win3(e):
        """acc <- acc^e with sliding windows of three bits (round 4): (x - 1)/3 = 0b100011 0^10 (10)^12 11 0 (10)^5 11 has
        28 set bits -- 27 dense products bit by bit, 14 + 2 with the odd powers b, b^3, b^5 (the only window values that
        occur) kept in slots 1, 0, 2 (free at this point of the script; b^2 passes through slot 4) and one more squaring"""
        slot_of = {1: 1, 3: 0, 5: 2}
        bits = bin(e)[2:]
        wins, i = [], 0
        while i < len(bits):
            if bits[i] == "0":
                i += 1
                continue
            n = min(3, len(bits) - i)
            while bits[i + n - 1] == "0":
                n -= 1
            wins.append((i, n, int(bits[i:i + n], 2)))
            i += n
        assert {v for _, _, v in wins} <= set(slot_of)
        S.extend([(ST, 1), (CSQ, 1), (ST, 4), (MUL, 1), (ST, 0), (MUL, 4), (ST, 2)])      # b, b^2, b^3, b^5
        pos, n, v = wins[0]
        assert pos == 0
        S.append((LD, slot_of[v]))
        done = n
        for pos, n, v in wins[1:]:
            S.append((CSQ, pos + n - done))
            S.append((MUL, slot_of[v]))
            done = pos + n
        if done < len(bits):
            S.append((CSQ, len(bits) - done))
    # easy part: f^-1 through norms, then f^(q^6 - 1), then (.)^(q^2 + 1)
    S += [(ST, 0), (CONJ, 0), (ST, 1), (MUL, 0), (ST, 2),        # M0 = f, M1 = conj f, M2 = N = f conj f
          (FROB, 1), (ST, 3), (LD, 2), (FROB, 2), (MUL, 3), (ST, 3),   # M3 = N^(q^2) N^(q^4)
          (MUL, 2), (TINV, 0), (MUL, 3), (MUL, 1),                # acc = f^-1
          (MUL, 1),                                               # acc = conj(f) f^-1
          (ST, 2), (FROB, 1), (MUL, 2), (ST, 3)]                  # acc = M3 = t
    e1 = (NX + 1) // 3
    pow_win3(e1)                                                  # s
    pow_acc(NX)
    S += [(MUL, 1), (ST, 4)]                                      # a
    pow_acc(NX)
    S += [(CONJ, 0), (ST, 2), (LD, 4), (FROB, 0), (MUL, 2), (ST, 4)]   # b = conj(a^|x|) frob1(a)
    pow_acc(NX)
    pow_acc(NX)
    S += [(ST, 2), (LD, 4), (FROB, 1), (MUL, 2), (ST, 2), (LD, 4), (CONJ, 0), (MUL, 2),   # c
          (MUL, 3), (END, 0)]
    assert all(0 <= a < 256 for _, a in S)
    return S


def run_script(f):
    """interpreter with the kernel's semantics (csrc/blsgpu_fexp.hip k_fexp_team)"""
    M = [None] * NSLOTS
    acc = list(f)
    for op, a in script():
        if op == MUL:
            acc = mul_dense(acc, M[a])
        elif op == CSQ:
            for _ in range(a):
                acc = cyc_sqr_lane_forms(acc)
        elif op == ST:
            M[a] = list(acc)
        elif op == LD:
            acc = list(M[a])
        elif op == CONJ:
            acc = conj6(acc)
        elif op == FROB:
            acc = frob(acc, FROB_POW[a])
        elif op == TINV:
            t = acc[0]
            ni = fq_inv((t[0] * t[0] + t[1] * t[1]) % Q)
            acc = [(t[0] * ni % Q, (-t[1]) * ni % Q)] + [(0, 0)] * 5
        else:
            break
    return acc
