"""Integer model of the LINE-STREAM multi-pairing (round 3; csrc/blsgpu_ml.hip).

The wavefront VM keeps a Miller accumulator per team and pays for it in linear-combination rounds.  The line-stream
form cuts the Miller loop of fields_t.py:1091-1111 into two data-parallel halves joined through HBM:

  A  (one pair per lane)   the twist-point chain T <- 2T (+ Q) alone, which does not depend on the accumulator; every
                           step writes its LINE l = l0 + l2 w^2 + l3 w^3 (three Fq2 values, already scaled by P).
  B  (six lanes per accumulator)  for every line index L the product of that line over the pairs of a group -- no
                           squarings, no dependency between steps: prod_i l_{i,L}.
  H  (Horner)              f = prod_s (M_s)^(2^(62 - s)) by  f <- f^2 M_s, M_s = tangent product x chord product of step s.

f is the same field element as the reference's loop gives up to the line scalings (Fq2 factors and w^3 per line,
programs.py header), so the final exponentiation returns the reference's bytes.

Basis used by B and H: Fq12 = Fq2[w] / (w^6 - xi), an element is (f_0 .. f_5), f = sum f_k w^k.  The reference's flat
order (fields.py:624-629, vmgen/tower.flat12) lists the w-powers 0, 2, 4, 1, 3, 5.

Everything here is plain integers mod q; tests/test_linestream_model.py checks it against the oracle and the golden
vectors, lane for lane the way the kernels compute (wrap selection, published values, term order).
"""
Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
NX = 0xd201000000010000                   # |x|  (fields_t.py:25)
FLAT_POW = (0, 2, 4, 1, 3, 5)             # flat Fq2 index -> power of w
LINE_POS = (0, 2, 3)                      # powers of w a line occupies


# ---- Fq2 -------------------------------------------------------------------------------------------------------
def f2(a, b=0):
    return (a % Q, b % Q)


def add2(x, y):
    return ((x[0] + y[0]) % Q, (x[1] + y[1]) % Q)


def sub2(x, y):
    return ((x[0] - y[0]) % Q, (x[1] - y[1]) % Q)


def mul2(x, y):
    return ((x[0] * y[0] - x[1] * y[1]) % Q, (x[0] * y[1] + x[1] * y[0]) % Q)


def scl2(x, k):
    return (x[0] * k % Q, x[1] * k % Q)


def xi2(x):
    """(1 + u) x"""
    return ((x[0] - x[1]) % Q, (x[0] + x[1]) % Q)


# ---- stage A: the point chain and its lines ---------------------------------------------------------------------
def line_schedule():
    """[(step, kind)] in execution order: 63 tangent lines, a chord line after the tangent of a set bit: 68 lines"""
    out = []
    for s, bit in enumerate(range(NX.bit_length() - 2, -1, -1)):
        out.append((s, "t"))
        if (NX >> bit) & 1:
            out.append((s, "c"))
    return out


def tangent(T, px3n, py):
    """programs.t_double: X3 = 2XY(B - F), Y3 = (B + F)^2 - 12 E^2, Z3 = 4 B H; line (B - E, XX (-3 px), H py)"""
    X, Y, Z = T
    A = mul2(X, Y)
    B = mul2(Y, Y)
    C = mul2(Z, Z)
    XX = mul2(X, X)
    H = scl2(mul2(Y, Z), 2)
    E = scl2(xi2(C), 12)
    F = scl2(E, 3)
    X3 = scl2(mul2(A, sub2(B, F)), 2)
    G = add2(B, F)
    Y3 = sub2(mul2(G, G), scl2(mul2(E, E), 12))
    Z3 = scl2(mul2(B, H), 4)
    return (X3, Y3, Z3), (sub2(B, E), scl2(XX, px3n), scl2(H, py))


def chord(T, Qa, px3n, py):
    """programs.t_add with px_is_m3: line times 3 = (3 (th xq - la yq), th (-3 px), 3 la py)"""
    X, Y, Z = T
    xq, yq = Qa
    th = sub2(Y, mul2(yq, Z))
    la = sub2(X, mul2(xq, Z))
    C = mul2(th, th)
    D = mul2(la, la)
    E = mul2(la, D)
    Fz = mul2(Z, C)
    G = mul2(X, D)
    H = sub2(add2(E, Fz), scl2(G, 2))
    X3 = mul2(la, H)
    Y3 = sub2(mul2(th, sub2(G, H)), mul2(E, Y))
    Z3 = mul2(Z, E)
    l0 = scl2(sub2(mul2(th, xq), mul2(la, yq)), 3)
    return (X3, Y3, Z3), (l0, scl2(th, px3n), scl2(la, 3 * py))


def pair_lines(P, Qa):
    """the 68 lines of one pair and the validity of the fast formulas (Q on the twist, final Z != 0)"""
    px, py = P
    xq, yq = Qa
    px3n = (-3 * px) % Q
    d = sub2(sub2(mul2(yq, yq), mul2(mul2(xq, xq), xq)), (4, 4))
    T = (xq, yq, (1, 0))
    lines = []
    for s, kind in line_schedule():
        if kind == "t":
            T, l = tangent(T, px3n, py)
        else:
            T, l = chord(T, Qa, px3n, py)
        lines.append(l)
    ok = d == (0, 0) and T[2] != (0, 0)
    return lines, ok


# ---- stage B: products in the w-power basis, the way a team of six lanes computes them ---------------------------
def one6():
    return [(1, 0)] + [(0, 0)] * 5


def line_to_dense(l):
    f = [(0, 0)] * 6
    for pos, c in zip(LINE_POS, l):
        f[pos] = c
    return f


def mul_terms(f, terms):
    """lane k: c_k = sum over (j, y_j) of F_{k - j} y_j with F_i = f_i (i >= 0), xi f_{i + 6} (i < 0): every lane
    publishes f_k and xi f_k, fetches both from lane (k - j) mod 6 and keeps the second iff j > k"""
    pub, pub_xi = list(f), [xi2(c) for c in f]
    out = []
    for k in range(6):
        acc = (0, 0)
        for j, y in terms:
            src = (k - j) % 6
            x = pub_xi[src] if j > k else pub[src]
            acc = add2(acc, mul2(x, y))
        out.append(acc)
    return out


def mul_sparse(f, l):
    return mul_terms(f, list(zip(LINE_POS, l)))


def mul_dense(f, g):
    return mul_terms(f, list(enumerate(g)))


def group_line_products(lines_of_pairs, chunk):
    """B: for every line index the products over chunks of `chunk` pairs (the first line of a chunk seeds the accumulator),
    then the merge over the chunks (dense products)"""
    n = len(lines_of_pairs)
    prods = []
    for L in range(len(line_schedule())):
        parts = []
        for c0 in range(0, n, chunk):
            acc = line_to_dense(lines_of_pairs[c0][L])
            for i in range(c0 + 1, min(n, c0 + chunk)):
                acc = mul_sparse(acc, lines_of_pairs[i][L])
            parts.append(acc)
        m = parts[0] if parts else one6()
        for p in parts[1:]:
            m = mul_dense(m, p)
        prods.append(m)
    return prods


def horner(prods):
    """H: f <- f^2 before every tangent product but the first, f <- f M for every line index"""
    f = None
    for (s, kind), m in zip(line_schedule(), prods):
        if f is None:
            f = m
            continue
        if kind == "t":
            f = mul_dense(f, f)
        f = mul_dense(f, m)
    return f


def to_flat12(f):
    """w-power order -> the reference's flat 12-tuple (fields.py:624-629)"""
    out = []
    for p in FLAT_POW:
        out += [f[p][0], f[p][1]]
    return out


def miller_product(pairs, chunk=4):
    """pairs: [((px, py), ((xq0, xq1), (yq0, yq1)))] all valid for the fast formulas -> flat 12-tuple of a value with the
    same final exponentiation as prod fq_miller_loop(P_i, Q_i)"""
    if not pairs:
        return to_flat12(one6())
    all_lines = []
    for P, Qa in pairs:
        lines, ok = pair_lines(P, Qa)
        assert ok, "degenerate pair: the slow program's business"
        all_lines.append(lines)
    return to_flat12(horner(group_line_products(all_lines, chunk)))


# ---- the reference-faithful lines (degenerate pairs) ---------------------------------------------------------------------
# The reference (fields_t.py:1035-1078, 641-686) is a total function of the coordinates: affine formulas with 0^-1 := 0
# and three branches in the chord step (vmgen/slow_programs.py spells them out).  Its line values are sparse in the
# w-power basis too: a tangent line is  py + c3 w^3 + c5 w^5  (c0 = py in Fq; 1/w^3 = xi^-1 w^3, 1/w = xi^-1 w^5), a
# chord line the same or, on the "vertical" branch,  px + c4 w^4  (1/w^2 = xi^-1 w^4).  k_ml_lines_exact writes them
# into the pair's line records, k_ml_accum multiplies them in like any other line, and the product over a group is the
# reference's own Miller value times the scaled lines of the ordinary pairs.
def inv_fq(a):
    return pow(a % Q, Q - 2, Q)                    # 0 -> 0


def inv2(z):
    ni = inv_fq(z[0] * z[0] + z[1] * z[1])
    return (z[0] * ni % Q, (-z[1]) * ni % Q)


def xi_inv2(a):
    """a / (1 + u) = a (1 - u) / 2"""
    half = (Q + 1) // 2
    return ((a[0] + a[1]) * half % Q, (a[1] - a[0]) * half % Q)


def neg2(a):
    return ((-a[0]) % Q, (-a[1]) % Q)


def exact_double(rx, ry):
    lam = mul2(scl2(mul2(rx, rx), 3), inv2(scl2(ry, 2)))
    xr = sub2(mul2(lam, lam), scl2(rx, 2))
    return lam, (xr, sub2(mul2(lam, sub2(rx, xr)), ry))


def exact_pair_lines(P, Qa, qinf=False):
    """[(positions, coefficients)] for the 68 lines of fq_miller_loop(P, Q): coefficient 0 is (c, 0) with c in Fq"""
    px, py = P
    qx, qy = Qa
    rx, ry = qx, qy
    out = []
    for s, kind in line_schedule():
        if kind == "t":
            lam, Rn = exact_double(rx, ry)
            l0 = sub2(mul2(lam, rx), ry)
            l1 = neg2(scl2(lam, px))
            out.append(((0, 3, 5), ((py % Q, 0), xi_inv2(l0), xi_inv2(l1))))
            rx, ry = Rn
        else:
            d, u = sub2(qx, rx), sub2(ry, qy)
            D = inv2(d)
            n1 = d != (0, 0)
            same = (not n1) and u == (0, 0)
            vert = add2(rx, qx) == (0, 0) and add2(ry, qy) == (0, 0)
            mu = mul2(neg2(u), D)
            nu = neg2(mul2(sub2(mul2(qy, rx), mul2(ry, qx)), D))
            if vert:
                out.append(((0, 4, 5), ((px % Q, 0), xi_inv2(neg2(rx)), (0, 0))))
            else:
                out.append(((0, 3, 5), ((py % Q, 0), xi_inv2(neg2(nu)), xi_inv2(neg2(scl2(mu, px))))))
            xc = sub2(sub2(mul2(mu, mu), rx), qx)
            yc = sub2(mul2(mu, sub2(rx, xc)), ry)
            _, (xd, yd) = exact_double(rx, ry)
            new = (xc, yc) if n1 else ((xd, yd) if same else ((0, 0), (0, 0)))
            if not qinf:
                rx, ry = new
    return out


def exact_miller(P, Qa, qinf=False):
    """fq_miller_loop(P, Q) itself (w-power order) as Horner over the exact lines"""
    f = None
    for (s, kind), (pos, cf) in zip(line_schedule(), exact_pair_lines(P, Qa, qinf)):
        m = [(0, 0)] * 6
        for p, c in zip(pos, cf):
            m[p] = c
        if f is None:
            f = m
            continue
        if kind == "t":
            f = mul_dense(f, f)
        f = mul_dense(f, m)
    return f
