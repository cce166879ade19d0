"""Checks the one-product dumps of fqmul_radix against Python integers (stdin: the program's output)."""
import sys
Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
ok = True
for line in sys.stdin:
    if not line.startswith("dump"):
        print(line.rstrip())
        continue
    head, vals = line.split(":")
    W = int(head.split("W=")[1].split()[0]); L = int(head.split("L=")[1])
    v = [int(x, 16) for x in vals.split()]
    mask = (1 << W) - 1
    a = [((0x12345678 * (j + 1) + 0x9abcdef) & 0xFFFFFFFF) & mask for j in range(L)]
    b = [((0x0fedcba9 * (j + 3) + 0x7654321) & 0xFFFFFFFF) & mask for j in range(L)]
    a[L - 1] &= 0xfffff; b[L - 1] &= 0xfffff
    A = sum(x << (W * j) for j, x in enumerate(a)); B = sum(x << (W * j) for j, x in enumerate(b))
    got = sum(x << (W * j) for j, x in enumerate(v[:L]))          # after one step: a <- a*b/R, b <- a
    want = A * B * pow(1 << (W * L), -1, Q) % Q
    good = got % Q == want and got < 2 * Q and sum(x << (W * j) for j, x in enumerate(v[L:])) == A
    ok = ok and good
    print("%s: product %s (result < 2q: %s)" % (head, "matches Python" if got % Q == want else "WRONG", got < 2 * Q))
sys.exit(0 if ok else 1)
