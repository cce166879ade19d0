"""Per-kernel timeline from a rocprofv3 --kernel-trace database (rocpd sqlite): for every run of consecutive launches
that ends in a final-exponentiation kernel, the kernels with their durations and the gaps between them.
usage: kernel_timeline.py results.db [how many sequences to print per distinct shape]"""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute("select k.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y from %s d join %s k on d.kernel_id = k.id order by d.start" % (disp, sym)).fetchall()
def short(n):
    m = re.search(r"k_[a-z0-9_]+", n)
    return m.group(0) if m else n[:32]
seq, seqs = [], []
for n, s, e, gx, gy in rows:
    seq.append((short(n), s, e, gx * gy))
    if short(n) in ("k_fexp_wide", "k_fexp_team", "k_final_groups", "k_ml_horner_fexp"):
        seqs.append(seq); seq = []
seen = collections.Counter()
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for sq in seqs:
    sq = [x for x in sq if not x[0].startswith("__amd")]
    key = tuple((n, g) for n, _, _, g in sq)
    seen[key] += 1
    if seen[key] != 3 or len(sq) < 2:          # the third occurrence: warmed up
        continue
    base = sq[0][1]
    print("---- %d kernels, %.1f us from first start to last end" % (len(sq), (sq[-1][2] - base) / 1e3))
    prev_end = None
    for n, s, e, g in sq:
        gap = "" if prev_end is None else "  gap %6.1f" % ((s - prev_end) / 1e3)
        print("%-22s grid %9d  start %9.1f  dur %8.1f us%s" % (n, g, (s - base) / 1e3, (e - s) / 1e3, gap))
        prev_end = e
