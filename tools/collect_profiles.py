"""Copy the summaries of gpurun_out/r01 (written by tools/profile_round.sh on the GPU box)
into profiles/ under round-1 names."""
import collections, csv, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", "r01")
P = os.path.join(ROOT, "profiles")


def json_line(path):
    for line in open(path):
        if line.startswith("{"):
            return line
    raise SystemExit("no JSON line in " + path)


shutil.copy(os.path.join(R, "kt", "kt_kernel_stats.csv"), os.path.join(P, "r01_bench_kernel_stats.csv"))
for src, dst in (("bench_default.json", "r01_bench_line.json"), ("kt_bench.json", "r01_bench_line_under_rocprof.json"),
                 ("bench_2rank_gloo_rehearsal.json", "r01_bench_line_2rank_gloo_rehearsal_on_one_gpu.json")):
    with open(os.path.join(P, dst), "w") as f:
        f.write(json_line(os.path.join(R, src)))
shutil.copy(os.path.join(R, "configs.jsonl"), os.path.join(P, "r01_configs_c3_c4_c5_h2c.jsonl"))
rows_out = []
for p, f in (("pmc_fetch", "f"), ("pmc_write", "w"), ("pmc_sq", "s")):
    rows = list(csv.DictReader(open(os.path.join(R, p, f + "_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        if "blsgpu" in k:
            for c, x in v.items():
                rows_out.append((k, c, len(n[k]), x / len(n[k])))
with open(os.path.join(P, "r01_bench_pmc_summary.csv"), "w") as f:
    f.write("kernel,counter,dispatches,average_per_dispatch\n")
    for k, c, d, x in rows_out:
        f.write("%s,%s,%d,%.6g\n" % (k, c, d, x))
print(open(os.path.join(P, "r01_bench_pmc_summary.csv")).read())
