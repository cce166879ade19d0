"""Copy the summaries of gpurun_out/<round> (written by tools/profile_round.sh on the GPU box)
into profiles/ under round names.  ROUND=r02 python tools/collect_profiles.py"""
import collections, csv, glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RN = os.environ.get("ROUND", "r05")
R = os.path.join(ROOT, "gpurun_out", RN)
P = os.path.join(ROOT, "profiles")


def json_line(path):
    for line in open(path):
        if line.startswith("{"):
            return line
    raise SystemExit("no JSON line in " + path)


def find(d, pat):
    hits = glob.glob(os.path.join(R, d, "**", pat), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s" % (pat, d))
    return hits[0]


shutil.copy(find("kt", "*kernel_stats.csv"), os.path.join(P, RN + "_bench_kernel_stats.csv"))
for c in ("c4", "c5", "h2c"):
    shutil.copy(find("kt_" + c, "*kernel_stats.csv"), os.path.join(P, "%s_%s_kernel_stats.csv" % (RN, c)))
for src, dst in (("bench_default.json", "_bench_line.json"), ("kt_bench.json", "_bench_line_under_rocprof.json"),
                 ("bench_2rank_gloo_rehearsal.json", "_bench_line_2rank_gloo_rehearsal_on_one_gpu.json"),
                 ("bench_rccl_1rank_rehearsal.json", "_bench_line_rccl_1rank_rehearsal.json"),
                 ("bench_c3.json", "_bench_c3_line.json"), ("bench_c3_2rank_gloo.json", "_bench_c3_line_2rank_gloo_rehearsal_on_one_gpu.json")):
    with open(os.path.join(P, RN + dst), "w") as f:
        f.write(json_line(os.path.join(R, src)))
with open(os.path.join(P, RN + "_configs_c4_c5_h2c.jsonl"), "w") as f:
    for c in ("c4", "c5", "h2c"):
        f.write(json_line(os.path.join(R, "bench_%s.json" % c)))
for src, dst in (("sweep_n.jsonl", "_scaling_curve_single_call.jsonl"), ("fexp_latency.jsonl", "_fexp_forms_latency.jsonl"),
                 ("fexpw_stamps.json", "_fexp_wide_cycles.json"), ("h2c_sweep.jsonl", "_h2c_sweep.jsonl"),
                 ("pipeline_rate.json", "_verify_pipeline_rate.json")):
    if os.path.exists(os.path.join(R, src)) and os.path.getsize(os.path.join(R, src)):
        shutil.copy(os.path.join(R, src), os.path.join(P, RN + dst))


def pmc_rows(dirs):
    rows_out = []
    for p in dirs:
        try:
            path = find(p, "*counter_collection.csv")
        except SystemExit:
            continue
        rows = list(csv.DictReader(open(path)))
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
    return rows_out


for name, dirs in ((RN + "_bench_pmc_summary.csv", ("pmc_fetch", "pmc_write", "pmc_sq")),
                   (RN + "_configs_pmc_summary.csv", ("pmc_c4", "pmc_c5", "pmc_h2c"))):
    with open(os.path.join(P, name), "w") as f:
        f.write("kernel,counter,dispatches,average_per_dispatch\n")
        for k, c, d, x in pmc_rows(dirs):
            f.write("%s,%s,%d,%.6g\n" % (k, c, d, x))
# the line-stream stage's rows once more, stamped with the build and the launch size they were measured on: bench.py
# reports roofline.traffic only when both match what it runs (never a stale figure)
import sys
sys.path.insert(0, ROOT)
import bench
pairs = json.loads(json_line(os.path.join(R, "pmc_fetch.json")))["config"]["pairs_per_step_per_gpu"]
with open(os.path.join(P, RN + "_ls_pmc_summary.csv"), "w") as f:
    f.write("# build %s pairs %d  (rocprofv3 --pmc, separate passes; KB per dispatch for FETCH_SIZE / WRITE_SIZE)\n" % (bench.source_hash(), pairs))
    f.write("kernel,counter,dispatches,average_per_dispatch\n")
    for k, c, d, x in pmc_rows(("pmc_fetch", "pmc_write", "pmc_sq")):
        if "k_ml_" in k:
            f.write("%s,%s,%d,%.6g\n" % (k.replace("void ", ""), c, d, x))
print(open(os.path.join(P, RN + "_ls_pmc_summary.csv")).read())


# HBM traffic per step of the secondary configs: counters of the dispatches BETWEEN the timed region's two marks
# (blsgpu::probe::k_mark, bench.py), divided by the steps of that run; stamped with the build like the line-stream rows
def region_total(d, counter):
    rows = [r for r in csv.DictReader(open(find(d, "*counter_collection.csv"))) if r["Counter_Name"] == counter]
    marks = sorted(int(r["Dispatch_Id"]) for r in rows if "k_mark" in r["Kernel_Name"])
    if len(marks) < 2:
        raise SystemExit("no marks in " + d)
    lo, hi = marks[-2], marks[-1]
    return sum(float(r["Counter_Value"]) for r in rows if lo < int(r["Dispatch_Id"]) < hi)


with open(os.path.join(P, RN + "_configs_traffic.csv"), "w") as f:
    f.write("# build %s  (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --config <name> --steps 2`; KB per STEP, "
            "summed over the dispatches between the timed region's marks)\n" % bench.source_hash())
    f.write("config,counter,steps,kb_per_step\n")
    for c in ("c4", "c5", "h2c"):
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = "pmc_%s_%s" % (c, ctr)
            try:
                steps = json.loads(json_line(os.path.join(R, d + ".json")))["steps"]
                f.write("%s,%s,%d,%.6g\n" % (c, ctr, steps, region_total(d, ctr) / steps))
            except (SystemExit, OSError, KeyError) as e:
                print("no traffic for", c, ctr, e)
print(open(os.path.join(P, RN + "_configs_traffic.csv")).read())
for src, dst in (("miller_wide_probe.jsonl", "_miller_wide_probe.jsonl"),):
    if os.path.exists(os.path.join(R, src)) and os.path.getsize(os.path.join(R, src)):
        shutil.copy(os.path.join(R, src), os.path.join(P, RN + dst))
