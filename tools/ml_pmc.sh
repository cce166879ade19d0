# PMC passes over the line-stream kernels (gpurun): SQ counters, then FETCH/WRITE sizes.  OUT=gpurun_out/ml_pmc
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${OUT:-gpurun_out/ml_pmc}
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/sq -o s --output-format csv -- python3 tools/ml_check.py --time-only > $O/sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 tools/ml_check.py --time-only > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 tools/ml_check.py --time-only > $O/write.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get("OUT", "gpurun_out/ml_pmc")
for d in ("sq", "fetch", "write"):
    for path in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_ml_" in k:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            for c, x in v.items():
                print("%s,%s,%d,%.6g" % (k, c, len(n[k]), x / len(n[k])))
PY
