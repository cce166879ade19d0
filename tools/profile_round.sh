# Runs on the GPU box (gpurun): the round's bench lines, rocprofv3 kernel stats and PMC passes.
# ROUND=r05 PART=A bash tools/profile_round.sh ; ROUND=r05 PART=B bash tools/profile_round.sh  (two gpurun calls: a call is limited to
# 20 minutes) ; then ROUND=r05 python tools/collect_profiles.py here copies the summaries into profiles/.
set -e
R=${ROUND:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
mkdir -p $O
PART=${PART:-AB}
if [[ $PART == *A* ]]; then
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
timeout -k 10 200 python bench.py --gpus 2 --backend gloo --steps 8 --warmup 2 > $O/bench_2rank_gloo_rehearsal.json 2> $O/bench_2rank.err
timeout -k 10 200 python bench.py --force-process-group --no-cpu-baseline --no-latency > $O/bench_rccl_1rank_rehearsal.json 2> $O/bench_rccl_1rank.err
timeout -k 10 200 python bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
timeout -k 10 200 python bench.py --config c3 --gpus 2 --backend gloo --steps 8 --warmup 2 > $O/bench_c3_2rank_gloo.json 2> $O/bench_c3_2rank.err
for c in c4 c5 h2c; do timeout -k 10 300 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; done
echo "config lines done"
# the timed region ALONE under the kernel trace (no latency probes, no secondary configs): per-kernel averages of this CSV are the bench's
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-latency --no-secondary > $O/kt_bench.json 2> $O/kt.err
for c in c4 c5 h2c; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$c -o kt --output-format csv -- python3 bench.py --config $c --steps 2 > $O/kt_$c.json 2> $O/kt_$c.err
done
echo "kernel traces done"
fi
if [[ $PART == *B* ]]; then
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --no-latency --no-secondary > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --no-latency --no-secondary > $O/pmc_write.json 2> $O/pmc_write.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/pmc_sq -o s --output-format csv -- python3 bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline --no-latency --no-secondary > $O/pmc_sq.json 2> $O/pmc_sq.err
for c in c4 c5 h2c; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS -d $O/pmc_$c -o p --output-format csv -- python3 bench.py --config $c --steps 1 > $O/pmc_$c.json 2> $O/pmc_$c.err || echo "pmc $c failed"
done
# HBM traffic of the secondary configs: FETCH_SIZE / WRITE_SIZE in passes of their own, cut to the timed region by the marks
for c in c4 c5 h2c; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_${c}_$ctr -o p --output-format csv -- python3 bench.py --config $c --steps 2 > $O/pmc_${c}_$ctr.json 2> $O/pmc_${c}_$ctr.err || echo "pmc $c $ctr failed"
  done
done
echo "pmc done"
timeout -k 10 200 python tools/miller_wide_probe.py > $O/miller_wide_probe.jsonl 2> $O/miller_wide_probe.err || echo "wide probe failed"
timeout -k 10 200 python tools/sweep_n.py > $O/sweep_n.jsonl 2> $O/sweep_n.err || echo "sweep failed"
timeout -k 10 200 python tools/fexp_latency.py > $O/fexp_latency.jsonl 2> $O/fexp_latency.err || echo "fexp latency failed"
timeout -k 10 100 python tools/fexpw_stamps.py > $O/fexpw_stamps.json 2> $O/fexpw_stamps.err || echo "stamps failed"
timeout -k 10 100 python tools/h2c_sweep.py 16384 65536 262144 > $O/h2c_sweep.jsonl 2> $O/h2c_sweep.err || echo "h2c sweep failed"
timeout -k 10 200 python tools/pipeline_rate.py 256 > $O/pipeline_rate.json 2> $O/pipeline_rate.err || echo "pipeline failed"
fi
cat $O/bench_default.json
