set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
timeout -k 10 300 python bench.py > gpurun_out/r01/bench_default.json 2> gpurun_out/r01/bench_default.err
echo "bench done"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r01/bench_2rank_gloo_rehearsal.json 2> gpurun_out/r01/bench_2rank.err
echo "2-rank rehearsal done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01/kt -o kt --output-format csv -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r01/kt_bench.json 2> gpurun_out/r01/kt.err
echo "kernel trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r01/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/r01/pmc_fetch.json 2> gpurun_out/r01/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r01/pmc_write -o w --output-format csv -- python3 bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/r01/pmc_write.json 2> gpurun_out/r01/pmc_write.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d gpurun_out/r01/pmc_sq -o s --output-format csv -- python3 bench.py --steps 4 --warmup 1 --streams 1 --no-cpu-baseline > gpurun_out/r01/pmc_sq.json 2> gpurun_out/r01/pmc_sq.err
echo "pmc done"
timeout -k 10 300 python tools/bench_configs.py > gpurun_out/r01/configs.jsonl 2> gpurun_out/r01/configs.err
echo "configs done"
cat gpurun_out/r01/bench_default.json
cat gpurun_out/r01/bench_2rank_gloo_rehearsal.json
cat gpurun_out/r01/configs.jsonl
