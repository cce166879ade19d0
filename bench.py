#!/usr/bin/env python3
"""bench.py -- BLS12-381 pairings/sec through the HIP multi-pairing engine.

One "step" = one pass of the hot path over one batch that is already resident in
HBM: B independent aggregate verifications (default 32), each the BASELINE.json
configs[1] shape -- fq_ate_pairing_multi over 1025 (pk, H(m)) pairs per GPU (1024
signatures + the (-G1, aggregate) pair): Miller loops, Fq12 product, its own
final exponentiation.  One launch sequence per step (blsgpu_miller_product_batch_dev
+ blsgpu_final_exp_product_batch_dev, the sharded form of blsgpu_pairing_multi_batch_dev).
With N ranks every rank holds 1025 pairs of EACH verification (weak scaling):
Miller products per rank, one RCCL all-gather of the B x 576-byte Fq12 partials,
B final exponentiations on every rank.  A single verification alone is latency
bound (one wavefront runs its final exponentiation); its latency is reported in
"single_verification".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "python-bls_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

PAIRS_PER_GPU = 1025
VERIFICATIONS_PER_STEP = 32
# algorithmic work, SURVEY.md section 8(d): 6754 Fq-mults per pairing at 300
# 32-bit MACs each, plus ~9.5k Fq-mults per final exponentiation
MAC_PER_PAIRING = 6754 * 300
MAC_PER_FINAL_EXP = 9500 * 300
HBM_BYTES_PER_PAIRING = 288
PEAK_TMACS = 34.65        # measured v_mad_u64_u32 rate, profiles/r01_intrate_microbench.txt
PEAK_HBM_GBS = 8000.0


def pmc_traffic(kernel, pairings_per_launch):
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 --pmc passes
    (FETCH_SIZE + WRITE_SIZE, KB per dispatch; profiles/r01_bench_pmc_summary.csv, collected by
    tools/profile_round.sh on this same workload).  Counters cannot be read inside this process,
    so the figure is only reported for the profiled shape (32 800 pairings per k_miller_mp launch);
    FETCH_SIZE is left uncorrected (dword table loads are outside the guide's x2 calibration)."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_summary.csv")
    if kernel != "k_miller_mp" or pairings_per_launch != 32800 or not os.path.exists(path):
        return None
    kb = {}
    with open(path) as f:
        for row in f:
            c = row.strip().split(",")
            if len(c) == 4 and c[0].endswith(kernel) and c[1] in ("FETCH_SIZE", "WRITE_SIZE"):
                kb[c[1]] = float(c[3])
    if len(kb) != 2:
        return None
    return int((kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024)


def cpu_baseline(g1, g2, n):
    """The oracle (CPU restatement of the reference's algorithm) on the host
    cores of this box; bounded sample = the same 1025-pair batch, once."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    O.build()
    cores = os.cpu_count() or 1
    t = time.perf_counter()
    out = O.pairing_multi(g1, g2, n, threads=cores)
    dt = time.perf_counter() - t
    return {"value": n / dt, "unit": "pairings/s", "cores": cores, "kind": "port",
            "sample": "one %d-pair multi-pairing (same batch), %.2f s wall" % (n, dt)}, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=PAIRS_PER_GPU, help="pairs per GPU per verification")
    ap.add_argument("--verifications", type=int, default=VERIFICATIONS_PER_STEP,
                    help="independent aggregate verifications per step (one launch sequence)")
    ap.add_argument("--streams", type=int, default=2,
                    help="steps in flight (the final exponentiations of one step overlap the Miller loops of the next)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: CPU-side collective, for rehearsing the multi-rank path on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from bls_py import _native

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    ndev = torch.cuda.device_count()
    local_dev = local % max(1, ndev)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    S = max(1, args.streams)
    B = max(1, args.verifications)
    engs = [_native.Engine(local_dev) for _ in range(S)]
    eng = engs[0]

    gold = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gold, "pairs_seed1_g1.bin"), "rb") as f:
        g1_all = f.read()
    with open(os.path.join(gold, "pairs_seed1_g2.bin"), "rb") as f:
        g2_all = f.read()
    n = args.pairs
    reps = (n + 1024) // 1025
    g1 = (g1_all * reps)[:96 * n]
    g2 = (g2_all * reps)[:192 * n]

    def shard(v, r):
        """Pairs of verification v held by rank r: a rotation of the seeded batch
        (different data per verification and per rank; a rotation keeps the product)."""
        rot = (v * 37 + r * 131) % n
        return g1[96 * rot:] + g1[:96 * rot], g2[192 * rot:] + g2[:192 * rot]
    mine = [shard(v, rank) for v in range(B)]
    t1 = torch.frombuffer(bytearray(b"".join(a for a, _ in mine)), dtype=torch.uint8).to(dev)
    t2 = torch.frombuffer(bytearray(b"".join(b for _, b in mine)), dtype=torch.uint8).to(dev)
    outs = [torch.zeros(B * 576, dtype=torch.uint8, device=dev) for _ in range(S)]
    parts = [torch.zeros(B * 144, dtype=torch.int32, device=dev) for _ in range(S)]
    gath = [torch.zeros(world * B * 144, dtype=torch.int32, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    for e in engs:
        e.reserve((n + 3) * B)
        e.set_mp_threshold(0 if n * B >= 2048 else 1 << 40)    # throughput kernel once the batch can fill the GPU

    # A step = Miller loops + per-verification products (stream k), then the B final
    # exponentiations.  Miller launches of consecutive steps are serialised with an
    # event (they would only stretch each other), so the final exponentiations of
    # step i (B wavefronts, latency bound) overlap the Miller loops of step i + 1.
    miller_done = [None]

    def step(i):
        k = i % S
        stream = streams[k]
        st = stream.cuda_stream
        if miller_done[0] is not None:
            stream.wait_event(miller_done[0])
        engs[k].miller_product_batch_dev(t1.data_ptr(), t2.data_ptr(), n, B, parts[k].data_ptr(), st)
        ev = torch.cuda.Event()
        ev.record(stream)
        miller_done[0] = ev
        if world == 1:
            engs[k].final_exp_product_batch_dev(parts[k].data_ptr(), 1, B, outs[k].data_ptr(), st)
        else:
            with torch.cuda.stream(stream):
                if args.backend == "nccl":
                    dist.all_gather_into_tensor(gath[k], parts[k])       # RCCL: B x 576 bytes per rank
                else:
                    stream.synchronize()
                    host = [torch.zeros(B * 144, dtype=torch.int32) for _ in range(world)]
                    dist.all_gather(host, parts[k].cpu())
                    gath[k].copy_(torch.cat(host).to(dev))
            engs[k].final_exp_product_batch_dev(gath[k].data_ptr(), world, B, outs[k].data_ptr(), st)

    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    for e in engs:
        e.timing_enable(True)         # HIP events around every kernel, on the stream it runs on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(streams[i % S])
        step(i)
        ev[i][1].record(streams[i % S])
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    results = [bytes(o.cpu().numpy()) for o in outs[:min(S, args.steps)]]
    step_ms = sorted(a.elapsed_time(b) for a, b in ev)
    ktimes = [t for e in engs for t in e.timing_read()]        # every kernel of the timed region
    for e in engs:
        e.timing_enable(False)
    miller_ms = [ms for k, ms in ktimes if k == 0]
    reduce_ms = [ms for k, ms in ktimes if k == 1]
    fexp_ms = [ms for k, ms in ktimes if k == 2]
    solo_out = torch.zeros(B * 576, dtype=torch.uint8, device=dev)
    # single verification alone on the GPU (latency view of the same workload)
    one_ms = []
    eng.set_mp_threshold(4096)
    for i in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(streams[0])
        eng.pairing_multi_dev(t1.data_ptr(), t2.data_ptr(), n, solo_out.data_ptr(), streams[0].cuda_stream)
        b.record(streams[0])
        streams[0].synchronize()
        one_ms.append(a.elapsed_time(b))
    assert all(r == results[0] for r in results), "streams disagree"
    result = results[0]

    if rank == 0:
        total_pairs = n * B * world * args.steps
        value = total_pairs / dt
        # correctness gate.  N=1 with the seeded 1025 pairs: every verification is a
        # rotation of the batch whose result the reference produced (committed golden
        # vector).  Other shapes: verification 0 against a single-GPU pass over the
        # concatenation of every rank's shard, and all B verifications the same power.
        check = "unchecked"
        per = [result[576 * v:576 * (v + 1)] for v in range(B)]
        if world == 1 and n == 1025:
            with open(os.path.join(gold, "pairing.json")) as f:
                want = bytes.fromhex(json.load(f)["seeded"]["1025"]["out"])
            check = "golden-ok" if all(p == want for p in per) else "MISMATCH"
        else:
            cat1 = b"".join(shard(0, r)[0] for r in range(world))
            cat2 = b"".join(shard(0, r)[1] for r in range(world))
            ok = eng.pairing_multi(cat1, cat2, n * world) == per[0] and all(p == per[0] for p in per)
            check = "single-gpu-ok" if ok else "MISMATCH"
        if check == "MISMATCH":
            raise SystemExit("result mismatch -- bench invalid")
        # dominant kernel = the Miller kernel (all the per-pairing work): algorithmic
        # MACs of one launch (B x n pairings) over its average duration in the timed
        # region (launches do not overlap each other, see step())
        miller_avg = sum(miller_ms) / len(miller_ms)
        ach = MAC_PER_PAIRING * n * B / (miller_avg * 1e-3) / 1e12
        kname = "k_miller_mp" if n * B >= 2048 else "k_miller"
        line = {
            "metric": "BLS12-381 pairings/sec (aggregate_verify multi-pairing)",
            "value": value, "unit": "pairings/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%d independent aggregate verifications per step, each a multi-pairing of %d (pk, H(m)) "
                                   "pairs per GPU with its own final exponentiation (BASELINE configs[1] shape x %d)" % (B, n, B),
                       "pairs_per_verification_per_gpu": n, "verifications_per_step": B, "pairs_per_step_per_gpu": n * B,
                       "parallelism": "shard%d+allgather%dB" % (world, 576 * B), "steps_in_flight": S, "check": check},
            "roofline": {"bound": "valu-int32-mac", "achieved": ach, "peak": PEAK_TMACS, "unit": "TMAC/s",
                         "frac": ach / PEAK_TMACS, "traffic": pmc_traffic(kname, n * B),
                         "traffic_unit": "bytes per launch, offline rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (profiles/r01_bench_pmc_summary.csv)",
                         "algorithmic_bytes_per_launch": (HBM_BYTES_PER_PAIRING * n * B + 576 * ((n + 2) // 3) * B),
                         "kernel": kname,
                         "kernel_launches_timed": len(miller_ms), "kernel_ms_avg": miller_avg,
                         "pairings_per_launch": n * B, "mac_per_pairing": MAC_PER_PAIRING,
                         "reduce_kernels_ms_per_step": sum(reduce_ms) / max(1, len(miller_ms)),
                         "final_exp_kernel_ms_avg": (sum(fexp_ms) / len(fexp_ms)) if fexp_ms else None,
                         "whole_step_TMACs": (MAC_PER_PAIRING * n + MAC_PER_FINAL_EXP) * B * world / (dt / args.steps) / 1e12,
                         "step_latency_ms_avg": sum(step_ms) / len(step_ms), "step_latency_ms_min": step_ms[0],
                         "hbm_GBps_algorithmic": HBM_BYTES_PER_PAIRING * n * B / (miller_avg * 1e-3) / 1e9,
                         "hbm_peak_GBps": PEAK_HBM_GBS},
            "single_verification": {"pairs": n, "latency_ms": min(one_ms), "pairings_per_s": n / (min(one_ms) * 1e-3)},
        }
        if not args.no_cpu_baseline and world == 1:
            cb, cpu_out = cpu_baseline(mine[0][0], mine[0][1], n)
            if cpu_out != per[0]:
                raise SystemExit("CPU oracle and GPU disagree -- bench invalid")
            cb["matches_gpu"] = True
            line["cpu_baseline"] = cb
        print(json.dumps(line))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
