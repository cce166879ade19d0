"""Engine clock and package power of the GPU while a command runs (samples of rocm-smi every ~0.25 s):
usage: python3 tools/clock_watch.py -- python3 bench.py --no-cpu-baseline --no-secondary --no-latency
Prints the command's stdout, then one JSON line with the samples' min / median / max."""
import json, re, statistics, subprocess, sys, threading, time
cmd = sys.argv[sys.argv.index("--") + 1:]
samples, stop = [], False


def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            j = json.loads(out)
            card = j.get("card0", next(iter(j.values())))
            sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
            mclk = next((v for k, v in card.items() if "mclk" in k.lower()), None)
            pw = next((v for k, v in card.items() if "power" in k.lower()), None)
            f = lambda s: float(re.search(r"[\d.]+", str(s)).group(0)) if s is not None and re.search(r"[\d.]+", str(s)) else None
            samples.append((time.time(), f(sclk), f(mclk), f(pw)))
        except Exception as e:
            samples.append((time.time(), None, None, repr(e)))
        time.sleep(0.2)


t = threading.Thread(target=sample, daemon=True)
t.start()
t0 = time.time()
p = subprocess.run(cmd, capture_output=True, text=True)
t1 = time.time()
stop = True
t.join(timeout=6)
sys.stdout.write(p.stdout)
sys.stderr.write(p.stderr[-2000:])
sc = [s[1] for s in samples if isinstance(s[1], float)]
pw = [s[3] for s in samples if isinstance(s[3], float)]
top = sorted(sc)[len(sc) // 2:] if sc else []
print(json.dumps({"clock_watch": {"samples": len(samples), "sclk_mhz_min": min(sc) if sc else None, "sclk_mhz_median": statistics.median(sc) if sc else None,
                                  "sclk_mhz_max": max(sc) if sc else None, "power_w_median": statistics.median(pw) if pw else None,
                                  "power_w_max": max(pw) if pw else None, "seconds": t1 - t0,
                                  "trace": [(round(s[0] - t0, 2), s[1], s[3]) for s in samples][:400], "errors": [s[3] for s in samples if isinstance(s[3], str)][:3]}}))
sys.exit(p.returncode)
