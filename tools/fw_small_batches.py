"""The finite-width leg of bench.py (Sycamore-53 supremacy sequence, max_width 32) at batches far below the chip: 256 ... 16 384
replicas -- move-evals/s and the per-step time of the staged moves and of the re-slice.  The moves are ~90 % of such a step;
up to two thirds of their kernel's wavefront slots (1 365 replicas) they run one replica per wavefront (sa_sweep.h, SPREAD).
Run on the GPU box: python tools/fw_small_batches.py"""
import json, subprocess, sys
for R in (256, 1024, 4096, 16384):
    out = subprocess.run([sys.executable, "bench.py", "--workload", "fw", "--replicas", str(R), "--pmc", "0", "--cpu-sample", "0", "--e2e", "0", "--steps", "10", "--warmup", "3"], capture_output=True, text=True, stdin=subprocess.DEVNULL)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1]); r = j["roofline"]
        print(R, "%.3e move-evals/s  %.2f ms/step  moves %.2f ms  re-slice %.2f ms  streams %s" % (j["value"], j["ms_per_step"], r["kernels"]["fw_move_kernel"]["ms_per_step"], r["kernels"]["fw_reslice_kernel"]["ms_per_step"], r.get("streams")), flush=True)
    except Exception as e:
        print(R, "failed", e, out.stderr[-500:])
