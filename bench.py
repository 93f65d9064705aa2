#!/usr/bin/env python3
"""bench.py -- SA move-evaluations/s of the HIP path on synthetic tensor networks.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it
is launched by torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

Headline leg ("im", the line's top-level fields).  A "step" = one launch of the sweep kernel:
`--sweeps-per-step` calls of Optimizer::update (include/tnco/optimize/infinite_memory/
optimizer.hpp:90-221 of the reference) on EVERY replica resident on the GPU.  The beta schedule is
linear 0 -> 100 over all (W + K) * sweeps_per_step sweeps, as tnco/app/infinite_memory/sa.py:147-156
builds it.  Inputs (trees, masks, PRNG state) are resident in HBM before the timed region.
Workload at N = 1: BASELINE.json configs[2], the configuration the metric is quoted on: 512-leaf
random 3-regular TN (bond dim 2), 65536 replicas.  For N > 1 every rank owns 65536 replicas of the
same TN (weak scaling, configs[3] at N = 8); the only collective is one RCCL all-reduce(min) of the
best cost inside the timed region.

Second leg ("fw" object of the same line): BASELINE.json configs[4], the memory-constrained
optimizer (finite_width/greedy/optimizer.hpp:117-390) on the Sycamore-53 supremacy circuit network
(depth 20, coupler sequence ABCDCDAB, 430 two-qubit gates), max_width 32, re-slicing every 10 sweeps,
65536 replicas per GPU; a step = one tnco_hip_run_fw call of `--sweeps-per-step` sweeps (11 move
launches + 10 re-slices).

`roofline.frac` is MEASURED traffic / device time / 8 TB/s: bytes from the L2's memory-side request
counters (every read request moves a 128-byte line: profiles/r04_pmc_calibration.md), so it is a
fraction; the contract's algorithmic bytes (no caching credit: the kernel carries 5 of the
7 masks of a move in registers from one level to the next) are kept as `frac_algorithmic`, and
`frac_compulsory` prices the traffic the kernel as built cannot avoid.

`roofline.traffic` and the request counts are measured in THIS run: after the timed legs, rank 0
(N = 1 only) re-runs the same command under `rocprofv3 --pmc` as child processes, one pass per
counter group, and reads the counters of the timed launches (`--pmc 0` skips it; a file under
profiles/ is used only if it was written for the same library version and workload).
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM"
# what the memory system delivers to RANDOM 128-byte lines (47e9 lines/s, tools/hbm_random.hip): the ceiling of a
# kernel whose every access is a dependent, data-chosen node of a tree -- `roofline.frac_of_achievable`
HBM_RANDOM_LINE_GBS = 6000.0
# Random fabric requests the chip retires per second whatever their size (32 / 64 / 128 B):
# tools/hbm_random.hip, profiles/r01v7_hbm_random.txt (47-52e9; 47e9 with dependent loads,
# tools/mem_latency.hip).  The sweep kernels are bound by THIS, not by bytes.
RANDOM_REQ_PEAK = 47e9
METRIC = "SA move-evaluations/s (whole node) + best log10(flops) vs ref, 512-leaf TN"
PMC_GROUPS = (("FETCH_SIZE",), ("WRITE_SIZE",), ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"),
              ("TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"), ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"))
KERNELS = ("sa_run_kernel", "fw_move_kernel", "fw_reslice_kernel", "fw_walk_kernel")
FW_MOVE_LAUNCHES = lambda sps, every: (sps + every - 1) // every + 1  # noqa: E731
FW_RESLICE_LAUNCHES = lambda sps, every: (sps + every - 1) // every   # noqa: E731


# what the library's four timers (tnco_hip_diag_kernel_times) cover, by the kernels' names in a rocprofv3 trace
TRACE_NAMES = {
    "sa_run_kernel": "sa_run_kernel<LOG2L, K, HYPER, GENERIC, false> (infinite-memory sweeps)",
    "fw_move_kernel": "sa_run_fw_kernel<LOG2L, K, SPREAD> (the finite-width moves of the plain cost model, 192 VGPRs; other cost models: "
                      "sa_run_kernel<LOG2L, K, HYPER, GENERIC, true>; fw_move_kernel<> with max_number_new_slices > 0)",
    "fw_reslice_kernel": "fw_wave_kernel<J, LOGT> + fw_reslice_a_kernel<> + fw_reslice_b_kernel<> (one wavefront per replica, "
                         "stragglers, end of sweep); fw_reslice_kernel<> in the walk + full-rebuild form",
    "fw_walk_kernel": "fw_walk2_kernel (walk + full-rebuild form only)",
}


def kernel_of(name: str):
    """Kernel name of a rocprofv3 row -> the library's timer it belongs to.  The staged sweep kernel also runs the
    moves of the finite-width optimizer: sa_run_kernel<LOG2L, K, HYPER, GENERIC, FW = true>."""
    if "sa_run_fw_kernel<" in name:
        return "fw_move_kernel"
    if "sa_run_kernel<" in name:
        targs = name.split("sa_run_kernel<", 1)[1].split(">", 1)[0].split(",")
        return "fw_move_kernel" if len(targs) >= 5 and targs[4].strip() == "true" else "sa_run_kernel"
    if "fw_move_kernel" in name:
        return "fw_move_kernel"
    if "fw_walk2_kernel" in name:
        return "fw_walk_kernel"
    if any(k in name for k in ("fw_wave_kernel", "fw_reslice_a_kernel", "fw_reslice_b_kernel", "fw_reslice_kernel")):
        return "fw_reslice_kernel"
    return None


def algorithmic_bytes_per_move(W: int, a: float, q: float) -> float:
    """B_move = 56W + 80 + a(24W + 64) + 8(2 + q)."""
    return 56 * W + 80 + a * (24 * W + 64) + 8 * (2 + q)


def algorithmic_bytes_per_move_fw(W: int, a: float, q: float) -> float:
    """The same move of the finite-width optimizer: + the slices mask read
    (8W, finite_width/greedy/optimizer.hpp:177,191-193) + the cached width written on accept (4a, :216)."""
    return algorithmic_bytes_per_move(W, a, q) + 8 * W + 4 * a


def compulsory_bytes_per_move(W: int, a: float, q: float, depth: float, fw: bool) -> dict:
    """HBM traffic per move evaluation the sweep kernel AS BUILT cannot avoid: every read is a
    128-byte line (profiles/r04_pmc_calibration.md), dirty data leaves the L2 in 64-byte sectors (a 4-byte update of a
    parent link: 32).  The working set (65 536 replicas x ~100 KB) is hundreds of times the L2, so every node first
    met is fetched and every line a move dirties is written back once.
      reads:  the next A's header line and the next C's line(s) -- unified layout (infinite memory): header + legs of a
              node in ceil((32 + 8W) / 128) lines; split layout (finite width): a 32-byte header in one line, the legs in
              ceil(8W / 128) more; + the leaf's parent line once per sweep (1 / depth per move); + 4 bytes of generator
              state per draw (2 + q per move, + 1 / depth for the leaf pick);
      writes: B's header sector (the partial sums change with every move evaluated; this move's A is the next move's
              B); per ACCEPTED move B's leg sectors beyond the header's and the parent words of C and E; the generator
              state back."""
    draws = 2.0 + q + 1.0 / max(depth, 1.0)
    if fw:
        legs_lines = -(-8 * W // 128)
        read = 128.0 * (1 + (1 + legs_lines)) + 128.0 / max(depth, 1.0) + 4.0 * draws
        write = 64.0 + a * (64.0 * -(-8 * W // 64) + 2 * 32.0) + 4.0 * draws
    else:
        node_lines = -(-(32 + 8 * W) // 128)
        read = 128.0 * (1 + node_lines) + 128.0 / max(depth, 1.0) + 4.0 * draws
        write = 64.0 + a * (64.0 * max(0, -(-(32 + 8 * W) // 64) - 1) + 2 * 32.0) + 4.0 * draws
    return {"read": read, "write": write, "total": read + write}


def algorithmic_bytes_per_reslice_repriced(n: int) -> float:
    """The re-slice in its re-priced form (no leg mask is read): the width cache (4N) and
    node links (12 per internal node) read by get_slices' ordering, and per internal node the old cost (8) and
    the two children's partial sums (16) read by the re-pricing.  The (cost, partial) pairs a KEPT re-slice
    writes back (16 per node, 25-55 % of the re-slices) and the too-wide tensors' masks are left out."""
    N = 2 * n - 1
    return 4 * N + (n - 1) * (12 + 8 + 16)


def algorithmic_bytes_per_reslice(n: int, W: int) -> float:
    """One replica's re-slice (greedy/optimizer.hpp:359-376), no caching credit: the width cache
    (4N) and node links (12 per internal node) read by get_slices' traverse; the CostCache rebuild
    reads two child masks + two child partial sums and writes (ccost, partial) per internal node.
    The too-wide tensors' masks (read twice, a data-dependent number) are left out."""
    N = 2 * n - 1
    return 4 * N + (n - 1) * (12 + 16 * W + 16 + 16)


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    c = min(c, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                    c = min(c, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return c


# ------------------------------------------------------------------------------------------------
# CPU baseline legs: the oracle (plain-C restatement of the reference), only here as the checker /
# the reported baseline -- never on the product path
# ------------------------------------------------------------------------------------------------
def cpu_baseline_im(prob, links, seeds, betas, n_sample, cores):
    """Only the update loops are timed (OpenMP over replicas inside oracle/tnco_oracle.c); tree
    flattening and cache construction are setup, as on the GPU side."""
    from oracle import oracle as orc
    orc.build()
    dt, _tot, mn, mv = orc.run_batch(links[:n_sample], prob.leaf_masks, seeds[:n_sample], betas,
                                     n_inds=prob.n_inds, dims=2, n_threads=cores)
    moves = int(mv.sum())
    return moves / dt, moves, dt, mn


def cpu_baseline_fw(prob, links, seeds, betas, n_sample, cores, max_width, every):
    """n_sample replicas of the finite-width optimizer through the oracle, one replica per host
    thread at a time (the C call releases the GIL); only the update loops are timed."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from tnco_amd import ctree
    orc.build()

    def make(r):
        l, rr, p = (np.ascontiguousarray(links[r, j]) for j in range(3))
        inds = ctree.derive_inds(l, rr, prob.leaf_masks, None)
        return orc.Oracle(l, rr, p, inds, n_inds=prob.n_inds, dims=2, seed=int(seeds[r]), max_width=max_width)

    def run(o):
        o.run(orc.PROB_MH, betas, update_slices_every=every)
        return o.counters()["moves"], o.min_total_cost

    with ThreadPoolExecutor(cores) as ex:
        states = list(ex.map(make, range(n_sample)))
        t0 = time.perf_counter()
        out = list(ex.map(run, states))
        dt = time.perf_counter() - t0
    moves = sum(m for m, _ in out)
    return moves / dt, moves, dt, np.array([c for _, c in out])


# ------------------------------------------------------------------------------------------------
# one timed leg on this rank's GPU
# ------------------------------------------------------------------------------------------------
class Leg:
    def __init__(self, kind, args, rank, world, local_rank, grouped=None):
        from tnco_amd import core, synthetic
        self.kind, self.args, self.rank, self.world, self.local_rank = kind, args, rank, world, local_rank
        self.grouped = world > 1 if grouped is None else grouped
        R = args.replicas
        if kind == "im":
            self.prob = synthetic.regular_problem(args.leaves, graph_seed=args.graph_seed)
            kw = {}
        else:
            self.prob = synthetic.sycamore_problem(args.fw_depth, args.fw_layout)
            kw = dict(max_width=args.fw_max_width)
        all_seeds = synthetic.replica_seeds(R * world, S=0)
        self.seeds = all_seeds[rank * R:(rank + 1) * R]
        if args.init == "greedy":  # as the reference starts every run (tnco/utils/tn.py:189-230), drawn on this rank's GPU
            self.links = core.greedy_trees(self.prob.ts_inds, self.prob.n_inds, self.seeds, device=local_rank)
        else:
            self.links = core.random_trees(self.prob.ts_inds, self.prob.n_inds, self.seeds)
        self.sps = args.sweeps_per_step
        self.total_sweeps = (args.warmup + args.steps) * self.sps
        self.betas = synthetic.linear_betas(0.0, 100.0, self.total_sweeps)
        self.opt = core.BatchedOptimizer(self.prob.leaf_masks, self.links, self.seeds, n_inds=self.prob.n_inds,
                                         dims=2, device=local_rank, **kw)

    def step(self, s):
        self.opt.run(self.betas[s * self.sps:(s + 1) * self.sps], update_slices_every=self.args.fw_update_slices)

    def run(self, barrier, dist):
        """W untimed steps, then exactly K timed ones between two barriers; the best-cost reduction
        (the path's only collective) is inside the timed region."""
        from tnco_amd import parallel
        a, opt = self.args, self.opt
        for s in range(a.warmup):
            self.step(s)
        barrier(opt)
        c0 = opt.counters()
        self.fw_stats0 = opt.fw_stats() if self.kind == "fw" else None
        opt.kernel_times_ms(reset=True)
        t0 = time.perf_counter()
        for s in range(a.warmup, a.warmup + a.steps):
            self.step(s)
        best = parallel.global_best(opt, rank=self.rank, world=self.world, device=self.local_rank, grouped=self.grouped)
        barrier(opt)
        dt = time.perf_counter() - t0
        c1 = opt.counters()
        kt = opt.kernel_times_ms()
        self.device_ms = opt.kernel_time_ms()[0]  # (all kernels of the timed steps; two streams: first launch -> last end)
        d = {k: c1[k] - c0[k] for k in c1}
        self.groups = opt.launch_groups
        self.repriced, self.fw_stats = False, None
        if self.kind == "fw":
            st1 = opt.fw_stats()
            self.fw_stats = {k: st1[k] - self.fw_stats0[k] for k in st1}  # (the timed steps' re-slices)
            self.repriced = self.fw_stats["repriced"] > 0 and self.fw_stats["full_rebuild_form"] == 0
        return dict(dt=dt, best=best, kt=kt, **d)


def reduce_legs(res, world, dist, torch, grouped=None):
    """max over ranks of the times, sum of the work; plus what every rank did (all-gather), so that
    the line itself shows how many ranks took part."""
    names = KERNELS
    mine = [res["dt"], float(res["moves"]), float(res["accepted"]), float(res["random_picks"]),
            float(res["improved"]), float(res["full_copies"])] + [res["kt"][k][0] for k in names] + [float(res.get("n_bad", -1))]
    per_rank = [mine]
    from tnco_amd import parallel
    if (world > 1 if grouped is None else grouped) and parallel._native is not None:
        per_rank = [[float(x) for x in v] for v in parallel._native.allgather_array(np.array(mine, np.float64))]
    elif world > 1 if grouped is None else grouped:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor(mine, dtype=torch.float64, device=dev)
        allv = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allv, t)
        per_rank = [[float(x) for x in v] for v in allv]
    arr = np.array(per_rank)
    out = dict(res)
    out["dt"] = float(arr[:, 0].max())
    for j, k in enumerate(("moves", "accepted", "random_picks", "improved", "full_copies")):
        out[k] = float(arr[:, 1 + j].sum())
    out["kt"] = {k: (float(arr[:, 6 + j].max()), res["kt"][k][1]) for j, k in enumerate(names)}
    nb = arr[:, 6 + len(names)]
    if (nb >= 0).all():  # (every rank validated its replicas: the line reports the sum over all of them)
        out["n_bad"] = int(nb.sum())
    else:
        out.pop("n_bad", None)
    out["per_rank"] = [dict(rank=i, wall_s=v[0], moves=v[1], kernel_ms=sum(v[6:6 + len(names)]), bad_replicas=int(v[6 + len(names)]))
                       for i, v in enumerate(per_rank)]
    return out


# ------------------------------------------------------------------------------------------------
# PMC passes (rank 0, N = 1): the same command under rocprofv3, one child process per counter group
# ------------------------------------------------------------------------------------------------
def pmc_passes(args, lib_version):
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = Path(tempfile.mkdtemp(prefix="tnco_pmc_", dir="/tmp"))
    env = dict(os.environ, TMPDIR="/tmp")
    cmd_tail = [sys.executable, str(ROOT / "bench.py"), "--steps", str(args.steps), "--warmup", str(args.warmup),
                "--sweeps-per-step", str(args.sweeps_per_step), "--leaves", str(args.leaves),
                "--replicas", str(args.replicas), "--graph-seed", str(args.graph_seed), "--init", args.init,
                "--workload", args.workload, "--fw-max-width", str(args.fw_max_width),
                "--fw-update-slices", str(args.fw_update_slices), "--fw-depth", str(args.fw_depth),
                "--fw-layout", args.fw_layout, "--cpu-sample", "0", "--pmc", "0", "--e2e", "0", "--no-validate"]
    vals = {}  # (kernel short name, counter) -> per-dispatch values in dispatch order
    names = {}  # kernel short name -> the kernels' names in the trace
    t0 = time.perf_counter()
    for gi, grp in enumerate(PMC_GROUPS):
        out = tmp / f"g{gi}"
        cmd = [exe, "--pmc", *grp, "--kernel-trace", "--output-format", "csv", "-d", str(out), "-o", "pmc", "--", *cmd_tail]
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=args.pmc_timeout)
        except subprocess.TimeoutExpired:
            shutil.rmtree(tmp, ignore_errors=True)
            return None, f"rocprofv3 pass {grp} timed out"
        files = glob.glob(str(out / "**" / "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            shutil.rmtree(tmp, ignore_errors=True)
            return None, f"rocprofv3 pass {grp} failed (rc {p.returncode}): {p.stderr[-300:]}"
        rows = []
        for f in files:
            with open(f) as fh:
                rows += list(csv.DictReader(fh))
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            short = kernel_of(r["Kernel_Name"])
            if short:
                vals.setdefault((short, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
                names.setdefault(short, set()).add(r["Kernel_Name"].split("(")[0].replace("tnco::", "").replace("(anonymous namespace)::", ""))
    shutil.rmtree(tmp, ignore_errors=True)
    every = args.fw_update_slices
    per_step = {"sa_run_kernel": 1, "fw_move_kernel": FW_MOVE_LAUNCHES(args.sweeps_per_step, every),
                "fw_reslice_kernel": FW_RESLICE_LAUNCHES(args.sweeps_per_step, every),
                "fw_walk_kernel": FW_RESLICE_LAUNCHES(args.sweeps_per_step, every)}
    res = {"library": lib_version, "seconds": None, "kernels": {}, "trace_names": {k: sorted(v) for k, v in names.items()}}
    for (short, ctr), v in vals.items():
        n = per_step[short]
        # (a step is two concurrent dispatches per kernel when the handle splits it over two streams; the re-slice by
        #  re-pricing is three kernels: fw_wave_kernel (order | get_slices | re-pricing of a replica in one wavefront) |
        #  its stragglers in fw_reslice_a_kernel | end of the sweep -- so 1, 2, 3 or 6 times n)
        mult, rest = divmod(len(v), n * (args.warmup + args.steps))
        if rest or mult < 1:
            continue
        n *= mult
        timed = v[args.warmup * n:(args.warmup + args.steps) * n]  # the timed steps' dispatches
        if len(timed) != args.steps * n:
            continue
        res["kernels"].setdefault(short, {})[ctr] = sum(timed) / args.steps  # per step
    res["seconds"] = time.perf_counter() - t0
    return res, None


def pmc_from_file(lib_version, key):
    """profiles/pmc_traffic.json, only when it was taken from the same library and workload."""
    f = ROOT / "profiles" / "pmc_traffic.json"
    try:
        j = json.loads(f.read_text())
    except (OSError, ValueError):
        return None
    if j.get("library") != lib_version or j.get("workload_key") != key:
        return None
    return j


def traffic_fields(k, moves_per_step, step_s):
    """From one kernel's per-step counters: the bytes that crossed the L2's memory side -- every read request is a
    128-byte line whatever the kernel asked for (RDREQ_128B == RDREQ on every pattern of profiles/r04_pmc_calibration.md;
    FETCH_SIZE tallies them at 64), writes are WRITE_SIZE as is -- and the request counts."""
    out = {}
    wr = k["WRITE_SIZE"] * 1024.0 if "WRITE_SIZE" in k else None
    rd = None
    if "TCC_EA0_RDREQ_128B_sum" in k and "TCC_EA0_RDREQ_64B_sum" in k and "TCC_EA0_RDREQ_sum" in k:
        r128, r64, r32 = k["TCC_EA0_RDREQ_128B_sum"], k["TCC_EA0_RDREQ_64B_sum"], k.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        rest = max(0.0, k["TCC_EA0_RDREQ_sum"] - r128 - r64 - r32)  # (requests in no size class: priced as lines)
        rd = 128.0 * (r128 + rest) + 64.0 * r64 + 32.0 * r32
        out["read_requests_by_size"] = {"128B": r128 / moves_per_step, "64B": r64 / moves_per_step, "32B": r32 / moves_per_step}
    elif "FETCH_SIZE" in k:
        rd = 2.0 * k["FETCH_SIZE"] * 1024.0
    if rd is not None and wr is not None:
        out["traffic"] = rd + wr
        out["traffic_read"], out["traffic_write"] = rd, wr
        out["traffic_per_move"] = {"read": rd / moves_per_step, "write": wr / moves_per_step}
        if "FETCH_SIZE" in k:
            out["traffic_uncorrected"] = k["FETCH_SIZE"] * 1024.0 + wr  # (FETCH_SIZE as printed: 64 bytes per read request)
        if step_s:
            out["traffic_frac"] = out["traffic"] / step_s / 1e9 / HBM_PEAK_GBS
    if "TCC_EA0_RDREQ_sum" in k and "TCC_EA0_WRREQ_sum" in k:
        rq, wq = k["TCC_EA0_RDREQ_sum"], k["TCC_EA0_WRREQ_sum"]
        out["requests_per_move"] = {"read": rq / moves_per_step, "write": wq / moves_per_step,
                                    "read_32B": k.get("TCC_EA0_RDREQ_32B_sum", 0.0) / moves_per_step,
                                    "write_64B": k.get("TCC_EA0_WRREQ_64B_sum", 0.0) / moves_per_step}
        if step_s:
            out["request_rate"] = (rq + wq) / step_s
            out["request_rate_frac"] = out["request_rate"] / RANDOM_REQ_PEAK
    return out


def transport_verdict(comm_kind, world, requested):
    """(transport, rccl_ranks, exit_code) of a grouped run.  A launch on N > 1 GPUs whose ranks did not talk through
    RCCL must not pass for an N-GPU measurement: non-zero exit unless sockets / gloo were ASKED for (tests)."""
    kind = comm_kind or "none"
    is_rccl = "rccl" in kind.lower() or "nccl" in kind.lower()
    rccl_ranks = world if is_rccl else 0
    asked_otherwise = requested in ("sockets", "gloo")
    code = 0 if (world <= 1 or is_rccl or asked_otherwise) else 3
    return kind, rccl_ranks, code


def end_to_end(args):
    """optimize() of the reference's plugin API on the headline network, wall time of the whole call: index-list spec
    -> TensorNetwork -> initial trees as the reference draws them (on the device) -> create -> sweeps -> the 16 best
    contraction paths as Python objects.  Informational (not `value`): the hot path is the `sweeps` part of it."""
    try:
        import warnings

        from tnco_amd import synthetic
        from tnco_amd.app import Optimizer
        ts, _d, _o = synthetic.random_regular_tn(args.leaves, 3, args.graph_seed)
        n_inds = max(max(x) for x in ts) + 1
        spec = [(2, *[f"t{t}" for t in range(len(ts)) if k in ts[t]]) for k in range(n_inds)]
        n_steps, times = 1000, []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(3):
                t0 = time.perf_counter()
                _tn, res = Optimizer(method="sa", seed=0).optimize(spec, betas=(0, 100), n_steps=n_steps, n_runs=args.replicas,
                                                                  top_k=16, fuse=None)
                times.append(time.perf_counter() - t0)
        return {"call": f"Optimizer(method='sa', seed=0).optimize(<{args.leaves}-tensor 3-regular network>, betas=(0, 100), "
                        f"n_steps={n_steps}, n_runs={args.replicas}, top_k=16, fuse=None)",
                "seconds": round(min(times[1:]), 4), "first_call_seconds": round(times[0], 4),
                "best_log2_cost": round(float(np.log2(float(res[0].cost))), 4)}
    except Exception as e:  # (informational: never fails the bench line)
        return {"error": f"{type(e).__name__}: {e}"}


def side_leg(label, prob, R, sweeps, local_rank, *, warm=50, cpu_n=64, dims=2, sparse_n_projs=None, init_seed_S=0):
    """One informational leg beside the two timed ones (N = 1): `sweeps` sweeps of R replicas of `prob` in ONE call,
    wall time around it; afterwards is_valid() of every replica on the device and `cpu_n` of the same replicas through
    the oracle, whole schedule, best costs compared bit for bit.  Never fails the bench line."""
    try:
        from concurrent.futures import ThreadPoolExecutor

        from tnco_amd import core, ctree, synthetic
        seeds = synthetic.replica_seeds(R, S=init_seed_S)
        om = prob.output_mask if prob.output_mask.any() else None
        links = core.greedy_trees(prob.ts_inds, prob.n_inds, seeds, output_mask=om, device=local_rank)
        betas = synthetic.linear_betas(0.0, 100.0, sweeps + warm)
        kw = dict(n_inds=prob.n_inds, dims=dims, output_mask=om)
        if prob.sparse_mask is not None:
            kw.update(sparse_mask=prob.sparse_mask, n_projs=sparse_n_projs)
        with core.BatchedOptimizer(prob.leaf_masks, links, seeds, device=local_rank, **kw) as opt:
            opt.run(betas[:warm])
            opt.sync()
            m0 = opt.counters()["moves"]
            t0 = time.perf_counter()
            opt.run(betas[warm:])
            opt.sync()
            dt = time.perf_counter() - t0
            moves = opt.counters()["moves"] - m0
            bad = int(opt.validate()[0])
            mn = opt.costs()[1]
            lds = opt.launch_groups == 0
        from oracle import oracle as orc
        orc.build()
        cpu_n = min(cpu_n, R)

        def one(r):
            l, rr, p = (np.ascontiguousarray(links[r, j]) for j in range(3))
            inds = ctree.derive_inds(l, rr, prob.leaf_masks, om)
            o = orc.Oracle(l, rr, p, inds, n_inds=prob.n_inds, dims=dims, seed=int(seeds[r]), sparse=prob.sparse_mask,
                           n_projs=sparse_n_projs or 0)
            o.run(orc.PROB_MH, betas)
            return o.min_total_cost

        with ThreadPoolExecutor(usable_cores()) as ex:
            omn = np.array(list(ex.map(one, range(cpu_n))))
        return {"workload": label, "value": moves / dt, "unit": "move-evals/s", "seconds": dt, "replicas": R, "sweeps": sweeps,
                "n_leaves": prob.n, "n_inds": prob.n_inds, "mask_words": prob.W, "validated_bad_replicas": bad,
                "kernel": "LDS-resident (sa_small_kernel / sa_lds_kernel)" if lds else "sa_run_kernel (trees in HBM)",
                "cpu_sample": cpu_n, "cpu_sample_min_cost_bit_exact": bool(np.array_equal(omn, mn[:cpu_n])),
                "best_log10_flops": float(np.log10(mn.min()))}
    except Exception as e:  # (informational: never fails the bench line)
        return {"workload": label, "error": f"{type(e).__name__}: {e}"}


def side_legs(local_rank, which):
    """The configurations the two timed legs do not show (N = 1, a few seconds each): BASELINE configs[1]; the reference
    loader's default for circuits, hyper-indices (tnco/app/app.py:351-358); per-index dims
    (infinite_memory/cost_model/simple.hpp:51-53); batches too small to fill the chip (the LDS-resident and the spread
    form); a network of 24 mask words."""
    from tnco_amd import synthetic
    out = {}
    if "c2" in which:
        out["c2"] = side_leg("C2: 64-leaf 3-regular TN d=2, 4096 replicas, MH, f64, 1000 sweeps (one launch)",
                             synthetic.regular_problem(64, graph_seed=7), 4096, 1000, local_rank)
    if "hyper" in which:
        hts, hd, hout = synthetic.random_hyper_tn(512, 768, k=3, n_output=8, seed=3)
        out["hyper"] = side_leg("hyper-index network: 512 tensors, 768 indices (3 tensors each) d=2, 8 open; 65536 replicas, 400 sweeps",
                                synthetic.Problem(hts, hd, hout, n_inds=768), 65536, 400, local_rank)
    if "dims" in which:
        prob = synthetic.regular_problem(512, graph_seed=11)
        dv = np.random.RandomState(0).choice([2, 3, 4], size=prob.n_inds).astype(np.uint64)
        out["dims"] = side_leg("C3's network with per-index dims in {2, 3, 4}; 65536 replicas, 400 sweeps", prob, 65536, 400, local_rank, dims=dv)
    if "small_batch" in which:
        prob = synthetic.regular_problem(512, graph_seed=11)
        out["small_batch"] = {
            "512_runs": side_leg("C3's network, 512 replicas, 1000 sweeps (one launch)", prob, 512, 1000, local_rank),
            "2048_runs": side_leg("C3's network, 2048 replicas, 1000 sweeps (one launch)", prob, 2048, 1000, local_rank)}
    if "wide" in which:
        out["wide"] = side_leg("1024-leaf 3-regular TN d=2 (24 mask words), 65536 replicas, 200 sweeps",
                               synthetic.regular_problem(1024, graph_seed=11), 65536, 200, local_rank)
    return out


def calibration(local_rank):
    """This box's memory system under the sweep kernels' access pattern (tools/box_probe.hip, built by
    __graft_entry__.build() as tools/libbox_probe.so), measured in this process: fresh boxes of the pool differ by ~10 %."""
    import ctypes
    try:
        L = ctypes.CDLL(str(ROOT / "tools" / "libbox_probe.so"))
        L.box_probe.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
        out = (ctypes.c_double * 5)()
        t0 = time.perf_counter()
        rc = L.box_probe(local_rank, 8.0, out)
        if rc:
            return {"error": f"box_probe: hip error {rc}"}
        return {"random_lines_per_s": out[0], "move_pattern_per_s": out[1], "stream_read_GBs": out[2] / 1e9,
                "valu_wave_insts_per_s_per_simd": out[3], "dependent_load_ns": out[4] * 1e9,
                "seconds": time.perf_counter() - t0,
                "what": "tools/box_probe.hip in this process, 8 GiB working set: random 128-byte lines read (4 lanes per line, 4 in "
                        "flight per group); the memory side of one infinite-memory move alone (2 random lines read, 1 header "
                        "sector written, 3 of 4 moves the rest of the line + two 4-byte parent words); streaming read; 32-bit shift-add "
                        "wave-instructions per second and SIMD under a full VALU load (clock / 4); one lane's dependent loads of cold lines on "
                        "an idle chip"}
    except OSError as e:
        return {"error": f"{type(e).__name__}: {e}"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sweeps-per-step", type=int, default=100)
    ap.add_argument("--leaves", type=int, default=512)
    ap.add_argument("--replicas", type=int, default=65536, help="replicas per GPU")
    ap.add_argument("--graph-seed", type=int, default=11)
    ap.add_argument("--init", choices=("kruskal", "greedy"), default="greedy",
                    help="initial trees: the reference's recipe (Random(seed).shuffle + opt_einsum greedy, restated, drawn "
                         "on the GPU) or the build's random-Kruskal generator (the line of round 1)")
    ap.add_argument("--workload", choices=("both", "im", "fw"), default="both",
                    help="im: the headline leg only; fw: the finite-width leg as the headline; both: im + 'fw' object")
    ap.add_argument("--fw-max-width", type=float, default=32.0)
    ap.add_argument("--fw-layout", choices=("supremacy", "alternating"), default="supremacy",
                    help="coupler patterns of the Sycamore-53 network: the supremacy experiment's (A, B one orientation on alternate "
                         "rows, C, D the other) or the easier assignment of rounds 1-3")
    ap.add_argument("--fw-update-slices", type=int, default=10)
    ap.add_argument("--fw-depth", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="replicas timed on the CPU oracle (0 = skip, -1 = as many as take ~10-15 s per leg)")
    ap.add_argument("--pmc", type=int, default=1, help="1: measure HBM traffic / fabric requests under rocprofv3 --pmc (N = 1)")
    ap.add_argument("--pmc-timeout", type=float, default=240.0)
    ap.add_argument("--pmc-out", default=None, help="also write the per-step PMC counters of the timed kernels to this file")
    ap.add_argument("--validate", action=argparse.BooleanOptionalAction, default=True,
                    help="device-side is_valid() of every replica after the timed region (default on)")
    ap.add_argument("--e2e", type=int, default=1,
                    help="1 (N = 1): also time app.Optimizer(method='sa').optimize() of the headline network end to end "
                         "(spec -> initial trees -> sweeps -> best paths): the `end_to_end` object, informational")
    ap.add_argument("--extras", default="c2,hyper,dims,small_batch,wide",
                    help="informational side legs of the line (N = 1; with --e2e 1): any of c2, hyper, dims, small_batch, wide; '' = none")
    ap.add_argument("--calibrate", type=int, default=1, help="1: measure this box's random-request rates in this process (`calibration`)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    # N > 1 (or TNCO_BENCH_FORCE_GROUP, the test of a group of ONE rank on a 1-GPU box): the ranks talk through RCCL bound
    # inside libtnco_hip.so (tnco_amd/parallel.py NativeComm: this process then holds ONE HIP runtime, torch is only the
    # launcher) -- or, if RCCL does not come up on every rank, through plain sockets (16 bytes per exchange; the line
    # says so); TNCO_BENCH_COMM=torch goes through torch.distributed ("nccl" = RCCL in PyTorch's bundled runtime) as
    # rounds 1-2 did; TNCO_BENCH_SHARE_GPU (tests: N ranks on one GPU) through gloo.
    from tnco_amd import parallel
    grouped = world > 1 or bool(os.environ.get("TNCO_BENCH_FORCE_GROUP"))
    torch = dist = None
    comm_kind, comm_note, comm_requested = None, None, None
    if grouped:
        # RCCL prints a version banner on the C-level stdout, flushed when the process ends -- after the JSON line.  The
        # line must be the only thing on stdout: everything written to file descriptor 1 from here on goes to stderr,
        # Python's own sys.stdout keeps the real one.
        sys.stdout.flush()
        real_stdout = os.dup(1)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(real_stdout, "w")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        want = "gloo" if os.environ.get("TNCO_BENCH_SHARE_GPU") else os.environ.get("TNCO_BENCH_COMM", "native")
        if os.environ.get("TNCO_BENCH_SHARE_GPU") == "sockets":
            # test knob: N ranks on ONE GPU over the socket transport (what a launch falls back to when RCCL does not
            # come up on every rank), no torch in the ranks
            want, local_rank = "native", 0
            os.environ["TNCO_COMM"] = "sockets"
        if want == "native":
            # RCCL if it comes up on every rank, else sockets on every rank (parallel.init_native: the ranks agree on a
            # side channel; ncclCommInitRank runs under a time limit) -- the line says which, and why
            c = parallel.init_native(rank, world, local_rank)
            comm_kind, comm_note = c.kind, getattr(c, "note", None)
        comm_requested = "sockets" if os.environ.get("TNCO_COMM") == "sockets" else want
        if want != "native":
            import torch
            import torch.distributed as dist
            if want == "gloo":
                # test knob (tests/test_gpu_two_ranks.py): N ranks on ONE GPU over gloo -- the N > 1 code of this
                # file on a 1-GPU box; RCCL needs a GPU per rank and is what the driver's scaling run uses
                local_rank = 0
                torch.cuda.set_device(0)
                dist.init_process_group("gloo")
            else:
                import datetime
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                        timeout=datetime.timedelta(seconds=1800))
            comm_kind = f"torch.distributed {dist.get_backend()}"
    else:
        import torch  # (one rank: nothing to exchange; torch only for the contract's torch.cuda.synchronize())

    from tnco_amd import _lib
    lib_version = _lib.load().tnco_hip_version().decode()

    def barrier(opt):
        opt.sync()
        if torch is not None:
            torch.cuda.synchronize()
        if parallel._native is not None:
            parallel._native.barrier()
        elif grouped:
            dist.barrier()
            torch.cuda.synchronize()

    legs = {"both": ("im", "fw"), "im": ("im",), "fw": ("fw",)}[args.workload]
    results, objs = {}, {}
    calib = calibration(local_rank) if (args.calibrate and rank == 0 and world == 1) else None
    for kind in legs:
        leg = Leg(kind, args, rank, world, local_rank, grouped)
        res = leg.run(barrier, dist)
        # after the timed region: is_valid(atol) of EVERY replica, recomputed on the device from the trees alone
        # (ContractionTree::is_valid + both caches against a from-scratch rebuild, infinite_memory/optimizer.hpp:223-251)
        # -- the line certifies the work it counted: every rank validates its own replicas, the counts are summed
        # with the moves (reduce_legs); --no-validate skips it
        if args.validate:
            tv = time.perf_counter()
            res["n_bad"] = int(leg.opt.validate()[0])
            res["validate_s"] = time.perf_counter() - tv
        res = reduce_legs(res, world, dist, torch, grouped)
        results[kind], objs[kind] = res, leg
        if not (rank == 0 and world == 1 and args.cpu_sample != 0):
            leg.opt.close()
            leg.opt = None

    devices = None
    if grouped:  # what the communicator saw: one entry per rank
        import ctypes
        nm = ctypes.create_string_buffer(256)
        _lib.load().tnco_hip_device_name(local_rank, nm, 256)
        me = dict(rank=rank, local_rank=local_rank, device=nm.value.decode(), backend=comm_kind)
        if parallel._native is not None:
            devices = parallel._native.allgather_object(me)
        else:
            devices = [None] * world
            dist.all_gather_object(devices, me)

    if rank == 0:
        R, sps, every = args.replicas, args.sweeps_per_step, args.fw_update_slices
        key = (f"{args.workload}/{args.leaves}/{R}/{sps}/{args.steps}/{args.warmup}/{args.fw_max_width}/{every}/"
               f"{args.fw_depth}/{args.fw_layout}/{args.init}")
        pmc, pmc_note = None, None
        if args.pmc and world == 1:
            for leg in objs.values():  # free the GPU memory of this process first
                if leg.opt is not None and args.cpu_sample == 0:
                    leg.opt.close()
                    leg.opt = None
            pmc, pmc_note = pmc_passes(args, lib_version)
            if pmc is not None:
                pmc["workload_key"] = key
                if args.pmc_out:
                    Path(args.pmc_out).write_text(json.dumps(pmc, indent=1) + "\n")
        if pmc is None and world == 1:
            pmc = pmc_from_file(lib_version, key)
            if pmc is not None:
                pmc_note = "profiles/pmc_traffic.json (same library and workload)"

        def leg_object(kind):
            res, leg = results[kind], objs[kind]
            prob = leg.prob
            moves = res["moves"]
            a, q = res["accepted"] / max(moves, 1), res["random_picks"] / max(moves, 1)
            kt = res["kt"]
            moves_per_step_gpu = moves / world / args.steps
            if kind == "im":
                kernels = ("sa_run_kernel",)
                bmove = algorithmic_bytes_per_move(prob.W, a, q)
                alg_per_step = bmove * moves_per_step_gpu
                extra = {"algorithmic_bytes_per_move": bmove}
            else:
                kernels = tuple(k for k in ("fw_move_kernel", "fw_reslice_kernel", "fw_walk_kernel") if kt[k][1] > 0)
                bmove = algorithmic_bytes_per_move_fw(prob.W, a, q)
                # (the full CostCache rebuild is credited only when it is what ran: VERDICT r02)
                bres = (algorithmic_bytes_per_reslice_repriced(prob.n) if leg.repriced
                        else algorithmic_bytes_per_reslice(prob.n, prob.W))
                n_res = FW_RESLICE_LAUNCHES(sps, every) * R
                alg_per_step = bmove * moves_per_step_gpu + bres * n_res
                extra = {"algorithmic_bytes_per_move": bmove, "algorithmic_bytes_per_reslice": bres,
                         "reslice_form": ("re-priced, one wavefront per replica: too-wide tensors ordered | get_slices | costs re-priced from the old ones "
                                          "(fw_wave_kernel; no walk, no leg masks read but the too-wide tensors')"
                                          if leg.repriced else "full CostCache rebuild"),
                         "reslices_per_step": n_res, "algorithmic_bytes_moves_only": bmove * moves_per_step_gpu}
            step_ms = sum(kt[k][0] for k in kernels) / args.steps  # device time of one step's kernels
            if kind == "fw" and leg.groups > 1 and world == 1:
                # two halves of the batch on two streams: their kernels overlap -- the leg's device time is the region's
                # (first launch of the timed steps to the end of the last), the per-kernel times are per-stream averages
                step_ms = leg.device_ms / args.steps
            achieved_alg = alg_per_step / (step_ms / 1e3) / 1e9
            depth = moves_per_step_gpu / max(R * sps, 1)  # moves per sweep = depth of the drawn leaf - 1, on average
            comp = compulsory_bytes_per_move(prob.W, a, q, depth, kind == "fw")
            comp_per_step = comp["total"] * moves_per_step_gpu
            if kind == "fw" and leg.repriced:  # + the re-slice: headers read once, a kept one rewrites them, legs of the too-wide tensors
                comp_per_step += (32.0 * (prob.n - 1) * 1.4 + 2500.0) * n_res
            roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                    "achieved_algorithmic": achieved_alg, "frac_algorithmic": achieved_alg / HBM_PEAK_GBS,
                    "compulsory_bytes_per_move": comp, "compulsory_bytes_per_step": comp_per_step,
                    "frac_compulsory": comp_per_step / (step_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                    "kernel": max(kernels, key=lambda k: kt[k][0]), "avg_launch_ms": step_ms,
                    "launches": args.steps, **extra,
                    "trace_names": {k: TRACE_NAMES[k] for k in kernels},
                    "note": "frac = traffic / device time / peak with traffic MEASURED in this run (rocprofv3 --pmc: 128 bytes per "
                            "read request of the L2's memory side -- profiles/r04_pmc_calibration.md -- + WRITE_SIZE); "
                            "frac_compulsory = the traffic the kernel as built cannot avoid (model) <= frac <= 1; frac_algorithmic = "
                            "the contract's bytes (SURVEY 8(d): 7 masks read per move, no caching credit) over the same time -- above 1 "
                            "where the kernel carries masks in registers from one level of the walk to the next; request_rate_frac = "
                            "fabric requests vs the 47e9/s random-request ceiling (tools/hbm_random.hip: 47e9 x 128 B = 6.0 TB/s)"}
            if kind == "fw":  # the moves' bytes alone over the leg's device time, and over the move kernel's
                roof["frac_algorithmic_moves_only"] = extra["algorithmic_bytes_moves_only"] / (step_ms / 1e3) / 1e9 / HBM_PEAK_GBS
                st = leg.fw_stats or {}
                tot_res = max(1, st.get("repriced", 0) + st.get("full_rebuild_form", 0))
                roof["reslices"] = dict(st, left_one_wavefront_path_frac=(st.get("fell_back", 0) + st.get("full_rebuild_form", 0)) / tot_res)
                if leg.groups > 1:
                    roof["streams"] = leg.groups
                    roof["note_streams"] = ("the two halves of the batch run the whole step -- moves, re-slice (fw_wave_kernel), "
                                            "end of sweep -- on streams of their own: request-bound moves of one half overlap the "
                                            "latency-bound re-slice kernels of the other; avg_launch_ms = device time of the timed "
                                            "region / steps, kernels[*].ms_per_step = average time a stream spent in that kernel")
            elif leg.groups > 1:
                roof["sub_launches_per_step"] = leg.groups
                roof["note_streams"] = (f"a step is {leg.groups} concurrent launches of sa_run_kernel over half of the replicas "
                                        "each, on two streams (no idle tail: 1024 blocks for 768 resident ones); avg_launch_ms = "
                                        "device time from the first launch of the timed region to the end of the last / steps -- "
                                        "in a kernel trace every sub-launch lasts about that long, two at a time")
            roof["kernels"] = {k: {"ms_per_step": kt[k][0] / args.steps, "launches_per_step": kt[k][1] / args.steps}
                               for k in kernels}
            if pmc is not None:
                tot = {}
                for k in kernels:
                    kc = pmc["kernels"].get(k)
                    if not kc:
                        continue
                    # (two streams: the kernels of the two halves overlap -- rates per kernel mean nothing, the leg's do)
                    overlapped = kind == "fw" and leg.groups > 1
                    f = traffic_fields(kc, moves_per_step_gpu, None if overlapped else kt[k][0] / args.steps / 1e3)
                    roof["kernels"][k].update(f)
                    for c, v in kc.items():
                        tot[c] = tot.get(c, 0.0) + v
                roof.update(traffic_fields(tot, moves_per_step_gpu, step_ms / 1e3))
                roof["traffic_source"] = pmc_note or f"rocprofv3 --pmc child passes of this run ({pmc.get('seconds', 0):.0f} s), {pmc['library']}"
                if pmc.get("trace_names"):
                    roof["trace_names_seen"] = {k: pmc["trace_names"].get(k) for k in kernels}
                    seen = pmc["trace_names"].get(roof["kernel"]) or []
                    # (the dominant kernel by its name in the trace: profiles/r04_kernel_stats.csv has this row)
                    dom = [x for x in seen if "fw_wave_kernel" in x or "sa_run_kernel" in x or "sa_run_fw_kernel" in x]
                    if dom:
                        roof["kernel"] = dom[0].replace("void ", "")
            else:
                roof["traffic_source"] = pmc_note
            if roof.get("traffic") is not None:
                roof["achieved"] = roof["traffic"] / (step_ms / 1e3) / 1e9
                roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
                roof["frac_of_achievable"] = roof["achieved"] / HBM_RANDOM_LINE_GBS
                if calib and calib.get("random_lines_per_s"):
                    # the same two fractions against what THIS box retires (the `calibration` object), not the constants
                    # measured on one box rounds ago: comparable between boxes
                    roof["frac_of_achievable_this_box"] = roof["achieved"] * 1e9 / (128.0 * calib["random_lines_per_s"])
                    if roof.get("request_rate"):
                        roof["request_rate_frac_this_box"] = roof["request_rate"] / calib["random_lines_per_s"]
                roof["frac_source"] = "measured traffic (this run's PMC passes)" if not pmc_note else f"measured traffic ({pmc_note})"
            else:  # no counters (rocprofv3 missing, --pmc 0, N > 1): the model of the compulsory traffic, said so
                roof["achieved"] = comp_per_step / (step_ms / 1e3) / 1e9
                roof["frac"] = roof["frac_compulsory"]
                roof["frac_of_achievable"] = roof["achieved"] / HBM_RANDOM_LINE_GBS
                roof["frac_source"] = "compulsory-traffic MODEL (no PMC counters in this run)"
            obj = {
                "value": moves / res["dt"], "unit": "move-evals/s", "ms_per_step": res["dt"] / args.steps * 1e3,
                "config": {
                    # (short keys first: everything that identifies the workload, nothing of it in a long sentence)
                    "workload": (f"C3: {prob.n}-leaf 3-regular TN d=2, {R} replicas/GPU, MH, f64, beta 0->100" if kind == "im" else
                                 f"C5: Sycamore-53 depth {args.fw_depth} ({args.fw_layout}), max_width {args.fw_max_width:g}, "
                                 f"{R} replicas/GPU, MH, f64"),
                    "network": (f"random 3-regular graph, seed {args.graph_seed}: {prob.n} tensors, {prob.n_inds} indices of dimension 2, "
                                f"{prob.W} mask words" if kind == "im" else
                                (f"Sycamore-53 supremacy circuit, depth {args.fw_depth}, coupler sequence ABCDCDAB (A, B: one coupler "
                                 f"orientation on alternate rows; C, D: the other), amplitude TN, single-qubit gates absorbed: "
                                 if args.fw_layout == "supremacy" else
                                 f"Sycamore-53-style depth-{args.fw_depth} circuit, the two coupler orientations ALTERNATING per cycle "
                                 f"(the easier network of rounds 1-3), amplitude TN: ") +
                                f"{prob.n} tensors, {sum(len(t) == 4 for t in prob.ts_inds)} two-qubit gates, {prob.n_inds} indices of "
                                f"dimension 2, {prob.W} mask words"),
                    "baseline_config": "configs[2]" if kind == "im" else "configs[4]",
                    "optimizer": "infinite_memory" if kind == "im" else "finite_width/greedy",
                    "rule": "MetropolisHastings", "cost_dtype": "float64",
                    "beta": [0.0, 100.0, leg.total_sweeps], "replicas_per_gpu": R, "n_leaves": prob.n, "n_inds": prob.n_inds,
                    "mask_words": prob.W,
                    **({} if kind == "im" else {"max_width": args.fw_max_width, "width_dtype": "float32",
                                                "update_slices_every": every}),
                    "replicas_total": R * world, "sweeps_per_step": sps, "moves_timed": moves, "accept_rate": a,
                    "random_pick_rate": q, "best_log10_flops": float(np.log10(res["best"])),
                    "improvements_timed": res["improved"], "full_tree_copies_timed": res["full_copies"],
                    "validated_bad_replicas": res.get("n_bad"),
                    "validated": (f"tnco_hip_validate(atol=1e-5) over all {R * world} replicas (every rank its own, counts summed) after the timed region, "
                                  f"{res.get('validate_s', 0.0):.2f} s (untimed)") if res.get("n_bad") is not None else None,
                    "library": lib_version,
                    "initial_trees": "random Kruskal (tnco_hip_random_trees)" if args.init == "kruskal" else
                                     "Random(seed).shuffle + opt_einsum greedy restated (tnco_hip_greedy_trees_device)",
                },
                "roofline": roof,
            }
            if grouped:
                obj["config"]["ranks"] = res["per_rank"]
            if args.cpu_sample != 0 and world == 1:
                cores = usable_cores()
                if kind == "im":
                    fn = lambda ns: cpu_baseline_im(prob, leg.links, leg.seeds, leg.betas, ns, cores)  # noqa: E731
                    probe, budget = min(256, R), 15.0
                else:
                    fn = lambda ns: cpu_baseline_fw(prob, leg.links, leg.seeds, leg.betas, ns, cores,  # noqa: E731
                                                    args.fw_max_width, every)
                    probe, budget = min(4 * cores, R), 10.0
                if args.cpu_sample > 0:
                    ns = min(args.cpu_sample, R)
                else:  # auto: a probe sets the sample so that the timed run is ~10-15 s of CPU work
                    pv, pm, _pt, _ = fn(probe)
                    ns = int(min(R, max(probe, budget * pv / (pm / probe))))
                v, m, t, cpu_min = fn(ns)
                gpu_min = leg.opt.costs()[1][:ns]
                obj["config"]["cpu_sample_min_cost_bit_exact"] = bool(np.array_equal(cpu_min, gpu_min))
                obj["cpu_baseline"] = {
                    "value": v, "unit": "move-evals/s", "cores": cores, "kind": "port",
                    "sample": f"oracle/tnco_oracle.c (plain-C restatement of the reference's "
                              f"{'infinite-memory' if kind == 'im' else 'finite-width'} optimizer), {ns} of the same "
                              f"replicas, full {leg.total_sweeps}-sweep schedule each, {cores} threads; {m} moves in {t:.1f} s",
                    "per_core": v / max(cores, 1),
                }
                if kind == "im":  # (what is known about port / reference: BASELINE.md sections 2-3)
                    obj["cpu_baseline"]["reference_note"] = (
                        "the real reference cannot be built here (Boost absent; no stand-ins), so rho = port / reference is "
                        "not measured; the survey-time probe of the reference's own classes (BASELINE.md section 2: stand-in "
                        "dynamic_bitset header, one update() per Python call) ran 2-2.5e6 move-evals/s per core on this "
                        "network -- the port, which allocates nothing per move, is several times faster per core: a "
                        "conservative baseline")
            return obj

        head_kind = legs[0]
        head = leg_object(head_kind)
        out = {
            "metric": METRIC, "value": head["value"], "unit": head["unit"], "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": head["config"],
            "roofline": head["roofline"],
        }
        if "cpu_baseline" in head:
            out["cpu_baseline"] = head["cpu_baseline"]
        # what carried the exchange between the ranks: an N > 1 line is an N-GPU measurement only if rccl_ranks == N
        out["transport"], out["rccl_ranks"], _code = transport_verdict(comm_kind if grouped else "none (one rank, nothing to exchange)",
                                                                        world, comm_requested)
        if devices is not None:
            out["config"]["devices"] = devices
            if comm_note:
                out["config"]["comm_note"] = comm_note
        for kind in legs[1:]:
            out[kind] = leg_object(kind)
        if args.e2e and world == 1 and args.replicas <= 131072:  # (beside the legs' own handles: not for the 100-GB batches)
            out["end_to_end"] = end_to_end(args)
            out.update(side_legs(local_rank, [x for x in args.extras.split(",") if x]))
        if calib is not None:
            out["calibration"] = calib
            if calib.get("move_pattern_per_s"):
                # the headline against the memory side of its own moves on this box (1.0 = the kernel costs nothing but
                # its memory requests); a second probe after everything else shows how far the box drifted meanwhile
                out["calibration"]["value_over_move_pattern"] = (results["im"]["moves"] / results["im"]["dt"] / calib["move_pattern_per_s"]
                                                                 if "im" in results else None)
                again = calibration(local_rank)
                out["calibration"]["after"] = {k: again.get(k) for k in ("random_lines_per_s", "move_pattern_per_s", "stream_read_GBs",
                                                                          "valu_wave_insts_per_s_per_simd", "dependent_load_ns", "error")
                                               if again.get(k) is not None}
        print(json.dumps(out), flush=True)
    for leg in objs.values():
        if leg.opt is not None:
            leg.opt.close()
    # A launch on N > 1 GPUs that did not talk through RCCL exits non-zero (after the line: the numbers are real, the
    # transport is not what was asked for), unless sockets / gloo were requested (tests).  A rank with a thread still
    # inside ncclCommInitRank cannot run RCCL's exit handlers: it leaves through os._exit, also non-zero.
    exit_code = transport_verdict(comm_kind, world, comm_requested)[2] if grouped else 0
    if parallel._native is not None:
        hung = bool(getattr(parallel._native, "hung", False))
        parallel._native.barrier()
        parallel.shutdown_native()
        if hung:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(exit_code or 3)
    elif grouped:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.stdout.flush()
        print(f"bench.py: {world} ranks exchanged their results over '{comm_kind}', not RCCL"
              + (f" ({comm_note})" if comm_note else "") + ": exit code 3", file=sys.stderr)
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
