"""Two ranks (torch.distributed, gloo rendezvous on 127.0.0.1) both driving cuda:0 run the plugin API
end to end: the run list is sharded, every rank gets the same merged results, and they are the
results of the single-process call -- results do not depend on the number of ranks
(tnco_amd/parallel.py; the reference fans runs out to processes and sorts, tnco/parallel.py:111-368,
tnco/app/infinite_memory/sa.py:243-257).  RCCL needs one GPU per rank: on this 1-GPU box it runs as a
group of ONE rank (the last test), N > 1 over RCCL is bench.py under the driver."""
import os
import socket
import sys
import warnings
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

SPEC_N, SPEC_SEED = 48, 5


def _spec():
    from tnco_amd import synthetic as syn
    ts, _d, _o = syn.random_regular_tn(SPEC_N, 3, SPEC_SEED)
    n_inds = 1 + max(i for xs in ts for i in xs)
    return [(2, *[f"t{t}" for t in range(SPEC_N) if k in ts[t]]) for k in range(n_inds)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(max_width):
    from tnco_amd.app import Optimizer
    kw = dict(update_slices=5) if max_width else {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _tn, res = Optimizer(method="sa", seed=9, max_width=max_width).optimize(
            _spec(), betas=(0, 60), n_steps=120, n_runs=37, top_k=12, device=0, fuse=None, **kw)
    return [(str(r.cost), [tuple(p) for p in r.path], sorted(getattr(r, "slices", ()))) for r in res]


def _worker(rank, world, port, q, max_width):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _run(max_width)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("max_width", [None, 6])
def test_two_ranks_equal_one_process(max_width):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, max_width)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = _run(max_width)  # (after the children: this process touches the GPU only now)
    assert len(want) == 12
    assert out[0] == want and out[1] == want


@pytest.mark.timeout(600)
def test_bench_line_of_a_two_rank_launch():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank), here
    with both ranks on the one GPU of the box over gloo (TNCO_BENCH_SHARE_GPU): the N > 1 code path
    -- sharded seeds, barriers, all-gathers, the per-rank record -- produces one valid line whose
    totals are the sums over the ranks."""
    import json
    import subprocess
    env = dict(os.environ, TNCO_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--replicas", "2048", "--sweeps-per-step", "20", "--pmc", "0", "--cpu-sample", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=540, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["vs_baseline"] is None
    for obj in (j, j["fw"]):
        ranks = obj["config"]["ranks"]
        assert [r["rank"] for r in ranks] == [0, 1] and all(r["moves"] > 0 for r in ranks)
        assert sum(r["moves"] for r in ranks) == obj["config"]["moves_timed"]
        assert obj["config"]["replicas_total"] == 4096
        assert obj["value"] > 0 and obj["roofline"]["frac"] > 0
    assert len(j["config"]["devices"]) == 2


@pytest.mark.timeout(600)
def test_bench_line_of_a_two_rank_launch_over_the_socket_transport():
    """The same launch with the ranks' exchanges on the transport of last resort (tnco_amd.parallel.SocketComm: what
    init_native falls back to on ALL ranks when RCCL does not come up on every one of them), both ranks on the one
    GPU: one valid line, the transport and the reason named in it."""
    import json
    import subprocess
    env = dict(os.environ, TNCO_BENCH_SHARE_GPU="sockets", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--replicas", "2048", "--sweeps-per-step", "20", "--pmc", "0", "--cpu-sample", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=540, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and len(j["config"]["devices"]) == 2
    assert all("tcp sockets" in d["backend"] for d in j["config"]["devices"])
    assert "TNCO_COMM=sockets" in j["config"]["comm_note"]
    for obj in (j, j["fw"]):
        ranks = obj["config"]["ranks"]
        assert [r["rank"] for r in ranks] == [0, 1] and sum(r["moves"] for r in ranks) == obj["config"]["moves_timed"]


def _run_with_traceback_on_timeout(cmd, timeout, env):
    """subprocess.run that, when the child does not finish, makes it say where it hangs: SIGABRT -> faulthandler
    prints every thread's Python stack, and the failure message carries it."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=str(ROOT))
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.send_signal(signal.SIGABRT)
        try:
            out, err = p.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
        pytest.fail(f"{' '.join(cmd[-12:])} with {[k for k in env if k.startswith('TNCO_')]} did not finish within "
                    f"{timeout} s; it was at:\n{err[-4000:]}")
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)


@pytest.mark.timeout(900)
def test_bench_line_through_rccl_group_of_one():
    """The RCCL side of bench.py on a 1-GPU box: a launch of one rank that still goes through the communicator
    (TNCO_BENCH_FORCE_GROUP) -- natively (RCCL bound inside libtnco_hip.so: no torch in the process, the best cost
    reduced on the device into the all-reduce's operand, barriers and all-gathers through the library) and, as
    rounds 1-2 did, through torch.distributed's "nccl" -- gives the line of the plain launch either way."""
    import json
    import subprocess
    tail = [str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--replicas", "2048",
            "--sweeps-per-step", "20", "--pmc", "0", "--cpu-sample", "0"]
    lines = []
    for extra in ({"TNCO_BENCH_FORCE_GROUP": "1", "MASTER_PORT": str(_free_port())},
                  {"TNCO_BENCH_FORCE_GROUP": "1", "MASTER_PORT": str(_free_port()), "TNCO_BENCH_COMM": "torch"}, {}):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        p = _run_with_traceback_on_timeout([sys.executable, "-X", "faulthandler", *tail], 280, env)
        assert p.returncode == 0, p.stderr[-2000:]
        assert len(p.stdout.strip().splitlines()) == 1, p.stdout[-500:]  # ONE line on stdout, nothing else (RCCL's banner: stderr)
        lines.append(json.loads(p.stdout))
    grp, grp_torch, plain = lines
    for g, backend in ((grp, "librccl.so bound inside libtnco_hip.so"), (grp_torch, "torch.distributed nccl")):
        assert g["n_gpus"] == 1 and len(g["config"]["devices"]) == 1 and "comm_note" not in g["config"]
        assert backend in g["config"]["devices"][0]["backend"] and g["config"]["devices"][0]["rank"] == 0
        assert "CUs" in g["config"]["devices"][0]["device"]
    assert "devices" not in plain["config"]
    for a, b in ((grp, plain), (grp["fw"], plain["fw"]), (grp_torch, plain), (grp_torch["fw"], plain["fw"])):
        assert [r["rank"] for r in a["config"]["ranks"]] == [0]
        assert a["config"]["ranks"][0]["moves"] == a["config"]["moves_timed"] == b["config"]["moves_timed"]
        assert a["config"]["best_log10_flops"] == b["config"]["best_log10_flops"]


def test_native_communicator_of_one_rank():
    """tnco_amd.parallel.NativeComm (csrc/host_comm.cpp) as a group of one: the collectives are RCCL's, on the library's
    own HIP runtime -- min of a handle's replicas reduced on the device, array and object all-gathers, barrier.  Run in
    a child process: it must work WITHOUT torch in the process."""
    import subprocess
    code = """
import sys
sys.path.insert(0, %r)
import numpy as np
from tnco_amd import core, parallel
from tests import helpers as H
assert 'torch' not in sys.modules
c = parallel.NativeComm(0, 1, 0, port=%d)
prob = H.regular_problem(32, graph_seed=2)
seeds = H.replica_seeds(512)
opt = core.BatchedOptimizer(prob.leaf_masks, prob.links(seeds), seeds, n_inds=prob.n_inds)
opt.run(H.linear_betas(0, 50, 40))
assert c.allreduce_min(opt) == opt.costs()[1].min() == opt.best(1)[0][0]
assert c.allreduce_min(3.5) == 3.5
a = np.arange(12, dtype=np.int32).reshape(3, 4)
assert np.array_equal(c.allgather_array(a), a[None])
assert c.allgather_object({'x': [1, 2, (3, 'y')]}) == [{'x': [1, 2, (3, 'y')]}]
c.barrier()
c.close()
assert 'torch' not in sys.modules
print('ok')
""" % (str(ROOT), _free_port())
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=str(ROOT))
    # (RCCL prints its version banner on stdout when the process ends)
    assert p.returncode == 0 and "ok" in p.stdout.split(), p.stderr[-2000:]
