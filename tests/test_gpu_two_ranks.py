"""Two ranks (torch.distributed, gloo rendezvous on 127.0.0.1) both driving cuda:0 run the plugin API
end to end: the run list is sharded, every rank gets the same merged results, and they are the
results of the single-process call -- results do not depend on the number of ranks
(tnco_amd/parallel.py; the reference fans runs out to processes and sorts, tnco/parallel.py:111-368,
tnco/app/infinite_memory/sa.py:243-257).  RCCL itself needs one GPU per rank and is exercised by
bench.py under the driver."""
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
