"""world_size-2 CPU test (gloo) of the multi-GPU path's host logic: shard, reduce, merge, broadcast."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from tnco_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert parallel.rank_world() == (rank, world)
        n_runs = 11
        lo, hi = parallel.shard_bounds(n_runs, world, rank)
        costs = np.array([50.0, 7.0, 9.0, 30.0, 7.0, 12.0, 99.0, 8.0, 41.0, 7.5, 60.0])  # run -> cost
        mine = costs[lo:hi]
        best = parallel.global_best(float(mine.min()), rank, world)
        gid = lo + int(np.argmin(mine))
        payload = np.full((3, 5), gid, np.int32)
        wc, wid, wp = parallel.global_winner(float(mine.min()), gid, payload, rank, world)
        local = sorted(((float(c), lo + k, [c], [[(0, 1)]]) for k, c in enumerate(mine)))[:3]
        merged = parallel.merge_heads(local, 3, rank, world)
        q.put((rank, best, wc, wid, wp.tolist(), [(c, g) for c, g, _, _ in merged]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_reduction_and_merge():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, best, wc, wid, wp, merged in out:
        assert best == 7.0 and wc == 7.0 and wid == 1          # tie 7.0 at runs 1 and 4 -> lowest id
        assert wp == [[1] * 5] * 3                              # winner's payload broadcast from rank 0
        assert merged == [(7.0, 1), (7.0, 4), (7.5, 9)]          # same head on every rank


class _StubOptimizer:
    """What bench.py's reduction sees of a handle: best(k) and the work counters."""

    def __init__(self, rank):
        self.rank = rank

    def best(self, k):
        return np.array([10.0 - 3 * self.rank]), np.array([5])


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import bench
    from tnco_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kt = {k: (1.0 + rank if k == "sa_run_kernel" else 0.0, 3 if k == "sa_run_kernel" else 0) for k in bench.KERNELS}
        res = dict(dt=0.5 + 0.25 * rank, moves=1000 * (rank + 1), accepted=100 * (rank + 1), random_picks=10,
                   improved=7, full_copies=0, kt=kt, best=parallel.global_best(_StubOptimizer(rank), rank, world))
        out = bench.reduce_legs(res, world, dist, torch)
        q.put((rank, out["dt"], out["moves"], out["accepted"], out["kt"]["sa_run_kernel"][0], out["best"],
               [(p["rank"], p["moves"]) for p in out["per_rank"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bench_reduction_over_two_ranks():
    """bench.py's N > 1 path without GPUs: the best cost is the min over ranks, the timed region the
    max, the work the sum, and the line lists what every rank did."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, dt, moves, acc, kms, best, per_rank in out:
        assert dt == 0.75 and moves == 3000 and acc == 300 and kms == 2.0 and best == 7.0
        assert per_rank == [(0, 1000.0), (1, 2000.0)]


class _FileCollectives:
    """Stand-in for the tnco_hip_comm_* entry points of libtnco_hip.so (csrc/host_comm.cpp: RCCL, needs a GPU per rank):
    the same C signatures over files in a shared directory, so that tnco_amd.parallel.NativeComm -- the rendezvous over
    a TCP socket, the two-phase object all-gather, the reductions built on it -- runs with world_size 2 on a CPU box."""

    def __init__(self, root, rank, world):
        import ctypes
        self.C, self.root, self.rank, self.world, self.seq, self.uid = ctypes, Path(root), rank, world, 0, None

    def tnco_hip_comm_last_error(self):
        return b"stand-in"

    def tnco_hip_comm_unique_id(self, uid):
        for i in range(128):
            uid[i] = (i * 7 + 3) & 0xFF
        return 0

    def tnco_hip_comm_init(self, rank, world, uid, device, out):
        self.uid = bytes(uid)
        out._obj.value = 1
        return 0

    def tnco_hip_comm_destroy(self, h):
        pass

    def _exchange(self, blob: bytes):
        import time
        self.seq += 1
        tmp = self.root / f"{self.seq}_{self.rank}.tmp"
        tmp.write_bytes(blob)
        tmp.rename(self.root / f"{self.seq}_{self.rank}.bin")
        out = []
        for k in range(self.world):
            f = self.root / f"{self.seq}_{k}.bin"
            t0 = time.monotonic()
            while not f.exists():
                assert time.monotonic() - t0 < 60
                time.sleep(0.005)
            out.append(f.read_bytes())
        return out

    def tnco_hip_comm_allgather(self, h, send, recv, nbytes):
        parts = self._exchange(self.C.string_at(send, nbytes))
        self.C.memmove(recv, b"".join(parts), nbytes * self.world)
        return 0

    def tnco_hip_comm_allreduce_min(self, h, handle, local, out):
        import struct
        assert handle is None
        vals = [struct.unpack("d", b)[0] for b in self._exchange(struct.pack("d", local))]
        out._obj.value = min(vals)
        return 0

    def tnco_hip_comm_barrier(self, h):
        self._exchange(b"x")
        return 0


def _native_worker(rank, world, port, root, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from tnco_amd import parallel
    fake = _FileCollectives(root, rank, world)
    parallel._native = parallel.NativeComm(rank, world, 0, lib=fake)
    try:
        assert "torch" not in sys.modules and parallel.rank_world() == (rank, world)
        assert fake.uid == bytes((i * 7 + 3) & 0xFF for i in range(128))  # rank 0's id reached every rank
        n_runs = 11
        lo, hi = parallel.shard_bounds(n_runs, world, rank)
        costs = np.array([50.0, 7.0, 9.0, 30.0, 7.0, 12.0, 99.0, 8.0, 41.0, 7.5, 60.0])
        mine = costs[lo:hi]
        best = parallel.global_best(float(mine.min()), rank, world)
        gid = lo + int(np.argmin(mine))
        wc, wid, wp = parallel.global_winner(float(mine.min()), gid, np.full((3, 5), gid, np.int32), rank, world)
        # (heads of different pickled sizes on the two ranks: the object all-gather pads to the longest)
        local = sorted(((float(c), lo + k, [c] * (1 + rank * 5), [[(0, 1)]]) for k, c in enumerate(mine)))[:3]
        merged = parallel.merge_heads(local, 3, rank, world)
        parallel._native.barrier()
        q.put((rank, best, wc, wid, wp.tolist(), [(c, g) for c, g, _, _ in merged]))
    finally:
        parallel.shutdown_native()


@pytest.mark.timeout(120)
def test_two_ranks_through_the_native_communicator_logic(tmp_path):
    """tnco_amd.parallel over NativeComm with world_size 2: socket rendezvous of the 128-byte id, rank_world,
    global_best / global_winner / merge_heads on the native branch -- the collectives themselves replaced by files
    (the real ones are RCCL's: tests/test_gpu_two_ranks.py runs them as a group of one on the GPU box)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_native_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, best, wc, wid, wp, merged in out:
        assert best == 7.0 and wc == 7.0 and wid == 1
        assert wp == [[1] * 5] * 3
        assert merged == [(7.0, 1), (7.0, 4), (7.5, 9)]


class _FlakyLib(_FileCollectives):
    """The stand-in library with RCCL failing in a chosen way on ONE rank."""

    def __init__(self, root, rank, world, mode, bad_rank):
        super().__init__(root, rank, world)
        self.mode, self.bad = mode, rank == bad_rank

    def tnco_hip_comm_last_error(self):
        return b"librccl.so: cannot open shared object file"

    def tnco_hip_comm_unique_id(self, uid):
        if self.bad and self.mode == "no_library":
            return 1
        return super().tnco_hip_comm_unique_id(uid)

    def tnco_hip_comm_init(self, rank, world, uid, device, out):
        if self.mode == "init_hangs":  # (ncclCommInitRank waits for a rank that never comes)
            import time
            time.sleep(3600 if not self.bad else 0)
            return 1
        return super().tnco_hip_comm_init(rank, world, uid, device, out)


def _init_native_worker(rank, world, port, root, mode, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from tnco_amd import parallel
    lib = _FlakyLib(root, rank, world, mode, bad_rank=1)
    c = parallel.init_native(rank, world, 0, lib=lib, timeout=3.0 if mode == "init_hangs" else 60.0)
    try:
        best = parallel.global_best(10.0 - rank, rank, world)
        wc, wid, wp = parallel.global_winner(10.0 - rank, 100 + rank, np.full(4, rank, np.int32), rank, world)
        merged = parallel.merge_heads([(10.0 - rank, 100 + rank, "x" * (rank * 50))], 2, rank, world)
        c.barrier()
        q.put((rank, type(c).__name__, c.kind, getattr(c, "note", None), best, wc, wid, wp.tolist(), [t[:2] for t in merged]))
    finally:
        parallel.shutdown_native()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("mode", ["fine", "no_library", "init_hangs"])
def test_three_ranks_agree_on_the_transport(tmp_path, mode):
    """init_native with world_size 3: RCCL (here the stand-in library) when it comes up on every rank; when ONE rank
    cannot load it, or ncclCommInitRank does not return, ALL ranks go through the sockets -- same results, the reason
    in `.note` -- instead of a launch that hangs."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_init_native_worker, args=(r, 3, port, str(tmp_path), mode, q), daemon=True) for r in range(3)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
    for rank, cls, kind, note, best, wc, wid, wp, merged in out:
        assert best == 8.0 and (wc, wid) == (8.0, 102) and wp == [2] * 4
        assert merged == [(8.0, 102), (9.0, 101)]
        if mode == "fine":
            assert cls == "NativeComm" and note is None and kind.startswith("rccl")
        else:
            assert cls == "SocketComm" and kind.startswith("tcp") and "rank" in note
            assert ("cannot open" in note) if mode == "no_library" else ("did not return" in note or "rank 1" in note)


def test_a_run_that_lost_rccl_cannot_pass_for_an_n_gpu_measurement():
    """bench.py's verdict on the transport: N > 1 ranks that exchanged their results over anything
    but RCCL exit non-zero -- unless sockets / gloo were ASKED for -- and the line says how many ranks RCCL carried."""
    sys.path.insert(0, str(ROOT))
    import bench
    rccl = "rccl (librccl.so bound inside libtnco_hip.so)"
    assert bench.transport_verdict(rccl, 8, "native") == (rccl, 8, 0)
    assert bench.transport_verdict("torch.distributed nccl", 4, "torch") == ("torch.distributed nccl", 4, 0)
    tcp = "tcp sockets through rank 0"
    assert bench.transport_verdict(tcp, 8, "native") == (tcp, 0, 3)            # RCCL did not come up: the launch fails
    assert bench.transport_verdict(tcp, 2, "sockets") == (tcp, 0, 0)           # TNCO_COMM=sockets: asked for
    assert bench.transport_verdict("torch.distributed gloo", 2, "gloo") == ("torch.distributed gloo", 0, 0)
    assert bench.transport_verdict("torch.distributed gloo", 2, "native")[2] == 3
    assert bench.transport_verdict(tcp, 1, "native")[1:] == (0, 0)             # a group of one: nothing to fake
    # ... and the exit path itself never leaves with 0 after a hung ncclCommInitRank
    src = (ROOT / "bench.py").read_text()
    assert "os._exit(0)" not in src and "os._exit(exit_code or 3)" in src


def _side_rank(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_RUN_ID="job-7")
    from tnco_amd import parallel
    c = parallel.SocketComm(rank, world, timeout=30.0)
    out = c.allgather_object({"rank": rank})
    q.put((rank, [o["rank"] for o in out], c.allreduce_min(10.0 - rank)))
    c.close()


@pytest.mark.timeout(120)
def test_side_channel_drops_strangers(tmp_path):
    """The TCP side channel: a connection that does not open with this job's token, claims a rank out of
    range or a rank that has already joined, or sends nothing at all, is dropped -- rank 0 neither registers it nor
    waits for it -- and messages are authenticated before they are unpickled."""
    import multiprocessing as mp
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    p0 = ctx.Process(target=_side_rank, args=(0, 3, port, q), daemon=True)
    p0.start()
    side = port + 18
    strangers = []
    t0 = time.monotonic()
    while time.monotonic() - t0 < 20:  # a silent stranger, a wrong token, and (below) a bad rank number
        try:
            s1 = socket.create_connection(("127.0.0.1", side), timeout=1.0)
            break
        except OSError:
            time.sleep(0.05)
    strangers.append(s1)
    s2 = socket.create_connection(("127.0.0.1", side), timeout=1.0)
    s2.sendall(b"x" * 16 + (1).to_bytes(4, "little") + b"y" * 16)
    strangers.append(s2)
    procs = [ctx.Process(target=_side_rank, args=(r, 3, port, q), daemon=True) for r in (1, 2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=60) for _ in range(3))
    for p in [p0] + procs:
        p.join(20)
    for s_ in strangers:
        s_.close()
    assert out == [(0, [0, 1, 2], 8.0), (1, [0, 1, 2], 8.0), (2, [0, 1, 2], 8.0)]
    # a forged message is refused, a replayed one too (the counter is part of the MAC), and nothing is unpickled
    sys.path.insert(0, str(ROOT))
    from tnco_amd import parallel
    a, b = socket.socketpair()
    tx, rx = parallel.SocketComm(1, 1), parallel.SocketComm(0, 1)
    tx._key = rx._key = b"k" * 32
    blob = parallel._encode(("hello", 1.5, frozenset({3}), np.arange(3)))
    a.sendall(len(blob).to_bytes(8, "little") + b"\0" * 32 + blob)
    with pytest.raises(ConnectionError):
        rx._recv_msg(b)
    rx._sess.clear()
    tx._send_msg(a, blob)
    first = parallel._decode(rx._recv_msg(b))
    assert first[:3] == ("hello", 1.5, frozenset({3})) and np.array_equal(first[3], np.arange(3))
    tx._sess[a.fileno()][1] = 0  # the same message once more: its number is 0 again, the receiver expects 1
    tx._send_msg(a, blob)
    with pytest.raises(ConnectionError):
        rx._recv_msg(b)
    assert "pickle" not in (ROOT / "tnco_amd" / "parallel.py").read_text().replace("never pickle", "").replace("unpickled", "")
    # off the loopback interface the launch parameters are not a secret: a token is demanded
    os.environ.pop("TNCO_COMM_TOKEN", None)
    with pytest.raises(RuntimeError, match="TNCO_COMM_TOKEN"):
        parallel.SocketComm(0, 2, addr="10.1.2.3", port=1)
    a.close(); b.close()


def test_config4_shards_of_eight(monkeypatch):
    """BASELINE config 4's split (524 288 runs over 8 ranks; tnco/app/infinite_memory/sa.py:237-257 draws ONE seed list
    and sorts ONE result list, tnco/parallel.py:330-341 fans the runs out): the eight shards are contiguous 65 536-run
    blocks whose seeds, concatenated, ARE the one-process list; the heads the ranks exchange merge into the head of
    the one-process sort (ties to the lower global run id); the minimum of the shard minima is the batch minimum --
    and bench.py slices its seeds the same way."""
    import random
    sys.path.insert(0, str(ROOT))
    from tnco_amd import parallel, synthetic
    from tnco_amd.app import _sa_driver
    n_runs, world = 524288, 8
    one = random.Random(0).choices(range(2**32), k=n_runs)          # the reference's line, one process
    assert _sa_driver.replica_seeds(random.Random(0), n_runs) == one  # (the driver's numpy shortcut for large k)
    assert synthetic.replica_seeds(n_runs, S=0) == one
    bounds = [parallel.shard_bounds(n_runs, world, k) for k in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n_runs and all(b[1] - b[0] == 65536 for b in bounds)
    assert all(bounds[k][1] == bounds[k + 1][0] for k in range(world - 1))
    assert [s for lo, hi in bounds for s in one[lo:hi]] == one
    # bench.py: rank k of an N-rank launch owns all_seeds[k * R:(k + 1) * R] of replica_seeds(R * N)
    assert all(one[k * 65536:(k + 1) * 65536] == one[slice(*bounds[k])] for k in range(world))
    # an uneven split (n_runs not a multiple of the world) still covers the list once, sizes differing by at most one
    ub = [parallel.shard_bounds(524289, world, k) for k in range(world)]
    assert [hi - lo for lo, hi in ub] == [65537] + [65536] * 7 and ub[-1][1] == 524289
    # results: every rank contributes its top_k (cost, global id, ...) tuples; all receive the head of sorted(results)
    rs = np.random.RandomState(5)
    cost = np.round(rs.random_sample(n_runs) * 50.0, 1)  # (many ties: 500 distinct values)
    top_k = 1024
    locals_ = []
    for lo, hi in bounds:
        order = np.lexsort((np.arange(lo, hi), cost[lo:hi]))[:top_k]
        locals_.append([(float(cost[lo + j]), int(lo + j), "payload") for j in order])

    class AllRanks:  # stand-in communicator: what an all-gather over the 8 ranks returns
        rank, world = 3, 8

        def allgather_object(self, obj):
            assert obj is locals_[self.rank]
            return locals_

    monkeypatch.setattr(parallel, "_native", AllRanks())
    merged = parallel.merge_heads(locals_[3], top_k, 3, world)
    head = np.lexsort((np.arange(n_runs), cost))[:top_k]
    assert [(c, g) for c, g, _ in merged] == [(float(cost[g]), int(g)) for g in head]
    assert min(float(cost[lo:hi].min()) for lo, hi in bounds) == float(cost.min()) == merged[0][0]


def test_wire_format_round_trips_what_the_ranks_exchange():
    """The fixed wire format of the exchanges (no pickle): everything merge_heads / the bench send survives a
    round trip with its types -- Decimals, tuples inside lists, frozensets of index names, infinities, numpy arrays and
    scalars, bytes, nested dicts -- and what it cannot carry is refused, not smuggled."""
    from decimal import Decimal
    sys.path.insert(0, str(ROOT))
    from tnco_amd import parallel
    head = [(Decimal("7.49888E+7"), 12345, [Decimal("1.5E+3"), 0], [[(0, 1), (0, 1)], []], [frozenset({"a", "b"}), frozenset()],
             [(2, 3), (0, 1)]),
            (Decimal("Infinity"), 7, [], [], [frozenset({("q", 1), 5})], [])]
    back = parallel._decode(parallel._encode(head))
    assert back == head and type(back[0]) is tuple and type(back[0][3][0][0]) is tuple and type(back[0][4][0]) is frozenset
    assert type(back[0][0]) is Decimal
    obj = {"rank": 3, "x": float("inf"), "y": -0.0, "z": float("1e-320"), "a": np.arange(6, dtype=np.int32).reshape(2, 3),
           "b": b"\x00\xff", "n": None, "t": True, "np": np.float64(2.5), "i": np.int64(-9), "s": {1, 2}}
    out = parallel._decode(parallel._encode(obj))
    assert out["rank"] == 3 and out["x"] == float("inf") and str(out["y"]) == "-0.0" and out["z"] == float("1e-320")
    assert np.array_equal(out["a"], obj["a"]) and out["a"].dtype == np.int32 and out["b"] == b"\x00\xff"
    assert out["n"] is None and out["t"] is True and out["np"] == 2.5 and out["i"] == -9 and out["s"] == {1, 2}
    assert np.isnan(parallel._decode(parallel._encode(float("nan"))))
    with pytest.raises(TypeError):
        parallel._encode(object())
    with pytest.raises(TypeError):
        parallel._encode(np.array([object()], dtype=object))
    with pytest.raises(ConnectionError):
        parallel._decode(b'{"!":"a","t":"|O","s":[1],"v":"00"}')
