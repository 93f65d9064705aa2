"""Replica sharding over the GPUs of one node and the best-cost reduction.

Replaces tnco/parallel.py (joblib/loky process fan-out + SharedMemory buffers,
/root/reference/tnco/parallel.py:111-368): replicas never interact, so rank k of
`world` owns a contiguous block of the run list and the only exchange is the
reduction of the best cost -- RCCL all-reduce(min) over xGMI -- plus the gather of
the heads of the result lists.  Two transports behind the same functions: the
NATIVE communicator (RCCL bound inside libtnco_hip.so, the default of a
multi-process launch) and a torch.distributed process group (gloo in the CPU
tests; nccl if a caller has initialised one).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import hmac
import ipaddress
import json
import os
import socket
import time
from decimal import Decimal

import numpy as np

__all__ = ["shard_bounds", "global_best", "global_winner", "NativeComm", "SocketComm", "init_native", "shutdown_native"]


class NativeComm:
    """One rank of the node-wide communicator, on RCCL bound INSIDE libtnco_hip.so (csrc/host_comm.cpp:
    dlopen of the ROCm installation's librccl.so, the HIP runtime the library itself links) -- nothing of
    PyTorch in the process, no pointer between two HIP runtimes.  The 128-byte ncclUniqueId travels from
    rank 0 to the others over a plain TCP socket on MASTER_ADDR : MASTER_PORT + 17 (`TNCO_COMM_PORT`
    overrides), so a launch by `torch.distributed.run` needs torch for the launcher only."""

    kind = "rccl (librccl.so bound inside libtnco_hip.so)"

    def __init__(self, rank: int, world: int, device: int, addr: str | None = None, port: int | None = None,
                 timeout: float = 90.0, lib=None, side: "SocketComm | None" = None):
        if lib is None:  # (tests pass a stand-in with the tnco_hip_comm_* entry points: the rendezvous and the
            from . import _lib  # host logic above the collectives run on a CPU box that way)
            lib = _lib.load()
        self._L = lib
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self._h = None
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port or os.environ.get("TNCO_COMM_PORT") or int(os.environ.get("MASTER_PORT", "29533")) + 17)
        uid = (C.c_uint8 * 128)()
        if side is not None:  # the id travels over the side channel the ranks already share (init_native)
            if self.rank == 0:
                self._check(self._L.tnco_hip_comm_unique_id(uid))
            data = side.allgather_object(bytes(uid) if self.rank == 0 else None)[0]
            C.memmove(uid, data, 128)
        elif self.rank == 0:
            self._check(self._L.tnco_hip_comm_unique_id(uid))
            if self.world > 1:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr, port))
                srv.listen(self.world)
                srv.settimeout(timeout)
                try:
                    for _ in range(self.world - 1):
                        conn, _peer = srv.accept()
                        with conn:
                            conn.sendall(bytes(uid))
                finally:
                    srv.close()
        else:
            t0, data = time.monotonic(), b""
            while len(data) < 128:
                try:
                    with socket.create_connection((addr, port), timeout=5.0) as c:
                        data = b""
                        while len(data) < 128:
                            chunk = c.recv(128 - len(data))
                            if not chunk:
                                break
                            data += chunk
                except OSError:
                    data = b""
                if len(data) < 128:
                    if time.monotonic() - t0 > timeout:
                        raise RuntimeError(f"rank {self.rank}: no ncclUniqueId from rank 0 at {addr}:{port}")
                    time.sleep(0.05)
            C.memmove(uid, data, 128)
        # ncclCommInitRank blocks until every rank has called it and has no time limit of its own: it runs on a
        # thread that is given `timeout` seconds (a rank that gives up says so to the others -- init_native)
        h = C.c_void_p()
        box: dict = {}

        def _init():
            try:
                box["rc"] = self._L.tnco_hip_comm_init(self.rank, self.world, uid, self.device, C.byref(h))
                if box["rc"]:  # (the text, from the thread that failed)
                    box["err"] = self._L.tnco_hip_comm_last_error().decode()
            except Exception as e:  # noqa: BLE001
                box["exc"] = e

        import threading
        th = threading.Thread(target=_init, name="tnco-rccl-init", daemon=True)
        th.start()
        th.join(timeout)
        if th.is_alive():
            # (the thread may still finish: init_native joins it once more after the ranks have voted and destroys the
            #  communicator if it did come up -- a rank that gave up must not leave a live communicator behind)
            _abandoned_inits.append((th, box, h, self._L))
            self.hung = True
            raise TimeoutError(f"rank {self.rank}: ncclCommInitRank did not return within {timeout:.0f} s")
        if "exc" in box:
            raise box["exc"]
        if box["rc"]:
            raise RuntimeError(box.get("err") or "ncclCommInitRank failed")
        self._h = h

    hung = False

    def _check(self, rc):
        if rc:
            raise RuntimeError(self._L.tnco_hip_comm_last_error().decode())

    def close(self):
        if self._h:
            self._L.tnco_hip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def barrier(self) -> None:
        self._check(self._L.tnco_hip_comm_barrier(self._h))

    def allreduce_min(self, opt_or_cost) -> float:
        """min over the ranks; an optimizer handle is reduced on the device straight into the collective's operand."""
        out = C.c_double(0.0)
        if hasattr(opt_or_cost, "_h"):
            self._check(self._L.tnco_hip_comm_allreduce_min(self._h, opt_or_cost._h, 0.0, C.byref(out)))
        else:
            self._check(self._L.tnco_hip_comm_allreduce_min(self._h, None, float(opt_or_cost), C.byref(out)))
        return out.value

    def allgather_array(self, a: np.ndarray) -> np.ndarray:
        """[world, *a.shape]: equal shapes and dtypes on every rank."""
        a = np.ascontiguousarray(a)
        out = np.empty((self.world,) + a.shape, a.dtype)
        self._check(self._L.tnco_hip_comm_allgather(self._h, a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                                    a.nbytes))
        return out

    def allgather_object(self, obj) -> list:
        blob = np.frombuffer(_encode(obj), np.uint8)
        sizes = self.allgather_array(np.array([len(blob)], np.int64))[:, 0]
        pad = np.zeros(int(sizes.max()), np.uint8)
        pad[:len(blob)] = blob
        allb = self.allgather_array(pad)
        return [_decode(allb[k, :int(sizes[k])].tobytes()) for k in range(self.world)]


def _is_loopback(addr: str) -> bool:
    if addr in ("localhost", "::1"):
        return True
    try:
        return ipaddress.ip_address(addr).is_loopback
    except ValueError:
        return False


def _single_node(addr: str) -> bool:
    """True when every rank of the job is on this machine: a loopback address, a launcher that says so
    (LOCAL_WORLD_SIZE == WORLD_SIZE: torchrun --standalone exports the node's FQDN as MASTER_ADDR), or a host name
    that resolves to loopback addresses only."""
    if _is_loopback(addr):
        return True
    lws, ws = os.environ.get("LOCAL_WORLD_SIZE"), os.environ.get("WORLD_SIZE")
    if lws and ws and lws == ws:
        return True
    try:
        found = {info[4][0] for info in socket.getaddrinfo(addr, None)}
    except OSError:
        return False
    return bool(found) and all(_is_loopback(a) for a in found)


def _encode(obj) -> bytes:
    """Fixed wire format of the side channel: JSON with tagged containers, numpy arrays as (dtype, shape, hex).
    Nothing received over a socket is ever unpickled."""
    def enc(x):
        if x is None or isinstance(x, (bool, str)):
            return x
        if isinstance(x, np.bool_):
            return bool(x)
        if isinstance(x, (int, np.integer)):
            return int(x)
        if isinstance(x, (float, np.floating)):
            return {"!": "f", "v": float(x).hex()}
        if isinstance(x, (bytes, bytearray)):
            return {"!": "b", "v": bytes(x).hex()}
        if isinstance(x, tuple):
            return {"!": "t", "v": [enc(y) for y in x]}
        if isinstance(x, list):
            return [enc(y) for y in x]
        if isinstance(x, (set, frozenset)):
            return {"!": "s" if isinstance(x, set) else "z", "v": [enc(y) for y in sorted(x, key=repr)]}
        if isinstance(x, dict):
            return {"!": "d", "v": [[enc(k), enc(v)] for k, v in x.items()]}
        if isinstance(x, Decimal):
            return {"!": "D", "v": str(x)}
        if isinstance(x, np.ndarray):
            if x.dtype.hasobject:
                raise TypeError("object arrays do not travel over the side channel")
            return {"!": "a", "t": x.dtype.str, "s": list(x.shape), "v": np.ascontiguousarray(x).tobytes().hex()}
        raise TypeError(f"side channel: cannot send a {type(x).__name__}")
    return json.dumps(enc(obj), separators=(",", ":")).encode()


def _decode(blob: bytes):
    def dec(x):
        if isinstance(x, list):
            return [dec(y) for y in x]
        if not isinstance(x, dict):
            return x
        k, v = x.get("!"), x.get("v")
        if k == "f":
            return float.fromhex(v)
        if k == "b":
            return bytes.fromhex(v)
        if k == "t":
            return tuple(dec(y) for y in v)
        if k == "s":
            return set(dec(y) for y in v)
        if k == "z":
            return frozenset(dec(y) for y in v)
        if k == "d":
            return {dec(a): dec(b) for a, b in v}
        if k == "D":
            return Decimal(v)
        if k == "a":
            dt = np.dtype(x["t"])
            if dt.hasobject:
                raise ConnectionError("side channel: object array refused")
            return np.frombuffer(bytes.fromhex(v), dt).reshape(x["s"]).copy()
        raise ConnectionError("side channel: unknown tag")
    return dec(json.loads(blob.decode()))


class SocketComm:
    """The same five calls over TCP through rank 0 (a star on MASTER_ADDR : MASTER_PORT + 18, `TNCO_COMM_SIDE_PORT`
    overrides): the side channel on which the ranks agree whether RCCL came up on ALL of them, and the transport of
    last resort when it did not -- what the ranks exchange is 16 bytes per launch chunk and the heads of the result
    lists at the end, so a launch on N GPUs still reports its numbers (and says which transport carried them).

    Trust.  One node is the design point: on a loopback rendezvous address the job key is derived from the
    launch (run id, address, port, world size) + `TNCO_COMM_TOKEN`; on any OTHER address `TNCO_COMM_TOKEN` -- a secret
    the launcher hands to every rank -- is mandatory.  Every connection opens with a challenge / response that mixes a
    nonce of each side into a per-connection session key, every message carries an HMAC under that key over (direction,
    message counter, body) -- a captured message can be neither replayed nor reflected --, and bodies are a fixed JSON
    format, never pickle: the worst a forger who knows the key can do is feed wrong numbers, not run code."""

    kind = "tcp sockets through rank 0"
    hung = False
    HELLO_TIMEOUT = 1.0  # a rank of this job sends its hello at once; a silent connection costs rank 0 this much

    def __init__(self, rank: int, world: int, addr: str | None = None, port: int | None = None, timeout: float = 90.0):
        self.rank, self.world = int(rank), int(world)
        self._peers: dict[int, socket.socket] = {}
        self._up: socket.socket | None = None
        self._sess: dict[int, list] = {}  # fileno -> [session key, messages sent, messages received]
        self._key = b""
        if self.world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port or os.environ.get("TNCO_COMM_SIDE_PORT") or int(os.environ.get("MASTER_PORT", "29533")) + 18)
        token = os.environ.get("TNCO_COMM_TOKEN", "")
        if not token and _single_node(str(addr)):
            addr = "127.0.0.1"  # (bind and connect on loopback, whatever name the launcher exported)
        if not token and not _is_loopback(str(addr)):
            raise RuntimeError(f"the side channel would listen on {addr}, which is not a loopback address: set TNCO_COMM_TOKEN "
                               "to a secret shared by the ranks of this job (the launch parameters alone can be guessed)")
        seed = "|".join([os.environ.get("TORCHELASTIC_RUN_ID", ""), str(addr), os.environ.get("MASTER_PORT", ""), str(self.world), token])
        self._key = hashlib.sha256(("tnco-side-channel|" + seed).encode()).digest()
        # (a collective waits for the slowest rank: TNCO_COMM_IO_TIMEOUT seconds, default one hour, 0 = for ever)
        io_timeout = float(os.environ.get("TNCO_COMM_IO_TIMEOUT", "3600")) or None
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world + 8)
            deadline = time.monotonic() + timeout
            try:
                while len(self._peers) < self.world - 1:
                    left = deadline - time.monotonic()
                    if left <= 0:
                        raise TimeoutError(f"rank 0: {self.world - 1 - len(self._peers)} ranks did not join the side channel at {addr}:{port}")
                    srv.settimeout(left)
                    conn, _peer = srv.accept()
                    try:
                        conn.settimeout(min(self.HELLO_TIMEOUT, max(left, 0.1)))
                        hello = self._recv_exact(conn, 36)  # nonce of the rank | rank | HMAC(job key, "hello" | nonce | rank)
                        nc, kb = hello[:16], hello[16:20]
                        k = int.from_bytes(kb, "little")
                        if (not hmac.compare_digest(hello[20:], self._mac(self._key, b"hello" + nc + kb)[:16])
                                or not 0 < k < self.world or k in self._peers):
                            raise ConnectionError("not a rank of this job")
                        ns = os.urandom(16)
                        conn.sendall(ns + self._mac(self._key, b"welcome" + nc + ns)[:16])
                    except (OSError, ConnectionError):
                        conn.close()
                        continue
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(io_timeout)
                    self._peers[k] = conn
                    self._sess[conn.fileno()] = [self._mac(self._key, b"session" + nc + ns), 0, 0]
            finally:
                srv.close()
        else:
            t0 = time.monotonic()
            while True:
                try:
                    c = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.monotonic() - t0 > timeout:
                        raise RuntimeError(f"rank {self.rank}: rank 0 does not answer at {addr}:{port}") from None
                    time.sleep(0.05)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(max(timeout, 5.0))
            nc, kb = os.urandom(16), self.rank.to_bytes(4, "little")
            c.sendall(nc + kb + self._mac(self._key, b"hello" + nc + kb)[:16])
            reply = self._recv_exact(c, 32)
            ns = reply[:16]
            if not hmac.compare_digest(reply[16:], self._mac(self._key, b"welcome" + nc + ns)[:16]):
                c.close()
                raise ConnectionError(f"rank {self.rank}: what answers at {addr}:{port} is not rank 0 of this job")
            c.settimeout(io_timeout)
            self._up = c
            self._sess[c.fileno()] = [self._mac(self._key, b"session" + nc + ns), 0, 0]

    @staticmethod
    def _mac(key: bytes, data: bytes) -> bytes:
        return hmac.new(key, data, hashlib.sha256).digest()

    @staticmethod
    def _recv_exact(c: socket.socket, n: int) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            chunk = c.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("peer closed the side channel")
            buf += chunk
        return bytes(buf)

    def _tag(self, c: socket.socket, sending: bool, blob: bytes) -> bytes:
        """HMAC under the connection's session key over (who speaks, message number, body)."""
        s = self._sess.setdefault(c.fileno(), [self._key, 0, 0])
        n = s[1] if sending else s[2]
        s[1 if sending else 2] += 1
        speaker = (self.rank == 0) == sending  # True: rank 0 is the speaker of this message
        return self._mac(s[0], (b"\1" if speaker else b"\0") + n.to_bytes(8, "little") + blob)

    def _send_msg(self, c: socket.socket, blob: bytes) -> None:
        c.sendall(len(blob).to_bytes(8, "little") + self._tag(c, True, blob) + blob)

    def _recv_msg(self, c: socket.socket) -> bytes:
        n = int.from_bytes(self._recv_exact(c, 8), "little")
        if n > (1 << 31):
            raise ConnectionError("side channel: message length out of range")
        mac = self._recv_exact(c, 32)
        blob = self._recv_exact(c, n)
        if not hmac.compare_digest(mac, self._tag(c, False, blob)):
            raise ConnectionError("side channel: message not from a rank of this job (or out of sequence)")
        return blob

    def allgather_object(self, obj) -> list:
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            parts = [obj] + [_decode(self._recv_msg(self._peers[k])) for k in range(1, self.world)]
            blob = _encode(parts)
            for k in range(1, self.world):
                self._send_msg(self._peers[k], blob)
            return _decode(blob)  # (what the other ranks see: the same types on every rank)
        self._send_msg(self._up, _encode(obj))
        return _decode(self._recv_msg(self._up))

    def allgather_array(self, a: np.ndarray) -> np.ndarray:
        return np.stack(self.allgather_object(np.ascontiguousarray(a)))

    def allreduce_min(self, opt_or_cost) -> float:
        c = float(opt_or_cost.best(1)[0][0]) if hasattr(opt_or_cost, "best") else float(opt_or_cost)
        return min(self.allgather_object(c))

    def barrier(self) -> None:
        self.allgather_object(None)

    def close(self) -> None:
        for c in list(self._peers.values()) + ([self._up] if self._up is not None else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._up, self._sess = {}, None, {}


_native: "NativeComm | SocketComm | None" = None
_side: SocketComm | None = None
_abandoned_inits: list = []  # (thread, result box, handle, library) of ncclCommInitRank calls that outlived their time limit


def _reap_abandoned_inits(grace: float = 5.0) -> bool:
    """After the ranks have agreed NOT to use RCCL: give the init threads that outlived their limit a last moment and
    destroy the communicators that did come up.  Returns whether a thread is still inside ncclCommInitRank."""
    still = False
    for th, box, h, lib in list(_abandoned_inits):
        th.join(grace)
        if th.is_alive():
            still = True
            continue
        _abandoned_inits.remove((th, box, h, lib))
        if box.get("rc") == 0 and h:
            lib.tnco_hip_comm_destroy(h)
    return still


def init_native(rank: int | None = None, world: int | None = None, device: int | None = None, **kw):
    """The process-wide communicator (rank / world / device default to the torchrun environment): RCCL inside the
    library if it comes up on EVERY rank, else the socket transport on all of them -- never a mix, which could only
    hang.  The ranks first meet on the side channel (SocketComm); each probes RCCL (a unique id can be made: the
    library is there); if all can, each runs ncclCommInitRank under a time limit (`TNCO_COMM_INIT_TIMEOUT`, 240 s)
    and they compare notes again.  `.kind` of the result says which transport it is, `.note` why RCCL was not."""
    global _native, _side
    if _native is not None:
        return _native
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
    device = local_device() if device is None else device
    lib = kw.pop("lib", None)
    timeout = float(kw.pop("timeout", os.environ.get("TNCO_COMM_INIT_TIMEOUT", "240")))
    side = SocketComm(rank, world, timeout=max(timeout, 90.0), **{k: v for k, v in kw.items() if k in ("addr",)})
    note = None
    try:
        if os.environ.get("TNCO_COMM", "") == "sockets":
            raise RuntimeError("TNCO_COMM=sockets")
        if lib is None:
            from . import _lib
            lib = _lib.load()
        probe = (C.c_uint8 * 128)()
        ok = lib.tnco_hip_comm_unique_id(probe) == 0
        why = None if ok else lib.tnco_hip_comm_last_error().decode()
    except Exception as e:  # noqa: BLE001 -- whatever it is, the other ranks must hear of it
        ok, why = False, repr(e)
    votes = side.allgather_object((ok, why))
    comm = None
    if all(v[0] for v in votes):
        try:
            comm = NativeComm(rank, world, device, lib=lib, side=side, timeout=timeout)
            ok, why = True, None
        except Exception as e:  # noqa: BLE001
            ok, why = False, repr(e)
        votes = side.allgather_object((ok, why))
    if all(v[0] for v in votes):
        _native, _side = comm, side
        comm.note = None
        return _native
    note = "; ".join(f"rank {k}: {v[1]}" for k, v in enumerate(votes) if not v[0])
    if comm is not None:
        comm.close()
    side.note = "RCCL not used -- " + note
    side.hung = _reap_abandoned_inits()  # (a thread of this process still sits in ncclCommInitRank)
    _native, _side = side, None
    return _native


def shutdown_native() -> None:
    global _native, _side
    for c in (_native, _side):
        if c is not None:
            c.close()
    _native = _side = None


def shard_bounds(n_runs: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of the run list owned by `rank`; sizes differ by at most one."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("'rank' / 'world' are not valid.")
    base, extra = divmod(int(n_runs), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _dist():
    import torch
    import torch.distributed as dist
    return torch, dist


def rank_world() -> tuple[int, int]:
    """(rank, world): the native communicator's if there is one; else a torch.distributed group's (the gloo
    tests, or a caller that initialised one); else, under a multi-process launch (WORLD_SIZE > 1 in the
    environment), the native communicator is created; else (0, 1)."""
    if _native is not None:
        return _native.rank, _native.world
    import sys
    if "torch" in sys.modules:
        _torch, dist = _dist()
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not os.environ.get("TNCO_COMM", "").startswith("torch"):
        c = init_native()
        return c.rank, c.world
    return 0, 1


def local_device() -> int:
    """GPU ordinal of this rank: LOCAL_RANK under torchrun, else 0."""
    return int(os.environ.get("LOCAL_RANK", "0"))


def merge_heads(local: list, k: int, rank: int, world: int) -> list:
    """Every rank contributes its best runs as tuples sorted by (cost, global run id, ...); all ranks
    receive the k best overall in that order (the head of `sorted(results)`,
    tnco/app/infinite_memory/sa.py:257)."""
    if world == 1:
        return sorted(local, key=lambda t: (t[0], t[1]))[:k]
    if _native is not None:
        gathered = _native.allgather_object(local)
    else:
        _torch, dist = _dist()
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
    merged = [t for part in gathered for t in part]
    return sorted(merged, key=lambda t: (t[0], t[1]))[:k]


def _tensor_device(dist, device):
    import torch
    if dist.get_backend() == "nccl":
        return torch.device("cuda", device)
    return torch.device("cpu")


def global_best(opt_or_cost, rank: int = 0, world: int = 1, device: int = 0, grouped: bool | None = None) -> float:
    """min over all ranks of the local best min_total_cost.

    Given an optimizer handle and an RCCL group, the local minimum is reduced on the device
    (tnco_hip_min_cost_device) straight into the tensor the all-reduce(min) runs on: 8 bytes over
    xGMI, no host round trip before the collective.  `grouped` (default: world > 1) = go through the
    process group; a group of ONE rank takes the same code path (tests/test_gpu_two_ranks.py runs
    RCCL that way on a 1-GPU box)."""
    is_opt = hasattr(opt_or_cost, "best")
    if not (world > 1 if grouped is None else grouped):
        return float(opt_or_cost.best(1)[0][0]) if is_opt else float(opt_or_cost)
    if _native is not None:  # RCCL inside the library: the device-side minimum is the collective's operand
        return _native.allreduce_min(opt_or_cost)
    torch, dist = _dist()
    dev = _tensor_device(dist, device)
    if is_opt and dev.type == "cuda" and hasattr(opt_or_cost, "min_cost_to_device"):
        t = torch.empty(1, dtype=torch.float64, device=dev)
        opt_or_cost.min_cost_to_device(t.data_ptr())
    else:
        c = float(opt_or_cost.best(1)[0][0]) if is_opt else float(opt_or_cost)
        t = torch.tensor([c], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t[0])


def global_winner(cost: float, global_id: int, payload: np.ndarray | None, rank: int, world: int,
                  device: int = 0):
    """(best cost, its global run id, winner's payload) on every rank.

    Ties go to the lowest global id (the order `sorted(results)` keeps,
    tnco/app/infinite_memory/sa.py:257).  `payload` is an int32 array of equal
    shape on every rank (e.g. the best tree's links), broadcast from the winner.
    """
    if world == 1:
        return cost, global_id, payload
    if _native is not None:
        pairs = _native.allgather_array(np.array([cost, float(global_id)], np.float64))
        best_cost, best_id, src = min((float(v[0]), int(v[1]), k) for k, v in enumerate(pairs))
        if payload is not None:
            payload = _native.allgather_array(np.ascontiguousarray(payload, np.int32))[src]
        return best_cost, best_id, payload
    torch, dist = _dist()
    dev = _tensor_device(dist, device)
    mine = torch.tensor([cost, float(global_id)], dtype=torch.float64, device=dev)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    pairs = [(float(v[0]), int(v[1]), k) for k, v in enumerate(allv)]
    best_cost, best_id, src = min(pairs)
    if payload is not None:
        t = torch.from_numpy(np.ascontiguousarray(payload, np.int32)).to(dev)
        dist.broadcast(t, src=src)
        payload = t.cpu().numpy()
    return best_cost, best_id, payload
