"""Device generator of the initial trees against the host generator on random multigraphs (no hyper-indices: the
graph form; multi-edges, output legs, dangling legs, now and then a second component) and random hypergraphs (the set
form):  python tools/fuzz_greedy.py [networks] [seeds per network] [max tensors]"""
import pathlib
import sys
import time

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tests.test_gpu_greedy import _random_multigraph  # noqa: E402
from tnco_amd import _lib, core, ctree as ct, synthetic as syn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
S = int(sys.argv[2]) if len(sys.argv) > 2 else 24
MAXN = int(sys.argv[3]) if len(sys.argv) > 3 else 70
rng = np.random.default_rng(4)
seeds = syn.replica_seeds(S, S=9)
L = _lib.load()
t0 = time.time()
done = redone = trees = 0
sizes = []
for it in range(N):
    if it % 5 == 4:  # a hypergraph: the set form
        ts, _d, out = syn.random_hyper_tn(int(rng.integers(6, 40)), int(rng.integers(10, 60)), k=3, n_output=int(rng.integers(0, 5)),
                                          seed=int(rng.integers(1 << 30)))
        n_inds = 1 + max(i for xs in ts for i in xs)
        cnt = [sum(i in xs for xs in ts) for i in range(n_inds)]
        out = [i for i in out if cnt[i] <= 1]
    else:
        ts, n_inds, out = _random_multigraph(rng)
        if MAXN > 70 and it % 3 == 0:  # a larger one: the same generator, more tensors
            big = [_random_multigraph(rng) for _ in range(int(rng.integers(2, 1 + MAXN // 35)))]
            ts, n_inds, out = [list(x) for x in big[0][0]], big[0][1], list(big[0][2])
            for ts2, n2, out2 in big[1:]:
                base_t, base_i = len(ts), n_inds
                ts += [[i + base_i for i in xs] for xs in ts2]
                out += [i + base_i for i in out2]
                n_inds += n2
                ts[int(rng.integers(0, base_t))].append(n_inds)  # joined by one more index
                ts[base_t + int(rng.integers(0, len(ts2)))].append(n_inds)
                n_inds += 1
    off, _h = core.holders_csr(ts, n_inds)
    if not L.tnco_hip_diag_greedy_device_supported(len(ts), n_inds, off.ctypes.data):
        continue
    om = ct.pack_masks([list(out)], n_inds)[0]
    host = core.greedy_trees(ts, n_inds, seeds, output_mask=om)
    dev = core.greedy_trees(ts, n_inds, seeds, output_mask=om, device=0)
    assert np.array_equal(host, dev), (it, len(ts), n_inds)
    done += 1
    trees += S
    redone += int(L.tnco_hip_diag_greedy_device_redone())
    sizes.append(len(ts))
print(f"{done} networks ({min(sizes)}-{max(sizes)} tensors, every fifth a hypergraph) x {S} seeds = {trees} trees: device == host, "
      f"{redone} trees handed to the host inside the call (second components, long lists), {time.time() - t0:.0f} s")
