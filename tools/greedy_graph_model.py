"""Model of the GRAPH form of the initial-tree generator (csrc/greedy_graph.h) in plain Python, checked
against the set form (`ctree.ssa_greedy`, the restatement of opt_einsum's greedy) -- and the numbers the
kernel's LDS budget is sized by (list lengths, entries created, depth of the parent chains).

Where every contractible index has exactly two holders and an output index one (no hyper-indices), the
published algorithm is a greedy over a MULTIGRAPH: a tensor = its id, its number of legs and a list of
(neighbour, shared legs); contracting u, v makes z with the union of their lists minus each other, and
|z| = kept(u) + kept(v) - 2 w(u, v).  Index sets are never needed: a dead set cannot come back (the leg the
contraction removed is gone for good) and two live tensors with equal sets are an isolated pair.

    python tools/greedy_graph_model.py [n_seeds]
"""
import heapq
import pathlib
import sys
from random import Random

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from tnco_amd import ctree as ct  # noqa: E402
from tnco_amd import synthetic as syn  # noqa: E402


def graph_greedy(inputs, output, stats=None, cap_slack=256):
    """ssa path of the greedy over `inputs` (index sets, shuffled order), graph form.  None: not a graph."""
    n = len(inputs)
    out = frozenset(output)
    holders = {}
    for t, s in enumerate(inputs):
        for i in s:
            holders.setdefault(i, []).append(t)
    for i, h in holders.items():
        if len(h) > 2 or (i in out and len(h) > 1):
            return None
    fp = [len(s) for s in inputs] + [0] * n
    kp = [sum(1 for i in s if i in out or len(holders[i]) == 2) for s in inputs] + [0] * n
    lists = [None] * (2 * n)
    for t in range(n):
        w = {}
        for i in inputs[t]:
            if i not in out and len(holders[i]) == 2:
                o = holders[i][0] + holders[i][1] - t
                w[o] = w.get(o, 0) + 1
        lists[t] = list(w.items())
    for t in range(n):
        for o, m in lists[t]:
            if m == fp[t] == fp[o]:
                return None  # equal index sets among the inputs
    par = list(range(2 * n))
    alive = [True] * n + [False] * n
    key = lambda a, b, c, i, j: ((1 << a) - (1 << b) - (1 << c), max(i, j), min(i, j))  # noqa: E731
    queue = []
    for i, h in holders.items():
        if i not in out and len(h) == 2:
            x, y = h
            m = dict(lists[x])[y]
            heapq.heappush(queue, key(kp[x] + kp[y] - 2 * m, fp[x], fp[y], x, y))
    path = []
    z = n
    made = live = sum(len(x) for x in lists[:n])
    bump = live
    cap = live + cap_slack
    compactions = 0
    while queue:
        _c, v, u = heapq.heappop(queue)
        if stats is not None:
            stats["pops"] = stats.get("pops", 0) + 1
        if not alive[u] or not alive[v]:
            continue
        alive[u] = alive[v] = False
        par[u] = par[v] = z
        path.append((u, v))
        acc = {}
        shared = 0
        pre = len(lists[u]) + len(lists[v])
        for src in (u, v):
            for e, m in lists[src]:
                r, hops = e, 0
                while par[r] != r:
                    r = par[r]
                    hops += 1
                par[e] = r if e != r else par[e]
                if stats is not None:
                    stats["hops"] = stats.get("hops", 0) + hops
                    stats["finds"] = stats.get("finds", 0) + 1
                    stats["maxhops"] = max(stats.get("maxhops", 0), hops)
                if r == z:
                    shared += m
                else:
                    acc[r] = acc.get(r, 0) + m
        shared //= 2
        live -= len(lists[u]) + len(lists[v])
        lists[z] = list(acc.items())
        fp[z] = kp[z] = kp[u] + kp[v] - 2 * shared
        alive[z] = True
        live += len(lists[z])
        made += len(lists[z])
        if bump + len(lists[z]) > cap:
            compactions += 1
            bump = live - len(lists[z])
        bump += len(lists[z])
        if stats is not None:
            stats["maxpre"] = max(stats.get("maxpre", 0), pre)
            stats["maxlen"] = max(stats.get("maxlen", 0), len(lists[z]))
            stats["maxmult"] = max([stats.get("maxmult", 0)] + [m for _, m in lists[z]])
            stats["maxfp"] = max(stats.get("maxfp", 0), fp[z])
        if lists[z]:
            heapq.heappush(queue, min(key(fp[z] + kp[y] - 2 * m, fp[z], fp[y], z, y) for y, m in lists[z]))
        z += 1
    if stats is not None:
        stats["made"] = stats.get("made", 0) + made
        stats["compactions"] = stats.get("compactions", 0) + compactions
        stats["trees"] = stats.get("trees", 0) + 1
    if sum(alive) != 1:
        return None
    return path


def check(name, prob, seeds):
    out = set(ct.unpack_mask(prob.output_mask)) if prob.output_mask is not None else set()
    held = {}
    for s in prob.ts_inds:
        for i in s:
            held[i] = held.get(i, 0) + 1
    out = {i for i in out if held.get(i, 0) <= 1}
    stats = {}
    same = 0
    for seed in seeds:
        order = list(range(prob.n))
        Random(seed).shuffle(order)
        inputs = [frozenset(prob.ts_inds[t]) for t in order]
        o = frozenset(out) & frozenset().union(*inputs)
        ref = ct.ssa_greedy(inputs, o)
        got = graph_greedy(inputs, o, stats)
        if got is None:
            print(f"{name}: seed {seed}: not a graph / disconnected")
            continue
        norm = lambda p: [(min(a, b), max(a, b)) for a, b in p]  # noqa: E731
        assert norm(got) == norm(ref), (name, seed)
        same += 1
    t = max(stats.get("trees", 1), 1)
    print(f"{name}: {same}/{len(seeds)} trees equal to the set form | per tree: pops {stats['pops'] / t:.0f}, "
          f"list entries made {stats['made'] / t:.0f}, compactions (slack 256) {stats['compactions'] / t:.1f}, "
          f"hops per find {stats['hops'] / max(stats['finds'], 1):.2f} (max {stats['maxhops']}) | max: "
          f"entries of u and v together {stats['maxpre']}, list {stats['maxlen']}, shared legs {stats['maxmult']}, legs {stats['maxfp']}")


if __name__ == "__main__":
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    seeds = syn.replica_seeds(k)
    check("3-regular, 512 tensors", syn.regular_problem(512, 11), seeds)
    check("3-regular, 64 tensors", syn.regular_problem(64, 7), seeds)
    check("Sycamore-53 depth 20", syn.sycamore_problem(20), seeds)
    check("Sycamore-53 depth 12", syn.sycamore_problem(12), seeds[:3])
    check("4-regular, 300 tensors", syn.regular_problem(300, 5, degree=4), seeds[:3])
