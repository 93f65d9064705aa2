"""Fuzz the GPU path against the oracle with fresh random option combinations (the bodies of
tests/test_gpu_random.py), printing the arguments of every failing case.

    python tools/fuzz_gpu.py [--cases 2000] [--seed 0] [--which im|fw|both]
"""
import argparse
import pathlib
import random
import sys
import traceback

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from oracle import oracle as orc  # noqa: E402  (checker: this tool is test infrastructure)
from tests import test_gpu_random as T  # noqa: E402
from tnco_amd import core  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--which", default="both")
    ap.add_argument("--nmin", type=int, default=0, help="override the range of the number of tensors")
    ap.add_argument("--nmax", type=int, default=0, help="(wide masks: 8- and 16-lane layouts from ~450 tensors on)")
    ap.add_argument("--dims", default="", help="restrict dims_kind (e.g. 'two': big random trees overflow otherwise)")
    ap.add_argument("--new-slices", type=int, default=-1, help="force max_number_new_slices of the finite-width cases")
    ap.add_argument("--verbose", action="store_true", help="print every case before it runs (a GPU fault kills the process)")
    a = ap.parse_args()
    orc.build()
    rng = random.Random(a.seed)
    im = T.test_random_infinite_memory.hypothesis.inner_test
    fw = T.test_random_finite_width.hypothesis.inner_test
    bad = skipped = 0
    for i in range(a.cases):
        if a.which in ("im", "both"):
            kw = dict(seed=rng.randrange(10**6), n=rng.randint(a.nmin or 4, a.nmax or 40), k=rng.choice([2, 3, 4]),
                      dims_kind=a.dims or rng.choice(["two", "three", "four", "vector"]), n_sparse=rng.choice([0, 0, 3, 8]),
                      cost_type=rng.choice(["float64", "float64", "float32"]),
                      kind=rng.choice(["mh", "mh", "greedy", "base"]), dsi=rng.random() < 0.5)
            if a.verbose:
                print("infinite_memory", kw, flush=True)
            try:
                im(core, orc, **kw)
            except Exception as e:
                if "Precision is too low" in str(e):
                    skipped += 1
                    continue
                bad += 1
                print("FAIL infinite_memory", kw)
                print("   ", traceback.format_exc().strip().splitlines()[-1])
        if a.which in ("fw", "both"):
            kw = dict(seed=rng.randrange(10**6), n=rng.randint(a.nmin or 6, a.nmax or 36), k=rng.choice([2, 3]),
                      dims_kind=a.dims or rng.choice(["two", "two", "four", "vector"]), n_sparse=rng.choice([0, 0, 4]),
                      frac=rng.uniform(0.3, 1.1), every=rng.choice([1, 3, 10]),
                      width_type=rng.choice(["float32", "float64"]), new_slices=rng.choice([0, 0, 2]))
            if a.new_slices >= 0:
                kw["new_slices"] = a.new_slices
            if a.verbose:
                print("finite_width", kw, flush=True)
            try:
                fw(core, orc, **kw)
            except Exception as e:
                if "Precision is too low" in str(e):  # (the reference's own verdict on costs beyond the float range)
                    skipped += 1
                    continue
                bad += 1
                print("FAIL finite_width", kw)
                print("   ", traceback.format_exc().strip().splitlines()[-1])
    print(f"{a.cases} cases each, {bad} failures, {skipped} skipped ('Precision is too low.')")


if __name__ == "__main__":
    main()
