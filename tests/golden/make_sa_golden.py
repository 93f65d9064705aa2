"""Golden state vectors of the SA path, produced by the CPU oracle
(oracle/tnco_oracle.c) in the build container and committed as tests/golden/sa_golden.json:

    python tests/golden/make_sa_golden.py

They do not pin the oracle to the reference (nothing in this image can: DESIGN.md section 5) -- they
pin BOTH the oracle and the HIP path to the state of the round they were generated in, so that a
later change to either that alters a single bit of a tree, a cost or the PRNG stream is caught even
if the two were changed together.  A case = inputs by recipe (synthetic generator + seeds, all in
tnco_amd/synthetic.py) + expected outputs: a 64-bit FNV-1a hash of (left, right, parent, legs) of the
current and of the best tree, the costs as hex floats, the PRNG position, slices, every `every` sweeps."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle as orc  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.golden_cases import CASES, problem_of, state_hash  # noqa: E402

orc.build()
out = []
for case in CASES:
    prob, seeds, links, betas, okw, every = problem_of(case)
    recs = []
    for r, s in enumerate(seeds):
        o = H.make_oracle(orc, prob, links[r], s, **okw)
        trace = []
        for k0 in range(0, len(betas), every):
            o.run(case.get("prob", 2), betas[k0:k0 + every], **({"update_slices_every": case["update_slices"]} if "max_width" in case else {}))
            trace.append(state_hash(o.tree(), o.tree(which_min=True), o.total_cost, o.min_total_cost, o.prng_state(),
                                    o.slices() if "max_width" in case else None))
        recs.append(trace)
    out.append({"case": case, "replicas": recs})
(Path(__file__).with_name("sa_golden.json")).write_text(json.dumps(out, indent=0) + "\n")
print(len(out), "cases,", sum(len(c["replicas"]) * len(c["replicas"][0]) for c in out), "state records")
