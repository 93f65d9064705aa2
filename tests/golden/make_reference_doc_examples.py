"""Collect the worked examples (`>>>` blocks) of the reference's docstrings that touch the hot path's
interfaces into
tests/golden/reference_doc_examples.json: inputs and expected outputs as DATA.

Run in the build container (reads /root/reference, which does not travel to the GPU box):
    python tests/golden/make_reference_doc_examples.py
The reference's modules cannot be imported here (more_itertools / opt_einsum / autoray are absent), so
the blocks are parsed as text; every `>>> call` / expected-output pair is kept verbatim together with
the file and line it comes from, and the statements that set the call's arguments up."""
import json
import re
from pathlib import Path

REF = Path("/root/reference/tnco")
WANT = {  # file -> functions / classes whose examples are kept
    "ctree.py": ["ContractionTree"],
    "utils/tn.py": ["get_random_contraction_path", "merge_contraction_paths", "split_contraction_path", "fuse", "contract"],
    "bitset.py": ["Bitset"],
    "optimize/prob.py": ["BaseProbability", "Greedy", "MetropolisHastings"],
    "app/app.py": ["load_tn", "Optimizer"],
}


def blocks(path):
    lines = path.read_text().split("\n")
    i = 0
    while i < len(lines):
        if lines[i].lstrip().startswith(">>>"):
            j = i
            stmts, pairs = [], []
            while j < len(lines) and lines[j].strip() and not lines[j].strip().startswith('"""'):
                s = lines[j].strip()
                if s.startswith(">>>"):
                    stmts.append([j + 1, s[3:].strip(), []])
                elif s.startswith("..."):
                    stmts[-1][1] += "\n" + s[3:].strip()
                else:
                    stmts[-1][2].append(s)
                j += 1
            yield i + 1, stmts
            i = j
        else:
            i += 1


out = []
for rel, names in WANT.items():
    for first, stmts in blocks(REF / rel):
        text = " ".join(s[1] for s in stmts)
        if not any(re.search(r"\b%s\b" % n, text) for n in names):
            continue
        out.append({
            "source": f"tnco/{rel}:{first}",
            "setup": [s[1] for s in stmts if not s[2] and not s[1].startswith(("from ", "import ", "#"))],
            "checks": [{"line": s[0], "expr": s[1], "expected": "\n".join(s[2])} for s in stmts if s[2]],
        })
Path(__file__).with_name("reference_doc_examples.json").write_text(json.dumps(out, indent=1) + "\n")
print(len(out), "examples,", sum(len(e["checks"]) for e in out), "checked expressions")
