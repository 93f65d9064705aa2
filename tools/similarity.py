"""Function-level similarity of this repository's Python to the reference's (runs only where /root/reference is).

For every function under tnco_amd/ and bench.py: the statements of its body (docstrings dropped, every statement
dumped from the AST, so layout and comments do not count -- identifiers do) as a multiset, against every function
of the reference's Python; reports the best match per function as common / own statements.  A function written
from the contract scores low; a transcription scores high.

    python tools/similarity.py [--min 0.3]
"""
from __future__ import annotations

import ast
import sys
from collections import Counter
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")


def _statements(fn: ast.AST) -> Counter:
    out = Counter()
    for node in ast.walk(fn):
        if node is fn or not isinstance(node, ast.stmt):
            continue
        if isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant) and isinstance(node.value.value, str):
            continue
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            continue
        # compound statements count by their header only (their bodies are walked on their own)
        if isinstance(node, (ast.For, ast.While, ast.If, ast.With, ast.Try)):
            head = {ast.For: lambda n: ("for", ast.dump(n.target), ast.dump(n.iter)),
                    ast.While: lambda n: ("while", ast.dump(n.test)),
                    ast.If: lambda n: ("if", ast.dump(n.test)),
                    ast.With: lambda n: ("with",) + tuple(ast.dump(i) for i in n.items),
                    ast.Try: lambda n: ("try",) + tuple(ast.dump(h.type) if h.type else "" for h in n.handlers)}[type(node)](node)
            out[head] += 1
        else:
            out[ast.dump(node)] += 1
    return out


def functions(paths):
    for path in paths:
        try:
            tree = ast.parse(path.read_text())
        except (SyntaxError, UnicodeDecodeError):
            continue
        for node in ast.walk(tree):
            if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
                st = _statements(node)
                if sum(st.values()) >= 4:
                    yield path, node.name, node.lineno, st


def report(min_frac: float = 0.0):
    ours = list(functions(sorted((ROOT / "tnco_amd").rglob("*.py")) + [ROOT / "bench.py"]))
    theirs = list(functions(sorted(REF.rglob("*.py"))))
    rows = []
    for path, name, line, st in ours:
        n = sum(st.values())
        best = (0.0, None)
        for rpath, rname, rline, rst in theirs:
            common = sum((st & rst).values())
            if common / n > best[0]:
                best = (common / n, (rpath, rname, rline, common))
        if best[1] and best[0] >= min_frac:
            rpath, rname, rline, common = best[1]
            rows.append((best[0], f"{path.relative_to(ROOT)}:{line} {name}", f"{rpath.relative_to(REF)}:{rline} {rname}", common, n))
    return sorted(rows, reverse=True)


if __name__ == "__main__":
    if not REF.exists():
        sys.exit("no /root/reference here")
    floor = float(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 0.2
    for frac, a, b, common, n in report(floor):
        print(f"{frac:5.2f}  {common:3d}/{n:<3d}  {a:60s} ~ {b}")
