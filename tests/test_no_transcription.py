"""Host code is written from the reference's contracts, not transcribed from its source: where the reference checkout
exists (this container; not the GPU box), no function of tnco_amd/ or bench.py with eight or more statements shares half
of them -- AST-normalised, identifiers included -- with any function of the reference (tools/similarity.py; VERDICT r05
found `merge_contraction_paths` at 19 of 20)."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


def test_no_function_is_a_transcription_of_the_reference():
    import similarity
    if not similarity.REF.exists():
        pytest.skip("no reference checkout on this machine")
    rows = similarity.report(0.5)
    close = [(round(frac, 2), ours, theirs) for frac, ours, theirs, _common, n in rows if n >= 8]
    assert not close, close
