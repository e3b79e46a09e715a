"""Rescoring path on the host: rs_rescore_lattice on the REFERENCE decoder's own lattices against
what the reference's tool chain makes of them (tests/golden/rescore, oracle/gen_rescore_golden.py):
    lattice-scale --lm-scale=0.0 | lattice-to-phone-lattice | lattice-compose - Ldet.fst | lattice-determinize |
    lattice-compose --phi-label=#0 - G.fst | lattice-add-trans-probs | lattice-to-nbest --n=5 | nbest-to-linear
n-best word ids equal, graph / acoustic costs within 1e-3 (float sums along different but equivalent machines).  No GPU: model
parsing and the lattice algebra run on the host."""
import json

import numpy as np
import pytest

from tests import cases

GOLDEN = cases.GOLDEN / "rescore"
RUNS = json.loads((GOLDEN / "cases.json").read_text())


@pytest.mark.parametrize("run", RUNS, ids=[f"{r['case']}-{r['lang']}" for r in RUNS])
def test_rescore_reference_lattice(case_cache, run):
    from rhasspy_speech_amd import _lib
    model_dir, graph_dir, _, _ = case_cache(run["case"])
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts())        # host-side parse only
    rs = _lib.Rescorer(model, GOLDEN / run["dir"])
    entry = (GOLDEN / run["dir"] / "decoder.lat").read_bytes()
    text, g, a = rs.rescore_lattice(entry, nbest=5, acoustic_scale=1.0)
    assert text.split() == run["nbest_text"].encode().split(), (text, run["nbest_text"])
    np.testing.assert_allclose(g, run["graph_cost"], rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(a, run["acoustic_cost"], rtol=1e-4, atol=2e-3)
    # n = 1 is the head of the list
    one, g1, a1 = rs.rescore_lattice(entry, nbest=1)
    assert one.split() == run["nbest_text"].encode().split(b"\n")[0].split()


def test_missing_phi_fails_like_the_reference(case_cache, tmp_path):
    import shutil
    from rhasspy_speech_amd import _lib
    model_dir, graph_dir, _, _ = case_cache("tiny_u0")
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts())
    lang = tmp_path / "lang"
    shutil.copytree(GOLDEN / RUNS[0]["dir"], lang)
    (lang / "words.txt").write_text("".join(l for l in (lang / "words.txt").read_text().splitlines(True) if not l.startswith("#0 ")))
    with pytest.raises(_lib.RsError) as ei:
        _lib.Rescorer(model, lang)
    assert "No value for disambiguation state (#0)" in str(ei.value)
