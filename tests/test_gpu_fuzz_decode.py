"""Randomised decode cases against the REFERENCE's binaries: 48 (model shape, graph kind, utterance length, decoder options)
combinations drawn from a seed (tests/fuzz_cases.py), offline and streaming, 5-best lists and their costs
(tests/golden/fuzz_decode.json, produced by oracle/gen_fuzz_decode_golden.py with oracle/_ref)."""
import json

import numpy as np
import pytest

from rhasspy_speech_amd import _lib

from . import cases, fuzz_cases

pytestmark = pytest.mark.gpu

GOLD = json.loads((cases.GOLDEN / "fuzz_decode.json").read_text())


def _opts(case):
    o = {k: v for k, v in case.get("opts", {}).items() if k in ("max_active", "min_active", "beam", "lattice_beam")}
    return _lib.default_opts(**o)


def _check(res, gold, what):
    assert gold["status"] == 0
    assert res.text(0, "utt").split() == gold["nbest_text"].encode().split(), what
    n = len(gold["graph_cost"])
    assert res.num_hyps(0) == n
    gc = np.array([res.costs(0, k)[0] for k in range(n)])
    ac = np.array([res.costs(0, k)[1] for k in range(n)])
    np.testing.assert_allclose(gc, gold["graph_cost"], rtol=2e-4, atol=2e-3, err_msg=what)
    np.testing.assert_allclose(ac, gold["acoustic_cost"], rtol=2e-4, atol=2e-3, err_msg=what)


@pytest.mark.parametrize("i", range(len(fuzz_cases.CASES)))
def test_random_case_equals_the_reference(i, tmp_path):
    case, gold = fuzz_cases.CASES[i], GOLD[i]
    model_dir, graph_dir, _wav, pcm = cases.build_case_files(case, tmp_path)
    if gold["offline"]["status"] != 0:
        # the reference's binaries aborted (decoder options its Check() refuses: min-active > max-active): so does the library
        with pytest.raises(_lib.RsError, match="min_active <= max_active"):
            _lib.Model(model_dir, graph_dir, _opts(case)).decode_batch([pcm])
        return
    model = _lib.Model(model_dir, graph_dir, _opts(case))
    _check(model.decode_batch([pcm], nbest=cases.NBEST), gold["offline"], f"case {i} offline {case}")
    if gold["stream"]["status"] == 0:
        st = _lib.Stream(model)
        raw = pcm.tobytes()
        for k in range(0, len(raw), 4096):           # 2048-sample deliveries, device work as they arrive
            st.accept(raw[k:k + 4096])
            st.advance()
        _check(st.finish(cases.NBEST, 1.0), gold["stream"], f"case {i} stream {case}")
    model.close()
