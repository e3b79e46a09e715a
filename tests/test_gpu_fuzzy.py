"""the fuzzy-matched transcription end to end -- wav -> HIP decode -> n-best -> fuzzy match
against <lang_dir>/G.fuzzy.fst -> List[str] -- against what the REFERENCE's own KaldiNnet3WavTranscriber.async_transcribe
returned for the same wav, model, graph, language directory and (nbest, max_fuzzy_cost, require_fuzzy) setting, with its
subprocess pipeline running the reference's Kaldi / OpenFst tools (oracle/gen_fuzzy_e2e_golden.py; tests/golden/fuzzy_e2e).
"""
import asyncio
import json

import pytest

from rhasspy_speech_amd import _lib
from rhasspy_speech_amd.meta import decode_meta, read_words_txt
from rhasspy_speech_amd.transcribe_stream import KaldiNnet3StreamTranscriber
from rhasspy_speech_amd.transcribe_wav import KaldiNnet3WavTranscriber

from . import cases

pytestmark = pytest.mark.gpu

E2E = cases.GOLDEN / "fuzzy_e2e"
RECORDS = json.loads((E2E / "cases.json").read_text())


@pytest.mark.parametrize("rec", RECORDS, ids=[r["case"] for r in RECORDS])
def test_fuzzy_transcription_equals_the_reference_transcriber(rec, tmp_path):
    case = cases.CASES[rec["case"]]
    model_dir, graph_dir, wav, pcm = cases.build_case_files(case, tmp_path)
    lang_dir = E2E / rec["case"]
    opts = {k: v for k, v in case.get("opts", {}).items() if k in ("max_active", "lattice_beam", "beam")}
    t = KaldiNnet3WavTranscriber(model_dir, graph_dir, **opts)
    for run in rec["runs"]:
        got = asyncio.run(t.async_transcribe(wav, lang_dir, nbest=run["nbest"], max_fuzzy_cost=run["max_fuzzy_cost"],
                                             require_fuzzy=run["require_fuzzy"]))
        assert got == run["texts"], (rec["case"], run, got)
    # the match folded into the result (rs_result_fuzzy) = the reference's get_fuzzy_text on the reference's own n-best
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts(**opts))
    res = model.decode_batch([pcm], nbest=cases.NBEST)
    matcher = _lib.FuzzyMatcher(lang_dir / "G.fuzzy.fst")
    hit = matcher.match_result(res, 0)
    want = rec["fuzzy_of_golden_nbest"]
    if want is None:
        assert hit is None
    else:
        words = read_words_txt(lang_dir / "words.txt")
        assert " ".join(words[i] for i in hit[0]) == want[0]
        assert hit[1] == want[1]              # an exact double: it is what is compared with max_fuzzy_cost
    assert hit == matcher.match(res.text(0, "utt"))
    matcher.close()
    model.close()


@pytest.mark.parametrize("rec", RECORDS, ids=[r["case"] for r in RECORDS])
def test_fuzzy_transcription_of_a_stream(rec, tmp_path):
    """transcribe_stream.py:101-129: the same tail behind the streaming decode.  Expected = the reference's get_fuzzy_text on
    the reference's streaming n-best (tests/golden/<case>.npz); where that is None (zam_u1, whose streaming transcript differs
    from the offline one) the n-best texts themselves come back."""
    import numpy as np
    case = cases.CASES[rec["case"]]
    model_dir, graph_dir, wav, pcm = cases.build_case_files(case, tmp_path)
    lang_dir = E2E / rec["case"]
    opts = {k: v for k, v in case.get("opts", {}).items() if k in ("max_active", "lattice_beam", "beam")}
    t = KaldiNnet3StreamTranscriber(model_dir, graph_dir, **opts)

    async def chunks():
        raw = pcm.tobytes()
        for i in range(0, len(raw), 2048):
            yield raw[i:i + 2048]

    got = asyncio.run(t.async_transcribe(chunks(), lang_dir, nbest=5, max_fuzzy_cost=100.0))
    want = rec["fuzzy_of_golden_stream_nbest"]
    if want is not None:
        assert got == [decode_meta(want[0])]
    else:
        words = read_words_txt(graph_dir / "words.txt")
        g = np.load(cases.GOLDEN / f"{rec['case']}.npz")
        lines = [ln.split() for ln in bytes(g["stream_nbest_text"]).decode().splitlines() if ln.startswith("utt-")]
        assert got == [decode_meta(" ".join(words[int(i)] for i in ln[1:])) for ln in lines if len(ln) > 1]
