"""Rescoring path end to end on the GPU (`-m gpu`): decode with the OLD graph on the HIP path (results keep their determinised
lattices), rs_rescore_result against the NEW language directory, compared with what the reference's tool chain -- its own
decoder included -- produces for the same wav (tests/golden/rescore; oracle/gen_rescore_golden.py).  The library's lattice is
equivalent to the reference's, not identical (state numbering, weight placement), so this also checks that nothing in the
chain depends on those."""
import asyncio
import json

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

GOLDEN = cases.GOLDEN / "rescore"
RUNS = json.loads((GOLDEN / "cases.json").read_text())


@pytest.mark.parametrize("run", RUNS, ids=[f"{r['case']}-{r['lang']}" for r in RUNS])
def test_rescore_end_to_end(case_cache, run):
    from rhasspy_speech_amd import _lib
    model_dir, graph_dir, _, pcm = case_cache(run["case"])
    o = dict(cases.CASES[run["case"]].get("opts", {}))
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts(emit_lattice=1, **o))
    rs = _lib.Rescorer(model, GOLDEN / run["dir"])
    res = model.decode_batch([pcm, pcm[: len(pcm) // 2]])
    text, g, a = rs.rescore(res, 0, nbest=5, acoustic_scale=1.0)
    assert text.split() == run["nbest_text"].encode().split(), (text, run["nbest_text"])
    np.testing.assert_allclose(g, run["graph_cost"], rtol=2e-4, atol=5e-3)
    np.testing.assert_allclose(a, run["acoustic_cost"], rtol=2e-4, atol=5e-3)
    # the library's own lattice bytes through the host entry point give the same answer
    text2, _, _ = rs.rescore_lattice(res.lattice(0), nbest=5)
    assert text2 == text
    # a stream of the same audio: the streaming transcriber's rescoring path (transcribe_stream.py:131-274) against the reference chain
    # run on the lattice of ITS streaming decoder (an iVector per chunk: other costs, sometimes another ranking, than the wav path's)
    st = _lib.Stream(model)
    for i in range(0, len(pcm), 4000):
        st.accept(pcm[i:i + 4000])
        st.advance()
    sres = st.finish()
    st.close()
    stext, sg, sa = rs.rescore(sres, 0, nbest=5, acoustic_scale=1.0)
    assert stext.split() == run["stream_nbest_text"].encode().split(), (stext, run["stream_nbest_text"])
    np.testing.assert_allclose(sg, run["stream_graph_cost"], rtol=2e-4, atol=5e-3)
    np.testing.assert_allclose(sa, run["stream_acoustic_cost"], rtol=2e-4, atol=5e-3)


def test_stream_transcriber_api_rescore(case_cache):
    """KaldiNnet3StreamTranscriber.async_transcribe_rescore end to end against the reference's streaming chain: words from the NEW
    words.txt; the audio arrives in the chunks an asyncio producer yields and the event loop stays free while the device works
    (accept / advance run in the executor, where the reference awaits the decoder's stdin drain: transcribe_stream.py:73-76)."""
    from rhasspy_speech_amd.transcribe_stream import KaldiNnet3StreamTranscriber
    from rhasspy_speech_amd.meta import read_words_txt
    run = next(r for r in RUNS if r["case"] == "zam_real_cold" and r["lang"] == "backoff")
    model_dir, graph_dir, _, pcm = case_cache(run["case"])
    tr = KaldiNnet3StreamTranscriber(model_dir, graph_dir)
    new_lang = GOLDEN / run["dir"]
    ticks = []

    async def audio():
        raw = pcm.astype("<i2").tobytes()
        for i in range(0, len(raw), 2048):
            yield raw[i:i + 2048]
            await asyncio.sleep(0)
        yield None

    async def main():
        async def heartbeat():
            while True:
                ticks.append(1)
                await asyncio.sleep(0.001)
        hb = asyncio.ensure_future(heartbeat())
        try:
            return await tr.async_transcribe_rescore(audio(), graph_dir, new_lang, nbest=5)
        finally:
            hb.cancel()
    got = asyncio.run(main())
    words = read_words_txt(new_lang / "words.txt")
    want = [" ".join(words[int(i)] for i in line.split()[1:]) for line in run["stream_nbest_text"].splitlines() if line.split()[1:]]
    assert got == want and len(want) == 2
    assert len(ticks) > 3            # (the loop ran other tasks while the stream was being decoded)


def test_transcriber_api_rescore(case_cache):
    """KaldiNnet3WavTranscriber.async_transcribe_rescore: same signature as the reference's (transcribe_wav.py:107-115), words
    come from the NEW words.txt."""
    import inspect
    from rhasspy_speech_amd.transcribe_wav import KaldiNnet3WavTranscriber
    from rhasspy_speech_amd.meta import read_words_txt
    sig = inspect.signature(KaldiNnet3WavTranscriber.async_transcribe_rescore)
    assert list(sig.parameters)[1:] == ["wav_path", "old_lang_dir", "new_lang_dir", "nbest", "max_fuzzy_cost", "require_fuzzy"]
    run = next(r for r in RUNS if r["case"] == "tiny_u0" and r["lang"] == "backoff")
    model_dir, graph_dir, wav, _ = case_cache("tiny_u0")
    tr = KaldiNnet3WavTranscriber(model_dir, graph_dir)
    new_lang = GOLDEN / run["dir"]
    got = asyncio.run(tr.async_transcribe_rescore(wav, graph_dir, new_lang, nbest=5))
    words = read_words_txt(new_lang / "words.txt")
    want = [" ".join(words[int(i)] for i in line.split()[1:]) for line in run["nbest_text"].splitlines() if line.split()[1:]]
    assert got == want
