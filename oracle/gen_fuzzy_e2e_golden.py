#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- end-to-end goldens for the fuzzy-matched transcription (build container only).

Runs the REFERENCE's own `KaldiNnet3WavTranscriber.async_transcribe` (rhasspy_speech/transcribe_wav.py:35-105, imported
from /root/reference with stubs for the two absent third-party packages) on parity-test cases: its subprocess pipeline
is the reference's `online2-wav-nnet3-latgen-faster | lattice-to-nbest | nbest-to-linear`, `utils/int2sym.pl` and the
seven OpenFst tools of `get_fuzzy_text`, all compiled from the reference's sources under oracle/_ref.  The language
directory of each case holds the graph's words.txt (plus meta labels) and a G.fuzzy.fst made with the reference's recipe
(kaldi.py:343-408, restated in gen_fuzzy_golden.compile_fuzzy) from a VARIANT of the case's grammar: every third sentence
unchanged, every third without its first word (the recognised word has to be dropped at cost 1), every third absent.

Committed: tests/golden/fuzzy_e2e/<case>/{words.txt,G.fuzzy.fst} and tests/golden/fuzzy_e2e/cases.json with, per case and
(nbest, max_fuzzy_cost, require_fuzzy) setting, the List[str] the reference returned, plus get_fuzzy_text's own
(text, cost) on the reference's offline and streaming n-best lists
(tests/golden/<case>.npz).  Nothing of the reference travels.
"""
import asyncio
import json
import shutil
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "oracle"))

import gen_fuzzy_golden as gf  # noqa: E402  (stubs + reference imports + compile_fuzzy + reference_tools)
from rhasspy_speech.transcribe_wav import KaldiNnet3WavTranscriber  # noqa: E402  (the reference's, /root/reference)
from rhasspy_speech_amd import synth  # noqa: E402
from tests import cases  # noqa: E402

OUT = REPO / "tests" / "golden" / "fuzzy_e2e"
CASES = ["tiny_u0", "tiny_real_hot", "tiny_real_time", "tinyf_u5", "zam_u0", "zam_u1", "zam_real_cold"]
SETTINGS = [(5, 0.0, False), (5, 1.05, False), (5, 100.0, False), (5, 0.5, True), (1, 2.5, True)]


def build_lang(lang_dir: Path, graph_dir: Path):
    words = [ln.split()[0] for ln in (graph_dir / "words.txt").read_text().splitlines() if ln.strip()]
    metas = [gf.meta("__output:", json.dumps({"text": f"slot{j}"})) for j in range(2)] + [gf.meta("__sentence_output:", "canned sentence text")]
    vocab = words + metas
    lang_dir.mkdir(parents=True, exist_ok=True)
    (lang_dir / "words.txt").write_text("".join(f"{w} {i}\n" for i, w in enumerate(vocab)))
    lines, finals, nstate = [], [], 1
    known = set(words)
    for i, sent in enumerate(s.split() for s in synth.DEFAULT_SENTENCES):
        if i % 3 == 2:
            continue
        if i % 3 == 1:
            sent = sent[1:]
        sent = [w for w in sent if w in known]
        if not sent:
            continue
        cur = 0
        for wd in sent:
            lines.append(f"{cur} {nstate} {wd} {wd}")
            cur, nstate = nstate, nstate + 1
        if i % 2 == 0:
            lines.append(f"{cur} {nstate} <eps> {metas[i % len(metas)]}")
            cur, nstate = nstate, nstate + 1
        finals.append(f"{cur}")
    gf.compile_fuzzy(lang_dir, vocab, lines + finals)


def main():
    if OUT.exists():
        shutil.rmtree(OUT)
    OUT.mkdir(parents=True)
    tools = gf.reference_tools()
    out = []
    for name in CASES:
        case = cases.CASES[name]
        td = Path(tempfile.mkdtemp())
        model_dir, graph_dir, wav, _pcm = cases.build_case_files(case, td)
        lang_dir = OUT / name
        build_lang(lang_dir, graph_dir)
        opts = case.get("opts", {})
        t = KaldiNnet3WavTranscriber(model_dir, graph_dir, tools, **{k: v for k, v in opts.items() if k in ("max_active", "lattice_beam", "beam")})
        import numpy as np
        g = np.load(cases.GOLDEN / f"{name}.npz")
        nbest_bytes = bytes(g["offline_nbest_text"])
        fz = asyncio.run(gf.get_fuzzy_text(nbest_bytes, lang_dir, tools))
        fzs = asyncio.run(gf.get_fuzzy_text(bytes(g["stream_nbest_text"]), lang_dir, tools))
        rec = {"case": name, "fuzzy_of_golden_nbest": None if fz is None else [fz[0], fz[1]],
               "fuzzy_of_golden_stream_nbest": None if fzs is None else [fzs[0], fzs[1]], "runs": []}
        for nbest, max_cost, require in SETTINGS:
            texts = asyncio.run(t.async_transcribe(wav, lang_dir, nbest=nbest, max_fuzzy_cost=max_cost, require_fuzzy=require))
            rec["runs"].append({"nbest": nbest, "max_fuzzy_cost": max_cost, "require_fuzzy": require, "texts": texts})
        out.append(rec)
        print(name, rec["fuzzy_of_golden_nbest"], [r["texts"] for r in rec["runs"]])
        shutil.rmtree(td)
    (OUT / "cases.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
