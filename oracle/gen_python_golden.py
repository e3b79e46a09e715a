#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden vectors for the Python layer of the boundary (build container only).

Imports the REFERENCE's own rhasspy_speech.transcribe_wav.KaldiNnet3WavTranscriber (with stub modules for the two
third-party packages missing here, hassil and unicode_rbnf) and drives its async_transcribe with a fake KaldiTools
whose pipeline returns canned `nbest-to-linear` bytes and runs the reference's real utils/int2sym.pl, so that the
"utt-" filtering, id->word mapping and decode_meta behaviour are captured as (inputs -> List[str]) pairs in
tests/golden/python_api.json.  Nothing from the reference travels: only these data pairs are committed.
"""
import asyncio
import base64
import json
import subprocess
import sys
import tempfile
import types
from pathlib import Path

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "python_api.json"

for name in ["hassil", "hassil.expression", "hassil.intents", "hassil.util", "hassil.recognize", "unicode_rbnf"]:
    m = types.ModuleType(name)

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    m.__getattr__ = lambda k, _A=_Any: _A     # any attribute is a permissive class
    sys.modules[name] = m
sys.path.insert(0, str(REF))
from rhasspy_speech.transcribe_wav import KaldiNnet3WavTranscriber  # noqa: E402
from rhasspy_speech.hassil_fst import encode_meta  # noqa: E402


class FakeTools:
    egs_utils_dir = REF / "kaldi" / "egs" / "wsj" / "s5" / "utils"

    def __init__(self, nbest_bytes):
        self.nbest = nbest_bytes

    async def async_run_pipeline(self, *commands, input=None, **kw):
        if commands[0][0] == "online2-wav-nnet3-latgen-faster":
            return self.nbest
        assert commands[0][0].endswith("int2sym.pl")
        return subprocess.run(["perl", *commands[0]], input=input, stdout=subprocess.PIPE, check=True).stdout


def b32json(obj):
    return json.dumps(obj)


WORDS = ["<eps>", "turn", "on", "the", "light", "__output:" + base64.b32encode(b32json({"text": "living room", "list": "area"}).encode()).decode(),
         "__output:" + base64.b32encode(b32json({"text": "ON"}).encode()).decode(),
         "__sentence_output:" + base64.b32encode("turn {area} lights".encode()).decode(), "off"]
CASES = [
    b"utt-1 1 2 3 4 \n",
    b"utt-1 1 2 3 4 \nutt-2 1 8 3 4 \nutt-3 \n",
    b"utt-1 \n",
    b"",
    b"utt-1 1 2 5 4 \n",
    b"utt-1 1 6 3 5 4 \n",
    b"utt-1 1 2 5 4 7 \nutt-2 1 2 3 4 \n",
    b"other-1 1 2 \nutt-1 3 4 \n",
]


def main():
    out = {"words": WORDS, "cases": []}
    with tempfile.TemporaryDirectory() as td:
        g = Path(td) / "graph"
        g.mkdir()
        (g / "words.txt").write_text("".join(f"{w} {i}\n" for i, w in enumerate(WORDS)))
        lang = Path(td) / "lang"
        lang.mkdir()
        for nb in CASES:
            tr = KaldiNnet3WavTranscriber(Path(td) / "model", g, FakeTools(nb))
            texts = asyncio.run(tr.async_transcribe(Path(td) / "x.wav", lang, nbest=3))
            out["cases"].append({"nbest_stdout": nb.decode(), "texts": texts})
    out["encode_meta"] = {"input": "hello world", "output": encode_meta("hello world")}
    OUT.write_text(json.dumps(out, indent=1))
    print(json.dumps(out["cases"], indent=1))


if __name__ == "__main__":
    main()
