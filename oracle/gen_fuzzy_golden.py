#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden vectors for the fuzzy matcher (build container only).

The reference post-processes an n-best list with `get_fuzzy_text` (rhasspy_speech/transcribe_util.py:11-88): the
hypotheses become a fan of weighted linear paths, composed with `G.fuzzy.fst` (the grammar plus word-deletion self loops,
kaldi.py:343-408) and pushed through `fstshortestpath | fstrmepsilon | fsttopsort | fstproject | fstprint`; the printed
arcs give the text and the cost that is compared with `max_fuzzy_cost`.

This script builds synthetic language directories, creates `G.fuzzy.fst` with the reference's recipe (the text
transformation of kaldi.py:358-389 restated here, compiled by the REFERENCE's fstcompile | fstarcsort from oracle/_ref)
and runs the REFERENCE's own `get_fuzzy_text` (imported with stubs for the absent third-party packages) with a KaldiTools
whose PATH holds the OpenFst tools compiled from the reference's vendored sources.  Committed: the language directories
(words.txt, G.fuzzy.fst) under tests/golden/fuzzy/<lang>/ and (n-best bytes -> (text, cost) | None) pairs in
tests/golden/fuzzy/cases.json.  Nothing of the reference travels.
"""
import asyncio
import base64
import json
import os
import subprocess
import sys
import types
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
BIN = REPO / "oracle" / "_ref" / "bin"
OUT = REPO / "tests" / "golden" / "fuzzy"

for name in ["hassil", "hassil.expression", "hassil.intents", "hassil.util", "hassil.recognize", "unicode_rbnf"]:
    m = types.ModuleType(name)

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    m.__getattr__ = lambda k, _A=_Any: _A
    sys.modules[name] = m
sys.path.insert(0, str(REF))
from rhasspy_speech.transcribe_util import get_fuzzy_text  # noqa: E402
from rhasspy_speech.tools import KaldiTools  # noqa: E402


def meta(prefix: str, payload: str) -> str:
    return prefix + base64.b32encode(payload.encode()).decode()


def compile_fuzzy(lang_dir: Path, vocab, text_fst):
    """kaldi.py:358-389: copy the transitions, then self loops on every source state: eps:eps/0 and word:eps/1 for every
    non-meta vocabulary word; compiled by the reference's fstcompile | fstarcsort."""
    states, fuzzy = [], []
    for ln in text_fst:
        fuzzy.append(ln)
        st = ln.split(maxsplit=1)[0]
        if st not in states:
            states.append(st)
    for st in states:
        fuzzy.append(f"{st} {st} <eps> <eps> 0.0")
        for wd in vocab:
            if wd[0] in ("<", "_"):
                continue
            fuzzy.append(f"{st} {st} {wd} <eps> 1.0")
    txt = lang_dir / "G.fuzzy.fst.txt"
    txt.write_text("\n".join(fuzzy) + "\n")
    env = dict(os.environ, PATH=f"{BIN}:{os.environ['PATH']}")
    cmd = (f"fstcompile --isymbols={lang_dir}/words.txt --osymbols={lang_dir}/words.txt --keep_isymbols=true --keep_osymbols=true "
           f"{txt} | fstarcsort --sort_type=ilabel - {lang_dir}/G.fuzzy.fst")     # (the vendored fstcompile takes no "-" output)
    subprocess.run(["bash", "-c", cmd], check=True, env=env)
    txt.unlink()


def build_lang(lang_dir: Path, seed: int, n_words: int, n_sents: int, with_eps_arcs: bool, weighted: bool):
    """A grammar acceptor/transducer in OpenFst text form: one path per sentence from a shared start state, word arcs
    (word:word), optional eps:meta output arcs, eps:eps arcs with a cost, sentence weights."""
    rng = np.random.default_rng(seed)
    words = ["<eps>"] + [f"w{i}" for i in range(1, n_words + 1)]
    metas = [meta("__output:", json.dumps({"text": f"slot{j}"})) for j in range(3)] + [meta("__sentence_output:", "canned sentence text")]
    vocab = words + metas
    lang_dir.mkdir(parents=True, exist_ok=True)
    (lang_dir / "words.txt").write_text("".join(f"{w} {i}\n" for i, w in enumerate(vocab)))
    lines, finals = [], []
    nstate = 1
    sents = []
    for _ in range(n_sents):
        L = int(rng.integers(2, 7))
        sent = [words[int(i)] for i in rng.integers(1, n_words + 1, L)]
        sents.append(sent)
        cur = 0
        w0 = float(np.round(rng.uniform(0.0, 2.0), 3)) if weighted else None
        for k, wd in enumerate(sent):
            nxt = nstate
            nstate += 1
            cost = w0 if (k == 0 and w0 is not None) else None
            lines.append(f"{cur} {nxt} {wd} {wd}" + (f" {cost}" if cost is not None else ""))
            cur = nxt
            if with_eps_arcs and rng.random() < 0.3:          # an output label on an epsilon-input arc
                nxt = nstate
                nstate += 1
                lines.append(f"{cur} {nxt} <eps> {metas[int(rng.integers(0, len(metas)))]}")
                cur = nxt
            if with_eps_arcs and rng.random() < 0.2:          # a costed eps:eps arc (folded by fstrmepsilon)
                nxt = nstate
                nstate += 1
                lines.append(f"{cur} {nxt} <eps> <eps> {float(np.round(rng.uniform(0.1, 0.9), 2))}")
                cur = nxt
        finals.append(f"{cur}" + (f" {float(np.round(rng.uniform(0.0, 0.5), 2))}" if weighted and rng.random() < 0.5 else ""))
    text_fst = lines + finals
    compile_fuzzy(lang_dir, vocab, text_fst)
    return vocab, sents


def nbest_cases(rng, vocab, sents, n_cases):
    ids = {w: i for i, w in enumerate(vocab)}
    n_plain = len([w for w in vocab if w.startswith("w")])
    out = []
    for _ in range(n_cases):
        hyps = []
        for _h in range(int(rng.integers(1, 6))):
            sent = list(sents[int(rng.integers(0, len(sents)))])
            r = rng.random()
            if r < 0.35:                       # insert spurious words (deletions in the match)
                for _k in range(int(rng.integers(1, 3))):
                    sent.insert(int(rng.integers(0, len(sent) + 1)), f"w{int(rng.integers(1, n_plain + 1))}")
            elif r < 0.5 and len(sent) > 2:    # drop a word (no grammar path unless another sentence matches)
                del sent[int(rng.integers(0, len(sent)))]
            elif r < 0.6:                      # substitute
                sent[int(rng.integers(0, len(sent)))] = f"w{int(rng.integers(1, n_plain + 1))}"
            hyps.append(sent)
        text = "".join(f"utt-{k + 1} " + "".join(f"{ids[w]} " for w in h) + "\n" for k, h in enumerate(hyps))
        out.append(text.encode())
    out += [b"", b"utt-1 \n"]
    return out


def reference_tools():
    """KaldiTools over oracle/_ref.  The reference's Python targets OpenFst >= 1.8 (`fstproject --project_type=output`); the
    OpenFst vendored under /root/reference/kaldi/openfst is older and spells the same switch `--project_output=true`.  An
    argument adapter in front of the vendored binary bridges the two spellings -- the projection itself is the vendored
    tool's.  utils/int2sym.pl is the reference's own script, run where it lies."""
    import tempfile
    root = Path(tempfile.mkdtemp())
    shim = root / "bin"
    shim.mkdir()
    (shim / "fstproject").write_text(f"#!/bin/bash\nargs=()\nfor a in \"$@\"; do [ \"$a\" = --project_type=output ] && a=--project_output=true; args+=(\"$a\"); done\n"
                                     f"exec {BIN}/fstproject \"${{args[@]}}\"\n")
    (shim / "fstproject").chmod(0o755)
    (root / "utils").symlink_to(REF / "kaldi" / "egs" / "wsj" / "s5" / "utils")
    return KaldiTools(kaldi_dir=root, openfst_dir=REPO / "oracle" / "_ref", opengrm_dir=REPO / "oracle" / "_ref",
                      phonetisaurus_bin=Path("/nonexistent"))


def main():
    if OUT.exists():
        import shutil
        shutil.rmtree(OUT)
    OUT.mkdir(parents=True)
    tools = reference_tools()
    all_cases = []
    specs = [("small", 1, 12, 6, False, False), ("eps", 2, 20, 15, True, False), ("weighted", 3, 25, 20, True, True), ("big", 4, 60, 70, True, True)]
    for lang, seed, n_words, n_sents, with_eps, weighted in specs:
        lang_dir = OUT / lang
        vocab, sents = build_lang(lang_dir, seed, n_words, n_sents, with_eps, weighted)
        rng = np.random.default_rng(100 + seed)
        for nb in nbest_cases(rng, vocab, sents, 40 if lang != "big" else 25):
            res = asyncio.run(get_fuzzy_text(nb, lang_dir, tools))
            all_cases.append({"lang": lang, "nbest": nb.decode(), "result": None if res is None else [res[0], res[1]]})
    (OUT / "cases.json").write_text(json.dumps(all_cases, indent=0))
    n_hit = sum(1 for c in all_cases if c["result"] is not None)
    print(f"{len(all_cases)} cases ({n_hit} with a fuzzy match) -> {OUT}")


if __name__ == "__main__":
    main()
