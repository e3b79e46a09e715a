#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- randomised comparison of the rescoring path with the REFERENCE's tool chain (build container only).

For each seed: one of the parity cases (its model, its old graph, its wav), a random NEW language directory (which sentences
survive, made-up sentences, grammar or back-off G: gen_rescore_golden.write_lang), the reference decoder's lattice and the
reference chain `lattice-scale | lattice-to-phone-lattice | lattice-compose Ldet | lattice-determinize | lattice-compose --phi-label
G | lattice-add-trans-probs | lattice-to-nbest | nbest-to-linear` on it (gen_rescore_golden.reference_rescore), against
rs_rescore_lattice on the same lattice: n-best word ids equal, costs within 2e-3.  usage: python oracle/fuzz_rescore.py [first [count]]
"""
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "oracle"))
from rhasspy_speech_amd import _lib, synth  # noqa: E402
from tests import cases  # noqa: E402
import gen_rescore_golden as rg  # noqa: E402

CASES = ["tiny_u0", "tiny_real_hot", "tiny_real_time", "tiny_arpa_u7", "tinyf_u5", "tiny_hmm_u6", "zam_u0", "zam_u1", "zam_real_cold"]
_cache = {}


def case_files(name):
    if name not in _cache:
        td = Path(tempfile.mkdtemp())
        case = cases.CASES[name]
        model_dir, graph_dir, wav, _ = cases.build_case_files(case, td)
        spec = cases.case_spec(case)
        g = case["graph"].split(":")
        if g[0] == "grammar":
            lex = synth.make_lexicon([s.split() for s in synth.DEFAULT_SENTENCES], spec, np.random.default_rng(11))
        else:
            lex = synth.make_lexicon([s.split() for s in synth.DEFAULT_SENTENCES], spec, np.random.default_rng(13), extra_words=int(g[1]))
        _cache[name] = (case, model_dir, graph_dir, wav, spec, lex, _lib.Model(model_dir, graph_dir, _lib.default_opts()))
    return _cache[name]


def one(seed):
    rng = np.random.default_rng(10_000 + seed)
    name = CASES[int(rng.integers(0, len(CASES)))]
    case, model_dir, graph_dir, wav, spec, lex, model = case_files(name)
    conf = dict(keep_every=int(rng.integers(1, 4)), extra_sentences=int(rng.integers(0, 25)), backoff=bool(rng.integers(0, 2)))
    td = Path(tempfile.mkdtemp())
    try:
        lang = td / "lang"
        rg.write_lang(lang, lex, spec, conf, rng)
        text, lm, ac, lat = rg.reference_rescore(model_dir, graph_dir, wav, lang, td, case)
        rs = _lib.Rescorer(model, lang)
        mine, g, a = rs.rescore_lattice(lat, nbest=rg.NBEST, acoustic_scale=1.0)
        if mine.split() != text.split():
            raise SystemExit(f"seed {seed} ({name}, {conf}): n-best differs\n reference: {text!r}\n mine:      {mine!r}")
        if len(g) != len(lm) or (len(g) and (np.abs(np.array(g) - np.array(lm)).max() > 2e-3 + 1e-4 * np.abs(lm).max()
                                             or np.abs(np.array(a) - np.array(ac)).max() > 2e-3 + 1e-4 * np.abs(ac).max())):
            raise SystemExit(f"seed {seed} ({name}, {conf}): costs differ\n reference: {lm} {ac}\n mine:      {g} {a}")
        shown = text.decode().strip().replace("\n", " | ") or "(nothing survives)"
        return f"seed {seed}: {name} {conf}: {len(lm)} hypotheses equal: {shown[:70]}"
    finally:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    for s in range(first, first + count):
        print(one(s), flush=True)
