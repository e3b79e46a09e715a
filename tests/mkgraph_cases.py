"""Graph-construction test cases shared by oracle/gen_mkgraph_golden.py and the tests.

A case = (synthetic acoustic model with a decision tree of some phonetic context, language-directory style, self-loop scale,
utterances decoded on the finished graph).  The language directory and the reference's outputs for each step of
utils/mkgraph.sh live in tests/golden/mkgraph/<case>/; the model directory (final.mdl, tree) is regenerated from the seed.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

from rhasspy_speech_amd import synth

GOLDEN = Path(__file__).resolve().parent / "golden" / "mkgraph"

CASES = {
    # monophone tree, grammar G (prefix tree of sentences), the self-loop scale rhasspy passes
    "mono_grammar": dict(spec=dict(context="mono"), lang="same_vocab", self_loop_scale=1.0, utts=[0, 1]),
    # left-biphone tree (the zamia chain models' shape), back-off bigram G with #0 arcs
    "bi_backoff": dict(spec=dict(context="biphone"), lang="backoff", self_loop_scale=1.0, utts=[2, 3]),
    # triphone tree (subsequential symbol, "#-1" pseudo epsilon), plain HMM topology, mkgraph.sh's default self-loop scale
    "tri_grammar_hmm": dict(spec=dict(context="triphone", chain_topology=False, seed=3), lang="same_vocab", self_loop_scale=0.1, utts=[4]),
    "tri_backoff": dict(spec=dict(context="triphone", seed=4), lang="backoff", self_loop_scale=1.0, utts=[5, 6]),
    # Kaldi's classic shape: three emitting states per phone (three pdf-classes under every tree leaf, an HMM of four states per
    # phone-in-context in H), triphone tree, the script's default scales
    "tri_hmm3_backoff": dict(spec=dict(context="triphone", chain_topology=False, hmm_states=3, seed=6), lang="backoff", self_loop_scale=0.1, utts=[8, 9]),
    # zamia-size model (1000 phones, left-biphone tree: 2668 pdfs), 660-word back-off bigram: only the language directory and
    # the reference's decodes are kept (its HCLG.fst is several MB)
    "zam_bi_arpa": dict(big=True, spec=dict(context="biphone"), lang="backoff", lang_conf=dict(keep_every=1, extra_sentences=1500, backoff=True),
                        extra_words=600, self_loop_scale=1.0, utts=[0, 1, 7]),
}


def _random_case(i: int) -> dict:
    """Seed-drawn tree / topology / lexicon / LM shapes ("light" cases: only the language directory, the reference's HCLG.fst and
    its decodes are kept, not every intermediate transducer)."""
    rng = np.random.default_rng(4200 + i)
    chain = bool(rng.integers(0, 2))
    spec = dict(context=["mono", "biphone", "triphone", "2,0", "3,2", "3,0", "4,1", "4,2"][int(rng.integers(0, 8))], chain_topology=chain,
                hmm_states=1 if chain else int(rng.integers(1, 4)), num_phones=int(rng.integers(12, 32)), seed=int(rng.integers(1, 500)))
    return dict(light=True, spec=spec, lang="backoff" if rng.random() < 0.5 else "same_vocab",
                lang_conf=dict(keep_every=int(rng.integers(1, 4)), extra_sentences=int(rng.integers(0, 25)), backoff=bool(rng.integers(0, 2))),
                lang_seed=int(rng.integers(1, 1000)), self_loop_scale=float([1.0, 0.1][int(rng.integers(0, 2))]),
                utts=[int(x) for x in rng.integers(10, 10000, 2)])


CASES.update({f"rnd{i:02d}": _random_case(i) for i in range(12)})
SMALL = [n for n, c in CASES.items() if not c.get("big")]                      # whole chain on the CPU
STEPWISE = [n for n, c in CASES.items() if not c.get("big") and not c.get("light")]      # every intermediate transducer committed


def case_spec(case: dict) -> synth.ModelSpec:
    return synth.ModelSpec(**case["spec"]) if case.get("big") else synth.tiny_spec(**case["spec"])


def case_lexicon(case: dict, spec: synth.ModelSpec) -> synth.Lexicon:
    """The parity cases' lexicon, plus what makes disambiguation symbols necessary: a pair of homophones and a word whose
    pronunciation is a prefix of another's."""
    sents = [s.split() for s in synth.DEFAULT_SENTENCES]
    lex = synth.make_lexicon(sents, spec, np.random.default_rng(11), extra_words=case.get("extra_words", 0))
    ids = {w: i for i, w in enumerate(lex.words)}
    lex.prons[ids["night"]] = list(lex.prons[ids["light"]])
    lex.prons[ids["time"]] = list(lex.prons[ids["timer"]][:2])
    lex.prons[ids["timer"]] = lex.prons[ids["time"]] + [lex.prons[ids["timer"]][-1]]
    return lex


def build_model_dir(case: dict, root: Path) -> Path:
    """-> <root>/model/model/model holding final.mdl and tree (plus the online conf beside it)."""
    synth.write_model_dir(root / "model", case_spec(case))
    return root / "model"
