"""SURVEY.md section 8(f2) end to end: a graph built by the library's mkgraph chain, decoded on the GPU, against the REFERENCE
decoder on the graph the REFERENCE's utils/mkgraph.sh built from the same language directory and model
(tests/golden/mkgraph/<case>/decode.json, oracle/gen_mkgraph_golden.py)."""
import json

import numpy as np
import pytest

from rhasspy_speech_amd import _lib, synth

from . import mkgraph_cases as mc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(mc.CASES))
def test_decoding_on_the_built_graph_equals_the_reference_on_its_graph(name, tmp_path):
    case = mc.CASES[name]
    g = mc.GOLDEN / name
    model_dir = mc.build_model_dir(case, tmp_path)
    graph_dir = tmp_path / "graph"
    _lib.mkgraph(g / "lang", model_dir / "model" / "model", graph_dir, self_loop_scale=case["self_loop_scale"])
    gold = json.loads((g / "decode.json").read_text())["decodes"]
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts())
    pcm = [synth.synth_utterance(d["utt"], 48000) for d in gold]
    res = model.decode_batch(pcm, nbest=5)
    for u, d in enumerate(gold):
        assert res.text(u, "utt").decode() == d["nbest_text"], (name, d["utt"])
        assert res.num_hyps(u) == len(d["graph_cost"])
        for k, (gc, ac) in enumerate(zip(d["graph_cost"], d["acoustic_cost"])):
            mg, ma = res.costs(u, k)
            # weights of both graphs sit on a 1/1024 grid that each chain reaches by its own float path
            assert abs(mg - gc) < 0.02 and abs(ma - ac) < 2e-3 * max(1.0, abs(ac)), (name, d["utt"], k, mg, gc, ma, ac)
    if case.get("big"):          # (the reference's multi-megabyte HCLG.fst of this case is not kept)
        model.close()
        return
    # the same decode on the reference's own HCLG.fst gives the same records (the two graphs are interchangeable)
    ref_graph = tmp_path / "ref_graph"
    ref_graph.mkdir()
    (ref_graph / "HCLG.fst").write_bytes((g / "ref" / "HCLG.fst").read_bytes())
    (ref_graph / "words.txt").write_bytes((g / "lang" / "words.txt").read_bytes())
    model2 = _lib.Model(model_dir, ref_graph, _lib.default_opts())
    res2 = model2.decode_batch(pcm, nbest=5)
    for u in range(len(gold)):
        assert res2.text(u, "utt") == res.text(u, "utt")
        for k in range(res.num_hyps(u)):
            (g1, a1), (g2, a2) = res.costs(u, k), res2.costs(u, k)
            assert abs(g1 - g2) < 0.02 and abs(a1 - a2) < 1e-2
    model.close()
    model2.close()
