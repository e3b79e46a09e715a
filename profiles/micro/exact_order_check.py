"""The grammar-graph search with the reference's order-dependent token creation (RS_EXACT_ORDER=1, decode_reg.hip) against the
reference's goldens of configs 1 and 3 (grammar-size graphs): which utterances' costs still differ, and by how much, with the
option off and on; and the search time either way.  GPU box: python profiles/micro/exact_order_check.py"""
import os, sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from tests import configs
from rhasspy_speech_amd import _lib

def deviations(words, costs, gold):
    gw, gg, ga = gold
    out = {}
    for u in range(len(gw)):
        if words[u] != gw[u]:
            out[u] = "WORDS"
        elif not (np.isclose(costs[u][0], gg[u], rtol=2e-4, atol=2e-3) and np.isclose(costs[u][1], ga[u], rtol=2e-4, atol=2e-3)):
            out[u] = round(float(costs[u][0] + costs[u][1] - gg[u] - ga[u]), 3)
    return out

tmp = Path(tempfile.mkdtemp())
jobs = [("c1_grammar", configs.build_grammar_model(tmp / "g"), configs.grammar_utterances(256, 0))]
names, pcms = configs.mixed_utterances()
for key, tag in (("de_DE-like", "c3_mixed_de"), ("fr_FR-like", "c3_mixed_fr")):
    m = configs.MIXED_MODELS[key]
    jobs.append((tag, configs.build_grammar_model(tmp / tag, m["model_seed"], m["graph_seed"]), [p for nm, p in zip(names, pcms) if nm == key]))
for tag, (md, gd), utts in jobs:
    model = _lib.Model(md, gd, _lib.default_opts())
    gold = configs.load_golden(tag)
    for mode in ("0", "1"):
        os.environ["RS_EXACT_ORDER"] = mode
        res = model.decode_batch(utts)
        t0 = time.perf_counter()
        for _ in range(5):
            res = model.decode_batch(utts)
        dt = (time.perf_counter() - t0) / 5
        dev = deviations([res.words(u) for u in range(len(utts))], [res.costs(u) for u in range(len(utts))], gold)
        print(f"{tag} exact_order={mode}: {len(utts)} utterances, cost deviations {dev}, search stage {res.timings()[4]:.3f} ms, call {dt * 1e3:.2f} ms", flush=True)
