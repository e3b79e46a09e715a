#!/usr/bin/env python3
"""Per-utterance work of the ARPA workload (BASELINE configs[2]) as the token-list search counts it: tokens over the whole
utterance, the largest frame's token count and the largest frame's candidate-record count (rs_result_counters [3], [4]).
Sizes the live-state table of the large-graph search.  Usage (GPU box): python profiles/micro/arpa_frame_sizes.py"""
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: F401,E402  (its HIP runtime first)
from rhasspy_speech_amd import _lib  # noqa: E402
from tests import configs  # noqa: E402

cache = Path(tempfile.gettempdir()) / "rs_bench_arpa_rank0"
model_dir, graph_dir = configs.build_arpa_model(cache)
pcms = configs.arpa_utterances(256)
model = _lib.Model(model_dir, graph_dir, _lib.default_opts())
print(model.describe().splitlines()[0:6])
res = model.decode_batch(pcms)
rows = np.array([res.counters(u) for u in range(len(pcms))], dtype=np.int64)
tok, big = rows[:, 3], rows[:, 4]
max_tok, max_cand = big & 0xFFFFFFFF, big >> 32
for name, v in (("tokens per utterance", tok), ("largest frame: tokens", max_tok), ("largest frame: candidate records", max_cand),
                ("arcs examined per utterance", rows[:, 1]), ("insertions per utterance", rows[:, 2])):
    q = np.percentile(v, [0, 25, 50, 75, 90, 99, 100]).astype(np.int64)
    print(f"{name:36s} min {q[0]} p25 {q[1]} median {q[2]} p75 {q[3]} p90 {q[4]} p99 {q[5]} max {q[6]}")
print("frames limited by max-active (sum over utterances):", int(rows[:, 5].sum()), "of", 256 * 298)
