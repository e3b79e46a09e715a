#!/usr/bin/env python3
"""The headline batch (256 x 3 s) on the full-size factorised TDNN (tests/configs.py: TDNNF_SPEC), N un-overlapped calls: what
`rocprofv3 --kernel-trace --stats` is pointed at for the per-kernel times of that model (profiles/r06/tdnnf_*).
usage: tdnnf_decode.py [calls] [frame_subsampling_factor] [utterances]"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rhasspy_speech_amd import _lib      # noqa: E402
from tests import configs                # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
fsf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
md, gd = configs.build_tdnnf_model(Path(tempfile.gettempdir()) / "rs_tdnnf_prof")
model = _lib.Model(md, gd, _lib.default_opts(frame_subsampling_factor=fsf))
pcms = configs.grammar_utterances(int(sys.argv[3]) if len(sys.argv) > 3 else 256)
model.decode_batch(pcms)
t0 = time.perf_counter()
st = np.zeros(8)
for _ in range(n):
    st += np.array(model.decode_batch(pcms).timings())
print(f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms per call; stages mfcc {st[1] / n:.3f} ivector {st[2] / n:.3f} nnet {st[3] / n:.3f} decode {st[4] / n:.3f}")
print([l for l in model.describe().splitlines() if l.startswith("layer_gemm")][0][:100])
