#!/usr/bin/env python3
"""What the split-fp16 layer GEMMs do to small operand rows when the precision flag is IGNORED (RS_GEMM_B3_NOUNDER=1: -DRS_TUNING builds
only; underflow_err.sh makes one in a scratch copy).  The model of tests/test_gpu_parity.py::test_activation_below_fp16_subnormal_range_...:
a hidden layer whose outputs are all of order `gain`, the next layer's weights divided by `gain` -- in exact arithmetic the same network
for every gain.  Prints max |log-likelihood - float64 oracle| for the split kernels with the flag ignored and for the exact kernels
(the flag honoured: tests/test_gpu_parity.py on the shipped build)."""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle import pipeline              # noqa: E402
from rhasspy_speech_amd import _lib, synth   # noqa: E402

pcms = [synth.synth_utterance(41, 32000), synth.synth_utterance(42, 20000)]
for gain in (1.0, 0.25, 0.0625, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
    td = Path(tempfile.mkdtemp())
    spec = synth.tiny_spec(hidden_dim=256, prefinal_dim=256, hidden_gain=gain, gain_compensated=True, seed=9)
    synth.write_model_dir(td / "model", spec)
    synth.make_grammar_graph(td / "graph", spec)
    orc = pipeline.Oracle(td / "model", td / "graph")
    saved = pipeline.Nnet3.matmul
    pipeline.Nnet3.matmul = staticmethod(lambda x, wt: (x.astype(np.float64) @ wt.astype(np.float64)).astype(np.float32))
    f64 = [orc.transcribe(p).loglikes for p in pcms]
    pipeline.Nnet3.matmul = saved
    os.environ["RS_GEMM_B3_NOUNDER"] = "1"
    m = _lib.Model(td / "model", td / "graph", _lib.default_opts(keep_intermediates=1))
    r = m.decode_batch(pcms)
    e_split = max(float(np.abs(r.matrix(i, 2) - f64[i]).max()) for i in range(2))
    os.environ["RS_GEMM_B3"] = "0"
    r = m.decode_batch(pcms)
    e_exact = max(float(np.abs(r.matrix(i, 2) - f64[i]).max()) for i in range(2))
    del os.environ["RS_GEMM_B3"]
    print(f"gain {gain:8.1e}: split kernels, flag ignored: max |ll - float64| {e_split:.2e}; exact-FP32 kernels {e_exact:.2e}")
