import sys, tempfile, os
from pathlib import Path
import numpy as np
sys.path.insert(0, ".")
import torch
from rhasspy_speech_amd import _lib
from tests import configs
md, gd = configs.build_grammar_model(Path(tempfile.gettempdir()) / "rs_dbg_grammar")
model = _lib.Model(md, gd, _lib.default_opts())
pcms = configs.grammar_utterances(3) + [np.zeros(200, np.int16)]
print("A plain 3", flush=True)
r = model.decode_batch(pcms[:3]); print(r.words(0), flush=True)
print("B with short", flush=True)
sys.stderr.write("=====B=====\n"); sys.stderr.flush()
r = model.decode_batch(pcms); print(r.words(0), flush=True)
print("done", flush=True)
