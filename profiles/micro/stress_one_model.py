"""One model, the same batch decoded repeatedly (two concurrent sub-batch groups inside the library): are results stable?"""
import sys, tempfile
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rhasspy_speech_amd import _lib, synth
with tempfile.TemporaryDirectory() as td:
    root = Path(td); spec = synth.ModelSpec()
    synth.write_model_dir(root / "m1", spec); synth.make_grammar_graph(root / "g1", spec)
    m1 = _lib.Model(root / "m1", root / "g1", _lib.default_opts(keep_intermediates=0))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    pcms = [synth.synth_utterance(9000 + u, 48000 - 700 * (u % 5)) for u in range(n)]
    ref = m1.decode_batch(pcms)
    bad = 0
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
        r = m1.decode_batch(pcms)
        for u in range(n):
            if r.words(u) != ref.words(u) or r.costs(u) != ref.costs(u):
                bad += 1
                if bad < 5: print("mismatch", it, u, r.costs(u), ref.costs(u), flush=True)
    print("mismatching results:", bad)
