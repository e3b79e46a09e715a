"""1024 ragged utterances in one call (0.3 - 6 s), checked against per-utterance decodes of a sample."""
import sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rhasspy_speech_amd import _lib, synth
with tempfile.TemporaryDirectory() as td:
    root = Path(td); spec = synth.ModelSpec()
    synth.write_model_dir(root / "m", spec); synth.make_grammar_graph(root / "g", spec)
    m = _lib.Model(root / "m", root / "g", _lib.default_opts())
    rng = np.random.default_rng(5)
    pcms = [synth.synth_utterance(40000 + u, int(rng.integers(4800, 96000))) for u in range(1024)]
    t = time.time(); r = m.decode_batch(pcms); dt = time.time() - t
    bad = 0
    for u in list(range(0, 1024, 37)) + [1023]:
        one = m.decode_batch([pcms[u]])
        if one.words(0) != r.words(u) or one.costs(0) != r.costs(u):
            bad += 1
    print(f"1024 utterances, {sum(len(p) for p in pcms) / 16000:.0f} s of audio in {dt * 1e3:.1f} ms; mismatches against single decodes: {bad}")
