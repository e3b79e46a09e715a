import sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rhasspy_speech_amd import _lib, synth
with tempfile.TemporaryDirectory() as td:
    root = Path(td); spec = synth.ModelSpec()
    synth.write_model_dir(root / "m", spec); synth.make_grammar_graph(root / "g", spec)
    m = _lib.Model(root / "m", root / "g", _lib.default_opts())
    pcms = [synth.synth_utterance(u, 48000) for u in range(8)]
    r = m.decode_batch(pcms)
    for u in range(4):
        c = r.counters(u); print(u, c, "alive/frame", c[3] / max(r.num_frames(u), 1), "arcs/frame", c[1] / max(r.num_frames(u), 1))
    print(m.describe().split("hclg")[1][:200])
