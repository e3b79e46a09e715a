"""Debug: concurrent decodes of different batches on one model, truncated after a pipeline stage (RS_DEBUG_STOP_AFTER) --
which kernels have to run side by side for two runs of the MFCC kernel to disagree (RS_DEBUG_MFCC2 lines on stderr)."""
import sys, tempfile, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rhasspy_speech_amd import _lib, synth
with tempfile.TemporaryDirectory() as td:
    root = Path(td); spec = synth.ModelSpec()
    synth.write_model_dir(root / "m", spec); synth.make_grammar_graph(root / "g", spec)
    m = _lib.Model(root / "m", root / "g", _lib.default_opts())
    batches = [[synth.synth_utterance(21000 + 100 * b + u, 48000 - 320 * ((u + b) % 11)) for u in range(24 + 16 * b)] for b in range(4)]
    def run(b):
        for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
            try:
                m.decode_batch(batches[b])
            except _lib.RsError:
                pass
    ts = [threading.Thread(target=run, args=(b,)) for b in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
