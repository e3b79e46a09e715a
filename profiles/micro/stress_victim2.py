"""Known-answer victim kernel (private LDS + registers per wave) beside several decode pipelines of one model."""
import ctypes, subprocess, sys, tempfile, threading
from pathlib import Path
HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))
from rhasspy_speech_amd import _lib, synth
so = Path(tempfile.mkdtemp()) / "libburner.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", str(so), str(HERE / "mfma_burner.hip")], check=True, stderr=subprocess.DEVNULL)
lib = ctypes.CDLL(str(so)); lib.run_victim.restype = ctypes.c_long
with tempfile.TemporaryDirectory() as td:
    root = Path(td); spec = synth.ModelSpec()
    synth.write_model_dir(root / "m", spec); synth.make_grammar_graph(root / "g", spec)
    m = _lib.Model(root / "m", root / "g", _lib.default_opts())
    batches = [[synth.synth_utterance(21000 + 100 * b + u, 48000 - 320 * ((u + b) % 11)) for u in range(24 + 16 * b)] for b in range(3)]
    ref = [m.decode_batch(p) for p in batches]
    print("victim alone:", lib.run_victim(2048, 200, 20), flush=True)
    stop = False
    bad = [0]
    def bg(b):
        while not stop:
            r = m.decode_batch(batches[b])
            for u in range(len(batches[b])):
                if r.costs(u) != ref[b].costs(u): bad[0] += 1
    ths = [threading.Thread(target=bg, args=(b,)) for b in range(3)]
    [t.start() for t in ths]
    tot = 0
    for i in range(80):
        tot += lib.run_victim(2048, 200, 20)
    stop = True; [t.join() for t in ths]
    print("victim beside 3 decode pipelines: corrupted words", tot, "; decode results that differed meanwhile:", bad[0], flush=True)
