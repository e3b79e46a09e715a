"""A kernel with a known answer (private LDS region + registers per wave, profiles/micro/mfma_burner.hip:victim) runs beside a
model's decode; reports how many of its words changed under it."""
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
    synth.write_model_dir(root / "m1", spec); synth.make_grammar_graph(root / "g1", spec)
    m1 = _lib.Model(root / "m1", root / "g1", _lib.default_opts())
    pcm1 = [synth.synth_utterance(9000 + u, 48000 - 700 * (u % 5)) for u in range(96)]
    print("victim alone:", lib.run_victim(2048, 200, 20), flush=True)
    stop = False
    def bg():
        while not stop:
            m1.decode_batch(pcm1)
    th = threading.Thread(target=bg); th.start()
    tot = 0
    for i in range(60):
        tot += lib.run_victim(2048, 200, 20)
    stop = True; th.join()
    print("victim beside the model's decode: corrupted words", tot, flush=True)
