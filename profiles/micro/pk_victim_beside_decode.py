"""The known-answer kernels of pk_repro2.hip (the instruction the ISA bisect singled out, and its clean commuted form) on their own
stream while three host threads keep decode calls of the library in flight on the same device: does the instruction misbehave beside
the REAL pipelines?   usage (GPU box): hipcc --offload-arch=gfx950 -O2 -DPK_LIBRARY -shared -fPIC -o /tmp/libpkvictim.so
profiles/micro/pk_repro2.hip && python profiles/micro/pk_victim_beside_decode.py"""
import ctypes as C
import sys, tempfile, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rhasspy_speech_amd import _lib, synth
vic = C.CDLL("/tmp/libpkvictim.so")
vic.pk_victim_run.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
with tempfile.TemporaryDirectory() as td:
    root = Path(td); spec = synth.ModelSpec()
    synth.write_model_dir(root / "m", spec); synth.make_grammar_graph(root / "g", spec)
    m = _lib.Model(root / "m", root / "g", _lib.default_opts())
    batches = [[synth.synth_utterance(21000 + 100 * b + u, 48000 - 320 * ((u + b) % 11)) for u in range(40 + 16 * b)] for b in range(3)]
    def decode(b):
        try:
            m.decode_batch(batches[b])
        except _lib.RsError:          # (pk_perturber2.sh runs this on builds whose GEMM is wrong by design)
            pass
    for b in range(3): decode(b)
    stop = False
    def run(b):
        while not stop:
            decode(b)
    for beside in (False, True):
        stop = False
        ts = [threading.Thread(target=run, args=(b,)) for b in range(3)] if beside else []
        [t.start() for t in ts]
        for seq, name in ((9, "v_pk_mul in place, source 1 half-swapped"), (11, "the same, next instruction overwrites source 1"),
                          (12, "operands commuted (clean in the bisect)"), (2, "v_pk_add, s_nop 0, 32-bit consumers")):
            w, p = C.c_uint(0), C.c_uint(0)
            rc = vic.pk_victim_run(seq, 20000, 24, C.byref(w), C.byref(p))
            print(f"{'beside three decode threads' if beside else 'alone':28s} {name:52s} rc {rc}  {w.value} wrong of {24 * 512 * 256 * 2 * 20000 / 1e9:.1f} G values ({p.value} the predicted wrong value)", flush=True)
        stop = True
        [t.join() for t in ts]
