"""rs_decode_batch(nbest = 5) of the headline batch against the number of host threads of the lattice tail (RS_LATTICE_THREADS, read at the
first call: one process per setting).  usage (GPU box): for t in 8 16 32 48; do RS_LATTICE_THREADS=$t python profiles/micro/nbest_threads.py; done"""
import os, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rhasspy_speech_amd import _lib
from tests import configs
md, gd = configs.build_grammar_model(Path(tempfile.mkdtemp()) / "c1")
m = _lib.Model(md, gd, _lib.default_opts())
pcms = configs.grammar_utterances(256)
for nb, lat in ((5, 0), ):
    for _ in range(3): m.decode_batch(pcms, nbest=nb)
    t = time.perf_counter()
    for _ in range(10): m.decode_batch(pcms, nbest=nb)
    r = m.decode_batch(pcms, nbest=nb)
    print("   timings [h2d, mfcc, ivector, nnet, search, lattice kernel + d2h, whole call, host lattice tail]:", [round(x, 2) for x in r.timings()])
    print(f"RS_LATTICE_THREADS={os.environ.get('RS_LATTICE_THREADS', 'default')}: nbest={nb} {1e3 * (time.perf_counter() - t) / 10:.2f} ms per call (one call at a time), {os.cpu_count()} cpus")
