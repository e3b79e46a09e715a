"""rs_decode_batch(nbest = 5) of the headline batch from k host threads (k calls in flight): the step time and the mean of the calls'
own stage timers -- which stage of a call stretches when other calls are in flight.  usage (GPU box): python profiles/micro/nbest_inflight.py [k ...]"""
import concurrent.futures, os, sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("RS_CONTEXTS", "8")
from rhasspy_speech_amd import _lib
from tests import configs
md, gd = configs.build_grammar_model(Path(tempfile.mkdtemp()) / "c1")
m = _lib.Model(md, gd, _lib.default_opts())
pcms = configs.grammar_utterances(256)
nb = int(os.environ.get("NBEST", "5"))
for k in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 6]:
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=k)
    fn = lambda: m.decode_batch(pcms, nbest=nb)
    for f in [pool.submit(fn) for _ in range(2 * k)]: f.result()
    n = 40
    t = time.perf_counter()
    tm = np.zeros(8)
    for f in [pool.submit(fn) for _ in range(n)]:
        tm += np.array(f.result().timings())
    dt = (time.perf_counter() - t) / n
    print(f"{k} in flight, nbest={nb}: {1e3 * dt:.2f} ms per step; mean call [h2d, mfcc, ivector, nnet, search, -, whole call, lattice tail] = {[round(x, 2) for x in tm / n]}")
    pool.shutdown()
