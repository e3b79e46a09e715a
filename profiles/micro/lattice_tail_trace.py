import sys, time, tempfile
from pathlib import Path
sys.path.insert(0, '.')
from rhasspy_speech_amd import _lib
from tests import configs
md, gd = configs.build_grammar_model(Path(tempfile.gettempdir()) / "rs_lt")
pcms = configs.grammar_utterances()
for name, opts, nbest in (("nbest5", {}, 5), ("emit_lattice", {"emit_lattice": 1}, 1)):
    m = _lib.Model(md, gd, _lib.default_opts(**opts))
    m.decode_batch(pcms, nbest=nbest)
    t0 = time.perf_counter()
    for _ in range(3):
        r = m.decode_batch(pcms, nbest=nbest)
    print(name, "un-overlapped call", (time.perf_counter() - t0) / 3 * 1e3, "ms", [round(x, 2) for x in r.timings()], flush=True)
