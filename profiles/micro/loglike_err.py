import sys, tempfile
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from tests import cases
from rhasspy_speech_amd import _lib
for name in ["zam_u0", "zam_u1", "zam_real_cold", "zam_long30", "tinyf_u5", "tiny_arpa_u7"]:
    g = np.load(cases.GOLDEN / f"{name}.npz")
    with tempfile.TemporaryDirectory() as td:
        md, gd, wav, pcm = cases.build_case_files(cases.CASES[name], Path(td))
        o = dict(keep_intermediates=1); o.update(cases.CASES[name].get("opts", {}))
        m = _lib.Model(md, gd, _lib.default_opts(**o))
        r = m.decode_batch([pcm], nbest=1)
        ll = r.matrix(0, 2); sr, sc = g["loglikes_stride"]
        d = np.abs(ll[::sr, ::sc] - g["offline_loglikes"])
        ref = [int(x) for x in bytes(g["offline_nbest_text"]).decode().splitlines()[0].split()[1:]]
        print(name, "max", d.max(), "mean", d.mean(), "words_ok", r.words(0, 0) == ref, flush=True)
