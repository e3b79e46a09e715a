"""Two models decoding concurrently from two host threads, compared with their sequential results (debug aid)."""
import sys, tempfile, threading
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rhasspy_speech_amd import _lib, synth
from tests import cases

with tempfile.TemporaryDirectory() as td:
    root = Path(td)
    spec = synth.ModelSpec()
    md2, gd2, _, _ = cases.build_case_files(cases.CASES["tinyf_u5"], root / "c2")
    synth.write_model_dir(root / "m1", spec); synth.make_grammar_graph(root / "g1", spec)
    import os
    if os.environ.get("STRESS_A") == "tinyf":
        m1 = _lib.Model(md2, gd2, _lib.default_opts())
    else:
        m1 = _lib.Model(root / "m1", root / "g1", _lib.default_opts())
    m2 = _lib.Model(md2, gd2, _lib.default_opts(keep_intermediates=1))
    pcm1 = [synth.synth_utterance(9000 + u, 48000 - 700 * (u % 5)) for u in range(96)]
    pcm2 = [synth.synth_utterance(9500 + u, 30000 + 900 * (u % 7)) for u in range(160)]
    ref1, ref2 = m1.decode_batch(pcm1), m2.decode_batch(pcm2)
    bad = 0
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        out = {}
        def run(key, model, pcms):
            out[key] = [model.decode_batch(pcms) for _ in range(3)]
        ts = [threading.Thread(target=run, args=("a", m1, pcm1)), threading.Thread(target=run, args=("b", m2, pcm2))]
        [t.start() for t in ts]; [t.join() for t in ts]
        for key, ref, n in ((("a", ref1, len(pcm1)),) if os.environ.get("STRESS_CHECK_A", "1") == "1" else ()) + (("b", ref2, len(pcm2)),):
            for res in out[key]:
                for u in range(n):
                    if res.words(u) != ref.words(u) or res.costs(u) != ref.costs(u):
                        bad += 1
                        msg = ""
                        if key == "b":
                            for kind, nm in ((0, "feat"), (1, "ivec"), (2, "loglikes")):
                                x, y = res.matrix(u, kind), ref.matrix(u, kind)
                                dd = np.abs(x - y)
                                rows_bad = np.nonzero(dd.reshape(len(dd), -1).max(axis=1) > 0)[0]
                                msg += f" {nm}: max {dd.max():.3g} rows {rows_bad[:6].tolist()}..{rows_bad[-3:].tolist()} of {len(dd)};"
                                if kind == 0:
                                    for rb in rows_bad[:2]:
                                        best = None
                                        for u2 in range(n):
                                            y2 = ref.matrix(u2, 0)
                                            e = np.abs(y2 - x[rb]).max(axis=1)
                                            if best is None or e.min() < best[0]:
                                                best = (float(e.min()), u2, int(e.argmin()))
                                        msg += f" [row {rb}: cols differing {int((dd[rb] > 0).sum())}/{dd.shape[1]}, closest reference frame: utt {best[1]} frame {best[2]} (err {best[0]:.3g})]"
                        print("mismatch", it, key, u, res.costs(u), ref.costs(u), msg, flush=True)
    print("iterations done, mismatching results:", bad)
