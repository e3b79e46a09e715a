"""Several host threads decoding DIFFERENT batches on one model at the same time; any result that differs from the
sequential one is reported with the first stage (features / iVector / log-likelihoods) in which it differs."""
import os, sys, tempfile, threading
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from rhasspy_speech_amd import _lib, synth
with tempfile.TemporaryDirectory() as td:
    root = Path(td); spec = synth.ModelSpec()
    synth.write_model_dir(root / "m", spec); synth.make_grammar_graph(root / "g", spec)
    m = _lib.Model(root / "m", root / "g", _lib.default_opts(keep_intermediates=int(os.environ.get('STRESS_KEEP', '1'))))
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    batches = [[synth.synth_utterance(21000 + 100 * b + u, 48000 - 320 * ((u + b) % 11)) for u in range(24 + 16 * b)] for b in range(nb)]
    ref = [m.decode_batch(p) for p in batches]
    if os.environ.get('STRESS_STOP_AFTER'):
        os.environ['RS_DEBUG_STOP_AFTER'] = os.environ['STRESS_STOP_AFTER']; os.environ['RS_DEBUG_STOP_MINUTTS'] = os.environ.get('STRESS_STOP_MIN', '0'); os.environ['RS_DEBUG_STOP_MAXUTTS'] = os.environ.get('STRESS_STOP_MAX', '60')
    bad = 0
    per_batch = [0] * nb
    per_it = []
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        out = [None] * nb
        def run(b):
            try:
                out[b] = m.decode_batch(batches[b])
            except _lib.RsError:
                out[b] = None
        ts = [threading.Thread(target=run, args=(b,)) for b in range(nb)]
        [t.start() for t in ts]; [t.join() for t in ts]
        before = bad
        for b in range(nb):
            if out[b] is None: continue
            for u in range(len(batches[b])):
                if out[b].words(u) != ref[b].words(u) or out[b].costs(u) != ref[b].costs(u):
                    bad += 1
                    per_batch[b] += 1
                    msg = ""
                    for kind, nm in () if os.environ.get('STRESS_KEEP', '1') == '0' else ((0, "feat"), (1, "ivec"), (2, "loglikes")):
                        d = np.abs(out[b].matrix(u, kind) - ref[b].matrix(u, kind))
                        rows = np.nonzero(d.reshape(len(d), -1).max(axis=1) > 0)[0]
                        msg += f" {nm}: max {d.max():.3g} rows {rows[:4].tolist()}..{rows[-2:].tolist()} ({len(rows)} of {len(d)});"
                    if bad <= 8: print(f"iteration {it} batch {b} utt {u}:{msg}", flush=True)
                    if bad <= 3 and os.environ.get('STRESS_SHOW_ROWS') and os.environ.get('STRESS_KEEP', '1') != '0':
                        # the first differing feature rows, both versions (which cepstra move, and how)
                        f1, f0 = out[b].matrix(u, 0), ref[b].matrix(u, 0)
                        for r in np.nonzero(np.abs(f1 - f0).max(axis=1) > 0)[0][:2]:
                            print(f"   row {r} got  {np.array2string(f1[r][:12], precision=3)}\n   row {r} want {np.array2string(f0[r][:12], precision=3)}", flush=True)
        per_it.append(bad - before)
    print("per iteration:", per_it, "per batch:", per_batch)
    print("mismatching results:", bad)
