"""RS_LDS_POISON debug aid: every model case decoded with and without LDS poisoning in front of each kernel; a difference
means some kernel reads LDS it did not write."""
import os, subprocess, sys, tempfile, json
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from rhasspy_speech_amd import _lib, synth
    from tests import cases
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name in ["tinyf_u5", "tiny_u0", "tiny_arpa_u7", "zam_u1"]:
            md, gd, _, pcm = cases.build_case_files(cases.CASES[name], Path(td) / name)
            m = _lib.Model(md, gd, _lib.default_opts(keep_intermediates=1))
            pcms = [synth.synth_utterance(9500 + u, 30000 + 900 * (u % 7)) for u in range(40)]
            r = m.decode_batch(pcms)
            out[name] = dict(costs=[list(r.costs(u)) for u in range(40)],
                             feat=[float(np.abs(r.matrix(u, 0)).sum()) for u in range(40)],
                             iv=[float(np.abs(r.matrix(u, 1)).sum()) for u in range(40)],
                             ll=[float(np.abs(r.matrix(u, 2)).sum()) for u in range(40)])
    print("JSON" + json.dumps(out))
else:
    res = {}
    for p in ("0", "1"):
        o = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, RS_LDS_POISON=p), capture_output=True, text=True)
        line = [l for l in o.stdout.splitlines() if l.startswith("JSON")]
        if not line:
            print(o.stderr[-2000:]); sys.exit(1)
        res[p] = json.loads(line[0][4:])
    for name in res["0"]:
        for key in ("feat", "iv", "ll", "costs"):
            a, b = np.array(res["0"][name][key]), np.array(res["1"][name][key])
            bad = np.nonzero(np.abs(a - b).reshape(len(a), -1).max(axis=1) > 0)[0]
            print(name, key, "differs for utts", bad.tolist() if len(bad) else "none")
