import sys, os, tempfile
from pathlib import Path
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tests import cases
from rhasspy_speech_amd import _lib
os.environ["RS_DECODER"] = sys.argv[1]
with tempfile.TemporaryDirectory() as td:
    md, gd, wav, pcm = cases.build_case_files(cases.CASES[sys.argv[2]], Path(td))
    m = _lib.Model(md, gd, _lib.default_opts(keep_intermediates=1))
    r = m.decode_batch([pcm])
    print("words", r.words(0), r.counters(0))
