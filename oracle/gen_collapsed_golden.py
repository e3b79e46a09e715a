#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- the network each parity case's model becomes under the reference's CollapseModel (nnet3/nnet-utils.cc:2116,
called by both decoder binaries at load): `rs-dump collapsed` = AmNnetSimple::Read + SetBatchnormTestMode + SetDropoutTestMode +
CollapseModel + Nnet::GetConfigLines on the reference's own classes (oracle/_ref; build container only).  Stored as
tests/golden/collapsed_configs.json {case: [config lines]}; tests/test_oracle_golden.py compares rs_nnet3_setup's text with it.
Usage: python oracle/gen_collapsed_golden.py"""
import json, os, subprocess, sys, tempfile
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from rhasspy_speech_amd import synth  # noqa: E402
from tests import cases  # noqa: E402
ENV = dict(os.environ, PATH=f"{REPO / 'oracle' / '_ref' / 'bin'}:{os.environ['PATH']}")
out = {}
with tempfile.TemporaryDirectory() as td:
    for name in sorted(cases.CASES):
        root = Path(td) / name
        synth.write_model_dir(root, cases.case_spec(cases.CASES[name]))
        conf = root / "model" / "online" / "conf" / "online.conf"
        r = subprocess.run(["rs-dump", f"--config={conf}", "collapsed", str(root / "model" / "model" / "final.mdl"), "-", "-"], env=ENV,
                           capture_output=True, text=True, check=True)
        out[name] = [l for l in r.stdout.splitlines() if l.strip()]
        print(name, len(out[name]), "lines")
(REPO / "tests" / "golden" / "collapsed_configs.json").write_text(json.dumps(out, indent=0))
