#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden vectors for randomised decode cases (build container only).

tests/fuzz_cases.py draws N random (acoustic-model shape, graph kind, utterance length, decoder options) combinations from a
seed; for each the REFERENCE's binaries (oracle/_ref) are run exactly as gen_golden.py runs them for the named parity cases --
`online2-wav-nnet3-latgen-faster | lattice-to-nbest --n=5 | nbest-to-linear` and the streaming `online2-cli-nnet3-decode-faster` --
and the n-best text and costs are stored in tests/golden/fuzz_decode.json (a few hundred bytes per case).
"""
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "oracle"))
from tests import cases, fuzz_cases  # noqa: E402
import gen_golden as gg  # noqa: E402

OUT = REPO / "tests" / "golden" / "fuzz_decode.json"
ENV = dict(os.environ, PATH=f"{gg.BIN}:{os.environ['PATH']}")


def run_case(case: dict) -> dict:
    out = {}
    with tempfile.TemporaryDirectory() as td:
        root = Path(td)
        model_dir, graph_dir, wav, pcm = cases.build_case_files(case, root)
        conf = model_dir / "model" / "online" / "conf" / "online.conf"
        mdl = model_dir / "model" / "model" / "final.mdl"
        for mode in ("offline", "stream"):
            lat = root / f"{mode}.lat"
            if mode == "offline":
                cmd = ["online2-wav-nnet3-latgen-faster", "--online=false", "--do-endpointing=false",
                       f"--word-symbol-table={graph_dir / 'words.txt'}", f"--config={conf}", *gg.decoder_args(case),
                       str(mdl), str(graph_dir / "HCLG.fst"), "ark:echo utt utt|", f"scp:echo utt {wav}|", f"ark:{lat}"]
                p = subprocess.run(cmd, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            else:
                cmd = ["online2-cli-nnet3-decode-faster", f"--config={conf}", *gg.decoder_args(case), str(mdl),
                       str(graph_dir / "HCLG.fst"), str(graph_dir / "words.txt"), f"ark:{lat}"]
                p = subprocess.run(cmd, env=ENV, input=pcm.astype("<i2").tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if p.returncode != 0:
                out[mode] = {"status": p.returncode}
                continue
            sh = (f"lattice-to-nbest --n={cases.NBEST} --acoustic-scale=1.0 ark:{lat} ark:- | "
                  f"nbest-to-linear ark:- ark:/dev/null ark,t:- ark,t:{root}/lm.txt ark,t:{root}/ac.txt")
            q = gg.run(["bash", "-c", sh], env=ENV)
            lm = gg.parse_vec_ark((root / "lm.txt").read_text())
            ac = gg.parse_vec_ark((root / "ac.txt").read_text())
            keys = sorted(lm, key=lambda k: int(k.split("-")[1]))
            out[mode] = {"status": 0, "nbest_text": q.stdout.decode(), "graph_cost": [lm[k][0] for k in keys], "acoustic_cost": [ac[k][0] for k in keys]}
    return out


def main():
    recs = []
    for i, case in enumerate(fuzz_cases.CASES):
        r = run_case(case)
        recs.append({"case": case, **r})
        off = r["offline"]
        print(i, case["graph"], case["audio"], case["spec"], "->", (off.get("nbest_text", "").strip().replace("\n", " | ") or f"status {off['status']}")[:80], flush=True)
    OUT.write_text(json.dumps(recs, indent=0))


if __name__ == "__main__":
    main()
