#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden vectors for the lattice wire format (build container only).

For a few parity cases the REFERENCE's online2-wav-nnet3-latgen-faster (oracle/_ref, built from /root/reference by
oracle/build_ref.sh) writes its CompactLattice; `lattice-to-nbest --n=N | nbest-to-linear` of the same build lists every
path of that lattice with its alignment (transition-ids), words and (graph, acoustic) costs.  Those lists are committed as
tests/golden/lattice/<case>.json; the GPU test renders the library's lattice (rs_result_lattice), pushes the bytes through
the same two reference tools and compares path by path.  Nothing of the reference travels.
"""
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from tests.cases import CASES, build_case_files  # noqa: E402
from oracle.gen_golden import BIN, decoder_args  # noqa: E402

OUT = REPO / "tests" / "golden" / "lattice"
LATTICE_CASES = ["tiny_u0", "tiny_u3_short", "tiny_real_hot", "tiny_noiv_u2", "tinyf_u5", "tiny_hmm_u6", "tiny_arpa_u7",
                 "tiny_arpa_prune_u8", "zam_u1"]
N_PATHS = 200


def paths_of_lattice(lat: Path, env, n: int = N_PATHS):
    """[(words, alignment, graph cost, acoustic cost)] in lattice-to-nbest order."""
    with tempfile.TemporaryDirectory() as td:
        sh = (f"lattice-to-nbest --n={n} --acoustic-scale=1.0 ark:{lat} ark:- | "
              f"nbest-to-linear ark:- ark,t:{td}/ali ark,t:{td}/words ark,t:{td}/lm ark,t:{td}/ac")
        subprocess.run(["bash", "-c", sh], env=env, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)

        def table(fn, conv):
            t = {}
            for line in Path(td, fn).read_text().splitlines():
                p = line.split()
                if p:
                    t[p[0]] = [conv(x) for x in p[1:]]
            return t
        ali, words, lm, ac = table("ali", int), table("words", int), table("lm", float), table("ac", float)
        keys = sorted(words, key=lambda k: int(k.rsplit("-", 1)[1]))
        return [dict(words=words[k], ali=ali[k], graph=lm[k][0], acoustic=ac[k][0]) for k in keys]


def main():
    env = dict(os.environ, PATH=f"{BIN}:{os.environ['PATH']}")
    OUT.mkdir(parents=True, exist_ok=True)
    for name in sys.argv[1:] or LATTICE_CASES:
        case = CASES[name]
        with tempfile.TemporaryDirectory() as td:
            root = Path(td)
            model_dir, graph_dir, wav, _ = build_case_files(case, root)
            conf = model_dir / "model" / "online" / "conf" / "online.conf"
            mdl = model_dir / "model" / "model" / "final.mdl"
            lat = root / "lat.ark"
            cmd = ["online2-wav-nnet3-latgen-faster", "--online=false", "--do-endpointing=false",
                   f"--word-symbol-table={graph_dir / 'words.txt'}", f"--config={conf}", *decoder_args(case),
                   str(mdl), str(graph_dir / "HCLG.fst"), "ark:echo utt utt|", f"scp:echo utt {wav}|", f"ark:{lat}"]
            subprocess.run(cmd, env=env, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            paths = paths_of_lattice(lat, env)
            (OUT / f"{name}.json").write_text(json.dumps(dict(n_requested=N_PATHS, paths=paths)))
            print(f"{name}: {len(paths)} paths, best {paths[0]['words']} total {paths[0]['graph'] + paths[0]['acoustic']:.3f}, "
                  f"worst total {paths[-1]['graph'] + paths[-1]['acoustic']:.3f}")


if __name__ == "__main__":
    main()
