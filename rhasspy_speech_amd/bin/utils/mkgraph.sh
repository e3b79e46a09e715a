#!/usr/bin/env python3
# utils/mkgraph.sh of the reference's Kaldi tree (kaldi/egs/wsj/s5/utils/mkgraph.sh) by name and argv, backed by rs_mkgraph:
#   mkgraph.sh [--remove-oov] [--transition-scale T] [--self-loop-scale S] <lang-dir> <model-dir> <graphdir>
# The reference's KaldiTrainer._mkgraph (rhasspy_speech/kaldi.py:409-425) runs `bash <kaldi_dir>/utils/mkgraph.sh ...`, i.e. a shell
# reads this file whatever its first line says: the file is a shell / Python polyglot, the next three lines make a shell
# re-execute it under python3 and are a string literal to Python.
"true" '''\'
exec python3 "$0" "$@"
'''
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent.parent))


def main(argv):
    tscale, loopscale = 1.0, 0.1
    args = list(argv)
    for _ in range(4):                     # the script looks at its options four times, in any order
        if args and args[0] in ("--mono", "--left-biphone", "--quinphone"):
            print("WARNING: the --mono, --left-biphone and --quinphone options are now deprecated and ignored.")
            args = args[1:]
        if args and args[0] == "--remove-oov":
            print("mkgraph.sh: --remove-oov is not supported by this implementation", file=sys.stderr)
            return 1
        if len(args) >= 2 and args[0] == "--transition-scale":
            tscale, args = float(args[1]), args[2:]
        if len(args) >= 2 and args[0] == "--self-loop-scale":
            loopscale, args = float(args[1]), args[2:]
    if len(args) != 3:
        print("Usage: utils/mkgraph.sh [options] <lang-dir> <model-dir> <graphdir>\n"
              "e.g.: utils/mkgraph.sh data/lang_test exp/tri1/ exp/tri1/graph\n"
              " Options:\n"
              " --transition-scale #  Scaling factor on transition probabilities.\n"
              " --self-loop-scale  #  Please see: http://kaldi-asr.org/doc/hmm.html#hmm_scale.")
        return 1
    from rhasspy_speech_amd import _lib
    lang, model, graph = args
    hclg = Path(graph) / "HCLG.fst"
    required = [Path(lang) / "L_disambig.fst", Path(lang) / "G.fst", Path(lang) / "words.txt", Path(lang) / "phones" / "disambig.int",
                Path(model) / "final.mdl", Path(model) / "tree"]
    if hclg.exists() and all(r.exists() and r.stat().st_mtime <= hclg.stat().st_mtime for r in required):
        print(f"{sys.argv[0]}: {hclg} is up to date.")        # mkgraph.sh:60-71
        return 0
    try:
        _lib.mkgraph(lang, model, graph, self_loop_scale=loopscale, transition_scale=tscale)
    except _lib.RsError as e:
        print(str(e), file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
