#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- randomised comparison of rs_mkgraph with the REFERENCE's utils/mkgraph.sh (build container only).

For each seed: a random acoustic-model shape (context width / central position, chain or HMM topology with 1-3 states, phone
count), a random lexicon (homophones and prefix pronunciations forced in) and a random grammar or back-off G; both chains are
run on the same language directory and model, and the two HCLG.fst are compared with the reference's own
`fstequivalent --random=true` and with the library's isomorphism check.  Prints one line per seed; exits non-zero on the first
difference.  usage: python oracle/fuzz_mkgraph.py [first_seed [count]]
"""
import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "oracle"))
from rhasspy_speech_amd import _lib, synth  # noqa: E402
import gen_rescore_golden as rg  # noqa: E402

BIN = REPO / "oracle" / "_ref" / "bin"
MKGRAPH = Path("/root/reference/kaldi/egs/wsj/s5/utils/mkgraph.sh")
ENV = dict(os.environ, PATH=f"{BIN}:{os.environ['PATH']}")


def sh(cmd, cwd=None):
    p = subprocess.run(["bash", "-c", "set -o pipefail; " + cmd], env=ENV, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if p.returncode != 0:
        raise RuntimeError(f"{cmd}\n{p.stdout.decode()[-1500:]}\n{p.stderr.decode()[-1500:]}")
    return p.stdout.decode()


def one(seed: int) -> str:
    rng = np.random.default_rng(seed)
    ctx = ["mono", "biphone", "triphone", "2,0", "3,2", "3,0", "4,1", "4,2"][int(rng.integers(0, 8))]
    chain = bool(rng.integers(0, 2))
    spec = synth.tiny_spec(context=ctx, chain_topology=chain, hmm_states=1 if chain else int(rng.integers(1, 4)),
                           num_phones=int(rng.integers(8, 30)), seed=seed)
    td = Path(tempfile.mkdtemp())
    try:
        synth.write_model_dir(td / "model", spec)
        mdl = td / "model" / "model" / "model"
        sents = [s.split() for s in synth.DEFAULT_SENTENCES]
        lex = synth.make_lexicon(sents, spec, rng)
        ids = {w: i for i, w in enumerate(lex.words)}
        vocab = lex.words[1:]
        for _ in range(int(rng.integers(0, 4))):          # homophones
            a, b = rng.choice(vocab, 2, replace=False)
            lex.prons[ids[a]] = list(lex.prons[ids[b]])
        for _ in range(int(rng.integers(0, 4))):          # prefixes
            a, b = rng.choice(vocab, 2, replace=False)
            lex.prons[ids[a]] = list(lex.prons[ids[b]][:max(1, len(lex.prons[ids[b]]) - 1)])
        conf = dict(keep_every=int(rng.integers(1, 4)), extra_sentences=int(rng.integers(0, 30)), backoff=bool(rng.integers(0, 2)))
        lang = td / "lang"
        rg.write_lang(lang, lex, spec, conf, rng)
        (lang / "phones.txt").write_text("<eps> 0\n" + "".join(f"p{i} {i}\n" for i in range(1, spec.num_phones + 1)))
        (lang / "phones" / "silence.csl").write_text(f"{lex.sil_phone}\n")
        loop = [1.0, 0.1][int(rng.integers(0, 2))]
        sh(f"bash {MKGRAPH} --self-loop-scale {loop} {lang} {mdl} {td}/ref 2>&1", cwd=td)
        _lib.mkgraph(lang, mdl, td / "mine", self_loop_scale=loop)
        sh(f"fstequivalent --random=true --delta=0.003 {td}/mine/HCLG.fst {td}/ref/HCLG.fst")
        try:
            _lib.fst_tool("fstisomorphic", td / "mine" / "HCLG.fst", td / "ref" / "HCLG.fst", param=1.5 / 1024)
            iso = "isomorphic"
        except _lib.RsError as e:
            iso = "equivalent, not isomorphic (" + str(e)[:60] + ")"
        n = (td / "ref" / "HCLG.fst").stat().st_size
        return f"seed {seed}: context {ctx} {'chain' if chain else f'hmm{spec.hmm_states}'} phones {spec.num_phones} {conf} loop {loop}: {n} bytes, {iso}"
    finally:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    for s in range(first, first + count):
        print(one(s), flush=True)
