#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden vectors for decoding-graph construction (build container only).

For every case: a synthetic acoustic model with a phonetic decision tree (rhasspy_speech_amd.synth: monophone, left-biphone or
triphone context), a language directory (lexicon transducer with disambiguation symbols + grammar / back-off G, written by
gen_rescore_golden.write_lang and compiled with the reference's fstcompile), and then the REFERENCE's own
`utils/mkgraph.sh --self-loop-scale 1.0 <lang> <model> <graph>` (kaldi/egs/wsj/s5/utils/mkgraph.sh, the call of
rhasspy_speech/kaldi.py:409-425) run where it lies, on the reference's tools compiled under oracle/_ref/bin.  The chain's
intermediate transducers are captured too (the script's own lang/tmp files, and Ha / HCLGa re-made with the script's
commands), and the reference decoder is run on the finished HCLG for a few utterances.

Committed under tests/golden/mkgraph/<case>/: lang/{L_disambig.fst,G.fst,words.txt,phones/disambig.int}, ref/{LG.fst,
CLG.fst,ilabels,Ha.fst,disambig_tid.int,HCLGa_det.fst,HCLGa.fst,HCLG.fst} and decode.json (n-best texts and costs of the
reference decoder on the reference graph).  Nothing of the reference travels: these are its outputs.
"""
import json
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
from rhasspy_speech_amd import synth  # noqa: E402
import gen_rescore_golden as rg  # noqa: E402  (write_lang)
from tests import mkgraph_cases as mc  # noqa: E402

BIN = REPO / "oracle" / "_ref" / "bin"
OUT = REPO / "tests" / "golden" / "mkgraph"
MKGRAPH = Path("/root/reference/kaldi/egs/wsj/s5/utils/mkgraph.sh")
ENV = dict(os.environ, PATH=f"{BIN}:{os.environ['PATH']}")


def sh(cmd: str, cwd=None) -> str:
    p = subprocess.run(["bash", "-c", "set -o pipefail; " + cmd], env=ENV, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if p.returncode != 0:
        raise RuntimeError(f"{cmd}\n{p.stdout.decode()[-2000:]}\n{p.stderr.decode()[-2000:]}")
    return p.stdout.decode()


def gen_case(name: str, case: dict) -> None:
    td = Path(tempfile.mkdtemp())
    spec = mc.case_spec(case)
    model_dir = td / "model"
    synth.write_model_dir(model_dir, spec)
    mdl_dir = model_dir / "model" / "model"
    lang = td / "lang"
    lex = mc.case_lexicon(case, spec)
    rg.write_lang(lang, lex, spec, case.get("lang_conf") or rg.LANGS[case["lang"]], np.random.default_rng(case.get("lang_seed", 7)))
    # the files mkgraph.sh insists on seeing (it only tests that they exist)
    (lang / "phones.txt").write_text("<eps> 0\n" + "".join(f"p{i} {i}\n" for i in range(1, spec.num_phones + 1)))
    (lang / "phones" / "silence.csl").write_text(f"{lex.sil_phone}\n")
    graph = td / "graph"
    import time
    t0 = time.time()
    log = sh(f"bash {MKGRAPH} --self-loop-scale {case['self_loop_scale']} {lang} {mdl_dir} {graph} 2>&1", cwd=td)
    ref_seconds = time.time() - t0
    n_ctx, p_ctx = synth.context_shape(spec)
    ref = (td / "ref") if (case.get("big") or case.get("light")) else (OUT / name / "ref")
    ref.mkdir(parents=True)
    shutil.copy(lang / "tmp" / "LG.fst", ref / "LG.fst")
    shutil.copy(lang / "tmp" / f"CLG_{n_ctx}_{p_ctx}.fst", ref / "CLG.fst")
    shutil.copy(lang / "tmp" / f"ilabels_{n_ctx}_{p_ctx}", ref / "ilabels")
    shutil.copy(graph / "HCLG.fst", ref / "HCLG.fst")
    shutil.copy(graph / "disambig_tid.int", ref / "disambig_tid.int")
    # Ha / HCLGa: the script deletes them; the same commands again (mkgraph.sh:128-150)
    sh(f"make-h-transducer --disambig-syms-out={td}/dis.int --transition-scale=1.0 {ref}/ilabels {mdl_dir}/tree {mdl_dir}/final.mdl > {ref}/Ha.fst")
    sh(f"fsttablecompose {ref}/Ha.fst {ref}/CLG.fst | fstdeterminizestar --use-log=true > {ref}/HCLGa_det.fst")
    sh(f"fstrmsymbols {td}/dis.int {ref}/HCLGa_det.fst | fstrmepslocal | fstminimizeencoded > {ref}/HCLGa.fst")
    # stage inputs for single-step tests: L o G before determinisation, LG before pushing
    sh(f"fsttablecompose {lang}/L_disambig.fst {lang}/G.fst > {ref}/LG_composed.fst")
    sh(f"fstdeterminizestar --use-log=true {ref}/LG_composed.fst > {ref}/LG_det.fst")
    sh(f"fstminimizeencoded {ref}/LG_det.fst > {ref}/LG_min.fst")
    # (check: the re-made HCLGa gives the script's HCLG)
    sh(f"add-self-loops --self-loop-scale={case['self_loop_scale']} --reorder=true {mdl_dir}/final.mdl {ref}/HCLGa.fst | fstconvert --fst_type=const > {td}/HCLG2.fst")
    assert (td / "HCLG2.fst").read_bytes() == (ref / "HCLG.fst").read_bytes(), "re-made chain differs from mkgraph.sh's output"
    if case.get("light"):          # light cases keep the finished graph only
        (OUT / name / "ref").mkdir(parents=True)
        for f in ["HCLG.fst", "disambig_tid.int"]:
            shutil.copy(ref / f, OUT / name / "ref" / f)
    dst_lang = OUT / name / "lang"
    (dst_lang / "phones").mkdir(parents=True)
    for f in ["L_disambig.fst", "G.fst", "words.txt", "phones/disambig.int"]:
        shutil.copy(lang / f, dst_lang / f)
    # ---- the reference decoder on the reference graph
    conf = model_dir / "model" / "online" / "conf" / "online.conf"
    dec = []
    for u in case["utts"]:
        pcm = synth.synth_utterance(u, 48000)
        wav = td / f"u{u}.wav"
        synth.write_wav(wav, pcm)
        lat = td / f"u{u}.lat"
        sh(f"online2-wav-nnet3-latgen-faster --online=false --do-endpointing=false --word-symbol-table={graph}/words.txt --config={conf} "
           f"--max-active=7000 --lattice-beam=8.0 --acoustic-scale=1.0 --beam=24.0 {mdl_dir}/final.mdl {graph}/HCLG.fst 'ark:echo utt utt|' "
           f"'scp:echo utt {wav}|' ark:{lat}")
        text = sh(f"lattice-to-nbest --n=5 --acoustic-scale=1.0 ark:{lat} ark:- | nbest-to-linear ark:- ark:/dev/null ark,t:- ark,t:{td}/lm.txt ark,t:{td}/ac.txt")
        lm = [float(ln.split()[1]) for ln in (td / "lm.txt").read_text().splitlines() if ln.strip()]
        ac = [float(ln.split()[1]) for ln in (td / "ac.txt").read_text().splitlines() if ln.strip()]
        dec.append({"utt": u, "nbest_text": text, "graph_cost": lm, "acoustic_cost": ac})
    from rhasspy_speech_amd import _lib
    t0 = time.time()
    _lib.mkgraph(lang, mdl_dir, td / "graph_mine", self_loop_scale=case["self_loop_scale"])
    my_seconds = time.time() - t0
    sh(f"fstequivalent --random=true --delta=0.003 {td}/graph_mine/HCLG.fst {graph}/HCLG.fst")       # the reference's own equivalence test
    sizes = {k: (ref / k).stat().st_size for k in ["LG.fst", "CLG.fst", "Ha.fst", "HCLGa.fst", "HCLG.fst"]}
    (OUT / name / "decode.json").write_text(json.dumps({
        "case": case, "mkgraph_log_tail": log[-300:], "decodes": dec, "reference_fst_bytes": sizes,
        "timing_note": "wall seconds in the build container (8 cores): the reference's mkgraph.sh process chain vs rs_mkgraph, same inputs",
        "reference_mkgraph_sh_seconds": round(ref_seconds, 2), "rs_mkgraph_seconds": round(my_seconds, 2)}, indent=1))
    print(name, sizes, f"ref {ref_seconds:.2f}s mine {my_seconds:.2f}s", [d["nbest_text"].splitlines()[0] for d in dec])
    shutil.rmtree(td)


def main():
    only = sys.argv[1:]
    for name, case in mc.CASES.items():
        if only and name not in only:
            continue
        if (OUT / name).exists():
            shutil.rmtree(OUT / name)
        gen_case(name, case)


if __name__ == "__main__":
    main()
