"""Decoding-graph construction on the host, step by step against the REFERENCE's tools.

tests/golden/mkgraph/<case>/ref/ holds what the reference's utils/mkgraph.sh and its individual commands produced
(oracle/gen_mkgraph_golden.py).  Every step of the library's chain is run on the reference's INPUT of that step and compared
with the reference's OUTPUT of it: as the same transducer up to state numbering where the step is canonical or order-preserving
(weights within 1e-4; the chain quantises to 1/1024 anyway), and as the same weighted relation (random paths, the
fstequivalent --random test) for the composition, whose state set is an implementation detail.  Then the whole chain from the
language directory and the model's tree, against the reference's finished HCLG.
"""
import pytest

from rhasspy_speech_amd import _lib

from . import mkgraph_cases as mc

TOL = 1e-4


@pytest.fixture(scope="module")
def models(tmp_path_factory):
    out = {}
    for name in mc.SMALL:
        case = mc.CASES[name]
        out[name] = mc.build_model_dir(case, tmp_path_factory.mktemp(name)) / "model" / "model"
    return out


@pytest.mark.parametrize("name", mc.STEPWISE)
def test_each_step_matches_the_reference_tool(name, models, tmp_path):
    case = mc.CASES[name]
    g = mc.GOLDEN / name
    ref, lang, mdl = g / "ref", g / "lang", models[name]
    t = _lib.fst_tool
    o = tmp_path
    # L o G
    t("fsttablecompose", lang / "L_disambig.fst", lang / "G.fst", o / "LG_composed.fst")
    t("fstequivalent", o / "LG_composed.fst", ref / "LG_composed.fst", param=TOL)
    # determinise (log semiring), minimise, push
    t("fstdeterminizestar", ref / "LG_composed.fst", out=o / "LG_det.fst", param=1.0)
    t("fstisomorphic", o / "LG_det.fst", ref / "LG_det.fst", param=TOL)
    t("fstminimizeencoded", ref / "LG_det.fst", out=o / "LG_min.fst")
    t("fstisomorphic", o / "LG_min.fst", ref / "LG_min.fst", param=TOL)
    t("fstpushspecial", ref / "LG_min.fst", out=o / "LG.fst")
    t("fstisomorphic", o / "LG.fst", ref / "LG.fst", param=TOL)
    # context
    from rhasspy_speech_amd import synth
    n_ctx, p_ctx = synth.context_shape(mc.case_spec(case))
    t("fstcomposecontext", ref / "LG.fst", lang / "phones" / "disambig.int", o / "CLG_unsorted.fst", aux=o / "ilabels", param=16 * n_ctx + p_ctx)
    t("fstarcsort", o / "CLG_unsorted.fst", out=o / "CLG.fst", aux="ilabel")
    assert (o / "ilabels").read_bytes() == (ref / "ilabels").read_bytes()
    t("fstisomorphic", o / "CLG.fst", ref / "CLG.fst", param=TOL)
    # H
    t("make-h-transducer", ref / "ilabels", mdl / "tree", o / "Ha.fst", aux=mdl / "final.mdl", param=1.0)
    assert (o / "Ha.fst.disambig").read_text() == (ref / "disambig_tid.int").read_text()
    t("fstisomorphic", o / "Ha.fst", ref / "Ha.fst", param=TOL)
    # H o CLG, determinise
    t("fsttablecompose", ref / "Ha.fst", ref / "CLG.fst", o / "HCLG_composed.fst")
    t("fstdeterminizestar", o / "HCLG_composed.fst", out=o / "HCLGa_det.fst", param=1.0)
    t("fstisomorphic", o / "HCLGa_det.fst", ref / "HCLGa_det.fst", param=TOL)
    # remove the disambiguation symbols, local epsilon removal, minimise
    t("fstrmsymbols", ref / "HCLGa_det.fst", out=o / "a.fst", aux=ref / "disambig_tid.int")
    t("fstrmepslocal", o / "a.fst", out=o / "b.fst")
    t("fstminimizeencoded", o / "b.fst", out=o / "HCLGa.fst")
    t("fstisomorphic", o / "HCLGa.fst", ref / "HCLGa.fst", param=TOL)
    # self loops
    t("add-self-loops", ref / "HCLGa.fst", out=o / "HCLG.fst", aux=mdl / "final.mdl", param=case["self_loop_scale"])
    t("fstisomorphic", o / "HCLG.fst", ref / "HCLG.fst", param=TOL)


@pytest.mark.parametrize("name", mc.SMALL)
def test_whole_chain_gives_the_reference_graph(name, models, tmp_path):
    case = mc.CASES[name]
    g = mc.GOLDEN / name
    _lib.mkgraph(g / "lang", models[name], tmp_path / "graph", self_loop_scale=case["self_loop_scale"], dump_dir=tmp_path / "dump")
    # same weighted relation as mkgraph.sh's HCLG.fst (2/1024: both chains quantise their weights to 1/1024 twice) ...
    _lib.fst_tool("fstequivalent", tmp_path / "graph" / "HCLG.fst", g / "ref" / "HCLG.fst", param=2.5 / 1024)
    if not case.get("light"):
        _lib.fst_tool("fstequivalent", tmp_path / "dump" / "LG.fst", g / "ref" / "LG.fst", param=2.5 / 1024)
    if case.get("light"):          # seed-drawn shapes: equivalence is what is guaranteed
        return
    # ... and, on these cases, even the same transducer up to state numbering (CLG / Ha differ in the numbering of the
    # phone-in-context labels, which is internal to the chain)
    for mine, theirs in [("dump/LG.fst", "LG.fst"), ("dump/HCLGa.fst", "HCLGa.fst"), ("graph/HCLG.fst", "HCLG.fst")]:
        _lib.fst_tool("fstisomorphic", tmp_path / mine, g / "ref" / theirs, param=1.5 / 1024)
    assert (tmp_path / "graph" / "words.txt").read_bytes() == (g / "lang" / "words.txt").read_bytes()
    assert (tmp_path / "graph" / "disambig_tid.int").read_text() == (g / "ref" / "disambig_tid.int").read_text()
    assert int((tmp_path / "graph" / "num_pdfs").read_text()) == mc.case_spec(case).num_pdfs


def test_mkgraph_reports_missing_inputs_like_the_script(tmp_path, models):
    with pytest.raises(_lib.RsError, match="expected .*L_disambig.fst to exist"):
        _lib.mkgraph(tmp_path / "nolang", models["mono_grammar"], tmp_path / "graph")


def test_trainer_mkgraph_mirror(tmp_path, models):
    """kaldi.py:409-425: KaldiTrainer._mkgraph builds <train_dir>/graph_<suffix> from <train_dir>/data/lang_<suffix> with
    --self-loop-scale 1.0, and only warns when the language directory is missing."""
    import asyncio
    import shutil
    from rhasspy_speech_amd.kaldi import KaldiTrainer
    name = "bi_backoff"
    model_dir = models[name].parent           # <model_dir>/model holds tree + final.mdl
    tr = KaldiTrainer(tmp_path / "train", model_dir)
    asyncio.run(tr._mkgraph("arpa"))           # no lang dir: a warning, nothing built
    assert not tr.graph_dir("arpa").exists()
    shutil.copytree(mc.GOLDEN / name / "lang", tr.lang_dir("grammar"))
    asyncio.run(tr._mkgraph("grammar"))
    _lib.fst_tool("fstisomorphic", tr.graph_dir("grammar") / "HCLG.fst", mc.GOLDEN / name / "ref" / "HCLG.fst", param=1.5 / 1024)


def test_mkgraph_sh_drop_in(tmp_path, models):
    """rhasspy_speech_amd/bin/utils/mkgraph.sh: the script's name and argv over rs_mkgraph, run the way the reference's trainer runs
    it (`bash <utils>/mkgraph.sh --self-loop-scale 1.0 <lang> <model> <graph>`, kaldi.py:415-424): same graph, the script's
    "up to date" shortcut and its usage error."""
    import subprocess
    from pathlib import Path
    sh = Path(_lib.__file__).resolve().parent / "bin" / "utils" / "mkgraph.sh"
    name = "tri_backoff"
    argv = ["bash", str(sh), "--self-loop-scale", "1.0", str(mc.GOLDEN / name / "lang"), str(models[name]), str(tmp_path / "graph")]
    p = subprocess.run(argv, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    _lib.fst_tool("fstisomorphic", tmp_path / "graph" / "HCLG.fst", mc.GOLDEN / name / "ref" / "HCLG.fst", param=1.5 / 1024)
    p = subprocess.run(argv, capture_output=True, text=True)
    assert p.returncode == 0 and "is up to date" in p.stdout
    p = subprocess.run(["bash", str(sh), "only-one-argument"], capture_output=True, text=True)
    assert p.returncode == 1 and p.stdout.startswith("Usage: utils/mkgraph.sh")
    p = subprocess.run(["bash", str(sh), str(tmp_path / "nolang"), str(models[name]), str(tmp_path / "g2")], capture_output=True, text=True)
    assert p.returncode == 1 and "expected" in p.stderr and "to exist" in p.stderr
