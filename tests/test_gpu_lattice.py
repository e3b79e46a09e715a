"""The lattice wire format (`-m gpu`): rs_result_lattice renders the utterance's determinised lattice as the binary
CompactLattice table entry the reference's online2-wav-nnet3-latgen-faster writes.  Checked two ways:

 * structurally, by a reader of the format written for the tests (tests/lattice_io.py: header, arc type, deterministic on word labels, acyclic);
 * by the REFERENCE's own tools (oracle/_ref, test infrastructure): the bytes go through `lattice-to-nbest |
   nbest-to-linear` and every path -- words, transition-id alignment, graph and acoustic cost -- is compared with the paths
   of the reference's lattice for the same case (tests/golden/lattice/*.json, oracle/gen_lattice_golden.py).
"""
import json
from pathlib import Path

import pytest

from tests import cases
from tests.lattice_io import read_compact_lattice, reference_paths

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parent.parent
LATTICE_GOLDEN = cases.GOLDEN / "lattice"
LATTICE_CASES = sorted(p.stem for p in LATTICE_GOLDEN.glob("*.json"))
BEAM_EDGE = 0.05        # paths this close to the lattice beam may fall on either side (log-likelihoods differ by ~1e-5)


@pytest.mark.parametrize("name", LATTICE_CASES)
def test_lattice_entry_matches_reference_lattice(case_cache, tmp_path, name):
    from rhasspy_speech_amd import _lib
    golden = json.loads((LATTICE_GOLDEN / f"{name}.json").read_text())
    model_dir, graph_dir, _, pcm = case_cache(name)
    o = dict(emit_lattice=1)
    o.update(cases.CASES[name].get("opts", {}))
    opts = _lib.default_opts(**o)
    model = _lib.Model(model_dir, graph_dir, opts)
    res = model.decode_batch([pcm], nbest=cases.NBEST)
    entry = res.lattice(0, "utt")
    # ---- structure
    start, finals, arcs = read_compact_lattice(entry, "utt")
    assert start >= 0 and finals
    for row in arcs:
        labels = [a[0] for a in row if a[0] != 0]
        assert len(labels) == len(set(labels)), "not deterministic on word labels"
    # ---- the reference's tools read it and list the same paths as for the reference's own lattice
    ark = tmp_path / "lat.ark"
    ark.write_bytes(entry)
    mine = reference_paths(ark, golden["n_requested"], tmp_path)
    ref = golden["paths"]
    best = ref[0]["graph"] + ref[0]["acoustic"]
    beam = float(opts.lattice_beam)
    by_words = {tuple(p["words"]): p for p in mine}
    assert len(by_words) == len(mine), "a word sequence appears twice: the lattice is not determinised"
    for p in ref:
        total = p["graph"] + p["acoustic"]
        q = by_words.get(tuple(p["words"]))
        if q is None:
            assert total > best + beam - BEAM_EDGE, f"path {p['words']} (total {total}) of the reference lattice is missing"
            continue
        assert q["ali"] == p["ali"], p["words"]
        assert abs(q["graph"] - p["graph"]) < 2e-3 * max(1.0, abs(p["graph"]))
        assert abs(q["acoustic"] - p["acoustic"]) < 2e-3 * max(1.0, abs(p["acoustic"]))
    if len(ref) < golden["n_requested"]:               # the golden lists the whole lattice: nothing extra either
        ref_words = {tuple(p["words"]) for p in ref}
        for q in mine:
            if tuple(q["words"]) not in ref_words:
                assert q["graph"] + q["acoustic"] > best + beam - BEAM_EDGE, f"extra path {q['words']}"
    # the n-best the library computes itself is the head of the same list
    assert [res.words(0, k) for k in range(res.num_hyps(0))] == [p["words"] for p in mine[:res.num_hyps(0)]]


def test_lattice_needs_the_option(case_cache):
    from rhasspy_speech_amd import _lib
    model_dir, graph_dir, _, pcm = case_cache("tiny_u0")
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts())
    res = model.decode_batch([pcm], nbest=2)
    with pytest.raises(_lib.RsError) as ei:
        res.lattice(0)
    assert "emit_lattice" in str(ei.value)


def test_lattice_batch_and_scale(case_cache, tmp_path):
    """A ragged batch gives one entry per utterance (concatenated = a valid table); the decodable's acoustic scale is
    undone on the way out like the reference does (online2-wav-nnet3-latgen-faster.cc:290-293)."""
    from rhasspy_speech_amd import _lib, synth
    model_dir, graph_dir, _, _ = case_cache("tiny_arpa_u7")
    pcms = [synth.synth_utterance(700 + i, n) for i, n in enumerate([48000, 20000, 33000])]
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts(emit_lattice=1))
    res = model.decode_batch(pcms, nbest=3)
    ark = tmp_path / "all.ark"
    ark.write_bytes(b"".join(res.lattice(u, f"utt{u}") for u in range(3)))
    paths = reference_paths(ark, 3, tmp_path)
    keys_seen = [p["words"] for p in paths]
    expect = [res.words(u, k) for u in range(3) for k in range(res.num_hyps(u))]
    assert keys_seen == expect
    half = _lib.Model(model_dir, graph_dir, _lib.default_opts(emit_lattice=1, acoustic_scale=0.5, beam=12.0, lattice_beam=4.0))
    r2 = half.decode_batch(pcms[:1], nbest=1)
    _, finals, arcs = read_compact_lattice(r2.lattice(0, "k"), "k")
    ark.write_bytes(r2.lattice(0, "k"))
    p = reference_paths(ark, 1, tmp_path)[0]
    gc, ac = r2.costs(0, 0)
    assert p["words"] == r2.words(0, 0)
    assert abs(p["graph"] - gc) < 2e-3 * max(1.0, abs(gc)) and abs(p["acoustic"] - ac) < 2e-3 * max(1.0, abs(ac))
    # un-scaled: the same audio decoded at scale 1 has about the same acoustic cost on that word sequence
    r1 = model.decode_batch(pcms[:1], nbest=1)
    if r1.words(0, 0) == r2.words(0, 0):
        assert abs(r1.costs(0, 0)[1] - ac) < 0.05 * abs(ac)
