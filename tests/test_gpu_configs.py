"""BASELINE.json configs[2..4] at their full sizes on the HIP path (`-m gpu`).

configs[1] is the bench line (bench.py); the other GPU configurations are parity cases:
  [2] ARPA-LM HCLG (larger FST), 256 x 3 s batch;
  [3] mixed-model batch (two different models decoded side by side, utterance-sharded);
  [4] streaming decode, 64 concurrent 30 s streams.
No golden from the reference exists at these sizes (the reference takes minutes per case), so they are checked through
size-independent properties -- every utterance of a big batch decodes exactly as it does alone, every decoder variant
agrees -- plus the CPU oracle (pinned to the reference by tests/test_oracle_golden.py) on a few sampled utterances.
"""
from __future__ import annotations

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOGLIKE_TOL = 1e-4


@pytest.fixture(scope="module")
def zam_arpa(tmp_path_factory):
    """zamia-like-S acoustic model + a back-off ARPA HCLG (a few thousand states: beyond the register-resident decoder)."""
    from rhasspy_speech_amd import synth
    root = tmp_path_factory.mktemp("zam_arpa")
    spec = synth.ModelSpec()
    synth.write_model_dir(root / "model", spec)
    synth.make_arpa_graph(root / "graph", spec, extra_words=600, num_random_sentences=4000)
    return root / "model", root / "graph"


@pytest.fixture(scope="module")
def zam_grammar(tmp_path_factory):
    from rhasspy_speech_amd import synth
    root = tmp_path_factory.mktemp("zam_grammar")
    spec = synth.ModelSpec()
    synth.write_model_dir(root / "model", spec)
    synth.make_grammar_graph(root / "graph", spec)
    return root / "model", root / "graph"


def _same_result(a, i, b, j):
    assert a.num_hyps(i) == b.num_hyps(j)
    for k in range(a.num_hyps(i)):
        assert a.words(i, k) == b.words(j, k)
    np.testing.assert_array_equal(a.costs(i), b.costs(j))


def test_config2_arpa_hclg_256x3s(zam_arpa):
    from rhasspy_speech_amd import _lib, synth
    from oracle import pipeline
    model_dir, graph_dir = zam_arpa
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts())
    desc = model.describe()
    n_states = int(desc.split("hclg: states=")[1].split()[0])
    assert n_states > 2000, desc           # a larger FST than the grammar graph (625 states)
    pcms = [synth.synth_utterance(7000 + u, 48000) for u in range(256)]
    batch = model.decode_batch(pcms)
    assert batch.num_utts == 256
    # every utterance of the batch decodes exactly as it does alone (sampled)
    for u in (0, 1, 77, 128, 255):
        one = model.decode_batch([pcms[u]])
        _same_result(batch, u, one, 0)
    # the CPU oracle on a few of them: transcripts exact
    orc = pipeline.Oracle(model_dir, graph_dir)
    for u in (3, 200):
        tr = orc.transcribe(pcms[u])
        assert batch.words(u) == tr.nbest[0].words
        np.testing.assert_allclose(batch.costs(u)[:2], [tr.nbest[0].graph_cost, tr.nbest[0].acoustic_cost], rtol=2e-4, atol=2e-3)
    # n-best through the lattice path on a slice of the batch
    nb = model.decode_batch(pcms[:16], nbest=3)
    for u in range(16):
        assert nb.words(u, 0) == batch.words(u)


def test_config3_mixed_models_side_by_side(zam_grammar, case_cache):
    """Two different models resident on the GPU, their batches decoded concurrently from two host threads (what a
    rank serving a mixed-model shard does).  Each result must equal the model's own sequential result."""
    from rhasspy_speech_amd import _lib, synth
    from tests import cases
    m1 = _lib.Model(*zam_grammar, _lib.default_opts())
    md2, gd2, _, _ = case_cache("tinyf_u5")
    m2 = _lib.Model(md2, gd2, _lib.default_opts(**cases.CASES["tinyf_u5"].get("opts", {})))
    pcm1 = [synth.synth_utterance(9000 + u, 48000 - 700 * (u % 5)) for u in range(96)]
    pcm2 = [synth.synth_utterance(9500 + u, 30000 + 900 * (u % 7)) for u in range(160)]
    ref1, ref2 = m1.decode_batch(pcm1), m2.decode_batch(pcm2)
    out = {}

    def run(key, model, pcms):
        out[key] = [model.decode_batch(pcms) for _ in range(3)]

    ts = [threading.Thread(target=run, args=("a", m1, pcm1)), threading.Thread(target=run, args=("b", m2, pcm2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for res in out["a"]:
        for u in range(len(pcm1)):
            _same_result(res, u, ref1, u)
    for res in out["b"]:
        for u in range(len(pcm2)):
            _same_result(res, u, ref2, u)
    # utterance-sharded gather of the mixed batch (shard.py is what bench.py / the multi-GPU path use)
    from rhasspy_speech_amd import shard
    ids = list(range(len(pcm1) + len(pcm2)))
    mine = [shard.shard_indices(len(ids), r, 8) for r in range(8)]
    assert sorted(i for part in mine for i in part) == ids
    # the mixed batch through the host entry point (shard.decode_mixed_sharded; one rank here): utterances of the two
    # models interleaved in one list, each decoded by its own model, results back in the caller's order
    names, pcm, src = [], [], []
    for u in range(max(len(pcm1), len(pcm2))):
        if u < len(pcm1):
            names.append("zam"); pcm.append(pcm1[u]); src.append((ref1, u))
        if u < len(pcm2):
            names.append("tinyf"); pcm.append(pcm2[u]); src.append((ref2, u))
    got = shard.decode_mixed_sharded({"zam": m1, "tinyf": m2}, names, pcm)
    assert sorted(got) == list(range(len(pcm)))
    for i, (ref, u) in enumerate(src):
        g, a = ref.costs(u)
        assert got[i][0] == ref.words(u)[:shard.MAX_WORDS] and got[i][1:] == (np.float32(g), np.float32(a)), i


def test_config4_64_streams_30s(zam_grammar):
    from rhasspy_speech_amd import _lib, synth
    from oracle import pipeline
    model_dir, graph_dir = zam_grammar
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts(keep_intermediates=1))
    n_streams, n_samples = 64, 30 * 16000
    rng = np.random.default_rng(4)
    pcms = [synth.synth_utterance(12000 + i, n_samples - 160 * int(rng.integers(0, 50))) for i in range(n_streams)]
    streams = [_lib.Stream(model) for _ in pcms]
    # ragged, interleaved delivery (2048-byte reads like transcribe_stream.py, but of varying size per stream)
    pos = [0] * n_streams
    while any(p < len(x) for p, x in zip(pos, pcms)):
        for i, s in enumerate(streams):
            if pos[i] < len(pcms[i]):
                n = int(rng.integers(512, 40000))
                s.accept(pcms[i][pos[i]:pos[i] + n].tobytes())
                pos[i] += n
    batch = _lib.finish_streams(streams)
    assert batch.num_utts == n_streams
    for i in (0, 31, 63):          # a stream alone gives the same result, bit for bit
        s = _lib.Stream(model)
        s.accept(pcms[i])
        one = s.finish()
        _same_result(batch, i, one, 0)
        np.testing.assert_array_equal(batch.matrix(i, 2), one.matrix(0, 2))
    # CPU oracle (streaming semantics of online2-cli-nnet3-decode-faster) on one stream
    orc = pipeline.Oracle(model_dir, graph_dir)
    tr = orc.transcribe_stream(pcms[5])
    assert batch.words(5) == tr.nbest[0].words
    assert np.abs(batch.matrix(5, 2) - tr.loglikes).max() < LOGLIKE_TOL


def test_split_bf16_gemm_matches_fp32_gemm(zam_grammar, monkeypatch):
    """The wide layers run on the bf16 matrix cores with every FP32 operand split into three bf16 parts (nnet_gemm_b3.hip).
    Same batch through that kernel and through the exact-FP32 MFMA kernel (RS_GEMM_B3=0): transcripts identical,
    log-likelihoods within half the 1e-4 bound (both are checked against the reference's values in test_gpu_parity)."""
    from rhasspy_speech_amd import _lib, synth
    model_dir, graph_dir = zam_grammar
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts(keep_intermediates=1))
    pcms = [synth.synth_utterance(15000 + u, 48000 - 640 * (u % 7)) for u in range(96)]
    monkeypatch.setenv("RS_GEMM_B3", "1")
    split = model.decode_batch(pcms)
    monkeypatch.setenv("RS_GEMM_B3", "0")
    exact = model.decode_batch(pcms)
    worst = 0.0
    for u in range(len(pcms)):
        assert split.words(u) == exact.words(u)
        a, b = split.matrix(u, 2), exact.matrix(u, 2)
        worst = max(worst, float(np.abs(a - b).max()))
        assert np.abs(split.costs(u)[1] - exact.costs(u)[1]) < 2e-3 * max(1.0, abs(exact.costs(u)[1]))
    assert 0.0 < worst < 5e-5, worst          # > 0: the two kernels really are different code paths


def test_pruned_output_layer_on_a_batch(zam_grammar):
    """prune_output_pdfs on the headline model / graph (362 of the 2000 pdfs are on HCLG arcs): a ragged batch decodes to
    the same words and costs as with the full output layer."""
    from rhasspy_speech_amd import _lib, synth
    model_dir, graph_dir = zam_grammar
    full = _lib.Model(model_dir, graph_dir, _lib.default_opts())
    pruned = _lib.Model(model_dir, graph_dir, _lib.default_opts(prune_output_pdfs=1))
    assert "pruned to the" in pruned.describe()
    pcms = [synth.synth_utterance(17000 + u, 48000 - 480 * (u % 9)) for u in range(80)]
    a, b = full.decode_batch(pcms), pruned.decode_batch(pcms)
    for u in range(len(pcms)):
        assert a.words(u) == b.words(u)
        np.testing.assert_allclose(b.costs(u), a.costs(u), rtol=2e-4, atol=2e-3)


def test_concurrent_calls_on_one_model(zam_grammar):
    """Several host threads decoding different batches on ONE model at the same time (each call gets its own decode
    context: streams, arenas, staging): every result equals the sequential one."""
    from rhasspy_speech_amd import _lib, synth
    model = _lib.Model(*zam_grammar, _lib.default_opts())
    batches = [[synth.synth_utterance(21000 + 100 * b + u, 48000 - 320 * ((u + b) % 11)) for u in range(40 + 24 * b)] for b in range(4)]
    ref = [model.decode_batch(pcms) for pcms in batches]
    out = [[] for _ in batches]

    def run(b):
        for _ in range(4):
            out[b].append(model.decode_batch(batches[b]))

    ts = [threading.Thread(target=run, args=(b,)) for b in range(len(batches))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for b, pcms in enumerate(batches):
        assert len(out[b]) == 4
        for res in out[b]:
            for u in range(len(pcms)):
                _same_result(res, u, ref[b], u)


def test_overlapped_contexts_use_the_cu_exclusive_gemm(zam_grammar, monkeypatch):
    """RS_GEMM_B3_EXCLUSIVE=1: the wide layers run as CU-exclusive 512-thread workgroups (nnet_gemm_b3.hip, WM = 2) when a model
    has several decode contexts.  Same results as the default model, sequentially and from four threads at once."""
    from rhasspy_speech_amd import _lib, synth
    default = _lib.Model(*zam_grammar, _lib.default_opts())
    batches = [[synth.synth_utterance(23000 + 100 * b + u, 48000 - 320 * ((u + b) % 11)) for u in range(24 + 16 * b)] for b in range(4)]
    ref = [default.decode_batch(p) for p in batches]
    monkeypatch.setenv("RS_GEMM_B3_EXCLUSIVE", "1")          # read at every launch
    overlapped = _lib.Model(*zam_grammar, _lib.default_opts())
    for b, p in enumerate(batches):
        one = overlapped.decode_batch(p)
        for u in range(len(p)):
            _same_result(one, u, ref[b], u)
    out = [[] for _ in batches]

    def run(b):
        for _ in range(3):
            out[b].append(overlapped.decode_batch(batches[b]))

    ts = [threading.Thread(target=run, args=(b,)) for b in range(len(batches))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for b, p in enumerate(batches):
        for res in out[b]:
            for u in range(len(p)):
                _same_result(res, u, ref[b], u)
