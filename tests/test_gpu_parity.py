"""GPU parity tests: the HIP path (through the C ABI) against the reference's golden vectors.

Tolerances: transcripts (word-id sequences) exact; acoustic log-likelihoods 1e-4 absolute (north-star); MFCC
features 5e-3 absolute worst case / 1e-3 at the 99th percentile on values up to ~1e2 (a different float32 FFT
factorisation; the log of near-empty mel bins amplifies FFT round-off); iVectors 1e-4.
"""
import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

LOGLIKE_TOL = 1e-4
# MFCC: the split-radix FFT restates the reference's float operations one for one and the mel filter edges come from the
# same libm logf; what is left is the summation order of the mel / DCT dot products times the cepstral lifter (<= 12) and
# libm-vs-device logf: observed max 8e-5 on c0 ~ 100
FEAT_TOL = 2e-4
FEAT_TOL_P99 = 1e-4
IVEC_TOL = 1e-4


def load_golden(name):
    return np.load(cases.GOLDEN / f"{name}.npz")


def parse_nbest(text: bytes):
    out = []
    for line in text.decode().splitlines():
        p = line.split()
        if p:
            out.append([int(x) for x in p[1:]])
    return out


def make_model(case_cache, name, **extra):
    from rhasspy_speech_amd import _lib
    model_dir, graph_dir, wav, pcm = case_cache(name)
    o = dict(keep_intermediates=1)
    o.update(cases.CASES[name].get("opts", {}))
    o.update(extra)
    return _lib.Model(model_dir, graph_dir, _lib.default_opts(**o)), pcm


@pytest.mark.parametrize("name", list(cases.CASES))
def test_offline_case(case_cache, name):
    g = load_golden(name)
    model, pcm = make_model(case_cache, name)
    res = model.decode_batch([pcm], nbest=1)
    assert res.num_frames(0) == int(g["offline_num_frames"])
    feats = res.matrix(0, 0)
    assert feats.shape == g["input"].shape
    fd = np.abs(feats - g["input"])
    assert fd.max() < FEAT_TOL and np.quantile(fd, 0.99) < FEAT_TOL_P99, (fd.max(), np.quantile(fd, 0.99))
    if "offline_ivector" in g:
        iv = res.matrix(0, 1)
        assert np.abs(iv - g["offline_ivector"]).max() < IVEC_TOL
    ll = res.matrix(0, 2)
    sr, sc = g["loglikes_stride"]
    diff = np.abs(ll[::sr, ::sc] - g["offline_loglikes"]).max()
    assert diff < LOGLIKE_TOL, f"log-likelihood max abs diff {diff}"
    ref = parse_nbest(bytes(g["offline_nbest_text"]))
    assert res.words(0, 0) == ref[0]
    gc, ac = res.costs(0, 0)
    assert abs(gc - g["offline_graph_cost"][0]) < 2e-3 * max(1.0, abs(gc))
    assert abs(ac - g["offline_acoustic_cost"][0]) < 2e-3 * max(1.0, abs(ac))
    assert res.text(0).split(b"\n")[0].split() == bytes(g["offline_nbest_text"]).split(b"\n")[0].split()


@pytest.mark.parametrize("name", ["tiny_u0", "tiny_fsf3_u16", "tiny_fsf2_noiv_u18", "zam_fsf3_u19"])
def test_batch_matches_single(case_cache, name):
    """Ragged batch: every utterance decodes exactly as it does alone (no cross-utterance leakage) -- also with the upper layers
    evaluated on every second / third row (--frame-subsampling-factor: the row lists then skip rows inside every utterance)."""
    from rhasspy_speech_amd import synth
    model, _ = make_model(case_cache, name)
    pcms = [synth.synth_utterance(20 + i, n) for i, n in enumerate([48000, 16000, 30000, 8000, 48000, 400, 24001])]
    batch = model.decode_batch(pcms)
    for i, p in enumerate(pcms):
        single = model.decode_batch([p])
        assert batch.words(i) == single.words(0)
        assert np.array_equal(batch.matrix(i, 2), single.matrix(0, 2))
        assert batch.costs(i) == single.costs(0)


@pytest.mark.parametrize("fsf", [2, 3, 5])
def test_subsampled_log_likelihoods_are_every_nth_row_of_the_dense_ones(case_cache, fsf):
    """--frame-subsampling-factor n: the decoder's frame f is output row n f, bit for bit the row the dense evaluation gives it (the
    layers that are evaluated on every n-th row only run the same kernels through a row list); a model without dither, whose noise
    would otherwise follow the factor through the reference's rand() count."""
    from rhasspy_speech_amd import synth
    pcms = [synth.synth_utterance(70 + i, n) for i, n in enumerate([36000, 16001, 5000, 700])]
    dense = make_model(case_cache, "tiny_nodither_u11", frame_subsampling_factor=1)[0].decode_batch(pcms)
    sub = make_model(case_cache, "tiny_nodither_u11", frame_subsampling_factor=fsf)[0].decode_batch(pcms)
    for u in range(len(pcms)):
        np.testing.assert_array_equal(sub.matrix(u, 0), dense.matrix(u, 0))
        np.testing.assert_array_equal(sub.matrix(u, 1), dense.matrix(u, 1))
        np.testing.assert_array_equal(sub.matrix(u, 2), dense.matrix(u, 2)[::fsf])
        assert sub.num_frames(u) == (dense.num_frames(u) + fsf - 1) // fsf


def test_too_short_utterance_fails_like_reference(case_cache):
    from rhasspy_speech_amd import _lib
    model, _ = make_model(case_cache, "tiny_u0")
    res = model.decode_batch([np.zeros(300, np.int16), np.zeros(0, np.int16)])
    for u in range(2):
        with pytest.raises(_lib.RsError) as ei:
            res.words(u)
        assert "decoded no frames" in str(ei.value)


@pytest.mark.parametrize("name", [n for n in cases.CASES])
def test_offline_nbest(case_cache, name):
    """n-best lists (lattice-to-nbest --n=5 | nbest-to-linear) with their graph / acoustic costs."""
    g = load_golden(name)
    model, pcm = make_model(case_cache, name, keep_intermediates=0)
    res = model.decode_batch([pcm], nbest=cases.NBEST)
    ref = parse_nbest(bytes(g["offline_nbest_text"]))
    got = [res.words(0, k) for k in range(res.num_hyps(0))]
    assert got == ref, (got, ref)
    gc = np.array([res.costs(0, k)[0] for k in range(len(got))])
    ac = np.array([res.costs(0, k)[1] for k in range(len(got))])
    np.testing.assert_allclose(gc, g["offline_graph_cost"], rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose(ac, g["offline_acoustic_cost"], rtol=2e-4, atol=2e-3)
    assert res.text(0).split() == bytes(g["offline_nbest_text"]).split()
    # the 1-best through the lattice equals the device traceback
    one = model.decode_batch([pcm], nbest=1)
    assert one.words(0) == got[0]


def test_gpu_matches_oracle_on_fresh_inputs(case_cache):
    """Seeded inputs that have no golden: HIP path vs the CPU oracle (oracle pinned by tests/test_oracle_golden.py)."""
    from oracle import pipeline
    from rhasspy_speech_amd import synth
    model_dir, graph_dir, _, _ = case_cache("tiny_u0")
    model, _ = make_model(case_cache, "tiny_u0")
    orc = pipeline.Oracle(model_dir, graph_dir)
    pcms = [synth.synth_utterance(100 + i, n) for i, n in enumerate([48000, 21000, 35000, 12345])]
    res = model.decode_batch(pcms, nbest=3)
    for i, p in enumerate(pcms):
        tr = orc.transcribe(p, nbest=3)
        assert np.abs(res.matrix(i, 2) - tr.loglikes).max() < LOGLIKE_TOL
        assert [res.words(i, k) for k in range(res.num_hyps(i))] == [q.words for q in tr.nbest]
        # decoder work counters: same number of frames' worth of live tokens as the sequential reference algorithm
        assert res.counters(i)[3] > 0


VARIANT_CASES = [("tiny_u0", {}), ("tiny_arpa_u7", {}), ("tiny_arpa_prune_u8", {}), ("tiny_noiv_u2", {}), ("zam_u0", {}),
                 # pruning paths of GetCutoff: max-active and min-active selection on most frames
                 ("zam_u0", dict(max_active=150, min_active=100, beam=10.0)),
                 ("zam_u0", dict(max_active=400, min_active=300, beam=6.0)),
                 ("zam_u1", dict(max_active=40, min_active=0, beam=16.0))]


@pytest.mark.parametrize("name,extra", VARIANT_CASES)
def test_decoder_variants_agree(case_cache, name, extra, monkeypatch):
    """The register-resident, the LDS-resident (pull) and the two general token-list decoders (dense per-state tables in HBM;
    live-state table in LDS) are the same search."""
    from rhasspy_speech_amd import synth
    models = {}
    for variant in ("sparse", "dense", "reg", "hash"):
        monkeypatch.setenv("RS_DECODER", variant)
        models[variant], pcm = make_model(case_cache, name, **extra)
    monkeypatch.delenv("RS_DECODER")
    pcms = [pcm] + [synth.synth_utterance(300 + i, n) for i, n in enumerate([48000, 17000, 33000])]
    ref = models["sparse"].decode_batch(pcms)
    for variant in ("dense", "reg", "hash"):
        got = models[variant].decode_batch(pcms)
        for u in range(len(pcms)):
            assert got.words(u) == ref.words(u), variant
            np.testing.assert_allclose(got.costs(u), ref.costs(u), rtol=1e-6)
            # same number of live tokens summed over frames => same token sets
            assert got.counters(u)[3] == ref.counters(u)[3], variant
            # the pruning branches were actually taken the same number of times
            assert got.counters(u)[5] == ref.counters(u)[5] and got.counters(u)[6] == ref.counters(u)[6], variant


@pytest.mark.parametrize("name,extra", [("tiny_u0", {}), ("tiny_hmm_u6", {}), ("zam_u0", {}), ("zam_u0", dict(max_active=150, min_active=100, beam=10.0)),
                                        ("zam_u0", dict(max_active=400, min_active=300, beam=6.0)), ("zam_u1", dict(max_active=40, min_active=0, beam=16.0)),
                                        ("zam_long30", dict(min_active=350))])
def test_exact_token_order_is_the_sequential_decoder(case_cache, name, extra, monkeypatch):
    """rs_decode_opts.exact_token_order: the grammar-graph search creates tokens against the reference's RUNNING cutoff in its
    HashList order (lattice-faster-decoder.cc:734-787, hash-list-inl.h:125-165).  Checked against oracle/decoder.c -- the
    sequential restatement that follows that order, pinned to the reference's costs by tests/test_oracle_golden.py -- on the SAME
    log-likelihoods (the GPU's own): the decisions are then the same float comparisons, so the best path's costs must agree to
    the last bits of their float sums, the token counts of every frame add up to the same total, and min-active / max-active
    bound on the same number of frames.  With pruning options that bind on most frames; batch, stream and time-slab paths."""
    from oracle import kaldi_formats as kf
    from oracle import pipeline
    from rhasspy_speech_amd import _lib, synth
    model_dir, graph_dir, _, pcm = case_cache(name)
    o = dict(cases.CASES[name].get("opts", {}))
    o.update(extra)
    model, _ = make_model(case_cache, name, exact_token_order=1, **extra)
    plain, _ = make_model(case_cache, name, **extra)
    pcms = [pcm] + [synth.synth_utterance(700 + i, n) for i, n in enumerate([48000, 17000])]
    res = model.decode_batch(pcms)
    assert "token_order: exact" in model.describe(), model.describe()
    base = plain.decode_batch(pcms)
    orc = pipeline.Oracle(model_dir, graph_dir, **o)
    for u in range(len(pcms)):
        lattice, ctr = pipeline.decode(orc.fst, orc.id2pdf, res.matrix(u, 2), **orc.opts)
        best = pipeline.lat.nbest(lattice, 1, orc.opts["lattice_beam"], 1.0)[0]
        assert res.words(u) == best.words
        np.testing.assert_allclose(res.costs(u), (best.graph_cost, best.acoustic_cost), rtol=2e-6, atol=1e-4)
        assert res.counters(u)[5] == ctr[5] and res.counters(u)[6] == ctr[6], (res.counters(u), ctr)
        assert res.counters(u)[3] == ctr[3] + ctr[7], (res.counters(u), ctr)  # tokens on the lists of all frames: the extras too
        assert res.words(u) == base.words(u)
    # the stream path parks and resumes the list order with the costs
    st = _lib.Stream(model)
    st.accept(pcm[: len(pcm) // 2]); st.advance(); st.accept(pcm[len(pcm) // 2:])
    got = st.finish()
    lattice, ctr = pipeline.decode(orc.fst, orc.id2pdf, got.matrix(0, 2), **orc.opts)
    best = pipeline.lat.nbest(lattice, 1, orc.opts["lattice_beam"], 1.0)[0]
    assert got.words(0) == best.words and got.counters(0)[3] == ctr[3] + ctr[7]
    np.testing.assert_allclose(got.costs(0), (best.graph_cost, best.acoustic_cost), rtol=2e-6, atol=1e-4)


def test_decoder_variants_agree_on_a_graph_of_a_few_thousand_states(tmp_path, monkeypatch):
    """A grammar graph of 2000-5000 states (a hundred-odd sentences): a back-pointer row no longer fits the 16-byte loads the
    traceback keeps in flight per thread, the register-resident search runs its wider shapes, the dense search one wave or four."""
    from rhasspy_speech_amd import _lib, synth
    spec = synth.ModelSpec(name="few-thousand-states", num_phones=120, hidden_dim=64, num_gauss=64, ivector_dim=30)
    rng = np.random.default_rng(5)
    vocab = sorted({w for s in synth.DEFAULT_SENTENCES for w in s.split()})
    sents = [" ".join(vocab[int(i)] for i in rng.integers(0, len(vocab), int(rng.integers(5, 9)))) for _ in range(140)]
    synth.write_model_dir(tmp_path / "m", spec)
    fst, _lex = synth.make_grammar_graph(tmp_path / "g", spec, sentences=sents)
    assert 2100 < fst.num_states <= 5000, fst.num_states
    pcms = [synth.synth_utterance(40 + i, n) for i, n in enumerate([48000, 9000, 30000])]
    res = {}
    for variant in ("sparse", "reg", "dense"):
        monkeypatch.setenv("RS_DECODER", variant)
        model = _lib.Model(tmp_path / "m", tmp_path / "g", _lib.default_opts(max_active=300, min_active=100))
        res[variant] = model.decode_batch(pcms)
        model.close()
    for variant in ("reg", "dense"):
        for u in range(len(pcms)):
            assert res[variant].words(u) == res["sparse"].words(u), (variant, u)
            np.testing.assert_allclose(res[variant].costs(u), res["sparse"].costs(u), rtol=1e-6)
            assert res[variant].counters(u)[3] == res["sparse"].counters(u)[3], (variant, u)
            assert res[variant].counters(u)[5:7] == res["sparse"].counters(u)[5:7], (variant, u)


@pytest.mark.parametrize("name,extra", [("tiny_u0", {}), ("zam_u0", {}), ("zam_u0", dict(max_active=150, min_active=100, beam=10.0)),
                                        ("zam_u1", dict(max_active=40, min_active=0, beam=16.0)), ("tiny_hmm_u6", {}), ("tinyf_u5", {}), ("tiny_vecfst_u9", {}),
                                        ("tiny_fsf3_u16", {})])
def test_lattice_from_the_register_resident_search_is_the_token_list_searchs(case_cache, name, extra, monkeypatch):
    """n-best / lattice calls on grammar graphs run the register-resident search, which leaves the costs of every (frame, state) pair
    beside its back-pointer rows, and the lattice pass on those rows (DenseLatticeKernel, round 5).  RS_LATTICE_KERNEL=tokens turns the
    rows into token lists and runs LatticeKernel on them (round 4's path); RS_LATTICE_SEARCH=tokens runs the token-list search in front
    of it, as every round before: the same lattice (arc count), the same n-best lists and costs, batch and stream."""
    from rhasspy_speech_amd import _lib, synth
    model, pcm = make_model(case_cache, name, **extra)
    pcms = [pcm] + [synth.synth_utterance(1300 + i, n) for i, n in enumerate([48000, 9000, 33000, 1700])]
    def run():
        b = model.decode_batch(pcms, nbest=5)
        st = _lib.Stream(model)
        raw = pcms[1].tobytes()
        for k in range(0, len(raw), 8192):
            st.accept(raw[k:k + 8192])
            st.advance()
        return b, st.finish(5, 1.0)
    got_b, got_s = run()
    monkeypatch.setenv("RS_LATTICE_KERNEL", "vote")      # (the dense kernel's closure pass until a round changes nothing, as for a cyclic epsilon subgraph)
    vote_b, vote_s = run()
    monkeypatch.setenv("RS_LATTICE_KERNEL", "tokens")
    mid_b, mid_s = run()
    monkeypatch.delenv("RS_LATTICE_KERNEL")
    monkeypatch.setenv("RS_LATTICE_SEARCH", "tokens")
    ref_b, ref_s = run()
    for got_b, got_s in ((got_b, got_s), (vote_b, vote_s), (mid_b, mid_s)):
        for got, ref, n in ((got_b, ref_b, len(pcms)), (got_s, ref_s, 1)):
            for u in range(n):
                assert got.num_hyps(u) == ref.num_hyps(u), u
                assert got.counters(u)[4] == ref.counters(u)[4], u          # arcs of the raw lattice
                for k in range(ref.num_hyps(u)):
                    assert got.words(u, k) == ref.words(u, k), (u, k)
                    np.testing.assert_allclose(got.costs(u, k), ref.costs(u, k), rtol=1e-6)


@pytest.mark.parametrize("name,extra", VARIANT_CASES)
def test_cutoff_from_the_commit_histogram_is_the_exact_selection(case_cache, name, extra, monkeypatch):
    """RegDecodeKernel's GetCutoff takes the max-active / min-active order statistic from the histogram the commit pass leaves
    (decode_reg.hip); RS_REG_NO_HIST=1 makes every frame select it the way rounds 2-3 did (a histogram pass over the tokens, then
    KthFromHist): the same cutoffs, so the same tokens, pruning branches and costs, bit for bit."""
    from rhasspy_speech_amd import _lib, synth
    monkeypatch.setenv("RS_DECODER", "reg")
    model, pcm = make_model(case_cache, name, **extra)
    pcms = [pcm] + [synth.synth_utterance(900 + i, n) for i, n in enumerate([48000, 17000, 33000, 2500])]
    got = model.decode_batch(pcms)
    monkeypatch.setenv("RS_REG_NO_HIST", "1")
    ref = model.decode_batch(pcms)
    for u in range(len(pcms)):
        assert got.words(u) == ref.words(u)
        np.testing.assert_array_equal(got.costs(u), ref.costs(u))
        assert got.counters(u)[:4] == ref.counters(u)[:4] and got.counters(u)[5:7] == ref.counters(u)[5:7]
    # streams re-enter the kernel every advance (no histogram on a window's first frame)
    res = {}
    for env in ("0", "1"):
        monkeypatch.setenv("RS_REG_NO_HIST", env)
        st = _lib.Stream(model)
        raw = pcms[1].tobytes()
        for k in range(0, len(raw), 8192):
            st.accept(raw[k:k + 8192])
            st.advance()
        res[env] = st.finish()
    assert res["0"].words(0) == res["1"].words(0)
    np.testing.assert_array_equal(res["0"].costs(0), res["1"].costs(0))
    assert res["0"].counters(0)[:4] == res["1"].counters(0)[:4] and res["0"].counters(0)[5:7] == res["1"].counters(0)[5:7]


@pytest.mark.parametrize("name,extra", [("tiny_arpa_u7", {}), ("tiny_arpa_prune_u8", {}), ("zam_u0", dict(max_active=150, min_active=100, beam=10.0))])
def test_live_state_table_search_leaves_the_same_lattice(case_cache, name, extra, monkeypatch):
    """n-best lists (LatticeKernel on the token lists the search leaves behind) from the live-state-table search, from the dense-table
    search, and from the former with a table so small that some or all utterances are handed to the latter on the device."""
    from rhasspy_speech_amd import synth
    monkeypatch.setenv("RS_DECODER", "sparse")
    ref_model, pcm = make_model(case_cache, name, **extra)
    pcms = [pcm] + [synth.synth_utterance(500 + i, n) for i, n in enumerate([48000, 9000, 33000])]
    ref = ref_model.decode_batch(pcms, nbest=5)
    monkeypatch.setenv("RS_DECODER", "hash")
    # (limit "lds4": only 16 entries of the LDS part of the table are used, so nearly every state lives in its second level in global
    # memory; "40" / "6": frames of more live states than that hand the utterance to the dense-table kernel on the device)
    for limit in (None, "40", "6", "lds4"):
        monkeypatch.delenv("RS_HASH_SLOT_LIMIT", raising=False)
        monkeypatch.delenv("RS_HASH_LDS_LOG", raising=False)
        if limit == "lds4":
            monkeypatch.setenv("RS_HASH_LDS_LOG", "4")
        elif limit is not None:
            monkeypatch.setenv("RS_HASH_SLOT_LIMIT", limit)
        got = make_model(case_cache, name, **extra)[0].decode_batch(pcms, nbest=5)
        for u in range(len(pcms)):
            assert got.num_hyps(u) == ref.num_hyps(u), (limit, u)
            for k in range(ref.num_hyps(u)):
                assert got.words(u, k) == ref.words(u, k), (limit, u, k)
                np.testing.assert_allclose(got.costs(u, k), ref.costs(u, k), rtol=1e-6)
            assert got.counters(u)[3] == ref.counters(u)[3], (limit, u)
    monkeypatch.delenv("RS_HASH_SLOT_LIMIT", raising=False)
    monkeypatch.delenv("RS_HASH_LDS_LOG", raising=False)


def test_nbest_on_a_grammar_graph_ignores_the_per_frame_token_limit(case_cache):
    """An n-best / lattice call on a grammar graph runs the register-resident search, which keeps every live state of a frame, and
    DenseToTokensKernel writes them all: the utterance's slice of the token array holds S per frame whatever
    rs_decode_opts.max_tokens_per_frame says (round 4 sized the slice by the option and wrote past it)."""
    from rhasspy_speech_amd import synth
    ref_model, pcm = make_model(case_cache, "zam_u0")
    pcms = [pcm] + [synth.synth_utterance(900 + i, n) for i, n in enumerate([48000, 21000])]
    ref = ref_model.decode_batch(pcms, nbest=5)
    got = make_model(case_cache, "zam_u0", max_tokens_per_frame=8)[0].decode_batch(pcms, nbest=5)
    for u in range(len(pcms)):
        assert got.num_hyps(u) == ref.num_hyps(u)
        for k in range(ref.num_hyps(u)):
            assert got.words(u, k) == ref.words(u, k)
            np.testing.assert_array_equal(got.costs(u, k), ref.costs(u, k))


def test_time_slab_pipeline_is_the_same_search(case_cache, monkeypatch):
    """RS_OVERLAP_SLABS=n: the output layer and the (resumable) register-resident search are pipelined over n time slabs."""
    from rhasspy_speech_amd import synth
    ref_model, pcm = make_model(case_cache, "zam_u0")
    pcms = [pcm] + [synth.synth_utterance(700 + i, n) for i, n in enumerate([48000, 5000, 33000, 90000, 1700])]
    ref = ref_model.decode_batch(pcms)
    for slabs in ("2", "5"):
        monkeypatch.setenv("RS_OVERLAP_SLABS", slabs)
        got = make_model(case_cache, "zam_u0")[0].decode_batch(pcms)
        for u in range(len(pcms)):
            assert got.words(u) == ref.words(u)
            np.testing.assert_array_equal(got.costs(u), ref.costs(u))
            assert got.counters(u)[:4] == ref.counters(u)[:4] and got.counters(u)[5:7] == ref.counters(u)[5:7]
            np.testing.assert_array_equal(got.matrix(u, 2), ref.matrix(u, 2))


STREAM_CASES = list(cases.CASES)


@pytest.mark.parametrize("mode", ["finish_only", "advance_every_accept", "advance_irregular"])
@pytest.mark.parametrize("name", STREAM_CASES)
def test_streaming_case(case_cache, name, mode):
    """online2-cli-nnet3-decode-faster semantics: 1024-sample ticks, one warm-started iVector per nnet chunk.  The device work
    happens in rs_streams_advance as the audio arrives (finish_only: everything at the end); how the caller slices the audio
    and when it advances must not change anything -- every mode is held to the same reference goldens."""
    from rhasspy_speech_amd import _lib
    g = load_golden(name)
    model, pcm = make_model(case_cache, name)
    st = _lib.Stream(model)
    # deliberately odd chunking: the result must not depend on it
    rng = np.random.default_rng(len(name))
    pos, k = 0, 0
    while pos < len(pcm):
        step = 777 if mode != "advance_irregular" else int(rng.integers(1, 9000))
        st.accept(pcm[pos:pos + step])
        pos += step
        k += 1
        if mode == "advance_every_accept" or (mode == "advance_irregular" and k % 3 == 0):
            st.advance()
    res = st.finish(nbest=cases.NBEST)
    assert res.num_frames(0) == int(g["stream_num_frames"])
    if "stream_ivector" in g:
        iv = res.matrix(0, 1)
        assert iv.shape == g["stream_ivector"].shape
        assert np.abs(iv - g["stream_ivector"]).max() < IVEC_TOL
    ll = res.matrix(0, 2)
    sr, sc = g["loglikes_stride"]
    assert np.abs(ll[::sr, ::sc] - g["stream_loglikes"]).max() < LOGLIKE_TOL
    ref = parse_nbest(bytes(g["stream_nbest_text"]))
    assert [res.words(0, k) for k in range(res.num_hyps(0))] == ref
    assert res.text(0).split() == bytes(g["stream_nbest_text"]).split()


@pytest.mark.parametrize("name", ["tiny_u0", "tiny_noiv_u2", "tiny_cmvn_u4", "zam_u1", "zam_s12005", "tiny_arpa_u7", "tiny_fsf3_chunk20_u17", "tiny_fsf2_noiv_u18",
                                  "zam_fsf3_u19"])
def test_incremental_stream_equals_batch_replay(case_cache, name, monkeypatch):
    """The incremental engine (stream.cc) against the batch replay of the same stream (DecodeGroup(streaming = true),
    RS_STREAM_BATCH=1): features, per-chunk iVectors and log-likelihoods bit for bit, same 1-best and n-best."""
    from rhasspy_speech_amd import _lib
    model, pcm = make_model(case_cache, name)

    def run(advance):
        st = _lib.Stream(model)
        for i in range(0, len(pcm), 2048):
            st.accept(pcm[i:i + 2048])
            if advance:
                st.advance()
        return st.finish(nbest=1), None

    inc, _ = run(True)
    monkeypatch.setenv("RS_STREAM_BATCH", "1")
    rep, _ = run(False)
    monkeypatch.delenv("RS_STREAM_BATCH")
    for kind in (0, 1, 2):
        if kind == 1 and cases.CASES[name]["spec"].get("ivector_dim", 1) == 0:
            continue
        np.testing.assert_array_equal(inc.matrix(0, kind), rep.matrix(0, kind))
    assert inc.words(0) == rep.words(0)
    assert inc.costs(0) == rep.costs(0)
    # n-best through the lattice at the end of an incrementally decoded stream (the token-list search then runs at finish)
    st = _lib.Stream(model)
    for i in range(0, len(pcm), 5000):
        st.accept(pcm[i:i + 5000])
        st.advance()
    nb = st.finish(nbest=cases.NBEST)
    ref = parse_nbest(bytes(load_golden(name)["stream_nbest_text"]))
    assert [nb.words(0, k) for k in range(nb.num_hyps(0))] == ref


def test_finish_after_incremental_advances_is_cheap(case_cache):
    """A 30 s stream advanced as its audio arrives: what is left for finish is the tail -- the last chunk, at most a second of audio
    a coalesced advance left for the next call, and the advances still in flight -- so its wall time must be a fraction of decoding
    the whole stream at the end (same stream, no advances)."""
    import time
    from rhasspy_speech_amd import _lib
    model, pcm = make_model(case_cache, "zam_long30", keep_intermediates=0)

    def run(advance):
        st = _lib.Stream(model)
        for i in range(0, len(pcm), 8192):
            st.accept(pcm[i:i + 8192])
            if advance:
                st.advance()
        t0 = time.perf_counter()
        res = st.finish()
        return res, time.perf_counter() - t0

    run(True), run(False)                       # warm both paths (arenas, pool)
    inc, t_inc = run(True)
    full, t_full = run(False)
    assert inc.words(0) == full.words(0) and inc.costs(0) == full.costs(0)
    assert t_inc < 0.5 * t_full, (t_inc, t_full)
    # ... the library's own clock around the finishing call (timings[7]); timings[1:5] are the stage times of the WHOLE
    # stream (advances included), which add up to about the same device work either way
    assert inc.timings()[7] < 0.5 * full.timings()[7], (inc.timings(), full.timings())
    assert sum(inc.timings()[1:5]) > 0.5 * sum(full.timings()[1:5])


@pytest.mark.parametrize("name", ["zam_u1", "zam_fsf3_u19", "tiny_arpa_u7"])
def test_a_stream_that_outgrows_its_rows_moves_and_goes_on(case_cache, name, monkeypatch):
    """A stream opened with room for 256 frames and fed 14 s round by round: its rows move to a range twice as long three times
    (StreamGrow: pool rows of features, log-likelihoods, back pointers, per-chunk iVectors) while advances are in flight on the
    issuing thread -- same result, bit for bit, as the same stream with room from the start."""
    from rhasspy_speech_amd import _lib, synth
    model, _ = make_model(case_cache, name)
    pcm = synth.synth_utterance(4242, 16000 * 14)

    def run():
        st = _lib.Stream(model)
        for k in range(0, len(pcm), 8192):
            st.accept(pcm[k:k + 8192])
            st.advance()
        return st.finish(nbest=cases.NBEST)
    ref = run()
    monkeypatch.setenv("RS_STREAM_INIT_FRAMES", "256")
    got = run()
    for kind in (0, 1, 2):
        if kind == 1 and cases.CASES[name]["spec"].get("ivector_dim", 1) == 0:
            continue
        np.testing.assert_array_equal(got.matrix(0, kind), ref.matrix(0, kind))
    assert got.num_hyps(0) == ref.num_hyps(0)
    for k in range(ref.num_hyps(0)):
        assert got.words(0, k) == ref.words(0, k)
        np.testing.assert_array_equal(got.costs(0, k), ref.costs(0, k))


def test_failed_advance_poisons_its_streams(case_cache, monkeypatch):
    """An advance that throws after the chunk schedule of some streams has moved on (here: the pool has no room for a stream
    that outgrew its row range) must not leave streams that look usable: every later call on the streams of that advance
    except rs_stream_free is refused; other streams and new ones work, and the pool's rows come back."""
    from rhasspy_speech_amd import _lib, synth
    monkeypatch.setenv("RS_STREAM_POOL_ROWS", "8192")
    monkeypatch.setenv("RS_STREAM_INIT_FRAMES", "4096")
    model, pcm = make_model(case_cache, "tiny_u0")
    ref = model.decode_batch([pcm]).words(0)
    a, b = _lib.Stream(model), _lib.Stream(model)           # 2 x 4096 rows: the pool is full
    long_audio = synth.synth_utterance(77, 16000 * 45)      # 4498 frames: stream a needs 8192 contiguous rows
    a.accept(long_audio)
    b.accept(pcm)
    with pytest.raises(_lib.RsError, match="pool exhausted"):
        _lib.advance_streams([a, b])
    for call in (lambda: _lib.advance_streams([a]), lambda: _lib.advance_streams([b]), lambda: a.finish(), lambda: _lib.finish_streams([a, b])):
        with pytest.raises(_lib.RsError, match="advance that failed"):
            call()
    a.close()
    b.close()
    c = _lib.Stream(model)                                   # the rows and slots are back
    c.accept(pcm)
    c.advance()
    assert c.finish().words(0) == ref


@pytest.mark.parametrize("name", ["tiny_u0", "tiny_fsf3_u16"])
def test_many_streams_one_batch(case_cache, name):
    from rhasspy_speech_amd import _lib, synth
    model, _ = make_model(case_cache, name)
    pcms = [synth.synth_utterance(500 + i, n) for i, n in enumerate([48000, 20000, 70000, 5000])]
    streams = [_lib.Stream(model) for _ in pcms]
    for s, p in zip(streams, pcms):
        s.accept(p.tobytes())
    batch = _lib.finish_streams(streams, nbest=2)
    for i, p in enumerate(pcms):
        s = _lib.Stream(model)
        s.accept(p)
        one = s.finish(nbest=2)
        assert [batch.words(i, k) for k in range(batch.num_hyps(i))] == [one.words(0, k) for k in range(one.num_hyps(0))]
        assert np.array_equal(batch.matrix(i, 1), one.matrix(0, 1))
        assert np.array_equal(batch.matrix(i, 2), one.matrix(0, 2))


@pytest.mark.parametrize("name", ["tiny_u0", "tiny_arpa_u7", "tiny_hmm_u6", "zam_u1", "tinyf_u5"])
def test_pruned_output_layer_gives_the_same_search(case_cache, name):
    """rs_decode_opts.prune_output_pdfs evaluates the output layer only for the pdfs on HCLG arcs: same words, same costs
    (to GEMM rounding: the narrower layer may take another tile shape), n-best included; nets ending in a log-softmax
    (tinyf_u5) must not be pruned."""
    from rhasspy_speech_amd import _lib
    model_dir, graph_dir, _, pcm = case_cache(name)
    o = dict(cases.CASES[name].get("opts", {}))
    full = _lib.Model(model_dir, graph_dir, _lib.default_opts(prune_output_pdfs=0, **o))
    pruned = _lib.Model(model_dir, graph_dir, _lib.default_opts(prune_output_pdfs=1, **o))
    assert "pruned to the" not in full.describe()
    if name == "tinyf_u5":
        assert "pruned to the" not in pruned.describe()
    elif name == "zam_u1":                # (the tiny models' graphs use most of their 48 pdfs: pruning is skipped below 30 %)
        assert "pruned to the" in pruned.describe(), pruned.describe()
    for nbest in (1, cases.NBEST):
        a, b = full.decode_batch([pcm], nbest=nbest), pruned.decode_batch([pcm], nbest=nbest)
        assert a.num_hyps(0) == b.num_hyps(0)
        for k in range(a.num_hyps(0)):
            assert a.words(0, k) == b.words(0, k)
            np.testing.assert_allclose(b.costs(0, k), a.costs(0, k), rtol=2e-4, atol=2e-3)
    ref = parse_nbest(bytes(load_golden(name)["offline_nbest_text"]))
    assert pruned.decode_batch([pcm]).words(0) == ref[0]


@pytest.mark.parametrize("dims", [dict(hidden_dim=200, prefinal_dim=216, num_phones=100),       # n = 200/216/200: 256-column tile, padded
                                  dict(hidden_dim=328, prefinal_dim=250, num_phones=260),       # n = 328 (fp32 path), 520 pdfs (3 tiles)
                                  dict(hidden_dim=250, prefinal_dim=250, num_phones=1000, ivector_dim=0)])
def test_split_fp16_gemm_odd_shapes_match_oracle(tmp_path, dims):
    """Layer shapes around the split-fp16 kernel's tile edges (output widths that leave padding in the 256-column tile,
    segment widths that are not multiples of the 16-wide k-step, a net without iVector input): log-likelihoods against the
    CPU oracle on a ragged batch, which also runs tiles with fewer than 64 live rows."""
    from oracle import pipeline
    from rhasspy_speech_amd import _lib, synth
    spec = synth.ModelSpec(**dims)
    synth.write_model_dir(tmp_path / "model", spec)
    synth.make_grammar_graph(tmp_path / "graph", spec)
    model = _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(keep_intermediates=1))
    assert "gemm" in model.describe()
    orc = pipeline.Oracle(tmp_path / "model", tmp_path / "graph")
    pcms = [synth.synth_utterance(300 + i, n) for i, n in enumerate([20000, 9000, 16000])]
    res = model.decode_batch(pcms)
    for i, p in enumerate(pcms):
        tr = orc.transcribe(p)
        diff = np.abs(res.matrix(i, 2) - tr.loglikes).max()
        assert diff < LOGLIKE_TOL, (dims, i, diff)
        assert res.words(i) == tr.nbest[0].words


def test_split_fp16_gemm_is_as_close_to_exact_as_fp32_on_heavy_tailed_weights(tmp_path, monkeypatch):
    """The layer GEMMs emulate FP32 with two fp16 parts per operand.  Weights drawn from a heavy-tailed distribution (Student t,
    2 degrees of freedom, |w| up to 1000 times the typical one in a row: the per-column weight scale is set by an outlier)
    instead of N(0, 1 / fan_in): the log-likelihoods must be as close to the EXACT ones (oracle with float64 products) as an
    FP32 BLAS and the library's own exact-FP32 kernels are -- within a factor of two of the larger of their errors -- and no call
    may have left the split kernels' range."""
    from oracle import pipeline
    from rhasspy_speech_amd import _lib, synth
    spec = synth.ModelSpec(hidden_dim=256, prefinal_dim=256, num_phones=130, ivector_dim=10, num_gauss=16, lda_dim=12, weight_dist="heavy",
                           layer_offsets=((0,), (-1, 0, 1), (-3, 0, 3)), seed=21)
    synth.write_model_dir(tmp_path / "model", spec)
    synth.make_grammar_graph(tmp_path / "graph", spec)
    pcms = [synth.synth_utterance(500 + i, n) for i, n in enumerate([24000, 16000])]
    orc = pipeline.Oracle(tmp_path / "model", tmp_path / "graph")
    f32 = [orc.transcribe(p).loglikes for p in pcms]
    monkeypatch.setattr(pipeline.Nnet3, "matmul", staticmethod(lambda x, wt: (x.astype(np.float64) @ wt.astype(np.float64)).astype(np.float32)))
    f64 = [orc.transcribe(p).loglikes for p in pcms]
    monkeypatch.undo()
    model = _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(keep_intermediates=1))
    split = model.decode_batch(pcms)
    assert "range_retries=0 precision_retries=0 exact_fp32=0" in model.describe()
    monkeypatch.setenv("RS_GEMM_B3", "0")
    exact = model.decode_batch(pcms)
    monkeypatch.delenv("RS_GEMM_B3")
    for i in range(len(pcms)):
        scale = np.abs(f64[i]).max()
        e_blas = np.abs(f32[i] - f64[i]).max()
        e_exact = np.abs(exact.matrix(i, 2) - f64[i]).max()
        e_split = np.abs(split.matrix(i, 2) - f64[i]).max()
        assert np.abs(split.matrix(i, 2) - exact.matrix(i, 2)).max() > 0       # (two different kernels did run)
        assert e_split <= 2.0 * max(e_blas, e_exact) + 1e-7 * scale, (i, scale, e_blas, e_exact, e_split)
        assert split.words(i) == exact.words(i)


def test_activation_beyond_fp16_range_repeats_the_call_on_the_exact_kernels(tmp_path, monkeypatch):
    """A layer whose output exceeds 65504 in magnitude cannot be carried by the two-fp16 split: the kernels raise the call's
    range flag, the call is repeated on the exact-FP32 kernels (its results are THEIR results, bit for bit), and a model whose
    calls keep doing that changes kernels for good.  A stream advance cannot be repeated: it fails, and the model changes at once."""
    from rhasspy_speech_amd import _lib, synth
    spec = synth.tiny_spec(hidden_dim=256, prefinal_dim=256, hidden_gain=3.0e5, seed=9)
    synth.write_model_dir(tmp_path / "model", spec)
    synth.make_grammar_graph(tmp_path / "graph", spec)
    pcm = synth.synth_utterance(41, 32000)
    monkeypatch.setenv("RS_GEMM_B3", "0")
    ref_model = _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(keep_intermediates=1))
    ref = ref_model.decode_batch([pcm])
    rs = _lib.Stream(ref_model)
    rs.accept(pcm)
    ref_stream = rs.finish()
    rs.close()
    monkeypatch.delenv("RS_GEMM_B3")
    model = _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(keep_intermediates=1))
    for k in range(1, 4):
        res = model.decode_batch([pcm])
        assert np.array_equal(res.matrix(0, 2), ref.matrix(0, 2))
        assert res.words(0) == ref.words(0)
        assert f"range_retries={k} precision_retries=0 exact_fp32={1 if k >= 3 else 0}" in model.describe(), model.describe()
    res = model.decode_batch([pcm])                                  # no further repetitions: the model is on the exact kernels
    assert np.array_equal(res.matrix(0, 2), ref.matrix(0, 2)) and "range_retries=3 precision_retries=0 exact_fp32=1" in model.describe()
    # streams
    model2 = _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(keep_intermediates=1))
    st = _lib.Stream(model2)
    st.accept(pcm)
    with pytest.raises(_lib.RsError, match="range of the split-fp16"):
        st.advance()
        st.finish()
    st.close()
    assert "exact_fp32=1" in model2.describe()
    st = _lib.Stream(model2)                                          # a new stream of the model runs on the exact kernels
    st.accept(pcm)
    got = st.finish()
    assert np.array_equal(got.matrix(0, 2), ref_stream.matrix(0, 2)) and got.words(0) == ref_stream.words(0)
    st.close()


@pytest.mark.parametrize("gain", [1e-3, 1e-6])
def test_activation_below_fp16_subnormal_range_repeats_the_call_on_the_exact_kernels(tmp_path, monkeypatch, gain):
    """The underflow twin of the test above.  A hidden layer whose outputs are all of order `gain` (no BatchNorm behind it, the next
    layer's weights divided by `gain`: in exact arithmetic the network of gain 1): the low fp16 part of |x| < 2^-3 is subnormal, so the
    split carries such a row to 2^-25 absolute instead of 2^-22 relative -- 15 bits at 1e-3, 4 at 1e-6.  The kernels take the largest
    |x| of every operand row they split and raise the call's precision flag for a non-zero row below 2^-3; the call is repeated on
    the exact-FP32 kernels (bit for bit THEIR result, within 1e-4 of the float64 oracle), describe() counts it, the third such call
    moves the model to the exact kernels for good; a stream advance fails with a message and the model changes at once.  With the
    flag ignored (RS_GEMM_B3_NOUNDER=1, -DRS_TUNING builds only) the split result at gain 1e-6 is off by more than the tolerance:
    measured in profiles/r06/underflow_notes.txt."""
    from oracle import pipeline
    from rhasspy_speech_amd import _lib, synth
    spec = synth.tiny_spec(hidden_dim=256, prefinal_dim=256, hidden_gain=gain, gain_compensated=True, seed=9)
    synth.write_model_dir(tmp_path / "model", spec)
    synth.make_grammar_graph(tmp_path / "graph", spec)
    pcms = [synth.synth_utterance(41, 32000), synth.synth_utterance(42, 20000)]
    orc = pipeline.Oracle(tmp_path / "model", tmp_path / "graph")
    monkeypatch.setattr(pipeline.Nnet3, "matmul", staticmethod(lambda x, wt: (x.astype(np.float64) @ wt.astype(np.float64)).astype(np.float32)))
    f64 = [orc.transcribe(p) for p in pcms]
    monkeypatch.undo()
    monkeypatch.setenv("RS_GEMM_B3", "0")
    ref_model = _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(keep_intermediates=1))
    ref = ref_model.decode_batch(pcms)
    rs = _lib.Stream(ref_model)
    rs.accept(pcms[0])
    ref_stream = rs.finish()
    rs.close()
    monkeypatch.delenv("RS_GEMM_B3")
    for i in range(len(pcms)):
        assert np.abs(ref.matrix(i, 2) - f64[i].loglikes).max() < LOGLIKE_TOL
    model = _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(keep_intermediates=1))
    assert "regime=split-fp16" in model.describe()
    for k in range(1, 4):
        res = model.decode_batch(pcms)
        for i in range(len(pcms)):
            assert np.array_equal(res.matrix(i, 2), ref.matrix(i, 2))
            assert np.abs(res.matrix(i, 2) - f64[i].loglikes).max() < LOGLIKE_TOL
            assert res.words(i) == f64[i].nbest[0].words
        assert f"range_retries={k} precision_retries={k} exact_fp32={1 if k >= 3 else 0}" in model.describe(), model.describe()
    res = model.decode_batch(pcms)
    assert np.array_equal(res.matrix(0, 2), ref.matrix(0, 2))
    assert "range_retries=3 precision_retries=3 exact_fp32=1 regime=exact-fp32" in model.describe(), model.describe()
    # streams
    model2 = _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(keep_intermediates=1))
    st = _lib.Stream(model2)
    st.accept(pcms[0])
    with pytest.raises(_lib.RsError, match="below the precision range of the split-fp16"):
        st.advance()
        st.finish()
    st.close()
    assert "precision_retries=1 exact_fp32=1" in model2.describe(), model2.describe()
    st = _lib.Stream(model2)
    st.accept(pcms[0])
    got = st.finish()
    assert np.array_equal(got.matrix(0, 2), ref_stream.matrix(0, 2)) and got.words(0) == ref_stream.words(0)
    st.close()


def test_activations_of_order_one_never_raise_the_precision_flag(case_cache):
    """The other side of the rule: on the zamia-size model (ReLU + BatchNorm hidden layers, MFCC + iVector input whose iVector part is
    small beside the cepstra) no row is below the split's precision range -- a batch with a too-short clip, digital silence and a
    few-LSB signal in it included (rows of halo and guard that hold zeros are not rows `below` anything)."""
    from rhasspy_speech_amd import synth
    model, pcm = make_model(case_cache, "zam_u0")
    quiet = (synth.synth_utterance(77, 24000).astype(np.float32) * 2e-4).astype(np.int16)
    res = model.decode_batch([pcm, np.zeros(100, np.int16), np.zeros(16000, np.int16), quiet, pcm[:5000]])
    assert res.num_frames(1) == 0 and res.num_frames(2) > 0
    assert "range_retries=0 precision_retries=0 exact_fp32=0 regime=split-fp16" in model.describe(), model.describe()
    st = __import__("rhasspy_speech_amd")._lib.Stream(model)
    st.accept(quiet)
    st.finish()
    st.close()
    assert "range_retries=0 precision_retries=0" in model.describe()


@pytest.mark.parametrize("name", ["tiny_u0", "zam_u0", "zam_long30"])
def test_ubm_posteriors_on_the_matrix_cores_are_bitwise_the_vector_ones(case_cache, name, monkeypatch):
    """UbmPostMfmaKernel (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain per element) against UbmPostKernel (the same chain on
    the vector unit): identical iVectors and log-likelihoods, offline and per streaming chunk."""
    from rhasspy_speech_amd import _lib
    model, pcm = make_model(case_cache, name)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RS_UBM_MFMA", flag)
        off = model.decode_batch([pcm, pcm[: len(pcm) // 3]])
        st = _lib.Stream(model)
        st.accept(pcm)
        out[flag] = (off, st.finish())
    for a, b in zip(out["1"], out["0"]):
        for u in range(a.num_utts):
            np.testing.assert_array_equal(a.matrix(u, 1), b.matrix(u, 1))
            np.testing.assert_array_equal(a.matrix(u, 2), b.matrix(u, 2))


@pytest.mark.parametrize("name", ["tiny_u0", "zam_u0", "zam_long30"])
def test_ivector_batch_products_on_the_fp64_matrix_cores(case_cache, name, monkeypatch):
    """IvecQuadMfmaKernel / IvecLinearMfmaKernel (v_mfma_f64_16x16x4_f64) against the vector kernels they replace: the same
    sums in the same order -- iVectors equal to fp64 rounding (1e-9 on O(1) values), offline and per streaming chunk, and
    against the reference's goldens in test_offline_case / test_streaming_case."""
    from rhasspy_speech_amd import _lib
    model, pcm = make_model(case_cache, name)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RS_IVEC_MFMA", flag)
        off = model.decode_batch([pcm, pcm[: len(pcm) // 3], pcm[: len(pcm) // 2]])
        st = _lib.Stream(model)
        st.accept(pcm)
        out[flag] = (off, st.finish())
    for a, b in zip(out["1"], out["0"]):
        for u in range(a.num_utts):
            np.testing.assert_allclose(a.matrix(u, 1), b.matrix(u, 1), rtol=0, atol=1e-6)
            assert np.abs(a.matrix(u, 2) - b.matrix(u, 2)).max() < 1e-5
            assert a.words(u) == b.words(u)


@pytest.mark.parametrize("name", ["zam_u0", "zam_long30"])
def test_ivector_quadratic_product_with_hand_placed_waits(case_cache, name, monkeypatch):
    """IvecQuadMfmaAsmKernel (loads in inline asm, counted s_waitcnt, num_gauss a multiple of 128) forms the sums of
    IvecQuadMfmaKernel in the same order: iVectors bit for bit, offline (a batch that leaves rows of the 64-utterance tile empty)
    and per streaming chunk."""
    from rhasspy_speech_amd import _lib
    model, pcm = make_model(case_cache, name)
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("RS_IVEC_ASM", flag)
        off = model.decode_batch([pcm, pcm[: len(pcm) // 3], pcm[: len(pcm) // 2]])
        st = _lib.Stream(model)
        st.accept(pcm)
        out[flag] = (off, st.finish())
    for a, b in zip(out["1"], out["0"]):
        for u in range(a.num_utts):
            np.testing.assert_array_equal(a.matrix(u, 1), b.matrix(u, 1))
            np.testing.assert_array_equal(a.matrix(u, 2), b.matrix(u, 2))
            assert a.words(u) == b.words(u)
