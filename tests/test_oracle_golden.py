"""CPU tests that PIN the oracle: every stage of oracle/pipeline.py against vectors produced by the reference's
own binaries (tests/golden/*.npz, see oracle/gen_golden.py).  Runs without a GPU."""
import numpy as np
import pytest

from tests import cases

FAST = ["tiny_u0", "tiny_u3_short", "tiny_real_hot", "tiny_real_time", "tiny_text_u1", "tiny_noiv_u2", "tiny_cmvn_u4", "tinyf_u5",
        "tiny_hmm_u6", "tiny_arpa_u7", "tiny_arpa_prune_u8", "tiny_vecfst_u9", "zam_real_cold", "zam_long30", "zam_s12005",
        "tiny_silence", "tiny_quiet_u10", "tiny_nodither_u11", "tiny_dither05_u12", "zam_quiet_u13", "tiny_confopts_u15", "tiny_win100_u21"]


def parse_nbest(text: bytes):
    return [[int(x) for x in ln.split()[1:]] for ln in text.decode().splitlines() if ln.split()]


@pytest.fixture(scope="module")
def oracles(case_cache):
    from oracle import pipeline
    built = {}

    def get(name):
        if name not in built:
            model_dir, graph_dir, wav, pcm = case_cache(name)
            o = cases.CASES[name].get("opts", {})
            built[name] = (pipeline.Oracle(model_dir, graph_dir, **o), pcm)
        return built[name]

    return get


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference(oracles, name):
    g = np.load(cases.GOLDEN / f"{name}.npz")
    orc, pcm = oracles(name)
    tr = orc.transcribe(pcm, nbest=cases.NBEST)
    assert tr.num_frames == int(g["offline_num_frames"])
    fd = np.abs(tr.feats - g["input"])
    # the dither is the reference's to the bit (oracle/dither.c), so is the frame sum behind the DC offset (pipeline.frame_sum),
    # the FFT is the reference's own, operation for operation, and the mel filter edges come from the same libm logf; what is
    # left is the order of the BLAS sums (mel, DCT) times the cepstral lifter (up to 12)
    assert fd.max() < 2e-4 and np.quantile(fd, 0.99) < 1e-4
    assert int(g["rand_calls"]) == orc.rand_calls or orc.mfcc.o.dither == 0.0
    if "offline_ivector" in g:
        assert np.abs(tr.ivector - g["offline_ivector"][0]).max() < 1e-4
    sr, sc = g["loglikes_stride"]
    assert np.abs(tr.loglikes[::sr, ::sc] - g["offline_loglikes"]).max() < 1e-4
    ref = parse_nbest(bytes(g["offline_nbest_text"]))
    got = [p.words for p in tr.nbest]
    assert got == ref, (got, ref)
    np.testing.assert_allclose([p.graph_cost for p in tr.nbest], g["offline_graph_cost"], rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose([p.acoustic_cost for p in tr.nbest], g["offline_acoustic_cost"], rtol=2e-4, atol=2e-3)
    assert tr.text().split() == bytes(g["offline_nbest_text"]).split()


def test_oracle_intermediate_ivector_features(oracles):
    """CMVN / splice+LDA streams of the iVector branch against the reference classes' own output."""
    from oracle import pipeline
    g = np.load(cases.GOLDEN / "tiny_u0.npz")
    orc, pcm = oracles("tiny_u0")
    feats = orc.features(pcm)
    cm = pipeline.online_cmvn(feats, orc.ie["gstats"])
    assert np.abs(cm - g["cmvn"]).max() < 5e-4
    raw = pipeline.lda_transform(pipeline.splice(feats, orc.ie["left"], orc.ie["right"]), orc.ie["lda"])
    nrm = pipeline.lda_transform(pipeline.splice(cm, orc.ie["left"], orc.ie["right"]), orc.ie["lda"])
    assert np.abs(raw - g["lda"]).max() < 2e-4
    assert np.abs(nrm - g["lda_norm"]).max() < 2e-4


@pytest.mark.parametrize("name", ["tiny_u0", "tiny_u3_short", "tiny_real_hot", "tinyf_u5", "tiny_noiv_u2", "tiny_cmvn_u4", "tiny_arpa_u7", "zam_long30", "zam_s12005",
                                  "tiny_silence", "tiny_quiet_u10", "tiny_dither05_u12"])
def test_oracle_streaming_matches_reference(oracles, name):
    """online2-cli-nnet3-decode-faster goldens: per-chunk iVectors, log-likelihoods, n-best text."""
    g = np.load(cases.GOLDEN / f"{name}.npz")
    orc, pcm = oracles(name)
    sched, L, R = orc.stream_schedule(len(pcm))
    if "stream_chunk_tick" in g:
        assert [j for j, _ in sched] == [int(x) for x in g["stream_chunk_tick"]]
    tr = orc.transcribe_stream(pcm, nbest=cases.NBEST)
    if "stream_ivector" in g:
        assert np.abs(tr.ivector - g["stream_ivector"]).max() < 1e-4
    sr, sc = g["loglikes_stride"]          # big cases keep a strided sample of the log-likelihood matrix
    assert np.abs(tr.loglikes[::sr, ::sc] - g["stream_loglikes"]).max() < 1e-4
    assert tr.text() == bytes(g["stream_nbest_text"])


# ---- full-size configurations (tests/configs.py): the restatement against the reference's per-utterance goldens on samples
@pytest.mark.parametrize("config,utts", [("c1_grammar", (0, 131, 255)), ("c2_arpa", (3, 162)), ("c3_mixed_de", (7,)), ("c3_mixed_fr", (110, 500)),
                                         ("c4_streams", (5,)), ("c5_tdnnf", (2, 40)), ("c5_tdnnf_fsf3", (9,)), ("c6_tdnnf1536", (3,))])
def test_oracle_matches_config_goldens(tmp_path_factory, config, utts):
    """c2_arpa utterance 162 and c3_mixed_fr utterance 110 are the cases where the reference's order-dependent pruning (tokens created under a running
    next_cutoff, lattice-faster-decoder.cc:774-787) changes the best path's cost: oracle/decoder.c follows the HashList
    iteration order and must land on the reference's cost there (the HIP kernels do not, see test_gpu_configs.py)."""
    from oracle import pipeline
    from tests import configs
    ref_words, ref_g, ref_a = configs.load_golden(config)
    root = tmp_path_factory.mktemp(config)
    if config == "c2_arpa":
        md, gd = configs.build_arpa_model(root)
        pcms = configs.arpa_utterances()
    elif config.startswith("c3_mixed"):
        key = "de_DE-like" if config.endswith("de") else "fr_FR-like"
        m = configs.MIXED_MODELS[key]
        md, gd = configs.build_grammar_model(root, m["model_seed"], m["graph_seed"])
        names, allp = configs.mixed_utterances()
        pcms = [p for nm, p in zip(names, allp) if nm == key]
    elif config.startswith("c5_tdnnf") or config == "c6_tdnnf1536":          # the full-size factorised TDNNs (also pin the oracle's log-likelihoods, below)
        md, gd = configs.build_tdnnf_model(root, conf_opts=configs.FSF3_CONF if config.endswith("fsf3") else None,
                                           spec_kw=configs.TDNNF1536_SPEC if config == "c6_tdnnf1536" else None)
        pcms = configs.grammar_utterances()
    else:
        md, gd = configs.build_grammar_model(root)
        pcms = configs.stream_utterances() if config == "c4_streams" else configs.grammar_utterances()
    orc = pipeline.Oracle(md, gd)
    for u in utts:
        tr = orc.transcribe_stream(pcms[u]) if config == "c4_streams" else orc.transcribe(pcms[u])
        assert tr.nbest[0].words == ref_words[u], (config, u)
        np.testing.assert_allclose([tr.nbest[0].graph_cost, tr.nbest[0].acoustic_cost], [ref_g[u], ref_a[u]], rtol=2e-4, atol=2e-3)
    if config.startswith("c5_tdnnf") or config == "c6_tdnnf1536":
        g = np.load(configs.GOLDEN / f"{config}_inter.npz")
        sr, sc = (int(x) for x in g["ll_stride"])
        k = 0                                        # (ll_utts[0] is utterance 0)
        tr = orc.transcribe(pcms[int(g["ll_utts"][k])])
        ll = tr.loglikes[::sr, ::sc]
        np.testing.assert_allclose(ll, g["ll"][k][:ll.shape[0]], rtol=0, atol=1e-4)
        np.testing.assert_allclose(tr.ivector[0] if tr.ivector.ndim == 2 else tr.ivector, g["ivector"][int(g["ll_utts"][k])], rtol=0, atol=1e-4)


@pytest.mark.parametrize("i", [0, 5, 11, 17, 23, 31, 37, 41])
def test_oracle_matches_random_case_goldens(tmp_path, i):
    """The CPU restatement on a sample of the randomised decode cases (tests/fuzz_cases.py; goldens of the reference's binaries in
    tests/golden/fuzz_decode.json): 5-best lists and costs."""
    import json
    from oracle import pipeline
    from tests import fuzz_cases
    case = fuzz_cases.CASES[i]
    gold = json.loads((cases.GOLDEN / "fuzz_decode.json").read_text())[i]["offline"]
    assert gold["status"] == 0
    model_dir, graph_dir, _wav, pcm = cases.build_case_files(case, tmp_path)
    orc = pipeline.Oracle(model_dir, graph_dir, **case.get("opts", {}))
    tr = orc.transcribe(pcm, nbest=cases.NBEST)
    assert tr.text().split() == gold["nbest_text"].encode().split()
    np.testing.assert_allclose([p.graph_cost for p in tr.nbest], gold["graph_cost"], rtol=2e-4, atol=2e-3)
    np.testing.assert_allclose([p.acoustic_cost for p in tr.nbest], gold["acoustic_cost"], rtol=2e-4, atol=2e-3)


# ------------------------------------------------------------------------------------------------ dither

def test_dither_table_is_the_references():
    """oracle/dither.c (glibc rand() + rand_r() + RandGauss restated) against the noise the reference's own Dither() produced in a
    fresh process (rs-dump dither, oracle/gen_golden.py): bit for bit."""
    from oracle import pipeline
    g = np.load(cases.GOLDEN / "dither_ref.npz")
    for key in g.files:
        ref = g[key]
        got = pipeline.dither_table(ref.shape[0], ref.shape[1], 0)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), key


def test_glibc_rand_restated():
    """The TYPE_3 generator of oracle/dither.c against libc itself (srand(1) = the state of a fresh process)."""
    import ctypes
    from oracle import nnet3_rand
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    want = [libc.rand() for _ in range(5000)]
    r = nnet3_rand.GlibcRand()
    assert [r.rand() for _ in range(5000)] == want


def test_frame_sum_is_the_blas_order():
    """pipeline.frame_sum against cblas_sdot(n, x, 1, &one, 0) of the OpenBLAS the reference build links (the one inside SciPy):
    VectorBase::Sum() of the reference, which decides the rounding of the DC offset once the samples carry dither noise."""
    import ctypes, glob, os
    import scipy
    from oracle import pipeline
    libs = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    if not libs:
        pytest.skip("SciPy's OpenBLAS not found")
    f = ctypes.CDLL(libs[0]).scipy_cblas_sdot
    f.restype = ctypes.c_float
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    one = np.ones(1, np.float32)
    rng = np.random.default_rng(5)
    for n in (400, 200, 275):
        x = (rng.integers(-8000, 8000, (64, n)) * (rng.random((64, 1)) < 0.8) + rng.standard_normal((64, n))).astype(np.float32)
        want = np.array([f(n, np.ascontiguousarray(r).ctypes.data, 1, one.ctypes.data, 0) for r in x], np.float32)
        assert np.array_equal(pipeline.frame_sum(x).view(np.uint32), want.view(np.uint32)), n


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_setup_rand_calls_match_reference(name):
    """How often the reference's model set-up calls rand() before the first frame (rs-dump randpos on the reference's own
    classes, stored by gen_golden.py): the oracle's restatement (oracle/nnet3_rand.py) and the library's (rs_nnet3_setup,
    csrc/nnet3_setup.cc; host code, no device needed) both reproduce it exactly, and the library's collapsed network is the
    reference's (one GEMM less: lda folded into the first affine layer)."""
    import ctypes as C
    import tempfile
    from pathlib import Path
    from oracle import kaldi_formats as kf, nnet3_rand
    from rhasspy_speech_amd import synth
    from rhasspy_speech_amd._lib import load_library
    g = np.load(cases.GOLDEN / f"{name}.npz")
    want = int(g["rand_calls"])
    with tempfile.TemporaryDirectory() as td:
        synth.write_model_dir(Path(td) / "m", cases.case_spec(cases.CASES[name]))
        mdl = Path(td) / "m" / "model" / "model" / "final.mdl"
        _, nf = kf.read_final_mdl(mdl)
        co = cases.CASES[name].get("conf_opts", {})
        chunk = int(co.get("frames-per-chunk", 24))      # (the looped computation is compiled for the chunk size ...
        fsf = int(co.get("frame-subsampling-factor", 1))      # ... and the output frames t = 0, fsf, 2 fsf, ...)
        assert nnet3_rand.setup_rand_calls(nf, chunk, 0, fsf) == want
        lib = load_library()
        lib.rs_nnet3_setup_subsampled.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_char_p, C.c_size_t]
        n, cert, buf = C.c_int64(), C.c_int32(), C.create_string_buffer(1 << 16)
        assert lib.rs_nnet3_setup_subsampled(str(mdl).encode(), chunk, fsf, C.byref(n), C.byref(cert), buf, len(buf)) == 0
        assert (n.value, cert.value) == (want, 1)
        cfg = buf.value.decode()
        assert "component-node name=lda " not in cfg and "component=lda.tdnn1.affine" in cfg
        # ... line for line the network the reference's CollapseModel leaves (oracle/gen_collapsed_golden.py)
        import json
        want_cfg = json.loads((cases.GOLDEN / "collapsed_configs.json").read_text())[name]
        assert [l for l in cfg.splitlines() if l.strip()] == want_cfg


def test_library_dither_noise_is_the_references():
    """rs_dither_noise (the table the MFCC kernel reads) against the reference's Dither() and, at an offset, the oracle's table."""
    import ctypes as C
    from oracle import pipeline
    from rhasspy_speech_amd._lib import load_library
    lib = load_library()
    lib.rs_dither_noise.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    g = np.load(cases.GOLDEN / "dither_ref.npz")
    for key in g.files:
        ref = g[key]
        got = np.zeros_like(ref)
        assert lib.rs_dither_noise(0, 0, ref.shape[0], ref.shape[1], got.ctypes.data) == 0
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), key
    want = pipeline.dither_table(40, 400, 7741)[30:40]
    got = np.zeros_like(want)
    assert lib.rs_dither_noise(7741, 30, 40, 400, got.ctypes.data) == 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
