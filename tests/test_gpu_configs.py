"""BASELINE.json configs[1..4] at their full sizes on the HIP path (`-m gpu`), every utterance checked against the REFERENCE.

tests/golden/configs/*.npz hold what the reference binaries (oracle/_ref, built from /root/reference; generator
oracle/gen_config_golden.py) produce for every utterance of
  [1] the headline grammar batch (256 x 3 s; the bench line),
  [2] the ARPA-LM HCLG batch (256 x 3 s),
  [3] the mixed-model batch: 1024 utterances, two independently seeded zamia-size model + graph sets, 512 each,
  [4] 64 concurrent 30 s streams (online2-cli-nnet3-decode-faster semantics),
namely 1-best word ids + nbest-to-linear's graph / acoustic cost.  Transcripts must be equal, costs within 2e-4 relative
(+2e-3 absolute: the reference's own BLAS-dependent rounding).  Property checks (a batch member
decodes as it does alone, decoder variants agree) stay beside them.
"""
from __future__ import annotations

import threading

import numpy as np
import pytest

from tests import configs

pytestmark = pytest.mark.gpu

LOGLIKE_TOL = 1e-4
COST_RTOL, COST_ATOL = 2e-4, 2e-3


@pytest.fixture(scope="module")
def zam_arpa(tmp_path_factory):
    """zamia-like-S acoustic model + a back-off ARPA HCLG (a few thousand states: beyond the register-resident decoder)."""
    return configs.build_arpa_model(tmp_path_factory.mktemp("zam_arpa"))


@pytest.fixture(scope="module")
def zam_tdnnf(tmp_path_factory):
    return configs.build_tdnnf_model(tmp_path_factory.mktemp("zam_tdnnf"))


@pytest.fixture(scope="module")
def zam_grammar(tmp_path_factory):
    return configs.build_grammar_model(tmp_path_factory.mktemp("zam_grammar"))


def _check_against_reference(name, words_of, costs_of, n):
    """Every utterance: transcript equal to the reference's; 1-best costs within tolerance -- except on the utterances listed in
    ORDER_DEPENDENT_COSTS (below), and on those by a few units of cost at most: the reference creates tokens while its running
    `next_cutoff` tightens (lattice-faster-decoder.cc:774-787), so which tokens beyond best + adaptive-beam exist depends on its
    HashList iteration order; the kernels prune with the final cutoff unless rs_decode_opts.exact_token_order is set (DESIGN.md
    section 2).  tests/test_oracle_golden.py pins the CPU oracle, which follows the hash order, to the reference's costs."""
    ref_words, ref_g, ref_a = configs.load_golden(name)
    assert len(ref_words) == n
    bad = [u for u in range(n) if words_of(u) != ref_words[u]]
    assert not bad, f"{name}: {len(bad)} of {n} transcripts differ from the reference, first {bad[:5]}: {words_of(bad[0])} vs {ref_words[bad[0]]}"
    got = np.array([costs_of(u) for u in range(n)], np.float64)
    ref_tot, tot = ref_g.astype(np.float64) + ref_a, got[:, 0] + got[:, 1]
    off = [u for u in range(n) if not (np.isclose(got[u, 0], ref_g[u], rtol=COST_RTOL, atol=COST_ATOL) and
                                       np.isclose(got[u, 1], ref_a[u], rtol=COST_RTOL, atol=COST_ATOL))]
    known = ORDER_DEPENDENT_COSTS.get(name, {})
    assert set(off) <= set(known), f"{name}: costs of utterances {sorted(set(off) - set(known))} differ from the reference's"
    for u in off:       # same words through an alignment the order-dependent pruning lost: a few units of cost at most
        assert abs(tot[u] - ref_tot[u]) < 4.0, (name, u, got[u], ref_g[u], ref_a[u])
    return off


def _same_result(a, i, b, j):
    assert a.num_hyps(i) == b.num_hyps(j)
    for k in range(a.num_hyps(i)):
        assert a.words(i, k) == b.words(j, k)
    np.testing.assert_array_equal(a.costs(i), b.costs(j))


def test_config1_headline_batch_vs_reference(zam_grammar):
    """The bench line's batch (rank 0): all 256 transcripts and costs equal the reference's, through both entry points."""
    import torch
    from rhasspy_speech_amd import _lib
    model = _lib.Model(*zam_grammar, _lib.default_opts(prune_output_pdfs=0))
    pcms = configs.grammar_utterances()
    res = model.decode_batch(pcms)
    _check_against_reference("c1_grammar", res.words, res.costs, len(pcms))
    d_pcm = torch.from_numpy(np.concatenate(pcms)).to("cuda:0")
    off = np.arange(len(pcms) + 1, dtype=np.int64) * configs.N_SAMPLES_3S
    dev = model.decode_batch_device(d_pcm.data_ptr(), off)
    _check_against_reference("c1_grammar", dev.words, dev.costs, len(pcms))
    # the output layer cut down to the pdfs on HCLG arcs (the library's default): same transcripts
    pruned_model = _lib.Model(*zam_grammar, _lib.default_opts())
    assert "pruned to the" in pruned_model.describe()
    pruned = pruned_model.decode_batch(pcms)
    _check_against_reference("c1_grammar", pruned.words, pruned.costs, len(pcms))


# The utterances of the full-size configs on which an order-dependent token of the reference (see _check_against_reference) changes a
# COST: same 5-best word sequences in the same order, the total of some hypothesis off by the amount noted (ours minus the
# reference's; the negative one is the reference pruning, with its larger token count, a token the kernels keep).  3 of 1600.
# (c2 162 and c3_fr 110: the CPU oracle, which follows the reference's hash order, lands on the reference's cost -- test_oracle_golden.py;
# c3_de 238: not a decoder effect -- on frame 55 the fifth and sixth best UBM Gaussians are 2.4e-6 apart in score, the reference's BLAS
# sums select the other one, the iVector moves by 2.6e-3; the oracle decoder on the reference's own log-likelihoods gives the
# reference's cost: profiles/r04/c3_de_238.txt.)
ORDER_DEPENDENT_COSTS = {"c2_arpa": {162: 2.05}, "c3_mixed_de": {238: 0.21}, "c3_mixed_fr": {110: -0.93}}


def _check_nbest_against_reference(name, res, n):
    """5-best lists of every utterance against the reference's (`lattice-to-nbest --n=5` on its own lattice, one process per
    utterance): the same word sequences in the same order for ALL utterances; costs within tolerance except on the utterances
    listed above, where the difference must be the one recorded."""
    ref = configs.load_golden_nbest(name)
    assert len(ref) == n
    wrong, off = [], {}
    for u in range(n):
        got = [(res.words(u, k),) + tuple(res.costs(u, k)) for k in range(res.num_hyps(u))]
        if [g[0] for g in got] != [r[0] for r in ref[u]]:
            wrong.append(u)
            continue
        for g, r in zip(got, ref[u]):
            if not (np.isclose(g[1], r[1], rtol=COST_RTOL, atol=COST_ATOL) and np.isclose(g[2], r[2], rtol=COST_RTOL, atol=COST_ATOL)):
                off[u] = (g[1] + g[2]) - (r[1] + r[2])
                break
    assert not wrong, f"{name}: the 5-best lists of {len(wrong)} of {n} utterances differ from the reference's, first {wrong[:8]}"
    known = ORDER_DEPENDENT_COSTS.get(name, {})
    assert set(off) <= set(known), f"{name}: 5-best costs differ from the reference's on {sorted(set(off) - set(known))}"
    for u, d in off.items():
        assert abs(d - known[u]) < 0.02, (name, u, d)


# ---- intermediate results at full size (round 5).  tests/golden/configs/<name>_inter.npz (oracle/gen_config_intermediates.py: `rs-dump`
# on the reference's own classes, a fresh process per utterance) hold the reference's iVector of every utterance (every eighth chunk's
# for the streams) and a strided sample of the log-likelihood matrices of 16 utterances per configuration.  Until round 4 only words and
# costs were compared at full size, so a discrete flip upstream (the UBM's top-5 selection is a discrete function of FP32 scores) was
# seen only when it moved a cost.  The utterances beyond tolerance are listed with their cause; everything else is within 1e-4.
IVEC_TOL = 1e-4
INTERMEDIATE_DEVIATIONS = {
    # Measured (round 5): 4 of the 1600 batch utterances and 1 of the 64 streams, each traced to ONE frame on which a discrete step of
    # the UBM posterior computation is decided inside the FP32 rounding of the scores (profiles/micro/ivector_near_ties.py: float64
    # scores beside the FP32 ones); the reference's BLAS sums land on one side, the kernels' k-ordered fmaf chain on the other, and
    # the iVector moves by 2e-4 .. 3e-3.  Every other iVector is within 2e-6 (median 6e-7).
    "c1_grammar": {117},      # frame 45: a posterior within 9e-6 (relative) of min_post (hmm/posterior.cc VectorToPosteriorEntry); |diff| 8.7e-4
    "c3_mixed_de": {238},     # frame 55: 5th / 6th best Gaussian 2.4e-6 apart in score (profiles/r04/c3_de_238.txt); 2.6e-3
    "c3_mixed_fr": {126},     # frame 209: 5th / 6th best Gaussian 7.0e-6 apart; 7.6e-4 (the numpy oracle lands on the kernels' side)
    "c4_streams": {22},       # frame 1259: 5th / 6th best Gaussian 9e-7 apart; chunks from 56 on differ by <= 1.8e-4
}


def _check_intermediates(name, res, n, stream=False):
    g = np.load(configs.GOLDEN / f"{name}_inter.npz")
    assert g["ivector"].shape[0] == n
    iv_err = np.zeros(n)
    for u in range(n):
        assert res.num_frames(u) == int(g["num_frames"][u]), (name, u)
        iv = res.matrix(u, 1)
        if stream:
            ref = g["chunk_iv"][u]
            ref = ref[~np.isnan(ref[:, 0])]
            got = iv[::int(g["chunk_stride"])]
            assert got.shape == ref.shape, (name, u, got.shape, ref.shape)
            iv_err[u] = max(np.abs(got - ref).max(), np.abs(iv[-1] - g["ivector"][u]).max())
        else:
            iv_err[u] = np.abs(iv[0] - g["ivector"][u]).max()
    off = sorted(int(u) for u in np.nonzero(iv_err >= IVEC_TOL)[0])
    known = INTERMEDIATE_DEVIATIONS.get(name, set())
    print(f"{name}: iVector max |diff| over {n} utterances: median {np.median(iv_err):.2e}, max outside the list {max([iv_err[u] for u in range(n) if u not in known], default=0):.2e}; "
          f"beyond {IVEC_TOL}: {[(u, float(iv_err[u])) for u in off]}")
    assert set(off) <= known, f"{name}: iVectors of utterances {sorted(set(off) - known)} differ from the reference's by {[float(iv_err[u]) for u in sorted(set(off) - known)]}"
    sr, sc = (int(x) for x in g["ll_stride"])
    worst = 0.0
    for k, u in enumerate(g["ll_utts"]):
        ll = res.matrix(int(u), 2)[::sr, ::sc]
        ref = g["ll"][k][:ll.shape[0]]
        assert ll.shape == ref.shape and not np.isnan(ref).any(), (name, u, ll.shape, ref.shape)
        err = float(np.abs(ll - ref).max())
        if int(u) not in known:
            assert err < LOGLIKE_TOL, f"{name}: log-likelihoods of utterance {u} differ from the reference's by {err}"
            worst = max(worst, err)
    print(f"{name}: log-likelihood samples of {len(g['ll_utts'])} utterances within {worst:.2e}")
    return off


def test_config1_intermediates_of_every_utterance(zam_grammar):
    from rhasspy_speech_amd import _lib
    pcms = configs.grammar_utterances()
    _check_intermediates("c1_grammar", _lib.Model(*zam_grammar, _lib.default_opts(keep_intermediates=1)).decode_batch(pcms), len(pcms))


def test_config2_intermediates_of_every_utterance(zam_arpa):
    from rhasspy_speech_amd import _lib
    pcms = configs.arpa_utterances()
    _check_intermediates("c2_arpa", _lib.Model(*zam_arpa, _lib.default_opts(keep_intermediates=1)).decode_batch(pcms), len(pcms))


def test_config3_intermediates_of_every_utterance(tmp_path_factory):
    from rhasspy_speech_amd import _lib
    names, pcms = configs.mixed_utterances()
    for key, tag in (("de_DE-like", "c3_mixed_de"), ("fr_FR-like", "c3_mixed_fr")):
        m = configs.MIXED_MODELS[key]
        md, gd = configs.build_grammar_model(tmp_path_factory.mktemp("im_" + tag), m["model_seed"], m["graph_seed"])
        mine = [p for nm, p in zip(names, pcms) if nm == key]
        _check_intermediates(tag, _lib.Model(md, gd, _lib.default_opts(keep_intermediates=1)).decode_batch(mine), len(mine))


def test_config4_intermediates_of_every_stream(zam_grammar):
    from rhasspy_speech_amd import _lib
    model = _lib.Model(*zam_grammar, _lib.default_opts(keep_intermediates=1))
    pcms = configs.stream_utterances()
    streams = [_lib.Stream(model) for _ in pcms]
    tick = 8 * 1024
    for r in range((max(len(p) for p in pcms) + tick - 1) // tick):
        _lib.accept_streams(streams, [p[r * tick:(r + 1) * tick] for p in pcms])
        _lib.advance_streams(streams)
    _check_intermediates("c4_streams", _lib.finish_streams(streams), len(pcms), stream=True)


def test_config1_five_best_of_every_utterance(zam_grammar):
    from rhasspy_speech_amd import _lib
    model = _lib.Model(*zam_grammar, _lib.default_opts())
    pcms = configs.grammar_utterances()
    _check_nbest_against_reference("c1_grammar", model.decode_batch(pcms, nbest=5), len(pcms))


def test_config2_five_best_of_every_utterance(zam_arpa):
    from rhasspy_speech_amd import _lib
    model = _lib.Model(*zam_arpa, _lib.default_opts())
    pcms = configs.arpa_utterances()
    _check_nbest_against_reference("c2_arpa", model.decode_batch(pcms, nbest=5), len(pcms))


def test_config3_five_best_of_every_utterance(tmp_path_factory):
    from rhasspy_speech_amd import _lib
    names, pcms = configs.mixed_utterances()
    for key, tag in (("de_DE-like", "c3_mixed_de"), ("fr_FR-like", "c3_mixed_fr")):
        m = configs.MIXED_MODELS[key]
        md, gd = configs.build_grammar_model(tmp_path_factory.mktemp("nb_" + tag), m["model_seed"], m["graph_seed"])
        mine = [p for nm, p in zip(names, pcms) if nm == key]
        _check_nbest_against_reference(tag, _lib.Model(md, gd, _lib.default_opts()).decode_batch(mine, nbest=5), len(mine))


def test_config4_five_best_of_every_stream(zam_grammar):
    from rhasspy_speech_amd import _lib
    model = _lib.Model(*zam_grammar, _lib.default_opts())
    pcms = configs.stream_utterances()
    streams = [_lib.Stream(model) for _ in pcms]
    tick = 8 * 1024
    for r in range((max(len(p) for p in pcms) + tick - 1) // tick):
        _lib.accept_streams(streams, [p[r * tick:(r + 1) * tick] for p in pcms])
        _lib.advance_streams(streams)
    _check_nbest_against_reference("c4_streams", _lib.finish_streams(streams, nbest=5), len(pcms))


def test_frame_subsampling_factor_3_at_config_size(tmp_path_factory):
    """--frame-subsampling-factor=3 in the headline model's online.conf (how a chain model is meant to be decoded; the upper layers then
    run on every third row): the 5-best lists of the first 128 utterances of configs[1] (one batch) and of the first 16 streams of
    configs[4] (8-tick rounds) against the reference binaries run with the same online.conf (oracle/gen_config_golden.py c1_fsf3 c4_fsf3)."""
    from rhasspy_speech_amd import _lib
    md, gd = configs.build_grammar_model(tmp_path_factory.mktemp("zam_fsf3"), conf_opts=configs.FSF3_CONF)
    model = _lib.Model(md, gd, _lib.default_opts())
    assert "frame_subsampling_factor=3" in model.describe() and "rows=every-3" in model.describe()
    pcms = configs.grammar_utterances()[:configs.N_FSF3_UTTS]
    res = model.decode_batch(pcms, nbest=5)
    assert all(res.num_frames(u) == 100 for u in range(len(pcms)))
    _check_nbest_against_reference("c1_fsf3", res, len(pcms))
    spcm = configs.stream_utterances()[:configs.N_FSF3_STREAMS]
    streams = [_lib.Stream(model) for _ in spcm]
    tick = 8 * 1024
    for r in range((max(len(p) for p in spcm) + tick - 1) // tick):
        _lib.accept_streams(streams, [p[r * tick:(r + 1) * tick] for p in spcm])
        _lib.advance_streams(streams)
    _check_nbest_against_reference("c4_fsf3", _lib.finish_streams(streams, nbest=5), len(spcm))


def test_streams_fed_through_the_array_entry_points(zam_grammar):
    """stream_handles / accept_streams_raw / advance_streams_raw (addresses and lengths as arrays the caller keeps, ragged rounds:
    streams that have run out of audio drop out of the accept call) against the per-object calls on the same audio."""
    from rhasspy_speech_amd import _lib
    model = _lib.Model(*zam_grammar, _lib.default_opts())
    pcms = [np.ascontiguousarray(p[:len(p) - 5000 * (i % 4)], dtype=np.int16) for i, p in enumerate(configs.stream_utterances()[:12])]
    tick = 8 * 1024
    n_rounds = (max(len(p) for p in pcms) + tick - 1) // tick
    ref_streams = [_lib.Stream(model) for _ in pcms]
    for r in range(n_rounds):
        live = [(s, p[r * tick:(r + 1) * tick]) for s, p in zip(ref_streams, pcms) if r * tick < len(p)]
        _lib.accept_streams([s for s, _ in live], [a for _, a in live])
        _lib.advance_streams(ref_streams)
    ref = _lib.finish_streams(ref_streams)
    streams = [_lib.Stream(model) for _ in pcms]
    handles = _lib.stream_handles(streams)
    base = np.array([p.__array_interface__["data"][0] for p in pcms], dtype=np.uintp)
    length = np.array([len(p) for p in pcms], dtype=np.int64)
    for r in range(n_rounds):
        left = length - r * tick
        live = left > 0
        _lib.accept_streams_raw(handles[live], (base + np.uintp(2 * r * tick))[live], np.minimum(left[live], tick).astype(np.int32))
        _lib.advance_streams_raw(handles)
    got = _lib.finish_streams(streams)
    for u in range(len(pcms)):
        assert got.words(u) == ref.words(u) and got.costs(u) == ref.costs(u)


def test_config2_arpa_hclg_256x3s(zam_arpa):
    from rhasspy_speech_amd import _lib
    model_dir, graph_dir = zam_arpa
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts())
    desc = model.describe()
    n_states = int(desc.split("hclg: states=")[1].split()[0])
    assert n_states > 2000, desc           # a larger FST than the grammar graph (625 states)
    pcms = configs.arpa_utterances()
    batch = model.decode_batch(pcms)
    assert batch.num_utts == 256
    _check_against_reference("c2_arpa", batch.words, batch.costs, len(pcms))
    # every utterance of the batch decodes exactly as it does alone (sampled)
    for u in (0, 1, 77, 128, 255):
        one = model.decode_batch([pcms[u]])
        _same_result(batch, u, one, 0)
    # n-best through the lattice path on a slice of the batch
    nb = model.decode_batch(pcms[:16], nbest=3)
    for u in range(16):
        assert nb.words(u, 0) == batch.words(u)


def test_config3_mixed_batch_1024_two_zamia_size_models(tmp_path_factory):
    """BASELINE configs[3] as specified: 1024 utterances, two independently seeded zamia-size model + graph sets (512 each,
    interleaved), through the sharded entry point (one rank here; ranks are covered by the gloo test on CPU).  Every
    transcript and cost against the reference's."""
    from rhasspy_speech_amd import _lib, shard
    models = {}
    for name, m in configs.MIXED_MODELS.items():
        md, gd = configs.build_grammar_model(tmp_path_factory.mktemp(name.replace("-", "_")), m["model_seed"], m["graph_seed"])
        models[name] = _lib.Model(md, gd, _lib.default_opts())
    names, pcms = configs.mixed_utterances()
    assert len(pcms) == 1024
    got = shard.decode_mixed_sharded(models, names, pcms)
    assert sorted(got) == list(range(len(pcms)))
    for key, tag in (("de_DE-like", "c3_mixed_de"), ("fr_FR-like", "c3_mixed_fr")):
        idx = [i for i, nm in enumerate(names) if nm == key]
        _check_against_reference(tag, lambda u: got[idx[u]][0], lambda u: got[idx[u]][1:], len(idx))
    # the C entry point (rs_decode_batch_sharded) on the same batch, as rank 1 of 2: its half only
    half = shard.decode_mixed_sharded(models, names, pcms, rank=1, world=2, gather=False)
    assert sorted(half) == list(range(1, len(pcms), 2))
    for i in half:
        assert half[i] == got[i]


def test_exact_token_order_closes_the_grammar_graph_deviation(tmp_path_factory):
    """Config 3 with rs_decode_opts.exact_token_order = 1 (the reference's running cutoff in its hash order, decode_reg.hip): of
    the two utterances of the 1024 whose costs differ from the reference's without it, c3_mixed_fr 110 -- the order-dependent one
    -- now has the reference's cost; c3_mixed_de 238 stays: the CPU oracle, which follows the order too, also lands on the
    kernels' cost there (a decision at the resolution of the log-likelihoods).  No other utterance moves."""
    from rhasspy_speech_amd import _lib
    names, pcms = configs.mixed_utterances()
    for key, tag, still in (("de_DE-like", "c3_mixed_de", {238}), ("fr_FR-like", "c3_mixed_fr", set())):
        m = configs.MIXED_MODELS[key]
        md, gd = configs.build_grammar_model(tmp_path_factory.mktemp(tag), m["model_seed"], m["graph_seed"])
        model = _lib.Model(md, gd, _lib.default_opts(exact_token_order=1))
        utts = [p for nm, p in zip(names, pcms) if nm == key]
        res = model.decode_batch(utts)
        ref_words, ref_g, ref_a = configs.load_golden(tag)
        off = set()
        for u in range(len(utts)):
            assert res.words(u) == ref_words[u]
            g, a = res.costs(u)
            if not (np.isclose(g, ref_g[u], rtol=COST_RTOL, atol=COST_ATOL) and np.isclose(a, ref_a[u], rtol=COST_RTOL, atol=COST_ATOL)):
                off.add(u)
        assert off == still, (tag, off)


def test_sharded_entry_point_issues_the_rccl_all_gather(zam_grammar):
    """rs_decode_batch_sharded with a real ncclComm_t (a one-rank RCCL communicator: the test box has one GPU): the records
    come back through ncclAllGather on the device and equal the ones of the collective-free call; a too-short utterance
    travels as a status, not as an exception."""
    import ctypes as C
    import torch  # noqa: F401  (loads the process's RCCL, the copy the library binds to)
    from rhasspy_speech_amd import _lib, shard
    rccl = C.CDLL("librccl.so.1")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        model = _lib.Model(*zam_grammar, _lib.default_opts())
        pcms = configs.grammar_utterances(24) + [np.zeros(200, np.int16)]
        rec, st, msg = _lib.decode_batch_sharded([model], [0] * len(pcms), pcms, 0, 1, comm.value)
        assert st == 0, msg
        plain, st2, _ = _lib.decode_batch_sharded([model], [0] * len(pcms), pcms, 0, 1, 0)
        np.testing.assert_array_equal(rec, plain)
        got = shard.unpack_records(rec, len(pcms))
        assert got.errors == {24: _lib.RS_ERR_DECODE}
        ref_words, ref_g, ref_a = configs.load_golden("c1_grammar")
        for u in range(24):
            assert got[u][0] == ref_words[u]
        # rank / world that contradict the communicator are refused before anything is launched
        _, st3, msg3 = _lib.decode_batch_sharded([model], [0] * len(pcms), pcms, 1, 2, comm.value)
        assert st3 != 0 and "communicator" in msg3
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_shard_gather_on_torch_distributed_communicator(zam_grammar):
    """The route bench.py takes at N > 1: torch.distributed's own RCCL communicator (ProcessGroupNCCL._comm_ptr(), here a one-rank
    group: the box has one GPU) handed to rs_shard_gather / rs_decode_batch_sharded -- the library binds the librccl.so.1 torch
    loaded, so the pointer is valid in it -- while torch keeps using the same communicator for its own collectives."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from rhasspy_speech_amd import _lib
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.ones(1, device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        comm = int(dist.distributed_c10d._get_default_group()._get_backend(torch.device("cuda", 0))._comm_ptr())
        assert comm != 0
        model = _lib.Model(*zam_grammar, _lib.default_opts())
        pcms = configs.grammar_utterances(16)
        own, st, msg = _lib.decode_batch_sharded([model], [0] * len(pcms), pcms, 0, 1, 0)
        assert st == 0, msg
        np.testing.assert_array_equal(_lib.shard_gather(own, 0, 0, 1, comm), own)
        both, st, msg = _lib.decode_batch_sharded([model], [0] * len(pcms), pcms, 0, 1, comm)
        assert st == 0, msg
        np.testing.assert_array_equal(both, own)
        dist.all_reduce(t)                  # torch's next collective on the same communicator
        torch.cuda.synchronize()
        assert float(t.item()) == 1.0
        with pytest.raises(_lib.RsError, match="communicator"):
            _lib.shard_gather(own, 0, 1, 2, comm)
    finally:
        dist.destroy_process_group()


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
    from rhasspy_speech_amd import _lib
    from oracle import pipeline
    model_dir, graph_dir = zam_grammar
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts(keep_intermediates=1))
    pcms = configs.stream_utterances()
    n_streams = len(pcms)
    rng = np.random.default_rng(4)
    streams = [_lib.Stream(model) for _ in pcms]
    # ragged, interleaved delivery (2048-byte reads like transcribe_stream.py, but of varying size per stream)
    pos = [0] * n_streams
    while any(p < len(x) for p, x in zip(pos, pcms)):
        for i, s in enumerate(streams):
            if pos[i] < len(pcms[i]):
                n = int(rng.integers(512, 40000))
                s.accept(pcms[i][pos[i]:pos[i] + n].tobytes())
                pos[i] += n
        _lib.advance_streams(streams)
    batch = _lib.finish_streams(streams)
    assert batch.num_utts == n_streams
    # every stream against the reference's streaming binary
    _check_against_reference("c4_streams", batch.words, batch.costs, n_streams)
    for i in (0, 31, 63):          # a stream alone gives the same result, bit for bit
        s = _lib.Stream(model)
        s.accept(pcms[i])
        one = s.finish()
        _same_result(batch, i, one, 0)
        np.testing.assert_array_equal(batch.matrix(i, 2), one.matrix(0, 2))
    # CPU oracle (streaming semantics of online2-cli-nnet3-decode-faster) on one stream: log-likelihoods
    orc = pipeline.Oracle(model_dir, graph_dir)
    tr = orc.transcribe_stream(pcms[5])
    assert batch.words(5) == tr.nbest[0].words
    assert np.abs(batch.matrix(5, 2) - tr.loglikes).max() < LOGLIKE_TOL


def test_split_fp16_gemm_matches_fp32_gemm(zam_grammar, monkeypatch):
    """The wide layers run on the fp16 matrix cores with every FP32 operand split into two fp16 parts (nnet_gemm_b3.hip).
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


def test_image_sourced_gemm_is_bitwise_the_split_gemm(zam_grammar, monkeypatch):
    """GemmKernelB3I (nnet_gemm_b3i.hip: the producing layer stored its result as fp16 operand images, the consumer copies
    fragments) against GemmKernelB3 (FP32 sources split inside the K-loop): the same MFMAs on the same operands in the same
    order, so log-likelihoods must be equal bit for bit -- on a ragged batch (partial tiles, both tile heights)."""
    from rhasspy_speech_amd import _lib, synth
    model = _lib.Model(*zam_grammar, _lib.default_opts(keep_intermediates=1))
    pcms = [synth.synth_utterance(31000 + u, 48000 - 640 * (u % 7)) for u in range(70)] + [synth.synth_utterance(31999, 1700)]
    monkeypatch.setenv("RS_GEMM_B3I", "1")
    img = model.decode_batch(pcms)
    monkeypatch.setenv("RS_GEMM_B3I", "0")
    plain = model.decode_batch(pcms)
    for u in range(len(pcms)):
        np.testing.assert_array_equal(img.matrix(u, 2), plain.matrix(u, 2))
        assert img.words(u) == plain.words(u) and img.costs(u) == plain.costs(u)


def test_all_dma_gemm_is_bitwise_the_image_gemm(zam_grammar, monkeypatch):
    """GemmKernelB3J (nnet_gemm_b3j.hip: both operands through LDS-DMA, hand-placed waits; one or two wave rows per workgroup,
    half-height tiles for the remainder) against GemmKernelB3I: same MFMAs, same order -> equal bit for bit.  The launch is made
    to believe the device runs 48 workgroups at a time so that this ragged batch (about 21 k rows) has whole rounds of
    full-height tiles AND a remainder of half-height ones."""
    from rhasspy_speech_amd import _lib, synth
    model = _lib.Model(*zam_grammar, _lib.default_opts(keep_intermediates=1))
    pcms = [synth.synth_utterance(32000 + u, 48000 - 640 * (u % 7)) for u in range(70)] + [synth.synth_utterance(32999, 1700)]
    monkeypatch.setenv("RS_GEMM_B3J", "0")
    ref = model.decode_batch(pcms)
    for wave_rows in ("1", "2"):                           # 128-row tiles, 2 workgroups per CU / 256-row tiles, 1 per CU
        for slots in ("48", "100000"):                     # full-height + half-height tiles; half-height tiles only
            monkeypatch.setenv("RS_GEMM_B3J", "2")        # any launch of at least 2 rows
            monkeypatch.setenv("RS_GEMM_B3J_WM", wave_rows)
            monkeypatch.setenv("RS_GEMM_B3J_SLOTS", slots)
            got = model.decode_batch(pcms)
            for u in range(len(pcms)):
                np.testing.assert_array_equal(got.matrix(u, 2), ref.matrix(u, 2))
                assert got.words(u) == ref.words(u) and got.costs(u) == ref.costs(u)


def test_160_row_tiles_are_bitwise_the_128_row_ones(zam_grammar, zam_tdnnf, monkeypatch):
    """GemmKernelB3J with five row blocks per wave (round 6: where a 160-row tile saves a round of tiles) against the 128-row shape:
    same MFMAs in the same k order per output element -> equal bit for bit.  RS_GEMM_B3J_MR forces either height.  Ragged batch
    with runs of too-short clips (the strip form's span bound for 160 list rows), the pruned and the full output layer (two / eight
    column tiles), and the factorised model (residual read through the image in the register epilogue of row block 4; the narrow
    bottleneck layers)."""
    from rhasspy_speech_amd import _lib, synth
    pcms = [synth.synth_utterance(34000 + u, 48000 - 640 * (u % 7)) for u in range(140)]
    for k in (5, 6, 60, 61, 62):
        pcms.insert(k, np.zeros(120, np.int16))
    for dirs, opts in ((zam_grammar, dict(keep_intermediates=1)), (zam_grammar, dict()), (zam_tdnnf, dict(keep_intermediates=1))):
        model = _lib.Model(*dirs, _lib.default_opts(**opts))
        some = pcms if dirs is zam_grammar else pcms[:80]
        out = {}
        for mr in ("4", "5"):
            monkeypatch.setenv("RS_GEMM_B3J_MR", mr)
            out[mr] = model.decode_batch(some)
        monkeypatch.delenv("RS_GEMM_B3J_MR")
        auto = model.decode_batch(some)
        for u in range(len(some)):
            if out["4"].num_frames(u) == 0:
                continue
            if opts:
                np.testing.assert_array_equal(out["5"].matrix(u, 2), out["4"].matrix(u, 2))
                np.testing.assert_array_equal(auto.matrix(u, 2), out["4"].matrix(u, 2))
            assert out["5"].words(u) == out["4"].words(u) == auto.words(u) and out["5"].costs(u) == out["4"].costs(u) == auto.costs(u)
        assert "range_retries=0 precision_retries=0" in model.describe()
        if dirs is zam_tdnnf:
            # the bottleneck layers (128 columns) on the 256 x 128 tile of two wave rows x two wave columns (the default) against the
            # 256-column shapes; also a batch large enough for more than one round of such tiles' slots (RS_GEMM_B3J_SLOTS)
            monkeypatch.setenv("RS_GEMM_B3J_NARROW", "0")
            wide = model.decode_batch(some)
            monkeypatch.delenv("RS_GEMM_B3J_NARROW")
            monkeypatch.setenv("RS_GEMM_B3J_SLOTS", "24")
            few_slots = model.decode_batch(some)
            monkeypatch.delenv("RS_GEMM_B3J_SLOTS")
            for u in range(len(some)):
                if auto.num_frames(u) == 0:
                    continue
                np.testing.assert_array_equal(auto.matrix(u, 2), wide.matrix(u, 2))
                np.testing.assert_array_equal(auto.matrix(u, 2), few_slots.matrix(u, 2))


def test_small_launches_on_the_all_dma_gemm_are_bitwise_the_image_gemm(zam_grammar, monkeypatch):
    """A launch of less than one round of tiles (a few utterances; a stream advance) runs GemmKernelB3J with 32-row tiles (round 5;
    GemmKernelB3I<1> before, RS_GEMM_B3J_SMALL=0): same MFMAs, same order -> equal bit for bit, batch and incremental stream."""
    from rhasspy_speech_amd import _lib, synth
    model = _lib.Model(*zam_grammar, _lib.default_opts(keep_intermediates=1))
    pcms = [synth.synth_utterance(33000 + u, n) for u, n in enumerate([48000, 30000, 1700, 20011])]

    def run():
        b = model.decode_batch(pcms)
        st = _lib.Stream(model)
        raw = pcms[0].tobytes()
        for k in range(0, len(raw), 16384):
            st.accept(raw[k:k + 16384])
            st.advance()
        return b, st.finish(1, 1.0)
    got_b, got_s = run()
    monkeypatch.setenv("RS_GEMM_B3J_SMALL", "0")
    ref_b, ref_s = run()
    for u in range(len(pcms)):
        np.testing.assert_array_equal(got_b.matrix(u, 2), ref_b.matrix(u, 2))
        assert got_b.words(u) == ref_b.words(u) and got_b.costs(u) == ref_b.costs(u)
    np.testing.assert_array_equal(got_s.matrix(0, 2), ref_s.matrix(0, 2))
    assert got_s.words(0) == ref_s.words(0) and got_s.costs(0) == ref_s.costs(0)


def test_pruned_output_layer_on_a_batch(zam_grammar):
    """prune_output_pdfs on the headline model / graph (362 of the 2000 pdfs are on HCLG arcs): a ragged batch decodes to
    the same words and costs as with the full output layer."""
    from rhasspy_speech_amd import _lib, synth
    model_dir, graph_dir = zam_grammar
    full = _lib.Model(model_dir, graph_dir, _lib.default_opts(prune_output_pdfs=0))
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


def test_lattice_arc_regions_grow_when_an_utterance_outgrows_its_share(zam_grammar):
    """Every utterance of an n-best call appends its lattice arcs to its own region of the context's arc buffer (1 M records
    shared out evenly at first): with 1360 utterances a region holds 771 arcs, less than a 3 s lattice of this graph has, so the
    first attempt reports the largest count, the buffer is re-allocated and the lattice pass repeated -- same 5-best lists as
    the same utterances in a small call."""
    from rhasspy_speech_amd import _lib
    model = _lib.Model(*zam_grammar, _lib.default_opts())
    base = configs.grammar_utterances()[:8]
    ref = model.decode_batch(base, nbest=5)
    assert max(ref.counters(u)[4] for u in range(8)) > 771          # (arcs of the raw lattice)
    res = _lib.Model(*zam_grammar, _lib.default_opts()).decode_batch(base * 170, nbest=5)
    for u in range(8 * 170):
        _same_result(res, u, ref, u % 8)


def test_streams_and_batch_calls_at_the_same_time(zam_grammar):
    """One model: a host thread advancing sixteen streams round by round (its advances' second halves issued by the pool's own
    thread, calls coalesced two by two) while two other threads decode batches -- every stream and every batch utterance equals its
    sequential result."""
    from rhasspy_speech_amd import _lib, synth
    model = _lib.Model(*zam_grammar, _lib.default_opts())
    spcm = [synth.synth_utterance(41000 + i, 16000 * 6 + 977 * i) for i in range(16)]
    batches = [[synth.synth_utterance(42000 + 100 * b + u, 48000 - 320 * ((u + b) % 11)) for u in range(48 + 16 * b)] for b in range(2)]

    def run_streams():
        sts = [_lib.Stream(model) for _ in spcm]
        pos = 0
        while pos < max(len(p) for p in spcm):
            for s, p in zip(sts, spcm):
                if pos < len(p):
                    s.accept(p[pos:pos + 8192])
            _lib.advance_streams([s for s, p in zip(sts, spcm) if pos < len(p) + 8192])
            pos += 8192
        return _lib.finish_streams(sts, nbest=1)

    ref_s = run_streams()
    ref_b = [model.decode_batch(pcms) for pcms in batches]
    out_s, out_b, errs = [], [[] for _ in batches], []

    def guard(fn):
        try:
            fn()
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    def streams_job():
        for _ in range(3):
            out_s.append(run_streams())

    def batch_job(b):
        for _ in range(6):
            out_b[b].append(model.decode_batch(batches[b]))

    ts = [threading.Thread(target=guard, args=(streams_job,))] + [threading.Thread(target=guard, args=(lambda b=b: batch_job(b),)) for b in range(len(batches))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert len(out_s) == 3
    for res in out_s:
        for i in range(len(spcm)):
            _same_result(res, i, ref_s, i)
    for b, pcms in enumerate(batches):
        for res in out_b[b]:
            for u in range(len(pcms)):
                _same_result(res, u, ref_b[b], u)


def test_two_utterance_groups_per_call(zam_grammar, monkeypatch):
    """RS_SUBBATCHES=2 (read at model load): a call of 32 or more utterances runs as two groups on two streams and two host
    threads.  The second group's stream must start behind the upload of the samples, which is queued on the first one's
    (round 3's advisor finding: nothing ordered them); results equal the one-group model's, call after call, from host memory
    (fresh upload every call) and with other calls in flight."""
    from rhasspy_speech_amd import _lib, synth
    monkeypatch.setenv("RS_SUBBATCHES", "2")
    two = _lib.Model(*zam_grammar, _lib.default_opts())
    monkeypatch.delenv("RS_SUBBATCHES")
    one = _lib.Model(*zam_grammar, _lib.default_opts())
    batches = [[synth.synth_utterance(23000 + 100 * b + u, 48000 - 320 * ((u + b) % 11)) for u in range(96 + 32 * b)] for b in range(3)]
    ref = [one.decode_batch(pcms) for pcms in batches]
    for rep in range(6):
        for b, pcms in enumerate(batches):
            got = two.decode_batch(pcms)
            for u in range(len(pcms)):
                _same_result(got, u, ref[b], u)
    out = [[] for _ in batches]

    def run(b):
        for _ in range(3):
            out[b].append(two.decode_batch(batches[b]))
    ts = [threading.Thread(target=run, args=(b,)) for b in range(len(batches))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for b, pcms in enumerate(batches):
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


# ---- the full-size factorised TDNN (round 6; tests/configs.py: TDNNF_SPEC).  Goldens: the reference's binaries, one process per
# utterance (c5_tdnnf.npz: 5-best lists and costs; c5_tdnnf_inter.npz: rs-dump's iVectors and log-likelihood samples).
def test_config5_tdnnf_full_size_vs_reference(zam_tdnnf, monkeypatch):
    """TdnnComponent + linear bottlenecks 1024 / 128 with Sum(Scale(0.66, .), .) residuals, 2 000 pdfs: every transcript, 5-best list
    and cost of the reference; iVectors and sampled log-likelihoods within 1e-4 -- on the split-fp16 layer GEMMs (the bottleneck
    outputs they read carry neither ReLU nor BatchNorm) with no call sent to the exact kernels, and the same on the exact kernels."""
    from rhasspy_speech_amd import _lib
    pcms = configs.grammar_utterances()[:configs.N_TDNNF_UTTS]
    model = _lib.Model(*zam_tdnnf, _lib.default_opts(keep_intermediates=1))
    res = model.decode_batch(pcms)
    _check_against_reference("c5_tdnnf", res.words, res.costs, len(pcms))
    assert _check_intermediates("c5_tdnnf", res, len(pcms)) == []
    assert "range_retries=0 precision_retries=0 exact_fp32=0 regime=split-fp16" in model.describe(), model.describe()
    _check_nbest_against_reference("c5_tdnnf", model.decode_batch(pcms, nbest=5), len(pcms))
    monkeypatch.setenv("RS_GEMM_B3", "0")
    exact = model.decode_batch(pcms)
    monkeypatch.delenv("RS_GEMM_B3")
    _check_against_reference("c5_tdnnf", exact.words, exact.costs, len(pcms))
    _check_intermediates("c5_tdnnf", exact, len(pcms))
    worst = max(float(np.abs(res.matrix(u, 2) - exact.matrix(u, 2)).max()) for u in range(len(pcms)))
    print(f"c5_tdnnf: split-fp16 against exact-FP32 layer GEMMs, all rows and pdfs of {len(pcms)} utterances: max |diff| {worst:.2e}")
    assert 0 < worst < 2 * LOGLIKE_TOL      # (each is within 1e-4 of the REFERENCE, above; this is their distance to each other)
    # The residual sums folded into the affine GEMMs' epilogues against the elementwise kernel they replace.  Reading the residual as
    # plain floats (RS_RESIDUAL_IMAGE=0): the same float operations, the same bits -- on the split kernels (GemmKernelB3J in registers,
    # the 32-row stream tiles, GemmKernelB3I + the residual pass) and on the exact ones.  The default reads it through the residual's
    # operand image -- the two-fp16-part value the next GEMM multiplies, within 2^-22 of the FP32 number -- so that no FP32 copy of the
    # activations exists at all: log-likelihoods move by less than the split kernels' own distance to the exact ones.
    assert sum("residual=0.66*" in l for l in model.describe().splitlines()) == 6, model.describe()
    monkeypatch.setenv("RS_FUSE_RESIDUAL", "0")
    plain_model = _lib.Model(*zam_tdnnf, _lib.default_opts(keep_intermediates=1))
    monkeypatch.delenv("RS_FUSE_RESIDUAL")
    assert "residual=" not in plain_model.describe() and sum(l.startswith("op: eltwise") for l in plain_model.describe().splitlines()) == 6
    monkeypatch.setenv("RS_RESIDUAL_IMAGE", "0")
    f32res_model = _lib.Model(*zam_tdnnf, _lib.default_opts(keep_intermediates=1))
    f32res_model.to_device()
    monkeypatch.delenv("RS_RESIDUAL_IMAGE")
    plain, f32res = plain_model.decode_batch(pcms), f32res_model.decode_batch(pcms)
    few = pcms[:20]                                   # (a launch of mid-size: GemmKernelB3I's 128-row tiles + the residual pass)
    plain_few, f32res_few, img_few = plain_model.decode_batch(few), f32res_model.decode_batch(few), model.decode_batch(few)
    monkeypatch.setenv("RS_GEMM_B3", "0")
    plain_exact = plain_model.decode_batch(pcms[:8])
    monkeypatch.delenv("RS_GEMM_B3")
    for u in range(len(pcms)):
        np.testing.assert_array_equal(f32res.matrix(u, 2), plain.matrix(u, 2))
    for u in range(len(few)):
        np.testing.assert_array_equal(f32res_few.matrix(u, 2), plain_few.matrix(u, 2))
        np.testing.assert_array_equal(f32res_few.matrix(u, 2), f32res.matrix(u, 2))      # (whichever kernel ran the layer)
        np.testing.assert_array_equal(img_few.matrix(u, 2), res.matrix(u, 2))
    for u in range(8):
        np.testing.assert_array_equal(exact.matrix(u, 2), plain_exact.matrix(u, 2))
    moved = max(float(np.abs(res.matrix(u, 2) - plain.matrix(u, 2)).max()) for u in range(len(pcms)))
    print(f"c5_tdnnf: residual through the operand image against the FP32 residual: max |diff| {moved:.2e}")
    assert 0 < moved < 6e-5
    for a_model, b_model in ((model, None), (f32res_model, plain_model)):
        sa = _lib.Stream(a_model)
        sa.accept(pcms[3])
        ra = sa.finish()
        sa.close()
        if b_model is None:
            continue
        sb = _lib.Stream(b_model)
        sb.accept(pcms[3])
        rb = sb.finish()
        sb.close()
        np.testing.assert_array_equal(ra.matrix(0, 2), rb.matrix(0, 2))


def test_config5_tdnnf_streams_and_subsampling(zam_tdnnf, tmp_path_factory):
    """The same model fed as streams (against the reference's streaming binary), and with --frame-subsampling-factor=3 in online.conf against
    the reference run that way: with the factor every layer above the (-1, 0, 1) ones -- bottlenecks, affines AND the residual sums --
    is evaluated on every third row (round 5 forced an elementwise op's operands dense without re-deriving what THEIR producers
    read)."""
    from rhasspy_speech_amd import _lib
    pcms = configs.grammar_utterances()[:configs.N_TDNNF_STREAMS]
    model = _lib.Model(*zam_tdnnf, _lib.default_opts())
    streams = [_lib.Stream(model) for _ in pcms]
    for r in range((configs.N_SAMPLES_3S + 8191) // 8192):
        _lib.accept_streams(streams, [p[r * 8192:(r + 1) * 8192] for p in pcms])
        _lib.advance_streams(streams)
    # (the reference's streaming binary on the same audio: an iVector per 24-frame chunk, so other costs than the offline decode's)
    _check_nbest_against_reference("c5_tdnnf_stream", _lib.finish_streams(streams, nbest=5), len(pcms))
    md3, gd3 = configs.build_tdnnf_model(tmp_path_factory.mktemp("zam_tdnnf_fsf3"), conf_opts=configs.FSF3_CONF)
    m3 = _lib.Model(md3, gd3, _lib.default_opts(keep_intermediates=1))
    assert "frame_subsampling_factor=3" in m3.describe()
    pcms3 = configs.grammar_utterances()[:configs.N_TDNNF_FSF3_UTTS]
    res3 = m3.decode_batch(pcms3)
    _check_against_reference("c5_tdnnf_fsf3", res3.words, res3.costs, len(pcms3))
    g = np.load(configs.GOLDEN / "c5_tdnnf_fsf3_inter.npz")
    sr, sc = (int(x) for x in g["ll_stride"])
    for k, u in enumerate(g["ll_utts"]):
        ll = res3.matrix(int(u), 2)[::sr, ::sc]
        np.testing.assert_allclose(ll, g["ll"][k][:ll.shape[0]], rtol=0, atol=LOGLIKE_TOL)
    assert "range_retries=0 precision_retries=0" in m3.describe(), m3.describe()
    _check_nbest_against_reference("c5_tdnnf_fsf3", m3.decode_batch(pcms3, nbest=5), len(pcms3))


def test_config6_tdnnf_1536_160_vs_reference(tmp_path_factory, monkeypatch):
    """A second factorised shape -- 1536-wide layers (six column tiles), 160-wide bottlenecks (NOT a whole number of 128-column tiles:
    the 256-column shapes at 37.5 % padding) -- against the reference's binaries: transcripts, 5-best lists, costs, iVectors,
    log-likelihood samples; the 128- and 160-row tile shapes agree bit for bit; one stream through the 32-row tiles."""
    from rhasspy_speech_amd import _lib
    md, gd = configs.build_tdnnf_model(tmp_path_factory.mktemp("zam_tdnnf1536"), spec_kw=configs.TDNNF1536_SPEC)
    pcms = configs.grammar_utterances()[:configs.N_TDNNF1536_UTTS]
    model = _lib.Model(md, gd, _lib.default_opts(keep_intermediates=1))
    res = model.decode_batch(pcms)
    _check_against_reference("c6_tdnnf1536", res.words, res.costs, len(pcms))
    assert _check_intermediates("c6_tdnnf1536", res, len(pcms)) == []
    _check_nbest_against_reference("c6_tdnnf1536", model.decode_batch(pcms, nbest=5), len(pcms))
    assert "range_retries=0 precision_retries=0 exact_fp32=0 regime=split-fp16" in model.describe(), model.describe()
    for mr in ("4", "5"):
        monkeypatch.setenv("RS_GEMM_B3J_MR", mr)
        forced = model.decode_batch(pcms)
        for u in range(len(pcms)):
            np.testing.assert_array_equal(forced.matrix(u, 2), res.matrix(u, 2))
    monkeypatch.delenv("RS_GEMM_B3J_MR")
    st = _lib.Stream(model)
    for k in range(0, len(pcms[5]), 6000):
        st.accept(pcms[5][k:k + 6000])
        st.advance()
    sres = st.finish()
    st.close()
    # (an iVector per chunk: not the batch decode's words in general; the stream path's parity on factorised models is
    # test_config5_tdnnf_streams_and_subsampling's -- here: the 32-row tiles with six column tiles run, and no range flag is raised)
    assert sres.num_frames(0) == res.num_frames(5) and len(sres.words(0)) > 0
    assert "range_retries=0 precision_retries=0" in model.describe()


def test_too_short_clips_inside_a_large_batch(zam_grammar):
    """Utterances without a single frame between the others of a batch large enough for the 128-row layer GEMM tiles (ADVICE r05:
    an empty utterance owns L + R halo rows but no entry of a row list, so the physical rows 128 list entries reach over were
    under-estimated and GemmKernelB3J's 192-row strip was read beyond its end): every non-empty utterance decodes as it does in a
    batch without the empty ones, bit for bit."""
    from rhasspy_speech_amd import _lib
    model = _lib.Model(*zam_grammar, _lib.default_opts(keep_intermediates=1))
    base = configs.grammar_utterances()[:96]
    rng = np.random.default_rng(6)
    empty = np.zeros(100, np.int16)
    pcms, where = [], []
    for i, p in enumerate(base):
        n = len(p) - 160 * int(rng.integers(0, 120))          # ragged: 178 .. 298 frames
        pcms.append(p[:n]); where.append(len(pcms) - 1)
        for _ in range(int(rng.integers(0, 4)) if i % 3 == 0 else 0):      # runs of up to three empty utterances
            pcms.append(empty)
    assert len(pcms) > len(base) + 20
    a = model.decode_batch(pcms)
    b = model.decode_batch([pcms[w] for w in where])
    for j, w in enumerate(where):
        assert a.num_frames(w) == b.num_frames(j) > 0
        np.testing.assert_array_equal(a.matrix(w, 2), b.matrix(j, 2))
        assert a.words(w) == b.words(j)
    for i in range(len(pcms)):
        if i not in where:      # (the reference's binary fails such an utterance: "You cannot get a lattice if you decoded no frames")
            assert a.num_frames(i) == 0
            with pytest.raises(_lib.RsError, match="decoded no frames"):
                a.words(i)
