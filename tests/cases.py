"""Parity-test case table shared by the GPU parity tests, the CPU oracle tests and oracle/gen_golden.py.

A case = (synthetic model spec, graph kind, audio source, decoder options).  Model / graph files are
regenerated from seeds with rhasspy_speech_amd.synth wherever the case is used; expected outputs of the
reference for each case live in tests/golden/<case>.npz (written by oracle/gen_golden.py in the build container).
"""
from __future__ import annotations

import wave
from pathlib import Path

import numpy as np

from rhasspy_speech_amd import synth

GOLDEN = Path(__file__).resolve().parent / "golden"

CASES = {
    # name: dict(spec=kwargs for tiny_spec/ModelSpec, big=bool, graph=..., audio=..., decoder opts, nbest)
    "tiny_u0": dict(spec=dict(), graph="grammar", audio="synth:0:48000"),
    "tiny_u3_short": dict(spec=dict(), graph="grammar", audio="synth:3:9000"),
    "tiny_real_hot": dict(spec=dict(), graph="grammar", audio="wav:how_hot_is_it.wav"),
    "tiny_real_time": dict(spec=dict(seed=5), graph="grammar", audio="wav:what_time_is_it.wav"),
    "tiny_text_u1": dict(spec=dict(binary=False), graph="grammar", audio="synth:1:40000"),
    "tiny_noiv_u2": dict(spec=dict(ivector_dim=0), graph="grammar", audio="synth:2:48000"),
    "tiny_cmvn_u4": dict(spec=dict(nnet_cmvn=True), graph="grammar", audio="synth:4:48000"),
    "tinyf_u5": dict(spec=dict(tdnnf=True, with_priors=True, with_log_softmax=True,
                               layer_offsets=((0,), (-1, 0, 1), (-1, 0, 1), (-3, 0, 3))), graph="grammar", audio="synth:5:48000"),
    "tiny_hmm_u6": dict(spec=dict(chain_topology=False, seed=3), graph="grammar", audio="synth:6:32000"),
    "tiny_arpa_u7": dict(spec=dict(num_phones=40), graph="arpa:60:200", audio="synth:7:48000"),
    "tiny_arpa_prune_u8": dict(spec=dict(num_phones=40), graph="arpa:300:1500", audio="synth:8:48000",
                               opts=dict(max_active=60, min_active=20, beam=12.0)),
    "tiny_vecfst_u9": dict(spec=dict(), graph="grammar:vector", audio="synth:9:30000"),
    "zam_u0": dict(big=True, spec=dict(), graph="grammar", audio="synth:0:48000"),
    "zam_u1": dict(big=True, spec=dict(), graph="grammar", audio="synth:1:48000"),
    "zam_real_cold": dict(big=True, spec=dict(), graph="grammar", audio="wav:how_cold_is_it.wav"),
    # 30 s: the CMVN window (600 frames) slides, max-count prior rescaling of the iVector stats saturates, 123 nnet chunks
    "zam_long30": dict(big=True, spec=dict(), graph="grammar", audio="synth:20:480000"),
    "zam_s12005": dict(big=True, spec=dict(), graph="grammar", audio="synth:12005:472320"),
    # dither (feature-window.cc:90-98; on by default): digital silence -- the features are the reference's noise and nothing else --
    # a few-LSB signal, --dither=0 and a --dither other than 1; every other case runs with the stock default 1.0
    "tiny_silence": dict(spec=dict(), graph="grammar", audio="zeros:32000"),
    "tiny_quiet_u10": dict(spec=dict(seed=7), graph="grammar", audio="synth:10:40000:0.002"),
    "tiny_nodither_u11": dict(spec=dict(dither=0.0), graph="grammar", audio="synth:11:36000"),
    "tiny_dither05_u12": dict(spec=dict(dither=0.5, ivector_dim=0), graph="grammar", audio="synth:12:36000:0.01"),
    "zam_quiet_u13": dict(big=True, spec=dict(), graph="grammar", audio="synth:13:48000:0.003"),
    # decoder / decodable options that arrive through online.conf ONLY (the reference registers them on the parser that reads --config,
    # online2-wav-nnet3-latgen-faster.cc:131-137): --min-active binds on most frames of this graph, --frames-per-chunk moves the
    # streaming iVector schedule (and the dither seeds), --beam-delta the adaptive beam; max-active / beam are on the command line as rhasspy
    # passes them.  With --min-active / --beam-delta at their defaults, or --frames-per-chunk at 24, the reference's 5-best lists differ
    # (offline and streaming: checked with the pinned oracle when the case was chosen)
    "tiny_confopts_u15": dict(spec=dict(num_phones=40), graph="arpa:300:1500", audio="synth:15:44000",
                              opts=dict(max_active=60, beam=12.0), conf_opts={"min-active": 20, "frames-per-chunk": 30, "beam-delta": 0.25}),
    # --frame-subsampling-factor (decodable-simple-looped.h:56; every chain recipe's value is 3; rhasspy leaves it at 1): the network is
    # evaluated for t = 0, 3, 6, ... only, the decoder's frames are those, the chunk is rounded up to a multiple of the factor (20 -> 21)
    "tiny_fsf3_u16": dict(spec=dict(layer_offsets=((0,), (-1, 0, 1), (-1, 0, 1), (-3, 0, 3))), graph="grammar", audio="synth:16:48000",
                          conf_opts={"frame-subsampling-factor": 3}),
    "tiny_fsf3_chunk20_u17": dict(spec=dict(num_phones=40), graph="arpa:300:1500", audio="synth:17:44000",
                                  opts=dict(max_active=200, beam=14.0), conf_opts={"frame-subsampling-factor": 3, "frames-per-chunk": 20}),
    "tiny_fsf2_noiv_u18": dict(spec=dict(ivector_dim=0, seed=9), graph="grammar", audio="synth:18:40001",
                               conf_opts={"frame-subsampling-factor": 2}),
    "zam_fsf3_u19": dict(big=True, spec=dict(), graph="grammar", audio="synth:19:48000", conf_opts={"frame-subsampling-factor": 3}),
    # --frame-length=100: a 1600-sample window, padded to 2048 points -- the other FFT size the HIP front end is built for (model.cc)
    "tiny_win100_u21": dict(spec=dict(frame_length=100.0, seed=11), graph="grammar", audio="synth:21:48000"),
}
NBEST = 5


def case_spec(case: dict) -> synth.ModelSpec:
    return synth.ModelSpec(**case["spec"]) if case.get("big") else synth.tiny_spec(**case["spec"])


def case_audio(case: dict, golden_dir: Path = GOLDEN) -> np.ndarray:
    kind, *rest = case["audio"].split(":")
    if kind == "synth":
        pcm = synth.synth_utterance(int(rest[0]), int(rest[1]))
        if len(rest) > 2:      # scaled down to a few LSB
            pcm = np.round(pcm.astype(np.float64) * float(rest[2])).astype(np.int16)
        return pcm
    if kind == "zeros":
        return np.zeros(int(rest[0]), np.int16)
    with wave.open(str(golden_dir / "wav" / rest[0]), "rb") as w:
        assert w.getframerate() == 16000 and w.getnchannels() == 1 and w.getsampwidth() == 2
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()


def build_case_files(case: dict, root: Path, golden_dir: Path = GOLDEN):
    """Writes model + graph + wav for a case under `root`; returns (model_dir, graph_dir, wav_path, pcm)."""
    spec = case_spec(case)
    model_dir, graph_dir = root / "model", root / "graph"
    synth.write_model_dir(model_dir, spec)
    g = case["graph"].split(":")
    if g[0] == "grammar":
        rng = np.random.default_rng(11)
        sents = [s.split() for s in synth.DEFAULT_SENTENCES]
        lex = synth.make_lexicon(sents, spec, rng)
        fst = synth.make_grammar_hclg(sents, lex, spec, rng)
        synth.write_graph_dir(graph_dir, fst, lex, const=(len(g) < 2 or g[1] != "vector"))
    else:
        synth.make_arpa_graph(graph_dir, spec, extra_words=int(g[1]), num_random_sentences=int(g[2]))
    if case.get("conf_opts"):
        conf = model_dir / "model" / "online" / "conf" / "online.conf"
        conf.write_text(conf.read_text() + "".join(f"--{k}={v}\n" for k, v in case["conf_opts"].items()))
    pcm = case_audio(case, golden_dir)
    wav = root / "utt.wav"
    synth.write_wav(wav, pcm)
    return model_dir, graph_dir, wav, pcm


