"""BASELINE.json configs[1..4] at their full sizes: one definition shared by bench.py, the GPU tests and
oracle/gen_config_golden.py (which runs the REFERENCE binaries on exactly these inputs and stores per-utterance
transcripts + costs under tests/golden/configs/).

Models, graphs and audio are regenerated from seeds wherever a config is used (numpy Generator streams are stable), so
the fixtures hold only the reference's outputs.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Tuple

import numpy as np

from rhasspy_speech_amd import synth

GOLDEN = Path(__file__).resolve().parent / "golden" / "configs"

N_SAMPLES_3S = 48000
N_SAMPLES_30S = 30 * 16000


def grammar_utterances(n_utts: int = 256, rank: int = 0) -> List[np.ndarray]:
    """configs[1] (the bench line): 256 synthetic 3 s utterances per rank."""
    return [synth.synth_utterance(rank * 100000 + u, N_SAMPLES_3S) for u in range(n_utts)]


def arpa_utterances(n_utts: int = 256) -> List[np.ndarray]:
    """configs[2]: 256 x 3 s on the ARPA-LM HCLG."""
    return [synth.synth_utterance(7000 + u, N_SAMPLES_3S) for u in range(n_utts)]


def stream_utterances(n_streams: int = 64) -> List[np.ndarray]:
    """configs[4]: 64 concurrent 30 s streams (lengths ragged by up to 49 frame shifts)."""
    rng = np.random.default_rng(4)
    return [synth.synth_utterance(12000 + i, N_SAMPLES_30S - 160 * int(rng.integers(0, 50))) for i in range(n_streams)]


# configs[3]: two independently seeded zamia-size model + graph sets standing in for de_DE-zamia / fr_FR-guyot
# (SURVEY.md section 8(d)), 512 utterances each, interleaved in one 1024-utterance batch.
MIXED_MODELS: Dict[str, dict] = {
    "de_DE-like": dict(model_seed=21, graph_seed=31, audio_base=30000),
    "fr_FR-like": dict(model_seed=22, graph_seed=32, audio_base=40000),
}


def mixed_utterances(n_per_model: int = 512) -> Tuple[List[str], List[np.ndarray]]:
    names, pcm = [], []
    for u in range(n_per_model):
        for name, m in MIXED_MODELS.items():
            names.append(name)
            pcm.append(synth.synth_utterance(m["audio_base"] + u, N_SAMPLES_3S - 320 * (u % 4)))
    return names, pcm


def build_grammar_model(root: Path, model_seed: int = 1, graph_seed: int = 11, conf_opts: dict = None) -> Tuple[Path, Path]:
    """conf_opts: decoder / decodable options appended to the model's online.conf (the reference reads them through --config)."""
    root = Path(root)
    model_dir, graph_dir = root / "model", root / "graph"
    if not (graph_dir / "HCLG.fst").exists():
        spec = synth.ModelSpec(seed=model_seed)
        synth.write_model_dir(model_dir, spec)
        synth.make_grammar_graph(graph_dir, spec, seed=graph_seed)
        if conf_opts:
            conf = model_dir / "model" / "online" / "conf" / "online.conf"
            conf.write_text(conf.read_text() + "".join(f"--{k}={v}\n" for k, v in conf_opts.items()))
    return model_dir, graph_dir


FSF3_CONF = {"frame-subsampling-factor": 3}      # c1_fsf3 / c4_fsf3: the headline model decoded the way a chain model is meant to be
N_FSF3_UTTS, N_FSF3_STREAMS = 128, 16


# "zamia-like-F" (the second net, "tdnn-f-like"): a full-size factorised TDNN -- the first hidden layer an
# ordinary affine + ReLU + BatchNorm, the six after it TdnnComponent bottlenecks 1024 -> 128 (no bias, offsets (-k, 0)) ->
# TdnnComponent 128 -> 1024 (offsets (0, k)) + ReLU + BatchNorm + dropout (identity at test time) with the residual
# Sum(Scale(0.66, previous), this) (nnet3/nnet-tdnn-component.cc:181-213; the xconfig tdnnf-layer), 2 000 pdfs, the headline's
# grammar graph shape.  The linear bottleneck outputs have neither ReLU nor BatchNorm behind them: the shape the split-fp16 layer
# GEMMs had never been pinned on at size.  c5_tdnnf: the first N_TDNNF_UTTS utterances of configs[1]; c5_tdnnf_fsf3: the first 32
# with --frame-subsampling-factor=3 (every layer above the (-1, 0, 1) ones on every third row, the residual sums included).
TDNNF_SPEC = dict(name="zamia-like-F", tdnnf=True, hidden_dim=1024, bottleneck_dim=128, seed=5)
N_TDNNF_UTTS, N_TDNNF_FSF3_UTTS, N_TDNNF_STREAMS = 64, 32, 16      # (c5_tdnnf_stream: the first 16 fed to the reference's streaming binary)


# A second factorised shape, the proportions of Kaldi's stock tdnn_f recipe (1536-wide layers, 160-wide bottlenecks): six column tiles
# per affine, and bottlenecks that are NOT a whole number of 128-column tiles (160 of 256 columns: the 256-column shapes at 37.5 %
# padding, where the 1024 / 128 model takes the 256 x 128 tile).  c6_tdnnf1536: the first N_TDNNF1536_UTTS utterances of configs[1].
TDNNF1536_SPEC = dict(name="zamia-like-F1536", tdnnf=True, hidden_dim=1536, bottleneck_dim=160, seed=6,
                      layer_offsets=((0,), (-1, 0, 1), (-1, 0, 1), (-3, 0, 3), (-3, 0, 3), (-3, 0, 3)))
N_TDNNF1536_UTTS = 24


def build_tdnnf_model(root: Path, conf_opts: dict = None, spec_kw: dict = None) -> Tuple[Path, Path]:
    root = Path(root)
    model_dir, graph_dir = root / "model", root / "graph"
    if not (graph_dir / "HCLG.fst").exists():
        spec = synth.ModelSpec(**(spec_kw or TDNNF_SPEC))
        synth.write_model_dir(model_dir, spec)
        synth.make_grammar_graph(graph_dir, spec, seed=11)
        if conf_opts:
            conf = model_dir / "model" / "online" / "conf" / "online.conf"
            conf.write_text(conf.read_text() + "".join(f"--{k}={v}\n" for k, v in conf_opts.items()))
    return model_dir, graph_dir


def build_arpa_model(root: Path) -> Tuple[Path, Path]:
    root = Path(root)
    model_dir, graph_dir = root / "model", root / "graph"
    if not (graph_dir / "HCLG.fst").exists():
        spec = synth.ModelSpec()
        synth.write_model_dir(model_dir, spec)
        synth.make_arpa_graph(graph_dir, spec, extra_words=600, num_random_sentences=4000)
    return model_dir, graph_dir


def load_golden(name: str):
    """-> (words: list of word-id lists, graph_cost, acoustic_cost) as the reference produced them."""
    g = np.load(GOLDEN / f"{name}.npz")
    off = g["word_offsets"]
    words = [g["words"][off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
    return words, g["graph_cost"], g["acoustic_cost"]


def load_golden_nbest(name: str):
    """-> per utterance: list of (word ids, graph cost, acoustic cost), best first, as `lattice-to-nbest --n=5 | nbest-to-linear`
    printed them for the reference's lattice (oracle/gen_config_golden.py)."""
    g = np.load(GOLDEN / f"{name}.npz")
    wo, uo = g["nbest_word_offsets"], g["nbest_utt_offsets"]
    out = []
    for u in range(len(uo) - 1):
        out.append([(g["nbest_words"][wo[k]:wo[k + 1]].tolist(), float(g["nbest_graph_cost"][k]), float(g["nbest_acoustic_cost"][k]))
                    for k in range(uo[u], uo[u + 1])])
    return out
