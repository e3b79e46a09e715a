"""Transcribe WAV files on an MI355X.

Same public surface as rhasspy_speech/transcribe_wav.py:15-105 (`KaldiNnet3WavTranscriber.__init__` and
`async_transcribe` keep their signatures, defaults and return type), but the three-process shell pipeline
`online2-wav-nnet3-latgen-faster | lattice-to-nbest | nbest-to-linear` is replaced by one call into
librhasspy_speech_hip.so, with the model and HCLG kept resident instead of being re-read per utterance.
`transcribe_many` is the batched entry point the GPU is built for.
"""
from __future__ import annotations

import asyncio
import logging
import wave
from pathlib import Path
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib
from .meta import decode_meta, int2sym, read_words_txt, texts_from_int2sym
from .transcribe_util import get_fuzzy_text
from .tools import KaldiTools

_LOGGER = logging.getLogger(__name__)


def read_wav_pcm16_rate(wav_path: Union[str, Path]):
    """WaveData::Read semantics (feat/wave-reader.cc): RIFF PCM 16-bit only, channel 0 is used.  -> (samples, sampling rate)"""
    with wave.open(str(wav_path), "rb") as w:
        if w.getsampwidth() != 2:
            raise RuntimeError(f"WaveData: can read only 16-bit PCM data, got {8 * w.getsampwidth()} bits: {wav_path}")
        data = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        ch, rate = w.getnchannels(), w.getframerate()
    return (np.ascontiguousarray(data.reshape(-1, ch)[:, 0]) if ch > 1 else data.copy()), rate


def read_wav_pcm16(wav_path: Union[str, Path]) -> np.ndarray:
    return read_wav_pcm16_rate(wav_path)[0]


class KaldiNnet3WavTranscriber:
    def __init__(
        self,
        model_dir: Union[str, Path],
        graph_dir: Union[str, Path],
        tools: Optional[KaldiTools] = None,
        max_active: int = 7000,
        lattice_beam: float = 8.0,
        acoustic_scale: float = 1.0,
        beam: float = 24.0,
        device_id: int = 0,
    ):
        self.model_dir = Path(model_dir)
        self.graph_dir = Path(graph_dir)
        self.tools = tools
        self.max_active = max_active
        self.lattice_beam = lattice_beam
        self.acoustic_scale = acoustic_scale
        self.beam = beam
        self.device_id = device_id
        self._model: Optional[_lib.Model] = None
        self._words = None
        self._lat_model: Optional[_lib.Model] = None      # the same files, results keep their lattices (rescoring path)
        self._rescorers = {}

    # the reference reloads everything per call; here it is loaded once, lazily
    def _ensure_loaded(self) -> _lib.Model:
        if self._model is None:
            opts = _lib.default_opts(max_active=self.max_active, lattice_beam=self.lattice_beam, beam=self.beam,
                                     acoustic_scale=1.0, device_id=self.device_id)   # "--acoustic-scale=1.0" is hard-coded upstream
            self._model = _lib.Model(self.model_dir, self.graph_dir, opts)
            self._words = read_words_txt(self.graph_dir / "words.txt")
        return self._model

    def _read_wav(self, wav_path) -> np.ndarray:
        """The file's samples, after the check the reference's binary makes on its header: a wav whose rate is not the model's
        --sample-frequency ends `online2-wav-nnet3-latgen-faster` with "Sampling frequency mismatch, expected 16000, got ..."
        (feat/online-feature.cc:86-101) and the Python with RuntimeError (tools.py:138-145)."""
        pcm, rate = read_wav_pcm16_rate(wav_path)
        try:
            self._ensure_loaded().check_sample_rate(rate)
        except _lib.RsError as e:
            raise RuntimeError(f"Unexpected error running command online2-wav-nnet3-latgen-faster (HIP): {e}") from e
        return pcm

    def _nbest_stdout(self, pcm_batch: Sequence[np.ndarray], nbest: int) -> List[bytes]:
        """One `nbest-to-linear ... ark,t:-` byte string per utterance (key "utt" like the reference)."""
        try:
            res = self._ensure_loaded().decode_batch(pcm_batch, nbest=nbest, lattice_acoustic_scale=self.acoustic_scale)
            return [res.text(u, "utt") for u in range(len(pcm_batch))]
        except _lib.RsError as e:
            # tools.py:138-145: non-zero exit status -> RuntimeError carrying the tool's stderr
            raise RuntimeError(f"Unexpected error running command online2-wav-nnet3-latgen-faster (HIP): {e}") from e

    async def async_transcribe(
        self,
        wav_path: Union[str, Path],
        lang_dir: Union[str, Path],
        nbest: int = 1,
        max_fuzzy_cost: Optional[float] = None,
        require_fuzzy: bool = False,
    ) -> List[str]:
        pcm = self._read_wav(wav_path)
        loop = asyncio.get_running_loop()
        nbest_stdout = (await loop.run_in_executor(None, self._nbest_stdout, [pcm], nbest))[0]
        return self._finish(nbest_stdout, Path(lang_dir), max_fuzzy_cost, require_fuzzy)

    def transcribe(self, wav_path, lang_dir, nbest: int = 1, max_fuzzy_cost=None, require_fuzzy: bool = False) -> List[str]:
        return self._finish(self._nbest_stdout([self._read_wav(wav_path)], nbest)[0], Path(lang_dir), max_fuzzy_cost, require_fuzzy)

    def transcribe_many(self, wav_paths: Sequence[Union[str, Path]], lang_dir, nbest: int = 1, max_fuzzy_cost=None,
                        require_fuzzy: bool = False) -> List[List[str]]:
        """Batched: all files decoded in one device pass."""
        outs = self._nbest_stdout([self._read_wav(p) for p in wav_paths], nbest)
        return [self._finish(o, Path(lang_dir), max_fuzzy_cost, require_fuzzy) for o in outs]

    # ---- rescoring path (transcribe_wav.py:107-232): decode with the old graph, re-rank the lattice with a NEW lexicon + LM
    def _rescored_stdout(self, pcm: np.ndarray, new_lang_dir: Path, nbest: int) -> bytes:
        if self._lat_model is None:
            opts = _lib.default_opts(max_active=self.max_active, lattice_beam=self.lattice_beam, beam=self.beam, acoustic_scale=1.0,
                                     device_id=self.device_id, emit_lattice=1)
            self._lat_model = _lib.Model(self.model_dir, self.graph_dir, opts)
        key = str(new_lang_dir)
        if key not in self._rescorers:
            try:
                self._rescorers[key] = _lib.Rescorer(self._lat_model, new_lang_dir)
            except _lib.RsError as e:
                if "No value for disambiguation state" in str(e):
                    raise ValueError("No value for disambiguation state (#0)") from e       # transcribe_wav.py:128-129
                raise
        try:
            res = self._lat_model.decode_batch([pcm], nbest=1)
            return self._rescorers[key].rescore(res, 0, nbest=nbest, acoustic_scale=self.acoustic_scale, key="utt")[0]
        except _lib.RsError as e:
            raise RuntimeError(f"Unexpected error running command online2-wav-nnet3-latgen-faster (HIP): {e}") from e

    async def async_transcribe_rescore(
        self,
        wav_path: Union[str, Path],
        old_lang_dir: Union[str, Path],
        new_lang_dir: Union[str, Path],
        nbest: int = 1,
        max_fuzzy_cost: Optional[float] = None,
        require_fuzzy: bool = False,
    ) -> List[str]:
        pcm = self._read_wav(wav_path)
        loop = asyncio.get_running_loop()
        nbest_stdout = await loop.run_in_executor(None, self._rescored_stdout, pcm, Path(new_lang_dir), nbest)
        # ids -> words with the NEW table, fuzzy matching against the OLD language directory (transcribe_wav.py:204-218)
        return self._finish(nbest_stdout, Path(old_lang_dir), max_fuzzy_cost, require_fuzzy, read_words_txt(Path(new_lang_dir) / "words.txt"))

    def _finish(self, nbest_stdout: bytes, lang_dir: Path, max_fuzzy_cost, require_fuzzy: bool, words=None) -> List[str]:
        int2sym_stdout = int2sym(nbest_stdout, self._words if words is None else words)
        _LOGGER.debug("nbest: %s", int2sym_stdout)
        fuzzy_result = get_fuzzy_text(nbest_stdout, lang_dir)      # transcribe_wav.py:87-92
        if fuzzy_result is not None:
            text, cost = fuzzy_result
            _LOGGER.debug("Fuzzy cost: %s", cost)
            if cost <= max_fuzzy_cost:       # (like the reference, a TypeError when max_fuzzy_cost is None)
                return [decode_meta(text)]
        if require_fuzzy:
            return []
        return texts_from_int2sym(int2sym_stdout)
