"""Transcribe an audio stream on an MI355X.

Same public surface as rhasspy_speech/transcribe_stream.py:18-129: `KaldiNnet3StreamTranscriber.async_transcribe`
consumes an async iterable of raw s16le 16 kHz mono chunks (the bytes the reference writes to the stdin of
`online2-cli-nnet3-decode-faster`, :76-82) and returns the decoded texts.  The library re-chunks to the binary's
fixed 1024-sample ticks, so -- like the reference -- the result does not depend on how the caller slices the audio.
"""
from __future__ import annotations

import asyncio
import logging
from collections.abc import AsyncIterable
from pathlib import Path
from typing import List, Optional, Union

from . import _lib
from .meta import decode_meta, int2sym, read_words_txt, texts_from_int2sym
from .transcribe_util import get_fuzzy_text
from .tools import KaldiTools

_LOGGER = logging.getLogger(__name__)


class KaldiNnet3StreamTranscriber:
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

    def _ensure_loaded(self) -> _lib.Model:
        if self._model is None:
            opts = _lib.default_opts(max_active=self.max_active, lattice_beam=self.lattice_beam, beam=self.beam,
                                     acoustic_scale=1.0, device_id=self.device_id)
            self._model = _lib.Model(self.model_dir, self.graph_dir, opts)
            self._words = read_words_txt(self.graph_dir / "words.txt")
        return self._model

    @staticmethod
    def _accept_and_advance(stream, chunk) -> None:
        stream.accept(chunk)
        stream.advance()

    async def async_transcribe(
        self,
        audio_stream: AsyncIterable[Optional[bytes]],
        lang_dir: Union[str, Path],
        nbest: int = 1,
        max_fuzzy_cost: Optional[float] = None,
        require_fuzzy: bool = False,
    ) -> List[str]:
        lang_dir = Path(lang_dir)
        stream = _lib.Stream(self._ensure_loaded())
        loop = asyncio.get_running_loop()
        try:
            async for chunk in audio_stream:
                if chunk:
                    # the reference writes the chunk to the decoder's stdin and awaits the drain (transcribe_stream.py:73-76) while the
                    # decoder decodes as it reads; here: hand the samples over and let the device do what they make possible (MFCC,
                    # iVector, nnet chunks, search) -- in the executor, so the event loop is not held while the library plans and
                    # issues the advance (the calls release the GIL)
                    await loop.run_in_executor(None, self._accept_and_advance, stream, chunk)
            _LOGGER.debug("Stream ended")
            try:
                res = await loop.run_in_executor(None, stream.finish, nbest, self.acoustic_scale)
                nbest_stdout = res.text(0, "utt")
            except _lib.RsError as e:
                # The reference never checks the decoder's exit status (transcribe_stream.py:82) and then fails in
                # lattice-to-nbest on the missing lattice; surface the decoder's message instead.
                raise RuntimeError(f"Unexpected error running command online2-cli-nnet3-decode-faster (HIP): {e}") from e
        finally:
            stream.close()
        int2sym_stdout = int2sym(nbest_stdout, self._words)
        _LOGGER.debug("nbest: %s", int2sym_stdout)
        fuzzy_result = get_fuzzy_text(nbest_stdout, lang_dir)      # transcribe_stream.py:111-116
        if fuzzy_result is not None:
            text, cost = fuzzy_result
            _LOGGER.debug("Fuzzy cost: %s", cost)
            if cost <= max_fuzzy_cost:       # (like the reference, a TypeError when max_fuzzy_cost is None)
                return [decode_meta(text)]
        if require_fuzzy:
            return []
        return texts_from_int2sym(int2sym_stdout)

    # ---- rescoring path (transcribe_stream.py:131-274): stream through the old graph, re-rank the lattice with a NEW lexicon + LM
    async def async_transcribe_rescore(
        self,
        audio_stream: AsyncIterable[Optional[bytes]],
        old_lang_dir: Union[str, Path],
        new_lang_dir: Union[str, Path],
        nbest: int = 1,
        max_fuzzy_cost: Optional[float] = None,
        require_fuzzy: bool = False,
    ) -> List[str]:
        old_lang_dir, new_lang_dir = Path(old_lang_dir), Path(new_lang_dir)
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
                    raise ValueError("No value for disambiguation state (#0)") from e       # transcribe_stream.py:150-151
                raise
        stream = _lib.Stream(self._lat_model)
        loop = asyncio.get_running_loop()
        try:
            async for chunk in audio_stream:
                if chunk:
                    await loop.run_in_executor(None, self._accept_and_advance, stream, chunk)
            try:
                res = await loop.run_in_executor(None, stream.finish, 1, 1.0)
                nbest_stdout = self._rescorers[key].rescore(res, 0, nbest=nbest, acoustic_scale=self.acoustic_scale, key="utt")[0]
            except _lib.RsError as e:
                raise RuntimeError(f"Unexpected error running command online2-cli-nnet3-decode-faster (HIP): {e}") from e
        finally:
            stream.close()
        int2sym_stdout = int2sym(nbest_stdout, read_words_txt(new_lang_dir / "words.txt"))
        _LOGGER.debug("nbest: %s", int2sym_stdout)
        fuzzy_result = get_fuzzy_text(nbest_stdout, old_lang_dir)      # transcribe_stream.py:255-260
        if fuzzy_result is not None:
            text, cost = fuzzy_result
            if cost <= max_fuzzy_cost:
                return [decode_meta(text)]
        if require_fuzzy:
            return []
        return texts_from_int2sym(int2sym_stdout)
