"""The graph-construction step of the reference's trainer on the library's in-process chain.

The reference's `KaldiTrainer._mkgraph` (rhasspy_speech/kaldi.py:409-425) runs
`bash utils/mkgraph.sh --self-loop-scale 1.0 <train_dir>/data/lang_<suffix> <model_dir>/model <train_dir>/graph_<suffix>`, a chain
of a dozen Kaldi / OpenFst processes; here the same directories go to `rs_mkgraph` (csrc/graph_build.cc, csrc/mkgraph.cc).  The
rest of the trainer (lexicon / G2P, prepare_lang.sh, grammar and ARPA compilation, prepare_online_decoding.sh) is outside the
hot path this package replaces: it produces the language directory this step reads.
"""
from __future__ import annotations

import asyncio
import logging
from pathlib import Path
from typing import Optional, Union

from . import _lib

_LOGGER = logging.getLogger(__name__)


class KaldiTrainer:
    """Directory layout and `_mkgraph` of the reference's trainer (kaldi.py:17-73, 409-425)."""

    def __init__(self, train_dir: Union[str, Path], model_dir: Union[str, Path]) -> None:
        self.train_dir = Path(train_dir).absolute()
        self.model_dir = Path(model_dir).absolute()

    def graph_dir(self, suffix: Optional[str] = None) -> Path:
        return self.train_dir / (f"graph_{suffix}" if suffix else "graph")

    @property
    def data_dir(self) -> Path:
        return self.train_dir / "data"

    def lang_dir(self, suffix: Optional[str] = None) -> Path:
        return self.data_dir / (f"lang_{suffix}" if suffix else "lang")

    async def _mkgraph(self, lang_type) -> None:
        """lang_type: a LangSuffix-like enum member or its string value ("grammar", "arpa", ...)."""
        suffix = getattr(lang_type, "value", lang_type)
        lang_dir = self.lang_dir(suffix)
        if not lang_dir.is_dir():
            _LOGGER.warning("Lang dir does not exist: %s", lang_dir)       # kaldi.py:411-413
            return
        loop = asyncio.get_running_loop()
        try:
            await loop.run_in_executor(None, mkgraph, lang_dir, self.model_dir / "model", self.graph_dir(suffix), 1.0)
        except _lib.RsError as e:
            # tools.py:82-90: a failing command surfaces as RuntimeError carrying its message
            raise RuntimeError(f"Unexpected error running command mkgraph.sh (in-process): {e}") from e


def mkgraph(lang_dir: Union[str, Path], model_dir: Union[str, Path], graph_dir: Union[str, Path], self_loop_scale: float = 0.1,
            transition_scale: float = 1.0) -> None:
    """utils/mkgraph.sh [--transition-scale T] [--self-loop-scale S] <lang-dir> <model-dir> <graphdir>; model-dir holds `tree` and
    `final.mdl`.  The script's defaults (tscale 1.0, loopscale 0.1); the reference passes --self-loop-scale 1.0."""
    _lib.mkgraph(lang_dir, model_dir, graph_dir, self_loop_scale=self_loop_scale, transition_scale=transition_scale)
