"""Counterpart of rhasspy_speech/tools.py for the HIP path.

The reference's KaldiTools locates directories of native binaries and runs them as subprocesses
(rhasspy_speech/tools.py:13-147).  Here the hot path is a shared library, so KaldiTools keeps the same
constructor / from_tools_dir signature (callers keep working) but only carries paths; nothing is spawned.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Union


@dataclass
class KaldiTools:
    kaldi_dir: Optional[Path] = None
    openfst_dir: Optional[Path] = None
    opengrm_dir: Optional[Path] = None
    phonetisaurus_bin: Optional[Path] = None

    @staticmethod
    def from_tools_dir(tools_dir: Union[str, Path]) -> "KaldiTools":
        tools_dir = Path(tools_dir).absolute()
        return KaldiTools(
            kaldi_dir=tools_dir / "kaldi",
            openfst_dir=tools_dir / "openfst",
            opengrm_dir=tools_dir / "opengrm",
            phonetisaurus_bin=tools_dir / "phonetisaurus",
        )

    @property
    def egs_utils_dir(self):
        return (self.kaldi_dir / "utils") if self.kaldi_dir else None

    @property
    def egs_steps_dir(self):
        return (self.kaldi_dir / "steps") if self.kaldi_dir else None
