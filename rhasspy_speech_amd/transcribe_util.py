"""`get_fuzzy_text` of the reference (rhasspy_speech/transcribe_util.py:11-88) on top of the library's host-side fuzzy
matcher (rs_fuzzy_*): same inputs (the n-best bytes, the language directory), same result -- `None` when there is no
`G.fuzzy.fst`, no path, or a path without words; otherwise (text, cost) with the words taken from `<lang_dir>/words.txt`
and the cost the reference sums from `fstprint` -- without the seven OpenFst processes per utterance."""
from __future__ import annotations

import threading
from pathlib import Path
from typing import Dict, Optional, Tuple, Union

from . import _lib
from .meta import read_words_txt

_cache: Dict[str, Tuple[float, "_lib.FuzzyMatcher", Dict[int, str]]] = {}
_lock = threading.Lock()


def _matcher(lang_dir: Path):
    fst = lang_dir / "G.fuzzy.fst"
    key, mtime = str(fst.resolve()), fst.stat().st_mtime
    with _lock:
        hit = _cache.get(key)
        if hit is None or hit[0] != mtime:      # parsed once per file version (the reference re-reads it on every call)
            hit = (mtime, _lib.FuzzyMatcher(fst), read_words_txt(lang_dir / "words.txt"))
            _cache[key] = hit
    return hit[1], hit[2]


def get_fuzzy_text(nbest_stdout: bytes, lang_dir: Union[str, Path]) -> Optional[Tuple[str, float]]:
    lang_dir = Path(lang_dir)
    if not (lang_dir / "G.fuzzy.fst").exists():
        return None
    matcher, words = _matcher(lang_dir)
    hit = matcher.match(nbest_stdout)
    if hit is None:
        return None
    ids, cost = hit
    return " ".join(words[i] for i in ids), cost
