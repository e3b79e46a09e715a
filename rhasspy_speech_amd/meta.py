"""Transcript post-processing shared by both transcribers.

`decode_meta` mirrors rhasspy_speech/hassil_fst.py:849-872: words of the form `__output:<BASE32>` carry a JSON
object {"text": ..., "list": ...} (slot value / list name), and `__sentence_output:<BASE32>` carries a Python
format string filled from the collected slots.  `int2sym` replaces the reference's extra
`utils/int2sym.pl -f 2- words.txt` subprocess (transcribe_wav.py:77-85).
"""
from __future__ import annotations

import base64
import json
import re
from pathlib import Path
from typing import Dict, List, Sequence, Union

OUTPUT_PREFIX = "__output:"
SENTENCE_OUTPUT = "__sentence_output:"


def decode_meta_single(text: str) -> str:
    return base64.b32decode(text.encode("utf-8")).strip().decode("utf-8")


def encode_meta(text: str, prefix: str = OUTPUT_PREFIX) -> str:
    return prefix + (base64.b32encode(text.encode("utf-8")).strip().decode("utf-8"))


def decode_meta(text: str) -> str:
    slots: Dict[str, str] = {}

    def handle_match(m) -> str:
        data = json.loads(decode_meta_single(m.group(1)))
        slot_name = data.get("list")
        slot_value = data["text"]
        if slot_name:
            slots[slot_name] = slot_value
        return slot_value

    text = re.sub(re.escape(OUTPUT_PREFIX) + "([0-9A-Z=]+)", handle_match, text)
    match = re.search(re.escape(SENTENCE_OUTPUT) + "([0-9A-Z=]+)", text)
    if match is None:
        return text
    sentence_output = decode_meta_single(match.group(1))
    return sentence_output.format(**slots)


def read_words_txt(path: Union[str, Path]) -> Dict[int, str]:
    table: Dict[int, str] = {}
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            parts = line.split()
            if len(parts) >= 2:
                table[int(parts[1])] = parts[0]
    return table


def int2sym(nbest_text: bytes, words: Dict[int, str]) -> str:
    """`int2sym.pl -f 2- words.txt`: field 1 (the key) is kept, every other field is mapped; an id missing
    from the table is an error there (the script dies) and here."""
    out: List[str] = []
    for line in nbest_text.decode().splitlines():
        parts = line.split()
        if not parts:
            continue
        mapped = [parts[0]]
        for p in parts[1:]:
            i = int(p)
            if i not in words:
                raise RuntimeError(f"int2sym: undefined symbol {i}")
            mapped.append(words[i])
        out.append(" ".join(mapped))
    return "\n".join(out) + ("\n" if out else "")


def texts_from_int2sym(int2sym_stdout: str) -> List[str]:
    """transcribe_wav.py:98-105: keep lines starting with "utt-", drop the key, drop empty hypotheses."""
    texts: List[str] = []
    for line in int2sym_stdout.splitlines():
        if line.startswith("utt-"):
            parts = line.strip().split(maxsplit=1)
            if len(parts) > 1:
                texts.append(decode_meta(parts[1]))
    return texts
