"""Transcript post-processing shared by both transcribers.

`decode_meta` gives the results of rhasspy_speech/hassil_fst.py:849-872 (written from the format, checked against
tests/golden/python_api.json): words of the form `__output:<BASE32>` carry a JSON object {"text": ..., "list": ...}
(slot value / list name), and `__sentence_output:<BASE32>` carries a Python format string filled from the collected slots.  `int2sym` replaces the reference's extra
`utils/int2sym.pl -f 2- words.txt` subprocess (transcribe_wav.py:77-85).
"""
from __future__ import annotations

import base64
import json
from pathlib import Path
from typing import Dict, Iterator, List, Tuple, Union

OUTPUT_PREFIX = "__output:"
SENTENCE_OUTPUT = "__sentence_output:"


# Characters a payload may consist of: the RFC 4648 base32 alphabet is A-Z 2-7 with '=' padding; like the reference's pattern
# (hassil_fst.py:861,866) every digit is accepted, so that a malformed payload fails in the decoder rather than being cut short.
_PAYLOAD_CHARS = frozenset("ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789=")


def _payloads(text: str, marker: str) -> Iterator[Tuple[int, int, str]]:
    """(start, end, payload) of every `marker` + non-empty payload in `text`, left to right, non-overlapping."""
    at = text.find(marker)
    while at >= 0:
        first = at + len(marker)
        last = first
        while last < len(text) and text[last] in _PAYLOAD_CHARS:
            last += 1
        if last > first:
            yield at, last, text[first:last]
            at = text.find(marker, last)
        else:
            at = text.find(marker, at + 1)


def decode_meta_single(text: str) -> str:
    """One base32 payload -> the UTF-8 string it carries (surrounding whitespace of the decoded bytes dropped)."""
    raw = base64.b32decode(bytes(text, "utf-8"))
    return raw.strip().decode("utf-8")


def encode_meta(text: str, prefix: str = OUTPUT_PREFIX) -> str:
    """The word the trainer puts on an output arc for `text` (hassil_fst.py:875-876): `prefix` + base32 of its UTF-8 bytes."""
    return prefix + str(base64.b32encode(bytes(text, "utf-8")).strip(), "utf-8")


def decode_meta(text: str) -> str:
    """A transcript with meta words -> what the user sees (the reference's behaviour, hassil_fst.py:849-872).

    Every `__output:<payload>` word is a JSON record {"text": value, "list": slot name or null}: it is replaced by its value,
    and a named slot remembers it (the last one wins).  If a `__sentence_output:<payload>` word is present after that, the
    whole transcript becomes its payload, a str.format template over the slot names; otherwise the substituted text is it."""
    out: List[str] = []
    slots: Dict[str, str] = {}
    done = 0
    for start, end, payload in _payloads(text, OUTPUT_PREFIX):
        record = json.loads(decode_meta_single(payload))
        value = record["text"]
        if record.get("list"):
            slots[record["list"]] = value
        out.append(text[done:start])
        out.append(value)
        done = end
    out.append(text[done:])
    plain = "".join(out)
    for _, _, payload in _payloads(plain, SENTENCE_OUTPUT):
        return decode_meta_single(payload).format(**slots)
    return plain


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
