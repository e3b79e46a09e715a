"""Executables with the names and command lines of the two Kaldi decoder binaries the reference spawns, backed by the HIP
library ("optional stronger drop-in"): with `rhasspy_speech_amd/bin` first on PATH the UNMODIFIED
reference Python (`rhasspy_speech/transcribe_wav.py:46-74`, `transcribe_stream.py:53-99`) runs on the GPU path --
these programs write the binary CompactLattice table the Kaldi ones write (`rs_result_lattice`), and the reference's
`lattice-to-nbest | nbest-to-linear` read it as before.

  online2-wav-nnet3-latgen-faster [--opts] final.mdl HCLG.fst <spk2utt-rspecifier> <wav-rspecifier> <lattice-wspecifier>
      (online2bin/online2-wav-nnet3-latgen-faster.cc:196-300; every utterance of the table is one device batch)
  online2-cli-nnet3-decode-faster [--opts] final.mdl HCLG.fst words.txt <lattice-wspecifier>      (s16le PCM on stdin)
      (online2bin/online2-cli-nnet3-decode-faster.cc:129-170; key "utt")

Like the binaries, a process loads the model, decodes and exits; errors go to stderr with a non-zero status
(tools.py:138-145 turns that into RuntimeError).  No CPU fallback: without a GPU the library's error is the program's.
"""
from __future__ import annotations

import subprocess
import sys
import wave
from pathlib import Path
from typing import Dict, List, Tuple

import numpy as np

# option -> (rs_decode_opts field, type).  A field the command line does not set stays RS_OPT_UNSET: online.conf's value, else the
# reference's default -- ParseOptions reads --config first and the command line overrides it (util/parse-options.cc:328-345).
_DECODE_OPTS = {"max-active": ("max_active", int), "min-active": ("min_active", int), "beam": ("beam", float),
                "lattice-beam": ("lattice_beam", float), "acoustic-scale": ("acoustic_scale", float), "beam-delta": ("beam_delta", float),
                "frames-per-chunk": ("frames_per_chunk", int), "frame-subsampling-factor": ("frame_subsampling_factor", int)}
_UNSET_FIELDS = [f for f, _ in _DECODE_OPTS.values()]
# accepted on the command line when they ask for what the library does anyway (the reference's defaults / rhasspy's flags); any other
# value is an error, as it is in online.conf (csrc/engine.cc: ResolveDecoderOptions)
_FIXED = {"online": ("false", "f", "0"), "do-endpointing": ("false", "f", "0"), "extra-left-context-initial": ("0",), "prune-interval": ("25",),
          "determinize-lattice": ("true", "t", "1", "")}
_FIXED_BIT = {"online": 1, "do-endpointing": 2, "extra-left-context-initial": 4, "prune-interval": 8, "determinize-lattice": 16}      # RS_FIXED_*
_IGNORED = {"word-symbol-table", "chunk-length", "num-threads-startup", "verbose", "hash-ratio", "minimize", "phone-determinize",
            "word-determinize", "max-mem", "debug-computation"}


def parse_command_line(argv: List[str]) -> Tuple[Dict[str, object], str, List[str]]:
    """(decode options, --config path, positional arguments); unknown options are errors, as in Kaldi's ParseOptions."""
    opts: Dict[str, object] = {}
    config, pos = None, []
    for a in argv:
        if a.startswith("--"):
            name, _, value = a[2:].partition("=")
            if name == "config":
                config = value
            elif name in _DECODE_OPTS:
                field, conv = _DECODE_OPTS[name]
                opts[field] = conv(value)
            elif name in _FIXED:
                if value.lower() not in _FIXED[name]:
                    raise ValueError(f"--{name}={value} is not supported by the HIP path")
                # given on the command line: overrides online.conf's value (util/parse-options.cc:328-345), so the library must not
                # refuse the model for what the file says (rs_decode_opts.command_line_fixed)
                opts["command_line_fixed"] = int(opts.get("command_line_fixed", 0)) | _FIXED_BIT[name]
            elif name in _IGNORED:
                pass
            else:
                raise ValueError(f"invalid option --{name}")
        else:
            pos.append(a)
    if config is None:
        raise ValueError("--config=<online.conf> is required")
    return opts, config, pos


def read_table(rspecifier: str) -> List[Tuple[str, str]]:
    """Text `ark:` / `scp:` rspecifier -> [(key, rest of line)]; `cmd|` forms are run through the shell as Kaldi does."""
    kind, _, rest = rspecifier.partition(":")
    if kind.split(",")[0] not in ("ark", "scp"):
        raise ValueError(f"unsupported rspecifier {rspecifier!r}")
    rest = rest.strip()
    if rest.endswith("|"):
        text = subprocess.run(["bash", "-c", rest[:-1]], check=True, stdout=subprocess.PIPE).stdout.decode()
    elif rest == "-":
        text = sys.stdin.read()
    else:
        text = Path(rest).read_text()
    rows = []
    for line in text.splitlines():
        key, _, value = line.strip().partition(" ")
        if key:
            rows.append((key, value.strip()))
    return rows


def read_wav(path: str, model=None) -> np.ndarray:
    """WaveData::Read as the binary uses it (16-bit PCM only, wave-reader.cc:199-200; channel 0 of a multi-channel file,
    online2-wav-nnet3-latgen-faster.cc:216-218) -- the same reader the transcriber uses; with `model`, the header's sampling rate
    is checked against the model's like the binary's feature pipeline does (feat/online-feature.cc:86-101)."""
    from .transcribe_wav import read_wav_pcm16_rate
    try:
        pcm, rate = read_wav_pcm16_rate(path)
    except RuntimeError as e:
        raise ValueError(str(e)) from e
    if model is not None:
        model.check_sample_rate(rate)           # RsError: exit status 1 with Kaldi's message on stderr (main)
    return pcm


def open_wspecifier(wspecifier: str):
    kind, _, rest = wspecifier.partition(":")
    if kind.split(",")[0] != "ark" or "t" in kind.split(",")[1:]:
        raise ValueError(f"unsupported lattice wspecifier {wspecifier!r} (binary ark only)")
    return sys.stdout.buffer if rest == "-" else open(rest, "wb")


def _load(opts, config, final_mdl, hclg):
    from . import _lib
    unset = {f: _lib.RS_OPT_UNSET for f in _UNSET_FIELDS}
    return _lib, _lib.Model(final_mdl=final_mdl, hclg=hclg, online_conf=config, opts=_lib.default_opts(emit_lattice=1, **{**unset, "command_line_fixed": 0, **opts}))


def wav_main(argv: List[str]) -> int:
    opts, config, pos = parse_command_line(argv)
    if len(pos) != 5:
        raise ValueError("usage: online2-wav-nnet3-latgen-faster [options] <nnet3-in> <fst-in> <spk2utt-rspecifier> <wav-rspecifier> <lattice-wspecifier>")
    final_mdl, hclg, spk2utt, wav_rspec, lat_wspec = pos
    wavs = dict(read_table(wav_rspec))
    spk = read_table(spk2utt)
    # The binary carries the iVector estimator's adaptation state from one utterance of a speaker to the next
    # (online2-wav-nnet3-latgen-faster.cc:203-205, 287-288: Get / SetAdaptationState around every utterance).  rhasspy never does that
    # -- one process, one speaker, one utterance (transcribe_wav.py:45-75) -- and the library starts every utterance fresh, so a table
    # that asks for the carry is refused rather than decoded differently.
    for s, us in spk:
        if len(us.split()) > 1:
            raise ValueError(f"speaker {s} has {len(us.split())} utterances: the reference decodes them with the iVector adaptation state carried "
                             "from one to the next, which this library does not do (every utterance starts fresh); list one utterance per speaker")
    utts = [u for _, us in spk for u in us.split()]
    missing = [u for u in utts if u not in wavs]
    if missing:
        raise ValueError(f"no wav for utterance {missing[0]}")
    _lib, model = _load(opts, config, final_mdl, hclg)
    res = model.decode_batch([read_wav(wavs[u], model) for u in utts])
    out = open_wspecifier(lat_wspec)
    for i, u in enumerate(utts):
        out.write(res.lattice(i, u))
    out.flush()
    return 0


def cli_main(argv: List[str]) -> int:
    opts, config, pos = parse_command_line(argv)
    if len(pos) != 4:
        raise ValueError("usage: online2-cli-nnet3-decode-faster [options] <nnet3-in> <fst-in> <word-symbol-table> <lattice-wspecifier>")
    final_mdl, hclg, _words, lat_wspec = pos
    _lib, model = _load(opts, config, final_mdl, hclg)
    stream = _lib.Stream(model)
    carry = b""
    while True:
        chunk = sys.stdin.buffer.read(2048)            # the reference's read size (transcribe_stream.py:70-76 feeds what it gets)
        if not chunk:
            break
        chunk = carry + chunk
        n = len(chunk) // 2 * 2
        carry = chunk[n:]
        if n:
            stream.accept(np.frombuffer(chunk[:n], dtype=np.int16))
    res = stream.finish()
    out = open_wspecifier(lat_wspec)
    out.write(res.lattice(0, "utt"))
    out.flush()
    return 0


def run(main, argv: List[str]) -> int:
    try:
        return main(argv)
    except Exception as e:               # like KALDI_ERR: message on stderr, status 1
        print(f"ERROR ({Path(sys.argv[0]).name}): {e}", file=sys.stderr)
        return 1
