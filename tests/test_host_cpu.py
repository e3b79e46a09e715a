"""CPU-only tests: the C-ABI library loads and exports what include/rhasspy_speech_hip.h declares, host-side model
parsing (no compute without a GPU), the Python layer against goldens captured from the reference's own Python,
and the multi-rank sharding/gather logic over gloo."""
import json
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests import cases

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from rhasspy_speech_amd import _lib
    header = (ROOT / "include" / "rhasspy_speech_hip.h").read_text()
    declared = set(re.findall(r"\b(rs_[a-z0-9_]+)\s*\(", header))
    lib = _lib.load_library()
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_default_opts_are_the_reference_flags():
    from rhasspy_speech_amd import _lib
    o = _lib.default_opts()
    # what rhasspy passes on the command line (transcribe_wav.py:46-55) is set; the rest is left to online.conf / the reference's defaults
    assert (o.beam, o.max_active, o.lattice_beam, o.acoustic_scale) == (24.0, 7000, 8.0, 1.0)
    assert (o.min_active, o.beam_delta, o.frames_per_chunk, o.frame_subsampling_factor) == (_lib.RS_OPT_UNSET,) * 4
    assert (o.emit_lattice, o.prune_output_pdfs, o.keep_intermediates) == (0, 1, 0)      # (pruned output layer: same words and costs)


def _with_online_conf(case_cache, tmp_path, extra_lines):
    """A copy of the tiny case's model directory whose online.conf has `extra_lines` appended."""
    import shutil
    model_dir, graph_dir, _, _ = case_cache("tiny_u0")
    dst = tmp_path / "model_dir"
    shutil.copytree(model_dir, dst)
    conf = dst / "model" / "online" / "conf" / "online.conf"
    text = conf.read_text().replace(str(model_dir), str(dst))
    conf.write_text(text + "".join(l + "\n" for l in extra_lines))
    for sub in conf.parent.glob("*.conf"):       # (the other config files name files of the directory by absolute path too)
        sub.write_text(sub.read_text().replace(str(model_dir), str(dst)))
    return dst, graph_dir


def _decoder_opts(describe: str) -> dict:
    line = [l for l in describe.splitlines() if l.startswith("decoder_opts:")][0]
    return dict(kv.split("=") for kv in line.split()[1:])


def test_decoder_options_of_online_conf_take_effect(case_cache, tmp_path):
    """The reference registers decoder / decodable options on the parser that reads --config (online2-wav-nnet3-latgen-faster.cc:131-137),
    config file first, command line overriding (util/parse-options.cc:328-345): a model directory whose online.conf sets them decodes
    with them unless rs_decode_opts (= the command line) sets the field."""
    from rhasspy_speech_amd import _lib
    plain = _decoder_opts(_lib.Model(*case_cache("tiny_u0")[:2]).describe())
    assert plain == dict(beam="24", max_active="7000", min_active="200", lattice_beam="8", beam_delta="0.5", acoustic_scale="1",
                         frames_per_chunk="24", frame_subsampling_factor="1")
    md, gd = _with_online_conf(case_cache, tmp_path, ["--min-active=17", "--beam-delta=0.25", "--frames-per-chunk=30", "--beam=13.0",
                                                      "--max-active=99", "--lattice-beam=5.5", "--acoustic-scale=0.7",
                                                      "--frame-subsampling-factor=1", "--extra-left-context-initial=0", "--prune-interval=25",
                                                      "--determinize-lattice=true", "--hash-ratio=3.0", "--minimize=false", "--max-mem=50000000"])
    got = _decoder_opts(_lib.Model(md, gd).describe())
    # not on rhasspy's command line: the file's values; on it (beam, max-active, lattice-beam, acoustic-scale): the command line's
    assert got == dict(plain, min_active="17", beam_delta="0.25", frames_per_chunk="30")
    # nothing on the command line: everything from the file
    unset = _lib.default_opts(beam=_lib.RS_OPT_UNSET, max_active=_lib.RS_OPT_UNSET, lattice_beam=_lib.RS_OPT_UNSET, acoustic_scale=_lib.RS_OPT_UNSET)
    got = _decoder_opts(_lib.Model(md, gd, unset).describe())
    assert got == dict(beam="13", max_active="99", min_active="17", lattice_beam="5.5", beam_delta="0.25", acoustic_scale="0.7",
                       frames_per_chunk="30", frame_subsampling_factor="1")
    # ... and the command line over the file
    got = _decoder_opts(_lib.Model(md, gd, _lib.default_opts(min_active=3, beam_delta=0.125, frames_per_chunk=18)).describe())
    assert (got["min_active"], got["beam_delta"], got["frames_per_chunk"]) == ("3", "0.125", "18")
    # no file value, nothing on the command line: the reference's defaults (lattice-faster-decoder.h:56-63, decodable-simple-looped.h:55-59)
    got = _decoder_opts(_lib.Model(*case_cache("tiny_u0")[:2], unset).describe())
    assert got == dict(beam="16", max_active=str(2**31 - 1), min_active="200", lattice_beam="10", beam_delta="0.5", acoustic_scale="0.1",
                       frames_per_chunk="24", frame_subsampling_factor="1")


@pytest.mark.parametrize("line,needle", [
    ("--frame-subsampling-factor=0", "KALDI_ASSERT: at Check:decodable-simple-looped.h:62"),
    ("--extra-left-context-initial=4", "--extra-left-context-initial=4 is not supported"),
    ("--prune-interval=10", "--prune-interval=10 is not supported"),
    ("--determinize-lattice=false", "--determinize-lattice=false is not supported"),
    ("--online=true", "--online=true is not supported"),
    ("--do-endpointing=true", "--do-endpointing=true is not supported"),
    ("--min-active=abc", 'Invalid integer option "abc"'),
    ("--beam-delta=x1", 'Invalid floating-point option "x1"'),
    ("--determinize-lattice=maybe", "Invalid format for boolean argument [expected true or false]: maybe"),
    ("--min-active=8000", "KALDI_ASSERT: at Check:lattice-faster-decoder.h:87"),       # min_active <= max_active (7000)
    ("--beam-delta=0", "KALDI_ASSERT: at Check:lattice-faster-decoder.h:87"),
    ("--frames-per-chunk=0", "KALDI_ASSERT: at Check:decodable-simple-looped.h:62"),
    ("--hash-ratio=0.5", "hash_ratio >= 1.0"),
    ("--no-such-option=1", "Invalid option --no-such-option=1"),
])
def test_online_conf_options_the_kernels_cannot_honour_fail_the_load(case_cache, tmp_path, line, needle):
    """... and what cannot be honoured is refused with a message, never dropped (round 4 accepted and ignored these)."""
    from rhasspy_speech_amd import _lib
    md, gd = _with_online_conf(case_cache, tmp_path, [line])
    with pytest.raises(_lib.RsError) as ei:
        _lib.Model(md, gd, _lib.default_opts(command_line_fixed=0))       # (a command line that does not repeat the option)
    assert needle in str(ei.value), str(ei.value)


def test_command_line_overrides_unsupported_values_in_online_conf(case_cache, tmp_path):
    """ParseOptions reads --config first and the command line overrides it (util/parse-options.cc:328-345): the reference runs with
    `--online=false --do-endpointing=false` on its command line (transcribe_wav.py:48-49) whatever online.conf says, so a model
    directory whose online.conf carries --online=true / --do-endpointing=true loads with rs_default_opts (= rhasspy's command line);
    an option the command line did not fix is still refused, and the shim passes down exactly what its argv holds."""
    from rhasspy_speech_amd import _lib, kaldi_cli
    md, gd = _with_online_conf(case_cache, tmp_path, ["--online=true", "--do-endpointing=true"])
    assert _lib.default_opts().command_line_fixed == _lib.FIXED_ONLINE | _lib.FIXED_DO_ENDPOINTING
    _lib.Model(md, gd)
    with pytest.raises(_lib.RsError, match="--online=true is not supported"):
        _lib.Model(md, gd, _lib.default_opts(command_line_fixed=_lib.FIXED_DO_ENDPOINTING))
    md2, gd2 = _with_online_conf(case_cache, tmp_path / "b", ["--prune-interval=10", "--determinize-lattice=false"])
    with pytest.raises(_lib.RsError, match="--prune-interval=10 is not supported"):
        _lib.Model(md2, gd2)
    _lib.Model(md2, gd2, _lib.default_opts(command_line_fixed=_lib.FIXED_PRUNE_INTERVAL | _lib.FIXED_DETERMINIZE_LATTICE))
    opts, _, _ = kaldi_cli.parse_command_line(["--config=x", "--online=false", "--prune-interval=25", "a"])
    assert opts["command_line_fixed"] == _lib.FIXED_ONLINE | _lib.FIXED_PRUNE_INTERVAL
    assert "command_line_fixed" not in kaldi_cli.parse_command_line(["--config=x", "--beam=13", "a"])[0]


def test_sampling_rate_is_checked_like_the_reference(case_cache, tmp_path):
    """feat/online-feature.cc:86-101: a waveform whose rate is not the model's --sample-frequency is an error with Kaldi's text (the
    reference's binary exits with status 1, its Python raises RuntimeError, tools.py:138-145) -- not a decode of garbage; and
    --allow-downsample (the reference then resamples) is refused with a message, not dropped."""
    import wave
    from rhasspy_speech_amd import _lib, transcribe_wav
    md, gd, wav, pcm = case_cache("tiny_u0")
    model = _lib.Model(md, gd)
    model.check_sample_rate(16000)
    with pytest.raises(_lib.RsError, match=r"Sampling frequency mismatch, expected 16000, got 8000\nPerhaps you want to use the options --allow_\{upsample,downsample\}"):
        model.check_sample_rate(8000)
    w8 = tmp_path / "u8k.wav"
    with wave.open(str(w8), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
        w.writeframes(pcm[::2].tobytes())
    assert transcribe_wav.read_wav_pcm16_rate(w8)[1] == 8000
    tr = transcribe_wav.KaldiNnet3WavTranscriber(md, gd)
    with pytest.raises(RuntimeError, match="Sampling frequency mismatch, expected 16000, got 8000"):
        tr.transcribe(w8, gd)
    md2, gd2 = _with_online_conf(case_cache, tmp_path / "c", [])
    mfcc = md2 / "model" / "online" / "conf" / "mfcc.conf"
    mfcc.write_text(mfcc.read_text() + "--allow-downsample=true\n")
    m2 = _lib.Model(md2, gd2)
    with pytest.raises(_lib.RsError, match="does not resample"):
        m2.check_sample_rate(44100)
    with pytest.raises(_lib.RsError, match="Perhaps you want to use the options"):
        m2.check_sample_rate(8000)             # (--allow-upsample was not given)


def test_frame_subsampling_factor_rounds_the_chunk_like_the_reference(case_cache, tmp_path):
    """--frame-subsampling-factor from online.conf is taken (round 5: refused) and the chunk becomes the advised size rounded up to a
    multiple of it (GetChunkSize, nnet-compile-looped.cc:81-94); an opts value overrides the file's."""
    from rhasspy_speech_amd import _lib
    md, gd = _with_online_conf(case_cache, tmp_path, ["--frame-subsampling-factor=3", "--frames-per-chunk=20"])
    got = _decoder_opts(_lib.Model(md, gd).describe())
    assert (got["frame_subsampling_factor"], got["frames_per_chunk"]) == ("3", "21")
    got = _decoder_opts(_lib.Model(md, gd, _lib.default_opts(frame_subsampling_factor=2)).describe())
    assert (got["frame_subsampling_factor"], got["frames_per_chunk"]) == ("2", "20")


def test_frame_subsampling_evaluates_the_upper_layers_on_every_third_row(case_cache):
    """Like the reference's compiler (the looped request asks for t = 0, 3, 6, ...: nnet-compile-looped.cc:111-128), the layer plan
    evaluates a layer only where something reads it: with offsets (0) (-1,0,1) (-1,0,1) (-3,0,3) x 4 everything from the third
    layer up is read at multiples of three only; with factor 1 every layer is dense, with factor 2 all but the last three."""
    from rhasspy_speech_amd import _lib
    md, gd = case_cache("zam_fsf3_u19")[:2]
    ops = [l for l in _lib.Model(md, gd).describe().splitlines() if l.startswith("op: gemm")]
    strided = ["rows=every-3" in l for l in ops]
    assert strided == [False, False] + [True] * (len(ops) - 2), ops
    ops1 = [l for l in _lib.Model(md, gd, _lib.default_opts(frame_subsampling_factor=1)).describe().splitlines() if l.startswith("op: gemm")]
    assert not any("rows=every" in l for l in ops1)
    ops2 = [l for l in _lib.Model(md, gd, _lib.default_opts(frame_subsampling_factor=2)).describe().splitlines() if l.startswith("op: gemm")]
    # (the last (-3,0,3) layer reads its input at odd and even rows; its own output, the pre-final and the output layer are read at even rows)
    assert [("rows=every-2" in l) for l in ops2] == [False] * (len(ops2) - 3) + [True, True, True], ops2


@pytest.mark.parametrize("name", ["tiny_u0", "tiny_text_u1", "tinyf_u5", "tiny_noiv_u2", "tiny_hmm_u6", "tiny_vecfst_u9", "tiny_arpa_u7"])
def test_model_parsing_host_side(case_cache, name):
    from rhasspy_speech_amd import _lib
    model_dir, graph_dir, _, _ = case_cache(name)
    m = _lib.Model(model_dir, graph_dir)
    d = m.describe()
    spec = cases.case_spec(cases.CASES[name])
    assert f"output_dim={spec.num_pdfs}" in d and f"pdfs={spec.num_pdfs}" in d
    assert f"ceps={spec.num_ceps}" in d
    if spec.ivector_dim:
        assert f"ivector: dim={spec.ivector_dim} gauss={spec.num_gauss}" in d
    else:
        assert "ivector: none" in d
    # relu + batchnorm are fused into their affine layer's GEMM epilogue
    assert "tdnn1.affine+tdnn1.relu+tdnn1.batchnorm" in d
    # the xent branch of chain models is not on the path
    assert "xent" not in d


def test_text_and_binary_models_parse_identically(case_cache):
    from rhasspy_speech_amd import _lib
    a = _lib.Model(*case_cache("tiny_u0")[:2]).describe()
    b = _lib.Model(*case_cache("tiny_text_u1")[:2]).describe()
    assert a == b


def test_no_gpu_means_loud_failure_not_fallback(case_cache):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rhasspy_speech_amd import _lib
    m = _lib.Model(*case_cache("tiny_u0")[:2])
    with pytest.raises(_lib.RsError) as ei:
        m.decode_batch([np.zeros(16000, np.int16)])
    assert ei.value.status == _lib.RS_ERR_DEVICE and "no CPU fallback" in str(ei.value)


def test_new_entry_points_validate_their_arguments():
    """rs_streams_accept / rs_shard_gather (round 3) refuse bad arguments before touching a device: status RS_ERR_ARG and a message."""
    import ctypes as C
    from rhasspy_speech_amd import _lib
    lib = _lib.lib()
    assert lib.rs_streams_accept(None, None, None, 0) == 0                       # nothing to hand over is not an error
    assert lib.rs_streams_accept(None, None, None, 3) == _lib.RS_ERR_ARG
    assert b"rs_streams_accept" in lib.rs_last_error()
    rec = (C.c_int32 * (2 * _lib.SHARD_RECORD_INTS))()
    for args in ((0, 2, 0, 1, None, rec),            # no communicator
                 (0, 2, 1, 1, C.c_void_p(1), rec),   # rank outside the world
                 (-1, 2, 0, 1, C.c_void_p(1), rec),  # no device
                 (0, 2, 0, 1, C.c_void_p(1), None)): # no records
        assert lib.rs_shard_gather(*args) == _lib.RS_ERR_ARG, args
        assert b"rs_shard_gather" in lib.rs_last_error()


def test_bad_model_files_fail_like_kaldi(tmp_path, case_cache):
    from rhasspy_speech_amd import _lib
    model_dir, graph_dir, _, _ = case_cache("tiny_u0")
    with pytest.raises(_lib.RsError) as ei:
        _lib.Model(tmp_path / "nope", graph_dir)
    assert ei.value.status == _lib.RS_ERR_MODEL
    bad = tmp_path / "HCLG.fst"
    bad.write_bytes(b"not an fst at all")
    with pytest.raises(_lib.RsError) as ei:
        _lib.Model(final_mdl=model_dir / "model/model/final.mdl", hclg=bad, online_conf=model_dir / "model/online/conf/online.conf")
    assert "Bad FST header" in str(ei.value)
    conf = tmp_path / "online.conf"
    conf.write_text("--feature-type=mfcc\n--no-such-option=1\n")
    with pytest.raises(_lib.RsError) as ei:
        _lib.Model(final_mdl=model_dir / "model/model/final.mdl", hclg=graph_dir / "HCLG.fst", online_conf=conf)
    assert "Invalid option" in str(ei.value)


def test_python_layer_matches_reference_python():
    from rhasspy_speech_amd import meta
    g = json.loads((cases.GOLDEN / "python_api.json").read_text())
    words = {i: w for i, w in enumerate(g["words"])}
    for c in g["cases"]:
        got = meta.texts_from_int2sym(meta.int2sym(c["nbest_stdout"].encode(), words))
        assert got == c["texts"], c
    assert meta.encode_meta(g["encode_meta"]["input"]) == g["encode_meta"]["output"]
    assert meta.decode_meta_single(g["encode_meta"]["output"].split(":", 1)[1]) == g["encode_meta"]["input"]


def test_meta_words_decode_like_the_reference():
    """200 random transcripts with `__output:` / `__sentence_output:` words (oracle/gen_meta_golden.py: the reference's own
    decode_meta on each, including the ones it raises on: a template naming a slot the transcript did not fill)."""
    from rhasspy_speech_amd import meta
    g = json.loads((cases.GOLDEN / "meta_vectors.json").read_text())
    assert len(g["decode_meta"]) == 200
    for v in g["decode_meta"]:
        if "raises" in v:
            with pytest.raises(Exception) as ei:
                meta.decode_meta(v["input"])
            assert type(ei.value).__name__ == v["raises"], v
        else:
            assert meta.decode_meta(v["input"]) == v["output"], v
    for v in g["encode_meta"]:
        assert meta.encode_meta(v["input"]) == v["output"]


def test_transcriber_signature_matches_reference():
    import inspect
    from rhasspy_speech_amd import KaldiNnet3WavTranscriber
    sig = inspect.signature(KaldiNnet3WavTranscriber.__init__)
    p = sig.parameters
    assert [p[k].default for k in ("max_active", "lattice_beam", "acoustic_scale", "beam")] == [7000, 8.0, 1.0, 24.0]
    sig = inspect.signature(KaldiNnet3WavTranscriber.async_transcribe)
    assert list(sig.parameters)[1:] == ["wav_path", "lang_dir", "nbest", "max_fuzzy_cost", "require_fuzzy"]
    assert inspect.iscoroutinefunction(KaldiNnet3WavTranscriber.async_transcribe)


def test_stream_transcriber_signature_matches_reference():
    import inspect
    from rhasspy_speech_amd.transcribe_stream import KaldiNnet3StreamTranscriber
    p = inspect.signature(KaldiNnet3StreamTranscriber.__init__).parameters
    assert [p[k].default for k in ("max_active", "lattice_beam", "acoustic_scale", "beam")] == [7000, 8.0, 1.0, 24.0]
    sig = inspect.signature(KaldiNnet3StreamTranscriber.async_transcribe)
    assert list(sig.parameters)[1:] == ["audio_stream", "lang_dir", "nbest", "max_fuzzy_cost", "require_fuzzy"]


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from rhasspy_speech_amd import shard
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world, n = dist.get_rank(), dist.get_world_size(), 11
idx = shard.shard_indices(n, rank, world)
words = [[i, i + 1, 7 * i % 13] [: 1 + i % 3] for i in idx]          # stand-in for the decoder's output
costs = [(float(i) + 0.5, -float(i) * 2.0) for i in idx]
got = shard.unpack_records(shard.gather_records(shard.pack_records(idx, words, costs), n), n)
assert sorted(got) == list(range(n)), got
for i in range(n):
    assert got[i] == ([i, i + 1, 7 * i % 13][: 1 + i % 3], float(i) + 0.5, -float(i) * 2.0), (i, got[i])
# a mixed-model batch: two stand-in "models" (the real ones need a GPU), utterance i belongs to model names[i % 3 == 0]
class Fake:
    def __init__(self, tag): self.tag, self.calls = tag, 0
    def decode_batch(self, pcm):
        self.calls += 1
        fake = self
        class R:
            def words(self, u): return [fake.tag, int(pcm[u][0]), len(pcm[u])]
            def costs(self, u): return (float(pcm[u][0]) / 4.0, -float(len(pcm[u])))
        return R()
models = {"de": Fake(1), "fr": Fake(2)}
n = 13
names = ["de" if i % 3 == 0 else "fr" for i in range(n)]
pcm = [np.full(5 + i, 100 + i, np.int16) for i in range(n)]
got = shard.decode_mixed_sharded(models, names, pcm, rank, world)
assert sorted(got) == list(range(n))
for i in range(n):
    assert got[i] == ([1 if i % 3 == 0 else 2, 100 + i, 5 + i], (100 + i) / 4.0, -float(5 + i)), (i, got[i])
assert models["de"].calls <= 1 and models["fr"].calls <= 1          # one batch per model and rank
try:
    shard.decode_mixed_sharded(models, ["es"] * n, pcm, rank, world)
    raise SystemExit("an unknown model name must be an error")
except KeyError:
    pass
# a hypothesis longer than the record: cut to MAX_WORDS ids and FLAGGED, not silently shortened
class Long(Fake):
    def decode_batch(self, pcm):
        class R:
            def words(self, u): return list(range(100 + u))
            def costs(self, u): return (1.0, 2.0)
        return R()
got = shard.decode_mixed_sharded({"de": Long(0)}, ["de"] * 4, pcm[:4], rank, world)
assert got.truncated == {0, 1, 2, 3} and got.num_words[3] == 101 and got[2][0] == list(range(shard.MAX_WORDS))
# one utterance fails on one rank (the reference's "decoded no frames"): reported per utterance on every rank, nobody hangs
class OneBad(Fake):
    def decode_batch(self, pcm):
        inner = super().decode_batch(pcm)
        class R:
            def words(self, u):
                if len(pcm[u]) == 5 + 3:
                    e = RuntimeError("You cannot get a lattice if you decoded no frames."); e.status = -4; raise e
                return inner.words(u)
            def costs(self, u): return inner.costs(u)
        return R()
got = shard.decode_mixed_sharded({"de": OneBad(1)}, ["de"] * n, pcm, rank, world)
assert got.errors == {3: -4} and sorted(got) == [i for i in range(n) if i != 3]
# a whole batch fails on ONE rank only: every rank takes part in the gather and every rank raises
class RankBad(Fake):
    def decode_batch(self, pcm):
        if rank == 1:
            raise RuntimeError("HIP error: device lost")
        return super().decode_batch(pcm)
try:
    shard.decode_mixed_sharded({"de": RankBad(1)}, ["de"] * n, pcm, rank, world)
    raise SystemExit("a failed rank must raise on every rank")
except shard.ShardError as e:
    assert "utterance 1" in str(e), str(e)
# gather=False: this rank's utterances only, no collective
got = shard.decode_mixed_sharded(models, names, pcm, rank, world, gather=False)
assert sorted(got) == list(range(rank, n, world))
dist.destroy_process_group()
print("ok", rank)
'''


def test_sharded_gather_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(ROOT)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err.decode()[-2000:]
        assert b"ok" in out


_WORKER8 = r'''
import concurrent.futures, os, sys, threading
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from rhasspy_speech_amd import _lib, shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
want = set(int(c) for c in os.environ["RS_BIND_CPULIST"].split(","))
assert _lib.bind_host_thread(0) == len(want) and os.sched_getaffinity(0) == want
seen = []
def init():
    _lib.bind_host_thread(0)
pool = concurrent.futures.ThreadPoolExecutor(max_workers=3, initializer=init)
n, steps = 37, 7                         # utterances per step (ragged over 8 ranks), steps; three decode calls in flight
def decode(step):                        # stand-in for rs_decode_batch_sharded(..., rccl_comm = NULL): this rank's records only
    seen.append((threading.get_ident(), frozenset(os.sched_getaffinity(0))))
    idx = shard.shard_indices(n, rank, world)
    words = [[step, i, (i * 7 + step) % 11][: 1 + (i + step) % 3] for i in idx]
    costs = [(float(i) + step, -float(i) * step) for i in idx]
    return shard.pack_records(idx, words, costs)
futures = [pool.submit(decode, k) for k in range(steps)]
for k in range(steps):                   # the gathers: one thread, step order, the same on every rank
    got = shard.unpack_records(shard.gather_records(futures[k].result(), n), n)
    assert sorted(got) == list(range(n))
    for i in range(n):
        assert got[i] == ([k, i, (i * 7 + k) % 11][: 1 + (i + k) % 3], float(i) + k, -float(i) * k), (k, i, got[i])
assert len({t for t, _ in seen}) >= 1 and all(a == want for _, a in seen), seen
dist.destroy_process_group()
print("ok", rank)
'''


def test_eight_ranks_three_steps_in_flight_bound_to_their_cpus(tmp_path):
    """The host side of an 8-GPU node without the GPUs: eight gloo ranks, each with three decode calls in flight on worker
    threads (stand-in decodes: the real ones need a GPU) and one thread gathering the records in step order, every thread of a
    rank restricted to the rank's CPUs by rs_bind_host_thread (RS_BIND_CPULIST stands in for the GPU's local_cpulist)."""
    ncpu = len(os.sched_getaffinity(0))
    cpus = sorted(os.sched_getaffinity(0))
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="8")
    procs = [subprocess.Popen([sys.executable, str(script), str(ROOT)], env=dict(env, RANK=str(r), RS_BIND_CPULIST=str(cpus[r % ncpu])),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(8)]
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err.decode()[-2000:]
        assert b"ok" in out


# ------------------------------------------------------------------------------------------ fuzzy matcher (host side)
def _fuzzy_cases():
    import json
    return json.loads((cases.GOLDEN / "fuzzy" / "cases.json").read_text())


def test_fuzzy_matcher_matches_reference():
    """rs_fuzzy_match vs the reference's get_fuzzy_text run on its own OpenFst tools (oracle/gen_fuzzy_golden.py): the text
    and the cost (an exact double: it is what gets compared with max_fuzzy_cost) of every case, including the ties between
    equally cheap paths, which fall the way fstcompose | fstshortestpath make them fall."""
    from rhasspy_speech_amd import transcribe_util
    n_match = 0
    for c in _fuzzy_cases():
        got = transcribe_util.get_fuzzy_text(c["nbest"].encode(), cases.GOLDEN / "fuzzy" / c["lang"])
        if c["result"] is None:
            assert got is None, c
        else:
            assert got is not None and got[0] == c["result"][0] and got[1] == c["result"][1], (c, got)
            n_match += 1
    assert n_match > 100


def test_fuzzy_control_flow_of_the_transcriber(tmp_path):
    """transcribe_wav.py:87-105: a fuzzy hit within max_fuzzy_cost replaces the n-best list by one decoded text; otherwise
    require_fuzzy empties the result; without G.fuzzy.fst nothing changes."""
    from rhasspy_speech_amd.transcribe_wav import KaldiNnet3WavTranscriber
    from rhasspy_speech_amd.meta import read_words_txt, decode_meta
    lang = cases.GOLDEN / "fuzzy" / "eps"
    hit = next(c for c in _fuzzy_cases() if c["lang"] == "eps" and c["result"] is not None and c["result"][1] > 0.5)
    t = KaldiNnet3WavTranscriber("unused", "unused")
    t._words = read_words_txt(lang / "words.txt")
    nb = hit["nbest"].encode()
    assert t._finish(nb, lang, hit["result"][1], False) == [decode_meta(hit["result"][0])]
    plain = t._finish(nb, tmp_path, None, False)                       # no G.fuzzy.fst there
    assert len(plain) == len([l for l in hit["nbest"].splitlines() if len(l.split()) > 1])
    assert t._finish(nb, lang, hit["result"][1] - 0.25, False) == plain   # too expensive: fall back to the n-best list
    assert t._finish(nb, lang, hit["result"][1] - 0.25, True) == []
    with pytest.raises(TypeError):
        t._finish(nb, lang, None, False)                               # the reference compares `cost <= None` too


@pytest.mark.parametrize("source,extra", [("feat_kernels.hip", ["-ffp-contract=off"]), ("nnet_kernels.hip", []),
                                           ("nnet_gemm_b3.hip", []), ("nnet_gemm_b3i.hip", []), ("nnet_gemm_b3j.hip", []), ("ivector_kernels.hip", []), ("decode_reg.hip", ["-ffp-contract=off"]),
                                           ("decode_kernels.hip", ["-ffp-contract=off"]), ("decode_dense.hip", ["-ffp-contract=off"])])
def test_no_packed_fp32_math_beside_the_gemm(source, extra, tmp_path):
    """Kernels that can share a CU with the MFMA GEMM of another decode call must not contain packed FP32 VALU math
    (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): a packed add / multiply whose LOW result takes the HIGH word of source 1
    (op_sel[1] = 1) returns 0.0 in 16-lane groups while another wave on the CU feeds VALU results into MFMA sources, as
    GemmKernelB3 does (profiles/r04/pk_interference.txt: bisected to the instruction, reproduced with two standalone kernels).
    Which operand selection the compiler picks is not ours to choose, so none of the packed forms is allowed: the compiler
    forms them on its own when it vectorises scalar float code, the Makefile builds these files with NOPACK, and this test
    compiles them the same way and looks at the ISA.
    (The UBM scoring, once the one deliberate v_pk_fma_f32 user, runs on the matrix cores now -- UbmPostMfmaKernel -- and its
    vector fallback uses scalar FMAs.)"""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc")
    if hipcc is None:
        pytest.skip("hipcc not on PATH")
    csrc = Path(__file__).resolve().parent.parent / "rhasspy_speech_amd" / "csrc"
    nopack = re.search(r"^NOPACK\s*=\s*(.*)$", (csrc / "Makefile").read_text(), re.M).group(1).split()
    assert "-fno-slp-vectorize" in nopack and "-fno-vectorize" in nopack
    flags = nopack if source in ("feat_kernels.hip", "nnet_kernels.hip", "ivector_kernels.hip", "decode_kernels.hip", "nnet_gemm_b3.hip",
                                 "nnet_gemm_b3i.hip", "nnet_gemm_b3j.hip") else []
    out = tmp_path / "k.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", *extra, *flags, "-S", "--cuda-device-only", str(csrc / source),
                    "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    packed = re.findall(r"v_pk_(?:add|mul|fma)_f32", out.read_text())
    assert not packed, f"{source}: {len(packed)} packed FP32 instructions"


def test_decoder_options_the_reference_asserts_on(tmp_path):
    """LatticeFasterDecoderConfig::Check (lattice-faster-decoder.h:86-91): the reference's decoder binaries abort on
    min-active > max-active, a non-positive beam / lattice beam, max-active <= 1; the library refuses the model load with the
    assertion's text (the randomised decode cases contain such a draw: tests/golden/fuzz_decode.json case 25)."""
    from rhasspy_speech_amd import _lib, synth
    spec = synth.tiny_spec()
    synth.write_model_dir(tmp_path / "model", spec)
    synth.make_grammar_graph(tmp_path / "graph", spec)
    for bad in (dict(max_active=100, min_active=200), dict(beam=0.0), dict(lattice_beam=-2.0), dict(max_active=1, min_active=0)):      # (exactly -1 is RS_OPT_UNSET)
        with pytest.raises(_lib.RsError, match="min_active <= max_active"):
            _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(**bad))
    _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(max_active=200, min_active=200)).close()


def test_library_asks_for_eight_hardware_queues_unless_told_otherwise():
    """Loading the library puts GPU_MAX_HW_QUEUES=8 into the process environment (api.cc: calls in flight use three streams each and
    the runtime's default of four hardware queues serialises them), and leaves a value that is already there alone.  (The C
    library's environment is read back through its own getenv: os.environ is a snapshot Python took at start-up.)"""
    import subprocess
    import sys
    code = """
import ctypes, os, sys
preset = sys.argv[1]
if preset == "-":
    os.environ.pop("GPU_MAX_HW_QUEUES", None)
else:
    os.environ["GPU_MAX_HW_QUEUES"] = preset
from rhasspy_speech_amd import _lib
_lib.lib()
getenv = ctypes.CDLL(None).getenv
getenv.restype = ctypes.c_char_p
print(getenv(b"GPU_MAX_HW_QUEUES").decode())
"""
    root = str(Path(__file__).resolve().parent.parent)
    for preset, want in (("-", "8"), ("5", "5")):
        out = subprocess.run([sys.executable, "-c", code, preset], cwd=root, capture_output=True, text=True, check=True).stdout.strip()
        assert out == want, (preset, out)


def test_frame_subsampling_strides_the_residual_sums_of_a_factorised_tdnn(tmp_path, monkeypatch):
    """TDNN-F with --frame-subsampling-factor=3: layer offsets (0) (-1,0,1) (-1,0,1) (-3,0,3) x 2, every layer a bottleneck + affine +
    Sum(Scale(0.66, previous), this).  Everything from the third layer's affine up is read at multiples of three only -- the
    residual sums too, whether they run as elementwise ops (RS_FUSE_RESIDUAL=0: they take row lists like the GEMMs; round 5 forced an
    elementwise op's buffers dense after the walk, leaving the layer that feeds it dense over an input evaluated on every third
    row: ADVICE r05) or folded into the affine GEMM's epilogue (the default)."""
    from rhasspy_speech_amd import _lib, synth
    spec = synth.tiny_spec(tdnnf=True, hidden_dim=64, bottleneck_dim=16, layer_offsets=((0,), (-1, 0, 1), (-1, 0, 1), (-3, 0, 3), (-3, 0, 3)))
    synth.write_model_dir(tmp_path / "model", spec)
    synth.make_grammar_graph(tmp_path / "graph", spec)
    dense = ["tdnn1.affine", "tdnnf2.linear", "tdnnf2.affine", "tdnnf2.noop", "tdnnf3.linear"]
    for fused in (False, True):
        if not fused:
            monkeypatch.setenv("RS_FUSE_RESIDUAL", "0")
        ops = [l for l in _lib.Model(tmp_path / "model", tmp_path / "graph", _lib.default_opts(frame_subsampling_factor=3)).describe().splitlines() if l.startswith("op:")]
        ops1 = [l for l in _lib.Model(tmp_path / "model", tmp_path / "graph").describe().splitlines() if l.startswith("op:")]
        monkeypatch.delenv("RS_FUSE_RESIDUAL", raising=False)
        names = [l.split()[2].split("+")[0] for l in ops]
        strided = {n: "rows=every-3" in l for n, l in zip(names, ops)}
        assert all(not strided[n] for n in dense if n in strided), ops
        assert all(v for n, v in strided.items() if n not in dense), ops
        if fused:
            assert not any(l.startswith("op: eltwise") for l in ops) and sum("residual=0.66*" in l for l in ops) == 4, ops
            assert sum("residual=" in l and "rows=every-3" in l for l in ops) == 3, ops
        else:
            assert sum(1 for l in ops if l.startswith("op: eltwise") and "rows=every-3" in l) == 3 and not any("residual=" in l for l in ops), ops
        assert not any("rows=every" in l for l in ops1)
