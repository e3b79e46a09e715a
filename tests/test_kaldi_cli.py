"""The Kaldi-named executables over the HIP library (rhasspy_speech_amd/bin, kaldi_cli.py): command lines exactly as the
reference builds them (transcribe_wav.py:46-74, transcribe_stream.py:53-64), output read by the REFERENCE's own
lattice-to-nbest | nbest-to-linear (oracle/_ref, test infrastructure) as in the reference's pipelines."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests import cases
from tests.lattice_io import BIN

REPO = Path(__file__).resolve().parent.parent
SHIMS = REPO / "rhasspy_speech_amd" / "bin"


def reference_wav_argv(model_dir, graph_dir, wav, max_active=7000, lattice_beam=8.0, beam=24.0):
    return ["online2-wav-nnet3-latgen-faster", "--online=false", "--do-endpointing=false",
            f"--word-symbol-table={graph_dir}/words.txt", f"--config={model_dir}/model/online/conf/online.conf",
            f"--max-active={max_active}", f"--lattice-beam={lattice_beam}", "--acoustic-scale=1.0", f"--beam={beam}",
            f"{model_dir}/model/model/final.mdl", f"{graph_dir}/HCLG.fst", "ark:echo utt utt|", f"scp:echo utt {wav}|", "ark:-"]


def test_command_line_of_the_reference_parses():
    from rhasspy_speech_amd import kaldi_cli
    opts, config, pos = kaldi_cli.parse_command_line(reference_wav_argv("/m", "/g", "/tmp/a.wav", 2500, 6.0, 13.0)[1:])
    # (--online=false --do-endpointing=false on the command line override online.conf: rs_decode_opts.command_line_fixed = ONLINE | DO_ENDPOINTING)
    assert opts == dict(max_active=2500, lattice_beam=6.0, acoustic_scale=1.0, beam=13.0, command_line_fixed=3)
    assert config == "/m/model/online/conf/online.conf"
    assert pos == ["/m/model/model/final.mdl", "/g/HCLG.fst", "ark:echo utt utt|", "scp:echo utt /tmp/a.wav|", "ark:-"]
    assert kaldi_cli.read_table("ark:echo utt utt|") == [("utt", "utt")]
    assert kaldi_cli.read_table("scp:echo utt /tmp/a.wav|") == [("utt", "/tmp/a.wav")]
    with pytest.raises(ValueError):
        kaldi_cli.parse_command_line(["--no-such-option=1", "--config=x"])
    with pytest.raises(ValueError):
        kaldi_cli.parse_command_line(["a", "b"])                         # no --config
    with pytest.raises(ValueError):
        kaldi_cli.open_wspecifier("ark,t:-")


def test_executables_fail_loudly_on_bad_input(tmp_path):
    """Non-zero status + a message on stderr (what tools.py:138-145 turns into RuntimeError); nothing on stdout."""
    for exe, argv in (("online2-wav-nnet3-latgen-faster", ["--config=/nonexistent.conf", "/no.mdl", "/no.fst", "ark:echo utt utt|",
                                                           "scp:echo utt /no.wav|", "ark:-"]),
                      ("online2-cli-nnet3-decode-faster", ["--config=/nonexistent.conf", "/no.mdl", "/no.fst", "/no.txt", f"ark:{tmp_path}/l"])):
        p = subprocess.run([sys.executable, str(SHIMS / exe), *argv], stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode != 0 and p.stdout == b"" and b"ERROR" in p.stderr, (exe, p.stderr)


def test_wav_executable_checks_what_the_binary_checks(case_cache, tmp_path):
    """The wav shim with the reference's argv: a wav whose sampling rate is not the model's ends it with Kaldi's "Sampling frequency
    mismatch" (feat/online-feature.cc:97-101) and status 1 -- before any device work, so this runs without a GPU -- and a spk2utt table
    that asks for the iVector adaptation state to be carried from one utterance to the next (online2-wav-nnet3-latgen-faster.cc:
    203-205) is refused rather than decoded differently."""
    import wave
    model_dir, graph_dir, wav, pcm = case_cache("tiny_u0")
    w8 = tmp_path / "u8k.wav"
    with wave.open(str(w8), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
        w.writeframes(pcm[::2].tobytes())
    argv = reference_wav_argv(model_dir, graph_dir, w8)
    p = subprocess.run([sys.executable, str(SHIMS / argv[0]), *argv[1:]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and p.stdout == b"" and b"Sampling frequency mismatch, expected 16000, got 8000" in p.stderr, p.stderr
    argv = reference_wav_argv(model_dir, graph_dir, wav)
    argv[-3], argv[-2] = "ark:echo spk a b|", f"scp:printf 'a {wav}\\nb {wav}\\n'|"
    p = subprocess.run([sys.executable, str(SHIMS / argv[0]), *argv[1:]], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and p.stdout == b"" and b"adaptation state" in p.stderr, p.stderr


def _nbest_text(lattice: bytes, n: int, env) -> bytes:
    sh = f"lattice-to-nbest --n={n} --acoustic-scale=1.0 ark:- ark:- | nbest-to-linear ark:- ark:/dev/null ark,t:-"
    p = subprocess.run(["bash", "-c", sh], input=lattice, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_u0", "zam_u1", "tiny_arpa_u7"])
def test_wav_executable_in_the_reference_pipeline(case_cache, name):
    from rhasspy_speech_amd import _lib
    if not (BIN / "lattice-to-nbest").exists():
        pytest.fail("oracle/_ref is not built (bash oracle/build_ref.sh in the build container; it travels with the snapshot)")
    model_dir, graph_dir, wav, pcm = case_cache(name)
    o = cases.CASES[name].get("opts", {})
    argv = reference_wav_argv(model_dir, graph_dir, wav, o.get("max_active", 7000), o.get("lattice_beam", 8.0), o.get("beam", 24.0))
    env = dict(os.environ, PATH=f"{SHIMS}:{BIN}:{os.environ['PATH']}")       # the executables first, the Kaldi tools behind them
    p = subprocess.run(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    text = _nbest_text(p.stdout, cases.NBEST, env)
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts(**o))
    res = model.decode_batch([pcm], nbest=cases.NBEST)
    assert text == res.text(0, "utt") and text.startswith(b"utt-1")


@pytest.mark.gpu
def test_stream_executable_in_the_reference_pipeline(case_cache, tmp_path):
    from rhasspy_speech_amd import _lib
    model_dir, graph_dir, _, pcm = case_cache("zam_u1")
    lat = tmp_path / "lat.ark"
    argv = ["online2-cli-nnet3-decode-faster", f"--config={model_dir}/model/online/conf/online.conf", "--max-active=7000",
            "--lattice-beam=8.0", "--acoustic-scale=1.0", "--beam=24.0", f"{model_dir}/model/model/final.mdl", f"{graph_dir}/HCLG.fst",
            f"{graph_dir}/words.txt", f"ark:{lat}"]
    env = dict(os.environ, PATH=f"{SHIMS}:{BIN}:{os.environ['PATH']}")
    p = subprocess.run(argv, env=env, input=np.asarray(pcm, dtype="<i2").tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    text = _nbest_text(lat.read_bytes(), 3, env)
    stream = _lib.Stream(_lib.Model(model_dir, graph_dir, _lib.default_opts()))
    stream.accept(pcm)
    assert text == stream.finish(nbest=3).text(0, "utt") and text.startswith(b"utt-1")
