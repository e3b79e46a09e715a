#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- golden-vector generator (runs in the build container only).

Runs the REFERENCE (Kaldi binaries compiled by oracle/build_ref.sh from /root/reference, plus our dump
driver oracle/drivers/rs-dump.cc linked against the same libraries) on synthetic models written by
rhasspy_speech_amd.synth and stores inputs' identity (case parameters) and expected outputs as small .npz
fixtures under tests/golden/.  The models themselves are NOT stored: tests regenerate them from the same
seeds (numpy Generator streams are stable), so fixtures stay small.

Pipeline per case = exactly the argv of rhasspy_speech/transcribe_wav.py:45-75 (offline) or
transcribe_stream.py:53-99 (stream), with nbest-to-linear's optional 4th/5th outputs for the costs.

Usage: python oracle/gen_golden.py [case ...]
"""
from __future__ import annotations

import json
import os
import shutil
import subprocess
import sys
import tempfile
import wave
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from rhasspy_speech_amd import synth  # noqa: E402

BIN = REPO / "oracle" / "_ref" / "bin"
GOLDEN = REPO / "tests" / "golden"
REF_WAVS = Path("/root/reference/tests/en_US-zamia")

from tests.cases import CASES, NBEST, build_case_files, case_audio, case_spec  # noqa: E402,F401


def decoder_args(case: dict):
    o = dict(max_active=7000, lattice_beam=8.0, beam=24.0)
    o.update({k: v for k, v in case.get("opts", {}).items() if k in o})
    args = [f"--max-active={o['max_active']}", f"--lattice-beam={o['lattice_beam']}", "--acoustic-scale=1.0", f"--beam={o['beam']}"]
    if "min_active" in case.get("opts", {}):
        args.append(f"--min-active={case['opts']['min_active']}")
    return args


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def parse_vec_ark(text: str):
    out = {}
    for line in text.splitlines():
        p = line.split()
        if p:
            out[p[0]] = [float(x) for x in p[1:]]
    return out


def gen_case(name: str, case: dict) -> None:
    env = dict(os.environ, PATH=f"{BIN}:{os.environ['PATH']}")
    with tempfile.TemporaryDirectory() as td:
        root = Path(td)
        model_dir, graph_dir, wav, pcm = build_case_files(case, root)
        conf = model_dir / "model" / "online" / "conf" / "online.conf"
        mdl = model_dir / "model" / "model" / "final.mdl"
        out = {}
        for mode in ("offline", "stream"):
            lat = root / f"{mode}.lat"
            if mode == "offline":
                cmd = ["online2-wav-nnet3-latgen-faster", "--online=false", "--do-endpointing=false",
                       f"--word-symbol-table={graph_dir / 'words.txt'}", f"--config={conf}", *decoder_args(case),
                       str(mdl), str(graph_dir / "HCLG.fst"), "ark:echo utt utt|", f"scp:echo utt {wav}|", f"ark:{lat}"]
                p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            else:
                cmd = ["online2-cli-nnet3-decode-faster", f"--config={conf}", *decoder_args(case), str(mdl),
                       str(graph_dir / "HCLG.fst"), str(graph_dir / "words.txt"), f"ark:{lat}"]
                p = subprocess.run(cmd, env=env, input=pcm.astype("<i2").tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if p.returncode != 0:
                out[f"{mode}_status"] = np.int32(p.returncode)
                out[f"{mode}_stderr"] = np.frombuffer(p.stderr[-400:], dtype=np.uint8)
                continue
            out[f"{mode}_status"] = np.int32(0)
            sh = (f"lattice-to-nbest --n={NBEST} --acoustic-scale=1.0 ark:{lat} ark:- | "
                  f"nbest-to-linear ark:- ark:/dev/null ark,t:- ark,t:{root}/lm.txt ark,t:{root}/ac.txt")
            q = run(["bash", "-c", sh], env=env)
            out[f"{mode}_nbest_text"] = np.frombuffer(q.stdout, dtype=np.uint8)
            lm = parse_vec_ark((root / "lm.txt").read_text())
            ac = parse_vec_ark((root / "ac.txt").read_text())
            keys = sorted(lm, key=lambda k: int(k.split("-")[1]))
            out[f"{mode}_graph_cost"] = np.array([lm[k][0] for k in keys], np.float32)
            out[f"{mode}_acoustic_cost"] = np.array([ac[k][0] for k in keys], np.float32)
            dump = root / f"dump_{mode}"
            dump.mkdir()
            run(["rs-dump", f"--config={conf}", "--acoustic-scale=1.0", mode, str(mdl), str(wav), str(dump)], env=env)
            ll = np.load(dump / "loglikes.npy")
            out[f"{mode}_num_frames"] = np.int32(ll.shape[0])
            if case.get("big"):
                out[f"{mode}_loglikes"] = ll[::8, ::4].copy()
                out["loglikes_stride"] = np.array([8, 4], np.int32)
            else:
                out[f"{mode}_loglikes"] = ll
                out["loglikes_stride"] = np.array([1, 1], np.int32)
            if mode == "offline":
                out["input"] = np.load(dump / "input.npy")
            if (dump / "ivector.npy").exists():
                iv = np.load(dump / "ivector.npy")
                out[f"{mode}_ivector"] = iv[:1] if mode == "offline" else iv
                out[f"{mode}_chunk_tick"] = np.load(dump / "chunk_tick.npy")[0].astype(np.int32)
            if mode == "offline" and (dump / "lda_norm.npy").exists() and not case.get("big"):
                out["cmvn"] = np.load(dump / "cmvn.npy")
                out["lda"] = np.load(dump / "lda.npy")
                out["lda_norm"] = np.load(dump / "lda_norm.npy")
        # rand() calls of the reference's model set-up (decides the dither seeds): oracle/nnet3_rand.py and
        # rhasspy_speech_amd/csrc/nnet3_setup.cc are checked against this
        rp = run(["rs-dump", f"--config={conf}", "randpos", str(mdl), "-", "-"], env=env)
        out["rand_calls"] = np.int64(int(rp.stdout))
        out["case_json"] = np.frombuffer(json.dumps(case, sort_keys=True).encode(), dtype=np.uint8)
        GOLDEN.mkdir(parents=True, exist_ok=True)
        np.savez_compressed(GOLDEN / f"{name}.npz", **out)
        txt = bytes(out.get("offline_nbest_text", np.zeros(0, np.uint8))).decode().strip().replace("\n", " | ")
        print(f"{name}: T={int(out.get('offline_num_frames', -1))} offline: {txt}")


def main() -> None:
    (GOLDEN / "wav").mkdir(parents=True, exist_ok=True)
    for c in CASES.values():
        if c["audio"].startswith("wav:"):
            fn = c["audio"].split(":")[1]
            if not (GOLDEN / "wav" / fn).exists():
                shutil.copy(REF_WAVS / fn, GOLDEN / "wav" / fn)   # data files of the reference's own tests
    # the reference's own Dither() noise in a fresh process (pins oracle/dither.c and rs_dither_noise bit for bit)
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, PATH=f"{BIN}:{os.environ['PATH']}")
        ref = {}
        for T, W in ((6, 400), (5, 200), (3, 275)):
            run(["rs-dump", "dither", str(T), str(W), f"{td}/d.npy"], env=env)
            ref[f"noise_{W}"] = np.load(f"{td}/d.npy")
        np.savez_compressed(GOLDEN / "dither_ref.npz", **ref)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        gen_case(n, CASES[n])


if __name__ == "__main__":
    main()
