#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- reference transcripts for the BASELINE.json configurations at their FULL sizes
(runs in the build container only; needs oracle/_ref built by oracle/build_ref.sh).

Every utterance of tests/configs.py's workloads goes through the REFERENCE binaries with the argv of
rhasspy_speech/transcribe_wav.py:45-75 (offline: online2-wav-nnet3-latgen-faster --online=false | lattice-to-nbest |
nbest-to-linear) or transcribe_stream.py:53-99 (streams: online2-cli-nnet3-decode-faster fed s16le on stdin).  The
offline binary is handed a table of utterances (one speaker per utterance, so nothing carries over) to pay the model
load once per worker instead of once per utterance; the streaming binary decodes one stdin per process, as it must.
Stored per config: the 1-best word ids of every utterance and nbest-to-linear's graph / acoustic cost
(tests/golden/configs/<name>.npz, a few KB each).

Usage: python oracle/gen_config_golden.py [c1_grammar c2_arpa c3_mixed_de c3_mixed_fr c4_streams]
"""
from __future__ import annotations

import concurrent.futures
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from rhasspy_speech_amd import synth  # noqa: E402
from tests import configs  # noqa: E402

BIN = REPO / "oracle" / "_ref" / "bin"
ENV = dict(os.environ, PATH=f"{BIN}:{os.environ['PATH']}", OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
DEC = "--max-active=7000 --lattice-beam=8.0 --acoustic-scale=1.0 --beam=24.0"
TAIL = "lattice-to-nbest --n=1 --acoustic-scale=1.0 ark:- ark:- | nbest-to-linear ark:- ark:/dev/null ark,t:- ark,t:{d}/lm.txt ark,t:{d}/ac.txt"


def parse_vec(text: str):
    out = {}
    for line in text.splitlines():
        p = line.split()
        if p:
            out[p[0]] = p[1:]
    return out


def offline_part(model_dir: Path, graph_dir: Path, pcms, ids, work: Path):
    work.mkdir(parents=True, exist_ok=True)
    for i, p in zip(ids, pcms):
        synth.write_wav(work / f"u{i:05d}.wav", p)
    (work / "wav.scp").write_text("".join(f"u{i:05d} {work}/u{i:05d}.wav\n" for i in ids))
    (work / "spk2utt").write_text("".join(f"u{i:05d} u{i:05d}\n" for i in ids))
    conf = model_dir / "model" / "online" / "conf" / "online.conf"
    cmd = (f"online2-wav-nnet3-latgen-faster --online=false --do-endpointing=false --word-symbol-table={graph_dir}/words.txt "
           f"--config={conf} {DEC} {model_dir}/model/model/final.mdl {graph_dir}/HCLG.fst ark:{work}/spk2utt scp:{work}/wav.scp ark:- | "
           + TAIL.format(d=work))
    r = subprocess.run(["bash", "-c", cmd], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError(r.stderr.decode()[-2000:])
    words = parse_vec(r.stdout.decode())
    lm, ac = parse_vec((work / "lm.txt").read_text()), parse_vec((work / "ac.txt").read_text())
    out = {}
    for i in ids:
        k = f"u{i:05d}-1"
        out[i] = ([int(x) for x in words[k]], float(lm[k][0]), float(ac[k][0]))
    return out


def stream_one(model_dir: Path, graph_dir: Path, pcm, i: int, work: Path):
    work.mkdir(parents=True, exist_ok=True)
    conf = model_dir / "model" / "online" / "conf" / "online.conf"
    cmd = (f"online2-cli-nnet3-decode-faster --config={conf} {DEC} {model_dir}/model/model/final.mdl {graph_dir}/HCLG.fst "
           f"{graph_dir}/words.txt ark:- | " + TAIL.format(d=work))
    r = subprocess.run(["bash", "-c", cmd], env=ENV, input=pcm.astype("<i2").tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError(r.stderr.decode()[-2000:])
    words = parse_vec(r.stdout.decode())
    lm, ac = parse_vec((work / "lm.txt").read_text()), parse_vec((work / "ac.txt").read_text())
    (k,) = list(words)
    return {i: ([int(x) for x in words[k]], float(lm[k][0]), float(ac[k][0]))}


def save(name: str, res: dict, n: int, note: str):
    words, off = [], [0]
    for i in range(n):
        words += res[i][0]
        off.append(len(words))
    configs.GOLDEN.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(configs.GOLDEN / f"{name}.npz", words=np.array(words, np.int32), word_offsets=np.array(off, np.int32),
                        graph_cost=np.array([res[i][1] for i in range(n)], np.float32),
                        acoustic_cost=np.array([res[i][2] for i in range(n)], np.float32),
                        note=np.frombuffer(note.encode(), np.uint8))
    nonempty = sum(1 for i in range(n) if res[i][0])
    print(f"{name}: {n} utterances, {nonempty} non-empty transcripts, {len(set(tuple(res[i][0]) for i in range(n)))} distinct")


def run_offline(name: str, model_dir: Path, graph_dir: Path, pcms, td: Path, workers: int = 8):
    n = len(pcms)
    parts = [list(range(w, n, workers)) for w in range(workers)]
    t0 = time.time()
    res = {}
    with concurrent.futures.ThreadPoolExecutor(workers) as ex:
        futs = [ex.submit(offline_part, model_dir, graph_dir, [pcms[i] for i in ids], ids, td / f"{name}_w{w}") for w, ids in enumerate(parts) if ids]
        for f in futs:
            res.update(f.result())
    save(name, res, n, "reference: online2-wav-nnet3-latgen-faster --online=false | lattice-to-nbest --n=1 | nbest-to-linear")
    print(f"  reference wall {time.time() - t0:.1f} s on {workers} processes")


def main():
    want = sys.argv[1:] or ["c1_grammar", "c2_arpa", "c3_mixed_de", "c3_mixed_fr", "c4_streams"]
    with tempfile.TemporaryDirectory() as tds:
        td = Path(tds)
        if "c1_grammar" in want:
            md, gd = configs.build_grammar_model(td / "zam")
            run_offline("c1_grammar", md, gd, configs.grammar_utterances(), td)
        if "c2_arpa" in want:
            md, gd = configs.build_arpa_model(td / "arpa")
            run_offline("c2_arpa", md, gd, configs.arpa_utterances(), td)
        if "c3_mixed_de" in want or "c3_mixed_fr" in want:
            names, pcms = configs.mixed_utterances()
            for key, tag in (("de_DE-like", "c3_mixed_de"), ("fr_FR-like", "c3_mixed_fr")):
                if tag not in want:
                    continue
                m = configs.MIXED_MODELS[key]
                md, gd = configs.build_grammar_model(td / key, m["model_seed"], m["graph_seed"])
                run_offline(tag, md, gd, [p for nm, p in zip(names, pcms) if nm == key], td)
        if "c4_streams" in want:
            md, gd = configs.build_grammar_model(td / "zam")
            pcms = configs.stream_utterances()
            t0 = time.time()
            res = {}
            with concurrent.futures.ThreadPoolExecutor(8) as ex:
                futs = [ex.submit(stream_one, md, gd, p, i, td / f"s{i}") for i, p in enumerate(pcms)]
                for f in futs:
                    res.update(f.result())
            save("c4_streams", res, len(pcms), "reference: online2-cli-nnet3-decode-faster (stdin s16le) | lattice-to-nbest --n=1 | nbest-to-linear")
            print(f"  reference wall {time.time() - t0:.1f} s on 8 processes")


if __name__ == "__main__":
    main()
