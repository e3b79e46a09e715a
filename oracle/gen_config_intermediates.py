#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- the REFERENCE's intermediate results for the BASELINE.json configurations at their FULL sizes
(runs in the build container only; needs oracle/_ref built by oracle/build_ref.sh).

tests/golden/configs/<name>.npz (oracle/gen_config_golden.py) hold words and costs only, so a difference in the features /
iVectors / log-likelihoods of a full-size utterance was only ever seen when it moved a cost (round 4: config 3 de #238, a UBM
Gaussian-selection near-tie).  This script stores, per configuration, what `rs-dump` (oracle/drivers/rs-dump.cc: our main() against
the reference's own classes, one fresh process per utterance like rhasspy runs the binaries) computes:
  ivector      float32 [n_utts, ivector_dim]      offline: the utterance's iVector; streams: the LAST chunk's
  chunk_iv     float32 [n_utts, K, ivector_dim]   streams only: the iVectors of every CHUNK_STRIDE-th nnet chunk (K per stream)
  ll_utts      int32   [16]                       the utterances whose log-likelihoods are sampled
  ll           float32 [16, ceil(T/LL_ROWS), ceil(P/LL_COLS)]   rows ::LL_ROWS, columns ::LL_COLS of their log-likelihood matrices
  num_frames   int32   [n_utts]
into tests/golden/configs/<name>_inter.npz.  tests/test_gpu_configs.py holds the HIP path to them (1e-4).

Usage: python oracle/gen_config_intermediates.py [c1_grammar c2_arpa c3_mixed_de c3_mixed_fr c4_streams c5_tdnnf c5_tdnnf_fsf3 c6_tdnnf1536]
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
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
LL_ROWS, LL_COLS, CHUNK_STRIDE, N_LL = 16, 25, 8, 16


def dump_one(model_dir: Path, pcm, i: int, work: Path, mode: str, want_ll: bool):
    work.mkdir(parents=True, exist_ok=True)
    wav, out = work / "u.wav", work / "dump"
    shutil.rmtree(out, ignore_errors=True)
    out.mkdir()
    synth.write_wav(wav, pcm)
    conf = model_dir / "model" / "online" / "conf" / "online.conf"
    r = subprocess.run(["rs-dump", f"--config={conf}", "--acoustic-scale=1.0", mode, str(model_dir / "model" / "model" / "final.mdl"), str(wav), str(out)],
                       env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError(r.stderr.decode()[-2000:])
    iv = np.load(out / "ivector.npy")
    ll = np.load(out / "loglikes.npy")
    rec = dict(i=i, T=ll.shape[0], ivector=iv[0].copy() if mode == "offline" else iv[-1].copy())
    if mode == "stream":
        rec["chunk_iv"] = iv[::CHUNK_STRIDE].copy()
    if want_ll:
        rec["ll"] = ll[::LL_ROWS, ::LL_COLS].copy()
    shutil.rmtree(out, ignore_errors=True)
    return rec


def run(name: str, model_dir: Path, pcms, td: Path, mode: str, workers: int = 8):
    n = len(pcms)
    ll_utts = sorted(set(np.linspace(0, n - 1, N_LL).astype(int).tolist()))
    t0 = time.time()
    recs = {}
    with concurrent.futures.ThreadPoolExecutor(workers) as ex:
        futs = [ex.submit(dump_one, model_dir, p, i, td / f"{name}_{i % workers}_{i}", mode, i in ll_utts) for i, p in enumerate(pcms)]
        for f in futs:
            r = f.result()
            recs[r["i"]] = r
    out = dict(ivector=np.stack([recs[i]["ivector"] for i in range(n)]).astype(np.float32),
               num_frames=np.array([recs[i]["T"] for i in range(n)], np.int32),
               ll_utts=np.array(ll_utts, np.int32), ll_stride=np.array([LL_ROWS, LL_COLS], np.int32))
    rows = max(recs[i]["ll"].shape[0] for i in ll_utts)
    ll = np.full((len(ll_utts), rows, recs[ll_utts[0]]["ll"].shape[1]), np.nan, np.float32)      # (ragged: NaN beyond an utterance's rows)
    for k, i in enumerate(ll_utts):
        ll[k, :recs[i]["ll"].shape[0]] = recs[i]["ll"]
    out["ll"] = ll
    if mode == "stream":
        K = max(recs[i]["chunk_iv"].shape[0] for i in range(n))
        civ = np.full((n, K, out["ivector"].shape[1]), np.nan, np.float32)
        for i in range(n):
            civ[i, :recs[i]["chunk_iv"].shape[0]] = recs[i]["chunk_iv"]
        out["chunk_iv"] = civ
        out["chunk_stride"] = np.int32(CHUNK_STRIDE)
    np.savez_compressed(configs.GOLDEN / f"{name}_inter.npz", **out)
    print(f"{name}: {n} utterances, iVectors {out['ivector'].shape}, log-likelihood samples {ll.shape}, reference wall {time.time() - t0:.1f} s on {workers} processes", flush=True)


def main():
    want = sys.argv[1:] or ["c1_grammar", "c2_arpa", "c3_mixed_de", "c3_mixed_fr", "c4_streams"]
    with tempfile.TemporaryDirectory() as tds:
        td = Path(tds)
        if "c1_grammar" in want:
            md, _ = configs.build_grammar_model(td / "zam")
            run("c1_grammar", md, configs.grammar_utterances(), td, "offline")
        if "c2_arpa" in want:
            md, _ = configs.build_arpa_model(td / "arpa")
            run("c2_arpa", md, configs.arpa_utterances(), td, "offline")
        if "c3_mixed_de" in want or "c3_mixed_fr" in want:
            names, pcms = configs.mixed_utterances()
            for key, tag in (("de_DE-like", "c3_mixed_de"), ("fr_FR-like", "c3_mixed_fr")):
                if tag in want:
                    m = configs.MIXED_MODELS[key]
                    md, _ = configs.build_grammar_model(td / key, m["model_seed"], m["graph_seed"])
                    run(tag, md, [p for nm, p in zip(names, pcms) if nm == key], td, "offline")
        if "c4_streams" in want:
            md, _ = configs.build_grammar_model(td / "zam")
            run("c4_streams", md, configs.stream_utterances(), td, "stream")
        if "c5_tdnnf" in want:
            md, _ = configs.build_tdnnf_model(td / "zamf")
            run("c5_tdnnf", md, configs.grammar_utterances()[:configs.N_TDNNF_UTTS], td, "offline")
        if "c6_tdnnf1536" in want:
            md, _ = configs.build_tdnnf_model(td / "zamf1536", spec_kw=configs.TDNNF1536_SPEC)
            run("c6_tdnnf1536", md, configs.grammar_utterances()[:configs.N_TDNNF1536_UTTS], td, "offline")
        if "c5_tdnnf_fsf3" in want:
            md, _ = configs.build_tdnnf_model(td / "zamf_fsf3", conf_opts=configs.FSF3_CONF)
            run("c5_tdnnf_fsf3", md, configs.grammar_utterances()[:configs.N_TDNNF_FSF3_UTTS], td, "offline")


if __name__ == "__main__":
    main()
