#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (build container; needs oracle/_ref) -- how much the REFERENCE's own intermediate results move when nothing
but the BLAS kernel family changes.

north_star holds the HIP path to 1e-4 on log-likelihoods, and tests/test_gpu_configs.py lists the utterances on which it cannot hold
(INTERMEDIATE_DEVIATIONS: a UBM Gaussian-selection / min-post decision that falls inside the FP32 rounding of the scores flips, the
iVector moves by 2e-4 .. 3e-3).  The reference computes those scores with cblas_sgemv (gmm/diag-gmm.cc:546-562) and links whatever
OpenBLAS the system has (README.md:47; version and kernel family unpinned; OpenBLAS picks the kernels by CPU at run time).  This script
runs `rs-dump` -- the reference's own classes -- on every utterance of the full-size configurations twice, with OPENBLAS_CORETYPE set to
two kernel families (the oracle's SciPy OpenBLAS is a DYNAMIC_ARCH build), and reports the utterances whose iVector differs by more
than 1e-4 between the two runs of THE REFERENCE ITSELF, i.e. the flip rate of the reference against itself on another x86 machine.

Usage: python oracle/blas_variability.py [SkylakeX Haswell] > profiles/r06/blas_variability.txt
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from rhasspy_speech_amd import synth  # noqa: E402
from tests import configs  # noqa: E402

BIN = REPO / "oracle" / "_ref" / "bin"


def dump(model_dir: Path, pcm, work: Path, core: str, mode: str):
    work.mkdir(parents=True, exist_ok=True)
    wav, out = work / "u.wav", work / "dump"
    shutil.rmtree(out, ignore_errors=True)
    out.mkdir()
    synth.write_wav(wav, pcm)
    env = dict(os.environ, PATH=f"{BIN}:{os.environ['PATH']}", OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1", OPENBLAS_CORETYPE=core)
    conf = model_dir / "model" / "online" / "conf" / "online.conf"
    r = subprocess.run(["rs-dump", f"--config={conf}", "--acoustic-scale=1.0", mode, str(model_dir / "model" / "model" / "final.mdl"), str(wav), str(out)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError(r.stderr.decode()[-1000:])
    iv, ll = np.load(out / "ivector.npy"), np.load(out / "loglikes.npy")
    shutil.rmtree(out, ignore_errors=True)
    return iv, ll


def one(args):
    model_dir, pcm, work, cores, mode = args
    a, b = dump(model_dir, pcm, work, cores[0], mode), dump(model_dir, pcm, work, cores[1], mode)
    n = min(len(a[0]), len(b[0]))
    return float(np.abs(a[0][:n] - b[0][:n]).max()), float(np.abs(a[1] - b[1]).max())


def main():
    cores = sys.argv[1:3] if len(sys.argv) >= 3 else ["SkylakeX", "Haswell"]
    print(f"# the reference (oracle/_ref: Kaldi from /root/reference, SciPy's DYNAMIC_ARCH OpenBLAS) against itself: OPENBLAS_CORETYPE={cores[0]} vs {cores[1]}")
    with tempfile.TemporaryDirectory() as tds:
        td = Path(tds)
        sets = []
        md, _ = configs.build_grammar_model(td / "zam")
        sets.append(("c1_grammar", md, configs.grammar_utterances(), "offline"))
        md2, _ = configs.build_arpa_model(td / "arpa")
        sets.append(("c2_arpa", md2, configs.arpa_utterances(), "offline"))
        names, pcms = configs.mixed_utterances()
        for key, tag in (("de_DE-like", "c3_mixed_de"), ("fr_FR-like", "c3_mixed_fr")):
            m = configs.MIXED_MODELS[key]
            mdk, _ = configs.build_grammar_model(td / key, m["model_seed"], m["graph_seed"])
            sets.append((tag, mdk, [p for nm, p in zip(names, pcms) if nm == key], "offline"))
        sets.append(("c4_streams", md, configs.stream_utterances(), "stream"))
        total = flips = 0
        for tag, mdir, ps, mode in sets:
            jobs = [(mdir, p, td / f"w_{tag}_{i % 8}_{i}", cores, mode) for i, p in enumerate(ps)]
            with concurrent.futures.ThreadPoolExecutor(8) as ex:
                res = list(ex.map(one, jobs))
            iv = np.array([r[0] for r in res]); ll = np.array([r[1] for r in res])
            big = [(i, float(iv[i]), float(ll[i])) for i in np.nonzero(iv >= 1e-4)[0]]
            total += len(ps); flips += len(big)
            print(f"{tag}: {len(ps)} utterances; iVector max |diff| between the two runs: median {np.median(iv):.2e}, max {iv.max():.2e}; log-likelihoods: median {np.median(ll):.2e}, "
                  f"max {ll.max():.2e}; iVector beyond 1e-4: {[(i, f'{a:.1e}', f'{b:.1e}') for i, a, b in big]}", flush=True)
        print(f"# {flips} of {total} utterances: the reference differs from ITSELF by more than 1e-4 in the iVector when only the BLAS kernel family changes")


if __name__ == "__main__":
    main()
