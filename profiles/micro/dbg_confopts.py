"""Debug aid (test infrastructure: uses the oracle): the streaming decode of parity case tiny_confopts_u14 on the GPU against the
CPU oracle's decoder run on the GPU's own log-likelihood rows, for the search kernels and option subsets."""
import os, sys, tempfile
from pathlib import Path
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch  # noqa
from tests import cases
from rhasspy_speech_amd import _lib
from oracle import pipeline, lattice as lat
case = cases.CASES["tiny_confopts_u14"]
g = np.load(cases.GOLDEN / "tiny_confopts_u14.npz")
print("reference stream:", bytes(g["stream_nbest_text"]).decode().replace("\n", " | "))
with tempfile.TemporaryDirectory() as td:
    md, gd, wav, pcm = cases.build_case_files(case, Path(td))
    orc = pipeline.Oracle(md, gd, **case["opts"])
    for dec in ("", "sparse", "hash", "reg"):
        if dec: os.environ["RS_DECODER"] = dec
        else: os.environ.pop("RS_DECODER", None)
        m = _lib.Model(md, gd, _lib.default_opts(keep_intermediates=1, **case["opts"]))
        print([l for l in m.describe().splitlines() if l.startswith("decoder_opts")][0])
        st = _lib.Stream(m)
        st.accept(pcm)
        res = st.finish(nbest=5)
        ll = res.matrix(0, 2)
        print(f"decoder={dec or 'auto'} gpu stream:", res.text(0).decode().replace("\n", " | "), [round(sum(res.costs(0, k)), 3) for k in range(res.num_hyps(0))])
        lattice, ctr = pipeline.decode(orc.fst, orc.id2pdf, ll, **orc.opts)
        paths = lat.nbest(lattice, 5, orc.opts["lattice_beam"], 1.0)
        print("   oracle decoder on the gpu's rows:", [(p.words, round(p.graph_cost + p.acoustic_cost, 3)) for p in paths])
        print("   max |ll - golden|", float(np.abs(ll[::int(g['loglikes_stride'][0]), ::int(g['loglikes_stride'][1])] - g["stream_loglikes"]).max()))
        r2 = m.decode_batch([pcm], nbest=5)
        print("   gpu offline:", r2.text(0).decode().replace("\n", " | "))
