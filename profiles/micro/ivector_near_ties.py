"""Full-size configs: the utterances whose iVector differs from the reference's by more than 1e-4 (tests/test_gpu_configs.py,
INTERMEDIATE_DEVIATIONS).  For one of them: the CPU oracle's iVector (numpy FP32 sums, a third summation order) beside the
reference's (tests/golden/configs/<tag>_inter.npz), and the frames on which the UBM's Gaussian selection -- a DISCRETE function of the
scores -- is decided inside the FP32 rounding of the scores (float64 scores beside).  CPU only (test infrastructure: uses the oracle).
usage: python profiles/micro/ivector_near_ties.py <c1_grammar|c2_arpa|c3_mixed_de|c3_mixed_fr|c4_streams> <utterance>"""
import sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import configs
from oracle import pipeline

tag, U = sys.argv[1], int(sys.argv[2])
tmp = Path(tempfile.mkdtemp())
if tag == "c1_grammar":
    md, gd = configs.build_grammar_model(tmp / "m"); pcm = configs.grammar_utterances()[U]
elif tag == "c2_arpa":
    md, gd = configs.build_arpa_model(tmp / "m"); pcm = configs.arpa_utterances()[U]
elif tag == "c4_streams":
    md, gd = configs.build_grammar_model(tmp / "m"); pcm = configs.stream_utterances()[U]
else:
    key = {"c3_mixed_de": "de_DE-like", "c3_mixed_fr": "fr_FR-like"}[tag]
    m = configs.MIXED_MODELS[key]
    md, gd = configs.build_grammar_model(tmp / "m", m["model_seed"], m["graph_seed"])
    names, pcms = configs.mixed_utterances()
    pcm = [p for nm, p in zip(names, pcms) if nm == key][U]
g = np.load(configs.GOLDEN / f"{tag}_inter.npz")
orc = pipeline.Oracle(md, gd)
feats = orc.features(np.asarray(pcm))
if tag == "c4_streams":
    _, ivs, _ = orc.loglikes_stream(feats, len(pcm))
    ref = g["chunk_iv"][U]; ref = ref[~np.isnan(ref[:, 0])]
    d = np.abs(np.asarray(ivs)[::int(g["chunk_stride"])] - ref).max(1)
    print(f"{tag} stream {U}: oracle vs reference iVectors of every {int(g['chunk_stride'])}th chunk, max |diff| per sampled chunk:", [float(f"{x:.1e}") for x in d])
else:
    _, iv, _ = orc.loglikes_offline(feats)
    print(f"{tag} utterance {U}: iVector oracle vs reference max |diff| {np.abs(np.asarray(iv).reshape(-1)[:g['ivector'].shape[1]] - g['ivector'][U]).max():.2e}")
ie = orc.ie
cm = pipeline.online_cmvn(feats, ie["gstats"])
nrm = pipeline.lda_transform(pipeline.splice(cm, ie["left"], ie["right"]), ie["lda"])
gm = ie["gmm"]
x64 = nrm.astype(np.float64)
ll64 = gm.gconsts[None, :].astype(np.float64) + x64 @ gm.means_invvars.T.astype(np.float64) - 0.5 * ((x64 * x64) @ gm.inv_vars.T.astype(np.float64))
logmp = np.log(ie["min_post"])
close = []
for t, row in enumerate(ll64):
    mx = row.max()
    srt = np.sort(row)[::-1]
    n = ie["num_gselect"]
    margins = {"candidate cut (like vs max + log min_post)": np.abs(row - (mx + logmp)).min(), f"rank {n} vs {n + 1}": srt[n - 1] - srt[n]}
    # VectorToPosteriorEntry (hmm/posterior.cc): of the num_gselect best, those whose posterior is below min_post are dropped (the best
    # one always stays) and the rest renormalised: decided by p_i / sum(p) against min_post
    p = np.exp(srt[:n] - mx); p = p / p.sum()
    margins["posterior vs min_post (relative)"] = float(np.abs(p[1:] - ie["min_post"]).min() / ie["min_post"]) if n > 1 else 1.0
    for what, mg in margins.items():
        if mg < (2e-4 if what.startswith("posterior") else 2e-5):
            close.append((t, what, float(mg)))
print(f"  frames whose Gaussian selection is decided within 2e-5 of the scores (FP32 rounding of a score of magnitude ~{np.abs(ll64).mean():.0f} is {np.abs(ll64).mean() * 6e-8:.1e}):")
for c in close:
    print("   ", c)
