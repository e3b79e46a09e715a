"""Config 3, de_DE-like model, utterance 238: the one utterance of the 1600 whose cost differs from the reference's although the
CPU oracle -- which follows the reference's token order -- agrees with the KERNELS.  Which side does the difference come from?
The sequential oracle decoder (oracle/decoder.c) is run twice on that utterance: on the oracle's own log-likelihoods (numpy FP32
sums) and on the REFERENCE's (its nnet3 computation with OpenBLAS, dumped by oracle/drivers/rs-dump.cc), and both best-path costs
are set beside the reference decoder's golden cost.  Build container only (needs oracle/_ref): python profiles/micro/c3_de_238.py"""
import os, subprocess, sys, tempfile
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import configs
from oracle import pipeline
from rhasspy_speech_amd import synth

U = int(sys.argv[1]) if len(sys.argv) > 1 else 238
KEY, TAG = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("de_DE-like", "c3_mixed_de")
tmp = Path(tempfile.mkdtemp())
m = configs.MIXED_MODELS[KEY]
md, gd = configs.build_grammar_model(tmp / "m", m["model_seed"], m["graph_seed"])
names, pcms = configs.mixed_utterances()
pcm = [p for nm, p in zip(names, pcms) if nm == KEY][U]
gw, gg, ga = configs.load_golden(TAG)
orc = pipeline.Oracle(md, gd)
tr = orc.transcribe(pcm)
own = tr.nbest[0]
# the reference's log-likelihoods for the same wav
bin_dir = ROOT / "oracle" / "_ref" / "bin"
env = dict(os.environ, PATH=f"{bin_dir}:{os.environ['PATH']}", OPENBLAS_NUM_THREADS="1")
wav = tmp / "u.wav"
synth.write_wav(wav, pcm)
dump = tmp / "dump"
dump.mkdir()
conf = md / "model" / "online" / "conf" / "online.conf"
subprocess.run(["rs-dump", f"--config={conf}", "--acoustic-scale=1.0", "offline", str(md / "model" / "model" / "final.mdl"), str(wav), str(dump)], env=env, check=True,
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
ref_ll = np.load(dump / "loglikes.npy")
lattice, ctr = pipeline.decode(orc.fst, orc.id2pdf, ref_ll, **orc.opts)
on_ref = pipeline.lat.nbest(lattice, 1, orc.opts["lattice_beam"], 1.0)[0]
d = np.abs(ref_ll - tr.loglikes)
print(f"{TAG} utterance {U}: {ref_ll.shape[0]} frames; log-likelihoods oracle vs reference: max |diff| {d.max():.2e}, mean {d.mean():.2e}")
print(f"  reference decoder (golden):                      words {list(gw[U])}  graph {gg[U]:.4f} acoustic {ga[U]:.4f} total {gg[U] + ga[U]:.4f}")
print(f"  oracle decoder on the ORACLE's log-likelihoods:  words {own.words}  graph {own.graph_cost:.4f} acoustic {own.acoustic_cost:.4f} total {own.graph_cost + own.acoustic_cost:.4f}")
print(f"  oracle decoder on the REFERENCE's log-likelihoods: words {on_ref.words}  graph {on_ref.graph_cost:.4f} acoustic {on_ref.acoustic_cost:.4f} total {on_ref.graph_cost + on_ref.acoustic_cost:.4f}")
ref_in = np.load(dump / "input.npy")
fd = np.abs(ref_in - tr.feats)
print(f"  features oracle vs reference: max |diff| {fd.max():.2e} (row {int(fd.max(1).argmax())}), 99th percentile {np.quantile(fd, 0.99):.2e}")
if (dump / "ivector.npy").exists():
    iv = np.load(dump / "ivector.npy")
    print(f"  iVector oracle vs reference: max |diff| {np.abs(iv[0] - tr.ivector).max():.2e}")
per_frame = d.max(1)
print("  frames with the largest log-likelihood differences:", [(int(i), float(f"{per_frame[i]:.2e}")) for i in np.argsort(-per_frame)[:6]])
# Where the iVectors part: the Gaussian posteriors are a DISCRETE function of the UBM scores (keep `like > max + log(min_post)`,
# the num_gselect best of those, drop the tail below min_post of the total).  Frames on which one of those comparisons is decided
# inside the FP32 rounding of the scores (float64 scores beside the FP32 ones):
ie = orc.ie
feats = tr.feats
cm = pipeline.online_cmvn(feats, ie["gstats"])
nrm = pipeline.lda_transform(pipeline.splice(cm, ie["left"], ie["right"]), ie["lda"])
g = ie["gmm"]
x64 = nrm.astype(np.float64)
ll64 = g.gconsts[None, :].astype(np.float64) + x64 @ g.means_invvars.T.astype(np.float64) - 0.5 * ((x64 * x64) @ g.inv_vars.T.astype(np.float64))
logmp = np.log(ie["min_post"])
close = []
for t, row in enumerate(ll64):
    mx = row.max()
    srt = np.sort(row)[::-1]
    margins = {"candidate cut (like vs max + log min_post)": np.abs(row - (mx + logmp)).min(),
               f"rank {ie['num_gselect']} vs {ie['num_gselect'] + 1}": srt[ie["num_gselect"] - 1] - srt[ie["num_gselect"]]}
    p = np.exp(srt[:ie["num_gselect"]] - mx)
    tot = p.sum()
    margins["tail below min_post of the total"] = np.abs(p - ie["min_post"] * tot).min() / max(p.min(), 1e-30) * 1.0
    for what, mg in margins.items():
        if mg < 2e-5:
            close.append((t, what, float(mg)))
print(f"  frames whose Gaussian selection is decided within 2e-5 of the scores (FP32 rounding of a score of magnitude ~{np.abs(ll64).mean():.0f} is {np.abs(ll64).mean() * 6e-8:.1e}):")
for c in close:
    print("   ", c)
