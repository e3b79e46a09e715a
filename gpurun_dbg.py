import sys, os, tempfile, pathlib, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import cases
from rhasspy_speech_amd import _lib
from oracle import pipeline
from oracle.pipeline import *
root = pathlib.Path(tempfile.mkdtemp())
model_dir, graph_dir, wav, pcm = cases.build_case_files(cases.CASES["tiny_u0"], root)
os.environ["RS_DEBUG_IVEC"] = "1"
m = _lib.Model(model_dir, graph_dir, _lib.default_opts(keep_intermediates=1))
res = m.decode_batch([pcm])
print("gpu ivec", res.matrix(0, 1)[0][:6])
o = pipeline.Oracle(model_dir, graph_dir)
feats = o.features(pcm)
ie = o.ie
cm = online_cmvn(feats, ie["gstats"])
T = feats.shape[0]
raw = lda_transform(splice(feats, ie["left"], ie["right"]), ie["lda"])
nrm = lda_transform(splice(cm, ie["left"], ie["right"]), ie["lda"])
posts = ubm_posteriors(nrm, ie["gmm"], ie["num_gselect"], ie["min_post"], ie["posterior_scale"])
for t in range(3): print("opost", t, [(g, float(w)) for g, w in posts[t]])
st = IvectorStats(ie["ext"], ie["max_count"])
st.acc(raw, posts)
print("o numf", st.num_frames, "lin", st.lin[:6], "quad", st.quad[0,0], st.quad[1,0], st.quad[1,1], st.quad[2,0], st.quad[2,1], st.quad[2,2])
print("o ivec", o.offline_ivector(feats)[:6])
