import sys, numpy as np
sys.path.insert(0, '/root/repo')
from tests import configs
from rhasspy_speech_amd import _lib
md, gd = configs.build_arpa_model('/tmp/arpa_model')
model = _lib.Model(md, gd, _lib.default_opts())
pcms = configs.arpa_utterances()
res = model.decode_batch(pcms)
w, g, a = configs.load_golden('c2_arpa')
for u in range(256):
    gc, ac = res.costs(u)
    if abs(ac - a[u]) > 2e-3 + 2e-4 * abs(a[u]) or abs(gc - g[u]) > 2e-3 + 2e-4*abs(g[u]) or res.words(u) != w[u]:
        print('MISMATCH', u, 'got', gc, ac, gc + ac, 'ref', g[u], a[u], g[u] + a[u], res.words(u) == w[u], res.counters(u))
        nb = model.decode_batch([pcms[u]], nbest=3)
        for k in range(nb.num_hyps(0)):
            print('  lattice path', k, nb.words(0, k) == w[u], nb.costs(0, k))
