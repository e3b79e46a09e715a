"""Headline batch with the samples resident in HBM (rs_decode_batch_device) and in host memory (rs_decode_batch) against the number of
calls in flight: ms per step over 300 steps each.  usage (GPU box): python profiles/micro/device_inflight.py"""
import concurrent.futures, os, sys, time, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import configs
from rhasspy_speech_amd import _lib

cache = pathlib.Path(os.environ.get("RS_BENCH_CACHE", "/tmp/rs_bench_cache")); cache.mkdir(parents=True, exist_ok=True)
model_dir, graph_dir = configs.build_grammar_model(cache)
pcms = configs.grammar_utterances(256, 0)
model = _lib.Model(model_dir, graph_dir, _lib.default_opts(device_id=0, prune_output_pdfs=1))
model.to_device()
d_pcm = torch.from_numpy(np.concatenate(pcms)).to("cuda:0")
offsets = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)
dev = lambda: model.decode_batch_device(d_pcm.data_ptr(), offsets)
host = lambda: model.decode_batch(pcms)

def run(fn, inflight, n):
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=inflight)
    list(pool.map(lambda _: fn().pack(64), range(2 * inflight)))
    torch.cuda.synchronize()
    t = time.perf_counter()
    fs = [pool.submit(fn) for _ in range(min(n, inflight))]
    for k in range(n):
        fs[k].result().pack(64)
        if len(fs) < n: fs.append(pool.submit(fn))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    pool.shutdown()
    return 1e3 * dt / n

for name, fn in (("device", dev), ("host", host)):
    for inflight in (2, 3, 4, 5, 6):
        print(name, "inflight", inflight, "ms/step %.3f" % run(fn, inflight, 300), flush=True)
