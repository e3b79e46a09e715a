"""Utterance sharding across the GPUs of one node (one process per GPU) and the path's single exchange step.

The reference has no parallelism at all (one process per utterance); utterances are fully
independent, so the multi-GPU design is: utterance i -> rank i % world, every rank runs the complete pipeline on
its shard with a replicated model, then ONE collective gathers fixed-size result records
(<= 62 word ids + 2 float costs = 264 B) to every rank: torch.distributed.all_gather, which is RCCL over xGMI
with the "nccl" backend on ROCm (and gloo on CPU in the tests).  Nothing else crosses GPUs.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

MAX_WORDS = 62
RECORD_INTS = 2 + MAX_WORDS + 2      # [global utt index, n_words, words..., graph cost bits, acoustic cost bits]


def shard_indices(n_utts: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_utts, world))


def pack_records(indices: Sequence[int], words: Sequence[Sequence[int]], costs: Sequence[Tuple[float, float]]) -> np.ndarray:
    rec = np.zeros((len(indices), RECORD_INTS), np.int32)
    for r, (i, w, c) in enumerate(zip(indices, words, costs)):
        w = list(w)[:MAX_WORDS]
        rec[r, 0] = i
        rec[r, 1] = len(w)
        rec[r, 2:2 + len(w)] = w
        rec[r, 2 + MAX_WORDS:] = np.array(c, np.float32).view(np.int32)
    return rec


def unpack_records(rec: np.ndarray):
    out = {}
    for row in rec:
        if row[1] < 0:
            continue
        n = int(row[1])
        g, a = row[2 + MAX_WORDS:].view(np.float32)
        out[int(row[0])] = ([int(x) for x in row[2:2 + n]], float(g), float(a))
    return out


def gather_records(local: np.ndarray, n_total: int, device=None):
    """all_gather of the per-rank record blocks (padded to the largest shard); returns {utt index: (words, g, a)}."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return unpack_records(local)
    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    pad = np.full((per, RECORD_INTS), 0, np.int32)
    pad[:, 1] = -1                     # padding rows are marked invalid
    pad[:local.shape[0]] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return unpack_records(torch.cat(outs).cpu().numpy())


def decode_mixed_sharded(models, utt_model: Sequence[str], pcm: Sequence[np.ndarray], rank: int = 0, world: int = 1, device=None):
    """A batch whose utterances name different models (BASELINE config 3: de_DE + fr_FR utterances in one batch), sharded
    over the ranks of one node -- SURVEY.md section 8(b)'s `rs_decode_batch_sharded`, on the host side, in the reference's
    own language; the communicator is torch.distributed's (RCCL with the "nccl" backend).

    `models` maps a name to a loaded model (`_lib.Model`, or anything with `decode_batch(list of int16 arrays)` returning
    an object with `words(u)` / `costs(u)`), resident on this rank's GPU; `utt_model[i]` names the model of utterance i.
    Utterance i belongs to rank i % world; a rank decodes its utterances of each model as ONE batch per model, the
    batches of different models concurrently from one host thread each (they overlap on the device), and one all_gather of
    the fixed-size 1-best records returns {utterance index: (word ids, graph cost, acoustic cost)} on every rank."""
    import threading
    if len(utt_model) != len(pcm):
        raise ValueError("utt_model and pcm differ in length")
    groups = {}
    for i in shard_indices(len(pcm), rank, world):
        if utt_model[i] not in models:
            raise KeyError(f"utterance {i} names model {utt_model[i]!r}, which is not loaded")
        groups.setdefault(utt_model[i], []).append(i)
    results, errors = {}, []

    def run(name, idx):
        try:
            results[name] = models[name].decode_batch([pcm[i] for i in idx])
        except Exception as e:          # re-raised on the calling thread, after every rank has reached the gather
            errors.append(e)

    threads = [threading.Thread(target=run, args=(name, idx)) for name, idx in groups.items()]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    indices, words, costs = [], [], []
    if not errors:
        for name, idx in groups.items():
            for u, i in enumerate(idx):
                indices.append(i)
                words.append(results[name].words(u))
                costs.append(results[name].costs(u))
    gathered = gather_records(pack_records(indices, words, costs), len(pcm), device)
    if errors:
        raise errors[0]
    return gathered
