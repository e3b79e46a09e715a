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
