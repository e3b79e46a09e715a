"""Utterance sharding across the GPUs of one node (one process per GPU) and the path's single exchange step.

The reference has no parallelism at all (one process per utterance); utterances are fully
independent, so the multi-GPU design is: utterance i -> rank i % world, every rank runs the complete pipeline on
its shard with a replicated model, then ONE collective gathers fixed-size result records (272 B: utterance index,
status, word count, <= 63 word ids, 2 float costs) to every rank.  Nothing else crosses GPUs.

Two routes to the same records:
  * `rs_decode_batch_sharded` (include/rhasspy_speech_hip.h), the C entry point: decode + ncclAllGather on an RCCL
    communicator handed in by the caller -- `decode_mixed_sharded(..., rccl_comm=ptr)`;
  * decode through the C entry point without a communicator, then `torch.distributed.all_gather` here ("nccl" backend =
    RCCL over xGMI on ROCm; gloo on CPU in the tests) -- the default when a process group is initialised.
Failures never go around the collective: every record carries a status, every rank takes part in the gather, and every
rank raises the same `ShardError` afterwards if any rank's batch failed.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

MAX_WORDS = 63
RECORD_INTS = 3 + MAX_WORDS + 2      # [utt index, status, n_words, words..., graph cost bits, acoustic cost bits] = rs_decode_batch_sharded's
STATUS_OK, STATUS_ABSENT = 0, 1


class ShardError(RuntimeError):
    """A rank's whole batch failed (device / model error).  Raised on EVERY rank after the gather."""


class ShardedResult(dict):
    """{utterance index: (word ids, graph cost, acoustic cost)} for the utterances that decoded; `errors` maps the others to
    their negative status (e.g. -4: "decoded no frames", the reference's per-utterance failure); `truncated` holds the
    utterances whose 1-best had more than MAX_WORDS words (ids cut to MAX_WORDS; the true count is in `num_words`)."""

    def __init__(self):
        super().__init__()
        self.errors: Dict[int, int] = {}
        self.truncated: set = set()
        self.num_words: Dict[int, int] = {}


def shard_indices(n_utts: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_utts, world))


def pack_records(indices: Sequence[int], words: Sequence[Optional[Sequence[int]]], costs: Sequence[Tuple[float, float]],
                 status: Optional[Sequence[int]] = None) -> np.ndarray:
    rec = np.zeros((len(indices), RECORD_INTS), np.int32)
    for r, (i, w, c) in enumerate(zip(indices, words, costs)):
        rec[r, 0] = i
        rec[r, 1] = STATUS_OK if status is None else status[r]
        if rec[r, 1] != STATUS_OK:
            continue
        w = list(w)
        rec[r, 2] = len(w)                      # the full length; more than MAX_WORDS = the ids below were cut
        w = w[:MAX_WORDS]
        rec[r, 3:3 + len(w)] = w
        rec[r, 3 + MAX_WORDS:] = np.array(c, np.float32).view(np.int32)
    return rec


def unpack_records(rec: np.ndarray, n_total: Optional[int] = None, require_all: bool = True) -> ShardedResult:
    out = ShardedResult()
    seen = set()
    for row in rec:
        i, st = int(row[0]), int(row[1])
        if i < 0 or st == STATUS_ABSENT:
            continue
        seen.add(i)
        if st != STATUS_OK:
            out.errors[i] = st
            continue
        n = int(row[2])
        out.num_words[i] = n
        if n > MAX_WORDS:
            out.truncated.add(i)
            n = MAX_WORDS
        g, a = row[3 + MAX_WORDS:].view(np.float32)
        out[i] = ([int(x) for x in row[3:3 + n]], float(g), float(a))
    if require_all and n_total is not None and len(seen) != n_total:
        missing = sorted(set(range(n_total)) - seen)
        raise ShardError(f"gather returned {len(seen)} of {n_total} utterances (first missing: {missing[:5]})")
    return out


def gather_records(local: np.ndarray, n_total: int, device=None) -> np.ndarray:
    """all_gather of the per-rank record blocks (padded to the largest shard) -> all ranks' records, padding rows marked
    absent.  One rank / no process group: the local block."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    pad = np.zeros((per, RECORD_INTS), np.int32)
    pad[:, 0] = -1
    pad[:, 1] = STATUS_ABSENT
    pad[:local.shape[0]] = local
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return torch.cat(outs).cpu().numpy()


def _decode_local_python(models, utt_model, pcm, rank, world) -> Tuple[np.ndarray, List[str]]:
    """This rank's records through `model.decode_batch` of arbitrary model objects (anything with decode_batch(list of int16
    arrays) returning an object with words(u) / costs(u)): used for stand-in models in the CPU tests."""
    import threading
    groups: Dict[str, List[int]] = {}
    for i in shard_indices(len(pcm), rank, world):
        groups.setdefault(utt_model[i], []).append(i)
    results, failed = {}, {}

    def run(name, idx):
        try:
            results[name] = models[name].decode_batch([pcm[i] for i in idx])
        except Exception as e:          # reported on every rank after the gather
            failed[name] = e

    threads = [threading.Thread(target=run, args=(name, idx)) for name, idx in groups.items()]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    indices, words, costs, status = [], [], [], []
    for name, idx in groups.items():
        for u, i in enumerate(idx):
            indices.append(i)
            if name in failed:
                words.append(None); costs.append((0.0, 0.0)); status.append(int(getattr(failed[name], "status", -3)) or -3)
                continue
            try:
                words.append(results[name].words(u)); costs.append(results[name].costs(u)); status.append(STATUS_OK)
            except Exception as e:      # a per-utterance failure (e.g. "decoded no frames")
                words.append(None); costs.append((0.0, 0.0)); status.append(int(getattr(e, "status", -4)) or -4)
    return pack_records(indices, words, costs, status), [f"{k}: {v}" for k, v in failed.items()]


def decode_mixed_sharded(models, utt_model: Sequence[str], pcm: Sequence[np.ndarray], rank: int = 0, world: int = 1, device=None,
                         rccl_comm: int = 0, gather: bool = True) -> ShardedResult:
    """A batch whose utterances name different models (BASELINE config 3: de_DE + fr_FR utterances in one batch), sharded
    over the ranks of one node.

    `models` maps a name to a loaded model resident on this rank's GPU; `utt_model[i]` names the model of utterance i.
    Utterance i belongs to rank i % world; a rank decodes its utterances of each model as ONE batch per model, the batches
    of different models concurrently, and one gather of the fixed-size 1-best records returns the whole batch on every rank
    (`gather=False`: this rank's utterances only, no collective).  `_lib.Model`s go through the C entry point
    `rs_decode_batch_sharded` -- with `rccl_comm` (an ncclComm_t as integer, e.g. ProcessGroupNCCL._comm_ptr()) the library
    issues the ncclAllGather itself; without it the records are gathered with torch.distributed here."""
    from . import _lib
    if len(utt_model) != len(pcm):
        raise ValueError("utt_model and pcm differ in length")
    for i, nm in enumerate(utt_model):
        if nm not in models:
            raise KeyError(f"utterance {i} names model {nm!r}, which is not loaded")
    n = len(pcm)
    native = all(isinstance(m, _lib.Model) for m in models.values())
    problems: List[str] = []
    if native:
        names = list(models)
        idx = {nm: k for k, nm in enumerate(names)}
        rec, st, msg = _lib.decode_batch_sharded([models[nm] for nm in names], [idx[nm] for nm in utt_model], pcm, rank, world,
                                                 rccl_comm if gather else 0)
        if st != 0:
            problems.append(msg)
        if gather and world > 1 and not rccl_comm:
            rec = gather_records(rec[rank::world], n, device)
    else:
        local, problems = _decode_local_python(models, utt_model, pcm, rank, world)
        rec = gather_records(local, n, device) if gather else local
    out = unpack_records(rec, n, require_all=gather)
    batch_failures = {i: s for i, s in out.errors.items() if s in (-1, -2, -3)}
    if batch_failures or problems:
        first = min(batch_failures) if batch_failures else -1
        raise ShardError(f"{len(batch_failures)} utterances were not decoded because a rank's batch failed (first: utterance {first}, "
                         f"status {batch_failures.get(first)}); local errors: {problems}")
    return out
