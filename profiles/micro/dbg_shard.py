import sys, tempfile, ctypes as C
from pathlib import Path
import numpy as np
sys.path.insert(0, ".")
import torch
from rhasspy_speech_amd import _lib, shard
from tests import configs
md, gd = configs.build_grammar_model(Path(tempfile.gettempdir()) / "rs_dbg_grammar")
model = _lib.Model(md, gd, _lib.default_opts())
pcms = configs.grammar_utterances(24) + [np.zeros(200, np.int16)]
print("A plain 24", flush=True)
r = model.decode_batch(pcms[:24]); print(r.words(0), flush=True)
print("B plain 25 with short", flush=True)
try:
    r = model.decode_batch(pcms); print(r.words(0), flush=True)
except Exception as e:
    print("exc", e, flush=True)
print("C sharded no comm", flush=True)
rec, st, msg = _lib.decode_batch_sharded([model], [0] * len(pcms), pcms, 0, 1, 0); print(st, msg, rec[:2, :6], flush=True)
rccl = C.CDLL("librccl.so.1")
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UniqueId()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
print("D sharded comm", flush=True)
rec2, st, msg = _lib.decode_batch_sharded([model], [0] * len(pcms), pcms, 0, 1, comm.value); print(st, msg, flush=True)
print("equal", np.array_equal(rec, rec2), flush=True)
print("E gather alone", flush=True)
rec3 = _lib.shard_gather(rec, 0, 0, 1, comm.value); print("equal", np.array_equal(rec, rec3), flush=True)
