"""ctypes binding of librhasspy_speech_hip.so (include/rhasspy_speech_hip.h).

This is the stub a maintainer of the reference would add in place of the subprocess pipeline of
rhasspy_speech/transcribe_wav.py:45-75.  The library is the only compute path: if it is
missing, import fails loudly; if no MI355X is present, every decode call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

_LIB_PATH = Path(__file__).resolve().parent / "librhasspy_speech_hip.so"

RS_OK, RS_ERR_ARG, RS_ERR_MODEL, RS_ERR_DEVICE, RS_ERR_DECODE = 0, -1, -2, -3, -4
RS_OPT_UNSET = -1      # rs_decode_opts field not given on the "command line": online.conf's value, else the reference's default


class DecodeOpts(C.Structure):
    _fields_ = [
        ("beam", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32), ("lattice_beam", C.c_float),
        ("beam_delta", C.c_float), ("acoustic_scale", C.c_float), ("frames_per_chunk", C.c_int32),
        ("frame_subsampling_factor", C.c_int32), ("device_id", C.c_int32), ("keep_intermediates", C.c_int32),
        ("max_tokens_per_frame", C.c_int32), ("emit_lattice", C.c_int32), ("prune_output_pdfs", C.c_int32), ("exact_token_order", C.c_int32), ("command_line_fixed", C.c_int32), ("stream_min_ticks", C.c_int32), ("reserved", C.c_int32 * 2),
    ]


FIXED_ONLINE, FIXED_DO_ENDPOINTING, FIXED_EXTRA_LEFT_CONTEXT_INITIAL, FIXED_PRUNE_INTERVAL, FIXED_DETERMINIZE_LATTICE = 1, 2, 4, 8, 16      # RS_FIXED_*

EXPORTS = [
    "rs_default_opts", "rs_last_error", "rs_model_load_files", "rs_model_load", "rs_model_to_device", "rs_model_free",
    "rs_model_describe", "rs_model_check_sample_rate", "rs_decode_batch", "rs_decode_batch_device", "rs_decode_batch_sharded", "rs_shard_gather", "rs_stream_open", "rs_stream_accept",
    "rs_stream_finish", "rs_stream_free", "rs_streams_accept", "rs_streams_advance", "rs_streams_finish", "rs_result_num_utts", "rs_result_num_hyps",
    "rs_result_num_frames", "rs_result_words", "rs_result_costs", "rs_result_text", "rs_result_lattice", "rs_result_matrix",
    "rs_result_counters", "rs_result_timings", "rs_result_pack", "rs_result_free",
    "rs_mkgraph", "rs_fst_tool", "rs_fuzzy_open", "rs_fuzzy_match", "rs_result_fuzzy", "rs_fuzzy_free", "rs_lattice_entry_from_raw",
    "rs_rescorer_open", "rs_rescore_result", "rs_rescore_lattice", "rs_rescorer_free",
    "rs_nnet3_setup", "rs_nnet3_setup_subsampled", "rs_dither_noise", "rs_bind_host_thread",
]


def load_library() -> C.CDLL:
    if not _LIB_PATH.exists():
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C rhasspy_speech_amd/csrc).  There is no CPU fallback.")
    lib = C.CDLL(str(_LIB_PATH))
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.rs_default_opts.argtypes = [C.POINTER(DecodeOpts)]
    lib.rs_last_error.restype = C.c_char_p
    lib.rs_model_load_files.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(DecodeOpts), C.POINTER(vp)]
    lib.rs_model_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(DecodeOpts), C.POINTER(vp)]
    lib.rs_model_to_device.argtypes = [vp]
    lib.rs_model_free.argtypes = [vp]
    lib.rs_model_free.restype = None
    lib.rs_model_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.rs_model_check_sample_rate.argtypes = [vp, f32]
    lib.rs_decode_batch.argtypes = [vp, C.POINTER(C.POINTER(C.c_int16)), C.POINTER(i32), i32, i32, f32, C.POINTER(vp)]
    lib.rs_decode_batch_device.argtypes = [vp, vp, C.POINTER(C.c_int64), i32, i32, f32, vp, C.POINTER(vp)]
    lib.rs_decode_batch_sharded.argtypes = [C.POINTER(vp), i32, C.POINTER(i32), C.POINTER(C.POINTER(C.c_int16)), C.POINTER(i32), i32, i32, i32, vp,
                                            C.POINTER(i32)]
    lib.rs_shard_gather.argtypes = [i32, i32, i32, i32, vp, C.POINTER(i32)]
    lib.rs_stream_open.argtypes = [vp, C.POINTER(vp)]
    lib.rs_stream_accept.argtypes = [vp, C.POINTER(C.c_int16), i32]
    lib.rs_stream_finish.argtypes = [vp, i32, f32, C.POINTER(vp)]
    lib.rs_stream_free.argtypes = [vp]
    lib.rs_stream_free.restype = None
    lib.rs_streams_accept.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), i32]
    lib.rs_streams_advance.argtypes = [C.POINTER(vp), i32]
    lib.rs_streams_finish.argtypes = [C.POINTER(vp), i32, i32, f32, C.POINTER(vp)]
    lib.rs_result_num_utts.argtypes = [vp]
    lib.rs_result_num_hyps.argtypes = [vp, i32]
    lib.rs_result_num_frames.argtypes = [vp, i32]
    lib.rs_result_words.argtypes = [vp, i32, i32, C.POINTER(C.POINTER(i32)), C.POINTER(i32)]
    lib.rs_result_costs.argtypes = [vp, i32, i32, C.POINTER(f32), C.POINTER(f32)]
    lib.rs_result_text.argtypes = [vp, i32, C.c_char_p, C.c_char_p, C.c_size_t]
    lib.rs_result_lattice.argtypes = [vp, i32, C.c_char_p, C.c_char_p, C.c_int64]
    lib.rs_result_lattice.restype = C.c_int64
    lib.rs_result_matrix.argtypes = [vp, i32, i32, C.POINTER(C.POINTER(f32)), C.POINTER(i32), C.POINTER(i32)]
    lib.rs_result_counters.argtypes = [vp, i32, C.POINTER(C.c_int64)]
    lib.rs_result_timings.argtypes = [vp, C.POINTER(f32)]
    lib.rs_result_pack.argtypes = [vp, i32, C.POINTER(i32)]
    lib.rs_result_free.argtypes = [vp]
    lib.rs_result_free.restype = None
    lib.rs_mkgraph.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_float, C.c_float, C.c_char_p]
    lib.rs_fst_tool.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_float]
    lib.rs_fuzzy_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.rs_fuzzy_match.argtypes = [vp, C.c_char_p, C.POINTER(i32), i32, C.POINTER(i32), C.POINTER(C.c_double)]
    lib.rs_result_fuzzy.argtypes = [vp, i32, vp, C.POINTER(i32), i32, C.POINTER(i32), C.POINTER(C.c_double)]
    lib.rs_fuzzy_free.argtypes = [vp]
    lib.rs_fuzzy_free.restype = None
    lib.rs_rescorer_open.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    lib.rs_rescore_result.argtypes = [vp, vp, i32, i32, f32, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(f32), C.POINTER(f32), C.POINTER(i32)]
    lib.rs_rescore_lattice.argtypes = [vp, C.c_char_p, C.c_size_t, i32, f32, C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(f32), C.POINTER(f32),
                                       C.POINTER(i32)]
    lib.rs_rescorer_free.argtypes = [vp]
    lib.rs_rescorer_free.restype = None
    pf, pi = C.POINTER(f32), C.POINTER(i32)
    lib.rs_lattice_entry_from_raw.argtypes = [i32, i32, pf, i32, pi, pi, pi, pi, pf, pf, f32, C.c_char_p, C.c_char_p, C.c_int64]
    lib.rs_lattice_entry_from_raw.restype = C.c_int64
    return lib


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib


class RsError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


def _check(status: int) -> None:
    if status != RS_OK:
        raise RsError(status, lib().rs_last_error().decode("utf-8", "replace"))


def default_opts(**overrides) -> DecodeOpts:
    o = DecodeOpts()
    _check(lib().rs_default_opts(C.byref(o)))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown decode option {k}")
        setattr(o, k, v)
    return o


class Result:
    """Owns an rs_result."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle

    def close(self) -> None:
        if self._h:
            lib().rs_result_free(self._h)
            self._h = None

    __del__ = close

    @property
    def num_utts(self) -> int:
        return lib().rs_result_num_utts(self._h)

    def num_hyps(self, utt: int) -> int:
        return lib().rs_result_num_hyps(self._h, utt)

    def num_frames(self, utt: int) -> int:
        return lib().rs_result_num_frames(self._h, utt)

    def words(self, utt: int, k: int = 0) -> List[int]:
        ids = C.POINTER(C.c_int32)()
        n = C.c_int32()
        _check(lib().rs_result_words(self._h, utt, k, C.byref(ids), C.byref(n)))
        return [ids[i] for i in range(n.value)]

    def costs(self, utt: int, k: int = 0) -> Tuple[float, float]:
        g, a = C.c_float(), C.c_float()
        _check(lib().rs_result_costs(self._h, utt, k, C.byref(g), C.byref(a)))
        return g.value, a.value

    def text(self, utt: int, key: str = "utt") -> bytes:
        """The bytes `nbest-to-linear ... ark,t:-` would print for this utterance."""
        n = lib().rs_result_text(self._h, utt, key.encode(), None, 0)
        if n < 0:
            _check(n)
        buf = C.create_string_buffer(n + 1)
        lib().rs_result_text(self._h, utt, key.encode(), buf, n + 1)
        return buf.value

    def lattice(self, utt: int, key: str = "utt") -> bytes:
        """One binary CompactLattice table entry (what online2-wav-nnet3-latgen-faster writes); needs emit_lattice=1."""
        n = lib().rs_result_lattice(self._h, utt, key.encode(), None, 0)
        if n < 0:
            _check(int(n))
        buf = C.create_string_buffer(int(n))
        lib().rs_result_lattice(self._h, utt, key.encode(), buf, n)
        return buf.raw

    def matrix(self, utt: int, kind: int) -> np.ndarray:
        data = C.POINTER(C.c_float)()
        r, c = C.c_int32(), C.c_int32()
        _check(lib().rs_result_matrix(self._h, utt, kind, C.byref(data), C.byref(r), C.byref(c)))
        if r.value * c.value == 0:
            return np.zeros((r.value, c.value), np.float32)
        return np.ctypeslib.as_array(data, shape=(r.value, c.value)).copy()

    def counters(self, utt: int) -> List[int]:
        out = (C.c_int64 * 8)()
        _check(lib().rs_result_counters(self._h, utt, out))
        return list(out)

    def pack(self, max_words: int = 62) -> np.ndarray:
        """(num_utts, max_words + 4) int32 records: status, n_words, words, graph/acoustic cost bits (for the gather)."""
        out = np.zeros((self.num_utts, max_words + 4), np.int32)
        _check(lib().rs_result_pack(self._h, max_words, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def timings(self) -> List[float]:
        out = (C.c_float * 8)()
        _check(lib().rs_result_timings(self._h, out))
        return list(out)


class Model:
    """Owns an rs_model (parsed Kaldi model + HCLG; device copy made on first decode)."""

    def __init__(self, model_dir: Optional[os.PathLike] = None, graph_dir: Optional[os.PathLike] = None,
                 opts: Optional[DecodeOpts] = None, *, final_mdl=None, hclg=None, online_conf=None):
        self._h = C.c_void_p()
        self.opts = opts or default_opts()
        if final_mdl is not None:
            _check(lib().rs_model_load_files(str(final_mdl).encode(), str(hclg).encode(), str(online_conf).encode(),
                                             C.byref(self.opts), C.byref(self._h)))
        else:
            _check(lib().rs_model_load(str(model_dir).encode(), str(graph_dir).encode(), C.byref(self.opts), C.byref(self._h)))

    def close(self) -> None:
        if self._h:
            lib().rs_model_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except TypeError:      # interpreter shutdown: the module's globals are gone already, and the process's memory goes with it
            pass

    def to_device(self) -> None:
        _check(lib().rs_model_to_device(self._h))

    def describe(self) -> str:
        n = lib().rs_model_describe(self._h, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib().rs_model_describe(self._h, buf, n + 1)
        return buf.value.decode()

    def check_sample_rate(self, sample_rate: float) -> None:
        """Raises RsError with Kaldi's "Sampling frequency mismatch ..." unless `sample_rate` is the model's --sample-frequency."""
        _check(lib().rs_model_check_sample_rate(self._h, float(sample_rate)))

    def decode_batch(self, pcm: Sequence[np.ndarray], nbest: int = 1, lattice_acoustic_scale: float = 1.0) -> Result:
        n = len(pcm)
        addr = np.zeros(max(n, 1), np.uintp)
        lens_np = np.zeros(max(n, 1), np.int32)
        keep = []
        for i, p in enumerate(pcm):       # (buffers that are int16 and contiguous already are handed over as they are)
            if not (isinstance(p, np.ndarray) and p.dtype == np.int16 and p.flags.c_contiguous):
                p = np.ascontiguousarray(p, dtype=np.int16)
                keep.append(p)
            addr[i] = p.__array_interface__["data"][0]
            lens_np[i] = p.shape[0]
        ptrs = addr.ctypes.data_as(C.POINTER(C.POINTER(C.c_int16)))
        lens = lens_np.ctypes.data_as(C.POINTER(C.c_int32))
        out = C.c_void_p()
        _check(lib().rs_decode_batch(self._h, ptrs, lens, n, nbest, lattice_acoustic_scale, C.byref(out)))
        return Result(out)

    def decode_batch_device(self, d_pcm_ptr: int, sample_offsets: np.ndarray, nbest: int = 1,
                            lattice_acoustic_scale: float = 1.0, stream: int = 0) -> Result:
        """d_pcm_ptr: device address (e.g. torch_tensor.data_ptr()) of all utterances back to back (int16)."""
        off = np.ascontiguousarray(sample_offsets, dtype=np.int64)
        out = C.c_void_p()
        _check(lib().rs_decode_batch_device(self._h, C.c_void_p(d_pcm_ptr), off.ctypes.data_as(C.POINTER(C.c_int64)),
                                            off.shape[0] - 1, nbest, lattice_acoustic_scale, C.c_void_p(stream), C.byref(out)))
        return Result(out)


SHARD_MAX_WORDS = 63
SHARD_RECORD_INTS = 3 + SHARD_MAX_WORDS + 2
SHARD_ABSENT = 1


def decode_batch_sharded(models: Sequence[Model], utt_model: Sequence[int], pcm: Sequence[Optional[np.ndarray]], rank: int = 0, world: int = 1,
                         rccl_comm: int = 0) -> Tuple[np.ndarray, int, str]:
    """rs_decode_batch_sharded: -> (records (n_utts, SHARD_RECORD_INTS) int32, status of this rank, its error text).
    pcm[i] may be None for utterances of other ranks.  rccl_comm = ncclComm_t as an integer (0: no collective, own records
    only).  Never raises for a decode failure: the status travels in the records so that every rank sees it."""
    n = len(pcm)
    # (a thousand utterances per call: no per-utterance numpy / ctypes object is made for buffers that are usable as they are)
    addr = np.zeros(max(n, 1), np.uintp)
    lens_np = np.zeros(max(n, 1), np.int32)
    keep = []
    for i in range(rank, n, world):
        p = pcm[i]
        if p is None:
            continue
        if not (isinstance(p, np.ndarray) and p.dtype == np.int16 and p.flags.c_contiguous):
            p = np.ascontiguousarray(p, dtype=np.int16)
            keep.append(p)
        addr[i] = p.__array_interface__["data"][0]
        lens_np[i] = p.shape[0]
    ptrs = addr.ctypes.data_as(C.POINTER(C.POINTER(C.c_int16)))
    lens = lens_np.ctypes.data_as(C.POINTER(C.c_int32))
    um = np.ascontiguousarray(utt_model, dtype=np.int32)
    handles = (C.c_void_p * len(models))(*[m._h for m in models])
    rec = np.zeros((n, SHARD_RECORD_INTS), np.int32)
    st = lib().rs_decode_batch_sharded(handles, len(models), um.ctypes.data_as(C.POINTER(C.c_int32)), ptrs, lens, n, rank, world,
                                       C.c_void_p(rccl_comm or None), rec.ctypes.data_as(C.POINTER(C.c_int32)))
    if st == RS_ERR_ARG:
        _check(st)
    return rec, st, ("" if st == RS_OK else lib().rs_last_error().decode("utf-8", "replace"))


def shard_gather(records: np.ndarray, device_id: int, rank: int, world: int, rccl_comm: int) -> np.ndarray:
    """rs_shard_gather: this rank's records (at their utterance indices, as decode_batch_sharded(..., rccl_comm=0) returns
    them) -> every rank's, by ONE ncclAllGather on `rccl_comm`.  Call it from one thread, in the same order on every rank."""
    rec = np.ascontiguousarray(records, dtype=np.int32).copy()
    _check(lib().rs_shard_gather(device_id, rec.shape[0], rank, world, C.c_void_p(rccl_comm), rec.ctypes.data_as(C.POINTER(C.c_int32))))
    return rec


def bind_host_thread(device_id: int) -> int:
    """rs_bind_host_thread: the calling thread (and the threads it starts from now on) stays on the CPUs local to GPU
    `device_id`; returns how many CPUs that is (0: the system names none, nothing changed).  RS_BIND_CPULIST overrides."""
    lib().rs_bind_host_thread.argtypes = [C.c_int32]
    n = lib().rs_bind_host_thread(device_id)
    if n < 0:
        _check(n)
    return n


class Stream:
    """Owns an rs_stream: the stdin of one online2-cli-nnet3-decode-faster process."""

    def __init__(self, model: Model):
        self.model = model
        self._h = C.c_void_p()
        _check(lib().rs_stream_open(model._h, C.byref(self._h)))

    def accept(self, pcm) -> None:
        a = np.ascontiguousarray(np.frombuffer(pcm, dtype="<i2") if isinstance(pcm, (bytes, bytearray, memoryview)) else pcm, dtype=np.int16)
        _check(lib().rs_stream_accept(self._h, a.ctypes.data_as(C.POINTER(C.c_int16)), a.shape[0]))

    def advance(self) -> None:
        """Run the device work the audio accepted so far makes possible (rs_streams_advance on this one stream)."""
        arr = (C.c_void_p * 1)(self._h)
        _check(lib().rs_streams_advance(arr, 1))

    def finish(self, nbest: int = 1, lattice_acoustic_scale: float = 1.0) -> Result:
        out = C.c_void_p()
        _check(lib().rs_stream_finish(self._h, nbest, lattice_acoustic_scale, C.byref(out)))
        return Result(out)

    def close(self) -> None:
        if self._h:
            lib().rs_stream_free(self._h)
            self._h = C.c_void_p()

    __del__ = close


def accept_streams(streams: Sequence[Stream], chunks: Sequence[np.ndarray]) -> None:
    """rs_streams_accept: chunks[i] (int16 samples, contiguous) to streams[i], one call for all of them."""
    n = len(streams)
    arr = (C.c_void_p * n)(*[s._h for s in streams])
    ptrs = (C.c_void_p * n)()
    lens = (C.c_int32 * n)()
    keep = []
    for i, a in enumerate(chunks):
        if a.dtype != np.int16 or not a.flags.c_contiguous:
            a = np.ascontiguousarray(a, dtype=np.int16)
            keep.append(a)
        ptrs[i] = a.ctypes.data
        lens[i] = a.shape[0]
    _check(lib().rs_streams_accept(arr, ptrs, lens, n))


def advance_streams(streams: Sequence[Stream]) -> None:
    """rs_streams_advance: run the device work that the audio accepted so far makes possible, batched over the streams."""
    arr = (C.c_void_p * len(streams))(*[s._h for s in streams])
    _check(lib().rs_streams_advance(arr, len(streams)))


def stream_handles(streams: Sequence[Stream]) -> np.ndarray:
    """The streams' rs_stream handles as an array of addresses, for the *_raw calls below."""
    return np.fromiter((s._h.value or 0 for s in streams), dtype=np.uintp, count=len(streams))


def accept_streams_raw(handles: np.ndarray, addrs: np.ndarray, lens: np.ndarray) -> None:
    """rs_streams_accept on arrays the caller keeps: handles[i] (stream_handles), addrs[i] = address of int16 samples, lens[i] =
    their number.  A host program that feeds many streams per round does this pointer arithmetic itself; building the arrays
    from Python objects per round (accept_streams) costs more than the call."""
    n = int(handles.shape[0])
    assert handles.dtype == np.uintp and addrs.dtype == np.uintp and lens.dtype == np.int32 and addrs.shape[0] == n and lens.shape[0] == n
    _check(lib().rs_streams_accept(handles.ctypes.data_as(C.POINTER(C.c_void_p)), addrs.ctypes.data_as(C.POINTER(C.c_void_p)),
                                   lens.ctypes.data_as(C.POINTER(C.c_int32)), n))


def advance_streams_raw(handles: np.ndarray) -> None:
    """rs_streams_advance on an array of handles (stream_handles)."""
    assert handles.dtype == np.uintp
    _check(lib().rs_streams_advance(handles.ctypes.data_as(C.POINTER(C.c_void_p)), int(handles.shape[0])))


def finish_streams(streams: Sequence[Stream], nbest: int = 1, lattice_acoustic_scale: float = 1.0) -> Result:
    """Ends all streams (EOF) and decodes them as one device batch; utterance i of the result = streams[i]."""
    arr = (C.c_void_p * len(streams))(*[s._h for s in streams])
    out = C.c_void_p()
    _check(lib().rs_streams_finish(arr, len(streams), nbest, lattice_acoustic_scale, C.byref(out)))
    return Result(out)


def mkgraph(lang_dir, model_dir, graph_dir, self_loop_scale: float = 0.1, transition_scale: float = 1.0, dump_dir=None) -> None:
    """utils/mkgraph.sh [--self-loop-scale S] [--transition-scale T] <lang_dir> <model_dir> <graph_dir> (defaults as the script's)."""
    _check(lib().rs_mkgraph(str(lang_dir).encode(), str(model_dir).encode(), str(graph_dir).encode(), transition_scale, self_loop_scale,
                            None if dump_dir is None else str(dump_dir).encode()))


def fst_tool(tool: str, in1=None, in2=None, out=None, aux=None, param: float = 0.0) -> None:
    """One step of the graph-construction chain on files (rs_fst_tool)."""
    enc = lambda p: None if p is None else str(p).encode()
    _check(lib().rs_fst_tool(tool.encode(), enc(in1), enc(in2), enc(out), enc(aux), param))


class FuzzyMatcher:
    """Owns an rs_fuzzy: <lang_dir>/G.fuzzy.fst parsed once (host side, no GPU)."""

    def __init__(self, fuzzy_fst_path):
        self._h = C.c_void_p()
        _check(lib().rs_fuzzy_open(str(fuzzy_fst_path).encode(), C.byref(self._h)))

    def match(self, nbest_text: bytes):
        """(output word ids, cost) of the cheapest fuzzy path, or None where the reference's get_fuzzy_text returns None."""
        cap = 256
        while True:
            buf = (C.c_int32 * cap)()
            n, cost = C.c_int32(), C.c_double()
            _check(lib().rs_fuzzy_match(self._h, bytes(nbest_text), buf, cap, C.byref(n), C.byref(cost)))
            if n.value < 0:
                return None
            if n.value <= cap:
                return list(buf[:n.value]), cost.value
            cap = n.value

    def match_result(self, result: "Result", utt: int = 0):
        """The same, on the hypotheses of a decode result (rs_result_fuzzy: no text round trip)."""
        cap = 256
        while True:
            buf = (C.c_int32 * cap)()
            n, cost = C.c_int32(), C.c_double()
            _check(lib().rs_result_fuzzy(result._h, utt, self._h, buf, cap, C.byref(n), C.byref(cost)))
            if n.value < 0:
                return None
            if n.value <= cap:
                return list(buf[:n.value]), cost.value
            cap = n.value

    def close(self) -> None:
        if self._h:
            lib().rs_fuzzy_free(self._h)
            self._h = C.c_void_p()

    __del__ = close


class Rescorer:
    """Owns an rs_rescorer: <new_lang_dir>/{L_disambig.fst, G.fst, words.txt, phones/disambig.int} parsed once (host side)."""

    def __init__(self, model: Model, new_lang_dir):
        self.model = model
        self._h = C.c_void_p()
        _check(lib().rs_rescorer_open(model._h, str(new_lang_dir).encode(), C.byref(self._h)))

    def _call(self, fn, *head, nbest: int, acoustic_scale: float, key: str):
        g, a, n = (C.c_float * nbest)(), (C.c_float * nbest)(), C.c_int32()
        size = fn(self._h, *head, nbest, acoustic_scale, key.encode(), None, 0, g, a, C.byref(n))
        if size < 0:
            _check(size)
        buf = C.create_string_buffer(size + 1)
        fn(self._h, *head, nbest, acoustic_scale, key.encode(), buf, size + 1, g, a, C.byref(n))
        return buf.value, list(g[:n.value]), list(a[:n.value])

    def rescore(self, result: Result, utt: int = 0, nbest: int = 1, acoustic_scale: float = 1.0, key: str = "utt"):
        """-> (nbest text bytes as nbest-to-linear prints them, graph costs, acoustic costs) for one utterance of a result decoded
        with emit_lattice = 1."""
        return self._call(lib().rs_rescore_result, result._h, utt, nbest=nbest, acoustic_scale=acoustic_scale, key=key)

    def rescore_lattice(self, entry: bytes, nbest: int = 1, acoustic_scale: float = 1.0, key: str = "utt"):
        """The same on one binary CompactLattice table entry (no GPU)."""
        return self._call(lib().rs_rescore_lattice, entry, len(entry), nbest=nbest, acoustic_scale=acoustic_scale, key=key)

    def close(self) -> None:
        if self._h:
            lib().rs_rescorer_free(self._h)
            self._h = C.c_void_p()

    __del__ = close


def lattice_entry_from_raw(num_states: int, start: int, final_cost, arcs, beam: float, key: str = "utt") -> bytes:
    """Host-only (no GPU): determinise a raw lattice and render the CompactLattice table entry.  `arcs` = iterable of
    (src, dst, word, transition_id, graph_cost, acoustic_cost); `final_cost[s]` = +inf for non-final states."""
    arcs = list(arcs)
    n = len(arcs)
    fc = np.ascontiguousarray(final_cost, dtype=np.float32)
    cols = [np.ascontiguousarray([a[k] for a in arcs], dtype=np.int32) for k in range(4)]
    g = np.ascontiguousarray([a[4] for a in arcs], dtype=np.float32)
    ac = np.ascontiguousarray([a[5] for a in arcs], dtype=np.float32)
    pi, pf = C.POINTER(C.c_int32), C.POINTER(C.c_float)
    args = [num_states, start, fc.ctypes.data_as(pf), n, *[c.ctypes.data_as(pi) for c in cols], g.ctypes.data_as(pf), ac.ctypes.data_as(pf),
            float(beam), key.encode()]
    size = lib().rs_lattice_entry_from_raw(*args, None, 0)
    if size < 0:
        _check(int(size))
    buf = C.create_string_buffer(int(size))
    lib().rs_lattice_entry_from_raw(*args, buf, size)
    return buf.raw
