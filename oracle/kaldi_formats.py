"""TEST INFRASTRUCTURE ONLY -- CPU oracle, part 1: readers for the Kaldi/OpenFst files of the hot path.

Independent (Python/numpy) restatement of the on-disk formats the reference reads; the product's loaders
live in rhasspy_speech_amd/csrc/{kaldi_io,model}.cc and never import this.  Citations: kaldi/src
base/io-funcs-inl.h:291-320 (binary header, size-prefixed basic types), matrix/kaldi-matrix.cc (FM/DM),
kaldi-vector.cc (FV/DV), packed-matrix.cc (FP/DP), hmm/hmm-topology.cc:39-161,
hmm/transition-model.cc:144-177,394-420, nnet3/nnet-nnet.cc:586-628, am-nnet-simple.cc:47-58, gmm/diag-gmm.cc:728-756,
ivector/ivector-extractor.cc:828-851, util/parse-options.cc:459-496, openfst lib/fst.cc:58-82, const-fst.h:192-232.
"""
from __future__ import annotations

import re
import struct
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np


class Reader:
    def __init__(self, path):
        self.name = str(path)
        self.b = Path(path).read_bytes()
        self.binary = self.b[:2] == b"\0B"
        self.p = 2 if self.binary else 0

    def _ws(self):
        while self.p < len(self.b) and self.b[self.p:self.p + 1].isspace():
            self.p += 1

    def peek_byte(self) -> int:
        if not self.binary:
            self._ws()
        return self.b[self.p] if self.p < len(self.b) else -1

    def token(self) -> str:
        if not self.binary:
            self._ws()
        s = self.p
        while self.p < len(self.b) and not self.b[self.p:self.p + 1].isspace():
            self.p += 1
        tok = self.b[s:self.p].decode()
        self.p += 1
        return tok

    def peek_token(self) -> str:
        save = self.p
        t = self.token()
        self.p = save
        return t

    def expect(self, tok: str):
        t = self.token()
        assert t == tok, f"{self.name}: expected {tok}, got {t!r}"

    def _text_number(self) -> float:
        self._ws()
        m = re.compile(rb"[-+]?(inf|nan|[0-9.]+([eE][-+]?[0-9]+)?)").match(self.b, self.p)
        assert m, f"{self.name}: number expected at {self.p}"
        self.p = m.end()
        return float(m.group(0))

    def int32(self) -> int:
        if self.binary:
            assert self.b[self.p] == 4, f"{self.name}: int32 size byte"
            v = struct.unpack_from("<i", self.b, self.p + 1)[0]
            self.p += 5
            return v
        return int(self._text_number())

    def real(self) -> float:
        if self.binary:
            sz = self.b[self.p]
            v = struct.unpack_from("<f" if sz == 4 else "<d", self.b, self.p + 1)[0]
            self.p += 1 + sz
            return v
        return self._text_number()

    def raw_basic(self) -> bytes:
        sz = self.b[self.p]
        raw = self.b[self.p + 1:self.p + 1 + sz]
        self.p += 1 + sz
        return raw

    def boolean(self) -> bool:
        if not self.binary:
            self._ws()
        c = self.b[self.p:self.p + 1]
        self.p += 1
        if not self.binary:
            self.p += 1
        assert c in (b"T", b"F")
        return c == b"T"

    def int_vector(self) -> np.ndarray:
        if self.binary:
            assert self.b[self.p] == 4
            n = struct.unpack_from("<i", self.b, self.p + 1)[0]
            v = np.frombuffer(self.b, "<i4", n, self.p + 5).copy()
            self.p += 5 + 4 * n
            return v
        self._ws()
        assert self.b[self.p:self.p + 1] == b"["
        e = self.b.index(b"]", self.p)
        v = np.array([int(x) for x in self.b[self.p + 1:e].split()], np.int32)
        self.p = e + 1
        return v

    def _text_matrix(self) -> np.ndarray:
        self._ws()
        assert self.b[self.p:self.p + 1] == b"[", f"{self.name}: '[' expected at {self.p}"
        e = self.b.index(b"]", self.p)
        body = self.b[self.p + 1:e].decode()
        self.p = e + 1
        rows = [r.split() for r in re.split(r"[\n;]", body)]
        rows = [r for r in rows if r]
        if not rows:
            return np.zeros((0, 0))
        return np.array([[float(x) for x in r] for r in rows], dtype=np.float64)

    def matrix(self) -> np.ndarray:
        """Returns float32 for FM, float64 for DM; text -> float64 (callers cast)."""
        if not self.binary:
            return self._text_matrix()
        tok = self.token()
        assert tok in ("FM", "DM"), f"{self.name}: matrix header {tok!r}"
        r, c = self.int32(), self.int32()
        dt = "<f4" if tok == "FM" else "<f8"
        m = np.frombuffer(self.b, dt, r * c, self.p).reshape(r, c).copy()
        self.p += m.nbytes
        return m

    def vector(self) -> np.ndarray:
        if not self.binary:
            m = self._text_matrix()
            return m.reshape(-1)
        tok = self.token()
        assert tok in ("FV", "DV"), f"{self.name}: vector header {tok!r}"
        n = self.int32()
        dt = "<f4" if tok == "FV" else "<f8"
        v = np.frombuffer(self.b, dt, n, self.p).copy()
        self.p += v.nbytes
        return v

    def sp_matrix(self) -> np.ndarray:
        """Packed symmetric matrix -> full (n x n) float64."""
        if not self.binary:
            self._ws()
            e = self.b.index(b"]", self.p)
            vals = np.array([float(x) for x in self.b[self.p + 1:e].split()])
            self.p = e + 1
            n = int(round((np.sqrt(8 * len(vals) + 1) - 1) / 2))
        else:
            tok = self.token()
            assert tok in ("FP", "DP")
            n = self.int32()
            cnt = n * (n + 1) // 2
            dt = "<f4" if tok == "FP" else "<f8"
            vals = np.frombuffer(self.b, dt, cnt, self.p).astype(np.float64)
            self.p += cnt * (4 if tok == "FP" else 8)
        m = np.zeros((n, n))
        m[np.tril_indices(n)] = vals
        return m + np.tril(m, -1).T

    def line(self) -> str:
        e = self.b.find(b"\n", self.p)
        if e < 0:
            e = len(self.b)
        s = self.b[self.p:e].decode().rstrip("\r")
        self.p = e + 1
        return s


def read_config(path) -> List[Tuple[str, str]]:
    out = []
    for line in Path(path).read_text().splitlines():
        line = line.split("#", 1)[0].strip()
        if not line:
            continue
        assert line.startswith("--"), f"{path}: bad config line {line!r}"
        k, _, v = line[2:].partition("=")
        out.append((k.replace("_", "-").lower(), v.strip() if _ else "true"))
    return out


def read_matrix_file(path) -> np.ndarray:
    return Reader(path).matrix()


# ---------------------------------------------------------------------------------------- transition model


def read_transition_model(r: Reader) -> np.ndarray:
    """Returns id2pdf (index = transition-id, entry 0 unused)."""
    r.expect("<TransitionModel>")
    r.expect("<Topology>")
    entries: List[List[Tuple[int, int, List[int]]]] = []   # per entry: per state (fwd, self, [dst...])
    phone2idx: Dict[int, int] = {}
    if r.binary:
        phones = r.int_vector()
        p2i = r.int_vector()
        phone2idx = {i: int(v) for i, v in enumerate(p2i) if v >= 0}
        sz = r.int32()
        is_hmm = True
        if sz == -1:
            is_hmm = False
            sz = r.int32()
        for _ in range(sz):
            states = []
            for _ in range(r.int32()):
                fwd = r.int32()
                slf = fwd if is_hmm else r.int32()
                dsts = []
                for _ in range(r.int32()):
                    dsts.append(r.int32())
                    r.real()
                states.append((fwd, slf, dsts))
            entries.append(states)
        r.expect("</Topology>")
    else:
        while True:
            t = r.token()
            if t == "</Topology>":
                break
            assert t == "<TopologyEntry>"
            r.expect("<ForPhones>")
            these = []
            while True:
                s = r.token()
                if s == "</ForPhones>":
                    break
                these.append(int(s))
            states = []
            t = r.token()
            while t != "</TopologyEntry>":
                assert t == "<State>"
                r.int32()
                fwd = slf = -1
                t = r.token()
                if t == "<PdfClass>":
                    fwd = slf = r.int32()
                    t = r.token()
                elif t == "<ForwardPdfClass>":
                    fwd = r.int32()
                    r.expect("<SelfLoopPdfClass>")
                    slf = r.int32()
                    t = r.token()
                dsts = []
                while t == "<Transition>":
                    dsts.append(r.int32())
                    r.real()
                    t = r.token()
                assert t == "</State>"
                states.append((fwd, slf, dsts))
                t = r.token()
            for p in these:
                phone2idx[p] = len(entries)
            entries.append(states)
    t = r.token()
    assert t in ("<Triples>", "<Tuples>")
    n = r.int32()
    id2pdf = [0]
    for _ in range(n):
        phone, st, fwd = r.int32(), r.int32(), r.int32()
        slf = r.int32() if t == "<Tuples>" else fwd
        for dst in entries[phone2idx[phone]][st][2]:
            id2pdf.append(slf if dst == st else fwd)
    r.token()
    r.expect("<LogProbs>")
    r.vector()
    r.expect("</LogProbs>")
    r.expect("</TransitionModel>")
    return np.array(id2pdf, np.int32)


# ---------------------------------------------------------------------------------------- nnet3

_INT_VEC_FIELDS = {"<TimeOffsets>", "<ColumnMap>", "<Sizes>"}
_INT_FIELDS = {"<Dim>", "<BlockDim>", "<InputDim>", "<OutputDim>", "<RankIn>", "<RankOut>", "<UpdatePeriod>", "<RankInOut>",
               "<TimePeriod>"}


@dataclass
class Comp:
    type: str
    fields: Dict[str, object] = field(default_factory=dict)


def _read_component(r: Reader) -> Comp:
    opening = r.token()
    typ = opening[1:-1]
    c = Comp(typ)
    closing = f"</{typ}>"
    while True:
        tok = r.token()
        if tok == closing:
            return c
        if tok in _INT_VEC_FIELDS:
            c.fields[tok] = r.int_vector()
            continue
        vals: List[object] = []
        while True:
            ch = r.peek_byte()
            if ch == ord("<"):
                break
            if r.binary:
                if ch in (4, 8):
                    raw = r.raw_basic()
                    if tok in _INT_FIELDS:
                        vals.append(struct.unpack("<i" if len(raw) == 4 else "<q", raw)[0])
                    else:
                        vals.append(struct.unpack("<f" if len(raw) == 4 else "<d", raw)[0])
                elif r.b[r.p:r.p + 3] in (b"FM ", b"DM "):
                    vals.append(r.matrix())
                elif r.b[r.p:r.p + 3] in (b"FV ", b"DV "):
                    vals.append(r.vector())
                elif ch in (ord("T"), ord("F")):
                    vals.append(r.boolean())
                else:
                    raise ValueError(f"{r.name}: cannot parse {tok} of {typ}")
            else:
                if ch == ord("["):
                    m = r._text_matrix()
                    vals.append(m if (m.shape[0] > 1 or tok in ("<LinearParams>", "<Params>")) else m.reshape(-1))
                elif r.peek_token() in ("T", "F"):
                    vals.append(r.boolean())
                else:
                    v = r._text_number()
                    vals.append(int(v) if tok in _INT_FIELDS else v)
        c.fields[tok] = vals[0] if len(vals) == 1 else (vals if vals else True)


@dataclass
class NnetFile:
    config: List[str]
    components: Dict[str, Comp]
    priors: np.ndarray


def read_final_mdl(path) -> Tuple[np.ndarray, NnetFile]:
    r = Reader(path)
    id2pdf = read_transition_model(r)
    r.expect("<Nnet3>")
    assert r.line().strip() == ""
    cfg = []
    while True:
        ln = r.line()
        if ln == "":
            break
        cfg.append(ln)
    r.expect("<NumComponents>")
    n = r.int32()
    comps = {}
    for _ in range(n):
        r.expect("<ComponentName>")
        name = r.token()
        comps[name] = _read_component(r)
    r.expect("</Nnet3>")
    r.expect("<LeftContext>")
    r.int32()
    r.expect("<RightContext>")
    r.int32()
    r.expect("<Priors>")
    pri = r.vector().astype(np.float32)
    return id2pdf, NnetFile(cfg, comps, pri)


# ---------------------------------------------------------------------------------------- extractor files


@dataclass
class DiagGmm:
    weights: np.ndarray
    means_invvars: np.ndarray
    inv_vars: np.ndarray
    gconsts: np.ndarray


def read_diag_gmm(path) -> DiagGmm:
    r = Reader(path)
    assert r.token() in ("<DiagGMM>", "<DiagGMMBegin>")
    t = r.token()
    if t == "<GCONSTS>":
        r.vector()
        r.expect("<WEIGHTS>")
    w = r.vector().astype(np.float32)
    r.expect("<MEANS_INVVARS>")
    miv = r.matrix().astype(np.float32)
    r.expect("<INV_VARS>")
    iv = r.matrix().astype(np.float32)
    D = miv.shape[1]
    # ComputeGconsts (gmm/diag-gmm.cc:114-150): float accumulator, double terms
    offset = np.float32(-0.5 * 1.8378770664093454835606594728112 * D)
    gc = np.zeros(len(w), np.float32)
    for g in range(len(w)):
        acc = np.float32(np.log(w[g])) + offset
        for d in range(D):
            acc = np.float32(np.float64(acc) + (0.5 * np.float64(np.log(iv[g, d])) - 0.5 * np.float64(miv[g, d]) * np.float64(miv[g, d]) / np.float64(iv[g, d])))
        gc[g] = acc
    return DiagGmm(w, miv, iv, gc)


@dataclass
class IvectorExtractorFile:
    M: np.ndarray            # G x D x I (float64)
    sigma_inv: np.ndarray    # G x D x D
    prior_offset: float


def read_ivector_extractor(path) -> IvectorExtractorFile:
    r = Reader(path)
    r.expect("<IvectorExtractor>")
    r.expect("<w>")
    w = r.matrix()
    assert w.shape[0] == 0, "iVector-dependent weights unsupported"
    r.expect("<w_vec>")
    r.vector()
    r.expect("<M>")
    n = r.int32()
    M = np.stack([r.matrix().astype(np.float64) for _ in range(n)])
    r.expect("<SigmaInv>")
    S = np.stack([r.sp_matrix() for _ in range(n)])
    r.expect("<IvectorOffset>")
    po = r.real()
    r.expect("</IvectorExtractor>")
    return IvectorExtractorFile(M, S, po)


# ---------------------------------------------------------------------------------------- FST


@dataclass
class Fst:
    start: int
    final: np.ndarray        # per state, inf = non-final
    arc_begin: np.ndarray    # n+1
    num_ieps: np.ndarray
    ilabel: np.ndarray
    olabel: np.ndarray
    weight: np.ndarray
    nextstate: np.ndarray


def read_fst(path) -> Fst:
    b = Path(path).read_bytes()
    p = 0

    def get(fmt):
        nonlocal p
        v = struct.unpack_from(fmt, b, p)
        p += struct.calcsize(fmt)
        return v[0] if len(v) == 1 else v

    def string():
        nonlocal p
        n = get("<i")
        s = b[p:p + n].decode()
        p += n
        return s

    assert get("<i") == 2125659606, "bad FST magic"
    fsttype, arctype = string(), string()
    version, flags = get("<i"), get("<i")
    get("<Q")
    start, ns, na = get("<q"), get("<q"), get("<q")
    assert arctype == "standard"
    assert not (flags & 3), "embedded symbol tables unsupported in the oracle"
    if fsttype == "const":
        if version == 1 or flags & 4:
            p = (p + 15) // 16 * 16
        st = np.frombuffer(b, np.dtype([("w", "<f4"), ("pos", "<u4"), ("n", "<u4"), ("ni", "<u4"), ("no", "<u4")]), ns, p)
        p += st.nbytes
        if version == 1 or flags & 4:
            p = (p + 15) // 16 * 16
        arcs = np.frombuffer(b, np.dtype([("il", "<i4"), ("ol", "<i4"), ("w", "<f4"), ("ns", "<i4")]), na, p)
        arc_begin = np.concatenate([st["pos"], [na]]).astype(np.int64)
        return Fst(start, st["w"].copy(), arc_begin, st["ni"].astype(np.int64), arcs["il"].copy(), arcs["ol"].copy(),
                   arcs["w"].copy(), arcs["ns"].copy())
    assert fsttype == "vector"
    final, begin, il, ol, w, nx, nie = [], [0], [], [], [], [], []
    for _ in range(ns):
        f, n = get("<f"), get("<q")
        final.append(f)
        arcs = [struct.unpack_from("<iifi", b, p + 16 * i) for i in range(n)]
        p += 16 * n
        arcs.sort(key=lambda a: a[0])     # the decoder only needs epsilons first; stable
        nie.append(sum(1 for a in arcs if a[0] == 0))
        for a in arcs:
            il.append(a[0]); ol.append(a[1]); w.append(a[2]); nx.append(a[3])
        begin.append(len(il))
    return Fst(start, np.array(final, np.float32), np.array(begin, np.int64), np.array(nie, np.int64), np.array(il, np.int32),
               np.array(ol, np.int32), np.array(w, np.float32), np.array(nx, np.int32))
