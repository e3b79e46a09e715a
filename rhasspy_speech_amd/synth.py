"""Synthetic "zamia-like" model / graph / audio generator in genuine Kaldi on-disk formats.

No acoustic model, HCLG or config ships with the reference, so the
benchmark and the parity tests run on synthetic models written in exactly the layout
the reference's hot path reads:

    <model_dir>/model/model/final.mdl                         TransitionModel + nnet3 (binary or text)
    <model_dir>/model/online/conf/{online,mfcc,ivector_extractor,splice,online_cmvn}.conf
    <model_dir>/model/online/ivector_extractor/{final.mat,final.ie,final.dubm,global_cmvn.stats}
    <graph_dir>/{HCLG.fst,words.txt}

Formats follow kaldi/src/base/io-funcs-inl.h (binary basic types), matrix/kaldi-matrix.cc
(FM/FV/DM/DV/DP), hmm/transition-model.cc:422-453, hmm/hmm-topology.cc:163-230,
nnet3/nnet-nnet.cc:631-656, nnet3/am-nnet-simple.cc:33-45, gmm/diag-gmm.cc (Write),
ivector/ivector-extractor.cc:805-825, openfst const-fst.h (ConstFst v2).  The files are
validated by the reference's own binaries in tests (oracle/_ref) in the build container.

Everything here is plain numpy; it runs on the GPU box (no Kaldi needed).
"""
from __future__ import annotations

import io
import math
import struct
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------- binary/text writers


class KaldiWriter:
    """Minimal Kaldi object writer (binary or text)."""

    def __init__(self, binary: bool = True):
        self.binary = binary
        self.buf = io.BytesIO()
        if binary:
            self.buf.write(b"\0B")

    def token(self, tok: str) -> "KaldiWriter":
        self.buf.write(tok.encode() + b" ")
        return self

    def nl(self) -> "KaldiWriter":
        if not self.binary:
            self.buf.write(b"\n")
        return self

    def raw(self, b: bytes) -> "KaldiWriter":
        self.buf.write(b)
        return self

    def i32(self, v: int) -> "KaldiWriter":
        if self.binary:
            self.buf.write(b"\x04" + struct.pack("<i", int(v)))
        else:
            self.buf.write(f"{int(v)} ".encode())
        return self

    def f32(self, v: float) -> "KaldiWriter":
        if self.binary:
            self.buf.write(b"\x04" + struct.pack("<f", float(v)))
        else:
            self.buf.write(f"{float(np.float32(v))!r} ".encode())
        return self

    def f64(self, v: float) -> "KaldiWriter":
        if self.binary:
            self.buf.write(b"\x08" + struct.pack("<d", float(v)))
        else:
            self.buf.write(f"{float(v)!r} ".encode())
        return self

    def boolean(self, v: bool) -> "KaldiWriter":
        self.buf.write(b"T" if v else b"F")
        if not self.binary:
            self.buf.write(b" ")
        return self

    def int_vector(self, v: Sequence[int]) -> "KaldiWriter":
        if self.binary:
            self.buf.write(b"\x04" + struct.pack("<i", len(v)))
            self.buf.write(np.asarray(v, dtype="<i4").tobytes())
        else:
            self.buf.write(("[ " + " ".join(str(int(x)) for x in v) + " ]\n").encode())
        return self

    def vector(self, v: np.ndarray, double: bool = False) -> "KaldiWriter":
        v = np.asarray(v)
        if self.binary:
            self.buf.write((b"DV " if double else b"FV ") + b"\x04" + struct.pack("<i", v.shape[0]))
            self.buf.write(v.astype("<f8" if double else "<f4").tobytes())
        else:
            vals = v.astype(np.float64 if double else np.float32)
            self.buf.write((" [ " + " ".join(repr(float(x)) for x in vals) + " ]\n").encode())
        return self

    def matrix(self, m: np.ndarray, double: bool = False) -> "KaldiWriter":
        m = np.asarray(m)
        if self.binary:
            self.buf.write((b"DM " if double else b"FM ") + b"\x04" + struct.pack("<i", m.shape[0])
                           + b"\x04" + struct.pack("<i", m.shape[1]))
            self.buf.write(np.ascontiguousarray(m.astype("<f8" if double else "<f4")).tobytes())
        else:
            vals = m.astype(np.float64 if double else np.float32)
            if vals.shape[0] == 0:
                self.buf.write(b" [ ]\n")
            else:
                rows = ["  " + " ".join(repr(float(x)) for x in r) for r in vals]
                self.buf.write((" [\n" + "\n".join(rows) + " ]\n").encode())
        return self

    def sp_matrix(self, m: np.ndarray, double: bool = True) -> "KaldiWriter":
        """Packed symmetric matrix (lower triangle, row-major): DP/FP."""
        n = m.shape[0]
        packed = m[np.tril_indices(n)]
        if self.binary:
            self.buf.write((b"DP " if double else b"FP ") + b"\x04" + struct.pack("<i", n))
            self.buf.write(packed.astype("<f8" if double else "<f4").tobytes())
        else:
            out = [" ["]
            k = 0
            for r in range(n):
                out.append("  " + " ".join(repr(float(x)) for x in packed[k:k + r + 1]))
                k += r + 1
            self.buf.write(("\n".join(out) + " ]\n").encode())
        return self

    def getvalue(self) -> bytes:
        return self.buf.getvalue()


def write_kaldi_matrix_file(path: Path, m: np.ndarray, binary: bool = True, double: bool = False) -> None:
    w = KaldiWriter(binary)
    w.matrix(m, double)
    Path(path).write_bytes(w.getvalue())


# --------------------------------------------------------------------------- model spec


@dataclass
class ModelSpec:
    """Dimensions of a synthetic acoustic model.  Defaults = "zamia-like-S" (SURVEY.md §8(d))."""

    name: str = "zamia-like-S"
    seed: int = 1
    num_ceps: int = 40
    num_mel_bins: int = 40
    ivector_dim: int = 100          # 0 => no iVector extractor at all
    num_gauss: int = 512
    lda_dim: int = 40
    splice_left: int = 3
    splice_right: int = 3
    num_phones: int = 1000          # chain topology: 2 pdfs per phone => P = 2*num_phones
    hidden_dim: int = 250
    # per hidden layer: time offsets of its input Append
    layer_offsets: Tuple[Tuple[int, ...], ...] = ((0,), (-1, 0, 1), (-1, 0, 1), (-3, 0, 3), (-3, 0, 3), (-3, 0, 3), (-3, 0, 3))
    lda_offsets: Tuple[int, ...] = (-1, 0, 1)
    prefinal_dim: int = 250
    with_priors: bool = False
    with_log_softmax: bool = False
    tdnnf: bool = False              # use TdnnComponent + LinearComponent bottlenecks (coverage net)
    bottleneck_dim: int = 64
    chain_topology: bool = True      # forward/self-loop pdf classes (nnet3 chain models)
    dither: Optional[float] = None   # None: no --dither line, i.e. the reference's default 1.0 like conf/mfcc_hires.conf (feature-window.h:57)
    frame_length: Optional[float] = None   # ms; None: no --frame-length line (25 ms: a 512-point FFT); 100 ms pads to 2048 points
    nnet_cmvn: bool = False          # --cmvn-config on the nnet input branch
    binary: bool = True
    xent_branch: bool = True         # extra output-xent branch (ignored by decoding), as chain recipes have
    input_scale: float = 0.04        # lda columns that see raw MFCCs (magnitude ~10-80) are scaled down, like a trained LDA whitens
    output_scale: float = 0.25       # keeps per-frame log-likelihood spread at a realistic few units
    # phonetic context of the decision tree written beside final.mdl (graph construction):
    # "mono" (N=1, P=0), "biphone" (N=2, P=1: every third phone has two pdf sets chosen by its left neighbour) or
    # "triphone" (N=3, P=1: additionally every third phone is split on its right neighbour)
    context: str = "mono"
    # layer weights: "normal" = N(0, 1 / fan_in); "heavy" = Student t with 2 degrees of freedom / sqrt(fan_in), clipped to
    # +-1000: rows with a few weights hundreds of times the typical one (the split-precision GEMMs scale every output column by
    # its largest weight)
    weight_dist: str = "normal"
    hidden_gain: float = 1.0         # the second hidden layer's weights times this (activations beyond fp16's range for ~1e5)
    # hidden_gain with the rest of the network making up for it: the second hidden layer's bias is scaled too, it has no BatchNorm
    # (its output is relu(gain * (W x + b)): every element of every row is of order gain), and the third layer's weights are
    # divided by gain -- in exact arithmetic the network of gain 1 without that BatchNorm; in a GEMM that carries small operands
    # to a fixed ABSOLUTE error only, a different one (tests: small activations and the split-fp16 layer GEMMs)
    gain_compensated: bool = False
    hmm_states: int = 1              # emitting states per phone (left-to-right; > 1 only with chain_topology = False, graphs by mkgraph)

    @property
    def num_pdfs(self) -> int:
        return len(context_tuples(self)) * (2 if self.chain_topology else 1)


TINY = dict(name="tiny", ivector_dim=10, num_gauss=16, lda_dim=12, num_phones=24, hidden_dim=32,
            layer_offsets=((0,), (-1, 0, 1), (-2, 0, 2)), prefinal_dim=24)


def tiny_spec(**kw) -> ModelSpec:
    d = dict(TINY)
    d.update(kw)
    return ModelSpec(**d)


# --------------------------------------------------------------------------- phonetic context

def context_shape(spec: "ModelSpec") -> Tuple[int, int]:
    """(context width N, central position P).  Besides the three named shapes any "N,P" is accepted (graph-construction tests)."""
    named = {"mono": (1, 0), "biphone": (2, 1), "triphone": (3, 1)}
    if spec.context in named:
        return named[spec.context]
    n, p = (int(x) for x in spec.context.split(","))
    if not (0 <= p < n):
        raise ValueError(f"bad context {spec.context}")
    return n, p


def context_splits(spec: "ModelSpec", phone: int):
    """None, or (tree key, sorted yes-set) of the one question asked about `phone`'s context window: every third phone is asked
    about its left neighbour (if the window has one), every third about its right neighbour (if it has one)."""
    n = spec.num_phones
    width, central = context_shape(spec)
    if central > 0 and phone % 3 == 0:
        return central - 1, [0] + [q for q in range(1, n + 1) if q % 2 == 1]           # left neighbour (0 = start of utterance)
    if central + 1 < width and phone % 3 == 1:
        return central + 1, [0] + [q for q in range(1, n + 1) if q % 2 == 0]           # right neighbour (0 = end of utterance)
    return None


def variant_pdfs(spec: "ModelSpec") -> Dict[int, List[List[int]]]:
    """phone -> context variants (yes branch first) -> pdf per pdf-class.  Pdf-classes: (forward, self-loop) of the one state of
    the chain topology, or one per emitting state of the HMM topology."""
    if spec.chain_topology and spec.hmm_states != 1:
        raise ValueError("hmm_states > 1 needs chain_topology = False")
    classes = 2 if spec.chain_topology else spec.hmm_states
    out, pdf = {}, 0
    for p in range(1, spec.num_phones + 1):
        out[p] = []
        for _ in range(2 if context_splits(spec, p) else 1):
            out[p].append(list(range(pdf, pdf + classes)))
            pdf += classes
    return out


def context_tuples(spec: "ModelSpec") -> List[Tuple[int, int, int, int]]:
    """The transition model's (phone, hmm-state, forward pdf, self-loop pdf) table, sorted as transition-model.cc:62-100 sorts
    it; a phone with a context question contributes two entries per state."""
    out = []
    for p, variants in variant_pdfs(spec).items():
        for pdfs in variants:
            if spec.chain_topology:
                out.append((p, 0, pdfs[0], pdfs[1]))
            else:
                out += [(p, hs, pdfs[hs], pdfs[hs]) for hs in range(spec.hmm_states)]
    return sorted(out)


def write_tree(path: Path, spec: "ModelSpec") -> None:
    """<model>/tree: a ContextDependency in text form (tree/context-dep.cc:143-156, tree/event-map.cc:55-205): a table on the
    central phone, under it a split on one neighbour where context_splits() asks one, under that the pdf per pdf-class."""
    n_ctx, p_ctx = context_shape(spec)
    vp = variant_pdfs(spec)

    def leaf(pdfs: List[int]) -> str:
        if len(pdfs) == 1:
            return f"CE {pdfs[0]} "
        return f"TE -1 {len(pdfs)} ( " + "".join(f"CE {x} " for x in pdfs) + ") "

    body = f"TE {p_ctx} {spec.num_phones + 1} ( NULL "
    for p in range(1, spec.num_phones + 1):
        q = context_splits(spec, p)
        if q is None:
            body += leaf(vp[p][0])
        else:
            key, yes = q
            body += f"SE {key} [ " + " ".join(map(str, yes)) + " ]\n{ " + leaf(vp[p][0]) + leaf(vp[p][1]) + "} "
    body += ") "
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    Path(path).write_text(f"ContextDependency {n_ctx} {p_ctx} ToPdf {body}\nEndContextDependency ")


# --------------------------------------------------------------------------- final.mdl


def _write_topology(w: KaldiWriter, spec: ModelSpec) -> None:
    phones = list(range(1, spec.num_phones + 1))
    w.token("<Topology>")
    if w.binary:
        w.int_vector(phones)
        phone2idx = [-1] + [0] * spec.num_phones
        w.int_vector(phone2idx)
        if spec.chain_topology:
            w.i32(-1)                    # marker: not a plain HMM (separate self-loop pdf class)
        w.i32(1)                         # one topology entry
        ns = 1 if spec.chain_topology else spec.hmm_states
        w.i32(ns + 1)                    # emitting states + the final one
        for st in range(ns):
            w.i32(st)                    # forward pdf class
            if spec.chain_topology:
                w.i32(1)                 # self-loop pdf class
            w.i32(2)
            w.i32(st).f32(0.5)
            w.i32(st + 1).f32(0.5)
        # final, non-emitting state
        w.i32(-1)
        if spec.chain_topology:
            w.i32(-1)
        w.i32(0)
        w.token("</Topology>")
    else:
        w.raw(b"\n<TopologyEntry>\n<ForPhones>\n" + " ".join(map(str, phones)).encode() + b"\n</ForPhones>\n")
        if spec.chain_topology:
            w.raw(b"<State> 0 <ForwardPdfClass> 0 <SelfLoopPdfClass> 1 <Transition> 0 0.5 <Transition> 1 0.5 </State>\n")
            w.raw(b"<State> 1 </State>\n</TopologyEntry>\n</Topology>\n")
        else:
            for st in range(spec.hmm_states):
                w.raw(f"<State> {st} <PdfClass> {st} <Transition> {st} 0.5 <Transition> {st + 1} 0.5 </State>\n".encode())
            w.raw(f"<State> {spec.hmm_states} </State>\n</TopologyEntry>\n</Topology>\n".encode())


def _write_transition_model(w: KaldiWriter, spec: ModelSpec) -> None:
    w.token("<TransitionModel>").nl()
    _write_topology(w, spec)
    tuples = context_tuples(spec)
    n = len(tuples)
    if spec.chain_topology:
        w.token("<Tuples>").i32(n).nl()
        for p, hs, fwd, slf in tuples:
            w.i32(p).i32(hs).i32(fwd).i32(slf).nl()
        w.token("</Tuples>").nl()
    else:
        w.token("<Triples>").i32(n).nl()
        for p, hs, fwd, _slf in tuples:
            w.i32(p).i32(hs).i32(fwd).nl()
        w.token("</Triples>").nl()
    w.token("<LogProbs>").nl()
    # index 0 unused; one entry per transition-id
    if spec.context == "mono" and spec.hmm_states == 1:
        probs = np.full(2 * n, 0.5)
    else:        # (context-dependent test models: a different self-loop probability per transition-state)
        sl = 0.35 + 0.05 * (np.arange(n) % 7)
        probs = np.stack([sl, 1.0 - sl], axis=1).reshape(-1)
    w.vector(np.concatenate([[0.0], np.log(probs)]).astype(np.float32))
    w.token("</LogProbs>").nl()
    w.token("</TransitionModel>").nl()


def transition_ids(spec: ModelSpec, phone: int) -> Tuple[int, int]:
    """(self-loop tid, forward tid) of 1-based `phone` for the synthetic topology
    (hmm/transition-model.cc:144-177: ids are assigned in tuple order, topology transition order).  Context-independent
    models only: the directly assembled synthetic graphs know nothing of phonetic context."""
    if context_shape(spec) != (1, 0) or spec.hmm_states != 1:
        raise ValueError("transition_ids: context-dependent or multi-state model; build its graph with mkgraph")
    return 2 * (phone - 1) + 1, 2 * (phone - 1) + 2


def tid_to_pdf(spec: ModelSpec) -> np.ndarray:
    tuples = context_tuples(spec)
    out = np.zeros(2 * len(tuples) + 1, dtype=np.int32)
    for k, (_p, _hs, fwd, slf) in enumerate(tuples):
        out[2 * k + 1] = slf         # transition 0 of the topology's state is the self loop
        out[2 * k + 2] = fwd
    return out


def _updatable_open(w: KaldiWriter, typ: str) -> None:
    w.token(f"<{typ}>").token("<LearningRate>").f32(0.001)


def _w_affine(w: KaldiWriter, typ: str, W: np.ndarray, b: np.ndarray) -> None:
    _updatable_open(w, typ)
    w.token("<LinearParams>").matrix(W).token("<BiasParams>").vector(b)
    if typ == "NaturalGradientAffineComponent":
        w.token("<RankIn>").i32(20).token("<RankOut>").i32(80)
        w.token("<UpdatePeriod>").i32(4).token("<NumSamplesHistory>").f32(2000.0).token("<Alpha>").f32(4.0)
    w.token(f"</{typ}>")


def _w_fixed_affine(w: KaldiWriter, W: np.ndarray, b: np.ndarray) -> None:
    w.token("<FixedAffineComponent>").token("<LinearParams>").matrix(W).token("<BiasParams>").vector(b)
    w.token("</FixedAffineComponent>")


def _w_nonlinear(w: KaldiWriter, typ: str, dim: int) -> None:
    w.token(f"<{typ}>").token("<Dim>").i32(dim)
    w.token("<ValueAvg>").vector(np.zeros(0, np.float32)).token("<DerivAvg>").vector(np.zeros(0, np.float32))
    w.token("<Count>").f64(0.0)
    w.token("<OderivRms>").vector(np.zeros(0, np.float32)).token("<OderivCount>").f64(0.0)
    w.token("<NumDimsSelfRepaired>").f64(0.0).token("<NumDimsProcessed>").f64(0.0)
    w.token(f"</{typ}>")


def _w_batchnorm(w: KaldiWriter, mean: np.ndarray, var: np.ndarray, eps: float = 1e-3, target_rms: float = 1.0) -> None:
    dim = mean.shape[0]
    w.token("<BatchNormComponent>").token("<Dim>").i32(dim).token("<BlockDim>").i32(dim)
    w.token("<Epsilon>").f32(eps).token("<TargetRms>").f32(target_rms).token("<TestMode>").boolean(False)
    w.token("<Count>").f32(1000.0).token("<StatsMean>").vector(mean).token("<StatsVar>").vector(var)
    w.token("</BatchNormComponent>")


def _w_tdnn(w: KaldiWriter, offsets: Sequence[int], W: np.ndarray, b: Optional[np.ndarray]) -> None:
    _updatable_open(w, "TdnnComponent")
    w.token("<TimeOffsets>").int_vector(list(offsets))
    w.token("<LinearParams>").matrix(W)
    w.token("<BiasParams>").vector(b if b is not None else np.zeros(0, np.float32))
    w.token("<OrthonormalConstraint>").f32(0.0).token("<UseNaturalGradient>").boolean(True)
    w.token("<NumSamplesHistory>").f32(2000.0).token("<AlphaInOut>").f32(4.0).f32(4.0)
    w.token("<RankInOut>").i32(20).i32(80)
    w.token("</TdnnComponent>")


def _w_linear(w: KaldiWriter, W: np.ndarray) -> None:
    _updatable_open(w, "LinearComponent")
    w.token("<Params>").matrix(W)
    w.token("<UseNaturalGradient>").boolean(True).token("<RankInOut>").i32(20).i32(80)
    w.token("<Alpha>").f32(4.0).token("<NumSamplesHistory>").f32(2000.0).token("<UpdatePeriod>").i32(4)
    w.token("</LinearComponent>")


def _w_general_dropout(w: KaldiWriter, dim: int) -> None:
    # nnet-general-component.cc:1674-1696 (test-mode/continuous are bare flag tokens)
    w.token("<GeneralDropoutComponent>").token("<Dim>").i32(dim).token("<BlockDim>").i32(dim)
    w.token("<TimePeriod>").i32(0).token("<DropoutProportion>").f32(0.2)
    w.token("<Continuous>")
    w.token("</GeneralDropoutComponent>")


def _w_noop(w: KaldiWriter, dim: int) -> None:
    # nnet-simple-component.cc:480-487
    w.token("<NoOpComponent>").token("<Dim>").i32(dim).token("<BackpropScale>").f32(1.0).token("</NoOpComponent>")


def _append_desc(src: str, offsets: Sequence[int]) -> str:
    parts = [src if o == 0 else f"Offset({src}, {o})" for o in offsets]
    return parts[0] if len(parts) == 1 else "Append(" + ", ".join(parts) + ")"


def build_nnet(spec: ModelSpec, rng: np.random.Generator):
    """Returns (config_lines, [(name, writer_fn)], params dict for the numpy oracle)."""
    cfg: List[str] = []
    comps: List[Tuple[str, object]] = []
    C, D = spec.num_ceps, spec.ivector_dim
    cfg.append(f"input-node name=input dim={C}")
    if D > 0:
        cfg.append(f"input-node name=ivector dim={D}")

    def randw(o, i):
        if spec.weight_dist == "heavy":
            return np.clip(rng.standard_t(2.0, (o, i)) / math.sqrt(i), -1000.0, 1000.0).astype(np.float32)
        return (rng.standard_normal((o, i)) / math.sqrt(i)).astype(np.float32)

    def randb(o, s=0.1):
        return (rng.standard_normal(o) * s).astype(np.float32)

    # lda
    lda_in = C * len(spec.lda_offsets) + D
    parts = [("input" if o == 0 else f"Offset(input, {o})") for o in spec.lda_offsets]
    if D > 0:
        parts.append("ReplaceIndex(ivector, t, 0)")
    lda_desc = "Append(" + ", ".join(parts) + ")" if len(parts) > 1 else parts[0]
    Wl, bl = randw(lda_in, lda_in), randb(lda_in)
    Wl[:, :C * len(spec.lda_offsets)] *= np.float32(spec.input_scale)
    comps.append(("lda", lambda w, W=Wl, b=bl: _w_fixed_affine(w, W, b)))
    cfg.append(f"component-node name=lda component=lda input={lda_desc}")
    prev, prev_dim = "lda", lda_in

    for li, offs in enumerate(spec.layer_offsets, start=1):
        H = spec.hidden_dim
        if spec.tdnnf and li > 1:
            # factorized layer: linear bottleneck (TdnnComponent w/o bias) -> affine (TdnnComponent) -> relu -> bn
            # -> dropout(test: identity) -> Sum(Scale(0.66, prev), this) residual through NoOp
            lo = [o for o in offs if o <= 0]
            ro = [o for o in offs if o >= 0]
            B = spec.bottleneck_dim
            W1 = randw(B, prev_dim * len(lo))
            comps.append((f"tdnnf{li}.linear", lambda w, o=lo, W=W1: _w_tdnn(w, o, W, None)))
            cfg.append(f"component-node name=tdnnf{li}.linear component=tdnnf{li}.linear input={prev}")
            W2, b2 = randw(H, B * len(ro)), randb(H)
            comps.append((f"tdnnf{li}.affine", lambda w, o=ro, W=W2, b=b2: _w_tdnn(w, o, W, b)))
            cfg.append(f"component-node name=tdnnf{li}.affine component=tdnnf{li}.affine input=tdnnf{li}.linear")
            comps.append((f"tdnnf{li}.relu", lambda w, d=H: _w_nonlinear(w, "RectifiedLinearComponent", d)))
            cfg.append(f"component-node name=tdnnf{li}.relu component=tdnnf{li}.relu input=tdnnf{li}.affine")
            mean, var = (0.4 + 0.1 * rng.standard_normal(H)).astype(np.float32), rng.uniform(0.1, 0.5, H).astype(np.float32)
            comps.append((f"tdnnf{li}.batchnorm", lambda w, m=mean, v=var: _w_batchnorm(w, m, v)))
            cfg.append(f"component-node name=tdnnf{li}.batchnorm component=tdnnf{li}.batchnorm input=tdnnf{li}.relu")
            comps.append((f"tdnnf{li}.dropout", lambda w, d=H: _w_general_dropout(w, d)))
            cfg.append(f"component-node name=tdnnf{li}.dropout component=tdnnf{li}.dropout input=tdnnf{li}.batchnorm")
            comps.append((f"tdnnf{li}.noop", lambda w, d=H: _w_noop(w, d)))
            if prev_dim == H:
                cfg.append(f"component-node name=tdnnf{li}.noop component=tdnnf{li}.noop "
                           f"input=Sum(Scale(0.66, {prev}), tdnnf{li}.dropout)")
            else:
                cfg.append(f"component-node name=tdnnf{li}.noop component=tdnnf{li}.noop input=tdnnf{li}.dropout")
            prev, prev_dim = f"tdnnf{li}.noop", H
            continue
        W, b = randw(H, prev_dim * len(offs)), randb(H)
        if li == 2 and spec.hidden_gain != 1.0:
            W = (W * np.float32(spec.hidden_gain)).astype(np.float32)
            if spec.gain_compensated:
                b = (b * np.float32(spec.hidden_gain)).astype(np.float32)
        if li == 3 and spec.hidden_gain != 1.0 and spec.gain_compensated:
            W = (W / np.float32(spec.hidden_gain)).astype(np.float32)
        comps.append((f"tdnn{li}.affine", lambda w, W=W, b=b: _w_affine(w, "NaturalGradientAffineComponent", W, b)))
        cfg.append(f"component-node name=tdnn{li}.affine component=tdnn{li}.affine input={_append_desc(prev, offs)}")
        comps.append((f"tdnn{li}.relu", lambda w, d=H: _w_nonlinear(w, "RectifiedLinearComponent", d)))
        cfg.append(f"component-node name=tdnn{li}.relu component=tdnn{li}.relu input=tdnn{li}.affine")
        if li == 2 and spec.gain_compensated:
            prev, prev_dim = f"tdnn{li}.relu", H
            continue
        mean, var = (0.4 + 0.1 * rng.standard_normal(H)).astype(np.float32), rng.uniform(0.1, 0.5, H).astype(np.float32)
        comps.append((f"tdnn{li}.batchnorm", lambda w, m=mean, v=var: _w_batchnorm(w, m, v)))
        cfg.append(f"component-node name=tdnn{li}.batchnorm component=tdnn{li}.batchnorm input=tdnn{li}.relu")
        prev, prev_dim = f"tdnn{li}.batchnorm", H

    # prefinal-chain: affine -> relu -> batchnorm
    Pd = spec.prefinal_dim
    W, b = randw(Pd, prev_dim), randb(Pd)
    comps.append(("prefinal-chain.affine", lambda w, W=W, b=b: _w_affine(w, "NaturalGradientAffineComponent", W, b)))
    cfg.append(f"component-node name=prefinal-chain.affine component=prefinal-chain.affine input={prev}")
    comps.append(("prefinal-chain.relu", lambda w, d=Pd: _w_nonlinear(w, "RectifiedLinearComponent", d)))
    cfg.append("component-node name=prefinal-chain.relu component=prefinal-chain.relu input=prefinal-chain.affine")
    mean, var = (0.4 + 0.1 * rng.standard_normal(Pd)).astype(np.float32), rng.uniform(0.1, 0.5, Pd).astype(np.float32)
    comps.append(("prefinal-chain.batchnorm", lambda w, m=mean, v=var: _w_batchnorm(w, m, v)))
    cfg.append("component-node name=prefinal-chain.batchnorm component=prefinal-chain.batchnorm input=prefinal-chain.relu")
    P = spec.num_pdfs
    W, b = randw(P, Pd) * np.float32(spec.output_scale), randb(P, 0.5)
    comps.append(("output.affine", lambda w, W=W, b=b: _w_affine(w, "NaturalGradientAffineComponent", W, b)))
    cfg.append("component-node name=output.affine component=output.affine input=prefinal-chain.batchnorm")
    out_src = "output.affine"
    if spec.with_log_softmax:
        comps.append(("output.log-softmax", lambda w, d=P: _w_nonlinear(w, "LogSoftmaxComponent", d)))
        cfg.append("component-node name=output.log-softmax component=output.log-softmax input=output.affine")
        out_src = "output.log-softmax"
    cfg.append(f"output-node name=output input={out_src} objective=linear")
    if spec.xent_branch:
        # chain recipes carry a second (xent) output; it must be ignored by decoding
        Wx, bx = randw(P, prev_dim), randb(P)
        comps.append(("output-xent.affine", lambda w, W=Wx, b=bx: _w_affine(w, "NaturalGradientAffineComponent", W, b)))
        cfg.append(f"component-node name=output-xent.affine component=output-xent.affine input={prev}")
        comps.append(("output-xent.log-softmax", lambda w, d=P: _w_nonlinear(w, "LogSoftmaxComponent", d)))
        cfg.append("component-node name=output-xent.log-softmax component=output-xent.log-softmax input=output-xent.affine")
        cfg.append("output-node name=output-xent input=output-xent.log-softmax objective=linear")
    return cfg, comps


def write_final_mdl(path: Path, spec: ModelSpec) -> None:
    rng = np.random.default_rng(spec.seed)
    w = KaldiWriter(spec.binary)
    _write_transition_model(w, spec)
    cfg, comps = build_nnet(spec, rng)
    w.token("<Nnet3>").raw(b"\n")
    w.raw(("\n".join(cfg) + "\n\n").encode())
    w.token("<NumComponents>").i32(len(comps)).nl()
    for name, fn in comps:
        w.token("<ComponentName>").token(name)
        fn(w)
        w.nl()
    w.token("</Nnet3>").nl()
    w.token("<LeftContext>").i32(0).token("<RightContext>").i32(0)   # recomputed by the reader (am-nnet-simple.cc:52)
    w.token("<Priors>")
    if spec.with_priors:
        pri = rng.uniform(0.2, 1.0, spec.num_pdfs)
        w.vector((pri / pri.sum()).astype(np.float32))
    else:
        w.vector(np.zeros(0, np.float32))
    w.nl()
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_bytes(w.getvalue())


# --------------------------------------------------------------------------- iVector extractor + confs


def write_ivector_extractor(ie_dir: Path, spec: ModelSpec) -> None:
    rng = np.random.default_rng(spec.seed + 7919)
    ie_dir.mkdir(parents=True, exist_ok=True)
    C, Dl, G, Di = spec.num_ceps, spec.lda_dim, spec.num_gauss, spec.ivector_dim
    nsp = spec.splice_left + 1 + spec.splice_right
    # final.mat: affine LDA, D_lda x (C*nsp + 1); scaled so outputs are O(1) for MFCC-sized inputs
    lda = (rng.standard_normal((Dl, C * nsp + 1)) * (0.05 / math.sqrt(C * nsp))).astype(np.float32)
    lda[:, -1] = (rng.standard_normal(Dl) * 0.1).astype(np.float32)
    write_kaldi_matrix_file(ie_dir / "final.mat", lda, spec.binary)
    # global_cmvn.stats: 2 x (C+1) double: sums, sumsq; count in [0][C]
    cnt = 10000.0
    mean = rng.standard_normal(C) * 2.0
    mean[0] += 60.0
    stats = np.zeros((2, C + 1))
    stats[0, :C] = mean * cnt
    stats[0, C] = cnt
    stats[1, :C] = (mean * mean + 25.0) * cnt
    write_kaldi_matrix_file(ie_dir / "global_cmvn.stats", stats, spec.binary, double=True)
    # final.dubm
    means = rng.standard_normal((G, Dl)) * 1.0
    inv_vars = 1.0 / rng.uniform(0.5, 2.0, (G, Dl))
    weights = rng.uniform(0.5, 1.5, G)
    weights /= weights.sum()
    w = KaldiWriter(spec.binary)
    w.token("<DiagGMM>").nl()
    gconsts = np.log(weights) - 0.5 * math.log(2 * math.pi) * Dl + 0.5 * np.log(inv_vars).sum(1) \
        - 0.5 * (means * means * inv_vars).sum(1)
    w.token("<GCONSTS>").vector(gconsts.astype(np.float32))
    w.token("<WEIGHTS>").vector(weights.astype(np.float32))
    w.token("<MEANS_INVVARS>").matrix((means * inv_vars).astype(np.float32))
    w.token("<INV_VARS>").matrix(inv_vars.astype(np.float32))
    w.token("</DiagGMM>").nl()
    (ie_dir / "final.dubm").write_bytes(w.getvalue())
    # final.ie
    w = KaldiWriter(spec.binary)
    w.token("<IvectorExtractor>")
    w.token("<w>").matrix(np.zeros((0, 0)), double=True)
    w.token("<w_vec>").vector(np.log(weights), double=True)
    w.token("<M>").i32(G)
    for g in range(G):
        M = rng.standard_normal((Dl, Di)) * 0.3
        M[:, 0] = means[g] / 5.0          # first column carries the (offset-scaled) mean, as in trained extractors
        w.matrix(M, double=True)
    w.token("<SigmaInv>")
    for g in range(G):
        A = rng.standard_normal((Dl, Dl)) * 0.1
        S = A @ A.T + np.diag(rng.uniform(0.5, 2.0, Dl))
        w.sp_matrix(S, double=True)
    w.token("<IvectorOffset>").f64(5.0)
    w.token("</IvectorExtractor>")
    (ie_dir / "final.ie").write_bytes(w.getvalue())


def write_model_dir(model_dir: Path, spec: ModelSpec) -> None:
    """Writes <model_dir>/model/{model,online} in the layout of SURVEY.md §3.4."""
    model_dir = Path(model_dir).absolute()
    write_final_mdl(model_dir / "model" / "model" / "final.mdl", spec)
    write_tree(model_dir / "model" / "model" / "tree", spec)
    conf = model_dir / "model" / "online" / "conf"
    conf.mkdir(parents=True, exist_ok=True)
    mfcc = ["--use-energy=false", f"--num-mel-bins={spec.num_mel_bins}", f"--num-ceps={spec.num_ceps}",
            "--low-freq=20", "--high-freq=-400", "--sample-frequency=16000"]
    if spec.dither is not None:
        mfcc.append(f"--dither={spec.dither}")
    if spec.frame_length is not None:
        mfcc.append(f"--frame-length={spec.frame_length}")
    (conf / "mfcc.conf").write_text("# hires MFCC (egs/wsj/s5/conf/mfcc_hires.conf)\n" + "\n".join(mfcc) + "\n")
    online = ["--feature-type=mfcc", f"--mfcc-config={conf / 'mfcc.conf'}"]
    if spec.ivector_dim > 0:
        ie = model_dir / "model" / "online" / "ivector_extractor"
        write_ivector_extractor(ie, spec)
        (conf / "splice.conf").write_text(f"--left-context={spec.splice_left}\n--right-context={spec.splice_right}\n")
        (conf / "online_cmvn.conf").write_text("# configuration file for apply-cmvn-online, used in the script ../local/run_online_decoding.sh\n")
        ivc = [f"--splice-config={conf / 'splice.conf'}", f"--cmvn-config={conf / 'online_cmvn.conf'}",
               f"--lda-matrix={ie / 'final.mat'}", f"--global-cmvn-stats={ie / 'global_cmvn.stats'}",
               f"--diag-ubm={ie / 'final.dubm'}", f"--ivector-extractor={ie / 'final.ie'}",
               "--num-gselect=5", "--min-post=0.025", "--posterior-scale=0.1",
               "--max-remembered-frames=1000", "--max-count=100", "--ivector-period=10"]
        (conf / "ivector_extractor.conf").write_text("\n".join(ivc) + "\n")
        online.append(f"--ivector-extraction-config={conf / 'ivector_extractor.conf'}")
    if spec.nnet_cmvn:
        if spec.ivector_dim <= 0:
            raise ValueError("nnet_cmvn needs the extractor's global_cmvn.stats")
        (conf / "nnet_cmvn.conf").write_text("--cmn-window=600\n")
        online.append(f"--cmvn-config={conf / 'nnet_cmvn.conf'}")
        online.append(f"--global-cmvn-stats={model_dir / 'model' / 'online' / 'ivector_extractor' / 'global_cmvn.stats'}")
    online.append("--endpoint.silence-phones=1")
    (conf / "online.conf").write_text("\n".join(online) + "\n")


# --------------------------------------------------------------------------- HCLG graphs

_FST_MAGIC = 2125659606


@dataclass
class Fst:
    """Mutable arc list used to assemble a graph before writing it as ConstFst."""
    arcs: List[List[Tuple[int, int, float, int]]] = field(default_factory=list)   # per state: (ilabel, olabel, cost, next)
    finals: Dict[int, float] = field(default_factory=dict)
    start: int = 0

    def add_state(self) -> int:
        self.arcs.append([])
        return len(self.arcs) - 1

    def add_arc(self, s: int, il: int, ol: int, cost: float, n: int) -> None:
        self.arcs[s].append((il, ol, float(np.float32(cost)), n))

    @property
    def num_states(self) -> int:
        return len(self.arcs)

    @property
    def num_arcs(self) -> int:
        return sum(len(a) for a in self.arcs)


def _fst_string(s: str) -> bytes:
    return struct.pack("<i", len(s)) + s.encode()


def write_const_fst(path: Path, fst: Fst) -> None:
    """OpenFst ConstFst<StdArc> v2 (unaligned) binary: header (kaldi/openfst/src/lib/fst.cc:84-96), then
    states {f32 final, u32 pos, u32 narcs, u32 nieps, u32 noeps}, then arcs {i32,i32,f32,i32}
    (include/fst/const-fst.h:102-110).  Arcs are ilabel-sorted per state like mkgraph.sh's output."""
    states = np.zeros(fst.num_states, dtype=[("w", "<f4"), ("pos", "<u4"), ("n", "<u4"), ("ni", "<u4"), ("no", "<u4")])
    arcs = np.zeros(fst.num_arcs, dtype=[("il", "<i4"), ("ol", "<i4"), ("w", "<f4"), ("ns", "<i4")])
    pos = 0
    for s, al in enumerate(fst.arcs):
        al = sorted(al, key=lambda a: (a[0], a[1], a[3]))
        states[s] = (fst.finals.get(s, np.inf), pos, len(al), sum(1 for a in al if a[0] == 0), sum(1 for a in al if a[1] == 0))
        for a in al:
            arcs[pos] = a
            pos += 1
    props = 0x1  # kExpanded; the decoder never looks at the rest
    hdr = struct.pack("<i", _FST_MAGIC) + _fst_string("const") + _fst_string("standard") + struct.pack(
        "<iiQqqq", 2, 0, props, fst.start, fst.num_states, fst.num_arcs)
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        f.write(hdr)
        f.write(states.tobytes())
        f.write(arcs.tobytes())


def write_vector_fst(path: Path, fst: Fst) -> None:
    """OpenFst VectorFst<StdArc> v2 binary (include/fst/vector-fst.h WriteFst): per state final f32,
    i64 narcs, arcs."""
    hdr = struct.pack("<i", _FST_MAGIC) + _fst_string("vector") + _fst_string("standard") + struct.pack(
        "<iiQqqq", 2, 0, 0x1, fst.start, fst.num_states, fst.num_arcs)
    with open(path, "wb") as f:
        f.write(hdr)
        for s, al in enumerate(fst.arcs):
            al = sorted(al, key=lambda a: (a[0], a[1], a[3]))
            f.write(struct.pack("<fq", fst.finals.get(s, np.inf), len(al)))
            for a in al:
                f.write(struct.pack("<iifi", *a))


DEFAULT_SENTENCES = """
turn on the garage light
turn off the garage light
turn on the living room lamp
turn off the living room lamp
turn on the bedroom light
turn off the bedroom light
what time is it
what is the temperature
how hot is it
how cold is it
is the garage door open
is the garage door closed
set the bedroom light to red
set the bedroom light to green
set the bedroom light to blue
set the living room lamp to red
set the living room lamp to green
set the living room lamp to blue
tell me the time
whats the temperature
open the garage door
close the garage door
set a timer for five minutes
set a timer for ten minutes
set a timer for twenty minutes
stop the timer
pause the music
play the music
next song
previous song
volume up
volume down
turn up the volume
turn down the volume
good morning
good night
""".strip().splitlines()


@dataclass
class Lexicon:
    words: List[str]                      # id = index (0 = <eps>)
    prons: Dict[int, List[int]]           # word id -> phone ids (1-based); phone 1 = silence
    sil_phone: int = 1


def make_lexicon(sentences: Sequence[Sequence[str]], spec: ModelSpec, rng: np.random.Generator,
                 extra_words: int = 0) -> Lexicon:
    vocab = sorted({w for s in sentences for w in s})
    vocab += [f"w{i:05d}" for i in range(extra_words)]
    words = ["<eps>"] + vocab
    prons = {}
    for i in range(1, len(words)):
        n = int(rng.integers(2, 7))
        prons[i] = [int(x) for x in rng.integers(2, spec.num_phones + 1, n)]
    return Lexicon(words, prons)


def _add_phone(fst: Fst, spec: ModelSpec, src: int, phone: int, olabel: int, entry_cost: float, dst: Optional[int] = None) -> int:
    """One HMM instance as add-self-loops leaves it for the 1-emitting-state topologies used here:
    forward arc (forward tid) into a state carrying the self-loop tid (cost -log 0.5 each,
    --self-loop-scale 1.0 / --transition-scale 1.0, kaldi.py:415-425)."""
    sl, fw = transition_ids(spec, phone)
    st = fst.add_state() if dst is None else dst
    fst.add_arc(src, fw, olabel, entry_cost + math.log(2.0), st)
    fst.add_arc(st, sl, 0, math.log(2.0), st)
    return st


def make_grammar_hclg(sentences: Sequence[Sequence[str]], lex: Lexicon, spec: ModelSpec,
                      rng: np.random.Generator) -> Fst:
    """Grammar-style HCLG: prefix tree over the sentences (a determinised G), each word expanded to its
    phone HMMs, optional silence between words through epsilon arcs, sentence-final states."""
    wid = {w: i for i, w in enumerate(lex.words)}
    fst = Fst()
    root = fst.add_state()
    fst.start = root
    # optional leading silence
    g0 = fst.add_state()
    fst.add_arc(root, 0, 0, math.log(2.0), g0)
    s_sil = _add_phone(fst, spec, root, lex.sil_phone, 0, math.log(2.0))
    fst.add_arc(s_sil, 0, 0, 0.0, g0)
    trie: Dict[Tuple[int, ...], int] = {(): g0}
    counts: Dict[Tuple[int, ...], int] = {}
    for s in sentences:
        ids = tuple(wid[w] for w in s)
        for k in range(len(ids) + 1):
            counts[ids[:k]] = counts.get(ids[:k], 0) + 1
    for s in sentences:
        ids = tuple(wid[w] for w in s)
        for k in range(1, len(ids) + 1):
            pre = ids[:k]
            if pre in trie:
                continue
            src = trie[pre[:-1]]
            cost = -math.log(counts[pre] / counts[pre[:-1]])
            cur = src
            for j, ph in enumerate(lex.prons[pre[-1]]):
                cur = _add_phone(fst, spec, cur, ph, pre[-1] if j == 0 else 0, cost if j == 0 else 0.0)
            # word end: optional silence, joined by epsilons at the next grammar state
            nxt = fst.add_state()
            fst.add_arc(cur, 0, 0, math.log(2.0), nxt)
            s_sil = _add_phone(fst, spec, cur, lex.sil_phone, 0, math.log(2.0))
            fst.add_arc(s_sil, 0, 0, 0.0, nxt)
            trie[pre] = nxt
    ends: Dict[Tuple[int, ...], int] = {}
    for s in sentences:
        ids = tuple(wid[w] for w in s)
        ends[ids] = ends.get(ids, 0) + 1
    for ids, c in ends.items():
        fst.finals[trie[ids]] = float(np.float32(-math.log(c / counts[ids])))
    return fst


def make_arpa_hclg(sentences: Sequence[Sequence[str]], lex: Lexicon, spec: ModelSpec,
                   rng: np.random.Generator) -> Fst:
    """ARPA-style HCLG: bigram back-off LM (history states, epsilon back-off arcs to a unigram state that
    fans out to the whole vocabulary) composed with the lexicon.  Exercises epsilon closure and, with a
    large vocabulary, max_active/min_active pruning."""
    wid = {w: i for i, w in enumerate(lex.words)}
    V = len(lex.words) - 1
    uni = np.ones(V + 1)
    uni[0] = 0
    big: Dict[int, Dict[int, int]] = {}
    for s in sentences:
        ids = [wid[w] for w in s]
        prev = 0
        for i in ids:
            uni[i] += 3
            big.setdefault(prev, {})[i] = big.get(prev, {}).get(i, 0) + 1
            prev = i
    uni_p = uni / uni.sum()
    fst = Fst()
    start = fst.add_state()       # sentence start history (<s>)
    fst.start = start
    ug = fst.add_state()          # unigram (back-off) state
    hist: Dict[int, int] = {0: start}

    def hstate(w: int) -> int:
        if w not in hist:
            hist[w] = fst.add_state()
        return hist[w]

    def add_word(src: int, w: int, cost: float, dst: int) -> None:
        cur = src
        pr = lex.prons[w]
        for j, ph in enumerate(pr):
            cur = _add_phone(fst, spec, cur, ph, w if j == 0 else 0, cost if j == 0 else 0.0)
        fst.add_arc(cur, 0, 0, math.log(2.0), dst)
        s_sil = _add_phone(fst, spec, cur, lex.sil_phone, 0, math.log(2.0))
        fst.add_arc(s_sil, 0, 0, 0.0, dst)

    for w in range(1, V + 1):
        add_word(ug, w, -math.log(uni_p[w]), hstate(w))
    for h, nxt in big.items():
        tot = sum(nxt.values())
        lam = tot / (tot + len(nxt))                 # Witten-Bell interpolation weight
        hs = hstate(h)
        for w, c in nxt.items():
            add_word(hs, w, -math.log(lam * c / tot + (1 - lam) * uni_p[w]), hstate(w))
        fst.add_arc(hs, 0, 0, -math.log(1 - lam), ug)
    for w, st in hist.items():
        if w not in big:
            fst.add_arc(st, 0, 0, 0.0, ug)           # unseen history: free back-off
        if w != 0:
            fst.finals[st] = float(np.float32(2.0 if w in big else 1.0))
    fst.finals[ug] = 4.0
    return fst


def write_graph_dir(graph_dir: Path, fst: Fst, lex: Lexicon, const: bool = True) -> None:
    graph_dir = Path(graph_dir)
    graph_dir.mkdir(parents=True, exist_ok=True)
    (write_const_fst if const else write_vector_fst)(graph_dir / "HCLG.fst", fst)
    (graph_dir / "words.txt").write_text("".join(f"{w} {i}\n" for i, w in enumerate(lex.words)))


def make_grammar_graph(graph_dir: Path, spec: ModelSpec, seed: int = 11, sentences: Optional[Sequence[str]] = None) -> Tuple[Fst, Lexicon]:
    rng = np.random.default_rng(seed)
    sents = [s.split() for s in (sentences or DEFAULT_SENTENCES)]
    lex = make_lexicon(sents, spec, rng)
    fst = make_grammar_hclg(sents, lex, spec, rng)
    write_graph_dir(graph_dir, fst, lex)
    return fst, lex


def make_arpa_graph(graph_dir: Path, spec: ModelSpec, seed: int = 13, extra_words: int = 2000,
                    num_random_sentences: int = 3000) -> Tuple[Fst, Lexicon]:
    rng = np.random.default_rng(seed)
    sents = [s.split() for s in DEFAULT_SENTENCES]
    lex = make_lexicon(sents, spec, rng, extra_words=extra_words)
    vocab = lex.words[1:]
    for _ in range(num_random_sentences):
        n = int(rng.integers(2, 8))
        sents.append([vocab[int(i)] for i in rng.integers(0, len(vocab), n)])
    fst = make_arpa_hclg(sents, lex, spec, rng)
    write_graph_dir(graph_dir, fst, lex)
    return fst, lex


# --------------------------------------------------------------------------- audio


def synth_utterance(u: int, num_samples: int = 48000, sample_rate: int = 16000) -> np.ndarray:
    """Deterministic int16 test signal: 5 sinusoids in [100, 4000] Hz with a 4 Hz
    amplitude envelope plus uniform noise, peak ~8000."""
    rng = np.random.default_rng(1000 + u)
    t = np.arange(num_samples, dtype=np.float64) / sample_rate
    x = np.zeros(num_samples)
    for _ in range(5):
        f = rng.uniform(100.0, 4000.0)
        ph = rng.uniform(0, 2 * math.pi)
        fm = rng.uniform(0.5, 3.0)
        x += np.sin(2 * math.pi * f * t + ph + 2.0 * np.sin(2 * math.pi * fm * t))
    env = 0.55 + 0.45 * np.sin(2 * math.pi * 4.0 * t + rng.uniform(0, 2 * math.pi))
    x = x * env * 1500.0 + rng.uniform(-200, 200, num_samples)
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)


def write_wav(path: Path, pcm: np.ndarray, sample_rate: int = 16000) -> None:
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(np.asarray(pcm, dtype="<i2").tobytes())
