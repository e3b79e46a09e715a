"""TEST INFRASTRUCTURE ONLY -- CPU restatement of *how many times the reference's model set-up calls rand()*.

Why this exists.  The reference dithers every frame with Gaussian noise seeded from glibc's rand()
(feat/feature-window.cc:90-98, base/kaldi-math.cc:59-70), rhasspy starts one decoder process per utterance
(rhasspy_speech/tools.py:117-147) and the reference's default mfcc configuration leaves dither on
(feat/feature-window.h:57).  The noise of frame t is therefore a function of (t, number of rand() calls made before the
first frame), and both decoder binaries make thousands of such calls while they set the model up -- all of them inside
nnet3, all of them deterministic functions of the network's structure:

  * nnet3/nnet-utils.cc:92-144    ComputeSimpleNnetContextForShift: `rand() % 10`, then a computation graph is built
                                  (called from AmNnetSimple::Read -> SetContext and again from
                                  DecodableNnetSimpleLoopedInfo::Init, decodable-simple-looped.cc:55-62)
  * nnet3/nnet-computation-graph.cc:462-493  ComputationGraphBuilder::Compute: one RandInt per BuildGraphOneIter round
                                  (none when the range is [1,1]) and one at the end
  * nnet3/nnet-computation-graph.cc:496-570  ComputationGraphBuilder::Check: two RandInt per visited cindex
  * nnet3/nnet-utils.cc:1777-1856 CollapseComponentsAffine: AffineComponent::Init -> two SetRandn -> two RandomState()
  * nnet3/nnet-tdnn-component.cc:553   TdnnComponent::PrecomputeIndexes: one RandInt per compiled step of such a component
  * nnet3/nnet-optimize-utils.cc:4654  InsertCommands (from SplitRowOps, :2883-2892): one RandInt when a multi-row command
                                  was split in two, which every looped computation with temporal context has

This module restates that machinery -- the descriptor algebra (nnet-descriptor.cc), the node list (nnet-nnet.cc:189-460),
CollapseModel's rewiring (nnet-utils.cc:1459-2110), the graph builder including its queue discipline, usable counts and
Prune(), and the request sequences of ComputeSimpleNnetContext and CompileLooped (nnet-compile-looped.cc:131-345) -- and
counts.  Pinned against the reference itself: `rs-dump randpos` (oracle/drivers/rs-dump.cc) reports the position of
rand() after the reference's own set-up; tests/test_oracle_golden.py compares on every model shape of the suite.
"""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import kaldi_formats as kf

HERE = Path(__file__).resolve().parent


# =========================================================================================== glibc rand()

class GlibcRand:
    """rand() of a fresh process (stdlib/random_r.c TYPE_3, default seed): values from oracle/dither.c, a call counter."""

    def __init__(self):
        self._vals = np.zeros(0, np.int32)
        self.pos = 0

    def _more(self, n: int):
        from .pipeline import decoder_lib
        lib = decoder_lib()
        lib.oracle_glibc_rand.argtypes = [C.c_int, C.c_void_p]
        lib.oracle_glibc_rand.restype = None
        v = np.zeros(n, np.int32)
        lib.oracle_glibc_rand(n, v.ctypes.data_as(C.c_void_p))
        self._vals = v

    def rand(self) -> int:
        if self.pos >= len(self._vals):
            self._more(max(1 << 16, 2 * len(self._vals)))
        v = int(self._vals[self.pos])
        self.pos += 1
        return v

    def rand_int(self, lo: int, hi: int) -> int:
        # base/kaldi-math.cc:95-119 (non-MSVC branch): no draw at all for an empty range
        if hi == lo:
            return lo
        return lo + self.rand() % (hi + 1 - lo)


# =========================================================================================== descriptors
# GeneralDescriptor (nnet-descriptor.cc:597-1010): parse, NormalizeAppend, Normalize to a fixpoint.  A node of the tree is
# a list [kind, children, v1, v2, alpha]; kinds as in the reference's enum.

_RESERVED = {"Append", "Sum", "Failover", "IfDefined", "Offset", "Switch", "Scale", "Const", "Round", "ReplaceIndex"}


def tokenize(s: str) -> List[str]:
    return re.findall(r"[(),]|[^\s(),]+", s)


class _G:
    __slots__ = ("kind", "ch", "v1", "v2", "alpha")

    def __init__(self, kind, ch=None, v1=-1, v2=-1, alpha=0.0):
        self.kind, self.ch, self.v1, self.v2, self.alpha = kind, (ch if ch is not None else []), v1, v2, alpha


def _parse(tok: List[str], p: int, names: List[str]) -> Tuple[_G, int]:
    t = tok[p]
    if t not in _RESERVED:
        if t not in names:
            raise ValueError(f"descriptor: unknown node {t!r}")
        return _G("Node", v1=names.index(t)), p + 1
    assert tok[p + 1] == "("
    p += 2
    g = _G(t)
    if t in ("Append", "Sum", "Switch"):
        while True:
            c, p = _parse(tok, p, names)
            g.ch.append(c)
            if tok[p] == ")":
                return g, p + 1
            assert tok[p] == ","
            p += 1
    if t == "Failover":
        a, p = _parse(tok, p, names)
        assert tok[p] == ","
        b, p = _parse(tok, p + 1, names)
        assert tok[p] == ")"
        g.ch = [a, b]
        return g, p + 1
    if t == "IfDefined":
        a, p = _parse(tok, p, names)
        assert tok[p] == ")"
        g.ch = [a]
        return g, p + 1
    if t == "Scale":
        g.alpha = float(tok[p])
        assert tok[p + 1] == ","
        a, p = _parse(tok, p + 2, names)
        assert tok[p] == ")"
        g.ch = [a]
        return g, p + 1
    if t == "Const":
        g.alpha = float(tok[p])
        g.v1 = int(tok[p + 2])
        assert tok[p + 3] == ")"
        return g, p + 4
    if t == "Offset":
        a, p = _parse(tok, p, names)
        assert tok[p] == ","
        g.v1 = int(tok[p + 1])
        p += 2
        g.v2 = 0
        if tok[p] == ",":
            g.v2 = int(tok[p + 1])
            p += 2
        assert tok[p] == ")"
        g.ch = [a]
        return g, p + 1
    if t == "Round":
        a, p = _parse(tok, p, names)
        g.v1 = int(tok[p + 1])
        assert tok[p + 2] == ")"
        g.ch = [a]
        return g, p + 3
    if t == "ReplaceIndex":
        a, p = _parse(tok, p, names)
        g.v1 = {"t": 0, "x": 1}[tok[p + 1]]     # variable
        g.v2 = int(tok[p + 3])
        assert tok[p + 4] == ")"
        g.ch = [a]
        return g, p + 5
    raise AssertionError(t)


def _num_append_terms(g: _G) -> int:
    if g.kind in ("Node", "Const"):
        return 1
    if g.kind == "Append":
        return sum(_num_append_terms(c) for c in g.ch)
    n = _num_append_terms(g.ch[0])
    assert all(_num_append_terms(c) == n for c in g.ch[1:])
    return n


def _append_term(g: _G, term: int) -> _G:
    if g.kind == "Node":
        return _G("Node", v1=g.v1)
    if g.kind == "Const":
        return _G("Const", v1=g.v1, alpha=g.alpha)
    if g.kind == "Append":
        for c in g.ch:
            n = _num_append_terms(c)
            if term < n:
                return _append_term(c, term)
            term -= n
        raise AssertionError
    return _G(g.kind, [_append_term(c, term) for c in g.ch], g.v1, g.v2, g.alpha)


def _take(dst: _G, src: _G):
    dst.kind, dst.ch, dst.v1, dst.v2, dst.alpha = src.kind, src.ch, src.v1, src.v2, src.alpha


def _normalize(d: _G) -> bool:
    """GeneralDescriptor::Normalize (nnet-descriptor.cc:818-958), one pass; True if something changed."""
    changed = False
    k = d.kind
    fall = False
    if k == "Offset":
        child = d.ch[0]
        if child.kind == "Offset":
            d.v1 += child.v1
            d.v2 += child.v2
            d.ch = [child.ch[0]]
            changed = True
            fall = True
        elif d.v1 == 0 and d.v2 == 0:
            _take(d, child)
            changed = True
        else:
            fall = True
    if fall or k in ("Switch", "Round", "ReplaceIndex"):
        child = d.ch[0]
        assert child.kind != "Append"
        if child.kind in ("Sum", "Failover", "IfDefined"):
            assert len(d.ch) == 1
            child.ch = [_G(d.kind, [gc], d.v1, d.v2, d.alpha) for gc in child.ch]
            d.kind, d.v1, d.v2 = child.kind, child.v1, child.v2
            d.ch = child.ch
            changed = True
    elif k == "Sum":
        if len(d.ch) == 1:
            _take(d, d.ch[0])
            changed = True
        elif len(d.ch) > 2:
            d.ch = [d.ch[0], _G("Sum", d.ch[1:])]
            changed = True
    elif k == "Scale":
        child = d.ch[0]
        if child.kind in ("Offset", "ReplaceIndex", "Round"):
            d.kind, child.kind = child.kind, d.kind
            d.alpha, child.alpha = child.alpha, d.alpha
            d.v1, child.v1 = child.v1, d.v1
            d.v2, child.v2 = child.v2, d.v2
            changed = True
        elif child.kind == "Sum":
            d.ch = [_G("Scale", [gc], -1, -1, d.alpha) for gc in child.ch]
            d.kind, d.alpha = "Sum", 0.0
            changed = True
        elif child.kind == "Scale":
            d.alpha *= child.alpha
            d.ch = [child.ch[0]]
            changed = True
        elif child.kind != "Node":
            raise ValueError("unhandled Scale() nesting")
    for c in d.ch:
        changed = changed or _normalize(c)      # (sic: the reference short-circuits the same way)
    return changed


# --- the normalised forms (ForwardingDescriptor / SumDescriptor / Descriptor)

Index = Tuple[int, int, int]            # (n, t, x)
Cindex = Tuple[int, Index]


class Fwd:
    """ForwardingDescriptor: chain of index maps ending in a node."""

    def __init__(self, g: _G):
        self.g = g

    def map(self, ind: Index) -> Cindex:
        g, (n, t, x) = self.g, ind
        while True:
            if g.kind == "Node":
                return (g.v1, (n, t, x))
            if g.kind == "Scale":
                g = g.ch[0]
            elif g.kind == "Offset":
                t, x = t + g.v1, x + g.v2
                g = g.ch[0]
            elif g.kind == "Round":
                t -= t % g.v1            # python's % is the mathematical modulus the reference spells out
                g = g.ch[0]
            elif g.kind == "ReplaceIndex":
                if g.v1 == 0:
                    t = g.v2
                else:
                    x = g.v2
                g = g.ch[0]
            elif g.kind == "Switch":
                g = g.ch[t % len(g.ch)]
            else:
                raise AssertionError(g.kind)

    def modulus(self, g=None) -> int:
        g = g or self.g
        if g.kind == "Node":
            return 1
        if g.kind == "Round":
            return g.v1
        if g.kind == "Switch":
            m = len(g.ch)
            for c in g.ch:
                m = _lcm(m, self.modulus(c))
            return m
        return self.modulus(g.ch[0])


def _lcm(a, b):
    from math import gcd
    return a * b // gcd(a, b)


class Sum:
    """SumDescriptor tree: Simple(fwd) | Optional(sum) | Binary(op, a, b) | Const."""

    def __init__(self, g: _G):
        self.g = g
        self.kind = g.kind
        if g.kind == "IfDefined":
            self.a = Sum(g.ch[0])
        elif g.kind in ("Sum", "Failover"):
            assert len(g.ch) == 2
            self.a, self.b = Sum(g.ch[0]), Sum(g.ch[1])
        elif g.kind == "Const":
            pass
        else:
            self.kind = "Simple"
            self.fwd = Fwd(g)

    def deps(self, ind: Index, out: List[Cindex]):
        if self.kind == "Simple":
            out.append(self.fwd.map(ind))
        elif self.kind == "IfDefined":
            self.a.deps(ind, out)
        elif self.kind in ("Sum", "Failover"):
            self.a.deps(ind, out)
            self.b.deps(ind, out)

    def computable(self, ind: Index, cset, used: Optional[List[Cindex]]) -> bool:
        if self.kind == "Simple":
            c = self.fwd.map(ind)
            ok = cset(c)
            if ok and used is not None:
                used.append(c)
            return ok
        if self.kind == "Const":
            return True
        if self.kind == "IfDefined":
            # OptionalSumDescriptor::IsComputable (nnet-descriptor.h): the source's used inputs if it is computable, always true
            tmp: List[Cindex] = []
            if self.a.computable(ind, cset, tmp if used is not None else None) and used is not None:
                used.extend(tmp)
            return True
        u1, u2 = ([], []) if used is not None else (None, None)
        c1, c2 = self.a.computable(ind, cset, u1), self.b.computable(ind, cset, u2)
        if self.kind == "Sum":
            if c1 and c2:
                if used is not None:
                    used.extend(u1)
                    used.extend(u2)
                return True
            return False
        if c1:
            if used is not None:
                used.extend(u1)
            return True
        if c2:
            if used is not None:
                used.extend(u2)
            return True
        return False

    def modulus(self) -> int:
        if self.kind == "Simple":
            return self.fwd.modulus()
        if self.kind == "Const":
            return 1
        if self.kind == "IfDefined":
            return self.a.modulus()
        return _lcm(self.a.modulus(), self.b.modulus())


def _print(g: _G, names: List[str]) -> str:
    k = g.kind
    if k == "Node":
        return names[g.v1]
    if k == "Const":
        return f"Const({g.alpha:g}, {g.v1})"
    if k == "Scale":
        return f"Scale({g.alpha:g}, {_print(g.ch[0], names)})"
    if k == "Offset":
        return f"Offset({_print(g.ch[0], names)}, {g.v1}" + (f", {g.v2}" if g.v2 != 0 else "") + ")"
    if k == "Round":
        return f"Round({_print(g.ch[0], names)}, {g.v1})"
    if k == "ReplaceIndex":
        return f"ReplaceIndex({_print(g.ch[0], names)}, {'t' if g.v1 == 0 else 'x'}, {g.v2})"
    return f"{k}(" + ", ".join(_print(c, names) for c in g.ch) + ")"


class Descriptor:
    def __init__(self, text: str, names: List[str]):
        tok = tokenize(text) + ["end of input"]
        g, p = _parse(tok, 0, names)
        assert tok[p] == "end of input", (text, tok[p])
        n = _num_append_terms(g)
        g = _append_term(g, 0) if n == 1 else _G("Append", [_append_term(g, i) for i in range(n)])
        while _normalize(g):
            pass
        self.g = g
        self.parts = [Sum(c) for c in g.ch] if g.kind == "Append" else [Sum(g)]

    def text(self, names: List[str]) -> str:
        return _print(self.g, names)

    def deps(self, ind: Index) -> List[Cindex]:
        out: List[Cindex] = []
        for p in self.parts:
            p.deps(ind, out)
        return out

    def computable(self, ind: Index, cset, used: Optional[List[Cindex]]) -> bool:
        if used is not None:
            used.clear()
        for p in self.parts:
            if not p.computable(ind, cset, used):
                if used is not None:
                    used.clear()
                return False
        return True

    def modulus(self) -> int:
        m = 1
        for p in self.parts:
            m = _lcm(m, p.modulus())
        return m

    def collapsible_node(self) -> int:
        """ModelCollapser::DescriptorIsCollapsible (nnet-utils.cc:1551-1565): every part is `foo` or `Offset(foo, k)` of ONE node."""
        ans = None
        for p in self.parts:
            n = -1
            if p.kind == "Simple":
                g = p.fwd.g
                if g.kind == "Offset":
                    g = g.ch[0]
                if g.kind == "Node":      # note: a Scale()d node is a SimpleForwardingDescriptor with a scale as well
                    n = g.v1
                elif g.kind == "Scale" and g.ch[0].kind == "Node":
                    n = g.ch[0].v1
            if ans is None:
                ans = n
            elif ans != -1 and n != ans:
                ans = -1
        return -1 if ans is None else ans


# =========================================================================================== the network's node list

@dataclass
class Node:
    kind: str                      # "input" | "descriptor" | "component" | "dimrange"
    name: str
    desc: Optional[Descriptor] = None
    comp: int = -1
    src: int = -1                  # dimrange
    line: Dict[str, str] = field(default_factory=dict)


def _config_fields(line: str) -> Tuple[str, Dict[str, str]]:
    first, rest = line.strip().split(None, 1)
    parts = re.split(r"\s+(?=[A-Za-z][A-Za-z0-9_\-]*=)", rest.strip())
    return first, {p.split("=", 1)[0]: p.split("=", 1)[1].strip() for p in parts}


_AFFINE = {"AffineComponent", "NaturalGradientAffineComponent"}      # dynamic_cast<AffineComponent*> succeeds


class Net:
    """Node list in the reference's order (nnet-nnet.cc:189-244: nodes appear in config-line order, a component-node is the
    pair `<name>_input` (descriptor), `<name>` (component))."""

    def __init__(self, nf: kf.NnetFile):
        self.comp_names = list(nf.components.keys())
        self.comps = [nf.components[k] for k in self.comp_names]
        lines = [_config_fields(ln) for ln in nf.config if ln.strip() and not ln.strip().startswith("#")]
        self.nodes: List[Node] = []
        for first, f in lines:
            if first == "input-node":
                self.nodes.append(Node("input", f["name"], line=f))
            elif first == "component-node":
                self.nodes.append(Node("descriptor", f["name"] + "_input", line=f))
                self.nodes.append(Node("component", f["name"], line=f))
            elif first == "output-node":
                self.nodes.append(Node("descriptor", f["name"], line=f))
            elif first == "dim-range-node":
                self.nodes.append(Node("dimrange", f["name"], line=f))
            else:
                raise ValueError(f"oracle: unexpected config line {first}")
        names = self.names()
        some = [n.name for n in self.nodes if not (n.kind == "descriptor" and n.name.endswith("_input") and "component" in n.line)]
        for i, n in enumerate(self.nodes):
            if n.kind == "descriptor":
                n.desc = Descriptor(n.line["input"], names)
            elif n.kind == "component":
                n.comp = self.comp_names.index(n.line["component"])
            elif n.kind == "dimrange":
                n.src = names.index(n.line["input-node"])
        del some

    def names(self) -> List[str]:
        return [n.name for n in self.nodes]

    def index(self, name: str) -> int:
        for i, n in enumerate(self.nodes):
            if n.name == name:
                return i
        return -1

    def is_output(self, i: int) -> bool:
        return self.nodes[i].kind == "descriptor" and (i + 1 == len(self.nodes) or self.nodes[i + 1].kind != "component")

    def modulus(self) -> int:
        m = 1
        for n in self.nodes:
            if n.kind == "descriptor":
                m = _lcm(m, n.desc.modulus())
        return m

    # ---- component properties the collapser and the graph builder look at
    def ctype(self, c: int) -> str:
        return self.comps[c].type

    def time_offsets(self, c: int) -> Optional[List[int]]:
        comp = self.comps[c]
        if comp.type == "TdnnComponent":
            return [int(v) for v in comp.fields["<TimeOffsets>"]]
        return None


def _affine_dims(comp: kf.Comp) -> Tuple[int, int]:
    m = comp.fields.get("<LinearParams>")
    if m is None:
        m = comp.fields["<Params>"]
    m = np.asarray(m)
    if comp.type == "FixedAffineComponent" and "<LinearParams>" not in comp.fields:
        raise ValueError("FixedAffineComponent without <LinearParams>")
    return int(m.shape[1]), int(m.shape[0])      # (input dim, output dim)


def _bn_scale_offset(comp: kf.Comp) -> Tuple[np.ndarray, np.ndarray]:
    """BatchNormComponent::ComputeDerived (nnet-normalize-component.cc:209-245) from the stored mean / variance (Read converts
    them to sums and back); only asked whether the transform is the identity."""
    f = comp.fields
    F32 = np.float32
    count, eps, rms = float(f["<Count>"]), F32(f["<Epsilon>"]), F32(f["<TargetRms>"])
    mean, var = np.asarray(f["<StatsMean>"], F32), np.asarray(f["<StatsVar>"], F32)
    sumsq = ((var + mean * mean).astype(F32) * F32(count)).astype(F32)
    ssum = (mean * F32(count)).astype(F32)
    off = (ssum * F32(-1.0 / count)).astype(F32)
    scl = (sumsq * F32(1.0 / count)).astype(F32)
    scl = (scl + F32(-1.0) * off * off).astype(F32)
    scl = np.maximum(scl, F32(0.0)) + eps
    scl = np.power(scl, F32(-0.5)).astype(F32) * rms
    return scl, (off * scl).astype(F32)


# =========================================================================================== CollapseModel (structure only)

def collapse_model(net: Net, rng: GlibcRand, all_four: bool = False) -> None:
    """ModelCollapser::Collapse (nnet-utils.cc:1459-1482) on the node list: which component nodes get bypassed, which
    components get created -- and the two rand() calls of every AffineComponent::Init (:1850).  Parameters are not
    touched here (the forward pass of oracle/pipeline.py evaluates the un-collapsed network; the difference is rounding)."""

    def comp_index(name: str) -> int:
        return net.comp_names.index(name) if name in net.comp_names else -1

    def add(name: str, comp: kf.Comp) -> int:
        net.comp_names.append(name)
        net.comps.append(comp)
        return len(net.comps) - 1

    def scaled(c2: int, scale: float) -> int:                 # GetScaledComponentIndex (:2062-2108)
        if scale == 1.0:
            return c2
        name = f"{net.comp_names[c2]}.scale{scale:.3g}"
        i = comp_index(name)
        if i >= 0:
            return i
        if net.ctype(c2) not in _AFFINE | {"LinearComponent", "TdnnComponent", "TimeHeightConvolutionComponent"}:
            return -1
        return add(name, net.comps[c2])

    def dropout(c1: int, c2: int) -> int:                     # CollapseComponentsDropout (:1700-1727)
        t = net.ctype(c1)
        if t == "DropoutComponent":
            p = float(net.comps[c1].fields.get("<DropoutProportion>", 0.0))
            return scaled(c2, float(np.float32(1.0) / (np.float32(1.0) - np.float32(p))))
        if t == "GeneralDropoutComponent":
            return scaled(c2, 1.0)
        return -1

    def batchnorm(c1: int, c2: int) -> int:                   # CollapseComponentsBatchnorm (:1742-1764) + (:1938-1996)
        if net.ctype(c1) != "BatchNormComponent":
            return -1
        scale, offset = _bn_scale_offset(net.comps[c1])
        if offset.max() == 0.0 and offset.min() == 0.0 and scale.max() == 1.0 and scale.min() == 1.0:
            return c2
        name = f"{net.comp_names[c1]}.{net.comp_names[c2]}"
        i = comp_index(name)
        if i >= 0:
            return i
        t2 = net.ctype(c2)
        if t2 in _AFFINE:
            return add(name, net.comps[c2])
        if t2 == "LinearComponent":
            # becomes a plain AffineComponent (constructor from parameters: no Init, no rand())
            lin = net.comps[c2]
            return add(name, kf.Comp("AffineComponent", {"<LinearParams>": lin.fields["<Params>"]}))
        if t2 == "TdnnComponent":
            return add(name, net.comps[c2])
        return -1

    def affine(c1: int, c2: int) -> int:                      # CollapseComponentsAffine (:1777-1856)
        t1, t2 = net.ctype(c1), net.ctype(c2)
        if t2 not in _AFFINE or (t1 != "FixedAffineComponent" and t1 not in _AFFINE):
            return -1
        name = f"{net.comp_names[c1]}.{net.comp_names[c2]}"
        i = comp_index(name)
        if i >= 0:
            return i
        in1, out1 = _affine_dims(net.comps[c1])
        if in1 > out1:
            return -1
        in2, out2 = _affine_dims(net.comps[c2])
        assert in2 % out1 == 0
        rng.rand()         # AffineComponent::Init (nnet-simple-component.cc): linear_params_.SetRandn() -> RandomState()
        rng.rand()         #                                                   bias_params_.SetRandn()   -> RandomState()
        mult = in2 // out1
        return add(name, kf.Comp("AffineComponent", {"<LinearParams>": np.zeros((out2, mult * in1), np.float32)}))

    def scale(c1: int, c2: int) -> int:                       # CollapseComponentsScale (:1872-1906)
        if net.ctype(c1) not in _AFFINE or net.ctype(c2) != "FixedScaleComponent":
            return -1
        _, out1 = _affine_dims(net.comps[c1])
        if out1 != len(np.asarray(net.comps[c2].fields["<Scales>"]).reshape(-1)):
            return -1
        name = f"{net.comp_names[c1]}.{net.comp_names[c2]}"
        i = comp_index(name)
        if i >= 0:
            return i
        return add(name, net.comps[c1])

    def collapse_components(c1: int, c2: int) -> int:
        # (:1505-1527) with the CollapseModelConfig() both binaries pass: collapse_dropout = collapse_batchnorm = false,
        # collapse_affine = collapse_scale = true (nnet-utils.h:240-249); the other two are restated for completeness
        enabled = (affine, scale) if not all_four else (dropout, batchnorm, affine, scale)
        for fn in enabled:
            ans = fn(c1, c2)
            if ans != -1:
                return ans
        return -1

    def optimize_node(i: int) -> bool:                        # OptimizeNode (:1640-1685)
        nodes = net.nodes
        if nodes[i].kind != "descriptor" or i + 1 >= len(nodes) or nodes[i + 1].kind != "component":
            return False
        src = nodes[i].desc.collapsible_node()
        if src == -1 or nodes[src].kind != "component":
            return False
        combined = collapse_components(nodes[src].comp, nodes[i + 1].comp)
        if combined == -1:
            return False
        nodes[i + 1].comp = combined
        # ReplaceNodeInDescriptor (:1571-1601): textual substitution of the bypassed node by its own input, re-parsed
        names = net.names()
        fake = list(names)
        fake[src] = nodes[src - 1].desc.text(names)
        nodes[i].desc = Descriptor(nodes[i].desc.text(fake), names)
        return True

    changed, iters = True, 0
    while changed:
        changed = False
        for i in range(len(net.nodes)):
            if optimize_node(i):
                changed = True
        iters += 1
        assert iters <= 11


# =========================================================================================== computation graph builder

K_UNKNOWN, K_COMPUTABLE, K_NOT = 0, 1, 2


class GraphBuilder:
    """ComputationGraph + ComputationGraphBuilder (nnet-computation-graph.cc), everything that decides how many cindexes
    exist after each BuildGraphOneIter round and how many rounds there are."""

    def __init__(self, net: Net, rng: GlibcRand):
        self.net, self.rng = net, rng
        self.cindexes: List[Cindex] = []
        self.ids: Dict[Cindex, int] = {}
        self.is_input: List[bool] = []
        self.deps: List[List[int]] = []
        self.segment_ends: List[int] = []
        self.computable: List[int] = []
        self.usable: List[int] = []
        self.queued: List[bool] = []
        self.deps_done: List[bool] = []
        self.depend_on_this: List[List[int]] = []
        self.cur: List[int] = []
        self.nxt: List[int] = []
        self.distance = -1
        self.is_out_node = [net.is_output(i) for i in range(len(net.nodes))]

    # -- graph
    def _get(self, c: Cindex, is_input: bool) -> Tuple[int, bool]:
        i = self.ids.get(c)
        if i is not None:
            return i, False
        i = len(self.cindexes)
        self.ids[c] = i
        self.cindexes.append(c)
        self.is_input.append(is_input)
        self.deps.append([])
        return i, True

    def _add_info(self):
        self.depend_on_this.append([])
        self.computable.append(K_UNKNOWN)
        self.usable.append(0)
        self.queued.append(False)
        self.deps_done.append(False)

    # -- Compute (:462-493)
    def compute(self, inputs: List[Tuple[str, List[Index]]], outputs: List[Tuple[str, List[Index]]]):
        start = len(self.cindexes)
        for name, idx in inputs:                                   # AddInputs (:262-285)
            n = self.net.index(name)
            for ind in idx:
                i, new = self._get((n, ind), True)
                assert new
                self._add_info()
                self.computable[-1] = K_COMPUTABLE
        for name, idx in outputs:                                  # AddOutputs (:287-314)
            n = self.net.index(name)
            for ind in idx:
                i, new = self._get((n, ind), False)
                assert new
                self._add_info()
                self.usable[-1] = 1
                self.queued[-1] = True
                self.nxt.append(i)
        self.distance = 0
        assert not self.cur
        self.cur, self.nxt = self.nxt, self.cur
        while self.distance < 10000:
            self._one_iter()
            if self.rng.rand_int(1, self.distance + 1) == 1:
                self.check(start)
            if not self.cur:
                break
        if self.rng.rand_int(1, 2 * (len(self.segment_ends) + 1)) == 1:
            self.check(start)

    def check(self, start: int):                                   # (:496-570): only the draws
        num = len(self.cindexes)
        i = start
        while i < num:
            self.rng.rand_int(0, i)
            i += 1 + self.rng.rand_int(0, num // 100)

    def _one_iter(self):                                           # BuildGraphOneIter (:893-913)
        cur = self.cur
        while cur:
            i = cur.pop()
            self.queued[i] = False
            if not self.deps_done[i] and self.usable[i] != 0:
                self.deps_done[i] = True
                self._add_deps(i)
                if not self.queued[i]:
                    self.queued[i] = True
                    self.nxt.append(i)
            elif self.computable[i] == K_UNKNOWN:
                self._update_computable(i)
        self.cur, self.nxt = self.nxt, self.cur
        self.distance += 1

    def _input_cindexes(self, i: int) -> List[Cindex]:
        node_i, ind = self.cindexes[i]
        node = self.net.nodes[node_i]
        if node.kind == "descriptor":
            return node.desc.deps(ind)
        if node.kind == "component":
            offs = self.net.time_offsets(node.comp)
            if offs is None:
                return [(node_i - 1, ind)]
            return [(node_i - 1, (ind[0], ind[1] + o, ind[2])) for o in offs]
        if node.kind == "dimrange":
            return [(node.src, ind)]
        return []

    def _add_deps(self, i: int):                                   # AddDependencies (:624-718)
        this = []
        for c in self._input_cindexes(i):
            d, new = self._get(c, False)
            this.append(d)
            if new:
                self._add_info()
                self.queued[-1] = True
                self.nxt.append(d)
        this = sorted(set(this))
        self.deps[i] = this
        for d in this:
            self.depend_on_this[d].append(i)
            self._inc_usable(d)

    def _inc_usable(self, i: int):                                 # (:856-874)
        self.usable[i] += 1
        if self.usable[i] == 1 and self.computable[i] != K_NOT:
            for d in self.deps[i]:
                self._inc_usable(d)
            if self.computable[i] == K_UNKNOWN and not self.queued[i]:
                self.queued[i] = True
                self.nxt.append(i)

    def _dec_usable(self, i: int):                                 # (:877-890)
        self.usable[i] -= 1
        if self.usable[i] == 0 and self.computable[i] != K_NOT:
            for d in self.deps[i]:
                self._dec_usable(d)

    def _cset(self, treat_unknown: bool):
        ids, comp = self.ids, self.computable

        def f(c: Cindex) -> bool:
            i = ids.get(c)
            if i is None:
                return False
            v = comp[i]
            return v == K_COMPUTABLE or (treat_unknown and v == K_UNKNOWN)
        return f

    def _component_computable(self, node_i: int, ind: Index, cset, used: Optional[List[Cindex]]) -> bool:
        offs = self.net.time_offsets(self.net.nodes[node_i].comp)
        want = [(node_i - 1, ind)] if offs is None else [(node_i - 1, (ind[0], ind[1] + o, ind[2])) for o in offs]
        if used is not None:
            used.clear()
        for c in want:
            if cset(c):
                if used is not None:
                    used.append(c)
            else:
                return False
        return True

    def _compute_computable(self, i: int) -> int:                  # ComputeComputableInfo (:721-785)
        node_i, ind = self.cindexes[i]
        node = self.net.nodes[node_i]
        if node.kind == "descriptor":
            if node.desc.computable(ind, self._cset(False), None):
                return K_COMPUTABLE
            if not node.desc.computable(ind, self._cset(True), None):
                return K_NOT
            return K_UNKNOWN
        if node.kind == "component":
            if self._component_computable(node_i, ind, self._cset(False), None):
                return K_COMPUTABLE
            if not self._component_computable(node_i, ind, self._cset(True), None):
                return K_NOT
            return K_UNKNOWN
        if node.kind == "dimrange":
            j = self.ids.get((node.src, ind))
            return self.computable[j] if j is not None else K_UNKNOWN
        return K_COMPUTABLE if self.is_input[i] else K_NOT

    def _update_computable(self, i: int):                          # UpdateComputableInfo (:813-853)
        if self.usable[i] == 0:
            return
        assert self.computable[i] == K_UNKNOWN
        out = self._compute_computable(i)
        self.computable[i] = out
        if out != K_UNKNOWN:
            for o in self.depend_on_this[i]:
                if self.computable[o] == K_UNKNOWN and not self.queued[o]:
                    self.queued[o] = True
                    self.nxt.append(o)
            if out == K_NOT and self.usable[i] != 0:
                for d in self.deps[i]:
                    self._dec_usable(d)

    def output_computable(self, name: str, idx: List[Index]) -> List[bool]:   # GetComputableInfo (:787-810)
        n = self.net.index(name)
        return [self.computable[self.ids[(n, ind)]] == K_COMPUTABLE for ind in idx]

    # -- Prune (:572-622)
    def prune(self):
        start = self.segment_ends[-1] if self.segment_ends else 0
        num = len(self.cindexes)
        for i in range(start, num):
            self._prune_deps(i)
        required = [False] * (num - start)                         # ComputeRequiredArray (:916-958)
        queue = []
        for c in range(start, num):
            if self.is_out_node[self.cindexes[c][0]]:
                required[c - start] = True
                queue.append(c)
        while queue:
            c = queue.pop()
            for d in self.deps[c]:
                if d >= start and not required[d - start]:
                    required[d - start] = True
                    queue.append(d)
        keep = [required[c - start] or self.is_input[c] for c in range(start, num)]
        for c in range(start, num):
            if keep[c - start]:
                assert self.computable[c] == K_COMPUTABLE, "Prune when not everything is computable"
        # Renumber (:53-122)
        old2new, new2old = {}, []
        for j, k in enumerate(keep):
            if k:
                old2new[j + start] = len(new2old) + start
                new2old.append(j + start)
        if len(new2old) != num - start:
            for old in range(start, num):
                if old not in old2new:
                    del self.ids[self.cindexes[old]]
            newc, newi, newd = [], [], []
            for old in new2old:
                c = self.cindexes[old]
                self.ids[c] = old2new[old]
                newc.append(c)
                newi.append(self.is_input[old])
                newd.append([d if d < start else old2new[d] for d in self.deps[old]])
            self.cindexes[start:] = newc
            self.is_input[start:] = newi
            self.deps[start:] = newd
        n2 = len(self.cindexes)
        self.computable[start:] = [K_COMPUTABLE] * (n2 - start)
        self.usable[start:] = [1] * (n2 - start)
        self.queued[start:] = [False] * (n2 - start)
        self.deps_done[start:] = [False] * (n2 - start)
        self.depend_on_this[start:] = [[] for _ in range(n2 - start)]
        self.segment_ends.append(n2)

    def _prune_deps(self, i: int):                                 # PruneDependencies (:352-447)
        if self.computable[i] == K_NOT or self.usable[i] == 0:
            self.deps[i] = []
            return
        assert self.computable[i] == K_COMPUTABLE
        node_i, ind = self.cindexes[i]
        node = self.net.nodes[node_i]
        if node.kind in ("dimrange", "input"):
            return
        used: List[Cindex] = []
        if node.kind == "descriptor":
            ok = node.desc.computable(ind, self._cset(False), used)
        else:
            ok = self._component_computable(node_i, ind, self._cset(False), used)
        assert ok
        self.deps[i] = sorted(set(self.ids[c] for c in used))


# =========================================================================================== the callers

def compute_simple_nnet_context(net: Net, rng: GlibcRand) -> Tuple[int, int]:
    """ComputeSimpleNnetContext (nnet-utils.cc:146-197)."""
    modulus = net.modulus()
    has_ivector = net.index("ivector") != -1
    window = 40
    while window < 800:
        lefts, rights, ok = [], [], True
        for start in range(modulus + 1):
            n = rng.rand() % 10                                    # (:107)
            idx = [(n, t, 0) for t in range(start, start + window)]
            inputs = [("input", idx)]
            if has_ivector:
                inputs.append(("ivector", [(n, t, 0) for t in range(start - modulus, start + window)]))
            b = GraphBuilder(net, rng)
            b.compute(inputs, [("output", idx)])                   # EvaluateComputationRequest (:64-84)
            okv = b.output_computable("output", idx)
            first_ok = okv.index(True) if True in okv else window
            first_not = first_ok + (okv[first_ok:].index(False) if False in okv[first_ok:] else window - first_ok)
            if first_ok == window or first_not <= first_ok:
                ok = False
                break
            lefts.append(first_ok)
            rights.append(window - first_not)
        if not ok:
            window *= 2
            continue
        return max(lefts), max(rights)
    raise ValueError("ComputeSimpleNnetContext failed")


def modify_ivector_period(net: Net, period: int) -> None:
    """ModifyNnetIvectorPeriod (nnet-compile-looped.cc:28-79): ReplaceIndex(<desc>, t, 0) -> Round(<desc>, period)."""
    names = net.names()
    for i, n in enumerate(net.nodes):
        if n.kind == "descriptor" and i + 1 < len(net.nodes) and net.nodes[i + 1].kind == "component":
            text = n.desc.text(names)
            pos = text.find("ReplaceIndex(")
            if pos >= 0:
                comma = text.find(", t, 0)", pos)
                if comma < 0:
                    raise ValueError("could not process the ReplaceIndex expression in " + text)
                inner = text[pos + len("ReplaceIndex("):comma]
                text = text[:pos] + f"Round({inner}, {period})" + text[comma + 7:]
                n.desc = Descriptor(text, names)


def looped_requests(net: Net, chunk: int, left: int, right: int, num_requests: int, fsf: int = 1):
    """CreateLoopedComputationRequest + the extrapolated requests of CompileLoopedInternal (nnet-compile-looped.cc:131-300);
    one sequence, outputs at the multiples of the frame-subsampling-factor (CreateComputationRequestInternal :111-128), ivector
    period = chunk (decodable-simple-looped.cc:68-70)."""
    has_ivector = net.index("ivector") != -1
    reqs = []
    seen: set = set()
    prev_times: List[int] = []
    for k in range(num_requests):
        in0 = -left if k == 0 else chunk + right + (k - 1) * chunk
        in1 = chunk + right if k == 0 else in0 + chunk
        inputs = [("input", [(0, t, 0) for t in range(in0, in1)])]
        if has_ivector:
            if k < 3:
                # (:158-180) the iVector at t rounded down to the period, where no earlier chunk asked for it already
                times = sorted({t - t % chunk for t in range(in0, in1)} - seen)
                seen.update(times)
            else:
                # ExtrapolateComputationRequest (:246-270): request k-1 shifted by the offset between k-2 and k-1
                times = [t + chunk for t in prev_times]
            prev_times = times
            if times:
                inputs.append(("ivector", [(0, t, 0) for t in times]))
        reqs.append((inputs, [("output", [(0, t, 0) for t in range(k * chunk, (k + 1) * chunk, fsf)])]))
    return reqs


def setup_rand_calls(nf: kf.NnetFile, frames_per_chunk: int = 24, extra_left_context_initial: int = 0, frame_subsampling_factor: int = 1) -> int:
    """Number of rand() calls both decoder binaries make before the first feature frame
    (online2-wav-nnet3-latgen-faster.cc:160-176, online2-cli-nnet3-decode-faster.cc:97-111)."""
    rng = GlibcRand()
    net = Net(nf)
    compute_simple_nnet_context(net, rng)                          # AmNnetSimple::Read -> SetContext (am-nnet-simple.cc:48,80-87)
    collapse_model(net, rng)                                       # CollapseModel(CollapseModelConfig(), &nnet)
    left, right = compute_simple_nnet_context(net, rng)            # DecodableNnetSimpleLoopedInfo::Init (:55-62)
    left += extra_left_context_initial
    modulus = net.modulus()
    chunk = frames_per_chunk
    while chunk % modulus != 0 or chunk % frame_subsampling_factor != 0:      # GetChunkSize (nnet-compile-looped.cc:82-96)
        chunk += 1
    if net.index("ivector") != -1:
        modify_ivector_period(net, chunk)
    num_requests = 5                                               # CompileLooped (:326-345): 5, then 10, 20, ... on failure
    b = GraphBuilder(net, rng)
    for inputs, outputs in looped_requests(net, chunk, left, right, num_requests, frame_subsampling_factor):   # Compiler::CreateComputation (nnet-compile.cc:50-62)
        b.compute(inputs, outputs)
        b.prune()
    # Compiler::SetUpPrecomputedIndexes (nnet-compile.cc:1239-1291): one step per (node, segment) in a feed-forward network;
    # TdnnComponent::PrecomputeIndexes draws once per step (nnet-tdnn-component.cc:553)
    begin = 0
    for end in b.segment_ends:
        tdnn_nodes = {c[0] for c in b.cindexes[begin:end]
                      if net.nodes[c[0]].kind == "component" and net.ctype(net.nodes[c[0]].comp) == "TdnnComponent"}
        for _ in tdnn_nodes:
            rng.rand()
        begin = end
    if left + right > 0:
        # Optimize -> SplitRowOps -> InsertCommands (nnet-optimize-utils.cc:2883-2892,4654): rows a chunk reads from the
        # chunk before it make two-piece multi-row commands, which are split, which draws once
        rng.rand()
    return rng.pos
