"""Test helpers for the lattice wire format: a reader of one binary CompactLattice table entry written from the format's
description (kaldi-lattice.cc:478-500, OpenFst VectorFst layout), and a wrapper that lists a lattice's paths with the
REFERENCE's own tools (oracle/_ref -- test infrastructure, built by oracle/build_ref.sh)."""
import os
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
BIN = REPO / "oracle" / "_ref" / "bin"


def read_compact_lattice(entry: bytes, key: str):
    """Parser of one binary table entry: returns (start, finals{state: (g, a, tids)}, arcs[state] = [(label, g, a, tids, dst)])."""
    head = key.encode() + b" "
    assert entry.startswith(head)
    pos = len(head)

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, entry, pos)
        pos += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def take_str():
        nonlocal pos
        n = take("i")
        s = entry[pos:pos + n]
        pos += n
        return s

    def take_weight():
        g, a = take("ff")
        n = take("i")
        tids = list(take(f"{n}i")) if n > 1 else ([take("i")] if n == 1 else [])
        return g, a, tids

    assert take("i") == 2125659606                       # OpenFst magic
    assert take_str() == b"vector"
    assert take_str() == b"compactlattice44"
    version, flags = take("ii")
    assert version == 2 and flags == 0
    take("Q")
    start, n_states, n_arcs = take("qqq")
    finals, arcs, seen = {}, [], 0
    for s in range(n_states):
        g, a, tids = take_weight()
        if np.isfinite(g):
            finals[s] = (g, a, tids)
        row = []
        for _ in range(take("q")):
            il, ol = take("ii")
            assert il == ol                              # acceptor
            g, a, tids = take_weight()
            row.append((il, g, a, tids, take("i")))
        seen += len(row)
        arcs.append(row)
    assert pos == len(entry) and seen == n_arcs
    return start, finals, arcs


def reference_paths(ark: Path, n: int, tmp: Path):
    if not (BIN / "lattice-to-nbest").exists():
        pytest.fail("oracle/_ref is not built (bash oracle/build_ref.sh in the build container; it travels with the snapshot)")
    env = dict(os.environ, PATH=f"{BIN}:{os.environ['PATH']}")
    sh = (f"lattice-to-nbest --n={n} --acoustic-scale=1.0 ark:{ark} ark:- | "
          f"nbest-to-linear ark:- ark,t:{tmp}/ali ark,t:{tmp}/words ark,t:{tmp}/lm ark,t:{tmp}/ac")
    p = subprocess.run(["bash", "-c", sh], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]

    def table(fn, conv):
        t = {}
        for line in (tmp / fn).read_text().splitlines():
            q = line.split()
            if q:
                t[q[0]] = [conv(x) for x in q[1:]]
        return t
    ali, words, lm, ac = table("ali", int), table("words", int), table("lm", float), table("ac", float)
    keys = sorted(words, key=lambda k: (k.rsplit("-", 1)[0], int(k.rsplit("-", 1)[1])))
    return [dict(words=words[k], ali=ali[k], graph=lm[k][0], acoustic=ac[k][0]) for k in keys]
