"""TEST INFRASTRUCTURE ONLY -- CPU oracle, part 2: numpy restatement of the transcribe hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product
(rhasspy_speech_amd + librhasspy_speech_hip.so) never does.  It restates, independently of the product's C++/HIP
code, what the reference computes between a waveform and the n-best word-id lists (citations: kaldi/src):

  MFCC            feat/feature-window.cc:90-224, feature-mfcc.cc:28-80, mel-computations.cc:33-142,226-259
  online CMVN     feat/online-feature.cc:337-452, transform/cmvn.cc:64-91
  splice / LDA    feat/online-feature.cc:504-554
  UBM posteriors  gmm/diag-gmm.cc:546-562, hmm/posterior.cc:440-509
  iVector         ivector/ivector-extractor.cc:182-218,611-668,732-756, matrix/optimization.cc:453-566,
                  online2/online-ivector-feature.cc:201-355 (offline: greedy, one estimate from all frames;
                  streaming: one estimate per nnet chunk from the frames available at that 1024-sample tick)
  nnet3 forward   nnet3/decodable-online-looped.cc:118-236 (+ the component Propagates), evaluated on the whole
                  edge-padded utterance: output[t] is a pure function of x[clamp(t-L .. t+R)]
  beam search     oracle/decoder.c (sequential, hash-order faithful restatement of lattice-faster-decoder.cc)
  n-best          oracle/lattice.py (determinised-lattice n-best semantics of lattice-to-nbest | nbest-to-linear)

Parity is PINNED: tests/test_oracle_golden.py checks every stage of this file against vectors produced by the
reference's own binaries (tests/golden/*.npz, written by oracle/gen_golden.py from oracle/_ref).
"""
from __future__ import annotations

import ctypes as C
import math
import re
import subprocess
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import kaldi_formats as kf
from . import lattice as lat

F32 = np.float32
HERE = Path(__file__).resolve().parent


# =========================================================================================== MFCC

@dataclass
class MfccOpts:
    samp_freq: float = 16000.0
    frame_shift_ms: float = 10.0
    frame_length_ms: float = 25.0
    dither: float = 1.0
    preemph: float = 0.97
    remove_dc: bool = True
    window_type: str = "povey"
    num_bins: int = 23
    low_freq: float = 20.0
    high_freq: float = 0.0
    num_ceps: int = 13
    use_energy: bool = True
    cepstral_lifter: float = 22.0

    @classmethod
    def from_conf(cls, path) -> "MfccOpts":
        o = cls()
        names = {"sample-frequency": "samp_freq", "frame-shift": "frame_shift_ms", "frame-length": "frame_length_ms",
                 "dither": "dither", "preemphasis-coefficient": "preemph", "num-mel-bins": "num_bins", "low-freq": "low_freq",
                 "high-freq": "high_freq", "num-ceps": "num_ceps", "cepstral-lifter": "cepstral_lifter"}
        for k, v in kf.read_config(path):
            if k in names:
                setattr(o, names[k], type(getattr(o, names[k]))(float(v)))
            elif k == "use-energy":
                o.use_energy = v == "true"
            elif k == "remove-dc-offset":
                o.remove_dc = v == "true"
            elif k == "window-type":
                o.window_type = v
            else:
                raise ValueError(f"oracle: unsupported mfcc option --{k}")
        return o


def _libm_f32(name: str):
    """cosf / sinf of the C library: the reference's tables are std::cos(float) / std::sin(float) values, and numpy's
    vectorised float32 trig functions differ from libm in the last bit for some arguments."""
    import ctypes
    import ctypes.util
    fn = getattr(ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6"), name)
    fn.restype, fn.argtypes = ctypes.c_float, [ctypes.c_float]
    return lambda x: F32(fn(float(x)))


class SplitRadixRealFft:
    """matrix/srfft.cc restated for a batch of frames: the in-place float32 split-radix complex FFT of N/2 points
    (ComputeRecursive :212-355, tables :78-115, BitReversePermute :185-209) and the real-FFT post-processing whose
    twiddle is advanced by a float32 recurrence (:360-417).  The recursion is unrolled into levels of independent
    butterflies (blocks of one recursion depth touch disjoint points); every butterfly performs the reference's float32
    operations on the same operands, numpy just applies it to all frames at once.  Output: the power spectrum of
    feature-functions.cc:29-51 (bins 0 .. N/2)."""

    def __init__(self, n_real: int):
        self.NR = n_real
        self.N = N = n_real // 2
        logn = N.bit_length() - 1
        two_pi = 6.283185307179586476925286766559005
        cosf, sinf = _libm_f32("cosf"), _libm_f32("sinf")
        levels = []
        cur = [(0, logn)]
        while cur:
            nxt, k0, k1, k2 = [], [], [], []
            for off, lg in cur:
                if lg >= 3:
                    m = 1 << lg
                    m2, m4, m8 = m // 2, m // 4, m // 8
                    for n in range(m4):
                        if n == 0:
                            tw, mode = [0.0] * 6, 0
                        elif n == m8:
                            tw, mode = [0.0] * 6, 1
                        else:
                            a = F32(n * two_pi / m)
                            c, sn = cosf(a), sinf(a)
                            a3 = F32(3 * n * two_pi / m)
                            c3, s3 = cosf(a3), sinf(a3)
                            tw, mode = [c, -(sn + c), sn - c, c3, -(s3 + c3), s3 - c3], 2
                        k0.append((off + n, off + n + m4, off + n + m2, off + n + m2 + m4, mode, tw))
                    nxt += [(off, lg - 1), (off + m2, lg - 2), (off + 3 * (m // 4), lg - 2)]
                elif lg == 2:
                    k1.append(off)
                elif lg == 1:
                    k2.append(off)
            if k0 or k1 or k2:
                e = np.array([t[:4] for t in k0], np.int64).reshape(-1, 4)
                mode = np.array([t[4] for t in k0], np.int64)
                tw = np.array([t[5] for t in k0], F32).reshape(-1, 6)
                levels.append((e, mode, tw, np.array(k1, np.int64), np.array(k2, np.int64)))
            cur = nxt
        self.levels = levels
        # bit-reversal pass as a gather
        lg2 = logn >> 1
        nn = 1 << lg2
        if logn & 1:
            lg2 += 1
        seed = [0] * (1 << lg2)
        if lg2 >= 1:
            seed[1] = 1
        for j in range(2, lg2 + 1):
            imax = 1 << (j - 1)
            for i in range(imax):
                seed[i] <<= 1
                seed[i + imax] = seed[i] + 1
        perm = list(range(N))
        if logn > 1:
            for off in range(1, nn):
                fj = nn * seed[off]
                perm[off], perm[fj] = perm[fj], perm[off]
                pp = off
                for gno in range(1, seed[off]):
                    pp += nn
                    j = fj + seed[gno]
                    perm[pp], perm[j] = perm[j], perm[pp]
        self.perm = np.array(perm, np.int64)
        # exp(-2 pi i k / NR), advanced by float32 complex multiplications
        x = F32(two_pi / n_real * -1)
        root_re, root_im = cosf(x), sinf(x)
        k_re, k_im = F32(1.0), F32(0.0)
        kn = [(k_re, k_im)]
        for _ in range(1, N // 2 + 1):
            t_re = F32(F32(k_re * root_re) - F32(k_im * root_im))
            k_im = F32(F32(k_re * root_im) + F32(k_im * root_re))
            k_re = t_re
            kn.append((k_re, k_im))
        self.kn = np.array(kn, F32)

    def power_spectrum(self, frames: np.ndarray) -> np.ndarray:
        """frames: T x NR float32 (windowed, zero padded).  Returns T x (NR/2 + 1) float32."""
        assert frames.dtype == F32
        xr = np.ascontiguousarray(frames[:, 0::2])
        xi = np.ascontiguousarray(frames[:, 1::2])
        sq = F32(0.70710678118654752440)
        for e, mode, tw, k1, k2 in self.levels:
            if len(e):
                e0, e1, e2, e3 = e[:, 0], e[:, 1], e[:, 2], e[:, 3]
                ar, ai, br, bi = xr[:, e0], xi[:, e0], xr[:, e1], xi[:, e1]
                cr, ci, dr, di = xr[:, e2], xi[:, e2], xr[:, e3], xi[:, e3]
                xr[:, e0], xi[:, e0] = ar + cr, ai + ci
                xr[:, e1], xi[:, e1] = br + dr, bi + di
                p_r, p_i, q_r, q_i = ar - cr, ai - ci, br - dr, bi - di
                r1, i2, i1, r2 = p_r + q_i, p_i + q_r, p_i - q_r, p_r - q_i
                # n == m/8
                s_r1, s_i1 = sq * (r1 + i1), sq * (i1 - r1)
                s_r2, s_i2 = sq * (i2 - r2), -sq * (r2 + i2)
                # general twiddle
                cn, spcn, smcn, c3n, spc3n, smc3n = (tw[None, :, j] for j in range(6))
                t2 = cn * (r1 + i1)
                g_i1 = spcn * r1 + t2
                g_r1 = smcn * i1 + t2
                t2 = c3n * (r2 + i2)
                g_i2 = spc3n * r2 + t2
                g_r2 = smc3n * i2 + t2
                m1, m2 = (mode == 1)[None, :], (mode == 2)[None, :]
                xr[:, e2] = np.where(m2, g_r1, np.where(m1, s_r1, r1))
                xi[:, e2] = np.where(m2, g_i1, np.where(m1, s_i1, i1))
                xr[:, e3] = np.where(m2, g_r2, np.where(m1, s_r2, r2))
                xi[:, e3] = np.where(m2, g_i2, np.where(m1, s_i2, i2))
            if len(k1):
                r0, r1, r2, r3 = (xr[:, k1 + j] for j in range(4))
                i0, i1, i2, i3 = (xi[:, k1 + j] for j in range(4))
                r0, r2 = r0 + r2, r0 - r2
                i0, i2 = i0 + i2, i0 - i2
                r1, r3 = r1 + r3, r1 - r3
                i1, i3 = i1 + i3, i1 - i3
                r0, r1 = r0 + r1, r0 - r1
                i0, i1 = i0 + i1, i0 - i1
                t1, t2 = r2 + i3, i2 + r3
                i2 = i2 - r3
                r3 = r2 - i3
                r2, i3 = t1, t2
                for j, (rv, iv) in enumerate(((r0, i0), (r1, i1), (r2, i2), (r3, i3))):
                    xr[:, k1 + j], xi[:, k1 + j] = rv, iv
            if len(k2):
                r0, r1, i0, i1 = xr[:, k2], xr[:, k2 + 1], xi[:, k2], xi[:, k2 + 1]
                xr[:, k2], xr[:, k2 + 1] = r0 + r1, r0 - r1
                xi[:, k2], xi[:, k2 + 1] = i0 + i1, i0 - i1
        xr, xi = xr[:, self.perm], xi[:, self.perm]
        N = self.N
        power = np.zeros((frames.shape[0], N + 1), F32)
        k = np.arange(1, N // 2 + 1)
        kd = N - k
        kn_re, kn_im = self.kn[k, 0][None, :], self.kn[k, 1][None, :]
        half = F32(0.5)
        ck_re, ck_im = half * (xr[:, k] + xr[:, kd]), half * (xi[:, k] - xi[:, kd])
        dk_re, dk_im = half * (xi[:, k] + xi[:, kd]), -half * (xr[:, k] - xr[:, kd])
        a_re = ck_re + (kn_re * dk_re - kn_im * dk_im)
        a_im = ck_im + (kn_re * dk_im + kn_im * dk_re)
        b_re = ck_re + ((-kn_re) * dk_re - kn_im * (-dk_im))
        b_im = -ck_im + ((-kn_re) * (-dk_im) + kn_im * dk_re)
        power[:, kd] = b_re * b_re + b_im * b_im          # k' = N/2 - k first: for k == k' the A_k value below wins
        power[:, k] = a_re * a_re + a_im * a_im
        zeroth, n2th = xr[:, 0] + xi[:, 0], xr[:, 0] - xi[:, 0]
        power[:, 0] = zeroth * zeroth
        power[:, N] = n2th * n2th
        return power


def frame_sum(fr: np.ndarray) -> np.ndarray:
    """Row sums of a float32 matrix in the order of cblas_sdot(n, x, 1, &one, 0) of OpenBLAS 0.3.x on x86-64."""
    n = fr.shape[1]
    pairs = (fr[:, 0:n - (n & 1):2] + fr[:, 1::2]).astype(F32).astype(np.float64)
    if n & 1:
        pairs = np.concatenate([pairs, fr[:, -1:].astype(np.float64)], axis=1)
    return np.add.accumulate(pairs, axis=1)[:, -1].astype(F32)


_dither_cache: dict = {}


def dither_table(T: int, win: int, offset: int = 0) -> np.ndarray:
    """[T, win] float32: RandGauss values Dither() draws for frame t, sample i in a reference process whose set-up called rand()
    `offset` times (oracle/dither.c)."""
    have = _dither_cache.get((win, offset))
    if have is None or have.shape[0] < T:
        n = max(T, 512, 2 * (have.shape[0] if have is not None else 0))
        have = np.zeros((n, win), F32)
        lib = decoder_lib()
        lib.oracle_dither_table.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.oracle_dither_table.restype = None
        lib.oracle_dither_table(offset, 0, n, win, have.ctypes.data_as(C.c_void_p))
        _dither_cache[(win, offset)] = have
    return have[:T]


class Mfcc:
    def __init__(self, o: MfccOpts, rand_offset: int = 0):
        assert not o.use_energy and o.window_type == "povey"
        self.o = o
        self.rand_offset = rand_offset
        self.win = int(o.samp_freq * 0.001 * o.frame_length_ms)
        self.shift = int(o.samp_freq * 0.001 * o.frame_shift_ms)
        self.padded = 1 << (self.win - 1).bit_length()
        i = np.arange(self.win, dtype=np.float64)
        self.window = np.power(0.5 - 0.5 * np.cos(2.0 * math.pi / (self.win - 1) * i), 0.85).astype(F32)
        # mel banks, float32 arithmetic like the reference
        nfft = self.padded // 2
        nyq = F32(0.5) * F32(o.samp_freq)
        high = F32(o.high_freq) if o.high_freq > 0 else nyq + F32(o.high_freq)
        # MelScale (mel-computations.h:82-84) with the C library's logf: numpy's vectorised float32 log differs from it in
        # the last bit for some arguments, which moves a filter edge by one ulp -- visible (1e-3 on the cepstra) whenever a
        # strong spectral line sits on that edge
        logf = _libm_f32("logf")
        mel = lambda f: (F32(1127.0) * np.array([logf(v) for v in np.atleast_1d(F32(1.0) + np.asarray(f, F32) / F32(700.0))], F32)).reshape(np.shape(f)) \
            if np.ndim(f) else F32(F32(1127.0) * logf(F32(1.0) + F32(f) / F32(700.0)))
        bw = F32(o.samp_freq) / F32(self.padded)
        ml, mh = mel(F32(o.low_freq)), mel(high)
        delta = F32((mh - ml) / F32(o.num_bins + 1))
        fmel = mel(bw * np.arange(nfft, dtype=F32))
        W = np.zeros((o.num_bins, nfft + 1), F32)
        for b in range(o.num_bins):
            left, center, right = F32(ml + F32(b) * delta), F32(ml + F32(b + 1) * delta), F32(ml + F32(b + 2) * delta)
            for k in range(nfft):
                m = fmel[k]
                if m > left and m < right:
                    W[b, k] = (m - left) / (center - left) if m <= center else (right - m) / (right - center)
        self.melW = W
        N = o.num_bins
        dct = np.zeros((N, N))
        dct[0, :] = math.sqrt(1.0 / N)
        n = np.arange(N)
        for k in range(1, N):
            dct[k, :] = math.sqrt(2.0 / N) * np.cos(math.pi / N * (n + 0.5) * k)
        self.dct = dct[:o.num_ceps].astype(F32)
        q = o.cepstral_lifter
        self.lifter = (1.0 + 0.5 * q * np.sin(math.pi * np.arange(o.num_ceps) / q)).astype(F32) if q != 0 else np.ones(o.num_ceps, F32)
        self.fft = SplitRadixRealFft(self.padded)

    def num_frames(self, n: int) -> int:
        return 0 if n < self.win else 1 + (n - self.win) // self.shift

    def compute(self, pcm: np.ndarray) -> np.ndarray:
        T = self.num_frames(len(pcm))
        if T == 0:
            return np.zeros((0, self.o.num_ceps), F32)
        x = pcm.astype(F32)
        idx = np.arange(T)[:, None] * self.shift + np.arange(self.win)[None, :]
        fr = x[idx]
        if self.o.dither != 0.0:
            # feature-window.cc:90-98 via ProcessWindow (:145-146): before DC removal; the noise of frame t is the same for every
            # utterance because the reference starts a fresh process (rand() at its default seed) per utterance -- oracle/dither.c
            fr = fr + dither_table(T, self.win, self.rand_offset) * F32(self.o.dither)
        if self.o.remove_dc:
            # window->Add(-window->Sum() / frame_length) (feature-window.cc:148-149).  VectorBase::Sum() is cblas_sdot against a
            # stride-0 one (kaldi-vector.cc), i.e. OpenBLAS's strided loop (kernel/x86_64/sdot.c, 0.3.x): adjacent pairs added in
            # float, the pair sums accumulated in a double, the result rounded to float.  Irrelevant while the samples were
            # integers; with dither a different rounding of the mean re-rounds every sample of a loud frame, which moves the
            # cepstra by 1e-3 wherever a mel bin sits in a spectral valley (pinned against the library itself in
            # tests/test_oracle_golden.py::test_frame_sum_is_the_blas_order)
            fr = fr - (frame_sum(fr) / F32(self.win))[:, None]
        pre = fr.copy()
        pre[:, 1:] = fr[:, 1:] - F32(self.o.preemph) * fr[:, :-1]
        pre[:, 0] = fr[:, 0] - F32(self.o.preemph) * fr[:, 0]
        pre *= self.window[None, :]
        pad = np.zeros((T, self.padded), F32)
        pad[:, :self.win] = pre
        power = self.fft.power_spectrum(pad)
        mel = power @ self.melW.T
        mel = np.log(np.maximum(mel, np.finfo(F32).eps)).astype(F32)
        return ((mel @ self.dct.T) * self.lifter[None, :]).astype(F32)


# =========================================================================================== iVector branch

def online_cmvn(feats: np.ndarray, global_stats: np.ndarray, cmn_window: int = 600, global_frames: int = 200) -> np.ndarray:
    T, C = feats.shape
    x = feats.astype(np.float64)
    cs = np.concatenate([np.zeros((1, C)), np.cumsum(x, 0)])
    t = np.arange(T)
    lo = np.maximum(0, t + 1 - cmn_window)
    s = cs[t + 1] - cs[lo]
    n = (t + 1 - lo).astype(np.float64)
    from_global = np.minimum(np.maximum(cmn_window - n, 0.0), float(global_frames))
    gcount = global_stats[0, C]
    a = from_global / gcount
    s = s + a[:, None] * global_stats[0, :C][None, :]
    n = n + a * gcount
    alpha = (-1.0 / n).astype(F32).astype(np.float64)
    offset = (alpha[:, None] * s).astype(F32)
    return (feats + offset).astype(F32)


def splice(feats: np.ndarray, left: int, right: int, T_ready: Optional[int] = None) -> np.ndarray:
    T = feats.shape[0] if T_ready is None else T_ready
    idx = np.clip(np.arange(T)[:, None] + np.arange(-left, right + 1)[None, :], 0, T - 1)
    return feats[idx].reshape(T, -1)


def lda_transform(sp: np.ndarray, lda: np.ndarray) -> np.ndarray:
    lda = lda.astype(F32)
    if lda.shape[1] == sp.shape[1] + 1:
        return (sp @ lda[:, :-1].T + lda[:, -1][None, :]).astype(F32)
    return (sp @ lda.T).astype(F32)


def ubm_posteriors(feats: np.ndarray, gmm: kf.DiagGmm, num_gselect: int, min_post: float, posterior_scale: float):
    ll = gmm.gconsts[None, :] + feats @ gmm.means_invvars.T
    ll = (ll + F32(-0.5) * ((feats * feats) @ gmm.inv_vars.T)).astype(F32)
    posts = []
    for row in ll:
        mx = row.max()
        cutoff = F32(mx + F32(math.log(min_post)))
        cand = np.nonzero(row > cutoff)[0]
        p = np.exp((row[cand] - mx).astype(np.float64)).astype(F32)
        order = np.argsort(-p, kind="stable")[:num_gselect]
        sel = [(int(cand[i]), F32(p[i])) for i in order]
        tot = F32(0.0)
        for _, w in sel:
            tot = F32(tot + w)
        thr = F32(min_post) * tot
        while len(sel) > 1 and sel[-1][1] < thr:
            tot = F32(tot - sel[-1][1])
            sel.pop()
        inv = F32(1.0 / float(tot))
        scale = F32(posterior_scale) * F32(1.0)
        posts.append([(g, F32(F32(w * inv) * scale)) for g, w in sel])
    return posts


class IvectorStats:
    """OnlineIvectorEstimationStats (ivector-extractor.cc:786-795, AccStats :611-668, GetIvector :732-756)."""

    def __init__(self, ie: kf.IvectorExtractorFile, max_count: float):
        self.ie = ie
        I = ie.M.shape[2]
        self.quad = np.eye(I)
        self.lin = np.zeros(I)
        self.lin[0] = ie.prior_offset
        self.num_frames = 0.0
        self.max_count = max_count
        self.sigma_inv_M = np.einsum("gde,gei->gdi", ie.sigma_inv, ie.M)
        self.U = np.einsum("gdi,gdj->gij", ie.M, self.sigma_inv_M)

    def acc(self, feats: np.ndarray, posts) -> None:
        G, D, I = self.ie.M.shape
        wf: Dict[int, np.ndarray] = {}
        tw: Dict[int, np.float32] = {}
        for t, post in enumerate(posts):
            for g, w in post:
                if g not in wf:
                    wf[g] = np.zeros(D)
                    tw[g] = F32(0.0)
                wf[g] += float(w) * feats[t].astype(np.float64)
                tw[g] = F32(tw[g] + w)
        tot = 0.0
        for g in wf:
            self.lin += self.sigma_inv_M[g].T @ wf[g]
            self.quad += float(tw[g]) * self.U[g]
            tot += float(tw[g])
        if self.max_count > 0:
            old, new = self.num_frames, self.num_frames + tot
            change = max(new, self.max_count) / self.max_count - max(old, self.max_count) / self.max_count
            if change != 0.0:
                self.lin[0] += self.ie.prior_offset * change
                self.quad[np.diag_indices_from(self.quad)] += change
        self.num_frames += tot

    def get_ivector(self, x: np.ndarray, num_cg_iters: int = 15) -> np.ndarray:
        if self.num_frames <= 0:
            out = np.zeros_like(x)
            out[0] = self.ie.prior_offset
            return out
        x = x.copy()
        if x[0] == 0.0:
            x[0] = self.ie.prior_offset
        return linear_cgd(self.quad, self.lin, x, num_cg_iters)


def linear_cgd(A: np.ndarray, b: np.ndarray, x: np.ndarray, max_iters: int) -> np.ndarray:
    """matrix/optimization.cc:453-566 (double)."""
    M = len(b)
    p = b - A @ x
    r = -p
    r_cur = float(r @ r)
    r_rec = r_cur
    rf = float(F32(0.01) * F32(0.01))
    k = 0
    while k < M + 5 and k != max_iters:
        Ap = A @ p
        alpha = -float(p @ r) / float(p @ Ap)
        x = x + alpha * p
        r = r + alpha * Ap
        r_next = float(r @ r)
        if r_next < rf * r_rec or r_next > (1.0 / rf) * r_rec:
            r = A @ x - b
            r_next = float(r @ r)
            r_rec = r_next
        if r_next <= np.finfo(np.float64).tiny:
            break
        beta = r_next / r_cur
        p = p * beta - r
        r_cur = r_next
        k += 1
    return x


# =========================================================================================== nnet3 forward

class Nnet3:
    def __init__(self, nf: kf.NnetFile):
        self.nf = nf
        self.nodes: Dict[str, Dict[str, str]] = {}
        for ln in nf.config:
            first, rest = ln.split(None, 1)
            if first == "component":
                continue
            kv = {}
            keys = [(m.start(1), m.end()) for m in re.finditer(r"(?:^|\s)([\w-]+)=", rest)]
            for i, (ks, ve) in enumerate(keys):
                end = keys[i + 1][0] if i + 1 < len(keys) else len(rest)
                kv[rest[ks:ve - 1]] = rest[ve:end].strip()
            kv["_type"] = first
            self.nodes[kv["name"]] = kv
        self.bn_cache: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
        # context bound: sum of all |offsets| appearing anywhere (upper bound of the true model context)
        offs = 0
        for kv in self.nodes.values():
            for m in re.finditer(r"Offset\([^,]+,\s*(-?\d+)", kv.get("input", "")):
                offs = max(offs, 0)
            desc = kv.get("input", "")
            o = [abs(int(m.group(1))) for m in re.finditer(r"Offset\([^()]*?,\s*(-?\d+)\s*[,)]", desc)]
            node_max = max(o) if o else 0
            if kv["_type"] == "component-node":
                c = nf.components[kv["component"]]
                if c.type == "TdnnComponent":
                    node_max += int(np.abs(c.fields["<TimeOffsets>"]).max())
            offs += node_max
        self.halo = offs + 2

    # ---- exact model context (ComputeSimpleNnetContext, nnet-utils.cc:146): how far the output at t looks
    def context(self) -> Tuple[int, int]:
        memo: Dict[str, Tuple[int, int]] = {}

        def desc_ctx(sd: str) -> Tuple[int, int]:
            sd = sd.strip()
            m = re.match(r"^(\w+)\((.*)\)$", sd, re.S)
            if m and m.group(1) in ("Append", "Sum", "Offset", "Scale", "ReplaceIndex", "IfDefined"):
                fn, args = m.group(1), self._split_args(m.group(2))
                if fn in ("Append", "Sum"):
                    cs = [desc_ctx(a) for a in args]
                    return max(c[0] for c in cs), max(c[1] for c in cs)
                if fn == "Offset":
                    l, r = desc_ctx(args[0])
                    o = int(args[1])
                    return l - o, r + o
                if fn == "Scale":
                    return desc_ctx(args[1])
                if fn == "ReplaceIndex":
                    return (-10 ** 6, -10 ** 6)      # constant over t
                return desc_ctx(args[0])
            return node_ctx(sd)

        def node_ctx(name: str) -> Tuple[int, int]:
            if name in memo:
                return memo[name]
            kv = self.nodes[name]
            if kv["_type"] == "input-node":
                out = (0, 0) if name == "input" else (-10 ** 6, -10 ** 6)
            elif kv["_type"] == "dim-range-node":
                out = node_ctx(kv["input-node"])
            else:
                l, r = desc_ctx(kv["input"])
                if kv["_type"] == "component-node":
                    c = self.nf.components[kv["component"]]
                    if c.type == "TdnnComponent":
                        offs = [int(x) for x in c.fields["<TimeOffsets>"]]
                        l, r = l - min(offs), r + max(offs)
                out = (l, r)
            memo[name] = out
            return out

        l, r = node_ctx("output")
        return max(l, 0), max(r, 0)

    # ---- descriptor evaluation on the padded time axis (rows = t in [-halo, T + halo))
    @staticmethod
    def matmul(x: np.ndarray, wt: np.ndarray) -> np.ndarray:
        """The layers' matrix products (FP32 BLAS, like the reference).  Tests that ask how far FP32 itself is from the exact
        result replace this with a float64 product."""
        return (x @ wt).astype(F32)

    def _shift(self, a: np.ndarray, o: int) -> np.ndarray:
        if o == 0:
            return a
        idx = np.clip(np.arange(a.shape[0]) + o, 0, a.shape[0] - 1)
        return a[idx]

    def _split_args(self, s: str) -> List[str]:
        out, depth, cur = [], 0, ""
        for ch in s:
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            if ch == "," and depth == 0:
                out.append(cur.strip())
                cur = ""
            else:
                cur += ch
        out.append(cur.strip())
        return out

    def _desc(self, s: str) -> np.ndarray:
        s = s.strip()
        m = re.match(r"^(\w+)\((.*)\)$", s, re.S)
        if m and m.group(1) in ("Append", "Sum", "Offset", "Scale", "ReplaceIndex", "IfDefined"):
            fn, args = m.group(1), self._split_args(m.group(2))
            if fn == "Append":
                return np.concatenate([self._desc(a) for a in args], axis=1)
            if fn == "Sum":
                out = self._desc(args[0])
                for a in args[1:]:
                    out = (out + self._desc(a)).astype(F32)
                return out
            if fn == "Offset":
                return self._shift(self._desc(args[0]), int(args[1]))
            if fn == "Scale":
                return (F32(float(args[0])) * self._desc(args[1])).astype(F32)
            if fn == "ReplaceIndex":
                assert args[1] == "t" and int(args[2]) == 0
                return self._desc(args[0])      # the iVector node is already constant over t
            return self._desc(args[0])
        return self._node(s)

    def _bn(self, name: str, c: kf.Comp):
        if name not in self.bn_cache:
            f = c.fields
            count, eps, rms = float(f["<Count>"]), F32(f["<Epsilon>"]), F32(f["<TargetRms>"])
            mean, var = np.asarray(f["<StatsMean>"], F32), np.asarray(f["<StatsVar>"], F32)
            sumsq = ((var + mean * mean).astype(F32) * F32(count)).astype(F32)
            ssum = (mean * F32(count)).astype(F32)
            off = (ssum * F32(-1.0 / count)).astype(F32)
            scl = (sumsq * F32(1.0 / count)).astype(F32)
            scl = (scl + F32(-1.0) * off * off).astype(F32)
            scl = np.maximum(scl, F32(0.0)) + eps
            scl = np.power(scl, F32(-0.5)).astype(F32) * rms
            off = (off * scl).astype(F32)
            dim = int(f["<Dim>"])
            reps = dim // len(scl)
            self.bn_cache[name] = (np.tile(scl, reps), np.tile(off, reps))
        return self.bn_cache[name]

    def _node(self, name: str) -> np.ndarray:
        if name in self.memo:
            return self.memo[name]
        kv = self.nodes[name]
        t = kv["_type"]
        if t == "output-node":
            out = self._desc(kv["input"])
        elif t == "dim-range-node":
            src = self._node(kv["input-node"])
            o, d = int(kv["dim-offset"]), int(kv["dim"])
            out = src[:, o:o + d]
        elif t == "component-node":
            c = self.nf.components[kv["component"]]
            x = self._desc(kv["input"])
            f = c.fields
            if c.type in ("AffineComponent", "NaturalGradientAffineComponent", "FixedAffineComponent"):
                out = (self.matmul(x, np.asarray(f["<LinearParams>"], F32).T) + np.asarray(f["<BiasParams>"], F32)[None, :]).astype(F32)
            elif c.type == "LinearComponent":
                out = self.matmul(x, np.asarray(f["<Params>"], F32).T)
            elif c.type == "TdnnComponent":
                W = np.asarray(f["<LinearParams>"], F32)
                b = np.asarray(f["<BiasParams>"], F32)
                d = x.shape[1]
                out = np.zeros((x.shape[0], W.shape[0]), F32)
                if b.size:
                    out += b[None, :]
                for i, o in enumerate(np.asarray(f["<TimeOffsets>"])):
                    out = (out + self.matmul(self._shift(x, int(o)), W[:, i * d:(i + 1) * d].T)).astype(F32)
            elif c.type == "RectifiedLinearComponent":
                out = np.maximum(x, F32(0.0))
            elif c.type == "BatchNormComponent":
                scl, off = self._bn(kv["component"], c)
                out = ((x * scl[None, :]).astype(F32) + off[None, :]).astype(F32)
            elif c.type in ("NoOpComponent", "DropoutComponent", "GeneralDropoutComponent"):
                out = x
            elif c.type == "LogSoftmaxComponent":
                mx = x.max(1, keepdims=True)
                out = (x - mx - np.log(np.exp(x - mx, dtype=F32).sum(1, keepdims=True, dtype=F32), dtype=F32)).astype(F32)
            elif c.type == "NormalizeComponent":
                rms = F32(f.get("<TargetRms>", 1.0))
                nrm = (x * x).sum(1, dtype=F32) / F32(x.shape[1] * rms * rms)
                nrm = np.power(np.maximum(nrm, F32(1.3552527156068805425e-20)), F32(-0.5))
                out = (x * nrm[:, None]).astype(F32)
            else:
                raise ValueError(f"oracle: unsupported component {c.type}")
        else:
            raise ValueError(name)
        self.memo[name] = out
        return out

    def forward(self, feats: np.ndarray, ivector_rows: Optional[np.ndarray], acoustic_scale: float = 1.0) -> np.ndarray:
        """feats: T x C; ivector_rows: None, or (T + 2*halo) x D giving the iVector seen by every padded row."""
        T, H = feats.shape[0], self.halo
        idx = np.clip(np.arange(-H, T + H), 0, T - 1)
        self.memo = {"input": feats[idx].astype(F32)}
        if ivector_rows is not None:
            self.memo["ivector"] = ivector_rows.astype(F32)
        out = self._node("output")[H:H + T]
        if self.nf.priors.size:
            out = (out + (-np.log(self.nf.priors, dtype=F32))[None, :]).astype(F32)
        return (out * F32(acoustic_scale)).astype(F32)


# =========================================================================================== decoder (C)

_dec_lib = None


def decoder_lib() -> C.CDLL:
    global _dec_lib
    if _dec_lib is None:
        so = HERE / "liboracle_decoder.so"
        if not so.exists():
            subprocess.run(["make", "-C", str(HERE)], check=True)
        lib = C.CDLL(str(so))
        lib.rs_oracle_decode.restype = C.c_void_p
        lib.rs_oracle_decode.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                                                 C.c_float, C.c_int, C.c_int, C.c_float, C.c_float]
        lib.rs_oracle_free.argtypes = [C.c_void_p]
        lib.rs_oracle_lattice_size.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        lib.rs_oracle_lattice_fill.argtypes = [C.c_void_p] * 10
        lib.rs_oracle_counters.argtypes = [C.c_void_p, C.c_void_p]
        _dec_lib = lib
    return _dec_lib


def decode(fst: kf.Fst, id2pdf: np.ndarray, loglikes: np.ndarray, beam=24.0, max_active=7000, min_active=200,
           lattice_beam=8.0, beam_delta=0.5) -> Tuple[lat.Lattice, List[int]]:
    lib = decoder_lib()
    ll = np.ascontiguousarray(loglikes, F32)
    keep = [np.ascontiguousarray(fst.final, F32), np.ascontiguousarray(fst.arc_begin, np.int64),
            np.ascontiguousarray(fst.num_ieps, np.int64), np.ascontiguousarray(fst.ilabel, np.int32),
            np.ascontiguousarray(fst.olabel, np.int32), np.ascontiguousarray(fst.weight, F32),
            np.ascontiguousarray(fst.nextstate, np.int32), ll, np.ascontiguousarray(id2pdf, np.int32)]
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    h = lib.rs_oracle_decode(len(fst.final), int(fst.start), *[p(a) for a in keep[:7]], p(ll), ll.shape[0], ll.shape[1], p(keep[8]),
                             float(beam), int(max_active), int(min_active), float(lattice_beam), float(beam_delta))
    try:
        na = C.c_int()
        ns = lib.rs_oracle_lattice_size(h, C.byref(na))
        na = na.value
        sf, fin = np.zeros(ns, np.int32), np.zeros(ns, F32)
        src, dst, il, ol = (np.zeros(na, np.int32) for _ in range(4))
        g, a = np.zeros(na, F32), np.zeros(na, F32)
        start = C.c_int32()
        lib.rs_oracle_lattice_fill(h, p(sf), p(fin), p(src), p(dst), p(il), p(ol), p(g), p(a), C.cast(C.byref(start), C.c_void_p))
        ctr = np.zeros(8, np.int64)
        lib.rs_oracle_counters(h, p(ctr))
    finally:
        lib.rs_oracle_free(h)
    return lat.Lattice(int(start.value), sf, fin, src, dst, il, ol, g, a), [int(x) for x in ctr]


# =========================================================================================== whole path

@dataclass
class Transcript:
    num_frames: int
    feats: np.ndarray
    ivector: Optional[np.ndarray]
    loglikes: np.ndarray
    nbest: List[lat.Path]
    lattice: Optional[lat.Lattice] = None
    counters: List[int] = field(default_factory=list)

    @property
    def words(self) -> List[int]:
        return self.nbest[0].words if self.nbest else []

    def text(self, key: str = "utt") -> bytes:
        return "".join(f"{key}-{k + 1} " + "".join(f"{w} " for w in p.words) + "\n" for k, p in enumerate(self.nbest)).encode()


class Oracle:
    """CPU restatement of one reference pipeline invocation (model + graph loaded once)."""

    def __init__(self, model_dir, graph_dir, beam=24.0, max_active=7000, min_active=None, lattice_beam=8.0, acoustic_scale=1.0,
                 frames_per_chunk=None, beam_delta=None, frame_subsampling_factor=None):
        model_dir, graph_dir = Path(model_dir), Path(graph_dir)
        conf = dict(kf.read_config(model_dir / "model" / "online" / "conf" / "online.conf"))
        # The binaries register decoder / decodable options on the parser that reads --config (online2-wav-nnet3-latgen-faster.cc:
        # 131-137): the config file is read first, the command line overrides it (util/parse-options.cc:328-345).  The keyword
        # arguments are the command line; beam / max_active / lattice_beam / acoustic_scale are always on rhasspy's
        # (transcribe_wav.py:46-55), the others only when given.
        if min_active is None:
            min_active = int(conf.get("min-active", 200))                  # lattice-faster-decoder.h:61
        if frames_per_chunk is None:
            frames_per_chunk = int(conf.get("frames-per-chunk", 24))      # decodable-simple-looped.h:57
        if beam_delta is None:
            beam_delta = float(conf.get("beam-delta", 0.5))                # lattice-faster-decoder.h:66
        if frame_subsampling_factor is None:
            frame_subsampling_factor = int(conf.get("frame-subsampling-factor", 1))      # decodable-simple-looped.h:56
        # the network is evaluated for the output frames t = 0, fsf, 2 fsf, ... and the decoder sees those as its frames
        # (decodable-online-looped.cc:56-84; nnet-compile-looped.cc:111-128); the chunk is the advised size rounded up to a
        # multiple of the factor (GetChunkSize, nnet-compile-looped.cc:81-94; every network here has modulus 1)
        self.fsf = frame_subsampling_factor
        while frames_per_chunk % self.fsf != 0:
            frames_per_chunk += 1
        self.opts = dict(beam=beam, max_active=max_active, min_active=min_active, lattice_beam=lattice_beam, beam_delta=beam_delta)
        self.acoustic_scale = acoustic_scale
        self.chunk = frames_per_chunk
        assert conf.get("feature-type", "mfcc") == "mfcc"
        self.id2pdf, nf = kf.read_final_mdl(model_dir / "model" / "model" / "final.mdl")
        # the dither of frame t is seeded by rand() value number (calls of the model set-up + t): oracle/nnet3_rand.py
        from . import nnet3_rand
        mo = MfccOpts.from_conf(conf["mfcc-config"]) if "mfcc-config" in conf else MfccOpts()
        self.rand_calls = nnet3_rand.setup_rand_calls(nf, frames_per_chunk, 0, self.fsf) if mo.dither != 0.0 else 0
        self.mfcc = Mfcc(mo, self.rand_calls)
        self.nnet = Nnet3(nf)
        self.fst = kf.read_fst(graph_dir / "HCLG.fst")
        self.nnet_cmvn = None
        if "cmvn-config" in conf:
            self.nnet_cmvn = kf.read_matrix_file(conf["global-cmvn-stats"]).astype(np.float64)
        self.ie = None
        if "ivector-extraction-config" in conf:
            ic = dict(kf.read_config(conf["ivector-extraction-config"]))
            sp = dict(kf.read_config(ic["splice-config"]))
            self.ie = dict(
                lda=kf.read_matrix_file(ic["lda-matrix"]).astype(F32), gstats=kf.read_matrix_file(ic["global-cmvn-stats"]).astype(np.float64),
                gmm=kf.read_diag_gmm(ic["diag-ubm"]), ext=kf.read_ivector_extractor(ic["ivector-extractor"]),
                left=int(sp.get("left-context", 0)), right=int(sp.get("right-context", 0)),
                num_gselect=int(ic.get("num-gselect", 5)), min_post=float(ic.get("min-post", 0.025)),
                posterior_scale=float(ic.get("posterior-scale", 0.1)), max_count=float(ic.get("max-count", 0.0)))

    # ---- iVector: frames [t0, t1) of the utterance accumulated into `stats`, with `T_ready` frames available
    def _ivector_acc(self, stats: IvectorStats, feats: np.ndarray, cm: np.ndarray, t0: int, t1: int, T_ready: int) -> None:
        ie = self.ie
        if t1 <= t0:
            return
        raw = lda_transform(splice(feats[:T_ready], ie["left"], ie["right"])[t0:t1], ie["lda"])
        nrm = lda_transform(splice(cm[:T_ready], ie["left"], ie["right"])[t0:t1], ie["lda"])
        posts = ubm_posteriors(nrm, ie["gmm"], ie["num_gselect"], ie["min_post"], ie["posterior_scale"])
        stats.acc(raw, posts)

    def features(self, pcm: np.ndarray) -> np.ndarray:
        return self.mfcc.compute(pcm)

    def offline_ivector(self, feats: np.ndarray) -> np.ndarray:
        ie = self.ie
        cm = online_cmvn(feats, ie["gstats"])
        st = IvectorStats(ie["ext"], ie["max_count"])
        self._ivector_acc(st, feats, cm, 0, feats.shape[0], feats.shape[0])
        x0 = np.zeros(ie["ext"].M.shape[2])
        x0[0] = ie["ext"].prior_offset
        x = st.get_ivector(x0)
        out = x.astype(F32)
        out[0] = F32(np.float64(out[0]) - ie["ext"].prior_offset)
        return out

    def loglikes_offline(self, feats: np.ndarray):
        nn_in = feats if self.nnet_cmvn is None else online_cmvn(feats, self.nnet_cmvn)
        iv = None
        rows = None
        if self.ie is not None:
            iv = self.offline_ivector(feats)
            rows = np.tile(iv[None, :], (feats.shape[0] + 2 * self.nnet.halo, 1))
        return nn_in, iv, self.nnet.forward(nn_in, rows, self.acoustic_scale)

    # ---- streaming semantics of online2-cli-nnet3-decode-faster (1024-sample ticks, one iVector per nnet chunk)
    def stream_schedule(self, n_samples: int, tick: int = 1024):
        """For every nnet chunk: (tick index at which it is computed, last frame whose stats the iVector has seen).
        decodable-online-looped.cc:56-84 (NumFramesReady), :186-194 (iVector frame), online2-cli...cc:143-161."""
        L, R = self.nnet.context()
        T = self.mfcc.num_frames(n_samples)
        nchunks = (T + self.chunk - 1) // self.chunk
        sched = []
        k = 0
        nt = (n_samples + tick - 1) // tick
        sr = self.ie["right"] if self.ie is not None else 0
        for j in range(nt):
            fr = self.mfcc.num_frames(min(tick * (j + 1), n_samples))
            ready = max(0, fr - R) // self.chunk
            while k < ready and k < nchunks:
                sched.append((j, min(fr - 1, fr - sr - 1)))
                k += 1
        while k < nchunks:
            sched.append((nt, T - 1))      # after InputFinished(): everything is available
            k += 1
        return sched, L, R

    def loglikes_stream(self, feats: np.ndarray, n_samples: int):
        nn_in = feats if self.nnet_cmvn is None else online_cmvn(feats, self.nnet_cmvn)
        sched, L, R = self.stream_schedule(n_samples)
        T = feats.shape[0]
        if self.ie is None:
            return nn_in, None, self.nnet.forward(nn_in, None, self.acoustic_scale)
        ie = self.ie
        cm = online_cmvn(feats, ie["gstats"])
        st = IvectorStats(ie["ext"], ie["max_count"])
        x = np.zeros(ie["ext"].M.shape[2])
        x[0] = ie["ext"].prior_offset
        done = 0            # frames already in the stats
        ivs = []
        for (_, last) in sched:
            if last + 1 > done:
                # frames done..last; their splice context is complete (or clamped at the true end after the flush)
                T_ready = T if last == T - 1 else last + 1 + ie["right"]
                self._ivector_acc(st, feats, cm, done, last + 1, min(T_ready, T))
                done = last + 1
                x = st.get_ivector(x)
            out = x.astype(F32)
            out[0] = F32(np.float64(out[0]) - ie["ext"].prior_offset)
            ivs.append(out)
        ivs = np.stack(ivs)
        # which chunk supplied the iVector "slot" of every padded row (nnet-compile-looped.cc:164-231)
        H = self.nnet.halo
        ts = np.arange(-H, T + H)
        slot = (ts // self.chunk) * self.chunk
        provider = {}
        ends = [self.chunk * (k + 1) + R for k in range(len(sched))]
        begin0 = -L
        for k in range(len(sched)):
            lo = begin0 if k == 0 else ends[k - 1]
            for t in range(lo, ends[k]):
                provider.setdefault((t // self.chunk) * self.chunk, k)
        maxk = len(sched) - 1
        idx = np.array([min(provider.get(int(sl), maxk if sl > 0 else 0), maxk) for sl in slot])
        return nn_in, ivs, self.nnet.forward(nn_in, ivs[idx], self.acoustic_scale)

    def transcribe_stream(self, pcm: np.ndarray, nbest: int = 1, lattice_acoustic_scale: float = 1.0) -> Transcript:
        pcm = np.asarray(pcm)
        feats = self.features(pcm)
        T = feats.shape[0]
        if T == 0:
            raise RuntimeError("You cannot get a lattice if you decoded no frames.")
        nn_in, ivs, ll = self.loglikes_stream(feats, len(pcm))
        ll = np.ascontiguousarray(ll[::self.fsf])
        lattice, ctr = decode(self.fst, self.id2pdf, ll, **self.opts)
        paths = lat.nbest(lattice, nbest, self.opts["lattice_beam"], lattice_acoustic_scale)
        return Transcript(ll.shape[0], nn_in, ivs, ll, paths, lattice, ctr)

    def transcribe(self, pcm: np.ndarray, nbest: int = 1, lattice_acoustic_scale: float = 1.0) -> Transcript:
        feats = self.features(np.asarray(pcm))
        T = feats.shape[0]
        if T == 0:
            raise RuntimeError("You cannot get a lattice if you decoded no frames.")
        nn_in, iv, ll = self.loglikes_offline(feats)
        ll = np.ascontiguousarray(ll[::self.fsf])
        lattice, ctr = decode(self.fst, self.id2pdf, ll, **self.opts)
        paths = lat.nbest(lattice, nbest, self.opts["lattice_beam"], lattice_acoustic_scale)
        return Transcript(ll.shape[0], nn_in, iv, ll, paths, lattice, ctr)
