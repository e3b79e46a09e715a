"""Patches a scratch copy of feat_kernels.hip for profiles/micro/pk_bisect3.sh: moves one region of MfccKernel into a
noinline + optnone helper (no SLP vectorizer there, everything else as built).  argv: <file> <variant: k1 | pw | none>"""
import sys
path, variant = sys.argv[1], sys.argv[2]
s = open(path).read()
if variant == "k1":
    old_head = "  } else if (kind == 1) {\n    // srfft.cc:227-264: the whole length-4 transform\n"
    i = s.index(old_head)
    j = s.index("  } else {\n    // srfft.cc:265-274: length 2", i)
    body = s[i + len(old_head):j]
    helper = ("__device__ __attribute__((noinline, optnone)) void SrfftLen4(float *xr, float *xi, int off) {\n" + body + "}\n\n")
    s = s[:i] + "  } else if (kind == 1) {\n    SrfftLen4(xr, xi, off);\n" + s[j:]
    k = s.index("__device__ __forceinline__ void SrfftRunTask")
    s = s[:k] + helper + s[k:]
elif variant == "pw":
    old_head = "    for (int k = lane + 1; 2 * k <= NC; k += RS_WAVE) {\n"
    i = s.index(old_head)
    j = s.index("    if (lane == 0) {\n      const float d0 = xr[m.fft_perm[0]]", i)
    body = s[i + len(old_head):j]
    assert body.rstrip().endswith("}")
    body = body.rstrip()[:-1]          # the loop's closing brace
    helper = ("template <int NC>\n__device__ __attribute__((noinline, optnone)) void PowerSpectrumK(const MfccDev &m, const float *xr, const float *xi, float *pwrow, int k) {\n"
              + body.replace("pw[wave][", "pwrow[") + "}\n\n")
    s = s[:i] + old_head + "      PowerSpectrumK<NC>(m, xr, xi, pw[wave], k);\n    }\n" + s[j:]
    k = s.index("template <int NFFT, int WPB>   // padded window (real points)")
    s = s[:k] + helper + s[k:]
open(path, "w").write(s)
