"""Does a model's result depend on what its kernels share a CU with?  One thread keeps a burner kernel (bf16 MFMA / f32 MFMA /
plain VALU; profiles/micro/mfma_burner.hip, compiled on the spot) running on its own stream while the main thread decodes
the same batch repeatedly and compares with the result obtained alone."""
import ctypes, os, subprocess, sys, tempfile, threading
from pathlib import Path
import numpy as np
HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))
from rhasspy_speech_amd import _lib, synth
from tests import cases
so = Path(tempfile.mkdtemp()) / "libburner.so"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", str(so), str(HERE / "mfma_burner.hip")], check=True, stderr=subprocess.DEVNULL)
burner = ctypes.CDLL(str(so))
name = sys.argv[1] if len(sys.argv) > 1 else "tinyf_u5"
with tempfile.TemporaryDirectory() as td:
    if name == "zam":
        spec = synth.ModelSpec(); md, gd = Path(td) / "m", Path(td) / "g"
        synth.write_model_dir(md, spec); synth.make_grammar_graph(gd, spec)
        pcms = [synth.synth_utterance(21300 + u, 48000 - 320 * ((u + 3) % 11)) for u in range(72)]
    else:
        md, gd, _, _ = cases.build_case_files(cases.CASES[name], Path(td))
        pcms = [synth.synth_utterance(9500 + u, 30000 + 900 * (u % 7)) for u in range(160)]
    m = _lib.Model(md, gd, _lib.default_opts(keep_intermediates=1))
    ref = m.decode_batch(pcms)
    for kind, label in ((2, "valu"), (1, "f32 mfma"), (0, "bf16 mfma 64 acc regs"), (3, "bf16 mfma 160 acc regs")):
        stop = False
        def bg():
            while not stop:
                burner.burn(kind, 512, 3000, 2)
        th = threading.Thread(target=bg); th.start()
        bad = 0
        for it in range(8):
            r = m.decode_batch(pcms)
            for u in range(len(pcms)):
                if r.costs(u) != ref.costs(u) or r.words(u) != ref.words(u):
                    bad += 1
                    if bad <= 3:
                        msg = ""
                        for k, nm in ((0, "feat"), (1, "ivec"), (2, "loglikes")):
                            d = np.abs(r.matrix(u, k) - ref.matrix(u, k))
                            msg += f" {nm} {d.max():.3g}"
                        print(f"  [{label}] iteration {it} utt {u}:{msg}", flush=True)
        stop = True; th.join()
        print(f"{name}: beside a {label} burner: {bad} of {8 * len(pcms)} utterance results differ", flush=True)
