"""Config 2 (ARPA-LM graph), utterance 162: the kernels' best-path cost is 2.05 above the reference decoder's, same words.  Which
decision differs?  The sequential oracle decoder (oracle/decoder.c, the reference's token order: HashList iteration, running
next_cutoff, stack-driven closure) is run twice on the oracle's log-likelihoods for that utterance:
  * as the reference runs (`tot_cost >= next_cutoff` tested against the RUNNING cutoff, lattice-faster-decoder.cc:774-787), and
  * with RS_ORACLE_FINAL_CUTOFF=1: every arc of the frame tested against the cutoff the loop ENDS with -- the kernels' rule,
and both are set beside the golden (the reference binary's) cost.  RS_ORACLE_TRACE lists, per frame, how many of the tokens
ProcessEmitting made lie at or above the frame's final cutoff (they exist only because an arc was looked at before the cutoff had
tightened -- which arcs those are depends on the hash list's order, and that order on this graph is the order in which the closure
of the frame before inserted its thousands of back-off states), and which tokens of the final best path were such tokens.
CPU only: python profiles/micro/c2_arpa_162.py [utterance]"""
import os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import configs
from oracle import pipeline

U = int(sys.argv[1]) if len(sys.argv) > 1 else 162
tmp = Path(tempfile.mkdtemp())
md, gd = configs.build_arpa_model(tmp / "m")
pcm = configs.arpa_utterances()[U]
gw, gg, ga = configs.load_golden("c2_arpa")
orc = pipeline.Oracle(md, gd)
tr = orc.transcribe(pcm)
print(f"c2_arpa utterance {U}: {tr.loglikes.shape[0]} frames, decoder options {orc.opts}")
print(f"  reference decoder (golden):  words {list(gw[U])}  graph {gg[U]:.4f} acoustic {ga[U]:.4f} total {gg[U] + ga[U]:.4f}")
for mode, what in (("0", "oracle decoder, reference order (running cutoff)"), ("1", "oracle decoder, final cutoff (the kernels' rule)")):
    trace = tmp / f"trace{mode}.txt"
    os.environ["RS_ORACLE_FINAL_CUTOFF"] = mode
    os.environ["RS_ORACLE_TRACE"] = str(trace)
    lattice, ctr = pipeline.decode(orc.fst, orc.id2pdf, tr.loglikes, **orc.opts)
    best = pipeline.lat.nbest(lattice, 1, orc.opts["lattice_beam"], 1.0)[0]
    print(f"  {what}:  words {best.words}  graph {best.graph_cost:.4f} acoustic {best.acoustic_cost:.4f} total {best.graph_cost + best.acoustic_cost:.4f}"
          f"  (tokens over all frames {ctr[3]}, frames max-active bound {ctr[5]}, min-active bound {ctr[6]})")
    lines = trace.read_text().splitlines()
    frames = [l.split() for l in lines if l.startswith("frame ")]
    ex = [(int(f[1]), int(f[3]), float(f[5]), float(f[7]), int(f[9]), int(f[11]), float(f[13])) for f in frames]
    tot_made, tot_ex = sum(e[4] for e in ex), sum(e[5] for e in ex)
    print(f"    tokens made by ProcessEmitting {tot_made}, of them at or above their frame's final cutoff {tot_ex} on {sum(1 for e in ex if e[5])} of {len(ex)} frames")
    bp = [l for l in lines if l.startswith("best_path frame")]
    print(f"    tokens of the best path that were born above the final cutoff: {len(bp)}")
    for l in bp:
        f = int(l.split()[2])
        e = ex[f - 1]
        nxt = ex[f] if f < len(ex) else None
        print(f"      {l[10:]}   [made on frame {e[0]}: adaptive beam {e[3]:.3f}, {e[5]} such tokens of {e[4]}; the frame after prunes at best + {nxt[2]:.3f} with {nxt[1]} tokens]" if nxt else f"      {l[10:]}")
