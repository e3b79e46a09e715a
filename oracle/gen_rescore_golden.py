#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- goldens for the rescoring path (runs in the build container only).

The reference's `async_transcribe_rescore` (rhasspy_speech/transcribe_wav.py:107-232) decodes with the old graph and then pipes
the lattice through
    lattice-scale --lm-scale=0.0 | lattice-to-phone-lattice final.mdl | lattice-compose - Ldet.fst | lattice-determinize |
    lattice-compose --phi-label=#0 - G.fst | lattice-add-trans-probs --transition-scale=1.0 --self-loop-scale=0.1 final.mdl |
    lattice-to-nbest --n=N --acoustic-scale=A | nbest-to-linear
with Ldet.fst = fstprint L_disambig.fst | awk (drop the arcs whose olabel is #0) | fstcompile | fstdeterminizestar | fstrmsymbols
disambig.int, made from the NEW language directory.  This script writes small NEW language directories (L_disambig.fst, G.fst,
words.txt, phones/disambig.int: data, committed under tests/golden/rescore/<lang>/) for the synthetic models of tests/cases.py,
runs exactly that pipeline with the reference's binaries (oracle/_ref) and stores, per (case, lang):
  * the lattice the reference decoder produced (binary CompactLattice entry) -- input of the host-only parity test,
  * the n-best text and nbest-to-linear's graph / acoustic costs after rescoring.

Usage: python oracle/gen_rescore_golden.py [lang style ...]      (the wav transcriber's path)
       python oracle/gen_rescore_golden.py --stream            (adds the streaming transcriber's runs to the committed cases.json)
"""
from __future__ import annotations

import json
import math
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from rhasspy_speech_amd import synth  # noqa: E402
from tests import cases  # noqa: E402

BIN = REPO / "oracle" / "_ref" / "bin"
OUT = REPO / "tests" / "golden" / "rescore"
ENV = dict(os.environ, PATH=f"{BIN}:{os.environ['PATH']}", OPENBLAS_NUM_THREADS="1")
NBEST = 5

# (name, parity case, sentences of the NEW grammar, extra made-up words, LM style)
LANGS = {
    # same vocabulary, other sentences: some of the old graph's sentences are not in the new G -> rejected or re-ranked
    "same_vocab": dict(keep_every=2, extra_sentences=6, backoff=False),
    # a back-off bigram G over the new sentences (phi arcs are taken), fewer words than the old graph knows
    "backoff": dict(keep_every=3, extra_sentences=10, backoff=True),
    # a grammar the way hassil_fst + KaldiTrainer._create_grammar (kaldi.py:311-341) make it: <eps> arcs (the weighted arc from the
    # start into each intent, bypasses of optional words, joins into the shared final state) run through
    # fstproject | fstdeterminize | fstminimize, which treat <eps> as a symbol and so KEEP it: G.fst has input-epsilon arcs
    "eps_grammar": dict(keep_every=2, extra_sentences=6, backoff=False, eps=True),
}
RUNS = [("tiny_u0", "eps_grammar"), ("zam_u0", "eps_grammar"), ("tiny_real_hot", "eps_grammar"), ("tiny_u0", "same_vocab"), ("tiny_u0", "backoff"), ("tiny_real_hot", "same_vocab"), ("tiny_arpa_u7", "backoff"),
        ("zam_u0", "same_vocab"), ("zam_u1", "backoff"), ("zam_real_cold", "backoff"), ("tinyf_u5", "same_vocab")]


def write_lang(lang_dir: Path, lex: synth.Lexicon, spec: synth.ModelSpec, conf: dict, rng: np.random.Generator) -> None:
    """L_disambig.fst (make_lexicon_fst.pl's shape: optional silence with probability 0.5, one disambiguation symbol per
    pronunciation that needs one, the #0 self-loop), G.fst (grammar or back-off bigram with #0 back-off arcs), words.txt,
    phones/disambig.int."""
    lang_dir.mkdir(parents=True, exist_ok=True)
    (lang_dir / "phones").mkdir(exist_ok=True)
    sents_all = [s.split() for s in synth.DEFAULT_SENTENCES]
    sents = [s for i, s in enumerate(sents_all) if i % conf["keep_every"] == 0]
    vocab_all = lex.words[1:]
    for _ in range(conf["extra_sentences"]):
        n = int(rng.integers(2, 6))
        sents.append([vocab_all[int(i)] for i in rng.integers(0, len(vocab_all), n)])
    vocab = sorted({w for s in sents for w in s})
    old_id = {w: i for i, w in enumerate(lex.words)}
    words = ["<eps>"] + vocab + ["#0"]
    wid = {w: i for i, w in enumerate(words)}
    (lang_dir / "words.txt").write_text("".join(f"{w} {i}\n" for i, w in enumerate(words)))
    # ---- lexicon: the OLD graph's pronunciations (same acoustic model, same phones)
    prons = {w: lex.prons[old_id[w]] for w in vocab}
    # disambiguation symbols: a pronunciation that equals or prefixes another one gets its own
    seqs = sorted(prons.items(), key=lambda kv: kv[1])
    need = {}
    nd = 0
    for w, p in prons.items():
        clash = any((q[:len(p)] == p) and (w2 != w) for w2, q in seqs)
        if clash:
            nd += 1
            need[w] = nd
    n_dis = max(nd, 1) + 1                       # (#1.. for words, the last one after optional silence)
    first_dis = spec.num_phones + 1            # phone ids 1..num_phones, then #0 #1 ...
    dis0 = first_dis                              # "#0" on the phone side
    sil_dis = first_dis + n_dis
    (lang_dir / "phones" / "disambig.int").write_text("".join(f"{first_dis + k}\n" for k in range(n_dis + 1)))
    nosil, sil = -math.log(0.5), -math.log(0.5)
    lines = []
    start, loop, silst, dsil = 0, 1, 2, 3
    nxt = 4
    lines.append(f"{start} {loop} 0 0 {nosil}")
    lines.append(f"{start} {silst} 0 0 {sil}")
    lines.append(f"{silst} {dsil} {lex.sil_phone} 0")
    lines.append(f"{dsil} {loop} {sil_dis} 0")
    for w in vocab:
        p = list(prons[w]) + ([first_dis + need[w]] if w in need else [])
        cur = loop
        for j, ph in enumerate(p):
            last = j == len(p) - 1
            ol = wid[w] if j == 0 else 0
            if not last:
                lines.append(f"{cur} {nxt} {ph} {ol}")
                cur = nxt
                nxt += 1
            else:
                lines.append(f"{cur} {loop} {ph} {ol} {nosil}")
                lines.append(f"{cur} {silst} {ph} {ol} {sil}")
    lines.append(f"{loop} {loop} {dis0} {wid['#0']}")
    lines.append(f"{loop} 0")
    subprocess.run(["fstcompile", "-", str(lang_dir / "L_disambig.fst")], input="\n".join(lines).encode() + b"\n", env=ENV, check=True)
    subprocess.run(["bash", "-c", f"fstarcsort --sort_type=olabel {lang_dir}/L_disambig.fst {lang_dir}/L_disambig.fst"], env=ENV, check=True)
    # ---- G
    g = []
    if conf.get("eps"):
        # intents = groups of sentences; start --<eps>/-log p(intent)--> intent start; words in sequence, every third word of a
        # sentence optional through an <eps> bypass; sentence end --<eps>--> one shared final state
        n_int = 3
        final = 1
        nxt_g = 2 + n_int
        for k in range(n_int):
            g.append(f"0 {2 + k} 0 0 {-math.log((k + 1.0) / (n_int * (n_int + 1) / 2.0))}")
        for i, s_ in enumerate(sents):
            cur = 2 + (i % n_int)
            for j, w in enumerate(s_):
                g.append(f"{cur} {nxt_g} {wid[w]} {wid[w]}")
                if j % 3 == 2:
                    g.append(f"{cur} {nxt_g} 0 0 {-math.log(0.25)}")
                cur = nxt_g
                nxt_g += 1
            g.append(f"{cur} {final} 0 0")
        g.append(f"{final} 0")
        # (the vendored OpenFst spells --project_type=input as the default of fstproject)
        sh = (f"fstcompile --keep_state_numbering=true - | fstproject | fstdeterminize | fstminimize | "
              f"fstarcsort --sort_type=ilabel - {lang_dir}/G.fst")
        subprocess.run(["bash", "-c", sh], input="\n".join(g).encode() + b"\n", env=ENV, check=True)
        return
    if not conf["backoff"]:
        # prefix tree of the sentences with relative-frequency costs (what rhasspy's grammar G looks like after determinisation)
        trie = {(): 0}
        counts = {}
        for s in sents:
            for k in range(len(s) + 1):
                counts[tuple(s[:k])] = counts.get(tuple(s[:k]), 0) + 1
        ends = {}
        for s in sents:
            ends[tuple(s)] = ends.get(tuple(s), 0) + 1
            for k in range(1, len(s) + 1):
                pre = tuple(s[:k])
                if pre in trie:
                    continue
                trie[pre] = len(trie)
                g.append(f"{trie[pre[:-1]]} {trie[pre]} {wid[pre[-1]]} {wid[pre[-1]]} {-math.log(counts[pre] / counts[pre[:-1]])}")
        for pre, c in ends.items():
            g.append(f"{trie[pre]} {-math.log(c / counts[pre])}")
    else:
        uni = {w: 1.0 for w in vocab}
        big = {}
        for s in sents:
            prev = "<s>"
            for w in s:
                uni[w] += 2.0
                big.setdefault(prev, {})[w] = big.get(prev, {}).get(w, 0) + 1
                prev = w
            big.setdefault(prev, {})["</s>"] = big.get(prev, {}).get("</s>", 0) + 1
        tot_u = sum(uni.values()) + 1.0
        st = {"<s>": 0, "<u>": 1}
        for w in vocab:
            st[w] = len(st)
        for w in vocab:
            g.append(f"{st['<u>']} {st[w]} {wid[w]} {wid[w]} {-math.log(uni[w] / tot_u)}")
        g.append(f"{st['<u>']} {-math.log(1.0 / tot_u)}")
        for h, nx in big.items():
            tot = sum(nx.values())
            lam = tot / (tot + len(nx))
            for w, c in nx.items():
                if w == "</s>":
                    g.append(f"{st[h]} {-math.log(lam * c / tot)}")
                else:
                    g.append(f"{st[h]} {st[w]} {wid[w]} {wid[w]} {-math.log(lam * c / tot + (1 - lam) * uni[w] / tot_u)}")
            g.append(f"{st[h]} {st['<u>']} {wid['#0']} 0 {-math.log(1 - lam)}")
        for w in vocab:
            if w not in big:
                g.append(f"{st[w]} {st['<u>']} {wid['#0']} 0 0.0")
    subprocess.run(["fstcompile", "-", str(lang_dir / "G.fst")], input="\n".join(g).encode() + b"\n", env=ENV, check=True)
    subprocess.run(["bash", "-c", f"fstarcsort --sort_type=ilabel {lang_dir}/G.fst {lang_dir}/G.fst"], env=ENV, check=True)


def phi_of(lang_dir: Path) -> int:
    for line in (lang_dir / "words.txt").read_text().splitlines():
        if line.startswith("#0 "):
            return int(line.split()[1])
    raise ValueError("no #0")


def reference_rescore(model_dir: Path, graph_dir: Path, wav: Path, lang_dir: Path, td: Path, case: dict):
    phi = phi_of(lang_dir)
    mdl = model_dir / "model" / "model" / "final.mdl"
    conf = model_dir / "model" / "online" / "conf" / "online.conf"
    # Ldet.fst exactly as transcribe_wav.py:131-142 makes it
    sh = (f"fstprint {lang_dir}/L_disambig.fst | awk '{{if($4 != {phi}){{print;}}}}' | fstcompile | fstdeterminizestar | "
          f"fstrmsymbols {lang_dir}/phones/disambig.int - {td}/Ldet.fst")
    subprocess.run(["bash", "-c", sh], env=ENV, check=True, stderr=subprocess.PIPE)
    o = dict(max_active=7000, lattice_beam=8.0, beam=24.0)
    o.update({k: v for k, v in case.get("opts", {}).items() if k in o})
    dec = (f"online2-wav-nnet3-latgen-faster --online=false --do-endpointing=false --word-symbol-table={graph_dir}/words.txt --config={conf} "
           f"--max-active={o['max_active']} --lattice-beam={o['lattice_beam']} --acoustic-scale=1.0 --beam={o['beam']} {mdl} {graph_dir}/HCLG.fst "
           f"'ark:echo utt utt|' 'scp:echo utt {wav}|' ark:{td}/dec.lat")
    subprocess.run(["bash", "-c", dec], env=ENV, check=True, stderr=subprocess.PIPE)
    tail = (f"lattice-scale --lm-scale=0.0 ark:{td}/dec.lat ark:- | lattice-to-phone-lattice {mdl} ark:- ark:- | lattice-compose ark:- {td}/Ldet.fst ark:- | "
            f"lattice-determinize ark:- ark:- | lattice-compose --phi-label={phi} ark:- {lang_dir}/G.fst ark:- | "
            f"lattice-add-trans-probs --transition-scale=1.0 --self-loop-scale=0.1 {mdl} ark:- ark:- | "
            f"lattice-to-nbest --n={NBEST} --acoustic-scale=1.0 ark:- ark:- | nbest-to-linear ark:- ark:/dev/null ark,t:- ark,t:{td}/lm.txt ark,t:{td}/ac.txt")
    r = subprocess.run(["bash", "-c", tail], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    text = r.stdout if r.returncode == 0 else b""

    def vec(p):
        out = {}
        if Path(p).exists():
            for line in Path(p).read_text().splitlines():
                q = line.split()
                if q:
                    out[q[0]] = float(q[1])
        return out
    lm, ac = vec(td / "lm.txt"), vec(td / "ac.txt")
    keys = sorted(lm, key=lambda k: int(k.split("-")[1]))
    return text, [lm[k] for k in keys], [ac[k] for k in keys], (td / "dec.lat").read_bytes()


def reference_rescore_stream(model_dir: Path, graph_dir: Path, pcm, lang_dir: Path, td: Path, case: dict):
    """The STREAMING transcriber's rescoring path (rhasspy_speech/transcribe_stream.py:131-274): the lattice comes from
    online2-cli-nnet3-decode-faster fed s16le on stdin (an iVector per chunk: other acoustic costs than the wav binary's), the tail
    is the same chain."""
    phi = phi_of(lang_dir)
    mdl = model_dir / "model" / "model" / "final.mdl"
    conf = model_dir / "model" / "online" / "conf" / "online.conf"
    sh = (f"fstprint {lang_dir}/L_disambig.fst | awk '{{if($4 != {phi}){{print;}}}}' | fstcompile | fstdeterminizestar | "
          f"fstrmsymbols {lang_dir}/phones/disambig.int - {td}/Ldet.fst")
    subprocess.run(["bash", "-c", sh], env=ENV, check=True, stderr=subprocess.PIPE)
    o = dict(max_active=7000, lattice_beam=8.0, beam=24.0)
    o.update({k: v for k, v in case.get("opts", {}).items() if k in o})
    dec = (f"online2-cli-nnet3-decode-faster --config={conf} --max-active={o['max_active']} --lattice-beam={o['lattice_beam']} --acoustic-scale=1.0 "
           f"--beam={o['beam']} {mdl} {graph_dir}/HCLG.fst {graph_dir}/words.txt ark:{td}/sdec.lat")
    subprocess.run(["bash", "-c", dec], env=ENV, check=True, input=np.asarray(pcm).astype("<i2").tobytes(), stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    tail = (f"lattice-scale --lm-scale=0.0 ark:{td}/sdec.lat ark:- | lattice-to-phone-lattice {mdl} ark:- ark:- | lattice-compose ark:- {td}/Ldet.fst ark:- | "
            f"lattice-determinize ark:- ark:- | lattice-compose --phi-label={phi} ark:- {lang_dir}/G.fst ark:- | "
            f"lattice-add-trans-probs --transition-scale=1.0 --self-loop-scale=0.1 {mdl} ark:- ark:- | "
            f"lattice-to-nbest --n={NBEST} --acoustic-scale=1.0 ark:- ark:- | nbest-to-linear ark:- ark:/dev/null ark,t:- ark,t:{td}/slm.txt ark,t:{td}/sac.txt")
    r = subprocess.run(["bash", "-c", tail], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    text = r.stdout if r.returncode == 0 else b""

    def vec(p):
        out = {}
        if Path(p).exists():
            for line in Path(p).read_text().splitlines():
                q = line.split()
                if q:
                    out[q[0]] = float(q[1])
        return out
    lm, ac = vec(td / "slm.txt"), vec(td / "sac.txt")
    keys = sorted(lm, key=lambda k: int(k.split("-")[1]))
    return text, [lm[k] for k in keys], [ac[k] for k in keys]


def add_stream_runs():
    """`gen_rescore_golden.py --stream`: adds stream_nbest_text / stream_graph_cost / stream_acoustic_cost to every entry of the
    committed cases.json, on the committed language directories (nothing else changes)."""
    index = json.loads((OUT / "cases.json").read_text())
    with tempfile.TemporaryDirectory() as tds:
        for e in index:
            case = cases.CASES[e["case"]]
            td = Path(tds) / e["dir"]
            td.mkdir()
            model_dir, graph_dir, wav, pcm = cases.build_case_files(case, td)
            text, lm, ac = reference_rescore_stream(model_dir, graph_dir, pcm, OUT / e["dir"], td, case)
            e["stream_nbest_text"], e["stream_graph_cost"], e["stream_acoustic_cost"] = text.decode(), lm, ac
            print(f"{e['case']} x {e['lang']} (stream): {text.decode().strip().replace(chr(10), ' | ') or '(nothing survives)'}")
    (OUT / "cases.json").write_text(json.dumps(index, indent=1))


def main():
    if sys.argv[1:] == ["--stream"]:
        add_stream_runs()
        return
    OUT.mkdir(parents=True, exist_ok=True)
    index = []
    only = set(sys.argv[1:])            # e.g. `gen_rescore_golden.py eps_grammar`: (re)generate that language style only
    if only and (OUT / "cases.json").exists():
        index = [e for e in json.loads((OUT / "cases.json").read_text()) if e["lang"] not in only]
    with tempfile.TemporaryDirectory() as tds:
        for case_name, lang in RUNS:
            if only and lang not in only:
                continue
            case = cases.CASES[case_name]
            td = Path(tds) / f"{case_name}_{lang}"
            td.mkdir()
            model_dir, graph_dir, wav, pcm = cases.build_case_files(case, td)
            spec = cases.case_spec(case)
            # the lexicon of the case's old graph (regenerated from the same seeds as build_case_files)
            g = case["graph"].split(":")
            if g[0] == "grammar":
                lex = synth.make_lexicon([s.split() for s in synth.DEFAULT_SENTENCES], spec, np.random.default_rng(11))
            else:
                lex = synth.make_lexicon([s.split() for s in synth.DEFAULT_SENTENCES], spec, np.random.default_rng(13), extra_words=int(g[1]))
            lang_dir = OUT / f"{case_name}__{lang}"
            write_lang(lang_dir, lex, spec, LANGS[lang], np.random.default_rng(sum(map(ord, lang)) + len(case_name)))
            text, lm, ac, lat = reference_rescore(model_dir, graph_dir, wav, lang_dir, td, case)
            (lang_dir / "decoder.lat").write_bytes(lat)
            index.append(dict(case=case_name, lang=lang, dir=lang_dir.name, nbest_text=text.decode(), graph_cost=lm, acoustic_cost=ac))
            print(f"{case_name} x {lang}: {text.decode().strip().replace(chr(10), ' | ') or '(nothing survives)'}")
    (OUT / "cases.json").write_text(json.dumps(index, indent=1))


if __name__ == "__main__":
    main()
