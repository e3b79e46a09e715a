#!/usr/bin/env python3
"""Headline benchmark: audio-seconds decoded per wall-second (RTF^-1) of the transcribe hot path.

Workload at N=1 = BASELINE.json configs[1]: "en_US-zamia grammar HCLG, batch of 256 synthetic 3 s utterances on 1
MI355X".  No real zamia model exists offline, so the model is the synthetic "zamia-like-S" of SURVEY.md section
8(d) written in genuine Kaldi formats by rhasspy_speech_amd.synth (40-dim hires MFCC, 100-dim iVector with a
512-Gaussian UBM, 7x250 TDNN + prefinal, 2000 pdfs) and a grammar HCLG; audio is synthetic (seeded).  A "step" is
one pass of the whole path (MFCC -> iVector -> TDNN -> beam search -> word ids) over the 256-utterance batch,
with the int16 samples already resident in HBM when the timed region starts.  The K timed steps are submitted from a few
host threads (`--inflight`, default 4) so that consecutive batches overlap on the device, as a serving process would run
them: the latency-bound search of one batch shares the CUs with the GEMMs of the next.  Every step's result records are
checked against the first step's (same input), and every step is complete before the closing synchronize + barrier.
Stage times and the roofline are taken from un-overlapped calls made right after the timed region (`--inflight 1` gives
the one-call-at-a-time figure for the whole step).

Multi-GPU (driver launches `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`):
utterances shard embarrassingly, one process per GPU, each rank decodes its own 256-utterance batch (weak
scaling, no data-path collective); fixed-size result records are gathered over RCCL (all_gather) after the
timed region's compute, inside the timed region.

Prints ONE JSON line (rank 0) with the fields the driver expects plus `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_UTTS = 256
N_SAMPLES = 48000          # 3 s @ 16 kHz
MAX_WORDS = 62             # result record: 64 x int32 = [n_words, words..., pad] + 2 floats


def build_workload(root: Path, n_utts: int, rank: int):
    from rhasspy_speech_amd import synth
    spec = synth.ModelSpec()
    model_dir, graph_dir = root / "model", root / "graph"
    if not (graph_dir / "HCLG.fst").exists():
        synth.write_model_dir(model_dir, spec)
        synth.make_grammar_graph(graph_dir, spec)
    pcm = np.stack([synth.synth_utterance(rank * 100000 + u, N_SAMPLES) for u in range(n_utts)])
    return spec, model_dir, graph_dir, pcm


def nnet_flops_per_row(desc: str) -> float:
    """2 * K * N summed over the GEMM ops listed by rs_model_describe (algorithmic FLOPs per frame row)."""
    fl = 0.0
    for line in desc.splitlines():
        if line.startswith("op: gemm"):
            parts = dict(p.split("=") for p in line.split() if "=" in p)
            fl += 2.0 * float(parts["k"]) * float(parts["out_dim"])
    return fl


def gemm_traffic_bytes(n_gemm: int):
    """HBM bytes per nnet GEMM launch from the committed PMC passes (profiles/collect.sh -> profiles/r01/bench_v4_pmc.json):
    FETCH_SIZE (KB, doubled: this rocprofv3 tallies the 128-B requests of a 16 B/lane streaming read at 64 B) + WRITE_SIZE
    (KB), averaged over the launches of the nnet stage.  None when the summary is absent."""
    path = ROOT / "profiles" / "r01" / "bench_v4_pmc.json"
    if not path.exists():
        return None
    ks = json.loads(path.read_text())["kernels"]
    tot, n = 0.0, 0
    for name, c in ks.items():
        if "GemmKernel" not in name or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or "<2, 4, 1" in name or "<1, 4, 1" in name or "GemmKernelDma" in name:
            continue        # the narrow (BN = 64) instantiations are the two iVector LDA launches, not the nnet stage
        tot += (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0 * c["FETCH_SIZE"]["launches"]
        n += c["FETCH_SIZE"]["launches"]
    return tot / n if n else None


def cpu_baseline(model_dir: Path, graph_dir: Path, pcm: np.ndarray, seconds_budget: float = 20.0):
    """Times the REFERENCE itself (oracle/_ref Kaldi binaries built from /root/reference by oracle/build_ref.sh)
    on this box's host cores on a bounded sample of the same workload: the 3-process pipeline of
    transcribe_wav.py:45-75, one utterance per pipeline invocation, one pipeline at a time (1 core)."""
    from rhasspy_speech_amd import synth
    bin_dir = ROOT / "oracle" / "_ref" / "bin"
    exe = bin_dir / "online2-wav-nnet3-latgen-faster"
    if not exe.exists():
        return None
    # one BLAS thread per process: "cores" below is then what the processes really use (the OpenBLAS the oracle build links
    # would otherwise start a thread per host core in every process)
    env = dict(os.environ, PATH=f"{bin_dir}:{os.environ['PATH']}", OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    conf = model_dir / "model" / "online" / "conf" / "online.conf"
    n_done, audio, t0 = 0, 0.0, time.perf_counter()
    with tempfile.TemporaryDirectory() as td:
        while n_done < pcm.shape[0] and (time.perf_counter() - t0) < seconds_budget and n_done < 64:
            wav = Path(td) / "u.wav"
            synth.write_wav(wav, pcm[n_done])
            cmd = (f"online2-wav-nnet3-latgen-faster --online=false --do-endpointing=false "
                   f"--word-symbol-table={graph_dir}/words.txt --config={conf} --max-active=7000 --lattice-beam=8.0 "
                   f"--acoustic-scale=1.0 --beam=24.0 {model_dir}/model/model/final.mdl {graph_dir}/HCLG.fst "
                   f"'ark:echo utt utt|' 'scp:echo utt {wav}|' ark:- | lattice-to-nbest --n=1 --acoustic-scale=1.0 ark:- ark:- | "
                   f"nbest-to-linear ark:- ark:/dev/null ark,t:-")
            r = subprocess.run(["bash", "-c", cmd], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if r.returncode != 0:
                return None
            n_done += 1
            audio += pcm.shape[1] / 16000.0
        wall = time.perf_counter() - t0
        # Beside it: the same binaries fed a table of utterances, so the model and HCLG load once -- the reference's decode
        # rate without its per-call start-up (not how rhasspy-speech calls them, but the fairer figure for the kernels).
        n_tab = min(16, pcm.shape[0])
        for i in range(n_tab):
            synth.write_wav(Path(td) / f"t{i}.wav", pcm[i])
        (Path(td) / "wav.scp").write_text("".join(f"utt{i} {td}/t{i}.wav\n" for i in range(n_tab)))
        (Path(td) / "spk2utt").write_text("".join(f"utt{i} utt{i}\n" for i in range(n_tab)))
        cmd = (f"online2-wav-nnet3-latgen-faster --online=false --do-endpointing=false "
               f"--word-symbol-table={graph_dir}/words.txt --config={conf} --max-active=7000 --lattice-beam=8.0 "
               f"--acoustic-scale=1.0 --beam=24.0 {model_dir}/model/model/final.mdl {graph_dir}/HCLG.fst "
               f"ark:{td}/spk2utt scp:{td}/wav.scp ark:- | lattice-to-nbest --n=1 --acoustic-scale=1.0 ark:- ark:- | "
               f"nbest-to-linear ark:- ark:/dev/null ark,t:-")
        t1 = time.perf_counter()
        r = subprocess.run(["bash", "-c", cmd], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall_tab = time.perf_counter() - t1
        steady = None
        if r.returncode == 0 and len(r.stdout.decode().splitlines()) == n_tab:
            steady = {"value": n_tab * pcm.shape[1] / 16000.0 / wall_tab, "unit": "audio-seconds/s", "cores": 1,
                      "sample": f"{n_tab} utterances through ONE pipeline invocation (model + HCLG loaded once)"}
        # ... and on all host cores, the way a Kaldi deployment scales: one such single-load pipeline per core, side by side
        # (SURVEY.md section 8(d)); every worker decodes the same table.
        n_workers = max(1, min(os.cpu_count() or 1, 64))
        all_cores = None
        if steady is not None:
            t2 = time.perf_counter()
            procs = [subprocess.Popen(["bash", "-c", cmd], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(n_workers)]
            outs = [p.communicate()[0] for p in procs]
            wall_all = time.perf_counter() - t2
            if all(p.returncode == 0 for p in procs) and all(len(o.decode().splitlines()) == n_tab for o in outs):
                all_cores = {"value": n_workers * n_tab * pcm.shape[1] / 16000.0 / wall_all, "unit": "audio-seconds/s", "cores": n_workers,
                             "sample": f"{n_workers} single-load pipelines side by side, {n_tab} utterances each"}
    return {"value": audio / wall, "unit": "audio-seconds/s", "cores": 1, "kind": "reference",
            "sample": f"{n_done} of the {pcm.shape[0]} utterances, one transcribe_wav.py-style 3-process pipeline per "
                      f"utterance (model + HCLG re-loaded every call, as the reference does)",
            "one_load": steady, "all_cores": all_cores}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--utts", type=int, default=N_UTTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=4,
                    help="decode calls in flight per rank (host threads on one model; the library gives each its own decode "
                         "context): the latency-bound search of one batch overlaps the GEMMs of the next.  1 = one call at a time")
    ap.add_argument("--prune-output", action="store_true",
                    help="rs_decode_opts.prune_output_pdfs=1: output layer only for the pdfs on HCLG arcs (NOT the default: the "
                         "headline line computes every pdf, as the reference does)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP library has no CPU fallback")
    # One rank per GPU.  (Test hook: RS_BENCH_BACKEND=gloo lets two ranks share the single GPU of a test box -- RCCL refuses
    # two ranks on one device -- to exercise the N > 1 control flow; the records then travel through host memory.)
    backend = os.environ.get("RS_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    comm_device = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from rhasspy_speech_amd import _lib
    cache = Path(tempfile.gettempdir()) / f"rs_bench_zamia_like_S_rank{rank}"
    spec, model_dir, graph_dir, pcm = build_workload(cache, args.utts, rank)
    model = _lib.Model(model_dir, graph_dir, _lib.default_opts(device_id=local_rank, prune_output_pdfs=1 if args.prune_output else 0))
    model.to_device()
    desc = model.describe()
    d_pcm = torch.from_numpy(pcm.reshape(-1)).to(f"cuda:{local_rank}")
    offsets = np.arange(args.utts + 1, dtype=np.int64) * N_SAMPLES
    audio_seconds = args.utts * N_SAMPLES / 16000.0

    def gather(res):
        rec = res.pack(MAX_WORDS)          # fixed 264-byte records: status, n_words, word ids, graph/acoustic cost
        if world > 1:
            t = torch.from_numpy(rec).to(comm_device)
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t)       # the path's one exchange step: fixed-size result records over RCCL/xGMI
            rec = torch.cat(out).cpu().numpy()
        return rec

    def decode():
        return model.decode_batch_device(d_pcm.data_ptr(), offsets)

    import concurrent.futures
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(args.inflight, 1))

    def run_steps(n, check_against=None):
        """n steps, at most --inflight decode calls in flight; results are consumed (and gathered) in step order."""
        futures = [pool.submit(decode) for _ in range(n)] if args.inflight > 1 else None
        last = None
        for k in range(n):
            res = futures[k].result() if futures else decode()
            rec = gather(res)
            if check_against is not None and not np.array_equal(rec, check_against):
                raise SystemExit(f"bench.py: step {k} produced different results from the first step on the same input")
            last = (res, rec)
        return last

    res, ref_rec = run_steps(1)
    run_steps(max(args.warmup - 1, 0), ref_rec)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, rec = run_steps(args.steps, ref_rec)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=comm_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # Stage times and the roofline come from un-overlapped calls made right after the timed region (with two calls in flight
    # the events around a stage also see the other call's kernels): same process, same buffers, one call at a time.
    n_iso = 5
    stage = np.zeros(8)
    counters = np.zeros(8)
    for _ in range(n_iso):
        r1 = decode()
        stage += np.array(r1.timings())
    stage /= n_iso
    for u in range(args.utts):
        counters += np.array(r1.counters(u), dtype=np.float64)

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / args.steps
        value = world * audio_seconds * args.steps / elapsed
        rows = args.utts * 298          # algorithmic: the real frames only (halo rows of the hidden layers are overhead)
        flops = nnet_flops_per_row(desc) * rows
        n_gemm = sum(1 for l in desc.splitlines() if l.startswith("op: gemm"))
        # decoder algorithmic bytes: arcs examined x (16 B arc + 4 B loglike), token
        # insertions x 16 B (8 B table key read-modify-write twice), tokens alive x 16 B token record
        dec_bytes = counters[1] * 20.0 + counters[2] * 16.0 + counters[3] * 16.0
        # dominant kernel: the segmented layer GEMM (one launch per affine layer; GemmKernelB3 for the wide layers: every FP32
        # product is six bf16 MFMAs on split operands, nnet_gemm_b3.hip).  achieved = ALGORITHMIC FP32 FLOPs of the stage's
        # launches / their duration, timed with HIP events on the library's stream (rs_result_timings).  peak = the dense
        # bf16 MFMA peak (2500 TFLOP/s, MI355X_MICROARCH.md) / 6 MFMAs per FP32 product = what this formulation can reach;
        # the exact-FP32 MFMA peak (157.3) is what the previous FP32-input kernel was priced against.
        split_bf16 = os.environ.get("RS_GEMM_B3", "1") != "0"
        peak = 2500.0 / 6.0 if split_bf16 else 157.3
        achieved = flops / (stage[3] * 1e-3) / 1e12
        roof_mfma = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": gemm_traffic_bytes(n_gemm),
                     "kernel": (f"GemmKernelB3 (FP32 operands split into 3 bf16 parts, 6 bf16 MFMAs per product, FP32 accumulate), "
                                f"{n_gemm} launches per step (the nnet stage)") if split_bf16 else f"GemmKernel, {n_gemm} launches per step",
                     "launches": n_gemm, "avg_launch_ms": float(stage[3]) / n_gemm, "flops_per_launch": flops / n_gemm,
                     "stage_ms": float(stage[3]), "frac_of_fp32_mfma_peak": achieved / 157.3}
        roof_dec = {"bound": "hbm", "achieved": dec_bytes / (stage[4] * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": dec_bytes / (stage[4] * 1e-3) / 1e9 / 8000.0, "traffic": None,
                    "kernel": "RegDecodeKernel (1 launch, one workgroup per utterance, latency-bound)", "stage_ms": float(stage[4])}
        roofline = roof_mfma if stage[3] >= stage[4] else roof_dec
        out = {
            "metric": "audio-seconds decoded/sec (RTF^-1) en_US-zamia grammar HCLG", "value": value, "unit": "audio-seconds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "FP32 results within the same 1e-4 bound as before; the wide layer GEMMs multiply 3-way bf16 splits of the FP32 operands (24 significand bits) on the bf16 matrix cores and accumulate in FP32",
            "config": {"workload": f"zamia-like-S synthetic Kaldi model (40-dim MFCC, 100-dim iVector, 7x250 TDNN, 2000 pdfs), "
                                   f"grammar HCLG, {args.utts} x 3 s utterances per GPU, beam 24 / max-active 7000 / lattice-beam 8",
                       "utts_per_gpu": args.utts, "seconds_per_utt": 3.0, "parallelism": f"utterance-sharded x{world}",
                       "output_layer": "pruned to the pdfs on HCLG arcs (--prune-output)" if args.prune_output else "all pdfs",
                       "calls_in_flight": args.inflight},
            "results_checked": "every step's result records equal the first step's (same input)",
            "stages_from": f"{n_iso} un-overlapped calls after the timed region",
            "roofline": roofline,
            "stages_ms": {"mfcc": float(stage[1]), "ivector": float(stage[2]), "nnet": float(stage[3]), "decode": float(stage[4]),
                          "d2h+host": float(stage[5]), "total_call": float(stage[6])},
            "other_roofline": roof_dec if roofline is roof_mfma else roof_mfma,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model_dir, graph_dir, pcm)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
