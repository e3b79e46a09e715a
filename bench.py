#!/usr/bin/env python3
"""Headline benchmark: audio-seconds decoded per wall-second (RTF^-1) of the transcribe hot path.

`--workload` selects one of BASELINE.json's GPU configurations (tests/configs.py holds their inputs; every utterance of
each is pinned to the reference's transcript in tests/test_gpu_configs.py):
  grammar (default, configs[1], the headline): zamia-like-S model, grammar HCLG, 256 x 3 s utterances per GPU;
  arpa    (configs[2]): the same model on the back-off ARPA HCLG, 256 x 3 s;
  mixed   (configs[3]): 1024 utterances naming two zamia-size models, sharded over the ranks, one gather;
  streams (configs[4]): 64 concurrent 30 s streams (rs_streams_* entry points) handed over round-robin in rounds of 8192 samples per stream
          (8 of the reference binary's 1024-sample reads; the library's default coalescing makes every second rs_streams_advance call do
          the device work of two) -- with `streams_per_tick` (1024-sample rounds, every call advances: SURVEY 8(d)'s wording) and
          `finish_latency_ms` (p50 / p99 of last-samples-in -> words-out at real-time arrival) beside it;
No real zamia model exists offline, so the model is the synthetic "zamia-like-S" of SURVEY.md section 8(d) written in
genuine Kaldi formats by rhasspy_speech_amd.synth (40-dim hires MFCC, 100-dim iVector with a 512-Gaussian UBM, 7x250
TDNN + prefinal, 2000 pdfs); audio is synthetic (seeded).

A "step" is one pass of the whole path (MFCC -> iVector -> TDNN -> beam search -> word ids) over the workload's batch.
`value` is SURVEY.md section 8(d)'s timed region: int16 PCM in pageable HOST memory -> word ids in host memory, model and
graph resident (rs_decode_batch; rs_decode_batch_sharded at N > 1; rs_stream_* for `streams`) -- PCIe-inclusive.  Beside it,
each over the same number of steps: `hbm_resident` (the samples already in HBM when the timed region starts,
rs_decode_batch_device), `reference_output_layer` (host PCM AND the output layer evaluated for all 2000 pdfs, which is
literally what the reference computes; the library default evaluates the 362 pdfs that occur on HCLG arcs: same words, same
costs) and `hbm_resident_all_pdfs`.  The K timed steps are submitted from a few host threads (`--inflight`, default 4; 3 on the ARPA graph and for the two-model batch) so
that consecutive batches overlap on the device, as a serving process would run them.  Every step's records are checked
against the first step's and -- rank 0 -- against the REFERENCE's transcripts (tests/golden/configs).  Stage times and the
roofline are taken from un-overlapped calls right after the timed region.

Multi-GPU (driver: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`): utterances shard
embarrassingly (utterance i -> rank i % N, rs_decode_batch_sharded), one process per GPU, weak scaling, no data-path
collective; the fixed-size result records are gathered inside the timed region by ONE ncclAllGather per step issued by the
library itself (rs_shard_gather on torch.distributed's RCCL communicator; torch's all_gather when that is not available).

Prints ONE JSON line (rank 0) with the fields the driver expects plus `roofline` and `cpu_baseline`.  A default one-GPU run of the
headline workload also carries `steady_state` (the same steps repeated for at least a second when the requested timed region is
shorter than that: the driver's `--steps 20` is 45 ms, inside which the pipeline of calls in flight fills and drains) and
`other_workloads`: short runs (>= 1 s of timed work each, transcripts checked against the reference's goldens like the headline's)
of the other three configurations -- arpa, mixed, streams -- each with its ms_per_step, value and the roofline of its dominant kernel.
"""
from __future__ import annotations

import argparse
import concurrent.futures
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

MAX_WORDS = 62             # rs_result_pack record: status, n_words, 62 word ids, 2 float costs = 264 B


def nnet_flops_per_row(desc: str) -> float:
    """2 * K * N summed over the GEMM ops listed by rs_model_describe (algorithmic FLOPs per frame row)."""
    fl = 0.0
    for line in desc.splitlines():
        if line.startswith("op: gemm"):
            parts = dict(p.split("=") for p in line.split() if "=" in p)
            fl += 2.0 * float(parts["k"]) * float(parts["out_dim"])
    return fl


def pmc_traffic(workload: str, kernel_substr: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS round
    (profiles/collect.sh -> profiles/r06/<workload>_pmc.json): (2 x FETCH_SIZE + WRITE_SIZE) KB, the doubling per
    MI355X_MICROARCH.md (128-B requests of streaming reads tallied at 64 B on gfx950).  None when no summary of this round
    is committed for the workload -- the line never carries a stale figure."""
    path = ROOT / "profiles" / "r06" / f"{workload}_pmc.json"
    if not path.exists():
        return None, None
    ks = json.loads(path.read_text())["kernels"]
    tot, n = 0.0, 0
    for name, c in ks.items():
        if kernel_substr not in name or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        tot += (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0 * c["FETCH_SIZE"]["launches"]
        n += c["FETCH_SIZE"]["launches"]
    return (tot / n, str(path.relative_to(ROOT))) if n else (None, None)


OTHER_WORKLOADS = {"arpa": (45, 7), "mixed": (120, 11), "streams": (30, 2)}      # (timed steps, warm-up): >= 1 s of timed work each


def other_workloads(timeout_s: float = 240.0):
    """BASELINE configs[2..4] as short runs of this script (one process each, one after the other, on the same GPU); what
    the line keeps of each: value, ms_per_step, steps, the golden check, stage times and the roofline of its dominant kernel."""
    out = {}
    for wl, (steps, warm) in OTHER_WORKLOADS.items():
        cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", wl, "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--no-side-figures"]
        if wl == "streams":
            cmd.append("--stream-figures")
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s, env=dict(os.environ, RS_BENCH_NO_OTHER="1"))
            line = json.loads(r.stdout.decode().strip().splitlines()[-1]) if r.returncode == 0 else None
        except (subprocess.TimeoutExpired, ValueError, IndexError):
            r, line = None, None
        if line is None:
            out[wl] = {"error": (r.stderr.decode()[-400:] if r is not None else f"no result within {timeout_s:.0f} s")}
            continue
        keep = {k: line.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "timed_seconds", "results_checked", "stages_ms", "roofline", "roofline_note")}
        for k in ("streams_per_tick", "finish_latency_ms"):
            if k in line:
                keep[k] = line[k]
        keep["workload"] = line["config"]["workload"]
        keep["calls_in_flight"] = line["config"]["calls_in_flight"]
        keep["golden_checked"] = "equal the reference's" in (line.get("results_checked") or "")
        keep["wall_seconds_incl_setup"] = time.perf_counter() - t0
        out[wl] = keep
    return out


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max), None when unlimited / unknown."""
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        return None


def cpu_baseline(model_dir: Path, graph_dir: Path, pcms, streaming: bool, seconds_budget: float = 20.0):
    """Times the REFERENCE itself (oracle/_ref Kaldi binaries built from /root/reference by oracle/build_ref.sh) on this
    box's host cores on a bounded sample of the same workload: the pipeline of transcribe_wav.py:45-75 (or, for streams,
    transcribe_stream.py:53-99), one utterance per pipeline invocation, one pipeline at a time (1 core)."""
    from rhasspy_speech_amd import synth
    bin_dir = ROOT / "oracle" / "_ref" / "bin"
    if not (bin_dir / "online2-wav-nnet3-latgen-faster").exists():
        return None
    # one BLAS thread per process: "cores" below is then what the processes really use
    env = dict(os.environ, PATH=f"{bin_dir}:{os.environ['PATH']}", OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1")
    conf = model_dir / "model" / "online" / "conf" / "online.conf"
    dec = "--max-active=7000 --lattice-beam=8.0 --acoustic-scale=1.0 --beam=24.0"
    tail = "lattice-to-nbest --n=1 --acoustic-scale=1.0 ark:- ark:- | nbest-to-linear ark:- ark:/dev/null ark,t:-"
    n_done, audio, t0 = 0, 0.0, time.perf_counter()
    with tempfile.TemporaryDirectory() as td:
        while n_done < len(pcms) and (time.perf_counter() - t0) < seconds_budget and n_done < 64:
            p = pcms[n_done]
            if streaming:
                cmd = (f"online2-cli-nnet3-decode-faster --config={conf} {dec} {model_dir}/model/model/final.mdl {graph_dir}/HCLG.fst "
                       f"{graph_dir}/words.txt ark:- | {tail}")
                r = subprocess.run(["bash", "-c", cmd], env=env, input=np.asarray(p, "<i2").tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            else:
                wav = Path(td) / "u.wav"
                synth.write_wav(wav, p)
                cmd = (f"online2-wav-nnet3-latgen-faster --online=false --do-endpointing=false --word-symbol-table={graph_dir}/words.txt "
                       f"--config={conf} {dec} {model_dir}/model/model/final.mdl {graph_dir}/HCLG.fst 'ark:echo utt utt|' 'scp:echo utt {wav}|' ark:- | {tail}")
                r = subprocess.run(["bash", "-c", cmd], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if r.returncode != 0:
                return None
            n_done += 1
            audio += len(p) / 16000.0
        wall = time.perf_counter() - t0
        steady = all_cores = None
        if not streaming:
            # Beside it: the same binaries fed a table of utterances, so the model and HCLG load once -- the reference's decode
            # rate without its per-call start-up (not how rhasspy-speech calls them, but the fairer figure for the kernels).
            n_tab = min(16, len(pcms))
            for i in range(n_tab):
                synth.write_wav(Path(td) / f"t{i}.wav", pcms[i])
            (Path(td) / "wav.scp").write_text("".join(f"utt{i} {td}/t{i}.wav\n" for i in range(n_tab)))
            (Path(td) / "spk2utt").write_text("".join(f"utt{i} utt{i}\n" for i in range(n_tab)))
            cmd = (f"online2-wav-nnet3-latgen-faster --online=false --do-endpointing=false --word-symbol-table={graph_dir}/words.txt "
                   f"--config={conf} {dec} {model_dir}/model/model/final.mdl {graph_dir}/HCLG.fst ark:{td}/spk2utt scp:{td}/wav.scp ark:- | {tail}")
            tab_audio = sum(len(pcms[i]) for i in range(n_tab)) / 16000.0
            t1 = time.perf_counter()
            r = subprocess.run(["bash", "-c", cmd], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            wall_tab = time.perf_counter() - t1
            if r.returncode == 0 and len(r.stdout.decode().splitlines()) == n_tab:
                steady = {"value": tab_audio / wall_tab, "unit": "audio-seconds/s", "cores": 1,
                          "sample": f"{n_tab} utterances through ONE pipeline invocation (model + HCLG loaded once)"}
                # ... and on all host cores, the way a Kaldi deployment scales: one such single-load pipeline per core
                # (every logical CPU of the box, whatever this process's threads are bound to: the children get the full affinity mask.
                # Until round 5 this was capped at 64 workers on boxes with 128+ CPUs per socket.)
                n_workers = max(1, os.cpu_count() or 1)
                # ... unless the container's CPU-time quota is smaller (cgroup v2 cpu.max: on the round-6 boxes 16 CPUs' worth of time on a
                # 256-thread host: 256 workers were throttled to the rate of 16, 692 audio-s/s against 705 with 64 workers)
                quota = cpu_quota()
                if quota is not None:
                    n_workers = max(1, min(n_workers, int(quota + 0.5)))

                def all_cpus():
                    try:
                        os.sched_setaffinity(0, range(os.cpu_count() or 1))
                    except OSError:
                        pass
                t2 = time.perf_counter()
                procs = [subprocess.Popen(["bash", "-c", cmd], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, preexec_fn=all_cpus) for _ in range(n_workers)]
                outs = [p.communicate()[0] for p in procs]
                wall_all = time.perf_counter() - t2
                if all(p.returncode == 0 for p in procs) and all(len(o.decode().splitlines()) == n_tab for o in outs):
                    all_cores = {"value": n_workers * tab_audio / wall_all, "unit": "audio-seconds/s", "cores": n_workers,
                                 "sample": f"{n_workers} single-load pipelines side by side, {n_tab} utterances each (workers = min(logical CPUs, cgroup CPU quota))"}
    cpu_model = "unknown"
    try:
        for line in subprocess.run(["lscpu"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().splitlines():
            if line.startswith("Model name:"):
                cpu_model = line.split(":", 1)[1].strip()
    except OSError:
        pass
    return {"value": audio / wall, "unit": "audio-seconds/s", "cores": 1, "kind": "reference", "cpu_model": cpu_model, "host_cpus": os.cpu_count(), "cgroup_cpu_quota": cpu_quota(),
            "sample": f"{n_done} of the {len(pcms)} utterances, one reference pipeline per utterance "
                      f"({'online2-cli-nnet3-decode-faster on stdin' if streaming else 'transcribe_wav.py-style 3-process pipeline'}; model + HCLG "
                      f"re-loaded every call, as the reference does)",
            "one_load": steady, "all_cores": all_cores}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: enough for >= 2 s of timed work)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["grammar", "arpa", "mixed", "streams"], default="grammar")
    ap.add_argument("--utts", type=int, default=None, help="utterances (streams) per GPU; default = the configuration's size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-figures", action="store_true", help="time the headline only (profiling runs)")
    ap.add_argument("--stream-figures", action="store_true", help="streams: streams_per_tick and finish_latency_ms even with --no-side-figures")
    ap.add_argument("--inflight", type=int, default=None,
                    help="decode calls in flight per rank (host threads on one model; the library gives each its own decode "
                         "context): the latency-bound search of one batch overlaps the GEMMs of the next.  1 = one call at a time")
    ap.add_argument("--all-pdfs", action="store_true",
                    help="rs_decode_opts.prune_output_pdfs=0 for the headline too: evaluate the output layer for every pdf like the "
                         "reference does.  The library's default evaluates it for the pdfs that occur on HCLG arcs only (the search can "
                         "read no others; same words, same costs); a default run reports the all-pdfs figure as `reference_output_layer`")
    args = ap.parse_args()
    wl = args.workload
    # (timed steps, warm-up, calls in flight).  Headline: four calls in flight since round 4 -- with the layer GEMMs at two thirds of
    # their round-3 time a fourth call's search fits under the others' stages: 2.08 -> 1.91 ms per step (2, 3, 4, 5 in flight: 2.41,
    # 2.08, 1.91, 1.91; a model has four decode contexts)
    defaults = {"grammar": (600, 20, 4), "arpa": (45, 7, 3), "mixed": (150, 11, 3), "streams": (40, 2, 1)}[wl]
    steps = args.steps if args.steps is not None else defaults[0]
    warmup = args.warmup if args.warmup is not None else defaults[1]
    inflight = args.inflight if args.inflight is not None else defaults[2]

    # The other three configurations first, while this process has not touched the GPU yet: a second process beside an idle HIP
    # context runs the two-model batch at 11.2 ms per step instead of 7.8 (measured; the runtime time-slices the queues of the two).
    others = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and wl == "grammar" and not args.no_side_figures and not os.environ.get("RS_BENCH_NO_OTHER"):
        others = other_workloads()
    # Calls in flight use three HIP streams each; the runtime maps streams onto 4 hardware queues by default and streams that share
    # a queue wait for each other's events in order.  8 queues: headline step 2.45 -> 2.31 ms (the library sets the same default
    # when it is loaded before the HIP runtime starts: api.cc); must be in the environment before the first HIP call.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
    comm_ptr, gather_by = 0, "none (one rank)"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # RCCL on ROCm
        else:
            dist.init_process_group(backend)
        gather_by = f"torch.distributed.all_gather ({backend})"
        if backend == "nccl" and os.environ.get("RS_BENCH_GATHER", "rccl") == "rccl":
            # the library issues the ncclAllGather itself on torch's communicator (the same librccl.so.1 in this process); every
            # rank has to agree on the route, so the ranks vote
            try:
                t = torch.zeros(1, device=comm_device)
                dist.all_reduce(t)                       # the communicator exists after the first collective
                torch.cuda.synchronize()
                pg = dist.distributed_c10d._get_default_group()
                comm_ptr = int(pg._get_backend(torch.device("cuda", local_rank))._comm_ptr())
            except Exception:                            # older torch: no _comm_ptr
                comm_ptr = 0
            ok = torch.tensor([1 if comm_ptr else 0], device=comm_device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                comm_ptr = 0
            else:
                gather_by = "rs_shard_gather: one ncclAllGather per step issued by the library on torch.distributed's RCCL communicator"

    from rhasspy_speech_amd import _lib, shard
    from tests import configs
    cache = Path(tempfile.gettempdir()) / f"rs_bench_{wl}_rank{rank}"
    opts = dict(device_id=local_rank, prune_output_pdfs=0 if args.all_pdfs else 1)
    golden = None
    sharded = world > 1 or wl == "mixed"        # through rs_decode_batch_sharded: records are (global utterances, 68)
    decode_dev = None
    d_pcm = offsets = None
    # ---- the workload: models, inputs, one step, and how its records compare with the reference's goldens
    if wl in ("grammar", "arpa"):
        n_utts = args.utts or 256
        model_dir, graph_dir = (configs.build_grammar_model if wl == "grammar" else configs.build_arpa_model)(cache)
        pcms = configs.grammar_utterances(n_utts, rank) if wl == "grammar" else configs.arpa_utterances(n_utts)
        model = _lib.Model(model_dir, graph_dir, _lib.default_opts(**opts))
        model.to_device()
        models = [model]
        audio_seconds = sum(len(p) for p in pcms) / 16000.0
        if rank == 0 and n_utts == 256:
            golden = configs.load_golden("c1_grammar" if wl == "grammar" else "c2_arpa")
        n_global = n_utts * world
        if not sharded:
            d_pcm = torch.from_numpy(np.concatenate(pcms)).to(f"cuda:{local_rank}")
            offsets = np.concatenate([[0], np.cumsum([len(p) for p in pcms])]).astype(np.int64)

            def decode():
                return model.decode_batch(pcms)

            def decode_dev():
                return model.decode_batch_device(d_pcm.data_ptr(), offsets)
        else:
            # weak scaling: 256 utterances per rank; global utterance i belongs to rank i % world and is that rank's utterance i // world
            g_pcms = [pcms[i // world] if i % world == rank else None for i in range(n_global)]
            g_model = [0] * n_global

            def decode():
                rec, st, msg = _lib.decode_batch_sharded(models, g_model, g_pcms, rank, world, 0)
                if st != 0:
                    raise SystemExit(f"bench.py: rs_decode_batch_sharded failed: {msg}")
                return rec
        workload_name = (f"zamia-like-S synthetic Kaldi model (40-dim MFCC with the reference's default dither, 100-dim iVector, 7x250 TDNN, 2000 pdfs), "
                         f"{'grammar' if wl == 'grammar' else 'back-off ARPA-LM'} HCLG, {n_utts} x 3 s utterances per GPU, beam 24 / max-active 7000 / lattice-beam 8")
    elif wl == "streams":
        n_utts = args.utts or 64
        model_dir, graph_dir = configs.build_grammar_model(cache)
        pcms = configs.stream_utterances(n_utts)
        model = _lib.Model(model_dir, graph_dir, _lib.default_opts(**opts))
        model.to_device()
        models = [model]
        audio_seconds = sum(len(p) for p in pcms) / 16000.0
        if rank == 0 and n_utts == 64:
            golden = configs.load_golden("c4_streams")
        tick = 1024 * 8          # samples handed over per stream and round: 8 of the binary's 1024-sample reads
        pcms = [np.ascontiguousarray(p, dtype=np.int16) for p in pcms]
        pcm_base = np.array([p.__array_interface__["data"][0] for p in pcms], dtype=np.uintp)
        pcm_len = np.array([len(p) for p in pcms], dtype=np.int64)
        n_rounds = (max(len(p) for p in pcms) + tick - 1) // tick
        sharded = False          # streams stay on their rank; at N > 1 the ranks run replicas and the records are gathered by torch
        n_global = n_utts * world

        def decode():
            trace = bool(os.environ.get("RS_BENCH_TRACE"))
            t_open = time.perf_counter()
            streams = [_lib.Stream(model) for _ in pcms]
            t_acc = t_adv = t_py = 0.0
            t_open = time.perf_counter() - t_open
            handles = _lib.stream_handles(streams)
            for r in range(n_rounds):
                t0 = time.perf_counter()
                # one round of audio for every stream that still has some: addresses and lengths by array arithmetic, as a host
                # program in C would (per-stream Python objects cost 85 us per round, more than the library call)
                left = pcm_len - r * tick
                live = left > 0
                addrs = (pcm_base + np.uintp(2 * r * tick))[live]
                lens = np.minimum(left[live], tick).astype(np.int32)
                t1 = time.perf_counter()
                _lib.accept_streams_raw(handles[live], addrs, lens)
                t2 = time.perf_counter()
                _lib.advance_streams_raw(handles)
                t3 = time.perf_counter()
                t_py += t1 - t0; t_acc += t2 - t1; t_adv += t3 - t2
            t0 = time.perf_counter()
            out = _lib.finish_streams(streams)
            if trace:
                sys.stderr.write(f"streams step: open {t_open * 1e3:.2f} ms, slicing {t_py * 1e3:.2f}, accept {t_acc * 1e3:.2f}, advance {t_adv * 1e3:.2f}, "
                                 f"finish {(time.perf_counter() - t0) * 1e3:.2f} ({n_rounds} rounds)\n")
            return out
        workload_name = (f"zamia-like-S synthetic Kaldi model, grammar HCLG, {n_utts} concurrent 30 s streams per GPU, audio handed over round-robin in rounds of "
                         f"{tick} samples per stream (8 of the reference binary's 1024-sample reads; rs_decode_opts.stream_min_ticks at its default 16: every second "
                         f"round's rs_streams_advance does the device work of two), online2-cli-nnet3-decode-faster semantics inside: the chunk / iVector schedule "
                         f"of 1024-sample ticks, one iVector per 24-frame nnet chunk; beside it `streams_per_tick` (1024-sample rounds, every call advances) and "
                         f"`finish_latency_ms` (real-time pacing)")

        def stream_figures(ref_rec):
            """What the headline's 8-tick rounds do not show: the same 64 streams fed the way SURVEY.md section 8(d) words it --
            1024-sample ticks round-robin, every rs_streams_advance call doing its tick's work (stream_min_ticks = 1) -- and the time
            from handing over a stream's last samples to holding its words when the audio arrives in real time."""
            tm = _lib.Model(model_dir, graph_dir, _lib.default_opts(**opts, stream_min_ticks=1))
            tm.to_device()

            def fed_by_tick(m, t_samples):
                streams = [_lib.Stream(m) for _ in pcms]
                handles = _lib.stream_handles(streams)
                for r in range((int(pcm_len.max()) + t_samples - 1) // t_samples):
                    left = pcm_len - r * t_samples
                    live = left > 0
                    _lib.accept_streams_raw(handles[live], (pcm_base + np.uintp(2 * r * t_samples))[live], np.minimum(left[live], t_samples).astype(np.int32))
                    _lib.advance_streams_raw(handles)
                return _lib.finish_streams(streams)
            out = {}
            r0 = fed_by_tick(tm, 1024)
            if not np.array_equal(r0.pack(MAX_WORDS), ref_rec):
                raise SystemExit("bench.py: streams fed tick by tick decode differently from the 8-tick rounds")
            n_t = 3
            t0 = time.perf_counter()
            for _ in range(n_t):
                fed_by_tick(tm, 1024)
            secs = time.perf_counter() - t0
            n_calls = (int(pcm_len.max()) + 1023) // 1024
            out["streams_per_tick"] = {"value": audio_seconds * n_t / secs, "unit": "audio-seconds/s", "ms_per_step": 1000.0 * secs / n_t, "steps": n_t,
                                       "advance_calls_per_step": n_calls, "ms_per_advance_call": 1000.0 * secs / n_t / n_calls,
                                       "note": "1024-sample ticks round-robin over the 64 streams, rs_decode_opts.stream_min_ticks = 1 (every rs_streams_advance call does its "
                                               "tick's work: the reference binary's cadence), not paced: the rate at which ticks can be absorbed; records equal the headline's (checked)"}

            def paced_tail(m, tail_ticks=24):
                """all but the last tail_ticks ticks as fast as they go, then one tick per 64 ms of wall clock (real time); a stream whose audio
                ends with a tick is finished right behind that tick's advance -- streams that end together in one rs_streams_finish call"""
                streams = [_lib.Stream(m) for _ in pcms]
                handles = _lib.stream_handles(streams)
                n_ticks = (pcm_len + 1023) // 1024                      # per stream: the tick that holds its last sample is n_ticks - 1
                first_paced = max(int(n_ticks.min()) - tail_ticks, 0)
                for r in range(0, first_paced, 8):
                    n_s = min(8, first_paced - r) * 1024
                    _lib.accept_streams_raw(handles, pcm_base + np.uintp(2 * r * 1024), np.full(len(pcms), n_s, np.int32))
                    _lib.advance_streams_raw(handles)
                time.sleep(0.25)                                         # (the device catches up: from here on the audio arrives in real time)
                lat = np.zeros(len(pcms))
                t_start = time.perf_counter()
                for k, r in enumerate(range(first_paced, int(n_ticks.max()))):
                    wait = t_start + k * 0.064 - time.perf_counter()
                    if wait > 0:
                        time.sleep(wait)
                    left = pcm_len - r * 1024
                    live = left > 0
                    t_in = time.perf_counter()
                    _lib.accept_streams_raw(handles[live], (pcm_base + np.uintp(2 * r * 1024))[live], np.minimum(left[live], 1024).astype(np.int32))
                    _lib.advance_streams_raw(handles[live])
                    ending = np.nonzero(n_ticks - 1 == r)[0]
                    if len(ending):
                        _lib.finish_streams([streams[i] for i in ending])
                        lat[ending] = (time.perf_counter() - t_in) * 1e3
                for st in streams:
                    st.close()
                return lat
            lat = {}
            for name, m in (("stream_min_ticks_1", tm), ("stream_min_ticks_16_default", model)):
                paced_tail(m)                                            # (warm: arenas of the small advances)
                v = np.concatenate([paced_tail(m) for _ in range(2)])
                lat[name] = {"p50": float(np.percentile(v, 50)), "p99": float(np.percentile(v, 99)), "max": float(v.max()), "samples": int(v.size)}
            lat["note"] = ("ms from handing a stream's LAST 1024 samples to rs_streams_accept until rs_streams_finish has returned its words, audio arriving at real time "
                           "(one tick per 64 ms) over the last 1.5 s of 64 concurrent streams whose lengths differ by up to 0.49 s; streams that end with the same tick "
                           "are finished in one call.  With the default coalescing the finishing call also does the up to 15 ticks of device work left undone.")
            out["finish_latency_ms"] = lat
            del tm
            return out
    else:   # mixed
        n_per = (args.utts or 1024) // 2
        names, pcms = configs.mixed_utterances(n_per)
        by_name = {}
        for key, m in configs.MIXED_MODELS.items():
            md, gd = configs.build_grammar_model(Path(str(cache) + "_" + key), m["model_seed"], m["graph_seed"])
            by_name[key] = _lib.Model(md, gd, _lib.default_opts(**opts))
            by_name[key].to_device()
        models = list(by_name.values())
        model, model_dir, graph_dir = models[0], md, gd
        n_utts = n_global = len(pcms)
        audio_seconds = sum(len(p) for i, p in enumerate(pcms) if i % world == rank) / 16000.0      # this rank's share (strong split)
        if rank == 0 and n_per == 512:
            gd_, gf_ = configs.load_golden("c3_mixed_de"), configs.load_golden("c3_mixed_fr")
            golden = ([None] * n_utts, np.zeros(n_utts, np.float32), np.zeros(n_utts, np.float32))
            it = {"de_DE-like": iter(zip(*gd_)), "fr_FR-like": iter(zip(*gf_))}
            for i, nm in enumerate(names):
                golden[0][i], golden[1][i], golden[2][i] = next(it[nm])
        name_idx = {nm: k for k, nm in enumerate(by_name)}
        utt_model = [name_idx[nm] for nm in names]

        def decode():
            rec, st, msg = _lib.decode_batch_sharded(models, utt_model, pcms, rank, world, 0)
            if st != 0:
                raise SystemExit(f"bench.py: rs_decode_batch_sharded failed: {msg}")
            return rec
        workload_name = (f"two independently seeded zamia-like-S model + grammar-HCLG sets (de_DE-like / fr_FR-like), {n_utts} x ~3 s utterances "
                         f"interleaved, utterance i -> rank i % {world} (rs_decode_batch_sharded), one record gather")

    def records(res):
        """this step's records of THIS rank: sharded -> the (global, 68) array with the rank's rows filled; else rs_result_pack."""
        return res if sharded else res.pack(MAX_WORDS)

    def gather(rec):
        """the path's one exchange step: fixed-size result records to every rank (RCCL over xGMI)"""
        if world == 1:
            return rec
        if sharded and comm_ptr:
            return _lib.shard_gather(rec, local_rank, rank, world, comm_ptr)
        local = rec[rank::world] if sharded else rec
        t = torch.from_numpy(np.ascontiguousarray(local)).to(comm_device)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return torch.cat(out).cpu().numpy()

    # Host placement: this rank's threads -- the workers that copy the step's PCM into pinned staging and the thread that gathers --
    # stay on the CPUs of the GPU's NUMA node (rs_bind_host_thread; RS_BENCH_BIND=0 switches it off).  At one rank the whole box
    # is the rank's and it changes nothing measurable; at eight, every rank's 10 GB/s of staging traffic stays off the socket links.
    bound = {"cpus": 0}
    bind_on = os.environ.get("RS_BENCH_BIND", "1") != "0"

    def bind_here():
        if bind_on:
            try:
                bound["cpus"] = _lib.bind_host_thread(local_rank)
            except Exception as e:           # placement is an optimisation: a box without the sysfs entries still runs the bench
                bound["error"] = str(e)
    bind_here()
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(inflight, 1), initializer=bind_here)

    default_stagger_ms = {"grammar": 0.6, "arpa": 0.0, "mixed": 0.0, "streams": 0.0}[wl]

    def run_steps(n, fn, check_against=None):
        """n steps, at most `inflight` decode calls in flight; results are consumed (and gathered) in step order."""
        t_start = time.perf_counter()
        if os.environ.get("RS_BENCH_TRACE"):
            inner = fn

            def fn():
                t_in = time.perf_counter()
                r = inner()
                sys.stderr.write(f"  call entered at {(t_in - t_start) * 1e3:.2f} ms, returned at {(time.perf_counter() - t_start) * 1e3:.2f} ms\n")
                return r
        # `inflight` calls are outstanding at any time, the next one is submitted when the oldest has been consumed (all n submitted
        # up front, the worker threads could not start before this thread had finished submitting: ~1 ms of an idle device per run)
        # From a standing start (the driver's `--steps 20` behind a device synchronisation) the first `inflight` calls are started a
        # fraction of a step apart instead of all at once: calls that start together run the same stage at the same moment and share
        # the device stage by stage -- the first one returns after ~5 ms instead of its un-overlapped 2.6 -- where calls a stage apart
        # fall into the staggered phases the steady state has anyway (a serving process never starts four identical calls in the
        # same microsecond either).  RS_BENCH_STAGGER_MS overrides (0: all at once).
        stagger = float(os.environ.get("RS_BENCH_STAGGER_MS", default_stagger_ms)) * 1e-3

        def delayed(i):
            if stagger > 0 and i > 0:
                time.sleep(i * stagger)
            return fn()
        futures = [pool.submit(delayed, i) for i in range(min(n, inflight))] if inflight > 1 else None
        last = None
        for k in range(n):
            res = futures[k].result() if futures else fn()
            if futures and len(futures) < n:
                futures.append(pool.submit(fn))
            if os.environ.get("RS_BENCH_TRACE"):
                tm_ = res.timings() if hasattr(res, "timings") else []
                sys.stderr.write(f"step {k} done at {(time.perf_counter() - t_start) * 1e3:.2f} ms  timings {[round(x, 2) for x in tm_]}\n")
            rec = gather(records(res))
            if check_against is not None and not np.array_equal(rec, check_against):
                raise SystemExit(f"bench.py: step {k} produced different results from the first step on the same input")
            last = (res, rec)
        return last

    def timed(n, fn, check_against):
        torch.cuda.synchronize()
        t = time.perf_counter()
        run_steps(n, fn, check_against)
        torch.cuda.synchronize()
        return time.perf_counter() - t

    res, ref_rec = run_steps(1, decode)
    # rank 0's transcripts against the REFERENCE's (tests/golden/configs: oracle/_ref binaries on the same inputs)
    checked_vs_reference = None
    if golden is not None:
        mine = records(res)
        if sharded:
            idx = list(range(rank, n_global, world))
            mine = mine[rank::world]
            gold_of = (lambda i: i) if wl == "mixed" else (lambda i: i // world)
        else:
            idx = list(range(n_utts))
            gold_of = lambda i: i
        wrong = 0
        for row, i in zip(mine, idx):
            if sharded:
                ok = row[1] == 0 and list(row[3:3 + row[2]]) == list(golden[0][gold_of(i)])
            else:
                ok = row[0] == 0 and list(row[2:2 + row[1]]) == list(golden[0][gold_of(i)])
            wrong += 0 if ok else 1
        if wrong and not os.environ.get("RS_BENCH_DEBUG_UNCHECKED"):      # (timing experiments on deliberately wrong scratch builds)
            raise SystemExit(f"bench.py: {wrong} of {len(idx)} transcripts differ from the reference's (tests/golden/configs)")
        checked_vs_reference = (f"all {len(idx)} transcripts of rank 0 equal the reference's (tests/golden/configs)" if not wrong else
                                f"NOT EQUAL: {wrong} of {len(idx)} transcripts differ (RS_BENCH_DEBUG_UNCHECKED)")
    run_steps(max(warmup - 1, 0), decode, ref_rec)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, rec = run_steps(steps, decode, ref_rec)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=comm_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    def figure(n, secs, note):
        return {"value": audio_seconds * n / secs, "unit": "audio-seconds/s", "ms_per_step": 1000.0 * secs / n, "steps": n, "note": note}

    # ---- side figures (one GPU, grammar / arpa), each over the same number of steps as the headline
    side = {}
    all_pdfs_stage = None
    if world == 1 and elapsed < 1.0 and not args.no_side_figures:
        # the requested timed region is shorter than a second (the driver's --steps 20: 45 ms, most of it the pipeline of calls in
        # flight filling and draining): the same steps repeated for at least a second, beside it
        n_steady = max(int(1.2 / max(elapsed / steps, 1e-4)), steps)
        side["steady_state"] = figure(n_steady, timed(n_steady, decode, ref_rec),
                                      f"the same step, {n_steady} times back to back (>= 1 s of timed work): what a serving process sees")
    if decode_dev is not None and not args.no_side_figures:
        n_warm = max(2, 2 * inflight)      # every decode context of the model has served a call (pinned staging, arena) before the clock starts
        run_steps(n_warm, decode_dev, ref_rec)
        side["hbm_resident"] = figure(steps, timed(steps, decode_dev, ref_rec),
                                      "rs_decode_batch_device: the int16 samples are resident in HBM when the timed region starts")
        if not args.all_pdfs:
            full = _lib.Model(model_dir, graph_dir, _lib.default_opts(device_id=local_rank, prune_output_pdfs=0))
            full.to_device()

            def full_host():
                return full.decode_batch(pcms)

            def full_dev():
                return full.decode_batch_device(d_pcm.data_ptr(), offsets)
            _, full_rec = run_steps(n_warm, full_host)
            # same transcripts (the costs may differ in the last bits: the narrower layer takes another GEMM tile shape)
            if not np.array_equal(full_rec[:, :2 + MAX_WORDS], ref_rec[:, :2 + MAX_WORDS]):
                raise SystemExit("bench.py: the all-pdfs model decodes different transcripts")
            side["reference_output_layer"] = figure(steps, timed(steps, full_host, full_rec),
                                                    "host PCM -> host word ids AND rs_decode_opts.prune_output_pdfs = 0 (output layer for all pdfs): "
                                                    "the reference's own computation; identical transcripts (checked)")
            run_steps(n_warm, full_dev, full_rec)
            side["hbm_resident_all_pdfs"] = figure(steps, timed(steps, full_dev, full_rec), "rs_decode_batch_device, output layer for all pdfs")
            # its nnet stage (un-overlapped calls): EVERY launch of it runs on the split-bf16 kernels -- the stage the roofline prices
            full_stage = np.zeros(8)
            for _ in range(5):
                full_stage += np.array(full_dev().timings())
            all_pdfs_stage = (float(full_stage[3] / 5), full.describe())
            del full
    if wl == "streams" and world == 1 and (args.stream_figures or not args.no_side_figures):
        side.update(stream_figures(ref_rec))
    # ---- the n-best / lattice tail (what every rescoring or fuzzy-matching call of the Python API asks for: nbest = 5; and the
    # determinised lattice itself): device-side lattice extraction + host determinisation and n-best, utterances of a call on a
    # few host threads.  Short runs: the tail is host work, an order of magnitude slower than the 1-best step.
    if wl == "grammar" and world == 1 and not args.no_side_figures and not sharded:
        n_tail = max(6, min(steps, 24))

        def nbest5():
            return model.decode_batch(pcms, nbest=5)
        r5 = nbest5()
        if [r5.words(u, 0) for u in range(n_utts)] != [res.words(u, 0) for u in range(n_utts)]:
            raise SystemExit("bench.py: the first of the 5-best differs from the 1-best")
        run_steps(2, nbest5)
        t5 = time.perf_counter()
        run_steps(n_tail, nbest5)
        side["nbest5"] = figure(n_tail, time.perf_counter() - t5, "rs_decode_batch(nbest = 5): lattice extraction on the device, determinisation + 5-best per utterance on host threads; "
                                f"{sum(r5.num_hyps(u) for u in range(n_utts)) / n_utts:.2f} hypotheses per utterance on average")
        lat_model = _lib.Model(model_dir, graph_dir, _lib.default_opts(device_id=local_rank, prune_output_pdfs=0 if args.all_pdfs else 1, emit_lattice=1))
        lat_model.to_device()

        def with_lattice():
            return lat_model.decode_batch(pcms, nbest=1)
        run_steps(2, with_lattice)
        tl = time.perf_counter()
        run_steps(n_tail, with_lattice)
        side["emit_lattice"] = figure(n_tail, time.perf_counter() - tl, "rs_decode_opts.emit_lattice = 1: every utterance's determinised CompactLattice kept with the result")
        del lat_model
        # --frame-subsampling-factor=3 (how a chain model is meant to be decoded; rhasspy leaves the factor at 1): the decoder sees every
        # third frame and the layers only those frames read run on a third of the rows.  Another search on other frames: no golden to
        # check against here (the parity cases of tests/cases.py hold the reference's results for it)
        fsf_model = _lib.Model(model_dir, graph_dir, _lib.default_opts(device_id=local_rank, prune_output_pdfs=0 if args.all_pdfs else 1, frame_subsampling_factor=3))
        fsf_model.to_device()

        def fsf3():
            return fsf_model.decode_batch(pcms)
        n_fsf = max(steps, 60)
        run_steps(max(2, 2 * inflight), fsf3)
        tf = time.perf_counter()
        run_steps(n_fsf, fsf3)
        side["frame_subsampling_factor_3"] = figure(n_fsf, time.perf_counter() - tf, "rs_decode_opts.frame_subsampling_factor = 3 on the same model and batch: "
                                                    "100 decoder frames per utterance instead of 298 (not the reference's configuration for this metric; rhasspy runs factor 1)")
        del fsf_model
        # The second acoustic model of SURVEY.md section 8(d), "tdnn-f-like", at full size (tests/configs.py: TDNNF_SPEC -- TdnnComponent
        # bottlenecks 1024 / 128 with 0.66-scaled residual sums, 2000 pdfs) on the same batch and graph shape; its first 64 utterances are
        # checked against the reference's transcripts (tests/golden/configs/c5_tdnnf.npz), all of them in tests/test_gpu_configs.py.
        f_md, f_gd = configs.build_tdnnf_model(Path(tempfile.gettempdir()) / f"rs_bench_tdnnf_rank{rank}")
        f_model = _lib.Model(f_md, f_gd, _lib.default_opts(device_id=local_rank, prune_output_pdfs=0 if args.all_pdfs else 1))
        f_model.to_device()

        def tdnnf():
            return f_model.decode_batch(pcms)
        rf = tdnnf()
        f_gold = configs.load_golden("c5_tdnnf")[0]
        n_chk = min(len(f_gold), n_utts) if n_utts == 256 and rank == 0 else 0
        if [rf.words(u) for u in range(n_chk)] != f_gold[:n_chk]:
            raise SystemExit("bench.py: the factorised-TDNN model's transcripts differ from the reference's")
        n_f = max(6, min(steps, 40))
        run_steps(max(2, 2 * inflight), tdnnf)
        tf = time.perf_counter()
        run_steps(n_f, tdnnf)
        side["tdnnf_model"] = figure(n_f, time.perf_counter() - tf, "the same batch on the full-size factorised TDNN (zamia-like-F: 1024 / 128 bottleneck TdnnComponents, "
                                     f"residual sums, 2000 pdfs); {n_chk} transcripts checked against the reference's")
        fst = np.zeros(8)
        for _ in range(3):
            fst += np.array(tdnnf().timings())
        side["tdnnf_model"]["nnet_stage_ms"] = float(fst[3] / 3)
        side["tdnnf_model"]["nnet_tflops"] = nnet_flops_per_row(f_model.describe()) * sum(1 + (len(p) - 400) // 160 for p in pcms) / (fst[3] / 3 * 1e-3) / 1e12
        side["tdnnf_model"]["layer_gemm"] = [l for l in f_model.describe().splitlines() if l.startswith("layer_gemm")][0][:110]
        del f_model
    # Stage times and the roofline come from un-overlapped calls made right after the timed region: same process, same
    # buffers, one call at a time.
    stage, counters, n_iso = np.zeros(8), np.zeros(8), 0
    staged = not sharded and world == 1
    if staged:
        n_iso = 5 if wl != "streams" else 2
        iso = decode_dev if decode_dev is not None else decode
        for _ in range(n_iso):
            r1 = iso()
            stage += np.array(r1.timings())
        stage /= n_iso
        for u in range(n_utts):
            counters += np.array(r1.counters(u), dtype=np.float64)

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / steps
        value = (world if wl != "mixed" else 1) * audio_seconds * steps / elapsed
        if wl == "mixed":
            value = sum(len(p) for p in pcms) / 16000.0 * steps / elapsed     # the whole 1024-utterance batch per step, all ranks together
        desc = model.describe()
        frames = sum(1 + (len(p) - 400) // 160 for p in pcms) if wl != "mixed" else 0
        flops = nnet_flops_per_row(desc) * frames       # algorithmic: the real frames only (halo rows are overhead)
        n_gemm = sum(1 for l in desc.splitlines() if l.startswith("op: gemm"))
        out = {
            "metric": "audio-seconds decoded/sec (RTF^-1) en_US-zamia grammar HCLG", "value": value, "unit": "audio-seconds/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if wl != "mixed" else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "FP32 results (log-likelihoods within 1e-4 of the reference's); the wide layer GEMMs multiply 2-way fp16 splits of the FP32 operands (22 significand bits, weights scaled per output column) on the fp16 matrix cores, three products per element, and accumulate in FP32; activations beyond fp16's range send the call to the exact-FP32 kernels",
            "config": {"workload": workload_name, "utts_per_gpu": n_utts if wl != "mixed" else n_utts // world, "parallelism": f"utterance-sharded x{world}",
                       "output_layer": "all pdfs (--all-pdfs)" if args.all_pdfs else "the pdfs that occur on HCLG arcs (library default)",
                       "calls_in_flight": inflight,
                       "inputs": "int16 PCM in pageable host memory -> word ids in host memory (SURVEY 8(d)'s timed region; PCIe-inclusive)",
                       "entry_point": ("rs_streams_accept / rs_streams_advance / rs_streams_finish" if wl == "streams" else
                                       "rs_decode_batch_sharded" if sharded else "rs_decode_batch"),
                       "record_gather": gather_by,
                       "host_affinity": (f"threads bound to the {bound['cpus']} CPUs local to the GPU (rs_bind_host_thread)" if bound["cpus"] else
                                         "not bound" + (f" ({bound['error']})" if "error" in bound else ""))},
            "timed_seconds": elapsed,
            "results_checked": "every step's result records equal the first step's (same input)" + (f"; {checked_vs_reference}" if checked_vs_reference else ""),
        }
        out.update(side)
        if staged:
            # decoder algorithmic bytes: arcs examined x (16 B arc + 4 B loglike), token insertions x 16 B,
            # tokens alive x 16 B token record
            dec_bytes = counters[1] * 20.0 + counters[2] * 16.0 + counters[3] * 16.0
            split_bf16 = os.environ.get("RS_GEMM_B3", "1") != "0"
            peak = 2500.0 / 3.0 if split_bf16 else 157.3          # three fp16 MFMAs per FP32 product (rounds 1-3: six bf16 ones, 417)
            nnet_ms, roof_on = float(stage[3]), "this run's model"
            if all_pdfs_stage is not None:
                # the default model evaluates its pruned output layer (362 of 2000 columns) on the exact-FP32 kernel; the roofline
                # is priced on the all-pdfs model of the same run, whose launches all run on the split-bf16 kernels
                # (round 1's definition of the stage)
                nnet_ms, roof_on = all_pdfs_stage[0], "the all-pdfs model of the same run (every launch of the stage on the split-bf16 kernels)"
                flops = nnet_flops_per_row(all_pdfs_stage[1]) * frames
            achieved = flops / (nnet_ms * 1e-3) / 1e12 if nnet_ms > 0 else 0.0
            traffic, traffic_from = pmc_traffic(wl, "GemmKernelB3" if split_bf16 else "GemmKernel")
            roof_mfma = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_from": traffic_from,
                         "kernel": (f"GemmKernelB3 family (B3 first layer, B3J image-fed layers: FP32 operands split into 2 fp16 parts, 3 fp16 MFMAs per product, FP32 accumulate), "
                                    f"{n_gemm} launches per step (the nnet stage)") if split_bf16 else f"GemmKernel, {n_gemm} launches per step",
                         "launches": n_gemm, "avg_launch_ms": nnet_ms / n_gemm, "flops_per_launch": flops / n_gemm,
                         "stage_ms": nnet_ms, "measured_on": roof_on, "frac_of_fp32_mfma_peak": achieved / 157.3,
                         "mfma_issue_frac": 3.0 * achieved / 2500.0 if split_bf16 else achieved / 157.3,
                         "peak_note": "2.5 PFLOP/s dense fp16 / 3 MFMAs per FP32 product (MI355X_MICROARCH.md); rounds 1-3 priced 6 bf16 MFMAs per product against 417"}
            # (the search kernel the workload runs: the register-resident one on the grammar graph, the live-state-table one on the ARPA graph)
            dtraffic, dtraffic_from = pmc_traffic(wl, "LiveDecodeKernel" if wl == "arpa" else "RegDecodeKernel")
            roof_dec = {"bound": "hbm", "achieved": dec_bytes / (stage[4] * 1e-3) / 1e9 if stage[4] > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                        "frac": dec_bytes / (stage[4] * 1e-3) / 1e9 / 8000.0 if stage[4] > 0 else 0.0, "traffic": dtraffic, "traffic_from": dtraffic_from,
                        "algorithmic_bytes": dec_bytes,
                        "kernel": ("LiveDecodeKernel" if wl == "arpa" else "RegDecodeKernel") + ": beam search (one workgroup per utterance, T sequential steps: bound by dependent instruction chains, not by bytes)", "stage_ms": float(stage[4])}
            # The line's roofline is that of the stage that bounds the overlapped step.  On the grammar graph that is the acoustic
            # model: with calls in flight the step is the sum of the device-filling kernels (features, iVector, layer GEMMs) and the
            # search -- one persistent workgroup per utterance, T dependent frames, its graph in registers -- runs under the next
            # calls' GEMMs; its figure is a latency (cycles per frame), reported in other_roofline.  On the ARPA graph the search is
            # the step.
            roof_dec["cycles_per_frame_at_2.4GHz"] = float(stage[4]) * 1e-3 * 2.4e9 / max(frames / max(n_utts, 1), 1)
            roofline = roof_mfma if (wl != "arpa" or stage[3] >= stage[4]) else roof_dec
            out["stages_from"] = f"{n_iso} un-overlapped calls after the timed region (samples resident in HBM)"
            out["roofline"] = roofline
            out["stages_ms"] = {"mfcc": float(stage[1]), "ivector": float(stage[2]), "nnet": float(stage[3]), "decode": float(stage[4]),
                                "d2h+host": float(stage[5]), "total_call": float(stage[6])}
            out["other_roofline"] = roof_dec if roofline is roof_mfma else roof_mfma
        else:
            out["roofline"] = None
            out["roofline_note"] = ("the mixed batch runs the grammar workload's kernels on two models side by side; see --workload grammar for the roofline"
                                    if wl == "mixed" else "stage times and the roofline are measured at N = 1 (un-overlapped calls)")
        if others is not None:
            out["other_workloads"] = others
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model_dir, graph_dir, pcms if wl != "mixed" else [p for nm, p in zip(names, pcms) if nm == list(by_name)[-1]],
                                               streaming=(wl == "streams"))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
