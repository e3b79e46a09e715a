"""`bench.py` under the driver's N > 1 launch line (`-m gpu`).

The driver starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
bench.py --gpus N ...`, one rank per GPU over RCCL.  A test box has ONE GPU and RCCL refuses two ranks on a device, so the
same launch line is run here with bench.py's test hook RS_BENCH_BACKEND=gloo: two ranks share the GPU, the barrier / MAX
over ranks / record gather travel through host memory, everything else (utterance sharding i mod N, per-rank model, the
transcripts of rank 0 checked against the reference's goldens inside the run, one JSON line from rank 0) is the N > 1 path.
"""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("workload,scaling,units", [("grammar", "weak", 2 * 256 * 3.0), ("mixed", "strong", None)])
def test_bench_two_ranks(workload, scaling, units):
    env = dict(os.environ, RS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--workload", workload, "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 2
    assert line["scaling"] == scaling and line["higher_is_better"] is True
    # value = the audio all ranks decoded per step / the slowest rank's time per step
    per_step = line["value"] * line["ms_per_step"] * 1e-3
    if units is not None:       # 256 x 3 s per rank
        assert "roofline" in line
        assert per_step == pytest.approx(units, rel=1e-6)
        assert line["config"]["utts_per_gpu"] == 256
    else:                       # the one 1024-utterance batch (~3 s each), split over the ranks
        assert 1024 * 2.5 < per_step < 1024 * 3.5
        assert line["config"]["utts_per_gpu"] == 512
    assert "equal the reference's" in line["results_checked"]
