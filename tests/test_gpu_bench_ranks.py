"""`bench.py --gpus 2` rehearsed on ONE GPU (PCC_BENCH_REHEARSE=1: both ranks on device 0, collectives over gloo): the ranks must
meet at the same collectives and rank 0 must print its line although it alone runs the event / strict / fp32 passes afterwards.
(Round 4 found a barrier inside a rank-0-only pass this way: with RCCL it would have met the other ranks' all_reduce.)  A
control-flow test -- the line it prints is marked as a rehearsal and is never a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_control_flow_on_one_gpu():
    env = dict(os.environ, PCC_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                      # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "rehearsal" in line
    ranks = line["config"]["per_rank"]
    assert [r["rank"] for r in ranks] == [0, 1]
    # disjoint core shares, both ranks timed the same barrier-to-barrier interval (they share the GPU, so within 20 %)
    assert ranks[0]["first_core"] != ranks[1]["first_core"]
    assert abs(ranks[0]["ms_per_step"] - ranks[1]["ms_per_step"]) <= 0.2 * ranks[0]["ms_per_step"]
    assert line["config"]["frames_per_step"] == 2 and line["value"] > 0
    assert line["config"]["ms_per_step_strict"] is not None        # the rank-0-only passes ran after the collectives
