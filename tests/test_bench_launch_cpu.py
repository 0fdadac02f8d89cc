"""CPU test (-m "not gpu") of bench.py's launcher path: `python bench.py --gpus 2` with no launcher around it must
start the ranks itself (python -m torch.distributed.run as a CHILD, never exec), and relay exactly one JSON line.
--stub swaps the HIP step for a stand-in so that the plumbing (rendezvous on 127.0.0.1, barrier, max over ranks,
rank-0 line, weak / strong batch split) runs on gloo ranks here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "3", "--warmup", "1", *flags],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling,per_rank", [("weak", 8), ("strong", 4), (None, 4)])
def test_self_launch_two_ranks(scaling, per_rank):
    """the default is BASELINE config 4's shape: a GLOBAL batch of 8 frames over the ranks (strong scaling)"""
    r = _run("--gpus", "2", *(("--scaling", scaling) if scaling else ()))
    scaling = scaling or "strong"
    assert r["n_gpus"] == 2 and r["ranks"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert r["scaling"] == scaling and r["config"]["frames_per_gpu"] == per_rank
    assert r["config"]["global_batch"] == 2 * per_rank and r["value"] > 0


def test_single_rank_needs_no_launcher():
    r = _run()
    assert r["n_gpus"] == 1 and r["config"]["frames_per_gpu"] == 8
