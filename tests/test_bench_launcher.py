"""not-gpu: `python bench.py --gpus N` must start its own ranks when no launcher did (the driver may call it either way).
The --rehearse mode replaces the train step by its only collective (one flat fp32 all-reduce) so that launcher,
rendezvous, barrier / MAX-over-ranks protocol and the single JSON line can be exercised on CPU ranks over gloo."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["AVSEP_DP_BACKEND"] = "gloo"
    env["OMP_NUM_THREADS"] = "2"
    return env


def _one_json_line(out):
    lines = [l for l in out.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse",
                        "--rehearse-elems", "100000"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _one_json_line(r.stdout)
    assert j["rehearsal"] is True and j["n_gpus"] == 2 and j["n_ranks_seen"] == 2 and j["backend"] == "gloo"
    assert j["allreduce_bytes_per_step"] == 400000 and j["value"] is None


def test_bench_under_external_launcher():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "1", "--warmup", "1", "--rehearse",
           "--rehearse-elems", "1000"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_json_line(r.stdout)["n_ranks_seen"] == 2


def test_bench_world_size_mismatch_is_an_error():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
