"""CPU: the multi-GPU plumbing of bench.py (replicas only) with world_size 2 over gloo: `--gpus 2` without
a launcher spawns two rank processes, under a launcher environment the ranks are taken from it, a mismatch between
--gpus and WORLD_SIZE fails loudly. No kernel runs here (`--plumbing-only` measures nothing and says so)."""
import json
import os
import socket
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_gpus_flag_spawns_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--plumbing-only"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 prints ONE line, the other rank nothing
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["elapsed_max_s"] >= 0


def test_launcher_environment_is_used():
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(_clean_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0", "--plumbing-only"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1000:] for o in outs]
    json_lines = [[ln for ln in o[0].splitlines() if ln.startswith("{")] for o in outs]
    assert len(json_lines[0]) == 1 and json.loads(json_lines[0][0])["n_gpus"] == 2
    assert json_lines[1] == []  # only rank 0 reports


def test_gpus_must_match_world_size():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--plumbing-only"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "does not match WORLD_SIZE" in p.stderr
