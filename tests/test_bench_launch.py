"""bench.py's own N > 1 entry (`python bench.py --gpus N` without a launcher): the parent stays a plain launcher that
never imports torch / touches HIP, starts N rank processes as children (no exec) which get RANK / LOCAL_RANK /
WORLD_SIZE and a 127.0.0.1 rendezvous, and rank 0's JSON line comes through with the process group's rank count.
CPU only: BENCH_REHEARSAL=cpu (gloo) and the `launchcheck` workload, which makes no solver call."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra_env=None, extra_args=()):
    env = dict(os.environ, BENCH_REHEARSAL="cpu")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "launchcheck", *extra_args],
                          env=env, capture_output=True, text=True, timeout=600)


def test_strong_scaling_flag_deals_one_grid_to_the_ranks():
    """--total-points: the SAME grid for every rank count, contiguous blocks, every point owned once (the blocks the grid leg
    would run; the leg itself needs a GPU: test_strong_scaling_grid_leg_on_one_gpu)."""
    r = _run(2, extra_args=("--total-points", "4097"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert [x["points"] for x in d["ranks"]] == [[0, 2049], [2049, 4097]]


def test_self_launch_two_ranks():
    r = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "torch imported in the launcher: False" in r.stderr          # the parent never initialises HIP
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                               # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collective_backend"] == "gloo"
    ranks = d["ranks"]
    assert [x["RANK"] for x in ranks] == ["0", "1"] and [x["LOCAL_RANK"] for x in ranks] == ["0", "1"]
    assert all(x["WORLD_SIZE"] == "2" and x["MASTER_ADDR"] == "127.0.0.1" for x in ranks)
    assert len({x["pid"] for x in ranks}) == 2                            # two child processes ...
    assert len({x["ppid"] for x in ranks}) == 1                           # ... of one launcher
    assert abs(d["max_over_ranks_check"] - 0.002) < 1e-12                 # MAX-reduce over the group works


def test_failing_children_exit_nonzero():
    # a world size the ranks do not agree with: every child refuses, the launcher passes the failure on
    env = dict(os.environ, BENCH_REHEARSAL="cpu", WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "launchcheck"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


import pytest


@pytest.mark.gpu
def test_two_rank_rehearsal_on_one_gpu_goes_through_the_launcher():
    """`python bench.py --gpus 2 --workload grid` with no launcher around it, on a one-GPU box: BENCH_REHEARSAL=1 lets
    both ranks share cuda:0 and rendezvous over gloo.  The parent launches, the ranks run the grid leg (a small share),
    rank 0 prints the configs[3] line with the group's rank count, whole-job steps/s and the MAX-over-ranks step time."""
    env = dict(os.environ, BENCH_REHEARSAL="1", BENCH_GRID_POINTS="64", BENCH_GRID_CHAINS="8")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "grid", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "torch imported in the launcher: False" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["collective_backend"] == "gloo"
    assert d["scaling"] == "weak" and d["unit"] == "steps/s" and d["config"]["chains_per_gpu"] == 512
    assert d["value"] > 0 and abs(d["value"] - 2 * 512 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
    assert d["ms_per_step_max_over_ranks"] >= d["ms_per_step"] * 0.999
    assert d["roofline"]["live_kernel_ms"] > 0


@pytest.mark.gpu
def test_strong_scaling_grid_leg_on_one_gpu():
    """`bench.py --workload grid --total-points N`: one grid whatever the rank count - on one GPU all N points, as two
    rehearsal ranks N/2 each; scaling "strong", value = Metropolis steps/s of the whole grid."""
    env = dict(os.environ, BENCH_GRID_CHAINS="8")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = {}
    for n in (1, 2):
        e = dict(env, BENCH_REHEARSAL="1") if n > 1 else env
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "grid", "--total-points", "128",
                            "--steps", "3", "--warmup", "1"], env=e, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        out[n] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    for n, d in out.items():
        assert d["scaling"] == "strong" and d["n_gpus"] == n and d["config"]["total_points"] == 128
        assert d["config"]["points_per_gpu"] == 128 // n and d["config"]["chains_per_gpu"] == 128 // n * 8
        assert abs(d["value"] - 128 * 8 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
