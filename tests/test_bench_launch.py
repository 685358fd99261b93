"""bench.py's own N > 1 entry (`python bench.py --gpus N` without a launcher): the parent stays a plain launcher that
never imports torch / touches HIP, starts N rank processes as children (no exec) which get RANK / LOCAL_RANK /
WORLD_SIZE and a 127.0.0.1 rendezvous, and rank 0's JSON line comes through with the process group's rank count.
CPU only: BENCH_REHEARSAL=cpu (gloo) and the `launchcheck` workload, which makes no solver call."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra_env=None):
    env = dict(os.environ, BENCH_REHEARSAL="cpu")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "launchcheck"],
                          env=env, capture_output=True, text=True, timeout=600)


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
