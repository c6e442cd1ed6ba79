"""bench.py's N > 1 legs run under a watchdog: a leg that never returns (a hung exchange) must not take the headline along.
The blocked call is played by a sleep; the process has to print the headline line with the reason and leave with status 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, time
sys.path.insert(0, {root!r})
import bench
line = {{"metric": "MPPI rollouts/s", "value": 1.0, "n_gpus": 2}}
wd = bench.LegWatchdog({rank}, line if {rank} == 0 else None, 0.3)
{body}
"""


def run(rank, body):
    code = SCRIPT.format(root=ROOT, rank=rank, body=body)
    env = dict(os.environ, TBNAV_BENCH_DETAIL=os.path.join(os.environ.get("TMPDIR", "/tmp"), f"tbnav_bench_detail_{os.getpid()}.json"))
    try:
        return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    finally:
        if os.path.exists(env["TBNAV_BENCH_DETAIL"]):
            os.remove(env["TBNAV_BENCH_DETAIL"])


def test_a_leg_that_hangs_leaves_the_headline_line():
    r = run(0, "time.sleep(60)\nprint('never')")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "never" not in r.stdout
    out = json.loads(lines[0])
    assert out["value"] == 1.0 and out["multi_gpu_legs"].startswith("skipped: not finished")


def test_other_ranks_leave_quietly():
    r = run(1, "time.sleep(60)\nprint('never')")
    assert r.returncode == 0 and r.stdout.strip() == ""


def test_a_failing_leg_expires_at_once_with_the_reason():
    r = run(0, "wd.expire_now('failed on rank 0: boom')")
    assert r.returncode == 0
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["multi_gpu_legs"] == "skipped: failed on rank 0: boom"


def test_a_cancelled_watchdog_does_nothing():
    r = run(0, "wd.cancel()\ntime.sleep(0.8)\nprint('done')")
    assert r.returncode == 0 and r.stdout.strip() == "done"
