"""One bounded randomized sweep inside the suite (round-4 review: the fuzz tools were only ever run by hand, their logs in untracked
scratch): tools/fuzz_parity.py — random sizes, gains, poses near +-pi, sensor models that make the run resample by itself, scan
matcher, ICP failures, band loop — 20 MPPI ticks-pairs + 40 RBPF runs + 10 pipelined replays from a FIXED seed, each against the
oracle with the suite's own assertions.  About 40 s.  The summary goes to gpurun_out/fuzz_in_suite.txt (scratch: what a gpurun call
carries back); the committed copy under profiles/ is refreshed from it by hand, never by the suite."""
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bounded_fuzz_sweep_of_both_paths_against_the_oracle(gpu_pkg):
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "20", "40", "2025", "", "10"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    dt = time.perf_counter() - t0
    lines = [l for l in r.stdout.splitlines() if l.startswith(("mppi:", "rbpf:", "batch:", "[FAIL]"))]
    summary = "\n".join([f"python tools/fuzz_parity.py 20 40 2025 '' 10   ({dt:.1f} s, exit code {r.returncode})"] + lines) + "\n"
    print("\n" + summary)
    try:   # (scratch only: the suite does not touch tracked files)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "fuzz_in_suite.txt"), "w") as f:
            f.write(summary)
    except OSError:
        pass
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert any(l.startswith("mppi: 20 cases done, failures so far 0") for l in lines)
    assert any(l.startswith("rbpf: 40 cases done, failures so far 0") for l in lines)
    assert any(l.startswith("batch: 10 cases done, failures so far 0") for l in lines)


def test_bounded_fuzz_sweep_of_the_reference_field_mode_against_the_oracle(gpu_pkg):
    """Round 6: the lazy brushfire under random reach, sharing, rooms that change in mid-run (passes resumed, proposals rerun),
    forced resamplings, ICP failures, empty scans and whole-field exports (lineages replayed): tools/fuzz_reffield.py, 24 cases from a
    fixed seed, every stage of every scan and every particle's field and log-odds against the oracle's filter, nothing injected."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_reffield.py"), "24", "2026"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("reffield:", "[FAIL]"))]
    print("\n" + "\n".join(lines[-3:]))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = lines[-1]
    assert last.startswith("reffield: 24 cases done, failures so far 0"), last
    import ast
    tot = ast.literal_eval(last.split("; ", 1)[1])
    assert tot["states_resumed"] > 0 and tot["proposals_rerun"] > 0 and tot["lineages_replayed"] > 0, tot   # the sweep really goes down those paths
