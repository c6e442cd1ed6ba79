"""csrc/ref_field.hpp (the reference-field mode's host side) writes libstdc++'s heap algorithms out by hand; tests/ref_field_check.cpp
holds that against std::priority_queue itself — the pop order of equal distances, which is what the reference's distance field
depends on (grid_mapper.cpp:399-433), and whole fields over scans with insertions, erasures, sharing and a resampling; and (round 6)
the LAZY passes — stopped a few cells out, resumed when a lookup lands on an unwritten cell, replayed where a stale cell is wanted —
against the eager brushfire: every written cell at every moment, whole fields after completion, sorted / reverse / history insert
orders, erase events, and the device-slot journal (plan_flush) applied to simulated slots.  Host code
only: runs without a GPU (the GPU suite holds the same fields against the oracle's, pinned to the compiled reference)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hand_written_heap_is_std_priority_queue(tmp_path):
    exe = tmp_path / "ref_field_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "ros-turtlebot-navigation_amd", "csrc"),
                    os.path.join(ROOT, "tests", "ref_field_check.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "heap order: 40 sequences of 20000 operations identical" in r.stdout and "fields: 5 maps x several scans identical" in r.stdout
    # (round 6) the lazy brushfire: truncated, resumed and replayed passes, and the journal that drives the device's field slots
    assert "lazy fields: 28 runs (truncated, resumed, replayed; journal-driven device slots) identical" in r.stdout, r.stdout
