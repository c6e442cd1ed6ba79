"""csrc/ref_field.hpp (the reference-field mode's host side) writes libstdc++'s heap algorithms out by hand; tests/ref_field_check.cpp
holds that against std::priority_queue itself — the pop order of equal distances, which is what the reference's distance field
depends on (grid_mapper.cpp:399-433), and whole fields over scans with insertions, erasures, sharing and a resampling.  Host code
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
