"""The measurement contract's plumbing, on the CPU: what bench.py / bench_rbpf.py read from profiles/ (bench_profiles.py) is
what tools/profile_round.sh's summarisers write (profiles/summarize_rocpd.py, tools/pmc_summary.py -> tools/assemble_profiles.py),
and the committed files of the newest round carry a row for every kernel a bench line points at (round-3 review, weak 12)."""
import csv
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_profiles as bp  # noqa: E402


def test_kernel_stats_table_written_by_the_summariser_is_what_the_bench_parses(tmp_path):
    db = tmp_path / "run_results.db"
    c = sqlite3.connect(db)
    c.execute("create table kernels (name text, grid_x int, grid_y int, grid_z int, workgroup_x int, duration int, vgpr_count int, "
              "sgpr_count int, lds_size int, scratch_size int)")
    rows = [("void (anonymous namespace)::mppi_rollout_fused<2, 8, 1, true>((anonymous namespace)::RolloutArgs, double const*)", 65536, 1, 1, 512, 5000),
            ("void (anonymous namespace)::mppi_rollout_fused<2, 8, 1, true>((anonymous namespace)::RolloutArgs, double const*)", 65536, 1, 1, 512, 6000),
            ("void tbnav_mk::mppi_rollout_fused<2, 8, 1, 1>(tbnav_mk::RolloutArgs, double const*)", 65536, 1, 1, 512, 7000),
            ("void tbnav_rk::rbpf_raycast_box<512>(tbnav_rk::ScanC)", 512512, 1, 1, 512, 48000),
            ("void (anonymous namespace)::rbpf_raycast_box<512>((anonymous namespace)::ScanC)", 512000, 1, 1, 512, 50000),
            ("mppi_partials(int, int)", 8192, 100, 1, 256, 26000)]
    c.executemany("insert into kernels values (?,?,?,?,?,?,28,80,0,0)", rows)
    c.commit(); c.close()
    md = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "summarize_rocpd.py"), str(db)], capture_output=True, text=True, check=True).stdout
    path = tmp_path / "r99_kernel_stats.md"
    path.write_text(md)
    parsed, _ = bp.kernel_stats_rows(str(path))
    by = {(r["kernel"], r["grid_threads"]): r for r in parsed}
    assert by[("mppi_rollout_fused<2, 8, 1, 1>", 65536)]["calls"] == 1   # (round 5: the MPPI kernels live in namespace tbnav_mk)
    assert by[("mppi_rollout_fused<2, 8, 1, true>", 65536)]["calls"] == 2 and abs(by[("mppi_rollout_fused<2, 8, 1, true>", 65536)]["avg_us"] - 5.5) < 1e-9
    assert by[("rbpf_raycast_box<512>", 512512)]["avg_us"] == 48.0 and by[("rbpf_raycast_box<512>", 512000)]["avg_us"] == 50.0
    assert by[("mppi_partials", 819200)]["wg"] == 256   # a two-dimensional grid: threads multiply


def test_pmc_summary_output_is_what_the_bench_parses(tmp_path):
    g = tmp_path / "gpurun_out"
    g.mkdir()
    for counter, kb in (("FETCH_SIZE", 100.0), ("WRITE_SIZE", 40.0)):
        with open(g / f"pmc_t_{counter}.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
            for _ in range(4):
                w.writerow(["void (anonymous namespace)::mppi_rollout_prefix<1>((anonymous namespace)::RolloutArgs)", counter, kb])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "t", "1"], capture_output=True, text=True, check=True, cwd=tmp_path).stdout
    row = json.loads(out)["mppi_rollout_prefix<1>"]
    assert {"launches", "read_bytes", "write_bytes", "hbm_bytes"} <= set(row)
    assert row["hbm_bytes"] == int(100.0 * 1024 * 2 + 40.0 * 1024)   # gfx950: FETCH_SIZE counts a 128-B request as 64 B


def test_committed_profiles_have_a_row_for_every_kernel_the_bench_lines_point_at():
    rows, src = bp.kernel_stats_rows()
    assert src and rows, "profiles/<round>_kernel_stats.md is missing"
    # (round 6: the driver's command no longer runs the large-K leg, so its kernels are in their workload's own table only)
    for kernel, grid, wl in (("mppi_rollout_fused<2, 8, 1, 2>", None, None), ("mppi_rollout_prefix<1>", 65536, "mppi_K65536_T100"), ("mppi_partials", None, "mppi_K65536_T100")):
        r = bp.rocprof_row(kernel, grid, wl)
        assert r is not None and r["avg_us"] > 0 and r["source"].startswith("profiles/"), kernel
    # the map update's instantiation is chosen per launch (launch_raycast): the committed bench line names the one that ran
    # (round 6: the line is the compact record the driver parses — bench.compact_line — and carries the same names)
    line_path = os.path.join(ROOT, "profiles", src.split("/")[-1].replace("kernel_stats.md", "bench_line.json"))
    with open(line_path) as f:
        text = f.read().strip().splitlines()[-1]
    assert len(text) < 6144, len(text)
    line = json.loads(text)
    assert line["roofline"]["kernel"] == "mppi_rollout_fused<2, 8, 1, 2>"   # the headline tick draws with the fp64 sampler (the reference's width)
    assert line["roofline"]["frac_rocprof"] is not None and line["roofline"]["traffic"] is not None and line["cpu_baseline"]["value"] > 0
    assert line["rbpf"]["modes"]["reference_equal"]["particle_updates_per_s"] >= 1e5   # bars (1) and (2) in the same mode, on the committed run
    k_raycast = line["rbpf"]["roofline"]["kernel"]
    assert k_raycast.startswith("rbpf_raycast_box<"), k_raycast
    r = bp.rocprof_row(k_raycast, 512 * 1001, "rbpf_N1000_k50_400x400_plain_scans_only")
    assert r and "plain_scans_only" in r["source"] and r["median_us"] and r["median_us"] <= r["avg_us"] * 1.5, (k_raycast, r)
    assert bp.rocprof_row("mppi_rollout_fused<2, 8, 1, 0>") != bp.rocprof_row("mppi_rollout_fused<2, 8, 1, 2>")   # never another instantiation's row
    assert bp.rocprof_row("no_such_kernel<1>") is None
    for wl, kernel in (("mppi_K1024_T50", "mppi_rollout_fused<2, 8, 1, 2>"), ("mppi_K65536_T100", "mppi_rollout_prefix<1>"),
                       ("rbpf_N1000_k50_400x400_plain_scans_only", k_raycast)):
        p = bp.pmc_row(wl, kernel)
        assert p is not None and p["hbm_bytes"] > 0 and wl in p["source"], (wl, kernel)
    assert bp.pmc_row("mppi_K1024_T50", "mppi_rollout_fused<2, 8, 1, 0>") is None


def test_every_baseline_shape_has_a_roofline_whose_frac_follows_from_one_named_row_each():
    """Round-4 review, item 1: a `frac` for every configs[*] shape, recomputable from ONE named row of a kernel-stats table (its
    AVERAGE) and ONE named entry of the traffic file — here recomputed from the committed run's full record (round 6: bench.py --detail
    writes everything it measured to bench_detail.json; the last stdout line is the compact record) and the committed profiles."""
    with open(os.path.join(ROOT, "profiles", "r06_bench_detail.json")) as f:
        line = json.load(f)
    objs = {"configs[1] K=1024,T=50": line["roofline"], "configs[3] K=65536,T=100": line["roofline_large"],
            "configs[3]/8 K=8192,T=100": line["configs3_shard_one_gpu"]["roofline"], "configs[2] bench room": line["rbpf"]["roofline"],
            "configs[2] SURVEY room": line["rbpf"]["survey_room"]["roofline"],
            "configs[4]/8 N=12500": line["rbpf"]["configs4_shard_one_gpu"]["roofline_leg"]["roofline"]}
    for name, o in objs.items():
        assert o["rocprof"] is not None and o["frac_rocprof"] is not None and o["traffic"] is not None, name
        assert o["frac_rocprof_of"] == "avg_us", name
        src, row = o["rocprof"]["source"], o["rocprof"]["row"]
        kernel, grid = [x.strip() for x in row.split("|")]
        rows, _ = bp.kernel_stats_rows(os.path.join(ROOT, src))
        hit = [r for r in rows if r["kernel"] == kernel and r["grid"] == grid]
        assert len(hit) == 1, (name, row, src)
        frac = o["algorithmic_bytes_per_launch"] / (hit[0]["avg_us"] * 1e-6) / 1e9 / 8000.0
        assert abs(frac - o["frac_rocprof"]) <= 1e-5 + 1e-4 * frac, (name, frac, o["frac_rocprof"])
        assert o["kernel"] == kernel, name
        tsrc = o["traffic_source"].split(" (")[0]
        path, key = tsrc.split(": workloads.")
        wl, kname = key.split(".", 1)
        with open(os.path.join(ROOT, path)) as f:
            assert json.load(f)["workloads"][wl][kname]["hbm_bytes"] == o["traffic"], name
