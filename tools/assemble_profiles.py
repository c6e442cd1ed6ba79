"""Builds profiles/<round>_traffic_pmc.json, <round>_sq_counters.json, <round>_kernel_stats.md and — round 5 — one
<round>_kernel_stats_<workload>.md per workload from what tools/profile_round.sh left in gpurun_out/ (run here after the GPU call has
merged its files back).  usage: python tools/assemble_profiles.py [r05]"""
import json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r05"
PREV = "r%02d" % (int(R[1:]) - 1)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = lambda f: os.path.join(ROOT, "gpurun_out", f)  # noqa: E731
P = lambda f: os.path.join(ROOT, "profiles", f)    # noqa: E731

# workload key (what the bench lines cite) -> tag of tools/profile_round.sh
TAGS = {"mppi_K1024_T50": "mppi_small_rng", "mppi_K1024_T50_resident_noise": "mppi_small", "mppi_K8192_T100": "mppi_mid_rng",
        "mppi_K65536_T100": "mppi_large", "rbpf_N1000_k50_400x400": "rbpf", "rbpf_N1000_k50_400x400_plain_scans_only": "rbpf_plain",
        "rbpf_N1000_k50_400x400_survey_room": "rbpf_survey", "rbpf_N12500_2000x2000_1080beams": "rbpf_cfg4"}
SQ_KEYS = {"mppi_K1024_T50_device_noise": "mppi_small_rng", "mppi_K8192_T100": "mppi_mid_rng", "mppi_K65536_T100": "mppi_large",
           "rbpf_N1000_k50_400x400": "rbpf", "rbpf_N1000_k50_400x400_survey_room": "rbpf_survey", "rbpf_N12500_2000x2000_1080beams": "rbpf_cfg4"}

old = json.load(open(P(f"{PREV}_traffic_pmc.json")))
about = old["_about"].replace("round " + PREV[1:].lstrip("0"), "round " + R[1:].lstrip("0"))
out = {"_about": about, "commands": ["tools/profile_round.sh  (on the GPU box; drivers tools/mppi_tick_driver.py, tools/rbpf_driver.py)",
                                     "python tools/pmc_summary.py <tag> 3", "python tools/assemble_profiles.py"],
       "workload_drivers": {k: f"tools/profile_round.sh: run_all {t} ..." for k, t in TAGS.items()}, "workloads": {}}
for key, tag in TAGS.items():
    if not os.path.exists(G(f"pmc_summary_{tag}.json")):
        print("missing", tag); continue
    wl = json.load(open(G(f"pmc_summary_{tag}.json")))
    for name, v in wl.items():
        if name.startswith("mppi_rollout_fused"):
            v["read_bytes"] = int(v["fetch_size_kb_raw"] * 1024)
            v["hbm_bytes"] = v["read_bytes"] + v["write_bytes"]
            v["note"] = "reads are 64- / 128-byte row segments (8 or 16 rollouts x 8 B per time step): FETCH_SIZE at face value, no x2"
        if name.startswith("rbpf_raycast") and key == "rbpf_N1000_k50_400x400":
            v["note"] = ("the run's 14 scans: <512, 6, true, 8> = the FIRST TWO (the LDS array not yet sized to the boxes' need, every tile a first touch of the zero tile); "
                         "<512, 8, false, 4> = the other twelve, of which the eighth (scan 9) follows the forced resampling of scan 8 and makes the written tiles of the shared maps "
                         "private — its own counters: _map_update_launches_in_order.  16-byte accesses, whole cache lines per wave: the x2 read correction and the 1:1 write reading "
                         "hold for this pattern (calibrated: profiles/r05_fetch_write_calibration.txt — whole 128-byte lines fetched, whole 32-byte sectors written)")
        elif name.startswith("rbpf_raycast"):
            v["note"] = ("no forced resample in this run: every launch is a plain scan (no tile clones); partial-line 16-byte accesses: FETCH_SIZE x 2 = whole 128-byte "
                         "lines fetched, WRITE_SIZE = whole 32-byte sectors written (calibrated: profiles/r05_fetch_write_calibration.txt)")
    out["workloads"][key] = wl
    if os.path.exists(G(f"kstats_{tag}.md")):
        shutil.copy(G(f"kstats_{tag}.md"), P(f"{R}_kernel_stats_{key}.md"))
json.dump(out, open(P(f"{R}_traffic_pmc.json"), "w"), indent=1)

olds = json.load(open(P(f"{PREV}_sq_counters.json")))
sq = {"_about": olds["_about"]}
for key, tag in SQ_KEYS.items():
    if os.path.exists(G(f"sq_summary_{tag}.json")):
        sq[key] = json.load(open(G(f"sq_summary_{tag}.json")))
json.dump(sq, open(P(f"{R}_sq_counters.json"), "w"), indent=1)
shutil.copy(G(f"{R}_kernel_stats.md"), P(f"{R}_kernel_stats.md"))
shutil.copy(G(f"{R}_bench_under_rocprof.json"), P(f"{R}_bench_under_rocprof.json"))
print("profiles/ refreshed")
