"""Builds profiles/<round>_traffic_pmc.json, <round>_sq_counters.json and <round>_kernel_stats.md from what tools/profile_round.sh
left in gpurun_out/ (run here after the GPU call has merged its files back).  usage: python tools/assemble_profiles.py [r03]"""
import json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
PREV = "r%02d" % (int(R[1:]) - 1)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = lambda f: os.path.join(ROOT, "gpurun_out", f)  # noqa: E731
P = lambda f: os.path.join(ROOT, "profiles", f)    # noqa: E731

old = json.load(open(P(f"{PREV}_traffic_pmc.json")))
old["_about"] = old["_about"].replace("round " + PREV[1:].lstrip("0"), "round " + R[1:].lstrip("0"))
tags = {"mppi_K1024_T50": "mppi_small_rng", "mppi_K1024_T50_resident_noise": "mppi_small", "mppi_K65536_T100": "mppi_large",
        "rbpf_N1000_k50_400x400": "rbpf", "rbpf_N1000_k50_400x400_plain_scans_only": "rbpf_plain"}
out = {"_about": old["_about"], "commands": old["commands"] + ["python tools/assemble_profiles.py"], "workloads": {}}
out["commands"] = sorted(set(out["commands"]), key=out["commands"].index)
for key, tag in tags.items():
    wl = json.load(open(G(f"pmc_summary_{tag}.json")))
    for name, v in wl.items():
        if name.startswith("mppi_rollout_fused"):
            v["read_bytes"] = int(v["fetch_size_kb_raw"] * 1024)
            v["hbm_bytes"] = v["read_bytes"] + v["write_bytes"]
            v["note"] = "reads are 64-byte row segments (8 rollouts x 8 B per time step): FETCH_SIZE at face value, no x2"
        if name.startswith("rbpf_raycast"):
            v["note"] = ("average over 11 launches of which ONE is the scan after a forced resample (15 tiles x 8 KB cloned per particle: "
                         "+123 MB written, +123 MB read in that launch); 16-byte accesses, whole cache lines per wave: the x2 read "
                         "correction and the 1:1 write reading are uncalibrated for this pattern (MI355X_MICROARCH.md, HBM)")
    out["workloads"][key] = wl
    if key.endswith("plain_scans_only"):
        for name, v in wl.items():
            if name.startswith("rbpf_raycast"):
                v["note"] = ("the same run WITHOUT the forced resamples: every launch is a plain scan (no tile clones) — the difference to "
                             "rbpf_N1000_k50_400x400 is what the post-resample scans' clones cost")
json.dump(out, open(P(f"{R}_traffic_pmc.json"), "w"), indent=1)

olds = json.load(open(P(f"{PREV}_sq_counters.json")))
sq = {"_about": olds["_about"]}
for key, tag in {"rbpf_N1000_k50_400x400": "rbpf", "mppi_K65536_T100": "mppi_large", "mppi_K1024_T50_device_noise": "mppi_small_rng"}.items():
    sq[key] = json.load(open(G(f"sq_summary_{tag}.json")))
json.dump(sq, open(P(f"{R}_sq_counters.json"), "w"), indent=1)
shutil.copy(G(f"{R}_kernel_stats.md"), P(f"{R}_kernel_stats.md"))
shutil.copy(G(f"{R}_bench_under_rocprof.json"), P(f"{R}_bench_under_rocprof.json"))
print("profiles/ refreshed")
