#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a small markdown table.

usage: python profiles/summarize_rocpd.py <results.db> [--match substr] > profiles/<name>.md

Groups dispatches by (kernel name, grid, workgroup) so that one bench run that launches the same
kernel on two workload sizes (K=1024/T=50 and K=65536/T=100) is reported per size; durations in µs.
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"(\(anonymous namespace\)|tbnav_rk|tbnav_mk)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main():
    db = sys.argv[1]
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, grid_x, grid_y, grid_z, workgroup_x, count(*), avg(duration), min(duration), max(duration), "
        "sum(duration), max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels "
        "group by name, grid_x, grid_y, grid_z, workgroup_x order by sum(duration) desc").fetchall()
    total = sum(r[9] for r in rows) or 1
    # the median beside the average: a kernel launched on two kinds of step (rbpf_raycast_box on a plain scan and on the scan after a
    # resampling, which also copies tiles) has an average that describes neither
    durs = {}
    for name, gx, gy, gz, wx, d in c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration from kernels"):
        durs.setdefault((name, gx, gy, gz, wx), []).append(d)
    print("| kernel | grid (threads) | wg | calls | avg µs | min µs | max µs | total ms | % | vgpr | sgpr | lds B | scratch | median µs |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        if match and match not in r[0]:
            continue
        dd = sorted(durs[(r[0], r[1], r[2], r[3], r[4])])
        med = dd[len(dd) // 2] if len(dd) % 2 else 0.5 * (dd[len(dd) // 2 - 1] + dd[len(dd) // 2])
        print(f"| {short(r[0])} | {r[1]}x{r[2]}x{r[3]} | {r[4]} | {r[5]} | {r[6]/1e3:.3f} | {r[7]/1e3:.3f} | {r[8]/1e3:.3f} | "
              f"{r[9]/1e6:.3f} | {100*r[9]/total:.1f} | {r[10]} | {r[11]} | {r[12]} | {r[13]} | {med/1e3:.3f} |")
    # the map update's launches one by one, in launch order (a run of a dozen scans): the first two (empty maps: every tile a first
    # touch), the plain ones and the one that follows a resampling (tiles of a shared map made private) are three different launches
    seq = {}
    cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
    order = next((k for k in ("start", "start_time", "dispatch_id", "id") if k in cols), None)
    for name, d in c.execute("select name, duration from kernels" + (f" order by {order}" if order else "")):
        if "rbpf_raycast_box" in name:
            seq.setdefault(short(name), []).append(d)
    if seq and sum(len(v) for v in seq.values()) <= 40:
        print()
        print("Map-update launches in launch order, µs (a kernel's own launches; tools/rbpf_driver.py says which scans resample):")
        for k, v in seq.items():
            print(f"- `{k}`: " + " ".join(f"{d/1e3:.1f}" for d in v))


if __name__ == "__main__":
    main()
