"""What bench.py / bench_rbpf.py read from profiles/: one named row per figure, so that every `frac` of a bench line can be
recomputed from the committed profiler output.

  profiles/<round>_kernel_stats.md   rocprofv3 --kernel-trace --stats of `python bench.py ...` (tools/profile_round.sh,
                                     profiles/summarize_rocpd.py): | kernel | grid (threads) | wg | calls | avg us | min | max | ... | median us
  profiles/<round>_kernel_stats_<workload>.md   (round 5) the same table from a --kernel-trace --stats pass of ONE workload's driver
                                     (tools/profile_round.sh): the per-GPU shard shapes and the SURVEY room, whose launches the bench run
                                     mixes with other legs' — looked up first when a workload is named
  profiles/<round>_traffic_pmc.json  separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/collect_pmc.sh,
                                     tools/pmc_summary.py, tools/assemble_profiles.py): workloads.<key>.<kernel>.hbm_bytes

The newest round that has the file is used and NAMED in the line (`source`); a kernel whose row is missing gets null — never
a number that belongs to another kernel or instantiation."""
from __future__ import annotations

import json
import os
import re

ROOT = os.path.dirname(os.path.abspath(__file__))
ROUNDS = ("r06", "r05", "r04", "r03")


def _first_existing(suffix):
    for r in ROUNDS:
        p = os.path.join(ROOT, "profiles", f"{r}_{suffix}")
        if os.path.exists(p):
            return p
    return None


def _grid_threads(grid: str) -> int:
    n = 1
    for part in grid.lower().split("x"):
        n *= int(part)
    return n


def kernel_stats_rows(path=None):
    """Every row of the kernel-stats table: dict(kernel, grid, grid_threads, wg, calls, avg_us, min_us, max_us)."""
    path = path or _first_existing("kernel_stats.md")
    rows = []
    if not path:
        return rows, None
    with open(path) as f:
        for line in f:
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            if len(cells) < 7 or not re.fullmatch(r"[0-9x]+", cells[1].lower() or "-"):
                continue
            try:
                rows.append(dict(kernel=cells[0], grid=cells[1], grid_threads=_grid_threads(cells[1]), wg=int(cells[2]), calls=int(cells[3]),
                                 avg_us=float(cells[4]), min_us=float(cells[5]), max_us=float(cells[6]),
                                 median_us=float(cells[13]) if len(cells) > 13 else None))
            except ValueError:
                continue
    return rows, os.path.relpath(path, ROOT)


def rocprof_row(kernel: str, grid_threads: int | None = None, workload: str | None = None):
    """The row of exactly this kernel instantiation (as the profiler spells it) — with this many threads in the grid when the
    kernel was launched in several shapes; None when there is no such row.  workload: look in that workload's own table
    (profiles/<round>_kernel_stats_<workload>.md) first; its AVERAGE is over that workload's launches only."""
    rows, src = (None, None)
    if workload:
        path = _first_existing(f"kernel_stats_{workload}.md")
        if path:
            rows, src = kernel_stats_rows(path)
    if not rows:
        rows, src = kernel_stats_rows()
    hits = [r for r in rows if r["kernel"] == kernel and (grid_threads is None or r["grid_threads"] == grid_threads)]
    if not hits:
        return None
    r = max(hits, key=lambda q: q["calls"])
    return dict(source=src, row=f"{r['kernel']} | {r['grid']}", avg_us=r["avg_us"], median_us=r["median_us"], min_us=r["min_us"], calls=r["calls"])


def sq_row(workload: str, kernel: str):
    """The SQ counter entry (instruction mix, busy cycles) of `kernel` in workload `workload` of the committed SQ pass; None if absent."""
    path = _first_existing("sq_counters.json")
    if not path:
        return None
    try:
        with open(path) as f:
            wl = json.load(f)[workload]
    except (OSError, KeyError, ValueError):
        return None
    v = wl.get(kernel)
    return None if v is None else dict(source=f"{os.path.relpath(path, ROOT)}: {workload}.{kernel}", **v)


def pmc_row(workload: str, kernel: str, prefix_ok: bool = False):
    """HBM bytes per launch of `kernel` in workload `workload` of the committed PMC passes; None if absent."""
    path = _first_existing("traffic_pmc.json")
    if not path:
        return None
    try:
        with open(path) as f:
            wl = json.load(f)["workloads"][workload]
    except (OSError, KeyError, ValueError):
        return None
    hits = [(n, v) for n, v in wl.items() if n == kernel or (prefix_ok and n.startswith(kernel))]
    if not hits:
        return None
    name, v = max(hits, key=lambda nv: nv[1].get("launches", 0))
    return dict(source=f"{os.path.relpath(path, ROOT)}: workloads.{workload}.{name}", hbm_bytes=v["hbm_bytes"],
                read_bytes=v.get("read_bytes"), write_bytes=v.get("write_bytes"), launches=v.get("launches"))
