"""A-B of a one-process MPPI group's exchange on this box's one device: direct stores + polling combines (kind 2) against the group's
all-gather (kind 1: event-ordered copies between members that share a device).  Says what the host-side enqueue and the launches
cost; it says nothing about xGMI."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

g.load_package()
if os.environ.get("TBNAV_DEV_LIB"):   # (A/B runs of two builds of the library in one gpurun call)
    from rtn_amd import capi
    capi.LIB_PATH = os.path.abspath(os.environ["TBNAV_DEV_LIB"])
from cases import WAYPOINTS, mppi_cfg  # noqa: E402
from rtn_amd.mppi import MPPIGroup, CartModel, LossFunc  # noqa: E402


def group(d, P, direct):
    m = MPPIGroup(CartModel(d["wheel_radius"], d["wheel_base"]), LossFunc(d["Q"], d["R"], d["P1"]), d["lam"], d["max_wheel_vel"],
                  d["ul_var"], d["ur_var"], d["horizon"], d["dt"], d["rollouts"], devices=[0] * P)
    if not direct:
        m.setOption(8, 0)
    m.setWaypoint(*WAYPOINTS[1])
    return m


for K, horizon, P in ((8 * 1024, 0.5, 8), (65536, 1.0, 8), (2 * 1024, 0.5, 2)):
    row = []
    for direct in (True, False):
        m = group(mppi_cfg(K, horizon), P, direct)
        m.enqueueRngBatch((0.0, 0.0, 0.0), 42, 0, 50); m.synchronize()
        t0 = time.perf_counter()
        m.enqueueRngBatch((0.0, 0.0, 0.0), 42, 50, 400); m.synchronize()
        row.append(((time.perf_counter() - t0) / 400 * 1e6, m.member(0).exchangeKind()))
        m.close()
    print(f"K={K} T={int(round(horizon / 0.01))} members={P}: direct {row[0][0]:.1f} us/tick (kind {row[0][1]}), all-gather {row[1][0]:.1f} us/tick (kind {row[1][1]})")
