#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the REAL reference classes.

Run in the build container only (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_golden.py
It calls oracle/_ref/libtbnav_ref.so — the reference's own unmodified grid_mapper.cpp,
sensor_model.cpp, rigid2d.cpp, diff_drive.cpp compiled by oracle/Makefile — on seeded inputs and
stores inputs + outputs as compressed .npz.  The fixtures are data only; no reference source text.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_api as orc  # noqa: E402


def sparse(a, fill):
    idx = np.flatnonzero(a != fill).astype(np.int32)
    return idx, a[idx]


def gridmapper_fixture():
    out = {}
    rng = np.random.default_rng(20260929)
    for tag, grid, walls in (("g80", (0.05, -2.0, 2.0, -2.0, 2.0), (-1.6, 1.5, -1.3, 1.7)),
                             ("g120", (0.05, -3.0, 3.0, -3.0, 3.0), (-2.4, 2.6, -2.2, 2.1))):
        g = orc.GridAPI("ref", grid=grid)
        out[f"{tag}_grid"] = np.array(grid)
        out[f"{tag}_size"] = np.array([g.xsize, g.ysize])
        out[f"{tag}_constants"] = g.constants()
        pose = np.zeros(3)
        scans, poses = [], []
        for s in range(5):
            scan = orc.room_scan(pose, walls=walls, rng=rng)
            if s == 2:
                scan[10:20] = 0.05   # below range_min: gated out
                scan[200:205] = 3.7  # beyond range_max: gated out
            scans.append(scan); poses.append(pose.copy())
            # candidate poses scored BEFORE this scan is integrated (what gaussianProposal does)
            cands = pose + rng.normal(0, [0.01, 0.02, 0.02], (10, 3))
            out[f"{tag}_s{s}_cand"] = cands
            out[f"{tag}_s{s}_lik"] = np.array([g.likelihood(scan, c)[0] for c in cands])
            assert g.integrate_scan(scan, pose) == 0
            if s in (0, 1, 4):
                d = g.dump()
                i, v = sparse(d["log_odds"], 0.0); out[f"{tag}_s{s}_lo_idx"], out[f"{tag}_s{s}_lo_val"] = i, v
                out[f"{tag}_s{s}_state"] = d["state"].astype(np.int8)
                out[f"{tag}_s{s}_prob"] = d["prob"]
                out[f"{tag}_s{s}_occ_dist"] = d["occ_dist"]
                out[f"{tag}_s{s}_occ_cells"] = g.occ_cells()
                out[f"{tag}_s{s}_gridmap"] = g.grid_map()
            pose = pose + np.array([0.06, 0.05, 0.03])
        out[f"{tag}_scans"] = np.stack(scans); out[f"{tag}_poses"] = np.stack(poses)
        # G-B1: end points and their cell indices for 3 poses
        ep_pose = np.array([[0.0, 0.0, 0.0], [0.7, 0.2, -0.1], [-2.5, -0.3, 0.25]])
        out[f"{tag}_ep_pose"] = ep_pose
        for n, ps in enumerate(ep_pose):
            xy = g.end_points(scans[0], ps)
            out[f"{tag}_ep{n}_xy"] = xy
            out[f"{tag}_ep{n}_idx"] = np.array([g.world2rowmajor(x, y) for x, y in xy], dtype=np.int64)
        g.close()
    # Bresenham free-cell lists: every octant, axis-aligned, 45 degrees, zero length (80x80 map)
    g = orc.GridAPI("ref", grid=(0.05, -2.0, 2.0, -2.0, 2.0))
    pose = np.array([0.3, 0.013, -0.021])
    pts = [(0.013 + r * np.cos(t), -0.021 + r * np.sin(t)) for t in np.linspace(0, 2 * np.pi, 33) for r in (0.0, 0.04, 0.6, 1.8)]
    pts += [(0.013, 1.0), (0.013, -1.0), (1.0, -0.021), (-1.0, -0.021), (1.013, 0.979), (-0.987, -1.021), (1.013, -1.021), (-0.987, 0.979)]
    pts = np.array(pts)
    lists = [g.free_index(p, pose) for p in pts]
    out["bres_pose"] = pose; out["bres_pts"] = pts
    out["bres_len"] = np.array([len(x) for x in lists], dtype=np.int32)
    out["bres_cells"] = np.concatenate(lists).astype(np.int32)
    g.close()
    return out


def rigid_fixture():
    out = {}
    r = orc.RigidAPI("ref")
    rng = np.random.default_rng(11)
    ang = np.concatenate([rng.uniform(-30, 30, 200), [0, np.pi, -np.pi, 3 * np.pi / 2, 7 * np.pi / 6, 8 * np.pi / 3]])
    out["ang_in"] = ang; out["ang_out"] = np.array([r.normalize_angle_PI(a) for a in ang])
    P, Q = rng.uniform(-3, 3, (60, 3)), rng.uniform(-3, 3, (60, 3))
    TW = rng.uniform(-2, 2, (60, 3)); TW[:5, 0] = 0.0; TW[0] = 0.0
    V = rng.uniform(-4, 4, (60, 2))
    out.update(P=P, Q=Q, TW=TW, V=V,
               compose=np.array([r.compose(p, q) for p, q in zip(P, Q)]),
               apply=np.array([r.apply(p, v) for p, v in zip(P, V)]),
               inv=np.array([r.inv(p) for p in P]),
               twist=np.array([r.integrate_twist(p, t) for p, t in zip(P, TW)]))
    # a DiffDrive trajectory: alternating feedforward / encoder odometry
    d = r.dd_create([0.0, 0.0, 0.0], 0.16, 0.033)
    cmds, states = [], []
    enc = np.zeros(2)
    for s in range(60):
        if s % 2 == 0:
            tw = np.array([rng.uniform(-1, 1), rng.uniform(-0.2, 0.3), 0.0]); r.dd_feedforward(d, tw)
            cmds.append(np.concatenate([[0.0], tw[:2]]))
        else:
            enc = enc + rng.uniform(-0.3, 0.7, 2); r.dd_update_odometry(d, *enc)
            cmds.append(np.concatenate([[1.0], enc]))
        states.append(r.dd_state(d))
    out["dd_cmds"] = np.array(cmds); out["dd_states"] = np.array(states)
    r.dd_destroy(d)
    # exact-arc plant steps (wheelsToTwist * dt -> feedforward), the dynamics of the arc-rollout option (N4)
    arc_pose = np.concatenate([rng.uniform(-3, 3, (80, 2)), rng.uniform(-7, 7, (80, 1))], 1)
    arc_wheels = rng.uniform(-6.35, 6.35, (80, 2))
    arc_wheels[:4] = [[1.0, 1.0], [0.0, 0.0], [2.0, 2.0 + 1e-11], [6.3, -6.3]]
    out["arc_pose"] = arc_pose; out["arc_wheels"] = arc_wheels
    out["arc_out"] = np.array([r.dd_arc_step(0.16, 0.033, 0.01, p, w)[0] for p, w in zip(arc_pose, arc_wheels)])
    # knife-edge thresholds (SURVEY.md hard part 2): prob(l) for l around the occupied / free cut-offs
    l_occ, l_free = r.prob_to_log_odds(0.90), r.prob_to_log_odds(0.35)
    ls = np.array([l_occ, np.nextafter(l_occ, 0), np.nextafter(l_occ, 9), l_free, np.nextafter(l_free, 0), np.nextafter(l_free, -9),
                   l_occ + l_free, 2 * l_free, 2 * l_occ, 0.0])
    out["knife_l"] = ls; out["knife_prob"] = np.array([r.log_odds_to_prob(l) for l in ls])
    out["pdf_in"] = np.stack([rng.normal(0, 1, 100), rng.uniform(0.01, 3, 100)], 1)
    out["pdf_out"] = np.array([r.pdf_normal(a, b)[0] for a, b in out["pdf_in"]])
    return out


if __name__ == "__main__":
    assert orc.ref_available(), "build oracle/_ref first: make -C oracle ref"
    np.savez_compressed(os.path.join(HERE, "ref_gridmapper.npz"), **gridmapper_fixture())
    np.savez_compressed(os.path.join(HERE, "ref_rigid2d.npz"), **rigid_fixture())
    for f in ("ref_gridmapper.npz", "ref_rigid2d.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
