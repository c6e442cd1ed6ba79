"""Synthetic RBPF workloads shared by the GPU tests, smoke and bench:
axis-aligned room, LDS-01 lidar (360 beams, 0.12-3.5 m, sigma 0.01), a short trajectory with
per-scan odometry increments; T_icp = the body-frame odometry increment (what a converged ICP
returns on this world)."""
import numpy as np


def compose(a, b):  # (theta, x, y) o (theta, x, y), rigid2d.cpp:214-224
    c, s = np.cos(a[0]), np.sin(a[0])
    return np.array([a[0] + b[0], c * b[1] - s * b[2] + a[1], s * b[1] + c * b[2] + a[2]])


def inverse(a):
    c, s = np.cos(a[0]), np.sin(a[0])
    return np.array([-a[0], -(c * a[1] + s * a[2]), -(-s * a[1] + c * a[2])])


def trajectory(n_scans, inc=(0.07, 0.10, 0.05), start=(0.0, 0.0, 0.0)):
    """List of (prev_odom, cur_odom, T_icp, u) per scan; the first scan has cur == prev == start."""
    poses = [np.array(start, dtype=np.float64)]
    for _ in range(n_scans - 1):
        poses.append(poses[-1] + np.array(inc))
    out = []
    for s in range(n_scans):
        prev = poses[s - 1] if s > 0 else poses[0] - np.array(inc)  # moving from the start: the reference's
        cur = poses[s]                                              # pdfNormal throws on zero motion
        t_icp = compose(inverse(prev), cur)
        u = np.array([t_icp[0], np.hypot(t_icp[1], t_icp[2]), 0.0])
        out.append((prev, cur, t_icp, u))
    return out, poses


ROOM_SMALL = (-1.6, 1.5, -1.3, 1.7)   # fits the shipped 80x80 map (+-2 m)
ROOM_SURVEY = (-3.0, 3.0, -2.5, 2.5)  # SURVEY.md 8-d, for the 400x400 map (+-10 m): 246 of 360 beams return inside range_max
# bench_rbpf.py's headline room and trajectory (it imports them from here, so that what is benched is what is tested): every corner
# is < 3.4 m from every pose of the trajectory — all 360 beams valid — and the scans' bounding boxes (94 x 88 cells) let the map
# update run four workgroups per CU (rbpf_raycast_box<512, 8, false, 4>)
ROOM_BENCH = (-2.2, 2.2, -2.0, 2.0)
TRAJ_BENCH = (0.07, 0.02, 0.01)
TRAJ_SURVEY = (0.07, 0.10, 0.05)
