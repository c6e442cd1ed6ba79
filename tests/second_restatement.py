"""TEST INFRASTRUCTURE — a SECOND, independent restatement of the two Eigen-dependent pieces of the reference.

oracle/mppi_oracle.cpp and the ParticleFilter part of oracle/rbpf_oracle.cpp cannot be pinned to the compiled reference
(mppi.cpp / rk4.cpp / particle_filter.cpp need Eigen 3.3, which this image does not have, and stand-ins are ruled out), and
the reference holds no test or golden vector for either.  What can be done without Eigen is an N-version check: this file
restates the same reference sources a second time, in numpy / plain Python, written from the reference's text and NOT from
the C++ oracle, with a different program structure (vectorised over rollouts; closed-form Cholesky), so that a slip of the
pen in either restatement shows up as a disagreement (tests/test_second_restatement.py).  It does not turn "parity
unpinned" into "pinned" and nothing says it does.

Each function cites the reference file:line it follows.  Nothing in the product, bench.py or smoke() imports this file.
"""
from __future__ import annotations

import math

import numpy as np

PI = 3.14159265358979323846  # rigid2d.hpp:18


# ---------------------------------------------------------------------------------------------------------------------
# controller::MPPI (controller/src/controller/mppi.cpp, rk4.cpp, include/controller/mppi.hpp)
# ---------------------------------------------------------------------------------------------------------------------
def mppi_steps(horizon: float, dt: float) -> int:
    return int(horizon / dt)  # mppi.cpp:47, rk4.cpp:58: static_cast<int>(horizon/dt)


def kinematic_cart(r, b, x, u):
    """mppi.hpp:41-48; x = (x, y, theta) rows, u = (uL, uR) rows, every column one rollout."""
    return np.stack([(r / 2.0) * (u[0] + u[1]) * np.cos(x[2]),
                     (r / 2.0) * (u[0] + u[1]) * np.sin(x[2]),
                     (r / b) * (u[1] - u[0])])


def rk4_integrate(r, b, h, x, u):
    """rk4.cpp:95-115."""
    k1 = kinematic_cart(r, b, x, u)
    k2 = kinematic_cart(r, b, x + h * (0.5 * k1), u)
    k3 = kinematic_cart(r, b, x + h * (0.5 * k2), u)
    k4 = kinematic_cart(r, b, x + h * k3, u)
    return x + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)


def mppi_new_controls(prm: dict, u: np.ndarray, uinit, xd, x0, noise: np.ndarray) -> dict:
    """mppi.cpp:72-140.  prm: wheel_radius, wheel_base, lam, max_wheel_vel, horizon, dt, Q, R, P1, rollouts.
    u [2][T] warm start; x0 / xd = (x, y, theta); noise [K][T][2] in the reference's draw order (mppi.cpp:173-184:
    rollout-major, per step uL then uR).  Returns loss_mat [T][K], J [T][K] (before the min is subtracted), u after
    the shift, out = (ul, ur)."""
    r, b, lam, umax = prm["wheel_radius"], prm["wheel_base"], prm["lam"], prm["max_wheel_vel"]
    h = prm["dt"]
    T = mppi_steps(prm["horizon"], h)
    K = prm["rollouts"]
    Q, R, P1 = (np.asarray(prm[n], dtype=np.float64) for n in ("Q", "R", "P1"))
    xd = np.asarray(xd, dtype=np.float64)
    u = np.array(u, dtype=np.float64)
    du = np.transpose(np.asarray(noise, dtype=np.float64), (2, 1, 0))  # [2][T][K]: duL.col(k) = pert.row(0), :82-83
    x = np.repeat(np.asarray(x0, dtype=np.float64)[:, None], K, axis=1)
    loss = np.zeros((T, K))
    for i in range(T):
        up = u[:, i:i + 1] + du[:, i, :]           # u_pert = u + pert (:87); rollout controls are NOT clamped
        x = rk4_integrate(r, b, h, x, up)          # traj.col(i) = state AFTER step i (rk4.cpp:62-66)
        e = x - xd[:, None]
        # (e^T Q e)(0) + (u^T R u)(0), diagonal Q / R (mppi.hpp:87-93)
        loss[i] = (e[0] * Q[0] * e[0] + e[1] * Q[1] * e[1] + e[2] * Q[2] * e[2]) + (up[0] * R[0] * up[0] + up[1] * R[1] * up[1])
    loss[T - 1] = e[0] * P1[0] * e[0] + e[1] * P1[1] * e[1] + e[2] * P1[2] * e[2]  # REPLACES the last row (:105)
    J = np.zeros((T, K))
    J[T - 1] = loss[T - 1]
    for i in range(T - 2, -1, -1):                 # cumSumCost (:15-25)
        J[i] = loss[i] + J[i + 1]
    J_out = J.copy()
    for i in range(T):                             # :112-126
        row = J[i] - J[i].min()
        w = np.exp(row * -1.0 / lam) + 1e-8
        w = w * (1.0 / w.sum())
        u[0, i] = min(max(u[0, i] + float(w @ du[0, i]), -umax), umax)
        u[1, i] = min(max(u[1, i] + float(w @ du[1, i]), -umax), umax)
    out = (u[0, 0], u[1, 0])                       # :129-131
    u[:, :-1] = u[:, 1:].copy()                    # :134
    u[0, -1], u[1, -1] = uinit                     # :136-137
    return {"loss": loss, "J": J_out, "u": u, "out": np.array(out)}


# ---------------------------------------------------------------------------------------------------------------------
# bmapping::ParticleFilter logic (bmapping/src/bmapping/particle_filter.cpp); GridMapper is NOT restated here: the scan
# likelihoods are taken from the oracle's GridMapper, which is pinned bit-exact to the compiled reference
# ---------------------------------------------------------------------------------------------------------------------
def normalize_angle_pi(rad: float) -> float:
    """rigid2d.hpp:52-64."""
    q = math.floor((rad + PI) / (2.0 * PI))
    rad = (rad + PI) - q * 2.0 * PI
    if rad < 0:
        rad += 2.0 * PI
    return rad - PI


def pdf_normal(a: float, b: float) -> float:
    """grid_mapper.cpp:18-28."""
    if abs(b) < 1e-12:
        raise ValueError("Variance in pdfNormal is 0")
    return (1.0 / math.sqrt(2.0 * PI * b)) * math.exp(-0.5 * (a * a) / b)


def compose(pose, t):
    """Transform2D(pose) * Transform2D(t), both (theta, x, y) (rigid2d.cpp:213-222): returns (theta, x, y)."""
    c, s = math.cos(pose[0]), math.sin(pose[0])
    return (pose[0] + t[0], c * t[1] - s * t[2] + pose[1], s * t[1] + c * t[2] + pose[2])


def cholesky_lower(a: np.ndarray) -> np.ndarray:
    """cov.llt().matrixL() for a 3 x 3 (particle_filter.cpp:44,57), textbook column Cholesky."""
    n = a.shape[0]
    L = np.zeros((n, n))
    for j in range(n):
        d = a[j, j] - sum(L[j, q] * L[j, q] for q in range(j))
        L[j, j] = math.sqrt(d)
        for i in range(j + 1, n):
            L[i, j] = (a[i, j] - sum(L[i, q] * L[j, q] for q in range(j))) / L[j, j]
    return L


def sample_mode(center, sample_range, z: np.ndarray) -> np.ndarray:
    """particle_filter.cpp:504-519: k samples mu + L z around the mode, L = LLT(diag) = sqrt(diag); theta wrapped."""
    sd = [math.sqrt(v) for v in sample_range]
    out = np.empty((z.shape[0], 3))
    for j in range(z.shape[0]):
        out[j] = [normalize_angle_pi(center[0] + sd[0] * z[j, 0]), center[1] + sd[1] * z[j, 1], center[2] + sd[2] * z[j, 2]]
    return out


def pose_likelihood_odom(a, cur, prev, cur_od, prev_od) -> float:
    """particle_filter.cpp:383-437 (Probabilistic Robotics table 5.5); a = (srr, srt, str, stt); poses (theta, x, y)."""
    a1, a2, a3, a4 = a
    nrm = normalize_angle_pi
    rot1 = math.atan2(cur_od[2] - prev_od[2], cur_od[1] - prev_od[1]) - prev_od[0]
    trans = math.sqrt((cur_od[1] - prev_od[1]) ** 2 + (cur_od[2] - prev_od[2]) ** 2)
    rot2 = nrm(nrm(cur_od[0]) - nrm(prev_od[0]) - rot1)
    rot1_h = math.atan2(cur[2] - prev[2], cur[1] - prev[1]) - prev[0]
    trans_h = math.sqrt((cur[1] - prev[1]) ** 2 + (cur[2] - prev[2]) ** 2)
    rot2_h = nrm(nrm(cur[0]) - nrm(prev[0]) - rot1_h)
    t1 = a1 * rot1_h * rot1_h + a2 * trans_h * trans_h
    t2 = a3 * trans_h * trans_h + a4 * rot1_h * rot1_h + a4 * rot2_h * rot2_h
    t3 = a1 * rot2_h * rot2_h + a2 * trans_h * trans_h
    return (pdf_normal(nrm(nrm(rot1) - nrm(rot1_h)), t1) * pdf_normal(trans - trans_h, t2)
            * pdf_normal(nrm(nrm(rot2) - nrm(rot2_h)), t3))


def gaussian_proposal(samples: np.ndarray, p_scan_raw: np.ndarray, p_pose_raw: np.ndarray, clamps):
    """particle_filter.cpp:522-599: clamp both likelihoods, eta, mean (theta wrapped AFTER the division), covariance
    (theta differences NOT wrapped).  Returns mu, sigma, eta."""
    smin, smax, pmin, pmax = clamps
    p = np.clip(p_scan_raw, smin, smax) * np.clip(p_pose_raw, pmin, pmax)
    mu = np.zeros(3)
    eta = 0.0
    for j in range(samples.shape[0]):
        mu = mu + samples[j] * p[j]
        eta += p[j]
    if abs(eta) < 1e-12:
        raise ValueError("eta is 0")
    mu = mu / eta
    mu[0] = normalize_angle_pi(mu[0])
    sigma = np.zeros((3, 3))
    for j in range(samples.shape[0]):
        d = samples[j] - mu
        sigma = sigma + np.outer(d, d) * p[j]
    return mu, sigma / eta, eta


def sample_motion_model(u, pose, motion_noise, z):
    """particle_filter.cpp:295-322 (the ICP-failure branch); u = (w, vx, vy); x / y use the ALREADY updated theta."""
    w = [math.sqrt(motion_noise[i]) * z[i] for i in range(3)]
    th, x, y = pose
    if abs(u[0]) < 1e-12:
        th = normalize_angle_pi(th + w[0])
        x += u[1] * math.cos(th) + w[1]
        y += u[1] * math.sin(th) + w[2]
    else:
        th = normalize_angle_pi(th + u[0] + w[0])
        x += (-u[1] / u[0]) * math.sin(th) + (u[1] / u[0]) * math.sin(th + u[0]) + w[1]
        y += (u[1] / u[0]) * math.cos(th) - (u[1] / u[0]) * math.cos(th + u[0]) + w[2]
    return np.array([th, x, y])


def normalize_and_neff(weights: np.ndarray):
    """particle_filter.cpp:442-465: sequential sums; resample iff int(1/sum w^2) < N / 2 (integer division)."""
    s = 0.0
    for w in weights:
        s += w
    out = np.empty_like(weights)
    sq = 0.0
    for i, w in enumerate(weights):
        out[i] = w / s
        sq += out[i] ** 2
    neff = int(1.0 / sq)
    return out, s, sq, neff, neff < (len(weights) // 2)


def low_variance_resampling(weights: np.ndarray, z: float) -> np.ndarray:
    """particle_filter.cpp:468-500: r = N(0,1)/N (may be negative), comb spacing 1/(N-1), index clamped at N-1."""
    n = len(weights)
    r = z / float(n)
    c = weights[0]
    i = 0
    idx = np.empty(n, dtype=np.int32)
    for m in range(n):
        U = r + float(m * (1.0 / (n - 1)))
        while U > c:
            i += 1
            if i > n - 1:
                i = n - 1
                break
            c += weights[i]
        idx[m] = i
    return idx
