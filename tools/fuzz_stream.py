"""Dev: randomized parity sweep of the round-3 MPPI paths against the oracle (run on the GPU box; not part of the suite):
the prefix-form streaming rollout kernel (K >= 32768 forced to the sequential kernel, random T % 4 == 0, dt, sampling variances
that mix the Taylor and the general round, start poses near and far from the waypoint) and the in-library sharded tick
(tbnav_mppi_group with 2..8 members on device 0, random K per member) — J <= 1e-10, controls at the north star's bar / 10.
usage: python tools/fuzz_stream.py [n_stream] [n_group] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
g.load_package()
import oracle_api as orc
from cases import MPPI_BASE, WAYPOINTS, make_mppi, rel_err
from rtn_amd.mppi import MPPIGroup, CartModel, LossFunc

n_stream = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_group = int(sys.argv[2]) if len(sys.argv) > 2 else 30
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
orc.lib().orc_set_threads(8)
fails = 0


def check(tag, desc, m_new, m_u, m_J, d, u, ui, xd, x0, nz):
    ref = orc.mppi_new_controls(d, u, ui, xd, x0, nz)
    got = m_new(nz)
    ej = max(rel_err(J, ref["J"][:, sl]) for J, sl in m_J())
    assert ej < 1e-10, (tag, "J", ej, desc)
    assert np.allclose(got, ref["out"], rtol=1e-6, atol=1e-8), (tag, "out", got, ref["out"], desc)
    assert np.allclose(m_u(), ref["u"], rtol=1e-6, atol=1e-8), (tag, "u", desc, float(np.abs(m_u() - ref["u"]).max()))
    return ref


for i in range(n_stream):
    rng = np.random.default_rng([seed, 7, i])
    K = int(rng.integers(32768, 50000))
    T = 4 * int(rng.integers(3, 51))
    dt = float(rng.choice([0.01, 0.01, 0.02, 0.1]))
    var = float(rng.choice([0.05, 0.9, 0.9, 25.0, 400.0]))
    d = dict(MPPI_BASE, rollouts=K, horizon=T * dt + 0.4 * dt, dt=dt, lam=float(rng.choice([0.01, 0.1, 1.0])), ul_var=var, ur_var=float(rng.uniform(0.05, 2.0)) * var)
    desc = dict(i=i, K=K, T=T, dt=dt, var=var, lam=d["lam"])
    try:
        m = make_mppi(None, d, kernel=0)
        assert m.steps == T == orc.mppi_steps(d), (m.steps, T)
        xd = WAYPOINTS[int(rng.integers(0, 5))]
        m.setWaypoint(*xd)
        ui = (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)))
        m.setInitialControls(*ui)
        u = np.zeros((2, T)); u[0] = ui[0]; u[1] = ui[1]
        near = rng.random() < 0.3
        x0 = (xd[0] + float(rng.uniform(-1e-3, 1e-3)), xd[1] + float(rng.uniform(-1e-3, 1e-3)), xd[2]) if near else \
             (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), float(rng.uniform(-3.2, 3.2)))
        desc["kernel"] = m.rollout_kernel; desc["near"] = bool(near)
        for tick in range(2):
            nz = np.random.default_rng([seed, 8, i, tick]).standard_normal((K, T, 2)) * np.sqrt([d["ul_var"], d["ur_var"]])
            ref = check("stream", desc, lambda z: m.newControls(*x0, z), m.getControls, lambda: [(m.costToGo(), slice(None))], d, u, ui, xd, x0, nz)
            u = ref["u"]; m.setControls(u)
            x0 = (x0[0] + 0.003, x0[1] - 0.002, x0[2] + 0.004)
        m.close()
    except AssertionError as e:
        fails += 1
        print("FAIL", e.args[0] if e.args else e, flush=True)

for i in range(n_group):
    rng = np.random.default_rng([seed, 9, i])
    P = int(rng.integers(2, 9))
    Ks = int(rng.choice([64, 512, 1024, 2048, 4096, 8192, 3000, 40000 // P]))
    K = Ks * P
    T = int(rng.choice([12, 25, 50, 64, 100, 128]))
    if K * T > 6_000_000:
        T = 12
    d = dict(MPPI_BASE, rollouts=K, horizon=T * 0.01 + 0.004, lam=float(rng.choice([0.01, 0.1])))
    desc = dict(i=i, P=P, Ks=Ks, T=T)
    try:
        grp = MPPIGroup(CartModel(d["wheel_radius"], d["wheel_base"]), LossFunc(d["Q"], d["R"], d["P1"]), d["lam"], d["max_wheel_vel"], d["ul_var"],
                        d["ur_var"], d["horizon"], d["dt"], K, devices=[0] * P)
        xd = WAYPOINTS[int(rng.integers(0, 5))]
        grp.setWaypoint(*xd)
        u = np.zeros((2, T)); x0 = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), float(rng.uniform(-3, 3)))
        members = [grp.member(r) for r in range(P)]
        desc["kernel"] = members[0].rollout_kernel
        for tick in range(2):
            nz = np.random.default_rng([seed, 10, i, tick]).standard_normal((K, T, 2)) * np.sqrt(0.9)
            ref = check("group", desc, lambda z: grp.newControls(*x0, z), grp.getControls,
                        lambda: [(members[r].costToGo(), slice(r * Ks, (r + 1) * Ks)) for r in range(P)], d, u, (0, 0), xd, x0, nz)
            for r in range(1, P):
                assert np.array_equal(members[r].getControls(), members[0].getControls()), ("group", "members disagree", desc)
            u = ref["u"]; grp.setControls(u)
        grp.close()
    except AssertionError as e:
        fails += 1
        print("FAIL", e.args[0] if e.args else e, flush=True)
print(f"fuzz_stream: {n_stream} streaming cases, {n_group} group cases, seed {seed}: {fails} failures")
