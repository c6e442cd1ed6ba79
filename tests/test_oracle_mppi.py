"""CPU tests of the MPPI oracle: the hand-checkable known answers of SURVEY.md's appendix (single
formulas restated from mppi.hpp:41-105, rk4.cpp:95-115, mppi.cpp:47,115-121) and self-consistency
of the sharded formulation.  The reference has no test for this path (parity unpinned)."""
import numpy as np
import pytest

import oracle_api as orc
from cases import MPPI_BASE, mppi_cfg, rel_err


def test_rk4_known_answers():
    d = MPPI_BASE
    assert orc.rk4_step(d, [0, 0, 0], [1, 1]).tolist() == [0.00033000000000000005, 0.0, 0.0]
    x1 = orc.rk4_step(d, [0.1, -0.2, 0.3], [2, 5])
    assert x1.tolist() == [0.10110235063078238, -0.19965526266418107, 0.3061875]
    x2 = orc.rk4_step(d, x1, [2, 5])
    assert x2.tolist() == [0.10220254711114436, -0.19930371117649695, 0.312375]


def test_loss_known_answers():
    d = MPPI_BASE
    x1 = orc.rk4_step(d, [0.1, -0.2, 0.3], [2, 5])
    x2 = orc.rk4_step(d, x1, [2, 5])
    assert orc.loss(d, x1, [1, 0, 0], [2, 5]) == 8481.785830295235
    assert orc.terminal_loss(d, x2, [1, 0, 0]) == 943.3403763274414


def test_softmin_known_answer():
    w, sl, _ = orc.softmin_step(0.01, [3.0, 3.004, 3.05, 10.0], [0.5, -0.25, 1.0, 2.0], [0, 0, 0, 0])
    exp_w = [0.5962822933386792, 0.3996999762868432, 0.004017724411654915, 5.9628228737585644e-09]
    assert rel_err(w, exp_w) < 1e-14
    assert abs(sl - 0.20223388893492947) < 1e-15


@pytest.mark.parametrize("h,dt,T", [(0.25, 0.01, 25), (0.5, 0.01, 50), (1.0, 0.01, 100),
                                     (0.29, 0.01, 28), (0.3, 0.1, 2)])
def test_steps_truncation(h, dt, T):  # mppi.cpp:47 static_cast<int>(horizon/dt)
    assert orc.mppi_steps(mppi_cfg(8, h, dt=dt)) == T


def test_normal_stream_fresh_distribution_per_draw():
    # utilities.cpp:20-24: a new std::normal_distribution per draw.  Deterministic in the seed,
    # scales with sigma, and NOT equal to numpy's stream.
    a = orc.normal_stream(42, 1000, 0.0, 1.0)
    b = orc.normal_stream(42, 1000, 0.0, 1.0)
    c = orc.normal_stream(42, 1000, 0.0, 2.0)
    assert np.array_equal(a, b) and np.allclose(c, 2.0 * a, rtol=1e-15)
    big = orc.normal_stream(7, 200000, 0.0, np.sqrt(0.9))
    assert abs(big.mean()) < 0.01 and abs(big.var() - 0.9) < 0.01


def test_tick_structure_terminal_overwrite_and_shift():
    d = mppi_cfg(16, 0.05)  # T = 5
    T, K = 5, 16
    noise = orc.normal_stream(3, K * T * 2, 0, np.sqrt(0.9)).reshape(K, T, 2)
    u0 = np.full((2, T), 0.3)
    r = orc.mppi_new_controls(d, u0, (0.7, -0.7), (1.0, 0.0, 1.5707), (0, 0, 0), noise)
    # cost-to-go is the suffix sum of the loss matrix (cumSumCost, mppi.cpp:15-25)
    assert np.array_equal(r["J"][T - 1], r["loss"][T - 1])
    assert rel_err(r["J"][0], r["loss"][::-1].cumsum(0)[-1]) < 1e-14
    # returned controls are column 0 before the shift; last column re-initialised (mppi.cpp:129-137)
    assert r["out"] == (r["u_upd"][0, 0], r["u_upd"][1, 0])
    assert np.array_equal(r["u"][:, :-1], r["u_upd"][:, 1:])
    assert r["u"][:, -1].tolist() == [0.7, -0.7]
    assert np.all(np.abs(r["u_upd"]) <= d["max_wheel_vel"])


@pytest.mark.parametrize("splits", [[64], [32, 32], [1, 63], [10, 20, 34]])
def test_sharded_formulation_equals_unsharded(splits):
    d = mppi_cfg(64, 0.25)
    T, K = 25, 64
    noise = orc.normal_stream(11, K * T * 2, 0, np.sqrt(0.9)).reshape(K, T, 2)
    u0 = np.zeros((2, T)); u0[0] = 1.0; u0[1] = 1.2
    full = orc.mppi_new_controls(d, u0, (0, 0), (1.0, 0.0, 1.5707), (0.1, 0.0, 0.2), noise)
    recs, k0 = [], 0
    for n in splits:
        ds = dict(d, rollouts=n)
        _, rec = orc.mppi_shard_partials(ds, u0, (1.0, 0.0, 1.5707), (0.1, 0.0, 0.2), noise[k0:k0 + n])
        recs.append(rec); k0 += n
    u_new, out = orc.mppi_combine(d, u0, (0, 0), np.stack(recs))
    assert rel_err(u_new[:, :-1], full["u"][:, :-1]) < 1e-10
    assert rel_err(out, full["out"]) < 1e-10
