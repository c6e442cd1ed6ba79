"""GPU parity tests of the MPPI path: HIP (through the C-ABI) vs the oracle on the same seeded
inputs.  Tolerances (north star): control vector <= 1e-5 relative; we assert far tighter where the
only difference is libm vs ocml sin/cos/exp: cost-to-go J <= 1e-12 relative, u <= 1e-9."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as orc
from cases import WAYPOINTS, make_mppi, mppi_cfg, rel_err

pytestmark = pytest.mark.gpu

J_RTOL = 1e-12      # cost-to-go, fp64, same association, different libm
U_RTOL = 1e-9       # updated controls (north-star bar is 1e-5)
U_ATOL = 1e-12


def _noise(seed, K, T, var=0.9):
    return orc.normal_stream(seed, K * T * 2, 0.0, np.sqrt(var)).reshape(K, T, 2)


def _check_tick(m, d, u_before, uinit, xd, x0, noise, dyn=0):
    ref = orc.mppi_new_controls(d, u_before, uinit, xd, x0, noise, dyn=dyn)
    got = m.newControls(*x0, noise)
    J = m.costToGo()
    assert rel_err(J, ref["J"]) < J_RTOL
    assert np.allclose(got, ref["out"], rtol=U_RTOL, atol=U_ATOL)
    u_after = m.getControls()
    assert np.allclose(u_after, ref["u"], rtol=U_RTOL, atol=U_ATOL)
    return ref


def test_cfg1_three_ticks_warm_start(gpu_pkg):
    """BASELINE configs[0]: K=64, T=25; three consecutive ticks exercise the warm-start shift."""
    d = mppi_cfg(64, 0.25)
    m = make_mppi(gpu_pkg, d)
    assert (m.steps, m.rollouts) == (25, 64)
    xd = WAYPOINTS[1]
    m.setWaypoint(*xd)
    u = np.zeros((2, 25)); x0 = (0.0, 0.0, 0.0)
    for tick in range(3):
        ref = _check_tick(m, d, u, (0.0, 0.0), xd, x0, _noise(42 + tick, 64, 25))
        u = ref["u"]
        x0 = (x0[0] + 0.001, x0[1], x0[2] + 0.002)


def test_cfg2_k1024_t50(gpu_pkg):
    """BASELINE configs[1]: K=1024, T=50 — control vector within 1e-5 (asserted at 1e-9)."""
    d = mppi_cfg(1024, 0.5)
    m = make_mppi(gpu_pkg, d)
    m.setWaypoint(*WAYPOINTS[1])
    m.setInitialControls(0.5, 0.6)
    u = np.zeros((2, 50)); u[0] = 0.5; u[1] = 0.6
    _check_tick(m, d, u, (0.5, 0.6), WAYPOINTS[1], (0.02, -0.01, 0.1), _noise(42, 1024, 50))


@pytest.mark.parametrize("K,horizon", [(1, 0.05), (1, 0.01), (3, 0.01), (2, 0.02), (5, 1.0), (63, 0.1), (65, 0.1), (100, 0.29),
                                        (2047, 0.05), (2049, 0.05), (4100, 0.02)])
def test_ragged_sizes(gpu_pkg, K, horizon):
    """K not a multiple of the wave (64) or of the K-slice (2048); the shipped rollouts=5, T=100;
    the int(horizon/dt) truncation case 0.29/0.01 -> 28 (mppi.cpp:47); a horizon of ONE step (the terminal loss is all there is,
    mppi.cpp:105) and of two, with one to three rollouts."""
    d = mppi_cfg(K, horizon)
    m = make_mppi(gpu_pkg, d)
    T = orc.mppi_steps(d)
    assert m.steps == T
    m.setWaypoint(*WAYPOINTS[2])
    _check_tick(m, d, np.zeros((2, T)), (0, 0), WAYPOINTS[2], (0.5, 0.2, 1.0), _noise(K, K, T))


@pytest.mark.parametrize("fused", ["0", "4", "8", "16"])
@pytest.mark.parametrize("K,horizon", [(1024, 0.5), (100, 1.0), (37, 1.28), (9, 0.65)])
def test_rollout_kernel_variants_agree_with_the_oracle(gpu_pkg, fused, K, horizon):
    """The small-K tick has two implementations: three kernels (time-parallel rollout, partials, combine;
    TBNAV_MPPI_FUSED=0) and the fused rollout+partials kernel (one wave per rollout, lanes over time; 4 / 8 / 16
    rollouts per workgroup).  Each must meet the oracle on its own: T = 50 (one step per lane), T = 100 and 128
    (two steps per lane, full last lane), T = 65 (ragged last lanes), K not a multiple of the workgroup tile, and
    three consecutive ticks so that the shift-on-read warm start is exercised in every variant."""
    d = mppi_cfg(K, horizon)
    m = make_mppi(gpu_pkg, d, kernel="scan" if fused == "0" else -int(fused))
    T = orc.mppi_steps(d)
    assert m.steps == T
    m.setWaypoint(*WAYPOINTS[2])
    u = np.zeros((2, T)); x0 = (0.5, 0.2, 1.0)
    for tick in range(3):
        ref = _check_tick(m, d, u, (0, 0), WAYPOINTS[2], x0, _noise(K + tick, K, T))
        u = ref["u"]
        x0 = (x0[0] + 0.002, x0[1] - 0.001, x0[2] + 0.003)


@pytest.mark.parametrize("K,horizon,x0", [(1024, 0.5, (0.5, 0.2, 1.0)), (64, 1.0, (0.0, 0.0, 3.1)), (37, 1.28, (-1.0, 2.0, -3.0)),
                                          (100, 4.0, (0.1, 0.1, 0.4)), (16640, 0.3, (0.3, -0.2, 2.0))])
def test_exact_arc_dynamics_option(gpu_pkg, K, horizon, x0):
    """rollouts integrated with the plant's own step (DiffDrive::feedforward of
    wheelsToTwist(u) * dt) instead of the RK4 cart.  The oracle's step is pinned bit for bit against the reference's
    DiffDrive class (tests/test_oracle_vs_reference.py, tests/golden/ref_rigid2d.npz); the device must meet the
    oracle at the usual tolerances in the fused kernel (T = 50, 100, 128), the sequential kernel (T = 400 and
    K = 16640 >= 2 waves per CU-SIMD pair) — including headings that cross the +-pi cut during the horizon — over
    three ticks of warm start."""
    d = mppi_cfg(K, horizon)
    m = make_mppi(gpu_pkg, d)
    m.setDynamics("arc")
    T = orc.mppi_steps(d)
    m.setWaypoint(*WAYPOINTS[2])
    m.setInitialControls(2.0, 3.5)             # a turning warm start: the heading sweeps ~0.3 rad/s
    u = np.zeros((2, T)); u[0] = 2.0; u[1] = 3.5
    for tick in range(3 if K <= 1024 else 1):
        ref = _check_tick(m, d, u, (2.0, 3.5), WAYPOINTS[2], x0, _noise(K + tick, K, T), dyn=1)
        u = ref["u"]
        x0 = (x0[0] + 0.002, x0[1] - 0.001, x0[2] + 0.003)


def test_arc_and_rk4_dynamics_agree_to_truncation_order(gpu_pkg):
    """The two dynamics describe the same cart: with zero-order-hold controls RK4's error per step is O(dt^5) in the
    position, so the cost-to-go of the two models differs by a hair, not by a modelling error (and not by zero:
    the option really is a different integrator)."""
    d = mppi_cfg(256, 0.5)
    nz = _noise(5, 256, 50)
    Js = []
    for model in ("rk4", "arc"):
        m = make_mppi(gpu_pkg, d)
        m.setDynamics(model)
        m.setWaypoint(*WAYPOINTS[1])
        m.newControls(0.1, 0.2, 0.3, nz)
        Js.append(m.costToGo().copy())
    rel = np.abs(Js[0] - Js[1]) / np.abs(Js[0])
    assert 0 < rel.max() < 1e-6


@pytest.mark.parametrize("sampler", [None, 0])
@pytest.mark.parametrize("K,horizon", [(1024, 0.5), (8192, 1.0), (100, 1.0), (37, 1.28), (200, 2.0)])
def test_in_kernel_noise_equals_sampled_noise(gpu_pkg, K, horizon, sampler):
    """Production tick (tbnav_mppi_new_controls_rng): the fused small-K kernel generates the perturbations of
    (seed, tick) itself instead of loading them.  They must be the values tbnav_mppi_sample_noise writes, so the tick
    equals "sample, then tick on the sampled arrays" bit for bit — and those arrays are what the oracle is fed.
    T = 200 has no fused kernel: the entry point then samples first, same contract.
    sampler None: the handle's default — since round 6 the fp64 sampler (the reference's std::normal_distribution<double> width,
    utilities.cpp:20-24), which is what bench.py's headline tick runs; 0: the fp32 sampler, now the option."""
    from rtn_amd import capi
    d = mppi_cfg(K, horizon)
    m_rng, m_ref = make_mppi(gpu_pkg, d), make_mppi(gpu_pkg, d)
    T = m_rng.steps
    rg = 2 if sampler is None else 1
    for m in (m_rng, m_ref):
        m.setWaypoint(*WAYPOINTS[1])
        if sampler is not None:
            m.setOption(capi.MPPI_OPT_SAMPLER, sampler)
    x0 = (0.1, -0.2, 0.3)
    u = np.zeros((2, T))
    for tick in range(3):
        got = m_rng.newControlsRng(x0, 77, tick)
        m_ref.sampleNoise(77, tick)
        want = m_ref.newControlsDev(x0, 0, 0)
        assert got == want and np.array_equal(m_rng.getControls(), m_ref.getControls())
        a, b = m_ref.getNoise()                      # [T][K] each -> the oracle's [K][T][2]
        ref = orc.mppi_new_controls(d, u, (0, 0), WAYPOINTS[1], x0, np.stack([a.T, b.T], axis=2))
        assert np.allclose(got, ref["out"], rtol=U_RTOL, atol=U_ATOL)
        assert rel_err(m_rng.costToGo(), ref["J"]) < J_RTOL
        u = ref["u"]
        x0 = (x0[0] + 0.002, x0[1], x0[2] + 0.001)
    if (K, horizon) == (1024, 0.5):   # BASELINE configs[1]: the kernels of bench.py's headline line, by name
        assert m_rng.lastKernelNames()[:2] == (f"mppi_rollout_fused<2, 8, 1, {rg}>", "mppi_combine<2, 0>"), m_rng.lastKernelNames()
    if (K, horizon) == (8192, 1.0):   # one GPU's share of configs[3] on 8: bench.py's configs3_shard_one_gpu leg
        assert m_rng.lastKernelNames()[:2] == (f"mppi_rollout_fused<2, 16, 2, {rg}>", "mppi_combine_wide"), m_rng.lastKernelNames()


@pytest.mark.parametrize("K,horizon,dyn", [(1024, 0.5, "rk4"), (4096, 1.0, "rk4"), (100, 1.28, "rk4"), (1024, 0.5, "arc"), (200, 2.0, "rk4")])
def test_fp64_sampler_in_kernel_equals_sampled_noise(gpu_pkg, K, horizon, dyn):
    """TBNAV_MPPI_OPT_SAMPLER = 1 (fp64 Box-Muller on 52-bit uniforms, the width of the reference's std::normal_distribution<double>,
    utilities.cpp:20-24): same contract as the default sampler — drawn inside the fused kernel (default dynamics) or sampled first
    (T = 200: no in-kernel form), the tick equals "sample, then tick on the sampled arrays" bit for bit, and the oracle
    fed those arrays agrees."""
    from rtn_amd import capi
    d = mppi_cfg(K, horizon)
    m_rng, m_ref = make_mppi(gpu_pkg, d), make_mppi(gpu_pkg, d)
    T = m_rng.steps
    for m in (m_rng, m_ref):
        m.setWaypoint(*WAYPOINTS[1]); m.setDynamics(dyn); m.setOption(capi.MPPI_OPT_SAMPLER, 1)
    x0 = (0.1, -0.2, 0.3)
    u = np.zeros((2, T))
    for tick in range(3):
        got = m_rng.newControlsRng(x0, 77, tick)
        if T <= 128:
            assert m_rng.lastKernelNames()[0].endswith(", 2>"), m_rng.lastKernelNames()   # mppi_rollout_fused<TRIG, R, TL, 2> (round 6: every dynamics)
        m_ref.sampleNoise(77, tick)
        want = m_ref.newControlsDev(x0, 0, 0)
        assert got == want and np.array_equal(m_rng.getControls(), m_ref.getControls())
        a, b = m_ref.getNoise()
        ref = orc.mppi_new_controls(d, u, (0, 0), WAYPOINTS[1], x0, np.stack([a.T, b.T], axis=2), dyn=1 if dyn == "arc" else 0)
        assert np.allclose(got, ref["out"], rtol=U_RTOL, atol=U_ATOL)
        u = ref["u"]
        x0 = (x0[0] + 0.002, x0[1], x0[2] + 0.001)
    # the two samplers share the Philox counters but not the bits they use: different perturbations
    m_ref.setOption(capi.MPPI_OPT_SAMPLER, 0); m_ref.sampleNoise(77, 0); a0, _ = m_ref.getNoise()
    m_ref.setOption(capi.MPPI_OPT_SAMPLER, 1); m_ref.sampleNoise(77, 0); a1, _ = m_ref.getNoise()
    assert not np.array_equal(a0, a1)
    m_rng.close(); m_ref.close()


def test_fp64_sampler_statistics_and_tails_to_six_sigma(gpu_pkg):
    """6.5 M pairs per draw x 8 draws = 1.05e8 normals: moments, and the tail counts beyond 3, 4, 5 and 6 sigma against the normal
    law (expected beyond 6 sigma: 0.21 of 1.05e8 — the default fp32 sampler's grid ENDS at 5.9; here a value out there is
    possible and the counts at 5 sigma (60 expected) must be right); values fill the fp64 grid (no 2^-24 lattice)."""
    from math import erfc, sqrt
    from rtn_amd import capi
    d = mppi_cfg(65536, 1.0, ul_var=1.0, ur_var=1.0)
    m = make_mppi(gpu_pkg, d)
    m.setOption(capi.MPPI_OPT_SAMPLER, 1)
    n = 0
    cnt = {3.0: 0, 4.0: 0, 5.0: 0, 6.0: 0}
    s1 = s2 = s4 = 0.0
    mx = 0.0
    for tick in range(8):
        m.sampleNoise(2024, tick)
        a, b = m.getNoise()
        z = np.concatenate([a.ravel(), b.ravel()])
        n += z.size; s1 += z.sum(); s2 += (z ** 2).sum(); s4 += (z ** 4).sum(); mx = max(mx, np.abs(z).max())
        for t in cnt:
            cnt[t] += int((np.abs(z) > t).sum())
        if tick == 0:
            assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 1e-3
            frac = np.mod(np.abs(z[:100000]) * 2.0 ** 24, 1.0)
            assert (frac != 0).mean() > 0.99   # not on the fp32 sampler's lattice
    mean, var = s1 / n, s2 / n
    assert abs(mean) < 5 / sqrt(n) and abs(var - 1.0) < 5 * sqrt(2.0 / n) and abs(s4 / n / var ** 2 - 3.0) < 5 * sqrt(96.0 / n)
    for t, c in cnt.items():
        exp = n * erfc(t / sqrt(2.0))
        print(f"[fp64 sampler] |z| > {t}: {c} (expected {exp:.2f})")
        assert abs(c - exp) <= 5 * sqrt(exp) + 3, (t, c, exp)
    assert 5.0 < mx < 8.6
    m.close()


@pytest.mark.parametrize("K,horizon", [(8192, 1.0), (6000, 0.5), (5000, 1.28)])
def test_mid_k_tick_with_four_waves_per_time_step_in_the_combine(gpu_pkg, K, horizon):
    """K = 4097 ... 8192 (what each of 8 GPUs runs for BASELINE configs[3]): the fused kernel leaves more than 256 soft-min records per time
    step and the combine takes four waves per step (mppi_combine_wide, round 5).  Against the oracle tick over three warm-started ticks,
    and against the one-wave-per-step combine (TBNAV_MPPI_OPT_WIDE_COMBINE = 0): same controls to rounding."""
    from rtn_amd import capi
    d = mppi_cfg(K, horizon)
    m, m1 = make_mppi(gpu_pkg, d), make_mppi(gpu_pkg, d)
    m1.setOption(capi.MPPI_OPT_WIDE_COMBINE, 0)
    T = m.steps
    for x in (m, m1):
        x.setWaypoint(*WAYPOINTS[1])
    u = np.zeros((2, T)); x0 = (0.05, -0.02, 0.1)
    for tick in range(3):
        nz = _noise(300 + tick, K, T)
        ref = _check_tick(m, d, u, (0.0, 0.0), WAYPOINTS[1], x0, nz)
        got1 = m1.newControls(*x0, nz)
        assert m.lastKernelNames()[1] == "mppi_combine_wide" and m1.lastKernelNames()[1].startswith("mppi_combine<"), (m.lastKernelNames(), m1.lastKernelNames())
        assert np.allclose(got1, ref["out"], rtol=U_RTOL, atol=U_ATOL) and np.allclose(m1.getControls(), m.getControls(), rtol=1e-12, atol=1e-14)
        u = ref["u"]
        x0 = (x0[0] + 0.002, x0[1], x0[2] + 0.001)
    m.close(); m1.close()


def test_long_horizon_uses_global_scratch_path(gpu_pkg):
    """T = 400 > 320: per-step losses no longer fit LDS ([T][64] doubles), J is the scratch."""
    d = mppi_cfg(96, 4.0)
    m = make_mppi(gpu_pkg, d)
    assert m.steps == 400
    m.setWaypoint(*WAYPOINTS[1])
    _check_tick(m, d, np.zeros((2, 400)), (0, 0), WAYPOINTS[1], (0, 0, 0), _noise(5, 96, 400))


def test_clamp_saturates(gpu_pkg):
    """Small max_wheel_vel: every updated control sits on the clamp (mppi.cpp:124-125)."""
    d = mppi_cfg(256, 0.25, max_wheel_vel=0.05)
    m = make_mppi(gpu_pkg, d)
    m.setWaypoint(*WAYPOINTS[1])
    ref = _check_tick(m, d, np.zeros((2, 25)), (0, 0), WAYPOINTS[1], (0, 0, 0), _noise(9, 256, 25))
    assert np.all(np.abs(ref["u_upd"]) <= 0.05 + 1e-15)


def test_zero_noise_leaves_controls_unchanged(gpu_pkg):
    d = mppi_cfg(128, 0.25)
    m = make_mppi(gpu_pkg, d)
    m.setInitialControls(1.0, 2.0)
    got = m.newControls(0, 0, 0, np.zeros((128, 25, 2)))
    assert got == (1.0, 2.0)
    assert np.array_equal(m.getControls(), np.array([[1.0] * 25, [2.0] * 25]))


def test_two_shards_on_one_gpu_equal_unsharded(gpu_pkg):
    """The multi-GPU formulation (partials -> all-gather -> combine) run as two K-slices on one
    device reproduces the unsharded tick and the oracle."""
    import torch
    d = mppi_cfg(3000, 0.5)
    T, K = 50, 3000
    noise = _noise(21, K, T)
    xd, x0 = WAYPOINTS[1], (0.0, 0.0, 0.0)
    ref = orc.mppi_new_controls(d, np.zeros((2, T)), (0, 0), xd, x0, noise)
    shards = []
    recs = []
    for lo, hi in ((0, 1000), (1000, 3000)):
        ds = mppi_cfg(hi - lo, 0.5)
        m = make_mppi(gpu_pkg, ds)
        m.setWaypoint(*xd)
        nz = torch.from_numpy(noise[lo:hi]).cuda()
        duL = nz[:, :, 0].t().contiguous(); duR = nz[:, :, 1].t().contiguous()
        rec = torch.zeros(T, m.records_per_step, 8, dtype=torch.float64, device="cuda")
        m.shardPartials(x0, duL.data_ptr(), duR.data_ptr(), rec.data_ptr())
        torch.cuda.synchronize()
        shards.append(m); recs.append(rec)
    # pad to a common records-per-step, as an all-gather of equal-size buffers would
    S = max(r.shape[1] for r in recs)
    assert all(m.records_per_step == S or True for m in shards)
    padded = torch.zeros(len(recs), T, S, 8, dtype=torch.float64, device="cuda")
    for g, r in enumerate(recs):
        padded[g, :, : r.shape[1]] = r
    m = make_mppi(gpu_pkg, mppi_cfg(S * 2048, 0.5))  # same S; K only sizes its own buffers
    assert m.records_per_step == S
    m.shardCombine(padded.data_ptr(), len(recs))
    out = m.lastControls()
    assert np.allclose(out, ref["out"], rtol=U_RTOL, atol=U_ATOL)
    assert np.allclose(m.getControls(), ref["u"], rtol=U_RTOL, atol=U_ATOL)


def test_device_resident_noise_and_async_ticks(gpu_pkg):
    """Noise already in HBM in the device layout; 4 enqueued ticks with state carried equal 4
    oracle ticks."""
    import torch
    d = mppi_cfg(512, 0.25)
    T, K = 25, 512
    m = make_mppi(gpu_pkg, d)
    m.setWaypoint(*WAYPOINTS[1])
    u = np.zeros((2, T))
    keep = []
    for t in range(4):
        nz = _noise(100 + t, K, T)
        ref = orc.mppi_new_controls(d, u, (0, 0), WAYPOINTS[1], (0, 0, 0), nz)
        u = ref["u"]
        tz = torch.from_numpy(nz).cuda()
        a, b = tz[:, :, 0].t().contiguous(), tz[:, :, 1].t().contiguous()
        keep.append((a, b))
        m.enqueueDev((0, 0, 0), a.data_ptr(), b.data_ptr())
    out = m.lastControls()
    assert np.allclose(out, ref["out"], rtol=U_RTOL, atol=U_ATOL)
    assert np.allclose(m.getControls(), u, rtol=U_RTOL, atol=U_ATOL)


def test_device_rng_statistics_and_determinism(gpu_pkg):
    """Production noise source (Philox + Box-Muller on the device): N(0, var) moments, the two
    wheels uncorrelated, reproducible in (seed, tick), different across ticks."""
    d = mppi_cfg(4096, 0.5, ul_var=0.9, ur_var=0.4)
    m = make_mppi(gpu_pkg, d)
    m.sampleNoise(1234, 0)
    a, b = m.getNoise()
    m.sampleNoise(1234, 0)
    a2, b2 = m.getNoise()
    m.sampleNoise(1234, 1)
    a3, _ = m.getNoise()
    assert np.array_equal(a, a2) and np.array_equal(b, b2) and not np.array_equal(a, a3)
    n = a.size
    assert abs(a.mean()) < 5 * np.sqrt(0.9 / n) and abs(b.mean()) < 5 * np.sqrt(0.4 / n)
    assert abs(a.var() - 0.9) < 0.02 and abs(b.var() - 0.4) < 0.01
    assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 0.01
    assert abs(np.mean(a ** 4) / a.var() ** 2 - 3.0) < 0.1  # kurtosis of a normal
    # and the tick runs on it
    m.setWaypoint(*WAYPOINTS[1])
    out = m.newControlsDev((0, 0, 0), 0, 0)
    assert np.all(np.isfinite(out))


def test_batched_production_ticks_equal_one_call_per_tick(gpu_pkg):
    """tbnav_mppi_enqueue_rng_batch is n tbnav_mppi_enqueue_rng calls made from C: same seed, same ticks -> the same
    control vector, bit for bit."""
    import torch
    d = mppi_cfg(1024, 0.5)
    a_, b_ = make_mppi(gpu_pkg, d), make_mppi(gpu_pkg, d)
    for m in (a_, b_):
        m.setWaypoint(*WAYPOINTS[2])
    x0 = (0.1, -0.2, 0.3)
    for t in range(12):
        a_.enqueueRng(x0, 99, 500 + t, 0)
    b_.enqueueRngBatch(x0, 99, 500, 5, 0)
    b_.enqueueRngBatch(x0, 99, 505, 7, 0)
    torch.cuda.synchronize()
    assert np.array_equal(a_.getControls(), b_.getControls())
    assert a_.lastControls(0) == b_.lastControls(0)
    a_.close(); b_.close()


def test_full_size_properties_k65536_t100(gpu_pkg):
    """BASELINE configs[3] size on one GPU (K=65536, T=100): size-independent properties —
    (1) permuting the rollouts leaves the update unchanged (soft-min is a symmetric function),
    (2) duplicating the ensemble leaves it unchanged up to the 1e-8 floor's renormalisation,
    (3) J is non-increasing along time (suffix sums of non-negative losses)."""
    import torch
    d = mppi_cfg(65536, 1.0)
    T, K = 100, 65536
    m = make_mppi(gpu_pkg, d)
    m.setWaypoint(*WAYPOINTS[1])
    g = torch.Generator(device="cuda").manual_seed(5)
    duL = torch.randn(T, K, dtype=torch.float64, device="cuda", generator=g) * np.sqrt(0.9)
    duR = torch.randn(T, K, dtype=torch.float64, device="cuda", generator=g) * np.sqrt(0.9)
    out1 = m.newControlsDev((0, 0, 0), duL.data_ptr(), duR.data_ptr())
    u1 = m.getControls()
    J = m.costToGo()
    assert np.all(np.diff(J, axis=0) <= 0.0)
    perm = torch.randperm(K, device="cuda", generator=g)
    pL, pR = duL[:, perm].contiguous(), duR[:, perm].contiguous()
    m.setInitialControls(0.0, 0.0)
    out2 = m.newControlsDev((0, 0, 0), pL.data_ptr(), pR.data_ptr())
    assert np.allclose(out1, out2, rtol=1e-9, atol=1e-12)
    assert np.allclose(u1, m.getControls(), rtol=1e-9, atol=1e-12)
    # oracle spot check on a 256-rollout sub-ensemble of the same arrays
    sub = mppi_cfg(256, 1.0)
    ms = make_mppi(gpu_pkg, sub)
    ms.setWaypoint(*WAYPOINTS[1])
    sL, sR = duL[:, :256].contiguous(), duR[:, :256].contiguous()
    got = ms.newControlsDev((0, 0, 0), sL.data_ptr(), sR.data_ptr())
    nz = torch.stack([sL.t(), sR.t()], dim=2).cpu().numpy()
    ref = orc.mppi_new_controls(sub, np.zeros((2, T)), (0, 0), WAYPOINTS[1], (0, 0, 0), nz)
    assert np.allclose(got, ref["out"], rtol=U_RTOL, atol=U_ATOL)
    assert rel_err(ms.costToGo(), ref["J"]) < J_RTOL


def test_null_arguments_are_rejected(gpu_pkg):
    L = gpu_pkg.capi.lib()
    m = make_mppi(gpu_pkg, mppi_cfg(64, 0.25))
    out = (C.c_double * 2)()
    assert L.tbnav_mppi_new_controls(m._h, None, None, out) == gpu_pkg.capi.ERR_INVALID_ARG
    x0 = (C.c_double * 3)(0, 0, 0)
    dummy = C.c_void_p(8)
    assert L.tbnav_mppi_new_controls_dev(m._h, x0, dummy, None, None, out) == gpu_pkg.capi.ERR_INVALID_ARG


def test_device_sincos_accuracy(gpu_pkg):
    """The rollout kernel's own sin/cos (FMA Cody-Waite + fdlibm kernels, no library call) against libm
    (numpy): <= 1.2e-16 absolute over every range a rollout heading can reach (|x| <= 1e5, quadrant
    boundaries, tiny arguments), graceful (documented) degradation up to 1e12, unit modulus always."""
    L = gpu_pkg.capi.lib()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-40, 40, 200000), rng.uniform(-1e5, 1e5, 50000), rng.uniform(-1e-3, 1e-3, 1000),
                        np.arange(-64, 65) * (np.pi / 4), np.arange(-64, 65) * (np.pi / 2) + 1e-9,
                        [0.0, -0.0, 1e-300, 1e5, -1e5]])
    s = np.empty_like(x); c = np.empty_like(x)
    gpu_pkg.capi.check(L.tbnav_mppi_debug_sincos(x.ctypes.data, x.size, s.ctypes.data, c.ctypes.data), "debug_sincos")
    es, ec = np.abs(s - np.sin(x)), np.abs(c - np.cos(x))
    print(f"\n[device sincos] |x| <= 1e5: max abs err sin {es.max():.3e} cos {ec.max():.3e}")
    assert es.max() <= 1.2e-16 and ec.max() <= 1.2e-16
    big = np.concatenate([rng.uniform(-1e12, 1e12, 20000), [3e7, -2.5e12, 1e15]])
    sb = np.empty_like(big); cb = np.empty_like(big)
    gpu_pkg.capi.check(L.tbnav_mppi_debug_sincos(big.ctypes.data, big.size, sb.ctypes.data, cb.ctypes.data), "debug_sincos")
    eb = np.maximum(np.abs(sb - np.sin(big)), np.abs(cb - np.cos(big)))
    print(f"[device sincos] |x| <= 1e12: max abs err {eb[:20000].max():.3e}; at 3e7 {eb[20000]:.1e}, -2.5e12 {eb[20001]:.1e}, 1e15 {eb[20002]:.1e}")
    assert eb[:20000].max() <= 5e-16
    assert np.allclose(sb * sb + cb * cb, 1.0, atol=1e-12)


def test_division_by_lambda_from_its_reciprocal_is_the_ieee_quotient_bit_for_bit(gpu_pkg):
    """The soft-min weights need (J - min) * -1.0 / lambda with the reference's association (mppi.cpp:117).  The kernels form the
    quotient from lambda's correctly rounded reciprocal — q = x r, e = fma(-q, lambda, x), fma(e, r, q): three dependent
    instructions instead of the division's ~25, and by Markstein's theorem the correctly rounded quotient, i.e. the SAME bits as
    x / lambda — provided nothing under- or overflows on the way (round-4 advisor finding): the statement is for x = 0 and
    2^-900 <= |x| <= 2^900.  Held here against numpy's IEEE division bit for bit inside that range: 2 x 10^5 values per lambda,
    zeros, subnormal QUOTIENTS, and arguments one ulp either side of exact products q * lambda (the nearest a quotient of doubles
    gets to a rounding boundary).  Outside the range (subnormal / huge arguments, infinities) what must hold — and is asserted — is
    that exp() of the two quotients is the same double: the only thing the kernels do with the result.  For the shipped lambda and
    awkward ones; a lambda whose significand is all ones takes the plain division."""
    import ctypes as C
    L = gpu_pkg.capi.lib()
    rng = np.random.default_rng(17)
    edge = np.array([0.0, np.inf, 1e-320, 5e-324, 1.7976931348623157e308, 1.0, 3.0, 1e-8, 0.01, 2.0 ** -1022, 2.0 ** -1074, 2.0 ** -1023,
                     2.0 ** -900, float(np.nextafter(2.0 ** -900, 0.0)), 2.0 ** 900, float(np.nextafter(2.0 ** 900, np.inf)), 1e308, 8.9e307,
                     2.0 ** 1023, 2.0 ** -1000, 3.0 * 2.0 ** -1060, 1.5e-310])
    used_any = False
    for lam in (0.01, 1e-3, 1.0, 3.0, 0.1, 7.3e-5, 1.9999999999999998, 2.0 ** -40, 123456.789, float(np.nextafter(1.0, 0.0)), 2.0 ** 40, 1e-30, 1e30):
        q0 = rng.uniform(0.5, 2.0, 20000) * 2.0 ** rng.integers(-30, 30, 20000)
        prod = q0 * lam
        near = np.concatenate([prod, np.nextafter(prod, np.inf), np.nextafter(prod, 0.0)])
        x = -np.abs(np.concatenate([rng.standard_normal(100000) * 10.0 ** rng.uniform(-300, 300, 100000), rng.uniform(0, 1e6, 99968), near, edge]))
        out = np.empty_like(x); used = C.c_int32()
        assert L.tbnav_mppi_debug_div_lambda(x.ctypes.data, x.size, C.c_double(lam), out.ctypes.data, C.byref(used)) == 0
        with np.errstate(over="ignore", under="ignore"):
            want = x / lam
            inside = (x == 0.0) | ((np.abs(x) >= 2.0 ** -900) & (np.abs(x) <= 2.0 ** 900))
            bad = inside & ~(out == want)
            assert not bad.any(), (lam, x[bad][:5], out[bad][:5], want[bad][:5])
            nz = inside & (x != 0.0)   # (a zero argument gives +0 where the division gives -0: equal as numbers, and exp of both is 1)
            assert np.array_equal(np.signbit(out[nz]), np.signbit(want[nz]))
            assert inside.sum() > 0.85 * x.size and (~inside).sum() > 1000
            assert np.array_equal(np.exp(out[~inside]), np.exp(want[~inside])), lam   # (1 for the tiny arguments, 0 for the huge ones)
        used_any |= bool(used.value)
        if lam in (1.9999999999999998, float(np.nextafter(1.0, 0.0))):
            assert used.value == 0   # significand all ones: the theorem's exception
            assert np.array_equal(out, want)
    assert used_any


def test_nan_state_gives_nan_controls_like_std_clamp(gpu_pkg):
    """mppi.cpp:124-125 clamps with std::clamp, which passes a NaN through (both comparisons are false).  fmin / fmax return the
    OTHER operand: round 4's clamp turned NaN controls into -max_wheel_vel — full reverse — silently (advisor finding)."""
    m = make_mppi(gpu_pkg, mppi_cfg(256, 0.25))
    m.setWaypoint(*WAYPOINTS[1])
    out = m.newControls(float("nan"), 0.0, 0.0, _noise(3, 256, m.steps))
    assert np.all(np.isnan(out)) and np.all(np.isnan(m.getControls()[:, :-1]))
    m.close()


@pytest.mark.parametrize("form", ["prefix", "general"])
@pytest.mark.parametrize("K,horizon", [(32768 + 37, 1.0), (32768, 0.6), (40000, 0.12), (33000, 0.32), (32800, 4.0)])
def test_streaming_rollout_kernel_against_the_oracle(gpu_pkg, K, horizon, form):
    """The large-K rollout kernels (K/64 >= 2 waves per CU-SIMD pair, T a multiple of 4) against the oracle tick.
    "prefix" = mppi_rollout_prefix (exclusive prefixes of the losses written as the rollout goes, exact suffix sums for the last
    4 / 8 / 12 steps, J = total - prefix formed by mppi_partials and the getter); "general" = mppi_rollout_cost, the one fallback
    (any T, either dynamics; round 2's mppi_rollout_cost_reg between the two was removed in round 4).
    T = 100 (late region of one group), 60 (three), 32 (two), 12 (too short for a round: the general kernel either way), 400 (no
    LDS stage could hold it); a ragged last wave; two ticks of warm start."""
    from rtn_amd import capi
    d = mppi_cfg(K, horizon)
    m = make_mppi(gpu_pkg, d, kernel=0)  # (at these sizes the handle's own choice is the time-parallel kernel: tested below)
    T = orc.mppi_steps(d)
    if form == "general":
        m.setOption(capi.MPPI_OPT_PREFIX_FORM, 0)
    want = {"prefix": "mppi_rollout_prefix" if T >= 16 else "mppi_rollout_cost", "general": "mppi_rollout_cost"}[form]
    assert m.steps == T and T % 4 == 0 and m.rollout_kernel == want, m.rollout_kernel
    m.setWaypoint(*WAYPOINTS[2])
    u = np.zeros((2, T)); x0 = (0.3, -0.2, 0.7)
    for tick in range(2):
        ref = _check_tick(m, d, u, (0, 0), WAYPOINTS[2], x0, _noise(90 + tick, K, T))
        u = ref["u"]
        x0 = (x0[0] + 0.002, x0[1] - 0.001, x0[2] + 0.003)
    assert m.lastKernelNames()[0].startswith(want + "<")
    m.close()


@pytest.mark.parametrize("lam", [1e-3, 1.0])
def test_prefix_form_across_lambda_and_an_overflowing_rollout(gpu_pkg, lam):
    """Round-3 advisor findings on the prefix form.  (1) Its prefix rows carry an absolute error of eps * S (S = the rollout's
    whole cost) where the suffix sums carried eps * J(i); the soft-min divides by lambda, so the bound moves with lambda: held
    here at a tenth of the shipped lambda (and at lambda = 1) — J within 1e-12 relative as everywhere, controls within 1e-6
    relative (north star: 1e-5; at the shipped 0.01 every test asserts 1e-9).  (2) A rollout whose total overflows to +inf must
    weigh NOTHING (J = +inf on every row, as the suffix-sum kernels give it) — its prefix rows alone are finite and would make
    it the cheapest rollout of the step: the tick with one such rollout equals the tick with that rollout's noise zeroed out
    of the sums, i.e. finite controls close to the ensemble's."""
    K, horizon = 32768 + 64, 1.0
    d = mppi_cfg(K, horizon, lam=lam)
    T = orc.mppi_steps(d)
    m = make_mppi(gpu_pkg, d, kernel=0)
    assert m.rollout_kernel == "mppi_rollout_prefix"
    m.setWaypoint(*WAYPOINTS[2])
    noise = _noise(5, K, T)
    x0 = (0.3, -0.2, 0.7)
    ref = orc.mppi_new_controls(d, np.zeros((2, T)), (0, 0), WAYPOINTS[2], x0, noise)
    got = m.newControls(*x0, noise)
    assert rel_err(m.costToGo(), ref["J"]) < J_RTOL
    assert np.allclose(got, ref["out"], rtol=1e-6, atol=1e-9) and np.allclose(m.getControls(), ref["u"], rtol=1e-6, atol=1e-9)
    # (2) one rollout with absurd late controls: its loss overflows in the horizon's second half (finite prefixes before that)
    m.setInitialControls(0.0, 0.0)
    bad = noise.copy()
    bad[17, T // 2:, :] = 1e160
    got_bad = m.newControls(*x0, bad)
    J = m.costToGo()
    assert np.all(np.isinf(J[:, 17])) and np.all(np.isfinite(np.delete(J, 17, axis=1)))
    assert np.all(np.isfinite(got_bad)) and np.all(np.isfinite(m.getControls()))
    ref_bad = orc.mppi_new_controls(d, np.zeros((2, T)), (0, 0), WAYPOINTS[2], x0, bad)
    if np.all(np.isfinite(ref_bad["u"])):   # (the reference's own arithmetic on such a rollout: weight exp(-inf) = 0 as well)
        assert np.allclose(m.getControls(), ref_bad["u"], rtol=1e-6, atol=1e-9)
    m.close()


def test_prefix_form_takes_the_general_round_for_large_wheel_speed_differences(gpu_pkg):
    """mppi_rollout_prefix decides ONCE per round of 12 steps whether every |h/2 * r/b * (ur - ul)| <= 2^-5 (straight-line Taylor
    round) or not (the general round with full sincos evaluations).  dt = 0.1 makes the general round the common case
    (|ur - ul| > 3 rad/s is enough), and a sampling variance of 400 mixes both inside one wave."""
    for dt, horizon, var in ((0.1, 4.0, 0.9), (0.01, 0.4, 400.0)):
        d = mppi_cfg(32768 + 64, horizon, dt=dt, ul_var=var, ur_var=var)
        m = make_mppi(gpu_pkg, d, kernel=0)
        T = orc.mppi_steps(d)
        assert m.rollout_kernel == "mppi_rollout_prefix" and T == 40
        m.setWaypoint(*WAYPOINTS[1])
        _check_tick(m, d, np.zeros((2, T)), (0, 0), WAYPOINTS[1], (0.1, 0.0, -0.4), _noise(7, d["rollouts"], T, var))
        m.close()


def test_hip_against_frozen_vectors_G_A1_G_A2(gpu_pkg):
    """The HIP path held against the committed vectors of tests/golden/path_mppi.npz (three
    warm-started cfg1 ticks with everything stored, one cfg2 tick by seed) — no oracle computation in the loop."""
    import os
    import zlib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "path_mppi.npz"))
    d = mppi_cfg(64, 0.25)
    m = make_mppi(gpu_pkg, d)
    m.setWaypoint(*g["a1_xd"])
    for t in range(3):
        assert np.allclose(m.getControls(), g[f"a1_u_before_{t}"], rtol=U_RTOL, atol=U_ATOL)
        got = m.newControls(*g[f"a1_x0_{t}"], g[f"a1_noise_{t}"])
        assert rel_err(m.costToGo(), g[f"a1_J_{t}"]) < J_RTOL
        assert np.allclose(got, g[f"a1_out_{t}"], rtol=U_RTOL, atol=U_ATOL)
        assert np.allclose(m.getControls(), g[f"a1_u_after_{t}"], rtol=U_RTOL, atol=U_ATOL)
    d = mppi_cfg(1024, 0.5)
    m2 = make_mppi(gpu_pkg, d)
    m2.setWaypoint(*WAYPOINTS[1])
    noise = _noise(int(g["a2_seed"]), 1024, 50)
    assert np.uint32(zlib.crc32(np.ascontiguousarray(noise).tobytes())) == g["a2_noise_crc"]
    got = m2.newControls(0.0, 0.0, 0.0, noise)
    J = m2.costToGo()
    assert rel_err(J[0], g["a2_J_row0"]) < J_RTOL and rel_err(J[-1], g["a2_J_last"]) < J_RTOL
    assert np.allclose(got, g["a2_out"], rtol=U_RTOL, atol=U_ATOL) and np.allclose(m2.getControls(), g["a2_u_after"], rtol=U_RTOL, atol=U_ATOL)


@pytest.mark.parametrize("K,horizon,want", [(1024, 0.25, "fused<8"), (2048, 0.5, "fused<8"), (4096, 1.0, "fused<16"), (8192, 1.0, "fused<16"), (16384 + 5, 0.5, "scan"),
                                            (40000, 0.25, "scan"), (45000, 0.12, "cost"), (45000, 0.6, "prefix")])
def test_default_kernel_choice_by_ensemble_size_against_the_oracle(gpu_pkg, K, horizon, want):
    """What the handle picks by itself across ensemble sizes (fused one-wave-per-rollout kernel while the chip would be
    empty otherwise, the time-parallel kernel in the middle, the sequential streaming kernel from ~2.5 one-wave workgroups
    per CU), each against the oracle: a resident-noise tick, then a device-noise tick that must equal sample-then-tick."""
    d = mppi_cfg(K, horizon)
    m = make_mppi(gpu_pkg, d)
    assert want in m.rollout_kernel, m.rollout_kernel
    T = orc.mppi_steps(d)
    m.setWaypoint(*WAYPOINTS[1])
    ref = _check_tick(m, d, np.zeros((2, T)), (0, 0), WAYPOINTS[1], (0.01, 0.0, 0.05), _noise(7, K, T))
    # production tick: in-kernel noise (K <= 8192) or sample + tick — either way the values tbnav_mppi_sample_noise writes
    m2 = make_mppi(gpu_pkg, d); m2.setWaypoint(*WAYPOINTS[1]); m2.setControls(m.getControls())
    m.sampleNoise(11, 3)
    a = m.newControlsDev((0.02, 0.0, 0.06), 0, 0)
    b = m2.newControlsRng((0.02, 0.0, 0.06), 11, 3)
    assert np.array_equal(np.array(a), np.array(b)) and np.array_equal(m.getControls(), m2.getControls())  # same kernels on both sides


@pytest.mark.parametrize("K,horizon", [(1024, 0.5), (1500, 0.25), (4096, 1.0)])
def test_batch_enqueue_by_graph_replay_equals_tick_by_tick(gpu_pkg, K, horizon):
    """tbnav_mppi_enqueue_rng_batch replays whole chunks of 100 ticks from a captured hipGraph (the tick number comes from device
    memory); the result must be the one of launching every tick by itself — bit for bit, over a run that is not a multiple of
    the chunk, called twice (the second call reuses the graph), and after set_controls (first tick unshifted)."""
    import torch
    from rtn_amd import capi
    d = mppi_cfg(K, horizon)
    ma, mb = make_mppi(gpu_pkg, d), make_mppi(gpu_pkg, d)
    mb.setOption(capi.MPPI_OPT_BATCH_GRAPH, 0)
    side = torch.cuda.Stream()
    st = side.cuda_stream
    x0 = (0.05, -0.02, 0.3)
    for m in (ma, mb):
        m.setWaypoint(*WAYPOINTS[1])
    for first, n in ((0, 250), (250, 330), (580, 4), (584, 23), (607, 10), (617, 11), (628, 1), (629, 20)):   # long chunks, short chunks, both parities of the double buffer
        ma.enqueueRngBatch(x0, 99, first, n, st)
        for i in range(n):
            mb.enqueueRng(x0, 99, first + i, st)
        torch.cuda.synchronize()
        assert ma.lastControls(st) == mb.lastControls(st)
        assert np.array_equal(ma.getControls(), mb.getControls())
    u = np.linspace(-1.0, 1.0, 2 * ma.steps).reshape(2, ma.steps)
    ma.setControls(u); mb.setControls(u)
    ma.enqueueRngBatch(x0, 7, 1000, 205, st)
    mb.enqueueRngBatch(x0, 7, 1000, 205, st)
    torch.cuda.synchronize()
    assert np.array_equal(ma.getControls(), mb.getControls())


def test_short_batches_replay_a_graph_of_their_own_length(gpu_pkg):
    """A batch shorter than a chunk of 100 — what is left of a long one, or the 20 ticks a harness times between two
    synchronisations — is replayed from ONE graph of its own length once a second batch in a row asks for the same length on the
    same parity of the controls' double buffer (csrc/mppi.hip: tgs).  Bit for bit the ticks launched one by one: repeated lengths
    (the graph is built by the second and replayed from then on), odd lengths (one plain tick behind the replay flips the parity,
    one in front of the next brings it back), a length change, a long batch with a short rest, a parameter change in between."""
    import torch
    from rtn_amd import capi
    d = mppi_cfg(1024, 0.5)
    ma, mb = make_mppi(gpu_pkg, d), make_mppi(gpu_pkg, d)
    mb.setOption(capi.MPPI_OPT_BATCH_GRAPH, 0)
    side = torch.cuda.Stream()
    st = side.cuda_stream
    x0 = (0.05, -0.02, 0.3)
    for m in (ma, mb):
        m.setWaypoint(*WAYPOINTS[1])
    first = 0
    replayed = [ma.graphReplayedTicks()]
    plan = [5] + [20] * 5 + [21] * 4 + [8] * 3 + [7] * 2 + [130] * 3 + [20] * 3
    for j, n in enumerate(plan):
        if j == 8:
            ma.setWaypoint(*WAYPOINTS[2]); mb.setWaypoint(*WAYPOINTS[2])   # (baked into the graphs: both are rebuilt)
        ma.enqueueRngBatch(x0, 31, first, n, st)
        for i in range(n):
            mb.enqueueRng(x0, 31, first + i, st)
        torch.cuda.synchronize()
        assert ma.lastControls(st) == mb.lastControls(st), (j, n)   # (the state carries: one wrong tick shows in every later one)
        first += n
        replayed.append(ma.graphReplayedTicks())
    assert np.array_equal(ma.getControls(), mb.getControls())
    # (getControls applies the owed shift for real: the next batch starts with one plain tick, its graph is one of 18 — same results)
    for n in (20, 20, 20):
        ma.enqueueRngBatch(x0, 31, first, n, st)
        for i in range(n):
            mb.enqueueRng(x0, 31, first + i, st)
        torch.cuda.synchronize()
        assert np.array_equal(ma.getControls(), mb.getControls())
        first += n
    per_call = np.diff(replayed)
    assert list(per_call[1:6]) == [0, 20, 20, 20, 20]          # the first batch of 20 asks, the second builds, all replay from then on
    assert list(per_call[6:10]) == [20, 20, 20, 20]            # 21 ticks: the graph of 20 + one plain tick, behind or in front (rebuilt at once after the waypoint change: the wish stands)
    assert list(per_call[10:13]) == [0, 8, 8] and list(per_call[13:15]) == [0, 0]   # 7 ticks: below the shortest graph
    assert list(per_call[15:18]) == [100, 130, 130]            # 100 by the chunk graph, the rest of 30 by its own from the second call
    assert mb.graphReplayedTicks() == 0
    ma.close(); mb.close()


def test_graph_replay_is_rebuilt_when_a_baked_parameter_changes(gpu_pkg):
    """The captured graph of ticks holds the waypoint, uinit, dynamics, trig, the rng shard and the record buffer BY VALUE
    (round-2 advisor finding): every setter that changes one of them must invalidate it.  Batch, change one, batch again —
    against a handle that launches every tick by itself."""
    import torch
    from rtn_amd import capi
    d = mppi_cfg(1024, 0.5)
    ma, mb = make_mppi(gpu_pkg, d), make_mppi(gpu_pkg, d)
    mb.setOption(capi.MPPI_OPT_BATCH_GRAPH, 0)
    side = torch.cuda.Stream()
    st = side.cuda_stream
    x0 = (0.05, -0.02, 0.3)
    changes = [
        lambda m: m.setWaypoint(*WAYPOINTS[1]),
        lambda m: m.setWaypoint(*WAYPOINTS[2]),
        lambda m: m.setOption(capi.MPPI_OPT_TRIG, 3),
        lambda m: m.setRngShard(2048, 8192),
        lambda m: m.setDynamics("arc"),
        lambda m: m.setDynamics("rk4"),
        lambda m: m.setOption(capi.MPPI_OPT_KERNEL, -8),   # frees and reallocates the record buffer the graph wrote to
        lambda m: m.setOption(capi.MPPI_OPT_KEEP_J, 1),
    ]
    first = 0
    for ch in changes:
        ch(ma); ch(mb)
        ma.enqueueRngBatch(x0, 5, first, 230, st)
        for i in range(230):
            mb.enqueueRng(x0, 5, first + i, st)
        torch.cuda.synchronize()
        assert ma.lastControls(st) == mb.lastControls(st)
        assert np.array_equal(ma.getControls(), mb.getControls())
        first += 230
    # setInitialControls changes uinit (read by the shifted warm start of every later tick)
    ma.setInitialControls(0.3, -0.2); mb.setInitialControls(0.3, -0.2)
    ma.enqueueRngBatch(x0, 5, first, 230, st); mb.enqueueRngBatch(x0, 5, first, 230, st)
    torch.cuda.synchronize()
    assert np.array_equal(ma.getControls(), mb.getControls())
    # the same graph on ANOTHER stream: the device tick word of the first stream's replay must not be trusted
    other = torch.cuda.Stream()
    ma.enqueueRngBatch(x0, 5, first + 230, 200, other.cuda_stream); mb.enqueueRngBatch(x0, 5, first + 230, 200, other.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(ma.getControls(), mb.getControls())
    ma.close(); mb.close()


def test_configs3_as_written_k65536_t100_split_8_ways_against_the_full_cpu_oracle(gpu_pkg):
    """BASELINE configs[3] AS WRITTEN: K = 65536, T = 100, the ensemble split 8 ways (8192 rollouts per shard: what each of
    8 GPUs runs) through the product's own sharded tick (tbnav_mppi_group: every member's rollouts + records, one all-gather,
    the combine of all 8 record sets on every member; the 8 members share the one device of this box, so the all-gather is the
    in-process copy transport — on 8 devices the same code issues one grouped ncclAllGather) — against
    (1) the unsharded K = 65536 tick on one handle (mppi_rollout_prefix + mppi_partials + mppi_combine) and
    (2) the CPU oracle's tick over all 65536 rollouts (mppi.cpp:72-140 restated; OpenMP over rollouts).
    Two ticks, warm start carried."""
    from rtn_amd.mppi import MPPIGroup, CartModel, LossFunc
    K, P, horizon = 65536, 8, 1.0
    d = mppi_cfg(K, horizon)
    T = orc.mppi_steps(d)
    Ks = K // P
    x0 = (0.02, -0.01, 0.05)
    whole = make_mppi(gpu_pkg, d)
    assert whole.rollout_kernel == "mppi_rollout_prefix"
    grp = MPPIGroup(CartModel(d["wheel_radius"], d["wheel_base"]), LossFunc(d["Q"], d["R"], d["P1"]), d["lam"], d["max_wheel_vel"],
                    d["ul_var"], d["ur_var"], horizon, d["dt"], K, devices=[0] * P)
    members = [grp.member(r) for r in range(P)]
    assert all(m.rollouts == Ks for m in members)
    whole.setWaypoint(*WAYPOINTS[1]); grp.setWaypoint(*WAYPOINTS[1])
    orc.lib().orc_set_threads(8)
    try:
        u = np.zeros((2, T))
        for tick in range(2):
            noise = np.random.default_rng(500 + tick).standard_normal((K, T, 2)) * np.sqrt(0.9)
            ref = orc.mppi_new_controls(d, u, (0, 0), WAYPOINTS[1], x0, noise)
            got_whole = whole.newControls(*x0, noise)
            assert rel_err(whole.costToGo(), ref["J"]) < J_RTOL
            assert np.allclose(got_whole, ref["out"], rtol=U_RTOL, atol=U_ATOL)
            assert np.allclose(whole.getControls(), ref["u"], rtol=U_RTOL, atol=U_ATOL)
            got = grp.newControls(*x0, noise)
            assert np.allclose(got, ref["out"], rtol=U_RTOL, atol=U_ATOL)
            u0 = members[0].getControls()
            assert np.allclose(u0, ref["u"], rtol=U_RTOL, atol=U_ATOL)
            for g, m in enumerate(members):
                assert rel_err(m.costToGo(), ref["J"][:, g * Ks:(g + 1) * Ks]) < J_RTOL
                assert np.array_equal(m.getControls(), u0)   # every rank ends with the same warm start, bit for bit
            u = ref["u"]
            whole.setControls(u); grp.setControls(u)   # carry the ORACLE's warm start on both sides (the comparison is per tick)
            x0 = (x0[0] + 0.002, x0[1] + 0.001, x0[2] + 0.004)
    finally:
        orc.lib().orc_set_threads(1)
    whole.close(); grp.close()
