"""Dev: randomized sweep of the REFERENCE distance-field mode (the lazy brushfire, csrc/ref_field.hpp) against the oracle's filter,
whose brushfire always runs to the end — nothing injected.  Every case draws its own ensemble size, samples per particle, grid,
reach (TBNAV_RBPF_OPT_REF_REACH, 0 = eager), sampling spread (how many particles share a state), rooms that CHANGE in mid-run (walls
that appear beyond the map: passes are resumed, proposals rerun), gated beams and empty scans, a forced resampling, ICP failures,
whole-field exports in mid-run (passes finished, lineages replayed).
usage: python tools/fuzz_reffield.py [n_cases] [seed] [only_case]      prints one line per failing case; exit code = failures"""
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
g.load_package()
import oracle_api as orc  # noqa: E402
import rbpf_cases as rc  # noqa: E402
from rtn_amd import capi  # noqa: E402
from rtn_amd.rbpf import ParticleFilter, default_params  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
totals = dict(passes=0, states_resumed=0, proposals_rerun=0, lineages_replayed=0, passes_completed=0)


def case(i):
    rng = np.random.default_rng([seed, 7, i])
    N = int(rng.choice([1, 3, 8, 20, 40]))
    k = int(rng.choice([4, 7, 20, 50]))
    half = float(rng.choice([2.0, 3.0, 5.0, 10.0]))
    if half == 10.0:
        N = min(N, 8)   # (the oracle's eager brushfire is 16 ms per particle and scan at 400 x 400)
    reach = int(rng.choice([0, 1, 1, 1, 2, 3, 6]))
    spread = float(rng.choice([1e-8, 1e-8, 1e-5, 1e-3]))
    n_scans = int(rng.integers(4, 8))
    small = (-float(rng.uniform(0.7, 1.2)), float(rng.uniform(0.7, 1.2)), -float(rng.uniform(0.6, 1.0)), float(rng.uniform(0.6, 1.1)))
    big = tuple(float(np.clip(w * rng.uniform(1.5, 2.4), -half + 0.3, half - 0.3)) for w in small)
    change_at = int(rng.integers(1, n_scans)) if rng.random() < 0.7 else n_scans + 1
    resample_at = int(rng.integers(1, n_scans)) if rng.random() < 0.6 and N >= 3 else -1
    icp_fail_at = int(rng.integers(0, n_scans)) if rng.random() < 0.3 else -1
    empty_at = int(rng.integers(0, n_scans)) if rng.random() < 0.2 else -1
    export_at = int(rng.integers(0, n_scans)) if rng.random() < 0.5 else -1
    desc = dict(case=i, N=N, k=k, half=half, reach=reach, spread=spread, n_scans=n_scans, change_at=change_at, resample_at=resample_at,
                icp_fail_at=icp_fail_at, empty_at=empty_at, export_at=export_at)
    extra = dict(sample_range=[spread * 0.1, spread, spread])
    pf_o = orc.PfAPI(orc.pf_params(N=N, k=k, map_min=-half, map_max=half, **extra))
    pf_d = ParticleFilter(default_params(N=N, k=k, map_min=-half, map_max=half, **extra), df_mode="reference")
    pf_d.setOption(capi.RBPF_OPT_REF_REACH, reach)
    inc = (float(rng.uniform(-0.05, 0.05)), float(rng.uniform(0.01, 0.05)), float(rng.uniform(-0.03, 0.03)))
    steps, poses = rc.trajectory(n_scans, inc=inc, start=(float(rng.uniform(-3.1, 3.1)), 0.0, 0.0))
    srng = np.random.default_rng([seed, 8, i])
    try:
        for s, (prev, cur, t_icp, u) in enumerate(steps):
            scan = orc.room_scan(poses[s], walls=small if s < change_at else big, rng=srng)
            if s == empty_at:
                scan[:] = 0.01
            elif rng.random() < 0.3:
                scan[srng.integers(0, 360, 25)] = 9.0   # gated beams
            icp_ok = s != icp_fail_at
            normals = orc.normal_stream(31 * i + s, pf_o.normals_per_scan(icp_ok), 0.0, 1.0)
            if s == resample_at:
                w = np.full(N, 0.2 / N); w[0] += 0.5; w[N // 2] += 0.3; w /= w.sum()
                pf_o.set_particles(w=w); pf_d.setParticles(w=w)
            tr_o = pf_o.slam(scan, u, cur, prev, icp_ok, t_icp, normals)
            try:
                st = pf_d.SLAM(scan, u, cur, prev, icp_ok, t_icp, normals)
            except capi.TbnavError as e:   # the wrapper raises on a non-zero status: the oracle must have refused the same scan the same way
                assert getattr(e, "status", None) == tr_o["rc"] != 0, ("status", s, getattr(e, "status", None), tr_o["rc"])
                totals["refused_alike"] = totals.get("refused_alike", 0) + 1
                return None
            assert st.status == 0 and tr_o["rc"] == 0, ("status", s, st.status, tr_o["rc"])
            tr_d = pf_d.trace()
            po, pvo, wo = pf_o.particles(); pd, pvd, wd = pf_d.particles()
            assert np.allclose(pd, po, rtol=1e-9, atol=1e-12), ("pose", s, float(np.abs(pd - po).max()))
            assert np.allclose(wd, wo, rtol=1e-8), ("weight", s, float(np.max(np.abs(wd - wo) / np.abs(wo))))
            if icp_ok:
                assert np.allclose(tr_d["p_scan"], tr_o["p_scan"], rtol=1e-9), ("p_scan", s)
            inv = 1.0 / tr_o["sq_sum"]
            if abs(inv - round(inv)) > 1e-9:   # (Neff = (int)(1 / sum w^2) on a knife edge can go either way on a last-bit difference)
                assert (st.neff, st.resampled) == (tr_o["neff"], tr_o["resampled"]), ("neff", s, st.neff, tr_o["neff"])
            if st.resampled:
                assert np.array_equal(tr_d["resample_idx"], tr_o["resample_idx"]), ("parents", s)
            if s == export_at:
                p = int(rng.integers(0, N))
                assert np.array_equal(pf_d.occDist(p), pf_o.grid(p).dump()["occ_dist"]), ("field in mid-run", s, p)
        for p in range(N):
            gd = pf_o.grid(p).dump()
            assert np.array_equal(pf_d.logOdds(p), gd["log_odds"]), ("log-odds", p)
            assert np.array_equal(pf_d.occDist(p), gd["occ_dist"]), ("field", p)
        st_ = pf_d.referenceFieldStats()
        for key in ("passes", "states_resumed", "proposals_rerun", "lineages_replayed", "passes_completed"):
            totals[key] += st_[key]
        return None
    except Exception as e:  # noqa: BLE001
        return (desc, repr(e), traceback.format_exc().splitlines()[-3:])
    finally:
        pf_d.close()


fails = 0
for i in ([only] if only >= 0 else range(n_cases)):
    r = case(i)
    if r is not None:
        fails += 1
        print("[FAIL]", r[0], r[1], flush=True)
        if only >= 0:
            print("\n".join(r[2]))
    if (i + 1) % 10 == 0 or only >= 0:
        print(f"reffield: {i + 1} cases done, failures so far {fails}; {totals}", flush=True)
print(f"reffield: {n_cases if only < 0 else 1} cases done, failures so far {fails}; {totals}", flush=True)
sys.exit(1 if fails else 0)
