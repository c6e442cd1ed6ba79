"""Dev: randomized sweep of the in-library sharded RBPF scan (tbnav_rbpf_group, 2..8 members on device 0) against ONE handle
holding all the particles — bit for bit: random shard counts and sizes, k, sensor spreads that make the run resample by itself,
forced skews (one or several heavy particles anywhere), host normals or device noise, ICP failures, 80x80 / 120x120 / 400x400
maps, tile pools small enough to matter.  Run on the GPU box; not part of the suite.  usage: python tools/fuzz_group.py [n] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
g.load_package()
import oracle_api as orc, rbpf_cases as rc
from rtn_amd.rbpf import ParticleFilter, ParticleFilterGroup, default_params

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fails = resamples = 0
for i in range(n_cases):
    rng = np.random.default_rng([seed, 21, i])
    P = int(rng.integers(2, 9))
    nl = int(rng.choice([1, 2, 3, 5, 8, 16, 40]))
    N, k = P * nl, int(rng.choice([2, 7, 20, 50]))
    half = float(rng.choice([2.0, 3.0, 10.0]))
    spread = float(rng.choice([1e-8, 1e-6, 1e-4]))
    dev_noise = bool(rng.random() < 0.5)
    n_scans = int(rng.integers(3, 8))
    kw = dict(map_min=-half, map_max=half, sample_range=[spread * 0.01, spread, spread], scan_likelihood_max=float(rng.choice([20.0, 1e6])))
    desc = dict(i=i, P=P, nl=nl, k=k, half=half, spread=spread, dev_noise=dev_noise, n_scans=n_scans)
    try:
        grp = ParticleFilterGroup(default_params(N=N, k=k, **kw), [0] * P)
        pf = ParticleFilter(default_params(N=N, k=k, **kw))
        grp.setSeed(1000 + i); pf.setSeed(1000 + i)
        steps, poses = rc.trajectory(n_scans, inc=(0.03, 0.02, 0.02))
        walls = rc.ROOM_SMALL if half < 5 else rc.ROOM_SURVEY
        for s, (prev, cur, t_icp, u) in enumerate(steps):
            scan = orc.room_scan(poses[s], walls=walls, rng=rng)
            ok = bool(rng.random() < 0.85)
            if rng.random() < 0.35 and N > 1:
                w = rng.random(N) * 0.01
                for _ in range(int(rng.integers(1, 4))):
                    w[int(rng.integers(0, N))] += float(rng.uniform(0.2, 1.0))
                w /= w.sum()
                pf.setParticles(w=w); grp.setParticles(w=w)
            nz = None if dev_noise else orc.normal_stream(77 * i + s, N * (3 * k + 3 if ok else 3) + 1, 0.0, 1.0)
            a = grp.SLAM(scan, u, cur, prev, ok, t_icp, nz)
            b = pf.SLAM(scan, u, cur, prev, ok, t_icp, nz)
            assert (a.status, a.neff, a.resampled, a.sum_w, a.sq_sum) == (b.status, b.neff, b.resampled, b.sum_w, b.sq_sum), ("stats", s, desc)
            resamples += a.resampled
            for x, y, name in zip(grp.particles(), pf.particles(), ("pose", "prev", "weight")):
                assert np.array_equal(x, y), (name, s, desc)
            for p in rng.choice(N, size=min(N, 6), replace=False):
                assert np.array_equal(grp.logOdds(int(p)), pf.logOdds(int(p))), ("map", int(p), s, desc)
        assert grp.getRobotState() == pf.getRobotState(), ("best", desc)
        grp.close(); pf.close()
    except AssertionError as e:
        fails += 1
        print("FAIL", e.args[0] if e.args else e, flush=True)
    except Exception as e:  # noqa: BLE001 (a status code the reference would throw, e.g. eta is 0 with k = 2: both sides must agree on it)
        print("EXC", type(e).__name__, e, desc, flush=True)
        fails += 1
print(f"fuzz_group: {n_cases} cases, seed {seed}: {fails} failures, {resamples} resampling scans")
