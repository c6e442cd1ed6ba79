"""RBPF half of __graft_entry__.smoke(): one small scan update on cuda:0 checked against the oracle."""
import numpy as np

import oracle_api as orc
import rbpf_cases as rc


def run():
    from rtn_amd.rbpf import ParticleFilter, default_params
    N, k = 8, 10
    pf_o = orc.PfAPI(orc.pf_params(N=N, k=k)); pf_d = ParticleFilter(default_params(N=N, k=k))
    steps, poses = rc.trajectory(2, inc=(0.03, 0.02, 0.02))
    rng = np.random.default_rng(0)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng)
        normals = orc.normal_stream(s, pf_o.normals_per_scan(True), 0.0, 1.0)
        for p in range(N):
            pf_d.setOccDist(p, pf_o.grid(p).dump()["occ_dist"])
        tr = pf_o.slam(scan, u, cur, prev, True, t_icp, normals)
        st = pf_d.SLAM(scan, u, cur, prev, True, t_icp, normals)
        assert (st.neff, st.resampled) == (tr["neff"], tr["resampled"])
        po, _, wo = pf_o.particles(); pd, _, wd = pf_d.particles()
        assert np.allclose(pd, po, rtol=1e-10, atol=1e-15) and np.allclose(wd, wo, rtol=1e-9)
        for p in range(N):
            assert np.array_equal(pf_d.logOdds(p), pf_o.grid(p).dump()["log_odds"])
    print("smoke RBPF ok: Neff", st.neff)
