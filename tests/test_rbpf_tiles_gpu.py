"""GPU tests of the tiled copy-on-write maps: a resample copies tile tables, the next scan
clones only the tiles it writes, tiles nobody names return to the pool — and the log-odds stay the reference's bits
(particle_filter.cpp:468-500 deep copies, grid_mapper.cpp:140-178 updates)."""
import os

import numpy as np
import pytest

import oracle_api as orc
import rbpf_cases as rc

pytestmark = pytest.mark.gpu


def _dev(gpu_pkg, pool_bytes=0, df_mode=None, **kw):
    from rtn_amd.rbpf import ParticleFilter, default_params
    return ParticleFilter(default_params(**kw), pool_bytes=pool_bytes, df_mode=df_mode)


def _oracle_map(grid, laser, scans, poses):
    """log-odds after integrating `scans` at `poses` (theta, x, y) with the oracle's GridMapper (no brushfire)."""
    g = orc.GridAPI("orc", grid=grid, laser=laser)
    for sc, po in zip(scans, poses):
        g.integrate_scan(sc, po, esdf=False)
    lo = g.dump()["log_odds"].copy()
    g.close()
    return lo


def test_resample_shares_tiles_and_the_next_scan_clones_what_it_writes(gpu_pkg):
    N, k, n_scans = 64, 8, 5
    pf = _dev(gpu_pkg, N=N, k=k, map_min=-10.0, map_max=10.0)
    cap, free0, tile_bytes = pf.poolStats()
    assert tile_bytes == 8192 and free0 == cap
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(3)
    scans = [orc.room_scan(poses[s], walls=rc.ROOM_SURVEY, rng=rng) for s in range(n_scans)]
    hist, used, parents = [], [], None
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        normals = orc.normal_stream(300 + s, pf.numNormals(True), 0.0, 1.0)
        if s == 2:  # force a resample: two heavy particles
            w = np.full(N, 1e-3); w[5] = 0.7; w[40] = 0.2; w /= w.sum()
            pf.setParticles(w=w)
        st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, normals)
        assert st.status == 0
        hist.append(pf.trace()["new_pose"].copy())  # the pose each slot integrated this scan at (before any resample)
        used.append(cap - pf.poolStats()[1])
        if s == 2:
            assert st.resampled == 1
            parents = pf.trace()["resample_idx"].copy()
        else:
            assert st.resampled == 0
    # scan 0 allocates the footprint, scans over the same area reuse it
    assert used[0] > 0 and used[1] - used[0] < 0.2 * used[0]
    # the resample freed the tiles of the particles that died: far fewer lineages are alive
    n_alive = len(set(parents.tolist()))
    assert n_alive < N // 2 and used[2] <= used[1] * (n_alive + 1) / N + 8
    # the scan after it cloned the shared tiles each slot wrote (about one footprint per slot again) ...
    assert used[3] > used[2] and used[3] <= used[2] + 1.2 * used[0]
    # ... and the one after that found them private already
    assert used[4] - used[3] < 0.2 * used[0]
    # bits: every slot's map equals the oracle's GridMapper fed the slot's lineage of poses
    grid = (0.05, -10.0, 10.0, -10.0, 10.0)
    for m in (0, 5, 17, 40, N - 1):
        q = int(parents[m])
        lineage = [hist[0][q], hist[1][q], hist[2][q], hist[3][m], hist[4][m]]
        assert np.array_equal(pf.logOdds(m), _oracle_map(grid, None, scans, lineage)), m
    pf.close()


def test_pool_exhaustion_is_reported_not_silent(gpu_pkg):
    from rtn_amd import capi
    pf = _dev(gpu_pkg, pool_bytes=8192 * 100, N=16, k=4, map_min=-10.0, map_max=10.0)  # 16 particles x ~25 tiles each do not fit
    steps, poses = rc.trajectory(1)
    prev, cur, t_icp, u = steps[0]
    scan = orc.room_scan(poses[0], walls=rc.ROOM_SURVEY, rng=np.random.default_rng(1))
    st = pf.SLAM(scan, u, cur, prev, True, t_icp, orc.normal_stream(1, pf.numNormals(True), 0.0, 1.0), check=False)
    assert st.status == capi.ERR_POOL_EXHAUSTED
    pf.close()


def test_configs4_shard_10000_particles_2000x2000_1080_beams(gpu_pkg):
    """BASELINE configs[4] per-GPU shard at scale: 10 000 particles on a 2000 x 2000 grid (dense that would be
    320 GB of log-odds), 1080-beam scans, three scans with a forced resample after the second; spot-checked bit for
    bit against the oracle's GridMapper along each spot particle's lineage."""
    N, k, n_scans, bd = 10000, 4, 3, 1.0 / 3.0
    pf = _dev(gpu_pkg, pool_bytes=24 << 30, N=N, k=k, map_min=-50.0, map_max=50.0, beam_delta_deg=bd)
    assert (pf.xsize, pf.ysize) == (2000, 2000)
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(8)
    scans = [orc.room_scan(poses[s], n_beams=1080, beam_delta_deg=bd, walls=rc.ROOM_SURVEY, rng=rng) for s in range(n_scans)]
    hist, parents = [], None
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        normals = np.random.default_rng(40 + s).standard_normal(pf.numNormals(True))
        if s == 1:
            w = np.full(N, 1e-6); w[[7, 1234, 5000, 9999]] = [0.4, 0.3, 0.2, 0.1]; w /= w.sum()
            pf.setParticles(w=w)
        st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, normals)
        assert st.status == 0 and st.n_valid_beams > 600
        hist.append(pf.trace()["new_pose"].copy())
        if s == 1:
            assert st.resampled == 1
            parents = pf.trace()["resample_idx"].copy()
    assert {7, 1234, 5000, 9999} <= set(parents.tolist()) and len(set(parents.tolist())) < 300  # the four heavy ones + ~1 % of the rest
    cap, free, _ = pf.poolStats()
    assert cap - free < 120 * N  # a few tens of tiles per particle, not 3969
    grid = (0.05, -50.0, 50.0, -50.0, 50.0)
    laser = orc.lds01_laser(bd)
    for m in (0, 4321, N - 1):
        q = int(parents[m])
        want = _oracle_map(grid, laser, scans, [hist[0][q], hist[1][q], hist[2][m]])
        assert np.array_equal(pf.logOdds(m), want), m
    (pose, idx) = pf.getRobotState()
    assert 0 <= idx < N and np.all(np.isfinite(pose))
    # a particle of this shard travels as its tiles (log-odds + occupancy bits): ~1 MB, not a 32 MB map + 512 KB bitmap
    import ctypes as C
    nbytes = C.c_uint64()
    assert pf._L.tbnav_rbpf_export_size(pf._h, 4321, C.byref(nbytes)) == 0
    assert nbytes.value < 120 * (8192 + 128) + 65536, nbytes.value
    pf.close()


@pytest.mark.parametrize("variant", ["box512", "box1024", "box_bands12", "box_bands5", "box512_cell16", "box_bands5_cell16", "beam_ordered", "form1"])
def test_raycast_kernel_variants_are_bit_exact(gpu_pkg, variant):
    """Every form of the map update the library ships — the box-counter kernel with 512 and 1024 threads, the same working its
    box through in bands of rows, and the beam-ordered kernel (by either option) — must leave the oracle's GridMapper bits.
    Scans with close obstacles (many events per end-point cell: the slot overflow path) and a long corridor.  (Round 2's first
    tile kernel, rbpf_raycast_tile, was removed in round 4: nothing selected it any more.)"""
    from rtn_amd import capi
    N, k, n_scans = 24, 6, 4
    pf = _dev(gpu_pkg, N=N, k=k, map_min=-10.0, map_max=10.0)
    cell16 = variant.endswith("_cell16")   # the 16-bit cell form (slots by table look-up), forced wherever it can run
    if cell16:
        variant = variant[:-len("_cell16")]
        pf.setOption(capi.RBPF_OPT_RAYCAST_CELL16, 2)
        pf.setOption(capi.RBPF_OPT_RAYCAST_THREADS, 512)
    if variant == "beam_ordered":
        pf.setOption(capi.RBPF_OPT_RAYCAST_ORDERED, 1)
    elif variant.startswith("box_bands"):  # the box-counter kernel working the box through in bands of ~12 / ~5 rows
        pf.setOption(capi.RBPF_OPT_RAYCAST_BAND_ROWS, int(variant[9:]))
    elif variant == "form1":   # the retired switch: 1 is refused loudly (it used to fall through to the beam-ordered kernel), 0 is the box kernel
        with pytest.raises(Exception):
            pf.setOption(capi.RBPF_OPT_RAYCAST_FORM, 1)
        pf.setOption(capi.RBPF_OPT_RAYCAST_FORM, 0)
    else:
        pf.setOption(capi.RBPF_OPT_RAYCAST_THREADS, int(variant[3:]))
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(12)
    scans = [orc.room_scan(poses[s], walls=(-0.4, 3.2, -0.35, 0.5), rng=rng) for s in range(n_scans)]  # a wall 7 cells away
    hist = []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, orc.normal_stream(60 + s, pf.numNormals(True), 0.0, 1.0))
        assert st.status == 0
        hist.append(pf.trace()["new_pose"].copy())
    grid = (0.05, -10.0, 10.0, -10.0, 10.0)
    for m in (0, 11, N - 1):
        assert np.array_equal(pf.logOdds(m), _oracle_map(grid, None, scans, [h[m] for h in hist])), (variant, m)
    if cell16:
        assert ", true, " in pf.lastKernelNames()[1], pf.lastKernelNames()
    pf.close()


def test_box_kernel_residency_forms_are_bit_exact_and_chosen_by_the_boxes_need(gpu_pkg):
    """rbpf_raycast_box sizes its LDS array by what the particles' boxes needed two scans ago and picks the residency that
    fits: <1024, 8> or <512, 6> (two / three per CU, by the scan's longest beam) before any need is known, then <512, 8, ., 8 or 4> (FOUR per CU, round 4; 4 = four-event slots, where only those fit) for a room whose box + 256 words
    fits a quarter of a CU's LDS, <512, 6> (three) when TBNAV_RBPF_OPT_RAYCAST_ADAPT = 2 forbids four.  Every form leaves the
    oracle's GridMapper bits."""
    from rtn_amd import capi
    N, k, n_scans = 24, 6, 6
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(12)
    scans = [orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng) for s in range(n_scans)]
    grid = (0.05, -10.0, 10.0, -10.0, 10.0)
    seen = {}
    for adapt in (1, 2):
        pf = _dev(gpu_pkg, N=N, k=k, map_min=-10.0, map_max=10.0)
        pf.setOption(capi.RBPF_OPT_RAYCAST_ADAPT, adapt)
        hist, names = [], []
        for s, (prev, cur, t_icp, u) in enumerate(steps):
            st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, orc.normal_stream(60 + s, pf.numNormals(True), 0.0, 1.0))
            assert st.status == 0
            hist.append(pf.trace()["new_pose"].copy())
            names.append(pf.lastKernelNames()[1])
        seen[adapt] = names
        for m in (0, 11, N - 1):
            assert np.array_equal(pf.logOdds(m), _oracle_map(grid, None, scans, [h[m] for h in hist])), (adapt, m)
        pf.close()
    assert not seen[1][0].startswith("rbpf_raycast_box<512, 8") and seen[1][-1] == "rbpf_raycast_box<512, 8, false, 8>", seen[1]   # (no need known at the first launch)
    assert seen[2][-1] == "rbpf_raycast_box<512, 6, false, 8>", seen[2]
    # the four-per-CU form keeps FOUR events per end-point slot (8 in the others): a corridor whose near walls are 0.3 m away puts
    # 5-10 beams into one cell, so most slots overflow (more than the 64 the list holds: every slot is scanned) and are replayed
    # exhaustively — same bits
    steps, poses = rc.trajectory(n_scans, inc=(0.01, 0.01, 0.05))
    corridor = (-0.3, 0.35, -1.6, 1.9)
    scans = [orc.room_scan(poses[s], walls=corridor, rng=rng) for s in range(n_scans)]
    pf = _dev(gpu_pkg, N=N, k=k, map_min=-10.0, map_max=10.0)
    pf.setOption(capi.RBPF_OPT_RAYCAST_ADAPT, 3)   # (left alone the selector keeps eight events a slot here: the small box fits four per CU with them)
    hist = []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        assert pf.SLAM(scans[s], u, cur, prev, True, t_icp, orc.normal_stream(70 + s, pf.numNormals(True), 0.0, 1.0)).status == 0
        hist.append(pf.trace()["new_pose"].copy())
    assert pf.lastKernelNames()[1] == "rbpf_raycast_box<512, 8, false, 4>", (pf.lastKernelNames(), pf.raycastBoxCells())
    for m in (0, 11, N - 1):
        assert np.array_equal(pf.logOdds(m), _oracle_map(grid, None, scans, [h[m] for h in hist])), ("corridor", m)
    pf.close()
    # a room whose boxes (13 860 cells) leave the 32-bit cell form at TWO workgroups per CU: the 16-bit cell form (slots by table
    # look-up) takes over with four — unless TBNAV_RBPF_OPT_RAYCAST_CELL16 0 forbids it
    steps, poses = rc.trajectory(n_scans, inc=(0.07, 0.02, 0.01))
    scans = [orc.room_scan(poses[s], walls=rc.ROOM_SURVEY, rng=rng) for s in range(n_scans)]
    for cell16, want in ((1, "rbpf_raycast_box<512, 8, true, "), (0, "rbpf_raycast_box<1024, 8, false, 8>")):
        pf = _dev(gpu_pkg, N=N, k=k, map_min=-10.0, map_max=10.0)
        pf.setOption(capi.RBPF_OPT_RAYCAST_CELL16, cell16)
        hist = []
        for s, (prev, cur, t_icp, u) in enumerate(steps):
            assert pf.SLAM(scans[s], u, cur, prev, True, t_icp, orc.normal_stream(80 + s, pf.numNormals(True), 0.0, 1.0)).status == 0
            hist.append(pf.trace()["new_pose"].copy())
        assert pf.lastKernelNames()[1].startswith(want), (cell16, pf.lastKernelNames(), pf.raycastBoxCells())   # (8 or 4 events a slot: whichever fits)
        for m in (0, 11, N - 1):
            assert np.array_equal(pf.logOdds(m), _oracle_map(grid, None, scans, [h[m] for h in hist])), (cell16, m)
        pf.close()


def test_a_box_that_outgrows_the_four_per_cu_array_is_worked_through_in_bands(gpu_pkg):
    """The four-per-CU form sizes its LDS array by what the boxes needed in the last scans plus three rows or so.  When the next
    scan's boxes are much larger (the robot leaves a small room for a hall), the launch still runs with the small array: every
    particle's box goes through in bands of rows, rays clipped per band — same bits as the oracle's GridMapper — and the array
    follows the need from the next launch on."""
    from rtn_amd import capi
    N, k, n_scans = 24, 6, 7
    steps, poses = rc.trajectory(n_scans, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(21)
    scans = [orc.room_scan(poses[s], walls=rc.ROOM_SMALL if s < 4 else rc.ROOM_SURVEY, rng=rng) for s in range(n_scans)]
    pf = _dev(gpu_pkg, N=N, k=k, map_min=-10.0, map_max=10.0)
    hist, names, cells = [], [], []
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        assert pf.SLAM(scans[s], u, cur, prev, True, t_icp, orc.normal_stream(40 + s, pf.numNormals(True), 0.0, 1.0)).status == 0
        hist.append(pf.trace()["new_pose"].copy())
        names.append(pf.lastKernelNames()[1]); cells.append(pf.raycastBoxCells())
    # scan 4: the hall's boxes (several times the small room's cells) met the array sized for the small room
    assert names[4].startswith("rbpf_raycast_box<512, 8, false"), names
    assert cells[4][1] < 0.6 * max(c[0] for c in cells[5:]), cells
    grid = (0.05, -10.0, 10.0, -10.0, 10.0)
    for m in (0, 11, N - 1):
        assert np.array_equal(pf.logOdds(m), _oracle_map(grid, None, scans, [h[m] for h in hist])), m
    pf.close()


def test_batched_export_equals_single_exports_and_imports_rebuild_the_particles(gpu_pkg):
    """tbnav_rbpf_export_batch_dev writes exactly the blobs tbnav_rbpf_export_particle_dev writes, back to back (a slot listed
    twice included); tbnav_rbpf_import_batch_dev into ANOTHER handle rebuilds pose / weight / map / occupied counts of every
    listed slot (one blob into two slots too), releases what the slots held before, and refuses a slot listed twice."""
    import ctypes as C
    import torch
    N, k = 12, 6
    src = _dev(gpu_pkg, N=N, k=k)
    dst = _dev(gpu_pkg, N=N, k=k)
    steps, poses = rc.trajectory(3)
    rng = np.random.default_rng(3)
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        scan = orc.room_scan(poses[s], walls=rc.ROOM_SMALL, rng=rng)
        for pf, seed in ((src, 70), (dst, 90)):  # two different filters: dst's slots hold maps of their own before the import
            st = pf.SLAM(scan, u, cur, prev, True, t_icp, orc.normal_stream(seed + s, N * (3 * k + 3) + 1, 0.0, 1.0))
            assert st.status == 0
    L = src._L
    slots = np.array([5, 0, 11, 5, 7], dtype=np.int32)
    sizes = np.zeros(len(slots), dtype=np.uint64)
    assert L.tbnav_rbpf_export_batch_sizes(src._h, len(slots), slots.ctypes.data, sizes.ctypes.data) == 0
    buf = torch.zeros(int(sizes.sum()), dtype=torch.uint8, device="cuda")
    offs = np.zeros(len(slots) + 1, dtype=np.uint64)
    assert L.tbnav_rbpf_export_batch_dev(src._h, len(slots), slots.ctypes.data, buf.data_ptr(), buf.numel(), offs.ctypes.data) == 0
    assert int(offs[-1]) == buf.numel() and np.array_equal(np.diff(offs), sizes)
    for i, sl in enumerate(slots):
        n = C.c_uint64()
        assert L.tbnav_rbpf_export_size(src._h, int(sl), C.byref(n)) == 0 and n.value == sizes[i]
        one = torch.zeros(n.value, dtype=torch.uint8, device="cuda")
        assert L.tbnav_rbpf_export_particle_dev(src._h, int(sl), one.data_ptr(), one.numel(), None) == 0
        assert torch.equal(one, buf[int(offs[i]):int(offs[i + 1])]), sl
    assert L.tbnav_rbpf_export_batch_dev(src._h, len(slots), slots.ctypes.data, buf.data_ptr(), buf.numel() - 8, offs.ctypes.data) != 0  # too small
    # import: dst slots 1, 2 <- src 5 (one blob twice), 3 <- src 0, 9 <- src 7
    into = np.array([1, 2, 3, 9], dtype=np.int32)
    frm = np.array([offs[0], offs[0], offs[1], offs[4]], dtype=np.uint64)
    cap0, free0, _ = dst.poolStats()
    assert L.tbnav_rbpf_import_batch_dev(dst._h, len(into), into.ctypes.data, buf.data_ptr(), buf.numel(), frm.ctypes.data) == 0
    sp, sv, sw = src.particles()
    dp, dv, dw = dst.particles()
    for d, s_ in zip(into, (5, 5, 0, 7)):
        assert np.array_equal(dp[d], sp[s_]) and np.array_equal(dv[d], sv[s_]) and dw[d] == sw[s_]
        assert np.array_equal(dst.logOdds(int(d)), src.logOdds(s_))
        assert np.array_equal(dst.distCode(int(d)), src.distCode(s_))   # materialised from the imported occupancy bits / counts
    untouched = dst.logOdds(0)
    dup = np.array([4, 4], dtype=np.int32)
    assert L.tbnav_rbpf_import_batch_dev(dst._h, 2, dup.ctypes.data, buf.data_ptr(), buf.numel(), frm.ctypes.data) != 0
    assert np.array_equal(dst.logOdds(0), untouched)
    # the next scan runs on the imported maps like on any other
    scan = orc.room_scan(poses[2], walls=rc.ROOM_SMALL, rng=rng)
    cur = steps[2][1]
    assert dst.SLAM(scan, (0.0, 0.0, 0.0), cur, cur, True, (0.0, 0.0, 0.0), orc.normal_stream(5, N * (3 * k + 3) + 1, 0.0, 1.0)).status == 0
    src.close(); dst.close()


def test_sixteen_free_lists_run_nearly_dry_then_report_exhaustion(gpu_pkg):
    """A pool of 16 384 tiles or more keeps its free tiles in sixteen lists (csrc/rbpf_device.hpp); a particle pops the tiles one map
    update makes private from ONE list and moves to the next when that one is short.  Here the particles need 96-98 % of the
    pool: the last workgroups find their own list too short and are served by another — every map still the oracle's bits, the
    counters add up — and with a few particles more than the pool holds the scan reports the exhaustion."""
    from rtn_amd import capi
    k, per_tile, cap_tiles = 4, 8192 + 128 + 4 + 4, 16400
    steps, poses = rc.trajectory(2, inc=(0.05, 0.04, 0.03))
    rng = np.random.default_rng(5)
    scans = [orc.room_scan(poses[s], walls=rc.ROOM_SURVEY, rng=rng) for s in range(2)]

    def run(N, n_scans=2):
        pf = _dev(gpu_pkg, pool_bytes=cap_tiles * per_tile, N=N, k=k, map_min=-10.0, map_max=10.0)
        cap, free0, _ = pf.poolStats()
        assert free0 == cap and (cap == cap_tiles - 1 or N < 48)   # (tile 0 is the shared zero tile; a few particles' maps could never need the budget)
        hist, sts = [], []
        for s, (prev, cur, t_icp, u) in enumerate(steps[:n_scans]):
            st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, orc.normal_stream(70 + s, pf.numNormals(True), 0.0, 1.0), check=False)
            sts.append(st.status)
            hist.append(pf.trace()["new_pose"].copy())
        return pf, cap, hist, sts

    pf, cap, _, sts = run(16, 1)                                # pilot: tiles one particle's first scan takes
    assert sts == [0]
    per_particle = (cap - pf.poolStats()[1]) / 16
    pf.close()
    assert 10 < per_particle < 40
    N_fit = int(0.97 * (cap_tiles - 1) / per_particle)
    pf, cap, hist, sts = run(N_fit)
    assert sts == [0, 0]
    used = cap - pf.poolStats()[1]
    assert 0.94 * cap < used <= cap, (used, cap)                 # nearly dry: most lists have fewer tiles left than one particle takes
    grid = (0.05, -10.0, 10.0, -10.0, 10.0)
    for m in (0, N_fit // 2, N_fit - 3, N_fit - 2, N_fit - 1):   # (the last workgroups are the ones that found their list short)
        assert np.array_equal(pf.logOdds(m), _oracle_map(grid, None, scans, [hist[0][m], hist[1][m]])), m
    pf.close()
    pf, cap, _, sts = run(int(1.05 * (cap_tiles - 1) / per_particle), 1)
    assert sts == [capi.ERR_POOL_EXHAUSTED]
    pf.close()


@pytest.mark.parametrize("cap,callers,each,same_hint", [
    (20000, 1000, 14, 0),    # sixteen lists, every caller served by the list of its number
    (20000, 1000, 14, 1),    # everybody starts at list 0 (1250 tiles): it runs dry after 89 callers, the rest move on
    (20000, 1500, 14, 0),    # 21 000 tiles asked of 19 999: some callers find no list long enough
    (20000, 400, 64, 1),     # the longest request a map update makes
    (16384, 2000, 9, 0),     # the smallest pool with sixteen lists, asked for more than it holds
    (5000, 300, 14, 0),      # one list
    (5000, 400, 14, 0),      # one list, exhausted: exactly floor(4999 / 14) callers are served
])
def test_free_lists_hand_every_tile_out_once_and_take_it_back(gpu_pkg, cap, callers, each, same_hint):
    """tile_pop_n / tile_at / tile_push on their own (csrc/rbpf_pool.hip): no tile is handed to two callers, a caller gets all its
    tiles or none, the counts add up, and three rounds of pops and pushes leave the lists whole."""
    import ctypes as C
    from rtn_amd import capi
    lib = capi.lib()
    ids = np.zeros(callers * each, dtype=np.uint32)
    f_pop, f_push = C.c_uint64(0), C.c_uint64(0)
    rc = lib.tbnav_rbpf_pool_selftest(cap, 3, callers, each, same_hint, ids.ctypes.data, C.addressof(f_pop), C.addressof(f_push))
    assert rc == 0
    per = ids.reshape(callers, each)
    served = (per != 0).all(axis=1)
    assert ((per != 0).any(axis=1) == served).all()                 # all of a caller's tiles or none
    got = per[served].ravel()
    assert got.size == np.unique(got).size and (got.size == 0 or (got.min() >= 1 and got.max() < cap))
    assert f_pop.value == cap - 1 - got.size and f_push.value == cap - 1
    n_served, fits = int(served.sum()), callers * each <= cap - 1
    # a request no list can fill at once is popped tile by tile from all of them: whatever fits is served; beyond that the callers
    # that run dry half way hold tiles while they fail (the hook pushes them back before it counts), so the count is a bound
    assert n_served == callers if fits else n_served <= (cap - 1) // each
    if cap < 16384 and not fits and not os.environ.get("TBNAV_POOL_SHARD_MIN"):
        assert n_served == (cap - 1) // each                        # one list: to the tile


def test_free_lists_on_a_small_pool_pop_tile_by_tile(gpu_pkg, monkeypatch):
    """TBNAV_POOL_SHARD_MIN (a test hook) puts a pool of 200 tiles on sixteen lists of 12-13: no list ever holds a request of 14,
    every caller is granted what its list has and takes the rest from the next lists — same guarantees."""
    import ctypes as C
    from rtn_amd import capi
    monkeypatch.setenv("TBNAV_POOL_SHARD_MIN", "32")
    lib = capi.lib()
    for cap, callers, each in ((200, 20, 14), (200, 10, 14), (1000, 40, 30), (1000, 90, 7)):
        ids = np.zeros(callers * each, dtype=np.uint32)
        f_pop, f_push = C.c_uint64(0), C.c_uint64(0)
        assert lib.tbnav_rbpf_pool_selftest(cap, 3, callers, each, 0, ids.ctypes.data, C.addressof(f_pop), C.addressof(f_push)) == 0
        per = ids.reshape(callers, each)
        served = (per != 0).all(axis=1)
        assert ((per != 0).any(axis=1) == served).all()
        got = per[served].ravel()
        assert got.size == np.unique(got).size and (got.size == 0 or (got.min() >= 1 and got.max() < cap))
        n_served = int(served.sum())
        # (asked for more than the pool holds, the callers race for the last tiles one by one and may ALL run dry half way: the scan
        #  that does this fails as a whole anyway; asked for no more than it holds, nobody fails)
        assert n_served == callers if callers * each <= cap - 1 else n_served <= (cap - 1) // each, (cap, callers, each, n_served)
        assert f_pop.value == cap - 1 - got.size and f_push.value == cap - 1


@pytest.mark.parametrize("N", [48, 300])
def test_no_tile_leaks_or_is_freed_twice_over_a_long_run(gpu_pkg, N):
    """The pool's whole life cycle, counted: 36 scans of a moving robot, a resampling with random weights every third scan (tables
    copied, counts adjusted, dead particles' tiles and shed notes pushed back to the free lists, the next scan's clones popped from
    them) — then every particle is made a copy of ONE, and the tiles in use must be exactly the tiles of that particle's map that
    hold a non-zero cell (a tile is taken only when a cell of it is written, and no sum of l_occ's and l_free's is zero).  A tile
    leaked anywhere leaves the count too high, one freed twice (or handed to two owners) too low or the maps wrong.
    N = 300: sixteen free lists (the pool holds 101 401 tiles); N = 48: one."""
    k, n_scans = 4, 36
    pf = _dev(gpu_pkg, N=N, k=k, map_min=-10.0, map_max=10.0)
    cap, free0, _ = pf.poolStats()
    assert free0 == cap
    steps, poses = rc.trajectory(n_scans, inc=(0.06, 0.05, 0.04))
    rng = np.random.default_rng(17)
    scans = [orc.room_scan(poses[s], walls=rc.ROOM_SURVEY, rng=rng) for s in range(n_scans)]
    resamples = 0
    for s, (prev, cur, t_icp, u) in enumerate(steps):
        if s % 3 == 2:
            w = rng.dirichlet(np.full(N, 0.05)) + 1e-9            # a few heavy particles, most of the rest die
            pf.setParticles(w=w / w.sum())
        st = pf.SLAM(scans[s], u, cur, prev, True, t_icp, orc.normal_stream(900 + s, pf.numNormals(True), 0.0, 1.0))
        assert st.status == 0
        resamples += int(st.resampled)
        used = cap - pf.poolStats()[1]
        assert 0 < used <= cap
    assert resamples >= 8
    w = np.full(N, 1e-12); w[N - 1] = 1.0      # (the LAST particle: the selection's walk past the end is clamped to it, particle_filter.cpp:483-491)
    pf.setParticles(w=w / w.sum())
    prev, cur, t_icp, u = steps[-1]
    st = pf.SLAM(scans[-1], u, cur, prev, True, t_icp, orc.normal_stream(999, pf.numNormals(True), 0.0, 1.0))
    assert st.resampled == 1 and set(pf.trace()["resample_idx"].tolist()) == {N - 1}
    lo = pf.logOdds(0)
    assert np.array_equal(lo, pf.logOdds(N - 1))
    G = pf.xsize
    pad = (-G) % 32
    tiles = np.pad(np.asarray(lo).reshape(G, G) != 0.0, ((0, pad), (0, pad))).reshape((G + pad) // 32, 32, (G + pad) // 32, 32).any(axis=(1, 3))
    assert cap - pf.poolStats()[1] == int(tiles.sum())
    pf.close()
