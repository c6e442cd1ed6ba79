"""Randomized sweep of the in-library sharded paths with one PROCESS per rank on this box's one device (IPC transport; the MPPI
records through the direct exchange AND through the communicator's all-gather): python tools/fuzz_procs.py [n_mppi] [n_rbpf] [seed]
Every case runs tests/dist_workers.py's workers, which assert against the oracle / the unsharded handles inside the ranks; here the
ranks' results are compared with each other and between the two exchanges, bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dist_workers import mppi_comm_worker, rbpf_comm_worker, run_spawn  # noqa: E402


def main():
    n_mppi = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n_rbpf = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    fails = 0
    for i in range(n_mppi):
        world = int(rng.integers(2, 5))
        K_local = int(rng.choice([64, 200, 512, 1024, 2048, 3000, 8192, 20000, 33000]))
        horizon = float(rng.choice([0.12, 0.25, 0.5, 0.65, 1.0])) if K_local <= 8192 else float(rng.choice([0.12, 0.24, 0.32]))
        try:
            a = run_spawn(mppi_comm_worker, world, K_local, horizon, 2)
            b = run_spawn(mppi_comm_worker, world, K_local, horizon, 2, False)
            assert all(a[r]["kind"] == 2 for r in range(world)) and all(b[r]["kind"] == 1 for r in range(world))
            for r in range(world):
                for x, y in zip(a[0]["rng"] + [a[0]["batch_u"]], a[r]["rng"] + [a[r]["batch_u"]]):
                    assert np.array_equal(x, y)
                for x, y in zip(a[r]["rng"] + [a[r]["batch_u"]], b[r]["rng"] + [b[r]["batch_u"]]):
                    assert np.array_equal(x, y)
                for (g0, u0), (g1, u1) in zip(a[r]["host"], b[r]["host"]):
                    assert np.array_equal(g0, g1) and np.array_equal(u0, u1)
        except AssertionError as e:
            fails += 1
            print(f"MPPI case {i} FAILED: world={world} K_local={K_local} horizon={horizon}: {str(e)[:400]}", flush=True)
    print(f"mppi: {n_mppi} cases, {fails} failures", flush=True)
    fr = 0
    for i in range(n_rbpf):
        world = int(rng.integers(2, 5))
        n_local = int(rng.integers(3, 12))
        N = world * n_local
        heavy = {int(j): float(w) for j, w in zip(rng.choice(N, size=int(rng.integers(1, 4)), replace=False), rng.uniform(0.2, 0.9, 3))}
        dev = bool(rng.integers(0, 2))
        try:
            out = run_spawn(rbpf_comm_worker, world, n_local, 8, heavy, dev)
            assert all(out[r]["stats"] == out[0]["stats"] for r in range(world))
        except AssertionError as e:
            fr += 1
            print(f"RBPF case {i} FAILED: world={world} n_local={n_local} heavy={heavy} device_noise={dev}: {str(e)[:400]}", flush=True)
    print(f"rbpf: {n_rbpf} cases, {fr} failures", flush=True)
    sys.exit(1 if fails or fr else 0)


if __name__ == "__main__":   # (the workers are spawned: they import this module)
    main()
