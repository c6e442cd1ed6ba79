"""Soak of the sharded MPPI tick with one process per rank on one device: python tools/soak_procs.py [ranks] [ticks]
n ticks through the direct exchange and through the communicator's all-gather; the final warm-start controls must agree bit for bit
between the two and across the ranks."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dist_workers import mppi_soak_worker, run_spawn  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    a = run_spawn(mppi_soak_worker, world, 1024, 0.5, ticks, True)
    b = run_spawn(mppi_soak_worker, world, 1024, 0.5, ticks, False)
    assert all(a[r]["kind"] == 2 and b[r]["kind"] == 1 for r in range(world))
    ok = all(np.array_equal(a[0]["u"], a[r]["u"]) and np.array_equal(a[r]["u"], b[r]["u"]) for r in range(world))
    print(f"soak: {world} ranks x {ticks} ticks: direct == all-gather == every rank: {ok}", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
