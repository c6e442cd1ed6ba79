"""(no GPU needed) The HBM traffic the map update CANNOT go below with this map layout, for the bench workload (bench_rbpf.layout_floor:
whole 128-byte lines read, whole 32-byte sectors written — profiles/r05_fetch_write_calibration.txt).
python tools/raycast_traffic_floor.py [bench|survey] [scans]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rbpf_cases as rc
import bench_rbpf_detail as bench_rbpf
room = sys.argv[1] if len(sys.argv) > 1 else "bench"
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 14
walls, inc = (rc.ROOM_BENCH, rc.TRAJ_BENCH) if room == "bench" else (rc.ROOM_SURVEY, rc.TRAJ_SURVEY)
f = bench_rbpf.layout_floor(walls, inc, n_scans)
alg, fl = f["algorithmic_bytes"], f["read_floor_bytes"] + f["write_floor_bytes"]
print(f"{room} room, scans 2..{n_scans - 1}: {f['valid_beams']:.0f} valid beams, {f['distinct_cells']:.0f} distinct cells, {f['sectors_32B']:.0f} 32-byte sectors, "
      f"{f['lines_128B']:.0f} 128-byte lines per particle and scan")
print(f"  algorithmic (distinct cells x 16 B): {alg / 1e3:.1f} KB per particle")
print(f"  layout floor: reads {f['read_floor_bytes'] / 1e3:.1f} KB (whole lines) + writes {f['write_floor_bytes'] / 1e3:.1f} KB (whole sectors) = {fl / 1e3:.1f} KB per particle = {fl / alg:.3f} x algorithmic")
