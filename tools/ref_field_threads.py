import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch, bench_rbpf
bench.graft.load_package()
print("affinity", len(os.sched_getaffinity(0)), "effective", bench_rbpf.effective_cores())
for th in (8, 16, 32, 64, 128, 200):
    r = bench_rbpf.reference_field_mode(torch.device("cuda", 0), 1000, 50, 10.0, bench_rbpf.ROOM_BENCH, 4, host_threads=th)
    print(th, r["particle_updates_per_s"], r["ms_per_scan"], flush=True)
