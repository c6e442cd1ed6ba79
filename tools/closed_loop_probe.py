"""Dev probe: MPPI closed loop (class surface) for a few (K, horizon) settings."""
import sys, os, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as g
pkg = g.load_package(); pkg.capi.lib()
from cases import MPPI_BASE, WAYPOINTS
L = C.CDLL(os.path.join(ROOT, 'ros-turtlebot-navigation_amd', 'lib', 'libtbnav_host.so'))
L.hst_last_error.restype = C.c_char_p
def P(a): return a.ctypes.data_as(C.c_void_p)
def params(d): return np.array([d["wheel_radius"], d["wheel_base"], d["lam"], d["max_wheel_vel"], d["ul_var"], d["ur_var"], d["horizon"], d["dt"]] + d["Q"] + d["R"] + d["P1"])
for K, hor in ((64, 0.25), (64, 1.0), (1024, 0.5), (1024, 1.0)):
    d = dict(MPPI_BASE, rollouts=K, horizon=hor)
    wp = np.array(WAYPOINTS, dtype=np.float64)
    mt = 40000
    traj = np.zeros((mt, 5)); reached = C.c_int()
    ticks = L.hst_mppi_closed_loop(P(params(d)), K, C.c_uint64(3), P(wp), 5, C.c_double(0.05), C.c_double(60.0), mt, P(traj), C.byref(reached))
    print(K, hor, "ticks", ticks, "reached", reached.value, "last", traj[max(ticks-1,0)], L.hst_last_error())
    if ticks>0:
        sel = traj[:ticks:max(1,ticks//12)]
        print(np.round(sel[:, :3], 3).tolist())
