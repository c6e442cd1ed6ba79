"""Dev probe: device sincos error on selected arguments."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); L = pkg.capi.lib()
x = np.array([1e5, 1.0e5 + 1.0, 2e5, 1e6, 3e7, -2.5e12, 1e15, 1e18, 1e22, 1e300])
s = np.empty_like(x); c = np.empty_like(x)
pkg.capi.check(L.tbnav_mppi_debug_sincos(x.ctypes.data, x.size, s.ctypes.data, c.ctypes.data), "dbg")
for xi, si, ci in zip(x, s, c):
    print(f"{xi:10.3e} sin err {abs(si-np.sin(xi)):.2e} cos err {abs(ci-np.cos(xi)):.2e}")
