"""MI355X-native hot paths of bostoncleek/ROS-Turtlebot-Navigation behind a C-ABI.

The directory name carries the reference's name (with hyphens), so it is loaded through
`__graft_entry__.load_package()` under the module name `rtn_amd` rather than a plain import.
Contents: csrc/ (HIP kernels + C-ABI), lib/ (built libtbnav_hip.so), host/ (C++ class shims with the
reference's surfaces), capi.py / mppi.py / rbpf.py (ctypes plumbing used by tests and bench).
"""
from . import capi  # noqa: F401
