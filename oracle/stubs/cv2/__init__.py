"""Import-time stand-in for OpenCV (absent in this image).

TEST INFRASTRUCTURE ONLY.  The reference (read-only, /root/reference) imports cv2
at module scope for constant tables (codes/options/options.py:4,11-36,
codes/dataops/common.py:7, codes/utils/util.py:5).  No cv2 *function* is on the
SR training hot path, so upper-case names resolve to distinct ints and anything
else raises.
"""
_consts = {}
__version__ = "4.5.5"     # extra_functional.py:23 compares the version at import time


def __getattr__(name):
    if name.isupper() or (name[:1].isupper() and "_" in name):
        if name not in _consts:
            _consts[name] = 1000 + len(_consts)
        return _consts[name]
    raise AttributeError("cv2 stub: %s is not available" % name)
