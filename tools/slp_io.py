"""Shim: the HDF5 side of the .slp reader / writer lives in the package now (`sleap_amd/io/_slp_io.py`). Kept for the fixture
scripts that `import slp_io` and for `python tools/slp_io.py read|write|frames ...`. Loaded by FILE PATH (interpreters with
h5py but without torch / sleap_amd)."""
import importlib.util
import os
import sys

_p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sleap_amd", "io", "_slp_io.py")
_s = importlib.util.spec_from_file_location("_sleap_amd_slp_io", _p)
_m = importlib.util.module_from_spec(_s)
_s.loader.exec_module(_m)
read, write, frames = _m.read, _m.write, getattr(_m, "frames", None)

if __name__ == "__main__":
    _m.main(sys.argv[1:]) if hasattr(_m, "main") else None
