"""Shim: the extractor lives in the package now (`sleap_amd/nn/_h5_extract.py`, so an installed sleap_amd reads a
`best_model.h5` without this repository's tools/). Kept for the fixture scripts that `import h5_extract` and for
`python tools/h5_extract.py best_model.h5 out.npz`. Loaded by FILE PATH: it must run under interpreters that have h5py but
neither torch nor sleap_amd."""
import importlib.util
import os
import sys

_p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sleap_amd", "nn", "_h5_extract.py")
_s = importlib.util.spec_from_file_location("_sleap_amd_h5_extract", _p)
_m = importlib.util.module_from_spec(_s)
_s.loader.exec_module(_m)
extract = _m.extract

if __name__ == "__main__":
    extract(sys.argv[1], sys.argv[2])
