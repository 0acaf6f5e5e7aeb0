"""HDF5 side of the .slp reader / writer, runnable under any interpreter that has h5py (this image: /opt/conda/bin/python3.9).

    python sleap_amd/io/_slp_io.py read  <file.slp> <out.npz>     tables + metadata -> npz
    python sleap_amd/io/_slp_io.py write <in.npz>   <file.slp>     npz (as produced by sleap_amd.io.slp) -> .slp
    python sleap_amd/io/_slp_io.py frames <file.h5> <dataset> <out.npy>   frames dataset of an HDF5 video -> memory-mappable .npy

Part of the package (round 5; it was tools/slp_io.py) so that an installed sleap_amd needs no repository checkout; self-contained
on purpose (numpy + h5py only, no package-relative imports): `sleap_amd.io.slp` runs this FILE under the interpreter named by
SLEAP_AMD_H5_PYTHON when the running one has no h5py, and that interpreter need not have torch or sleap_amd (`python -m
sleap_amd.io._slp_io ...` works where it has).

Dataset layout as written by the reference (sleap/io/format/hdf5.py:265-575): group `metadata` with attrs `format_id`
(float) and `json` (bytes); string datasets `videos_json`, `tracks_json`, `suggestions_json` (one JSON document per
element, float64 when empty); structured datasets `frames`, `instances`, `points`, `pred_points`, all 1-D with
maxshape (None,).
"""
import sys

import numpy as np


def read(path, out):
    import h5py

    d = {}
    with h5py.File(path, "r") as f:
        for k in ("frames", "instances", "points", "pred_points"):
            d[k] = f[k][:]
        for k in ("videos_json", "tracks_json", "suggestions_json"):
            arr = f[k][:] if k in f else np.zeros((0,))
            d[k] = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in arr], dtype=object) if len(arr) else \
                np.zeros((0,), dtype=object)
        m = f["metadata"].attrs
        d["format_id"] = np.float64(m["format_id"])
        j = m["json"]
        d["json"] = np.array(j.decode() if isinstance(j, bytes) else (j.tobytes().decode() if hasattr(j, "tobytes") else str(j)),
                             dtype=object)
    np.savez(out, **{k: (v if v.dtype != object else np.array(v.tolist(), dtype=str)) for k, v in d.items()})


def write(npz, path):
    import os

    import h5py

    z = np.load(npz, allow_pickle=False)
    if os.path.exists(path):
        os.unlink(path)  # hdf5.py:282-283
    with h5py.File(path, "a") as f:
        g = f.require_group("metadata")
        g.attrs["format_id"] = float(z["format_id"])
        for key in ("videos_json", "tracks_json", "suggestions_json"):
            data = [np.bytes_(s) for s in z[key].tolist()]
            f.create_dataset(key, data=data, maxshape=(None,))
        g.attrs["json"] = np.bytes_(str(z["json"]))
        for key in ("points", "pred_points", "instances", "frames"):
            f.create_dataset(key, data=z[key], maxshape=(None,), dtype=z[key].dtype)


def frames(path, dataset, out):
    """Copy a frames dataset of an HDF5 video to a .npy file (memory-mappable) without loading it whole."""
    import h5py

    with h5py.File(path, "r") as f:
        d = f[dataset]
        mm = np.lib.format.open_memmap(out, mode="w+", dtype=d.dtype, shape=d.shape)
        step = max(1, (64 << 20) // max(1, int(np.prod(d.shape[1:])) * d.dtype.itemsize))
        for i in range(0, d.shape[0], step):
            mm[i:i + step] = d[i:i + step]
        mm.flush()


def main(argv):
    if argv[0] == "frames":
        frames(argv[1], argv[2], argv[3])
    else:
        {"read": read, "write": write}[argv[0]](argv[1], argv[2])


if __name__ == "__main__":
    main(sys.argv[1:])
