"""Extract a Keras `best_model.h5` into a self-contained `.npz` (graph JSON + weights).

Runs under any interpreter that has h5py (in this image: /opt/conda/bin/python3.9).
The product loader (`sleap_amd.nn.model_io.load_keras_model`) imports `extract` when the running
interpreter has h5py and otherwise runs THIS FILE as a subprocess under the interpreter named by
SLEAP_AMD_H5_PYTHON. Part of the package (round 5; it was tools/h5_extract.py) so that an installed
sleap_amd reads a `best_model.h5` without the repository checkout; self-contained on purpose (numpy +
h5py, no package-relative imports): the other interpreter need not have torch or sleap_amd.
`python -m sleap_amd.nn._h5_extract best_model.h5 out.npz` works where it has.

Schema read (reference: sleap/nn/inference.py:3204-3209 which calls
`tf.keras.models.load_model(best_model.h5)`):
  root attr `model_config`   JSON functional graph
  group `model_weights/<layer>/<layer>/{kernel:0,bias:0,gamma:0,beta:0,moving_mean:0,moving_variance:0}`

Output npz keys:
  `__model_config__`  uint8 array holding the UTF-8 JSON of `model_config`
  `<layer>/<weight>`  float32 arrays (Keras layouts untouched)

usage: python sleap_amd/nn/_h5_extract.py best_model.h5 out.npz
"""
import sys
import json
import numpy as np


def extract(h5_path, out_path):
    import h5py

    out = {}
    with h5py.File(h5_path, "r") as f:
        cfg = f.attrs["model_config"]
        if isinstance(cfg, bytes):
            cfg = cfg.decode("utf-8")
        json.loads(cfg)  # validate
        out["__model_config__"] = np.frombuffer(cfg.encode("utf-8"), dtype=np.uint8)
        mw = f["model_weights"]

        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                parts = name.split("/")
                layer = parts[0]
                wname = parts[-1].split(":")[0]
                out[f"{layer}/{wname}"] = np.asarray(obj[()], dtype=np.float32)

        mw.visititems(visit)
    np.savez_compressed(out_path, **out)


if __name__ == "__main__":
    extract(sys.argv[1], sys.argv[2])
