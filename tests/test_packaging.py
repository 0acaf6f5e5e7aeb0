"""An INSTALLED sleap_amd (the package directory alone, no repository checkout beside it) must read a SLEAP model folder:
`best_model.h5` through the package's own `_h5_extract.py`, `.slp` files through `_slp_io.py` (both lived
under tools/ and the loader reached outside the package). CPU only; needs an interpreter with h5py (SLEAP_AMD_H5_PYTHON)."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H5PY = os.environ.get("SLEAP_AMD_H5_PYTHON", "/opt/conda/bin/python3.9")

_MAKE_H5 = r"""
import json, sys, numpy as np, h5py
z = np.load(sys.argv[1])
with h5py.File(sys.argv[2], "w") as f:
    f.attrs["model_config"] = bytes(z["__model_config__"]).decode("utf-8")   # tf.keras.Model.save(..., save_format="h5")
    mw = f.create_group("model_weights")
    for k in z.files:
        if k == "__model_config__":
            continue
        layer, w = k.split("/")
        mw.require_group(layer).require_group(layer).create_dataset(w + ":0", data=z[k])
"""


def _package_copy(tmp_path):
    """the package tree as `pip install` would lay it down: sleap_amd/ only -- no tools/, tests/, oracle/ beside it"""
    site = tmp_path / "site"
    shutil.copytree(os.path.join(ROOT, "sleap_amd"), site / "sleap_amd",
                    ignore=shutil.ignore_patterns("__pycache__", "*.so", "*.o", "csrc", "lib", "data"))
    assert not (site / "tools").exists()
    return site


@pytest.mark.skipif(not os.path.exists(H5PY), reason="no interpreter with h5py (SLEAP_AMD_H5_PYTHON)")
def test_installed_package_reads_best_model_h5_without_the_repo(tmp_path):
    src = os.path.join(ROOT, "tests", "golden", "models", "minimal_instance.UNet.bottomup")
    folder = tmp_path / "model"
    folder.mkdir()
    shutil.copy(os.path.join(src, "training_config.json"), folder / "training_config.json")
    subprocess.run([H5PY, "-c", _MAKE_H5, os.path.join(src, "best_model.npz"), str(folder / "best_model.h5")], check=True)
    site = _package_copy(tmp_path)
    code = ("import sys, json, numpy as np; sys.path.insert(0, sys.argv[1]);"
            "from sleap_amd.nn import model_io; import sleap_amd;"
            "assert sleap_amd.__file__.startswith(sys.argv[1]), sleap_amd.__file__;"
            "cfg, w = model_io.load_keras_model(sys.argv[2]);"
            "np.savez(sys.argv[3], **w); print(json.dumps(cfg)[:40])")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["SLEAP_AMD_H5_PYTHON"] = H5PY
    r = subprocess.run([sys.executable, "-c", code, str(site), str(folder), str(tmp_path / "w.npz")], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (folder / "best_model.npz").exists()  # the extraction is cached beside the .h5
    want, got = np.load(os.path.join(src, "best_model.npz")), np.load(tmp_path / "w.npz")
    keys = sorted(k for k in want.files if k != "__model_config__")
    assert sorted(got.files) == keys and len(keys) > 10
    for k in keys:
        np.testing.assert_array_equal(got[k], want[k])
    # the graph came through too
    cfg = json.loads(bytes(np.load(folder / "best_model.npz")["__model_config__"]).decode())
    assert cfg["class_name"] in ("Functional", "Model")


@pytest.mark.skipif(not os.path.exists(H5PY), reason="no interpreter with h5py (SLEAP_AMD_H5_PYTHON)")
def test_installed_package_round_trips_an_slp_file_without_the_repo(tmp_path):
    site = _package_copy(tmp_path)
    code = ("import sys, numpy as np; sys.path.insert(0, sys.argv[1]);"
            "from sleap_amd.io import slp;"
            "outs = [dict(instance_peaks=np.array([[[[1., 2.], [3., 4.]]]], np.float32), instance_peak_vals=np.ones((1, 1, 2), np.float32),"
            "             instance_scores=np.ones((1, 1), np.float32), frame_ind=np.array([0]))];"
            "slp.write_slp(sys.argv[2], outs, ['a', 'b'], [[0, 1]]);"
            "t = slp.read_slp(sys.argv[2]);"
            "assert len(t['frames']) == 1 and len(t['pred_points']) == 2, t; print('ok')")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["SLEAP_AMD_H5_PYTHON"] = H5PY
    r = subprocess.run([sys.executable, "-c", code, str(site), str(tmp_path / "p.slp")], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_export_model_attribute_path_exists_and_says_out_of_scope():
    """`Predictor.export_model` / `InferenceModel.export_model` exist with the reference's signatures
    (inference.py:533-542, 1092-1100) and raise a clear NotImplementedError."""
    import inspect

    from sleap_amd.nn.inference import InferenceModel, Predictor

    assert list(inspect.signature(Predictor.export_model).parameters) == [
        "self", "save_path", "signatures", "save_traces", "model_name", "tensors", "unrag_outputs", "max_instances"]
    assert list(inspect.signature(InferenceModel.export_model).parameters) == [
        "self", "save_path", "signatures", "save_traces", "model_name", "tensors", "unrag_outputs"]
    with pytest.raises(NotImplementedError, match="out of scope"):
        Predictor.export_model(object.__new__(Predictor), "/tmp/x")
    with pytest.raises(NotImplementedError, match="out of scope"):
        InferenceModel.export_model(object.__new__(InferenceModel), "/tmp/x")
