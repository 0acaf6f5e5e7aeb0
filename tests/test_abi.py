"""The C-ABI library loads and exports every symbol include/sleap_amd.h declares (no GPU needed)."""
import os
import re

from sleap_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sleap_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sa_[a-z0-9_]+)\s*\(", src)))


import pytest


@pytest.mark.parametrize("dtype", _lib.DTYPES)
def test_all_declared_symbols_exported_and_bound(dtype):
    """Both builds of the library (bf16 and fp16 storage, csrc/bf16.h) export the same ABI."""
    names = _declared()
    assert len(names) >= 15
    h = _lib.lib(dtype)
    for n in names:
        assert hasattr(h, n), f"{n} declared in sleap_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)
    assert h.sa_abi_version() == 8
    assert h.sa_storage_dtype().decode() == dtype


def test_argument_validation_without_gpu():
    h = _lib.lib()
    # bad shapes are rejected before any HIP call
    rc = h.sa_find_local_peaks(None, None, 0, 4, 4, 1, 0.2, 0, 5, 1.0, 16, None, None, None, None, None, None, 0, None)
    assert rc == -1
    assert b"bad shape" in h.sa_last_error()
    assert h.sa_find_local_peaks_workspace(4, 100) == 1600


def test_network_plan_validation_without_gpu():
    """sa_network_create parses the plan on the host: malformed word streams are rejected, a minimal well-formed one yields a
    handle whose shape queries work without a device."""
    import ctypes as C

    import numpy as np

    h = _lib.lib()
    out = C.c_void_p()
    bad = np.array([1, 2, 3, 4, 5, 6, 7, 8], np.int64)
    assert h.sa_network_create(bad.ctypes.data_as(C.c_void_p), bad.size, C.byref(out)) == -1
    assert b"bad magic" in h.sa_last_error()
    MAGIC = 0x53414E4554303032
    # header (magic, buffers, outputs, ops, input channels, max stride, layout); buffers: 0 = 16-bit 16ch @ 1/1, 1 = f32 head
    # 3ch @ 1/2 (virtual ids are shape-only); one head op reading 0 -> 1
    words = [MAGIC, 2, 1, 1, 1, 2, 0, 16, 1, 1, 0, 3, 1, 2, 1, 1, 3, 1, 11, 6, 0, 0, 0, 3, 0, 1]
    ok = np.array(words, np.int64)
    assert h.sa_network_create(ok.ctypes.data_as(C.c_void_p), ok.size, C.byref(out)) == 0, h.sa_last_error()
    assert h.sa_network_n_outputs(out) == 1 and h.sa_network_in_channels(out) == 1 and h.sa_network_max_stride(out) == 2
    assert h.sa_network_layout(out) == _lib.LAYOUT_NHWC
    oh, ow, oc = C.c_int(), C.c_int(), C.c_int()
    assert h.sa_network_output_shape(out, 0, 8, 12, C.byref(oh), C.byref(ow), C.byref(oc)) == 0
    assert (oh.value, ow.value, oc.value) == (4, 6, 3)
    assert h.sa_network_workspace_bytes(out, 2, 8, 12) == 2 * 8 * 12 * 16 * 2 + 768  # 16-bit tensor + 256-aligned f32 head (576 B)
    # wrong channel count / odd size / small workspace are refused before any launch
    one = (C.c_void_p * 1)(1)
    assert h.sa_network_forward(out, 1, 1, 2, 8, 12, 3, one, 1, 1 << 20, None) == -1 and b"input channels" in h.sa_last_error()
    assert h.sa_network_forward(out, 1, 1, 2, 7, 12, 1, one, 1, 1 << 20, None) == -1 and b"multiple of the model stride" in h.sa_last_error()
    assert h.sa_network_forward(out, 1, 1, 2, 8, 12, 1, one, 1, 16, None) == -4
    h.sa_network_destroy(out)
    planes = list(words)
    planes[6] = _lib.LAYOUT_PLANES16  # round 3: the un-fused matrix-core head reads 16-channel planes (<= 64 maps)
    planes = np.array(planes, np.int64)
    assert h.sa_network_create(planes.ctypes.data_as(C.c_void_p), planes.size, C.byref(out)) == 0, h.sa_last_error()
    assert h.sa_network_layout(out) == _lib.LAYOUT_PLANES16
    h.sa_network_destroy(out)
    pool = list(words[:18]) + [12, 2, 0, 0]  # round 4: a stand-alone MaxPool2D launch runs per plane (a plane of a frame = a frame)
    pool[6] = _lib.LAYOUT_PLANES16
    pool = np.array(pool, np.int64)
    assert h.sa_network_create(pool.ctypes.data_as(C.c_void_p), pool.size, C.byref(out)) == 0, h.sa_last_error()
    h.sa_network_destroy(out)
    pool = list(words[:18]) + [7, 5, 0, 0, 0, 0, 0]  # the fixture-only VALU transposed conv only exists for NHWC tensors
    pool[6] = _lib.LAYOUT_PLANES16
    pool = np.array(pool, np.int64)
    assert h.sa_network_create(pool.ctypes.data_as(C.c_void_p), pool.size, C.byref(out)) == -3 and b"PLANES16" in h.sa_last_error()
    trunc = np.array(words[:-2], np.int64)
    assert h.sa_network_create(trunc.ctypes.data_as(C.c_void_p), trunc.size, C.byref(out)) == -1
    assert b"truncated" in h.sa_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["minimal_instance.UNet.bottomup", "min_tracks_2node.UNet.bottomup_multiclass",
                                   "minimal_robot.UNet.single_instance"])
def test_c_executor_equals_python_launch_loop(model):
    """DeviceNetwork.forward runs the plan inside the library (sa_network_forward); the per-launch Python loop kept for
    profiling issues the same launches: outputs bitwise equal."""
    import numpy as np
    import torch

    from sleap_amd.nn import model_io
    from sleap_amd.nn.engine import DeviceNetwork

    d = os.path.join(ROOT, "tests", "golden", "models", model)
    net = DeviceNetwork(*model_io.load_keras_model(d))
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.integers(0, 256, (2, 96, 128, net.in_channels), dtype=np.uint8)).cuda()
    a = [t.clone() for t in net.forward(x)]
    prof = []
    b = [t.clone() for t in net.forward(x, profile=prof)]
    assert len(prof) == len(net.plan)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    xf = x.float() / 255
    for u, v in zip(net.forward(xf), [t.clone() for t in net.forward(xf, profile=[])]):
        assert torch.equal(u, v)


@pytest.mark.gpu
def test_bottomup_fixture_through_ctypes_only():
    """SURVEY 8(b) last row: the hot path from C -- sa_network_create + sa_bottomup_predict with caller-owned buffers and a
    caller-owned workspace, no sleap_amd.nn objects in the call -- gives what BottomUpInferenceModel.call gives."""
    import ctypes as C

    import numpy as np
    import torch

    from sleap_amd.nn.inference import load_model
    from sleap_amd.synth import render_frames

    p = load_model(os.path.join(ROOT, "tests", "golden", "models", "minimal_instance.UNet.bottomup"), batch_size=3)
    layer = p.inference_model.bottomup_layer
    layer.peak_threshold = 0.5
    frames = render_frames(3, 384, 384, n_animals=2, seed=11)[0]
    want = p.inference_model.call(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    net, sc = layer.keras_model, layer.paf_scorer
    h = _lib.lib(net.dtype)

    class Params(C.Structure):
        _fields_ = [("confmaps_ind", C.c_int), ("pafs_ind", C.c_int), ("offsets_ind", C.c_int), ("peak_threshold", C.c_float),
                    ("refinement", C.c_int), ("integral_patch_size", C.c_int), ("cm_output_stride", C.c_float),
                    ("pafs_stride", C.c_float), ("n_nodes", C.c_int), ("n_edges", C.c_int), ("edges", C.c_void_p),
                    ("sorted_edge_inds", C.c_void_p), ("n_sorted", C.c_int), ("max_edge_length_ratio", C.c_float),
                    ("dist_penalty_weight", C.c_float), ("n_points", C.c_int), ("min_line_scores", C.c_float),
                    ("min_instance_peaks", C.c_int), ("max_peaks", C.c_int), ("max_node_peaks", C.c_int), ("max_instances", C.c_int)]

    words = net.plan_words()
    handle = C.c_void_p()
    assert h.sa_network_create(words.ctypes.data_as(C.c_void_p), words.size, C.byref(handle)) == 0
    edges = torch.tensor(sc.edge_inds, dtype=torch.int32).cuda()
    sorted_e = torch.tensor(list(sc.sorted_edge_inds), dtype=torch.int32).cuda()
    I = sc.max_instances
    q = Params(layer.confmaps_ind, layer.pafs_ind, -1 if layer.offsets_ind is None else layer.offsets_ind, layer.peak_threshold,
               _lib.REFINE[layer.refinement], layer.integral_patch_size, float(layer.cm_output_stride), float(sc.pafs_stride),
               sc.n_nodes, sc.n_edges, edges.data_ptr(), sorted_e.data_ptr(), len(sc.sorted_edge_inds), sc.max_edge_length_ratio,
               sc.dist_penalty_weight, sc.n_points, sc.min_line_scores, 0, layer.max_peaks, sc.max_node_peaks, I)
    B, H, W, Cc = frames.shape
    n = h.sa_bottomup_workspace_bytes(handle, C.byref(q), B, H, W)
    ws = torch.zeros((n,), dtype=torch.uint8, device="cuda")
    x = torch.from_numpy(frames).cuda()
    ip = torch.empty((B, I, sc.n_nodes, 2), dtype=torch.float32, device="cuda")
    iv = torch.empty((B, I, sc.n_nodes), dtype=torch.float32, device="cuda")
    isc = torch.empty((B, I), dtype=torch.float32, device="cuda")
    ni = torch.empty((B,), dtype=torch.int32, device="cuda")
    stt = torch.empty((B,), dtype=torch.int32, device="cuda")
    rc = h.sa_bottomup_predict(handle, C.byref(q), x.data_ptr(), 1, B, H, W, Cc, ip.data_ptr(), iv.data_ptr(), isc.data_ptr(),
                               ni.data_ptr(), stt.data_ptr(), ws.data_ptr(), n, None)
    assert rc == 0, h.sa_last_error()
    torch.cuda.synchronize()
    h.sa_network_destroy(handle)
    nv = want["n_valid"].cpu().numpy() if "n_valid" in want else None
    got_n = ni.cpu().numpy()
    assert got_n.sum() > 0
    wp = want["instance_peaks"].cpu().numpy()
    for b in range(B):
        k = int(got_n[b])
        if nv is not None:
            assert k == int(nv[b])
        np.testing.assert_array_equal(ip.cpu().numpy()[b, :k], wp[b, :k])
        np.testing.assert_array_equal(isc.cpu().numpy()[b, :k], want["instance_scores"].cpu().numpy()[b, :k])
    np.testing.assert_array_equal(stt.cpu().numpy(), want["status"].cpu().numpy())  # the same capacity / OOB flags, frame by frame
