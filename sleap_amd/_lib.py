"""ctypes binding of libsleap_amd.so (the C ABI declared in include/sleap_amd.h).

There is NO CPU fallback: if the shared object can not be loaded the import raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libsleap_amd.so")

SA_OK = 0
STATUS_PEAK_OVERFLOW = 1
STATUS_NODE_PEAK_OVERFLOW = 2
STATUS_INSTANCE_OVERFLOW = 4
STATUS_LSA_INFEASIBLE = 8
STATUS_PAF_OOB = 16
STATUS_NONFINITE = 32

REFINE = {None: 0, "none": 0, "integral": 1, "local": 2, "offsets": 3}

SRC1_NONE, SRC1_DIRECT, SRC1_UPSAMPLE2X, SRC0_POOL2X = 0, 1, 2, 4
LAYOUT_NHWC, LAYOUT_PLANES16 = 0, 0x100  # SA_LAYOUT_*: 16-bit activation tensors as [B,H,W,CP] or as 16-channel planes [B,CP/16,H,W,16]

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/sleap_amd.h
SIGNATURES = {
    "sa_abi_version": (_i, []),
    "sa_storage_dtype": (C.c_char_p, []),
    "sa_last_error": (C.c_char_p, []),
    "sa_device_info": (_i, [_i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "sa_find_local_peaks_workspace": (_sz, [_i, _i]),
    "sa_find_local_peaks_rough": (_i, [_p, _i, _i, _i, _i, _f, _i, _p, _p, _p, _p]),
    "sa_find_local_peaks": (_i, [_p, _p, _i, _i, _i, _i, _f, _i, _i, _f, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "sa_find_global_peaks": (_i, [_p, _p, _i, _i, _i, _i, _f, _i, _i, _f, _p, _p, _p]),
    "sa_select_centroids": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p]),
    "sa_finish_instance_peaks": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "sa_crop_and_resize": (_i, [_p, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p]),
    "sa_paf_score": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _i, _f, _f, _f, _i, _p, _p, _p, _p, _p]),
    "sa_paf_match": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "sa_paf_workspace": (_sz, [_i, _i, _i, _i]),
    "sa_paf_group_connections": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _f, _i, _i, _p, _p, _p,
                                      _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "sa_bottomup_postproc_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "sa_bottomup_postproc": (_i, [_p, _p, _i, _i, _i, _i, _f, _i, _i, _f, _i, _p, _i, _i, _i, _p, _p, _i, _i, _i, _f, _f, _f, _i,
                                  _f, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "sa_paf_line_subs": (_i, [_p, _p, _p, _i, _i, _f, _p, _p]),
    "sa_gather_nd3": (_i, [_p, _i, _i, _i, _p, _i, _p, _p, _p]),
    "sa_paf_line_scores": (_i, [_p, _p, _p, _i, _i, _f, _f, _p, _p]),
    "sa_distance_penalty": (_i, [_p, _i, _f, _f, _p, _p]),
    "sa_paf_group": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "sa_lsa_host": (_i, [_p, _i, _i, _p, _p]),
    "sa_lsa_host_wave": (_i, [_p, _i, _i, _p, _p]),
    "sa_stem_conv3x3": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p]),
    "sa_conv3x3_bf16": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "sa_stem_conv3x3x2_bf16": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p, _i, _i, _p, _p, _i, _p]),
    "sa_stem16_u8_bf16": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p]),
    "sa_stem16_pack": (_i, [_p, _p, _i, _i, _p, _p, _i, _p]),
    "sa_stem16_blob_bytes": (_sz, []),
    "sa_conv3x3_ex_bf16": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _p]),
    "sa_pointwise_packed_elems": (_sz, [_i, _i]),
    "sa_pack_pointwise_weights": (_i, [_p, _i, _i, _i, _i, _p]),
    "sa_conv3x3_bneck_bf16": (_i, [_p, _i, _i, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p,
                                   _i, _i, _i, _p, _p]),
    "sa_conv3x3_heads_bf16": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "sa_conv3x3_ex_heads_bf16": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "sa_pack_conv3x3_weights": (_i, [_p, _i, _i, _i, _i, _i, _i, _p]),
    "sa_conv3x3_packed_elems": (_sz, [_i, _i, _i]),
    "sa_conv3x3_set_grid_limit": (_i, [_i]),
    "sa_conv3x3_set_persistent": (_i, [_i]),
    "sa_convt3x3s2_bf16": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "sa_image_conv_bf16": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p, _p, _p]),
    "sa_imgconv_packed_elems": (C.c_size_t, [_i, _i, _i]),
    "sa_imgconv_pack": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "sa_imgconv_pack_tiled": (_i, [_p, _i, _i, _i, _p, _p, _p, _p]),
    "sa_imgconv_u8_bf16": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "sa_conv3x3_pair_bf16": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p]),
    "sa_add_bf16": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "sa_conv1x1_bf16": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p]),
    "sa_flow_scaled_size": (_i, [_i, _i, C.c_double, _p, _p]),
    "sa_flow_pyramid_build_scaled": (_i, [_p, _i, _i, _i, _i, C.c_double, _i, _i, _p, _p, _p]),
    "sa_convk_bf16": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p]),
    "sa_convt_s2_bf16": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p]),
    "sa_convt_s2_phase_taps": (_i, [_i, _i, _p, _p]),
    "sa_tapconv_packed_elems": (C.c_size_t, [_i, _i, _i]),
    "sa_pack_tapconv_weights": (_i, [_p, _i, _i, _i, _i, _i, _p]),
    "sa_maxpool_bf16": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "sa_tracker_create": (_p, [_p]),
    "sa_tracker_destroy": (None, [_p]),
    "sa_tracker_reset": (_i, [_p]),
    "sa_tracker_n_tracks": (_i, [_p]),
    "sa_tracker_last_first_choice": (_i, [_p]),
    "sa_tracker_track": (_i, [_p, _i, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "sa_tracker_track_frames": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "sa_tracker_set_image": (_i, [_p, _p, _i, _i, _i, _p]),
    "sa_tracker_track_frames_images": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _i, _i, _i, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "sa_connect_single_track_breaks": (_i, [_i, _i, _p, _p, _i]),
    "sa_maxpool2x2_bf16": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "sa_upsample2x_bf16": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "sa_conv1x1_head": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "sa_resize_bilinear_f32": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "sa_resize_bilinear_u8_f32": (_i, [_p, _i, _i, _i, _i, _i, _i, _f, _p, _p]),
    "sa_f32_to_bf16_padded": (_i, [_p, _i, _i, _i, _p, _p]),
    "sa_bf16_to_f32": (_i, [_p, _i, _i, _i, _p, _p]),
    "sa_tensor_absmax": (_i, [_p, _sz, _i, _p, _p]),
    "sa_network_create": (_i, [_p, _sz, _p]),
    "sa_network_destroy": (None, [_p]),
    "sa_network_n_outputs": (_i, [_p]),
    "sa_network_in_channels": (_i, [_p]),
    "sa_network_max_stride": (_i, [_p]),
    "sa_network_layout": (_i, [_p]),
    "sa_flow_pyramid_levels": (_i, [_i, _i, _i, _i]),
    "sa_flow_pyramid_bytes": (_sz, [_i, _i, _i, _i]),
    "sa_flow_pyramid_build": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "sa_flow_pyramid_build_batch": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "sa_flow_lk": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _f, _p]),
    "sa_flow_lk_pairs": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _f, _p]),
    "sa_network_output_shape": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "sa_network_workspace_bytes": (_sz, [_p, _i, _i, _i]),
    "sa_network_buffer": (_p, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "sa_network_forward": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "sa_bottomup_workspace_bytes": (_sz, [_p, _p, _i, _i, _i]),
    "sa_bottomup_predict": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "sa_h264_decode_slice": (_i, [_p, _p, C.c_int64, _p, _p, _p, _p]),
    "sa_yuv420_to_bgr": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _i]),
}

_lib = None
MISSING = []


class SleapAmdError(RuntimeError):
    pass


DTYPES = ("bf16", "fp16")  # 16-bit storage type of the network kernels: one library build each (csrc/bf16.h)
# fp16 is the default because it is the one that agrees END TO END with the reference's fp32 network (11 mantissa bits: heads
# within 0.1-0.5 % of range; bf16's 8 bits: 1-4 %, which flips marginal peak / matching decisions -- DESIGN.md section 4).
# bf16 has fp32's range and is the fallback for networks whose activations exceed 65504 (SA_STATUS_NONFINITE reports that).
DEFAULT_DTYPE = os.environ.get("SLEAP_AMD_DTYPE", "fp16")
if DEFAULT_DTYPE not in DTYPES:
    raise ValueError(f"SLEAP_AMD_DTYPE={DEFAULT_DTYPE!r}: expected one of {DTYPES}")
_libs = {}


def lib_path(dtype: str = None) -> str:
    dtype = dtype or DEFAULT_DTYPE
    override = os.environ.get(f"SLEAP_AMD_LIB_{dtype.upper()}")  # A/B experiments: a differently built library file
    if override:
        return override
    return LIB_PATH if dtype == "bf16" else os.path.join(HERE, "lib", f"libsleap_amd_{dtype}.so")


def lib(dtype: str = None):
    """Load (once) and return the ctypes handle of the library variant whose network kernels store activations and weights
    as `dtype` (None: DEFAULT_DTYPE). Post-processing, tracker and host entry points are identical in every variant.
    Raises if the HIP library is unavailable: there is no CPU fallback."""
    global _lib
    dtype = dtype or DEFAULT_DTYPE
    if dtype not in DTYPES:
        raise ValueError(f"unknown storage dtype {dtype!r}; one of {DTYPES}")
    if dtype in _libs:
        return _libs[dtype]
    # torch ships its own libamdhip64; it must be the first (and only) HIP runtime in the process, otherwise
    # device pointers handed over by torch belong to a different runtime than the one launching our kernels.
    import torch  # noqa: F401
    path = lib_path(dtype)
    if not os.path.exists(path):
        try:
            from . import build as _build

            _build.build(verbose=False, dtypes=(dtype,))
        except Exception as e:  # noqa: BLE001
            raise ImportError(
                f"{os.path.basename(path)} not found at {path} and could not be built ({e}); "
                "run `python -m sleap_amd.build` (needs hipcc). There is no CPU fallback."
            ) from e
    try:
        h = C.CDLL(path)
    except OSError as e:
        raise ImportError(f"could not load {path}: {e}. There is no CPU fallback.") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name, None)
        if fn is None:  # tests/test_abi.py asserts that this never happens for a released build
            MISSING.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    _libs[dtype] = h
    if dtype == DEFAULT_DTYPE:
        _lib = h
    return h


def check(code, what=""):
    if code != SA_OK:
        msg = lib().sa_last_error()
        raise SleapAmdError(f"{what} failed ({code}): {msg.decode() if msg else ''}")
