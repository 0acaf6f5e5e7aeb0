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

REFINE = {None: 0, "none": 0, "integral": 1, "local": 2, "offsets": 3}

SRC1_NONE, SRC1_DIRECT, SRC1_UPSAMPLE2X, SRC0_POOL2X = 0, 1, 2, 4

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/sleap_amd.h
SIGNATURES = {
    "sa_abi_version": (_i, []),
    "sa_last_error": (C.c_char_p, []),
    "sa_device_info": (_i, [_i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "sa_find_local_peaks_workspace": (_sz, [_i, _i]),
    "sa_find_local_peaks": (_i, [_p, _p, _i, _i, _i, _i, _f, _i, _i, _f, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "sa_find_global_peaks": (_i, [_p, _p, _i, _i, _i, _i, _f, _i, _i, _f, _p, _p, _p]),
    "sa_crop_and_resize": (_i, [_p, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p]),
    "sa_paf_score": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _i, _f, _f, _f, _i, _p, _p, _p, _p, _p]),
    "sa_paf_match": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "sa_paf_workspace": (_sz, [_i, _i, _i, _i]),
    "sa_paf_group": (_i, [_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "sa_lsa_host": (_i, [_p, _i, _i, _p, _p]),
    "sa_stem_conv3x3": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p]),
    "sa_conv3x3_bf16": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "sa_stem_conv3x3x2_bf16": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p, _i, _i, _p, _p, _p]),
    "sa_stem16_u8_bf16": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p]),
    "sa_stem16_pack": (_i, [_p, _p, _i, _i, _p, _p, _i, _p]),
    "sa_stem16_blob_bytes": (_sz, []),
    "sa_conv3x3_ex_bf16": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _p]),
    "sa_conv3x3_heads_bf16": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "sa_pack_conv3x3_weights": (_i, [_p, _i, _i, _i, _i, _i, _i, _p]),
    "sa_conv3x3_packed_elems": (_sz, [_i, _i, _i]),
    "sa_convt3x3s2_bf16": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "sa_image_conv_bf16": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _i, _p, _p, _p, _p]),
    "sa_imgconv_packed_elems": (C.c_size_t, [_i, _i, _i]),
    "sa_imgconv_pack": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p]),
    "sa_imgconv_u8_bf16": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "sa_conv3x3_pair_bf16": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "sa_add_bf16": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "sa_conv1x1_bf16": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p]),
    "sa_convt_s2_bf16": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p]),
    "sa_convt_s2_phase_taps": (_i, [_i, _i, _p, _p]),
    "sa_tapconv_packed_elems": (C.c_size_t, [_i, _i, _i]),
    "sa_pack_tapconv_weights": (_i, [_p, _i, _i, _i, _i, _i, _p]),
    "sa_maxpool_bf16": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "sa_tracker_create": (_p, [_p]),
    "sa_tracker_destroy": (None, [_p]),
    "sa_tracker_reset": (_i, [_p]),
    "sa_tracker_n_tracks": (_i, [_p]),
    "sa_tracker_track": (_i, [_p, _i, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "sa_tracker_track_frames": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "sa_connect_single_track_breaks": (_i, [_i, _i, _p, _p, _i]),
    "sa_maxpool2x2_bf16": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "sa_upsample2x_bf16": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "sa_conv1x1_head": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "sa_resize_bilinear_f32": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "sa_f32_to_bf16_padded": (_i, [_p, _i, _i, _i, _p, _p]),
    "sa_bf16_to_f32": (_i, [_p, _i, _i, _i, _p, _p]),
}

_lib = None
MISSING = []


class SleapAmdError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle. Raises if the HIP library is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; it must be the first (and only) HIP runtime in the process, otherwise
    # device pointers handed over by torch belong to a different runtime than the one launching our kernels.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        try:
            from . import build as _build

            _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise ImportError(
                f"libsleap_amd.so not found at {LIB_PATH} and could not be built ({e}); "
                "run `python -m sleap_amd.build` (needs hipcc). There is no CPU fallback."
            ) from e
    try:
        h = C.CDLL(LIB_PATH)
    except OSError as e:
        raise ImportError(f"could not load {LIB_PATH}: {e}. There is no CPU fallback.") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name, None)
        if fn is None:  # tests/test_abi.py asserts that this never happens for a released build
            MISSING.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = h
    return h


def check(code, what=""):
    if code != SA_OK:
        msg = lib().sa_last_error()
        raise SleapAmdError(f"{what} failed ({code}): {msg.decode() if msg else ''}")
