"""Frame sources and the per-rank frame feed.

The reference reads frames one at a time on a single Python thread (`tf.py_function(video.get_frame)`,
sleap/nn/data/providers.py:371-439, "we don't parallelize here for thread safety"), which caps any fast predictor. Here:

* `Video` -- the thin facade of sleap/io/video.py:1023-1508 over array-like backends: `NumpyVideo` (in-memory array or a
  memory-mapped `.npy`, sleap/io/video.py:511-590) and `HDF5Video` (dataset of frames in an HDF5 file with the reference's
  `input_format` / `convert_range` options, sleap/io/video.py:47-338). `MediaVideo` (sleap/io/video.py:340-504, cv2 / FFmpeg
  there): no decoder exists in this image or on the GPU box, so round 6 brought its own -- progressive 8-bit 4:2:0 Baseline / Main / High-
  profile H.264 (I, P and B pictures, CAVLC and CABAC, 8x8 transform) in MP4 (io/_h264.py over io/_h264_intra.py, native engine
  csrc/h264dec.hip).
* `VideoReader` -- the provider surface (`videos`, `example_indices`, `len`, `make_dataset()` yielding the same example
  dictionaries).
* `FramePrefetcher` -- the throughput piece: a producer thread reads whole batches ahead of the consumer into a small ring
  of page-locked host buffers; the predictor uploads them with asynchronous copies on its network stream, so reading,
  H2D transfer, network and post-processing of consecutive batches overlap. Under torch.distributed every rank runs its
  own prefetcher over its own contiguous slice of each global batch (no rank reads another rank's frames).
"""
import os
import queue
import threading
from typing import Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np


class NumpyVideo:
    """sleap/io/video.py:511-590: frames as an array (frames, height, width, channels)."""

    def __init__(self, filename: Union[str, np.ndarray]):
        if isinstance(filename, str):
            self.filename = filename
            self._data = np.load(filename, mmap_mode="r")
        else:
            self.filename = "numpy array"
            self._data = filename
        if self._data.ndim == 3:
            self._data = self._data[..., None]
        if self._data.ndim != 4:
            raise ValueError(f"video array must have shape (frames, height, width, channels), got {self._data.shape}")

    frames = property(lambda self: self._data.shape[0])
    height = property(lambda self: self._data.shape[1])
    width = property(lambda self: self._data.shape[2])
    channels = property(lambda self: self._data.shape[3])
    dtype = property(lambda self: self._data.dtype)

    def get_frame(self, idx: int) -> np.ndarray:
        return np.asarray(self._data[idx])

    def get_frames(self, lo: int, hi: int) -> np.ndarray:
        return np.asarray(self._data[lo:hi])

    READ_THREADS = int(os.environ.get("SLEAP_AMD_READ_THREADS", "8"))

    def read_into(self, lo: int, hi: int, out: np.ndarray) -> bool:
        """Frames [lo, hi) of a `.npy` FILE straight into `out` (C-contiguous, same dtype) with positional reads, a few threads
        wide (1 thread 6.5 k, 4: 8.5 k, 8: 9.9 k, 12: 9.3 k frames/s of 1024 x 1024 from the page cache). Copying out of the memory map instead faults the file in page by page: 6.9 GB/s = 6.5 k frames/s
        on the GPU box, less than the network consumes; pread from the page cache is not bound by that. -> False when the
        source is not a plain C-ordered file-backed array (the caller falls back to `get_frames`)."""
        d = self._data
        if not isinstance(d, np.memmap) or not d.flags.c_contiguous or out.dtype != d.dtype or not out.flags.c_contiguous:
            return False
        frame_bytes = int(np.prod(d.shape[1:])) * d.dtype.itemsize
        if out.nbytes != (hi - lo) * frame_bytes:
            return False
        if getattr(self, "_fd", None) is None:
            self._fd = os.open(d.filename, os.O_RDONLY)
            self._pool = None
        base = d.offset + lo * frame_bytes
        mv = memoryview(out).cast("B")
        n = out.nbytes

        def rd(a, b):
            pos = a
            while pos < b:
                got = os.preadv(self._fd, [mv[pos:b]], base + pos)
                if got <= 0:
                    raise IOError(f"short read from {d.filename}")
                pos += got

        k = max(1, min(self.READ_THREADS, n >> 22))
        if k == 1:
            rd(0, n)
            return True
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(max_workers=self.READ_THREADS)
        step = -(-n // k)
        step += (-step) % 4096
        list(self._pool.map(lambda a: rd(a, min(a + step, n)), range(0, n, step)))
        return True


class HDF5Video:
    """sleap/io/video.py:47-338: a frames dataset inside an HDF5 file.

    input_format "channels_last" = (frames, height, width, channels), "channels_first" = (frames, channels, width, height)
    (video.py:114-126 transposes 2 <-> 1 and moves channels last); convert_range: float data in [0, 1] is scaled to
    uint8 [0, 255] (video.py:306-312). Read with h5py when importable, else through a one-time .npy copy."""

    def __init__(self, filename: str, dataset: str, input_format: str = "channels_last", convert_range: bool = True):
        if input_format not in ("channels_last", "channels_first"):
            raise ValueError(f"unknown input_format {input_format}")
        self.filename, self.dataset, self.input_format, self.convert_range = filename, dataset, input_format, convert_range
        try:
            import h5py

            self._f = h5py.File(filename, "r")
            self._d = self._f[dataset]
        except ImportError:
            # no h5py in this interpreter: the frames dataset is copied ONCE to a memory-mapped .npy next to the file by
            # sleap_amd/io/_slp_io.py under the interpreter that has h5py (SLEAP_AMD_H5_PYTHON), as model_io does for best_model.h5
            import subprocess

            from .slp import _h5_python, _tool

            cache = f"{filename}.{dataset.strip('/').replace('/', '_')}.npy"
            if not os.path.exists(cache) or os.path.getmtime(cache) < os.path.getmtime(filename):
                subprocess.run([_h5_python(), _tool(), "frames", filename, dataset, cache], check=True)
            self._d = np.load(cache, mmap_mode="r")
        s = self._d.shape
        if len(s) != 4:
            raise ValueError(f"dataset {dataset} must be 4-D, got {s}")
        self._shape = (s[0], s[1], s[2], s[3]) if input_format == "channels_last" else (s[0], s[3], s[2], s[1])

    frames = property(lambda self: self._shape[0])
    height = property(lambda self: self._shape[1])
    width = property(lambda self: self._shape[2])
    channels = property(lambda self: self._shape[3])

    @property
    def dtype(self):
        return np.dtype(np.uint8) if (self.convert_range and self._d.dtype.kind == "f") else self._d.dtype

    def _fix(self, x):
        if self.input_format == "channels_first":
            x = np.transpose(x, (0, 3, 2, 1))
        if self.convert_range and x.dtype.kind == "f" and np.max(x, initial=0.0) <= 1.0:
            x = (x * 255).astype(np.uint8)
        return x

    def get_frame(self, idx: int) -> np.ndarray:
        return self._fix(self._d[idx:idx + 1])[0]

    def get_frames(self, lo: int, hi: int) -> np.ndarray:
        return self._fix(self._d[lo:hi])


class SingleImageVideo:
    """sleap/io/video.py:803-998: a list of image files (jpg / jpeg / png / tif / tiff) presented as a video.

    The reference decodes with `cv2.imread` (always 3 channels, 8 bit, BGR) and flips to RGB; here Pillow decodes and
    converts to 8-bit RGB, which is the same array for 8-bit grayscale / RGB / RGBA PNG and TIFF files (lossless formats).
    JPEG: both sit on libjpeg(-turbo), but their IDCT / upsampling choices are not guaranteed bit-identical -- parity unpinned
    for lossy files. `grayscale=None` detects it from the first frame as the reference does (channel 0 == channel 2
    everywhere -> one channel is presented); missing files are looked up next to `filename` (video.py:845-856)."""

    EXTS = ("jpg", "jpeg", "png", "tif", "tiff")
    DECODE_THREADS = int(os.environ.get("SLEAP_AMD_READ_THREADS", "8"))

    def __init__(self, filename: Optional[str] = None, filenames: Optional[Sequence[str]] = None, height_: Optional[int] = None,
                 width_: Optional[int] = None, channels_: Optional[int] = None, grayscale: Optional[bool] = None):
        filenames = list(filenames or [])
        if not filename and filenames:
            filename = filenames[0]
        elif filename and not filenames:
            filenames = [filename]
        if not filenames:
            raise ValueError("SingleImageVideo needs at least one image file")
        self.filename, self.filenames = filename, filenames
        self.height_, self.width_, self.channels_ = height_, width_, channels_
        self._detect_grayscale = grayscale is None
        self.grayscale = bool(grayscale) if grayscale is not None else False
        self._probed = False
        self._pool = None

    def _get_filename(self, idx: int) -> str:
        f = self.filenames[idx]
        if os.path.exists(f):
            return f
        g = os.path.join(os.path.dirname(self.filename), os.path.basename(f))
        if os.path.exists(g):
            return g
        raise FileNotFoundError(f"Unable to locate file {idx}: {self.filenames[idx]}")

    def _load_idx(self, idx: int) -> np.ndarray:
        from PIL import Image

        with Image.open(self._get_filename(idx)) as im:
            return np.asarray(im.convert("RGB"))

    def _probe(self):
        if not self._probed:
            f = self._load_idx(0)
            if self._detect_grayscale:
                self.grayscale = bool((f[..., 0] == f[..., -1]).all())
            self.height_ = f.shape[0] if self.height_ is None else self.height_
            self.width_ = f.shape[1] if self.width_ is None else self.width_
            self.channels_ = f.shape[2] if self.channels_ is None else self.channels_
            self._probed = True

    frames = property(lambda self: len(self.filenames))
    dtype = property(lambda self: np.dtype(np.uint8))

    @property
    def height(self):
        self._probe()
        return self.height_

    @property
    def width(self):
        self._probe()
        return self.width_

    @property
    def channels(self):
        self._probe()
        return 1 if self.grayscale else self.channels_

    def get_frame(self, idx: int, grayscale: Optional[bool] = None) -> np.ndarray:
        self._probe()
        frame = self._load_idx(int(idx))
        if self.grayscale if grayscale is None else grayscale:
            frame = frame[..., 0][..., None]
        return frame

    def get_frames(self, lo: int, hi: int) -> np.ndarray:
        """Decoded by a few threads (Pillow releases the GIL while it decodes)."""
        self._probe()
        n = hi - lo
        if n <= 2 or self.DECODE_THREADS <= 1:
            return np.stack([self.get_frame(i) for i in range(lo, hi)])
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(max_workers=self.DECODE_THREADS)
        return np.stack(list(self._pool.map(self.get_frame, range(lo, hi))))

    def backend_dict(self) -> dict:
        """The attrs fields of the reference's backend, as its `.slp` files store them."""
        return {"filename": self.filename, "filenames": list(self.filenames), "height_": self.height, "width_": self.width,
                "channels_": self.channels_, "grayscale": self.grayscale}


class MediaVideo:
    """sleap/io/video.py:340-504 (`MediaVideo`: cv2.VideoCapture over FFmpeg) for H.264 in MP4 / MOV through the package's own
    decoder (neither this image nor the GPU box holds one: profiles/r06_decoder_probe.txt): io/_h264.py decodes the I, P and B
    pictures of progressive 8-bit 4:2:0 Baseline / Main / High-profile streams (CAVLC and CABAC, 8x8 transform: all six .mp4
    files of the reference's test data); scaling matrices, fields and several slices per picture raise NotImplementedError naming
    the missing coding tool. Frame k is the k-th picture in PRESENTATION order (the MP4's composition
    times), as cv2 numbers frames. The macroblock layer runs in the package's library (`sa_h264_decode_slice`, host C++:
    325-545 pictures/s on one core); sequential reads decode every picture once, a jump decodes from the key frame in front of the target (`keyframes`). Colour conversion, channel handling and the
    `grayscale` / `bgr` attributes follow the reference: BGR as libswscale delivers it to cv2, `grayscale` "auto" = detected on
    the first frame (all channels equal), a grayscale video yields channel 0, `bgr=True` reverses the channel order of colour
    frames. An index past the end raises `KeyError` like the reference's failed read (video.py:497-498)."""

    EXTS = ("mp4", "mov", "m4v")

    def __init__(self, filename: str, grayscale: Optional[bool] = None, bgr: bool = True, workers: Optional[int] = None):
        from . import _h264, _h264_intra

        if not os.path.isfile(filename):
            raise FileNotFoundError(f"Could not find filename video filename named {filename}")
        self.filename, self.bgr = filename, bgr
        self.dataset, self.input_format = "", ""
        self._track = _h264_intra.Mp4H264(filename)
        self._reader = _h264.H264Reader(self._track)
        # `workers` > 1: whole GOPs decoded ahead on threads (the native macroblock layer runs without the GIL); default = the
        # CPU quota up to 8, 1 = the sequential reader only
        if workers is None:
            try:
                workers = min(len(os.sched_getaffinity(0)), 8)
            except AttributeError:
                workers = 1
        self._blue = _h264_intra.swscale_blue
        self._gops = _h264.GopPool(self._track, workers, convert_pic=self._convert_pic) if workers > 1 else None
        if self._gops is not None and not self._gops.closed:
            self._gops = None
        self._swscale = _h264_intra.swscale_bgr
        self._cache = {}
        self._lock = threading.Lock()  # (the decoder is a sequential state machine)
        self.grayscale = grayscale
        if grayscale is None:  # (video.py:391-396: detect on the first frame)
            t = self._bgr_frame(0, want_bgr=True)
            self.grayscale = bool(np.all(t[..., 0] == t[..., -1]))

    frames = property(lambda self: len(self._track))
    height = property(lambda self: self._track.height)
    width = property(lambda self: self._track.width)
    channels = property(lambda self: 1 if self.grayscale else 3)
    dtype = property(lambda self: np.dtype(np.uint8))
    fps = property(lambda self: self._track.fps)
    keyframes = property(lambda self: [self._track.display_order.index(s) for s in self._track.sync])

    def _convert(self, y, cb, cr):
        """planes -> what the reference's reader yields before its channel handling: BGR, or -- once the video is known to be
        grayscale -- its channel 0 alone (one table look-up for a grey stream). Runs on the decode threads."""
        if self.grayscale:
            return self._blue(y, cb)[..., None]
        return self._swscale(y, cb, cr)

    def _convert_pic(self, pic):
        """the same conversion straight from a natively decoded picture's buffers by the library (`sa_yuv420_to_bgr`: no GIL, no
        intermediate copies) -- what the decode threads run"""
        import ctypes as C

        from .. import _lib

        cl, cr, ct, cb = self._track.sps["crop"]
        h, w = self._track.height, self._track.width
        ch = 1 if self.grayscale else 3
        out = np.empty((h, w, ch), np.uint8)
        ys, cs = pic.Y.shape[1], pic.C[0].shape[1]
        rc = _lib.lib().sa_yuv420_to_bgr(C.c_void_p(pic.Y.ctypes.data + 2 * ct * ys + 2 * cl), C.c_void_p(pic.C[0].ctypes.data + ct * cs + cl),
                                         C.c_void_p(pic.C[1].ctypes.data + ct * cs + cl), w, h, ys, cs, C.c_void_p(out.ctypes.data), ch)
        _lib.check(rc, "sa_yuv420_to_bgr")
        return out

    def _bgr_frame(self, idx: int, want_bgr: bool = False) -> np.ndarray:
        """-> (H, W, 3) BGR, or (H, W, 1) channel 0 for a grayscale video unless `want_bgr`"""
        with self._lock:
            key = (idx, bool(want_bgr or not self.grayscale))
            if key not in self._cache:
                if len(self._cache) >= 16:
                    self._cache.pop(next(iter(self._cache)))
                try:
                    if key[1] and self.grayscale is not False:  # all three channels of a (possibly) grayscale video: from the planes
                        y, cb, cr = self._reader.frame(idx)
                        self._cache[key] = self._swscale(y, cb, cr)
                    elif self._gops is not None:
                        self._cache[key] = self._gops.frame(idx)
                    else:
                        self._cache[key] = self._convert(*self._reader.frame(idx))
                except IndexError as e:
                    raise KeyError(f"Unable to load frame {idx} from {self.filename}: the video has {len(self._track)} frames") from e
            return self._cache[key]

    def get_frame(self, idx: int, grayscale: Optional[bool] = None) -> np.ndarray:
        gray = self.grayscale if grayscale is None else grayscale
        frame = self._bgr_frame(int(idx), want_bgr=not gray)
        if gray and frame.shape[-1] != 1:
            frame = frame[..., 0][..., None]
        if self.bgr and frame.shape[-1] == 3:
            frame = frame[..., ::-1]
        return np.ascontiguousarray(frame)

    def get_frames(self, lo: int, hi: int) -> np.ndarray:
        return np.stack([self.get_frame(i) for i in range(lo, hi)]) if hi > lo else np.zeros((0, self.height, self.width, self.channels), np.uint8)

    def backend_dict(self) -> dict:
        return {"filename": str(self.filename), "grayscale": bool(self.grayscale), "bgr": bool(self.bgr), "dataset": "", "input_format": ""}


class Video:
    """sleap/io/video.py:1023-1508 (`sleap.Video`): the facade the predictor, the providers and the writer talk to."""

    def __init__(self, backend):
        self.backend = backend

    @classmethod
    def from_numpy(cls, filename: Union[str, np.ndarray], *args, **kwargs) -> "Video":
        return cls(NumpyVideo(filename))

    @classmethod
    def from_hdf5(cls, dataset: str, filename: str, input_format: str = "channels_last", convert_range: bool = True) -> "Video":
        return cls(HDF5Video(filename, dataset, input_format, convert_range))

    @classmethod
    def from_image_filenames(cls, filenames: Sequence[str], height: Optional[int] = None, width: Optional[int] = None,
                             *args, **kwargs) -> "Video":
        """video.py:1227-1241: individual image files as one video."""
        return cls(SingleImageVideo(filenames=list(filenames), height_=height, width_=width))

    @classmethod
    def from_media(cls, filename: str, *args, **kwargs) -> "Video":
        """video.py:1211-1225: a media file (here: Baseline / Main-profile H.264 in MP4 / MOV)."""
        return cls(MediaVideo(filename, *args, **kwargs))

    @classmethod
    def from_filename(cls, filename: str, dataset: Optional[str] = None, input_format: str = "channels_last", **kwargs) -> "Video":
        ext = os.path.splitext(filename)[1].lower()
        if ext.lstrip(".") in SingleImageVideo.EXTS:
            return cls(SingleImageVideo(filename=filename, grayscale=kwargs.get("grayscale")))
        if ext == ".npy":
            return cls.from_numpy(filename)
        if ext in (".h5", ".hdf5", ".slp"):
            if dataset is None:
                raise ValueError("an HDF5 video needs the name of its frames dataset")
            return cls.from_hdf5(dataset, filename, input_format)
        if ext.lstrip(".") in MediaVideo.EXTS:
            return cls(MediaVideo(filename, grayscale=kwargs.get("grayscale"), bgr=kwargs.get("bgr", True)))
        if ext in (".avi", ".mj2", ".mkv"):
            raise NotImplementedError("only H.264 in MP4 / MOV containers can be read (sleap_amd.io.video.MediaVideo): "
                                      "no general video decoder (cv2 / ffmpeg) exists in this environment")
        raise ValueError(f"Could not detect backend for specified filename: {filename}")

    num_frames = frames = property(lambda self: self.backend.frames)
    height = property(lambda self: self.backend.height)
    width = property(lambda self: self.backend.width)
    channels = property(lambda self: self.backend.channels)
    dtype = property(lambda self: self.backend.dtype)
    shape = property(lambda self: (self.backend.frames, self.backend.height, self.backend.width, self.backend.channels))
    filename = property(lambda self: self.backend.filename)

    def __len__(self) -> int:
        return self.backend.frames

    def get_frame(self, idx: int) -> np.ndarray:
        if idx < 0 or idx >= len(self):
            raise KeyError(f"Unable to load frame {idx} from {type(self.backend).__name__}.")  # video.py:403-405 wording
        return self.backend.get_frame(int(idx))

    def get_frames(self, idxs) -> np.ndarray:
        idxs = list(idxs)
        if idxs and idxs == list(range(idxs[0], idxs[0] + len(idxs))) and 0 <= idxs[0] and idxs[-1] < len(self):
            return self.backend.get_frames(idxs[0], idxs[-1] + 1)
        return np.stack([self.get_frame(i) for i in idxs]) if idxs else np.zeros((0,) + self.shape[1:], self.dtype)

    def __getitem__(self, key):
        if isinstance(key, slice):
            lo, hi, step = key.indices(len(self))
            if step == 1:
                return self.backend.get_frames(lo, hi)
            return self.get_frames(range(lo, hi, step))
        if isinstance(key, (int, np.integer)):
            return self.get_frame(int(key) % len(self) if key < 0 else int(key))
        return self.get_frames(key)

    def backend_dict(self) -> dict:
        """The `backend` record the `.slp` writer stores in `videos_json`."""
        if hasattr(self.backend, "backend_dict"):
            return self.backend.backend_dict()
        return {"filename": str(self.backend.filename), "grayscale": self.channels == 1, "bgr": True,
                "dataset": getattr(self.backend, "dataset", ""), "input_format": getattr(self.backend, "input_format", "")}


class VideoReader:
    """sleap/nn/data/providers.py:301-439: `VideoReader(video, example_indices=None)`."""

    def __init__(self, video: Video, example_indices: Optional[Sequence[int]] = None):
        self.video = video
        self.example_indices = None if example_indices is None else [int(i) for i in example_indices]

    @classmethod
    def from_filepath(cls, filename: str, example_indices=None, **kwargs) -> "VideoReader":
        return cls(Video.from_filename(filename, **kwargs), example_indices)

    @property
    def videos(self) -> List[Video]:
        return [self.video]

    @property
    def output_keys(self) -> List[str]:
        return ["image", "raw_image_size", "video_ind", "frame_ind", "scale"]

    def indices(self) -> List[int]:
        return list(range(len(self.video))) if self.example_indices is None else self.example_indices

    def __len__(self) -> int:
        return len(self.indices())

    def make_dataset(self) -> Iterator[dict]:
        for i in self.indices():
            img = self.video.get_frame(i)
            yield {"image": img, "raw_image_size": np.array(img.shape, dtype=np.int32), "video_ind": 0,
                   "frame_ind": np.int64(i), "scale": np.ones((2,), np.float32)}


class FramePrefetcher:
    """Reads batches `[(lo, hi), ...]` (positions in `reader.indices()`) ahead of the consumer.

    Yields `(lo, hi, frame_inds, batch)`; `batch` is a page-locked uint8/float32 torch tensor (a plain one without CUDA)
    that stays valid until the next-but-`depth - 1` item has been requested -- call `release(event)` with a recorded
    torch.cuda.Event after queueing the upload if the copy is asynchronous. A `KeyError("Unable to load frame ...")`
    from the source ends the stream quietly, as the reference does (inference.py:3333-3339); other errors re-raise."""

    def __init__(self, reader: Union[VideoReader, Video, np.ndarray], ranges: Sequence[Tuple[int, int]], depth: int = 3,
                 pin_memory: Optional[bool] = None):
        import torch

        if isinstance(reader, np.ndarray):
            reader = Video.from_numpy(reader)
        if isinstance(reader, Video):
            reader = VideoReader(reader)
        self.reader, self.ranges, self.depth = reader, list(ranges), max(2, int(depth))
        self._idx = reader.indices()
        self._pin = torch.cuda.is_available() if pin_memory is None else bool(pin_memory)
        self._q: "queue.Queue" = queue.Queue(maxsize=self.depth - 1)
        self._free: "queue.Queue" = queue.Queue()
        self._bufs = []
        self._events = {}
        bmax = max((hi - lo for lo, hi in self.ranges), default=0)
        v = reader.video
        for k in range(self.depth):
            # page-locked straight from torch's caching host allocator: the blocks of the previous predict() call are handed
            # out again (no new page locking); `empty().pin_memory()` locked fresh pages AND copied 64 MB per buffer on every
            # call -- 3 buffers = ~90 ms of a 205 ms predict() over 1280 frames (tools/predict_e2e.py)
            t = torch.empty((bmax, v.height, v.width, v.channels), dtype=torch.uint8 if v.dtype == np.uint8 else torch.float32,
                            pin_memory=bool(self._pin and bmax))
            self._bufs.append(t)
            self._free.put(k)
        self._pool, self.copy_threads = None, int(os.environ.get("SLEAP_AMD_COPY_THREADS", "1"))
        self._thread = threading.Thread(target=self._produce, daemon=True)
        self._pending_release = None
        self._thread.start()

    def _produce(self):
        try:
            for lo, hi in self.ranges:
                inds = self._idx[lo:hi]
                k = self._free.get()
                ev = self._events.pop(k, None)
                if ev is not None:
                    ev.synchronize()  # the upload that read this buffer last has finished
                buf = self._bufs[k][: hi - lo]
                try:
                    rd = getattr(self.reader.video.backend, "read_into", None)
                    contiguous = len(inds) > 0 and int(inds[-1]) - int(inds[0]) + 1 == len(inds) and \
                        (len(inds) == 1 or bool((np.diff(np.asarray(inds)) == 1).all()))
                    if not (rd is not None and contiguous and 0 <= int(inds[0]) and int(inds[-1]) < len(self.reader.video)
                            and rd(int(inds[0]), int(inds[-1]) + 1, buf.numpy())):
                        frames = self.reader.video.get_frames(inds)
                        self._stage(buf.numpy(), frames)  # straight into the page-locked buffer (no temporary)
                except KeyError as e:
                    if "Unable to load frame" in str(e):
                        break
                    raise
                self._q.put((lo, hi, np.asarray(inds, dtype=np.int64), buf, k))
            self._q.put(None)
        except BaseException as e:  # noqa: BLE001 - handed to the consumer thread
            self._q.put(e)

    def _stage(self, dst: np.ndarray, src: np.ndarray):
        """Copy (and cast) one batch into the page-locked buffer. One thread moves 64 frames of 1024 x 1024 in 2.5 ms
        (measured on the GPU box), well under the 6.9 ms the network takes, so the default is a plain copy;
        SLEAP_AMD_COPY_THREADS > 1 splits the batch over threads (NumPy releases the GIL in its copy loops) for slower
        sources -- mind container CPU quotas."""
        n = len(src)
        if n < 8 or src.nbytes < (8 << 20) or self.copy_threads <= 1:
            np.copyto(dst, src, casting="unsafe")
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(max_workers=self.copy_threads)
        step = -(-n // self.copy_threads)
        list(self._pool.map(lambda a: np.copyto(dst[a:a + step], src[a:a + step], casting="unsafe"), range(0, n, step)))

    def __iter__(self):
        while True:
            if self._pending_release is not None:  # the previous item was not released explicitly: assume a synchronous use
                self._free.put(self._pending_release)
                self._pending_release = None
            item = self._q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            lo, hi, inds, buf, k = item
            self._pending_release = k
            yield lo, hi, inds, buf

    def hold(self):
        """Take responsibility for the most recently yielded buffer: it is NOT handed back when the next item is requested;
        call `release_key(key, event)` later (a consumer that keeps several batches in flight)."""
        k, self._pending_release = self._pending_release, None
        return k

    def release_key(self, k, event=None):
        if k is not None:
            if event is not None:
                self._events[k] = event
            self._free.put(k)

    def release(self, event=None):
        """Hand the most recently yielded buffer back; `event` (torch.cuda.Event) marks the end of its asynchronous upload."""
        k = self._pending_release
        if k is not None:
            if event is not None:
                self._events[k] = event
            self._free.put(k)
            self._pending_release = None
