"""Frame sources and the prefetching feed (sleap_amd/io/video.py; reference sleap/io/video.py, nn/data/providers.py)."""
import os

import numpy as np
import pytest

from sleap_amd.io.video import FramePrefetcher, Video, VideoReader


@pytest.fixture
def clip():
    rng = np.random.default_rng(0)
    return rng.integers(0, 256, (23, 12, 16, 1), dtype=np.uint8)


def test_video_facade(clip, tmp_path):
    v = Video.from_numpy(clip)
    assert len(v) == 23 and v.shape == (23, 12, 16, 1) and (v.height, v.width, v.channels) == (12, 16, 1)
    assert np.array_equal(v.get_frame(5), clip[5]) and np.array_equal(v[3:9], clip[3:9]) and np.array_equal(v[-1], clip[-1])
    assert np.array_equal(v.get_frames([2, 7, 3]), clip[[2, 7, 3]]) and np.array_equal(v[::5], clip[::5])
    with pytest.raises(KeyError, match="Unable to load frame"):
        v.get_frame(23)
    np.save(tmp_path / "clip.npy", clip)
    m = Video.from_filename(str(tmp_path / "clip.npy"))  # memory mapped
    assert len(m) == 23 and np.array_equal(m[10:14], clip[10:14]) and m.backend_dict()["grayscale"]
    assert Video.from_numpy(clip[..., 0]).shape == (23, 12, 16, 1)  # (frames, h, w) gets a channel axis
    with pytest.raises(FileNotFoundError):  # (MediaVideo exists since round 6: a missing file fails like the reference's, video.py:370)
        Video.from_filename("movie.mp4")
    with pytest.raises(ValueError, match="Could not detect backend"):
        Video.from_filename("movie.xyz")


def test_video_reader_examples(clip):
    r = VideoReader(Video.from_numpy(clip), example_indices=[4, 9, 2])
    exs = list(r.make_dataset())
    assert len(r) == 3 and [int(e["frame_ind"]) for e in exs] == [4, 9, 2]
    e = exs[1]
    assert set(e) == set(r.output_keys) and np.array_equal(e["image"], clip[9])
    assert e["raw_image_size"].tolist() == [12, 16, 1] and e["raw_image_size"].dtype == np.int32
    assert e["video_ind"] == 0 and e["scale"].tolist() == [1.0, 1.0] and r.videos[0] is r.video
    assert len(VideoReader(Video.from_numpy(clip))) == 23


@pytest.mark.parametrize("depth", [2, 3, 5])
def test_prefetcher_order_and_content(clip, depth):
    ranges = [(i, min(i + 4, 23)) for i in range(0, 23, 4)]  # ragged last batch
    got = []
    pf = FramePrefetcher(clip, ranges, depth=depth, pin_memory=False)
    for lo, hi, inds, batch in pf:
        got.append((lo, hi, inds.tolist(), batch.numpy().copy()))
        pf.release()
    assert [(a, b) for a, b, _, _ in got] == ranges
    for lo, hi, inds, b in got:
        assert inds == list(range(lo, hi)) and np.array_equal(b, clip[lo:hi])


def test_prefetcher_sharded_indices_and_implicit_release(clip):
    r = VideoReader(Video.from_numpy(clip), example_indices=list(range(20, 2, -3)))  # 20, 17, 14, 11, 8, 5
    out = [(inds.tolist(), b.numpy().copy()) for _, _, inds, b in FramePrefetcher(r, [(1, 3), (3, 6)], pin_memory=False)]
    assert out[0][0] == [17, 14] and out[1][0] == [11, 8, 5]
    assert np.array_equal(out[1][1], clip[[11, 8, 5]])


def test_prefetcher_error_handling(clip):
    class Broken(Video):
        def get_frames(self, idxs):
            idxs = list(idxs)
            if 8 in idxs:
                raise KeyError("Unable to load frame 8 from MediaVideo.")
            return super().get_frames(idxs)

    # a seeking error ends the stream quietly (inference.py:3333-3339) ...
    got = [lo for lo, _, _, _ in FramePrefetcher(VideoReader(Broken.from_numpy(clip)), [(0, 4), (4, 8), (8, 12), (12, 16)],
                                                 pin_memory=False)]
    assert got == [0, 4]

    class Worse(Video):
        def get_frames(self, idxs):
            raise RuntimeError("disk on fire")

    with pytest.raises(RuntimeError, match="disk on fire"):  # ... anything else reaches the caller
        list(FramePrefetcher(VideoReader(Worse.from_numpy(clip)), [(0, 4)], pin_memory=False))


def test_hdf5_video(clip, tmp_path):
    """HDF5Video (sleap/io/video.py:47-338): channels_last / channels_first layouts and float [0,1] -> uint8 range conversion.
    The file is written by the interpreter that has h5py; reading works with or without h5py in this one."""
    import os
    import subprocess

    from sleap_amd.io.slp import _h5_python

    if not os.path.exists(_h5_python()):
        pytest.skip("no interpreter with h5py")
    path = str(tmp_path / "vid.h5")
    np.save(tmp_path / "a.npy", clip)
    code = ("import h5py, numpy as np, sys; a = np.load(sys.argv[1]); f = h5py.File(sys.argv[2], 'w'); "
            "f['box'] = a; f['cf'] = np.transpose(a, (0, 3, 2, 1)); f['flt'] = a.astype('float32') / 255.0; f.close()")
    subprocess.run([_h5_python(), "-c", code, str(tmp_path / "a.npy"), path], check=True)
    v = Video.from_hdf5("box", path)
    assert v.shape == (23, 12, 16, 1) and np.array_equal(v[4:9], clip[4:9]) and np.array_equal(v.get_frame(22), clip[22])
    cf = Video.from_hdf5("cf", path, input_format="channels_first")
    assert cf.shape == (23, 12, 16, 1) and np.array_equal(cf[0:3], clip[0:3])
    fl = Video.from_filename(path, dataset="flt")
    got = fl[2:6]
    assert got.dtype == np.uint8 and np.abs(got.astype(int) - clip[2:6].astype(int)).max() <= 1
    assert fl.backend_dict()["dataset"] == "flt"
    out = [b.numpy().copy() for _, _, _, b in FramePrefetcher(VideoReader(v), [(0, 5), (5, 10)], pin_memory=False)]
    assert np.array_equal(out[1], clip[5:10])


def test_npy_file_positional_reads_equal_memory_map(tmp_path):
    """NumpyVideo.read_into (pread, several threads) == slicing the memory map; declines what it can not serve."""
    from sleap_amd.io.video import FramePrefetcher, NumpyVideo, Video

    a = np.random.default_rng(5).integers(0, 256, (37, 96, 128, 1), dtype=np.uint8)
    path = str(tmp_path / "v.npy")
    np.save(path, a)
    nv = NumpyVideo(path)
    out = np.empty((9, 96, 128, 1), np.uint8)
    assert nv.read_into(5, 14, out) and np.array_equal(out, a[5:14])
    big = np.empty((37, 96, 128, 1), np.uint8)
    nv.READ_THREADS = 3
    assert nv.read_into(0, 37, big) and np.array_equal(big, a)
    assert not nv.read_into(0, 9, np.empty((9, 96, 128, 1), np.float32))  # dtype mismatch -> caller falls back
    assert not nv.read_into(0, 8, out)  # size mismatch
    assert not NumpyVideo(a).read_into(0, 9, out)  # in-memory array: nothing to pread
    got = [b.numpy()[: hi - lo].copy() for lo, hi, _, b in
           FramePrefetcher(Video.from_filename(path), [(0, 16), (16, 32), (32, 37)], depth=3, pin_memory=False)]
    assert np.array_equal(np.concatenate(got), a)


def test_prefetcher_hold_and_release_key_keep_buffers_until_released():
    """A consumer with several batches in flight (`Predictor._predict_generator`) takes buffers with hold() and hands them
    back later with release_key(): a held buffer must not be refilled, and the stream must still complete."""
    from sleap_amd.io.video import FramePrefetcher, Video

    a = np.arange(12 * 4 * 4, dtype=np.uint8).reshape(12, 4, 4, 1)
    fp = FramePrefetcher(Video.from_numpy(a), [(i, i + 2) for i in range(0, 12, 2)], depth=4, pin_memory=False)
    held, seen = [], []
    for lo, hi, inds, buf in fp:
        key = fp.hold()
        held.append((key, lo, hi, buf))
        if len(held) > 2:  # two batches stay in flight
            k, l, h, b = held.pop(0)
            assert np.array_equal(b.numpy()[: h - l], a[l:h])  # not overwritten while held
            seen.append((l, h))
            fp.release_key(k)
    for k, l, h, b in held:
        assert np.array_equal(b.numpy()[: h - l], a[l:h])
        seen.append((l, h))
        fp.release_key(k)
    assert seen == [(i, i + 2) for i in range(0, 12, 2)]


# ------------------------------------------------------------------------------------------ image-sequence videos
def test_images_video_reference_expectations(tmp_path):
    """The expectations of the reference's tests/io/test_video.py:359-369 and :30-41 (three 560 x 320 colour JPEGs: frames,
    height, width, channels, the shape of `vid[0]`, backend detection by extension), on JPEGs written here."""
    from PIL import Image

    from sleap_amd.io.video import SingleImageVideo, Video

    rng = np.random.default_rng(1)
    filenames = []
    for i in range(3):
        base = np.kron(rng.integers(0, 256, (20, 35, 3), dtype=np.uint8), np.ones((16, 16, 1), np.uint8))  # 320 x 560 blocks
        f = str(tmp_path / f"robot{i}.jpg")
        Image.fromarray(base).save(f, quality=90)
        filenames.append(f)
    vid = Video.from_image_filenames(filenames)
    assert vid.frames == len(filenames) and vid.height == 320 and vid.width == 560 and vid.channels == 3
    assert vid[0:1].shape == (1, 320, 560, 3) and vid.get_frame(1).dtype == np.uint8
    assert type(Video.from_filename(filenames[0]).backend) is SingleImageVideo
    with pytest.raises(ValueError):
        Video.from_filename("this_has_no_video_extension")
    d = vid.backend_dict()  # the attrs fields the reference serialises for this backend
    assert d["filenames"] == filenames and d["grayscale"] is False and (d["height_"], d["width_"], d["channels_"]) == (320, 560, 3)


def test_image_sequence_lossless_round_trip_grayscale_detection_and_feed(tmp_path):
    """PNG / TIFF are lossless: frames come back exactly; a gray sequence is detected (channel 0 == channel 2) and presented
    with one channel as the reference does; a moved folder is found through `filename`; the prefetcher feeds from it."""
    from PIL import Image

    from sleap_amd.io.video import FramePrefetcher, SingleImageVideo, Video

    rng = np.random.default_rng(0)
    gray = rng.integers(0, 256, (7, 40, 56), dtype=np.uint8)
    rgb = rng.integers(0, 256, (3, 40, 56, 3), dtype=np.uint8)
    gfiles, cfiles = [], []
    for i, g in enumerate(gray):
        f = str(tmp_path / f"g{i:02d}.{'png' if i % 2 else 'tif'}")
        Image.fromarray(g).save(f)
        gfiles.append(f)
    for i, c in enumerate(rgb):
        f = str(tmp_path / f"c{i}.png")
        Image.fromarray(c).save(f)
        cfiles.append(f)
    v = Video.from_image_filenames(gfiles)
    assert v.shape == (7, 40, 56, 1) and v.backend.grayscale is True
    assert np.array_equal(v[0:7], gray[..., None]) and np.array_equal(v.get_frame(3), gray[3][..., None])
    assert np.array_equal(Video.from_image_filenames(cfiles)[0:3], rgb)
    assert SingleImageVideo(filenames=gfiles, grayscale=False).get_frame(0).shape == (40, 56, 3)  # explicit override
    got = [b.numpy()[: hi - lo].copy() for lo, hi, _, b in FramePrefetcher(v, [(0, 4), (4, 7)], depth=3, pin_memory=False)]
    assert np.array_equal(np.concatenate(got), gray[..., None])
    moved = SingleImageVideo(filename=gfiles[0], filenames=["/nonexistent/" + os.path.basename(f) for f in gfiles])
    assert np.array_equal(moved.get_frame(5)[..., 0], gray[5])  # video.py:845-856: looked up next to `filename`
    with pytest.raises(FileNotFoundError):
        SingleImageVideo(filename=gfiles[0], filenames=["/nonexistent/zzz.png"]).get_frame(0)
