"""Native slice decoder (csrc/h264dec.hip) against the Python one (io/_h264.py), picture by picture: planes and motion data must be
equal.    python tools/h264_native_vs_python.py <file.mp4> [n_samples] [first sample: a key frame]     (result: profiles/r06_h264_native_vs_python.txt)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from sleap_amd.io import _h264 as D
from sleap_amd.io import _h264_intra as H

path = sys.argv[1]
tr = H.Mp4H264(path)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
assert first in tr.sync, tr.sync
n = min(int(sys.argv[2]) if len(sys.argv) > 2 else len(tr), len(tr) - first)
a, b = D.H264Decoder(tr.sps, tr.pps, "native"), D.H264Decoder(tr.sps, tr.pps, "python")
ta = tb = 0.0
types = {"I": 0, "P": 0, "B": 0}
for i in range(first, first + n):
    nals = tr.nal_units(i)
    t0 = time.perf_counter()
    pa = a.decode_sample(nals, i)
    t1 = time.perf_counter()
    pb = b.decode_sample(nals, i)
    t2 = time.perf_counter()
    ta, tb = ta + t1 - t0, tb + t2 - t1
    types[pa.stats["type"]] += 1
    assert pa.stats == pb.stats, (i, pa.stats, pb.stats)
    assert np.array_equal(pa.Y, pb.Y) and np.array_equal(pa.C[0], pb.C[0]) and np.array_equal(pa.C[1], pb.C[1]), f"sample {i}: planes differ"
    assert np.array_equal(pa.mv, pb.mv) and np.array_equal(pa.ref, pb.ref) and np.array_equal(pa.intra4.astype(bool), pb.intra4), f"sample {i}: motion data differs"
    if i % 100 == 0:
        print(i, pa.stats, f"native {ta:.2f} s, python {tb:.1f} s", flush=True)
print(f"{path}: {n} pictures {types} equal in planes and motion data; native {n / ta:.0f} pictures/s, python {n / tb:.1f} pictures/s")
