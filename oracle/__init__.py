"""CPU oracle for the SLEAP bottom-up inference hot path.

TEST INFRASTRUCTURE ONLY. Nothing under `sleap_amd/` may import this package; only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
only as the checker.

Every function restates, in NumPy float32 (network: torch-CPU float32), the algorithm of
the reference function named in its docstring (file:line relative to the reference
checkout of talmolab/sleap v1.4.1).

Pinning status (see DESIGN.md "Oracle"):
  * peak_finding / paf_grouping restatements: pinned by the reference's own known-answer
    tests (tests/nn/test_peak_finding.py, tests/nn/test_paf_grouping.py), re-expressed in
    tests/test_oracle_peak_finding.py and tests/test_oracle_paf_grouping.py.
  * network forward (Keras graph interpreter): PINNED to TensorFlow since round 6 for the layer kinds of a SLEAP UNet with a
    decoder (Conv2D, MaxPooling2D, Conv2DTranspose(k3, s2, same), Concatenate, the 1x1 heads; odd channel counts): on frame 0 of the reference's
    centered_pair_low_quality.mp4 the oracle reproduces the predictions TensorFlow wrote into
    tests/data/models/minimal_instance.UNet.bottomup/labels_pr.val.slp to 2e-5 px and 1e-6 in the scores
    (tests/test_frame0_golden.py), and the encoder-only robot model's within the reference's own tolerance
    (tests/test_oracle_network_pin.py); the centered-instance model's peaks on ground-truth crops likewise (3e-5 px).
    UpSampling2D(bilinear) -- the benchmark network's decoder -- BatchNormalization and the ResNet / hourglass graphs stay
    pinned to hand-derived vectors only (tests/layer_pin_vectors.py): no reference golden runs them.
  * NOT in this package: the checker of the native H.264 slice decoder (csrc/h264dec.hip). It is the pure-Python decoder
    sleap_amd/io/_h264.py + _h264_intra.py, which ships with the package because it is also `MediaVideo`'s selectable second engine
    (`engine="python"`); the product default is the native engine and there is no silent fall-back between them
    (tests/test_h264_native.py compares the two picture by picture; what pins the Python decoder: tests/test_frame0_golden.py,
    tests/test_h264_inter.py).
"""
