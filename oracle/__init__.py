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
  * network forward (Keras graph interpreter): parity UNPINNED numerically -- the
    reference can not be imported here (no TensorFlow) and its tests pin only layer
    shapes / parameter counts, which tests/test_oracle_keras_graph.py checks.
"""
