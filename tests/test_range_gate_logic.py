"""The fp16 range gate's decisions under torch.distributed, on CPU (inside an initialised process group `_range_gate`
only recorded its scan, even when the scan saw inf / NaN, for ANY caller -- now only for callers that promised an agreement).
`DeviceNetwork._range_gate` is exercised as a plain function on a stand-in object: the decisions do not touch the device."""
import types
import warnings

import pytest


def _net(scan):
    """a stand-in with the attributes `_range_gate` reads; `scan` = (largest finite value, inf / NaN seen, offending tensor)"""
    from sleap_amd.nn.engine import DeviceNetwork

    n = types.SimpleNamespace(_pending_scan=None, _pending_imgs=None, _dist_agreed=False, _dist_defer=False, _range_checked=True,
                              range_log2_scale=None, _range_calibrated=False, range_safe=True, scans=[scan])
    n._check_fp16_range = lambda bufs: n.scans[-1]
    n._apply_range_scaling = lambda imgs, must, quiet=False: pytest.fail("nothing may be rescaled inside a process group")
    n.gate = lambda imgs="batch": DeviceNetwork._range_gate(n, {}, imgs)
    n.defer = lambda: DeviceNetwork.defer_range_agreement(n)
    return n


@pytest.fixture
def world2(monkeypatch):
    from sleap_amd.nn import range_scaling

    monkeypatch.setattr(range_scaling, "dist_world", lambda: 2)


def test_without_a_promise_an_overflow_inside_a_process_group_raises_at_once(world2):
    n = _net((1e9, True, 7))
    with pytest.raises(FloatingPointError, match="defer_range_agreement"):
        n.gate()
    assert n._pending_scan is None and n._pending_imgs is None and n._range_checked is False  # nothing recorded, scanned again next time


def test_without_a_promise_a_fitting_network_is_left_alone_and_a_near_overflow_warns(world2):
    n = _net((100.0, False, None))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert n.gate() is False
    n = _net((30000.0, False, None))
    with pytest.warns(UserWarning, match="within 4x"):
        assert n.gate() is False
    assert n._pending_scan is None


def test_deferred_mode_records_one_batch_and_raises_when_the_promise_is_not_kept(world2):
    n = _net((1e9, True, 3))
    n.defer()
    assert n.gate("first batch") is False           # recorded, not raised: the caller promised dist_agree_range()
    assert n._pending_scan == (1e9, True) and n._pending_imgs == "first batch"
    with pytest.raises(FloatingPointError, match="dist_agree_range"):
        n.gate("second batch")                       # a second scan while an overflow waits for the agreement
    # a finite first scan followed by other shapes: maxima accumulate, ONE batch is kept -- the batch of the WORST scan (ADVICE r5:
    # the first batch was kept, so an overflow coming from a later shape was calibrated on a batch that had not overflowed)
    n = _net((10.0, False, None))
    n.defer()
    assert n.gate("a") is False
    n.scans.append((20.0, False, None))
    assert n.gate("b") is False
    assert n._pending_scan == (20.0, False) and n._pending_imgs == "b"
    n.scans.append((15.0, False, None))
    assert n.gate("c") is False                      # a milder batch does not replace it
    assert n._pending_scan == (20.0, False) and n._pending_imgs == "b"
    n.scans.append((1.0, True, 4))
    assert n.gate("d") is False                      # the first non-finite scan does, whatever its finite maximum
    assert n._pending_scan == (20.0, True) and n._pending_imgs == "d"


def test_after_the_agreement_an_overflow_raises_and_a_fitting_batch_passes(world2):
    n = _net((5.0, False, None))
    n._dist_agreed = True
    assert n.gate() is False
    n.scans.append((1e9, True, 2))
    with pytest.raises(FloatingPointError, match="after the ranks agreed"):
        n.gate()
