import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Outputs of the compiled reference (tests/golden/make_ns_golden.py)."""
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "ns_golden.npz")))


@pytest.fixture(scope="session")
def built_lib():
    """Path of the in-tree HIP library, building it if it is absent."""
    from audiosignalprocess_amd.build import LIB, build_library

    if not os.path.exists(LIB):
        build_library()
    return LIB


def check_free_running(rel, what=""):
    """SURVEY 8(c) item (3) for a free-running comparison with the reference: per-stream relative
    L2 <= 1e-4 at the median and for all but at most one stream in eight, and never beyond 4e-4.
    The suppressor is chaotic at the 1e-4 level against ITSELF: the reference rebuilt with
    -O3 -ffp-contract=fast differs from its -O2 build by up to 3.6e-4 per stream (SURVEY 0.4,
    branch flips in the quantile tracker and the histogram thresholds), so a reduction order other
    than the reference's sequential one shows the same rare events (stream 6 of the eight-stream
    fixture flips at frame 844 under the 32-lane and the 64-lane pair orders: 1.4e-4 over the run)."""
    rel = np.asarray(rel, np.float64)
    assert np.isfinite(rel).all(), (what, rel)
    assert np.median(rel) <= 1e-5, (what, rel)
    assert (rel > 1e-4).sum() <= max(1, rel.size // 8), (what, rel)
    assert rel.max() <= 4e-4, (what, rel)


def state_from_bytes(buf):
    from audiosignalprocess_amd._abi import AspNsState

    return AspNsState.from_buffer_copy(bytes(buf))


# fields of AspNsState that exist on the device only in part (see ns_layout.h)
LIVE = {"analyzeBuf": slice(160, 256), "dataBuf": slice(160, 256), "syntBuf": slice(0, 96)}
SKIP_FIELDS = {"speechProb"}


def state_diff(a, b, skip=SKIP_FIELDS):
    """{field: (n_mismatch, max_abs_diff)} over the live parts of two AspNsState."""
    from audiosignalprocess_amd._abi import STATE_FIELDS

    da, db = a.to_dict(), b.to_dict()
    bad = {}
    for k in STATE_FIELDS:
        if k in skip:
            continue
        x, y = da[k], db[k]
        if k in LIVE:
            x, y = x[LIVE[k]], y[LIVE[k]]
        if not np.array_equal(x, y):
            d = np.abs(x.astype(np.float64) - y.astype(np.float64))
            bad[k] = (int(np.sum(x != y)), float(d.max()))
    return bad


def rel_l2_per_stream(a, b):
    """Per-stream relative L2 distance of [F][S][160] outputs (b is the truth)."""
    num = np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(axis=(0, 2)))
    den = np.sqrt((b.astype(np.float64) ** 2).sum(axis=(0, 2)))
    return num / den
