import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# Order of the GPU suite: the hot path first.  The driver runs `pytest -x`, so a failure in a peripheral test must
# not be able to hide the replays of the reference-dumped vectors (round 3: a race in an entropy-coder test stopped
# the run in front of all of tests/test_gpu_parity.py).  Rows of SURVEY.md 8: a-1..a-6 (KATs, fuzz, full-size
# properties of compose and effects), then the boundary (drop-in, CLI), then the rows either side (entropy coders,
# ordering, multi-device).  Tests keep their file order inside a group.
_GPU_ORDER = [
    ("tests/test_gpu_parity.py", None),
    ("tests/test_gpu_fullsize.py", lambda name: "entropy" not in name),
    ("tests/test_dropin_shim.py", lambda name: "entropy" not in name),
    ("tests/test_cli.py", None),
    ("tests/test_gpu_fullsize.py", None),
    ("tests/test_dropin_shim.py", None),
    ("tests/test_gpu_huffman.py", None),
    ("tests/test_gpu_huffdec.py", None),
    ("tests/test_gpu_ordering.py", None),
    ("tests/test_gpu_multidevice.py", None),
]


def _gpu_rank(item):
    path = item.nodeid.split("::")[0]
    name = item.nodeid.split("::", 1)[1] if "::" in item.nodeid else ""
    for rank, (file, pred) in enumerate(_GPU_ORDER):
        if path == file and (pred is None or pred(name)):
            return rank
    return len(_GPU_ORDER)


def pytest_collection_modifyitems(config, items):
    gpu = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu:
        return
    order = {id(it): i for i, it in enumerate(items)}
    gpu_sorted = sorted(gpu, key=lambda it: (_gpu_rank(it), order[id(it)]))
    rest = [it for it in items if it.get_closest_marker("gpu") is None]
    items[:] = rest + gpu_sorted
