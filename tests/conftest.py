import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Collection order of the GPU suite (the driver runs `pytest tests/ -x -q -m gpu`): op-level parity first, then the tests
# shaped like what bench.py runs (configs[1] at full size, depth 12 / 6), then the repeat tests, then everything else in
# file order -- a stop at the first failure then costs the least evidence.
_EARLY = [
    "test_gpu_parity.py::test_logmel", "test_gpu_parity.py::test_stack_layout", "test_gpu_parity.py::test_encoder",
    "test_gpu_parity.py::test_predictor_and_joint", "test_gpu_parity.py::test_offline_transcribe",
    "test_gpu_parity.py::test_streaming_matches_reference",
    "test_gpu_round3.py::test_config1_64_rows_depth_12",
    "test_gpu_round2.py::test_config1_all_64_streams",
    "test_gpu_soak.py::",
]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        nid = item.nodeid.split("/")[-1]
        for i, pre in enumerate(_EARLY):
            if nid.startswith(pre):
                return i
        return len(_EARLY)
    items.sort(key=rank)          # stable: file order within a rank
