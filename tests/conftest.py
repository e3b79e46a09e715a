import sys
from pathlib import Path

import pytest

# PyTorch's wheel bundles its own ROCr/HIP runtime (SONAME libamdhip64.so, not the system's libamdhip64.so.7), so a process that
# uses both the library and torch holds two runtimes.  Seen on the MI355X boxes: if the system runtime comes up first, torch's then
# reports "No HIP GPUs are available"; the other order works.  The tests that hand torch device buffers to the library therefore
# need torch loaded before the first rs_* call whatever order the files run in (INTEGRATION.md section 2 says the same to callers).
try:
    import torch  # noqa: F401
except ImportError:      # the library itself does not need it
    pass

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def case_cache(tmp_path_factory):
    """Builds (once per session) the model/graph/wav files of a parity case."""
    from tests import cases
    built = {}

    def get(name):
        if name not in built:
            root = tmp_path_factory.mktemp(name)
            built[name] = cases.build_case_files(cases.CASES[name], root)
        return built[name]

    return get
