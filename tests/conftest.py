import sys
from pathlib import Path

import pytest

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
