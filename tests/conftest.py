import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("VHR_TEST_LIB"):       # tests/test_sanitizers.py: the host-only tests on the AddressSanitizer build of the library
        from vulkanhybridrenderer_amd import lib
        lib.LIB_PATH = os.path.abspath(os.environ["VHR_TEST_LIB"])


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/libvhr_oracle.so on demand."""
    from oracle import binding
    binding.build()
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def vhr():
    """The product library binding; the .so must have been built (python -c 'import __graft_entry__ as g; g.build()')."""
    from vulkanhybridrenderer_amd import lib
    lib.load()
    return lib
