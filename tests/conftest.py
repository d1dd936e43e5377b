import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


# (No "torch first" fixture any more: simple_spectral_amd/_capi.py puts ONE HIP runtime into the process whatever the import order,
# and ssx_create refuses to run with two -- tests/test_host_and_abi.py::test_one_hip_runtime_whatever_the_import_order.)


@pytest.fixture(scope="session", autouse=True)
def _own_jit_cache(tmp_path_factory):
    """Every pytest session gets its own disk cache for the run-time compiled kernels (csrc/ssx_jit.h: $SSX_CACHE_DIR, else ~/.cache/ssx).
    With the shared default a suite's SECOND run on a box differs from its first: test_pass1_compiled_at_upload_for_any_topology stores
    the code object of "the Cornell box with one corner moved", and the next process's test_pass1_variants_..._generic_otherwise, which
    uploads that same pattern expecting the generic kernel, starts specialised from the disk hit and fails -- the one red of
    profiles/r05/parity_per_kernel_variant.log, reproduced 9 times out of 10 in profiles/r06/formal_repeat_before.log (the first run of a
    box is green).  Child processes inherit the variable; tests that manage a cache directory of their own set it themselves."""
    old = os.environ.get("SSX_CACHE_DIR")
    os.environ["SSX_CACHE_DIR"] = str(tmp_path_factory.mktemp("ssx_jit_cache"))
    yield
    if old is None:
        os.environ.pop("SSX_CACHE_DIR", None)
    else:
        os.environ["SSX_CACHE_DIR"] = old
