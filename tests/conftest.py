import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracleload
    return oracleload.Oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference (oracle/_ref). Built here from /root/reference when missing; on the GPU box the
    prebuilt library travels with the tree."""
    import subprocess
    import refload
    if not os.path.exists(refload.REF_SO):
        if os.path.exists("/root/reference/src/relateHMM.cpp"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref/librelate_ref.so not present and /root/reference not available")
    return refload.Ref()


@pytest.fixture(scope="session")
def gpu():
    """Session-wide Relate handle on cuda:0 -- fails loudly if the HIP library or the device is missing."""
    import relate_amd
    return relate_amd.Relate(0)
