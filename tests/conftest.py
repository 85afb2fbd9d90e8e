import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def port():
    """The plain-C restatement (oracle/msm_oracle.c), built on demand."""
    from oracle import port as p

    p.build()
    p.lib()
    return p


@pytest.fixture(scope="session")
def ref():
    """The reference itself (oracle/_ref/libff_ref.so); skip where it cannot exist."""
    from oracle import ref as r

    if not r.available():
        if os.path.isdir("/root/reference/libff"):
            import subprocess

            subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")])
        if not r.available():
            pytest.skip("oracle/_ref/libff_ref.so not built (reference not mounted here)")
    r.lib()
    return r


@pytest.fixture(scope="session")
def engine():
    """The HIP engine on cuda:0 -- no fallback: the test fails if it cannot be created."""
    import libff_amd

    return libff_amd.Engine(0)
