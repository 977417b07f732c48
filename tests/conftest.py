import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "computational-chemistry-ai_amd", "python"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def free_port():
    """A TCP port the kernel just handed out (for torch.distributed rendezvous on 127.0.0.1)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


MOLECULES = {
    "h2o": "O 0 0 0; H 0 -0.757 0.587; H 0 0.757 0.587",
    "h2co": "C 0 0 0; O 1.2 0 0; H -0.5 0.9 0; H -0.5 -0.9 0",  # reference README.md:187-192
    "hf": "H 0 0 0; F 0 0 1.1",
    "ch4": "C 0 0 0; H 0.629 0.629 0.629; H -0.629 -0.629 0.629; H -0.629 0.629 -0.629; H 0.629 -0.629 -0.629",
    "c2h4": "C 0 0 0.6695; C 0 0 -0.6695; H 0 0.9289 1.2321; H 0 -0.9289 1.2321; H 0 0.9289 -1.2321; H 0 -0.9289 -1.2321",
}


@pytest.fixture(scope="session")
def molecules():
    return MOLECULES
