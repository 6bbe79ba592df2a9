import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """liboracle.so is test infrastructure; build it on demand (plain gcc, seconds)."""
    import orcbind
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return orcbind.lib()


@pytest.fixture(scope="session")
def gold():
    class G:
        dir = GOLD

        @staticmethod
        def path(name):
            return os.path.join(GOLD, name)

        @staticmethod
        def npz(name):
            return dict(np.load(os.path.join(GOLD, name)))

        @staticmethod
        def json_gz(name):
            with gzip.open(os.path.join(GOLD, name), "rt") as f:
                return json.load(f)

        @staticmethod
        def text_gz(name):
            with gzip.open(os.path.join(GOLD, name), "rb") as f:
                return f.read()

        @staticmethod
        def fastq_nt6(name):
            tab = np.full(256, 5, dtype=np.uint8)
            for ch, v in zip(b"ACGTacgt", [1, 2, 3, 4, 1, 2, 3, 4]):
                tab[ch] = v
            lines = G.text_gz(name).split(b"\n")
            return [tab[np.frombuffer(lines[i], dtype=np.uint8)] for i in range(1, len(lines) - 1, 4)]
    return G


@pytest.fixture(scope="session")
def tiny_oracle(oracle_lib, gold):
    import orcbind
    return orcbind.OrcIndex(gold.path("tiny.fmd"))


def _gpu_available():
    try:
        from fermi_amd import api
        return api.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """On a GPU box the HIP library MUST be present and working: fail loudly, never skip."""
    from fermi_amd import api
    api.lib()
    assert api.device_count() > 0, "pytest -m gpu needs a GPU; libfmdhip has no CPU fallback"
    return api
