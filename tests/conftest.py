import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")


def _ensure_built():
    lib = os.path.join(ROOT, "iyokan_amd", "lib")
    need = [os.path.join(lib, f) for f in ("libiyokan_hip.so", "libiyokan_client.so", "libiyk_emul.so")]
    need.append(os.path.join(ROOT, "oracle", "libiyk_oracle.so"))
    if not all(os.path.exists(f) for f in need):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session", autouse=True)
def built():
    _ensure_built()


@pytest.fixture(scope="session")
def keys128(built):
    from iyokan_amd import client
    from iyokan_amd.params import params_128bit

    return client.keygen(params_128bit(), seed=1)


@pytest.fixture(scope="session")
def keys80(built):
    from iyokan_amd import client
    from iyokan_amd.params import params_80bit

    return client.keygen(params_80bit(), seed=1)


@pytest.fixture(scope="session")
def oracle128(keys128):
    import oracle_lib

    o = oracle_lib.Oracle(keys128)
    yield o
    o.close()


@pytest.fixture(scope="session")
def oracle80(keys80):
    import oracle_lib

    o = oracle_lib.Oracle(keys80)
    yield o
    o.close()
