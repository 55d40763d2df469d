import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """Build libt2v_hip.so once per session if it is missing (hipcc cross-compiles without a GPU)."""
    from text2video_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle (stock torch convolutions) runs 2.7x SLOWER on the 128-256 threads torch takes by default on the GPU
    boxes' hosts than on 8-32 (bench.py's cpu_baseline.by_threads, rounds 3-5: 0.11 fps at 128 threads, 0.29-0.31 at 32 / 8):
    the oracle-heavy GPU tests were paying for that.  16 threads unless T2V_TEST_CPU_THREADS says otherwise; a no-op on hosts
    with fewer cores (this only changes how the checker is scheduled, not what it computes in fp64; fp32 summation order inside
    torch's kernels may differ with the thread count -- the tolerances are for "another correct fp32 implementation")."""
    import torch
    want = int(os.environ.get("T2V_TEST_CPU_THREADS", "16"))
    if want > 0 and torch.get_num_threads() > want:
        torch.set_num_threads(want)
    yield


@pytest.fixture(autouse=True)
def _seeded():
    """Every test starts from the same torch / numpy global RNG state: modules built with torch's default
    initialisers (e.g. the conv biases that vid2vid's weights_init leaves alone) are then the same in every
    process, so a run is reproducible instead of a fresh random draw."""
    import numpy as np
    import torch
    torch.manual_seed(0)
    np.random.seed(0)
    yield


@pytest.fixture
def t2v_env():
    """t2v_env(name, value): set a T2V_* switch for this test.  libt2v_hip.so reads its switches once (t2v_create), so
    every change -- and the restore at the end of the test -- is followed by t2v_reload_env(); the switches train.py reads
    from os.environ at run time see the change directly."""
    saved = {}

    def reload():
        from text2video_amd import _lib
        if os.path.exists(_lib.LIB_PATH):
            _lib.load().t2v_reload_env()

    def setenv(name, value):
        saved.setdefault(name, os.environ.get(name))
        os.environ[name] = str(value)
        reload()
    yield setenv
    for name, old in saved.items():
        if old is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = old
    if saved:
        reload()
