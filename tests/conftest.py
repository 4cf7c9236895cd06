import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): compiled on demand with gcc."""
    from oracle import oracle as orc
    orc.build()
    return orc


# Collection order of the GPU suite (the driver runs `pytest -x`): the oracle comparisons first, every test that starts
# other processes last, so that an infrastructure failure (a socket, a launcher, a compiler) can never hide the parity
# results again (round 3: EADDRINUSE in a launcher test stopped the run before tests/test_gpu_parity.py was reached).
_FILE_ORDER = ["test_gpu_parity.py", "test_gpu_fuzz_slice.py", "test_gpu_dict_stream.py", "test_gpu_gauss_seidel.py",
               "test_gpu_threads.py", "test_gpu_dist.py", "test_gpu_p2p_allreduce.py", "test_gpu_dist_multirank.py"]
_SPAWNING = ("test_c_program_through_the_abi", "test_host_owned_recurrence", "test_bench_")


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        fname = os.path.basename(str(item.fspath))
        if not fname.startswith("test_gpu_"):
            return (0, 0, 0)                               # CPU tests keep their order, in front
        rank = _FILE_ORDER.index(fname) if fname in _FILE_ORDER else len(_FILE_ORDER)
        spawns = any(s in item.name for s in _SPAWNING) or fname in ("test_gpu_dist_multirank.py", "test_gpu_p2p_allreduce.py")
        return (1, 1 if spawns else 0, rank)
    items.sort(key=key)                                    # stable: the order within a file is kept
