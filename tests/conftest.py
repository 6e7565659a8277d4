import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The sparse head's row capacity is pinned to "every site" for the suite: the bit-for-bit comparisons (eager == capture == replay, operand path ==
# stored form, rank == rank) hold for ONE capacity setting -- a capacity sizes the persistent grids, and with them the order in which partial sums
# over the live rows meet. The product default ('auto': capacities follow the workload) is covered where it is set explicitly:
# test_gpu_model.py::test_auto_sparse_capacity_follows_the_workload, test_gpu_determinism.py::test_train_step_is_bit_reproducible[...auto],
# test_gpu_fullsize.py::test_train_step_512_fp32_matches_oracle, and by smoke() / bench.py, which run the default.
os.environ.setdefault('MAGGIE_SPARSE_CAPACITY', '1.0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


# Oracle / golden parity files run FIRST, graph and multi-process infrastructure LAST: under `pytest -x` a late infrastructure failure can then
# never hide a parity result (round 3: one flaky graph test in an alphabetically early file kept 148 parity tests from running).
_FILE_ORDER = ['test_gpu_model.py', 'test_gpu_kernels.py', 'test_gpu_conv.py', 'test_gpu_determinism.py', 'test_gpu_fullsize.py',
               'test_region_oracle.py', 'test_oracle_golden.py', 'test_host_cpu.py', 'test_gpu_graphs.py']


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_FILE_ORDER)}

    def key(item):
        return rank.get(os.path.basename(str(item.fspath)), len(_FILE_ORDER) - 1)

    items.sort(key=key)                     # stable: the order inside a file is kept


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(autouse=True)
def _quiesce_between_tests():
    """Models of a finished test (and the hipGraphs they captured) die HERE, with the device idle -- not whenever the cyclic collector happens to run
    inside a later test's forward pass. Twice in ~20 full-suite runs of round 6 the suite died with a fatal fault inside
    test_parked_slab_reductions_leave_the_step_unchanged[video] (a forward of a freshly built model, four test files into the run); the test alone (24 x),
    its file (10 x) and its file behind test_gpu_conv.py (7 x) never did. What the collector destroys mid-step depends on everything allocated before --
    this fixture takes that dependence out."""
    yield
    torch = sys.modules.get('torch')
    if torch is not None and torch.cuda.is_available():
        import gc
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
