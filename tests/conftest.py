import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Order of the -m gpu tier: the kernel-level HIP-vs-oracle parity suite first, then the solver / full-size files, and the tests
# that start other processes (C++ mirror, RCCL with one rank, several ranks on one GPU) last -- so that `pytest -x` can never
# again stop in a subprocess test before the parity suite has run ("What's weak" 2).
_ORDER = ["test_gpu_0_parity", "test_golden", "test_gpu_cgs_tfqmr", "test_gpu_fgmres", "test_gpu_fullsize", "test_gpu_fullsize_oracle",
          "test_gpu_x_cpp_mirror", "test_gpu_y_dist_single", "test_gpu_z_multirank_shim", "test_gpu_z_bench_rehearsal"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else len(_ORDER) // 2
    items.sort(key=key)          # stable: the order inside a file is kept


# LDS is not cleared between kernels, and what an earlier process left there differs from GPU box to GPU box: a kernel that reads LDS it has
# not written is right on one box and wrong on the next (round 4: tri_box.h's ring, found by a test that failed on one box in five).
# Every -m gpu test therefore starts with NaNs in every compute unit's LDS (kryst_bench_poison_lds) and in 4 GB of just-freed device memory.
import pytest

# ... and every device block that enters the library's pool of destroyed ILU preconditioners' storage is filled with 0xFF bytes (NaNs / -1) before the
# next set-up is handed it uncleared (ctx.cpp: pool_free; read once per process, so it is set before the library loads -- child processes inherit it)
os.environ.setdefault("KRYST_DEV_POOL_POISON", "1")

_poison_ctx = None


@pytest.fixture(autouse=True)
def _nans_in_lds_before_every_gpu_test(request):
    global _poison_ctx
    if request.node.get_closest_marker("gpu") is not None and os.environ.get("KRYST_TEST_POISON_LDS", "1") != "0":
        import kryst_amd as K
        if _poison_ctx is None:
            _poison_ctx = K.Context(0)
        _poison_ctx.poison_lds()
        # ... and NaNs in the device memory the test is about to be handed (hipMalloc does not clear either)
        junk = [_poison_ctx.vec(1 << 28).fill(float("nan")) for _ in range(2)]
        _poison_ctx.synchronize()
        del junk
    yield
