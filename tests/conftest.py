"""pytest configuration: registers the `gpu` marker and puts the repo's python helpers on sys.path.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol export (no GPU needed).
`-m gpu`:       parity tests proper, through the C-ABI on a real MI355X.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "performance-test_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """ZZZ_TEST_ORDER=reverse | shuffle:<seed>: run the collected tests in another order (soak runs: faults that depend on
    what was allocated and freed before -- an out-of-bounds read that only sometimes leaves a mapping -- show up this way)."""
    order = os.environ.get("ZZZ_TEST_ORDER", "")
    if order == "reverse":
        items.reverse()
    elif order.startswith("shuffle:"):
        import random

        random.Random(int(order.split(":", 1)[1])).shuffle(items)
