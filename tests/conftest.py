import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a stuck test (a device call that never returns) must fail with a stack dump, not hold the whole run: pytest-timeout is
    # part of the image. The longest test (the 30 M-cell shard) takes about a minute — but the first process on a fresh box
    # pages the image in while it runs, and the first test that initialises torch's device runtime has been seen to stand for
    # more than ten minutes there (three of ~25 first runs; never in a second process on the same box): the limit is wide.
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 2400.0
        config.option.timeout_method = "thread"  # a call stuck inside the library never returns to the interpreter: only a watchdog thread can end it
    # `kill -USR1 <pid>` (or `timeout -s USR1 ...`) dumps every thread's Python stack, also while the main thread sits in a C call
    import faulthandler
    import signal

    if hasattr(signal, "SIGUSR1"):
        # into a file of its own: pytest's capture holds stderr
        config._scanrs_fault_file = open(os.environ.get("SCANRS_FAULT_LOG", "/tmp/scanrs_pytest_stacks.log"), "w")
        faulthandler.register(signal.SIGUSR1, file=config._scanrs_fault_file, all_threads=True, chain=False)


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")) as f:
        return json.load(f)
