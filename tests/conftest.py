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
    # part of the image; the longest test (the 30 M-cell shard) takes about a minute
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900.0
        config.option.timeout_method = "thread"  # a call stuck inside the library never returns to the interpreter: only a watchdog thread can end it
    # `kill -USR1 <pid>` (or `timeout -s USR1 ...`) dumps every thread's Python stack, also while the main thread sits in a C call
    import faulthandler
    import signal

    if hasattr(signal, "SIGUSR1"):
        faulthandler.register(signal.SIGUSR1, all_threads=True, chain=False)


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")) as f:
        return json.load(f)
