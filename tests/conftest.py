import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # `kill -USR1 <pid>` (or `timeout -s USR1 ...`) dumps every thread's Python stack, also while the main thread sits in a C call.
    # The file is per process and opened when the signal handler is first needed (a GPU run), never for the CPU-only suite.
    config._scanrs_fault_file = None


GPU_TEST_TIMEOUT_S = 900.0  # the longest gpu test (the 1 M-cell fixture: 10^9 nonzeros generated on the host first) takes ~2 min on a warm box


def _fault_log(config):
    import faulthandler
    import signal

    if config._scanrs_fault_file is None and hasattr(signal, "SIGUSR1"):
        path = os.environ.get("SCANRS_FAULT_LOG", f"/tmp/scanrs_pytest_stacks.{os.getpid()}.log")
        config._scanrs_fault_file = open(path, "w")  # into a file of its own: pytest's capture holds stderr
        faulthandler.register(signal.SIGUSR1, file=config._scanrs_fault_file, all_threads=True, chain=False)


def pytest_collection_modifyitems(config, items):
    # A stuck gpu test (a device call that never returns) must fail with a stack dump, not hold the whole run: pytest-timeout
    # is part of the image. Only gpu-marked tests get the wide limit (a call stuck inside the library never returns to the
    # interpreter, so the watchdog has to be a thread); the library's own waits are bounded since round 4 ("sync_timeout_s"),
    # so this is the second line of defence. CPU tests keep whatever the command line says.
    has_timeout = config.pluginmanager.hasplugin("timeout")
    any_gpu = False
    for it in items:
        if it.get_closest_marker("gpu") is None:
            continue
        any_gpu = True
        if has_timeout and it.get_closest_marker("timeout") is None and not getattr(config.option, "timeout", None):
            it.add_marker(pytest.mark.timeout(GPU_TEST_TIMEOUT_S, method="thread"))
    if any_gpu and "not gpu" not in (config.getoption("markexpr", "") or ""):
        _fault_log(config)


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")) as f:
        return json.load(f)
