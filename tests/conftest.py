import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_device_temporaries():
    yield
    import util
    util.release()


# ---- parity summary (VERDICT r5 #2 / weak #9e): every test prints its measured numbers as `[parity] ...` lines, which `pytest -q`
# swallows.  They are collected here from the captured stdout of each test and (1) printed once in the terminal summary -- so the
# driver's own `pytest -m gpu -q` run leaves the measured rel_max / rel_rms / agreement values in its log -- and (2) written as JSON to
# gpurun_out/parity_summary.json (scratch; a copy of the builder's last full pass is committed as profiles/r6_parity_summary.json).
_PARITY = {}


def pytest_runtest_logreport(report):
    if report.when != "call":
        return
    lines = [ln[len("[parity] "):] for ln in (report.capstdout or "").splitlines() if ln.startswith("[parity] ")]
    if lines:
        _PARITY[report.nodeid] = {"outcome": report.outcome, "lines": lines}


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _PARITY:
        return
    import json
    terminalreporter.section("parity numbers measured in this run ([parity] lines of the tests)")
    for nodeid, rec in _PARITY.items():
        terminalreporter.write_line(f"{nodeid} [{rec['outcome']}]")
        for ln in rec["lines"]:
            terminalreporter.write_line("    " + ln)
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_summary.json"), "w") as f:
            json.dump({"exitstatus": int(exitstatus), "tests": _PARITY}, f, indent=1)
    except OSError:
        pass
