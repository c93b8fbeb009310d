import os
import sys

import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
sys.path.insert(0, os.path.dirname(_HERE))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    if "s" not in (config.option.reportchars or ""):
        config.option.reportchars = (config.option.reportchars or "") + "s"      # a skipped test prints its reason, also under -q


def _have_gpu():
    try:
        from speech_amd import _lib
        return _lib.load().speech_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a machine without a device must fail loudly, not skip: the product has no fallback.
    pass


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """how many parity checks of this session passed by the tolerance itself and how many needed the float32-noise clause"""
    try:
        import compare
    except Exception:
        return
    log = compare.PARITY_LOG
    if not log:
        return
    n1 = sum(1 for r in log if r[2] == 1)
    n2 = sum(1 for r in log if r[2] == 2)
    n0 = sum(1 for r in log if r[2] == 0)
    worst = max((r for r in log if r[2] == 2), key=lambda r: r[3], default=None)
    terminalreporter.write_line("parity checks: %d, by the tolerance (clause 1): %d, by the float32-noise clause (2): %d, failed: %d%s"
                                % (len(log), n1, n2, n0, "" if worst is None else "; largest clause-2 hip-vs-oracle %.3g (%s)" % (worst[3], worst[1])))
    out = os.path.join(os.path.dirname(_HERE), "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "parity_clauses.json"), "w") as f:
            json.dump({"checks": len(log), "clause1": n1, "clause2": n2, "failed": n0,
                       "clause2_cases": [dict(label=str(r[0]), kind=r[1], e_go=r[3], e_gt=r[4], e_ot=r[5], rms_ratio=r[6]) for r in log if r[2] == 2]}, f, indent=1)
