import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- achieved parity margins: tests call report(name, max_abs, tolerance); the table is printed at the end of the run, so the tail of a
# `pytest -q` log (GPUTEST_rNN.json keeps only that) carries the numbers and not just dots
PARITY_REPORT = []


def report(name, value, tol):
    PARITY_REPORT.append((str(name), float(value), float(tol)))
    return float(value)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not PARITY_REPORT:
        return
    tr = terminalreporter
    tr.write_sep("=", "parity margins (max-abs vs golden / oracle; tolerance; fraction of tolerance used)")
    worst = {}
    for name, v, tol in PARITY_REPORT:
        if name not in worst or v / tol > worst[name][0] / worst[name][1]:
            worst[name] = (v, tol)
    for name, (v, tol) in worst.items():
        tr.write_line(f"  {name:<74s} {v:9.3e}  tol {tol:7.1e}  {100.0 * v / tol:5.1f} %")


GOLDEN = os.path.join(ROOT, "tests", "golden")

NARROW_CFG = dict(image_size=32, in_channels=4, out_channels=4, model_channels=32,
                  attention_resolutions=[1, 2], num_res_blocks=1, channel_mult=[1, 2],
                  num_heads=2, use_scale_shift_norm=True, resblock_updown=True, cond_model=False)
SHALLOW_CFG = dict(image_size=32, in_channels=4, out_channels=4, model_channels=32,
                   attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4],
                   num_heads=8, use_scale_shift_norm=True, resblock_updown=True, cond_model=False)
BASE_CFG = dict(image_size=32, in_channels=4, out_channels=4, model_channels=128,
                attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4],
                num_heads=8, use_scale_shift_norm=True, resblock_updown=True, cond_model=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
