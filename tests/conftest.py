import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
