import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from g2o_frontend_amd import _lib
        return _lib.lib().pwn_hip_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- shared synthetic cases ------------------------------------------------------------------------------
# "small": 120x160, the reference's imageScale-4 configuration (pwn_core/conf/pwn_aligner_1_4.conf)
# "vga":   480x640, pwn_core/conf/pwn_aligner_1_1.conf
CASES = {
    "small": dict(rows=120, cols=160, scale=4),
    "vga": dict(rows=480, cols=640, scale=1),
    "k2": dict(rows=960, cols=1280, scale=1),      # BASELINE configs[4]: Kinect-2-scale frames
}


def case_params(name):
    from g2o_frontend_amd import synth
    from oracle import oracle as O
    c = CASES[name]
    if name == "k2":     # K = (1050,1050,639.5,479.5), stats radii x2 (min 20, max 60, minPoints 200), rest as VGA
        return c["rows"], c["cols"], synth.K_1280, dict(O.VGA_CONF_CONVERTER, min_image_radius=20, max_image_radius=60, min_points=200), dict(O.VGA_CONF_ALIGNER)
    K = synth.scaled_K(synth.K_VGA, c["scale"]) if c["scale"] != 1 else synth.K_VGA
    conv = dict(O.QVGA4_CONF_CONVERTER if c["scale"] == 4 else O.VGA_CONF_CONVERTER)
    alig = dict(O.QVGA4_CONF_ALIGNER if c["scale"] == 4 else O.VGA_CONF_ALIGNER)
    return c["rows"], c["cols"], K, conv, alig


def make_depth_pair(name, seed):
    """float32 depth images of the seeded pair + true transform (rendered at the case's resolution)."""
    from g2o_frontend_amd import synth
    from oracle import oracle as O
    rows, cols, K, _, _ = case_params(name)
    ref_mm, cur_mm, T = synth.make_pair(seed, rows, cols, K)
    return O.convert_16u_to_32f(ref_mm), O.convert_16u_to_32f(cur_mm), T, ref_mm, cur_mm


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O
