"""The host mirrors (g2o_frontend_amd/api.py, g2o_frontend_amd/host/pwn_hip.hpp) carry the public accessor names of the reference classes they stand
for, so that configuration and caller code written against pwn_core reads the same here (same names, argument meaning).  The names
are taken from the `inline` members of the reference headers where they lie under /root/reference (studied as text; skipped on the GPU box)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/g2o_frontend/pwn_core"
TRK = "/root/reference/g2o_frontend/pwn_tracker"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference headers under /root/reference")

CLASSES = [("Aligner", "aligner.h"), ("CorrespondenceFinder", "correspondencefinder.h"), ("Linearizer", "linearizer.h"),
           ("DepthImageConverterIntegralImage", "depthimageconverter.h"), ("StatsCalculatorIntegralImage", "statscalculatorintegralimage.h"),
           ("PointInformationMatrixCalculator", "informationmatrixcalculator.h"), ("PinholePointProjector", "pinholepointprojector.h"),
           ("PinholePointProjector", "pointprojector.h"), ("Cloud", "cloud.h"), ("Merger", "merger.h"), ("VoxelCalculator", "voxelcalculator.h"),
           ("PwnMatcherBase", "pwn_matcher_base.h"), ("PwnTracker", "pwn_tracker.h")]       # the callers of the path (pwn_tracker/): SURVEY.md 8(f)
# members whose reference form returns an internal container the device path has no host copy of (documented per class in the mirrors)
NOT_MIRRORED = {"StatsCalculatorIntegralImage": {"integralImage", "intervalImage"},      # the converter's interval image: DepthImageConverter.intervalImage(); integral planes: StatsCalculatorIntegralImage.integralImage(cloud, indexImage)
                "PinholePointProjector": {"project", "unProject", "projectInterval"}}      # single-point forms: projectPoint / unProjectPixel / projectInterval in Python (no overloading); same names in C++


def _inline_names(header):
    text = open(os.path.join(TRK if header.startswith("pwn_") else REF, header)).read()
    names = set()
    for m in re.finditer(r"inline\s+[^;{(]*?[\s&*]([A-Za-z]\w*)\s*\(", text):
        n = m.group(1)
        if not n.startswith("_") and n != "operator":
            names.add(n)
    return names


@pytest.mark.parametrize("cls,header", CLASSES)
def test_python_mirror_has_the_reference_accessors(cls, header):
    from g2o_frontend_amd import api
    c = getattr(api, cls)
    missing = [n for n in sorted(_inline_names(header)) if not hasattr(c, n) and n not in NOT_MIRRORED.get(cls, set())]
    assert not missing, (cls, header, missing)


@pytest.mark.parametrize("cls,header", CLASSES)
def test_cpp_mirror_has_the_reference_accessors(cls, header):
    text = open(os.path.join(ROOT, "g2o_frontend_amd", "host", "pwn_hip.hpp")).read()
    skip = NOT_MIRRORED.get(cls, set()) - {"project", "unProject", "projectInterval"}
    missing = [n for n in sorted(_inline_names(header)) if not re.search(r"\b%s\s*\(" % re.escape(n), text) and n not in skip]
    assert not missing, (cls, header, missing)
