"""The drop-in boundary against the reference's OWN caller sources: the reference's
src/DenseSLAM/InfiniTamDriver.cpp -- with InfiniTamDriver.h, Input.h, Utils.h, DepthProvider.h, Defines.h,
PreviewType.h, VoxelDecayParams.h it includes -- is put in front of the ITMLib mirror
(denseslam-global-consistency-h_amd/itmlib) with `g++ -std=c++11 -fsyntax-only`.  The file is read where it lies under
/root/reference (nothing of it is copied into this repository); the third-party headers the image lacks (OpenCV, Eigen,
Pangolin, gflags) are minimal declarations under tests/stubs/.  Every error this prints is a source-compatibility gap
of the mirror.  Skipped where /root/reference does not exist (the GPU box).

src/DenseSLAM/DenseSlam.cpp is NOT reachable this way: it is written against ORB-SLAM2-API-M -- a second un-vendored,
empty submodule (System, Tracking, MapDrawer, KeyFrame ...), not a third-party header -- so for it, DenseSLAMGUI.cpp and
SystemEntry.cpp the check is by name: every ITMLib identifier those files mention must be declared by the mirror."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/DenseSLAM"
MIRROR = os.path.join(ROOT, "denseslam-global-consistency-h_amd", "itmlib")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")


def _syntax_check(source):
    cmd = ["g++", "-std=c++11", "-fsyntax-only", os.path.join(REF, source),
           "-I", os.path.join(ROOT, "tests", "stubs", "thirdparty"),
           "-I", os.path.join(ROOT, "tests", "stubs", "refroot", "DenseSLAM"),  # "../InfiniTAM-Global-Consistency-h/..." resolves from here
           "-I", MIRROR, "-I", os.path.join(ROOT, "include")]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_reference_infinitam_driver_compiles_against_the_mirror():
    res = _syntax_check("InfiniTamDriver.cpp")
    errors = [line for line in res.stderr.splitlines() if "error" in line]
    assert res.returncode == 0 and not errors, "source-compatibility gaps of the ITMLib mirror:\n" + "\n".join(errors[:40])


def test_the_check_would_notice_a_gap(tmp_path):
    """The syntax check is not vacuous: a caller that names something the mirror does not have fails it."""
    probe = tmp_path / "probe.cpp"
    probe.write_text('#include "InfiniTamDriver.h"\n'
                     "void f(SparsetoDense::drivers::InfiniTamDriver *d) { d->ThisMethodDoesNotExist(); }\n")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", str(probe), "-I", REF,
           "-I", os.path.join(ROOT, "tests", "stubs", "thirdparty"), "-I", os.path.join(ROOT, "tests", "stubs", "refroot", "DenseSLAM"),
           "-I", MIRROR, "-I", os.path.join(ROOT, "include")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode != 0 and "ThisMethodDoesNotExist" in res.stderr


def _mirror_text():
    out = []
    for base, _dirs, files in os.walk(MIRROR):
        for f in files:
            if f.endswith(".h"):
                out.append(open(os.path.join(base, f), errors="replace").read())
    return "\n".join(out)


@pytest.mark.parametrize("source", ["DenseSlam.h", "DenseSlam.cpp", "DenseSLAMGUI.cpp", "SystemEntry.cpp", "InfiniTamDriver.h"])
def test_every_itmlib_name_the_callers_mention_exists_in_the_mirror(source):
    text = open(os.path.join(REF, source), errors="replace").read()
    text = re.sub(r"//[^\n]*", "", text)            # (commented-out calls are not calls)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(ITM[A-Za-z0-9_]+|InfiniTAM_IMAGE_[A-Z_]+|MEMORYDEVICE_[A-Z]+|SDF_[A-Z0-9_]+)\b", text))
    # methods called through the engine objects the driver owns (denseMapper->X(, trackingController->X( ...)
    names |= set(re.findall(r"\b(?:denseMapper|trackingController|viewBuilder|mapManager|visualisationEngine|swappingEngine|meshingEngine)"
                            r"\s*->\s*([A-Za-z_][A-Za-z0-9_]*)\s*\(", text))
    mirror = _mirror_text()
    missing = sorted(n for n in names if not re.search(r"\b" + re.escape(n) + r"\b", mirror))
    assert not missing, f"{source} names ITMLib symbols the mirror does not declare: {missing}"
