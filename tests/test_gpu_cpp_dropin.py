"""Builds and runs tests/cpp/cedr_test_hip: a C++ caller written like the reference's
harness, using cedr::qlt::QLT<>, cedr::caas::CAAS<>, the shared_ptr tree API and a
DeviceOp copied by value into the caller's own HIP kernels."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "cedr_test_hip")


@pytest.mark.parametrize("args", [["-nc", "1"], ["-nc", "21"], ["-nc", "777"], ["-nc", "1000", "--imbalanced"],
                                  ["-nc", "1024", "--prefer-mass-con"]])
def test_cpp_caller(gpu, args):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(EXE)])
    res = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "PASS" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
