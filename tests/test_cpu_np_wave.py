"""The straight-line node problem the root->leaves kernels run (cedr_dev::node_problem_mass_wave,
compose_amd/csrc/cedr_node_math.hpp) against the branchy form that mirrors the reference's control flow
(cedr_qlt_inl.hpp:120-173, cedr_local_inl.hpp:69-165) and is compared bitwise with the oracle on the
GPU: compiled for the HOST, 8M problems of every kind, bit for bit. No GPU needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_straight_line_node_problem_equals_the_branchy_one():
    d = os.path.join(ROOT, "tests", "cpp")
    subprocess.check_call(["make", "-s", "-C", d, "np_wave_check"])
    res = subprocess.run([os.path.join(d, "np_wave_check"), "4000000"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert " 0 mismatches" in res.stdout
