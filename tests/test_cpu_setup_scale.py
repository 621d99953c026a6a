"""f2: tree::analyze at scale. The array-based analyze of the product and the oracle's
analyze (the reference's algorithm, cedr_tree.cpp:55-213) give identical integers -- slot count, node table,
level lists -- for 6M cells on one rank (BASELINE config D's global size), in seconds. CPU only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_analyze_6M_cells_identical_integers():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "setup_time.py"), "6000000"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "6000000 cells" in res.stdout and "identical integers: True" in res.stdout
