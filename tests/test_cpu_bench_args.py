"""bench.py's argument rules that need no GPU: the one-GPU rehearsal flags are refused at N > 1 before anything is spawned."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=120)


def test_rehearsal_flags_need_one_gpu():
    for extra in (["--force-rccl"], ["--pretend-ranks", "8"]):
        r = _run(["--gpus", "2"] + extra)
        assert r.returncode == 2 and "rehearsals on ONE GPU" in r.stderr, r.stderr[-500:]
    r = _run(["--pretend-ranks", "0"])
    assert r.returncode == 2 and "at least 1" in r.stderr


def test_check_needs_cst_types():
    r = _run(["--check", "--types", "mix"])
    assert r.returncode == 2 and "cst" in r.stderr


def test_two_one_rank_communicators_at_once_are_refused():
    r = _run(["--force-rccl", "--comm", "peer-flags"])
    assert r.returncode == 2 and "two different one-rank communicators" in r.stderr


def test_peer_flags_python_surface_checks_the_handles():
    """Parallel.peer_flags refuses a handle list of the wrong length before it calls into the library."""
    sys.path.insert(0, ROOT)
    import pytest
    import compose_amd as ca
    assert ca.Parallel.PEER_HANDLE_BYTES == 96
    with pytest.raises(ca.CedrError):
        ca.Parallel.peer_flags(None, [b"x" * 96], 0, 2)
