"""The C-ABI library loads without a GPU and exports every symbol include/cedr_hip.h
declares; compute entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "cedr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(cedr_[a-zA-Z0-9_]+)\s*\(", src))
    names -= {n for n in names if n.endswith("_fn")}  # callback typedefs
    return sorted(names)


def test_all_declared_symbols_are_exported(ca):
    lib = ctypes.CDLL(ca.LIB_PATH)
    names = declared_functions()
    assert len(names) > 80
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_covers_header(ca):
    names = set(declared_functions())
    bound = set(ca._SIGS)
    assert bound <= names, sorted(bound - names)


def test_no_cpu_fallback(ca):
    if ca.device_count() > 0:
        pytest.skip("a GPU is present")
    q = ca.QLT(ca.Parallel.self_(), 8, ca.Tree.make_1d(8))
    q.declare_tracer(7, 0)
    q.end_tracer_declarations()
    with pytest.raises(ca.CedrError):
        q.finish_setup()  # needs device memory: must raise, never fall back
    c = ca.CAAS(ca.Parallel.self_(), 8)
    c.declare_tracer(7, 0)
    c.end_tracer_declarations()
    with pytest.raises(ca.CedrError):
        c.finish_setup()


def test_library_carries_the_sha_of_its_kernel_source():
    """cedr_hip_kernels_sha256() is stamped in at build time (csrc/Makefile): it must be the sha256 of the cedr_kernels.hip in the
    tree, i.e. the library is not stale; bench.py and every profiles/r05_* file quote it, and bench.py refuses a PMC traffic
    file made with other kernel code (a kernel change therefore needs tools/final_profiles.sh run again before
    roofline.traffic is reported again)."""
    import hashlib
    import compose_amd as ca
    src = os.path.join(os.path.dirname(os.path.abspath(ca.__file__)), "csrc", "cedr_kernels.hip")
    assert ca.kernels_sha256() == hashlib.sha256(open(src, "rb").read()).hexdigest()


def test_rehearsal_communicator_surface():
    """make_parallel_rehearsal (cedr_mpi.hpp): rank 0 of P over a communicator of ONE real rank; anything else is refused."""
    import compose_amd as ca
    p = ca.Parallel.rehearsal(ca.Parallel.self_(), 8)
    assert p.size() == 8 and p.rank() == 0
    with pytest.raises(ca.CedrError, match="one rank"):
        ca.Parallel.rehearsal(ca.Parallel.callbacks(0, 2, lambda *a: 0, lambda *a: 0), 4)
    with pytest.raises(ca.CedrError):
        ca.Parallel.rehearsal(ca.Parallel.self_(), 0)
    # a QLT over it plans as rank 0 of 8: one eighth of the cells, boundary exchange + apex tile
    q = ca.QLT(p, 8000, ca.Tree.make_1d(8000, 8, False, False))
    assert q.nlclcells() == 1000 and q.plan()["nb_max"] >= 1
