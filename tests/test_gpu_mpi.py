"""QLT, CAAS and the BFB reducer over a REAL MPI communicator: mpiexec starts 2 and 3 processes that
share GPU 0, cedr::mpi::make_parallel(MPI_COMM_WORLD) (cedr_mpi_adapter.hpp, host staged: the
image's MPICH is not GPU aware) carries the collectives, and every rank compares its cells with
a single-rank run of the same problem (tests/cpp/cedr_test_mpi.cpp). The reference's
counterpart is `mpirun -np 8 cedr_test -t` (cedr/CMakeLists.txt:14-16)."""
import os
import subprocess

import pytest

from test_cpu_mpi_adapter import MPIEXEC, ROOT, mpi_env

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.path.exists(MPIEXEC), reason="no MPI in this image")
@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_cdrs_over_mpi(gpu, tmp_path, nranks):
    exe = os.path.join(ROOT, "tests", "cpp", "cedr_test_mpi")
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe), "cedr_test_mpi"])
    res = subprocess.run([MPIEXEC, "-np", str(nranks), exe], capture_output=True, text=True, timeout=600,
                         env=mpi_env(tmp_path))
    assert res.returncode == 0 and "CEDR_MPI_OK %d ranks" % nranks in res.stdout, \
        res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.skipif(not os.path.exists(MPIEXEC), reason="no MPI in this image")
@pytest.mark.parametrize("nranks", [2, 4])
def test_cdrs_over_peer_flags_set_up_through_mpi(gpu, tmp_path, nranks):
    """The same C++ caller with `--peer-flags`: MPI only gathers the 96-byte mailbox handles; QLT, CAAS
    and the BFB reducer then run over cedr::mpi::make_parallel_peer_flags, bitwise against the single-rank run -- partial
    trees included (per-level point-to-point rounds over the mailboxes)."""
    exe = os.path.join(ROOT, "tests", "cpp", "cedr_test_mpi")
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe), "cedr_test_mpi"])
    res = subprocess.run([MPIEXEC, "-np", str(nranks), exe, "--peer-flags"], capture_output=True, text=True, timeout=600,
                         env=mpi_env(tmp_path))
    assert res.returncode == 0 and "CEDR_MPI_OK %d ranks" % nranks in res.stdout, \
        res.stdout[-2000:] + res.stderr[-2000:]
