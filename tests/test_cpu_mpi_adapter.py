"""The opt-in MPI adapter (compose_amd/csrc/cedr_mpi_adapter.hpp) compiled against the image's
MPICH and run with mpiexec on 3 ranks: reference-style calling code -- make_parallel(MPI_Comm),
`MPI_Comm c = p->comm()`, all_reduce / isend / irecv / waitall / gather(v) / all_ok / Op
(reference cedr/cedr_mpi.hpp:17-85) -- compiles unchanged and behaves as MPI does. Host only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPIEXEC = "/opt/conda/bin/mpiexec"


def mpi_env(tmp):
    """Only MPICH's own libraries on the path: conda's lib directory holds an older libstdc++."""
    d = os.path.join(str(tmp), "mpilib")
    os.makedirs(d, exist_ok=True)
    for l in ("libmpi.so.12", "libgfortran.so.4", "libquadmath.so.0"):
        dst = os.path.join(d, l)
        if not os.path.exists(dst):
            os.symlink(os.path.join("/opt/conda/lib", l), dst)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = d + (":" + env["LD_LIBRARY_PATH"] if env.get("LD_LIBRARY_PATH") else "")
    return env


@pytest.mark.skipif(not os.path.exists(MPIEXEC), reason="no MPI in this image")
def test_mpi_adapter_three_ranks(ca, tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "mpi_adapter_test"])
    res = subprocess.run([MPIEXEC, "-np", "3", os.path.join(ROOT, "tests", "cpp", "mpi_adapter_test")],
                         capture_output=True, text=True, timeout=300, env=mpi_env(tmp_path))
    assert res.returncode == 0 and "MPI_ADAPTER_OK 3 ranks" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]
