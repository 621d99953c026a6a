"""cedr::mpi's RCCL communicator class with MORE THAN ONE rank on the one-GPU box: ranks as threads, the
nccl* entry points served by a test double (tests/cpp/fake_rccl.cpp) that checks datatypes, counts, peers
and the pairing of every group's sends and receives. RCCL itself refuses two ranks on a device, so this is
as close as the class's N > 1 code gets to running here; the double completes every call before returning
and says nothing about RCCL's asynchronous execution."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_rccl_comm_class_over_the_double(gpu):
    dbl = os.path.join(HERE, "cpp", "libfake_rccl.so")
    assert os.path.exists(dbl), "tests/cpp/libfake_rccl.so is not built (__graft_entry__.build())"
    env = dict(os.environ, CEDR_HIP_RCCL_LIB=dbl)
    for k in ("CEDR_HIP_EXCHANGE",):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(HERE, "rccl_double_worker.py")], env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    st = json.loads(res.stdout.strip().splitlines()[-1])
    assert st["errors"] == 0 and st["allgather"] > 0 and st["allreduce"] > 0 and st["sends"] > 0


def test_named_rccl_library_that_does_not_load_is_an_error(gpu):
    """CEDR_HIP_RCCL_LIB names a library that cannot be loaded: no silent fallback to another librccl."""
    code = ("import compose_amd as ca\n"
            "try:\n"
            "    ca.Parallel.rccl_unique_id()\n"
            "except ca.CedrError as e:\n"
            "    print('refused')\n")
    env = dict(os.environ, CEDR_HIP_RCCL_LIB="/nonexistent/librccl.so")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(HERE))
    assert res.returncode == 0 and "refused" in res.stdout, res.stdout + res.stderr
