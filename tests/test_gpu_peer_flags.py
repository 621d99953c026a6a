"""The peer-flag transport (compose_amd/csrc/cedr_mpi.hpp: stream-ordered stores into the peers' IPC-mapped mailboxes +
one hipStreamWaitValue64 per collective; behind a knob, RCCL stays the default) with MORE THAN ONE rank on the one-GPU
box: ranks as processes (tests/peer_flags_worker.py), every rank bitwise against the single-rank oracle."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_ranks(size, env=None, timeout=420):
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "peer_flags_worker.py"), str(r), str(size), d],
                                  env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                  text=True) for r in range(size)]
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=timeout))
        finally:
            for p in procs:  # exactly the children started above
                if p.poll() is None:
                    p.kill()
        res = []
        for r, (p, (so, se)) in enumerate(zip(procs, outs)):
            assert p.returncode == 0, "rank %d: exit %s\n%s\n%s" % (r, p.returncode, so[-2000:], se[-4000:])
            res.append(json.loads(so.strip().splitlines()[-1]))
        return res


@pytest.mark.parametrize("size,env", [(1, {}), (2, {}), (4, {}), (3, {"PEER_NCELLS": "700"}),
                                      (2, {"PEER_SLOT_DOUBLES": "64"})])
def test_peer_flags_ranks_as_processes_bitwise(gpu, size, env):
    """(slots of 64 doubles: QLT's all-gather and CAAS's all-reduce run in several rounds each)"""
    for r in run_ranks(size, env):
        assert r["bad"] == [], r
