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


# default: put, wait and take as ONE launch that waits for the arrivals itself; STREAM: put launch, hipStreamWaitValue64, take launch
STREAM = {"CEDR_HIP_PEER_WAIT": "stream"}


@pytest.mark.parametrize("size,env", [(1, {}), (2, {}), (4, {}), (3, {"PEER_NCELLS": "700"}),
                                      (2, {"PEER_SLOT_DOUBLES": "64"}),
                                      (1, STREAM), (2, STREAM), (4, STREAM),
                                      (3, dict(STREAM, PEER_SLOT_DOUBLES="100", PEER_NCELLS="700"))])
def test_peer_flags_ranks_as_processes_bitwise(gpu, size, env):
    """(slots of 64 doubles: QLT's all-gather and CAAS's all-reduce run in several rounds each)"""
    for r in run_ranks(size, env):
        assert r["bad"] == [], r


def test_a_peer_that_never_arrives_ends_in_nan_not_in_a_hang(gpu):
    """The one-launch form waits in the kernel: with a bounded number of polls (here 2^14; the default is minutes) the
    launch ends on its own and says so with NaN."""
    res = run_ranks(2, {"PEER_MODE": "giveup", "CEDR_HIP_PEER_GIVE_UP_LOG2": "14"}, timeout=200)
    assert res[0]["gave_up_nan"] is True and res[0]["seconds"] < 30, res


ROOT = os.path.dirname(HERE)


def _bench(args, timeout=600, env=None):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT, env=dict(os.environ, **(env or {})))
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("nranks,extra,env", [(2, [], {}), (4, [], {}), (2, ["--serial"], {}), (2, ["--caas-bfb", "4"], {}),
                                              (2, [], STREAM), (4, ["--serial"], STREAM)])
def test_bench_over_peer_flags_ranks_on_one_gpu(gpu, nranks, extra, env):
    """`bench.py --gpus N --same-gpu --comm peer-flags`: N rank processes on GPU 0, the step's two collectives over the
    peer-flag transport, two streams by default; the results pass the checks and the line names the transport."""
    res, d = _bench(["--gpus", str(nranks), "--comm", "peer-flags", "--same-gpu", "--steps", "3", "--warmup", "1",
                     "--ncells", "48000", "--ntracers", "6", "--check", "--no-cpu-baseline"] + extra, env=env)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert d["n_gpus"] == nranks and d["check_ok"] is True
    assert d["config"]["streams"] == (1 if "--serial" in extra else 2)
    assert d["config"]["transport"].startswith("peer-flags: one communicator of %d ranks (ALL ON GPU 0" % nranks)
    assert d["config"]["rccl_nranks"] == 0
    # the line also carries what one collective of this communicator costs (and that its results were right)
    assert d["collective"]["results_ok"] is True and 0 < d["collective"]["allreduce_us"] < 5000


@pytest.mark.parametrize("env", [{}, STREAM])
@pytest.mark.parametrize("extra", [[], ["--serial"]])
def test_bench_rank_0_of_8_over_a_one_rank_peer_flag_communicator(gpu, extra, env):
    """The one-GPU rehearsal of rank 0 of 8 (bench.py --pretend-ranks 8) with the peer-flag communicator of ONE rank
    inside: the exchange launch (or put, stream wait and take) is really enqueued for every collective."""
    res, d = _bench(["--steps", "3", "--warmup", "1", "--ncells", "240000", "--ntracers", "6", "--check",
                     "--no-cpu-baseline", "--comm", "peer-flags", "--pretend-ranks", "8"] + extra, env=env)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert d["n_gpus"] == 1 and d["check_ok"] is True
    assert d["config"]["transport"].startswith("peer-flags communicator of ONE rank, rehearsing rank 0 of 8")
