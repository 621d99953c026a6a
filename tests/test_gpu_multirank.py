"""Multi-rank paths on the one-GPU box: (1) the RCCL communicator with a single rank
(dlopen, ncclCommInitRank, all-gather / all-reduce plumbing), (2) bench.py with two
processes sharing GPU 0 over the host-staged gloo transport: the N > 1 control flow
(sharded tree, subtree-root exchange, replicated top, CAAS all-reduce) end to end on
the HIP kernels, with the result properties checked."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import common
import randomized as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_single_rank(gpu, oracle):
    ca = gpu
    par = ca.Parallel.rccl(ca.Parallel.rccl_unique_id(), 0, 1)
    assert par.size() == 1 and par.rank() == 0
    trs = R.caas_tracers(R.tracers_vector())
    v = R.make_values(trs, 3000, seed=8)
    types = [t.problem_type for t in trs]
    c = ca.CAAS(par, 3000)
    for t in types:
        c.declare_tracer(t, 0)
    c.end_tracer_declarations()
    c.finish_setup()
    d = [ca.DeviceArray.from_numpy(x) for x in (v.rhom, v.Qm, v.Qm_min, v.Qm_max, v.Qm_prev)]
    c.set_rhom(d[0])
    c.set_Qm(d[1], d[2], d[3], d[4])
    c.run()  # reduce_globally goes through ncclAllReduce
    out = ca.DeviceArray(v.Qm.size)
    c.get_Qm(out)
    ca.synchronize()
    want, _ = common.run_oracle_caas(oracle, [3000], types, v)
    got = out.numpy(v.Qm.shape)
    assert np.abs(got - want).max() / np.abs(want).max() <= 1e-13


@pytest.mark.parametrize("nranks", [2, 3])
def test_bench_multiprocess_same_gpu(gpu, nranks):
    port = 29500 + nranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(nranks), "--steps", "2", "--warmup", "1", "--ncells", "30000", "--ntracers", "6",
           "--comm", "gloo", "--same-gpu", "--check", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == nranks and d["check_ok"] is True and d["scaling"] == "strong"


def test_bench_rccl_failure_falls_back_to_host_staged(gpu):
    """Two ranks on ONE GPU make ncclCommInitRank fail (duplicate device): bench.py must
    notice on every rank, switch the collectives to the host-staged transport, still
    produce correct results, and say which transport it measured."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29507", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--ncells", "30000", "--ntracers", "6",
           "--comm", "rccl", "--same-gpu", "--check", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["check_ok"] is True
    assert d["config"]["transport"].startswith("gloo-host-staged (RCCL failed")
