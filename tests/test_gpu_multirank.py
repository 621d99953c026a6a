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
    assert common.rel_per_tracer(got, want).max() <= 1e-13


def test_rccl_point_to_point_group(gpu):
    """The level-mode transport on RCCL: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd. One
    device takes one rank, so the round is a message to self; it still runs the real entry points."""
    ca = gpu
    par = ca.Parallel.rccl(ca.Parallel.rccl_unique_id(), 0, 1)
    a = np.arange(1000, dtype=np.float64) * 0.5
    src, dst = ca.DeviceArray.from_numpy(a), ca.DeviceArray.from_numpy(np.zeros(1000))
    assert par.exchange([(0, src.ptr, 1000, True), (0, dst.ptr, 1000, False)]) == 0
    ca.synchronize()
    assert np.array_equal(dst.numpy(), a)
    assert par.exchange([]) == 0


def _bench(args, timeout=600, env=None):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT, env=dict(os.environ, **(env or {})))
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


SMALL = ["--steps", "2", "--warmup", "1", "--ncells", "30000", "--ntracers", "6", "--same-gpu", "--check",
         "--no-cpu-baseline"]


@pytest.mark.parametrize("nranks,extra", [(2, []), (3, []), (2, ["--serial"])])
def test_bench_self_spawned_ranks_same_gpu(gpu, nranks, extra):
    """`python bench.py --gpus N` exactly as typed: bench.py starts its own rank processes. N > 1 runs
    the two CDRs on two streams by default; --serial puts them back to back on one."""
    res, d = _bench(["--gpus", str(nranks), "--comm", "gloo"] + SMALL + extra)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert d["n_gpus"] == nranks and d["check_ok"] is True and d["scaling"] == "strong"
    assert d["config"]["streams"] == (1 if extra else 2)
    assert d["config"]["rccl_nranks"] == 0 and d["config"]["transport"].startswith("gloo-host-staged")


@pytest.mark.parametrize("extra,streams", [([], 2), (["--serial"], 1), (["--overlap", "down"], 2),
                                           (["--overlap", "mid"], 2)])
def test_bench_one_gpu_stream_modes(gpu, extra, streams):
    """One GPU: QLT::run and CAAS::run on two streams by default (both start together; or CAAS waits for the event
    QLT::run records before its first-stage root->leaves sweep / after its first-stage leaves->root sweep), or back
    to back on one stream. Every mode passes the result checks."""
    res, d = _bench(["--steps", "3", "--warmup", "1", "--ncells", "40000", "--ntracers", "6", "--check",
                     "--no-cpu-baseline"] + extra)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert d["n_gpus"] == 1 and d["check_ok"] is True and d["config"]["streams"] == streams
    assert d["collective"] is None  # (the single-rank communicator's collectives are copies: not reported)


def test_bench_stdout_is_the_json_line_and_a_stalled_two_stream_run_still_reports(gpu):
    """N > 1: the standard output is exactly ONE line, the JSON (gloo and others print to file descriptor 1 too: it is
    pointed at stderr), and it carries the same steps' one-stream time next to the two-stream value. A two-stream run that
    stalls over the transport (test hook) ends, by every rank's watchdog, with the one-stream line measured before it --
    flagged as such -- and exit code 0 instead of hanging until the caller's timeout."""
    res, d = _bench(["--gpus", "2", "--comm", "gloo"] + SMALL)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert len(res.stdout.strip().splitlines()) == 1 and d["config"]["streams"] == 2 and "note" not in d
    assert d["serial"]["ms_per_step"] > 0 and d["roofline"] is not None
    res, d = _bench(["--gpus", "2", "--comm", "gloo"] + SMALL, timeout=120,
                    env={"CEDR_BENCH_SIMULATE_STALL": "1", "CEDR_BENCH_WATCHDOG_S": "6"})
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert len(res.stdout.strip().splitlines()) == 1
    assert d["n_gpus"] == 2 and d["config"]["streams"] == 1 and "did not complete" in d["note"] and d["value"] > 0


def test_bench_under_torch_distributed_run(gpu):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29502", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--comm", "gloo"] + SMALL
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["check_ok"] is True


def test_bench_rccl_failure_is_a_failed_run(gpu):
    """Two ranks on ONE GPU make ncclCommInitRank fail (duplicate device): every rank must
    notice, no JSON line with a value may appear, and the exit code must be non-zero. Only with
    --allow-host-fallback does the launcher start FRESH ranks on the host-staged transport,
    and the line then says so."""
    res, d = _bench(["--gpus", "2", "--comm", "rccl"] + SMALL)
    assert res.returncode != 0 and d is None, res.stdout[-1500:]
    assert "RCCL transport failed" in res.stderr
    res, d = _bench(["--gpus", "2", "--comm", "rccl", "--allow-host-fallback"] + SMALL)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert d["check_ok"] is True and d["config"]["rccl_nranks"] == 0
    assert d["config"]["transport"].startswith("gloo-host-staged")


@pytest.mark.parametrize("extra,streams", [([], 2), (["--serial"], 1)])
@pytest.mark.parametrize("bfb", [[], ["--caas-bfb", "4"]])
def test_bench_force_rccl_rank_0_of_8(gpu, extra, streams, bfb):
    """Real librccl in the loop on one GPU: bench.py --force-rccl --pretend-ranks 8 runs the step as rank 0 of 8 over an
    RCCL communicator of one rank -- QLT::run's ncclAllGather + apex and CAAS::run's ncclAllReduce (or the BFB
    reducer's all-gather) are really enqueued, on two streams through the library's cross-stream ordering events
    (reference cedr_qlt.cpp:327-337,432-439, cedr_caas.cpp:203-209 are the MPI calls they replace). The results pass the
    checks, and the line says what it is."""
    res, d = _bench(["--steps", "3", "--warmup", "1", "--ncells", "240000", "--ntracers", "6", "--check",
                     "--no-cpu-baseline", "--force-rccl", "--pretend-ranks", "8"] + extra + bfb)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-1500:]
    assert d["n_gpus"] == 1 and d["check_ok"] is True and d["config"]["streams"] == streams
    assert d["config"]["transport"].startswith("rccl communicator of ONE rank, rehearsing rank 0 of 8")
    assert d["config"]["cells_per_gpu"] == 30000 and "rank 0 of 8" in d["config"]["rehearsal"]
    assert abs(d["value"] - 2 * 30000 * 6 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["kernels_sha256"] == gpu.kernels_sha256() and len(d["kernels_sha256"]) == 64
    assert 0 < d["qlt"]["frac"] < 1 and 0 < d["caas"]["frac"] < 1 and d["placement"]["tries_env"] == "4"
    assert 0 < d["collective"]["allreduce_us"] < 1000 and 0 < d["collective"]["allgather_us"] < 1000  # one real RCCL call each
    if bfb:
        assert d["caas"]["fused_with_bfb_reducer"] is True and d["caas"]["bytes_per_cell_tracer"] == 64.0


def test_rehearsal_over_rccl_matches_the_stub_bitwise(gpu):
    """QLT as rank 0 of 8 over the rehearsal communicator: the same bits whether its inner communicator is the
    single-rank one (plain copies) or an RCCL communicator of one rank, on one stream and with the CAAS of the same
    communicator running on a second stream (the collectives of one communicator are then ordered by events)."""
    import torch
    ca = gpu
    n, nt, P = 160000, 5, 8
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    rhom, Qm, lo, hi, prev = bench.gen_inputs(torch, n // P, nt, 99, dev)
    outs = []
    for inner_kind in ("self", "rccl"):
        for two_streams in (False, True):
            inner = ca.Parallel.self_() if inner_kind == "self" else ca.Parallel.rccl(ca.Parallel.rccl_unique_id(), 0, 1)
            par = ca.Parallel.rehearsal(inner, P)
            assert par.size() == P and par.rank() == 0
            q = ca.QLT(par, n, ca.Tree.make_1d(n, P, False, False))
            c = ca.CAAS(par, n // P)
            for x in (q, c):
                for _ in range(nt):
                    x.declare_tracer(7, 0)
                x.end_tracer_declarations()
                x.finish_setup()
            assert q.nlclcells() == n // P
            s = [torch.cuda.Stream(device=dev) for _ in range(2)] if two_streams else None
            if s:
                q.set_stream(s[0].cuda_stream)
                c.set_stream(s[1].cuda_stream)
            torch.cuda.synchronize()
            q.set_rhom(rhom); q.set_Qm(Qm, lo, hi, prev)
            c.set_rhom(rhom); c.set_Qm(Qm, lo, hi, prev); c.set_Qm_source(Qm)
            for _ in range(3):
                q.run()
                c.run()
            oq, oc = torch.empty_like(Qm), torch.empty_like(Qm)
            q.get_Qm(oq); c.get_Qm(oc)
            torch.cuda.synchronize()
            outs.append((oq, oc))
            assert bool(((oq >= lo) & (oq <= hi)).all()) and bool(((oc >= lo) & (oc <= hi)).all())
            del q, c
    for oq, oc in outs[1:]:
        assert torch.equal(oq, outs[0][0]) and torch.equal(oc, outs[0][1])
