#!/usr/bin/env python3
"""bench.py -- cell-tracer-updates/s of the CEDR limiter hot path (QLT + CAAS) on MI355X.

One "step" = one QLT::run() plus one CAAS::run() over the same synthetic
(ncells x ntracers) problem, all tracers conserve|shapepreserve|consistent, inputs
resident in HBM before the timed region. value = cell-tracer
updates by both algorithms per second = 2*ncells*ntracers / time per step.

N GPUs: `python bench.py --gpus N ...` starts its own N rank processes (fresh children,
spawned before anything touches a GPU; rank 0's JSON line is relayed), and
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` works too.
The global problem is sharded contiguously over ranks (strong scaling). The data
path uses the library's own RCCL communicator (one all-gather per QLT run, one
all-reduce per CAAS run; the two CDRs run on two streams by default, at every N, and the
library orders the communicator's collectives across them with events, --serial = one stream);
torch.distributed (gloo) is only the control plane (unique-id broadcast, barriers,
max over ranks). A transport that fails ends the run with a non-zero exit code.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# dmabuf IPC is the only mode the hosts support (RCCL across processes needs it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
CST = 7


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ncells", type=int, default=1500000)
    ap.add_argument("--ntracers", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-cells", type=int, default=0,
                    help="cells of the CPU-baseline run (same tracer count); 0 = the configuration itself")
    ap.add_argument("--check", action="store_true", help="verify properties of the results")
    ap.add_argument("--overlap", nargs="?", const="start", default=None, choices=["start", "down", "mid"],
                    help="QLT and CAAS on two concurrent streams instead of back to back on one: 'start' = both "
                         "begin together (measured on one GPU: 2-3 %% at 1.5M cells, 10 %% at the 187.5k-cell "
                         "shard of an 8-GPU run); 'down' = CAAS::run begins when QLT::run reaches its first-stage "
                         "root->leaves sweep (fp64-issue bound, so the memory-bound CAAS kernels fit beside it). With N > 1 "
                         "the two CDRs still share ONE communicator, whose collectives the library orders "
                         "with events across streams; 'mid' = CAAS::run begins when QLT::run has finished its "
                         "first-stage leaves->root sweep. Default: 'start' at every N (round 3, with CAAS's small "
                         "workgroups: 1.61 -> 1.56 ms per step at 1.5M cells on one GPU, 0.239 -> 0.221 ms at the 187.5k-cell "
                         "shard; at N > 1 the latency of each CDR's one small collective also hides behind the other "
                         "CDR's kernels). The roofline leg always runs on one stream.")
    ap.add_argument("--serial", action="store_true",
                    help="QLT::run then CAAS::run back to back on one stream")
    ap.add_argument("--cu-split", type=int, default=int(os.environ.get("CEDR_BENCH_CU_SPLIT", "0")), choices=[0, 1, 2, 3],
                    help="two-stream modes: CAAS::run on this many of every four groups of eight compute units, "
                         "QLT::run on the others (CU-masked streams); 0 = both on all of them")
    ap.add_argument("--comm", choices=["rccl", "gloo", "peer-flags"], default="rccl",
                    help="data-path transport for N > 1: the library's RCCL communicator (default); for rehearsals on "
                         "one GPU, host-staged torch.distributed/gloo callbacks; or the library's peer-flag transport "
                         "(stream-ordered stores into the peers' IPC-mapped mailboxes + one hipStreamWaitValue64 per "
                         "collective, no communicator library; works with --same-gpu, ranks as processes; at --gpus 1 a "
                         "communicator of ONE rank over the same code path, like --force-rccl)")
    ap.add_argument("--allow-host-fallback", action="store_true",
                    help="launcher mode only: if the RCCL run fails, start a FRESH set of rank processes "
                         "with --comm gloo (host-staged) and say so in the JSON line; without this flag a "
                         "failed transport is a failed run")
    ap.add_argument("--caas-bfb", type=int, default=0, metavar="ACCUM",
                    help="CAAS with the BFB tree all-reduce as its UserAllReducer (the production pairing): "
                         "ACCUM cells per reducer leaf (n_accum_in_place), e.g. 4; 0 = CAAS's own all-reduce")
    ap.add_argument("--types", choices=["cst", "mix"], default="cst",
                    help="QLT tracer problem types: all cst (BASELINE's metric) or the harness's six types "
                         "cst, st, ct, t, nn, cnn in turn (diagnostic)")
    ap.add_argument("--same-gpu", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (needs --comm gloo or --comm peer-flags)")
    ap.add_argument("--force-rccl", action="store_true",
                    help="--gpus 1 only: the CDRs get an RCCL communicator of ONE rank instead of the single-rank one, so "
                         "that every collective of a step really goes through librccl's enqueue path and, on two streams, "
                         "through the library's cross-stream ordering of the shared communicator (CAAS::run: one "
                         "ncclAllReduce; QLT::run has no collective on one rank -- see --pretend-ranks)")
    ap.add_argument("--pretend-ranks", type=int, default=1, metavar="P",
                    help="--gpus 1 only, rehearsal: run as rank 0 of P -- the tree and the cells are sharded P ways and "
                         "this process sweeps the first shard, with QLT::run's all-gather + apex and CAAS::run's all-reduce "
                         "(through RCCL with --force-rccl; the other ranks' all-gather blocks are copies of this rank's). "
                         "`value` then counts the cells of this shard only; what one GPU can measure of an N-GPU step")
    a = ap.parse_args()
    if (a.force_rccl or a.pretend_ranks > 1) and a.gpus != 1:
        ap.error("--force-rccl and --pretend-ranks are rehearsals on ONE GPU (--gpus 1)")
    if a.force_rccl and a.comm == "peer-flags":
        ap.error("--force-rccl and --comm peer-flags are two different one-rank communicators")
    if a.pretend_ranks < 1:
        ap.error("--pretend-ranks must be at least 1")
    if a.check and a.types != "cst":
        ap.error("--check verifies bounds and mass conservation, the properties of cst tracers; mixed types are "
                 "checked against the oracle in tests/")
    return a


def gen_inputs(torch, n, nt, seed, device):
    """SURVEY.md 8(d) synthetic inputs for one shard: arrays [tracer][cell] on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    f64 = torch.float64
    rhom = 0.5 * (1 + torch.rand(n, generator=g, device=device, dtype=f64))
    q_min = -0.75 + torch.rand(nt, n, generator=g, device=device, dtype=f64)
    q_max = q_min + torch.rand(nt, n, generator=g, device=device, dtype=f64)
    q = q_min + (q_max - q_min) * torch.rand(nt, n, generator=g, device=device, dtype=f64)
    Qm_min = q_min * rhom
    Qm_max = q_max * rhom
    Qm_prev = torch.maximum(Qm_min, torch.minimum(Qm_max, q * rhom))
    del q_min, q_max, q
    # perturbation type 1 (cedr_test_randomized.cpp:109-118): permute Qm within the shard
    Qm = torch.empty_like(Qm_prev)
    mode = os.environ.get("CEDR_BENCH_PERTURB", "permute")  # diagnostics only
    for t in range(nt):
        if mode == "none":
            Qm[t] = Qm_prev[t]
        elif mode == "small":  # mass-neutral nudges that stay inside the bounds almost everywhere
            d = 1e-3 * (Qm_max[t] - Qm_min[t]) * (torch.rand(n, generator=g, device=device, dtype=f64) - 0.5)
            Qm[t] = Qm_prev[t] + d - d.roll(1)
        else:
            Qm[t] = Qm_prev[t][torch.randperm(n, generator=g, device=device)]
    return rhom, Qm.contiguous(), Qm_min.contiguous(), Qm_max.contiguous(), Qm_prev.contiguous()


def cpu_baseline(ncells, nt):
    """The oracle (CPU port of the reference algorithm, built by oracle/Makefile; the flags are read back from the library) timed on this
    host on the metric's own configuration: the reported CPU baseline, never the thing measured as
    `value`. One thread (the reference's Kokkos-Serial shape) and the box's CPU share (OpenMP over
    level nodes / tracers, as cedr_qlt.cpp:391-393,570-572 and cedr_caas.cpp:187)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    rng = np.random.default_rng(12345)
    rhom = 0.5 * (1 + rng.random(ncells))
    q_min = -0.75 + rng.random((nt, ncells))
    q_max = q_min + rng.random((nt, ncells))
    q = q_min + (q_max - q_min) * rng.random((nt, ncells))
    Qm_min, Qm_max = q_min * rhom, q_max * rhom
    Qm_prev = np.maximum(Qm_min, np.minimum(Qm_max, q * rhom))
    del q_min, q_max, q
    Qm = np.empty_like(Qm_prev)
    for t in range(nt):
        Qm[t] = Qm_prev[t][rng.permutation(ncells)]
    # threads for the parallel leg: the CPUs this process may use, at most the GPU box's share of
    # host cores for one GPU (16)
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    nthr = max(1, min(o.lib().or_max_threads(), allowed, 16))
    tree = o.Tree.make_1d(ncells)
    qlt = o.OracleQLT(tree, ncells)
    qlt.declare([CST] * nt)
    qlt.set(0, rhom, Qm, Qm_min, Qm_max, Qm_prev)
    caas = o.OracleCAAS([ncells], [CST] * nt)
    caas.set(0, rhom, Qm, Qm_min, Qm_max, Qm_prev)
    # repetitions sized for ~10-20 s of CPU work in all at 1.5M x 40 (one serial QLT run is ~4 s)
    scale = max(1, int(1500000 * 40 / max(1, ncells * nt)))
    reps = {"1": (1 * scale, 3 * scale), "all": (3 * scale, 10 * scale)}
    res = {}
    for label, threads in (("1", 1), ("all", nthr)):
        qlt.set_threads(threads)
        caas.set_threads(threads)
        if label == "all":
            qlt.run()  # warm-up of the thread team
        nq, nc = reps[label]
        tq = qlt.time_run(nq) / nq
        tc = caas.time_run(Qm, nc) / nc
        res[label] = dict(threads=threads, qlt_s=tq, caas_s=tc, qlt_runs=nq, caas_runs=nc,
                          value=2.0 * ncells * nt / (tq + tc),
                          qlt_updates_per_s=ncells * nt / tq, caas_updates_per_s=ncells * nt / tc)
        if nthr == 1:
            res["all"] = res["1"]
            break
    return res, nthr, o.lib().or_build_flags().decode()


def spawn_ranks(a):
    """`python bench.py --gpus N` typed as is: start N fresh rank processes (one per GPU) and
    relay rank 0's JSON line. Runs before this process has imported torch or touched a GPU;
    nothing is ever exec'ed from a process that has."""
    import signal
    import socket
    import subprocess

    def run_once(extra):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                       LOCAL_WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            # rank 0 writes the JSON line to our stdout; the other ranks' stdout goes to stderr
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + extra,
                                          env=env, stdout=None if r == 0 else sys.stderr))
        rc, deadline = 0, None
        while any(p.poll() is None for p in procs):
            for p in procs:
                if p.poll() not in (None, 0) and rc == 0:
                    rc = p.returncode
                    deadline = time.monotonic() + 20  # peers may sit in a collective: do not wait for ever
            if deadline is not None and time.monotonic() > deadline:
                for p in procs:
                    if p.poll() is None:
                        p.send_signal(signal.SIGKILL)  # exactly the children started above
            time.sleep(0.2)
        for p in procs:
            if p.returncode != 0 and rc == 0:
                rc = p.returncode
        return rc

    rc = run_once([])
    if rc != 0 and a.allow_host_fallback and a.comm == "rccl":
        print("bench.py: the RCCL run failed (exit %d); --allow-host-fallback: fresh ranks with --comm gloo" % rc,
              file=sys.stderr)
        rc = run_once(["--comm", "gloo"])
    sys.exit(rc)


def main():
    a = parse()
    if a.overlap is None and not a.serial:
        a.overlap = "start"  # at every N, so that the N = 1 and N > 1 numbers are the same program
    if a.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(a)  # does not return
    # The JSON line is the ONLY thing this program writes to its standard output: libraries underneath print there too
    # (gloo: "[Gloo] Rank 0 is connected to 1 peer ranks ..." at N > 1), so file descriptor 1 is pointed at stderr for the rest
    # of the run and the line goes out through a duplicate of the original.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import compose_amd as ca

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    if a.same_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        if torch.cuda.device_count() == 1 and world > 1:
            # a launcher that hands every rank its own GPU through the visibility mask (ROCR / HIP_VISIBLE_DEVICES):
            # the one device this process sees is this rank's
            local_rank = 0
        else:
            raise SystemExit("rank %d: no GPU %d on this node (%d visible); --gpus must not exceed the GPU count"
                             % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    torch.zeros(1, device=device)  # initialise the HIP context before the library is used

    ncells, nt = a.ncells, a.ntracers
    keep = []  # objects the callbacks refer to
    keep_streams = []
    # The l2r / r2l buffers: the library takes the first allocation unless asked (CEDR_HIP_PLACE_TRIES); this benchmark
    # asks for the best of four and reports what the tries found (`placement` in the line).
    os.environ.setdefault("CEDR_HIP_PLACE_TRIES", "4")
    nshards = world if world > 1 else a.pretend_ranks  # ranks the tree and the cells are sharded over

    def make_cdrs(par, par2):
        tree = ca.Tree.make_1d(ncells, nshards, False, False)  # contiguous ownership, balanced
        qlt = ca.QLT(par, ncells, tree)
        mix = [7, 2, 5, 4, 8, 9]  # cst, s(t), ct, t, nn, cnn
        for i in range(nt):
            qlt.declare_tracer(CST if a.types == "cst" else mix[i % 6], 0)
        qlt.end_tracer_declarations()
        qlt.finish_setup()
        if a.caas_bfb:
            acc = a.caas_bfb
            if ncells % (acc * nshards):
                raise SystemExit("--caas-bfb %d needs ncells divisible by %d" % (acc, acc * nshards))
            btree = ca.Tree.make_1d(ncells // acc, nshards, False, False)
            bfb = ca.BfbTreeAllReducer(par2, btree, ncells // acc, 4 * nt)
            keep.append((btree, bfb))
            # The reducer object itself is CAAS's UserAllReducer (caas::BfbUserAllReducer): run() fuses its reduce pass
            # with the first stage of the reducer's tree (CEDR_HIP_CAAS_BFB_FUSE=0: the two separate launches)
            caas = ca.CAAS(par2, qlt.nlclcells(), bfb, acc)
        else:
            caas = ca.CAAS(par2, qlt.nlclcells())
        for _ in range(nt):
            caas.declare_tracer(CST, 0)
        caas.end_tracer_declarations()
        caas.finish_setup()
        return qlt, caas

    def host_staged():
        def ag_host(x):
            bufs = [torch.zeros(len(x), dtype=torch.float64) for _ in range(world)]
            dist.all_gather(bufs, torch.from_numpy(x))
            return torch.cat(bufs).numpy()

        def ar_host(x):
            t = torch.from_numpy(x.copy())
            dist.all_reduce(t)
            return t.numpy()
        return ca.Parallel.host_staged(rank, world, ag_host, ar_host)

    transport = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if a.comm in ("rccl", "peer-flags"):
            # ONE communicator for both CDRs: the all-gather of QLT::run and the all-reduce of CAAS::run are enqueued in
            # the same order on every rank (and, on two streams, executed in that order: the library's events).
            err = None
            if a.comm == "rccl":
                uid = [ca.Parallel.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
            else:
                try:
                    mailbox, handle = ca.Parallel.peer_mailbox(world)
                except ca.CedrError as e:
                    mailbox, handle, err = None, None, "CedrError: %s" % e
                handles = [None] * world
                dist.all_gather_object(handles, handle)
                if any(h is None for h in handles):
                    err = err or "a peer has no mailbox"
                if not err:
                    try:
                        par = par2 = ca.Parallel.peer_flags(mailbox, handles, rank, world)
                    except ca.CedrError as e:
                        err = "CedrError: %s" % e
                # nobody stores into a mailbox before every rank has mapped them all (and a rank that could not leaves
                # with the others: the gather is the barrier)
                errs = [None] * world
                dist.all_gather_object(errs, err)
                err = next((e for e in errs if e), None)
            try:
                if err:
                    raise RuntimeError(err)
                if a.comm == "rccl":
                    par = par2 = ca.Parallel.rccl(uid[0], rank, world)
                qlt, caas = make_cdrs(par, par2)
                # one untimed exchange through the communicator before anything is measured
                z = torch.zeros((nt, qlt.nlclcells()), dtype=torch.float64, device=device)
                o = torch.ones(qlt.nlclcells(), dtype=torch.float64, device=device)
                qlt.set_rhom(o); qlt.set_Qm(z, z, z, z); qlt.run()
                caas.set_rhom(o); caas.set_Qm(z, z, z, z); caas.run()
                torch.cuda.synchronize()
                del z, o
            except Exception as e:  # noqa: BLE001 -- any failure of the RCCL transport
                err = "%s: %s" % (type(e).__name__, e)
            errs = [None] * world
            dist.all_gather_object(errs, err)
            if any(errs):
                # A transport that does not work is a failed run: never report host-staged
                # numbers as multi-GPU results (a fresh `--comm gloo` run is the caller's choice).
                if rank == 0:
                    print("bench.py: %s transport failed: %s" % ("RCCL" if a.comm == "rccl" else a.comm,
                                                                 next(e for e in errs if e)), file=sys.stderr)
                dist.destroy_process_group()
                sys.exit(3)
            form = ""
            if a.comm == "peer-flags":
                form = (" [put launch + hipStreamWaitValue64 + take launch per collective]"
                        if os.environ.get("CEDR_HIP_PEER_WAIT") == "stream" else " [one launch per collective]")
            transport = ("%s: one communicator of %d ranks%s%s, %s, collectives in program order"
                         % (a.comm, world, " (ALL ON GPU 0: a rehearsal)" if a.same_gpu else "", form,
                            "two streams (event ordered)" if a.overlap else "one stream"))
        else:
            par = par2 = host_staged()
            qlt, caas = make_cdrs(par, par2)
            transport = "gloo-host-staged (rehearsal transport, not RCCL)"
    else:
        if a.comm == "peer-flags":
            mailbox, handle = ca.Parallel.peer_mailbox(1)
            inner = ca.Parallel.peer_flags(mailbox, [handle], 0, 1)
        else:
            inner = ca.Parallel.rccl(ca.Parallel.rccl_unique_id(), 0, 1) if a.force_rccl else ca.Parallel.self_()
        par = par2 = ca.Parallel.rehearsal(inner, a.pretend_ranks) if a.pretend_ranks > 1 else inner
        keep.append((inner, None))
        qlt, caas = make_cdrs(par, par2)
        if a.force_rccl or a.pretend_ranks > 1 or a.comm == "peer-flags":
            transport = ("%s communicator of ONE rank%s, %s" %
                         ("rccl" if a.force_rccl else ("peer-flags" if a.comm == "peer-flags" else "single-rank"),
                          (", rehearsing rank 0 of %d (the other ranks' all-gather blocks are copies of this rank's)"
                           % a.pretend_ranks) if a.pretend_ranks > 1 else "",
                          "two streams (event ordered)" if a.overlap else "one stream"))
    n = qlt.nlclcells()
    # cells `value` counts: the whole job's -- except in a rehearsal of rank 0 of P on one GPU, which sweeps (and counts) one shard
    ncount = n if (world == 1 and a.pretend_ranks > 1) else ncells

    rhom, Qm, Qm_min, Qm_max, Qm_prev = gen_inputs(torch, n, nt, 0x5EED0000 + 1000 * 2 + rank, device)
    qlt.set_rhom(rhom)
    qlt.set_Qm(Qm, Qm_min, Qm_max, Qm_prev)
    caas.set_rhom(rhom)
    caas.set_Qm(Qm, Qm_min, Qm_max, Qm_prev)
    caas.set_Qm_source(Qm)  # every run() reads the pristine Qm: identical work each step
    torch.cuda.synchronize()
    # Default: each CDR on its own stream, both starting together; --serial: QLT::run then CAAS::run on one stream.
    streams = None
    if a.overlap:
        # (experiment knob: CEDR_BENCH_PRIO="q,c" = HIP stream priorities of the QLT and the CAAS stream,
        # lower number = higher priority, torch's range is -1..0)
        pq, pc = [int(x) for x in os.environ.get("CEDR_BENCH_PRIO", "0,0").split(",")]
        streams = (torch.cuda.Stream(device=device, priority=pq), torch.cuda.Stream(device=device, priority=pc))
        if a.cu_split:
            # Each CDR on its own share of the compute units: of every four groups of eight CUs CAAS::run gets
            # cu_split, QLT::run the rest (uniform over the XCDs whichever way the mask's bits map to them).
            ncu = torch.cuda.get_device_properties(device).multi_processor_count
            words = [[0] * ((ncu + 31) // 32) for _ in range(2)]
            for cu in range(ncu):
                words[1 if (cu // 8) % 4 < a.cu_split else 0][cu // 32] |= 1 << (cu % 32)
            masked = [ca.MaskedStream(w) for w in words]
            keep_streams.append(masked)
            streams = tuple(torch.cuda.ExternalStream(m.handle, device=device) for m in masked)

    def use_streams(on):
        qlt.set_stream(streams[0].cuda_stream if on else None)
        caas.set_stream(streams[1].cuda_stream if on else None)
        for _, bfb in keep:
            if bfb is not None:
                bfb.set_stream(streams[1].cuda_stream if on else None)  # the reducer runs inside caas.run()
    if streams and world == 1:
        use_streams(True)

    def barrier():
        if world > 1:
            dist.barrier()

    down_ev = qlt.sweep_down_event() if (streams and a.overlap == "down") else None
    if streams and a.overlap == "mid":  # CAAS::run starts when QLT::run has finished its first-stage leaves->root sweep
        down_ev = qlt.sweep_mid_event()

    def step():
        qlt.run()
        if down_ev:  # CAAS::run starts when QLT::run reaches its first-stage root->leaves sweep
            ca.stream_wait_event(streams[1].cuda_stream, down_ev)
        caas.run()

    def json_head(value_, ms_, nstreams):
        return {
            "metric": "cell-tracer-updates/sec (QLT+CAAS), 1.5M cells \u00d7 40 tracers; % HBM roofline",  # BASELINE.json's string
            "value": value_, "unit": "cell-tracer-updates/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "QLT::run + CAAS::run%s, %d cells x %d cst tracers, balanced 1-D tree, "
                                   "contiguous %d-way sharding" % (" on two concurrent streams" if nstreams == 2 else "",
                                                                   ncells, nt, nshards),
                       "rehearsal": ("rank 0 of %d on one GPU: value counts this shard's %d cells only" % (nshards, n))
                       if (world == 1 and nshards > 1) else None,
                       "ncells": ncells, "ntracers": nt, "cells_per_gpu": n,
                       "transport": transport, "rccl_nranks": world if (world > 1 and a.comm == "rccl") else 0,
                       "streams": nstreams,
                       "cu_split": ("CAAS on %d of every 4 groups of 8 CUs, QLT on the rest" % a.cu_split)
                       if (nstreams == 2 and a.cu_split) else "none",
                       "caas_reducer": ("bfb_tree_allreduce, %d cells per leaf" % a.caas_bfb) if a.caas_bfb
                       else "own all-reduce"}}

    def warm_and_check(what, on_stall):
        """W untimed steps; at N > 1 they must complete (a communicator whose collectives stall -- a rank that died, an
        ordering problem across the two streams -- would otherwise sit in synchronize() until the caller's timeout)."""
        for _ in range(max(a.warmup, 1) if world > 1 else a.warmup):
            step()
        if world > 1:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
            evs = [ev]
            for st in (streams or ()):
                e2 = torch.cuda.Event()
                e2.record(st)
                evs.append(e2)
            deadline = time.monotonic() + float(os.environ.get("CEDR_BENCH_WARMUP_S", "120"))
            while not all(e.query() for e in evs):
                if time.monotonic() > deadline:
                    print("bench.py: rank %d: the warm-up steps (%s) did not complete in time over '%s'" %
                          (rank, what, transport), file=sys.stderr, flush=True)
                    on_stall()
                time.sleep(0.002)

    def timed_steps():
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        barrier()
        d = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([d], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt[0])
        return d

    # N > 1 on two streams: the SAME K steps are first timed on one stream (no cross-stream ordering of the shared
    # communicator's collectives involved), so that a two-stream run that stalls over a real transport still ends with a
    # measured line -- flagged as the one-stream program -- instead of no result: every rank's watchdog then prints / exits.
    dt_one = None
    watchdog = None
    if world > 1 and streams:
        warm_and_check("one stream", lambda: os._exit(6))
        dt_one = timed_steps()

        def fall_back():
            if rank == 0:
                line = json_head(2.0 * ncount * nt / (dt_one / a.steps), 1e3 * dt_one / a.steps, 1)
                line.update(roofline=None, cpu_baseline=None, degraded=True,
                            note="the two-stream run (the default program) did not complete over this transport within "
                                 "its time limit; this is the same K steps back to back on ONE stream, measured before it")
                print(json.dumps(line), file=json_out, flush=True)
            os._exit(0)
        import threading
        # (limit: 150 s, or more for a long run -- 20 x what the one-stream warm-up + K steps just took)
        limit = max(150.0, 20.0 * dt_one * (1.0 + max(a.warmup, 1) / max(a.steps, 1)))
        watchdog = threading.Timer(float(os.environ.get("CEDR_BENCH_WATCHDOG_S", limit)), fall_back)
        watchdog.daemon = True
        watchdog.start()
        use_streams(True)
        if os.environ.get("CEDR_BENCH_SIMULATE_STALL"):  # (test hook: behave like a two-stream run that hangs)
            time.sleep(1e9)
        warm_and_check("two streams", fall_back)
    else:
        warm_and_check("one stream" if not streams else "two streams", lambda: os._exit(6))
    dt = timed_steps()
    ms_per_step = 1e3 * dt / a.steps
    value = 2.0 * ncount * nt / (dt / a.steps)
    if watchdog:  # the two-stream program has completed its timed steps: nothing after this point can stall on it
        watchdog.cancel()

    # Split of the step between the two algorithms (same loop shape, timed apart).
    def timed(fn):
        torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
        s = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize(); barrier()
        e = time.perf_counter() - s
        if world > 1:
            tt = torch.tensor([e], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e = float(tt[0])
        return e / a.steps
    t_qlt = timed(qlt.run)
    t_caas = timed(caas.run)

    # Roofline leg: HIP events around every kernel launch, on the launch stream,
    # over `steps` more steps of the same loop, with the two CDRs on ONE stream so
    # that a kernel's duration is not stretched by a concurrent one.
    if streams:
        torch.cuda.synchronize()
        use_streams(False)
    # The same K steps back to back on ONE stream (the shape of rounds 1-2, of the CPU baseline and of the
    # roofline leg below), reported next to the two-stream `value` at every N: the two programs differ only in
    # the overlap, so a regression of the cross-stream ordering over real RCCL shows up as a difference here.
    t_serial = (dt_one / a.steps) if dt_one is not None else (timed(step) if streams else dt / a.steps)

    # What ONE collective of this run's communicator costs, the way a step uses it: 160 doubles (CAAS::run's all-reduce at 40
    # tracers), 200 calls back to back on one stream, max over ranks; results checked. Only for a real transport (RCCL or
    # peer flags, also as the one-rank communicator of a rehearsal): after the timed steps, which the same collectives
    # have been through already, on the stream the one-stream steps have just used. Reported next to the step, never part of `value`.
    coll = None
    if (world > 1 and a.comm in ("rccl", "peer-flags")) or a.force_rccl or (world == 1 and a.comm == "peer-flags"):
        try:
            P = par.size()
            cnt, calls = 160, 200
            base = torch.arange(cnt, dtype=torch.float64, device=device)
            send = base + 1000.0 * par.rank()
            rs = torch.empty(cnt, dtype=torch.float64, device=device)
            rg = torch.empty(cnt * P, dtype=torch.float64, device=device)
            coll = {"doubles": cnt, "calls_back_to_back": calls}
            for name, fn in (("allreduce", lambda: par.allreduce_sum(send, rs, cnt)), ("allgather", lambda: par.allgather(send, rg, cnt))):
                for _ in range(20):
                    fn()
                torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(calls):
                    fn()
                torch.cuda.synchronize()
                e = time.perf_counter() - t0
                if world > 1:
                    tt = torch.tensor([e], dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    e = float(tt[0])
                coll[name + "_us"] = 1e6 * e / calls
            if world > 1 or a.pretend_ranks == 1:  # (a rehearsal's other blocks are copies of this rank's: nothing to compare with)
                want_sum = base * P + 1000.0 * sum(range(P))
                want_blocks = torch.cat([base + 1000.0 * r for r in range(P)])
                coll["results_ok"] = bool(torch.equal(rs, want_sum) and torch.equal(rg, want_blocks))
        except Exception as e:  # noqa: BLE001 -- an extra: never the reason the line is lost (every rank runs the same code)
            coll = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    ca.profile_start()
    for _ in range(a.steps):
        step()
    recs = ca.profile_stop()
    per = {}
    for name, aux, ms in recs:
        k = (name, aux)
        c = per.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += ms
    # Algorithmic HBM bytes per launch (DESIGN.md "Roofline accounting"), cst tracers:
    #   tree_up stage 1  : reads the 4 l2r fields of every cell-tracer + rhom      = n*(32*nt + 8)
    #   qlt_down stage 1 : reads lo/orig/hi + rhom, writes the cell's new mass     = n*(32*nt + 8)
    #   caas_reduce      : reads Qm, Qm_min, Qm_max, Qm_prev, writes clipped Qm    = n*nt*40
    #   caas_finish      : reads Qm and one bound, writes Qm                       = n*nt*24
    plan = qlt.plan()
    ntiles1 = len(plan["local"][0]["tiles"]) if plan["local"] else 0
    alg = {("tree_up_kernel", ntiles1): n * (32.0 * nt + 8), ("qlt_down_kernel", ntiles1): n * (32.0 * nt + 8),
           ("caas_reduce_kernel", n): n * nt * 40.0, ("caas_finish_kernel", n): n * nt * 24.0,
           # user-reducer variant: the same streams + 4 partial sums per group of caas_bfb cells
           ("caas_reduce_accum_kernel", n): n * nt * (40.0 + 32.0 / max(a.caas_bfb, 1)),
           # ... fused with the BFB reducer's first stage: nothing but the cells' own streams (+ KBs of tile outputs)
           ("caas_reduce_bfb_kernel", n): n * nt * 40.0}
    kernels = []
    for (name, aux), (cnt, tot) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        avg_ms = tot / cnt
        e = dict(kernel=name, launch=aux, launches_per_step=cnt / a.steps, avg_ms=avg_ms,
                 ms_per_step=tot / a.steps)
        if (name, aux) in alg:
            gbs = alg[(name, aux)] / (avg_ms * 1e-3) / 1e9
            e.update(algorithmic_bytes=alg[(name, aux)], achieved_GBs=gbs, frac=gbs / HBM_PEAK_GBS)
        kernels.append(e)
    # HBM traffic per launch: NOT measured by this run. It is read from the committed PMC profile of
    # this same command (profiles/rNN_traffic.json, made by tools/profile_traffic.sh; the newest
    # round's file whose configuration matches), and the line says so in roofline.traffic_source.
    traffic_source = None
    traffic_refused = []
    ksha = ca.kernels_sha256()
    for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles"))
                        if f.startswith("r") and f.endswith("_traffic.json")), reverse=True):
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", cand)))
            if world != 1 or a.pretend_ranks > 1 or tr["config"] != {"ncells": ncells, "ntracers": nt}:
                continue
            if tr.get("kernels_sha256") != ksha:
                # counters of OTHER kernel code say nothing about this run's: refused, and the line says so
                traffic_refused.append("profiles/%s (kernels_sha256 %s)" % (cand, str(tr.get("kernels_sha256"))[:12]))
                continue
            for k in kernels:
                t = tr["kernels"].get(k["kernel"])
                if t and "algorithmic_bytes" in k:
                    k["traffic"] = t["hbm_bytes"]
            traffic_source = "profiles/" + cand + " (PMC run of this command, not this run)"
            break
        except (OSError, ValueError, KeyError):
            continue
    grids = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_kernel_stats_by_grid.csv"))
    by_grid = "profiles/" + grids[-1] if grids else "a by-grid split of the kernel trace"
    dom = next((k for k in kernels if "achieved_GBs" in k), None)
    roofline = None
    if dom:
        roofline = dict(bound="hbm", kernel=dom["kernel"], achieved=dom["achieved_GBs"],
                        peak=HBM_PEAK_GBS, unit="GB/s", frac=dom["frac"], traffic=dom.get("traffic"),
                        traffic_source=traffic_source if dom.get("traffic") is not None else None,
                        traffic_refused=traffic_refused or None,
                        avg_ms=dom["avg_ms"], algorithmic_bytes_per_launch=dom["algorithmic_bytes"],
                        launch="first stage: %d workgroups (tiles)" % dom["launch"],
                        rocprof="every stage runs the same sweep kernels (qlt_down_sp_kernel / tree_up_micro_kernel),"
                                " so rocprofv3 --stats averages all stages under one name; "
                                "%s splits the same trace by kernel instantiation "
                                "and launch grid (tools/kernel_stats_by_grid.py)" % by_grid)
        if dom["kernel"] == "qlt_down_kernel":
            roofline["bound_note"] = ("reported against the HBM roofline, the metric's roofline. Since round 5 the first-stage tiles of a "
                                      "list of up to 64 shape-preserving tracers keep THREE LDS levels: this launch walks one more "
                                      "dependent level per chunk (0.39 -> 0.43 ms, 61 -> 56 % of 8 TB/s) and QLT::run as a whole gets "
                                      "faster (half the output lines of the leaves->root launch, half the second stage, no third: "
                                      "qlt.frac; profiles/r05_experiments.md 3). With two levels and no node problems the kernel takes "
                                      "0.36 ms: what the memory system delivers to its access pattern (three read rows and one "
                                      "written row per tracer in 1.4 KB pieces 12 MB apart) at five workgroups per CU "
                                      "(profiles/r04_experiments.md 1, 5)")
        elif dom["kernel"] == "tree_up_kernel":
            roofline["bound_note"] = ("the reads stream at the copy rate (0.317 ms without the output stores); the gap is the 1.3 M "
                                      "isolated 128-byte output lines written between them: free to issue (a build that writes "
                                      "them all to the same lines runs at the no-store speed), 0.07-0.11 ms on the memory side "
                                      "depending on which device memory the allocation got, and an eighth of them already "
                                      "costs half of that (profiles/r04_experiments.md 3; r03_experiments.md 12). Whichever of the "
                                      "two first-stage sweeps is slower in this process is reported here")

    # Pack / unpack either side of run() (reported apart from `value`).
    def timed_sync(fn, reps=5):
        fn(); torch.cuda.synchronize()
        s0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - s0) / reps
    out_buf = torch.empty_like(Qm)
    t_set = timed_sync(lambda: qlt.set_Qm(Qm, Qm_min, Qm_max, Qm_prev))
    t_get = timed_sync(lambda: qlt.get_Qm(out_buf))
    # Zero-copy mode: the first-stage kernels sweep the caller's arrays themselves.
    t_run_packed = timed_sync(qlt.run)
    qlt.bind_arrays(rhom, Qm, Qm_min, Qm_max, Qm_prev, out_buf)
    t_run_bound = timed_sync(qlt.run)
    qlt.bind_arrays(None, None, None, None, None, None)
    pack = {"qlt_set_Qm_ms": 1e3 * t_set, "qlt_set_Qm_GBs": 64.0 * n * nt / t_set / 1e9,
            "qlt_get_Qm_ms": 1e3 * t_get, "qlt_get_Qm_GBs": 16.0 * n * nt / t_get / 1e9,
            "qlt_run_ms": 1e3 * t_run_packed,
            "qlt_set_run_get_ms": 1e3 * (t_set + t_run_packed + t_get),
            "qlt_zero_copy_run_ms": 1e3 * t_run_bound,
            "note": "bulk [tracer][cell] arrays <-> field-major QLT buffers; with bind_arrays QLT::run "
                    "reads and writes the caller's arrays itself (no set_Qm / get_Qm pass), as CAAS does"}
    del out_buf

    ok = None
    if a.check:
        out = torch.empty_like(Qm)
        qlt.get_Qm(out)
        torch.cuda.synchronize()
        ok = bool(((out >= Qm_min) & (out <= Qm_max)).all())
        mass = torch.stack([out.sum(1), Qm_prev.sum(1), Qm_prev.abs().sum(1)])
        if world > 1:
            mass = mass.cpu(); dist.all_reduce(mass)
        ok = ok and bool(((mass[0] - mass[1]).abs() / mass[2]).max() < 1e-13)
        caas.get_Qm(out)
        torch.cuda.synchronize()
        ok = ok and bool(((out >= Qm_min) & (out <= Qm_max)).all())
        if world > 1:  # every rank's verdict
            okt = torch.tensor([1 if ok else 0]); dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            ok = bool(int(okt[0]))

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            cc = a.cpu_cells or ncells
            res, nthr, oflags = cpu_baseline(cc, nt)
            cpu = dict(value=res["all"]["value"], unit="cell-tracer-updates/s", cores=res["all"]["threads"],
                       kind="port",
                       sample="oracle (CPU port of the reference algorithm, built with '%s' as the library itself "
                              "reports, OpenMP over level nodes / tracers) QLT::run + CAAS::run on %d cells x %d cst tracers%s: "
                              "%d + %d runs on %d threads (this box's CPU share for one GPU; the host reports "
                              "%d CPUs), and %d + %d runs on 1 thread (single_thread)"
                              % (oflags, cc, nt, "" if cc == ncells else " (a sample of the %d-cell configuration)" % ncells,
                                 res["all"]["qlt_runs"], res["all"]["caas_runs"], nthr, os.cpu_count() or 0,
                                 res["1"]["qlt_runs"], res["1"]["caas_runs"]),
                       host_cores=os.cpu_count(), single_thread=res["1"], all_threads=res["all"])
        except Exception as e:  # the baseline must never take the bench down
            cpu = dict(value=None, unit="cell-tracer-updates/s", cores=0, kind="port",
                       sample="failed: %r" % (e,))

    caas_fused = bool(a.caas_bfb) and caas.fused_with_bfb()
    caas_bytes = 64.0 + (64.0 / a.caas_bfb if (a.caas_bfb and not caas_fused) else 0.0)
    # what QLT::finish_setup's placement tries found: time of the first-stage sweeps over every candidate allocation
    rep = qlt.placement_report()
    placement = {"report": rep, "tries_env": os.environ.get("CEDR_HIP_PLACE_TRIES")}
    for part in rep.split(";"):
        w = part.split()
        if len(w) > 2 and w[1] == "ms:" and "kept" in w:
            ms = [float(x) for x in w[2:w.index("kept")]]
            placement[w[0]] = {"candidates_ms": ms, "kept": int(w[w.index("kept") + 1]),
                               "spread": (max(ms) - min(ms)) / min(ms) if ms else None}
    if rank == 0:
        line = json_head(value, ms_per_step, 2 if streams else 1)
        line.update({
            "serial": {"ms_per_step": 1e3 * t_serial, "value": 2.0 * ncount * nt / t_serial,
                       "note": "the same steps with QLT::run and CAAS::run back to back on one stream"},
            # whole-run fractions of the HBM roofline: algorithmic bytes per cell-tracer x this GPU's cells x
            # tracers / the run's time / 8 TB/s (QLT: every cell record read on the way up and again on the way down + the
            # result written, rhom twice per cell; CAAS: 40 B in the reduce pass + 24 B in the finish pass, + the
            # per-group partial sums written and read back when a BFB reducer is NOT fused with the reduce pass)
            "qlt": {"ms_per_run": 1e3 * t_qlt, "updates_per_s": ncount * nt / t_qlt,
                    "bytes_per_cell_tracer": 64.0 + 16.0 / nt,
                    "frac": (64.0 + 16.0 / nt) * n * nt / t_qlt / 1e9 / HBM_PEAK_GBS},
            "caas": {"ms_per_run": 1e3 * t_caas, "updates_per_s": ncount * nt / t_caas,
                     "bytes_per_cell_tracer": caas_bytes, "frac": caas_bytes * n * nt / t_caas / 1e9 / HBM_PEAK_GBS,
                     "fused_with_bfb_reducer": caas_fused},
            "collective": coll, "placement": placement, "kernels_sha256": ksha,
            "pack_unpack": pack, "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
        })
        if ok is not None:
            line["check_ok"] = ok
        print(json.dumps(line), file=json_out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if ok is False:
        sys.exit(4)  # --check failed: the JSON line says check_ok false, and the run fails


if __name__ == "__main__":
    main()
