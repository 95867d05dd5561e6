#!/usr/bin/env python3
"""bench.py -- deskewed images/s of the projection-std-dev angle sweep on MI355X.

Workload (BASELINE.json configs[1], "C2"): 2480x3508 8-bit gray scans, +-10 deg @ 0.05 deg sweep
(400 candidates: the reference's half-open range, packages/lib/src/projection.rs:36-38).  One
"step" = one pass of the hot path (threshold-fused bit-pack -> fused rotate/project sweep ->
std-dev -> arg-max) over a batch of `--scans` synthetic cards that are resident in HBM before
the timed region starts.

N > 1: one process per GPU, every rank sweeps its own shard of the batch (weak scaling, no data-path
collective; scans are independent, task.rs:19-38); the only exchange is the gather of the best
indices after the last step.  The ranks come either from the driver's torchrun (RANK / LOCAL_RANK /
WORLD_SIZE in the environment) or -- when `--gpus N` is given without that environment -- from this
script itself, which then starts N children BEFORE it touches the GPU and relays rank 0's line.
`--gpus N` that disagrees with WORLD_SIZE is an error, never a silent single-GPU run.

`--dry-run` rehearses exactly that launch on CPU (gloo, no HIP call, no compute): shard sizes,
barrier/max timing and the gather, so the N-rank plumbing is testable without GPUs.

Prints ONE JSON line (rank 0) with the contract's fields plus `roofline` and `cpu_baseline`.

roofline: the sweep kernel never moves SURVEY 8(d)'s algorithmic bytes through HBM (the scan is a
1-bit/px image staged in LDS, 32 px per lane-operation), so that figure is kept as the labelled side
value `algorithmic_GBps`, NOT as the fraction.  `bound` names the resource the kernel's own counters
show closest to its peak (VALU issue, SCALAR issue, the LDS array or HBM), `achieved`/`peak`/`frac` are that
resource's measured rate against its peak (<= 1), `wait_frac` is the share of wave cycles spent in s_waitcnt,
and `traffic` / `hbm_frac` are the HBM bytes per launch from
rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this very run (child processes started before the parent
touches the GPU), or -- if rocprofv3 cannot run -- the figure from profiles/ with its source named.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "omr-img-corrector_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

ROWS, COLS = 3508, 2480
MAX_ANGLE, STEP = 10, 0.05
# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0   # 8.0 TB/s spec
N_CU, N_SIMD, CLOCK_GHZ = 256, 1024, 2.4
VALU_ISSUE_CYCLES = 2.0  # cycles a plain wave64 VOP2 holds its SIMD with >= 2 waves resident (constants table)
VALU_LANE_MIX_CYCLES = 3.7  # scan-lane kernel: 31 VALU per row of which 27 are 4-cycle forms (v_alignbit, v_and_or, v_bitop3, v_bcnt) and 4 v_mov
VALU_MIX_CYCLES = 3.86   # the same for THIS kernel's instruction mix (v_alignbit, v_bfi, v_and_or, v_lshlrev, VOP3 forms take 4:
#                          tools/valu_issue.hip "sweep-kernel mix", profiles/r02_valu_issue.md): what `bound` is decided with
METRIC = "deskewed images/sec, 2480x3508 gray, +-10deg@0.05deg sweep; HBM GB/s vs roofline"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scans", type=int, default=512,
                    help="scans per GPU per step (default = C3's per-GPU share: 4096 scans / 8 GPUs, 4.45 GB of u8)")
    ap.add_argument("--distinct", type=int, default=64,
                    help="distinct seeded cards per GPU; the batch repeats them (SURVEY.md 8d: 64 seeds x repeats)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams per GPU; the sweep kernel fills the chip by itself, so 1 keeps the "
                         "per-kernel HIP-event / rocprof durations free of cross-stream overlap")
    ap.add_argument("--group", type=int, default=32,
                    help="scans carried by one launch of each kernel (a multiple of 8 keeps one scan per XCD; measured on "
                         "one box: 8 -> 4.61 k, 16 -> 4.78 k, 32 -> 4.79 k images/s: the partial last wave of workgroups "
                         "of a launch is amortised over more work)")
    ap.add_argument("--lanes", type=int, default=512,
                    help="scan-lane sweep (DESIGN.md 4.6): scans carried by one launch, 64 scans per wavefront "
                         "(omr_batch_set_lanes); 0 = the run-merging path with --group scans per launch")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-clock budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes")
    ap.add_argument("--no-deskew", action="store_true", help="skip the legs that also produce the deskewed images")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-memory end-to-end leg (e2e_host_images_per_s)")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the N-rank launch (gloo, no GPU work)")
    ap.add_argument("--share-gpus", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r computes on device r %% device_count and the "
                         "result gather runs over gloo (RCCL refuses two ranks on one device); never used for reported numbers")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cards", default=None, help=argparse.SUPPRESS)  # .npy of pre-generated cards (PMC children)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: `--gpus N` without a torchrun environment -> N children, one per GPU
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Parent of N rank processes.  It never imports torch / touches HIP: it only spawns, watches ALL of them,
    relays rank 0's JSON line and the worst exit code.  A rank that dies early (bad device, out of memory, RCCL
    initialisation) would leave the others in init_process_group / a barrier until the store times out, minutes
    later: the first non-zero exit ends the run -- the survivors are terminated, nothing is retried."""
    import threading
    n = args.gpus
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        # the host driver of this pool only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's
        # hipIpcGetMemHandle fails with "invalid argument" (task environment notes); an explicit setting wins
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("OMR_BENCH_TIMEOUT_S", "3000"))
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = "rank %d exited with code %d" % bad[0]
        elif time.monotonic() > deadline:
            failed = "timeout"
        if failed or all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    if failed:
        for p in procs:  # exactly the children started above, by handle
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    out0 = b"".join(c for c in chunks if c)
    codes = [p.poll() for p in procs]
    if failed:
        sys.stderr.write("bench.py: %s; rank exit codes %s\n" % (failed, codes))
        sys.stdout.write(out0.decode("utf-8", "replace"))
        raise SystemExit(max([abs(c) for c in codes if c] or [1]))
    line = None
    for ln in out0.decode("utf-8", "replace").splitlines():
        if ln.startswith("{"):
            line = ln
    if line is None:
        raise SystemExit("bench.py: rank 0 printed no JSON line")
    rec = json.loads(line)
    if rec.get("n_gpus") != n:
        raise SystemExit("bench.py: asked for %d ranks, rank 0 reports n_gpus=%s" % (n, rec.get("n_gpus")))
    rec["launcher"] = "bench.py spawned %d rank processes (RANK/LOCAL_RANK/WORLD_SIZE, 127.0.0.1:%d)" % (n, port)
    print(json.dumps(rec))


# ------------------------------------------------------------------------------------------------
def base_record(args, world, value, elapsed, B, A, G, D=None, path="run-merging"):
    D = B if D is None else D
    return {
        "metric": METRIC,
        "value": value,
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / max(1, args.steps) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "C2: 2480x3508 8-bit gray scan, +-10deg @ 0.05deg = %d candidates "
                               "(reference half-open range), %d scans/GPU/step resident in HBM "
                               "(%d distinct seeded cards, repeated), projection-std-dev sweep (threshold fused)"
                               % (A, B, D),
                   "scans_per_gpu_per_step": B, "distinct_cards_per_gpu": D, "candidates": A, "global_batch": B * world,
                   "parallelism": "scan-sharded x%d, host-side gather" % world, "streams_per_gpu": args.streams,
                   "scans_per_kernel_launch": G, "sweep_kernel": path},
    }


def dry_run(args):
    """The N-rank launch without GPUs: gloo ranks, round-robin shards, barrier + max timing, gather."""
    import torch
    from oics import dist as odist
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:  # before the rendezvous: a mismatch must not hang
        raise SystemExit("--gpus %d does not match WORLD_SIZE %s" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
    if os.environ.get("OMR_BENCH_TEST_FAIL_RANK") == os.environ.get("RANK", "0"):  # tests: a rank that dies before the rendezvous
        raise SystemExit(3)
    rank, local_rank, world = odist.init(backend="gloo")
    B = args.scans
    n_total = B * world
    mine = odist.shard_indices(n_total, rank, world)
    assert len(mine) == B, "weak scaling: every rank owns exactly --scans scans"
    best = torch.tensor(mine, dtype=torch.int32)  # stand-in result = the scan's global index
    cpu = torch.device("cpu")

    def fence():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        pass
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    fence()
    elapsed = odist.barrier_max_seconds(time.perf_counter() - t0, cpu)
    full = odist.gather_results(best, n_total, rank, world)
    ok = full.tolist() == list(range(n_total))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if not ok:
        raise SystemExit("dry run: gather order wrong on rank %d" % rank)
    if rank == 0:
        rec = base_record(args, world, n_total * args.steps / elapsed, elapsed, B, 400, min(args.group, B))
        rec.update(dry_run=True, data="none (launch rehearsal on CPU: gloo ranks, no sweep is run, value is not a measurement)",
                   gathered_results=int(full.numel()), shard_sizes=[len(odist.shard_indices(n_total, r, world)) for r in range(world)])
        print(json.dumps(rec))


# ------------------------------------------------------------------------------------------------
def card_order(B, D):
    """card of scan i: (i + 7 * (i // 64)) % D -- the 64 lanes of every scan group carry different cards than the same
    lanes of the other groups"""
    return [(i + 7 * (i // 64)) % D for i in range(B)]


def _one_card(seed):
    from oics import synth
    return synth.make_card(ROWS, COLS, seed)


def make_cards(n, seed0):
    """n distinct seeded cards (numpy, ~2 s each: a process pool, forked before anything touches the GPU)."""
    import multiprocessing as mp
    workers = max(1, min(n, (os.cpu_count() or 2) // max(1, int(os.environ.get("WORLD_SIZE", "1"))), 16))
    seeds = [seed0 + i for i in range(n)]
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_one_card, seeds)
    else:
        res = [_one_card(sd) for sd in seeds]
    return np.stack([r[0] for r in res]), [r[1] for r in res]


def cpu_baseline(checks, budget_s):
    """The oracle ("port" of the reference CPU path, oracle/oracle.c) timed on this box's host
    cores on a bounded sample of the same workload: whole 400-candidate sweeps of one scan,
    angle-parallel over all cores, repeated until about 2/3 of `budget_s` seconds are spent; then
    16-candidate slices on one thread for the rest.  Also the run's parity check: every candidate's
    scores must equal the GPU's bit for bit -- `checks` = [(scan index, gray card, GPU v_sd, GPU h_sd)]: the timed
    sweeps go round the list, so scans of BOTH quads of scan groups of the launch are compared (index 0 and one >= 256)."""
    from oracle import oracle as orc
    orc.build()
    bins = [orc.threshold_binary(g) for _, g, _, _ in checks]
    b = bins[0]
    Ms = orc.rotation_matrices(ROWS, COLS, MAX_ANGLE, STEP)
    A = Ms.shape[0]
    cores = min(os.cpu_count() or 1, 64)
    orc.sweep_matrices(b, Ms[:2], threads=1, want_proj=False, fast=True)  # warm-up
    sweeps, t_all, ok, compared = 0, 0.0, True, []
    while sweeps < len(checks) or t_all < budget_s * 2.0 / 3.0:
        k = sweeps % len(checks)
        t0 = time.perf_counter()
        _, _, vs, hs = orc.sweep_matrices(bins[k], Ms, threads=cores, want_proj=False, fast=True)
        t_all += time.perf_counter() - t0
        sweeps += 1
        ok = ok and bool((vs.view(np.uint64) == checks[k][2].view(np.uint64)).all()
                         and (hs.view(np.uint64) == checks[k][3].view(np.uint64)).all())
        if checks[k][0] not in compared:
            compared.append(checks[k][0])
    n1, t_1 = 0, 0.0
    while n1 < 16 or (t_1 < budget_s / 3.0 and n1 + 16 <= A):
        t0 = time.perf_counter()
        orc.sweep_matrices(b, Ms[n1:n1 + 16], threads=1, want_proj=False, fast=True)
        t_1 += time.perf_counter() - t0
        n1 += 16
    return {
        "value": sweeps / t_all,
        "unit": "images/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d whole sweep(s) of one scan (all %d candidates), angle-parallel OpenMP over all host cores, %.1f s "
                  "(oracle/oracle.c -O3: warp + clone + 2 projection passes + 2 std-devs per candidate)"
                  % (sweeps, A, t_all),
        "single_thread_value": n1 / (t_1 * A),
        "single_thread_sample": "%d candidates, 1 thread, %.1f s (threads=1 is what every reference caller passes)"
                                % (n1, t_1),
        "parity_vs_gpu": ok,
        "parity_scans": compared,
    }


def e2e_host(cards, B, order, repeats=3):
    """SURVEY.md 8(d)'s end-to-end figure (the reference times image-in-memory to result, packages/core/src/main.rs:68-95):
    the same B scans, binarised, starting in HOST memory, through omr_host_batch_run -- H2D of every 8.7 MB scan and D2H
    of the results inside the timed region; plan, pinned ring and device stages created outside it.  Two variants:
    pageable source (copier threads -> pinned ring -> DMA) and page-locked source (DMA straight from the caller's memory)."""
    import torch
    from oics import projection
    D = cards.shape[0]
    binc = [np.where(cards[i] > 127, 255, 0).astype(np.uint8) for i in range(D)]
    pageable = [binc[order[i]].copy() for i in range(B)]
    pinned_t = torch.empty((B, ROWS, COLS), dtype=torch.uint8).pin_memory()
    pinned_np = pinned_t.numpy()
    for i in range(B):
        pinned_np[i] = pageable[i]
    pinned = [pinned_np[i] for i in range(B)]
    t0 = time.perf_counter()
    hb = projection.HostBatch(ROWS, COLS, MAX_ANGLE, STEP, B, n_devices=1)
    create_s = time.perf_counter() - t0
    _, spl, lane = hb.info()
    out = {"what": "omr_host_batch_run: %d binarised scans from host memory to best angles on the host, 1 GPU; context "
                   "(plan, pinned ring of 3 x 16 scans, 2 device stages of one launch) created outside the timed region; best of %d runs; "
                   "three transfer modes: pageable u8 (copier threads -> pinned ring -> DMA), page-locked u8 (DMA from the caller's memory), "
                   "packed (the copier threads pack to 1 bit per pixel, 1/8 of the bytes cross the link)" % (B, repeats),
           "scans_per_launch": spl, "scan_lane": lane, "context_creation_s": create_s}
    ref = None
    nw = (COLS + 31) // 32
    # (name, source, pinned, packed, scans per launch): the u8 modes are bound by the copies -- 64-scan launches keep the
    # exposed last sweep short; packed transfers (1 bit per pixel made by the copier threads, 1/8 of the bytes uploaded) are
    # bound by the sweep and the host's packing rate -- 128-scan launches sweep faster per scan
    for name, src, pin, pk, launch in (("pageable", pageable, False, False, 64), ("pinned", pinned, True, False, 64),
                                       ("packed", pageable, False, True, 128)):
        hb.set_launch(min(launch, 64 * ((B + 63) // 64)))
        hb.run(src[:min(B, launch)], pinned=pin, packed=pk)
        best_t = None
        for _ in range(repeats):
            t0 = time.perf_counter()
            best, _, _, _ = hb.run(src, pinned=pin, packed=pk)
            dt = time.perf_counter() - t0
            best_t = dt if best_t is None else min(best_t, dt)
        if ref is None:
            ref = best
        out[name + "_images_per_s"] = B / best_t
        out[name + "_h2d_GBps"] = B * (ROWS * nw * 4 if pk else ROWS * COLS) / best_t / 1e9
        out[name + "_scans_per_launch"] = hb.info()[1]
        out[name + "_agrees"] = bool((best == ref).all())
    out["packed_host_read_GBps"] = B * ROWS * COLS / (B / out["packed_images_per_s"]) / 1e9
    hb.close()
    return out, ref


def pmc_passes(args, cards_path, workdir):
    """HBM + SQ counter passes of the sweep kernel, each a rocprofv3 child running this script with
    --pmc-child on the same cards (started before this process touches the GPU)."""
    from oics import pmc
    if pmc.rocprof() is None:
        return {"error": "rocprofv3 not found"}
    # (a scan-lane launch carries the whole batch: the children sweep the same number of scans as the parent so that
    # "per launch" means the same launch)
    argv = [os.path.abspath(__file__), "--pmc-child", "--steps", "3", "--warmup", "1",
            "--scans", str(args.scans if args.lanes > 0 else min(args.scans, 64)),
            "--group", str(args.group), "--lanes", str(args.lanes), "--streams", str(args.streams), "--cards", cards_path]
    res = {"passes": {}}
    for name, ctrs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("sq", pmc.SQ_PASS)):
        c, d = pmc.run_pass(ctrs, argv, os.path.join(workdir, name), timeout_s=240,
                            log=os.path.join(workdir, name + ".log"))
        if c is None:
            res["passes"][name] = {"error": d}
            continue
        k = pmc.pick(c.keys(), "slane_kernel") or pmc.pick(c.keys(), "runs_kernel") or pmc.pick(c.keys(), "sweep_lds_kernel")
        if k is None:
            res["passes"][name] = {"error": "sweep kernel not in the counter file"}
            continue
        # first dispatch = the plan's dry run on one white scan: dropped from the counters AND from the durations
        vals = {n: pmc.mean(v[1:] if len(v) > 2 else v) for n, v in c[k].items()}
        durs = d.get(k, [])
        res["passes"][name] = {"kernel": k, "dispatches": len(next(iter(c[k].values()))), "per_launch": vals,
                               "kernel_us_profiled": pmc.mean(durs[1:] if len(durs) > 2 else durs)}
    return res


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    in_launcher_env = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_launcher_env:
        return launch_ranks(args, argv)
    if args.dry_run:
        return dry_run(args)

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world_env))
    rank_env = int(os.environ.get("RANK", "0"))

    # ---- everything that must happen before this process touches the GPU
    B = args.scans
    D = max(1, min(args.distinct, B))  # distinct cards; scan i of the batch is card order[i]
    if args.cards:
        cards = np.load(args.cards)
        D = cards.shape[0]
        thetas = None  # (cards from a file carry no injected angles: the accuracy check is the un-profiled run's)
    else:
        cards, thetas = make_cards(D, 2 + rank_env * D)  # seed 2 = C2's card
    pmc_res = None
    tmpdir = None
    if world_env == 1 and not args.pmc_child and not args.no_pmc:
        tmpdir = tempfile.mkdtemp(prefix="omr_bench_pmc_", dir="/tmp")
        cards_path = os.path.join(tmpdir, "cards.npy")
        np.save(cards_path, cards)
        try:
            pmc_res = pmc_passes(args, cards_path, tmpdir)
        except Exception as e:  # noqa: BLE001  (profiling must never take the bench line down)
            pmc_res = {"error": repr(e)}
        try:
            os.remove(cards_path)
        except OSError:
            pass

    import torch
    from oics import dist as odist
    from oics import projection
    import oics

    rank, local_rank, world = odist.init(backend="gloo" if args.share_gpus else None)
    if not torch.cuda.is_available() or oics.lib().omr_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    if args.share_gpus:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = torch.device("cpu") if args.share_gpus else dev  # where the (tiny) exchanged tensors live

    N, A = projection.candidate_count(MAX_ANGLE, STEP)
    # the batch: B scans, each its own buffer in HBM (repeats of the D cards).  Every scan group of 64 holds the cards in
    # another lane order (round-4 verdict, item 1d): identical groups would hide a launch that mixed up groups or lanes
    order = card_order(B, D)
    scans = torch.from_numpy(cards).to(dev)[torch.tensor(order, device=dev)].contiguous()
    best = torch.full((B,), -1, dtype=torch.int32, device=dev)
    vs = torch.zeros((B, A), dtype=torch.float64, device=dev)
    hs = torch.zeros((B, A), dtype=torch.float64, device=dev)
    batch = projection.Batch(ROWS, COLS, MAX_ANGLE, STEP, device=local_rank, n_streams=args.streams)
    lanes_mode = False
    if args.lanes > 0:
        try:
            batch.set_lanes(min(B, args.lanes))  # the programs of every strip are generated on the device, once
            lanes_mode = True
        except oics.OmrError as e:
            if e.code != -213:
                raise
    if lanes_mode:
        G = 64 * ((min(B, args.lanes) + 63) // 64)  # scans per launch of the scan-lane kernel
        G = min(G, B) if B % 64 == 0 else G
    else:
        G = max(1, min(args.group, B))
        batch.set_group(G)  # scans per kernel launch

    def step():
        # black_max = 127 fuses transfer_gray_image_to_thresh_binary into the bit-pack
        batch.run_device(scans.data_ptr(), ROWS * COLS, COLS, B, 127, best.data_ptr(), vs.data_ptr(), hs.data_ptr())

    def fence():
        batch.sync()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = odist.barrier_max_seconds(time.perf_counter() - t0, xdev)

    if args.pmc_child:
        batch.close()
        print(json.dumps({"pmc_child": True, "steps": args.steps}))
        return

    # the path's only exchange: gather of the per-scan results (outside the per-step loop, as in
    # a real batch job where it happens once)
    all_best = odist.gather_results(best.to(xdev), B * world, rank, world)

    # ---- the whole unit of work of the reference (omr.rs:339-452, core/src/main.rs:69-92): the same steps with the
    # deskewed image produced INSIDE the timed region -- sweep -> arg-max -> CONTAIN warp by the detected angle
    # (LINEAR as core/src/main.rs:72-81; NEAREST as omr.rs:408-445), winners read on the device.  `value` stays the
    # angle-detection rate (comparable across rounds); these are reported beside it.
    deskew = {}
    if not args.no_deskew:
        dr, dc = batch.deskew_canvas()
        canvases = torch.empty((B, dr, dc), dtype=torch.uint8, device=dev)
        sizes = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        for name, interp in (("linear", 1), ("nearest", 0)):
            def dstep():
                batch.deskew_device(scans.data_ptr(), ROWS * COLS, COLS, B, 127, interp, 255, canvases.data_ptr(), dr * dc, dc,
                                    sizes.data_ptr(), best.data_ptr())
            dstep()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dstep()
            fence()
            d_el = odist.barrier_max_seconds(time.perf_counter() - t0, xdev)
            deskew[name] = B * world * args.steps / d_el
        sz = sizes.cpu().numpy().astype(np.int64)
        deskew["canvas_bytes_per_scan_mean"] = float((sz[:, 0] * sz[:, 1]).mean())
        del canvases

    # roofline leg: duration of the sweep stage of a launch group (G scans) from HIP events recorded on
    # the stream the kernels are launched on, over min(K, 50) more steps.  The stage is `launches` kernel
    # launches: the run-merging kernel (G scans, both projections, one launch) plus one gather launch
    # per scan when some candidates do not qualify for run-merging; the dominant kernel's mean launch
    # time is stage / launches.
    n_runs, n_gather = batch.info()
    launches = 1 if lanes_mode else (1 if n_runs > 0 else 0) + (G if n_gather > 0 else 0)
    batch.set_timing(True)
    for _ in range(min(args.steps, 50)):
        step()
    k_sum_ms, k_n = batch.kernel_ms()
    batch.set_timing(False)
    stage_ms = k_sum_ms / max(1, k_n)
    kernel_ms = stage_ms / launches
    kernel_s = kernel_ms * 1e-3
    algo_bytes_scan = float(A) * ROWS * COLS  # binarised image streamed once per candidate
    algo_bytes = G * algo_bytes_scan / launches

    total_scans = B * world * args.steps
    value = total_scans / elapsed
    out = None
    if rank == 0:
        detected = [(int(k) - N) * STEP for k in best.cpu().tolist()]
        acc_ok = all(abs(d - thetas[order[i]]) < 0.5 for i, d in enumerate(detected)) if thetas else None  # lib.rs:103-113
        out = base_record(args, world, value, elapsed, B, A, G, D, "scan-lane" if lanes_mode else "run-merging")
        if deskew:
            out["deskew_images_per_s"] = deskew["linear"]
            out["deskew"] = {"what": "sweep -> arg-max -> CONTAIN warp of every scan by its detected angle inside the timed "
                                     "region (omr_batch_deskew_device), same steps / warmup as `value`",
                             "linear_images_per_s": deskew["linear"], "nearest_images_per_s": deskew["nearest"],
                             "linear_over_value": deskew["linear"] / value, "nearest_over_value": deskew["nearest"] / value,
                             "warp_hbm_bytes_per_scan": float(ROWS * COLS) + deskew["canvas_bytes_per_scan_mean"]}
        kname = "omr::slane_kernel" if lanes_mode else "omr::runs_kernel" if n_runs > 0 else "omr::sweep_lds_kernel"
        # headline keys FIRST (the driver's parsed copy keeps the first ~20 keys of this object); strings, raw counters
        # and side figures go under "detail"
        roof = {"kernel": kname, "kernel_ms": kernel_ms, "bound": None, "frac": None, "achieved": None, "peak": None,
                "unit": None, "traffic": None, "hbm_frac": None, "valu_frac": None, "scalar_frac": None, "lds_frac": None,
                "wait_frac": None, "scans_per_launch": G, "kernel_ms_per_8_scans": kernel_ms * 8.0 / G,
                "algorithmic_GBps": algo_bytes / kernel_s / 1e9, "compulsory_bytes_per_launch": None}
        det = {"kernel_what": "scan-lane rotate+project: 64 scans per wavefront, geometry as wave-uniform programs, both "
                              "projections per launch" if lanes_mode
                              else "run-merging rotate+project, both projections per launch" if n_runs > 0 else "gather rotate+project",
               "launches_per_group": launches, "sweep_stage_ms_per_scan": stage_ms / G, "launch_groups_timed": k_n,
               "candidates_run_merged": 0 if lanes_mode else n_runs, "candidates_gathered": 0 if lanes_mode else n_gather,
               "candidates_scan_lane": A if lanes_mode else 0,
               "algorithmic_bytes_per_launch": algo_bytes, "algorithmic_bytes_per_scan": algo_bytes_scan,
               "algorithmic_note": "SURVEY 8(d) bytes (u8 image streamed once per candidate) / kernel time: a labelled "
                                   "side figure, not a roofline fraction -- the kernel reads a 1-bit/px image, 32 px per "
                                   "lane-operation"}
        roof["detail"] = det
        traffic, source = None, None
        sq = None
        if pmc_res and "passes" in pmc_res:
            ps = pmc_res["passes"]
            if "per_launch" in ps.get("fetch", {}) and "per_launch" in ps.get("write", {}):
                from oics import pmc
                f_kb = ps["fetch"]["per_launch"].get("FETCH_SIZE")
                w_kb = ps["write"]["per_launch"].get("WRITE_SIZE")
                if f_kb is not None and w_kb is not None:
                    traffic = pmc.hbm_bytes(f_kb, w_kb)
                    source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes "
                              "(3 steps each), mean per launch, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B")
                    det["hbm_counters_KB"] = {"FETCH_SIZE": f_kb, "WRITE_SIZE": w_kb}
            if "per_launch" in ps.get("sq", {}):
                sq = ps["sq"]
            errs = {k: v["error"] for k, v in ps.items() if "error" in v}
            if errs:
                det["pmc_errors"] = errs
        elif pmc_res:
            det["pmc_errors"] = pmc_res
        if traffic is None:
            tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tpath):
                try:
                    rec = json.load(open(tpath))
                    traffic = rec.get("sweep_kernel_hbm_bytes_per_launch")
                    source = "NOT measured in this run: copied from profiles/hbm_traffic.json (%s)" % rec.get("measured_at", "round 1")
                except Exception:  # noqa: BLE001
                    traffic = None
        roof["traffic"] = traffic
        det["traffic_source"] = source
        if traffic is not None:
            det["hbm_measured_GBps"] = traffic / kernel_s / 1e9
            roof["hbm_frac"] = traffic / kernel_s / 1e9 / HBM_PEAK_GBPS
            if lanes_mode:
                # what one launch has to move at least: every strip's program once PER QUAD OF SCAN GROUPS (a workgroup
                # sweeps four scan groups with one copy of the program; a launch of 512 scans reads the plan twice), the
                # interleaved bit images once, the row counts (u16 pairs, read-modify-write) and the counter dumps once
                nw = (COLS + 31) // 32
                used = (G + 63) // 64
                plan_reads = 1 if used <= 2 else (used + 3) // 4
                det["plan_bytes"] = float(batch.lanes_program_bytes())
                det["plan_reads_per_launch"] = plan_reads
                roof["compulsory_bytes_per_launch"] = float(batch.lanes_program_bytes()) * plan_reads + float(G) * (
                    ROWS * nw * 4 + A * ROWS * 2 * 2 + A * nw * 17 * 4)
            else:
                roof["compulsory_bytes_per_launch"] = float(G) * (ROWS * ((COLS + 127) // 128 * 16) + A * (ROWS + COLS) * 4)
            det["traffic_over_compulsory"] = traffic / roof["compulsory_bytes_per_launch"]
        if sq is not None:
            c = sq["per_launch"]
            t_prof = sq["kernel_us_profiled"] * 1e-6  # the counters belong to the profiled launch: use ITS duration
            valu_rate = c["SQ_INSTS_VALU"] / t_prof            # wave-instructions / s
            valu_peak = N_SIMD * CLOCK_GHZ * 1e9 / VALU_ISSUE_CYCLES  # hardware constant: 2 cycles per wave64 op
            lds_rate = c["SQ_LDS_IDX_ACTIVE"] / t_prof          # LDS-array cycles / s, summed over CUs
            lds_peak = N_CU * CLOCK_GHZ * 1e9
            # the scalar unit: one per CU, one SALU / SMEM instruction per cycle (SQ_ACTIVE_INST_SCA / SQ_BUSY_CU_CYCLES gives
            # the same fraction within 2 %, profiles/r04_pmc_sweep.md)
            sca_rate = (c["SQ_INSTS_SALU"] + c.get("SQ_INSTS_SMEM", 0.0)) / t_prof
            sca_peak = N_CU * CLOCK_GHZ * 1e9
            mix = VALU_LANE_MIX_CYCLES if lanes_mode else VALU_MIX_CYCLES
            roof.update(valu_frac=valu_rate / valu_peak, scalar_frac=sca_rate / sca_peak, lds_frac=lds_rate / lds_peak,
                        wait_frac=c["SQ_WAIT_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"]) if "SQ_WAIT_ANY" in c else None)
            det.update(valu_issue_frac_at_2_cycles=valu_rate / valu_peak,
                       valu_issue_frac_at_mix=valu_rate / (N_SIMD * CLOCK_GHZ * 1e9 / mix),
                       valu_mix_cycles=mix,
                       scalar_frac_of_busy_cu_cycles=(c["SQ_INSTS_SALU"] + c.get("SQ_INSTS_SMEM", 0.0)) / max(1.0, c.get("SQ_BUSY_CU_CYCLES", 0.0)),
                       lds_conflict_frac=c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]),
                       sq_counters_per_launch=c, kernel_us_under_profiler=sq["kernel_us_profiled"],
                       valu_insts_per_dst_word=c["SQ_INSTS_VALU"] * 64.0 / (G * A * ROWS * ((COLS + 31) // 32)),
                       scalar_insts_per_dst_word=(c["SQ_INSTS_SALU"] + c.get("SQ_INSTS_SMEM", 0.0)) * 64.0 / (G * A * ROWS * ((COLS + 31) // 32)),
                       peaks="frac / peak / bound are priced against hardware constants only (round-3 advice): VALU = %d SIMDs "
                             "x %.1f GHz / 2 cycles per wave64 op; SCALAR = %d CUs x %.1f GHz, one SALU / SMEM instruction per "
                             "cycle and CU; LDS = %d CUs x %.1f GHz array cycles; HBM = %.0f GB/s (MI355X_MICROARCH.md).  "
                             "`bound` names the LARGEST of these busy fractions; wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES says "
                             "how much of the waves' time is spent in s_waitcnt: when no fraction is near 1 the kernel is "
                             "issue- and latency-bound, not saturated.  valu_issue_frac_at_mix prices the same instruction "
                             "count at the issue cost of THIS kernel's opcodes (%.2f cycles: v_alignbit / v_bitop3 / v_bcnt "
                             "are 4-cycle VOP3 forms, profiles/r02_valu_issue.md) -- a side figure"
                             % (N_SIMD, CLOCK_GHZ, N_CU, CLOCK_GHZ, N_CU, CLOCK_GHZ, HBM_PEAK_GBPS, mix))
            cands = [("valu", roof["valu_frac"], valu_rate / 1e9, valu_peak / 1e9, "G wave-instr/s"),
                     ("scalar", roof["scalar_frac"], sca_rate / 1e9, sca_peak / 1e9, "G scalar-instr/s"),
                     ("lds", roof["lds_frac"], lds_rate / 1e9, lds_peak / 1e9, "G LDS-array cycles/s")]
            if traffic is not None:
                cands.append(("hbm", roof["hbm_frac"], det["hbm_measured_GBps"], HBM_PEAK_GBPS, "GB/s"))
            bnd = max(cands, key=lambda t: t[1])
            roof.update(bound=bnd[0], frac=bnd[1], achieved=bnd[2], peak=bnd[3], unit=bnd[4])
        elif traffic is not None:
            # no SQ counters: fall back to the measured HBM fraction (<= 1 by construction)
            roof.update(bound="hbm", achieved=det["hbm_measured_GBps"], peak=HBM_PEAK_GBPS, unit="GB/s",
                        frac=roof["hbm_frac"])
        else:
            roof.update(bound="hbm", achieved=None, peak=HBM_PEAK_GBPS, unit="GB/s", frac=None)
        out["roofline"] = roof
        if args.share_gpus:
            out["rehearsal"] = "--share-gpus: %d ranks on %d device(s), gloo gather; not a scaling measurement" % (world, torch.cuda.device_count())
        out["accuracy_ok"] = acc_ok
        out["gathered_results"] = int(all_best.numel())
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            # scan 0 (first quad of scan groups of the launch) and one scan of the last scan group
            picks = sorted({0, B - 1 - (B - 1) % 64 + min(44, (B - 1) % 64)})
            out["cpu_baseline"] = cpu_baseline([(i, cards[order[i]], vs[i].cpu().numpy(), hs[i].cpu().numpy()) for i in picks],
                                               args.cpu_seconds)
    batch.close()
    if rank == 0 and world == 1 and not args.no_e2e:
        del scans, vs, hs
        torch.cuda.empty_cache()
        try:
            e2e, e2e_best = e2e_host(cards, B, order)
            # the detected angles must be the HBM-resident run's (same cards, thresholded on the host instead of in the pack)
            e2e["agrees_with_resident_run"] = bool((e2e_best == np.array(all_best.cpu().tolist()[:B], np.int32)).all())
            out["e2e_host"] = e2e
            out["e2e_host_images_per_s"] = e2e["pageable_images_per_s"]
            out["e2e_host_pinned_images_per_s"] = e2e["pinned_images_per_s"]
            out["e2e_host_h2d_GBps"] = e2e["pinned_h2d_GBps"]
            out["e2e_host_packed_images_per_s"] = e2e["packed_images_per_s"]
            out["e2e_host_packed_h2d_GBps"] = e2e["packed_h2d_GBps"]
        except Exception as e:  # noqa: BLE001  (a side leg must never take the bench line down)
            out["e2e_host"] = {"error": repr(e)}
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if tmpdir:
        import shutil
        keep = os.path.join(ROOT, "gpurun_out", "bench_pmc")
        try:  # keep the raw counter files when running from a writable checkout (gpurun merges gpurun_out/ back)
            if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                shutil.rmtree(keep, ignore_errors=True)
                shutil.copytree(tmpdir, keep)
        except Exception:  # noqa: BLE001
            pass
        shutil.rmtree(tmpdir, ignore_errors=True)
    if rank == 0:
        print(json.dumps(out))
        # the bench fails on its own checks (round-4 verdict, weak 10): no `value` with exit code 0 from a kernel that
        # misses the reference's criterion (lib.rs:103-113) or differs from the oracle
        problems = []
        if out.get("accuracy_ok") is False:
            problems.append("accuracy_ok is false (a detected angle is >= 0.5 deg from the injected one)")
        if "cpu_baseline" in out and not out["cpu_baseline"].get("parity_vs_gpu", True):
            problems.append("cpu_baseline.parity_vs_gpu is false (GPU scores differ from the oracle's)")
        e2 = out.get("e2e_host", {})
        if e2 and "error" not in e2 and not (e2.get("agrees_with_resident_run", True) and e2.get("pinned_agrees", True) and e2.get("packed_agrees", True)):
            problems.append("e2e_host results differ from the HBM-resident run")
        if problems:
            raise SystemExit("bench.py: " + "; ".join(problems))


if __name__ == "__main__":
    main()
