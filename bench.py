#!/usr/bin/env python3
"""bench.py -- deskewed images/s of the projection-std-dev angle sweep on MI355X.

Workload (BASELINE.json configs[1], "C2"): 2480x3508 8-bit gray scans, +-10 deg @ 0.05 deg sweep
(400 candidates: the reference's half-open range, packages/lib/src/projection.rs:36-38).  One
"step" = one pass of the hot path (threshold-fused bit-pack -> fused rotate/project sweep ->
std-dev -> arg-max) over a batch of `--scans` synthetic cards that are resident in HBM before
the timed region starts.  N > 1: one process per GPU (torchrun), every rank sweeps its own shard
of the batch (weak scaling, no data-path collective); the only exchange is the gather of the
best indices after the last step.

Prints ONE JSON line (rank 0) with the contract's fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "omr-img-corrector_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROWS, COLS = 3508, 2480
MAX_ANGLE, STEP = 10, 0.05
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def make_scans(n, seed0, dev):
    """n distinct seeded cards -> [n, ROWS, COLS] u8 on `dev` + their injected skews."""
    from oics import synth
    cards, thetas = [], []
    for i in range(n):
        g, th = synth.make_card(ROWS, COLS, seed0 + i)
        cards.append(torch.from_numpy(g))
        thetas.append(th)
    return torch.stack(cards).to(dev), thetas


def cpu_baseline(gray, gpu_vs, gpu_hs, budget_s):
    """The oracle ("port" of the reference CPU path, oracle/oracle.c) timed on this box's host
    cores on a bounded sample of the same workload: whole 400-candidate sweeps of one scan,
    angle-parallel over all cores, repeated until about 2/3 of `budget_s` seconds are spent; then
    16-candidate slices on one thread for the rest.  Also the run's parity check: every candidate's
    scores must equal the GPU's bit for bit."""
    from oracle import oracle as orc
    orc.build()
    b = orc.threshold_binary(gray)
    Ms = orc.rotation_matrices(ROWS, COLS, MAX_ANGLE, STEP)
    A = Ms.shape[0]
    cores = min(os.cpu_count() or 1, 64)
    orc.sweep_matrices(b, Ms[:2], threads=1, want_proj=False, fast=True)  # warm-up
    sweeps, t_all, ok = 0, 0.0, True
    while sweeps < 1 or t_all < budget_s * 2.0 / 3.0:
        t0 = time.perf_counter()
        _, _, vs, hs = orc.sweep_matrices(b, Ms, threads=cores, want_proj=False, fast=True)
        t_all += time.perf_counter() - t0
        sweeps += 1
        ok = ok and bool((vs.view(np.uint64) == gpu_vs.view(np.uint64)).all()
                         and (hs.view(np.uint64) == gpu_hs.view(np.uint64)).all())
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
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scans", type=int, default=8, help="scans per GPU per step")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams per GPU; the sweep kernel fills the chip by itself, so 1 keeps the "
                         "per-kernel HIP-event / rocprof durations free of cross-stream overlap")
    ap.add_argument("--group", type=int, default=8, help="scans carried by one launch of each kernel")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-clock budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from oics import dist as odist
    from oics import projection
    import oics

    rank, local_rank, world = odist.init()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available() or oics.lib().omr_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    N, A = projection.candidate_count(MAX_ANGLE, STEP)
    B = args.scans
    scans, thetas = make_scans(B, 2 + rank * B, dev)  # seed 2 = C2's card
    best = torch.full((B,), -1, dtype=torch.int32, device=dev)
    vs = torch.zeros((B, A), dtype=torch.float64, device=dev)
    hs = torch.zeros((B, A), dtype=torch.float64, device=dev)
    batch = projection.Batch(ROWS, COLS, MAX_ANGLE, STEP, device=local_rank, n_streams=args.streams)
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
    elapsed = odist.barrier_max_seconds(time.perf_counter() - t0, dev)

    # the path's only exchange: gather of the per-scan results (outside the per-step loop, as in
    # a real batch job where it happens once)
    all_best = odist.gather_results(best, B * world, rank, world)

    # roofline leg: duration of the sweep stage of a launch group (G scans) from HIP events recorded on
    # the stream the kernels are launched on, over the same K steps.  The stage is `launches` kernel
    # launches: the run-merging kernel (G scans, both projections, one launch) plus one gather launch
    # per scan when some candidates do not qualify for run-merging; the dominant kernel's mean launch
    # time is stage / launches and one launch carries G / launches scans' algorithmic bytes.
    n_runs, n_gather = batch.info()
    launches = (1 if n_runs > 0 else 0) + (G if n_gather > 0 else 0)
    batch.set_timing(True)
    for _ in range(args.steps):
        step()
    k_sum_ms, k_n = batch.kernel_ms()
    batch.set_timing(False)
    stage_ms = k_sum_ms / max(1, k_n)
    kernel_ms = stage_ms / launches
    algo_bytes_scan = float(A) * ROWS * COLS  # binarised image streamed once per candidate
    algo_bytes = G * algo_bytes_scan / launches
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9

    total_scans = B * world * args.steps
    value = total_scans / elapsed
    out = None
    if rank == 0:
        detected = [(int(k) - N) * STEP for k in best.cpu().tolist()]
        acc_ok = all(abs(d - t) < 0.5 for d, t in zip(detected, thetas))  # lib.rs:103-113
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("sweep_kernel_hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "deskewed images/sec, 2480x3508 gray, +-10deg@0.05deg sweep; HBM GB/s vs roofline",
            "value": value,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "C2: 2480x3508 8-bit gray scan, +-10deg @ 0.05deg = %d candidates "
                                   "(reference half-open range), %d scans/GPU/step resident in HBM, "
                                   "projection-std-dev sweep (threshold fused)" % (A, B),
                       "scans_per_gpu_per_step": B, "candidates": A, "global_batch": B * world,
                       "parallelism": "scan-sharded x%d, host-side gather" % world, "streams_per_gpu": args.streams,
                       "scans_per_kernel_launch": G},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "omr::runs_kernel (run-merging rotate+project, both projections per launch)"
                                   if n_runs > 0 else "omr::sweep_lds_kernel (gather rotate+project)",
                         "kernel_ms": kernel_ms, "scans_per_launch": G, "launches_per_group": launches,
                         "sweep_stage_ms_per_scan": stage_ms / G,
                         "algorithmic_bytes_per_launch": algo_bytes, "algorithmic_bytes_per_scan": algo_bytes_scan,
                         "launch_groups_timed": k_n, "candidates_run_merged": n_runs, "candidates_gathered": n_gather},
            "accuracy_ok": acc_ok,
            "gathered_results": int(all_best.numel()),
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(scans[0].cpu().numpy(), vs[0].cpu().numpy(), hs[0].cpu().numpy(),
                                               args.cpu_seconds)
    batch.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
