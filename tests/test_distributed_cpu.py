"""world_size-2 gloo test of the multi-GPU layout (oics/dist.py): round-robin sharding of a
batch and the gather of per-scan results -- the only exchange the path has.  The per-scan
work is done by the CPU oracle here (no GPU in this tier); bench.py uses the same functions with
the HIP path on each rank."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oics import dist as odist
    from oics import synth
    from oracle import oracle as orc
    r, lr, w = odist.init(backend="gloo")
    assert (r, w) == (rank, world)
    mine = odist.shard_indices(n, rank, world)
    local = []
    for i in mine:
        b, _ = synth.make_binary_card(96, 80, 100 + i)
        _, _, vs, hs = orc.sweep(b, 5, 0.5, want_proj=False)
        local.append(orc.argmax_path1(vs, hs)[0])
    full = odist.gather_results(torch.tensor(local, dtype=torch.int32), n, rank, world)
    t = odist.barrier_max_seconds(1.0 + rank, torch.device("cpu"))
    assert t == float(world)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), full.numpy())
    torch.distributed.destroy_process_group()


def test_shard_and_gather_two_ranks(tmp_path, oracle):
    sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
    from oics import dist as odist
    from oics import synth
    n = 5  # ragged: ranks get 3 and 2 scans
    assert odist.shard_indices(n, 0, 2) == [0, 2, 4] and odist.shard_indices(n, 1, 2) == [1, 3]
    assert odist.shard_indices(0, 0, 2) == []
    mp.spawn(_worker, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    expect = []
    for i in range(n):
        b, _ = synth.make_binary_card(96, 80, 100 + i)
        _, _, vs, hs = oracle.sweep(b, 5, 0.5, want_proj=False)
        expect.append(oracle.argmax_path1(vs, hs)[0])
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "r%d.npy" % r))
        assert got.tolist() == expect


def test_single_process_gather_is_identity():
    sys.path.insert(0, os.path.join(ROOT, "omr-img-corrector_amd"))
    from oics import dist as odist
    x = torch.arange(7, dtype=torch.int32)
    assert odist.gather_results(x, 7, 0, 1).tolist() == list(range(7))
