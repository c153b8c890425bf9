"""The N>1 path on CPU: world_size-2 gloo run of the scatter(frames)/gather(detections) harness
(vti_amd.dataparallel) that bench.py uses over RCCL.  No GPU, no compute kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vti_amd import dataparallel as dp
        per, tail = 3, (8, 8, 3)
        frames = None
        if rank == 0:
            frames = torch.arange(world * per * 8 * 8 * 3, dtype=torch.int64).remainder(251).to(torch.uint8).view(world * per, *tail)
        shard = dp.scatter_frames(frames, per, tail, torch.device("cpu"))
        lo, hi = dp.shard_range(world * per, world, rank)
        expect = torch.arange(world * per * 8 * 8 * 3, dtype=torch.int64).remainder(251).to(torch.uint8).view(world * per, *tail)[lo:hi]
        ok = torch.equal(shard, expect)
        # each rank "detects": counts = frame index, dets filled with rank; rank r has 2 + r live mask slots of 5
        live = 2 + rank
        masks = torch.full((5, 8, 1), 200 + rank, dtype=torch.uint8)        # dead slots: must not travel
        masks[:live] = rank
        out = dict(dets=torch.full((per, 4, 38), float(rank)), counts=torch.arange(lo, hi, dtype=torch.int32),
                   xyxy=torch.full((per, 4, 4), float(rank)), masks=masks,
                   offsets=torch.tensor([0, 1, 2, live], dtype=torch.int32),
                   stats=torch.full((5, 5), rank, dtype=torch.int64), envelope=torch.full((per, 16), rank - 1, dtype=torch.int32))
        got = dp.gather_detections(out)
        if rank == 0:
            ok &= set(got) == {"dets", "counts", "xyxy", "offsets", "stats", "envelope"}       # compact payload: no mask buffer
            ok &= got["counts"].tolist() == list(range(world * per))
            ok &= got["dets"].shape == (world * per, 4, 38) and float(got["dets"][per:].min()) == 1.0
            ok &= got["stats"].shape == (world * 5, 5) and int(got["stats"][5:].min()) == 1
            ok &= got["envelope"].shape == (world * per, 16) and got["envelope"][per:].tolist() == [[0] * 16] * per
        else:
            ok &= got is None
        few = dp.gather_detections(dict(dets=out["dets"], counts=out["counts"], xyxy=out["xyxy"], offsets=out["offsets"]))
        ok &= (few is None) if rank else (set(few) == {"dets", "counts", "xyxy", "offsets"})
        lm = dp.gather_live_masks(out["masks"], out["offsets"])
        if rank == 0:
            buf, n_live = lm
            ok &= n_live.tolist() == [2, 3] and buf.shape == (5, 8, 1)
            ok &= buf[:2].eq(0).all().item() and buf[2:].eq(1).all().item()
        else:
            ok &= lm is None
        ok &= dp.max_over_ranks(10.0 + rank, torch.device("cpu")) == 10.0 + world - 1
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_shard_range_is_a_balanced_partition():
    from vti_amd.dataparallel import shard_range
    for total, world in ((512, 8), (64, 8), (10, 4), (3, 4), (0, 2)):
        parts = [shard_range(total, world, r) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == total
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in parts]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(120)
def test_scatter_gather_two_ranks_gloo(lib_built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(30)
    assert res == [(0, True), (1, True)]
