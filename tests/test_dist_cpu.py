"""N > 1 path on CPU: 2 ranks over gloo exercise the image sharding and the detection gather used by bench.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_dets(image_idx, n):
    rng = np.random.default_rng(image_idx)
    d = rng.uniform(0, 300, (n, 5))
    d[:, 4] = np.sort(rng.uniform(0, 1, n))[::-1]
    return d, np.arange(n, dtype=np.int32) + 10 * image_idx


def _worker(rank, world, port, num_images, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mscnn_amd import dist as mdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gather = mdist.DetectionGather(max_det=16, device="cpu")
    mine = mdist.shard(num_images, rank, world)
    seen = []
    steps = (num_images + world - 1) // world
    for s in range(steps):
        if s < len(mine):
            dets, ids = _fake_dets(mine[s], 3 + 5 * mine[s])      # image 3 has 18 > max_det detections: tail cut
        else:
            dets, ids = np.zeros((0, 5)), np.zeros(0, np.int32)   # ragged last step: this rank has no image
        gather(dets, ids)
        seen.append(gather.result())
    dist.barrier()
    if rank == 0:
        torch.save(seen, out)
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    world, num_images = 2, 5
    out = str(tmp_path / "seen.pt")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, num_images, out), nprocs=world, join=True)
    seen = torch.load(out, weights_only=False)
    from mscnn_amd import dist as mdist
    assert mdist.shard(5, 0, 2) == [0, 2, 4] and mdist.shard(5, 1, 2) == [1, 3]
    got = {}
    for s, per_rank in enumerate(seen):
        for r, (dets, ids) in enumerate(per_rank):
            idx = s * world + r
            if idx < num_images:
                got[idx] = (dets, ids)
            else:
                assert len(dets) == 0
    assert sorted(got) == list(range(num_images))
    for idx, (dets, ids) in got.items():
        ref_d, ref_i = _fake_dets(idx, 3 + 5 * idx)
        n = min(len(ref_d), 16)
        assert len(dets) == n
        assert np.allclose(dets, ref_d[:n].astype(np.float32)) and np.array_equal(ids, ref_i[:n])
