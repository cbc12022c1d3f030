"""N > 1 path on CPU: 2 ranks over gloo exercise the image sharding and the fixed-size detection-pack exchange of
mscnn_amd/dist.py (the same bytes libmscnn_dist.so moves with ncclAllGather on GPUs): what every rank receives must be
BIT-IDENTICAL to what the owning rank would have returned in a single-GPU run (float64 detections, int32 ROI rows), and a
pack written for more ROIs than the agreed capacity must be refused, not truncated."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAP = 40


def _fake_dets(image_idx, n):
    rng = np.random.default_rng(image_idx)
    d = rng.uniform(0, 300, (n, 5)) * np.pi          # full float64 mantissas: an fp32 round trip would not survive the check
    d[:, 4] = np.sort(rng.uniform(0, 1, n))[::-1]
    return d, np.arange(n, dtype=np.int32) + 10 * image_idx


def _host_pack(dets, ids, num_rois, cap):
    """The layout mscnn_net_detect_device writes (include/mscnn_net.h): [count, R, cap, 0 | cap x 5 f64 | cap x i32]."""
    sys.path.insert(0, ROOT)
    from mscnn_amd import net as mnet
    buf = np.zeros(mnet.detect_pack_bytes(cap), np.uint8)
    buf[:16].view(np.int32)[:] = [len(dets), num_rois, cap, 0]
    buf[16:16 + 40 * cap].view(np.float64)[:5 * len(dets)] = dets.reshape(-1)
    buf[16 + 40 * cap:16 + 44 * cap].view(np.int32)[:len(ids)] = ids
    return buf


def _worker(rank, world, port, num_images, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from mscnn_amd import dist as mdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gather = mdist.TorchGather(CAP, "cpu")
    mine = mdist.shard(num_images, rank, world)
    seen = []
    steps = (num_images + world - 1) // world
    for s in range(steps):
        if s < len(mine):
            dets, ids = _fake_dets(mine[s], 3 + 5 * mine[s])
            pack = _host_pack(dets, ids, len(dets) + 2, CAP)
        else:                                                     # ragged last step: this rank has no image
            pack = _host_pack(np.zeros((0, 5)), np.zeros(0, np.int32), 0, CAP)
        seen.append(gather(torch.from_numpy(pack)))
    dist.barrier()
    torch.save(seen, out + f".{rank}")
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_bit_identical(tmp_path):
    world, num_images = 2, 5
    out = str(tmp_path / "seen.pt")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, num_images, out), nprocs=world, join=True)
    from mscnn_amd import dist as mdist
    assert mdist.shard(5, 0, 2) == [0, 2, 4] and mdist.shard(5, 1, 2) == [1, 3]
    per_rank_view = [torch.load(out + f".{r}", weights_only=False) for r in range(world)]
    for seen in per_rank_view:                                    # every rank holds every image's detections
        got = {}
        for s, per_rank in enumerate(seen):
            for r, (dets, ids, R) in enumerate(per_rank):
                idx = s * world + r
                if idx < num_images:
                    got[idx] = (dets, ids, R)
                else:
                    assert len(dets) == 0 and R == 0
        assert sorted(got) == list(range(num_images))
        for idx, (dets, ids, R) in got.items():
            ref_d, ref_i = _fake_dets(idx, 3 + 5 * idx)           # = what the single-GPU run of image idx returns
            assert dets.dtype == np.float64 and dets.tobytes() == ref_d.tobytes()
            assert np.array_equal(ids, ref_i) and R == len(ref_d) + 2


def test_pack_capacity_is_enforced():
    from mscnn_amd import net as mnet
    d, i = _fake_dets(1, 6)
    pack = _host_pack(d, i, 9, CAP)
    dets, ids, R = mnet.unpack_detections(pack, CAP)
    assert dets.tobytes() == d.tobytes() and R == 9
    with pytest.raises(mnet.NetError, match="another capacity"):
        mnet.unpack_detections(pack[:mnet.detect_pack_bytes(8)].copy(), 8)     # reader and writer disagree on the capacity
    bad = pack.copy()
    bad[:16].view(np.int32)[:] = [CAP + 1, CAP + 1, CAP, 0]                      # more rows than the pack can hold
    with pytest.raises(mnet.NetError, match="corrupt detection pack"):
        mnet.unpack_detections(bad, CAP)
