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


# ---- the product's own exchange code at world size 2 -------------------------------------------------------------------------------
# libmscnn_dist.so (mscnn_dist_unique_id / _init / _all_gather / _barrier) and mscnn_amd.dist.RcclGather, unchanged, in two
# processes: the collective library is the shared-memory stand-in tests/stub/fake_rccl.c (mscnn_dist_use_transport) and the HIP
# runtime calls land in tests/stub/fake_hip.c (LD_PRELOAD: "device" memory is host memory).  What this pins without a second GPU:
# the rendezvous-id hand-over, rank / world bookkeeping, buffer sizing and offsets of the gathered packs, the pinned-buffer views,
# the barrier, and that an over-capacity rank fails on EVERY rank after the exchange instead of hanging the others.
_RCCL_WORKER = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import ctypes as C
from mscnn_amd import dist as mdist
from test_dist_cpu import _fake_dets, _host_pack, CAP
rank, world, tmp, overflow, pipelined = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4] == "1", sys.argv[4] == "2"
assert mdist.dist_lib().mscnn_dist_use_transport({stub!r}.encode()) == 0
idfile = os.path.join(tmp, "id.bin")
def exchange(mine):                      # rank 0's bytes to every rank, through a file (any out-of-band channel does)
    if rank == 0:
        open(idfile + ".tmp", "wb").write(mine); os.rename(idfile + ".tmp", idfile)
        return mine
    for _ in range(600):
        if os.path.exists(idfile):
            return open(idfile, "rb").read()
        time.sleep(0.05)
    raise SystemExit("no id")
try:
    g = mdist.RcclGather(rank, world, 0, CAP, exchange)
except mdist.DistError as e:
    print("INITERROR", str(e)); sys.stdout.flush()
    raise SystemExit(0)
assert (g.comm_world, g.comm_rank) == (world, rank)      # from ncclCommCount / ncclCommUserRank, not an echo of the arguments
g.barrier()
seen = []
if pipelined:        # begin(i) ... begin(i + 1) ... end() -> step i: the product's double-buffered exchange on its own stream
    packs = []
    for step in range(5):
        img = step * world + rank
        dets, ids = _fake_dets(img, 2 + 3 * img)
        packs.append(_host_pack(dets, ids, len(dets) + 1, CAP))
        g.begin(packs[-1].ctypes.data)
        packs[-1][16:] = 0xAB                  # the caller may overwrite its pack right after begin(): begin() took a copy
        if step >= 1:
            seen.append([(d.copy(), i.copy(), R) for d, i, R in g.end()])
    seen.append([(d.copy(), i.copy(), R) for d, i, R in g.end()])
    try:
        g.end()
    except mdist.DistError as e:
        print("DISTERROR", str(e)); sys.stdout.flush()
    g.begin(packs[0].ctypes.data); g.begin(packs[1].ctypes.data)
    try:
        g.begin(packs[2].ctypes.data)
    except mdist.DistError as e:
        print("DISTERROR", str(e)); sys.stdout.flush()
    g.end(); g.end()
for step in range(0 if pipelined else 3):
    img = step * world + rank
    dets, ids = _fake_dets(img, 2 + 3 * img)
    pack = _host_pack(dets, ids, len(dets) + 1, CAP)
    if overflow and step == 2 and rank == 1:
        pack[:16].view(np.int32)[:] = [-1, CAP + 7, CAP, 0]      # what mscnn_net_detect_device writes for R > cap
    try:
        per_rank = g(pack.ctypes.data)
    except mdist.DistError as e:
        print("DISTERROR", step, str(e)); sys.stdout.flush()
        break
    seen.append([(d.copy(), i.copy(), R) for d, i, R in per_rank])
if seen and not overflow:
    assert g.ranks_seen == set(range(world)), g.ranks_seen      # every slot of every exchange carried its own rank's stamp
g.barrier()
g.close()
np.save(os.path.join(tmp, "seen%d.npy" % rank), np.array(seen, dtype=object), allow_pickle=True)
'''


def _build_stubs(tmp_path, rccl_flags=()):
    import subprocess
    src = os.path.join(ROOT, "tests", "stub")
    out = {}
    for name in ("fake_hip", "fake_rccl"):
        so = str(tmp_path / f"lib{name}.so")
        subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", so, os.path.join(src, name + ".c")] + (list(rccl_flags) if name == "fake_rccl" else []))
        out[name] = so
    return out


def _run_rccl_workers(tmp_path, overflow, mode=None, rccl_flags=()):
    import subprocess
    stubs = _build_stubs(tmp_path, rccl_flags)
    code = _RCCL_WORKER.format(root=ROOT, stub=stubs["fake_rccl"])
    env = dict(os.environ, LD_PRELOAD=stubs["fake_hip"])
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), "2", str(tmp_path), mode or ("1" if overflow else "0")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    return [so for so, _ in outs]


def test_rccl_gather_code_path_world2_bit_identical(tmp_path):
    _run_rccl_workers(tmp_path, overflow=False)
    for rank in range(2):
        seen = np.load(str(tmp_path / f"seen{rank}.npy"), allow_pickle=True)
        assert len(seen) == 3
        for step, per_rank in enumerate(seen):
            assert len(per_rank) == 2
            for r, (dets, ids, R) in enumerate(per_rank):
                img = step * 2 + r
                ref_d, ref_i = _fake_dets(img, 2 + 3 * img)
                assert dets.dtype == np.float64 and dets.tobytes() == ref_d.tobytes()      # float64 bit patterns, both ranks
                assert np.array_equal(ids, ref_i) and R == len(ref_d) + 1


def test_communicator_that_reports_another_size_is_refused(tmp_path):
    """mscnn_dist_world / _rank come from ncclCommCount / ncclCommUserRank: a transport whose communicator reports three ranks when two
    were asked for makes mscnn_dist_init fail on every rank -- the number a scaling line quotes is the library's, never the launcher's."""
    outs = _run_rccl_workers(tmp_path, overflow=False, rccl_flags=("-DFAKE_COUNT_BIAS=1",))
    for rank, so in enumerate(outs):
        assert f"INITERROR the communicator reports rank {rank} of 3, the caller asked for rank {rank} of 2" in so, so


def test_pack_from_the_wrong_rank_is_refused():
    """Every pack that went through libmscnn_dist.so carries its sender's rank in header word 3; split_packs(stamped=True) refuses a
    gathered buffer whose slot r was not written by rank r."""
    from mscnn_amd import dist as mdist
    a, b = _fake_dets(0, 3), _fake_dets(1, 4)
    packs = [_host_pack(a[0], a[1], 4, CAP), _host_pack(b[0], b[1], 5, CAP)]
    for r, p in enumerate(packs):
        p[:16].view(np.int32)[3] = r
    ok = mdist.split_packs(np.concatenate(packs), 2, CAP, stamped=True)
    assert [R for _, _, R in ok] == [4, 5]
    with pytest.raises(mdist.DistError, match="slot 0 of the gathered buffer carries a pack stamped by rank 1"):
        mdist.split_packs(np.concatenate(packs[::-1]), 2, CAP, stamped=True)


def test_rccl_gather_pipelined_world2_bit_identical(tmp_path):
    """mscnn_dist_all_gather_begin / _end: five steps with two exchanges in flight arrive in order and bit-identical on both
    ranks; end() with nothing in flight and a third begin() are refused."""
    outs = _run_rccl_workers(tmp_path, overflow=False, mode="2")
    for so in outs:
        assert "nothing in flight" in so and "two exchanges already in flight" in so, so
    for rank in range(2):
        seen = np.load(str(tmp_path / f"seen{rank}.npy"), allow_pickle=True)
        assert len(seen) == 5
        for step, per_rank in enumerate(seen):
            for r, (dets, ids, R) in enumerate(per_rank):
                img = step * 2 + r
                ref_d, ref_i = _fake_dets(img, 2 + 3 * img)
                assert dets.tobytes() == ref_d.tobytes() and np.array_equal(ids, ref_i) and R == len(ref_d) + 1


def test_rccl_gather_overflow_fails_on_every_rank(tmp_path):
    """One rank's image has more ROIs than the agreed capacity: it marks its pack and still takes part in the all-gather, so
    both ranks raise the same error after the exchange (round 2: only the overflowing rank failed, the other hung in the
    collective)."""
    outs = _run_rccl_workers(tmp_path, overflow=True)
    for so in outs:
        assert "DISTERROR 2 rank 1:" in so and "exceed the detection pack capacity" in so, so
    for rank in range(2):
        assert len(np.load(str(tmp_path / f"seen{rank}.npy"), allow_pickle=True)) == 2


# ---- bench.py's own launch path: a plain `python bench.py --gpus N` must measure N GPUs or fail loudly ----------
def _bench(args, env=None, timeout=300):
    import subprocess
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_n_without_enough_devices_fails_loudly():
    """No launcher in the environment, fewer GPUs visible than asked for (none in this container): non-zero exit and a message naming
    both counts -- never a silent n_gpus = 1 line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import torch as _t
    if _t.cuda.is_available() and _t.cuda.device_count() >= 2:
        pytest.skip("two real GPUs visible: the refusal path cannot be provoked here")
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], env=env)
    assert r.returncode != 0 and "--gpus 2 but only" in r.stderr and "refusing" in r.stderr, (r.returncode, r.stderr[-800:])
    assert "n_gpus" not in r.stdout


def test_bench_launched_with_another_world_size_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = _bench(["--gpus", "4", "--steps", "2", "--warmup", "1"], env=env)
    assert r.returncode != 0 and "--gpus 4 but the launcher started WORLD_SIZE=1" in r.stderr, r.stderr[-800:]
    assert "n_gpus" not in r.stdout


def test_bench_gpus_2_launches_itself_with_two_ranks(tmp_path):
    """`python bench.py --gpus 2 --launch-check` with NO launcher variables: bench.py re-executes itself under torch.distributed.run
    with two ranks, the ranks rendezvous, the product's RcclGather / libmscnn_dist.so runs its pipelined exchange at world 2 on the
    transport stub (tests/stub/fake_rccl.c over fake_hip.c), and rank 0 prints ONE JSON line with n_gpus = 2."""
    import json
    stubs = _build_stubs(tmp_path)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LD_PRELOAD"] = stubs["fake_hip"]
    r = _bench(["--gpus", "2", "--steps", "5", "--warmup", "1", "--launch-check", "--transport", stubs["fake_rccl"]], env=env)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launch_check"] is True and out["value"] is None
    assert "2 ranks in the communicator (ncclCommCount)" in out["config"]["gather"] and "pipelined" in out["config"]["gather"]
    assert out["config"]["comm_count"] == 2 and out["config"]["ranks_seen"] == [0, 1]


def test_bench_gpus_8_launch_check_with_host_placement(tmp_path):
    """The 8-rank launch nobody can rehearse on hardware here: `python bench.py --gpus 8 --launch-check` starts EIGHT
    ranks under torch.distributed.run, every rank takes its CPU slice from the product's planner (mscnn_dist_plan_cpus) on a synthetic
    two-socket topology laid over this container's CPUs -- ranks 0-3 local to the first half, 4-7 to the second -- applies it, and the
    pipelined exchange of libmscnn_dist.so runs at world 8 on the transport stub.  The line: n_gpus 8, eight senders seen in every
    exchange, eight disjoint CPU slices, each inside its socket's half."""
    import json
    stubs = _build_stubs(tmp_path)
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 8:
        pytest.skip("fewer than 8 CPUs: eight disjoint slices do not exist")
    half = len(cpus) // 2

    def fmt(v):
        return ",".join(str(c) for c in v)
    lists = [fmt(cpus[:half])] * 4 + [fmt(cpus[half:])] * 4
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LD_PRELOAD"] = stubs["fake_hip"]
    env["OMP_NUM_THREADS"] = "1"
    r = _bench(["--gpus", "8", "--steps", "6", "--warmup", "1", "--launch-check", "--transport", stubs["fake_rccl"],
                "--fake-cpulists", ";".join(lists)], env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["comm_count"] == 8 and out["config"]["ranks_seen"] == list(range(8))
    per = out["config"]["per_rank"]
    assert [p["rank"] for p in per] == list(range(8))
    seen = set()
    for p in per:
        mine = set(p["applied"])
        assert mine and not (mine & seen), (p, seen)                     # disjoint
        assert mine <= set(cpus[:half] if p["rank"] < 4 else cpus[half:]), p   # inside its own "socket"
        seen |= mine


def test_plan_cpus_two_socket_node():
    """mscnn_dist_plan_cpus on the topology the 8-GPU node is expected to have (4 GPUs per socket, SMT siblings listed as a second run):
    every rank gets cores AND their siblings of its own socket, slices are disjoint and cover the node; an `allowed` mask narrower than
    the node (container) is respected; unknown locality falls back to the allowed set; more ranks than CPUs share."""
    from mscnn_amd import dist as md
    node = ["0-63,128-191"] * 4 + ["64-127,192-255"] * 4
    got = [md.parse_cpulist(md.plan_cpus(node, r, allowed="0-255")) for r in range(8)]
    assert got[0] == set(range(0, 16)) | set(range(128, 144)) and got[5] == set(range(80, 96)) | set(range(208, 224))
    assert all(len(g) == 32 for g in got) and set().union(*got) == set(range(256))
    assert sum(len(g) for g in got) == 256                                  # disjoint
    inside = [md.parse_cpulist(md.plan_cpus(node, r, allowed="0-31")) for r in range(8)]
    assert all(g <= set(range(32)) and g for g in inside)
    # (the second socket's ranks find none of their CPUs in the mask and fall back to it: all eight then share 0-31, four CPUs each)
    assert inside[0] == {0, 1, 2, 3} and inside[7] == {28, 29, 30, 31} and sum(len(g) for g in inside) == 32
    assert md.parse_cpulist(md.plan_cpus(["", ""], 1, allowed="0-7")) == {4, 5, 6, 7}
    assert md.parse_cpulist(md.plan_cpus(["0-1"] * 4, 3, allowed="0-1")) == {1} or md.parse_cpulist(md.plan_cpus(["0-1"] * 4, 3, allowed="0-1")) == {0, 1}
    with pytest.raises(md.DistError):
        md.plan_cpus(["0-3", "x"], 0, allowed="0-7")
