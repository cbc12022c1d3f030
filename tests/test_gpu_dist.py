"""The multi-GPU exchange on real hardware, at the one world size a single-GPU box allows: libmscnn_dist.so's ncclAllGather
(RCCL, world = 1) of the device-resident detection pack must hand back exactly what the single-GPU path
(mscnn_net_detect) returns -- same float64 bytes, same ROI rows; and the C++ host driver (one thread + replica per GPU)
must run end to end.  World sizes > 1: tests/test_dist_cpu.py (pack / shard logic over gloo) and the driver's scaling run."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from mscnn_amd import dist as mdist, net as mnet, synth, zoo   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_gather_world1_is_bit_identical_to_single_gpu_detect():
    if not torch.cuda.is_available():
        pytest.fail("needs a MI355X")
    n = mnet.Net(prototxt_text=zoo.prototxt("kitti_car/mscnn-7s-576", height=192, width=640, max_nms_num=300))
    synth.load_into(n, "mid")
    kw = dict(cls_id=2, ratios=(192 / 375.0, 640 / 1242.0), org_hw=(375, 1242))
    cap = 300
    gather = mdist.RcclGather(0, 1, 0, cap, exchange_id=lambda b: b)
    assert gather.pack_bytes == mnet.detect_pack_bytes(cap) == (16 + 44 * cap + 15) // 16 * 16
    for seed in (1, 2, 3):
        n.set_blob("data", synth.frame(192, 640, seed=seed))
        n.forward()
        dets, ids, R = n.detect(**kw)
        (gd, gi, gR), = gather(n.detect_device(cap, **kw))
        assert gR == R and len(gd) == len(dets) > 0
        assert gd.tobytes() == dets.tobytes() and np.array_equal(gi, ids)
    # the pipelined form (exchange i on the communicator's own stream under image i + 1): same bytes, in order
    want = []
    for seed in (4, 5, 6, 7):
        n.set_blob("data", synth.frame(192, 640, seed=seed))
        n.forward()
        d_, i_, _ = n.detect(**kw)
        want.append((d_.copy(), i_.copy()))
        gather.begin(n.detect_device(cap, **kw))
        if len(want) >= 2:
            (gd, gi, _), = gather.end()
            assert gd.tobytes() == want[-2][0].tobytes() and np.array_equal(gi, want[-2][1])
    (gd, gi, _), = gather.end()
    assert gd.tobytes() == want[-1][0].tobytes() and np.array_equal(gi, want[-1][1])
    with pytest.raises(mdist.DistError, match="nothing in flight"):
        gather.end()
    gather.barrier()
    gather.close()
    # capacity below the ROI count is an error, not a truncation -- and (round 3) not an error on the overflowing rank ALONE, which
    # would leave the other ranks blocked in the collective: the pack is marked {-1, R, cap}, travels, and every rank's unpack raises
    _, _, R = n.detect(**kw)      # the ROI count of the frame the net holds now
    small = mdist.RcclGather(0, 1, 0, R - 1, exchange_id=lambda b: b)
    with pytest.raises(mdist.DistError, match="exceed the detection pack capacity"):
        small(n.detect_device(R - 1, **kw))
    small.close()


def test_bench_under_the_launcher_takes_the_distributed_path(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, RANK / MASTER_* in the environment), with the one
    rank a single-GPU box allows: process group, id exchange over the store, direct RCCL gather, barrier, max-over-ranks."""
    if not torch.cuda.is_available():
        pytest.fail("needs a MI355X")
    import json
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29671", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--model", "caltech/mscnn-7s-480", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 50 and d["config"]["gather"].startswith("libmscnn_dist"), d["config"]
    assert "pipelined" in d["config"]["gather"], d["config"]
    # the second route to the same bytes (taken by itself when the direct communicator cannot be set up), end to end
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29673", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--model", "caltech/mscnn-7s-480", "--no-cpu-baseline", "--no-alt", "--gather", "torch"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d2 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["config"]["gather"].startswith("torch.distributed") and d2["value"] > 50, d2["config"]
    assert abs(d2["config"]["mean_detections"] - d["config"]["mean_detections"]) < 1e-9      # the same packs came back
    from bench import _CudaPtr
    t = torch.arange(64, dtype=torch.uint8, device="cuda")
    v = torch.as_tensor(_CudaPtr(t.data_ptr(), 64), device="cuda")
    assert v.data_ptr() == t.data_ptr() and torch.equal(v, t)


def test_cpp_multi_gpu_host_driver(tmp_path):
    """mscnn_amd/detect_multi_gpu (host/tools/detect_multi_gpu.cpp): threads + net replicas + RCCL gather through the C ABIs."""
    if not torch.cuda.is_available():
        pytest.fail("needs a MI355X")
    exe = os.path.join(ROOT, "mscnn_amd/detect_multi_gpu")
    assert os.path.exists(exe), "detect_multi_gpu not built"
    proto = tmp_path / "deploy.prototxt"
    proto.write_text(zoo.prototxt("kitti_car/mscnn-7s-576", height=192, width=640, max_nms_num=300))
    r = subprocess.run([exe, str(proto), "--gpus", "1", "--images", "3", "--cap", "300"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("image")]
    assert len(lines) == 3 and "3 images on 1 GPU(s)" in r.stdout
    assert all(int(l.split("->")[1].split()[0]) > 0 for l in lines), r.stdout      # every frame yields detections


def test_bench_plain_gpus_n_refuses_on_a_box_with_fewer_devices():
    """`python bench.py --gpus N` with no launcher variables launches N ranks by itself; with fewer than N devices visible (this box has
    one) it must exit non-zero with a message and print no result line -- never n_gpus: 1."""
    if not torch.cuda.is_available():
        pytest.fail("needs a MI355X")
    import sys
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and f"--gpus {n} but only {n - 1} GPU(s) visible" in r.stderr, (r.returncode, r.stderr[-1000:])
    assert "n_gpus" not in r.stdout


def test_pin_host_thread_on_the_real_topology():
    """mscnn_dist_pin_host_thread on the box's own sysfs (round 6): GPU 0's PCI address and NUMA node are found, the CPUs it assigns are
    a non-empty subset of the mask the process had and of the GPU's local_cpulist, every thread of the process is moved, and as rank
    1 of 2 on the same device (a process that sees only its own GPU) it gets the OTHER half of the same node.  Run in a child process:
    the affinity of the test runner is not touched."""
    import json
    import sys
    code = r'''
import json, os, sys
sys.path.insert(0, %r)
from mscnn_amd import dist as md
before = sorted(os.sched_getaffinity(0))
rep = md.pin_host_thread(0, 0, 1)
after = sorted(os.sched_getaffinity(0))
bdf = rep["pci"]
try:
    local = open("/sys/bus/pci/devices/%%s/local_cpulist" %% bdf).read().strip()
except OSError:
    local = ""                      # (no sysfs entry visible in this sandbox: the planner then stays inside the given mask)
os.sched_setaffinity(0, before)
half = [md.pin_host_thread(0, r, 2) for r in (0,)]      # rank 0 of 2 on the one visible device: one slice of two
h0 = sorted(os.sched_getaffinity(0))
print(json.dumps({"before": before, "after": after, "rep": rep, "local": sorted(md.parse_cpulist(local)), "half0": h0, "rep_half": half[0]}))
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    rep = out["rep"]
    assert rep["device"] == 0 and rep["pci"].count(":") == 2 and rep["n_cpus"] == len(out["after"]) >= 1
    assert set(out["after"]) <= set(out["before"])
    if out["local"] and set(out["local"]) & set(out["before"]):
        assert set(out["after"]) <= set(out["local"])                      # on its GPU's NUMA node
    assert rep["threads_pinned"] == rep["threads"] >= 1
    assert sorted(mdist.parse_cpulist(rep["cpus"])) == out["after"]
    # two ranks on the node: rank 0 gets half of what one rank got (when there is more than one CPU to share)
    if len(out["after"]) >= 2:
        assert set(out["half0"]) < set(out["after"]) and out["rep_half"]["ranks_sharing_node"] == 2
