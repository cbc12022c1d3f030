"""Image-parallel inference across the GPUs of one node.

The path shards by independent units: image k goes to rank k mod world, every rank owns a full net replica (weights are
0.31 GB) and there is NO activation or weight traffic between GPUs.  The only exchange is one all_gather per step of the
fixed-size DEVICE-RESIDENT detection pack the final stage leaves in HBM (mscnn_net_detect_device: [count, R, cap, 0 |
cap x 5 float64 | cap x int32], 16 + 44 cap bytes): lossless (float64 as computed), no host bounce before the collective,
and a rank whose ROI count exceeds the agreed capacity raises instead of truncating.

  RcclGather   -- the product path: libmscnn_dist.so (include/mscnn_dist.h) calls ncclAllGather directly (RCCL over xGMI);
                  the 128-byte ncclUniqueId travels over the launcher's torch.distributed store.
  TorchGather  -- the same exchange through torch.distributed.all_gather_into_tensor on a uint8 tensor: "nccl" (= RCCL) on
                  GPUs as a second route to the same bytes, "gloo" in the CPU tests of the sharding / pack logic.
"""
import ctypes as C
import os

import numpy as np

from . import net as mnet

_HERE = os.path.dirname(os.path.abspath(__file__))
DIST_LIB_PATH = os.path.join(_HERE, "libmscnn_dist.so")
_dlib = None


class DistError(RuntimeError):
    pass


def dist_lib():
    global _dlib
    if _dlib is None:
        if not os.path.exists(DIST_LIB_PATH):
            raise DistError(f"{DIST_LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(DIST_LIB_PATH)
        L.mscnn_dist_last_error.restype = C.c_char_p
        L.mscnn_dist_unique_id.argtypes = [C.c_void_p]
        L.mscnn_dist_use_transport.argtypes = [C.c_char_p]
        L.mscnn_dist_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
        L.mscnn_dist_destroy.argtypes = [C.c_void_p]
        L.mscnn_dist_destroy.restype = None
        L.mscnn_dist_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mscnn_dist_barrier.argtypes = [C.c_void_p, C.c_void_p]
        L.mscnn_dist_all_gather_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mscnn_dist_all_gather_end.argtypes = [C.c_void_p, C.c_void_p]
        L.mscnn_dist_world.argtypes = [C.c_void_p]
        L.mscnn_dist_rank.argtypes = [C.c_void_p]
        L.mscnn_dist_plan_cpus.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]
        L.mscnn_dist_pin_host_thread.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint, C.c_char_p, C.c_char_p, C.c_size_t]
        _dlib = L
    return _dlib


def _dcheck(rc):
    if rc != 0:
        raise DistError(dist_lib().mscnn_dist_last_error().decode())


def plan_cpus(local_cpulists, rank, allowed=None):
    """mscnn_dist_plan_cpus: the CPU slice (kernel list syntax, e.g. "0-3,128-131") of rank `rank` given every rank's GPU-local CPU
    list; ranks on the same NUMA node get disjoint slices inside `allowed` (default: this thread's affinity mask)."""
    world = len(local_cpulists)
    arr = (C.c_char_p * world)(*[s.encode() for s in local_cpulists])
    out = C.create_string_buffer(8192)
    _dcheck(dist_lib().mscnn_dist_plan_cpus(arr, world, rank, allowed.encode() if allowed else None, out, len(out)))
    return out.value.decode()


def parse_cpulist(text):
    cpus = set()
    for part in text.replace("\n", "").split(","):
        if part.strip():
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def pin_host_thread(device, rank, world, sysfs_root=None):
    """mscnn_dist_pin_host_thread: confine this process (all its threads, and every thread created later) to the rank's slice of the
    CPUs of its GPU's NUMA node.  Returns the library's report as a dict.  Call before the net / communicator exist."""
    import json
    rep = C.create_string_buffer(2048)
    _dcheck(dist_lib().mscnn_dist_pin_host_thread(device, rank, world, 1, sysfs_root.encode() if sysfs_root else None, rep, len(rep)))      # 1 = MSCNN_DIST_PIN_PROCESS
    return json.loads(rep.value.decode())


def shard(num_images, rank, world):
    """Indices of the images rank `rank` processes (round-robin, like the reference's one-image-per-forward loop)."""
    return list(range(rank, num_images, world))


def split_packs(gathered, world, cap, copy=True, stamped=False):
    """world concatenated packs (bytes-like) -> [(dets float64 [D,5], ids int32 [D], R)] per rank.  stamped: the packs went through
    libmscnn_dist.so, which writes the sender's communicator rank into header word 3 -- slot r must then carry rank r (anything else:
    the collective did not deliver what the line is about to claim, and this raises).  The layout is the one
    mscnn_net_detect_device writes (include/mscnn_net.h; C callers use mscnn_net_unpack_detections); here the rows are numpy
    VIEWS of the gathered buffer (copy=False: valid until the next gather) so that the per-step host cost stays a few us."""
    pb = mnet.detect_pack_bytes(cap)
    buf = np.frombuffer(gathered, np.uint8) if not isinstance(gathered, np.ndarray) else gathered.reshape(-1)
    assert buf.size == world * pb, (buf.size, world, pb)
    rows = max(cap, 1)
    out = []
    for r in range(world):
        p = buf[r * pb:(r + 1) * pb]
        D, R, c, who = (int(v) for v in p[:16].view(np.int32))
        if stamped and who != r:
            raise DistError(f"slot {r} of the gathered buffer carries a pack stamped by rank {who}")
        if c == cap and D == -1:      # the writer's overflow mark (mscnn_net_detect_device): raised on EVERY rank after the exchange
            raise DistError(f"rank {r}: {R} ROIs exceed the detection pack capacity {cap} (size it by BoxOutput's max_nms_num)")
        if c != cap or not (0 <= D <= R <= cap):
            raise DistError(f"rank {r}: corrupt detection pack ({D} detections, {R} ROIs, written for capacity {c}, read with {cap})")
        dets = p[16:16 + 40 * D].view(np.float64).reshape(D, 5)
        ids = p[16 + 40 * rows:16 + 40 * rows + 4 * D].view(np.int32)
        out.append((dets.copy(), ids.copy(), R) if copy else (dets, ids, R))
    return out


class RcclGather:
    """ncclAllGather of the device pack through libmscnn_dist.so.  `exchange_id(b)`: callable that returns rank 0's bytes on
    every rank (e.g. a torch.distributed broadcast_object_list or a TCPStore get/set)."""

    def __init__(self, rank, world, device, cap, exchange_id):
        L = dist_lib()
        self.cap, self.world, self.rank = cap, world, rank
        self.pack_bytes = mnet.detect_pack_bytes(cap)
        idb = (C.c_ubyte * 128)()
        # every rank takes part in the exchange whatever happened on rank 0, so that a failure there raises everywhere
        # instead of leaving the other ranks waiting
        mine = b""
        if rank == 0 and L.mscnn_dist_unique_id(idb) == 0:
            mine = bytes(idb)
        raw = exchange_id(mine)
        if len(raw) != 128:
            raise DistError("rank 0 could not create the RCCL rendezvous id: " + L.mscnn_dist_last_error().decode())
        idb = (C.c_ubyte * 128).from_buffer_copy(raw)
        self._h = C.c_void_p()
        _dcheck(L.mscnn_dist_init(idb, rank, world, device, self.pack_bytes, C.byref(self._h)))
        # what the collective library reports about the communicator it built (ncclCommCount / ncclCommUserRank), not what we asked for
        self.comm_world, self.comm_rank = L.mscnn_dist_world(self._h), L.mscnn_dist_rank(self._h)
        self.ranks_seen = set()          # senders' ranks found in the packs of every exchange so far (split_packs checks slot r == rank r)

    def _split(self, host):
        per_rank = split_packs(np.frombuffer(host, np.uint8), self.world, self.cap, copy=False, stamped=True)      # views of the pinned buffer
        if len(per_rank) != self.comm_world:
            raise DistError(f"{len(per_rank)} packs gathered, the communicator has {self.comm_world} ranks")
        self.ranks_seen.update(range(len(per_rank)))
        return per_rank

    def __call__(self, pack_dev_ptr, stream=None):
        """pack_dev_ptr: device address from Net.detect_device(cap, ...).  Returns the per-rank list of (dets, ids, R)."""
        out = C.c_void_p()
        _dcheck(dist_lib().mscnn_dist_all_gather(self._h, C.c_void_p(pack_dev_ptr), C.c_void_p(stream or 0), C.byref(out)))
        host = (C.c_ubyte * (self.world * self.pack_bytes)).from_address(out.value)
        return self._split(host)

    def begin(self, pack_dev_ptr, stream=None):
        """Pipelined form: enqueue this step's exchange on the communicator's own stream and return at once."""
        _dcheck(dist_lib().mscnn_dist_all_gather_begin(self._h, C.c_void_p(pack_dev_ptr), C.c_void_p(stream or 0)))

    def end(self):
        """Wait for the oldest exchange in flight; returns its per-rank list of (dets, ids, R) (zero-copy views of the pinned buffer: valid only until the next begin() -- copy what must live longer)."""
        out = C.c_void_p()
        _dcheck(dist_lib().mscnn_dist_all_gather_end(self._h, C.byref(out)))
        host = (C.c_ubyte * (self.world * self.pack_bytes)).from_address(out.value)
        return self._split(host)

    def barrier(self, stream=None):
        _dcheck(dist_lib().mscnn_dist_barrier(self._h, C.c_void_p(stream or 0)))

    def close(self):
        if self._h:
            dist_lib().mscnn_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TorchGather:
    """The same fixed-size byte exchange through torch.distributed (backend "nccl" = RCCL on GPUs, "gloo" on CPU)."""

    def __init__(self, cap, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.cap, self.group = cap, group
        self.world = dist.get_world_size(group)
        self.pack_bytes = mnet.detect_pack_bytes(cap)
        self.device = device
        self.recv = torch.zeros(self.world * self.pack_bytes, dtype=torch.uint8, device=device)

    def __call__(self, pack):
        """pack: uint8 tensor of pack_bytes on self.device (a view of the device pack, or a host pack in the CPU tests)."""
        assert pack.dtype == self.torch.uint8 and pack.numel() == self.pack_bytes
        self.dist.all_gather_into_tensor(self.recv, pack, group=self.group)
        return split_packs(self.recv.cpu().numpy(), self.world, self.cap)
