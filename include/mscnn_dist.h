/*
 * mscnn_dist.h -- C ABI of libmscnn_dist.so: the ONE exchange step of multi-GPU MS-CNN inference.
 *
 * Images are independent (the only per-image coupling in the reference is BoxOutput's `for i < num` loop,
 * box_output_layer.cpp:107): image k runs on GPU k mod G with its own net replica -- one host thread (or process) per
 * GPU, mirroring the reference's thread-local Caffe singleton (src/caffe/common.cpp:13-20) -- and nothing but the final
 * detections ever crosses xGMI.  This library gathers the DEVICE-RESIDENT detection pack that
 * mscnn_net_detect_device() (include/mscnn_net.h) leaves behind: one fixed-size ncclAllGather per step straight out of /
 * into HBM, then one D2H copy of the gathered packs.  No fp32 narrowing, no host bounce before the collective; a rank
 * whose detections do not fit the agreed capacity is an ERROR, never a silent truncation.
 *
 * RCCL (librccl.so.1) is resolved at run time with dlopen, so the library loads on machines without it and does not
 * collide with an RCCL another component (e.g. PyTorch) already mapped.
 *
 * All functions return 0 on success; mscnn_dist_last_error() has the text otherwise.
 */
#ifndef MSCNN_DIST_H_
#define MSCNN_DIST_H_

#include <stddef.h>

#if defined(__GNUC__)
#define MSCNN_DIST_API __attribute__((visibility("default")))
#else
#define MSCNN_DIST_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mscnn_dist mscnn_dist;
#define MSCNN_DIST_ID_BYTES 128          /* sizeof(ncclUniqueId) */

MSCNN_DIST_API const char* mscnn_dist_last_error(void);

/* Optional, before the first other call of the process: resolve the collective library from this path instead of the default
 * search (librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so.1) -- an RCCL build in a non-standard place, or the transport stub
 * of the CPU tests (tests/stub/fake_rccl.c).  It must export ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather,
 * ncclAllReduce and ncclGetErrorString with RCCL's signatures. */
MSCNN_DIST_API int mscnn_dist_use_transport(const char* library_path);

/* Rank 0 creates the rendezvous id and hands the 128 bytes to every other rank by any out-of-band channel (the launcher's
 * store, a file, a socket); every rank then calls mscnn_dist_init with the same id. */
MSCNN_DIST_API int mscnn_dist_unique_id(unsigned char id_out[MSCNN_DIST_ID_BYTES]);

/* One communicator per GPU.  `device` must be the HIP device of the calling thread's net replica; pack_bytes is the fixed
 * per-rank message size (mscnn_net_detect_pack_bytes(cap)).  Collective: returns when all `world` ranks have called it. */
MSCNN_DIST_API int mscnn_dist_init(const unsigned char id[MSCNN_DIST_ID_BYTES], int rank, int world, int device,
                                   size_t pack_bytes, mscnn_dist** out);
MSCNN_DIST_API void mscnn_dist_destroy(mscnn_dist* d);
/* What the collective library reports about the communicator (ncclCommUserRank / ncclCommCount, queried in mscnn_dist_init and
 * required to equal the arguments) -- NOT an echo of what the caller asked for: a scaling run quotes these to show RCCL saw N ranks.
 * Every pack that goes through the exchanges below also carries its sender's rank in header word 3 (the final stage writes 0
 * there), so a receiver can check that slot r of the gathered buffer came from rank r. */
MSCNN_DIST_API int mscnn_dist_rank(const mscnn_dist* d);
MSCNN_DIST_API int mscnn_dist_world(const mscnn_dist* d);

/* The exchange: ncclAllGather of send_dev[pack_bytes] (device memory, e.g. the pack of mscnn_net_detect_device) on
 * `stream` into the communicator's device buffer, one asynchronous D2H copy of all `world` packs into the
 * communicator's pinned host buffer, then a wait on the stream.  *gathered_host = world * pack_bytes bytes, rank-major,
 * valid until the next call. */
MSCNN_DIST_API int mscnn_dist_all_gather(mscnn_dist* d, const void* send_dev, void* stream, const void** gathered_host);
/* Same without the D2H copy / wait: *gathered_dev = the device buffer (the caller orders later work on `stream`). */
MSCNN_DIST_API int mscnn_dist_all_gather_device(mscnn_dist* d, const void* send_dev, void* stream, const void** gathered_dev);
/* The same exchange pipelined: begin() copies the pack (device to device, on `stream`) and enqueues the ncclAllGather + the D2H copy
 * on the communicator's OWN stream behind it, then returns -- the compute stream goes on with the next image while the collective
 * runs; end() waits for the OLDEST exchange in flight and hands out its world * pack_bytes host bytes, valid ONLY UNTIL THE NEXT
 * begin() (two slots: with one exchange still in flight the next begin() re-uses the slot end() just handed out and overwrites these
 * bytes asynchronously -- consume or copy them first).  At most two exchanges may be in flight; every rank must call begin() / end() in the same order. */
MSCNN_DIST_API int mscnn_dist_all_gather_begin(mscnn_dist* d, const void* send_dev, void* stream);
MSCNN_DIST_API int mscnn_dist_all_gather_end(mscnn_dist* d, const void** gathered_host);
/* Barrier over the communicator (a 4-byte ncclAllReduce + stream wait): brackets the timed region of the benchmark. */
MSCNN_DIST_API int mscnn_dist_barrier(mscnn_dist* d, void* stream);

/* ---- host placement of the ranks (no reference counterpart: the reference's only multi-GPU code is the training-time P2PSync,
 * src/caffe/parallel.cpp; this belongs to the one-thread-per-GPU inference design above) ------------------------------------------
 * A rank's driver thread issues ~45 launches and one stream synchronisation per 4.5 ms frame: it must sit on the CPUs of its GPU's
 * NUMA node, and ranks that share a node must not share cores (their BLAS / OpenMP / RCCL helper threads inherit the mask).
 *
 * mscnn_dist_plan_cpus: pure planning.  local_cpulists[r] = the CPUs local to rank r's GPU in the kernel's list syntax
 *   ("0-15,128-143"; "" = unknown), allowed_cpulist = the CPUs this process may use (NULL / "" = the calling thread's current
 *   affinity mask).  Ranks with the same effective list (local AND allowed) share it in equal slices, cut inside every run of
 *   consecutive CPU numbers (so a core and its SMT sibling stay together); *out receives rank's slice in the same syntax.
 * mscnn_dist_pin_host_thread: reads <sysfs_root>/bus/pci/devices/<hipDeviceGetPCIBusId>/local_cpulist of the ranks' devices
 *   (rank r <-> device r when all `world` devices are visible to the process and device == rank; else only this process's device is
 *   known and its node is cut into `world` slices), plans as above inside the thread's current mask, and applies the result with
 *   sched_setaffinity to the calling thread (and, with MSCNN_DIST_PIN_PROCESS, to the threads the process already has); threads the
 *   calling thread creates later inherit it.  Call it
 *   first thing in a rank, before the net / communicator are created.  sysfs_root NULL = "/sys".  report (optional) receives one
 *   JSON object: rank, device, pci, numa_node, cpus, n_cpus, ranks_sharing_node, all_devices_visible, threads_pinned, threads. */
MSCNN_DIST_API int mscnn_dist_plan_cpus(const char* const* local_cpulists, int world, int rank, const char* allowed_cpulist,
                                        char* out, size_t out_bytes);
#define MSCNN_DIST_PIN_PROCESS 1u      /* one process per GPU: also move the threads the process already has (thread-per-GPU drivers: 0) */
MSCNN_DIST_API int mscnn_dist_pin_host_thread(int device, int rank, int world, unsigned flags, const char* sysfs_root, char* report,
                                              size_t report_bytes);

#ifdef __cplusplus
}
#endif
#endif
