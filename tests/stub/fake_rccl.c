/* Test infrastructure (never shipped): a stand-in for librccl.so.1 with the eight entry points libmscnn_dist.so resolves, moving the
 * bytes through a file-backed shared mapping between the ranks' processes.  It lets the CPU tests drive the product's own
 * rendezvous-id exchange, communicator setup, all-gather and barrier at world size > 1 (tests/test_dist_cpu.py). */
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
#define SLOT (1 << 20)
struct shared { volatile int arrived; volatile int generation; int pad[14]; unsigned char slots[]; };
struct comm { struct shared* sh; int rank, world; size_t bytes; };
typedef struct comm* ncclComm_t;

static void barrier(struct comm* c) {
  const int gen = __atomic_load_n(&c->sh->generation, __ATOMIC_ACQUIRE);
  if (__atomic_add_fetch(&c->sh->arrived, 1, __ATOMIC_ACQ_REL) == c->world) {
    __atomic_store_n(&c->sh->arrived, 0, __ATOMIC_RELEASE);
    __atomic_add_fetch(&c->sh->generation, 1, __ATOMIC_ACQ_REL);
  } else {
    while (__atomic_load_n(&c->sh->generation, __ATOMIC_ACQUIRE) == gen) usleep(50);
  }
}
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  snprintf(id->internal, sizeof(id->internal), "/tmp/mscnn_fake_rccl_%d_%ld", (int)getpid(), (long)time(NULL));
  return 0;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
  if (world < 1 || world > 16 || rank < 0 || rank >= world) return 4;
  struct comm* c = calloc(1, sizeof(*c));
  c->rank = rank; c->world = world; c->bytes = sizeof(struct shared) + (size_t)world * SLOT;
  int fd = open(id.internal, O_RDWR | O_CREAT, 0600);
  if (fd < 0) return 2;
  if (ftruncate(fd, (off_t)c->bytes) != 0) return 2;
  c->sh = mmap(NULL, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->sh == MAP_FAILED) return 2;
  barrier(c);                       /* collective, like the real call */
  if (rank == 0) unlink(id.internal);
  *out = c;
  return 0;
}
#ifndef FAKE_COUNT_BIAS
#define FAKE_COUNT_BIAS 0      /* -DFAKE_COUNT_BIAS=1: a communicator that reports another size than it was asked for (negative test) */
#endif
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->world + FAKE_COUNT_BIAS; return 0; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return 0; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { if (c) { munmap(c->sh, c->bytes); free(c); } return 0; }
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, void* stream) {
  (void)t; (void)stream;
  if (count > SLOT) return 4;
  memcpy(c->sh->slots + (size_t)c->rank * SLOT, send, count);
  barrier(c);
  for (int r = 0; r < c->world; ++r) memcpy((char*)recv + (size_t)r * count, c->sh->slots + (size_t)r * SLOT, count);
  barrier(c);
  return 0;
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, void* stream) {
  (void)t; (void)op; (void)stream;
  if (count != 1) return 4;
  memcpy(c->sh->slots + (size_t)c->rank * SLOT, send, sizeof(int));
  barrier(c);
  int sum = 0;
  for (int r = 0; r < c->world; ++r) sum += *(int*)(c->sh->slots + (size_t)r * SLOT);
  *(int*)recv = sum;
  barrier(c);
  return 0;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : r == 2 ? "fake rccl: system error" : "fake rccl: invalid argument"; }
