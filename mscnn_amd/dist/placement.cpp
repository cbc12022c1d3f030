// Host placement of the ranks of a multi-GPU run (include/mscnn_dist.h: mscnn_dist_plan_cpus / mscnn_dist_pin_host_thread).
//
// One host thread (or process) per GPU drives a 4.5 ms frame with ~45 kernel launches and one stream synchronisation; on a two-socket
// node a driver thread that runs on the other socket pays the cross-socket hop on every doorbell write, on the pinned detection pack
// and on the BoxOutput row-count read, and eight unpinned ranks' BLAS / OpenMP helper threads wander over each other's cores.  So each
// rank is confined to the CPUs of ITS GPU's NUMA node, and the ranks that share a node get disjoint slices of it.  The reference has
// nothing of the kind (its multi-GPU code is the training-only P2PSync, src/caffe/parallel.cpp): this belongs to the one-exchange
// inference design of mscnn_dist.h, not to a reference interface.
//
// Sources of truth: hipDeviceGetPCIBusId -> <sysfs>/bus/pci/devices/<bdf>/{local_cpulist, numa_node}; the calling thread's current
// affinity mask (a container / cgroup / taskset restriction is respected: the plan only ever narrows it).
#include <dirent.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mscnn_dist.h"

extern "C" void mscnn_dist_set_error_text(const char* text);      // dist.cpp (the thread's last-error buffer)

namespace {

void fail(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  mscnn_dist_set_error_text(buf);
}

// "0-15,128-143" -> sorted unique CPU numbers; false on a malformed list
bool parse_cpulist(const char* s, std::vector<int>* out) {
  out->clear();
  if (!s) return true;
  const char* p = s;
  while (*p) {
    while (*p == ' ' || *p == ',' || *p == '\n' || *p == '\t') ++p;
    if (!*p) break;
    if (!std::isdigit((unsigned char)*p)) return false;
    char* end = nullptr;
    const long a = std::strtol(p, &end, 10);
    long b = a;
    p = end;
    if (*p == '-') {
      ++p;
      if (!std::isdigit((unsigned char)*p)) return false;
      b = std::strtol(p, &end, 10);
      p = end;
    }
    if (a < 0 || b < a || b >= 1 << 16) return false;
    for (long c = a; c <= b; ++c) out->push_back((int)c);
  }
  std::sort(out->begin(), out->end());
  out->erase(std::unique(out->begin(), out->end()), out->end());
  return true;
}

std::string format_cpulist(const std::vector<int>& v) {
  std::string s;
  for (size_t i = 0; i < v.size();) {
    size_t j = i;
    while (j + 1 < v.size() && v[j + 1] == v[j] + 1) ++j;
    if (!s.empty()) s += ",";
    s += std::to_string(v[i]);
    if (j > i) s += "-" + std::to_string(v[j]);
    i = j + 1;
  }
  return s;
}

std::vector<int> intersect(const std::vector<int>& a, const std::vector<int>& b) {
  std::vector<int> r;
  std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(r));
  return r;
}

// Slice k of m of a CPU set, taken inside every run of consecutive CPU numbers: a node's list is usually
// "<cores>,<their SMT siblings>" (0-15,128-143), so cutting each run the same way keeps a core and its sibling in one slice.
std::vector<int> slice_of(const std::vector<int>& cpus, int k, int m) {
  std::vector<int> r;
  for (size_t i = 0; i < cpus.size();) {
    size_t j = i;
    while (j + 1 < cpus.size() && cpus[j + 1] == cpus[j] + 1) ++j;
    const size_t n = j - i + 1;
    for (size_t q = n * k / m; q < n * (k + 1) / m; ++q) r.push_back(cpus[i + q]);
    i = j + 1;
  }
  return r;
}

std::vector<int> current_mask() {
  std::vector<int> v;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) != 0) return v;
  for (int c = 0; c < CPU_SETSIZE; ++c)
    if (CPU_ISSET(c, &set)) v.push_back(c);
  return v;
}

bool read_small_file(const std::string& path, std::string* out) {
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[4096];
  const size_t n = std::fread(buf, 1, sizeof(buf) - 1, f);
  std::fclose(f);
  buf[n] = 0;
  *out = buf;
  while (!out->empty() && (out->back() == '\n' || out->back() == ' ')) out->pop_back();
  return true;
}

int plan(const std::vector<std::vector<int>>& local, int rank, const std::vector<int>& allowed, std::vector<int>* mine, int* sharers) {
  const int world = (int)local.size();
  // a rank without locality information (no sysfs entry, numa_node -1 with an empty list) counts as local to everything allowed
  std::vector<std::vector<int>> eff(world);
  for (int r = 0; r < world; ++r) {
    eff[r] = local[r].empty() ? allowed : intersect(local[r], allowed);
    if (eff[r].empty()) eff[r] = allowed;      // its node's CPUs are all outside the mask we were given: anything allowed
  }
  int k = 0, m = 0;
  for (int r = 0; r < world; ++r)
    if (eff[r] == eff[rank]) { if (r < rank) ++k; ++m; }
  *sharers = m;
  *mine = slice_of(eff[rank], k, m);
  if (mine->empty()) *mine = eff[rank];          // more ranks than CPUs on the node: share it
  return 0;
}

}  // namespace

extern "C" {

int mscnn_dist_plan_cpus(const char* const* local_cpulists, int world, int rank, const char* allowed_cpulist, char* out, size_t out_bytes) {
  if (!local_cpulists || world < 1 || rank < 0 || rank >= world || !out || out_bytes < 2) { fail("plan_cpus: bad argument"); return 1; }
  std::vector<std::vector<int>> local(world);
  for (int r = 0; r < world; ++r)
    if (!parse_cpulist(local_cpulists[r], &local[r])) { fail("plan_cpus: malformed CPU list of rank %d: '%s'", r, local_cpulists[r]); return 1; }
  std::vector<int> allowed;
  if (allowed_cpulist && allowed_cpulist[0]) {
    if (!parse_cpulist(allowed_cpulist, &allowed)) { fail("plan_cpus: malformed allowed list '%s'", allowed_cpulist); return 1; }
  } else allowed = current_mask();
  if (allowed.empty()) { fail("plan_cpus: no CPU allowed"); return 1; }
  std::vector<int> mine;
  int sharers = 0;
  plan(local, rank, allowed, &mine, &sharers);
  const std::string s = format_cpulist(mine);
  if (s.size() + 1 > out_bytes) { fail("plan_cpus: %zu bytes needed", s.size() + 1); return 1; }
  std::memcpy(out, s.c_str(), s.size() + 1);
  return 0;
}

int mscnn_dist_pin_host_thread(int device, int rank, int world, unsigned flags, const char* sysfs_root, char* report, size_t report_bytes) {
  if (world < 1 || rank < 0 || rank >= world || device < 0) { fail("pin_host_thread: bad argument (device %d, rank %d of %d)", device, rank, world); return 1; }
  const std::string root = sysfs_root && sysfs_root[0] ? sysfs_root : "/sys";
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) { fail("pin_host_thread: device %d of %d visible", device, ndev); return 2; }
  // rank r of this node drives device r (torch.distributed.run's LOCAL_RANK convention, detect_multi_gpu's threads) when all of them
  // are visible here; a process that sees only its own device knows only its own node and takes the rank-th of `world` slices of it
  const bool all_visible = device == rank && ndev >= world;
  std::vector<std::vector<int>> local(world);
  std::string my_bdf, my_node = "?";
  for (int r = 0; r < world; ++r) {
    const int dev = all_visible ? r : device;
    char bdf[64] = "";
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), dev) != hipSuccess) continue;
    for (char* c = bdf; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
    std::string text;
    if (read_small_file(root + "/bus/pci/devices/" + bdf + "/local_cpulist", &text)) parse_cpulist(text.c_str(), &local[r]);
    if (r == rank) {
      my_bdf = bdf;
      if (!read_small_file(root + "/bus/pci/devices/" + bdf + "/numa_node", &my_node)) my_node = "?";
    }
  }
  const std::vector<int> allowed = current_mask();
  if (allowed.empty()) { fail("pin_host_thread: sched_getaffinity failed"); return 1; }
  std::vector<int> mine;
  int sharers = 0;
  plan(local, rank, allowed, &mine, &sharers);
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : mine)
    if (c < CPU_SETSIZE) CPU_SET(c, &set);
  // the calling thread and -- one process per GPU (MSCNN_DIST_PIN_PROCESS) -- every thread the process already has (runtime helper
  // threads started before this call); threads created from now on by the calling thread inherit its mask
  int pinned = 0, threads = 0;
  if (flags & MSCNN_DIST_PIN_PROCESS)
   if (DIR* d = opendir("/proc/self/task")) {
    while (dirent* e = readdir(d)) {
      if (!std::isdigit((unsigned char)e->d_name[0])) continue;
      ++threads;
      if (sched_setaffinity((pid_t)std::atoi(e->d_name), sizeof(set), &set) == 0) ++pinned;
    }
    closedir(d);
  }
  if (sched_setaffinity(0, sizeof(set), &set) != 0) { fail("pin_host_thread: sched_setaffinity(%s) failed", format_cpulist(mine).c_str()); return 1; }
  if (report && report_bytes) {
    std::snprintf(report, report_bytes,
                  "{\"rank\": %d, \"device\": %d, \"pci\": \"%s\", \"numa_node\": \"%s\", \"cpus\": \"%s\", \"n_cpus\": %zu, "
                  "\"ranks_sharing_node\": %d, \"all_devices_visible\": %s, \"threads_pinned\": %d, \"threads\": %d}",
                  rank, device, my_bdf.c_str(), my_node.c_str(), format_cpulist(mine).c_str(), mine.size(), sharers,
                  all_visible ? "true" : "false", pinned, threads);
  }
  return 0;
}

}  // extern "C"
