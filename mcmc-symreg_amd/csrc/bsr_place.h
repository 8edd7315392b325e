// Which CPUs the library's threads of one rank run on: an L3 domain (CCX) of the host that belongs to the NUMA node of
// the rank's GPU.  Pure host logic over sysfs (no HIP): compiled into libbsr_hip.so (bsr_place.hip) and, on a faked
// sysfs tree, into tests/native/place_shim.cpp.
//
// The input block of every batch is written by a submission thread straight into device memory through the PCIe BAR
// and the results come back into pinned host memory: a rank whose threads sit on the far socket pays the inter-socket
// hop on every batch.  Dealing L3 domains by LOCAL_RANK over the CPU numbering (round 3) only lines up with the GPUs'
// sockets by luck.
#pragma once
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

namespace bsr_place {

// "0-7,128-135" -> CPU set; false when nothing parses
inline bool parse_cpulist(const char* txt, cpu_set_t* set) {
  CPU_ZERO(set);
  int n = 0;
  for (const char* p = txt; p && *p;) {
    while (*p == ',' || *p == ' ' || *p == '\n') ++p;
    if (*p < '0' || *p > '9') break;
    char* end = nullptr;
    long a = strtol(p, &end, 10), b = a;
    if (end && *end == '-') b = strtol(end + 1, &end, 10);
    for (long i = a; i <= b && i < CPU_SETSIZE; ++i) {
      CPU_SET((int)i, set);
      ++n;
    }
    p = end;
  }
  return n > 0;
}
inline bool read_line(const std::string& path, char* buf, size_t len) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  const bool ok = fgets(buf, (int)len, f) != nullptr;
  fclose(f);
  return ok;
}
inline bool l3_domain_of(const std::string& root, int cpu, cpu_set_t* set) {
  char buf[1024];
  return read_line(root + "/sys/devices/system/cpu/cpu" + std::to_string(cpu) + "/cache/index3/shared_cpu_list", buf, sizeof buf) &&
         parse_cpulist(buf, set);
}
// NUMA node of the PCI device `bdf` ("0000:05:00.0"), -1: unknown (no such file, or the platform reports -1)
inline int numa_of_device(const std::string& root, const char* bdf) {
  if (!bdf || !*bdf) return -1;
  std::string b(bdf);
  for (char& ch : b) if (ch >= 'A' && ch <= 'F') ch = (char)(ch - 'A' + 'a');   // sysfs names are lower case
  char buf[64];
  if (!read_line(root + "/sys/bus/pci/devices/" + b + "/numa_node", buf, sizeof buf)) return -1;
  return atoi(buf);
}

// The CPUs for local rank `lr` of `lw` whose GPU is `bdf`: among the allowed CPUs of the GPU's NUMA node (all allowed
// CPUs where the node is unknown or holds none of them), the L3 domains in CPU order; rank lr takes domain lr mod their
// number (ranks whose GPUs share a node sit next to each other in LOCAL_RANK order on the boxes this runs on), a single
// rank the domain it is running in if that one qualifies.  false: no placement (sysfs unreadable, fewer than 4 CPUs).
inline bool pick_cpus(const std::string& root, const char* bdf, const cpu_set_t& allowed, int lr, int lw, int cur_cpu,
                      cpu_set_t* out, int* numa_out) {
  const int numa = numa_of_device(root, bdf);
  if (numa_out) *numa_out = numa;
  cpu_set_t pool = allowed;
  if (numa >= 0) {
    char buf[4096];
    cpu_set_t node, both;
    if (read_line(root + "/sys/devices/system/node/node" + std::to_string(numa) + "/cpulist", buf, sizeof buf) &&
        parse_cpulist(buf, &node)) {
      CPU_AND(&both, &node, &allowed);
      if (CPU_COUNT(&both) >= 4) pool = both;
    }
  }
  std::vector<cpu_set_t> doms;
  cpu_set_t seen;
  CPU_ZERO(&seen);
  int cur_dom = -1;
  for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) {
    if (!CPU_ISSET(cpu, &pool) || CPU_ISSET(cpu, &seen)) continue;
    cpu_set_t dset;
    if (!l3_domain_of(root, cpu, &dset)) return false;
    CPU_OR(&seen, &seen, &dset);
    CPU_AND(&dset, &dset, &pool);
    if (cur_cpu >= 0 && CPU_ISSET(cur_cpu, &dset)) cur_dom = (int)doms.size();
    doms.push_back(dset);
  }
  if (doms.empty()) return false;
  size_t pickd;
  // GPU's node known: the node's domains in order, one per rank whose GPU sits there.  Unknown (sysfs says -1: common in
  // VMs and containers): the pool is every allowed CPU, and the ranks are dealt evenly over ALL its domains -- rank lr
  // mod their number would pack a node's ranks onto the first domains of socket 0.
  if (lw > 1 && lr >= 0) pickd = numa >= 0 ? (size_t)lr % doms.size() : ((size_t)lr * doms.size()) / (size_t)lw % doms.size();
  else pickd = cur_dom >= 0 ? (size_t)cur_dom : 0;
  *out = doms[pickd];
  return CPU_COUNT(out) >= 4;   // fewer is not worth it (and a submission thread needs a core of its own)
}

}  // namespace bsr_place
