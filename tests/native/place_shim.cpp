// C shim around csrc/bsr_place.h for tests/test_place_host.py (CPU only): the library's CPU choice on a faked sysfs tree.
#include "../../mcmc-symreg_amd/csrc/bsr_place.h"

extern "C" {
// allowed: cpulist text; returns the number of CPUs picked (0: no placement) and writes them (ascending) to out[]
int place_pick(const char* root, const char* bdf, const char* allowed_list, int lr, int lw, int cur_cpu, int* out, int cap,
               int* numa) {
  cpu_set_t allowed, want;
  if (!bsr_place::parse_cpulist(allowed_list, &allowed)) return -1;
  CPU_ZERO(&want);
  if (!bsr_place::pick_cpus(root, bdf, allowed, lr, lw, cur_cpu, &want, numa)) return 0;
  int n = 0;
  for (int cpu = 0; cpu < CPU_SETSIZE && n < cap; ++cpu)
    if (CPU_ISSET(cpu, &want)) out[n++] = cpu;
  return n;
}
}
