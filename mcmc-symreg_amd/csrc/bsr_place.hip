// Where the library's own threads run (submission threads, the sampler's workers): one L3 domain of the host per local
// rank.  The caller's affinity is left alone unless BSR_PIN=1 asks for it.
#include "bsr_ctx.h"

std::atomic<bool> g_pinned{false};   // BSR_PIN=1: this process confined itself to the library's CPUs (choose_lib_cpus)
// "0-7,128-135" -> CPU set; false when nothing parses
static bool parse_cpulist(const char* txt, cpu_set_t* set) {
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
static bool l3_domain_of(int cpu, cpu_set_t* set) {
  char path[128], buf[1024];
  snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
  FILE* f = fopen(path, "r");
  if (!f) return false;
  const bool ok = fgets(buf, sizeof buf, f) != nullptr;
  fclose(f);
  return ok && parse_cpulist(buf, set);
}
// CPU placement.  A batch passes through the caller, a submission thread and (native sampler) a worker thread; left to
// roam two sockets and sixteen L3 domains the pipelined step of the C2 bench measures anything from 17.9 to 21.6 us
// run by run, with those threads inside ONE L3 domain (a CCX: 8 cores and their SMT siblings on the EPYC hosts of
// MI355X boxes) 17.3 us every time (tools/probes/taskset_ab.sh).  The library therefore places ITS OWN threads
// (submission threads, sampler workers: bsr_internal_place_thread, called by each of them) on the L3 domain the
// context was created from -- with several ranks on the node (LOCAL_RANK / LOCAL_WORLD_SIZE) the domains of the
// allowed CPUs are dealt evenly by local rank.  The CALLER's affinity is not touched: a drop-in library must not
// narrow the CPU set of the host application's later threads.  A process that wants the whole effect for itself
// (bench.py does) sets BSR_PIN=1: the calling thread is then confined too, once per process.  BSR_PIN=0: no placement
// at all; BSR_PIN_CPUS="0-7,128-135": this list instead of an L3 domain.
cpu_set_t g_lib_cpus;
std::atomic<bool> g_lib_cpus_ok{false};
void choose_lib_cpus() {
  static std::atomic<bool> done{false};
  if (done.exchange(true)) return;
  const int mode = env_int("BSR_PIN", -1);   // -1 (unset): the library's threads only; 0: nothing; 1: the caller as well
  if (mode == 0) return;
  cpu_set_t allowed, want;
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
  CPU_ZERO(&want);
  const char* list = getenv("BSR_PIN_CPUS");
  if (list && *list) {
    if (!parse_cpulist(list, &want)) return;
  } else {
    const int lr = env_int("LOCAL_RANK", -1), lw = env_int("LOCAL_WORLD_SIZE", env_int("WORLD_SIZE", 1));
    if (lw > 1 && lr >= 0) {
      std::vector<cpu_set_t> doms;
      cpu_set_t seen;
      CPU_ZERO(&seen);
      for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) {
        if (!CPU_ISSET(cpu, &allowed) || CPU_ISSET(cpu, &seen)) continue;
        cpu_set_t dset;
        if (!l3_domain_of(cpu, &dset)) return;
        CPU_OR(&seen, &seen, &dset);
        doms.push_back(dset);
      }
      if (doms.empty()) return;
      const size_t nd = doms.size();
      want = doms[nd >= (size_t)lw ? ((size_t)lr * nd) / (size_t)lw : (size_t)lr % nd];
    } else {
      const int cpu = sched_getcpu();
      if (cpu < 0 || !l3_domain_of(cpu, &want)) return;
    }
  }
  CPU_AND(&want, &want, &allowed);
  if (CPU_COUNT(&want) < 4) return;   // not worth it (and a submission thread needs a core of its own)
  g_lib_cpus = want;
  g_lib_cpus_ok.store(true, std::memory_order_release);
  if (mode == 1 && sched_setaffinity(0, sizeof want, &want) == 0) g_pinned.store(true);
}
// called by every thread the library starts: confines that thread (and nothing else) to the library's CPUs
__attribute__((visibility("hidden"))) void bsr_internal_place_thread() {
  if (!g_lib_cpus_ok.load(std::memory_order_acquire)) return;
  (void)pthread_setaffinity_np(pthread_self(), sizeof g_lib_cpus, &g_lib_cpus);
}

