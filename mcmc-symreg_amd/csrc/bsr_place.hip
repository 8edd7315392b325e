// Where the library's own threads run (submission threads, the sampler's workers): one L3 domain of the host per local
// rank.  The caller's affinity is left alone unless BSR_PIN=1 asks for it.
#include "bsr_ctx.h"
#include "bsr_place.h"
#include <dirent.h>

std::atomic<bool> g_pinned{false};   // BSR_PIN=1: this process confined itself to the library's CPUs (choose_lib_cpus)
// CPU placement.  A batch passes through the caller, a submission thread and (native sampler) a worker thread; left to
// roam two sockets and sixteen L3 domains the pipelined step of the C2 bench measures anything from 17.9 to 21.6 us
// run by run, with those threads inside ONE L3 domain (a CCX: 8 cores and their SMT siblings on the EPYC hosts of
// MI355X boxes) 17.3 us every time (tools/probes/taskset_ab.sh).  The library therefore places ITS OWN threads
// (submission threads, sampler workers: bsr_internal_place_thread, called by each of them) on an L3 domain of the NUMA
// node of the context's GPU (bsr_place.h: /sys/bus/pci/devices/<bdf>/numa_node; a single rank stays in the domain it
// is running in when that one belongs to the node) -- with several ranks on the node (LOCAL_RANK / LOCAL_WORLD_SIZE) rank
// r takes domain r mod (domains of its GPU's node).  The CALLER's affinity is not touched: a drop-in library must not
// narrow the CPU set of the host application's later threads.  A process that wants the whole effect for itself
// (bench.py does) sets BSR_PIN=1: the calling thread is then confined too, once per process.  BSR_PIN=0: no placement
// at all; BSR_PIN_CPUS="0-7,128-135": this list instead of an L3 domain.
cpu_set_t g_lib_cpus;
std::atomic<bool> g_lib_cpus_ok{false};
int g_lib_numa = -2;   // NUMA node of the device the placement was made for (-1: unknown to sysfs, -2: no placement)
// `device`: the HIP device of the context being created (its PCI address names the NUMA node).  Called once per
// process, from the first bsr_ctx_create -- after the device count is known, so the runtime's own helper threads exist
// already: BSR_PIN=1 therefore confines EVERY thread of the process (/proc/self/task), not only the caller.
void choose_lib_cpus(int device) {
  static std::atomic<bool> done{false};
  if (done.exchange(true)) return;
  const int mode = env_int("BSR_PIN", -1);   // -1 (unset): the library's threads only; 0: nothing; 1: the caller as well
  if (mode == 0) return;
  cpu_set_t allowed, want;
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
  CPU_ZERO(&want);
  const char* list = getenv("BSR_PIN_CPUS");
  if (list && *list) {
    if (!bsr_place::parse_cpulist(list, &want)) return;
    CPU_AND(&want, &want, &allowed);
    if (CPU_COUNT(&want) < 4) return;
  } else {
    char bdf[64] = "";
    if (device >= 0 && hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) bdf[0] = 0;
    const int lr = env_int("LOCAL_RANK", -1), lw = env_int("LOCAL_WORLD_SIZE", env_int("WORLD_SIZE", 1));
    const char* root = getenv("BSR_SYSFS_ROOT");   // (test hook: a faked /sys)
    if (!bsr_place::pick_cpus(root ? root : "", bdf, allowed, lr, lw, sched_getcpu(), &want, &g_lib_numa)) return;
  }
  g_lib_cpus = want;
  g_lib_cpus_ok.store(true, std::memory_order_release);
  if (mode == 1) {
    bool any = false;
    if (DIR* dir = opendir("/proc/self/task")) {
      while (struct dirent* e = readdir(dir)) {
        const int tid = atoi(e->d_name);
        if (tid > 0 && sched_setaffinity(tid, sizeof want, &want) == 0) any = true;
      }
      closedir(dir);
    }
    if (any || sched_setaffinity(0, sizeof want, &want) == 0) g_pinned.store(true);
  }
}
// called by every thread the library starts: confines that thread (and nothing else) to the library's CPUs
__attribute__((visibility("hidden"))) void bsr_internal_place_thread() {
  if (!g_lib_cpus_ok.load(std::memory_order_acquire)) return;
  (void)pthread_setaffinity_np(pthread_self(), sizeof g_lib_cpus, &g_lib_cpus);
}

// CPUs of the set the library's threads are confined to (0: no placement): the sampler sizes its helper threads by it
__attribute__((visibility("hidden"))) int bsr_internal_placed_cpus() {
  return g_lib_cpus_ok.load(std::memory_order_acquire) ? CPU_COUNT(&g_lib_cpus) : 0;
}
