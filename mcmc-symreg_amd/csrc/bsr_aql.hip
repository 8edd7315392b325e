// Direct AQL dispatch: see bsr_aql.h.  Host code only (ROCr + a little of the HIP runtime API).
#include "bsr_aql.h"

#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <immintrin.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Queue {
  hsa_queue_t* q = nullptr;
  std::atomic_flag lock = ATOMIC_FLAG_INIT;   // a batch's packets stay contiguous, doorbell values ascend
  std::atomic<int> error{0};
};

}  // namespace

struct AqlDevice {
  int hip_device = -1;
  hsa_agent_t agent{};
  std::vector<hsa_executable_t> exes;
  std::vector<Queue*> queues;
  double us_per_tick = 0.0;   // of the dispatch timestamps (hsa_amd_profiling_get_dispatch_time)
  std::atomic<unsigned> next_queue{0};
  std::mutex mu;   // kernel table
  std::unordered_map<const void*, AqlKernel*> kernels;
  std::string err;
};

namespace {

std::mutex g_mu;
std::unordered_map<int, AqlDevice*> g_devices;
std::unordered_map<int, std::string> g_refused;   // device -> why not (asked once)
std::vector<unsigned char>* g_image = nullptr;     // this library's file: the code objects are loaded from it
std::vector<std::pair<size_t, size_t>> g_code_objects;   // (offset, size) of the gfx950 code objects in g_image

thread_local AqlBatch* tl_target = nullptr;
thread_local AqlDevice* tl_device = nullptr;

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// The fat binary of a HIP shared library is a sequence of clang offload bundles (section .hip_fatbin), one per
// translation unit: a 24-byte magic, the number of entries, then per entry (offset, size, length of the target string,
// the target string); offsets count from the bundle's first byte.
bool find_code_objects(std::string* why) {
  if (g_image) return !g_code_objects.empty();
  g_image = new std::vector<unsigned char>();
  Dl_info info;
  if (!dladdr(reinterpret_cast<void*>(&find_code_objects), &info) || !info.dli_fname) {
    *why = "dladdr did not name this library";
    return false;
  }
  FILE* f = fopen(info.dli_fname, "rb");
  if (!f) {
    *why = std::string("cannot read ") + info.dli_fname;
    return false;
  }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  g_image->resize(n > 0 ? (size_t)n : 0);
  const size_t got = n > 0 ? fread(g_image->data(), 1, (size_t)n, f) : 0;
  fclose(f);
  if (got != g_image->size() || got < 64) {
    *why = std::string("short read of ") + info.dli_fname;
    return false;
  }
  // (the magic, assembled here so that this file's own constants do not look like a bundle)
  char magic[25];
  snprintf(magic, sizeof magic, "%s%s%s", "__CLANG_", "OFFLOAD_", "BUNDLE__");
  const unsigned char* p = g_image->data();
  const size_t N = g_image->size();
  for (size_t at = 0; at + 32 <= N; ++at) {
    if (p[at] != '_' || memcmp(p + at, magic, 24) != 0) continue;
    uint64_t n_entries = 0;
    memcpy(&n_entries, p + at + 24, 8);
    if (n_entries == 0 || n_entries > 64) continue;
    size_t o = at + 32;
    for (uint64_t e = 0; e < n_entries && o + 24 <= N; ++e) {
      uint64_t off = 0, size = 0, tl = 0;
      memcpy(&off, p + o, 8);
      memcpy(&size, p + o + 8, 8);
      memcpy(&tl, p + o + 16, 8);
      o += 24;
      if (tl > 256 || o + tl > N) break;
      const std::string target(reinterpret_cast<const char*>(p + o), (size_t)tl);
      o += tl;
      if (size == 0 || at + off + size > N) continue;
      if (target.find("amdgcn") != std::string::npos && target.find("gfx950") != std::string::npos &&
          memcmp(p + at + off, "\177ELF", 4) == 0)
        g_code_objects.emplace_back(at + off, (size_t)size);
    }
  }
  if (g_code_objects.empty()) *why = "no gfx950 code object in this library's fat binary";
  return !g_code_objects.empty();
}

struct FindAgent {
  int domain, bus, dev;
  hsa_agent_t agent;
  bool found;
};
hsa_status_t agent_cb(hsa_agent_t a, void* data) {
  FindAgent* fa = static_cast<FindAgent*>(data);
  hsa_device_type_t t;
  if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
  uint32_t bdf = 0, domain = 0;
  if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
  if ((int)((bdf >> 8) & 0xFF) == fa->bus && (int)((bdf >> 3) & 0x1F) == fa->dev && (int)domain == fa->domain) {
    fa->agent = a;
    fa->found = true;
    return HSA_STATUS_INFO_BREAK;
  }
  return HSA_STATUS_SUCCESS;
}

void queue_error_cb(hsa_status_t status, hsa_queue_t*, void* data) {
  static_cast<Queue*>(data)->error.store((int)status ? (int)status : -1);
}

const char* hsa_msg(hsa_status_t st) {
  const char* m = nullptr;
  return (hsa_status_string(st, &m) == HSA_STATUS_SUCCESS && m) ? m : "unknown HSA status";
}

AqlDevice* open_device(int hip_device, std::string* why) {
  if (!env_int("BSR_AQL", 1)) {
    *why = "BSR_AQL=0";
    return nullptr;
  }
  if (!find_code_objects(why)) return nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, hip_device) != hipSuccess) {
    (void)hipGetLastError();
    *why = "hipGetDeviceProperties failed";
    return nullptr;
  }
  hsa_status_t st = hsa_init();   // (reference counted: the HIP runtime holds the first reference)
  if (st != HSA_STATUS_SUCCESS) {
    *why = std::string("hsa_init: ") + hsa_msg(st);
    return nullptr;
  }
  FindAgent fa{prop.pciDomainID, prop.pciBusID, prop.pciDeviceID, {}, false};
  (void)hsa_iterate_agents(agent_cb, &fa);
  if (!fa.found) {
    *why = "no ROCr agent at the HIP device's PCI address";
    return nullptr;
  }
  AqlDevice* d = new AqlDevice();
  d->hip_device = hip_device;
  d->agent = fa.agent;
  auto fail = [&](const std::string& m) -> AqlDevice* {
    *why = m;
    for (Queue* q : d->queues) {
      if (q->q) (void)hsa_queue_destroy(q->q);
      delete q;
    }
    for (hsa_executable_t e : d->exes) (void)hsa_executable_destroy(e);
    delete d;
    return nullptr;
  };
  for (const auto& co : g_code_objects) {
    hsa_code_object_reader_t reader;
    st = hsa_code_object_reader_create_from_memory(g_image->data() + co.first, co.second, &reader);
    if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_code_object_reader_create_from_memory: ") + hsa_msg(st));
    hsa_executable_t exe;
    st = hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe);
    if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_executable_create_alt: ") + hsa_msg(st));
    d->exes.push_back(exe);
    st = hsa_executable_load_agent_code_object(exe, d->agent, reader, nullptr, nullptr);
    if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_executable_load_agent_code_object: ") + hsa_msg(st));
    st = hsa_executable_freeze(exe, nullptr);
    if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_executable_freeze: ") + hsa_msg(st));
    (void)hsa_code_object_reader_destroy(reader);
  }
  // four queues, as many as the GPU has pipes to run them on (and as the HIP runtime spreads its streams over): with
  // more, queues that wait on a barrier hold up the ones behind them on their pipe -- 8.9 us/step with four, 16 with
  // six, 18 with eight (2 048 rows, six batches in flight; HIP streams over GPU_MAX_HW_QUEUES=8 the same)
  const int nq = std::max(1, std::min(16, env_int("BSR_AQL_QUEUES", 4)));
  for (int i = 0; i < nq; ++i) {
    Queue* q = new Queue();
    d->queues.push_back(q);
    st = hsa_queue_create(d->agent, 1024, HSA_QUEUE_TYPE_MULTI, queue_error_cb, q, UINT32_MAX, UINT32_MAX, &q->q);
    if (st != HSA_STATUS_SUCCESS) return fail(std::string("hsa_queue_create: ") + hsa_msg(st));
    // start / end timestamps of a dispatch land in its completion signal (only timed batches hang one on their row pass)
    (void)hsa_amd_profiling_set_profiler_enabled(q->q, 1);
  }
  uint64_t hz = 0;
  if (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &hz) == HSA_STATUS_SUCCESS && hz > 0) d->us_per_tick = 1e6 / (double)hz;
  return d;
}

}  // namespace

AqlDevice* aql_device(int hip_device, const char** err) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_devices.find(hip_device);
  if (it != g_devices.end()) return it->second;
  auto r = g_refused.find(hip_device);
  if (r == g_refused.end()) {
    std::string why;
    AqlDevice* d = open_device(hip_device, &why);
    if (d) {
      g_devices[hip_device] = d;
      return d;
    }
    r = g_refused.emplace(hip_device, why).first;
  }
  if (err) *err = r->second.c_str();
  return nullptr;
}

int aql_n_queues(AqlDevice* d) { return d ? (int)d->queues.size() : 0; }

const AqlKernel* aql_kernel(AqlDevice* d, const void* host_fn) {
  std::lock_guard<std::mutex> lk(d->mu);
  auto it = d->kernels.find(host_fn);
  if (it != d->kernels.end()) return it->second;
  AqlKernel* k = nullptr;
  const char* name = hipKernelNameRefByPtr(host_fn, nullptr);
  if (name && *name) {
    const std::string sym = std::string(name) + ".kd";
    for (hsa_executable_t exe : d->exes) {
      hsa_executable_symbol_t s;
      if (hsa_executable_get_symbol_by_name(exe, sym.c_str(), &d->agent, &s) != HSA_STATUS_SUCCESS) continue;
      AqlKernel t;
      if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &t.object) != HSA_STATUS_SUCCESS) continue;
      (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &t.kernarg_bytes);
      (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &t.lds_bytes);
      (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &t.scratch_bytes);
      if (t.object != 0) {
        k = new AqlKernel(t);
        break;
      }
    }
  } else {
    (void)hipGetLastError();
  }
  if (!k && env_int("BSR_AQL_VERBOSE", 0)) fprintf(stderr, "bsr aql: no kernel descriptor for %s\n", name ? name : "(unnamed function)");
  d->kernels[host_fn] = k;   // (a miss is remembered too: the batch falls back to the stream)
  return k;
}

int aql_slot_init(AqlDevice* d, AqlSlot* s, int index) {
  hsa_signal_t sig;
  // (polled by the host, never waited on through the runtime: no interrupt needed behind the packet)
  if (hsa_amd_signal_create(0, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &sig) != HSA_STATUS_SUCCESS) return -1;
  s->signal = sig.handle;
  hsa_signal_t sig2;
  if (hsa_signal_create(0, 0, nullptr, &sig2) != HSA_STATUS_SUCCESS) {   // (a plain one: its timestamps are read through the runtime)
    (void)hsa_signal_destroy(sig);
    s->signal = 0;
    return -1;
  }
  s->signal_row = sig2.handle;
  s->queue = index % (int)d->queues.size();   // (rewritten by every submission: chains go to the queues in turn)
  if (hipMalloc((void**)&s->d_kernarg, (size_t)BSR_AQL_MAX_PACKETS * BSR_AQL_KERNARG_BYTES) != hipSuccess) {
    (void)hipGetLastError();
    (void)hsa_signal_destroy(sig);
    (void)hsa_signal_destroy(sig2);
    s->signal = s->signal_row = 0;
    s->d_kernarg = nullptr;
    return -1;
  }
  return 0;
}

void aql_slot_destroy(AqlDevice*, AqlSlot* s) {
  if (s->signal) (void)hsa_signal_destroy(hsa_signal_t{s->signal});
  if (s->signal_row) (void)hsa_signal_destroy(hsa_signal_t{s->signal_row});
  s->signal_row = 0;
  if (s->d_kernarg) (void)hipFree(s->d_kernarg);
  s->signal = 0;
  s->d_kernarg = nullptr;
}

AqlBatch*& aql_target() { return tl_target; }
AqlDevice*& aql_target_device() { return tl_device; }

// Implicit kernel arguments (code object v5): a fixed block behind the explicit ones, 8-byte aligned; a kernel's
// kernarg segment is cut behind the last one it reads.
void aql_append(AqlBatch* b, AqlDevice* d, const void* host_fn, dim3 grid, dim3 block, unsigned dyn_lds, const unsigned char* args,
                size_t explicit_bytes) {
  const AqlKernel* k = d ? aql_kernel(d, host_fn) : nullptr;
  const size_t hidden_at = (explicit_bytes + 7) / 8 * 8;
  if (!k || b->n >= BSR_AQL_MAX_PACKETS || hidden_at + 256 > BSR_AQL_KERNARG_BYTES || k->kernarg_bytes < explicit_bytes ||
      k->kernarg_bytes > BSR_AQL_KERNARG_BYTES) {
    b->failed = true;
    return;
  }
  AqlBatch::Item& it = b->item[b->n];
  unsigned char* dst = b->args[b->n];
  memcpy(dst, args, explicit_bytes);
  memset(dst + explicit_bytes, 0, hidden_at + 256 - explicit_bytes);
  unsigned char* h = dst + hidden_at;
  const uint32_t bc[3] = {grid.x, grid.y, grid.z};
  const uint16_t gs[3] = {(uint16_t)block.x, (uint16_t)block.y, (uint16_t)block.z};
  memcpy(h + 0, bc, 12);      // hidden_block_count_{x,y,z}
  memcpy(h + 12, gs, 6);      // hidden_group_size_{x,y,z}; remainders (18..23) and global offsets (40..63) stay zero
  const uint16_t dims = 3;
  memcpy(h + 64, &dims, 2);   // hidden_grid_dims
  memcpy(h + 120, &dyn_lds, 4);   // hidden_dynamic_lds_size
  it.k = k;
  it.grid[0] = grid.x * block.x; it.grid[1] = grid.y * block.y; it.grid[2] = grid.z * block.z;
  it.block[0] = block.x; it.block[1] = block.y; it.block[2] = block.z;
  it.dyn_lds = dyn_lds;
  it.arg_bytes = (uint32_t)std::max<size_t>(k->kernarg_bytes, explicit_bytes);
  ++b->n;
}

void aql_stage_args(AqlDevice*, AqlSlot* s, const AqlBatch& b) {
  for (int i = 0; i < b.n; ++i)
    memcpy(s->d_kernarg + (size_t)i * BSR_AQL_KERNARG_BYTES, b.args[i], (b.item[i].arg_bytes + 15) / 16 * 16);
}

void aql_flush_writes(const void* last_device_word) {
  _mm_sfence();   // drain the write-combining buffers ...
  const volatile uint64_t* w = static_cast<const volatile uint64_t*>(last_device_word);
  (void)*w;       // ... and a read through the same mapping cannot pass the posted writes in front of it
  std::atomic_thread_fence(std::memory_order_seq_cst);
}

namespace {

// Reserves n consecutive packets of a queue; the caller fills them (headers last); the doorbell is rung when the
// reservation goes out of scope.  Held under the queue's lock: a batch's packets stay together, doorbell values ascend.
struct Reserve {
  Queue* Q;
  uint64_t first;
  int n;
  Reserve(Queue* q, int count) : Q(q), n(count) {
    first = hsa_queue_add_write_index_relaxed(Q->q, (uint64_t)n);
    while (first + (uint64_t)n - hsa_queue_load_read_index_scacquire(Q->q) > Q->q->size) _mm_pause();   // (1024 packets: never in practice)
  }
  void* packet(int i) const {
    return static_cast<unsigned char*>(Q->q->base_address) + ((first + (uint64_t)i) & (Q->q->size - 1)) * 64;
  }
  ~Reserve() { hsa_signal_store_screlease(Q->q->doorbell_signal, (hsa_signal_value_t)(first + (uint64_t)n - 1)); }
};

void write_dispatch(void* slot, const AqlBatch::Item& it, void* kernarg, uint64_t completion, bool barrier, int acq, int rel) {
  hsa_kernel_dispatch_packet_t* p = static_cast<hsa_kernel_dispatch_packet_t*>(slot);
  p->workgroup_size_x = (uint16_t)it.block[0];
  p->workgroup_size_y = (uint16_t)it.block[1];
  p->workgroup_size_z = (uint16_t)it.block[2];
  p->reserved0 = 0;
  p->grid_size_x = it.grid[0];
  p->grid_size_y = it.grid[1];
  p->grid_size_z = it.grid[2];
  p->private_segment_size = it.k->scratch_bytes;
  p->group_segment_size = it.k->lds_bytes + it.dyn_lds;
  p->kernel_object = it.k->object;
  p->kernarg_address = kernarg;
  p->reserved2 = 0;
  p->completion_signal.handle = completion;
  const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                     (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
  const uint16_t setup = (uint16_t)(3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
  __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
}

}  // namespace

// A batch is a chain: row pass -> k_solve -> residual pass (-> finalise, events).  A hardware queue runs its packets in
// order, and the GPU runs four queues side by side (its four compute pipes; the HIP runtime spreads its streams over
// four queues for the same reason): with a batch's chain in one queue -- what a HIP stream is -- four chains are in
// flight at most and the pipelined step costs a quarter of a chain's latency: 45 us / 4 = 11.3 us at C2, whatever the
// kernels' share of the machine (measured: the step follows latency / depth up to depth 4 and stops there, with HIP
// streams and with packets of our own alike; more queues than pipes make it worse; the row pass in a queue of its own
// with the tail behind a barrier-AND packet on its completion signal: worse again, 15 against 11 us; batches that wait
// together sent as one chain, level by level: no change at C2, where the CUs are what is short by then --
// profiles/r05_direct_dispatch_ab.txt).
int aql_submit(AqlDevice* d, AqlSlot* s, const AqlBatch& b, bool time_row) {
  if (b.n <= 0 || b.failed) return -1;
  // fences.  Results are read by the host: system-scope release behind the batch's last kernel.  Everything else at
  // agent scope, as the HIP runtime dispatches kernels of one stream: the data between the kernels stays on the device,
  // and the host's BAR stores (input block, kernel arguments) are in device memory before the doorbell rings
  // (aql_flush_writes) where an agent-scope acquire finds them.  A system-scope acquire in front of the row pass cost
  // 10 us per batch; system scope on every packet 14.0 against 8.8 us/step at 2 048 rows
  // (profiles/r05_direct_dispatch_ab.txt).
  const int qi = (int)(d->next_queue.fetch_add(1, std::memory_order_relaxed) % d->queues.size());   // chains go to the queues in turn
  Queue* Q = d->queues[qi];
  if (Q->error.load(std::memory_order_relaxed) != 0) return -1;
  s->queue = qi;
  hsa_signal_store_relaxed(hsa_signal_t{s->signal}, 1);
  if (time_row && b.n > 1) hsa_signal_store_relaxed(hsa_signal_t{s->signal_row}, 1);
  // A batch that would wrap around the end of the ring goes out as two reservations, each with its own doorbell write:
  // the hardware does not care, but a profiler's intercepting queue (rocprofv3 with --stats or --pmc) hands "the
  // packets written so far" to its tool as one contiguous block and ran off the end of the ring on the first batch
  // that straddled it.
  while (Q->lock.test_and_set(std::memory_order_acquire)) _mm_pause();
  const uint64_t at = hsa_queue_load_write_index_relaxed(Q->q) & (Q->q->size - 1);
  const int n_first = (int)std::min<uint64_t>((uint64_t)b.n, Q->q->size - at);
  for (int lo = 0; lo < b.n; lo += (lo == 0 ? n_first : b.n)) {
    const int cnt = (lo == 0) ? n_first : b.n - n_first;
    Reserve r(Q, cnt);
    for (int i = lo; i < lo + cnt; ++i) {
      const bool last = i == b.n - 1;
      const int acq = HSA_FENCE_SCOPE_AGENT;
      const int rel = last ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
      // the first packet depends on nothing in its queue (another slot's batch may be in front of it and run beside
      // it); every later one waits for the packets before it
      write_dispatch(r.packet(i - lo), b.item[i], s->d_kernarg + (size_t)i * BSR_AQL_KERNARG_BYTES,
                     last ? s->signal : ((i == 0 && time_row) ? s->signal_row : 0), i > 0, acq, rel);
    }
  }
  Q->lock.clear(std::memory_order_release);
  return 0;
}

// duration of the last batch's row pass by the packet processor's own timestamps, us (the batch was submitted with
// time_row and is complete); < 0 if the runtime has none
double aql_row_us(AqlDevice* d, AqlSlot* s, bool single_packet) {
  hsa_amd_profiling_dispatch_time_t t{};
  const hsa_signal_t sig{single_packet ? s->signal : s->signal_row};
  if (d->us_per_tick <= 0.0 || hsa_amd_profiling_get_dispatch_time(d->agent, sig, &t) != HSA_STATUS_SUCCESS || t.end < t.start) return -1.0;
  return (double)(t.end - t.start) * d->us_per_tick;
}

int aql_poll(AqlDevice* d, AqlSlot* s, const char** err) {
  if (hsa_signal_load_scacquire(hsa_signal_t{s->signal}) <= 0) return 0;
  int e = d->queues[s->queue]->error.load(std::memory_order_relaxed);
  if (e != 0) {
    if (err) *err = hsa_msg((hsa_status_t)e);
    return -1;
  }
  return 1;
}
