// Direct AQL dispatch of a batch's kernels (bsr_aql.hip): the library writes the kernel-dispatch packets of a scoring
// batch into user-mode queues of its own (ROCr: hsa_queue_create) and rings the doorbell once per batch, instead of
// three to five hipLaunchKernel calls at ~2.9 us each -- which, serialised inside the HIP runtime, were what bounded the
// pipelined step ("issuing a batch's HIP calls").  The kernels are the same code objects: the loader reads
// the gfx950 code objects out of this library's own fat binary and loads them once per device through ROCr, and a
// launch site names its kernel by the host-side function pointer as before (bsr_launch below).
//
// Ordering: the packets of a batch go into one queue in launch order, every packet but the first with the barrier bit
// (it waits for everything in front of it in that queue): the dependencies of back-to-back launches on one HIP
// stream; chains go to the device's four queues in turn.  A batch's completion is an HSA signal on its last packet,
// polled by the waiter.  Nothing here orders against HIP streams: bsr_api.hip only takes this path for a slot whose stream is idle
// (BatchSlot::stream_dirty) and whose input block went to the device by BAR stores, and falls back to the stream otherwise.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <type_traits>

struct AqlKernel {   // what a dispatch packet needs to know about a kernel (from the loaded code object's descriptor)
  uint64_t object = 0;
  uint32_t kernarg_bytes = 0, lds_bytes = 0, scratch_bytes = 0;
};

#define BSR_AQL_MAX_PACKETS 8
#define BSR_AQL_KERNARG_BYTES 2048   // per packet: TileArgs<double> is 1232 bytes + 256 of implicit arguments

// One batch's packets, collected while issue_batch runs its launch functions, then written to a queue in one go.
struct AqlBatch {
  int n = 0;
  bool failed = false;
  struct Item {
    const AqlKernel* k;
    uint32_t grid[3], block[3], dyn_lds;
    uint32_t arg_bytes;   // explicit + implicit, as laid out in `args`
  } item[BSR_AQL_MAX_PACKETS];
  alignas(16) unsigned char args[BSR_AQL_MAX_PACKETS][BSR_AQL_KERNARG_BYTES];
};

struct AqlDevice;   // per HIP device: agent, code objects, queues
struct AqlSlot {    // per batch slot: completion signal, kernarg block in device memory
  uint64_t signal = 0;            // hsa_signal_t::handle: the batch's last packet
  uint64_t signal_row = 0;        // ... its row pass, for timed batches (the dispatch's start / end timestamps land in it)
  unsigned char* d_kernarg = nullptr;   // device memory, host-writable through the BAR: BSR_AQL_MAX_PACKETS * BSR_AQL_KERNARG_BYTES
  int queue = 0;                  // which of the device's queues the batch in flight went to
};

// nullptr + message in *err when direct dispatch is not available (no large BAR, ROCr refuses, code objects not found ...)
AqlDevice* aql_device(int hip_device, const char** err);
const AqlKernel* aql_kernel(AqlDevice* d, const void* host_fn);
int aql_slot_init(AqlDevice* d, AqlSlot* s, int index);
void aql_slot_destroy(AqlDevice* d, AqlSlot* s);
// copies the batch's kernel arguments into the slot's block (BAR stores; the caller fences and reads back once for this
// and its own input block: aql_flush_writes), then writes the packets and rings the doorbell
void aql_stage_args(AqlDevice* d, AqlSlot* s, const AqlBatch& b);
void aql_flush_writes(const void* last_device_word);
int aql_submit(AqlDevice* d, AqlSlot* s, const AqlBatch& b, bool time_row);
double aql_row_us(AqlDevice* d, AqlSlot* s, bool single_packet);
// 0: complete; 1: not yet; < 0: the queue reported an error (message in *err)
int aql_poll(AqlDevice* d, AqlSlot* s, const char** err);
int aql_n_queues(AqlDevice* d);

// The launch sites' side.  A thread that is collecting a batch (issue_batch sets the target around its launch calls)
// gets its launches appended to that batch; everywhere else this is hipLaunchKernelGGL.
AqlBatch*& aql_target();
AqlDevice*& aql_target_device();
void aql_append(AqlBatch* b, AqlDevice* d, const void* host_fn, dim3 grid, dim3 block, unsigned dyn_lds, const unsigned char* args,
                size_t explicit_bytes);

namespace bsr_aql_detail {
template <typename P, typename A>
inline void pack_one(unsigned char* buf, size_t& off, A&& a) {
  static_assert(std::is_trivially_copyable<P>::value, "kernel parameters are plain data");
  const P v = static_cast<P>(a);
  off = (off + alignof(P) - 1) / alignof(P) * alignof(P);
  memcpy(buf + off, &v, sizeof(P));
  off += sizeof(P);
}
}  // namespace bsr_aql_detail

template <typename... Params, typename... Args>
inline void bsr_launch(void (*kernel)(Params...), dim3 grid, dim3 block, unsigned dyn_lds, hipStream_t st, Args&&... args) {
  static_assert(sizeof...(Params) == sizeof...(Args), "one argument per kernel parameter");
  if (AqlBatch* b = aql_target()) {
    alignas(16) unsigned char buf[BSR_AQL_KERNARG_BYTES];
    size_t off = 0;
    (bsr_aql_detail::pack_one<Params>(buf, off, args), ...);
    aql_append(b, aql_target_device(), reinterpret_cast<const void*>(kernel), grid, block, dyn_lds, buf, off);
    return;
  }
  hipLaunchKernelGGL(kernel, grid, block, dyn_lds, st, static_cast<Params>(args)...);
}
