// afe_aql.h -- a user-mode AQL queue of the engine's own for the resident step grid (afe_engine.cpp persist_*).
//
// Why not a HIP stream: a resident grid is a kernel that does not end while the host keeps authorising steps.  On a HIP
// stream every hipDeviceSynchronize / hipStreamSynchronize of the process (torch.cuda.synchronize, an allocator's implicit
// synchronisation, the host's own bracket around a block of steps) waits for that kernel to END -- so the grid has to be
// parked for each of them and started again for the next step: ~20-25 us per synchronised block, 1.3 us per step in
// blocks of 20 (round-3 review).  The runtime underneath HIP (ROCr) gives every process user-mode queues; a grid
// dispatched on a queue HIP does not know about is invisible to HIP's synchronisation, and the engine's own completion
// word (written after every worker's stores of a step) says when the authorised steps are done.  The kernel is the one
// HIP loaded (same code object, found through the loader extension by its symbol name): nothing is compiled twice.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace afe {

struct AqlKernel {
  uint64_t object = 0;          // kernel descriptor address (hsa_kernel_dispatch_packet_t::kernel_object)
  uint32_t kernarg_bytes = 0;   // size of the kernel-argument segment the code object declares
  uint32_t group_bytes = 0, private_bytes = 0;
};

struct AqlQueue;   // opaque

// nullptr when the runtime cannot be reached (why says so); the caller then stays with HIP launches
AqlQueue *aql_open(int hip_device, std::string *why);
// false: a dispatch was still in flight after 30 s (or the queue had failed under it) -- NOTHING was freed, and the caller
// must not free what that dispatch can reach either
bool aql_close(AqlQueue *q);
// the kernel behind a HIP __global__ function's host address, as loaded by HIP on this queue's device
bool aql_find_kernel(AqlQueue *q, const void *hip_host_function, AqlKernel *out, std::string *why);
// one dispatch of `workgroups` x `workgroup_size` work-items; kernarg is copied (bytes must equal k.kernarg_bytes).
// System-scope acquire at its start, system-scope release at its end.  One dispatch in flight per queue.
bool aql_dispatch(AqlQueue *q, const AqlKernel &k, const void *kernarg, size_t bytes, uint32_t workgroups, uint32_t workgroup_size, std::string *why);
bool aql_in_flight(const AqlQueue *q);
// waits for the dispatch in flight: 0 complete, 1 still running after timeout_us, -1 the queue reported an error (why)
int aql_wait(AqlQueue *q, uint64_t timeout_us, std::string *why);
// device time of the last completed dispatch in nanoseconds (0 when unknown)
uint64_t aql_last_duration_ns(const AqlQueue *q);
}  // namespace afe
