// Internal view of the tensor-parallel communicator (tp_comm.hip) for the runner.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stddef.h>

struct FerrumHipComm;

namespace fh {
int comm_world(const FerrumHipComm* c);
// fp16 sum all-reduce in place on `s`: one-shot peer kernel when the message fits and the policy picks it, else RCCL.
int comm_all_reduce_f16(FerrumHipComm* c, __half* buf, size_t count, hipStream_t s);
bool comm_oneshot_fits(const FerrumHipComm* c, size_t count);
int comm_all_reduce_add_rms_norm_f16(FerrumHipComm* c, const __half* x, __half* residual, const __half* w, float eps, __half* norm_out,
                                     int rows, int dim, int* fused, hipStream_t s);
// small all-gather: `bytes` (multiple of 8) per rank → out[world][bytes] in rank order
// new give-ups of the one-shot transport since the last call (the transport is switched off when there are any)
unsigned comm_take_timeouts(FerrumHipComm* c);
int comm_all_gather_bytes(FerrumHipComm* c, const void* in, void* out, size_t bytes, hipStream_t s);
}  // namespace fh
