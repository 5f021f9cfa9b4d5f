// examples/gather_poses_rccl.cpp -- the C++-side gather of BASELINE config 3: every rank (one process per GPU) registers its own
// shard of independent (scan, submap) pairs with pcm_align_batch, which leaves the packed pcm_result records in a DEVICE buffer the
// caller names; ONE ncclAllGather (RCCL over xGMI) per batch then makes every rank hold every pose.  The library owns no
// communicator: the collective, its stream and its order among the caller's other collectives stay
// with the host code -- here a ROS node would own `comm`.
//
// What a maintainer of the reference writes instead of the per-object loop of fast_gicp/src/align.cpp:61-99 when the pairs of a
// batch are spread over the GPUs of one node.  Compile check (no GPU needed): tests/test_adapter_compiles.py builds this file
// against /opt/rocm/include/rccl/rccl.h; it is not part of libpcm_amd.so.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <vector>

#include "pcm_amd.h"

// `ctxs`: this rank's n registration objects (targets and sources set); `guesses`: n x 16 floats; `all`: device buffer of
// world * n records; `stream`: the caller's stream for the collective.  Returns the library's status or -100 on an RCCL error.
int align_shard_and_gather(pcm_ctx* const* ctxs, int n, const float* guesses, ncclComm_t comm, int world, pcm_result* d_mine /* n records */,
                           pcm_result* d_all /* world * n records */, hipStream_t stream, std::vector<pcm_result>* host_all) {
  // 1. the hot path: all n Gauss-Newton loops advance on the device, results stay in HBM (device_out), no host copy requested
  const int rc = pcm_align_batch(ctxs, n, guesses, /*host_out=*/nullptr, /*device_out=*/d_mine);
  if (rc != PCM_OK && rc != PCM_ERR_NOT_CONVERGED) return rc;   // pcm_align_batch returns after its stream has drained: d_mine is complete
  // 2. one small collective per batch: 512 bytes per pair, latency-bound -- issue it once, not per pair
  if (ncclAllGather(d_mine, d_all, (size_t)n * sizeof(pcm_result), ncclChar, comm, stream) != ncclSuccess) return -100;
  // 3. (optional) the poses of every rank on the host: rank-major, pair-minor
  if (host_all) {
    host_all->resize((size_t)world * n);
    if (hipMemcpyAsync(host_all->data(), d_all, host_all->size() * sizeof(pcm_result), hipMemcpyDeviceToHost, stream) != hipSuccess) return PCM_ERR_HIP;
  }
  return hipStreamSynchronize(stream) == hipSuccess ? rc : PCM_ERR_HIP;
}
