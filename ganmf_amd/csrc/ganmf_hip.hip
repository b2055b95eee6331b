// libganmf_hip.so — handle, step orchestration and the C ABI declared in include/ganmf_hip.h.
//
// One handle = one GPU = one HIP stream (+ one RCCL communicator when data-parallel).  All state
// of a fit() lives in HBM: the CSR user x item matrix, every parameter with its Adam moments and
// best-snapshot twin, the minibatch work buffers.  The host sends one permutation per epoch and
// receives the per-minibatch losses; nothing else crosses PCIe inside the training loop.
//
// Layout in HBM.  Every matrix is row-major with a leading dimension rounded up to 64 floats
// (256-byte rows, whole K-tiles); pad columns of every buffer that is consumed along K are zero
// and stay zero (Adam on a zero gradient of a zero parameter is a fixed point; GEMM / reduce
// epilogues never store into pad columns).
// Bias folding: the encoder bias is stored as row N of We_ext [N+1, e] and the decoder bias as
// row e of Wd_ext [e+1, N]; minibatch matrices carry a ones column (XF[:, N] = 1, E[:, e] = 1).
// The bias add then rides inside the forward GEMMs and both bias gradients fall out of the weight
// gradient GEMMs as their last row, so biases need no kernels of their own (and Adam / L2 see
// them as part of the same tensor, exactly as the reference regularises biases, GANMF.py:131-132).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/ganmf_hip.h"
#include "gemm_f32.hpp"
#include "gemm_bf16s.hpp"
#include "gemm_bf16k.hpp"
#include "gemm_persist.hpp"
#include "gemm_bf16p.hpp"
#include "kernels.hpp"
#include "gemm_multi.hpp"
#ifdef GANMF_PERSIST_DIAG_BUILD
#include "wgrad_stream.hpp"      // experiment (profiles/r04_wgrad_stream.md)
#endif

using namespace ganmf;

namespace {
#include "lib/base.inc"
}  // namespace

#include "lib/handle.inc"

namespace {
#include "lib/support.inc"
#include "lib/dataparallel.inc"
#include "lib/gemm_run.inc"
#include "lib/step_ganmf.inc"
#include "lib/step_disganmf.inc"
#include "lib/epoch.inc"
}  // namespace

// =================================================================================================
extern "C" {
#include "lib/abi_core.inc"
#include "lib/abi_train.inc"
#include "lib/abi_score.inc"
#include "lib/abi_misc.inc"
}  // extern "C"
