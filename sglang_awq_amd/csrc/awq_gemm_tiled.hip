// Tiled MFMA fused AWQ GEMM for large M (prefill shapes).  Placeholder until the LDS-staged kernel
// lands: reports "unsupported" so awq_gemm falls through to the generic kernel.
#include "awq_kernels.h"

namespace awq {

bool tiled_supported(const GemmArgs&) { return false; }
int launch_gemm_tiled(const GemmArgs&) { return AWQ_ERR_BAD_VARIANT; }

}  // namespace awq
