// Entry points declared in include/vpx.h whose kernels are not written yet: they fail loudly (never a CPU fallback).
#include "vpx_internal.h"
using namespace vpx;
extern "C" {
size_t vpx_decouple_workspace_bytes(int, int, int, int) { return 0; }
int vpx_decouple_fwd(const float*, const float*, const float*, float*, int, int, int, int, void*, size_t, void*) {
    set_error("vpx_decouple_fwd: not implemented yet"); return VPX_ERR_UNSUPPORTED; }
int vpx_decouple_bwd(const float*, const float*, const float*, const float*, float*, float*, float*, int, int, int, int,
                     void*, size_t, void*) { set_error("vpx_decouple_bwd: not implemented yet"); return VPX_ERR_UNSUPPORTED; }
}
