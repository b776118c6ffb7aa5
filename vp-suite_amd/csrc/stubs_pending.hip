// Entry points declared in include/vpx.h whose kernels are not written yet: they fail loudly (never a CPU fallback).
#include "vpx_internal.h"
using namespace vpx;
extern "C" {
size_t vpx_stlstm_workspace_bytes(const vpx_stlstm_desc*) { return 0; }
size_t vpx_stlstm_reserve_bytes(const vpx_stlstm_desc*) { return 0; }
int vpx_stlstm_step_fwd(const vpx_stlstm_desc*, const float*, const float*, const float*, const float*, const float*,
                        const float*, const float*, const float*, const float*, const float* const*, float*, float*,
                        float*, float*, float*, void*, size_t, void*, size_t, void*) {
    set_error("vpx_stlstm_step_fwd: not implemented yet"); return VPX_ERR_UNSUPPORTED; }
int vpx_stlstm_step_bwd(const vpx_stlstm_desc*, const float*, const float*, const float*, const float*, const float*,
                        const float*, const float*, const float*, const float*, const void*, size_t, const float*,
                        const float*, const float*, const float*, const float*, float*, float*, float*, float*, float*,
                        float*, float*, float*, float*, void*, size_t, void*) {
    set_error("vpx_stlstm_step_bwd: not implemented yet"); return VPX_ERR_UNSUPPORTED; }
size_t vpx_decouple_workspace_bytes(int, int, int, int) { return 0; }
int vpx_decouple_fwd(const float*, const float*, const float*, float*, int, int, int, int, void*, size_t, void*) {
    set_error("vpx_decouple_fwd: not implemented yet"); return VPX_ERR_UNSUPPORTED; }
int vpx_decouple_bwd(const float*, const float*, const float*, const float*, float*, float*, float*, int, int, int, int,
                     void*, size_t, void*) { set_error("vpx_decouple_bwd: not implemented yet"); return VPX_ERR_UNSUPPORTED; }
}
