// cell2_dev.h — device-side helpers shared by the operand-format kernels (cell2.hip, convq.hip): vector types, the split-bf16
// encoding, the LDS-DMA wrapper, the geometry of the 32x16-pixel activation stage image and of a K = 32 weight chunk.
#pragma once
#include "vpx_internal.h"

namespace vpx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int C2_HALO_W = 18;                   // 16 + 3 - 1
constexpr int C2_NPOS = 34 * C2_HALO_W;         // 612 halo positions of a 32x16 tile
constexpr int C2_PLANE_POS = 640;               // padded so that 4 planes are exactly 5 pieces per thread (512 threads)
constexpr int C2_PLANE = C2_PLANE_POS * 16;
constexpr int C2_ABUF = 4 * C2_PLANE;           // 40960 B
constexpr int C2_WCHUNK = 3 * 2 * 2 * 128 * 16; // 24576 B
constexpr int C2_LDS = 2 * C2_ABUF + 3 * C2_WCHUNK;  // 155648 B <= 160 KiB

// 16x16x32 form ("q form", cell2_kernel_q below): same activation stage image; one weight chunk = one K = 32 step
constexpr int CQ_WCHUNK = 2 * 2 * 4 * 64 * 16;  // 16384 B: [half = n >> 6][part][k group = tap half * 2 + channel half][n & 63][16 B], n = gate*32 + j
constexpr int CQ_LDS = 2 * C2_ABUF + 3 * CQ_WCHUNK;  // 131072 B (the epilogue's transposition space needs 8 x 16 KiB as well)

// Geometry of the q-form workgroup tile: NW waves own 4 * NW tile rows of 16 pixels.
//   NW = 8: the 32x16 tile, one workgroup per CU, three whole weight chunks in the ring.
//   NW = 4: the "half tile" (16x16 pixels, 256 threads, 80 KiB of LDS): TWO workgroups per CU, so that one's epilogue and
//           prologue run under the other's MFMAs; the weight ring holds two chunks and turns over in halves.
template <int NW>
struct CQGeom {
    static constexpr int NT = 64 * NW;                          // threads
    static constexpr int TH = 4 * NW;                           // tile rows
    static constexpr int NPOS = (TH + 2) * C2_HALO_W;           // halo positions (612 | 324)
    static constexpr int PLANE_POS = NW == 8 ? 640 : 384;       // padded: 4 planes = NPIECE pieces per thread exactly
    static constexpr int PLANE = PLANE_POS * 16;
    static constexpr int ABUF = 4 * PLANE;                      // 40960 | 24576 B
    static constexpr int NPIECE = 4 * PLANE_POS / NT;           // stage-copy DMAs per thread: 5 | 6
    static constexpr int WSLOTS = NW == 8 ? 3 : 2;
    static constexpr int WPIECE = CQ_WCHUNK / 2 / (NT * 16);    // DMAs per thread and half chunk: 1 | 2
    static constexpr int LDS = 2 * ABUF + WSLOTS * CQ_WCHUNK;   // 131072 | 81920 B (the epilogue needs NW x 16 KiB)
};

// q form, step schedule of a two-stage period (nine K = 32 steps): step p (0..8), lane half tsel (k groups 0,1 | 2,3) -> which stage of the
// period (0 even, 1 odd) and which tap; fragment addressing of the step (slot of a tap in the 18-wide halo image, base kind, offset)
__host__ __device__ constexpr int cq_stage_of(int p, int tsel) { return p < 4 ? 0 : (p == 4 ? tsel : 1); }
__host__ __device__ constexpr int cq_tap_of(int p, int tsel) { return p < 4 ? 2 * p + tsel : (p == 4 ? (tsel ? 0 : 8) : 2 * (p - 5) + 1 + tsel); }
__host__ __device__ constexpr int cq_slot(int tap) { return (tap / 3) * C2_HALO_W + tap % 3; }
__host__ __device__ constexpr int cq_kind(int p) { return p == 4 ? 2 : ((p == 1 || p == 7) ? 1 : 0); }   // slot distance tB - tA: 1 | 16 | other buffer
__host__ __device__ constexpr int cq_aoff(int p, int abuf) { return (p >= 5 ? abuf : 0) + cq_slot(cq_tap_of(p, 0)) * 16; }

__device__ const float c2_zero16[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};  // source of out-of-image pieces

__device__ __forceinline__ unsigned short c2_bf16_bits(float v) {
    __bf16 h = (__bf16)v;  // v_cvt_pk_bf16_f32: round to nearest even (same split as conv_gemm.hip's split_bf16)
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ void c2_split(float v, unsigned& hi, unsigned& lo) {
    const unsigned short h = c2_bf16_bits(v);
    hi = h;
    lo = c2_bf16_bits(v - __builtin_bit_cast(float, (unsigned)h << 16));
}

__device__ __forceinline__ void c2_dma16(const char* g, char* lds_wave_base) {
    // 64 lanes x 16 B: lane l's bytes land at lds_wave_base + 16*l (M0 = wave-uniform base).
    // Inline asm on purpose: while a compiler-visible LDS-DMA (__builtin_amdgcn_global_load_lds) is pending, hipcc's
    // waitcnt pass treats it as an access to BOTH memory and LDS and degrades every later wait to lgkmcnt(0) / vmcnt(0)
    // — the MFMA loop then waited for the fragments it had just requested (72 % matrix-pipe use by a lone wave, measured).
    // Hidden in asm, the loop's ds_reads get counted waits; the DMA's own completion is waited for by hand (C2_WAIT_VM)
    // at the sync points, and compiler-made vmcnt waits (epilogue loads) only become stricter by the extra queue entries.
    // M0 is an INPUT operand bound to the physical register ("{m0}"): the compiler emits the s_mov to m0 itself and knows
    // about it — nothing reserved is clobbered behind its back. The s_nop covers the "SALU writes M0 -> LDS-DMA" wait state,
    // which the hazard recognizer cannot see through the asm.
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(g), "{m0}"(__builtin_amdgcn_readfirstlane(lds)) : "memory");
}

__device__ __forceinline__ int c2_px(int i) { return (i & 16) ? ((i + 14) & 15) : (i & 15); }

#define C2_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ void c2_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


}  // namespace vpx
