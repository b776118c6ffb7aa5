// Probe of ds_read_b64_tr_b16 semantics on gfx950 (development aid): checks the mapping described in
// cdna_hip_programming.md T10 against a known LDS image.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short M[64 * 64];  // M[row][col], pitch 64
    for (int i = threadIdx.x; i < 64 * 64; i += 64) M[i] = (short)((i / 64) * 100 + (i % 64));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, L = l & 15, q = L >> 2, p = L & 3;
    const int r0 = 5 + 7 * g, c0 = 16 * (g & 1);  // arbitrary row start per group, column block per group
    const short* addr = &M[(r0 + q) * 64 + c0 + 4 * p];
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)addr);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = r[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * sizeof(short));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, L = l & 15, r0 = 5 + 7 * g, c0 = 16 * (g & 1);
        for (int e = 0; e < 4; ++e) {
            const int want = (r0 + e) * 100 + c0 + L;
            if (h[l * 4 + e] != want) { if (bad < 8) printf("lane %d elem %d got %d want %d\n", l, e, h[l * 4 + e], want); ++bad; }
        }
    }
    printf("tr16_b64 mapping check: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK: lane i of a 16-lane group gets column c0+i of rows r0..r0+3", bad);
    return 0;
}
