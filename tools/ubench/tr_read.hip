// What ds_read_b64_tr_b16 delivers (gfx950): image I[row][col] of 16-bit elements, value = 256 * row + col, 64-byte rows.
// Per 16-lane group: lane 4q + p supplies the address of row q, columns 4p .. 4p+3 of a 4 x 16 block; prints the four
// elements every lane receives.  Build: hipcc --offload-arch=gfx950 -O3 tr_read.hip -o tr_read.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long *out) {
    __shared__ short lds[64 * 32];
    for (int i = threadIdx.x; i < 64 * 32; i += 64) lds[i] = (short)(256 * (i / 32) + (i % 32));
    __syncthreads();
    const int lane = threadIdx.x, grp = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    // group g reads the block of rows 8 g .. 8 g + 3, columns 0 .. 15
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + (8 * grp + q) * 32 + 4 * p));
    out[lane] = __builtin_bit_cast(unsigned long long, v);
}
int main() {
    unsigned long long *d, h[64];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) { int v = (h[l] >> (16 * e)) & 0xffff; printf("  (r%d,c%d)", v / 256, v % 256); }
        printf("\n");
    }
    return 0;
}
