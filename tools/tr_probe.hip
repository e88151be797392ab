// tools/tr_probe.hip — what ds_read_b64_tr_b16 (gfx950) returns: LDS holds a [16][16] u16 matrix with value = 16 * row + col; lane l = (i = l & 15,
// g = l >> 4) passes the address of the 4-element row piece (row 4 g + (i >> 2), columns 4 (i & 3) ..) and prints the 4 values it gets back.
// Build: hipcc -O2 --offload-arch=gfx950 tools/tr_probe.hip -o tools/_bin/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t *out) {
    __shared__ __attribute__((aligned(16))) uint16_t m[16 * 16];
    for (int k = threadIdx.x; k < 256; k += 64) m[k] = (uint16_t)k;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const uint16_t *p = &m[(4 * g + (i >> 2)) * 16 + 4 * (i & 3)];
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)p);
    uint64_t bits = __builtin_bit_cast(uint64_t, v);
    for (int j = 0; j < 4; j++) out[l * 4 + j] = (uint16_t)(bits >> (16 * j));
}
int main() {
    uint16_t *d, h[256];
    hipMalloc(&d, 512);
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) {
        printf("lane %2d (i=%2d g=%d):", l, l & 15, l >> 4);
        for (int j = 0; j < 4; j++) printf("  r%2d c%2d", h[l * 4 + j] / 16, h[l * 4 + j] % 16);
        printf("\n");
    }
    return 0;
}
