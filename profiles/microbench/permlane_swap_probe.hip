// What v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950) do to the four rows of 16 lanes of a wave, and the sequence the tile
// kernel's table form uses to turn "lane L holds the weight of position L" into "every row holds chunk c's 16 weights" (tab_spread in
// tools/gen_tile_dense_asm.py). hipcc --offload-arch=gfx950 -O2 permlane_swap_probe.hip -o permlane_swap_probe && ./permlane_swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out) {
    unsigned g = threadIdx.x; // lane L holds L
    unsigned w0, w1, w2, w3;
    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\ts_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\t"
                 "v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %2\n\t"
                 "v_permlane32_swap_b32 %1, %3\n\t"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
                 : "v"(g));
    out[threadIdx.x] = w0;
    out[64 + threadIdx.x] = w1;
    out[128 + threadIdx.x] = w2;
    out[192 + threadIdx.x] = w3;
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int c = 0; c < 4; c++) {
        printf("W%d:", c);
        for (int l = 0; l < 64; l++) {
            printf(" %u", h[64 * c + l]);
            bad += h[64 * c + l] != (unsigned)(16 * c + l % 16);
        }
        printf("\n");
    }
    printf(bad ? "MISMATCH: %d lanes\n" : "ok: W_c = row c of the source in all four rows\n", bad);
    return bad != 0;
}
