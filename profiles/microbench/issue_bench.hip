// issue_bench.hip — issue cost of the instructions of the tile kernel's position pipeline (tiles.hip) on gfx950, with 1, 2
// and 4 waves per SIMD: v_readlane_b32, v_add_u32 (SGPR operand), v_fmac_f64 (SGPR-pair operand), ds_read_b128, and the
// 7-instruction mix of one position. Prints shader cycles (s_memtime) per instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 issue_bench.hip -o issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE>
__global__ void k(unsigned long long *out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const unsigned lane = threadIdx.x & 63;
    double a0 = lane, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double x0 = 1.5, x1 = 2.5;
    unsigned v = lane * 16, r0 = 0, r1 = 0, rb = 800, zero = 0;
    lds[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { // 64 v_readlane
            asm volatile(REP16("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 4\n v_readlane_b32 s22, %0, 5\n v_readlane_b32 s23, %0, 6\n") ::"v"(v) : "s20", "s21", "s22", "s23");
        } else if (MODE == 1) { // 64 v_add_u32 with SGPR
            asm volatile(REP16("v_add_u32 %0, s20, %2\n v_add_u32 %1, s21, %2\n v_add_u32 %0, s22, %2\n v_add_u32 %1, s23, %2\n") : "+v"(r0), "+v"(r1) : "v"(v) : "s20", "s21", "s22", "s23");
        } else if (MODE == 2) { // 64 v_fmac_f64, 8 independent accumulators
            asm volatile(REP4(REP4("v_fmac_f64 %0, s[20:21], %8\n v_fmac_f64 %1, s[22:23], %9\n v_fmac_f64 %2, s[20:21], %8\n v_fmac_f64 %3, s[22:23], %9\n") REP4("v_fmac_f64 %4, s[20:21], %8\n v_fmac_f64 %5, s[22:23], %9\n v_fmac_f64 %6, s[20:21], %8\n v_fmac_f64 %7, s[22:23], %9\n"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(x0), "v"(x1)
                         : "s20", "s21", "s22", "s23");
        } else if (MODE == 3) { // 64 ds_read_b128 (4 destinations), drained at the end
            asm volatile(REP16("ds_read_b128 v[40:43], %0\n ds_read_b128 v[44:47], %0 offset:1024\n ds_read_b128 v[48:51], %0 offset:2048\n ds_read_b128 v[52:55], %0 offset:3072\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(v)
                         : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        } else if (MODE == 4) { // 16 positions of the pipeline mix: 3 readlane, v_add, ds_read_b128, 2 fmac (data from a read 4 positions back)
            asm volatile(REP4("v_readlane_b32 s24, %2, 3\n v_readlane_b32 s20, %2, 4\n v_readlane_b32 s21, %2, 5\n v_add_u32 %3, s24, %2\n ds_read_b128 v[40:43], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[20:21], v[44:45]\n v_fmac_f64 %1, s[20:21], v[46:47]\n"
                              "v_readlane_b32 s24, %2, 6\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %2, 8\n v_add_u32 %3, s24, %2\n ds_read_b128 v[44:47], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[22:23], v[48:49]\n v_fmac_f64 %1, s[22:23], v[50:51]\n"
                              "v_readlane_b32 s24, %2, 9\n v_readlane_b32 s20, %2, 10\n v_readlane_b32 s21, %2, 11\n v_add_u32 %3, s24, %2\n ds_read_b128 v[48:51], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[20:21], v[52:53]\n v_fmac_f64 %1, s[20:21], v[54:55]\n"
                              "v_readlane_b32 s24, %2, 12\n v_readlane_b32 s22, %2, 13\n v_readlane_b32 s23, %2, 14\n v_add_u32 %3, s24, %2\n ds_read_b128 v[52:55], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[22:23], v[40:41]\n v_fmac_f64 %1, s[22:23], v[42:43]\n")
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0)
                         : "s20", "s21", "s22", "s23", "s24", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        } else if (MODE == 5) { // the same without the readlanes of the weight (1 readlane, v_add, ds_read, 2 fmac)
            asm volatile(REP4("v_readlane_b32 s24, %2, 3\n v_add_u32 %3, s24, %2\n ds_read_b128 v[40:43], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[20:21], v[44:45]\n v_fmac_f64 %1, s[20:21], v[46:47]\n"
                              "v_readlane_b32 s24, %2, 6\n v_add_u32 %3, s24, %2\n ds_read_b128 v[44:47], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[22:23], v[48:49]\n v_fmac_f64 %1, s[22:23], v[50:51]\n"
                              "v_readlane_b32 s24, %2, 9\n v_add_u32 %3, s24, %2\n ds_read_b128 v[48:51], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[20:21], v[52:53]\n v_fmac_f64 %1, s[20:21], v[54:55]\n"
                              "v_readlane_b32 s24, %2, 12\n v_add_u32 %3, s24, %2\n ds_read_b128 v[52:55], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[22:23], v[40:41]\n v_fmac_f64 %1, s[22:23], v[42:43]\n")
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0)
                         : "s20", "s21", "s22", "s23", "s24", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        } else if (MODE == 6) { // 64 v_fma_f64 with VGPR weight
            asm volatile(REP4(REP4("v_fma_f64 %0, %10, %8, %0\n v_fma_f64 %1, %10, %9, %1\n v_fma_f64 %2, %10, %8, %2\n v_fma_f64 %3, %10, %9, %3\n") REP4("v_fma_f64 %4, %10, %8, %4\n v_fma_f64 %5, %10, %9, %5\n v_fma_f64 %6, %10, %8, %6\n v_fma_f64 %7, %10, %9, %7\n"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(x0), "v"(x1), "v"(a0 * 0 + 1.25));
        } else if (MODE == 7) {
            asm volatile("s_mov_b32 s26, 800\n" REP4("v_readlane_b32 s25, %2, 3\n s_bfe_u32 s24, s25, 0x80000\n s_mul_i32 s24, s24, s26\n v_readlane_b32 s20, %2, 4\n v_readlane_b32 s21, %2, 5\n v_add_u32 %3, s24, %2\n ds_read_b128 v[40:43], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[20:21], v[44:45]\n v_fmac_f64 %1, s[20:21], v[46:47]\n"
"s_bfe_u32 s24, s25, 0x80008\n s_mul_i32 s24, s24, s26\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %2, 8\n v_add_u32 %3, s24, %2\n ds_read_b128 v[44:47], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[22:23], v[48:49]\n v_fmac_f64 %1, s[22:23], v[50:51]\n"
"s_bfe_u32 s24, s25, 0x80010\n s_mul_i32 s24, s24, s26\n v_readlane_b32 s20, %2, 10\n v_readlane_b32 s21, %2, 11\n v_add_u32 %3, s24, %2\n ds_read_b128 v[48:51], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[20:21], v[52:53]\n v_fmac_f64 %1, s[20:21], v[54:55]\n"
"s_bfe_u32 s24, s25, 0x80018\n s_mul_i32 s24, s24, s26\n v_readlane_b32 s22, %2, 13\n v_readlane_b32 s23, %2, 14\n v_add_u32 %3, s24, %2\n ds_read_b128 v[52:55], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[22:23], v[40:41]\n v_fmac_f64 %1, s[22:23], v[42:43]\n"
)
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0), "v"(rb), "v"(zero)
                         : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
        } else if (MODE == 8) {
            asm volatile("s_mov_b32 s26, 800\n" REP4("v_readlane_b32 s25, %2, 3\n s_bfe_u32 s24, s25, 0x80008\n v_readlane_b32 s20, %2, 4\n v_readlane_b32 s21, %2, 5\n v_mad_u32_u24 %3, s24, %4, %2\n ds_read_b128 v[40:43], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[20:21], v[44:45]\n v_fmac_f64 %1, s[20:21], v[46:47]\n"
"s_bfe_u32 s24, s25, 0x80008\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %2, 8\n v_mad_u32_u24 %3, s24, %4, %2\n ds_read_b128 v[44:47], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[22:23], v[48:49]\n v_fmac_f64 %1, s[22:23], v[50:51]\n"
"s_bfe_u32 s24, s25, 0x80008\n v_readlane_b32 s20, %2, 10\n v_readlane_b32 s21, %2, 11\n v_mad_u32_u24 %3, s24, %4, %2\n ds_read_b128 v[48:51], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[20:21], v[52:53]\n v_fmac_f64 %1, s[20:21], v[54:55]\n"
"s_bfe_u32 s24, s25, 0x80008\n v_readlane_b32 s22, %2, 13\n v_readlane_b32 s23, %2, 14\n v_mad_u32_u24 %3, s24, %4, %2\n ds_read_b128 v[52:55], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[22:23], v[40:41]\n v_fmac_f64 %1, s[22:23], v[42:43]\n"
)
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0), "v"(rb), "v"(zero)
                         : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
        } else if (MODE == 9) {
            asm volatile("s_mov_b32 s26, 800\n" REP4("v_readlane_b32 s24, %2, 3\n v_add_u32 %3, s24, %2\n ds_read_b128 v[40:43], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[44:45]\n v_add_f64 %1, %1, v[46:47]\n"
"v_readlane_b32 s24, %2, 6\n v_add_u32 %3, s24, %2\n ds_read_b128 v[44:47], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[48:49]\n v_add_f64 %1, %1, v[50:51]\n"
"v_readlane_b32 s24, %2, 9\n v_add_u32 %3, s24, %2\n ds_read_b128 v[48:51], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[52:53]\n v_add_f64 %1, %1, v[54:55]\n"
"v_readlane_b32 s24, %2, 12\n v_add_u32 %3, s24, %2\n ds_read_b128 v[52:55], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[40:41]\n v_add_f64 %1, %1, v[42:43]\n"
)
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0), "v"(rb), "v"(zero)
                         : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
        } else if (MODE == 10) {
            asm volatile("s_mov_b32 s26, 800\n" REP4("v_readlane_b32 s25, %2, 3\n s_bfe_u32 s24, s25, 0x80008\n v_mad_u32_u24 %3, s24, %4, %2\n ds_read_b128 v[40:43], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[44:45]\n v_add_f64 %1, %1, v[46:47]\n"
"s_bfe_u32 s24, s25, 0x80008\n v_mad_u32_u24 %3, s24, %4, %2\n ds_read_b128 v[44:47], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[48:49]\n v_add_f64 %1, %1, v[50:51]\n"
"s_bfe_u32 s24, s25, 0x80008\n v_mad_u32_u24 %3, s24, %4, %2\n ds_read_b128 v[48:51], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[52:53]\n v_add_f64 %1, %1, v[54:55]\n"
"s_bfe_u32 s24, s25, 0x80008\n v_mad_u32_u24 %3, s24, %4, %2\n ds_read_b128 v[52:55], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[40:41]\n v_add_f64 %1, %1, v[42:43]\n"
)
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0), "v"(rb), "v"(zero)
                         : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
        } else if (MODE == 11) {
            asm volatile("s_mov_b32 s26, 800\n" REP4("v_readlane_b32 s24, %2, 3\n v_readlane_b32 s20, %2, 4\n v_readlane_b32 s21, %2, 5\n v_add_u32 %3, s24, %2\n ds_read_b128 v[40:43], %3\n s_waitcnt lgkmcnt(2)\n v_fmac_f64 %0, s[20:21], v[44:45]\n v_fmac_f64 %1, s[20:21], v[46:47]\n"
"v_readlane_b32 s24, %2, 6\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %2, 8\n v_add_u32 %3, s24, %2\n ds_read_b128 v[44:47], %3\n v_fmac_f64 %0, s[22:23], v[48:49]\n v_fmac_f64 %1, s[22:23], v[50:51]\n"
"v_readlane_b32 s24, %2, 9\n v_readlane_b32 s20, %2, 10\n v_readlane_b32 s21, %2, 11\n v_add_u32 %3, s24, %2\n ds_read_b128 v[48:51], %3\n s_waitcnt lgkmcnt(2)\n v_fmac_f64 %0, s[20:21], v[52:53]\n v_fmac_f64 %1, s[20:21], v[54:55]\n"
"v_readlane_b32 s24, %2, 12\n v_readlane_b32 s22, %2, 13\n v_readlane_b32 s23, %2, 14\n v_add_u32 %3, s24, %2\n ds_read_b128 v[52:55], %3\n v_fmac_f64 %0, s[22:23], v[40:41]\n v_fmac_f64 %1, s[22:23], v[42:43]\n"
)
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0), "v"(rb), "v"(zero)
                         : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
        } else if (MODE == 12) {
            asm volatile("s_mov_b32 s26, 800\n" REP4("v_readlane_b32 s24, %2, 3\n v_add_u32 %3, s24, %2\n ds_read_b128 v[40:43], %3\n ds_read_b64 v[58:59], %5 offset:0\n s_waitcnt lgkmcnt(6)\n v_fma_f64 %0, v[56:57], v[44:45], %0\n v_fma_f64 %1, v[56:57], v[46:47], %1\n"
"v_readlane_b32 s24, %2, 6\n v_add_u32 %3, s24, %2\n ds_read_b128 v[44:47], %3\n ds_read_b64 v[56:57], %5 offset:8\n s_waitcnt lgkmcnt(6)\n v_fma_f64 %0, v[58:59], v[48:49], %0\n v_fma_f64 %1, v[58:59], v[50:51], %1\n"
"v_readlane_b32 s24, %2, 9\n v_add_u32 %3, s24, %2\n ds_read_b128 v[48:51], %3\n ds_read_b64 v[58:59], %5 offset:16\n s_waitcnt lgkmcnt(6)\n v_fma_f64 %0, v[56:57], v[52:53], %0\n v_fma_f64 %1, v[56:57], v[54:55], %1\n"
"v_readlane_b32 s24, %2, 12\n v_add_u32 %3, s24, %2\n ds_read_b128 v[52:55], %3\n ds_read_b64 v[56:57], %5 offset:24\n s_waitcnt lgkmcnt(6)\n v_fma_f64 %0, v[58:59], v[40:41], %0\n v_fma_f64 %1, v[58:59], v[42:43], %1\n"
)
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0), "v"(rb), "v"(zero)
                         : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
        } else if (MODE == 13) { // 64 v_fmac_f64 with the weight broadcast inside the instruction (DPP row_newbcast: lane n of every row of 16)
            asm volatile(REP4(REP4("v_fmac_f64_dpp %0, %10, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %10, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %10, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %10, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n")
                              REP4("v_fmac_f64_dpp %4, %10, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %10, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %6, %10, %8 row_newbcast:9 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %10, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf\n"))
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(x0), "v"(x1), "v"(a0 * 0 + 1.25));
        } else if (MODE == 14) { // general position, new form: v_mad_u32_u16 (SGPR row) + ds_read_b128 + 2 v_fmac_f64_dpp
            asm volatile("s_mov_b32 s24, 0x00030002\n" REP4("v_mad_u32_u16 %3, s24, %4, %2\n ds_read_b128 v[40:43], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64_dpp %0, %6, v[44:45] row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %6, v[46:47] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
"v_mad_u32_u16 %3, s24, %4, %2 op_sel:[1,0,0,0]\n ds_read_b128 v[44:47], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64_dpp %0, %6, v[48:49] row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %6, v[50:51] row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
"v_mad_u32_u16 %3, s24, %4, %2\n ds_read_b128 v[48:51], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64_dpp %0, %6, v[52:53] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %6, v[54:55] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
"v_mad_u32_u16 %3, s24, %4, %2 op_sel:[1,0,0,0]\n ds_read_b128 v[52:55], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64_dpp %0, %6, v[40:41] row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %6, v[42:43] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
)
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0), "v"(rb), "v"(zero), "v"(x0)
                         : "scc", "s24", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        } else if (MODE == 15) { // unit position, new form: v_mad_u32_u16 (SGPR row) + ds_read_b128 + 2 v_add_f64
            asm volatile("s_mov_b32 s24, 0x00030002\n" REP4("v_mad_u32_u16 %3, s24, %4, %2\n ds_read_b128 v[40:43], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[44:45]\n v_add_f64 %1, %1, v[46:47]\n"
"v_mad_u32_u16 %3, s24, %4, %2 op_sel:[1,0,0,0]\n ds_read_b128 v[44:47], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[48:49]\n v_add_f64 %1, %1, v[50:51]\n"
"v_mad_u32_u16 %3, s24, %4, %2\n ds_read_b128 v[48:51], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[52:53]\n v_add_f64 %1, %1, v[54:55]\n"
"v_mad_u32_u16 %3, s24, %4, %2 op_sel:[1,0,0,0]\n ds_read_b128 v[52:55], %3\n s_waitcnt lgkmcnt(3)\n v_add_f64 %0, %0, v[40:41]\n v_add_f64 %1, %1, v[42:43]\n"
)
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0), "v"(rb), "v"(zero)
                         : "scc", "s24", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        } else if (MODE == 16) { // general position, shipped form of round 3/4: .5 rdl (row pair) + 2 rdl (weight) + mad + read + 2 fma
            asm volatile(REP4("v_readlane_b32 s24, %2, 3\n v_readlane_b32 s20, %2, 4\n v_readlane_b32 s21, %2, 5\n v_mad_u32_u16 %3, s24, %4, %2\n ds_read_b128 v[40:43], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[20:21], v[44:45]\n v_fmac_f64 %1, s[20:21], v[46:47]\n"
"v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %2, 8\n v_mad_u32_u16 %3, s24, %4, %2 op_sel:[1,0,0,0]\n ds_read_b128 v[44:47], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[22:23], v[48:49]\n v_fmac_f64 %1, s[22:23], v[50:51]\n"
"v_readlane_b32 s24, %2, 9\n v_readlane_b32 s20, %2, 10\n v_readlane_b32 s21, %2, 11\n v_mad_u32_u16 %3, s24, %4, %2\n ds_read_b128 v[48:51], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[20:21], v[52:53]\n v_fmac_f64 %1, s[20:21], v[54:55]\n"
"v_readlane_b32 s22, %2, 13\n v_readlane_b32 s23, %2, 14\n v_mad_u32_u16 %3, s24, %4, %2 op_sel:[1,0,0,0]\n ds_read_b128 v[52:55], %3\n s_waitcnt lgkmcnt(3)\n v_fmac_f64 %0, s[22:23], v[40:41]\n v_fmac_f64 %1, s[22:23], v[42:43]\n"
)
                         : "+v"(a0), "+v"(a1)
                         : "v"(v & 1023u), "v"(r0), "v"(rb), "v"(zero)
                         : "scc", "s20", "s21", "s22", "s23", "s24", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + r0 + r1 == 12345.678) out[0] = 0;
}

template <int MODE>
void run(const char *name, int per_iter) {
    unsigned long long *d;
    hipMalloc(&d, 8 * 256 * 16);
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * 4 * wps);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto c : h) avg += (double)c;
        avg /= h.size();
        // cycles per instruction(-group) per SIMD = wave cycles / (iters * per_iter * waves per SIMD)
        printf("%-28s waves/SIMD %d: %.2f cycles per unit per wave, %.2f per unit per SIMD, %.3f ms (%.2f GHz)\n", name, wps, avg / (iters * (double)per_iter),
               avg / (iters * (double)per_iter * wps), ms, avg / (ms * 1e6));
        fflush(stdout);
    }
    hipFree(d);
}

int main(int argc, char **argv) {
    auto want = [&](int m) {
        if (argc < 2) return true;
        for (int i = 1; i < argc; i++)
            if (atoi(argv[i]) == m) return true;
        return false;
    };
    if (want(0)) run<0>("v_readlane_b32", 64);
    if (want(1)) run<1>("v_add_u32 (sgpr)", 64);
    if (want(2)) run<2>("v_fmac_f64 (sgpr weight)", 64);
    if (want(6)) run<6>("v_fma_f64 (vgpr weight)", 64);
    if (want(3)) run<3>("ds_read_b128", 64);
    if (want(4)) run<4>("position mix (7 instr)", 16);
    if (want(5)) run<5>("position mix, 1 readlane", 16);
    if (want(7)) run<7>("v2: 2.25 rdl + bfe + mul + add", 16);
    if (want(8)) run<8>("v3: 2.25 rdl + bfe + mad", 16);
    if (want(9)) run<9>("unit: 1 rdl + add, v_add_f64", 16);
    if (want(10)) run<10>("unit packed: .25 rdl+bfe+mad", 16);
    if (want(11)) run<11>("v1 mix, wait every 2nd", 16);
    if (want(12)) run<12>("weights via ds_read_b64", 16);
    if (want(13)) run<13>("v_fmac_f64_dpp row_newbcast", 64);
    if (want(16)) run<16>("general, r4: 2.5 rdl+mad+2 fma", 16);
    if (want(14)) run<14>("general, new: mad+2 fmac_dpp", 16);
    if (want(15)) run<15>("unit, new: mad + 2 add", 16);
    return 0;
}
