// idx_bench.hip — round 5, VERDICT r4 item 1 stage (i): what does a DYNAMICALLY selected accumulator cost in the tile kernel's
// position pipeline? All 64 lanes of a wave work the same record (lanes = column pairs), so the destination of a position's two
// f64 FMAs is wave-uniform: VGPR index mode (s_set_gpr_idx_on / _idx / _off, M0[7:0] = register offset, DST_REL) picks it.
//
// Stream per position (record dword in an SGPR: bits 7:0 = 4 * slot, bits 31:16 = ring row):
//   A: v_mad_u32_u16 addr, rec.hi, rowbytes, ring          (index mode must be OFF: DST_REL would move this destination too)
//   L: ds_read_b128 x, addr
//   F: s_set_gpr_idx_idx rec ; v_fmac_f64_dpp acc.x, w, x.lo row_newbcast:p ; v_fmac_f64_dpp acc.y, w, x.hi row_newbcast:p
// in batches of B positions: [on, B x F, off, B x A, B x L], 2 B reads in flight.
// modes: 0 static destinations (the round-4 stream), 1 indexed B = 4, 2 indexed B = 8, 3 indexed B = 8 with the address from a
// vector of row offsets by DPP (v_add_u32_dpp row_newbcast: no SGPR row), 4 = mode 2 + records and weights reloaded per chunk
// (s_load_dwordx16 + global_load_dwordx2, one lgkmcnt(0) per chunk), 5 = mode 2 with v_fma_f64 (VOP3, SRC2_REL | DST_REL) and the
// weight in an SGPR pair (no DPP; reference for the DPP form's correctness).
// Every mode checks its sums against a host evaluation. Prints shader cycles per position per SIMD (8 waves per CU = 2 per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 idx_bench.hip -o idx_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d16 __attribute__((ext_vector_type(16)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define LDX 100u // panel columns
#define ROWB (LDX * 8u)
#define NROWS 193u

// x registers: v[64 + 4 k : 64 + 4 k + 3], k = 0 .. 15; addresses v[48 + k]; weights v[44:45]; ring v46, rowbytes v47
// record SGPRs s[36 + p], p = 0 .. 15
#define XR(k) "v[64+4*" #k ":64+4*" #k "+1]"
#define XRH(k) "v[64+4*" #k "+2:64+4*" #k "+3]"
#define XQ(k) "v[64+4*" #k ":64+4*" #k "+3]"
#define AD(k) "v[48+" #k "]"
#define REC(p) "s[36+" #p "]"

#define A_EVEN(p, k) "v_mad_u32_u16 " AD(k) ", " REC(p) ", v47, v46 op_sel:[1,0,0,0]\n"
#define L_(p, k) "ds_read_b128 " XQ(k) ", " AD(k) "\n"
// static destination: slot = p (accumulator v[128 + 4 p ..])
#define F_STATIC(p, k)                                                                                     \
    "v_fmac_f64_dpp v[128+4*" #p ":128+4*" #p "+1], v[44:45], " XR(k) " row_newbcast:" #p " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp v[128+4*" #p "+2:128+4*" #p "+3], v[44:45], " XRH(k) " row_newbcast:" #p " row_mask:0xf bank_mask:0xf\n"
#define F_IDX_BODY(p, k)                                                                    \
    "v_fmac_f64_dpp v[128:129], v[44:45], " XR(k) " row_newbcast:" #p " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp v[130:131], v[44:45], " XRH(k) " row_newbcast:" #p " row_mask:0xf bank_mask:0xf\n"
#define F_ON(p, k) "s_set_gpr_idx_on " REC(p) ", 0x8\n" F_IDX_BODY(p, k)
#define F_IDX(p, k) "s_set_gpr_idx_idx " REC(p) "\n" F_IDX_BODY(p, k)
#define F_OFF "s_set_gpr_idx_off\n"
// VOP3 form: weight p in s[52 + 2 p : 53 + 2 p]?? (16 pairs = 32 SGPRs: s[52:83])
#define F3_BODY(p, k)                                                              \
    "v_fma_f64 v[128:129], s[52+2*" #p ":52+2*" #p "+1], " XR(k) ", v[128:129]\n" \
    "v_fma_f64 v[130:131], s[52+2*" #p ":52+2*" #p "+1], " XRH(k) ", v[130:131]\n"
#define F3_ON(p, k) "s_set_gpr_idx_on " REC(p) ", 0xc\n" F3_BODY(p, k)
#define F3_IDX(p, k) "s_set_gpr_idx_idx " REC(p) "\n" F3_BODY(p, k)
// address from a vector of row byte offsets (v43: lane L holds the offset of record L % 16)
#define A_DPP(p, k) "v_add_u32_dpp " AD(k) ", v43, v46 row_newbcast:" #p " row_mask:0xf bank_mask:0xf\n"


// packed records (mode 6): position p in half (p & 1) of s[36 + p / 2]: bits 7:0 of the half = 4 x slot, bits 15:8 = ring row.
// A: s_bfe_u32 (row) + v_mad_u32_u24; F: low half -> s_set_gpr_idx_idx reads bits 7:0 directly, high half -> s_lshr_b32 first
#define PK_A(p, k, sh) "s_bfe_u32 s84, s[36+" #p "/2], " sh "\n v_mad_u32_u24 " AD(k) ", s84, v47, v46\n"
#define PK_A_LO(p, k) PK_A(p, k, "0x80008")
#define PK_A_HI(p, k) PK_A(p, k, "0x80018")
#define PK_F_LO_ON(p, k) "s_set_gpr_idx_on s[36+" #p "/2], 0x8\n" F_IDX_BODY(p, k)
#define PK_F_LO(p, k) "s_set_gpr_idx_idx s[36+" #p "/2]\n" F_IDX_BODY(p, k)
#define PK_F_HI(p, k) "s_lshr_b32 s85, s[36+" #p "/2], 16\n s_set_gpr_idx_idx s85\n" F_IDX_BODY(p, k)

#define CLOBBERS                                                                                                                          \
    "scc", "m0", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54",  \
        "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", \
        "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",     \
        "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73",     \
        "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93",     \
        "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",      \
        "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "memory"

template <int MODE>
__global__ __launch_bounds__(512, 2) void kb(const uint32_t *__restrict__ recs, const double *__restrict__ wts, double *__restrict__ out,
                                             unsigned long long *__restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // the panel rows: value of (row r, column c) = 1 + r + c / 128
    for (uint32_t i = threadIdx.x; i < NROWS * LDX; i += blockDim.x) reinterpret_cast<double *>(lds)[i] = 1.0 + (double)(i / LDX) + (double)(i % LDX) / 128.0;
    __syncthreads();
    // idle lanes re-read the first active lane of their own ds_read_b128 group (tiles.hip)
    uint32_t src_lane = lane;
    if (lane * 2u >= LDX) {
        const uint32_t grp = lane < 32u ? (((lane >= 4u && lane < 12u) || (lane >= 16u && lane < 20u) || lane >= 28u) ? 1u : 0u)
                                        : (((lane >= 36u && lane < 44u) || (lane >= 48u && lane < 52u) || lane >= 60u) ? 3u : 2u);
        const uint32_t first = grp == 0u ? 0u : grp == 1u ? 4u : grp == 2u ? 32u : 36u;
        src_lane = first * 2u < LDX ? first : 0u;
    }
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds + src_lane * 16u;
    d16 a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0; // 4 x 32 VGPRs = 32 slots x 2 f64 pairs
    const uint32_t *rp = recs + (size_t)(blockIdx.x * 8u + wave) * 16u;
    const double *wp = wts + (size_t)(blockIdx.x * 8u + wave) * 16u;
    const double wv = wp[lane & 15u];
    const uint32_t roff = (rp[lane & 15u] >> 16) * ROWB;
    unsigned long long t0 = 0, t1 = 0;
    // prologue: registers, records (s_load), weights as SGPR pairs for mode 5
    asm volatile("v_mov_b32 v46, %[ring]\n v_mov_b32 v47, %[rowb]\n v_mov_b32 v44, %[wlo]\n v_mov_b32 v45, %[whi]\n v_mov_b32 v43, %[roff]\n"
                 "s_load_dwordx16 s[36:51], %[rp], 0x0\n s_load_dwordx16 s[52:67], %[wp], 0x0\n s_load_dwordx16 s[68:83], %[wp], 0x40\n s_waitcnt lgkmcnt(0)\n"
                 :
                 : [ring] "v"(ring), [rowb] "v"(ROWB), [wlo] "v"(__double2loint(wv)), [whi] "v"(__double2hiint(wv)), [roff] "v"(roff), [rp] "s"(rp), [wp] "s"(wp)
                 : CLOBBERS);
    t0 = __builtin_amdgcn_s_memtime();
#define ACC_OPS "+{v[128:159]}"(a0), "+{v[160:191]}"(a1), "+{v[192:223]}"(a2), "+{v[224:255]}"(a3)
    // prime: reads of positions 0 .. B-1 are in flight when an iteration's first F batch... (each iteration is self-contained but
    // keeps 2 B reads in flight inside; its last B positions are drained at its end: 1 drain per 16 positions, as a chunk-level
    // s_load certificate would force)
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            asm volatile(A_EVEN(0, 0) A_EVEN(1, 1) A_EVEN(2, 2) A_EVEN(3, 3) L_(0, 0) L_(1, 1) L_(2, 2) L_(3, 3) //
                         A_EVEN(4, 4) A_EVEN(5, 5) A_EVEN(6, 6) A_EVEN(7, 7) L_(4, 4) L_(5, 5) L_(6, 6) L_(7, 7) //
                         "s_waitcnt lgkmcnt(4)\n" F_STATIC(0, 0) F_STATIC(1, 1) F_STATIC(2, 2) F_STATIC(3, 3)    //
                         A_EVEN(8, 0) A_EVEN(9, 1) A_EVEN(10, 2) A_EVEN(11, 3) L_(8, 0) L_(9, 1) L_(10, 2) L_(11, 3) //
                         "s_waitcnt lgkmcnt(4)\n" F_STATIC(4, 4) F_STATIC(5, 5) F_STATIC(6, 6) F_STATIC(7, 7)    //
                         A_EVEN(12, 4) A_EVEN(13, 5) A_EVEN(14, 6) A_EVEN(15, 7) L_(12, 4) L_(13, 5) L_(14, 6) L_(15, 7) //
                         "s_waitcnt lgkmcnt(4)\n" F_STATIC(8, 0) F_STATIC(9, 1) F_STATIC(10, 2) F_STATIC(11, 3) //
                         "s_waitcnt lgkmcnt(0)\n" F_STATIC(12, 4) F_STATIC(13, 5) F_STATIC(14, 6) F_STATIC(15, 7)
                         : ACC_OPS
                         :
                         : CLOBBERS);
        } else if (MODE == 1) {
            asm volatile(A_EVEN(0, 0) A_EVEN(1, 1) A_EVEN(2, 2) A_EVEN(3, 3) L_(0, 0) L_(1, 1) L_(2, 2) L_(3, 3) //
                         A_EVEN(4, 4) A_EVEN(5, 5) A_EVEN(6, 6) A_EVEN(7, 7) L_(4, 4) L_(5, 5) L_(6, 6) L_(7, 7) //
                         "s_waitcnt lgkmcnt(4)\n" F_ON(0, 0) F_IDX(1, 1) F_IDX(2, 2) F_IDX(3, 3) F_OFF             //
                         A_EVEN(8, 0) A_EVEN(9, 1) A_EVEN(10, 2) A_EVEN(11, 3) L_(8, 0) L_(9, 1) L_(10, 2) L_(11, 3) //
                         "s_waitcnt lgkmcnt(4)\n" F_ON(4, 4) F_IDX(5, 5) F_IDX(6, 6) F_IDX(7, 7) F_OFF             //
                         A_EVEN(12, 4) A_EVEN(13, 5) A_EVEN(14, 6) A_EVEN(15, 7) L_(12, 4) L_(13, 5) L_(14, 6) L_(15, 7) //
                         "s_waitcnt lgkmcnt(4)\n" F_ON(8, 0) F_IDX(9, 1) F_IDX(10, 2) F_IDX(11, 3) F_OFF           //
                         "s_waitcnt lgkmcnt(0)\n" F_ON(12, 4) F_IDX(13, 5) F_IDX(14, 6) F_IDX(15, 7) F_OFF
                         : ACC_OPS
                         :
                         : CLOBBERS);
        } else if (MODE == 2 || MODE == 4) {
            if (MODE == 4) // the next chunk's records and weights (the same addresses: timing only), certified by the lgkmcnt(0) at the end
                asm volatile("s_load_dwordx16 s[36:51], %[rp], 0x0\n global_load_dwordx2 v[44:45], %[wa], off\n s_waitcnt vmcnt(0) lgkmcnt(0)\n"
                             :
                             : [rp] "s"(rp), [wa] "v"(wp + (lane & 15u))
                             : CLOBBERS);
            asm volatile(A_EVEN(0, 0) A_EVEN(1, 1) A_EVEN(2, 2) A_EVEN(3, 3) A_EVEN(4, 4) A_EVEN(5, 5) A_EVEN(6, 6) A_EVEN(7, 7) //
                         L_(0, 0) L_(1, 1) L_(2, 2) L_(3, 3) L_(4, 4) L_(5, 5) L_(6, 6) L_(7, 7)                                    //
                         A_EVEN(8, 8) A_EVEN(9, 9) A_EVEN(10, 10) A_EVEN(11, 11) A_EVEN(12, 12) A_EVEN(13, 13) A_EVEN(14, 14) A_EVEN(15, 15) //
                         L_(8, 8) L_(9, 9) L_(10, 10) L_(11, 11) L_(12, 12) L_(13, 13) L_(14, 14) L_(15, 15)                           //
                         "s_waitcnt lgkmcnt(8)\n" F_ON(0, 0) F_IDX(1, 1) F_IDX(2, 2) F_IDX(3, 3) F_IDX(4, 4) F_IDX(5, 5) F_IDX(6, 6) F_IDX(7, 7) F_OFF //
                         "s_waitcnt lgkmcnt(0)\n" F_ON(8, 8) F_IDX(9, 9) F_IDX(10, 10) F_IDX(11, 11) F_IDX(12, 12) F_IDX(13, 13) F_IDX(14, 14) F_IDX(15, 15) F_OFF
                         : ACC_OPS
                         :
                         : CLOBBERS);
        } else if (MODE == 3) {
            asm volatile(A_DPP(0, 0) A_DPP(1, 1) A_DPP(2, 2) A_DPP(3, 3) A_DPP(4, 4) A_DPP(5, 5) A_DPP(6, 6) A_DPP(7, 7) //
                         L_(0, 0) L_(1, 1) L_(2, 2) L_(3, 3) L_(4, 4) L_(5, 5) L_(6, 6) L_(7, 7)                            //
                         A_DPP(8, 8) A_DPP(9, 9) A_DPP(10, 10) A_DPP(11, 11) A_DPP(12, 12) A_DPP(13, 13) A_DPP(14, 14) A_DPP(15, 15) //
                         L_(8, 8) L_(9, 9) L_(10, 10) L_(11, 11) L_(12, 12) L_(13, 13) L_(14, 14) L_(15, 15)                   //
                         "s_waitcnt lgkmcnt(8)\n" F_ON(0, 0) F_IDX(1, 1) F_IDX(2, 2) F_IDX(3, 3) F_IDX(4, 4) F_IDX(5, 5) F_IDX(6, 6) F_IDX(7, 7) F_OFF //
                         "s_waitcnt lgkmcnt(0)\n" F_ON(8, 8) F_IDX(9, 9) F_IDX(10, 10) F_IDX(11, 11) F_IDX(12, 12) F_IDX(13, 13) F_IDX(14, 14) F_IDX(15, 15) F_OFF
                         : ACC_OPS
                         :
                         : CLOBBERS);
        } else if (MODE == 5) {
            asm volatile(A_EVEN(0, 0) A_EVEN(1, 1) A_EVEN(2, 2) A_EVEN(3, 3) A_EVEN(4, 4) A_EVEN(5, 5) A_EVEN(6, 6) A_EVEN(7, 7) //
                         L_(0, 0) L_(1, 1) L_(2, 2) L_(3, 3) L_(4, 4) L_(5, 5) L_(6, 6) L_(7, 7)                                    //
                         A_EVEN(8, 8) A_EVEN(9, 9) A_EVEN(10, 10) A_EVEN(11, 11) A_EVEN(12, 12) A_EVEN(13, 13) A_EVEN(14, 14) A_EVEN(15, 15) //
                         L_(8, 8) L_(9, 9) L_(10, 10) L_(11, 11) L_(12, 12) L_(13, 13) L_(14, 14) L_(15, 15)                           //
                         "s_waitcnt lgkmcnt(8)\n" F3_ON(0, 0) F3_IDX(1, 1) F3_IDX(2, 2) F3_IDX(3, 3) F3_IDX(4, 4) F3_IDX(5, 5) F3_IDX(6, 6) F3_IDX(7, 7) F_OFF //
                         "s_waitcnt lgkmcnt(0)\n" F3_ON(8, 8) F3_IDX(9, 9) F3_IDX(10, 10) F3_IDX(11, 11) F3_IDX(12, 12) F3_IDX(13, 13) F3_IDX(14, 14) F3_IDX(15, 15) F_OFF
                         : ACC_OPS
                         :
                         : CLOBBERS);
        } else if (MODE == 6) {
            asm volatile(PK_A_LO(0, 0) PK_A_HI(1, 1) PK_A_LO(2, 2) PK_A_HI(3, 3) PK_A_LO(4, 4) PK_A_HI(5, 5) PK_A_LO(6, 6) PK_A_HI(7, 7) //
                         L_(0, 0) L_(1, 1) L_(2, 2) L_(3, 3) L_(4, 4) L_(5, 5) L_(6, 6) L_(7, 7)                                    //
                         PK_A_LO(8, 8) PK_A_HI(9, 9) PK_A_LO(10, 10) PK_A_HI(11, 11) PK_A_LO(12, 12) PK_A_HI(13, 13) PK_A_LO(14, 14) PK_A_HI(15, 15) //
                         L_(8, 8) L_(9, 9) L_(10, 10) L_(11, 11) L_(12, 12) L_(13, 13) L_(14, 14) L_(15, 15)                           //
                         "s_waitcnt lgkmcnt(8)\n" PK_F_LO_ON(0, 0) PK_F_HI(1, 1) PK_F_LO(2, 2) PK_F_HI(3, 3) PK_F_LO(4, 4) PK_F_HI(5, 5) PK_F_LO(6, 6) PK_F_HI(7, 7) F_OFF //
                         "s_waitcnt lgkmcnt(0)\n" PK_F_LO_ON(8, 8) PK_F_HI(9, 9) PK_F_LO(10, 10) PK_F_HI(11, 11) PK_F_LO(12, 12) PK_F_HI(13, 13) PK_F_LO(14, 14) PK_F_HI(15, 15) F_OFF
                         : ACC_OPS
                         :
                         : CLOBBERS);
        }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8u + wave] = t1 - t0;
    // slot s = registers v[128 + 4 s .. +3] = doubles 2 s, 2 s + 1 of the 64-double accumulator file a0 | a1 | a2 | a3
    double *o = out + ((size_t)(blockIdx.x * 8u + wave) * 32u) * 128u; // [slot][lane][2]
    d16 *acc[4] = {&a0, &a1, &a2, &a3};
#pragma unroll
    for (int s = 0; s < 32; s++) {
        const d16 &t = *acc[s / 8];
        o[(size_t)s * 128u + lane * 2u] = t[(s % 8) * 2];
        o[(size_t)s * 128u + lane * 2u + 1u] = t[(s % 8) * 2 + 1];
    }
}

template <int MODE>
void run(const char *name) {
    const int n_wg = 256, iters = 4000;
    const size_t n_waves = (size_t)n_wg * 8;
    std::vector<uint32_t> recs(n_waves * 16);
    std::vector<double> wts(n_waves * 16);
    srand(12345 + MODE);
    for (size_t w = 0; w < n_waves; w++)
        for (int p = 0; p < 16; p++) {
            const uint32_t slot = MODE == 0 ? (uint32_t)p : (uint32_t)(rand() % 32);
            const uint32_t row = (uint32_t)(rand() % NROWS);
            recs[w * 16 + p] = (slot * 4u) | (1u << 8) | (row << 16);
            wts[w * 16 + p] = 0.5 + (rand() % 1000) / 1000.0;
        }
    std::vector<uint32_t> unpacked = recs;
    if (MODE == 6)
        for (size_t w = 0; w < n_waves; w++) {
            uint32_t pk[8];
            for (int p = 0; p < 16; p += 2) {
                const uint32_t a = recs[w * 16 + p], b = recs[w * 16 + p + 1];
                pk[p / 2] = ((a & 0xFFu) | ((a >> 16) << 8)) | (((b & 0xFFu) | ((b >> 16) << 8)) << 16);
            }
            for (int p = 0; p < 16; p++) recs[w * 16 + p] = p < 8 ? pk[p] : 0u;
        }
    uint32_t *d_recs;
    double *d_wts, *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_recs, recs.size() * 4 + 4096);
    hipMalloc(&d_wts, wts.size() * 8 + 4096);
    hipMalloc(&d_out, n_waves * 32 * 128 * 8);
    hipMalloc(&d_cyc, n_waves * 8);
    hipMemcpy(d_recs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_wts, wts.data(), wts.size() * 8, hipMemcpyHostToDevice);
    const size_t shmem = 160 * 1024;
    hipFuncSetAttribute((const void *)kb<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kb<MODE>, dim3(n_wg), dim3(512), shmem, 0, d_recs, d_wts, d_out, d_cyc, iters);
        hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) {
            printf("%-44s launch failed: %s\n", name, hipGetErrorString(hipGetLastError()));
            return;
        }
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> cyc(n_waves);
    std::vector<double> out(n_waves * 32 * 128);
    hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto c : cyc) avg += (double)c;
    avg /= cyc.size();
    // check: slot sums over the 16 records, iters times
    double worst = 0;
    for (size_t w = 0; w < n_waves; w += 37) {
        for (int s = 0; s < 32; s++)
            for (uint32_t lane = 0; lane < 50; lane++)
                for (int h = 0; h < 2; h++) {
                    double per = 0;
                    for (int p = 0; p < 16; p++) {
                        const uint32_t r = unpacked[w * 16 + p];
                        if ((r & 0xFFu) / 4u != (uint32_t)s) continue;
                        const uint32_t row = r >> 16, c = lane * 2u + h;
                        per += wts[w * 16 + p] * (1.0 + (double)row + (double)c / 128.0);
                    }
                    const double want = per * iters, got = out[(w * 32 + s) * 128 + lane * 2 + h];
                    const double e = fabs(got - want) / (fabs(want) + 1.0);
                    if (e > worst) worst = e;
                }
    }
    const double per_wave = avg / ((double)iters * 16.0);
    printf("%-44s %6.2f clk per position per wave, %6.2f per SIMD (2 waves), %6.2f per CU; %.3f ms (%.2f GHz); max rel err %.1e %s\n", name, per_wave,
           per_wave / 2.0, per_wave / 8.0, ms, avg / (ms * 1e6), worst, worst < 1e-9 ? "ok" : "WRONG");
    fflush(stdout);
    hipFree(d_recs);
    hipFree(d_wts);
    hipFree(d_out);
    hipFree(d_cyc);
}


// ---- 16 waves per workgroup (4 per SIMD, <= 128 VGPRs): 16 slots per wave in v[64:127], x v[32:63] (2 sets x 4 positions), addresses v[24:31],
// weights v[20:21], ring v22, rowbytes v23; batches of 4
#define YR(k) "v[32+4*" #k ":32+4*" #k "+1]"
#define YRH(k) "v[32+4*" #k "+2:32+4*" #k "+3]"
#define YQ(k) "v[32+4*" #k ":32+4*" #k "+3]"
#define BD(k) "v[24+" #k "]"
#define A16(p, k) "v_mad_u32_u16 " BD(k) ", " REC(p) ", v23, v22 op_sel:[1,0,0,0]\n"
#define L16(p, k) "ds_read_b128 " YQ(k) ", " BD(k) "\n"
#define F16_STATIC(p, k)                                                                                   \
    "v_fmac_f64_dpp v[64+4*" #p ":64+4*" #p "+1], v[20:21], " YR(k) " row_newbcast:" #p " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp v[64+4*" #p "+2:64+4*" #p "+3], v[20:21], " YRH(k) " row_newbcast:" #p " row_mask:0xf bank_mask:0xf\n"
#define F16_BODY(p, k)                                                                   \
    "v_fmac_f64_dpp v[64:65], v[20:21], " YR(k) " row_newbcast:" #p " row_mask:0xf bank_mask:0xf\n" \
    "v_fmac_f64_dpp v[66:67], v[20:21], " YRH(k) " row_newbcast:" #p " row_mask:0xf bank_mask:0xf\n"
#define F16_ON(p, k) "s_set_gpr_idx_on " REC(p) ", 0x8\n" F16_BODY(p, k)
#define F16_IDX(p, k) "s_set_gpr_idx_idx " REC(p) "\n" F16_BODY(p, k)
#define CLOB16                                                                                                                             \
    "scc", "m0", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "v20", "v21", "v22", "v23", \
        "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44",   \
        "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "memory"

template <int MODE, int NWV>
__global__ __launch_bounds__(64 * NWV) void kb16(const uint32_t *__restrict__ recs, const double *__restrict__ wts, double *__restrict__ out,
                                                 unsigned long long *__restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (uint32_t i = threadIdx.x; i < NROWS * LDX; i += blockDim.x) reinterpret_cast<double *>(lds)[i] = 1.0 + (double)(i / LDX) + (double)(i % LDX) / 128.0;
    __syncthreads();
    uint32_t src_lane = lane;
    if (lane * 2u >= LDX) {
        const uint32_t grp = lane < 32u ? (((lane >= 4u && lane < 12u) || (lane >= 16u && lane < 20u) || lane >= 28u) ? 1u : 0u)
                                        : (((lane >= 36u && lane < 44u) || (lane >= 48u && lane < 52u) || lane >= 60u) ? 3u : 2u);
        const uint32_t first = grp == 0u ? 0u : grp == 1u ? 4u : grp == 2u ? 32u : 36u;
        src_lane = first * 2u < LDX ? first : 0u;
    }
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds + src_lane * 16u;
    d16 a0 = 0.0, a1 = 0.0; // 2 x 32 VGPRs = 16 slots x 2 f64 pairs
    const uint32_t *rp = recs + (size_t)(blockIdx.x * NWV + wave) * 16u;
    const double *wp = wts + (size_t)(blockIdx.x * NWV + wave) * 16u;
    const double wv = wp[lane & 15u];
    asm volatile("v_mov_b32 v22, %[ring]\n v_mov_b32 v23, %[rowb]\n v_mov_b32 v20, %[wlo]\n v_mov_b32 v21, %[whi]\n"
                 "s_load_dwordx16 s[36:51], %[rp], 0x0\n s_waitcnt lgkmcnt(0)\n"
                 :
                 : [ring] "v"(ring), [rowb] "v"(ROWB), [wlo] "v"(__double2loint(wv)), [whi] "v"(__double2hiint(wv)), [rp] "s"(rp)
                 : CLOB16);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define ACC16 "+{v[64:95]}"(a0), "+{v[96:127]}"(a1)
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            asm volatile(A16(0, 0) A16(1, 1) A16(2, 2) A16(3, 3) L16(0, 0) L16(1, 1) L16(2, 2) L16(3, 3) //
                         A16(4, 4) A16(5, 5) A16(6, 6) A16(7, 7) L16(4, 4) L16(5, 5) L16(6, 6) L16(7, 7) //
                         "s_waitcnt lgkmcnt(4)\n" F16_STATIC(0, 0) F16_STATIC(1, 1) F16_STATIC(2, 2) F16_STATIC(3, 3) //
                         A16(8, 0) A16(9, 1) A16(10, 2) A16(11, 3) L16(8, 0) L16(9, 1) L16(10, 2) L16(11, 3) //
                         "s_waitcnt lgkmcnt(4)\n" F16_STATIC(4, 4) F16_STATIC(5, 5) F16_STATIC(6, 6) F16_STATIC(7, 7) //
                         A16(12, 4) A16(13, 5) A16(14, 6) A16(15, 7) L16(12, 4) L16(13, 5) L16(14, 6) L16(15, 7) //
                         "s_waitcnt lgkmcnt(4)\n" F16_STATIC(8, 0) F16_STATIC(9, 1) F16_STATIC(10, 2) F16_STATIC(11, 3) //
                         "s_waitcnt lgkmcnt(0)\n" F16_STATIC(12, 4) F16_STATIC(13, 5) F16_STATIC(14, 6) F16_STATIC(15, 7)
                         : ACC16
                         :
                         : CLOB16);
        } else {
            asm volatile(A16(0, 0) A16(1, 1) A16(2, 2) A16(3, 3) L16(0, 0) L16(1, 1) L16(2, 2) L16(3, 3) //
                         A16(4, 4) A16(5, 5) A16(6, 6) A16(7, 7) L16(4, 4) L16(5, 5) L16(6, 6) L16(7, 7) //
                         "s_waitcnt lgkmcnt(4)\n" F16_ON(0, 0) F16_IDX(1, 1) F16_IDX(2, 2) F16_IDX(3, 3) F_OFF //
                         A16(8, 0) A16(9, 1) A16(10, 2) A16(11, 3) L16(8, 0) L16(9, 1) L16(10, 2) L16(11, 3) //
                         "s_waitcnt lgkmcnt(4)\n" F16_ON(4, 4) F16_IDX(5, 5) F16_IDX(6, 6) F16_IDX(7, 7) F_OFF //
                         A16(12, 4) A16(13, 5) A16(14, 6) A16(15, 7) L16(12, 4) L16(13, 5) L16(14, 6) L16(15, 7) //
                         "s_waitcnt lgkmcnt(4)\n" F16_ON(8, 0) F16_IDX(9, 1) F16_IDX(10, 2) F16_IDX(11, 3) F_OFF //
                         "s_waitcnt lgkmcnt(0)\n" F16_ON(12, 4) F16_IDX(13, 5) F16_IDX(14, 6) F16_IDX(15, 7) F_OFF
                         : ACC16
                         :
                         : CLOB16);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * NWV + wave] = t1 - t0;
    double *o = out + ((size_t)(blockIdx.x * NWV + wave) * 16u) * 128u;
    d16 *acc[2] = {&a0, &a1};
#pragma unroll
    for (int s2 = 0; s2 < 16; s2++) {
        const d16 &t = *acc[s2 / 8];
        o[(size_t)s2 * 128u + lane * 2u] = t[(s2 % 8) * 2];
        o[(size_t)s2 * 128u + lane * 2u + 1u] = t[(s2 % 8) * 2 + 1];
    }
}

template <int MODE, int NWV>
void run16(const char *name) {
    const int n_wg = 256, iters = 4000;
    const size_t n_waves = (size_t)n_wg * NWV;
    std::vector<uint32_t> recs(n_waves * 16);
    std::vector<double> wts(n_waves * 16);
    srand(777 + MODE);
    for (size_t w = 0; w < n_waves; w++)
        for (int p = 0; p < 16; p++) {
            const uint32_t slot = MODE == 0 ? (uint32_t)p : (uint32_t)(rand() % 16);
            const uint32_t row = (uint32_t)(rand() % NROWS);
            recs[w * 16 + p] = (slot * 4u) | (1u << 8) | (row << 16);
            wts[w * 16 + p] = 0.5 + (rand() % 1000) / 1000.0;
        }
    uint32_t *d_recs;
    double *d_wts, *d_out;
    unsigned long long *d_cyc;
    (void)hipMalloc(&d_recs, recs.size() * 4 + 4096);
    (void)hipMalloc(&d_wts, wts.size() * 8 + 4096);
    (void)hipMalloc(&d_out, n_waves * 16 * 128 * 8);
    (void)hipMalloc(&d_cyc, n_waves * 8);
    (void)hipMemcpy(d_recs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_wts, wts.data(), wts.size() * 8, hipMemcpyHostToDevice);
    const size_t shmem = 160 * 1024;
    (void)hipFuncSetAttribute((const void *)kb16<MODE, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((kb16<MODE, NWV>), dim3(n_wg), dim3(64 * NWV), shmem, 0, d_recs, d_wts, d_out, d_cyc, iters);
        (void)hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) {
            printf("%-44s launch failed: %s\n", name, hipGetErrorString(hipGetLastError()));
            return;
        }
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> cyc(n_waves);
    std::vector<double> out(n_waves * 16 * 128);
    (void)hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto c : cyc) avg += (double)c;
    avg /= cyc.size();
    double worst = 0;
    for (size_t w = 0; w < n_waves; w += 37)
        for (int s2 = 0; s2 < 16; s2++)
            for (uint32_t lane = 0; lane < 50; lane++)
                for (int h = 0; h < 2; h++) {
                    double per = 0;
                    for (int p = 0; p < 16; p++) {
                        const uint32_t r = recs[w * 16 + p];
                        if ((r & 0xFFu) / 4u != (uint32_t)s2) continue;
                        per += wts[w * 16 + p] * (1.0 + (double)(r >> 16) + (double)(lane * 2u + h) / 128.0);
                    }
                    const double want = per * iters, got = out[(w * 16 + s2) * 128 + lane * 2 + h];
                    const double e = fabs(got - want) / (fabs(want) + 1.0);
                    if (e > worst) worst = e;
                }
    const double per_wave = avg / ((double)iters * 16.0);
    printf("%-44s %6.2f clk per position per wave, %6.2f per SIMD (%d waves), %6.2f per CU; %.3f ms (%.2f GHz); max rel err %.1e %s\n", name, per_wave,
           per_wave / (NWV / 4.0), NWV / 4, per_wave / NWV, ms, avg / (ms * 1e6), worst, worst < 1e-9 ? "ok" : "WRONG");
    fflush(stdout);
    (void)hipFree(d_recs);
    (void)hipFree(d_wts);
    (void)hipFree(d_out);
    (void)hipFree(d_cyc);
}

int main() {
    run<0>("static destinations (round-4 stream)");
    run<1>("indexed, batches of 4");
    run<2>("indexed, batches of 8");
    run<3>("indexed, batches of 8, DPP address");
    run<4>("indexed, B = 8, s_load + weights per chunk");
    run<5>("indexed VOP3 fma, SGPR weights, B = 8");
    run<6>("indexed, B = 8, 16-bit packed records");
    run16<0, 16>("16 waves x 16 slots: static, B = 4");
    run16<1, 16>("16 waves x 16 slots: indexed, B = 4");
    run16<0, 12>("12 waves x 16 slots: static, B = 4");
    run16<1, 12>("12 waves x 16 slots: indexed, B = 4");
    run16<1, 8>("8 waves x 16 slots: indexed, B = 4");
    return 0;
}
