#!/usr/bin/env python3
"""Writes scan-rs_amd/csrc/tile_dense_body.inc: the round loop of the dense tile kernel (tiles_dense.inc, spmm_tile_dense_kernel)
as one hand-scheduled gfx950 instruction stream.

Why generated: the stream is unrolled over the 64 record positions of a round with four entry points, every register is a fixed
physical one (the accumulators are picked through VGPR index mode, which shifts EVERY vector destination while it is on, so no
compiler-scheduled instruction may sit between s_set_gpr_idx_on and _off), and LDS reads, scalar loads and LDS-DMA all have to
be counted by hand. A generator keeps the numbering honest; the output is committed next to it.

Register map (all clobbered by the asm statement; the compiler keeps v0-v25 and s0-s16):
  v[88:215]   accumulators: slot q = v[88 + 4 q : 88 + 4 q + 3] (two f64: columns 2 lane, 2 lane + 1)
  v[56:87]    panel rows of 8 positions in flight (4 registers each)
  v[48:55]    their LDS addresses
  (216 registers in all: two tile waves per SIMD leave 80 for a wave of the overflow gather, which runs beside this kernel)
  v[40:47]    weights of the round's 4 chunks: lane L holds weight L % 16 of the chunk (one register pair per chunk)
  v[38:39]    this lane's address in the weight stream: 8 L behind the END of the NEXT round's weights
  v26 4 x lane, v27 sink of the touch load, v30 4 x (lane % 16); v[28:29] the NEXT round's 64 weights as loaded (lane L: weight L)
  v36 ring base of the lane (LDS address of its column pair in ring row 0), v37 row pitch in bytes
  v[31:35]    LDS-DMA source offsets of the wave's 5 staging chunks
  s[36:99]    the round's records: bits 7:0 = 4 x slot (the VGPR index), 15:8 = raw count (weight refresh only), 31:16 = ring row
  s[20:21] END of the current round's records, s[22:23] round table, s[24:25] next tile to stage, s26 rounds left, s27 tile bytes,
  s28 header of the current round (7:0 chunks, bit 8 = first round of a visit), s29 its chunk count, s30 / s31 / s[32:33] scratch,
  s34 next header (s101: the one after), s100 chunks of the next round, s35 LDS address of the buffer of the tile staged last, s17 wave << 10, s18 ring base, s19 ring end
"""
import os
import sys

SKIP = set((os.environ.get("GEN_SKIP") or "").split(","))  # timing experiments (wrong results): w = weights, r = records, b = barrier, d = staging

BP = int(os.environ.get("GEN_BP") or 4)   # positions per batch (two batches of row reads in flight); 8 is as fast but takes 40 registers more
NB = 64 // BP     # batches per round
BPC = 16 // BP    # batches per chunk
ACC0 = 88
X0 = 56
A0 = 48
W0 = 40
R0 = 36


def xr(p, half):
    k = p % (2 * BP)
    b = X0 + 4 * k + 2 * half
    return f"v[{b}:{b + 1}]"


def xq(p):
    k = p % (2 * BP)
    return f"v[{X0 + 4 * k}:{X0 + 4 * k + 3}]"


def ad(p):
    return f"v{A0 + p % (2 * BP)}"


def rec(p):
    return f"s{R0 + p}"


def wreg(p):
    c = p // 16
    return f"v[{W0 + 2 * c}:{W0 + 2 * c + 1}]"


def batch_AL(b, out):
    for j in range(BP):
        p = b * BP + j
        out.append(f"v_mad_u32_u16 {ad(p)}, {rec(p)}, v37, v36 op_sel:[1,0,0,0]")
    for j in range(BP):
        p = b * BP + j
        out.append(f"ds_read_b128 {xq(p)}, {ad(p)}")


def batch_F(b, out):
    for j in range(BP):
        p = b * BP + j
        if j == 0:
            out.append(f"s_set_gpr_idx_on {rec(p)}, 0x8")
        else:
            out.append(f"s_set_gpr_idx_idx {rec(p)}")
        q = p % 16
        out.append(f"v_fmac_f64_dpp v[{ACC0}:{ACC0 + 1}], {wreg(p)}, {xr(p, 0)} row_newbcast:{q} row_mask:0xf bank_mask:0xf")
        out.append(f"v_fmac_f64_dpp v[{ACC0 + 2}:{ACC0 + 3}], {wreg(p)}, {xr(p, 1)} row_newbcast:{q} row_mask:0xf bank_mask:0xf")
    out.append("s_set_gpr_idx_off")


def dma(i, out):
    """staging chunk i of the wave: LDS destination in M0 (one wait state before the LDS-DMA reads it), source = tile base + lane offset"""
    if "d" in SKIP:
        out.append("s_nop 0")
        return
    out.append(f"s_add_u32 s30, s17, {i * 8192}")
    out.append("s_min_u32 s30, s30, s31")
    out.append("s_add_u32 m0, s30, s35")
    out.append("s_nop 0")
    out.append(f"global_load_lds_dwordx4 v{31 + i}, s[24:25]")


def gen():
    """Vector memory per round: one touch load (next round's records into L2), ONE weight load for the next round (lane L takes
    weight L of the 64 positions that END at the round's last one: 512 contiguous bytes; four loads of 16 replicated weights each
    cost four times the return bytes on the CU's fetch path, which the staging shares: -1.7 ms per pass without them), five staging
    loads. All of it is issued at the round's start and waited for (vmcnt(0)) at the next round's start. The weights reach the
    form the FMAs want - every row of 16 lanes holds the 16 weights of ONE chunk - by eight ds_bpermute_b32 (LDS crossbar)."""
    o = []
    a = o.append
    # ---- inputs into the fixed registers ----
    a("s_mov_b64 s[20:21], %[rec]")
    a("s_mov_b64 s[22:23], %[rtab]")
    a("s_mov_b64 s[24:25], %[src]")
    a("s_mov_b32 s26, %[nrounds]")
    a("s_mov_b32 s27, %[tb]")
    a("s_mov_b32 s35, %[dst]")
    a("s_mov_b32 s17, %[wave10]")
    a("s_mov_b32 s18, %[lds0]")
    a("s_lshl_b32 s19, s27, 2")
    a("s_add_u32 s19, s19, s18")
    a("s_sub_u32 s31, s27, 0x400")
    a("v_mov_b32 v38, %[pwlo]")
    a("v_mov_b32 v39, %[pwhi]")
    a("v_mov_b32 v36, %[ring]")
    a("v_mov_b32 v37, %[rowb]")
    a("v_mov_b32 v26, %[lane4]")
    a("v_and_b32 v30, 60, v26")  # 4 (lane % 16): ds_bpermute address of this lane's weight inside a chunk
    for i in range(5):
        a(f"v_mov_b32 v{31 + i}, %[voff{i}]")
    a("s_cmp_eq_u32 s26, 0")
    a("s_cbranch_scc1 LDONE%=")
    a("s_load_dwordx2 s[28:29], s[22:23], 0x0")  # headers of rounds 0 and 1
    a("s_waitcnt lgkmcnt(0)")
    a("s_mov_b32 s34, s29")
    # weights of round 0
    a("s_and_b32 s29, s28, 0xff")
    a("s_lshl_b32 s30, s29, 7")
    a("v_add_co_u32 v38, vcc, s30, v38")
    a("v_addc_co_u32 v39, vcc, 0, v39, vcc")
    a("global_load_dwordx2 v[28:29], v[38:39], off offset:-512")
    a("LROUND%=:")
    a("s_and_b32 s29, s28, 0xff")   # chunks of this round
    a("s_and_b32 s100, s34, 0xff")  # ... of the next one
    a("s_lshl_b32 s30, s29, 6")
    a("s_add_u32 s20, s20, s30")
    a("s_addc_u32 s21, s21, 0")
    a("s_sub_u32 s32, s20, 0x100")
    a("s_subb_u32 s33, s21, 0")
    rb = "s[32:33]" if "r" not in SKIP else "%[rec]"  # experiment: always the item's first chunks (scalar-cache hits)
    a(f"s_load_dwordx16 s[36:51], {rb}, 0x0")
    a(f"s_load_dwordx16 s[52:67], {rb}, 0x40")
    a(f"s_load_dwordx16 s[68:83], {rb}, 0x80")
    a(f"s_load_dwordx16 s[84:99], {rb}, 0xc0")
    a("s_load_dword s101, s[22:23], 0x8")  # header of the round after the next
    a("s_add_u32 s22, s22, 4")
    a("s_addc_u32 s23, s23, 0")
    a("s_lshl_b32 s30, s100, 7")
    # This round's weights must be there, and the tile staged during the visit BEFORE the last one (first read now); the five
    # staging loads of the last round - if it was the first of its visit (bit 9) they are the youngest operations - may still be in flight
    a("s_bitcmp1_b32 s28, 9")
    a("s_cbranch_scc0 LW0%=")
    a("s_waitcnt vmcnt(5)")
    a("s_branch LW1%=")
    a("LW0%=:")
    a("s_waitcnt vmcnt(0)")
    a("LW1%=:")
    a("s_bitcmp1_b32 s28, 8")
    a("s_cbranch_scc0 LNOBAR%=")
    if "b" not in SKIP:
        a("s_barrier")
    a("s_add_u32 s24, s24, s27")
    a("s_addc_u32 s25, s25, 0")
    a("s_add_u32 s35, s35, s27")
    a("s_cmp_eq_u32 s35, s19")
    a("s_cselect_b32 s35, s18, s35")
    a("LNOBAR%=:")
    # the round's weights: row r of the loaded register pair holds chunk r's 16 weights -> four pairs in which EVERY row holds one chunk's
    for c in range(4):
        a(f"ds_bpermute_b32 v{W0 + 2 * c}, v30, v28 offset:{64 * c}")
        a(f"ds_bpermute_b32 v{W0 + 2 * c + 1}, v30, v29 offset:{64 * c}")
    a("global_load_dword v27, v26, s[20:21]" if "t" not in SKIP else "s_nop 0")  # touch: the next round's records into L2
    a("v_add_co_u32 v38, vcc, s30, v38")  # -> behind the next round's weights
    a("v_addc_co_u32 v39, vcc, 0, v39, vcc")
    a("s_waitcnt lgkmcnt(0)")  # records, headers, the permuted weights (v[28:29] is free again)
    a("global_load_dwordx2 v[28:29], v[38:39], off offset:-512" if "w" not in SKIP else "s_nop 0")
    a("s_bitcmp1_b32 s28, 8")
    a("s_cbranch_scc0 LNODMA%=")
    a("s_cmp_eq_u32 s29, 4")
    a("s_cbranch_scc1 LPRO0%=")
    a("s_cmp_eq_u32 s29, 3")
    a("s_cbranch_scc1 LPRO1%=")
    a("s_cmp_eq_u32 s29, 2")
    a("s_cbranch_scc1 LPRO2%=")
    a("s_cmp_eq_u32 s29, 1")
    a("s_cbranch_scc1 LPRO3%=")
    # an empty round (a visit nobody has work in yet: the first two of a part)
    for i in range(5):
        dma(i, o)
    a("s_branch LEND%=")
    # a later round of a long visit: nothing to stage
    a("LNODMA%=:")
    a("s_cmp_eq_u32 s29, 4")
    a("s_cbranch_scc1 LQRO0%=")
    a("s_cmp_eq_u32 s29, 3")
    a("s_cbranch_scc1 LQRO1%=")
    a("s_cmp_eq_u32 s29, 2")
    a("s_cbranch_scc1 LQRO2%=")
    a("s_cmp_eq_u32 s29, 1")
    a("s_cbranch_scc1 LQRO3%=")
    a("s_branch LEND%=")
    # prologues: rows of the first two batches of the round (with the staging loads between them), then into the steady stream
    for with_dma in (False, True):
        for c in (3, 2, 1, 0):
            a(f"L{'P' if with_dma else 'Q'}RO{c}%=:")
            blocks = []
            for b in (BPC * c, BPC * c + 1):
                blk = []
                for j in range(BP):
                    p = b * BP + j
                    blk.append(f"v_mad_u32_u16 {ad(p)}, {rec(p)}, v37, v36 op_sel:[1,0,0,0]")
                blocks.append(blk)
                blk = []
                for j in range(BP):
                    p = b * BP + j
                    blk.append(f"ds_read_b128 {xq(p)}, {ad(p)}")
                blocks.append(blk)
            for i, blk in enumerate(blocks):
                o.extend(blk)
                if with_dma:
                    dma(i, o)
            if with_dma:
                dma(4, o)
            if not (with_dma and c == 0):
                a(f"s_branch LS{BPC * c}%=")
    for b in range(NB):
        a(f"LS{b}%=:")
        a(f"s_waitcnt lgkmcnt({BP})" if b < NB - 1 else "s_waitcnt lgkmcnt(0)")
        batch_F(b, o)
        if b + 2 < NB:
            batch_AL(b + 2, o)
    a("LEND%=:")
    a("s_and_b32 s30, s28, 0x100")
    a("s_lshl_b32 s30, s30, 1")
    a("s_mov_b32 s28, s34")
    a("s_or_b32 s28, s28, s30")  # bit 9: the round before was the first of its visit (it staged a tile)
    a("s_mov_b32 s34, s101")
    a("s_sub_u32 s26, s26, 1")
    a("s_cmp_lg_u32 s26, 0")
    a("s_cbranch_scc1 LROUND%=")
    a("LDONE%=:")
    return o


def main():
    lines = gen()
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "..", "scan-rs_amd", "csrc", "tile_dense_body.inc")
    with open(path, "w") as f:
        f.write("// generated by tools/gen_tile_dense_asm.py - do not edit\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    print(f"{len(lines)} instructions / labels -> {os.path.normpath(path)}")


if __name__ == "__main__":
    main()
