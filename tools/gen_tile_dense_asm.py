#!/usr/bin/env python3
"""Writes scan-rs_amd/csrc/tile_dense_body.inc: the round loop of the dense tile kernel (tiles_dense.inc, spmm_tile_dense_kernel)
as one hand-scheduled gfx950 instruction stream.

Why generated: the stream is unrolled over the 64 record positions of a round with four entry points, every register is a fixed
physical one (the accumulators are picked through VGPR index mode, which shifts EVERY vector destination while it is on, so no
compiler-scheduled instruction may sit between s_set_gpr_idx_on and _off), and LDS reads, scalar loads and LDS-DMA all have to
be counted by hand. A generator keeps the numbering honest; the output is committed next to it.

Register map (all clobbered by the asm statement; the compiler keeps v0-v25 and s0-s11; operands arrive in vector registers and are
read with v_readfirstlane_b32):
  v[88:215]   accumulators: slot q = v[88 + 4 q : 88 + 4 q + 3] (two f64: columns 2 lane, 2 lane + 1)
  v[56:87]    panel rows of 8 positions in flight (4 registers each)
  v[48:55]    their LDS addresses (v48 also: the offset of the records' "touch" load at the round's start)
  v[40:47]    weights of the round's 4 chunks: lane L holds weight L % 16 of the chunk (one register pair per chunk)
  v[38:39]    this lane's address in the weight stream: 8 (L % 16) behind the END of the NEXT round's weights
  v36 ring base of the lane (LDS address of its column pair in ring row 0), v37 row pitch in bytes
  v[26:33]    the NEXT round's weights as loaded: one register pair per chunk, lane L: weight L % 16 of the chunk (moved to v[40:47] at the
              round's start). v34 16 x lane: the lane's offset inside a 1 KB staging chunk (the chunk's own offset rides in the scalar base)
              (GEN_WDIRECT=0, the round's first form: v[26:27] next weights lane L = weight L, v[28:29] this round's, v30 the ds_bpermute
              address 4 x (lane % 16), v[31:35] the five staging chunks' offsets)
  (216 registers in all: two tile waves per SIMD leave 80 for a wave of the overflow gather, which runs beside this kernel)
  s[36:99]    the round's records: bits 7:0 = 4 x slot (the VGPR index), 15:8 = raw count (weight refresh only), 31:16 = ring row
  s[13:16]    copies of the last batch's four records (their chunk slot is reloaded before its FMAs)
  s17 wave << 10, s18 ring base, s19 ring end, s[20:21] END of the current round's records, s[22:23] round table,
  s[24:25] next tile to stage, s26 rounds left, s27 tile bytes, s31 tile bytes - 1024, s35 LDS address of the buffer staged into,
  s28 header of the current round (7:0 chunks, bit 8 = first round of a visit), s29 its chunk count, s34 / s101 / s12 the next
  three headers, s100 chunks of the next round, s30 / s[32:33] scratch (s[32:33]: where the next round's chunk slots come from)
  (stamp build: s[2:9] time stamps and sums, s[10:11] saved EXEC)

GEN_WSRC=tabo / tabi (round 6): the map is evaluated INSIDE the kernel - no weight stream. A position's weight is one entry of a table the
normalize refreshes (16 doubles per place of the layout when the outer side owns the map's nonlinear links, "tabo"; 16 planes over the
inner positions otherwise, "tabi"; entry 0 = 0.0 for null records, counts above 15 live in the overflow part), gathered by the
record's own (slot, count) or (ring row, count). Per round ONE vector load of the 64 records of the round after next (lane L = record
L of its four chunk slots; it also pulls them into L2 for the scalar loads, the stream form's "touch"), ONE 64-lane gather of the next
round's weights (lane L = weight of position L), and at the round's start the gathered register pair is spread into the form the
FMAs read (every row of 16 lanes = one chunk's 16 weights) by v_permlane16_swap / v_permlane32_swap (gfx950), no LDS involved:
  v[26:27]    the NEXT round's weights as gathered (lane L: position L); v28 the records of the round after next (lane L: record L)
  v[29:31]    address arithmetic; v32 tabi: 8 x plane stride; v33 4 x lane; v35 tabi: -1536 (one ring of 192 rows back, in bytes)
  v[38:39]    the table's base for this wave (tabo: the 32 places of its group; tabi: plane 0)
  s31         tabi: visit index of the current round (the staging code derives tile bytes - 1024 where it needs it)
"""
import os
import sys

PRIO = int(os.environ.get("GEN_PRIO") or 0)  # experiments (no gain, profiles/HISTORY.md): 1 = waves 4-7 at priority 1, 2 = two copies of the round whose priorities alternate per batch, opposite in the two waves of a SIMD
STAMP = bool(os.environ.get("GEN_STAMP"))  # diagnostic build: s_memtime stamps around the round's body, the staging wait and the barrier, summed per wave
SKIP = set((os.environ.get("GEN_SKIP") or "").split(","))  # timing experiments (wrong results): w = weights, r = records, b = barrier, d = staging, p = weight spreading (ds_bpermute)

PRE = (os.environ.get("GEN_PRE") or "0") != "0"  # experiment (slower by 0.3 ms per pass, profiles/HISTORY.md): the row reads of a round's first two batches are issued at the boundary in front of it (0: in the round's prologue)
WSRC = os.environ.get("GEN_WSRC") or "stream"  # where a position's weight comes from: stream = one f64 per position (pw, refreshed per normalize); tabo / tabi = gathered from the map's table by the record itself (see above)
TAB = WSRC in ("tabo", "tabi")
SYNC = os.environ.get("GEN_SYNC") or "bar"  # how the 8 waves of an item hand ring buffers to one another: bar = one s_barrier per visit; cnt = readiness / release counters in LDS (table forms only, see sync_* below)
CNT = SYNC == "cnt"
ALFIRST = (os.environ.get("GEN_ALFIRST") or "1") != "0"  # a round issues the row reads of its first two batches BEFORE it spreads its weights and starts the next round's loads: that work (45 instructions) then runs beside the reads' latency instead of in front of it - 0.5 ms per pass (round 6: 14.4 / 15.7 -> 13.9 / 15.2 on one box); 0: the round-5 order. No effect on the flow loop (its rounds have no barrier in front of them to line the waves up)
ROTFIRST = (os.environ.get("GEN_ROTFIRST") or "0") != "0" and not CNT  # experiment: the boundary's register rotation (and its wait for the next round's records) in FRONT of the visit's barrier instead of behind it
ALFIRST = ALFIRST and not CNT  # (the counter form's round top carries a label and an early poll of its own: left in the round-5 order)
ONCE = (os.environ.get("GEN_ONCE") or "0") != "0"  # barrier form: a visit's tile is staged by its FIRST round only (the default re-stages it in every round of the visit so that one vmcnt count fits every boundary; a visit of more than 4 chunks then moves its 38 KB twice)
assert SYNC in ("bar", "cnt") and (not CNT or WSRC in ("tabo", "tabi"))
SPIN_MAX = int(os.environ.get("GEN_SPIN_MAX") or (1 << 22))  # polls of a counter before a wave gives up waiting (no hang on a bug: the result is then wrong and the parity tests say so)
assert WSRC in ("stream", "tabo", "tabi")
WDIRECT = (os.environ.get("GEN_WDIRECT") or "1") != "0"  # a round's weights are loaded in the form the FMAs read (lane L: weight L % 16 of each chunk: four loads); 0: one load + eight ds_bpermute_b32
DMA_TOP = (os.environ.get("GEN_DMA") or "tail") == "top"  # staging loads at the round's start instead of in its last batches
BP = int(os.environ.get("GEN_BP") or 4)   # positions per batch (two batches of row reads in flight); 8 is as fast but takes 40 registers more
assert not TAB or (WDIRECT and not PRE and not DMA_TOP and BP == 4)
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


# timing experiment (wrong results): the positions p with p % 4 in SHARE do not read their panel row but use the row registers of the
# position in front of them - what the stream would cost if that share of the nonzeros found its row already in registers (a wave's
# nonzeros that name the same panel row: 61 % of them have such a partner at 32 cells x 3 %)
SHARE = {int(x) for x in (os.environ.get("GEN_SHARE") or "").split(",") if x}
RB = BP - len(SHARE)  # row reads per batch


def batch_AL(b, out):
    for j in range(BP):
        p = b * BP + j
        if j in SHARE:
            continue
        out.append(f"v_mad_u32_u16 {ad(p)}, {rec(p)}, v37, v36 op_sel:[1,0,0,0]")
    for j in range(BP):
        p = b * BP + j
        if j in SHARE:
            continue
        out.append(f"ds_read_b128 {xq(p)}, {ad(p)}")


def batch_F(b, out, tail=False):
    for j in range(BP):
        p = b * BP + j
        r = f"s{13 + j}" if tail else rec(p)
        if j == 0:
            out.append(f"s_set_gpr_idx_on {r}, 0x8")
        else:
            out.append(f"s_set_gpr_idx_idx {r}")
        q = p % 16
        ps = p - 1 if j in SHARE else p  # (experiment: the row of the position in front)
        out.append(f"v_fmac_f64_dpp v[{ACC0}:{ACC0 + 1}], {wreg(p)}, {xr(ps, 0)} row_newbcast:{q} row_mask:0xf bank_mask:0xf")
        out.append(f"v_fmac_f64_dpp v[{ACC0 + 2}:{ACC0 + 3}], {wreg(p)}, {xr(ps, 1)} row_newbcast:{q} row_mask:0xf bank_mask:0xf")
    out.append("s_set_gpr_idx_off")


def dma(i, out):
    """staging chunk i of the wave: LDS destination in M0 (one wait state before the LDS-DMA reads it), source = tile base + lane offset"""
    if "d" in SKIP:
        out.append("s_nop 0")
        return
    out.append(f"s_add_u32 s30, s17, {i * 8192}")
    if TAB:
        out.append("s_sub_u32 vcc_lo, s27, 0x400")
        out.append("s_min_u32 s30, s30, vcc_lo")
    else:
        out.append("s_min_u32 s30, s30, s31")
    out.append("s_add_u32 m0, s30, s35")
    if WDIRECT:
        # (the chunk's offset goes into the scalar base - VCC is free between two rounds' tops - and every chunk uses the one lane offset v34:
        # the five offset registers of the first form hold the next round's weights now)
        out.append("s_add_u32 vcc_lo, s24, s30")
        out.append("s_addc_u32 vcc_hi, s25, 0")
        out.append("global_load_lds_dwordx4 v34, vcc")
    else:
        out.append("s_nop 0")
        out.append(f"global_load_lds_dwordx4 v{31 + i}, s[24:25]")


def reload_slot(c, out):
    """records of chunk slot c for the NEXT round (s[32:33]: 256 bytes in front of its last record's end)"""
    out.append(f"s_load_dwordx16 s[{R0 + 16 * c}:{R0 + 16 * c + 15}], s[32:33], {hex(64 * c)}")


# ---- ring hand-over by counters (GEN_SYNC=cnt) -------------------------------------------------------------------------------
# One s_barrier per visit makes the 8 waves of an item wait for the slowest at every tile (420 clk of a round's 2 780, lopsided: the
# older wave of a SIMD wins every issue conflict, arrives early and waits while its partner runs alone). What the barrier orders is
# narrower: (1) a tile staged by all 8 waves during visit v - 1 may be read from visit v + 1 on; (2) the buffer of tile v - 3 may be
# overwritten (tile v + 1, during visit v) once all 8 waves have finished visit v - 1. Two arrays of four counters behind the ring:
#   ready[b] += 1 by a wave when its share of the tile in buffer b has landed (certified by the counted vmcnt wait of the boundary
#               one round after the staging round - never a wait for a load of the round it closes);
#   done[b]  += 1 by a wave when it has finished a visit v with (v - 2) % 4 == b (its last read of that tile has returned).
# The counters only grow; the item's prologue presets them as if every visit before the part's first had happened, so the thresholds
# do not depend on the part: entering visit v a wave needs ready[(v - 1) % 4] >= 8 ((v - 1) / 4 + 1); before its staging loads of
# visit v it needs done[(v + 1) % 4] >= 8 ((v - 1) / 4 + 1). A wave can run ahead of the slowest by about a round; both polls are
# issued early (their LDS round trip runs beside other work) and re-polled with s_sleep only when the early value falls short.
# Staging happens in the FIRST round of a visit only (the barrier form re-stages the same tile in every round of a visit so that one
# vmcnt count fits all boundaries).
CNT_READY = 0x400   # byte offsets behind the ring's end (s19): ready[4]; the row of zeros sits in the first KB
CNT_DONE = 0x410


def sync_signal(out, addr_s):
    """one lane adds 1 to the counter at LDS address `addr_s` (v29 / v30: free between a round's top and the next)"""
    a = out.append
    a(f"v_mov_b32 v29, {addr_s}")
    a("v_mov_b32 v30, 1")
    a("s_mov_b64 exec, 1")
    a("ds_add_u32 v29, v30")
    a("s_mov_b64 exec, -1")


def sync_wait(out, tag, addr_s, thr_s, early_reg=None):
    """until counter[addr_s] >= thr_s. early_reg: a register that already holds a value read earlier (its ds_read has returned)"""
    a = out.append
    if early_reg:
        a(f"v_readfirstlane_b32 s13, {early_reg}")
        a(f"s_cmp_ge_u32 s13, {thr_s}")
        a(f"s_cbranch_scc1 LW{tag}OK%=")
    a(f"v_mov_b32 v29, {addr_s}")
    a(f"s_mov_b32 vcc_hi, {SPIN_MAX}")
    a(f"LW{tag}%=:")
    a("ds_read_b32 v30, v29")
    a("s_waitcnt lgkmcnt(0)")
    a("v_readfirstlane_b32 s13, v30")  # (s13: a copy register of the last batch, free everywhere else)
    a(f"s_cmp_ge_u32 s13, {thr_s}")
    a(f"s_cbranch_scc1 LW{tag}OK%=")
    a("s_sleep 1")
    a("s_sub_u32 vcc_hi, vcc_hi, 1")
    a("s_cmp_eq_u32 vcc_hi, 0")
    a(f"s_cbranch_scc0 LW{tag}%=")
    a(f"LW{tag}OK%=:")


def tab_gather(out, next_round):
    """v[26:27] <- the table entries of the 64 records in v28 (lane L: record L of a round's four chunk slots).
    tabo: byte offset = slot x 128 + count x 8 inside the wave's 4 KB of the place table (16 doubles per place).
    tabi: entry = plane count, inner position = ring row + 192 q - (192 if the row's buffer index lies above m), with u' = visit - 1,
    q = u' / 4, m = u' % 4: the record's tile is the latest one at or below u' that sits in its ring buffer (the tiles a visit reads are
    u' and u' - 1). The compare works on the whole record (ring row = its upper half), row x 8 = record >> 13 (counts stay below 32).
    next_round: the visit is the NEXT round's (s31 + bit 8 of its header s34); else the current one (entry: round 0)."""
    a = out.append
    if WSRC == "tabo":
        a("v_and_b32 v30, 0xfc, v28")               # 4 x slot
        a("v_bfe_u32 v29, v28, 8, 8")               # count
        a("v_lshlrev_b32 v29, 3, v29")
        a("v_lshl_add_u32 v30, v30, 5, v29")        # slot x 128 + count x 8
    else:
        if next_round:
            a("s_bfe_u32 s30, s34, 0x10008")        # the next round opens a visit?
            a("s_add_u32 s30, s31, s30")
        else:
            a("s_mov_b32 s30, s31")
        a("s_max_u32 s30, s30, 1")
        a("s_sub_u32 s30, s30, 1")                  # u' (a part's first visit has no records: clamped)
        a("s_lshr_b32 m0, s30, 2")
        a("s_mul_i32 m0, m0, 0x600")                # 192 q rows, in bytes
        a("s_and_b32 s30, s30, 3")
        a("s_add_u32 s30, s30, 1")
        a("s_mul_i32 s30, s30, 0x300000")           # 48 (m + 1) << 16: first ring row of the buffers above m, as a record
        a("v_bfe_u32 v29, v28, 8, 8")               # count
        a("v_mul_lo_u32 v29, v29, v32")             # plane
        a("v_lshrrev_b32 v30, 13, v28")             # ring row x 8
        a("v_cmp_le_u32 vcc, s30, v28")
        a("v_cndmask_b32 v31, 0, v35, vcc")         # -1536: the buffer holds a tile of the ring's previous turn
        a("v_add3_u32 v30, v30, v29, v31")
        a("v_add_u32 v30, m0, v30")
    a("v_add_co_u32 v30, vcc, v38, v30")            # (address pairs are even-aligned on gfx90a and later)
    a("v_addc_co_u32 v31, vcc, 0, v39, vcc")
    a("global_load_dwordx2 v[26:27], v[30:31], off")


def tab_spread(out):
    """v[26:27] (lane L = weight of position L: row c of 16 lanes = chunk slot c) -> v[40 + 2 c : 41 + 2 c] = chunk c's row in all four
    rows. v_permlane16_swap a, b: a's odd rows <-> b's even rows; v_permlane32_swap a, b: a's upper half <-> b's lower half.
    [r0 r1 r2 r3] x 2 -> [r0 r0 r2 r2], [r1 r1 r3 r3] -> copies -> [r0 x 4], [r2 x 4], [r1 x 4], [r3 x 4]. A swap may not read a
    register a vector instruction wrote in the two instructions in front of it (check_swap_hazards)."""
    a = out.append
    w = lambda c, h: f"v{W0 + 2 * c + h}"
    for h in (0, 1):
        a(f"v_mov_b32 {w(0, h)}, v{26 + h}")
        a(f"v_mov_b32 {w(1, h)}, v{26 + h}")
    a(f"v_permlane16_swap_b32 {w(0, 0)}, {w(1, 0)}")
    a("s_nop 0")
    a(f"v_permlane16_swap_b32 {w(0, 1)}, {w(1, 1)}")
    for h in (0, 1):
        a(f"v_mov_b32 {w(2, h)}, {w(0, h)}")
        a(f"v_mov_b32 {w(3, h)}, {w(1, h)}")
    for h in (0, 1):
        a(f"v_permlane32_swap_b32 {w(0, h)}, {w(2, h)}")
        a(f"v_permlane32_swap_b32 {w(1, h)}, {w(3, h)}")


def gen():
    """One round = up to 4 chunks of 16 record positions, worked in batches of BP positions with two batches of row reads in flight.

    What is loaded when (everything for round r + 1 travels while round r computes; the boundary between two rounds waits for nothing
    that was not issued a good part of a round earlier):
      * records (64 scalar registers, chunk slot c = s[36 + 16 c ..]): slot c of the next round is reloaded as soon as this round has
        issued the last FMA that names it (behind F of batch 4 c + 3; slots the round does not enter: in its prologue; slot 3: in front
        of the last batch, whose four accumulator indices are copied to s[13:16] first). Scalar loads return out of order, so only the
        boundary's lgkmcnt(0) certifies them; the counted waits of the row reads in between stay correct (a count that includes scalar
        loads can only wait for MORE row reads than needed, one per scalar load still in flight);
      * weights: four 128-byte loads per round into v[26:33] (lane L: weight L % 16 of each of the four chunk slots that end at the next
        round's last one), moved to v[40:47] by eight v_mov at the next round's start (every row of 16 lanes then holds a chunk's 16
        weights, which v_fmac_f64_dpp row_newbcast picks from);
      * panel tiles: five LDS-DMA loads per wave and round, in the round's last four batches and behind it. A round that is not the
        first of its visit stages the same tile once more (same bytes, same place; nobody reads it before the visit after the next):
        with the same five loads in every round the boundary's vmcnt(5) means the same thing everywhere;
      * headers: three rounds ahead (s28 / s34 / s101 / s12)."""
    o = []
    a = o.append
    # ---- inputs into the fixed registers ----
    # (every operand arrives in a vector register: the compiler has a dozen scalar registers left beside this statement's)
    for name, lo in (("rec", 20), ("rtab", 22), ("src", 24)):
        a(f"v_readfirstlane_b32 s{lo}, %[{name}lo]")
        a(f"v_readfirstlane_b32 s{lo + 1}, %[{name}hi]")
    a("v_readfirstlane_b32 s26, %[nrounds]")  # (uniform values handed over in vector registers: the compiler has few scalar ones left for operands)
    a("v_readfirstlane_b32 s27, %[rowb]")
    a("s_mul_i32 s27, s27, 48")  # bytes of a tile of 48 rows
    if TAB:
        a("v_readfirstlane_b32 s31, %[t0]")  # first visit of the item (tabi: the visit counter); the ring buffer of its first tile follows from it
    else:
        a("v_readfirstlane_b32 s35, %[dst]")  # (uniform values handed over in vector registers: the compiler has few scalar ones left for operands)
    a("v_readfirstlane_b32 s17, %[wave10]")  # (uniform values handed over in vector registers: the compiler has few scalar ones left for operands)
    a("v_readfirstlane_b32 s18, %[lds0]")  # (uniform values handed over in vector registers: the compiler has few scalar ones left for operands)
    a("s_lshl_b32 s19, s27, 2")
    a("s_add_u32 s19, s19, s18")
    if TAB:
        a("s_and_b32 s30, s31, 3")
        a("s_mul_i32 s30, s30, s27")
        a("s_add_u32 s35, s18, s30")
        a("v_mov_b32 v38, %[tablo]")
        a("v_mov_b32 v39, %[tabhi]")
        a("v_mov_b32 v33, %[lane4]")
        if WSRC == "tabi":
            a("v_mov_b32 v32, %[stride8]")
            a("v_mov_b32 v35, 0xfffffa00")
    else:
        a("s_sub_u32 s31, s27, 0x400")
        a("v_mov_b32 v38, %[pwlo]")
        a("v_mov_b32 v39, %[pwhi]")
    a("v_mov_b32 v36, %[ring]")
    a("v_mov_b32 v37, %[rowb]")
    if WDIRECT:
        a("v_lshlrev_b32 v34, 2, %[lane4]")  # 16 x lane: the lane's offset inside a 1 KB staging chunk
    else:
        a("v_and_b32 v30, 60, %[lane4]")  # 4 (lane % 16): ds_bpermute address of this lane's weight inside a chunk
        a("v_lshlrev_b32 v28, 2, %[lane4]")  # 16 x lane (v28: free until the first round)
        for i in range(5):  # source offsets of the wave's five staging chunks: min((wave + 8 i) 1024, tile bytes - 1024) + 16 lane
            a(f"s_add_u32 s30, s17, {i * 8192}")
            a("s_min_u32 s30, s30, s31")
            a(f"v_add_u32 v{31 + i}, s30, v28")
    if PRIO == 1:
        a("s_cmp_ge_u32 s17, 0x1000")
        a("s_cbranch_scc0 LP0%=")
        a("s_setprio 1")
        a("LP0%=:")
    if STAMP:
        for r_ in (6, 7, 8, 9):
            a(f"s_mov_b32 s{r_}, 0")
    a("s_cmp_eq_u32 s26, 0")
    a("s_cbranch_scc1 LDONE%=")
    a("s_load_dwordx2 s[28:29], s[22:23], 0x0")  # headers of rounds 0 and 1
    a("s_load_dword s101, s[22:23], 0x8")        # ... 2
    a("s_waitcnt lgkmcnt(0)")
    a("s_mov_b32 s34, s29")
    a("s_and_b32 s29, s28, 0xff")                # chunks of round 0
    a("s_lshl_b32 s30, s29, 6")
    a("s_add_u32 s20, s20, s30")
    a("s_addc_u32 s21, s21, 0")
    a("s_sub_u32 s32, s20, 0x100")
    a("s_subb_u32 s33, s21, 0")
    for c in range(4):
        reload_slot(c, o)
    if TAB:
        a("global_load_dword v28, v33, s[32:33]")   # the records of round 0, lane L = record L of its four chunk slots
        a("s_waitcnt vmcnt(0)")
        tab_gather(o, next_round=False)             # weights of round 0
        a("s_and_b32 s100, s34, 0xff")
        a("s_lshl_b32 s30, s100, 6")
        a("s_add_u32 vcc_lo, s32, s30")
        a("s_addc_u32 vcc_hi, s33, 0")
        a("s_nop 3")                                # (a vector write of VCC five wait states in front of a vector-memory read of it)
        a("global_load_dword v28, v33, vcc")        # the records of round 1
    elif WDIRECT:
        a("s_lshl_b32 s30, s29, 7")
        a("v_add_co_u32 v38, vcc, s30, v38")
        a("v_addc_co_u32 v39, vcc, 0, v39, vcc")
        for c in range(4):
            a(f"global_load_dwordx2 v[{26 + 2 * c}:{27 + 2 * c}], v[38:39], off offset:{-512 + 128 * c}")  # weights of round 0
    else:
        a("s_lshl_b32 s30, s29, 7")
        a("v_add_co_u32 v38, vcc, s30, v38")
        a("v_addc_co_u32 v39, vcc, 0, v39, vcc")
        a("global_load_dwordx2 v[26:27], v[38:39], off offset:-512")  # weights of round 0
    if CNT:
        # The ring's hand-over counters, preset by wave 0 as if every visit before the item's first (t0) had happened, so that the
        # thresholds do not depend on the part: lanes 0-3 ready[b] = 8 x tiles below t0 that live in buffer b (+ 8 for tile t0 itself:
        # staged by the caller, certified by the wait and the barrier below), lanes 4-7 done[b] = 8 x visits v below t0 with
        # (v - 2) % 4 == b. tiles / visits below t0 congruent to c modulo 4: (t0 + 3 - c) / 4.
        a("s_cmp_eq_u32 s17, 0")
        a("s_cbranch_scc0 LINITX%=")
        a("v_mbcnt_lo_u32_b32 v29, -1, 0")
        a("v_mbcnt_hi_u32_b32 v29, -1, v29")         # lane
        a("v_and_b32 v30, 3, v29")                   # b
        a("v_add_u32 v31, 2, v30")
        a("v_and_b32 v31, 3, v31")                   # (b + 2) % 4
        a("v_cmp_gt_u32 vcc, 4, v29")
        a("v_cndmask_b32 v31, v31, v30, vcc")        # c = b for the ready counters, (b + 2) % 4 for the done counters
        a("s_add_u32 s30, s31, 3")
        a("v_sub_u32 v31, s30, v31")
        a("v_lshrrev_b32 v31, 2, v31")
        a("s_and_b32 s30, s31, 3")
        a("v_cmp_eq_u32 vcc, s30, v29")              # the ready counter of tile t0's buffer (lanes 4-7 never match: t0 % 4 < 4)
        a("v_addc_co_u32 v31, vcc, 0, v31, vcc")
        a("v_lshlrev_b32 v31, 3, v31")
        a("v_lshlrev_b32 v29, 2, v29")
        a(f"s_add_u32 s30, s19, {CNT_READY}")
        a("v_add_u32 v29, s30, v29")
        a("s_mov_b64 exec, 0xff")
        a("ds_write_b32 v29, v31")
        a("s_mov_b64 exec, -1")
        a("LINITX%=:")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")           # ... and the item's first tile (staged by the caller)
    if not PRE:
        if "b" not in SKIP:
            a("s_barrier")                            # round 0 is the first of its visit (the counter form keeps this one: it publishes the item's preset counters)
        a("s_add_u32 s24, s24, s27")
        a("s_addc_u32 s25, s25, 0")
        a("s_add_u32 s35, s35, s27")
        a("s_cmp_eq_u32 s35, s19")
        a("s_cselect_b32 s35, s18, s35")
        if CNT:
            a("s_mov_b32 s29, 0")                     # s29: the ready counter this wave bumps at the next boundary (0: none). The round's chunk count is taken from its header where needed
    else:
        a("s_branch LENTER%=")                        # round 0 is the first of its visit: barrier, then its first row reads

    def advance():
        if "b" not in SKIP and not CNT:
            a("s_barrier")
        a("s_add_u32 s24, s24, s27")
        a("s_addc_u32 s25, s25, 0")
        a("s_add_u32 s35, s35, s27")
        a("s_cmp_eq_u32 s35, s19")
        a("s_cselect_b32 s35, s18, s35")

    def top():
        """what a round starts for the rounds behind it: the next round's weights, the touch of the records two rounds ahead, the header three ahead"""
        a("s_and_b32 s100, s34, 0xff")               # chunks of the next round
        if TAB:
            a("s_lshl_b32 s30, s100, 6")                 # where the next round's four chunk slots come from
            a("s_add_u32 s32, s20, s30")
            a("s_addc_u32 s33, s21, 0")
            a("s_sub_u32 s32, s32, 0x100")
            a("s_subb_u32 s33, s33, 0")
            tab_gather(o, next_round=True)               # the next round's weights, by its records (loaded a round ago)
            a("s_and_b32 s30, s101, 0xff")               # ... and the records of the round after it: its four chunk slots end 64 x its chunks further on
            a("s_lshl_b32 s30, s30, 6")
            a("s_load_dword s12, s[22:23], 0xc")         # header three rounds ahead
            a("s_add_u32 vcc_lo, s32, s30")
            a("s_addc_u32 vcc_hi, s33, 0")
            a("global_load_dword v28, v33, vcc")         # (this load is also what pulls those records into L2 for their scalar loads: the stream form's touch)
            if CNT:
                # first round of a visit: it will stage tile v + 1 over tile v - 3 - the release counter of that buffer, read now, is in v29 when the staging block asks
                a("s_bitcmp1_b32 s28, 8")
                a("s_cbranch_scc0 LT2%=")
                a("s_add_u32 s30, s31, 1")
                a("s_and_b32 s30, s30, 3")
                a("s_lshl_b32 s30, s30, 2")
                a("s_add_u32 s30, s30, s19")
                a("v_mov_b32 v29, s30")
                a(f"ds_read_b32 v29, v29 offset:{CNT_DONE}")
                a("LT2%=:")
            return
        a("s_lshl_b32 s30, s100, 7")
        a("v_add_co_u32 v38, vcc, s30, v38")         # behind the next round's weights
        a("v_addc_co_u32 v39, vcc, 0, v39, vcc")
        if "w" in SKIP:
            a("s_nop 0")
        elif WDIRECT:
            # lane L: weight L % 16 of each of the next round's four chunk slots (v[38:39] carries 8 (L % 16)): the form the FMAs' DPP
            # broadcast reads. Four loads of 128 distinct bytes each where one load + eight ds_bpermute_b32 stood: the permutes cost
            # 1.5 ms per pass (timing experiment GEN_SKIP=p), at the top of every round, in front of its first FMA
            for c in range(4):
                a(f"global_load_dwordx2 v[{26 + 2 * c}:{27 + 2 * c}], v[38:39], off offset:{-512 + 128 * c}")
        else:
            a("global_load_dwordx2 v[26:27], v[38:39], off offset:-512")
        a("s_lshl_b32 s30, s100, 6")                 # where the next round's four chunk slots come from
        a("s_add_u32 s32, s20, s30")
        a("s_addc_u32 s33, s21, 0")
        a("s_sub_u32 s32, s32, 0x100")
        a("s_subb_u32 s33, s33, 0")
        if "t" not in SKIP:
            # touch: the records of the round after the next one into L2 (their scalar loads, issued during the next round, then hit there:
            # without it a boundary waits for slot 3's load to come from HBM, +350 clk per round). An LDS-DMA load into 256 scratch bytes of
            # this wave behind the ring: it needs no register to land in, and it is this round's OLDEST vector-memory operation
            # (v48: an address register; an instruction reads its address operand at issue, so the row reads that named it are past it)
            if WDIRECT:
                a("v_lshrrev_b32 v48, 2, v34")       # 4 x lane: 256 bytes = four chunk slots
            else:
                a("s_min_u32 s30, s17, s31")
                a("v_sub_u32 v48, v31, s30")         # 16 x lane
                a("v_lshrrev_b32 v48, 2, v48")       # 4 x lane: 256 bytes = four chunk slots
            a("s_lshr_b32 s30, s17, 2")
            a("s_add_u32 s30, s30, s19")
            a("s_add_u32 m0, s30, 0x400")            # behind the ring and the row of zeros
            a("s_nop 0")
            a("global_load_lds_dword v48, s[32:33] offset:256")
        if DMA_TOP:
            for i_ in range(5):
                dma(i_, o)
        a("s_load_dword s12, s[22:23], 0xc")         # header three rounds ahead

    def spread_weights():
        if TAB:
            tab_spread(o)
            return
        if WDIRECT:
            for i_ in range(8):
                a(f"v_mov_b32 v{W0 + i_}, v{26 + i_}")   # (vector moves: more than the two wait states a DPP read of them needs lie before the first FMA)
            return
        a("v_mov_b32 v28, v26")
        a("v_mov_b32 v29, v27")
        if "p" in SKIP:  # timing experiment (wrong results): no spreading of the weights through ds_bpermute
            return
        for c in range(4):
            a(f"ds_bpermute_b32 v{W0 + 2 * c}, v30, v28 offset:{64 * c}")
            a(f"ds_bpermute_b32 v{W0 + 2 * c + 1}, v30, v29 offset:{64 * c}")

    def stage_block(tag):
        """counter form: the five staging loads of a visit, in its first round only, once the buffer's last readers are through"""
        a("s_bitcmp1_b32 s28, 8")
        a(f"s_cbranch_scc0 LSTX{tag}%=")
        a("s_max_u32 s30, s31, 1")
        a("s_sub_u32 s30, s30, 1")
        a("s_lshr_b32 s30, s30, 2")
        a("s_add_u32 s30, s30, 1")
        a("s_lshl_b32 s30, s30, 3")                   # 8 ((v - 1) / 4 + 1): every wave has finished visit v - 1
        a("s_cmp_eq_u32 s31, 0")
        a("s_cselect_b32 s30, 0, s30")                # (visit 0 overwrites nothing)
        a("s_add_u32 vcc_lo, s31, 1")
        a("s_and_b32 vcc_lo, vcc_lo, 3")
        a("s_lshl_b32 vcc_lo, vcc_lo, 2")
        a("s_add_u32 vcc_lo, vcc_lo, s19")
        a(f"s_add_u32 vcc_lo, vcc_lo, {CNT_DONE}")
        sync_wait(o, "D" + tag, "vcc_lo", "s30", early_reg="v29")
        for i_ in range(5):
            dma(i_, o)
        a(f"LSTX{tag}%=:")

    def batches():
        for b in range(NB):
            a(f"LS{b}%=:")
            last = b == NB - 1
            a(f"s_waitcnt lgkmcnt({RB})" if not last else "s_waitcnt lgkmcnt(0)")
            if last:
                # the last batch names its accumulators through copies, so that slot 3 can be reloaded before its FMAs instead of behind them
                for j_ in range(BP):
                    a(f"s_mov_b32 s{13 + j_}, {rec(b * BP + j_)}")
                reload_slot(3, o)
                batch_F(b, o, tail=True)
            else:
                batch_F(b, o)
            if b % BPC == BPC - 1 and b // BPC < 3:
                reload_slot(b // BPC, o)
            if CNT:
                if b == NB - 4:
                    stage_block("b")
            elif ONCE:
                if b == NB - 4:
                    a("s_bitcmp1_b32 s28, 8")
                    a("s_cbranch_scc0 LONCEX%=")
                    for i_ in range(5):
                        dma(i_, o)
                    a("LONCEX%=:")
            elif b >= NB - 4 and not DMA_TOP:
                dma(b - (NB - 4), o)
            if b + 2 < NB:
                batch_AL(b + 2, o)
        if not DMA_TOP and not CNT and not ONCE:
            dma(4, o)

    def rotate():
        a("s_waitcnt lgkmcnt(0)")                    # the next round's records and the header
        if WSRC == "tabi" or CNT:
            a("s_bfe_u32 s30, s34, 0x10008")         # the visit counter follows the round that becomes current
            a("s_add_u32 s31, s31, s30")
        a("s_mov_b32 s28, s34")
        a("s_mov_b32 s34, s101")
        a("s_mov_b32 s101, s12")
        a("s_add_u32 s22, s22, 4")
        a("s_addc_u32 s23, s23, 0")
        if not CNT:
            a("s_mov_b32 s29, s100")
        a("s_lshl_b32 s30, s100, 6")
        a("s_add_u32 s20, s20, s30")
        a("s_addc_u32 s21, s21, 0")

    def stamp_barrier_begin():
        if STAMP:
            a("s_memtime s[2:3]")

    def stamp_barrier_end():
        if STAMP:
            a("s_memtime s[4:5]")
            a("s_waitcnt lgkmcnt(0)")
            a("s_sub_u32 s10, s4, s2")
            a("s_add_u32 s8, s8, s10")   # barrier (with GEN_PRE the stamp's lgkmcnt(0) also waits for row reads in flight: an upper bound)

    if not PRE:
        # ---- one round (round-5 first form: everything a round needs is started at its top) ----
        a("LROUND%=:")
        if STAMP:
            a("s_memtime s[2:3]")
        if not ALFIRST:
            spread_weights()
            top()
        nch = "s29"
        if CNT:
            a("s_and_b32 s30, s28, 0xff")
            nch = "s30"
        a(f"s_cmp_eq_u32 {nch}, 4")
        a("s_cbranch_scc1 LPRO0%=")
        a(f"s_cmp_eq_u32 {nch}, 3")
        a("s_cbranch_scc1 LPRO1%=")
        a(f"s_cmp_eq_u32 {nch}, 2")
        a("s_cbranch_scc1 LPRO2%=")
        a(f"s_cmp_eq_u32 {nch}, 1")
        a("s_cbranch_scc1 LPRO3%=")
        # an empty round (a visit nobody has work in yet: the first two of a part)
        if ALFIRST:
            spread_weights()
            top()
        for c in range(4):
            reload_slot(c, o)
        if CNT:
            a("s_waitcnt lgkmcnt(0)")                 # (the early poll)
            stage_block("e")
        elif not DMA_TOP:
            for i_ in range(5):
                dma(i_, o)
        a("s_branch LBND%=")
        # prologues: the slots this round does not enter, then the rows of its first two batches
        for c in (3, 2, 1, 0):
            a(f"LPRO{c}%=:")
            if not ALFIRST:
                for cc in range(c):
                    reload_slot(cc, o)
            batch_AL(BPC * c, o)
            batch_AL(BPC * c + 1, o)
            if ALFIRST:
                spread_weights()
                top()
                for cc in range(c):  # (behind top(): it says where the next round's slots come from)
                    reload_slot(cc, o)
            if c != 0 or ALFIRST:
                a(f"s_branch LS{BPC * c}%=")
        batches()
        # ---- boundary ----
        a("LBND%=:")
        if STAMP:
            a("s_memtime s[4:5]")
            a("s_waitcnt lgkmcnt(0)")
            a("s_sub_u32 s10, s4, s2")
            a("s_add_u32 s6, s6, s10")   # body
            a("s_add_u32 s9, s9, 1")     # rounds
        a("s_sub_u32 s26, s26, 1")
        a("s_cmp_eq_u32 s26, 0")
        a("s_cbranch_scc1 LDONE%=")
        if CNT:
            # early poll for the case that the next round opens a visit: ready counter of the tile that visit reads first (tile v of visit v + 1)
            a("s_and_b32 s30, s31, 3")
            a("s_lshl_b32 s30, s30, 2")
            a("s_add_u32 s30, s30, s19")
            a("v_mov_b32 v31, s30")
            a(f"ds_read_b32 v31, v31 offset:{CNT_READY}")
            # the next round's weights and records; every staging load but the five of THIS round - which only a visit's first round has
            a("s_bitcmp1_b32 s28, 8")
            a("s_cbranch_scc0 LV0%=")
            a("s_waitcnt vmcnt(5)")
            a("s_branch LV1%=")
            a("LV0%=:")
            a("s_waitcnt vmcnt(0)")
            a("LV1%=:")
        elif ONCE:
            a("s_bitcmp1_b32 s28, 8")                    # only a visit's first round has five staging loads behind the next round's weights
            a("s_cbranch_scc0 LV0%=")
            a("s_waitcnt vmcnt(5)")
            a("s_branch LV1%=")
            a("LV0%=:")
            a("s_waitcnt vmcnt(0)")
            a("LV1%=:")
        else:
            a("s_waitcnt vmcnt(5)")                      # the next round's weights; every staging load but this round's five
        if STAMP:
            a("s_memtime s[2:3]")
            a("s_waitcnt lgkmcnt(0)")
            a("s_sub_u32 s10, s2, s4")
            a("s_add_u32 s7, s7, s10")   # wait for vector memory (+ the scalar loads the stamp forces)
        if CNT:
            # the tile this wave staged a round ago has landed (the wait above covered its loads): tell the others
            a("s_cmp_eq_u32 s29, 0")
            a("s_cbranch_scc1 LNP%=")
            sync_signal(o, "s29")
            a("LNP%=:")
            a("s_mov_b32 s29, 0")
            a("s_bitcmp1_b32 s28, 8")                # this round staged tile v + 1: certified at the next boundary
            a("s_cbranch_scc0 LNS%=")
            a("s_add_u32 s30, s31, 1")
            a("s_and_b32 s30, s30, 3")
            a("s_lshl_b32 s30, s30, 2")
            a("s_add_u32 s30, s30, s19")
            a(f"s_add_u32 s29, s30, {CNT_READY}")
            a("LNS%=:")
        if ROTFIRST:
            rotate()
            a("s_bitcmp1_b32 s28, 8")
            a("s_cbranch_scc0 LROUND%=")
            advance()
            stamp_barrier_end()
            a("s_branch LROUND%=")
        a("s_bitcmp1_b32 s34, 8")
        a("s_cbranch_scc0 LNB%=")
        if CNT:
            # this wave has finished visit v: its last read of tile v - 2 has returned (the last batch's lgkmcnt(0))
            a("s_sub_u32 s30, s31, 2")
            a("s_and_b32 s30, s30, 3")
            a("s_lshl_b32 s30, s30, 2")
            a("s_add_u32 s30, s30, s19")
            a(f"s_add_u32 s30, s30, {CNT_DONE}")
            sync_signal(o, "s30")
            # visit v + 1 reads tile v: landed for all 8 waves?
            a("s_lshr_b32 s30, s31, 2")
            a("s_add_u32 s30, s30, 1")
            a("s_lshl_b32 s30, s30, 3")               # 8 (v / 4 + 1)
            a("s_and_b32 vcc_lo, s31, 3")
            a("s_lshl_b32 vcc_lo, vcc_lo, 2")
            a("s_add_u32 vcc_lo, vcc_lo, s19")
            a(f"s_add_u32 vcc_lo, vcc_lo, {CNT_READY}")
            a("s_waitcnt lgkmcnt(0)")                 # (the early poll; the next round's records too: rotate() would wait for them anyway)
            sync_wait(o, "R", "vcc_lo", "s30", early_reg="v31")
        advance()                                     # next round = first of a visit: the tile it reads first has landed for everybody, nobody reads the oldest one any more
        stamp_barrier_end()
        a("LNB%=:")
        rotate()
        a("s_branch LROUND%=")
    else:
        # ---- one round, second form: what the round needs FIRST is started at the boundary in front of it. Entering a round (LENTER:
        # its header in s28, its chunk count in s29, its records certified): weights into their rows; the loads for the rounds behind
        # it; the record slots it will not enter, reloaded; the row reads of its first two batches — in FRONT of the barrier when the
        # header says so (bit 9, set by the builder: none of those 8 positions names the tile that becomes readable behind this barrier
        # in any of the item's 8 waves), else behind it. The latency of those reads and the scalar work then run beside the barrier wait.
        a("LBATCH%=:")
        batches()
        a("LBND%=:")
        if STAMP:
            a("s_memtime s[4:5]")
            a("s_waitcnt lgkmcnt(0)")
            a("s_sub_u32 s10, s4, s2")
            a("s_add_u32 s6, s6, s10")   # body (from the end of the barrier)
            a("s_add_u32 s9, s9, 1")     # rounds
        a("s_sub_u32 s26, s26, 1")
        a("s_cmp_eq_u32 s26, 0")
        a("s_cbranch_scc1 LDONE%=")
        a("s_waitcnt vmcnt(5)")                      # the next round's weights; every staging load but this round's five
        if STAMP:
            a("s_memtime s[2:3]")
            a("s_waitcnt lgkmcnt(0)")
            a("s_sub_u32 s10, s2, s4")
            a("s_add_u32 s7, s7, s10")   # wait for vector memory
        rotate()
        a("LENTER%=:")
        spread_weights()
        top()
        a("s_cmp_eq_u32 s29, 0")
        a("s_cbranch_scc1 LEMPTY%=")
        a("s_bitcmp1_b32 s28, 8")                    # first round of a visit?
        a("s_cbranch_scc0 LPF%=")                    # no: no barrier
        a("s_bitcmp1_b32 s28, 9")                    # its first reads may go ahead of the barrier?
        a("s_cbranch_scc1 LPF%=")
        stamp_barrier_begin()
        advance()
        stamp_barrier_end()
        a("s_andn2_b32 s28, s28, 0x100")             # (the barrier is behind us)
        a("LPF%=:")
        for n, lab in ((4, 0), (3, 1), (2, 2)):
            a(f"s_cmp_eq_u32 s29, {n}")
            a(f"s_cbranch_scc1 LPF{lab}%=")
        for c in (3, 2, 1, 0):
            a(f"LPF{c}%=:")
            for cc in range(c):
                reload_slot(cc, o)
            batch_AL(BPC * c, o)
            batch_AL(BPC * c + 1, o)
            a("s_bitcmp1_b32 s28, 8")                # still a barrier to pass (reads went ahead of it)?
            a(f"s_cbranch_scc0 LS{BPC * c}%=")
            stamp_barrier_begin()
            advance()
            stamp_barrier_end()
            a(f"s_branch LS{BPC * c}%=")
        # an empty round (a visit nobody has work in yet: the first two of a part)
        a("LEMPTY%=:")
        a("s_bitcmp1_b32 s28, 8")
        a("s_cbranch_scc0 LEMPTY2%=")
        stamp_barrier_begin()
        advance()
        stamp_barrier_end()
        a("LEMPTY2%=:")
        for c in range(4):
            reload_slot(c, o)
        if not DMA_TOP:
            for i_ in range(5):
                dma(i_, o)
        a("s_branch LBND%=")
    a("LDONE%=:")
    if PRIO:
        a("s_setprio 0")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")           # nothing of this statement may land in a register later
    if STAMP:
        a("v_mov_b32 v30, 0")
        a("v_mov_b32 v29, 0")
        a("v_readfirstlane_b32 s2, %[stamplo]")
        a("v_readfirstlane_b32 s3, %[stamphi]")
        a("s_mov_b64 s[10:11], exec")
        a("s_mov_b64 exec, 1")  # one lane adds the wave's sums
        for i_, r_ in enumerate((6, 7, 8, 9)):
            a(f"v_mov_b32 v28, s{r_}")
            a(f"global_atomic_add_x2 v30, v[28:29], s[2:3] offset:{8 * i_}")
        # the barrier wait by wave number (8 more sums behind the four)
        a("s_lshr_b32 s12, s17, 7")  # 8 x wave
        a("v_mov_b32 v30, s12")
        a("v_mov_b32 v28, s8")
        a("global_atomic_add_x2 v30, v[28:29], s[2:3] offset:32")
        a("s_waitcnt vmcnt(0)")
        a("s_mov_b64 exec, s[10:11]")
    return o



# ---- the flow layout's round loop (GEN_FLOW=1; tiles_flow.inc) --------------------------------------------------------------------
# Every round is 4 chunks of 16 record positions of ONE wave's stream; the ring of 6 tiles of 32 rows is handed over at TICKS in front
# of chunks (round header: ticks in front of chunk j in bits 2j+1:2j, their sum in bits 11:8). One array of six counters behind the
# ring, P[u % 6] += 1 when a wave passes tick K(u). Passing K(v) means: the wave has released tile v - 2 (its records lie in front of
# the tick) and its share of tile v + 1 - staged by its previous tick - has landed (a counted vmcnt wait: behind those loads only the
# two loads of every round top since). At K(v) a wave
#   A (in front of the chunk): certifies + announces (one ds_add), reads P[(v - 1) % 6], stages its four KB of tile v + 2 over tile
#     v - 4 - allowed because its previous tick saw every wave past K(v - 2) - and goes on with the chunk's first 12 positions, which the
#     builder keeps clear of tile v;
#   B (one batch later, when the counted LDS wait has covered that read): checks P[(v - 1) % 6] >= 8 ((v - 1) / 6 + 1): every wave is past
#     K(v - 1), so tile v has landed for all of them and tile v - 3's readers are gone. Polls with s_sleep only if not.
# A wave may run one tile ahead of the slowest; no wait sits on a tick's fast path except the vmcnt one.
# Timing experiments on the first form (profiles/HISTORY.md): the round loop WITHOUT ticks runs a pass in 10.6 / 11.2 ms against the
# dense layout's 14.4 / 15.8 - the ticks are what the flow form costs.
FLOW = (os.environ.get("GEN_FLOW") or "0") != "0"
FSKIP = set((os.environ.get("GEN_FSKIP") or "").split(","))  # timing experiments on the flow loop (wrong results): p = no counter check, v = no vmcnt wait in a tick, d = no staging loads, t = no ticks at all
FL_P = 0x400      # byte offset of P[6] behind the ring's end (s19); the row of zeros sits in the first KB
FL_GUARD_BATCHES = 3  # batches of a chunk worked before phase B: the builder's FL_GUARD = 4 x this
MAGIC6 = "0xaaaaaaab"  # x / 6 = (x * 0xaaaaaaab) >> 34


def flow_div6(out, q, r, x, tmp):
    """q = x / 6, r = x % 6 (scalar registers; r may be x)"""
    a = out.append
    a(f"s_mul_hi_u32 {q}, {x}, {MAGIC6}")
    a(f"s_lshr_b32 {q}, {q}, 2")
    a(f"s_mul_i32 {tmp}, {q}, 6")
    a(f"s_sub_u32 {r}, {x}, {tmp}")


def flow_gather(out, vs_from_header):
    """v[26:27] <- the table entries of the 64 records in v28 (temporaries: v28 itself, v32 / v33 - a tick's registers, idle at a round's top). tabi: a
    round's records name tiles vs - 2 .. vs + 3 (vs = the tile the round's first tick opens; at most 4 ticks per round), 6 consecutive
    ones, so the ring buffer (ring row / 32) says which: with u = vs - 2 = 6 q + m, buffer b holds tile 6 q + b when b >= m and
    6 (q + 1) + b otherwise. vs_from_header: the NEXT round's (s31 + the ticks of the current one); else the current s31 (entry)."""
    a = out.append
    if WSRC == "tabo":
        a("v_and_b32 v32, 0xfc, v28")               # 4 x slot
        a("v_bfe_u32 v28, v28, 8, 8")               # count (the records are done with: in place)
        a("v_lshlrev_b32 v28, 3, v28")
        a("v_lshl_add_u32 v32, v32, 5, v28")        # slot x 128 + count x 8
    else:
        if vs_from_header:
            a("s_bfe_u32 s30, s28, 0x40008")
            a("s_add_u32 s30, s31, s30")
        else:
            a("s_mov_b32 s30, s31")
        a("s_max_u32 s30, s30, 2")
        a("s_sub_u32 s30, s30, 2")                  # u
        flow_div6(out, "m0", "s30", "s30", "vcc_lo")  # q, m
        a("s_mul_i32 m0, m0, 0x600")                # 192 q rows, in bytes
        a("s_lshl_b32 s30, s30, 21")                # (32 m) << 16: first ring row of buffer m, as a record
        a("v_lshrrev_b32 v32, 13, v28")             # ring row x 8
        a("v_mov_b32 v33, 0x600")
        a("v_cmp_gt_u32 vcc, s30, v28")             # a buffer below m: the ring's next turn
        a("v_cndmask_b32 v33, 0, v33, vcc")
        a("v_bfe_u32 v28, v28, 8, 8")               # count (the records are done with: in place)
        a("v_mul_lo_u32 v28, v28, s101")            # plane
        a("v_add3_u32 v32, v32, v28, v33")
        a("v_add_u32 v32, m0, v32")
    a("v_add_co_u32 v32, vcc, v38, v32")
    a("v_addc_co_u32 v33, vcc, 0, v39, vcc")
    a("global_load_dwordx2 v[26:27], v[32:33], off")


def flow_spin(out, tag):
    """until P[(v - 1) % 6] >= s13 for the tick just worked (s31 = v + 1): the slow path of a tick's check"""
    a = out.append
    a("s_sub_u32 s14, s31, 2")                      # v - 1 (a tick with v = 0 never comes here: its threshold is 0)
    flow_div6(out, "s15", "s14", "s14", "s16")
    a("s_lshl_b32 s14, s14, 2")
    a("s_add_u32 s14, s14, s19")
    a("v_mov_b32 v33, s14")
    a(f"s_mov_b32 vcc_hi, {SPIN_MAX}")
    a(f"LSP{tag}%=:")
    a("s_sleep 1")
    a(f"ds_read_b32 v32, v33 offset:{FL_P}")
    a("s_waitcnt lgkmcnt(0)")
    a("v_readfirstlane_b32 s16, v32")
    a("s_cmp_ge_u32 s16, s13")
    a(f"s_cbranch_scc1 LSP{tag}X%=")
    a("s_sub_u32 vcc_hi, vcc_hi, 1")
    a("s_cmp_eq_u32 vcc_hi, 0")
    a(f"s_cbranch_scc0 LSP{tag}%=")
    a(f"LSP{tag}X%=:")


def flow_tick_a(out, tag):
    """phase A of tick K(v), v = s31 (see above). Leaves the check's threshold in s13 and the counter as read in v32; v += 1.
    Temporaries: s13-s16 (the last batch's copy registers: no tick stands inside a batch), s30, m0, v33."""
    a = out.append
    if "v" not in FSKIP:  # the previous tick's staging loads have landed
        a("s_cmp_eq_u32 s29, 0")
        a(f"s_cbranch_scc1 LC{tag}0%=")
        a("s_cmp_eq_u32 s29, 1")
        a(f"s_cbranch_scc1 LC{tag}1%=")
        a("s_waitcnt vmcnt(4)")
        a(f"s_branch LC{tag}X%=")
        a(f"LC{tag}1%=:")
        a("s_waitcnt vmcnt(2)")
        a(f"s_branch LC{tag}X%=")
        a(f"LC{tag}0%=:")
        a("s_waitcnt vmcnt(0)")
        a(f"LC{tag}X%=:")
    if "m" in FSKIP:  # timing experiment: no address / threshold arithmetic (what a tick would cost with that work done in the shadow of the FMAs in front of it)
        a("v_mov_b32 v33, s19")
        a("s_mov_b64 exec, 1")
        a(f"ds_add_u32 v33, v34 offset:{FL_P}")
        a("s_mov_b64 exec, -1")
        a("v_mov_b32 v32, s19")
        a(f"ds_read_b32 v32, v32 offset:{FL_P}")
        a("s_mov_b32 s13, 0")
    else:
        flow_div6(out, "s13", "s14", "s31", "s30")      # q, b
        a("s_lshl_b32 s15, s14, 2")
        a("s_add_u32 s15, s15, s19")                    # P[b]
        a("v_mov_b32 v33, s15")
        a("s_mov_b64 exec, 1")
        a(f"ds_add_u32 v33, v34 offset:{FL_P}")         # this wave is past K(v)
        a("s_mov_b64 exec, -1")
        a("s_sub_u32 s15, s15, 4")                      # P[(v - 1) % 6]: the counter in front, or the last one
        a("s_add_u32 s16, s19, 20")
        a("s_cmp_eq_u32 s14, 0")
        a("s_cselect_b32 s15, s16, s15")
        a("s_cselect_b32 s16, 0, 1")
        a("v_mov_b32 v32, s15")
        a(f"ds_read_b32 v32, v32 offset:{FL_P}")
        a("s_add_u32 s13, s13, s16")                    # (v - 1) / 6 + 1 = q + 1, or q when b = 0 (0 for v = 0: nothing to wait for)
        a("s_lshl_b32 s13, s13, 3")
    # this wave's KBs of tile v + 2: chunks wave, wave + 8, ... below the tile's end - s17 of them, 3 at 100 columns (4 for wave 0); the first form
    # issued 4 from every wave, the surplus rewriting the tile's last KB: 32 loads for 25 KB (s12 / s32 / s33 / s100: the chunks' offsets
    # inside a tile; v29 / v30 / v31 / v35: the same + 16 x lane)
    a("s_add_u32 s31, s31, 1")
    a("s_mov_b32 s29, 0")
    dm = (("s12", "v29"), ("s32", "v30"), ("s33", "v31"), ("s100", "v35"))
    for i in (3, 2, 1, 0):
        off, vo = dm[i]
        a(f"s_cmp_lt_u32 s17, {i + 1}")
        a(f"s_cbranch_scc1 LDM{tag}{i}%=")
        a(f"s_add_u32 m0, s35, {off}")
        a("s_nop 0")                                # (one wait state between a write of M0 and the LDS-DMA that reads it)
        a(f"global_load_lds_dwordx4 {vo}, s[24:25]" if "d" not in FSKIP else "s_nop 0")
        a(f"LDM{tag}{i}%=:")
    a("s_add_u32 s24, s24, s27")
    a("s_addc_u32 s25, s25, 0")
    a("s_add_u32 s35, s35, s27")
    a("s_cmp_eq_u32 s35, s19")
    a("s_cselect_b32 s35, s18, s35")


def flow_tick_b(out, tag):
    """phase B: the counter read in phase A (v32; its LDS read has returned) against the threshold in s13"""
    a = out.append
    if "p" in FSKIP:
        return
    a("v_readfirstlane_b32 s16, v32")
    a("s_cmp_ge_u32 s16, s13")
    a(f"s_cbranch_scc1 LB{tag}X%=")
    flow_spin(out, f"B{tag}")
    a(f"LB{tag}X%=:")


def gen_flow():
    """The flow layout's round loop. Registers as in the table forms of gen() except: s31 = the tile the next tick opens; s29 = round
    tops since the last tick; s12 / s32 / s33 / s100 = this wave's four chunk offsets inside a tile and v29 / v30 / v31 / v35 the same
    + 16 x lane (the staging loads' addresses); s101 = tabi: bytes between two planes; s34 the next round's header; s[20:21] = END of
    the current round's records (= where the next round's begin); v32 a tick's counter as read; v33 a tick's temporary; v34 = 1."""
    assert TAB and WDIRECT and BP == 4 and not STAMP
    o = []
    a = o.append
    for name, lo in (("rec", 20), ("rtab", 22), ("src", 24)):
        a(f"v_readfirstlane_b32 s{lo}, %[{name}lo]")
        a(f"v_readfirstlane_b32 s{lo + 1}, %[{name}hi]")
    a("v_readfirstlane_b32 s26, %[nrounds]")
    a("v_readfirstlane_b32 s27, %[rowb]")
    a("s_lshl_b32 s27, s27, 5")                  # bytes of a tile of 32 rows
    a("v_readfirstlane_b32 s31, %[t0]")
    a("v_readfirstlane_b32 s17, %[wave10]")
    a("v_readfirstlane_b32 s18, %[lds0]")
    a("s_mul_i32 s19, s27, 6")
    a("s_add_u32 s19, s19, s18")
    a("s_add_u32 s30, s31, 2")                   # tile t0 + 2 is staged first: its ring buffer
    flow_div6(o, "s13", "s14", "s30", "s15")
    a("s_mul_i32 s14, s14, s27")
    a("s_add_u32 s35, s18, s14")
    a("s_sub_u32 s30, s27, 0x400")
    a("v_lshlrev_b32 v34, 2, %[lane4]")          # 16 x lane: the lane's offset inside a 1 KB staging chunk
    for i, (r_, vo) in enumerate((("s12", "v29"), ("s32", "v30"), ("s33", "v31"), ("s100", "v35"))):
        a(f"s_add_u32 {r_}, s17, {i * 8192}")
        a(f"s_min_u32 {r_}, {r_}, s30")
        a(f"v_add_u32 {vo}, {r_}, v34")
    a("v_mov_b32 v38, %[tablo]")
    a("v_mov_b32 v39, %[tabhi]")
    a("v_mov_b32 v33, %[lane4]")                 # (4 x lane: the address of the record loads below; a tick's temporary afterwards - the round top makes it again)
    a("s_mov_b32 s16, s17")                      # (wave << 10, once more: wave 0 presets the counters below)
    # s17 = the chunks of a tile this wave stages: chunks wave + 8 i below ceil(tile bytes / 1 KB)
    a("s_add_u32 s30, s27, 0x3ff")
    a("s_lshr_b32 s30, s30, 10")                 # chunks of a tile
    a("s_lshr_b32 s15, s17, 10")                 # wave
    a("s_add_u32 s30, s30, 7")
    a("s_sub_u32 s30, s30, s15")
    a("s_lshr_b32 s17, s30, 3")                  # (chunks + 7 - wave) / 8
    a("s_min_u32 s17, s17, 4")
    if WSRC == "tabi":
        a("v_readfirstlane_b32 s101, %[stride8]")
    a("v_mov_b32 v36, %[ring]")
    a("v_mov_b32 v37, %[rowb]")
    a("s_mov_b32 s29, 0")                        # round tops since the last tick
    a("s_cmp_eq_u32 s26, 0")
    a("s_cbranch_scc1 LDONE%=")
    a("s_load_dword s28, s[22:23], 0x0")         # header of round 0
    for c in range(4):
        a(f"s_load_dwordx16 s[{R0 + 16 * c}:{R0 + 16 * c + 15}], s[20:21], {hex(64 * c)}")
    a("global_load_dword v28, v33, s[20:21]")    # the records of round 0, lane L = record L
    a("s_add_u32 s20, s20, 0x100")
    a("s_addc_u32 s21, s21, 0")
    # The counters, preset by wave 0 as if every tick before K(t0) had happened (so that the thresholds do not depend on the part):
    # P[b] = 8 x ticks u < t0 with u % 6 == b = 8 ((t0 + 5 - b) / 6). (Tiles t0 and t0 + 1 were staged by the caller and are certified by the
    # wait and the barrier below; every wave's first tick announces tile t0 + 1 like any tile "staged by the tick before".)
    a("s_cmp_eq_u32 s16, 0")
    a("s_cbranch_scc0 LINITX%=")
    a("v_mbcnt_lo_u32_b32 v56, -1, 0")
    a("v_mbcnt_hi_u32_b32 v56, -1, v56")         # lane = b
    a("s_add_u32 s30, s31, 5")
    a("v_sub_u32 v57, s30, v56")
    a(f"s_mov_b32 s30, {MAGIC6}")
    a("v_mul_hi_u32 v57, v57, s30")
    a("v_lshrrev_b32 v57, 2, v57")
    a("v_lshlrev_b32 v57, 3, v57")
    a("v_lshlrev_b32 v56, 2, v56")
    a("v_add_u32 v56, s19, v56")
    a("s_mov_b64 exec, 0x3f")
    a(f"ds_write_b32 v56, v57 offset:{FL_P}")
    a("s_mov_b64 exec, -1")
    a("LINITX%=:")
    a("v_mov_b32 v34, 1")                        # (what a tick adds to a counter)
    a("s_waitcnt vmcnt(0)")                      # (round 0's records as a vector; the caller's two tiles)
    flow_gather(o, vs_from_header=False)         # weights of round 0
    a("v_subrev_u32 v33, s12, v29")              # (the gather used v33)
    a("v_lshrrev_b32 v33, 2, v33")               # 4 x lane
    a("global_load_dword v28, v33, s[20:21]")    # the records of round 1
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a("s_barrier")                               # the only one: it publishes the preset counters, the row of zeros and the first two tiles

    a("LROUND%=:")
    if ALFIRST:
        batch_AL(0, o)
        batch_AL(1, o)
    tab_spread(o)
    a("s_add_u32 s29, s29, 1")
    a("s_load_dword s34, s[22:23], 0x4")         # the next round's header
    flow_gather(o, vs_from_header=True)          # the next round's weights, by its records (loaded a round ago)
    a("v_subrev_u32 v33, s12, v29")              # 16 x lane
    a("v_lshrrev_b32 v33, 2, v33")               # 4 x lane
    a("global_load_dword v28, v33, s[20:21] offset:256")  # the records of the round after it (this load also pulls them into L2 for their scalar loads)
    if not ALFIRST:
        batch_AL(0, o)
        batch_AL(1, o)
    for b in range(NB):
        a(f"LS{b}%=:")
        j = b // BPC
        fld = hex((2 << 16) | (2 * j))
        if b % BPC == 0 and "t" not in FSKIP:
            a(f"s_bfe_u32 s30, s28, {fld}")
            a("s_cmp_eq_u32 s30, 0")
            a(f"s_cbranch_scc0 LTK{j}%=")
            a(f"LTKR{j}%=:")
        last = b == NB - 1
        a(f"s_waitcnt lgkmcnt({RB})" if not last else "s_waitcnt lgkmcnt(0)")
        if b % BPC == FL_GUARD_BATCHES - 2 and "t" not in FSKIP:
            # phase B of a tick in front of this chunk: its counter read lies behind this batch's rows and in front of the next one's, the
            # wait above has covered it; the rows of the batch after the next are issued below, behind the check
            a(f"s_bfe_u32 s30, s28, {fld}")
            a("s_cmp_eq_u32 s30, 0")
            a(f"s_cbranch_scc1 LNB{j}%=")
            flow_tick_b(o, f"{j}")
            a(f"LNB{j}%=:")
        if last:
            for j_ in range(BP):
                a(f"s_mov_b32 s{13 + j_}, {rec(b * BP + j_)}")
            a(f"s_load_dwordx16 s[{R0 + 48}:{R0 + 63}], s[20:21], 0xc0")
            batch_F(b, o, tail=True)
        else:
            batch_F(b, o)
        if b % BPC == BPC - 1 and b // BPC < 3:
            c = b // BPC
            a(f"s_load_dwordx16 s[{R0 + 16 * c}:{R0 + 16 * c + 15}], s[20:21], {hex(64 * c)}")
        if b + 2 < NB:
            batch_AL(b + 2, o)
    # ---- boundary ----
    # the next round's weights and records must have arrived; behind them (newer) only the staging loads of this round's ticks, at least 3
    # each (a wave that stages fewer - panels below 72 columns - waits for everything)
    a("s_bfe_u32 s30, s28, 0x40008")
    a("s_cmp_lt_u32 s17, 3")
    a("s_cselect_b32 s30, 0, s30")
    for n, lab in ((0, "LV0"), (1, "LV1"), (2, "LV2")):
        a(f"s_cmp_eq_u32 s30, {n}")
        a(f"s_cbranch_scc1 {lab}%=")
    a("s_waitcnt vmcnt(9)")
    a("s_branch LVX%=")
    a("LV2%=:")
    a("s_waitcnt vmcnt(6)")
    a("s_branch LVX%=")
    a("LV1%=:")
    a("s_waitcnt vmcnt(3)")
    a("s_branch LVX%=")
    a("LV0%=:")
    a("s_waitcnt vmcnt(0)")
    a("LVX%=:")
    a("s_sub_u32 s26, s26, 1")
    a("s_cmp_eq_u32 s26, 0")
    a("s_cbranch_scc1 LDONE%=")                  # (every tile another wave still reads was announced by the tick behind the one that staged it)
    a("s_waitcnt lgkmcnt(0)")                    # the next round's records and header
    a("s_mov_b32 s28, s34")
    a("s_add_u32 s22, s22, 4")
    a("s_addc_u32 s23, s23, 0")
    a("s_add_u32 s20, s20, 0x100")
    a("s_addc_u32 s21, s21, 0")
    a("s_branch LROUND%=")
    # ---- ticks in front of chunk j (out of line) ----
    for j in range(4):
        fld = hex((2 << 16) | (2 * j))
        a(f"LTK{j}%=:")
        flow_tick_a(o, f"{j}")
        a(f"s_bfe_u32 s30, s28, {fld}")          # ticks still standing in front of this chunk, the one just worked included
        a("s_cmp_eq_u32 s30, 1")
        a(f"s_cbranch_scc1 LTKR{j}%=")           # the last one: its check follows a batch later (phase B)
        a(f"s_sub_u32 s28, s28, {hex(1 << (2 * j))}")  # one more to go: counted down inside the header (nothing else reads this field; the sum in bits 11:8 stays)
        a("s_waitcnt lgkmcnt(0)")                # ... and this one's check right here: the next tick stages on the strength of it
        flow_tick_b(o, f"M{j}")
        a(f"s_branch LTK{j}%=")
    a("LDONE%=:")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")           # nothing of this statement may land in a register later
    return o


def check_dpp_hazards(lines):
    """gfx950: a VALU write of a VGPR needs two wait states before a DPP instruction reads it, a VALU write of EXEC five. The compiler's
    hazard recognizer does not look into an asm statement, so the stream is checked here: the DPP sources of every
    v_fmac_f64_dpp (weights v40-v47: written by ds_bpermute_b32; rows v56-v87: written by ds_read_b128) must not be the destination
    of one of the two vector-ALU instructions in front of it, and no vector instruction may write EXEC at all."""
    import re

    def regs(tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r"v(\d+)", tok)
        return {int(m.group(1))} if m else set()

    insts = [ln for ln in lines if not ln.endswith(":")]
    for i, ln in enumerate(insts):
        ops = [t.strip() for t in ln.split(None, 1)[1].split(",")] if " " in ln else []
        if ln.startswith("v_") and ops and ops[0].startswith("exec"):
            raise SystemExit(f"vector write of EXEC: {ln}")
        if not ln.startswith("v_fmac_f64_dpp"):
            continue
        src = regs(ops[1]) | regs(ops[2].split()[0])
        for back in (1, 2):
            if i - back < 0:
                break
            prev = insts[i - back]
            if not prev.startswith("v_") or prev.startswith("v_readfirstlane"):
                continue
            pops = [t.strip() for t in prev.split(None, 1)[1].split(",")]
            if regs(pops[0]) & src:
                raise SystemExit(f"DPP hazard: '{prev}' writes a source of '{ln}' {back} instruction(s) earlier")


def check_swap_hazards(lines):
    """gfx950: v_permlane16_swap_b32 / v_permlane32_swap_b32 read (and write) both operands; a vector-ALU write of either needs two wait
    states before the swap (the compiler puts s_nop 1 there; it does not look into an asm statement)."""
    import re

    insts = [ln for ln in lines if not ln.endswith(":")]
    for i, ln in enumerate(insts):
        if not ln.startswith("v_permlane"):
            continue
        ops = {t.strip() for t in ln.split(None, 1)[1].split(",")}
        states = 0
        for back in range(1, 4):
            if i - back < 0 or states >= 2:
                break
            prev = insts[i - back]
            m = re.fullmatch(r"s_nop (\d+)", prev)
            if prev.startswith("v_") and not prev.startswith("v_readfirstlane"):
                dst = {t.strip() for t in prev.split(None, 1)[1].split(",")[: 2 if prev.startswith("v_permlane") else 1]}
                if dst & ops:
                    raise SystemExit(f"swap hazard: '{prev}' writes an operand of '{ln}' {states} wait state(s) earlier")
            states += 1 + (int(m.group(1)) if m else 0)


def main():
    lines = gen_flow() if FLOW else gen()
    check_dpp_hazards(lines)
    check_swap_hazards(lines)
    here = os.path.dirname(os.path.abspath(__file__))
    name = "tile_dense_body.inc" if WSRC == "stream" else f"tile_dense_body_{WSRC}{'_cnt' if CNT else ''}.inc"
    if FLOW:
        name = f"tile_flow_body_{WSRC}.inc"
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "scan-rs_amd", "csrc", name)
    with open(path, "w") as f:
        f.write("// generated by tools/gen_tile_dense_asm.py - do not edit\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    print(f"{len(lines)} instructions / labels -> {os.path.normpath(path)}")


if __name__ == "__main__":
    main()
