// tiles.hip — the hybrid sparse x dense product for panels of up to 104 columns: LDS-staged panel tiles for the bulk of the
// nonzeros AND the L2-blocked row gather (kernels.hip) for the rest, both at the same time.
//
// Why two pipes. The gather form pulls one 800-byte panel row per nonzero through the CU's texture addresser (64 B/clk,
// busy 91-96 % of a launch: DESIGN.md section 4) while the LDS (256 B/clk for ds_read_b128) idles; an LDS-only form (round 2,
// quad.hip) pays for every nonzero that does not fit its fixed record structure. Here the matrix is SPLIT once per map:
//   * tile part: the panel is walked in tiles of T rows through a RING of B tile buffers in LDS (B T <= 192 rows of 800 B);
//     visit v of an outer vector happens when tile v has just landed, with tiles v-B+2 .. v resident, and owns exactly K
//     record positions; the vector's nonzeros are dealt to the positions first come first served — a nonzero of tile t may
//     be worked at any visit t .. t+B-2 — as (ring row u8, weight f64); the weight is the whole map chain evaluated once at
//     build time (so a padded position costs no logarithm), unused positions hold weight 0. Letting a nonzero wait for up
//     to B-2 visits is what keeps the overflow small at a fixed number of positions per nonzero: at 3 % density, T = 48,
//     K = 2, B = 4 leave 5.7 % of the nonzeros over (K = 4 positions per 96-row tile with two buffers: 13.5 %);
//   * overflow part: the nonzeros no visit had room for form an ordinary compressed matrix of the same shape (indptr /
//     indices / f64 weights) that the L2-blocked gather kernel walks unchanged.
// product = tile kernel (this file) + gather over the overflow part, launched on two streams so that their workgroups share
// the CUs: the tile kernel is PERSISTENT (one workgroup of 8 waves per CU, resident for the whole product, all of the LDS,
// <= 192 VGPRs so that 2 waves / SIMD leave registers for gather waves); the gather kernel's workgroups fill what is left
// of every SIMD. The tile kernel keeps the LDS pipe and the f64 FMA pipe busy, the gather kernel the texture addresser.
//
// Tile kernel: a wave owns S outer vectors ("slots") of one group for a whole item (a range of tiles): 2 f64 accumulators
// per lane per slot (lane = column pair), every FMA destination a compile-time register because the code is unrolled over
// the positions; the 64 lanes of a record set hold the set's positions (slot q, position j -> lane j SPS + q), so the serial
// part reads them with v_readlane at immediate lanes (the four row bytes of a lane quad with ONE v_readlane, taken apart by
// scalar instructions: v_readlane is the slowest instruction of the loop, profiles/microbench/issue_bench.hip); tiles are
// staged by LDS-DMA (global_load_lds, one barrier per visit), a few chunks at a time between the positions. The positions
// of a visit run as a software pipeline (readlane / address / ds_read_b128 / FMA of four different positions per step).
// Sums are bit-reproducible: positions in order, visits ascending, parts and the overflow sum added in a fixed order.
#include "common.hpp"
#include "device_map.hpp"

#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace scanrs {

namespace {

#ifndef TL_NW_DEF
#define TL_NW_DEF 8
#endif
constexpr uint32_t TL_NW = TL_NW_DEF;  // waves per workgroup (8: 2 per SIMD; 12 with the 168-register kernel: 3 per SIMD, an experiment)
constexpr uint32_t TL_LMAX = 104;      // widest panel: 192 ring rows x 104 x 8 B = 159744 B of LDS
constexpr uint32_t TL_LDS = 160u << 10;
constexpr int TL_W = 6;                // LDS row reads in flight per wave
#ifndef TL_SROWS
#define TL_SROWS 1
#endif
#ifndef TL_DPPW
#define TL_DPPW 1
#endif
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
constexpr uint32_t TL_TABC = 8;        // unit mode: counts 1 .. 8 take their weight from the per-position quotient table (one 64-byte line per position)

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef __attribute__((address_space(3))) const char *lds_cptr_t;

struct TileShape {
    uint32_t T, B, K, KU, S, sps, nset, nt, n_parts, tpp; // KU of the K positions of a (slot, visit) pair are "unit" positions
};
struct DenseItem; // tiles_dense.inc

static inline dim3 grid_1d(uint64_t n) { // one work-item per element; a dispatch holds fewer than 2^32 of them
    const uint64_t blocks = (n + 255) / 256;
    if (blocks * 256 >= (1ull << 32)) fail(SCANRS_ERR_DEVICE, "a per-element launch over %llu elements exceeds the 2^32 work-items of one dispatch", (unsigned long long)n);
    return dim3((unsigned)blocks);
}

// ---- builder -------------------------------------------------------------------------------------------------------
// One thread per (outer vector, part): walks the part's nonzeros in order and deals them to visits first come first served.
// FILL = false: counts the nonzeros left over (ovc[outer * n_parts + part]); FILL = true: writes records and overflow.
// Record index of (group g, visit v, set b, lane): ((g nt + v) nset + b) 64 + lane, lane = j sps + q for position j of slot
// b sps + q (position-major: the weights of the general positions of a set are contiguous, so the weight refresh writes
// whole cache lines).
// The layout is STRUCTURE (this kernel: which nonzero sits where; row bytes and raw counts) and WEIGHTS (tile_weights_kernel:
// the map chain evaluated per position) — a re-normalized handle keeps the structure and refreshes the weights in one
// streaming pass.
template <bool FILL>
__global__ __launch_bounds__(256) void tile_assign_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                          const uint32_t *__restrict__ values, uint64_t n_outer, TileShape sh,
                                                          unsigned long long *__restrict__ ovc, const unsigned long long *__restrict__ ov_off,
                                                          uint16_t *__restrict__ prow, uint8_t *__restrict__ pcnt,
                                                          uint32_t *__restrict__ ov_indices, uint32_t *__restrict__ ov_values) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_outer * sh.n_parts) return;
    const uint64_t o = e / sh.n_parts;
    const uint32_t part = (uint32_t)(e % sh.n_parts);
    const uint32_t t0 = part * sh.tpp, t1 = min(sh.nt, t0 + sh.tpp);
    const uint64_t s = indptr[o], end = indptr[o + 1];
    // first nonzero of the part
    uint64_t lo = s, hi = end;
    const uint64_t key = (uint64_t)t0 * sh.T;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if ((uint64_t)indices[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    const uint64_t g = o / sh.S;
    const uint32_t sl = (uint32_t)(o % sh.S), b = sl / sh.sps, q = sl % sh.sps;
    const uint64_t idx_end = (uint64_t)t1 * sh.T;
    // Two first-come-first-served queues: the KU unit positions of a visit take only nonzeros of count 1 (whose weight is the
    // product of a per-outer and a per-inner factor: the kernel adds their panel rows without a weight), the K - KU general
    // positions take anything. A count-1 nonzero goes where it is served first (unit queue on a tie).
    const uint32_t KG = sh.K - sh.KU;
    uint32_t vu = t0, cu = sh.KU, vg = t0, cg = KG;
    unsigned long long n_ov = 0;
    unsigned long long op = FILL ? ov_off[e] : 0ull;
    for (uint64_t p = lo; p < end; p++) {
        const uint32_t idx = indices[p];
        if ((uint64_t)idx >= idx_end) break;
        const uint32_t tau = idx / sh.T, cnt = values[p];
        const uint32_t last = min(tau + sh.B - 2u, t1 - 1u); // last visit with tile tau in the ring (and inside the part)
        // where each queue would serve it
        uint32_t au = vu, bu = cu, ag = vg, bg = cg;
        if (tau > au) {
            au = tau;
            bu = sh.KU;
        }
        if (bu == 0) {
            au++;
            bu = sh.KU;
        }
        if (tau > ag) {
            ag = tau;
            bg = KG;
        }
        if (bg == 0) {
            ag++;
            bg = KG;
        }
        const bool can_u = sh.KU > 0 && cnt == 1u && au <= last;
        const bool can_g = KG > 0 && cnt <= 255u && ag <= last;
        if (!can_u && !can_g) { // no visit has room while tile tau is in the ring (or the count does not fit a byte): overflow part
            if (FILL) {
                ov_indices[op] = idx;
                ov_values[op] = cnt;
                op++;
            } else {
                n_ov++;
            }
            continue;
        }
        const bool use_u = can_u && (!can_g || au <= ag);
        uint32_t v, j;
        if (use_u) {
            v = au;
            j = sh.KU - bu;
            vu = au;
            cu = bu - 1u;
        } else {
            v = ag;
            j = sh.KU + (KG - bg);
            vg = ag;
            cg = bg - 1u;
        }
        if (FILL) {
            const uint64_t rec = ((g * sh.nt + v) * sh.nset + b) * 64u + (uint64_t)j * sh.sps + q;
            prow[rec] = (uint16_t)((tau % sh.B) * sh.T + (idx - tau * sh.T));
            pcnt[rec] = (uint8_t)cnt;
        }
    }
    if (!FILL) ovc[e] = n_ov;
}

// ---- slots: the outer vectors as the layout sees them ----------------------------------------------------------------------
// A slot owns K record positions per visit — right for a vector with 1-2 nonzeros per panel tile. Real count matrices are not like
// that in the gene-major orientation: a few thousand genes are detected in 10-90 % of the cells (5-40 nonzeros per 48-cell tile)
// and most of the others in well under 1 % — with one slot per vector the dense genes overflow (70 % of the nonzeros on
// tools/pass_bench.py's heavy-tailed model) while the sparse ones are all padding. So a vector gets as many slots as its density
// asks for: with x = (its nonzeros) T / n_inner expected nonzeros per tile, V = round(x / x_target) slots, at least one; its
// nonzeros are dealt to its V slots round-robin in storage order (slot r takes positions s + r, s + r + V, ... of the vector), each
// slot then runs the same first-come-first-served assignment as a vector of its own; V = 0 below x_min: all of the vector goes to
// the overflow part (two positions per visit for a third of a nonzero cost more than its row gathers). The product kernel sees
// n_slots independent "outer vectors"; tile_finish_kernel adds the slots of a vector.
__global__ void tile_mult_kernel(const uint64_t *__restrict__ indptr, uint64_t n_outer, double tiles_inv, double x_target, double x_min,
                                 uint32_t v_max, uint32_t *__restrict__ mult) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o > n_outer) return;
    if (o == n_outer) {
        mult[o] = 0;
        return;
    }
    const double x = (double)(indptr[o + 1] - indptr[o]) * tiles_inv; // expected nonzeros per tile
    uint32_t v = 1;
    if (x_target > 0.0) {
        if (x < x_min)
            v = 0;
        else {
            const double q = floor(x / x_target + 0.5);
            v = q < 1.0 ? 1u : q > (double)v_max ? v_max : (uint32_t)q;
        }
    }
    mult[o] = v;
}
__global__ void tile_slotvec_kernel(const uint32_t *__restrict__ slot_first, uint64_t n_outer, uint32_t *__restrict__ slot_vec) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_outer) return;
    for (uint32_t sidx = slot_first[o]; sidx < slot_first[o + 1]; sidx++) slot_vec[sidx] = (uint32_t)o;
}
// vectors without a slot: their nonzeros of every part are overflow (count, then copy)
template <bool FILL>
__global__ void tile_slotless_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices, const uint32_t *__restrict__ values,
                                     const uint32_t *__restrict__ slot_first, uint64_t n_outer, TileShape sh, unsigned long long *__restrict__ ovc,
                                     const unsigned long long *__restrict__ ov_off, uint32_t *__restrict__ ov_indices, uint32_t *__restrict__ ov_values) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_outer * sh.n_parts) return;
    const uint64_t o = e / sh.n_parts;
    if (slot_first[o + 1] != slot_first[o]) return;
    const uint32_t part = (uint32_t)(e - o * sh.n_parts);
    const uint64_t k0 = (uint64_t)part * sh.tpp * sh.T, k1 = (uint64_t)min(sh.nt, (part + 1u) * sh.tpp) * sh.T;
    const uint64_t s = indptr[o], end = indptr[o + 1];
    auto lower = [&](uint64_t key) {
        uint64_t lo = s, hi = end;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if ((uint64_t)indices[mid] < key)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    };
    const uint64_t a = lower(k0), b = lower(k1);
    if (!FILL) {
        ovc[e] = b - a;
    } else {
        unsigned long long op = ov_off[e];
        for (uint64_t p = a; p < b; p++, op++) {
            ov_indices[op] = indices[p];
            ov_values[op] = values[p];
        }
    }
}

// The same assignment, one LANE per slot and the wave in lock-step over the visits (round 4; the default for K = 2, B = 4, S = 32).
// The per-thread walk above reads its vector 4 bytes at a time from wherever it stands and writes 3-byte records all
// over the layout: every 128-byte line travels from the L2 to a CU a dozen times and the record rows are patched together in
// the L2 (157 GB of HBM traffic and 47-62 ms per launch at 10^9 nonzeros for 6 GB of output). Here
//   * a wave owns two groups (64 slots) and one part; visit v is worked by all lanes together: every lane takes the nonzeros
//     of its slot that lie in tile v (the wave loops while any lane has one: about 5 trips at 1.4 nonzeros per slot and tile)
//     and deals them to the positions of visits v .. v+2 exactly as above (same two queues, same tie rule) — but into a WINDOW of
//     three visits held in registers; when tile v is done, the row of visit v is complete and leaves as whole 64-byte runs
//     (32 lanes x u16 rows, 32 x u8 counts per group and position): every record is written exactly once, used or not, so the
//     layout needs no initialisation pass;
//   * a wave whose slots own at most every 4th nonzero of their vector (V <= 4: the usual case) reads in aligned 16-byte chunks (4 indices, 4 counts)
//     through two chunk registers per lane; the chunk after the current one is requested at the END of a visit and first looked
//     at in the next one, so the wave does not stand on a load it has just issued; only a slot with more than 4-8 nonzeros in ONE
//     tile waits inside a visit. A wave with split vectors walks element by element with the next element requested one step ahead.
// FILL = false counts the nonzeros no visit has room for (per vector and part), FILL = true writes records and overflow part
// (the overflow nonzeros of a split vector land in its segment in arrival order of its lanes: tile_layout_build sorts those segments).
struct AssignState {
    uint32_t vu, cu, vg, cg;
    uint32_t w00, w01, w10, w11, w20, w21; // window: visit v + d, position j -> code | count << 16 (0: free)
};
template <bool FILL, uint32_t T, uint32_t KU>
__device__ __forceinline__ bool tile_assign_one(AssignState &a, uint32_t v, uint32_t t1, uint32_t idx, uint32_t cnt) {
    constexpr uint32_t KG = 2u - KU, B = 4u;
    const uint32_t tau = v;
    const uint32_t last = min(tau + B - 2u, t1 - 1u);
    uint32_t au = a.vu, bu = a.cu, ag = a.vg, bg = a.cg;
    if (tau > au) {
        au = tau;
        bu = KU;
    }
    if (bu == 0) {
        au++;
        bu = KU;
    }
    if (tau > ag) {
        ag = tau;
        bg = KG;
    }
    if (bg == 0) {
        ag++;
        bg = KG;
    }
    const bool can_u = KU > 0 && cnt == 1u && au <= last;
    const bool can_g = KG > 0 && cnt <= 255u && ag <= last;
    if (!can_u && !can_g) return false; // overflow
    const bool use_u = can_u && (!can_g || au <= ag);
    uint32_t vis, j;
    if (use_u) {
        vis = au;
        j = KU - bu;
        a.vu = au;
        a.cu = bu - 1u;
    } else {
        vis = ag;
        j = KU + (KG - bg);
        a.vg = ag;
        a.cg = bg - 1u;
    }
    if (FILL) {
        const uint32_t rec = ((tau & 3u) * T + (idx - tau * T)) | (cnt << 16);
        const uint32_t d = vis - v;
        if (d == 0u) {
            if (j == 0u) a.w00 = rec; else a.w01 = rec;
        } else if (d == 1u) {
            if (j == 0u) a.w10 = rec; else a.w11 = rec;
        } else {
            if (j == 0u) a.w20 = rec; else a.w21 = rec;
        }
    }
    return true;
}
template <bool FILL, uint32_t T, uint32_t KU>
__global__ __launch_bounds__(64) void tile_assign_wave_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                              const uint32_t *__restrict__ values, const uint32_t *__restrict__ slot_vec,
                                                              const uint32_t *__restrict__ slot_first, uint64_t n_slots, uint64_t n_groups, TileShape sh,
                                                              unsigned long long *__restrict__ ovc, const unsigned long long *__restrict__ ov_off,
                                                              unsigned long long *__restrict__ ov_cursor, uint16_t *__restrict__ prow,
                                                              uint8_t *__restrict__ pcnt, uint32_t *__restrict__ ov_indices, uint32_t *__restrict__ ov_values) {
    constexpr uint32_t KG = 2u - KU, B = 4u;
    const uint32_t lane = threadIdx.x;
    const uint64_t item = blockIdx.x;
    const uint64_t gp = item / sh.n_parts;
    const uint32_t part = (uint32_t)(item - gp * sh.n_parts);
    const uint64_t slot = gp * 64u + lane;
    const bool valid = slot < n_slots;
    const uint64_t o = valid ? slot_vec[slot] : 0ull;
    const uint32_t sf = valid ? slot_first[o] : 0u;
    const uint32_t V = valid ? slot_first[o + 1] - sf : 1u, r = valid ? (uint32_t)slot - sf : 0u;
    const uint64_t g = gp * 2u + (lane >> 5);
    const uint32_t q = lane & 31u;
    const uint32_t t0 = part * sh.tpp, t1 = min(sh.nt, t0 + sh.tpp);
    const uint64_t s = valid ? indptr[o] : 0ull, end = valid ? indptr[o + 1] : 0ull;
    uint64_t p = s;
    if (t0 > 0) { // first nonzero of the part
        uint64_t lo = s, hi = end;
        const uint64_t key = (uint64_t)t0 * T;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if ((uint64_t)indices[mid] < key)
                lo = mid + 1;
            else
                hi = mid;
        }
        p = lo;
    }
    const uint64_t part_lo = p; // the part's first nonzero in the source arrays
    if (V > 1u) p += (r + V - (uint32_t)((p - s) % V)) % V; // this slot's first position of the part: p = s + r (mod V)
    const uint64_t e = o * sh.n_parts + part;
    AssignState a{t0, KU, t0, KG, 0, 0, 0, 0, 0, 0};
    unsigned long long n_ov = 0;
    // FILL with ov_off == nullptr: ONE pass (no counting pass in front): the overflow nonzeros of a (vector, part) segment go into arrays
    // indexed like the SOURCE, from the segment's first nonzero on (they are at most as many as the segment has), the count of the
    // segment into ov_cursor (zeroed before; the lanes of a split vector add theirs up as they go); tile_ov_compact_kernel then
    // moves them to their final places.
    const bool one_pass = FILL && ov_off == nullptr;
    unsigned long long op = (FILL && !one_pass && valid && V == 1u) ? ov_off[e] : 0ull;
    auto overflow = [&](uint32_t idx, uint32_t cnt) {
        if (FILL) {
            unsigned long long at;
            if (one_pass)
                at = part_lo + (V == 1u ? n_ov++ : atomicAdd(&ov_cursor[e], 1ull));
            else
                at = V == 1u ? op++ : atomicAdd(&ov_cursor[e], 1ull);
            ov_indices[at] = idx;
            ov_values[at] = cnt;
        } else {
            n_ov++;
        }
    };
    auto emit = [&](uint32_t v) { // the row of visit v: whole 64-byte runs per group and position
        if (FILL && g < n_groups) {
            const uint64_t row = (g * sh.nt + v) * 64u;
            const uint32_t e0 = a.w00 ? a.w00 : (KU > 0 ? B * T : (v & 3u) * T); // an unused unit position reads the row of zeros behind the ring,
            const uint32_t e1 = a.w01 ? a.w01 : (v & 3u) * T;                    // an unused general one row 0 of the visit's own tile (weight 0)
            prow[row + q] = (uint16_t)e0;
            prow[row + 32u + q] = (uint16_t)e1;
            pcnt[row + q] = (uint8_t)(e0 >> 16);
            pcnt[row + 32u + q] = (uint8_t)(e1 >> 16);
        }
        a.w00 = a.w10;
        a.w01 = a.w11;
        a.w10 = a.w20;
        a.w11 = a.w21;
        a.w20 = 0;
        a.w21 = 0;
    };
    if (!__builtin_amdgcn_ballot_w64(valid && V > 4u)) {
        // ---- every slot of the wave is a whole vector or one of at most 4 of its vector (the next element of a slot is then in the
        // current chunk or the one behind it): chunked reads ----
        const uint4 *I4 = reinterpret_cast<const uint4 *>(indices), *V4 = reinterpret_cast<const uint4 *>(values);
        uint64_t ca = p >> 2; // chunk held in (ai, av); (bi, bv) holds chunk ca + 1. Reads run at most two chunks past a vector's end: the arrays are padded (DevBuf)
        uint4 ai = I4[ca], av = V4[ca], bi = I4[ca + 1], bv = V4[ca + 1];
        // element k of the chunk the position is in: selects only (an indexed uint4 would live in scratch memory)
        auto pick = [](bool second, uint32_t k, uint4 x, uint4 y) {
            const uint32_t x0 = second ? y.x : x.x, x1 = second ? y.y : x.y, x2 = second ? y.z : x.z, x3 = second ? y.w : x.w;
            const uint32_t lo = (k & 1u) ? x1 : x0, hi = (k & 1u) ? x3 : x2;
            return (k & 2u) ? hi : lo;
        };
        for (uint32_t v = t0; v < t1; v++) {
            const uint64_t lim = (uint64_t)(v + 1u) * T; // nonzeros below it belong to tile v (everything below v T has been worked)
            for (;;) {
                for (;;) {
                    const uint64_t ck = p >> 2;
                    const bool inb = ck != ca;
                    const bool avail = ck <= ca + 1u;
                    const uint32_t k = (uint32_t)p & 3u;
                    const uint32_t idx = pick(inb, k, ai, bi), cnt = pick(inb, k, av, bv);
                    const bool has = p < end && avail && (uint64_t)idx < lim;
                    if (!__builtin_amdgcn_ballot_w64(has)) break;
                    if (has) {
                        if (!tile_assign_one<FILL, T, KU>(a, v, t1, idx, cnt)) overflow(idx, cnt);
                        p += V;
                    }
                }
                // a lane that ran out of chunks inside the tile (more than 4-8 nonzeros of one vector in one tile): fetch and go on
                const uint64_t ck = p >> 2;
                const bool starved = p < end && ck > ca + 1u;
                if (!__builtin_amdgcn_ballot_w64(starved)) break;
                if (starved) {
                    ca = ck;
                    ai = I4[ca];
                    av = V4[ca];
                    bi = I4[ca + 1];
                    bv = V4[ca + 1];
                }
            }
            emit(v);
            // advance the chunk registers here, a visit ahead of their use
            const uint64_t ck = p >> 2;
            if (ck == ca + 1u) {
                ai = bi;
                av = bv;
                ca = ck;
                bi = I4[ca + 1];
                bv = V4[ca + 1];
            } else if (ck > ca + 1u) {
                ca = ck;
                ai = I4[ca];
                av = V4[ca];
                bi = I4[ca + 1];
                bv = V4[ca + 1];
            }
        }
    } else {
        // ---- split vectors among the slots: every lane steps through its own positions (stride V), the next element requested one
        // step ahead; the V lanes of a vector read neighbouring elements, so their loads share cache lines ----
        const uint64_t last_el = end ? end - 1 : 0; // reads are clamped into the vector (lanes past their end are masked by p < end)
        uint32_t ci = indices[p < end ? p : last_el], cc = values[p < end ? p : last_el];
        uint64_t pn = p + V;
        uint32_t ni = indices[pn < end ? pn : last_el], nc = values[pn < end ? pn : last_el];
        for (uint32_t v = t0; v < t1; v++) {
            const uint64_t lim = (uint64_t)(v + 1u) * T;
            for (;;) {
                const bool has = p < end && (uint64_t)ci < lim;
                if (!__builtin_amdgcn_ballot_w64(has)) break;
                if (has) {
                    if (!tile_assign_one<FILL, T, KU>(a, v, t1, ci, cc)) overflow(ci, cc);
                    p = pn;
                    ci = ni;
                    cc = nc;
                    pn = p + V;
                    ni = indices[pn < end ? pn : last_el];
                    nc = values[pn < end ? pn : last_el];
                }
            }
            emit(v);
        }
    }
    if (!FILL && valid) {
        if (V == 1u)
            ovc[e] = n_ov;
        else if (n_ov)
            atomicAdd(&ovc[e], n_ov);
    }
    if (one_pass && valid && V == 1u) ov_cursor[e] = n_ov;
}

// One-pass build: the overflow nonzeros of segment e = (vector, part) stand in `tmp_*` from the segment's first source position on
// (a vector without slots: ALL its nonzeros of the part are overflow and are taken from the source arrays themselves); their final
// place starts at ov_off[e]. A wave takes 64 segments: every lane looks up one of them, then the wave copies them one after the other.
__global__ __launch_bounds__(256) void tile_ov_compact_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ indices,
                                                              const uint32_t *__restrict__ values, const uint32_t *__restrict__ slot_first,
                                                              uint64_t n_outer, TileShape sh, const unsigned long long *__restrict__ ov_off,
                                                              const uint32_t *__restrict__ tmp_indices, const uint32_t *__restrict__ tmp_values,
                                                              uint32_t *__restrict__ ov_indices, uint32_t *__restrict__ ov_values) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t n_seg = n_outer * sh.n_parts;
    const uint64_t e0 = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u;
    if (e0 >= n_seg) return;
    const uint64_t e = e0 + lane;
    uint64_t src = 0, dst = 0;
    uint32_t cnt = 0, from_source = 0;
    if (e < n_seg) {
        dst = ov_off[e];
        cnt = (uint32_t)(ov_off[e + 1] - dst);
        if (cnt) {
            const uint64_t o = e / sh.n_parts;
            const uint32_t part = (uint32_t)(e - o * sh.n_parts);
            uint64_t lo = indptr[o], hi = indptr[o + 1];
            const uint64_t key = (uint64_t)part * sh.tpp * sh.T;
            if (part > 0)
                while (lo < hi) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if ((uint64_t)indices[mid] < key)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
            src = lo;
            from_source = slot_first[o + 1] == slot_first[o] ? 1u : 0u;
        }
    }
    uint64_t busy = __builtin_amdgcn_ballot_w64(cnt != 0u);
    while (busy) {
        const uint32_t j = (uint32_t)__builtin_ctzll(busy);
        busy &= busy - 1;
        const uint32_t n = rdlane(cnt, j);
        const uint64_t sj = ((uint64_t)rdlane((uint32_t)(src >> 32), j) << 32) | rdlane((uint32_t)src, j);
        const uint64_t dj = ((uint64_t)rdlane((uint32_t)(dst >> 32), j) << 32) | rdlane((uint32_t)dst, j);
        const bool fs = rdlane(from_source, j) != 0u;
        const uint32_t *si = fs ? indices : tmp_indices, *sv = fs ? values : tmp_values;
        for (uint32_t k = lane; k < n; k += 64u) {
            ov_indices[dj + k] = si[sj + k];
            ov_values[dj + k] = sv[sj + k];
        }
    }
}

// An unused general position reads a row that is certainly in the ring at its visit — row 0 of the visit's own tile — with
// weight 0; an unused unit position (no weight) reads the row of zeros that the kernel keeps behind the ring (row B T).
__global__ void tile_init_rows_kernel(uint16_t *__restrict__ prow, uint64_t n_rec, TileShape sh) {
    // 4 records per step, grid-stride (a dispatch holds fewer than 2^32 work-items)
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e * 4u < n_rec; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = (uint32_t)((e * 4u / (sh.nset * 64u)) % sh.nt);
        const uint32_t own = (v % sh.B) * sh.T, zero = sh.B * sh.T;
        uint32_t word[2] = {0u, 0u};
        for (uint32_t i = 0; i < 4u; i++) {
            const uint32_t lane = (uint32_t)((e * 4u + i) & 63u);
            word[i >> 1] |= ((lane / sh.sps) < sh.KU ? zero : own) << (16u * (i & 1u));
        }
        reinterpret_cast<uint2 *>(prow)[e] = make_uint2(word[0], word[1]);
    }
}

// pw[rec] = the map chain at the record's (count, outer, inner), 0 for an unused position. One work-item per position that
// carries a weight: all 64 of a record row, or — unit mode (uo / vi given) — the lanes behind the `skip` unit positions
// (never read: no weight). Every such position is written, used or not: whole cache lines leave the CU (with the unused
// ones skipped the partial-line stores cost twice the HBM traffic of the full ones). nset * 64 is a power of two and
// (group, visit) pairs fit 32 bits (checked by the builder).
// Unit mode: the stored weight is w / (uo[outer] * vi[inner]) — the kernel works on panel rows scaled by vi and scales a
// vector's sum by uo at the end.
__global__ __launch_bounds__(256) void tile_weights_kernel(const uint16_t *__restrict__ prow, const uint8_t *__restrict__ pcnt,
                                                           double *__restrict__ pw, uint64_t n_rec, const uint32_t *__restrict__ slot_vec, uint64_t n_slots, TileShape sh, DevMap map,
                                                           const double *__restrict__ uo, const double *__restrict__ vi, uint32_t skip,
                                                           const double *__restrict__ tab, int tab_outer, uint32_t vmajor_groups) {
    const uint32_t per_row = 64u - skip; // 64 or 32 (K = 2, one unit position)
    // The index arithmetic below would be five integer divisions per position (~25 VALU instructions each — as much as the
    // logarithm and the division of the weight together): quotients through a rounded-down product with the reciprocal
    // instead. floor((x + 0.5) * (1 / d)) is exact as long as the rounding error stays below the 0.5 / d margin: single
    // precision for x < 2^16, double for the 32-bit (group, visit) index (x / d < 2^22, d < 2^22: error 2^-30 against 2^-23).
    const float inv_T = 1.0f / (float)sh.T, inv_sps = 1.0f / (float)sh.sps;
    const double inv_nt = 1.0 / (double)sh.nt;
    const uint32_t nset_shift = 31u - (uint32_t)__clz((int)sh.nset);
    // A thread works the same lane of WU consecutive record rows at a time: their counts and row codes are loaded first, then
    // the table lookups, then the stores — one position per trip made the pass a chain of three dependent loads per store
    // (1.5 TB/s). Grid-stride: a launch holds fewer than 2^32 work-items, a layout more records.
    constexpr int WU = 4;
    const uint64_t n_rows = n_rec >> 6, n_quads = (n_rows + WU - 1) / WU;
    // Visit-major order (vmajor_groups = the number of groups; grid.y = quad of visits, grid.x over groups x lanes; one set per
    // visit): the table is indexed by the INNER position there, and with the records in storage order every wave would be
    // at a different place of it (64 MB at 10^6 cells: every lookup an L2 miss, 8.2 ms); in this order the waves in flight
    // work the same few hundred inner positions.
    const uint64_t n_lin = vmajor_groups ? (uint64_t)vmajor_groups * per_row : n_quads * per_row;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lin; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t gq = per_row == 64u ? (i >> 6) : (i >> 5); // linear: quad of rows; visit-major: group
        const uint32_t lane = skip + (uint32_t)(i & (per_row - 1u));
        // first row of the quad and the rows it may use: linear -> up to the end of the layout; visit-major -> up to the end of the group
        const uint64_t row0 = vmajor_groups ? gq * sh.nt + (uint64_t)blockIdx.y * WU : gq * WU;
        const uint64_t row_end = vmajor_groups ? (gq + 1) * sh.nt : n_rows;
        const uint32_t qs = (uint32_t)(((float)lane + 0.5f) * inv_sps);
        const uint32_t slot = lane - qs * sh.sps;
        uint32_t cnt[WU], code[WU];
        bool ok[WU];
#pragma unroll
        for (int u = 0; u < WU; u++) {
            const uint64_t sv64 = row0 + u;
            ok[u] = sv64 < row_end;
            const uint64_t e = ((ok[u] ? sv64 : n_rows - 1) << 6) + lane;
            cnt[u] = ok[u] ? pcnt[e] : 0u; // a row past the end has no positions
            code[u] = prow[e];
        }
        double w[WU];
        uint32_t oo[WU], in[WU];
#pragma unroll
        for (int u = 0; u < WU; u++) {
            const uint32_t sv = (uint32_t)(row0 + u);
            const uint32_t b = sv & (sh.nset - 1u);
            const uint32_t gv = sv >> nset_shift;
            const uint32_t g = (uint32_t)(((double)gv + 0.5) * inv_nt), v = gv - g * sh.nt;
            const uint32_t gs = g * sh.S + b * sh.sps + slot; // slot of the layout -> the outer vector it belongs to (padding slots of the last group: unused positions only)
            oo[u] = gs < n_slots ? slot_vec[gs] : 0u;
            const uint32_t bufi = (uint32_t)(((float)code[u] + 0.5f) * inv_T), r = code[u] - bufi * sh.T;
            const uint32_t vmod = sh.B == 4u ? (v & 3u) : v % sh.B; // the default ring has 4 buffers
            const uint32_t d = vmod >= bufi ? vmod - bufi : vmod + sh.B - bufi; // visits the nonzero waited
            in[u] = (v - d) * sh.T + r;
            w[u] = 0.0;
            if (tab && cnt[u] != 0u && cnt[u] <= TL_TABC) // unit mode: the quotient depends on the count and on one side's position only (tile_ratio_table_kernel)
                w[u] = tab[(size_t)(tab_outer ? oo[u] : in[u]) * TL_TABC + (cnt[u] - 1u)];
        }
#pragma unroll
        for (int u = 0; u < WU; u++) {
            if (cnt[u] != 0u && !(tab && cnt[u] <= TL_TABC)) {
                double x = eval_map(map, cnt[u], oo[u], in[u]);
                if (uo) {
                    const double d2 = uo[oo[u]] * vi[in[u]];
                    x = (d2 != 0.0 && isfinite(d2)) ? x / d2 : 0.0; // a zero unit weight means a zero weight for every count (log1p, square, scale)
                }
                w[u] = x;
            }
            // unit mode keeps weights for the general half of a record row only (32 per row: the unit positions have none)
            if (ok[u]) pw[skip ? ((row0 + u) << 5) + (lane - skip) : ((row0 + u) << 6) + lane] = w[u];
        }
    }
}

// The same refresh for the usual case — unit mode on the default shape (K = 2, S = 32, B = 4: weights of the 32 general positions
// of a record row only, quotient table present) — with wide accesses: a thread owns FOUR neighbouring general positions of a row
// (one 4-byte load of their counts, one 8-byte load of their row codes, one 16-byte load of their slots' vectors, 32 bytes of
// weights stored) and WU rows of them; the one-position-per-thread form above moved 1-2 bytes per lane and load. The order of the
// rows is the same: linear for a table indexed by the outer vector, visit-major (grid.y = quad of visits) for one indexed by the
// inner position.
template <uint32_t T>
__global__ __launch_bounds__(256) void tile_weights_unit_kernel(const uint16_t *__restrict__ prow, const uint8_t *__restrict__ pcnt,
                                                                double *__restrict__ pw, uint64_t n_rows, const uint32_t *__restrict__ slot_vec,
                                                                uint64_t n_slots, uint32_t nt, DevMap map, const double *__restrict__ uo,
                                                                const double *__restrict__ vi, const double *__restrict__ tab, int tab_outer,
                                                                uint32_t vmajor_groups) {
    constexpr int WU = 4;
    const double inv_nt = 1.0 / (double)nt;
    const float inv_T = 1.0f / (float)T;
    const uint64_t n_quads = (n_rows + WU - 1) / WU;
    const uint64_t n_lin = (vmajor_groups ? (uint64_t)vmajor_groups : n_quads) * 8u; // 8 threads per row: 4 general lanes each
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lin; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t gq = i >> 3;
        const uint32_t j4 = (uint32_t)(i & 7u) * 4u; // first of this thread's four general positions = slots j4 .. j4 + 3 of the group
        const uint64_t row0 = vmajor_groups ? gq * nt + (uint64_t)blockIdx.y * WU : gq * WU;
        const uint64_t row_end = vmajor_groups ? (gq + 1) * nt : n_rows;
        uint32_t cnt4[WU];
        uint2 code4[WU];
        bool ok[WU];
#pragma unroll
        for (int u = 0; u < WU; u++) {
            ok[u] = row0 + u < row_end;
            const uint64_t e = ((ok[u] ? row0 + u : n_rows - 1) << 6) + 32u + j4;
            cnt4[u] = ok[u] ? *reinterpret_cast<const uint32_t *>(pcnt + e) : 0u;
            code4[u] = *reinterpret_cast<const uint2 *>(prow + e);
        }
#pragma unroll
        for (int u = 0; u < WU; u++) {
            if (!ok[u]) continue;
            const uint32_t gv = (uint32_t)(row0 + u);
            const uint32_t g = (uint32_t)(((double)gv + 0.5) * inv_nt), v = gv - g * nt;
            const uint64_t gs = (uint64_t)g * 32u + j4;
            uint32_t oo[4] = {0u, 0u, 0u, 0u};
            if (gs + 3u < n_slots) {
                const uint4 sv = *reinterpret_cast<const uint4 *>(slot_vec + gs);
                oo[0] = sv.x;
                oo[1] = sv.y;
                oo[2] = sv.z;
                oo[3] = sv.w;
            } else {
                for (uint32_t k = 0; k < 4u; k++) oo[k] = gs + k < n_slots ? slot_vec[gs + k] : 0u; // padding slots of the last group: unused positions only
            }
            const uint32_t codes[4] = {code4[u].x & 0xFFFFu, code4[u].x >> 16, code4[u].y & 0xFFFFu, code4[u].y >> 16};
            double w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t c = (cnt4[u] >> (8 * k)) & 0xFFu;
                const uint32_t bufi = (uint32_t)(((float)codes[k] + 0.5f) * inv_T), r = codes[k] - bufi * T;
                const uint32_t d = ((v & 3u) - bufi) & 3u; // visits the nonzero waited (ring of 4 buffers)
                const uint32_t in = (v - d) * T + r;
                double x = 0.0;
                if (c != 0u && c <= TL_TABC) {
                    x = tab[(size_t)(tab_outer ? oo[k] : in) * TL_TABC + (c - 1u)];
                } else if (c != 0u) {
                    x = eval_map(map, c, oo[k], in);
                    const double d2 = uo[oo[k]] * vi[in];
                    x = (d2 != 0.0 && isfinite(d2)) ? x / d2 : 0.0;
                }
                w[k] = x;
            }
            double *dst = pw + ((row0 + u) << 5) + j4;
            *reinterpret_cast<d2 *>(dst) = (d2){w[0], w[1]};
            *reinterpret_cast<d2 *>(dst + 2) = (d2){w[2], w[3]};
        }
    }
}

// The factor of the weight of a count-1 nonzero that depends on one side only (outer or inner position): the chain run on
// x = 1 with the links of the other side left out; the side that owns the links in front of the nonlinear links also owns
// those (nl_outer). uo[o] * vi[i] == f(1, o, i) up to rounding for the chains tile_map_separable() accepts.
__global__ void tile_unit_factor_kernel(DevMap map, int side_outer, int nl_outer, uint64_t n, double *__restrict__ out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    double x = 1.0;
    const bool owns_nl = (side_outer != 0) == (nl_outer != 0);
    for (int i = 0; i < map.n; i++) {
        const DevOp &op = map.ops[i];
        switch (op.kind) {
        case OP_SCALE_AXIS:
            if ((op.a_outer != 0) == (side_outer != 0)) x = op.a[idx] * x;
            break;
        case OP_LN_1P:
            if (owns_nl) x = map_ln(x + 1.0);
            break;
        case OP_LOG2_1P:
            if (owns_nl) x = map_log2(x + 1.0);
            break;
        case OP_LOG10_1P:
            if (owns_nl) x = map_log10(x + 1.0);
            break;
        case OP_SQUARE:
            if (owns_nl) x = x * x;
            break;
        default:
            break;
        }
    }
    out[idx] = x;
}

// Unit mode: the stored weight of a general position, f(c, o, i) / (uo[o] vi[i]), is N(c a) / N(a) with a = the scales in
// front of the nonlinear links — all on ONE side for a separable chain — and N those links; every scale behind them cancels.
// It therefore depends on the count and on that side's position only, and 60 % of the counts above 1 are 2, 95 % at most 4:
// a table of the quotient for counts 1 .. 8 per position of that side (one 64-byte line; 8 logarithms per cell instead of one per
// nonzero per orientation: 8e6 against 5.6e8 at 10^6 cells; counts above 8 — 0.07 % — are evaluated directly) turns the weight refresh into a streaming pass with a cached lookup.
// G(c) = the chain on x = c with the other side's links left out; tab[idx][c - 1] = G(c) / G(1).
__global__ void tile_ratio_table_kernel(DevMap map, int nl_outer, uint64_t n, double *__restrict__ tab) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * TL_TABC) return;
    const uint64_t idx = e / TL_TABC;
    const uint32_t c = (uint32_t)(e - idx * TL_TABC) + 1u;
    double g[2];
    for (int which = 0; which < 2; which++) {
        double x = which ? (double)c : 1.0;
        for (int i = 0; i < map.n; i++) {
            const DevOp &op = map.ops[i];
            switch (op.kind) {
            case OP_SCALE_AXIS:
                if ((op.a_outer != 0) == (nl_outer != 0)) x = op.a[idx] * x;
                break;
            case OP_LN_1P:
                x = map_ln(x + 1.0);
                break;
            case OP_LOG2_1P:
                x = map_log2(x + 1.0);
                break;
            case OP_LOG10_1P:
                x = map_log10(x + 1.0);
                break;
            case OP_SQUARE:
                x = x * x;
                break;
            default:
                break;
            }
        }
        g[which] = x;
    }
    tab[e] = (g[0] != 0.0 && isfinite(g[0])) ? g[1] / g[0] : 0.0;
}

// Xc[r, 0:l] = vi[r] * X[r, 0:l] in compact rows of ldc columns (the panel the unit-mode kernel stages)
__global__ void tile_scale_panel_kernel(const double *__restrict__ X, uint32_t ldx, uint64_t rows, uint32_t l, uint32_t ldc,
                                        const double *__restrict__ vi, double *__restrict__ Xc) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * ldc) return;
    const uint64_t r = e / ldc;
    const uint32_t c = (uint32_t)(e - r * ldc);
    Xc[e] = c < l ? vi[r] * X[r * ldx + c] : 0.0;
}

__global__ void tile_gather_slots_kernel(const double *__restrict__ per_vec, const uint32_t *__restrict__ slot_vec, uint64_t n_slots, double *__restrict__ per_slot) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slots) per_slot[i] = per_vec[slot_vec[i]];
}

// segment bounds of the overflow sort: a vector with one slot (or none) wrote its overflow in index order already -> length 0
struct SplitSegment {
    const uint64_t *indptr;
    const uint32_t *slot_first;
    int end;
    __host__ __device__ unsigned long long operator()(unsigned o) const {
        const bool split = slot_first[o + 1] - slot_first[o] > 1u;
        return indptr[o + ((end && split) ? 1u : 0u)];
    }
};
__global__ void tile_split_copyback_kernel(const uint64_t *__restrict__ indptr, const uint32_t *__restrict__ slot_first, uint64_t n_outer,
                                           const uint32_t *__restrict__ si, const uint32_t *__restrict__ sv, uint32_t *__restrict__ di,
                                           uint32_t *__restrict__ dv) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n_outer || slot_first[o + 1] - slot_first[o] <= 1u) return;
    for (uint64_t p = indptr[o]; p < indptr[o + 1]; p++) {
        di[p] = si[p];
        dv[p] = sv[p];
    }
}

__global__ void tile_ovptr_kernel(const unsigned long long *__restrict__ ov_off, uint64_t n_outer, uint32_t n_parts,
                                  uint64_t *__restrict__ indptr) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o <= n_outer) indptr[o] = ov_off[o * n_parts];
}

} // namespace

// the dense record layout serves the default shape (tiles_dense.inc)
static inline bool tile_dense_wanted(const Storage &st) {
    return st.tile_dense != 0 && st.tile_builder != 0 && st.tile_k == 2u && st.tile_b == 4u && st.tile_s == 32u && st.tile_t == 48u;
}
// ... in its flow form (tiles_flow.inc): the map must come from a table inside the kernel, the build is the one-walk build
static inline bool tile_flow_wanted(const Storage &st, const SparseCopy &cp) {
    return tile_dense_wanted(st) && st.tile_flow != 0 && st.tile_wtab != 0 && st.tile_fold != 0 && st.tile_one_walk != 0 && !cp.tile_flow_refused;
}
// the tile layout of one orientation under one map
struct TileLayout {
    TileShape sh{};
    uint64_t n_groups = 0;
    uint64_t n_slots = 0;          // "outer vectors" of the product kernel: a vector owns slot_first[o] .. slot_first[o + 1] (none: all of it is overflow)
    uint32_t max_mult = 1;         // most slots one vector owns
    DevBuf<uint32_t> slot_first;   // [n_outer + 1]
    DevBuf<uint32_t> slot_vec;     // [n_slots]: the vector a slot belongs to
    double split_x = 0.0, split_min = 0.0; // the rule the slots were made with (0: one slot per vector)
    DevBuf<uint16_t> prow; // [group][visit][set][64]: ring row of the position (16 bits: the kernel multiplies a half of a
                           // scalar register by the row pitch in one v_mad_u32_u16 — a byte would cost a scalar extract per position)
    DevBuf<uint8_t> pcnt;  // same index: the raw count (0 = unused position; counts above 255 live in the overflow part)
    DevBuf<double> pw;     // same index: the weight under the map `sig_*`
    SparseCopy ov;         // the overflow part: indptr / indices / values (counts) / fvals (weights)
    // unit mode (sh.KU > 0 and a separable map): per-outer / per-inner factors of the weight of a count-1 nonzero
    bool unit_mode = false;
    DevBuf<double> uo, vi;
    DevBuf<double> uo_slot;   // uo by slot (what the product kernel scales a slot's sums by)
    DevBuf<double> ratio_tab; // unit mode: quotient of the weight of counts 1 .. TL_TABC per position of the side that owns the nonlinear links
    // dense layout (round 5, tiles_dense.inc): record streams [item][wave][chunk][16] instead of fixed positions per (slot, visit)
    bool dense = false;
    bool flow = false;             // dense, flow form (tiles_flow.inc): sh.T = 32, sh.B = 6, per-wave streams (fwaves), rtab = tick counts per round
    DevBuf<char> fwaves;           // FlowWave per (item, wave)
    uint32_t dn_wgg = 0, dn_items = 0;
    uint64_t dn_chunks = 0;        // chunks of 16 positions in drec / pw (cmeta: one entry each)
    uint64_t dn_served = 0;        // nonzeros that own a record
    DevBuf<uint32_t> drec;         // 4 x slot-in-group | raw count << 8 | ring row << 16
    DevBuf<uint64_t> cmeta;        // group << 32 | visit of a chunk (weight refresh)
    DevBuf<char> ditems;           // DenseItem per workgroup item
    DevBuf<uint32_t> rtab;         // rounds: chunks | first-of-visit << 8
    int slot_mode = 0;             // dense: the "tile_sort_slots" form slot_order was built in (1 consecutive ranks per group, 2 dealt)
    DevBuf<uint32_t> slot_order;   // dense: the slot at place p of the layout (group p / 32, accumulator p % 32), slots sorted by load; empty: place = slot
    DevBuf<uint32_t> slot_pos;     // ... and the place of slot s
    DevBuf<double> w_place, w_inner; // dense, separable map: the weight's factor by place ([8] per place when the outer side owns the nonlinear links) / [8] per inner position otherwise
    bool fold_inner = false, fold_outer = false; // dense, separable map: the side without the nonlinear links keeps its factor out of the weights (tiles_dense.inc, dense_weights_kernel)
    bool separable = false;        // dense: the map's count-1 value is uo[outer] vi[inner] (tables in uo / vi / ratio_tab)
    int wsrc = 0;                  // dense: where the product kernel takes a position's weight from - 0 the stream pw, 1 wtab by (place, count), 2 wtab by (count, inner position): the map evaluated inside the kernel (round 6)
    DevBuf<double> wtab;           // ... the table: 16 doubles per place of every workgroup item / 192 zeros + 16 planes of wtab_stride inner positions
    uint64_t wtab_stride = 0;
    // identity of the map the weights were evaluated under (MapOp ids are never reused); -1: none yet
    int sig_n = -1;
    uint32_t sig_id[MAX_OPS] = {};
    int sig_outer[MAX_OPS] = {};
    double bytes() const {
        return (double)prow.n * 2.0 + (double)pcnt.n + (double)pw.n * 8.0 + (double)ov.nnz * 16.0 + (double)ratio_tab.n * 8.0 + (double)drec.n * 4.0 + (double)cmeta.n * 8.0 +
               (double)rtab.n * 4.0 + (double)wtab.n * 8.0 + (double)(w_place.n + w_inner.n) * 8.0 + (double)fwaves.n;
    }
    bool structure_matches(const Storage &st, const SparseCopy &cp) const {
        if (dense != tile_dense_wanted(st)) return false;
        if (flow != tile_flow_wanted(st, cp)) return false;
        if (flow) { // (its own tile shape, whatever tile_t / tile_b say)
            const double want_x = st.tile_split ? st.tile_split_x : 0.0, want_min = st.tile_split ? st.tile_split_min : 0.0;
            return (slot_order.n != 0) == (st.tile_sort_slots != 0 && n_slots > 1) && (slot_order.n == 0 || slot_mode == st.tile_sort_slots) && split_x == want_x && split_min == want_min;
        }
        if (dense && (slot_order.n != 0) != (st.tile_sort_slots != 0 && n_slots > 1)) return false;
        if (dense && slot_order.n != 0 && slot_mode != st.tile_sort_slots) return false;
        const bool splittable = st.tile_builder != 0 && sh.K == 2u && sh.B == 4u && sh.S == 32u && sh.T == 48u; // the wave-level builder's shape
        const double want_x = splittable && st.tile_split ? st.tile_split_x : 0.0, want_min = splittable && st.tile_split ? st.tile_split_min : 0.0;
        return sh.K == st.tile_k && sh.S == st.tile_s && sh.T == st.tile_t && sh.B == st.tile_b && sh.KU == (st.tile_k == 2u && st.tile_ku ? 1u : 0u) &&
               split_x == want_x && split_min == want_min;
    }
    bool weights_match(const DevMap &map) const {
        if (sig_n != map.n) return false;
        for (int i = 0; i < map.n; i++)
            if (sig_id[i] != map.ops[i].id || sig_outer[i] != map.ops[i].a_outer) return false;
        return true;
    }
};

void tile_layout_free(TileLayout *t) { delete t; }
// an option that changes the FORM of the weights (tile_fold, tile_wtab) was set: the next product evaluates them again
void tile_layout_forget_weights(TileLayout *t) {
    if (t) t->sig_n = -1;
}
// record positions the tile kernel works per pass, nonzeros among them, nonzeros left to the overflow gather (scanrs_mat_get_counter)
void tile_layout_stats(const TileLayout *t, uint64_t out[3]) {
    out[0] = out[1] = out[2] = 0;
    if (!t) return;
    out[2] = t->ov.nnz;
    if (t->dense) {
        out[0] = t->dn_chunks * 16u;
        out[1] = t->dn_served;
    } else {
        out[0] = t->prow.n;
    }
}

uint32_t ensure_bounds_public(Storage &st, SparseCopy &cp, hipStream_t s); // kernels.hip
void materialize_map_values(Storage &st, SparseCopy &cp, const DevMap &map, double *fout); // kernels.hip

// unit positions exist for K = 2 only (one unit + one general position per visit)
bool tile_shape_ok(uint32_t K, uint32_t S, uint32_t T, uint32_t B) {
    const bool ks = (K == 2 && (S == 32 || S == 28)) || (K == 3 && S == 32) || (K == 4 && (S == 32 || S == 28));
    return ks && B >= 2 && T >= 8 && T <= 24u * K && B * T <= 192;
}

static TileLayout *dense_layout_build(Storage &st, const SparseCopy &cp, double max_overflow, hipStream_t s, std::unique_ptr<TileLayout> tl,
                                      const std::function<void(const char *)> &lap); // tiles_dense.inc
static thread_local bool tl_flow_fallback = false; // dense_layout_build gave up on the FLOW form (not on the layout): build the dense form instead
// max_overflow > 0: give up (nullptr) when more than that share of the nonzeros would land in the overflow part — known after the
// counting pass, before anything large is allocated.
TileLayout *tile_layout_build(Storage &st, const SparseCopy &cp, double max_overflow, hipStream_t stream = nullptr) {
    Tick tick("tile layout build");
    if (!tile_shape_ok(st.tile_k, st.tile_s, st.tile_t, st.tile_b))
        fail(SCANRS_ERR_ARGUMENT, "unsupported tile shape K=%u S=%u T=%u B=%u", st.tile_k, st.tile_s, st.tile_t, st.tile_b);
    auto tl = std::make_unique<TileLayout>();
    TileShape &sh = tl->sh;
    sh.K = st.tile_k;
    sh.KU = st.tile_k == 2u && st.tile_ku ? 1u : 0u;
    sh.S = st.tile_s;
    sh.T = st.tile_t;
    sh.B = st.tile_b;
    tl->flow = tile_flow_wanted(st, cp);
    if (tl->flow) { // the flow form of the dense layout: the same 192 ring rows as 6 tiles of 32 (tiles_flow.inc)
        sh.T = 32u;
        sh.B = 6u;
    }
    sh.sps = 64u / sh.K;
    sh.nset = (sh.S + sh.sps - 1) / sh.sps;
    sh.nt = (uint32_t)((cp.n_inner + sh.T - 1) / sh.T);
    hipStream_t s = stream ? stream : st.stream;
    auto lap = [&](const char *what) { // SCANRS_TRACE: where a build spends its time (forces a sync per phase)
        static thread_local std::chrono::steady_clock::time_point t_prev;
        if (!trace_on()) return;
        (void)wait_stream_quiet(s);
        const auto t_now = std::chrono::steady_clock::now();
        if (what) fprintf(stderr, "[scanrs trace]   layout: %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t_now - t_prev).count());
        t_prev = t_now;
    };
    lap(nullptr);
    // the wave-level builder serves the default shape; other shapes (experiments) keep the per-thread walk, one slot per vector
    const bool wave_builder = st.tile_builder != 0 && sh.K == 2u && ((sh.B == 4u && sh.T == 48u) || tl->flow) && sh.S == 32u && sh.nset == 1u;
    const bool split = wave_builder && st.tile_split != 0;
    tl->split_x = split ? st.tile_split_x : 0.0;
    tl->split_min = split ? st.tile_split_min : 0.0;
    // ---- slots ----
    if (cp.n_outer + 1 > 0xFFFFFFFFull) fail(SCANRS_ERR_SHAPE, "matrix too large for the tile layout's 32-bit slot index");
    DevBuf<uint32_t> mult(cp.n_outer + 1);
    tl->slot_first.alloc(cp.n_outer + 1);
    // (the slot rule is stated in nonzeros per 48 inner positions; the flow form walks tiles of 32 rows but deals its slots by the same densities:
    // a slot then holds two thirds of the records per tile, and no vector that owned a slot loses it)
    hipLaunchKernelGGL(tile_mult_kernel, grid_1d(cp.n_outer + 1), dim3(256), 0, s, cp.indptr.p, cp.n_outer, (double)(tl->flow ? 48u : sh.T) / (double)std::max<uint64_t>(1, cp.n_inner),
                       tl->split_x, tl->split_min, sh.S, mult.p);
    size_t tmp_bytes_u = 0;
    SCANRS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes_u, mult.p, tl->slot_first.p, 0u, (size_t)cp.n_outer + 1, rocprim::plus<uint32_t>(), s));
    DevBuf<char> tmp_u(std::max<size_t>(tmp_bytes_u, 16));
    SCANRS_HIP(rocprim::exclusive_scan(tmp_u.p, tmp_bytes_u, mult.p, tl->slot_first.p, 0u, (size_t)cp.n_outer + 1, rocprim::plus<uint32_t>(), s));
    uint32_t n_slots32 = 0, *d_max = nullptr;
    DevBuf<uint32_t> maxbuf(1);
    d_max = maxbuf.p;
    size_t tmp_bytes_m = 0;
    SCANRS_HIP(rocprim::reduce(nullptr, tmp_bytes_m, mult.p, d_max, 0u, (size_t)cp.n_outer + 1, rocprim::maximum<uint32_t>(), s));
    DevBuf<char> tmp_m(std::max<size_t>(tmp_bytes_m, 16));
    SCANRS_HIP(rocprim::reduce(tmp_m.p, tmp_bytes_m, mult.p, d_max, 0u, (size_t)cp.n_outer + 1, rocprim::maximum<uint32_t>(), s));
    n_slots32 = SCANRS_D2H_VALUE(tl->slot_first.p + cp.n_outer, s);
    tl->max_mult = SCANRS_D2H_VALUE(d_max, s);
    tl->n_slots = n_slots32; // (a u32 scan: a sum past 2^32 would need > 2^32 vectors x 32 slots; the visit-index check below bounds it far lower)
    tl->slot_vec.alloc(std::max<uint64_t>(tl->n_slots, 1));
    hipLaunchKernelGGL(tile_slotvec_kernel, grid_1d(std::max<uint64_t>(cp.n_outer, 1)), dim3(256), 0, s, tl->slot_first.p, cp.n_outer, tl->slot_vec.p);
    tl->n_groups = (tl->n_slots + sh.S - 1) / sh.S;
    // items = (workgroup of TL_NW groups, part of the tile range), dealt round-robin to one persistent workgroup per CU: about
    // 16 rounds, the part count chosen so that the last round is nearly full. A nonzero never waits across a part boundary.
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t wgg = (uint32_t)((tl->n_groups + TL_NW - 1) / TL_NW);
    uint32_t parts = std::max<uint32_t>(1u, (uint32_t)(16ull * (uint32_t)n_cu / std::max<uint32_t>(1u, wgg)));
    parts = std::min<uint32_t>(parts, std::max<uint32_t>(1u, sh.nt / 16u));
    sh.tpp = (sh.nt + parts - 1) / parts;
    sh.n_parts = (sh.nt + sh.tpp - 1) / sh.tpp;
    const uint64_t n_rec = tl->n_groups * sh.nt * sh.nset * 64u;
    if (tl->n_groups * sh.nt > 0xFFFFFFFFull || (sh.nset & (sh.nset - 1u))) fail(SCANRS_ERR_SHAPE, "matrix too large for the tile layout's 32-bit visit index");
    const uint64_t n_seg = cp.n_outer * sh.n_parts;
    lap("slots");
    if (tile_dense_wanted(st)) {
        const bool was_flow = tl->flow;
        tl_flow_fallback = false;
        TileLayout *t = dense_layout_build(st, cp, max_overflow, s, std::move(tl), lap);
        if (!t && was_flow && tl_flow_fallback) {
            if (trace_on()) fprintf(stderr, "[scanrs trace] tile layout: the flow form cannot serve this copy -> dense form\n");
            cp.tile_flow_refused = true;
            return tile_layout_build(st, cp, max_overflow, stream);
        }
        return t;
    }
    // ---- count the overflow per (vector, part) ----
    // one_pass (the wave builder's default): no counting pass - the fill pass writes the records AND the overflow nonzeros (into
    // temporaries indexed like the source, counted per segment as they come), a compaction moves them to their places; the
    // max_overflow verdict then comes after the records were written (a rejected layout is rare and costs one fill pass)
    bool one_pass = wave_builder && st.tile_build_one_pass != 0;
    DevBuf<uint32_t> tmp_oi, tmp_ovv;
    DevBuf<unsigned long long> ovc(n_seg + 1), ovo(n_seg + 1);
    SCANRS_HIP(hipMemsetAsync(ovc.p, 0, (n_seg + 1) * 8, s)); // the slots of a split vector add up theirs
    const dim3 grid((unsigned)((n_seg + 255) / 256));
    const uint64_t n_witems = ((tl->n_groups + 1) / 2) * sh.n_parts;
    if (wave_builder && n_witems > 0x7FFFFFFFull) fail(SCANRS_ERR_SHAPE, "matrix too large for the tile layout builder's grid");
    // waves per CU of the builder: its lanes read 64 different places of the matrix, so the lines in flight (2 x 128 B per lane) of
    // all resident waves must fit the L2 or they come from HBM several times; a dummy LDS allocation per wave caps the occupancy
    const size_t wb_lds = st.tile_build_waves ? std::min<size_t>(65536, TL_LDS / st.tile_build_waves) : 0;
#define SCANRS_ASSIGN(FILLV, KUV, OVC, OFF, CUR, PROW, PCNT, OVI, OVV)                                                                          \
    hipLaunchKernelGGL((tile_assign_wave_kernel<FILLV, 48, KUV>), dim3((unsigned)n_witems), dim3(64), wb_lds, s, cp.indptr.p, cp.indices.p, cp.values.p, \
                       tl->slot_vec.p, tl->slot_first.p, tl->n_slots, tl->n_groups, sh, OVC, OFF, CUR, PROW, PCNT, OVI, OVV)
    if (one_pass) { // its temporaries (8 bytes per nonzero) must fit beside the layout: if they cannot be had, the two-walk form
        try {
            tmp_oi.alloc(std::max<uint64_t>(cp.nnz, 1));
            tmp_ovv.alloc(std::max<uint64_t>(cp.nnz, 1));
        } catch (const Failure &) {
            tmp_oi.release();
            tmp_ovv.release();
            one_pass = false;
        }
    }
    if (one_pass) {
        if (n_rec) {
            tl->prow.alloc(n_rec);
            tl->pcnt.alloc(n_rec);
        }
        lap("hipMalloc of records");
        if (n_witems) {
            if (sh.KU)
                SCANRS_ASSIGN(true, 1, (unsigned long long *)nullptr, (const unsigned long long *)nullptr, ovc.p, tl->prow.p, tl->pcnt.p, tmp_oi.p, tmp_ovv.p);
            else
                SCANRS_ASSIGN(true, 0, (unsigned long long *)nullptr, (const unsigned long long *)nullptr, ovc.p, tl->prow.p, tl->pcnt.p, tmp_oi.p, tmp_ovv.p);
        }
        if (split && n_seg)
            hipLaunchKernelGGL((tile_slotless_kernel<false>), grid, dim3(256), 0, s, cp.indptr.p, cp.indices.p, cp.values.p, tl->slot_first.p, cp.n_outer, sh,
                               ovc.p, (const unsigned long long *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
        lap("fill (one pass)");
    } else if (wave_builder) {
        if (n_witems) {
            if (sh.KU)
                SCANRS_ASSIGN(false, 1, ovc.p, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint16_t *)nullptr, (uint8_t *)nullptr,
                              (uint32_t *)nullptr, (uint32_t *)nullptr);
            else
                SCANRS_ASSIGN(false, 0, ovc.p, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, (uint16_t *)nullptr, (uint8_t *)nullptr,
                              (uint32_t *)nullptr, (uint32_t *)nullptr);
        }
        if (split && n_seg)
            hipLaunchKernelGGL((tile_slotless_kernel<false>), grid, dim3(256), 0, s, cp.indptr.p, cp.indices.p, cp.values.p, tl->slot_first.p, cp.n_outer, sh,
                               ovc.p, (const unsigned long long *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
    } else if (n_seg) {
        hipLaunchKernelGGL((tile_assign_kernel<false>), grid, dim3(256), 0, s, cp.indptr.p, cp.indices.p, cp.values.p, cp.n_outer, sh, ovc.p,
                           (const unsigned long long *)nullptr, (uint16_t *)nullptr, (uint8_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
    }
    size_t tmp_bytes = 0;
    SCANRS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, ovc.p, ovo.p, 0ull, (size_t)n_seg + 1, rocprim::plus<unsigned long long>(), s));
    DevBuf<char> tmp(std::max<size_t>(tmp_bytes, 16));
    SCANRS_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, ovc.p, ovo.p, 0ull, (size_t)n_seg + 1, rocprim::plus<unsigned long long>(), s));
    unsigned long long n_ov = 0;
    n_ov = SCANRS_D2H_VALUE(ovo.p + n_seg, s);
    lap("count + scan");
    if (max_overflow > 0.0 && (double)n_ov > max_overflow * (double)cp.nnz) {
        if (trace_on())
            fprintf(stderr, "[scanrs trace] tile layout: %llu outer x %llu inner: %.1f %% of the nonzeros would overflow (limit %.0f %%) -> gather kernels for this orientation\n",
                    (unsigned long long)cp.n_outer, (unsigned long long)cp.n_inner, 100.0 * (double)n_ov / (double)cp.nnz, 100.0 * max_overflow);
        return nullptr;
    }
    if (n_rec && !one_pass) { // (the weights are allocated with the first map: 32 or 64 per record row, tile_layout_weights)
        tl->prow.alloc(n_rec);
        tl->pcnt.alloc(n_rec);
    }
    if (!one_pass) lap("hipMalloc of records");
    // (the weights need no initialisation: tile_weights_kernel writes every position that is ever read, used or not)
    if (!wave_builder && n_rec) { // the per-thread walk patches records into an initialised layout; the wave builder writes every record itself
        hipLaunchKernelGGL(tile_init_rows_kernel, dim3((unsigned)std::min<uint64_t>((n_rec / 4 + 255) / 256, 1u << 23)), dim3(256), 0, s, tl->prow.p, n_rec, sh);
        SCANRS_HIP(hipMemsetAsync(tl->pcnt.p, 0, n_rec, s));
        lap("init rows + counts");
    }
    SparseCopy &ov = tl->ov;
    ov.n_outer = cp.n_outer;
    ov.n_inner = cp.n_inner;
    ov.nnz = n_ov;
    ov.indptr.alloc(cp.n_outer + 1);
    hipLaunchKernelGGL(tile_ovptr_kernel, dim3((unsigned)((cp.n_outer + 256) / 256)), dim3(256), 0, s, ovo.p, cp.n_outer, sh.n_parts, ov.indptr.p);
    ov.indices.alloc(std::max<uint64_t>(n_ov, 1));
    ov.values.alloc(std::max<uint64_t>(n_ov, 1));
    ov.fvals.alloc(std::max<uint64_t>(n_ov, 1));
    if (one_pass) {
        if (n_ov)
            hipLaunchKernelGGL(tile_ov_compact_kernel, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, s, cp.indptr.p, cp.indices.p, cp.values.p,
                               tl->slot_first.p, cp.n_outer, sh, ovo.p, tmp_oi.p, tmp_ovv.p, ov.indices.p, ov.values.p);
    } else if (wave_builder) {
        // the slots of a split vector take their places in the vector's overflow segment from a cursor (ovc, now a copy of the offsets)
        if (tl->max_mult > 1u) SCANRS_HIP(hipMemcpyAsync(ovc.p, ovo.p, (n_seg + 1) * 8, hipMemcpyDeviceToDevice, s));
        if (n_witems) {
            if (sh.KU)
                SCANRS_ASSIGN(true, 1, (unsigned long long *)nullptr, ovo.p, ovc.p, tl->prow.p, tl->pcnt.p, ov.indices.p, ov.values.p);
            else
                SCANRS_ASSIGN(true, 0, (unsigned long long *)nullptr, ovo.p, ovc.p, tl->prow.p, tl->pcnt.p, ov.indices.p, ov.values.p);
        }
        if (split && n_seg)
            hipLaunchKernelGGL((tile_slotless_kernel<true>), grid, dim3(256), 0, s, cp.indptr.p, cp.indices.p, cp.values.p, tl->slot_first.p, cp.n_outer, sh,
                               (unsigned long long *)nullptr, ovo.p, ov.indices.p, ov.values.p);
    } else if (n_seg) {
        hipLaunchKernelGGL((tile_assign_kernel<true>), grid, dim3(256), 0, s, cp.indptr.p, cp.indices.p, cp.values.p, cp.n_outer, sh,
                           (unsigned long long *)nullptr, ovo.p, tl->prow.p, tl->pcnt.p, ov.indices.p, ov.values.p);
    }
#undef SCANRS_ASSIGN
    SCANRS_HIP(hipGetLastError());
    lap(one_pass ? "overflow compaction" : "fill");
    if (n_ov && tl->max_mult > 1u) {
        // the overflow nonzeros of a split vector arrived in the order of its lanes: back into ascending index order (what the gather
        // kernel's bounds table and the reference's accumulation order want) by a segmented sort — over the split vectors only (the
        // others are segments of length 0 to the sort: a tenth of the cells of the headline matrix are split, 4.5 -> 1.5 ms)
        if (n_ov > 0xFFFFFFFFull) fail(SCANRS_ERR_SHAPE, "overflow part of a split layout beyond 2^32-1 nonzeros is not supported");
        DevBuf<uint32_t> ki(n_ov), vi2(n_ov);
        unsigned end_bit = 1;
        while (end_bit < 32u && (cp.n_inner >> end_bit) != 0) end_bit++;
        const SplitSegment seg_begin{ov.indptr.p, tl->slot_first.p, 0}, seg_end{ov.indptr.p, tl->slot_first.p, 1};
        auto it_b = rocprim::make_transform_iterator(rocprim::make_counting_iterator<unsigned>(0u), seg_begin);
        auto it_e = rocprim::make_transform_iterator(rocprim::make_counting_iterator<unsigned>(0u), seg_end);
        size_t tb = 0;
        SCANRS_HIP(rocprim::segmented_radix_sort_pairs(nullptr, tb, ov.indices.p, ki.p, ov.values.p, vi2.p, (unsigned)n_ov, (unsigned)cp.n_outer, it_b, it_e, 0u,
                                                       end_bit, s));
        DevBuf<char> tsort(std::max<size_t>(tb, 16));
        SCANRS_HIP(rocprim::segmented_radix_sort_pairs(tsort.p, tb, ov.indices.p, ki.p, ov.values.p, vi2.p, (unsigned)n_ov, (unsigned)cp.n_outer, it_b, it_e, 0u,
                                                       end_bit, s));
        hipLaunchKernelGGL(tile_split_copyback_kernel, grid_1d(cp.n_outer), dim3(256), 0, s, ov.indptr.p, tl->slot_first.p, cp.n_outer, ki.p, vi2.p,
                           ov.indices.p, ov.values.p);
        SCANRS_SYNC(s);
        lap("overflow sort");
    }
    if (n_ov) ensure_bounds_public(st, ov, s);
    SCANRS_SYNC(s); // the temporaries are released on return
    lap("overflow bounds");
    if (trace_on())
        fprintf(stderr, "[scanrs trace] tile layout: %llu outer x %llu inner, T %u x B %u, K %u, S %u, %llu slots (most per vector %u) in %llu groups x %u tiles in %u parts, nnz %llu, overflow %llu (%.1f %%), positions per nonzero %.2f, %.2f GB\n",
                (unsigned long long)cp.n_outer, (unsigned long long)cp.n_inner, sh.T, sh.B, sh.K, sh.S, (unsigned long long)tl->n_slots, tl->max_mult,
                (unsigned long long)tl->n_groups, sh.nt, sh.n_parts, (unsigned long long)cp.nnz, (unsigned long long)n_ov,
                100.0 * (double)n_ov / (double)std::max<uint64_t>(1, cp.nnz), ((double)tl->n_slots * sh.nt * sh.K) / (double)std::max<uint64_t>(1, cp.nnz),
                tl->bytes() / 1e9);
    return tl.release();
}

// Is f(1, outer, inner) a product of an outer and an inner factor? Yes when the links in front of the first nonlinear link
// (log, square) index one side only and no binomial residual link takes part; nl_outer = the side that owns them.
static bool tile_map_separable(const DevMap &map, int &nl_outer) {
    bool nonlinear = false, pre_outer = false, pre_inner = false, post_scale = false;
    for (int i = 0; i < map.n; i++) {
        const int k = map.ops[i].kind;
        if (k == OP_BINOM_DEV || k == OP_BINOM_PEARSON) return false;
        if (k == OP_LN_1P || k == OP_LOG2_1P || k == OP_LOG10_1P || k == OP_SQUARE) {
            if (post_scale) return false; // a scale between two nonlinear links does not factor out of the second one
            nonlinear = true;
        }
        if (k == OP_SCALE_AXIS) {
            if (!nonlinear)
                (map.ops[i].a_outer ? pre_outer : pre_inner) = true;
            else
                post_scale = true;
        }
    }
    if (nonlinear && pre_outer && pre_inner) return false;
    nl_outer = pre_inner ? 0 : 1;
    return true;
}

static void dense_layout_weights(Storage &st, TileLayout &tl, const SparseCopy &cp, const DevMap &map); // below
// weights of every position and of the overflow part under `map`
static void tile_layout_weights(Storage &st, TileLayout &tl, const SparseCopy &cp, const DevMap &map) {
    Tick tick("tile layout weights");
    if (tl.dense) {
        dense_layout_weights(st, tl, cp, map);
        return;
    }
    const uint64_t n_rec = tl.prow.n;
    int nl_outer = 1;
    tl.unit_mode = tl.sh.KU > 0 && tile_map_separable(map, nl_outer);
    if (tl.unit_mode) {
        if (tl.uo.n != cp.n_outer) tl.uo.alloc(std::max<uint64_t>(cp.n_outer, 1));
        if (tl.vi.n != cp.n_inner) tl.vi.alloc(std::max<uint64_t>(cp.n_inner, 1));
        hipLaunchKernelGGL(tile_unit_factor_kernel, dim3((unsigned)((cp.n_outer + 255) / 256)), dim3(256), 0, st.stream, map, 1, nl_outer, cp.n_outer, tl.uo.p);
        hipLaunchKernelGGL(tile_unit_factor_kernel, dim3((unsigned)((cp.n_inner + 255) / 256)), dim3(256), 0, st.stream, map, 0, nl_outer, cp.n_inner, tl.vi.p);
        if (tl.uo_slot.n != std::max<uint64_t>(tl.n_slots, 1)) tl.uo_slot.alloc(std::max<uint64_t>(tl.n_slots, 1));
        if (tl.n_slots)
            hipLaunchKernelGGL(tile_gather_slots_kernel, grid_1d(tl.n_slots), dim3(256), 0, st.stream, tl.uo.p, tl.slot_vec.p, tl.n_slots, tl.uo_slot.p);
        const uint64_t n_side = nl_outer ? cp.n_outer : cp.n_inner;
        if (tl.ratio_tab.n != n_side * TL_TABC) tl.ratio_tab.alloc(std::max<uint64_t>(n_side * TL_TABC, 1));
        hipLaunchKernelGGL(tile_ratio_table_kernel, grid_1d(n_side * TL_TABC), dim3(256), 0, st.stream, map, nl_outer, n_side, tl.ratio_tab.p);
    }
    if (n_rec) {
        // the unit positions are lanes [0, KU sps) of a record row; the lanes behind them must be a power of two for the kernel's index split
        const uint32_t skip = tl.unit_mode && tl.sh.KU * tl.sh.sps == 32u ? 32u : 0u;
        const uint64_t n_w = skip ? n_rec / 2 : n_rec; // unit mode: weights of the general half only (5.8 GB less per layout at 10^9 nonzeros)
        if (tl.pw.n != n_w) tl.pw.alloc(n_w);
        const uint64_t n_work = (((n_rec >> 6) + 3) / 4) * (64u - skip); // 4 record rows per work-item
        const uint32_t vquads = (tl.sh.nt + 3u) / 4u;
        // the table indexed by the inner position: visit-major order (see the kernel)
        const bool vmajor = tl.unit_mode && !nl_outer && tl.sh.nset == 1u && vquads <= 65535u && tl.n_groups <= 0xFFFFFFFFull / 64u;
        dim3 grid((unsigned)std::min<uint64_t>((n_work + 255) / 256, 1u << 23));
        if (vmajor) grid = dim3((unsigned)((tl.n_groups * (64u - skip) + 255) / 256), vquads);
        if (skip == 32u && st.tile_weights_wide && tl.sh.K == 2u && tl.sh.S == 32u && tl.sh.B == 4u && tl.sh.T == 48u && tl.sh.nset == 1u && (!vmajor || true)) {
            // the wide form (four positions per thread): same rows in the same order
            const uint64_t n_rows = n_rec >> 6;
            const uint64_t lin = (vmajor ? tl.n_groups : (n_rows + 3) / 4) * 8u;
            dim3 g2((unsigned)std::min<uint64_t>((lin + 255) / 256, 1u << 23), vmajor ? vquads : 1u);
            hipLaunchKernelGGL((tile_weights_unit_kernel<48>), g2, dim3(256), 0, st.stream, tl.prow.p, tl.pcnt.p, tl.pw.p, n_rows, tl.slot_vec.p, tl.n_slots,
                               tl.sh.nt, map, tl.uo.p, tl.vi.p, tl.ratio_tab.p, nl_outer, vmajor ? (uint32_t)tl.n_groups : 0u);
        } else
        hipLaunchKernelGGL(tile_weights_kernel, grid, dim3(256), 0, st.stream, tl.prow.p, tl.pcnt.p, tl.pw.p, n_rec,
                           tl.slot_vec.p, tl.n_slots, tl.sh, map, tl.unit_mode ? tl.uo.p : (const double *)nullptr, tl.unit_mode ? tl.vi.p : (const double *)nullptr, skip,
                           tl.unit_mode ? tl.ratio_tab.p : (const double *)nullptr, nl_outer, vmajor ? (uint32_t)tl.n_groups : 0u);
    }
    if (tl.ov.nnz) materialize_map_values(st, tl.ov, map, tl.ov.fvals.p);
    SCANRS_HIP(hipGetLastError());
    if (trace_on()) (void)wait_stream_quiet(st.stream);
    tl.sig_n = map.n;
    for (int i = 0; i < map.n; i++) {
        tl.sig_id[i] = map.ops[i].id;
        tl.sig_outer[i] = map.ops[i].a_outer;
    }
}

static void flow_layout_streams(Storage &st, TileLayout &tl, const uint32_t *cnt, const uint64_t *off, const uint32_t *sorted, uint64_t n_served, hipStream_t s,
                                const std::function<void(const char *)> &lap); // tiles_flow.inc
#include "tiles_dense.inc"
#include "tiles_flow.inc"

// ---- product -------------------------------------------------------------------------------------------------------
namespace {

struct TileArgs {
    const uint16_t *prow;
    const double *pw;
    const double *uo; // unit mode: factor applied to a vector's sum at the end
    uint32_t *next_item; // items are taken from this counter (zeroed before the launch)
    uint64_t n_groups, n_outer, n_inner;
    TileShape sh;
};

// parts[part][outer][:] = sum over the part's visits of weight * X[inner, :]
// KU > 0 (unit mode): position j < KU of every (slot, visit) pair holds a count-1 nonzero or nothing: its row is ADDED (no
// weight, no v_readlane of one: the panel staged here was scaled by the per-inner factor, the sums are scaled by the
// per-outer factor at the end); an unused unit position reads a row of zeros kept behind the ring.
template <int K, int S, int KU>
__device__ __forceinline__ void spmm_tile_body(const TileArgs &ta, const double *__restrict__ X, uint32_t ldx, uint32_t l,
                                               double *__restrict__ parts, uint32_t ldo, uint64_t part_stride, uint32_t n_items) {
    constexpr int SPS = 64 / K;
    constexpr int NSET = (S + SPS - 1) / SPS;
    // weights per record row: unit mode (K = 2, one unit position per slot) stores those of the general half only — lanes 0-31 load the
    // same words as lanes 32-63 and never use them
    constexpr uint32_t WPR = KU > 0 ? 32u : 64u;
    constexpr bool SROWS = TL_SROWS && NSET == 1 && K == 2;
    // DPPW: lane L holds the weights L % 16, 16 + L % 16, ... of the record row (one register pair per 16 weights), every row of 16
    // lanes the same ones: a general position's FMA takes its weight from lane n of the lane's own row (fmac_bcast)
    constexpr bool DPPW = TL_DPPW != 0 && (KU > 0 || S > 28); // (the 168-register kernels without unit positions have no room for 4 weight pairs)
    constexpr int NWV = DPPW ? (int)(WPR / 16u) : 1;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t rowbytes = ldx * 8u;
    const uint32_t tile_bytes = ta.sh.T * rowbytes;
    const uint32_t tpp = ta.sh.tpp, nt = ta.sh.nt, nbuf = ta.sh.B;
    const uint32_t n_wgg = (uint32_t)((ta.n_groups + TL_NW - 1) / TL_NW);
    const char *Xb = reinterpret_cast<const char *>(X);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
    if (ta.sh.KU) { // the row of zeros behind the ring (unused unit positions); visible to everybody after the first barrier
        const uint32_t zoff = nbuf * tile_bytes;
        for (uint32_t i = threadIdx.x * 16u; i < rowbytes; i += blockDim.x * 16u) *reinterpret_cast<d2 *>(lds + zoff + i) = (d2){0.0, 0.0};
    }

    // LDS-DMA staging of tile t into ring buffer `buf`: 1 KB per wave-instruction, chunk i of this wave. No branches (a join
    // makes the compiler wait for ALL outstanding LDS reads at the next use) and hand-issued for the same reason (with the
    // builtin anywhere in the block of the position pipeline the compiler stops counting LDS reads): M0 = LDS destination of
    // lane 0, 16 bytes per lane. No lane is ever switched off (round 3/4 masked the lanes past the end of the tile or of the
    // panel through EXEC: v_cmpx + two s_mov + wait states, 7 instructions per chunk and 12 per visit for the limit, a tenth
    // of the instructions of a visit): a chunk that would reach past the end of the tile — the next ring buffer is live — is
    // moved back to END at the tile's end (it writes some bytes a neighbour writes too, the same values), one that does not
    // exist becomes that last chunk once more, and the panel the kernel reads is always the library's own compact copy, which
    // has two tiles of slack behind it (launch_spmm_tiles): what is staged from there lands in ring rows no record refers to
    // (rows past the panel's end; the tile after the last one goes into a free buffer).
    constexpr int CH = (int)((24u * K * TL_LMAX * 8u / 1024u + TL_NW - 1) / TL_NW); // chunks per wave and tile (tiles of <= 24 K rows)
    uint32_t coff[CH], voff[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        coff[i] = min((wave + (uint32_t)i * TL_NW) * 1024u, tile_bytes - 1024u);
        voff[i] = coff[i] + lane * 16u;
        asm volatile("" : "+s"(coff[i]), "+v"(voff[i])); // computed once, kept in registers
    }
    auto stage_set = [&](uint32_t buf, uint32_t i) { // (M0 is nobody else's on gfx9: LDS instructions do not use it)
        const uint32_t base = lds0 + buf * tile_bytes;
        asm volatile("s_add_u32 m0, %0, %1" ::"s"(base), "s"(coff[i]) : "memory", "scc");
    };
    auto stage_go = [&](uint32_t t, uint32_t i) { // at least one instruction after stage_set (M0 write -> LDS-DMA: 1 wait state)
        const char *sbase = Xb + (uint64_t)t * tile_bytes;
        asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff[i]), "s"(sbase) : "memory");
    };
    auto stage_chunk = [&](uint32_t t, uint32_t buf, uint32_t i) {
        stage_set(buf, i);
        asm volatile("s_nop 0" ::: "memory");
        stage_go(t, i);
    };
    // Lanes that own no column pair still take part in every ds_read_b128, which the LDS serves in four groups of 16 lanes
    // ({0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63}: MI355X_MICROARCH.md, LDS): an idle
    // lane that re-read lane 0's bytes hit the banks of an active lane of ITS group at another address — a 2-way conflict in
    // two of the four groups, 6 LDS cycles per row instead of 4 (measured: SQ_LDS_BANK_CONFLICT = 1/3 of SQ_LDS_IDX_ACTIVE at
    // l = 100). It re-reads the first active lane of its own group instead: same address, broadcast, no extra cycle.
    uint32_t src_lane = lane;
    if (lane * 2u >= l) {
        const uint32_t grp = lane < 32u ? (((lane >= 4u && lane < 12u) || (lane >= 16u && lane < 20u) || lane >= 28u) ? 1u : 0u)
                                        : (((lane >= 36u && lane < 44u) || (lane >= 48u && lane < 52u) || lane >= 60u) ? 3u : 2u);
        const uint32_t first = grp == 0u ? 0u : grp == 1u ? 4u : grp == 2u ? 32u : 36u;
        src_lane = first * 2u < l ? first : 0u;
    }
    const uint32_t lcol16 = src_lane * 16u;
    const lds_cptr_t ring = (lds_cptr_t)(lds + lcol16);
    uint32_t rowbytes_v = rowbytes;
    asm volatile("" : "+v"(rowbytes_v)); // a vector register: v_mad_u32_u16 takes one scalar operand
    // record rows are 16 bits; a lane loads the aligned dword that holds the two rows of its pair
    const uint32_t *prow32 = reinterpret_cast<const uint32_t *>(ta.prow);

    __shared__ uint32_t s_item;
    for (;;) {
        // Items are taken from a counter, in order: a workgroup that starts late (its CU was still running a dense kernel of the
        // auxiliary stream) or draws slow items simply takes fewer — with a fixed stride the launch ended when its unluckiest
        // workgroup did (the pass after the last Gram-Schmidt block ran 22.3 instead of 20.4 ms).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // everyone is done with the ring of the previous item and has read s_item
        if (threadIdx.x == 0) s_item = atomicAdd(ta.next_item, 1u);
        __syncthreads();
        const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_item);
        if (item >= n_items) break;
        // part-major: the workgroups resident at one time walk the SAME range of tiles, so a tile comes from HBM once and from
        // L2 / Infinity Cache for everybody else (with the parts interleaved every workgroup streamed its own range of the
        // 800 MB panel of the gene-major product: 118 GB of staging reads per pass, 5 TB/s, and that bound the pass)
        const uint32_t part = item / n_wgg, wgg = item - part * n_wgg;
        const uint32_t t0 = part * tpp, t1 = min(nt, t0 + tpp);
        const uint64_t group_raw = (uint64_t)wgg * TL_NW + wave;
        const bool live = group_raw < ta.n_groups;
        const uint64_t group = live ? group_raw : ta.n_groups - 1; // idle waves shadow the last group (they still stage tiles and meet the barriers)
        uint32_t bufn = t0 % nbuf; // ring buffer of the tile being staged
#pragma unroll
        for (int i = 0; i < CH; i++) stage_chunk(t0, bufn, i);

        d2 acc[S];
#pragma unroll
        for (int sl = 0; sl < S; sl++) acc[sl] = (d2){0.0, 0.0};

        const size_t vbase = (size_t)group * nt;
        uint32_t crow[NSET], nrow[NSET];
        double cw[NSET][NWV], nw[NSET][NWV];
#pragma unroll
        for (int b = 0; b < NSET; b++) {
            crow[b] = prow32[((vbase + t0) * NSET + b) * 32u + (lane >> 1)]; // (SROWS: brings the first visits' rows into L2)
#pragma unroll
            for (int v = 0; v < NWV; v++)
                cw[b][v] = ta.pw[((vbase + t0) * NSET + b) * WPR + (DPPW ? v * 16u + (lane & 15u) : (lane & (WPR - 1u)))];
        }

        for (uint32_t t = t0; t < t1; t++) {
            // SROWS: the visit's 64 row numbers are the same for every lane — two scalar loads put them into 32 SGPRs, where the
            // address instruction of a position reads its half-word directly (no v_readlane per pair of positions: 32 of the ~300
            // issue slots of a visit). Hand-issued: the compiler knows nothing of them (it would wait for ALL LDS reads wherever
            // a scalar load is outstanding); they are in flight across the DMA wait and the barrier only, where no LDS read is,
            // and the wait behind the barrier hands the registers to the compiler. The lines were brought into L2 by the vector
            // load of a lane pair two visits earlier (nrow below).
            u32x16 ra, rb;
            if constexpr (SROWS) {
                const uint32_t *rp = prow32 + (vbase + t) * 32u;
                asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(ra), "=&s"(rb) : "s"(rp) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's LDS-DMA chunks of tile t (the compiler does not see them)
            if constexpr (SROWS) {
#pragma unroll
                for (int b = 0; b < NSET; b++) asm volatile("" : "+v"(crow[b])); // keeps the L2 prefetch of the rows alive
            }
#ifndef TL_EXPERIMENT_NO_BARRIER // timing experiment only (wrong results): what the barrier per visit costs
            if constexpr (SROWS) {
                // no fence: the LDS reads of visit t - 1 were consumed by its FMAs, the DMA writes were waited for above
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ra), "+s"(rb)::"memory");
            } else
                __syncthreads(); // tile t is in the ring; everyone is done with visit t - 1, so the buffer of tile t + 1 - B is free
#endif
            bufn = bufn + 1u == nbuf ? 0u : bufn + 1u;
            // the records of the next visit(s), whether they are this item's or not (or nobody's: behind the last group's last visit
            // lies DevBuf's slack): no selection, no branch
            const uint32_t tr = t + 1, tp = t + 2;
            // the next visit's records: nothing in this visit waits for them
#pragma unroll
            for (int b = 0; b < NSET; b++) {
                nrow[b] = prow32[((vbase + (SROWS ? tp : tr)) * NSET + b) * 32u + (lane >> 1)];
#pragma unroll
                for (int v = 0; v < NWV; v++)
                    nw[b][v] = ta.pw[((vbase + tr) * NSET + b) * WPR + (DPPW ? v * 16u + (lane & 15u) : (lane & (WPR - 1u)))];
            }
            // The visit's S K positions as one software pipeline over position g (set g / (SPS K), lane g % (SPS K)): per step
            //   R(g)          v_readlane: the weight's halves (and, every 4th position, the quad's row bytes) into SGPRs; row
            //                 byte -> ring offset by scalar instructions
            //   A(g - 1)      LDS address of the row
            //   L(g - 2)      ds_read_b128 of the row
            //   F(g - 2 - W)  the two FMAs
            // so that no instruction of a step depends on another one of the same step (a wave issues in order and two waves
            // share a SIMD: dependent neighbours leave the vector unit idle), and the next tile's LDS-DMA chunks are issued a
            // few at a time between the steps instead of as one burst that queues on the texture path.
            uint32_t wlo[NSET], whi[NSET];
#pragma unroll
            for (int b = 0; b < NSET; b++) {
                wlo[b] = (uint32_t)__double2loint(cw[b][0]);
                whi[b] = (uint32_t)__double2hiint(cw[b][0]);
            }
            constexpr int PPS = SPS * K;      // positions per full set
            constexpr int NPT = S * K;        // positions per visit
            constexpr int WR = TL_W + 3;      // weights live from R(g) to F(g)
            // the next tile's LDS-DMA chunks go out in the first third of the visit, a few steps apart (all at once they queue on
            // the texture path; spread over the whole visit the last ones are still in flight at the barrier: -1.5 ms per pass)
            constexpr int DM = NPT / (3 * CH) > 1 ? NPT / (3 * CH) : 1;
            uint32_t rows4 = 0;
            uint32_t offs[2];
            lds_cptr_t addr[2];
            double wq[WR];
            d2 x[TL_W];
#pragma unroll
            for (int i = 0; i < NPT + TL_W + 2; i++) {
                // position k of the visit -> set, position j of slot q (j-major inside a set; a last, partial set has fewer slots)
                auto set_of = [](int k) constexpr { return k / PPS; };
                auto ns_of = [](int k) constexpr { return S - SPS * (k / PPS) < SPS ? S - SPS * (k / PPS) : SPS; };
                auto lane_of = [&](int k) constexpr { return ((k % PPS) / ns_of(k)) * SPS + (k % PPS) % ns_of(k); };
                auto slot_of = [&](int k) constexpr { return set_of(k) * SPS + (k % PPS) % ns_of(k); };
                auto unit_of = [&](int k) constexpr { return (k % PPS) / ns_of(k) < KU; };
                if (i >= 2 + TL_W) { // F(i - 2 - W)
                    const int g = i - 2 - TL_W;
                    const int sl = slot_of(g);
                    if (unit_of(g)) { // unit position: the row as it is
                        acc[sl].x += x[g % TL_W].x;
                        acc[sl].y += x[g % TL_W].y;
                    } else if constexpr (DPPW) {
                        const int wi = lane_of(g) & (int)(WPR - 1u);
                        acc[sl].x = fmac_bcast(acc[sl].x, cw[set_of(g)][wi / 16], x[g % TL_W].x, wi % 16);
                        acc[sl].y = fmac_bcast(acc[sl].y, cw[set_of(g)][wi / 16], x[g % TL_W].y, wi % 16);
                    } else {
                        acc[sl].x = fma(wq[g % WR], x[g % TL_W].x, acc[sl].x);
                        acc[sl].y = fma(wq[g % WR], x[g % TL_W].y, acc[sl].y);
                    }
                    // pin the FMAs here (pure arithmetic: without a use the compiler sinks them to the end of the kernel); the DPP form
                    // is an asm statement already (and an asm that reads its result makes the compiler put an s_nop between them)
                    if (unit_of(g) || !DPPW) asm volatile("" : "+v"(acc[sl].x), "+v"(acc[sl].y));
                }
                if (i >= 1 && i - 1 < NPT) { // A(i - 1)
                    const int g = i - 1;
                    // ring + row * rowbytes in one vector instruction (the address now, not in front of the read); the row is the low
                    // or the high half of the pair's scalar register, picked by op_sel: no scalar extract
                    uint32_t sr;
                    if constexpr (SROWS) {
                        const int d = lane_of(g) >> 1;
                        sr = d < 16 ? ra[d & 15] : rb[d & 15];
                    } else
                        sr = offs[g % 2];
                    if (lane_of(g) % 2 == 0)
                        asm volatile("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(addr[g % 2]) : "s"(sr), "v"(rowbytes_v), "v"(ring));
                    else
                        asm volatile("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(addr[g % 2]) : "s"(sr), "v"(rowbytes_v), "v"(ring));
                }
                if (i >= 2 && i - 2 < NPT) { // L(i - 2)
                    const int g = i - 2;
                    x[g % TL_W] = *(const __attribute__((address_space(3))) d2 *)addr[g % 2];
                }
                if (i < NPT) { // R(i)
                    const int b = set_of(i), p = lane_of(i);
                    if constexpr (!SROWS) {
                        if (p % 2 == 0 || (i % PPS) % ns_of(i) == 0) {
                            asm volatile("" : "+v"(crow[b])); // read the pair's rows here, not 64 steps early (they would fill the SGPR file)
                            rows4 = rdlane(crow[b], p & ~1);
                        }
                        offs[i % 2] = rows4;
                    }
                    if constexpr (!DPPW)
                        if (!unit_of(i)) wq[i % WR] = __hiloint2double((int)rdlane(whi[b], p), (int)rdlane(wlo[b], p));
                }
                if constexpr (DM >= 2) { // M0 in one step, the load in the next: no wait state to pay for
                    if (i % DM == 0 && i / DM < CH) stage_set(bufn, i / DM);
                    if (i % DM == 1 && i / DM < CH) stage_go(t + 1, i / DM);
                } else if (i / DM < CH)
                    stage_chunk(t + 1, bufn, i / DM);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int b = 0; b < NSET; b++) {
                crow[b] = nrow[b];
#pragma unroll
                for (int v = 0; v < NWV; v++) cw[b][v] = nw[b][v];
            }
        }
        if (live && lane * 2u < l) {
            double *dst = parts + (size_t)part * part_stride;
#pragma unroll
            for (int sl = 0; sl < S; sl++) {
                const uint64_t o = group * S + sl;
                if (o < ta.n_outer) {
                    d2 r = acc[sl];
                    if (KU > 0) { // the per-outer factor of the unit weight
                        const double uf = ta.uo[o];
                        r.x *= uf;
                        r.y *= uf;
                    }
                    *reinterpret_cast<d2 *>(dst + o * ldo + lane * 2u) = r;
                }
            }
        }
    }
}

template <int K, int S, int KU>
__global__ __launch_bounds__(64 * TL_NW, 2) void spmm_tile_kernel(TileArgs ta, const double *__restrict__ X, uint32_t ldx, uint32_t l,
                                                                  double *__restrict__ parts, uint32_t ldo, uint64_t part_stride,
                                                                  uint32_t n_items) {
    spmm_tile_body<K, S, KU>(ta, X, ldx, l, parts, ldo, part_stride, n_items);
}
// <= 168 VGPRs: 2 tile waves + 2 gather waves of 88 per SIMD
template <int K, int S, int KU>
__global__ __launch_bounds__(64 * TL_NW) __attribute__((amdgpu_waves_per_eu(3, 3))) void spmm_tile_kernel_r168(
    TileArgs ta, const double *__restrict__ X, uint32_t ldx, uint32_t l, double *__restrict__ parts, uint32_t ldo, uint64_t part_stride,
    uint32_t n_items) {
    spmm_tile_body<K, S, KU>(ta, X, ldx, l, parts, ldo, part_stride, n_items);
}

// out[o, :] = sum over the vector's slots (in order) of the sum over parts (in order) + overflow sum + LowRankOffset term
// (sqz/src/low_rank_offset.rs:76-80)
__global__ __launch_bounds__(256) void tile_finish_kernel(const double *__restrict__ parts, uint32_t n_parts, uint64_t part_stride,
                                                          const uint32_t *__restrict__ slot_first, const uint32_t *__restrict__ slot_pos,
                                                          const double *__restrict__ ovout, uint64_t n_outer,
                                                          uint32_t l, uint32_t ldp, uint32_t ldo, double *__restrict__ out, const double *__restrict__ off_a,
                                                          uint32_t rank, const double *__restrict__ off_w, uint32_t ldw, const uint64_t *__restrict__ ov_indptr,
                                                          const double *__restrict__ uo) {
    const uint32_t hp = (l + 1u) / 2u; // column pairs
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_outer * hp) return;
    const uint64_t o = e / hp;
    const uint32_t c = (uint32_t)(e % hp) * 2u;
    d2 s = (d2){0.0, 0.0};
    const uint32_t s0 = slot_first[o], s1 = slot_first[o + 1];
    for (uint32_t sl = s0; sl < s1; sl++) {
        const size_t at = slot_pos ? slot_pos[sl] : sl; // where the slot's sums lie (dense layout: slots sorted by load)
        d2 t = *reinterpret_cast<const d2 *>(parts + at * ldp + c);
        for (uint32_t p = 1; p < n_parts; p++) {
            const d2 u = *reinterpret_cast<const d2 *>(parts + (size_t)p * part_stride + at * ldp + c);
            t.x += u.x;
            t.y += u.y;
        }
        if (sl == s0) {
            s = t; // (one slot per vector: bit for bit the sum the unsplit layout gave)
        } else {
            s.x += t.x;
            s.y += t.y;
        }
    }
    if (uo) { // dense layout, outer side's factor kept out of the weights (TileLayout::fold_outer)
        const double f = uo[o];
        s.x *= f;
        s.y *= f;
    }
    if (ovout && ov_indptr[o + 1] > ov_indptr[o]) { // (the gather writes the rows of vectors that have overflow nonzeros, and only those)
        const d2 t = *reinterpret_cast<const d2 *>(ovout + o * ldp + c);
        s.x += t.x;
        s.y += t.y;
    }
    for (uint32_t q = 0; q < rank; q++) {
        const double aq = off_a[o * rank + q];
        s.x += aq * off_w[(size_t)q * ldw + c];
        if (c + 1 < l) s.y += aq * off_w[(size_t)q * ldw + c + 1];
    }
    *reinterpret_cast<d2 *>(out + o * ldo + c) = s;
}

template <int K, int S, int KU>
void launch_tile_kernel(Storage &st, const TileArgs &ta, const double *X, uint32_t ldx, uint32_t l, double *parts, uint32_t ldo,
                        uint64_t part_stride, uint32_t n_items, uint32_t grid) {
    const size_t shmem = ((size_t)ta.sh.B * ta.sh.T + (ta.sh.KU ? 1u : 0u)) * ldx * 8; // the ring (+ the row of zeros)
    if constexpr (S <= 28) {
        SCANRS_HIP(hipFuncSetAttribute((const void *)spmm_tile_kernel_r168<K, S, KU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((spmm_tile_kernel_r168<K, S, KU>), dim3(grid), dim3(64 * TL_NW), shmem, st.stream, ta, X, ldx, l, parts, ldo, part_stride,
                           n_items);
    } else {
        SCANRS_HIP(hipFuncSetAttribute((const void *)spmm_tile_kernel<K, S, KU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((spmm_tile_kernel<K, S, KU>), dim3(grid), dim3(64 * TL_NW), shmem, st.stream, ta, X, ldx, l, parts, ldo, part_stride,
                           n_items);
    }
}

} // namespace

bool spmm_tiles_ok(const Storage &st, const SparseCopy &cp, uint32_t ldx, uint32_t l) {
    return l >= 16 && l <= TL_LMAX && (ldx & 1u) == 0 && cp.n_outer > 0 && cp.n_inner > 0 && cp.nnz > 0 &&
           ((size_t)st.tile_b * st.tile_t + 1u) * even_up(l) * 8 <= TL_LDS && (size_t)st.tile_t * even_up(l) * 8 >= 1024u; // (a tile is at least one staging chunk)
}

// Auto path (spmm_path 0): the hybrid product serves a large matrix once its layout under this map exists — built when a
// solver announces many products (Storage::tile_hint) or when the same map comes by a second time, and only if the device
// has room for it (about 16 bytes per nonzero per orientation).
static uint64_t tile_shape_signature(const Storage &st) {
    return 1ull + st.tile_k + 16ull * st.tile_s + 4096ull * st.tile_t + (1ull << 24) * st.tile_b + (1ull << 32) * st.tile_ku +
           (1ull << 34) * (uint64_t)(st.tile_max_overflow * 1000.0);
}
// large enough, the visit index fits, and this shape has not been found wanting for this copy before
static bool tile_auto_candidate(const Storage &st, const SparseCopy &cp) {
    if (!st.tile_auto || cp.nnz < std::max<uint64_t>(st.blocked_min_nnz, 1ull << 24)) return false;
    if (!tile_shape_ok(st.tile_k, st.tile_s, st.tile_t, st.tile_b)) return false;
    if (((cp.n_outer + st.tile_s - 1) / st.tile_s) * ((cp.n_inner + st.tile_t - 1) / st.tile_t) > 0xFFFFFFFFull) return false; // 32-bit visit index
    return cp.tile_rejected_shape != tile_shape_signature(st);
}
bool tile_layout_build_auto(Storage &st, SparseCopy &cp, hipStream_t s) {
    if (!tile_auto_candidate(st, cp)) return false;
    if (cp.tiles && cp.tiles->structure_matches(st, cp)) return true;
    // room: records (11 B per position: row 2, count 1, weight 8) + overflow + the build's temporaries + the partial-sum buffers, and 8 GB for the solver
    const double nt = (double)((cp.n_inner + st.tile_t - 1) / st.tile_t);
    // (dense layout: 12.5 B per position at about 1.1 positions per nonzero, 4 B per nonzero of build temporaries and the per-(group, visit) tables)
    const double need = tile_dense_wanted(st)
                            ? 18.0 * (double)cp.nnz + 40.0 * (double)((cp.n_outer + st.tile_s - 1) / st.tile_s + 8) * (nt + 1.0) + 2.0 * (double)cp.n_outer * 104.0 * 8.0 * 2.0
                            : 11.0 * 64.0 * (double)((cp.n_outer + st.tile_s - 1) / st.tile_s) * nt * ((st.tile_s + 64 / st.tile_k - 1) / (64 / st.tile_k)) +
                                  0.2 * 12.0 * (double)cp.nnz + 32.0 * (double)cp.n_outer * 64.0 + 2.0 * (double)cp.n_outer * 104.0 * 8.0 * 2.0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    // what the driver has free, what a stale layout gives back, and the released blocks the library keeps for reuse (device_alloc
    // hands them back to the driver before it gives up): without the last term a shard whose first PCA ran with one orientation on
    // the gather kernels for lack of room never got that layout later either (2.6 10^9 nonzeros, k = 100: 2.26 instead of 1.86 s per step).
    // The unused part of a reserve made ahead of time is NOT counted: the solver's panels will want it.
    // Round 5: a reserve counts too, minus what a solver that has not started yet will want from it - 26 bytes per nonzero cover the
    // panels of a top-100 PCA on the 30 M-cell shard (23 B per nonzero there). Without it the shard's second layout was refused during
    // its first call (the reserve held the memory the driver no longer reported free) and built in the second call, outside the reserve.
    // (the helper thread, which may run ahead of the solver, and a build outside a solver; the main thread inside a solver has its panels already)
    const bool panels_taken = st.tile_hint > 0 && s == st.stream;
    const double from_reserve = std::max(0.0, (double)device_reserve_unused_bytes() - (panels_taken ? 0.0 : 26.0 * (double)cp.nnz));
    const double have = (double)free_b + (double)device_cache_bytes() + (cp.tiles ? cp.tiles->bytes() : 0.0) + from_reserve;
    if (!(have > need + 8.0 * (double)(1ull << 30))) {
        if (trace_on())
            fprintf(stderr,
                    "[scanrs trace] tile layout: %llu outer x %llu inner not built: needs %.1f GB + 8.6 of headroom, has %.1f GB (driver %.1f, cached blocks %.1f, "
                    "reserve %.1f of %.1f unused%s)\n",
                    (unsigned long long)cp.n_outer, (unsigned long long)cp.n_inner, need / 1e9, have / 1e9, (double)free_b / 1e9, (double)device_cache_bytes() / 1e9,
                    from_reserve / 1e9, (double)device_reserve_unused_bytes() / 1e9, panels_taken ? ", the solver's panels exist" : ", less 26 B per nonzero for a solver's panels");
        return false;
    }
    // Build now: the layout is only worth having when not too many nonzeros miss it (the overflow gather runs at two waves
    // per SIMD beside the tile kernel). At 27-28 % overflow (100 k x 20 k at 5 % density, both orientations) the hybrid product
    // still wins, 50 against 61 ms per PCA; with the heavy-tailed gene profile of
    // real data (genes detected in most cells: tens of nonzeros per 48-cell tile against 2 positions) the gene-major layout of
    // a 10^6 x 33 k matrix overflowed by 70 % and its pass took 109 ms against the gather kernels' 39 (tools/pass_bench.py
    // gene_shape=0.1 shared_profile=1; the cell-major layout of the same matrix: 5.8 %, 19.5 ms). Such an orientation stays on
    // the gather kernels, and is not tried again until the tile shape changes.
    cp.tiles.reset();
    TileLayout *t = nullptr;
    try {
        t = tile_layout_build(st, cp, st.tile_max_overflow, s);
    } catch (const Failure &e) {
        // the room check above is an estimate (cached blocks of other devices, blocks carved from a reserve): when the build
        // itself runs out of device memory the product falls back to the gather kernels instead of failing the solver. Other failures pass.
        if (e.code != SCANRS_ERR_DEVICE || !strstr(scanrs_last_error(), "hipMalloc")) throw;
        if (trace_on()) fprintf(stderr, "[scanrs trace] tile layout: not built, the device ran out of memory during the build (%s) -> gather kernels\n", scanrs_last_error());
        (void)wait_stream_quiet(s);
        return false; // (not remembered as rejected: a later call, with the solver's temporaries gone, may have the room)
    }
    if (!t) {
        cp.tile_rejected_shape = tile_shape_signature(st);
        return false;
    }
    cp.tiles.reset(t, tile_layout_free);
    return true;
}

bool spmm_tiles_auto(Storage &st, SparseCopy &cp, const DevMap &map) {
    if (!tile_auto_candidate(st, cp)) return false;
    if (cp.tiles && cp.tiles->structure_matches(st, cp)) return true; // the weights follow the map in one streaming pass
    bool seen = cp.tsig_n == map.n;
    for (int i = 0; seen && i < map.n; i++) seen = cp.tsig_id[i] == map.ops[i].id && cp.tsig_outer[i] == map.ops[i].a_outer;
    cp.tsig_n = map.n;
    for (int i = 0; i < map.n; i++) {
        cp.tsig_id[i] = map.ops[i].id;
        cp.tsig_outer[i] = map.ops[i].a_outer;
    }
    if (!seen && st.tile_hint <= 0) return false;
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = tile_layout_build_auto(st, cp, st.stream);
    st.t_layout_us += (uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return ok;
}

void launch_gather2d_ov(Storage &st, hipStream_t s, SparseCopy &ov, const double *X, uint32_t ldx, uint32_t l, double *out, uint32_t ldo); // kernels.hip

void launch_spmm_tiles(Storage &st, SparseCopy &cp, const DevMap &map, const double *X, uint32_t ldx, uint32_t l, double *out,
                       uint32_t ldo, const double *off_a, uint32_t rank, const double *off_w, uint32_t ldw) {
    if ((ldx & 1u) || (ldo & 1u)) fail(SCANRS_ERR_ARGUMENT, "panel leading dimensions must be even");
    if (!cp.tiles || !cp.tiles->structure_matches(st, cp)) {
        cp.tiles.reset(); // free the old layout before the new one is allocated
        cp.tiles.reset(tile_layout_build(st, cp, 0.0), tile_layout_free); // forced (spmm_path 3): whatever the overflow
    }
    if (!cp.tiles->weights_match(map)) tile_layout_weights(st, *cp.tiles, cp, map);
    if (cp.tiles->flow && cp.tiles->wsrc == 0) { // a map that does not come from a table (GLM-PCA residuals, tile_fold 0): the dense form and its weight stream
        if (trace_on()) fprintf(stderr, "[scanrs trace] tile layout: the map does not separate -> this orientation is rebuilt in the dense form\n");
        cp.tile_flow_refused = true;
        cp.tiles.reset();
        cp.tiles.reset(tile_layout_build(st, cp, 0.0), tile_layout_free);
        tile_layout_weights(st, *cp.tiles, cp, map);
    }
    TileLayout &tl = *cp.tiles;
    const TileShape &sh = tl.sh;
    // the kernels want compact panel rows (a tile is one contiguous run of bytes): a block of a wider panel is copied first
    const uint32_t ldc = even_up(l);
    const double *Xov = X; // the overflow part works on the panel as it came
    const uint32_t ldxov = ldx;
    // The tile kernel always reads the library's own compact copy: its staging never switches a lane off, so it reads up to two
    // tiles past the panel's end (spmm_tile_body) — the copy has that much slack behind it.
    double *xc = st.scratch.get<double>("tile_xc", ((size_t)cp.n_inner + 3u * sh.T) * ldc); // (the dense kernel stages one tile further: a part ends with a visit of its own)
    // (mat_apply may have filled the copy while it read the panel for the offset term's column sums: tile_panel_copy_target)
    const bool copied = st.tile_xc_src == X && st.tile_xc_l == l && ldc == l && tl.dense && !tl.unit_mode && !tl.fold_inner;
    st.tile_xc_src = nullptr;
    if (tl.unit_mode || (tl.dense && tl.fold_inner)) { // the tile kernel's panel carries the per-inner factor of the weight
        const uint64_t n = cp.n_inner * (uint64_t)ldc;
        hipLaunchKernelGGL(tile_scale_panel_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st.stream, X, ldx, cp.n_inner, l, ldc, tl.vi.p, xc);
    } else {
        if (!copied) launch_copy_cols(st, X, ldx, xc, ldc, cp.n_inner, l);
        if (ldx != ldc) Xov = xc;
    }
    X = xc;
    ldx = ldc;
    int dev = 0, n_cu = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t wgg = (uint32_t)((tl.n_groups + TL_NW - 1) / TL_NW);
    const uint32_t n_items = wgg * sh.n_parts;
    // As many workgroups as the same number of item rounds needs, not as many as there are CUs: 3 977 items of equal size take 16 rounds on
    // 256 workgroups (the last one 47 % empty) and 16 rounds on 249 — and the 7 CUs left over run the side streams' small kernels (the
    // Gram-Schmidt chain of the previous Krylov block) DURING the pass instead of behind it, where a workgroup per CU left them no room
    // (option "tile_spare_cus" 0: one workgroup per CU)
    uint32_t grid = std::min<uint32_t>(n_items, (uint32_t)n_cu);
    if (st.tile_spare_cus && grid == (uint32_t)n_cu) {
        const uint32_t rounds = (n_items + grid - 1u) / grid;
        grid = std::max<uint32_t>((n_items + rounds - 1u) / rounds, grid - grid / 16u); // (never more than a sixteenth of the CUs)
    }
    const uint64_t part_stride = std::max<uint64_t>(tl.n_slots, 1) * (uint64_t)ldc; // partial sums per slot, in compact rows of ldc columns
    double *pbuf = st.scratch.get<double>("tile_parts", (size_t)sh.n_parts * part_stride);
    double *ovout = nullptr;
    if (tl.ov.nnz) { // the overflow part through the texture path, beside the tile kernel
        ovout = st.scratch.get<double>("tile_ovout", (size_t)cp.n_outer * ldc); // the overflow sum per vector
        if (st.tile_overlap) {
            hipStream_t ovs = st.ov();
            SCANRS_HIP(hipEventRecord(st.ev_in, st.stream)); // the panel (and the scratch zero-fills) are ready
            SCANRS_HIP(hipStreamWaitEvent(ovs, st.ev_in, 0));
            launch_gather2d_ov(st, ovs, tl.ov, Xov, Xov == X ? ldx : ldxov, l, ovout, ldc);
            SCANRS_HIP(hipEventRecord(st.ev_ov, ovs));
        } else {
            launch_gather2d_ov(st, st.stream, tl.ov, Xov, Xov == X ? ldx : ldxov, l, ovout, ldc);
        }
    }
    uint32_t *next_item = st.scratch.get<uint32_t>("tile_next_item", 1);
    SCANRS_HIP(hipMemsetAsync(next_item, 0, sizeof(uint32_t), st.stream));
    TileArgs ta{tl.prow.p, tl.pw.p, tl.unit_mode ? tl.uo_slot.p : nullptr, next_item, tl.n_groups, tl.n_slots, cp.n_inner, sh};
    // algorithmic bytes of the nonzeros this kernel works: 8 B each + indptr + the two panels
    const double bytes = (double)(cp.nnz - tl.ov.nnz) * 8.0 + (double)(cp.n_outer + 1) * 8.0 + (double)cp.n_inner * l * 8.0 + (double)cp.n_outer * l * 8.0;
    const bool long_outer = cp.n_outer >= cp.n_inner;
    if (st.prof.on)
        st.prof.begin(st.stream, long_outer ? "spmm_tile_kernel/long-outer" : "spmm_tile_kernel/short-outer", bytes,
                      tl.dense ? (double)tl.dn_served * 8.0 * l /* rows of SERVED nonzeros: padding positions move no useful byte */ : (double)tl.n_groups * sh.nt * sh.S * sh.K * 8.0 * l);
#define SCANRS_TILE(KK, SS, UU) launch_tile_kernel<KK, SS, UU>(st, ta, X, ldx, l, pbuf, ldc, part_stride, n_items, grid)
    const bool um = tl.unit_mode; // a layout with unit positions under a map that does not separate runs the weighted kernel
    if (n_items == 0) { // every vector went to the overflow part: nothing for the tile kernel
    } else if (tl.flow) {
        launch_tile_flow_kernel(st, tl, X, ldx, l, pbuf, ldc, part_stride, next_item, grid);
    } else if (tl.dense) {
        launch_tile_dense_kernel(st, tl, X, ldx, l, pbuf, ldc, part_stride, next_item, grid);
    } else if (sh.K == 2 && sh.S == 32)
        um ? SCANRS_TILE(2, 32, 1) : SCANRS_TILE(2, 32, 0);
    else if (sh.K == 2 && sh.S == 28)
        um ? SCANRS_TILE(2, 28, 1) : SCANRS_TILE(2, 28, 0);
    else if (sh.K == 3 && sh.S == 32)
        SCANRS_TILE(3, 32, 0);
    else if (sh.K == 4 && sh.S == 32)
        SCANRS_TILE(4, 32, 0);
    else if (sh.K == 4 && sh.S == 28)
        SCANRS_TILE(4, 28, 0);
    else
        fail(SCANRS_ERR_ARGUMENT, "unsupported tile shape K=%u S=%u", sh.K, sh.S);
#undef SCANRS_TILE
    if (st.prof.on) st.prof.end(st.stream);
    if (ovout && st.tile_overlap) SCANRS_HIP(hipStreamWaitEvent(st.stream, st.ev_ov, 0));
    const uint64_t n = cp.n_outer * (uint64_t)((l + 1u) / 2u);
    hipLaunchKernelGGL(tile_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st.stream, pbuf, sh.n_parts, part_stride, tl.slot_first.p, tl.slot_pos.p, ovout,
                       cp.n_outer, l, ldc, ldo, out, off_a, rank, off_w, ldw, tl.ov.indptr.p, (tl.dense && tl.fold_outer) ? tl.uo.p : (const double *)nullptr);
    SCANRS_HIP(hipGetLastError());
}

double *tile_panel_copy_target(Storage &st, SparseCopy &cp, uint32_t l) {
    if ((l & 1u) || l < 16u || l > TL_LMAX || !cp.tiles || !cp.tiles->dense || cp.tiles->unit_mode || cp.tiles->fold_inner || !cp.tiles->structure_matches(st, cp)) return nullptr;
    if (!(st.spmm_path == 3 || (st.spmm_path == 0 && st.tile_auto && st.panel_precision == 0))) return nullptr;
    return st.scratch.get<double>("tile_xc", ((size_t)cp.n_inner + 3u * cp.tiles->sh.T) * l); // (the size launch_spmm_tiles asks for: the same buffer)
}

// scanrs_init(): one empty launch per translation unit makes the runtime load this file's code object now instead of inside the
// first real call
__global__ void warm_tiles_kernel() {}
void warm_tiles(hipStream_t s) { hipLaunchKernelGGL(warm_tiles_kernel, dim3(1), dim3(64), 0, s); }

} // namespace scanrs
