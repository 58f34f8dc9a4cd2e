// decode.hip — sqz::AdaptiveVec decode on the device.
//
// A sqz::AdaptiveMat is a slice of AdaptiveVec, each in one of eight encodings (sqz/src/vec.rs:1029-1053). The host
// hands the encoded buffers over as they are (≈4 kB per cell instead of 8 kB per cell for the decoded triplet); they
// are packed into two device arenas and expanded here to the (indptr u64, indices u32, counts u32) triplet every
// other kernel works on. The walk is `AbsIter::next` (vec.rs:96-117): ascending positions, stored zeros skipped.
//
// Work unit = one position (dense encodings D3/D4/D8/D16) or one stored entry (V, S3/S4/S8). A chunk of CH units
// is one workgroup; pass 1 counts the nonzero units of every chunk, an exclusive scan turns the counts into output
// offsets (and indptr), pass 2 decodes again and writes — integer work, bit-exact by construction.
#include <cstring>

#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace scanrs {

namespace {

constexpr uint32_t CH = 2048;     // units per chunk
constexpr uint32_t PER_T = CH / 256;

struct AVDesc {
    uint32_t kind, pad;
    uint64_t len, n_units;
    uint64_t data_off;              // byte arena
    uint64_t ib_off;                // byte arena (S*)
    uint64_t fb_idx_off, fb_val_off, n_fb; // word arena
    uint64_t bs_off, n_bs;          // word arena (S*)
    uint64_t chunk0;                // first chunk of this vector
};

enum { K_D3 = 0, K_D4, K_D8, K_D16, K_V, K_S3, K_S4, K_S8 };

__device__ __forceinline__ uint32_t fallback_get(const uint32_t *__restrict__ fi, const uint32_t *__restrict__ fv, uint64_t n, uint32_t key) {
    // SimpleSparse::get (vec.rs:141-148): binary search, zero when absent
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (fi[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (lo < n && fi[lo] == key) ? fv[lo] : 0u;
}

// one unit -> (position, value); value 0 = nothing to emit
__device__ __forceinline__ void decode_unit(const AVDesc &d, const uint8_t *__restrict__ bytes, const uint32_t *__restrict__ words,
                                            uint64_t u, uint32_t &pos, uint32_t &val) {
    const uint32_t *fi = words + d.fb_idx_off, *fv = words + d.fb_val_off;
    if (d.kind == K_V) { // SimpleSparse::get_nonzero (vec.rs:165-167)
        pos = fi[u];
        val = fv[u];
        return;
    }
    const uint8_t *data = bytes + d.data_off;
    uint32_t raw, th;
    switch (d.kind) {
    case K_D3:
    case K_S3: { // Dense3::get (vec.rs:913-926)
        const uint64_t w = reinterpret_cast<const uint64_t *>(data)[u / 21u];
        raw = (uint32_t)((w >> (3u * (uint32_t)(u % 21u))) & 7ull);
        th = 7u;
        break;
    }
    case K_D4:
    case K_S4: { // Dense4::get (vec.rs:779-791)
        const uint8_t b = data[u >> 1];
        raw = (u & 1ull) ? (uint32_t)(b >> 4) : (uint32_t)(b & 15u);
        th = 15u;
        break;
    }
    case K_D8:
    case K_S8: // DenseW<u8,u32>::get (vec.rs:678-685)
        raw = data[u];
        th = 255u;
        break;
    default: // K_D16
        raw = reinterpret_cast<const uint16_t *>(data)[u];
        th = 65535u;
        break;
    }
    val = raw == th ? fallback_get(fi, fv, d.n_fb, (uint32_t)u) : raw;
    if (d.kind <= K_D16) {
        pos = (uint32_t)u;
    } else { // CompressedIndexSparse::get_nonzero (vec.rs:292-299): block = the one whose entry range holds u
        const uint32_t *bs = words + d.bs_off;
        uint64_t lo = 0, hi = d.n_bs; // first block start > u
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if ((uint64_t)bs[mid] <= u)
                lo = mid + 1;
            else
                hi = mid;
        }
        const uint64_t block = lo - 1;
        pos = (uint32_t)((block << 8) | (uint64_t)bytes[d.ib_off + u]);
    }
}

__device__ __forceinline__ uint32_t find_vec(const AVDesc *__restrict__ desc, uint64_t n_vecs, uint64_t chunk) {
    uint64_t lo = 0, hi = n_vecs; // last vector with chunk0 <= chunk (vectors without units own no chunk)
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (desc[mid].chunk0 <= chunk)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (uint32_t)(lo - 1);
}

template <bool WRITE>
__global__ __launch_bounds__(256) void decode_kernel(const AVDesc *__restrict__ desc, uint64_t n_vecs, const uint8_t *__restrict__ bytes,
                                                     const uint32_t *__restrict__ words, unsigned long long *__restrict__ counts,
                                                     const unsigned long long *__restrict__ offs, uint32_t *__restrict__ out_idx,
                                                     uint32_t *__restrict__ out_val, unsigned long long *__restrict__ bad) {
    __shared__ uint32_t wave_tot[4];
    __shared__ uint32_t vsh;
    const uint64_t chunk = blockIdx.x;
    if (threadIdx.x == 0) vsh = find_vec(desc, n_vecs, chunk);
    __syncthreads();
    const AVDesc d = desc[vsh];
    const uint64_t u0 = (chunk - d.chunk0) * CH + (uint64_t)threadIdx.x * PER_T;
    uint32_t pos[PER_T], val[PER_T];
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t q = 0; q < PER_T; q++) {
        pos[q] = 0;
        val[q] = 0;
        if (u0 + q < d.n_units) {
            decode_unit(d, bytes, words, u0 + q, pos[q], val[q]);
            if (val[q] != 0u && (uint64_t)pos[q] >= d.len) atomicAdd(bad, 1ull);
        }
        cnt += val[q] != 0u;
    }
    // block-wide exclusive scan of cnt (wave shuffles + 4 wave totals)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
        if ((int)lane >= o) incl += t;
    }
    if (lane == 63u) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < 4u; w++) {
        if (w < wave) before += wave_tot[w];
        total += wave_tot[w];
    }
    if (!WRITE) {
        if (threadIdx.x == 0) counts[chunk] = total;
        return;
    }
    unsigned long long o = offs[chunk] + before + incl - cnt;
#pragma unroll
    for (uint32_t q = 0; q < PER_T; q++)
        if (val[q] != 0u) {
            out_idx[o] = pos[q];
            out_val[o] = val[q];
            o++;
        }
}

__global__ void indptr_kernel(const AVDesc *__restrict__ desc, uint64_t n_vecs, uint64_t n_chunks, const unsigned long long *__restrict__ offs,
                              unsigned long long total, uint64_t *__restrict__ indptr) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v > n_vecs) return;
    if (v == n_vecs) {
        indptr[v] = total;
        return;
    }
    const uint64_t c = desc[v].chunk0;
    indptr[v] = c < n_chunks ? offs[c] : total;
}

template <class T>
static void upload(DevBuf<T> &dst, const std::vector<T> &src) {
    dst.alloc(src.size() ? src.size() : 1);
    if (!src.empty()) SCANRS_HIP(hipMemcpy(dst.p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
}

} // namespace

// Expands the vectors into a device triplet (freshly allocated DevBufs). Returns the number of nonzeros.
uint64_t decode_adaptive_vectors(const scanrs_adaptive_vec *vecs, uint64_t n_vecs, uint64_t vec_len, DevBuf<uint64_t> &indptr,
                                 DevBuf<uint32_t> &indices, DevBuf<uint32_t> &values) {
    std::vector<AVDesc> desc(n_vecs);
    uint64_t n_bytes = 0, n_words = 0, n_chunks = 0;
    auto align8 = [](uint64_t x) { return (x + 7ull) & ~7ull; };
    // pass A: sizes and offsets
    for (uint64_t i = 0; i < n_vecs; i++) {
        const scanrs_adaptive_vec &v = vecs[i];
        AVDesc &d = desc[i];
        std::memset(&d, 0, sizeof(d));
        if (v.kind > 7u) fail(SCANRS_ERR_ARGUMENT, "unknown AdaptiveVec encoding");
        if (v.len != vec_len) fail(SCANRS_ERR_SHAPE, "every AdaptiveVec must have the length of the inner dimension");
        d.kind = v.kind;
        d.len = v.len;
        d.n_units = v.n_units;
        const bool sparse = v.kind >= K_S3, simple = v.kind == K_V;
        if (!sparse && !simple && v.n_units != v.len) fail(SCANRS_ERR_SHAPE, "a dense AdaptiveVec stores one field per position");
        if (v.n_units > 0xFFFFFFFFull) fail(SCANRS_ERR_SHAPE, "AdaptiveVec too long");
        uint64_t need = 0; // bytes of `data` the decode may touch
        switch (v.kind) {
        case K_D3: case K_S3: need = (v.n_units / 21u + 1u) * 8u; break;
        case K_D4: case K_S4: need = v.n_units / 2u + 1u; break;
        case K_D8: case K_S8: need = v.n_units; break;
        case K_D16: need = v.n_units * 2u; break;
        default: need = 0; break;
        }
        if (need > 0 && v.n_units > 0) {
            if (!v.data || v.data_bytes < need) fail(SCANRS_ERR_ARGUMENT, "AdaptiveVec data buffer too short for its encoding");
            d.data_off = align8(n_bytes);
            n_bytes = d.data_off + need;
        }
        if (v.n_fallback > 0 && (!v.fallback_indexes || !v.fallback_values)) fail(SCANRS_ERR_ARGUMENT, "null fallback arrays");
        if (simple && v.n_fallback != v.n_units) fail(SCANRS_ERR_SHAPE, "V: n_units must equal the number of stored entries");
        d.n_fb = v.n_fallback;
        d.fb_idx_off = n_words;
        d.fb_val_off = n_words + v.n_fallback;
        n_words += 2 * v.n_fallback;
        if (sparse) {
            const uint64_t want = (v.len + 255u) / 256u + 1u;
            if (!v.block_starts || v.n_block_starts < want) fail(SCANRS_ERR_ARGUMENT, "S*: block_starts must hold round_up(len,256)/256 + 1 entries");
            if (v.n_units > 0 && !v.index_bytes) fail(SCANRS_ERR_ARGUMENT, "S*: null index_bytes");
            d.bs_off = n_words;
            d.n_bs = v.n_block_starts;
            n_words += v.n_block_starts;
            d.ib_off = n_bytes;
            n_bytes += v.n_units;
        }
        d.chunk0 = n_chunks;
        n_chunks += (v.n_units + CH - 1) / CH;
    }
    // pass B: pack
    std::vector<uint8_t> hb(align8(n_bytes) + 8, 0);
    std::vector<uint32_t> hw(n_words, 0);
    for (uint64_t i = 0; i < n_vecs; i++) {
        const scanrs_adaptive_vec &v = vecs[i];
        const AVDesc &d = desc[i];
        uint64_t need = 0;
        switch (v.kind) {
        case K_D3: case K_S3: need = (v.n_units / 21u + 1u) * 8u; break;
        case K_D4: case K_S4: need = v.n_units / 2u + 1u; break;
        case K_D8: case K_S8: need = v.n_units; break;
        case K_D16: need = v.n_units * 2u; break;
        default: break;
        }
        if (need > 0 && v.n_units > 0) std::memcpy(hb.data() + d.data_off, v.data, need);
        if (v.n_fallback) {
            std::memcpy(hw.data() + d.fb_idx_off, v.fallback_indexes, v.n_fallback * 4);
            std::memcpy(hw.data() + d.fb_val_off, v.fallback_values, v.n_fallback * 4);
        }
        if (v.kind >= K_S3) {
            std::memcpy(hw.data() + d.bs_off, v.block_starts, v.n_block_starts * 4);
            if (v.n_units) std::memcpy(hb.data() + d.ib_off, v.index_bytes, v.n_units);
            if (v.block_starts[v.n_block_starts - 1] != v.n_units || v.block_starts[0] != 0)
                fail(SCANRS_ERR_ARGUMENT, "S*: block_starts must run from 0 to the number of stored entries");
        }
    }
    DevBuf<AVDesc> d_desc;
    DevBuf<uint8_t> d_bytes;
    DevBuf<uint32_t> d_words;
    upload(d_desc, desc);
    upload(d_bytes, hb);
    upload(d_words, hw);
    DevBuf<unsigned long long> counts, offs, bad;
    counts.alloc(n_chunks + 1);
    offs.alloc(n_chunks + 1);
    bad.alloc(1);
    SCANRS_HIP(hipMemset(bad.p, 0, 8));
    SCANRS_HIP(hipMemset(counts.p, 0, (n_chunks + 1) * 8));
    unsigned long long total = 0;
    if (n_chunks > 0) {
        hipLaunchKernelGGL((decode_kernel<false>), dim3((unsigned)n_chunks), dim3(256), 0, 0, d_desc.p, n_vecs, d_bytes.p, d_words.p, counts.p,
                           (const unsigned long long *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, bad.p);
        size_t tmp_bytes = 0;
        SCANRS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, counts.p, offs.p, 0ull, (size_t)n_chunks + 1, rocprim::plus<unsigned long long>(), 0));
        DevBuf<unsigned char> tmp;
        tmp.alloc(tmp_bytes ? tmp_bytes : 1);
        SCANRS_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, counts.p, offs.p, 0ull, (size_t)n_chunks + 1, rocprim::plus<unsigned long long>(), 0));
        SCANRS_HIP(hipMemcpy(&total, offs.p + n_chunks, 8, hipMemcpyDeviceToHost));
    }
    indptr.alloc(n_vecs + 1);
    indices.alloc(total ? total : 1);
    values.alloc(total ? total : 1);
    if (n_chunks > 0)
        hipLaunchKernelGGL((decode_kernel<true>), dim3((unsigned)n_chunks), dim3(256), 0, 0, d_desc.p, n_vecs, d_bytes.p, d_words.p, counts.p,
                           offs.p, indices.p, values.p, bad.p);
    hipLaunchKernelGGL(indptr_kernel, dim3((unsigned)((n_vecs + 256) / 256)), dim3(256), 0, 0, d_desc.p, n_vecs, n_chunks, offs.p, total, indptr.p);
    SCANRS_HIP(hipGetLastError());
    unsigned long long n_bad = 0;
    SCANRS_HIP(hipMemcpy(&n_bad, bad.p, 8, hipMemcpyDeviceToHost));
    if (n_bad) fail(SCANRS_ERR_ARGUMENT, "AdaptiveVec holds a position beyond its length");
    return total;
}

// scanrs_init(): one empty launch per translation unit makes the runtime load this file's code object now instead of inside the
// first real call
__global__ void warm_decode_kernel() {}
void warm_decode(hipStream_t s) { hipLaunchKernelGGL(warm_decode_kernel, dim3(1), dim3(64), 0, s); }

} // namespace scanrs
