// A small read-only HDF5 parser: exactly the subset of the file format that 10x Genomics matrix / analysis files use.
//
// The reference reads these files through the `hdf5` crate (libhdf5) in hdf5-io/src/matrix.rs and analysis.rs; this
// image has no libhdf5 on the library path, so the bytes are parsed here from the published format specification
// ("HDF5 File Format Specification Version 3.0"):
//   superblock v0-v3; object headers v1 and v2 (with continuation blocks); old-style groups (symbol table message ->
//   v1 B-tree of SNOD nodes + local heap) and new-style compact groups (link messages); dataspace v1/v2 (scalar, simple,
//   null); datatypes: fixed-point 1-8 bytes either endianness, IEEE f32/f64, fixed-length strings; data layout v3
//   (compact, contiguous, chunked through a v1 B-tree) and v4 (single-chunk, implicit, fixed-array and extensible-array
//   chunk indexes);
//   filter pipeline v1/v2 with deflate, shuffle and fletcher32.
// Anything else (dense link storage in fractal heaps, v2-B-tree chunk indexes, variable-length
// data, compound types, szip/lzf, external or virtual storage) is refused with a message naming the feature.
// Checked against files written by the real library: tests/golden/make_h5_fixtures.py, tests/test_h5_reader.py.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_set>
#include <vector>

namespace scanrs {
namespace h5 {

struct TypeInfo {
    enum Class { FIXED = 0, FLOAT = 1, STRING = 3 };
    int cls = FIXED;
    uint32_t size = 0;     // bytes per element
    bool is_signed = false;
    bool big_endian = false;
    int str_pad = 0;       // 0 null-terminated, 1 null-padded, 2 space-padded
};

struct DatasetInfo {
    std::vector<uint64_t> dims; // empty: scalar (one element) or null (no elements)
    std::vector<uint64_t> max_dims; // UINT64_MAX: unlimited
    bool null_space = false;
    TypeInfo type;
    // UINT64_MAX when the product of the extents does not fit 64 bits (a corrupt shape; every caller bounds it)
    uint64_t n_elements() const {
        if (null_space) return 0;
        uint64_t n = 1;
        for (uint64_t d : dims)
            if (__builtin_mul_overflow(n, d, &n)) return UINT64_MAX;
        return n;
    }
};

class File {
  public:
    explicit File(const std::string &path);
    ~File();
    File(const File &) = delete;
    File &operator=(const File &) = delete;

    typedef uint64_t Object; // address of an object header

    Object root() const { return root_; }
    // `path` is a '/'-separated list of link names relative to `from`; fails when a component is missing
    Object open(Object from, const std::string &path) const;
    bool exists(Object from, const std::string &path) const;
    // link names of a group in ascending byte order (the order libhdf5 iterates old-style groups in)
    std::vector<std::string> member_names(Object group) const;

    DatasetInfo info(Object dataset) const;
    // rows [start, end) of the first dimension (everything when the dataset is scalar); raw stored bytes, row-major
    std::vector<uint8_t> read_raw(Object dataset, uint64_t start, uint64_t end, DatasetInfo *info_out = nullptr) const;

    // numeric reads with the library's "hard conversion" semantics: integers saturate at the target's range,
    // float -> integer truncates towards zero and saturates
    template <typename T>
    std::vector<T> read(Object dataset, uint64_t start = 0, uint64_t end = UINT64_MAX, DatasetInfo *info_out = nullptr) const;
    // fixed-length strings, cut at the first NUL (null-terminated / null-padded) or stripped of trailing blanks
    std::vector<std::string> read_strings(Object dataset, uint64_t start = 0, uint64_t end = UINT64_MAX) const;

    const std::string &path() const { return path_; }

  private:
    struct Msg {
        uint16_t type;
        uint8_t flags;
        const uint8_t *data;
        uint32_t size;
    };
    struct Chunk {
        uint64_t addr;
        uint32_t size;
        uint32_t filter_mask;
        std::vector<uint64_t> offset; // rank entries, in elements
    };
    struct Filter {
        uint16_t id;
        std::vector<uint32_t> cd;
    };
    struct Layout;

    const uint8_t *at(uint64_t off, uint64_t len) const;
    uint64_t rd(const uint8_t *p, unsigned n) const;
    uint64_t rdO(const uint8_t *p) const { return rd(p, O_); }
    uint64_t rdL(const uint8_t *p) const { return rd(p, L_); }
    bool undefined(uint64_t a) const;
    std::vector<Msg> messages(Object obj) const;
    const Msg *find(const std::vector<Msg> &m, uint16_t type) const;
    void links(Object group, std::vector<std::pair<std::string, Object>> &out) const;
    void group_btree(uint64_t node, uint64_t heap_data, uint64_t heap_size, std::vector<std::pair<std::string, Object>> &out, int depth) const;
    void chunk_btree(uint64_t node, unsigned rank, std::vector<Chunk> &out, int depth) const;
    // B-tree walks: every node may be entered once (a node that is its own descendant, or a DAG, would otherwise be walked
    // exponentially often), and there cannot be more nodes than the file has room for
    void visit_node(uint64_t node) const;
    uint64_t checked_mul(uint64_t a, uint64_t b, const char *what) const;
    void parse_layout(const std::vector<Msg> &m, const DatasetInfo &di, Layout &lay) const;
    void unfilter(std::vector<uint8_t> &buf, const std::vector<Filter> &filters, uint32_t mask, uint64_t limit) const;

    std::string path_;
    int fd_ = -1;
    const uint8_t *base_ = nullptr;
    uint64_t size_ = 0;
    unsigned O_ = 8, L_ = 8;
    uint64_t base_addr_ = 0;
    Object root_ = 0;
    mutable std::unordered_set<uint64_t> visited_; // nodes of the B-tree walk in progress (cleared per walk)
};

} // namespace h5
} // namespace scanrs
