// Read-only HDF5 parser for 10x matrix / analysis files; see h5lite.hpp for the supported subset.
#include "h5lite.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <type_traits>

#include "common_err.hpp"

namespace scanrs {
GlobalOptions &global_options() { // lives with the host-only code so that the sanitizer builds of the readers link without the HIP part
    static GlobalOptions go;
    return go;
}
namespace h5 {

namespace {
constexpr int MAX_DEPTH = 64;
constexpr uint16_t MSG_DATASPACE = 0x01, MSG_LINK_INFO = 0x02, MSG_DATATYPE = 0x03, MSG_LINK = 0x06, MSG_LAYOUT = 0x08,
                   MSG_FILTERS = 0x0B, MSG_CONTINUATION = 0x10, MSG_SYMBOL_TABLE = 0x11;
} // namespace

#define H5FAIL(...) ::scanrs::fail(SCANRS_ERR_IO, __VA_ARGS__)

File::File(const std::string &path) : path_(path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) H5FAIL("unable to open file %s: %s", path.c_str(), strerror(errno));
    struct stat sb;
    if (fstat(fd_, &sb) != 0 || sb.st_size <= 0) {
        ::close(fd_);
        H5FAIL("unable to open file %s: empty or unreadable", path.c_str());
    }
    size_ = (uint64_t)sb.st_size;
    void *p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (p == MAP_FAILED) {
        ::close(fd_);
        H5FAIL("unable to map file %s: %s", path.c_str(), strerror(errno));
    }
    base_ = (const uint8_t *)p;
    try {
        static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        uint64_t sb_off = 0;
        bool found = false;
        for (uint64_t off = 0; off + 8 <= size_; off = off ? off * 2 : 512) {
            if (memcmp(base_ + off, sig, 8) == 0) {
                sb_off = off;
                found = true;
                break;
            }
        }
        if (!found) H5FAIL("%s is not an HDF5 file (no superblock signature; a git-LFS pointer?)", path.c_str());
        const uint8_t *s = at(sb_off, 16);
        const unsigned ver = s[8];
        if (ver == 0 || ver == 1) {
            s = at(sb_off, 24 + (ver == 1 ? 4 : 0) + 4 * 8 + 2 * 8 + 24);
            O_ = s[13];
            L_ = s[14];
            if ((O_ != 4 && O_ != 8) || (L_ != 4 && L_ != 8)) H5FAIL("%s: unsupported offset/length size %u/%u", path.c_str(), O_, L_);
            const uint8_t *q = s + 24 + (ver == 1 ? 4 : 0);
            base_addr_ = rdO(q);
            q += 4 * O_;               // base, free-space info, end of file, driver info
            root_ = rdO(q + O_);       // root symbol table entry: link name offset, object header address
        } else if (ver == 2 || ver == 3) {
            O_ = s[9];
            L_ = s[10];
            if ((O_ != 4 && O_ != 8) || (L_ != 4 && L_ != 8)) H5FAIL("%s: unsupported offset/length size %u/%u", path.c_str(), O_, L_);
            const uint8_t *q = at(sb_off + 12, 4 * O_);
            base_addr_ = rdO(q);
            root_ = rdO(q + 3 * O_);
        } else {
            H5FAIL("%s: unsupported superblock version %u", path.c_str(), ver);
        }
        if (undefined(base_addr_)) base_addr_ = 0;
        base_addr_ += 0; // addresses in the file are relative to the base address
        if (undefined(root_)) H5FAIL("%s: no root group", path.c_str());
    } catch (...) {
        munmap((void *)base_, size_);
        ::close(fd_);
        throw;
    }
}

File::~File() {
    if (base_) munmap((void *)base_, size_);
    if (fd_ >= 0) ::close(fd_);
}

const uint8_t *File::at(uint64_t off, uint64_t len) const {
    if (off > size_ || len > size_ - off) H5FAIL("%s: truncated or corrupt file (read of %llu bytes at %llu past the end)", path_.c_str(), (unsigned long long)len, (unsigned long long)off);
    return base_ + off;
}

uint64_t File::rd(const uint8_t *p, unsigned n) const {
    uint64_t v = 0;
    for (unsigned i = 0; i < n; i++) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

bool File::undefined(uint64_t a) const { return O_ == 8 ? a == UINT64_MAX : a == 0xFFFFFFFFull; }

uint64_t File::checked_mul(uint64_t a, uint64_t b, const char *what) const {
    uint64_t r;
    if (__builtin_mul_overflow(a, b, &r)) H5FAIL("%s: %s overflows 64 bits (corrupt shape)", path_.c_str(), what);
    return r;
}

void File::visit_node(uint64_t node) const {
    if (visited_.size() > size_ / 24 + 64) H5FAIL("%s: more B-tree nodes than the file can hold", path_.c_str());
    if (!visited_.insert(node).second) H5FAIL("%s: B-tree node %llu is reachable twice (cycle)", path_.c_str(), (unsigned long long)node);
}

// ---- object headers -----------------------------------------------------------------------------------------------
std::vector<File::Msg> File::messages(Object obj) const {
    std::vector<Msg> out;
    const uint64_t addr = base_addr_ + obj;
    const uint8_t *p = at(addr, 16);
    struct Block {
        uint64_t off, len;
    };
    std::vector<Block> blocks;
    // a continuation that points at a block already queued is a loop; a header cannot hold more messages than the file has
    // room for (4 bytes is the smallest message header)
    auto push_block = [&](uint64_t off, uint64_t len) {
        for (const Block &b : blocks)
            if (b.off == off) H5FAIL("%s: object header continuation loop", path_.c_str());
        if (blocks.size() > 4096) H5FAIL("%s: object header with too many continuation blocks", path_.c_str());
        blocks.push_back({off, len});
    };
    const uint64_t max_msgs = size_ / 4 + 16;
    if (p[0] == 1) { // version 1: 16-byte prefix, 8-byte message headers, 8-byte aligned bodies
        const unsigned n_msgs = (unsigned)rd(p + 2, 2);
        push_block(addr + 16, rd(p + 8, 4));
        for (size_t b = 0; b < blocks.size(); b++) {
            uint64_t pos = blocks[b].off;
            const uint64_t end = blocks[b].off + blocks[b].len;
            at(blocks[b].off, blocks[b].len);
            while (pos + 8 <= end && out.size() < n_msgs) {
                const uint8_t *h = at(pos, 8);
                Msg m{(uint16_t)rd(h, 2), h[4], nullptr, (uint32_t)rd(h + 2, 2)};
                if (pos + 8 + m.size > end) H5FAIL("%s: object header message runs past its block", path_.c_str());
                m.data = at(pos + 8, m.size);
                pos += 8 + m.size;
                if (m.type == MSG_CONTINUATION) {
                    if (m.size < O_ + L_) H5FAIL("%s: short continuation message", path_.c_str());
                    push_block(base_addr_ + rdO(m.data), rdL(m.data + O_));
                }
                out.push_back(m);
            }
        }
    } else if (memcmp(p, "OHDR", 4) == 0) { // version 2
        if (p[4] != 2) H5FAIL("%s: unsupported object header version %u", path_.c_str(), p[4]);
        const uint8_t flags = p[5];
        uint64_t pos = addr + 6;
        if (flags & 0x20) pos += 16; // access, modification, change, birth times
        if (flags & 0x10) pos += 4;  // max compact / min dense attributes
        const unsigned nsz = 1u << (flags & 3);
        const uint64_t chunk0 = rd(at(pos, nsz), nsz);
        pos += nsz;
        const unsigned hdr = 4 + ((flags & 0x04) ? 2 : 0);
        push_block(pos, chunk0);
        for (size_t b = 0; b < blocks.size(); b++) {
            uint64_t q = blocks[b].off;
            const uint64_t end = blocks[b].off + blocks[b].len;
            at(blocks[b].off, blocks[b].len);
            while (q + hdr <= end) {
                const uint8_t *h = at(q, hdr);
                Msg m{h[0], h[3], nullptr, (uint32_t)rd(h + 1, 2)};
                if (q + hdr + m.size > end) break; // gap at the end of the chunk
                m.data = at(q + hdr, m.size);
                q += hdr + m.size;
                if (m.type == MSG_CONTINUATION) {
                    if (m.size < O_ + L_) H5FAIL("%s: short continuation message", path_.c_str());
                    const uint64_t coff = base_addr_ + rdO(m.data), clen = rdL(m.data + O_);
                    if (clen < 8 || memcmp(at(coff, 4), "OCHK", 4) != 0) H5FAIL("%s: bad object header continuation block", path_.c_str());
                    push_block(coff + 4, clen - 8); // signature in front, checksum behind
                }
                if (out.size() >= max_msgs) H5FAIL("%s: object header with more messages than the file can hold", path_.c_str());
                out.push_back(m);
            }
        }
    } else {
        H5FAIL("%s: object header at %llu has an unknown format", path_.c_str(), (unsigned long long)addr);
    }
    return out;
}

const File::Msg *File::find(const std::vector<Msg> &m, uint16_t type) const {
    for (const Msg &x : m)
        if (x.type == type) {
            if (x.flags & 0x02) H5FAIL("%s: shared object header messages are not supported", path_.c_str());
            return &x;
        }
    return nullptr;
}

// ---- groups -------------------------------------------------------------------------------------------------------
void File::group_btree(uint64_t node, uint64_t heap_data, uint64_t heap_size, std::vector<std::pair<std::string, Object>> &out, int depth) const {
    if (depth > MAX_DEPTH) H5FAIL("%s: group B-tree too deep", path_.c_str());
    visit_node(node); // a node reachable twice is a cycle (or a DAG that would be walked exponentially often)
    const uint8_t *p = at(base_addr_ + node, 8 + 2 * O_);
    if (memcmp(p, "TREE", 4) == 0) {
        if (p[4] != 0) H5FAIL("%s: group B-tree node has type %u", path_.c_str(), p[4]);
        const unsigned level = p[5], n = (unsigned)rd(p + 6, 2);
        const uint8_t *q = at(base_addr_ + node + 8 + 2 * O_, (uint64_t)n * (L_ + O_) + L_);
        for (unsigned i = 0; i < n; i++) {
            const uint64_t child = rdO(q + L_ + (uint64_t)i * (L_ + O_));
            (void)level;
            group_btree(child, heap_data, heap_size, out, depth + 1);
        }
    } else if (memcmp(p, "SNOD", 4) == 0) {
        const unsigned n = (unsigned)rd(p + 6, 2);
        const unsigned esz = 2 * O_ + 24;
        const uint8_t *q = at(base_addr_ + node + 8, (uint64_t)n * esz);
        for (unsigned i = 0; i < n; i++) {
            const uint8_t *e = q + (uint64_t)i * esz;
            const uint64_t name_off = rdO(e), obj = rdO(e + O_);
            const uint32_t cache = (uint32_t)rd(e + 2 * O_, 4);
            if (name_off >= heap_size) H5FAIL("%s: link name outside the group's heap", path_.c_str());
            const char *s = (const char *)at(heap_data + name_off, 1);
            const size_t maxlen = (size_t)(heap_size - name_off);
            const size_t len = strnlen(s, maxlen);
            at(heap_data + name_off, len);
            if (cache == 2) continue; // symbolic link: no object behind it
            out.emplace_back(std::string(s, len), obj);
        }
    } else {
        H5FAIL("%s: bad group B-tree node signature", path_.c_str());
    }
}

void File::links(Object group, std::vector<std::pair<std::string, Object>> &out) const {
    const std::vector<Msg> m = messages(group);
    if (const Msg *st = find(m, MSG_SYMBOL_TABLE)) {
        if (st->size < 2 * O_) H5FAIL("%s: short symbol table message", path_.c_str());
        const uint64_t btree = rdO(st->data), heap = rdO(st->data + O_);
        const uint8_t *h = at(base_addr_ + heap, 8 + 2 * L_ + O_);
        if (memcmp(h, "HEAP", 4) != 0) H5FAIL("%s: bad local heap signature", path_.c_str());
        const uint64_t hsize = rdL(h + 8), hdata = base_addr_ + rdO(h + 8 + 2 * L_);
        at(hdata, hsize);
        visited_.clear();
        group_btree(btree, hdata, hsize, out, 0);
        return;
    }
    bool any = false;
    for (const Msg &x : m) {
        if (x.type == MSG_LINK_INFO) {
            any = true;
            // version, flags, [max creation index 8], fractal heap address, name index address
            if (x.size < 2 + 2 * O_) H5FAIL("%s: short link info message", path_.c_str());
            const unsigned skip = 2 + ((x.data[1] & 1) ? 8 : 0);
            if (!undefined(rdO(x.data + skip))) H5FAIL("%s: group with dense link storage (fractal heap) is not supported by this reader", path_.c_str());
        }
        if (x.type != MSG_LINK) continue;
        any = true;
        const uint8_t *d = x.data;
        const uint8_t *e = d + x.size;
        if (x.size < 4 || d[0] != 1) H5FAIL("%s: unsupported link message version", path_.c_str());
        const uint8_t fl = d[1];
        d += 2;
        unsigned ltype = 0;
        if (fl & 0x08) ltype = *d++;
        if (fl & 0x04) d += 8;
        if (fl & 0x10) d += 1;
        const unsigned lsz = 1u << (fl & 3);
        if (d + lsz > e) H5FAIL("%s: short link message", path_.c_str());
        const uint64_t nlen = rd(d, lsz);
        d += lsz;
        if (nlen > (uint64_t)(e - d)) H5FAIL("%s: short link message", path_.c_str());
        std::string name((const char *)d, (size_t)nlen);
        d += nlen;
        if (ltype != 0) continue; // soft / external link
        if (d + O_ > e) H5FAIL("%s: short link message", path_.c_str());
        out.emplace_back(std::move(name), rdO(d));
    }
    if (!any) H5FAIL("%s: object is not a group", path_.c_str());
}

File::Object File::open(Object from, const std::string &path) const {
    Object cur = from;
    size_t pos = 0;
    while (pos < path.size()) {
        size_t next = path.find('/', pos);
        if (next == std::string::npos) next = path.size();
        const std::string name = path.substr(pos, next - pos);
        pos = next + 1;
        if (name.empty()) continue;
        std::vector<std::pair<std::string, Object>> ls;
        links(cur, ls);
        bool found = false;
        for (auto &kv : ls)
            if (kv.first == name) {
                cur = kv.second;
                found = true;
                break;
            }
        if (!found) H5FAIL("unable to open '%s' in %s: object '%s' doesn't exist", path.c_str(), path_.c_str(), name.c_str());
    }
    return cur;
}

bool File::exists(Object from, const std::string &path) const {
    try {
        (void)open(from, path);
        return true;
    } catch (const Failure &) {
        return false;
    }
}

std::vector<std::string> File::member_names(Object group) const {
    std::vector<std::pair<std::string, Object>> ls;
    links(group, ls);
    std::vector<std::string> out;
    for (auto &kv : ls) out.push_back(kv.first);
    std::sort(out.begin(), out.end());
    return out;
}

// ---- datasets -----------------------------------------------------------------------------------------------------
struct File::Layout {
    int cls = -1; // 0 compact, 1 contiguous, 2 chunked
    const uint8_t *compact = nullptr;
    uint64_t compact_size = 0;
    uint64_t addr = 0, size = 0;
    std::vector<uint64_t> chunk_dims; // rank entries
    std::vector<Chunk> chunks;
    std::vector<Filter> filters;
};

DatasetInfo File::info(Object dataset) const {
    const std::vector<Msg> m = messages(dataset);
    DatasetInfo di;
    const Msg *sp = find(m, MSG_DATASPACE), *ty = find(m, MSG_DATATYPE);
    if (!sp || !ty) H5FAIL("%s: object is not a dataset", path_.c_str());
    {
        const uint8_t *d = sp->data;
        if (sp->size < 4) H5FAIL("%s: short dataspace message", path_.c_str());
        const unsigned ver = d[0], rank = d[1], flags = d[2];
        unsigned off;
        if (ver == 1)
            off = 8;
        else if (ver == 2) {
            off = 4;
            if (d[3] == 2) di.null_space = true;
        } else
            H5FAIL("%s: unsupported dataspace version %u", path_.c_str(), ver);
        if (sp->size < off + (uint64_t)rank * L_) H5FAIL("%s: short dataspace message", path_.c_str());
        for (unsigned i = 0; i < rank; i++) di.dims.push_back(rdL(d + off + (uint64_t)i * L_));
        if (flags & 1) { // maximum extents follow (all ones: unlimited)
            if (sp->size < off + 2ull * rank * L_) H5FAIL("%s: short dataspace message", path_.c_str());
            for (unsigned i = 0; i < rank; i++) {
                const uint64_t v = rdL(d + off + (uint64_t)(rank + i) * L_);
                di.max_dims.push_back((L_ == 4 && v == 0xFFFFFFFFull) ? UINT64_MAX : v);
            }
        } else {
            di.max_dims = di.dims;
        }
    }
    {
        const uint8_t *d = ty->data;
        if (ty->size < 8) H5FAIL("%s: short datatype message", path_.c_str());
        const unsigned cls = d[0] & 0x0F;
        di.type.size = (uint32_t)rd(d + 4, 4);
        if (cls == 0) {
            di.type.cls = TypeInfo::FIXED;
            di.type.big_endian = d[1] & 1;
            di.type.is_signed = (d[1] & 8) != 0;
            if (di.type.size != 1 && di.type.size != 2 && di.type.size != 4 && di.type.size != 8) H5FAIL("%s: %u-byte integers are not supported", path_.c_str(), di.type.size);
        } else if (cls == 1) {
            di.type.cls = TypeInfo::FLOAT;
            di.type.big_endian = d[1] & 1;
            if (di.type.size != 4 && di.type.size != 8) H5FAIL("%s: %u-byte floats are not supported", path_.c_str(), di.type.size);
        } else if (cls == 3) {
            di.type.cls = TypeInfo::STRING;
            di.type.str_pad = d[1] & 0x0F;
        } else {
            static const char *names[] = {"fixed-point", "float", "time", "string", "bitfield", "opaque", "compound", "reference", "enum", "variable-length", "array"};
            H5FAIL("%s: datatype class '%s' is not supported by this reader", path_.c_str(), cls < 11 ? names[cls] : "unknown");
        }
    }
    return di;
}

void File::chunk_btree(uint64_t node, unsigned rank, std::vector<Chunk> &out, int depth) const {
    if (depth > MAX_DEPTH) H5FAIL("%s: chunk B-tree too deep", path_.c_str());
    visit_node(node);
    if (out.size() > size_ / 8 + 64) H5FAIL("%s: more chunks than the file can hold", path_.c_str());
    const uint8_t *p = at(base_addr_ + node, 8 + 2 * O_);
    if (memcmp(p, "TREE", 4) != 0 || p[4] != 1) H5FAIL("%s: bad chunk B-tree node", path_.c_str());
    const unsigned level = p[5], n = (unsigned)rd(p + 6, 2);
    const uint64_t key = 8 + 8ull * (rank + 1);
    const uint8_t *q = at(base_addr_ + node + 8 + 2 * O_, (uint64_t)n * (key + O_) + key);
    for (unsigned i = 0; i < n; i++) {
        const uint8_t *k = q + (uint64_t)i * (key + O_);
        const uint64_t child = rdO(k + key);
        if (level > 0) {
            chunk_btree(child, rank, out, depth + 1);
        } else {
            Chunk c;
            c.size = (uint32_t)rd(k, 4);
            c.filter_mask = (uint32_t)rd(k + 4, 4);
            c.addr = child;
            for (unsigned d = 0; d < rank; d++) c.offset.push_back(rd(k + 8 + 8ull * d, 8));
            out.push_back(std::move(c));
        }
    }
}

void File::parse_layout(const std::vector<Msg> &m, const DatasetInfo &di, Layout &lay) const {
    const Msg *lm = find(m, MSG_LAYOUT);
    if (!lm || lm->size < 2) H5FAIL("%s: dataset without a data layout", path_.c_str());
    const uint8_t *d = lm->data, *e = lm->data + lm->size;
    const unsigned ver = d[0];
    const unsigned rank = (unsigned)di.dims.size();
    const uint64_t elem = di.type.size;
    if (const Msg *fm = find(m, MSG_FILTERS)) {
        const uint8_t *f = fm->data, *fe = fm->data + fm->size;
        if (fm->size < 2) H5FAIL("%s: short filter pipeline message", path_.c_str());
        const unsigned fver = f[0], nf = f[1];
        f += fver == 1 ? 8 : 2;
        if (fver != 1 && fver != 2) H5FAIL("%s: filter pipeline version %u is not supported", path_.c_str(), fver);
        if (f > fe) H5FAIL("%s: short filter pipeline message", path_.c_str());
        auto need = [&](uint64_t n) { // every field is checked against the end of the message before it is read
            if ((uint64_t)(fe - f) < n) H5FAIL("%s: short filter pipeline message", path_.c_str());
        };
        for (unsigned i = 0; i < nf; i++) {
            Filter fl;
            need(2);
            fl.id = (uint16_t)rd(f, 2);
            f += 2;
            unsigned name_len = 0;
            if (fver == 1 || fl.id >= 256) {
                need(2);
                name_len = (unsigned)rd(f, 2);
                f += 2;
            }
            need(4);
            f += 2; // flags
            const unsigned ncd = (unsigned)rd(f, 2);
            f += 2;
            if (fver == 1) name_len = (name_len + 7) & ~7u;
            need(name_len);
            f += name_len;
            need(4ull * ncd + ((fver == 1 && (ncd & 1)) ? 4 : 0));
            for (unsigned c = 0; c < ncd; c++) fl.cd.push_back((uint32_t)rd(f + 4ull * c, 4));
            f += 4ull * ncd;
            if (fver == 1 && (ncd & 1)) f += 4;
            if (fl.id != 1 && fl.id != 2 && fl.id != 3) H5FAIL("%s: filter %u (%s) is not supported by this reader", path_.c_str(), fl.id, fl.id == 4 ? "szip" : fl.id == 32000 ? "lzf" : "unknown");
            lay.filters.push_back(std::move(fl));
        }
    }
    if (ver != 3 && ver != 4) H5FAIL("%s: data layout version %u is not supported", path_.c_str(), ver);
    lay.cls = d[1];
    if (lay.cls == 0) {
        if (lm->size < 4) H5FAIL("%s: short layout message", path_.c_str());
        lay.compact_size = rd(d + 2, 2);
        if (d + 4 + lay.compact_size > e) H5FAIL("%s: compact data runs past its message", path_.c_str());
        lay.compact = d + 4;
    } else if (lay.cls == 1) {
        if (lm->size < 2 + O_ + L_) H5FAIL("%s: short layout message", path_.c_str());
        lay.addr = rdO(d + 2);
        lay.size = rdL(d + 2 + O_);
    } else if (lay.cls == 2 && ver == 3) {
        const unsigned nd = d[2];
        if (nd != rank + 1) H5FAIL("%s: chunk dimensionality %u does not match the dataspace rank %u", path_.c_str(), nd, rank);
        if (lm->size < 3 + O_ + 4ull * nd) H5FAIL("%s: short layout message", path_.c_str());
        const uint64_t btree = rdO(d + 3);
        for (unsigned i = 0; i < rank; i++) lay.chunk_dims.push_back(rd(d + 3 + O_ + 4ull * i, 4));
        if (rd(d + 3 + O_ + 4ull * rank, 4) != elem) H5FAIL("%s: chunk element size does not match the datatype", path_.c_str());
        if (!undefined(btree)) {
            visited_.clear();
            chunk_btree(btree, rank, lay.chunks, 0);
        }
    } else if (lay.cls == 2 && ver == 4) {
        const uint8_t cflags = d[2];
        const unsigned nd = d[3], enc = d[4];
        if (nd != rank + 1 || enc == 0 || enc > 8) H5FAIL("%s: bad version-4 chunk layout", path_.c_str());
        const uint8_t *q = d + 5;
        if (q + (uint64_t)nd * enc + 1 > e) H5FAIL("%s: short layout message", path_.c_str());
        for (unsigned i = 0; i < rank; i++) lay.chunk_dims.push_back(rd(q + (uint64_t)i * enc, enc));
        q += (uint64_t)nd * enc;
        const unsigned index_type = *q++;
        if ((cflags & 0x01) && !lay.filters.empty()) H5FAIL("%s: unfiltered partial edge chunks are not supported", path_.c_str());
        uint64_t chunk_bytes = elem;
        for (uint64_t c : lay.chunk_dims) chunk_bytes = checked_mul(chunk_bytes, c, "chunk size");
        if (chunk_bytes > 0xFFFFFFFFull) H5FAIL("%s: chunks of 4 GiB or more are not supported", path_.c_str());
        std::vector<uint64_t> grid(rank);
        uint64_t n_chunks = 1;
        for (unsigned i = 0; i < rank; i++) {
            if (lay.chunk_dims[i] == 0) H5FAIL("%s: zero chunk dimension", path_.c_str());
            grid[i] = di.dims[i] / lay.chunk_dims[i] + (di.dims[i] % lay.chunk_dims[i] ? 1 : 0);
            n_chunks = checked_mul(n_chunks, grid[i], "chunk grid");
        }
        // every chunk of the grid costs at least one index entry (>= 8 bytes) in the file: a grid the file has no room for is a
        // corrupt shape, not a dataset (the walks below are linear in n_chunks)
        if (index_type != 1 && n_chunks > size_ / 8 + 64) H5FAIL("%s: chunk grid of %llu chunks cannot be indexed by a %llu-byte file", path_.c_str(), (unsigned long long)n_chunks, (unsigned long long)size_);
        auto offset_of = [&](uint64_t linear) {
            std::vector<uint64_t> off(rank);
            for (int i = (int)rank - 1; i >= 0; i--) {
                off[i] = (linear % grid[i]) * lay.chunk_dims[i];
                linear /= grid[i];
            }
            return off;
        };
        if (index_type == 1) { // single chunk
            Chunk c;
            c.size = (uint32_t)chunk_bytes;
            c.filter_mask = 0;
            if (cflags & 0x02) {
                if (q + L_ + 4 > e) H5FAIL("%s: short layout message", path_.c_str());
                c.size = (uint32_t)rdL(q);
                c.filter_mask = (uint32_t)rd(q + L_, 4);
                q += L_ + 4;
            } else if (!lay.filters.empty()) {
                c.filter_mask = 0;
            }
            if (q + O_ > e) H5FAIL("%s: short layout message", path_.c_str());
            c.addr = rdO(q);
            c.offset.assign(rank, 0);
            if (!undefined(c.addr)) lay.chunks.push_back(std::move(c));
        } else if (index_type == 2) { // implicit: all chunks allocated back to back, never filtered
            if (q + O_ > e) H5FAIL("%s: short layout message", path_.c_str());
            const uint64_t a0 = rdO(q);
            if (!undefined(a0))
                for (uint64_t i = 0; i < n_chunks; i++) {
                    Chunk c;
                    c.addr = a0 + i * chunk_bytes;
                    c.size = (uint32_t)chunk_bytes;
                    c.filter_mask = 0;
                    c.offset = offset_of(i);
                    lay.chunks.push_back(std::move(c));
                }
        } else if (index_type == 3) { // fixed array
            if (q + 1 + O_ > e) H5FAIL("%s: short layout message", path_.c_str());
            q += 1; // page bits (repeated in the header)
            const uint64_t hdr = rdO(q);
            if (!undefined(hdr)) {
                const uint8_t *h = at(base_addr_ + hdr, 8 + L_ + O_ + 4);
                if (memcmp(h, "FAHD", 4) != 0) H5FAIL("%s: bad fixed array header", path_.c_str());
                const unsigned client = h[5], esz = h[6], page_bits = h[7];
                if (page_bits > 31) H5FAIL("%s: bad fixed array page size", path_.c_str());
                const uint64_t nel = rdL(h + 8), dblk = rdO(h + 8 + L_);
                if (nel < n_chunks) H5FAIL("%s: fixed array smaller than the chunk grid", path_.c_str());
                if ((client == 0 && esz != O_) || (client == 1 && (esz < O_ + 5 || esz > O_ + 12))) H5FAIL("%s: bad fixed array entry size", path_.c_str());
                if (!undefined(dblk)) {
                    const uint8_t *b = at(base_addr_ + dblk, 6 + O_);
                    if (memcmp(b, "FADB", 4) != 0) H5FAIL("%s: bad fixed array data block", path_.c_str());
                    uint64_t pos = base_addr_ + dblk + 6 + O_;
                    const uint64_t page_n = 1ull << page_bits;
                    const bool paged = nel > page_n;
                    uint64_t n_pages = 0;
                    std::vector<uint8_t> bitmap;
                    if (paged) {
                        n_pages = (nel + page_n - 1) / page_n;
                        const uint64_t bm = (n_pages + 7) / 8;
                        const uint8_t *bp = at(pos, bm + 4);
                        bitmap.assign(bp, bp + bm);
                        pos += bm + 4; // bitmap, then the data block's checksum
                    }
                    for (uint64_t i = 0; i < n_chunks; i++) {
                        uint64_t epos;
                        if (paged) {
                            const uint64_t pg = i / page_n;
                            if (!(bitmap[pg / 8] & (0x80u >> (pg % 8)))) continue; // page never written
                            epos = pos + pg * (page_n * esz + 4) + (i % page_n) * esz;
                        } else {
                            epos = pos + i * esz;
                        }
                        const uint8_t *el = at(epos, esz);
                        Chunk c;
                        c.addr = rdO(el);
                        if (undefined(c.addr)) continue;
                        c.size = (uint32_t)chunk_bytes;
                        c.filter_mask = 0;
                        if (client == 1) {
                            const unsigned csz = esz - O_ - 4;
                            c.size = (uint32_t)rd(el + O_, csz);
                            c.filter_mask = (uint32_t)rd(el + O_ + csz, 4);
                        }
                        c.offset = offset_of(i);
                        lay.chunks.push_back(std::move(c));
                    }
                }
            }
        } else if (index_type == 4) { // extensible array: exactly one unlimited dimension
            if (q + 5 + O_ > e) H5FAIL("%s: short layout message", path_.c_str());
            const unsigned max_bits = q[0], idx_elmts = q[1], min_ptrs = q[2], min_elmts = q[3], page_bits = q[4];
            const uint64_t hdr = rdO(q + 5);
            if (!undefined(hdr)) {
                auto ilog2 = [&](uint64_t v) {
                    unsigned r = 0;
                    while (v > 1) {
                        v >>= 1;
                        r++;
                    }
                    return r;
                };
                if (min_elmts == 0 || (min_elmts & (min_elmts - 1)) || min_ptrs < 2 || (min_ptrs & (min_ptrs - 1)) || max_bits == 0 || max_bits > 64 ||
                    max_bits < ilog2(min_elmts) || page_bits > 31)
                    H5FAIL("%s: bad extensible array parameters", path_.c_str());
                const uint8_t *h = at(base_addr_ + hdr, 12 + 6 * L_ + O_ + 4);
                if (memcmp(h, "EAHD", 4) != 0) H5FAIL("%s: bad extensible array header", path_.c_str());
                const unsigned client = h[5], esz = h[6];
                if (h[7] != max_bits || h[8] != idx_elmts || h[9] != min_elmts || h[10] != min_ptrs || h[11] != page_bits)
                    H5FAIL("%s: extensible array header disagrees with the layout message", path_.c_str());
                if ((client == 0 && esz != O_) || (client == 1 && (esz < O_ + 5 || esz > O_ + 12))) H5FAIL("%s: bad extensible array entry size", path_.c_str());
                const uint64_t iblk = rdO(h + 12 + 6 * L_);
                const unsigned off_sz = (max_bits + 7) / 8;
                const unsigned nsblks = 1 + (max_bits - ilog2(min_elmts));
                const unsigned ib_nsblks = 2 * ilog2(min_ptrs);
                const uint64_t ib_ndblk = 2ull * (min_ptrs - 1);
                const unsigned ib_nsblk_addrs = nsblks > ib_nsblks ? nsblks - ib_nsblks : 0;
                const uint64_t page_n = 1ull << page_bits;
                // which dimension is unlimited, and the linear index of a chunk: the unlimited dimension counts slowest
                // (H5VM_swizzle_coords), the others by the chunk counts of their MAXIMUM extents
                int unlim = -1;
                for (unsigned i = 0; i < rank; i++)
                    if (i < di.max_dims.size() && di.max_dims[i] == UINT64_MAX) {
                        if (unlim >= 0) H5FAIL("%s: extensible array with two unlimited dimensions", path_.c_str());
                        unlim = (int)i;
                    }
                if (unlim < 0) H5FAIL("%s: extensible array chunk index without an unlimited dimension", path_.c_str());
                std::vector<unsigned> order;
                order.push_back((unsigned)unlim);
                for (unsigned i = 0; i < rank; i++)
                    if ((int)i != unlim) order.push_back(i);
                std::vector<uint64_t> max_chunks(rank);
                for (unsigned i = 0; i < rank; i++) {
                    const uint64_t md = (int)i == unlim ? di.dims[i] : di.max_dims[i];
                    max_chunks[i] = (md + lay.chunk_dims[i] - 1) / lay.chunk_dims[i];
                }
                auto element = [&](const uint8_t *el, uint64_t linear, const std::vector<uint64_t> &off) {
                    (void)linear;
                    Chunk c;
                    c.addr = rdO(el);
                    if (undefined(c.addr)) return;
                    c.size = (uint32_t)chunk_bytes;
                    c.filter_mask = 0;
                    if (client == 1) {
                        const unsigned csz = esz - O_ - 4;
                        c.size = (uint32_t)rd(el + O_, csz);
                        c.filter_mask = (uint32_t)rd(el + O_ + csz, 4);
                    }
                    c.offset = off;
                    lay.chunks.push_back(std::move(c));
                };
                if (!undefined(iblk) && n_chunks > 0) {
                    const uint8_t *ib = at(base_addr_ + iblk, 6 + O_);
                    if (memcmp(ib, "EAIB", 4) != 0) H5FAIL("%s: bad extensible array index block", path_.c_str());
                    const uint64_t ib_elems = base_addr_ + iblk + 6 + O_;
                    const uint64_t ib_dblks = ib_elems + (uint64_t)idx_elmts * esz;
                    const uint64_t ib_sblks = ib_dblks + ib_ndblk * O_;
                    at(ib_elems, (uint64_t)idx_elmts * esz + ib_ndblk * O_ + (uint64_t)ib_nsblk_addrs * O_ + 4);
                    // super block geometry
                    std::vector<uint64_t> sb_ndblks(nsblks), sb_dnel(nsblks), sb_start_idx(nsblks), sb_start_dblk(nsblks);
                    {
                        uint64_t si = 0, sd = 0;
                        for (unsigned u = 0; u < nsblks; u++) {
                            sb_ndblks[u] = 1ull << (u / 2);
                            sb_dnel[u] = (uint64_t)min_elmts << ((u + 1) / 2);
                            sb_start_idx[u] = si;
                            sb_start_dblk[u] = sd;
                            if (u < 62) {
                                si += sb_ndblks[u] * sb_dnel[u];
                                sd += sb_ndblks[u];
                            }
                        }
                    }
                    // walk the chunk grid; the per-chunk lookup is the library's H5EA__lookup_elmt
                    std::vector<uint64_t> scaled(rank);
                    for (uint64_t g = 0; g < n_chunks; g++) {
                        const std::vector<uint64_t> off = offset_of(g);
                        for (unsigned i = 0; i < rank; i++) scaled[i] = off[i] / lay.chunk_dims[i];
                        uint64_t idx = 0;
                        for (unsigned i = 0; i < rank; i++) idx = (i == 0 ? 0 : idx * max_chunks[order[i]]) + scaled[order[i]];
                        if (idx < idx_elmts) {
                            element(at(ib_elems + idx * esz, esz), idx, off);
                            continue;
                        }
                        uint64_t ei = idx - idx_elmts;
                        const unsigned sb = ilog2(ei / min_elmts + 1);
                        if (sb >= nsblks) H5FAIL("%s: chunk index past the extensible array's range", path_.c_str());
                        ei -= sb_start_idx[sb];
                        const uint64_t dnel = sb_dnel[sb];
                        uint64_t dblk_addr;
                        const uint64_t dblk_in_sb = ei / dnel;
                        const uint64_t in_dblk = ei % dnel;
                        const bool paged = dnel > page_n;
                        if (sb < ib_nsblks) {
                            const uint64_t di_ = sb_start_dblk[sb] + dblk_in_sb;
                            if (di_ >= ib_ndblk) H5FAIL("%s: bad extensible array geometry", path_.c_str());
                            dblk_addr = rdO(at(ib_dblks + di_ * O_, O_));
                        } else {
                            const uint64_t sa = rdO(at(ib_sblks + (uint64_t)(sb - ib_nsblks) * O_, O_));
                            if (undefined(sa)) continue;
                            const uint8_t *sp = at(base_addr_ + sa, 6 + O_ + off_sz);
                            if (memcmp(sp, "EASB", 4) != 0) H5FAIL("%s: bad extensible array super block", path_.c_str());
                            uint64_t pos = base_addr_ + sa + 6 + O_ + off_sz;
                            if (paged) {
                                const uint64_t npages = dnel / page_n, bm = (npages + 7) / 8;
                                // one bit per page, indexed dblk * npages + page across the whole region (H5EA__lookup_elmt),
                                // although the region is sized ndblks * ceil(npages / 8) bytes
                                const uint8_t *bits = at(pos, sb_ndblks[sb] * bm);
                                const uint64_t pg = dblk_in_sb * npages + in_dblk / page_n;
                                if (!(bits[pg / 8] & (0x80u >> (pg % 8)))) continue; // page never written
                                pos += sb_ndblks[sb] * bm;
                            }
                            dblk_addr = rdO(at(pos + dblk_in_sb * O_, O_));
                        }
                        if (undefined(dblk_addr)) continue;
                        const uint8_t *dp = at(base_addr_ + dblk_addr, 6 + O_ + off_sz);
                        if (memcmp(dp, "EADB", 4) != 0) H5FAIL("%s: bad extensible array data block", path_.c_str());
                        uint64_t epos = base_addr_ + dblk_addr + 6 + O_ + off_sz;
                        if (paged) {
                            epos += 4; // the data block's own checksum precedes its pages
                            epos += (in_dblk / page_n) * (page_n * esz + 4) + (in_dblk % page_n) * esz;
                        } else {
                            epos += in_dblk * esz;
                        }
                        element(at(epos, esz), idx, off);
                    }
                }
            }
        } else {
            H5FAIL("%s: chunk index type %u (%s) is not supported by this reader", path_.c_str(), index_type,
                   index_type == 5 ? "version-2 B-tree" : "unknown");
        }
    } else {
        H5FAIL("%s: data layout class %d is not supported", path_.c_str(), lay.cls);
    }
}

void File::unfilter(std::vector<uint8_t> &buf, const std::vector<Filter> &filters, uint32_t mask, uint64_t limit) const {
    for (int i = (int)filters.size() - 1; i >= 0; i--) {
        if (mask & (1u << i)) continue;
        const Filter &f = filters[i];
        if (f.id == 1) { // deflate
            // the declared chunk size (+ checksum) in one allocation — but never more than deflate can produce from this
            // many input bytes (1032:1), so a corrupt chunk shape cannot ask for gigabytes
            limit = std::min<uint64_t>(limit, (uint64_t)buf.size() * 1100 + 4096);
            std::vector<uint8_t> out((size_t)limit);
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit(&zs) != Z_OK) H5FAIL("%s: zlib initialisation failed", path_.c_str());
            zs.next_in = buf.data();
            zs.avail_in = (uInt)buf.size();
            size_t produced = 0;
            for (;;) {
                zs.next_out = out.data() + produced;
                zs.avail_out = (uInt)(out.size() - produced);
                const int rc = inflate(&zs, Z_NO_FLUSH);
                produced = out.size() - zs.avail_out;
                if (rc == Z_STREAM_END) break;
                if (rc != Z_OK && rc != Z_BUF_ERROR) {
                    inflateEnd(&zs);
                    H5FAIL("%s: corrupt deflate stream in a chunk", path_.c_str());
                }
                if (zs.avail_out == 0) {
                    if (out.size() >= limit) { // a chunk never holds more than its declared shape (+ checksum)
                        inflateEnd(&zs);
                        H5FAIL("%s: chunk inflates past its declared size", path_.c_str());
                    }
                    out.resize((size_t)std::min<uint64_t>(out.size() * 2, limit));
                } else if (zs.avail_in == 0) {
                    inflateEnd(&zs);
                    H5FAIL("%s: truncated deflate stream in a chunk", path_.c_str());
                }
            }
            inflateEnd(&zs);
            out.resize(produced);
            buf.swap(out);
        } else if (f.id == 2) { // shuffle: byte b of element i was stored at b * n + i
            const size_t es = f.cd.empty() ? 0 : f.cd[0];
            if (es > 1 && buf.size() >= es) {
                const size_t n = buf.size() / es;
                std::vector<uint8_t> out(buf.size());
                if (es == 4) {
                    const uint8_t *b0 = buf.data(), *b1 = b0 + n, *b2 = b1 + n, *b3 = b2 + n;
                    uint8_t *o = out.data();
                    for (size_t i = 0; i < n; i++, o += 4) {
                        o[0] = b0[i];
                        o[1] = b1[i];
                        o[2] = b2[i];
                        o[3] = b3[i];
                    }
                } else if (es == 8) {
                    const uint8_t *bb[8];
                    for (int b = 0; b < 8; b++) bb[b] = buf.data() + (size_t)b * n;
                    uint8_t *o = out.data();
                    for (size_t i = 0; i < n; i++, o += 8)
                        for (int b = 0; b < 8; b++) o[b] = bb[b][i];
                } else {
                    for (size_t b = 0; b < es; b++)
                        for (size_t i = 0; i < n; i++) out[i * es + b] = buf[b * n + i];
                }
                for (size_t r = n * es; r < buf.size(); r++) out[r] = buf[r];
                buf.swap(out);
            }
        } else if (f.id == 3) { // fletcher32 checksum behind the data
            if (buf.size() < 4) H5FAIL("%s: chunk shorter than its checksum", path_.c_str());
            buf.resize(buf.size() - 4);
        }
    }
}

std::vector<uint8_t> File::read_raw(Object dataset, uint64_t start, uint64_t end, DatasetInfo *info_out) const {
    const std::vector<Msg> m = messages(dataset);
    const DatasetInfo di = info(dataset);
    if (info_out) *info_out = di;
    const unsigned rank = (unsigned)di.dims.size();
    const uint64_t elem = di.type.size;
    if (di.null_space || elem == 0) return {};
    const uint64_t d0 = rank ? di.dims[0] : 1;
    if (end > d0) end = d0;
    if (start > end) H5FAIL("%s: slice start %llu is past the end %llu of the dataset", path_.c_str(), (unsigned long long)start, (unsigned long long)end);
    uint64_t row_elems = 1; // elements per index of the first dimension
    for (unsigned i = 1; i < rank; i++) row_elems = checked_mul(row_elems, di.dims[i], "dataset shape");
    const uint64_t row_bytes = checked_mul(row_elems, elem, "dataset shape");
    if (di.n_elements() == UINT64_MAX) H5FAIL("%s: dataset shape overflows 64 bits (corrupt shape)", path_.c_str());
    if (row_bytes && (end - start) > (UINT64_MAX / 2) / row_bytes) H5FAIL("%s: dataset too large", path_.c_str());
    // deflate expands at most ~1032:1, so nothing a file describes can be larger than this (a corrupt shape would
    // otherwise turn into a huge zero-filled allocation)
    if ((end - start) * row_bytes > size_ * 1100 + (1u << 20)) H5FAIL("%s: dataset of %llu bytes cannot come from a %llu-byte file", path_.c_str(), (unsigned long long)((end - start) * row_bytes), (unsigned long long)size_);
    std::vector<uint8_t> out((size_t)((end - start) * row_bytes), 0);
    if (out.empty()) return out;
    Layout lay;
    parse_layout(m, di, lay);
    const uint64_t total = checked_mul(di.n_elements(), elem, "dataset size");
    if (lay.cls == 0) {
        if (lay.compact_size < total) H5FAIL("%s: compact dataset shorter than its dataspace", path_.c_str());
        memcpy(out.data(), lay.compact + start * row_bytes, out.size());
    } else if (lay.cls == 1) {
        if (undefined(lay.addr)) return out; // never written: fill value (zeros)
        if (lay.size < total) H5FAIL("%s: contiguous dataset shorter than its dataspace", path_.c_str());
        memcpy(out.data(), at(base_addr_ + lay.addr + start * row_bytes, out.size()), out.size());
    } else {
        uint64_t chunk_elems = 1;
        for (uint64_t c : lay.chunk_dims) {
            if (c == 0) H5FAIL("%s: zero chunk dimension", path_.c_str());
            chunk_elems = checked_mul(chunk_elems, c, "chunk size");
        }
        const uint64_t chunk_size_bytes = checked_mul(chunk_elems, elem, "chunk size");
        if (chunk_size_bytes > 0xFFFFFFFFull) H5FAIL("%s: chunks of 4 GiB or more are not supported", path_.c_str());
        // the chunks that intersect the slice; each lands in its own region of `out`, so they can be inflated side by side.
        // A chunk's offset comes from the file (B-tree key / index position): it must sit on the chunk grid inside the
        // dataspace — the copy below trusts it.
        std::vector<const Chunk *> todo;
        for (const Chunk &c : lay.chunks) {
            if (c.offset.size() != rank) H5FAIL("%s: chunk with the wrong number of coordinates", path_.c_str());
            bool inside = true;
            for (unsigned i = 0; i < rank; i++) {
                if (c.offset[i] % lay.chunk_dims[i]) H5FAIL("%s: chunk offset off the chunk grid", path_.c_str());
                inside = inside && c.offset[i] < di.dims[i];
            }
            if (!inside) continue;
            if (c.offset[0] >= end || lay.chunk_dims[0] > UINT64_MAX - c.offset[0] || c.offset[0] + lay.chunk_dims[0] <= start) continue;
            todo.push_back(&c);
        }
        auto place = [&](const Chunk &c, std::vector<uint64_t> &idx) {
            const uint8_t *src = at(base_addr_ + c.addr, c.size);
            std::vector<uint8_t> buf(src, src + c.size);
            unfilter(buf, lay.filters, c.filter_mask, chunk_size_bytes + 8);
            if (buf.size() < chunk_size_bytes) H5FAIL("%s: chunk shorter than its declared shape", path_.c_str());
            // every copy is checked against both buffers (defence in depth behind the offset validation above)
            auto copy = [&](uint64_t dst_byte, uint64_t src_byte, uint64_t n) {
                if (dst_byte > out.size() || n > out.size() - dst_byte || src_byte > buf.size() || n > buf.size() - src_byte)
                    H5FAIL("%s: chunk does not fit its dataset (corrupt chunk index)", path_.c_str());
                memcpy(out.data() + dst_byte, buf.data() + src_byte, (size_t)n);
            };
            // copy the runs along the last dimension; idx walks the other dimensions of the chunk
            const uint64_t last = rank - 1;
            const uint64_t run = std::min(lay.chunk_dims[last], di.dims[last] - c.offset[last]);
            std::fill(idx.begin(), idx.end(), 0);
            for (;;) {
                bool in_range = true;
                uint64_t dst_elem = 0, src_elem = 0;
                for (unsigned i = 0; i < rank; i++) {
                    const uint64_t g = c.offset[i] + (i == last ? 0 : idx[i]);
                    if (g >= di.dims[i]) in_range = false;
                    dst_elem = dst_elem * di.dims[i] + g;
                    src_elem = src_elem * lay.chunk_dims[i] + (i == last ? 0 : idx[i]);
                }
                if (in_range) {
                    const uint64_t g0 = rank == 1 ? c.offset[0] : c.offset[0] + idx[0];
                    if (rank == 1) { // the run itself crosses the slice
                        const uint64_t lo = std::max(start, c.offset[0]), hi = std::min(end, c.offset[0] + run);
                        if (lo < hi) copy((lo - start) * elem, (lo - c.offset[0]) * elem, (hi - lo) * elem);
                    } else if (g0 >= start && g0 < end) {
                        copy((dst_elem - start * row_elems) * elem, src_elem * elem, run * elem);
                    }
                }
                int k = (int)rank - 2; // odometer over dimensions 0 .. rank-2
                for (; k >= 0; k--) {
                    if (++idx[k] < lay.chunk_dims[k]) break;
                    idx[k] = 0;
                }
                if (k < 0) break;
            }
        };
        // Cell Ranger's `data` / `indices` of a 10^6-cell matrix are ~10^4 deflated chunks (12 GB inflated): zlib on one
        // core is the whole load time, so large reads deal the chunks over a few threads (libhdf5 reads them one by one)
        unsigned n_thr = 1;
        if (!lay.filters.empty() && todo.size() >= 16 && out.size() >= (64u << 20)) {
            const unsigned cap = (unsigned)std::max(1, global_options().h5_threads);
            n_thr = std::max(1u, std::min({cap, std::thread::hardware_concurrency(), (unsigned)(todo.size() / 4)}));
        }
        if (n_thr == 1) {
            std::vector<uint64_t> idx(rank);
            for (const Chunk *c : todo) place(*c, idx);
        } else {
            std::atomic<size_t> next{0};
            std::atomic<bool> failed{false};
            std::mutex err_m;
            std::string err_msg;
            int err_code = SCANRS_ERR_IO;
            auto worker = [&] {
                std::vector<uint64_t> idx(rank);
                try {
                    for (;;) {
                        const size_t i = next.fetch_add(1, std::memory_order_relaxed);
                        if (i >= todo.size() || failed.load(std::memory_order_relaxed)) return;
                        place(*todo[i], idx);
                    }
                } catch (const Failure &f) { // the message lives in this thread's error slot: carry it over
                    std::lock_guard<std::mutex> lk(err_m);
                    if (!failed.exchange(true)) {
                        err_msg = scanrs_last_error();
                        err_code = f.code;
                    }
                } catch (const std::exception &e) {
                    std::lock_guard<std::mutex> lk(err_m);
                    if (!failed.exchange(true)) err_msg = e.what();
                }
            };
            std::vector<std::thread> pool;
            try {
                for (unsigned t = 1; t < n_thr; t++) pool.emplace_back(worker);
            } catch (...) { // fewer helpers than planned: the caller's thread finishes the queue
            }
            worker();
            for (auto &th : pool) th.join();
            if (failed.load()) ::scanrs::fail(err_code, "%s", err_msg.c_str());
        }
    }
    return out;
}

namespace {
template <typename T, typename S>
inline T saturate(S v) {
    if constexpr (std::is_floating_point<T>::value) {
        return (T)v;
    } else if constexpr (std::is_floating_point<S>::value) {
        if (!(v == v)) return 0;
        if (v <= (S)std::numeric_limits<T>::min()) return std::numeric_limits<T>::min();
        if (v >= (S)std::numeric_limits<T>::max()) return std::numeric_limits<T>::max();
        return (T)v;
    } else if constexpr (std::is_signed<S>::value) {
        if constexpr (std::is_signed<T>::value) {
            if (v < (S)std::numeric_limits<T>::min() && sizeof(S) > sizeof(T)) return std::numeric_limits<T>::min();
            if (v > (S)std::numeric_limits<T>::max() && sizeof(S) > sizeof(T)) return std::numeric_limits<T>::max();
            return (T)v;
        } else {
            if (v < 0) return 0;
            if ((uint64_t)v > (uint64_t)std::numeric_limits<T>::max()) return std::numeric_limits<T>::max();
            return (T)v;
        }
    } else {
        if ((uint64_t)v > (uint64_t)std::numeric_limits<T>::max()) return std::numeric_limits<T>::max();
        return (T)v;
    }
}
} // namespace

template <typename T>
std::vector<T> File::read(Object dataset, uint64_t start, uint64_t end, DatasetInfo *info_out) const {
    DatasetInfo di;
    const std::vector<uint8_t> raw = read_raw(dataset, start, end, &di);
    if (info_out) *info_out = di;
    if (di.type.cls == TypeInfo::STRING) H5FAIL("%s: dataset holds strings, numbers were asked for", path_.c_str());
    const size_t es = di.type.size;
    const size_t n = es ? raw.size() / es : 0;
    std::vector<T> out(n);
    // little-endian stored types (every 10x file): typed loads instead of assembling bytes
    if (!di.type.big_endian && n) {
        const uint8_t *r = raw.data();
        auto typed = [&](auto tag) {
            typedef decltype(tag) S;
            for (size_t i = 0; i < n; i++) {
                S v;
                memcpy(&v, r + i * sizeof(S), sizeof(S));
                out[i] = saturate<T, typename std::conditional<std::is_floating_point<S>::value, double,
                                                               typename std::conditional<std::is_signed<S>::value, int64_t, uint64_t>::type>::type>(v);
            }
        };
        bool done = true;
        if (di.type.cls == TypeInfo::FLOAT && es == 8) typed(double());
        else if (di.type.cls == TypeInfo::FLOAT && es == 4) typed(float());
        else if (di.type.cls == TypeInfo::FIXED && di.type.is_signed && es == 8) typed(int64_t());
        else if (di.type.cls == TypeInfo::FIXED && di.type.is_signed && es == 4) typed(int32_t());
        else if (di.type.cls == TypeInfo::FIXED && !di.type.is_signed && es == 8) typed(uint64_t());
        else if (di.type.cls == TypeInfo::FIXED && !di.type.is_signed && es == 4) typed(uint32_t());
        else done = false;
        if (done) return out;
    }
    for (size_t i = 0; i < n; i++) {
        uint64_t bits = 0;
        const uint8_t *p = raw.data() + i * es;
        if (di.type.big_endian)
            for (size_t b = 0; b < es; b++) bits = (bits << 8) | p[b];
        else
            for (size_t b = 0; b < es; b++) bits |= (uint64_t)p[b] << (8 * b);
        if (di.type.cls == TypeInfo::FLOAT) {
            double v;
            if (es == 4) {
                float f;
                const uint32_t b32 = (uint32_t)bits;
                memcpy(&f, &b32, 4);
                v = f;
            } else {
                memcpy(&v, &bits, 8);
            }
            out[i] = saturate<T, double>(v);
        } else if (di.type.is_signed) {
            int64_t v = (int64_t)bits;
            if (es < 8) { // sign-extend
                const unsigned sh = 64 - 8 * (unsigned)es;
                v = (int64_t)(bits << sh) >> sh;
            }
            out[i] = saturate<T, int64_t>(v);
        } else {
            out[i] = saturate<T, uint64_t>(bits);
        }
    }
    return out;
}

template std::vector<uint8_t> File::read<uint8_t>(Object, uint64_t, uint64_t, DatasetInfo *) const;
template std::vector<int16_t> File::read<int16_t>(Object, uint64_t, uint64_t, DatasetInfo *) const;
template std::vector<uint16_t> File::read<uint16_t>(Object, uint64_t, uint64_t, DatasetInfo *) const;
template std::vector<int32_t> File::read<int32_t>(Object, uint64_t, uint64_t, DatasetInfo *) const;
template std::vector<uint32_t> File::read<uint32_t>(Object, uint64_t, uint64_t, DatasetInfo *) const;
template std::vector<int64_t> File::read<int64_t>(Object, uint64_t, uint64_t, DatasetInfo *) const;
template std::vector<uint64_t> File::read<uint64_t>(Object, uint64_t, uint64_t, DatasetInfo *) const;
template std::vector<float> File::read<float>(Object, uint64_t, uint64_t, DatasetInfo *) const;
template std::vector<double> File::read<double>(Object, uint64_t, uint64_t, DatasetInfo *) const;

std::vector<std::string> File::read_strings(Object dataset, uint64_t start, uint64_t end) const {
    DatasetInfo di;
    const std::vector<uint8_t> raw = read_raw(dataset, start, end, &di);
    if (di.type.cls != TypeInfo::STRING) H5FAIL("%s: dataset does not hold fixed-length strings", path_.c_str());
    const size_t es = di.type.size;
    const size_t n = es ? raw.size() / es : 0;
    std::vector<std::string> out(n);
    for (size_t i = 0; i < n; i++) {
        const char *p = (const char *)raw.data() + i * es;
        size_t len = strnlen(p, es);
        if (di.type.str_pad == 2)
            while (len > 0 && p[len - 1] == ' ') len--;
        out[i].assign(p, len);
    }
    return out;
}

} // namespace h5
} // namespace scanrs
