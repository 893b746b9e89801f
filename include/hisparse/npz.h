// hisparse/npz.h — minimal reader for NumPy .npy arrays inside .npz (zip) archives.
//
// The reference ingests scipy.sparse.save_npz archives through the third-party cnpy library
// (sw/data_loader.h:7,53-68; cnpy is un-vendored and unpinned, Readme.md:32-41).  cnpy is not in
// this image, so this is a from-scratch reader of the two published formats involved:
//   * ZIP (PKWARE APPNOTE): end-of-central-directory record -> central directory -> local header;
//     methods 0 (stored) and 8 (deflate, via zlib raw inflate).  Sizes are taken from the central
//     directory (with the zip64 extra field when present) because numpy writes every member with
//     force_zip64, which leaves 0xffffffff placeholders in the local header.
//   * NPY 1.0/2.0/3.0 (numpy.lib.format): magic, version, little-endian header length, a Python
//     dict literal with 'descr', 'fortran_order', 'shape'.
// Only little-endian 4- and 8-byte integer and 4/8-byte float payloads are accepted — the layouts
// scipy emits for CSR matrices.
#ifndef HISPARSE_NPZ_H_
#define HISPARSE_NPZ_H_

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace hisparse {
namespace npz {

struct Array {
    std::string descr;           // e.g. "<f4", "<i4", "<i8"
    std::vector<uint64_t> shape;
    std::vector<uint8_t> bytes;  // raw little-endian payload
    size_t word_size() const { return descr.size() >= 3 ? size_t(std::stoul(descr.substr(2))) : 0; }
    char kind() const { return descr.size() >= 2 ? descr[1] : '?'; }
    uint64_t count() const {   // saturates instead of wrapping (a hostile header cannot make count() * word_size() look small)
        uint64_t n = 1;
        for (uint64_t s : shape) {
            if (s != 0 && n > (uint64_t(1) << 60) / s) return uint64_t(1) << 60;
            n *= s;
        }
        return n;
    }
    // element i widened to 64-bit integer (for i4/u4/i8/u8 payloads)
    int64_t as_int(uint64_t i) const {
        const uint8_t* p = bytes.data() + i * word_size();
        if (word_size() == 4) {
            if (kind() == 'u') { uint32_t v; std::memcpy(&v, p, 4); return int64_t(v); }
            int32_t v; std::memcpy(&v, p, 4); return v;
        }
        int64_t v; std::memcpy(&v, p, 8); return v;
    }
    float as_float(uint64_t i) const {
        const uint8_t* p = bytes.data() + i * word_size();
        if (kind() == 'f' && word_size() == 4) { float v; std::memcpy(&v, p, 4); return v; }
        if (kind() == 'f' && word_size() == 8) { double v; std::memcpy(&v, p, 8); return float(v); }
        return float(as_int(i));
    }
};

namespace detail {

inline uint16_t rd16(const uint8_t* p) { return uint16_t(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }
inline uint64_t rd64(const uint8_t* p) { return uint64_t(rd32(p)) | (uint64_t(rd32(p + 4)) << 32); }

inline std::vector<uint8_t> read_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("npz: cannot open " + path);
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(n > 0 ? size_t(n) : 0);
    size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size()) throw std::runtime_error("npz: short read on " + path);
    return buf;
}

inline std::vector<uint8_t> inflate_raw(const uint8_t* src, uint64_t src_len, uint64_t dst_len) {
    std::vector<uint8_t> out(dst_len);
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) throw std::runtime_error("npz: inflateInit2 failed");
    uint64_t in_off = 0, out_off = 0;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        uint64_t in_chunk = src_len - in_off, out_chunk = dst_len - out_off;
        if (in_chunk > (1u << 30)) in_chunk = 1u << 30;
        if (out_chunk > (1u << 30)) out_chunk = 1u << 30;
        zs.next_in = const_cast<Bytef*>(src + in_off);
        zs.avail_in = uInt(in_chunk);
        zs.next_out = out.data() + out_off;
        zs.avail_out = uInt(out_chunk);
        rc = inflate(&zs, Z_NO_FLUSH);
        in_off += in_chunk - zs.avail_in;
        out_off += out_chunk - zs.avail_out;
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); throw std::runtime_error("npz: inflate error"); }
        if (rc == Z_OK && in_off >= src_len && out_off >= dst_len) break;
    }
    inflateEnd(&zs);
    if (out_off != dst_len) throw std::runtime_error("npz: inflated size mismatch");
    return out;
}

// value of a key in the header dict literal, e.g. key "descr" -> "<f4", key "shape" -> "(3, 4)"
inline std::string dict_value(const std::string& hdr, const std::string& key) {
    size_t k = hdr.find("'" + key + "'");
    if (k == std::string::npos) throw std::runtime_error("npy: header lacks " + key);
    size_t c = hdr.find(':', k);
    if (c == std::string::npos) throw std::runtime_error("npy: malformed header near " + key);
    size_t b = hdr.find_first_not_of(" ", c + 1);
    if (b == std::string::npos) throw std::runtime_error("npy: malformed header near " + key);
    size_t e;
    if (hdr[b] == '\'') {
        if ((e = hdr.find('\'', b + 1)) == std::string::npos) throw std::runtime_error("npy: unterminated string in header");
        return hdr.substr(b + 1, e - b - 1);
    }
    if (hdr[b] == '(') {
        if ((e = hdr.find(')', b)) == std::string::npos) throw std::runtime_error("npy: unterminated tuple in header");
        return hdr.substr(b, e - b + 1);
    }
    if ((e = hdr.find_first_of(",}", b)) == std::string::npos) throw std::runtime_error("npy: malformed header near " + key);
    return hdr.substr(b, e - b);
}

}  // namespace detail

inline Array parse_npy(const uint8_t* p, uint64_t n) {
    using namespace detail;
    if (n < 10 || std::memcmp(p, "\x93NUMPY", 6) != 0) throw std::runtime_error("npy: bad magic");
    unsigned major = p[6];
    uint64_t hlen, hoff;
    if (major == 1) { hlen = rd16(p + 8); hoff = 10; } else { hlen = rd32(p + 8); hoff = 12; }
    if (hlen > n || hoff + hlen > n) throw std::runtime_error("npy: truncated header");
    std::string hdr(reinterpret_cast<const char*>(p + hoff), hlen);
    Array a;
    a.descr = dict_value(hdr, "descr");
    if (a.descr.size() < 3 || (a.descr[0] != '<' && a.descr[0] != '|'))
        throw std::runtime_error("npy: unsupported byte order in " + a.descr);
    if (dict_value(hdr, "fortran_order").find("True") != std::string::npos)
        throw std::runtime_error("npy: fortran order unsupported");
    std::string shp = dict_value(hdr, "shape");
    for (size_t i = 0; i < shp.size();) {
        if (shp[i] >= '0' && shp[i] <= '9') {
            size_t j = i;
            while (j < shp.size() && shp[j] >= '0' && shp[j] <= '9') ++j;
            if (j - i > 18) throw std::runtime_error("npy: dimension too large");
            a.shape.push_back(std::stoull(shp.substr(i, j - i)));
            i = j;
        } else {
            ++i;
        }
    }
    size_t ws = a.word_size();
    if ((a.kind() != 'f' && a.kind() != 'i' && a.kind() != 'u') || (ws != 4 && ws != 8))
        throw std::runtime_error("npy: unsupported dtype " + a.descr);
    const uint64_t room = n - hoff - hlen;
    if (a.count() > room / ws) throw std::runtime_error("npy: truncated payload");
    const uint64_t need = a.count() * ws;
    a.bytes.assign(p + hoff + hlen, p + hoff + hlen + need);
    return a;
}

// Load every member "<name>.npy" of the archive; keys have the ".npy" suffix stripped.
inline std::map<std::string, Array> load(const std::string& path) {
    using namespace detail;
    std::vector<uint8_t> z = read_file(path);
    const uint64_t n = z.size();
    if (n < 22) throw std::runtime_error("npz: file too small");
    // end-of-central-directory: scan backwards for PK\5\6
    int64_t eocd = -1;
    for (int64_t i = int64_t(n) - 22; i >= 0 && i >= int64_t(n) - 22 - 65536; --i)
        if (rd32(&z[i]) == 0x06054b50u) { eocd = i; break; }
    if (eocd < 0) throw std::runtime_error("npz: no end-of-central-directory record");
    uint64_t entries = rd16(&z[eocd + 10]);
    uint64_t cd_off = rd32(&z[eocd + 16]);
    if (entries == 0xffff || cd_off == 0xffffffffu) {  // zip64 end-of-central-directory
        if (eocd < 20 || rd32(&z[eocd - 20]) != 0x07064b50u) throw std::runtime_error("npz: zip64 locator missing");
        uint64_t e64 = rd64(&z[eocd - 20 + 8]);
        if (e64 > n || n - e64 < 56 || rd32(&z[e64]) != 0x06064b50u) throw std::runtime_error("npz: bad zip64 EOCD");
        entries = rd64(&z[e64 + 32]);
        cd_off = rd64(&z[e64 + 48]);
    }
    std::map<std::string, Array> out;
    uint64_t p = cd_off;
    for (uint64_t e = 0; e < entries; ++e) {
        if (p > n || n - p < 46 || rd32(&z[p]) != 0x02014b50u) throw std::runtime_error("npz: bad central directory");
        unsigned method = rd16(&z[p + 10]);
        uint64_t csize = rd32(&z[p + 20]), usize = rd32(&z[p + 24]);
        unsigned nlen = rd16(&z[p + 28]), xlen = rd16(&z[p + 30]), clen = rd16(&z[p + 32]);
        uint64_t lho = rd32(&z[p + 42]);
        if (p + 46 + uint64_t(nlen) + xlen + clen > n) throw std::runtime_error("npz: central directory entry exceeds file");
        std::string name(reinterpret_cast<const char*>(&z[p + 46]), nlen);
        // zip64 extra field (id 0x0001): present values appear in the order usize, csize, offset
        uint64_t x = p + 46 + nlen, xend = x + xlen;
        while (x + 4 <= xend) {
            unsigned id = rd16(&z[x]), sz = rd16(&z[x + 2]);
            if (x + 4 + sz > xend) throw std::runtime_error("npz: extra field exceeds its record");
            if (id == 0x0001) {
                uint64_t q = x + 4;
                const uint64_t qend = x + 4 + sz;
                auto take64 = [&](uint64_t& v) {
                    if (q + 8 > qend) throw std::runtime_error("npz: zip64 extra field too short");
                    v = rd64(&z[q]);
                    q += 8;
                };
                if (usize == 0xffffffffu) take64(usize);
                if (csize == 0xffffffffu) take64(csize);
                if (lho == 0xffffffffu) take64(lho);
            }
            x += 4 + sz;
        }
        p += 46 + uint64_t(nlen) + xlen + clen;
        if (lho > n || n - lho < 30 || rd32(&z[lho]) != 0x04034b50u) throw std::runtime_error("npz: bad local header");
        const uint64_t data = lho + 30 + rd16(&z[lho + 26]) + rd16(&z[lho + 28]);
        if (data > n || csize > n - data) throw std::runtime_error("npz: member exceeds file");
        if (usize > (uint64_t(1) << 40)) throw std::runtime_error("npz: member larger than 1 TiB");
        std::vector<uint8_t> raw;
        const uint8_t* src = &z[data];
        if (method == 8) { raw = inflate_raw(src, csize, usize); src = raw.data(); }
        else if (method != 0) throw std::runtime_error("npz: unsupported compression method");
        std::string key = name;
        if (key.size() > 4 && key.compare(key.size() - 4, 4, ".npy") == 0) key.resize(key.size() - 4);
        if (name.size() > 4 && name.compare(name.size() - 4, 4, ".npy") == 0) {
            // 'format' in scipy archives is a unicode/bytes scalar ("|S3"/"<U3"): skip non-numeric members
            try { out[key] = parse_npy(src, usize); } catch (const std::runtime_error&) { /* non-numeric member */ }
        }
    }
    return out;
}

}  // namespace npz
}  // namespace hisparse

#endif  // HISPARSE_NPZ_H_
