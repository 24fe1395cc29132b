// k_io.hip -- data ingest in front of the hot path, native host code + HIP copy engine.
//   bx_io_probe / bx_io_read_xyz   point-cloud readers for the three formats the reference's loaders open:
//                                  .ply  (dataset/threedmatch.py:75-79 via open3d.io.read_point_cloud)
//                                  .pcd  (dataset/tiers.py:72-73 via open3d.io.read_point_cloud)
//                                  .bin  (dataset/kitti.py:76-80: np.fromfile(float32).reshape(-1, 4)[:, :3])
//   bx_prefetch_*                  a worker thread parses the files of the NEXT pairs straight into pinned host memory and
//                                  uploads them with hipMemcpyAsync on its own stream while the GPU registers the current pair;
//                                  the consumer stream only waits on an event.
// The reference reads every pair synchronously on the main thread (DataLoader num_workers = 0, config/indoor_config.py:23) and
// uploads it with a blocking .cuda(); at > 20 pairs/s per GPU that is the ceiling ("Data t" in test.py:237).
// Formats follow their public specifications (PLY 1.0: ascii / binary_little_endian / binary_big_endian, scalar and list
// properties; PCD 0.7: ascii / binary / binary_compressed (LZF)).  Open3D itself is not in this image: parity is pinned
// against files written by tests/ (numpy) and, for .bin, against np.fromfile.
#include "bx_common.h"

#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct FileBuf {
    std::vector<unsigned char> d;
    size_t file_size = 0;
    bool load(const char* path, size_t max_bytes = ~(size_t)0)    // max_bytes: header-only probes read the first 64 KiB
    {
        FILE* f = fopen(path, "rb");
        if (!f) { bx_set_error("bx_io: cannot open %s: %s", path, strerror(errno)); return false; }
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz < 0) { fclose(f); bx_set_error("bx_io: cannot size %s", path); return false; }
        file_size = (size_t)sz;
        if ((size_t)sz > max_bytes) sz = (long)max_bytes;
        d.resize((size_t)sz);
        const size_t got = sz ? fread(d.data(), 1, (size_t)sz, f) : 0;
        fclose(f);
        if (got != (size_t)sz) { bx_set_error("bx_io: short read on %s", path); return false; }
        return true;
    }
};

bool ends_with(const std::string& s, const char* suf)
{
    const size_t n = strlen(suf);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; ++i) {
        char a = s[s.size() - n + i], b = suf[i];
        if (a >= 'A' && a <= 'Z') a = (char)(a - 'A' + 'a');
        if (a != b) return false;
    }
    return true;
}

std::vector<std::string> split_ws(const std::string& line)
{
    std::vector<std::string> out;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && (line[i] == ' ' || line[i] == '\t' || line[i] == '\r')) ++i;
        size_t j = i;
        while (j < line.size() && line[j] != ' ' && line[j] != '\t' && line[j] != '\r') ++j;
        if (j > i) out.push_back(line.substr(i, j - i));
        i = j;
    }
    return out;
}

// next text line starting at pos (without the newline); advances pos behind it
bool next_line(const FileBuf& fb, size_t& pos, std::string& line)
{
    if (pos >= fb.d.size()) return false;
    size_t e = pos;
    while (e < fb.d.size() && fb.d[e] != '\n') ++e;
    line.assign(reinterpret_cast<const char*>(fb.d.data()) + pos, e - pos);
    if (!line.empty() && line.back() == '\r') line.pop_back();
    pos = e < fb.d.size() ? e + 1 : e;
    return true;
}

enum Scalar { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_BAD };

int scalar_size(Scalar t) { static const int s[] = {1, 1, 2, 2, 4, 4, 4, 8, 0}; return s[t]; }

Scalar ply_type(const std::string& n)
{
    if (n == "char" || n == "int8") return T_I8;
    if (n == "uchar" || n == "uint8") return T_U8;
    if (n == "short" || n == "int16") return T_I16;
    if (n == "ushort" || n == "uint16") return T_U16;
    if (n == "int" || n == "int32") return T_I32;
    if (n == "uint" || n == "uint32") return T_U32;
    if (n == "float" || n == "float32") return T_F32;
    if (n == "double" || n == "float64") return T_F64;
    return T_BAD;
}

double load_scalar(const unsigned char* p, Scalar t, bool swap)
{
    unsigned char b[8];
    const int n = scalar_size(t);
    if (swap) for (int i = 0; i < n; ++i) b[i] = p[n - 1 - i];
    else memcpy(b, p, (size_t)n);
    switch (t) {
    case T_I8: return (double)*reinterpret_cast<signed char*>(b);
    case T_U8: return (double)b[0];
    case T_I16: { int16_t v; memcpy(&v, b, 2); return (double)v; }
    case T_U16: { uint16_t v; memcpy(&v, b, 2); return (double)v; }
    case T_I32: { int32_t v; memcpy(&v, b, 4); return (double)v; }
    case T_U32: { uint32_t v; memcpy(&v, b, 4); return (double)v; }
    case T_F32: { float v; memcpy(&v, b, 4); return (double)v; }
    case T_F64: { double v; memcpy(&v, b, 8); return v; }
    default: return 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------- PLY
struct PlyProp { std::string name; bool is_list; Scalar count_t, t; };
struct PlyElem { std::string name; int64_t count; std::vector<PlyProp> props; };

int read_ply(const FileBuf& fb, const char* path, float* out, int64_t cap, int64_t* n_out)
{
    size_t pos = 0;
    std::string line;
    if (!next_line(fb, pos, line) || line != "ply") { bx_set_error("bx_io: %s is not a PLY file", path); return BX_ERR_ARG; }
    int fmt = -1;   // 0 ascii, 1 little, 2 big
    std::vector<PlyElem> elems;
    bool ended = false;
    while (next_line(fb, pos, line)) {
        const std::vector<std::string> w = split_ws(line);
        if (w.empty()) continue;
        if (w[0] == "format" && w.size() >= 2) fmt = w[1] == "ascii" ? 0 : (w[1] == "binary_little_endian" ? 1 : (w[1] == "binary_big_endian" ? 2 : -1));
        else if (w[0] == "element" && w.size() >= 3) {
            PlyElem e; e.name = w[1]; e.count = atoll(w[2].c_str());
            if (e.count < 0 || e.count > ((int64_t)1 << 40)) { bx_set_error("bx_io: %s: element count %s out of range", path, w[2].c_str()); return BX_ERR_ARG; }
            elems.push_back(e);
        }
        else if (w[0] == "property" && !elems.empty()) {
            PlyProp p;
            if (w.size() >= 5 && w[1] == "list") { p.is_list = true; p.count_t = ply_type(w[2]); p.t = ply_type(w[3]); p.name = w[4]; }
            else if (w.size() >= 3) { p.is_list = false; p.count_t = T_BAD; p.t = ply_type(w[1]); p.name = w[2]; }
            else { bx_set_error("bx_io: %s: malformed property line", path); return BX_ERR_ARG; }
            if (p.t == T_BAD || (p.is_list && p.count_t == T_BAD)) { bx_set_error("bx_io: %s: unknown PLY type in '%s'", path, line.c_str()); return BX_ERR_ARG; }
            elems.back().props.push_back(p);
        } else if (w[0] == "end_header") { ended = true; break; }
    }
    if (!ended || fmt < 0) { bx_set_error("bx_io: %s: incomplete PLY header", path); return BX_ERR_ARG; }
    int vi = -1;
    for (size_t i = 0; i < elems.size(); ++i) if (elems[i].name == "vertex") vi = (int)i;
    if (vi < 0) { *n_out = 0; return BX_OK; }
    const int64_t n = elems[vi].count;
    *n_out = n;
    if (!out) return BX_OK;
    if (n > cap) { bx_set_error("bx_io: %s holds %lld points, buffer holds %lld", path, (long long)n, (long long)cap); return BX_ERR_ARG; }
    const bool swap = fmt == 2;
    for (int ei = 0; ei <= vi; ++ei) {
        const PlyElem& e = elems[ei];
        // every row of a non-empty element occupies at least one byte of the body (a line feed / one scalar)
        if (!e.props.empty() && e.count > (int64_t)(fb.d.size() - pos)) {
            bx_set_error("bx_io: %s: element '%s' declares %lld rows, %lld bytes remain", path, e.name.c_str(), (long long)e.count, (long long)(fb.d.size() - pos));
            return BX_ERR_ARG;
        }
        int ix[3] = {-1, -1, -1};
        bool has_list = false;
        for (size_t k = 0; k < e.props.size(); ++k) {
            if (e.props[k].is_list) has_list = true;
            if (e.props[k].name == "x") ix[0] = (int)k;
            if (e.props[k].name == "y") ix[1] = (int)k;
            if (e.props[k].name == "z") ix[2] = (int)k;
        }
        const bool want = ei == vi;
        if (want && (ix[0] < 0 || ix[1] < 0 || ix[2] < 0)) { bx_set_error("bx_io: %s: vertex element without x/y/z", path); return BX_ERR_ARG; }
        if (fmt == 0) {
            for (int64_t r = 0; r < e.count; ++r) {
                if (!next_line(fb, pos, line)) { bx_set_error("bx_io: %s: truncated PLY body", path); return BX_ERR_ARG; }
                if (!want) continue;
                const char* p = line.c_str();
                char* endp = nullptr;
                for (size_t k = 0; k < e.props.size(); ++k) {
                    if (e.props[k].is_list) {
                        // the per-row list length comes from the file: a value is at least two characters ("0 "), so a count
                        // that the rest of the line cannot hold is corrupt (and would otherwise spin for up to 2^63 iterations)
                        const long cnt = strtol(p, &endp, 10);
                        if (endp == p || cnt < 0 || (size_t)cnt > (size_t)(line.c_str() + line.size() - endp)) {
                            bx_set_error("bx_io: %s: bad list length on vertex line %lld", path, (long long)r);
                            return BX_ERR_ARG;
                        }
                        p = endp;
                        for (long q = 0; q < cnt; ++q) {
                            (void)strtod(p, &endp);
                            if (endp == p) { bx_set_error("bx_io: %s: short list on vertex line %lld", path, (long long)r); return BX_ERR_ARG; }
                            p = endp;
                        }
                        continue;
                    }
                    const double v = strtod(p, &endp);
                    if (endp == p) { bx_set_error("bx_io: %s: malformed vertex line %lld", path, (long long)r); return BX_ERR_ARG; }
                    p = endp;
                    for (int c = 0; c < 3; ++c) if (ix[c] == (int)k) out[r * 3 + c] = (float)v;
                }
            }
        } else if (!has_list) {
            size_t stride = 0;
            std::vector<size_t> off(e.props.size());
            for (size_t k = 0; k < e.props.size(); ++k) { off[k] = stride; stride += (size_t)scalar_size(e.props[k].t); }
            if (pos + stride * (size_t)e.count > fb.d.size()) { bx_set_error("bx_io: %s: truncated PLY body", path); return BX_ERR_ARG; }
            if (want) {
                const unsigned char* base = fb.d.data() + pos;
                const Scalar t0 = e.props[ix[0]].t, t1 = e.props[ix[1]].t, t2 = e.props[ix[2]].t;
                const size_t o0 = off[ix[0]], o1 = off[ix[1]], o2 = off[ix[2]];
                if (!swap && t0 == T_F32 && t1 == T_F32 && t2 == T_F32) {     // the common case: plain float copies
                    for (int64_t r = 0; r < n; ++r) {
                        const unsigned char* q = base + (size_t)r * stride;
                        memcpy(out + r * 3, q + o0, 4); memcpy(out + r * 3 + 1, q + o1, 4); memcpy(out + r * 3 + 2, q + o2, 4);
                    }
                } else {
                    for (int64_t r = 0; r < n; ++r) {
                        const unsigned char* q = base + (size_t)r * stride;
                        out[r * 3] = (float)load_scalar(q + o0, t0, swap);
                        out[r * 3 + 1] = (float)load_scalar(q + o1, t1, swap);
                        out[r * 3 + 2] = (float)load_scalar(q + o2, t2, swap);
                    }
                }
            }
            pos += stride * (size_t)e.count;
        } else {
            for (int64_t r = 0; r < e.count; ++r)
                for (size_t k = 0; k < e.props.size(); ++k) {
                    const PlyProp& pr = e.props[k];
                    if (pr.is_list) {
                        if (pos + (size_t)scalar_size(pr.count_t) > fb.d.size()) { bx_set_error("bx_io: %s: truncated PLY body", path); return BX_ERR_ARG; }
                        const double cntf = load_scalar(fb.d.data() + pos, pr.count_t, swap);
                        const size_t left = fb.d.size() - pos - (size_t)scalar_size(pr.count_t);
                        // a negative count would move pos backwards, a huge one wrap it: both are corrupt files
                        if (!(cntf >= 0.0) || cntf > (double)(left / (size_t)scalar_size(pr.t))) {
                            bx_set_error("bx_io: %s: bad list length in element '%s' row %lld", path, e.name.c_str(), (long long)r);
                            return BX_ERR_ARG;
                        }
                        pos += (size_t)scalar_size(pr.count_t) + (size_t)cntf * (size_t)scalar_size(pr.t);
                    } else {
                        if (pos + (size_t)scalar_size(pr.t) > fb.d.size()) { bx_set_error("bx_io: %s: truncated PLY body", path); return BX_ERR_ARG; }
                        if (want) for (int c = 0; c < 3; ++c) if (ix[c] == (int)k) out[r * 3 + c] = (float)load_scalar(fb.d.data() + pos, pr.t, swap);
                        pos += (size_t)scalar_size(pr.t);
                    }
                    if (pos > fb.d.size()) { bx_set_error("bx_io: %s: truncated PLY body", path); return BX_ERR_ARG; }
                }
        }
    }
    return BX_OK;
}

// ---------------------------------------------------------------------------------------------------- PCD
// LZF decompression (the format PCL writes for DATA binary_compressed): control byte < 32 -> literal run of ctrl+1 bytes;
// otherwise a back reference of length (ctrl >> 5) + 2 (7 = extended by one more byte) at distance ((ctrl & 31) << 8 | next) + 1.
bool lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len)
{
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        const unsigned ctrl = in[ip++];
        if (ctrl < 32) {
            const size_t run = ctrl + 1;
            if (ip + run > in_len || op + run > out_len) return false;
            memcpy(out + op, in + ip, run);
            ip += run; op += run;
        } else {
            size_t len = ctrl >> 5;
            if (len == 7) { if (ip >= in_len) return false; len += in[ip++]; }
            if (ip >= in_len) return false;
            const size_t dist = ((size_t)(ctrl & 31) << 8 | in[ip++]) + 1;
            len += 2;
            if (dist > op || op + len > out_len) return false;
            for (size_t i = 0; i < len; ++i, ++op) out[op] = out[op - dist];   // may overlap: byte by byte
        }
    }
    return op == out_len;
}

int read_pcd(const FileBuf& fb, const char* path, float* out, int64_t cap, int64_t* n_out)
{
    size_t pos = 0;
    std::string line;
    std::vector<std::string> fields, types;
    std::vector<int> sizes, counts;
    int64_t width = 0, height = 1, points = -1;
    std::string data;
    while (next_line(fb, pos, line)) {
        if (line.empty() || line[0] == '#') continue;
        const std::vector<std::string> w = split_ws(line);
        if (w.empty()) continue;
        if (w[0] == "FIELDS" || w[0] == "COLUMNS") fields.assign(w.begin() + 1, w.end());
        else if (w[0] == "SIZE") { sizes.clear(); for (size_t i = 1; i < w.size(); ++i) sizes.push_back(atoi(w[i].c_str())); }
        else if (w[0] == "TYPE") types.assign(w.begin() + 1, w.end());
        else if (w[0] == "COUNT") { counts.clear(); for (size_t i = 1; i < w.size(); ++i) counts.push_back(atoi(w[i].c_str())); }
        else if (w[0] == "WIDTH" && w.size() > 1) width = atoll(w[1].c_str());
        else if (w[0] == "HEIGHT" && w.size() > 1) height = atoll(w[1].c_str());
        else if (w[0] == "POINTS" && w.size() > 1) points = atoll(w[1].c_str());
        else if (w[0] == "DATA" && w.size() > 1) { data = w[1]; break; }
    }
    if (data.empty() || fields.empty() || sizes.size() != fields.size() || types.size() != fields.size()) {
        bx_set_error("bx_io: %s: incomplete PCD header", path);
        return BX_ERR_ARG;
    }
    if (counts.size() != fields.size()) counts.assign(fields.size(), 1);
    const int64_t n = points >= 0 ? points : width * height;
    if (n < 0 || n > ((int64_t)1 << 40) || width < 0 || height < 0) { bx_set_error("bx_io: %s: point count out of range", path); return BX_ERR_ARG; }
    for (size_t k = 0; k < fields.size(); ++k)
        if (sizes[k] < 1 || sizes[k] > 8 || (counts.size() == fields.size() && (counts[k] < 1 || counts[k] > 4096))) {
            bx_set_error("bx_io: %s: field size / count out of range", path);
            return BX_ERR_ARG;
        }
    *n_out = n;
    if (!out) return BX_OK;
    if (n > cap) { bx_set_error("bx_io: %s holds %lld points, buffer holds %lld", path, (long long)n, (long long)cap); return BX_ERR_ARG; }
    int fi[3] = {-1, -1, -1};
    std::vector<size_t> off(fields.size());
    size_t stride = 0;
    int ncol = 0;
    std::vector<int> col0(fields.size());
    for (size_t k = 0; k < fields.size(); ++k) {
        off[k] = stride; stride += (size_t)sizes[k] * (size_t)counts[k];
        col0[k] = ncol; ncol += counts[k];
        if (fields[k] == "x") fi[0] = (int)k;
        if (fields[k] == "y") fi[1] = (int)k;
        if (fields[k] == "z") fi[2] = (int)k;
    }
    if (fi[0] < 0 || fi[1] < 0 || fi[2] < 0) { bx_set_error("bx_io: %s: PCD without x/y/z fields", path); return BX_ERR_ARG; }
    auto ftype = [&](int k) -> Scalar {
        const char t = types[k][0];
        const int s = sizes[k];
        if (t == 'F') return s == 4 ? T_F32 : (s == 8 ? T_F64 : T_BAD);
        if (t == 'I') return s == 1 ? T_I8 : (s == 2 ? T_I16 : (s == 4 ? T_I32 : T_BAD));
        if (t == 'U') return s == 1 ? T_U8 : (s == 2 ? T_U16 : (s == 4 ? T_U32 : T_BAD));
        return T_BAD;
    };
    const Scalar tx = ftype(fi[0]), ty = ftype(fi[1]), tz = ftype(fi[2]);
    if (tx == T_BAD || ty == T_BAD || tz == T_BAD) { bx_set_error("bx_io: %s: unsupported PCD field type", path); return BX_ERR_ARG; }
    if (data == "ascii") {
        for (int64_t r = 0; r < n; ++r) {
            if (!next_line(fb, pos, line)) { bx_set_error("bx_io: %s: truncated PCD body", path); return BX_ERR_ARG; }
            const char* p = line.c_str();
            char* endp = nullptr;
            for (int c = 0; c < ncol; ++c) {
                const double v = strtod(p, &endp);
                if (endp == p) { bx_set_error("bx_io: %s: malformed PCD line %lld", path, (long long)r); return BX_ERR_ARG; }
                p = endp;
                for (int a = 0; a < 3; ++a) if (col0[fi[a]] == c) out[r * 3 + a] = (float)v;
            }
        }
    } else if (data == "binary") {
        if (pos + stride * (size_t)n > fb.d.size()) { bx_set_error("bx_io: %s: truncated PCD body", path); return BX_ERR_ARG; }
        const unsigned char* base = fb.d.data() + pos;
        if (tx == T_F32 && ty == T_F32 && tz == T_F32) {
            const size_t o0 = off[fi[0]], o1 = off[fi[1]], o2 = off[fi[2]];
            for (int64_t r = 0; r < n; ++r) {
                const unsigned char* q = base + (size_t)r * stride;
                memcpy(out + r * 3, q + o0, 4); memcpy(out + r * 3 + 1, q + o1, 4); memcpy(out + r * 3 + 2, q + o2, 4);
            }
        } else
        for (int64_t r = 0; r < n; ++r) {
            const unsigned char* q = base + (size_t)r * stride;
            out[r * 3] = (float)load_scalar(q + off[fi[0]], tx, false);
            out[r * 3 + 1] = (float)load_scalar(q + off[fi[1]], ty, false);
            out[r * 3 + 2] = (float)load_scalar(q + off[fi[2]], tz, false);
        }
    } else if (data == "binary_compressed") {
        if (pos + 8 > fb.d.size()) { bx_set_error("bx_io: %s: truncated PCD body", path); return BX_ERR_ARG; }
        uint32_t csz, usz;
        memcpy(&csz, fb.d.data() + pos, 4); memcpy(&usz, fb.d.data() + pos + 4, 4);
        pos += 8;
        if (pos + csz > fb.d.size() || (size_t)usz != stride * (size_t)n) { bx_set_error("bx_io: %s: inconsistent compressed PCD sizes", path); return BX_ERR_ARG; }
        std::vector<unsigned char> raw(usz);
        if (!lzf_decompress(fb.d.data() + pos, csz, raw.data(), usz)) { bx_set_error("bx_io: %s: LZF stream is corrupt", path); return BX_ERR_ARG; }
        // field-major layout: all values of field 0, then field 1, ...
        std::vector<size_t> fbase(fields.size());
        size_t acc = 0;
        for (size_t k = 0; k < fields.size(); ++k) { fbase[k] = acc; acc += (size_t)sizes[k] * (size_t)counts[k] * (size_t)n; }
        const Scalar tt[3] = {tx, ty, tz};
        for (int a = 0; a < 3; ++a) {
            const size_t es = (size_t)sizes[fi[a]] * (size_t)counts[fi[a]];
            const unsigned char* b = raw.data() + fbase[fi[a]];
            for (int64_t r = 0; r < n; ++r) out[r * 3 + a] = (float)load_scalar(b + (size_t)r * es, tt[a], false);
        }
    } else {
        bx_set_error("bx_io: %s: unknown PCD DATA mode '%s'", path, data.c_str());
        return BX_ERR_ARG;
    }
    return BX_OK;
}

// ---------------------------------------------------------------------------------------------------- KITTI .bin
int read_bin(const FileBuf& fb, const char* path, float* out, int64_t cap, int64_t* n_out)
{
    if (fb.file_size % 16 != 0) { bx_set_error("bx_io: %s: size %zu is not a multiple of 16 (float32 x 4 records)", path, fb.file_size); return BX_ERR_ARG; }
    const int64_t n = (int64_t)(fb.file_size / 16);
    *n_out = n;
    if (!out) return BX_OK;
    if (n > cap) { bx_set_error("bx_io: %s holds %lld points, buffer holds %lld", path, (long long)n, (long long)cap); return BX_ERR_ARG; }
    const float* s = reinterpret_cast<const float*>(fb.d.data());
    for (int64_t r = 0; r < n; ++r) { out[r * 3] = s[r * 4]; out[r * 3 + 1] = s[r * 4 + 1]; out[r * 3 + 2] = s[r * 4 + 2]; }
    return BX_OK;
}

int read_any_impl(const char* path, float* out, int64_t cap, int64_t* n_out);

// nothing may unwind through the C boundary or out of the worker thread (a corrupt header can ask for an absurd allocation)
int read_any(const char* path, float* out, int64_t cap, int64_t* n_out)
{
    try {
        return read_any_impl(path, out, cap, n_out);
    } catch (const std::exception& e) {
        bx_set_error("bx_io: %s: %s", path ? path : "(null)", e.what());
    } catch (...) {
        bx_set_error("bx_io: %s: unexpected exception", path ? path : "(null)");
    }
    return BX_ERR_ARG;
}

int read_any_impl(const char* path, float* out, int64_t cap, int64_t* n_out)
{
    if (!path || !n_out) { bx_set_error("bx_io: null argument"); return BX_ERR_ARG; }
    FileBuf fb;
    if (!fb.load(path, out ? ~(size_t)0 : (size_t)65536)) return BX_ERR_ARG;
    const std::string p(path);
    if (ends_with(p, ".ply")) return read_ply(fb, path, out, cap, n_out);
    if (ends_with(p, ".pcd")) return read_pcd(fb, path, out, cap, n_out);
    if (ends_with(p, ".bin")) return read_bin(fb, path, out, cap, n_out);
    bx_set_error("bx_io: %s: unknown extension (.ply, .pcd, .bin)", path);
    return BX_ERR_ARG;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- prefetcher
struct bx_prefetch {
    struct Slot {
        float *h_src = nullptr, *h_tgt = nullptr;     // pinned
        float *d_src = nullptr, *d_tgt = nullptr;
        int64_t n_src = 0, n_tgt = 0;
        hipEvent_t uploaded = nullptr, consumed = nullptr;
        int64_t ticket = -1;
        int state = 0;        // 0 free, 1 queued, 2 uploaded (or failed), 3 handed to the consumer
        int rc = BX_OK;
        bool consumed_valid = false;
        std::string src, tgt, err;
    };
    int device = 0;
    int64_t max_points = 0;
    std::vector<Slot> slots;
    std::deque<int> queue;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::thread worker;
    hipStream_t copy_stream = nullptr;
    int64_t next_ticket = 0;
    bool stop = false;

    void run()
    {
        (void)hipSetDevice(device);
        for (;;) {
            int si;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !queue.empty(); });
                if (stop && queue.empty()) return;
                si = queue.front();
                queue.pop_front();
            }
            Slot& s = slots[(size_t)si];
            int rc = BX_OK;
            if (s.consumed_valid) (void)hipEventSynchronize(s.consumed);   // the previous user of these device buffers is done
            int64_t ns = 0, nt = 0;
            rc = read_any(s.src.c_str(), s.h_src, max_points, &ns);
            if (rc == BX_OK) rc = read_any(s.tgt.c_str(), s.h_tgt, max_points, &nt);
            std::string err;
            if (rc == BX_OK) {
                hipError_t e = hipMemcpyAsync(s.d_src, s.h_src, sizeof(float) * 3 * (size_t)ns, hipMemcpyHostToDevice, copy_stream);
                if (e == hipSuccess) e = hipMemcpyAsync(s.d_tgt, s.h_tgt, sizeof(float) * 3 * (size_t)nt, hipMemcpyHostToDevice, copy_stream);
                if (e == hipSuccess) e = hipEventRecord(s.uploaded, copy_stream);
                if (e != hipSuccess) { rc = BX_ERR_HIP; err = hipGetErrorString(e); }
            } else {
                err = bx_last_error();     // thread-local message of the reader
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                s.n_src = ns; s.n_tgt = nt; s.rc = rc; s.err = err; s.state = 2;
            }
            cv_done.notify_all();
        }
    }
};

extern "C" {

int bx_io_probe(const char* path, int64_t* n_points) { return read_any(path, nullptr, 0, n_points); }

int bx_io_read_xyz(const char* path, float* xyz_out, int64_t capacity, int64_t* n_points)
{
    if (!xyz_out) { bx_set_error("bx_io_read_xyz: null output buffer"); return BX_ERR_ARG; }
    return read_any(path, xyz_out, capacity, n_points);
}

int bx_prefetch_create(int32_t device, int32_t slots, int64_t max_points, bx_prefetch** out)
{
    if (!out || slots < 1 || slots > 64 || max_points < 1) { bx_set_error("bx_prefetch_create: bad argument"); return BX_ERR_ARG; }
    BxDevScope ds(device);
    bx_prefetch* p = new bx_prefetch();
    p->device = device; p->max_points = max_points;
    p->slots.resize((size_t)slots);
    const size_t bytes = sizeof(float) * 3 * (size_t)max_points;
    hipError_t e = hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking);
    for (auto& s : p->slots) {
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&s.h_src), bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&s.h_tgt), bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s.d_src), bytes);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s.d_tgt), bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        bx_set_error("bx_prefetch_create: %s", hipGetErrorString(e));
        for (auto& s : p->slots) { (void)hipHostFree(s.h_src); (void)hipHostFree(s.h_tgt); (void)hipFree(s.d_src); (void)hipFree(s.d_tgt); }
        delete p;
        return BX_ERR_HIP;
    }
    p->worker = std::thread([p] { p->run(); });
    *out = p;
    return BX_OK;
}

int bx_prefetch_submit(bx_prefetch* p, const char* src_path, const char* tgt_path, int64_t* ticket)
{
    if (!p || !src_path || !tgt_path || !ticket) { bx_set_error("bx_prefetch_submit: null argument"); return BX_ERR_ARG; }
    std::lock_guard<std::mutex> lk(p->mu);
    for (size_t i = 0; i < p->slots.size(); ++i)
        if (p->slots[i].state == 0) {
            auto& s = p->slots[i];
            s.state = 1; s.src = src_path; s.tgt = tgt_path; s.ticket = p->next_ticket++;
            *ticket = s.ticket;
            p->queue.push_back((int)i);
            p->cv_work.notify_one();
            return BX_OK;
        }
    bx_set_error("bx_prefetch_submit: all %zu slots are in use (release a ticket first)", p->slots.size());
    return BX_ERR_STATE;
}

int bx_prefetch_wait(bx_prefetch* p, int64_t ticket, void* stream, const float** src_dev, int64_t* n_src, const float** tgt_dev,
                     int64_t* n_tgt)
{
    if (!p || !src_dev || !n_src || !tgt_dev || !n_tgt) { bx_set_error("bx_prefetch_wait: null argument"); return BX_ERR_ARG; }
    std::unique_lock<std::mutex> lk(p->mu);
    bx_prefetch::Slot* s = nullptr;
    for (auto& c : p->slots) if (c.ticket == ticket && c.state != 0) s = &c;
    if (!s) { bx_set_error("bx_prefetch_wait: unknown ticket %lld", (long long)ticket); return BX_ERR_ARG; }
    p->cv_done.wait(lk, [&] { return s->state >= 2; });
    if (s->rc != BX_OK) { bx_set_error("bx_prefetch: %s", s->err.c_str()); const int rc = s->rc; s->state = 0; s->ticket = -1; return rc; }
    s->state = 3;
    lk.unlock();
    BX_HIP(hipStreamWaitEvent((hipStream_t)stream, s->uploaded, 0));      // the consumer stream waits for the DMA, not the host
    *src_dev = s->d_src; *n_src = s->n_src; *tgt_dev = s->d_tgt; *n_tgt = s->n_tgt;
    return BX_OK;
}

int bx_prefetch_release(bx_prefetch* p, int64_t ticket, void* stream)
{
    if (!p) { bx_set_error("bx_prefetch_release: null argument"); return BX_ERR_ARG; }
    std::lock_guard<std::mutex> lk(p->mu);
    for (auto& s : p->slots)
        if (s.ticket == ticket && s.state == 3) {
            // the slot's device buffers may be overwritten once everything queued on `stream` so far has run
            BX_HIP(hipEventRecord(s.consumed, (hipStream_t)stream));
            s.consumed_valid = true;
            s.state = 0; s.ticket = -1;
            return BX_OK;
        }
    bx_set_error("bx_prefetch_release: ticket %lld is not handed out", (long long)ticket);
    return BX_ERR_ARG;
}

int bx_prefetch_destroy(bx_prefetch* p)
{
    if (!p) return BX_OK;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_work.notify_all();
    if (p->worker.joinable()) p->worker.join();
    BxDevScope ds(p->device);
    (void)hipStreamSynchronize(p->copy_stream);
    for (auto& s : p->slots) {
        if (s.consumed_valid) (void)hipEventSynchronize(s.consumed);
        (void)hipHostFree(s.h_src); (void)hipHostFree(s.h_tgt); (void)hipFree(s.d_src); (void)hipFree(s.d_tgt);
        (void)hipEventDestroy(s.uploaded); (void)hipEventDestroy(s.consumed);
    }
    (void)hipStreamDestroy(p->copy_stream);
    delete p;
    return BX_OK;
}

}  // extern "C"
