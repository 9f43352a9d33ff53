// htk.cpp -- see htk.h
#include "htk.h"

#include <cstdint>
#include <cstdio>
#include <cstring>

namespace phnrec {

namespace {
inline uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
inline uint16_t bswap16(uint16_t v) { return __builtin_bswap16(v); }
}  // namespace

namespace {
bool write_header(FILE *f, int rows, int cols)
{
    unsigned char hdr[12];
    const uint32_t n = bswap32((uint32_t)rows), per = bswap32(100000u);   // defaults matrix.h:411-423
    const uint16_t sz = bswap16((uint16_t)(cols * 4)), kind = bswap16(6);
    memcpy(hdr, &n, 4); memcpy(hdr + 4, &per, 4); memcpy(hdr + 8, &sz, 2); memcpy(hdr + 10, &kind, 2);
    return fwrite(hdr, 1, 12, f) == 12;
}
}  // namespace

bool SaveHTKRaw(const std::string &path, const void *be_words, int rows, int cols)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const size_t n = (size_t)rows * cols;
    const bool ok = write_header(f, rows, cols) && fwrite(be_words, 4, n, f) == n;
    return fclose(f) == 0 && ok;
}

FILE *BeginHTKRaw(const std::string &path, int rows, int cols)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (f && !write_header(f, rows, cols)) { fclose(f); f = nullptr; }
    return f;
}

bool AppendHTKRaw(FILE *f, const void *be_words, int rows, int cols)
{
    const size_t n = (size_t)rows * cols;
    return fwrite(be_words, 4, n, f) == n;
}

bool SaveHTK(const std::string &path, const float *data, int rows, int cols)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = write_header(f, rows, cols);
    std::vector<uint32_t> buf((size_t)rows * cols);
    for (size_t i = 0; i < buf.size(); i++) {
        uint32_t v;
        memcpy(&v, &data[i], 4);
        buf[i] = bswap32(v);
    }
    ok = ok && fwrite(buf.data(), 4, buf.size(), f) == buf.size();
    fclose(f);
    return ok;
}

bool LoadHTK(const std::string &path, std::vector<float> &data, int *rows, int *cols)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char hdr[12];
    if (fread(hdr, 1, 12, f) != 12) { fclose(f); return false; }
    uint32_t n; uint16_t sz;
    memcpy(&n, hdr, 4); memcpy(&sz, hdr + 8, 2);
    *rows = (int)bswap32(n);
    *cols = (int)bswap16(sz) / 4;
    if (*rows < 0 || *cols <= 0) { fclose(f); return false; }
    std::vector<uint32_t> buf((size_t)*rows * *cols);
    const size_t got = fread(buf.data(), 4, buf.size(), f);
    fclose(f);
    if (got != buf.size()) return false;
    data.resize(buf.size());
    for (size_t i = 0; i < buf.size(); i++) {
        const uint32_t v = bswap32(buf[i]);
        memcpy(&data[i], &v, 4);
    }
    return true;
}

}  // namespace phnrec
