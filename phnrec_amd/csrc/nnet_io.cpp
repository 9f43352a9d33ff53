// nnet_io.cpp -- see nnet_io.h
#include "nnet_io.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace phnrec {

const char *net_status_str(NetStatus s)
{
    switch (s) {
    case NET_OK: return "ok";
    case NET_NOWEIGHTS: return "weight file not found";
    case NET_BADWEIGHTS: return "malformed weight file";
    case NET_NONORMS: return "norm file not found";
    case NET_BADNORMS: return "malformed norm file";
    case NET_MEMORY: return "out of memory";
    case NET_CREATEERR: return "cannot create file";
    case NET_WRITEERR: return "short read/write";
    }
    return "?";
}

namespace {

struct File {
    FILE *f;
    explicit File(const char *p, const char *m) : f(fopen(p, m)) {}
    ~File() { if (f) fclose(f); }
};

// Reads `rows` rows of `cols` floats out of a file that stores them with a
// row stride of `stride` floats (the reference pads every dimension to x4).
bool read_strided(FILE *f, std::vector<float> &dst, int rows, int cols, int stride)
{
    std::vector<float> row(stride);
    dst.resize((size_t)rows * cols);
    for (int r = 0; r < rows; r++) {
        if (fread(row.data(), sizeof(float), stride, f) != (size_t)stride) return false;
        memcpy(&dst[(size_t)r * cols], row.data(), sizeof(float) * cols);
    }
    return true;
}

bool skip_floats(FILE *f, long n) { return fseek(f, n * (long)sizeof(float), SEEK_CUR) == 0; }

bool write_strided(FILE *f, const std::vector<float> &src, int rows, int cols, int stride,
                   int total_rows, float pad)
{
    std::vector<float> row(stride);
    for (int r = 0; r < total_rows; r++) {
        for (int c = 0; c < stride; c++)
            row[c] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : pad;
        if (fwrite(row.data(), sizeof(float), stride, f) != (size_t)stride) return false;
    }
    return true;
}

}  // namespace

// Layout (nn.cpp:464-531): int32 nlayers(=2); int32 nInp,nHid,nOut; then
// W1[nHid16][nInp16] W2[nOut16][nHid16] b1[nHid16] b2[nOut16] mean[nInp16] dev[nInp16]
// with X16 = X rounded up to 4 floats; host (little-endian) byte order.
NetStatus load_nbin(const std::string &path, HostNet &net)
{
    File fp(path.c_str(), "rb");
    if (!fp.f) return NET_NOWEIGHTS;
    int32_t nl = 0, sz[3] = {0, 0, 0};
    if (fread(&nl, 4, 1, fp.f) != 1 || nl != 2) return NET_BADWEIGHTS;
    if (fread(sz, 4, 3, fp.f) != 3) return NET_WRITEERR;
    if (sz[0] <= 0 || sz[1] <= 0 || sz[2] <= 0 || sz[0] > (1 << 20) || sz[1] > (1 << 20) ||
        sz[2] > (1 << 20))
        return NET_BADWEIGHTS;
    net.n_inp = sz[0]; net.n_hid = sz[1]; net.n_out = sz[2];
    const int i16 = pad4(net.n_inp), h16 = pad4(net.n_hid), o16 = pad4(net.n_out);
    // W1 has nHid16 rows in the file (nn.cpp:497), W2 nOut16 rows
    if (!read_strided(fp.f, net.w1, net.n_hid, net.n_inp, i16)) return NET_WRITEERR;
    if (!skip_floats(fp.f, (long)(h16 - net.n_hid) * i16)) return NET_WRITEERR;
    if (!read_strided(fp.f, net.w2, net.n_out, net.n_hid, h16)) return NET_WRITEERR;
    if (!skip_floats(fp.f, (long)(o16 - net.n_out) * h16)) return NET_WRITEERR;
    if (!read_strided(fp.f, net.b1, 1, net.n_hid, h16)) return NET_WRITEERR;
    if (!read_strided(fp.f, net.b2, 1, net.n_out, o16)) return NET_WRITEERR;
    if (!read_strided(fp.f, net.mean, 1, net.n_inp, i16)) return NET_WRITEERR;
    if (!read_strided(fp.f, net.dev, 1, net.n_inp, i16)) return NET_WRITEERR;
    net.has_norms = true;
    return NET_OK;
}

NetStatus save_nbin(const std::string &path, const HostNet &net)
{
    File fp(path.c_str(), "wb");
    if (!fp.f) return NET_CREATEERR;
    const int i16 = pad4(net.n_inp), h16 = pad4(net.n_hid), o16 = pad4(net.n_out);
    int32_t hdr[4] = {2, net.n_inp, net.n_hid, net.n_out};
    if (fwrite(hdr, 4, 4, fp.f) != 4) return NET_WRITEERR;
    std::vector<float> mean = net.mean, dev = net.dev;
    if (!net.has_norms) { mean.assign(net.n_inp, 0.0f); dev.assign(net.n_inp, 1.0f); }
    bool ok = write_strided(fp.f, net.w1, net.n_hid, net.n_inp, i16, h16, 0.0f) &&
              write_strided(fp.f, net.w2, net.n_out, net.n_hid, h16, o16, 0.0f) &&
              write_strided(fp.f, net.b1, 1, net.n_hid, h16, 1, 0.0f) &&
              write_strided(fp.f, net.b2, 1, net.n_out, o16, 1, 0.0f) &&
              write_strided(fp.f, mean, 1, net.n_inp, i16, 1, 0.0f) &&
              write_strided(fp.f, dev, 1, net.n_inp, i16, 1, 1.0f);   // dev pad = 1 (nn.cpp:344-348)
    return ok ? NET_OK : NET_WRITEERR;
}

namespace {

// Whitespace-token reader over a whole text file.
struct Tokens {
    std::string text;
    size_t pos = 0;
    bool load(const std::string &path)
    {
        std::ifstream in(path.c_str(), std::ios::binary);
        if (!in) return false;
        std::ostringstream ss;
        ss << in.rdbuf();
        text = ss.str();
        return true;
    }
    bool next(const char *&b, size_t &len)
    {
        while (pos < text.size() && strchr(" \t\n\r", text[pos])) pos++;
        if (pos >= text.size()) return false;
        size_t s = pos;
        while (pos < text.size() && !strchr(" \t\n\r", text[pos])) pos++;
        b = text.data() + s;
        len = pos - s;
        return true;
    }
    bool header(const char *kw, long &count)
    {
        const char *b; size_t n;
        if (!next(b, n) || n < strlen(kw) || strncmp(b, kw, strlen(kw)) != 0) return false;
        if (!next(b, n)) return false;
        char buf[64];
        if (n >= sizeof buf) return false;
        memcpy(buf, b, n); buf[n] = 0;
        int v;
        if (sscanf(buf, "%d", &v) != 1) return false;
        count = v;
        return true;
    }
    bool skip(long n)
    {
        const char *b; size_t len;
        for (long i = 0; i < n; i++) if (!next(b, len)) return false;
        return true;
    }
    bool floats(std::vector<float> &dst, long n)
    {
        dst.resize((size_t)n);
        const char *b; size_t len;
        char buf[128];
        for (long i = 0; i < n; i++) {
            if (!next(b, len) || len >= sizeof buf) return false;
            memcpy(buf, b, len); buf[len] = 0;
            if (sscanf(buf, "%e", &dst[(size_t)i]) != 1) return false;   // nn.cpp:953-966
        }
        return true;
    }
};

}  // namespace

// "weigvec <nHid*nInp>" values (row = hidden unit), "weigvec <nOut*nHid>" values
// (row = output unit), "biasvec <nHid>", "biasvec <nOut>"; nInp = n1 / nHid
// (nn.cpp:116-197).  Norms: "vec <nInp>" means, "vec <nInp>" multipliers.
NetStatus load_ascii(const std::string &weights, const std::string &norms, HostNet &net)
{
    Tokens t;
    if (!t.load(weights)) return NET_NOWEIGHTS;
    long n1, n2, nb1, nb2;
    if (!t.header("weigvec", n1) || !t.skip(n1) || !t.header("weigvec", n2) || !t.skip(n2) ||
        !t.header("biasvec", nb1) || !t.skip(nb1) || !t.header("biasvec", nb2) || !t.skip(nb2))
        return NET_BADWEIGHTS;
    if (nb1 <= 0 || nb2 <= 0 || n1 <= 0) return NET_BADWEIGHTS;
    net.n_out = (int)nb2; net.n_hid = (int)nb1; net.n_inp = (int)(n1 / nb1);
    if (net.n_inp <= 0) return NET_BADWEIGHTS;
    t.pos = 0;
    long c;
    if (!t.header("weigvec", c) || !t.floats(net.w1, (long)net.n_hid * net.n_inp)) return NET_BADWEIGHTS;
    if (!t.header("weigvec", c) || !t.floats(net.w2, (long)net.n_out * net.n_hid)) return NET_BADWEIGHTS;
    if (!t.header("biasvec", c) || !t.floats(net.b1, net.n_hid)) return NET_BADWEIGHTS;
    if (!t.header("biasvec", c) || !t.floats(net.b2, net.n_out)) return NET_BADWEIGHTS;
    net.mean.assign(net.n_inp, 0.0f);
    net.dev.assign(net.n_inp, 1.0f);
    net.has_norms = false;
    if (!norms.empty()) {
        Tokens n;
        if (!n.load(norms)) return NET_NONORMS;
        if (!n.header("vec", c) || !n.floats(net.mean, net.n_inp) || !n.header("vec", c) ||
            !n.floats(net.dev, net.n_inp))
            return NET_BADNORMS;
        net.has_norms = true;
    }
    return NET_OK;
}

NetStatus load_net(const std::string &weights, const std::string &norms, HostNet &net,
                   bool write_cache)
{
    std::string bin = weights;
    size_t dot = bin.rfind('.'), slash = bin.find_last_of("/\\");
    if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) bin.erase(dot);
    bin += ".nbin";
    const NetStatus sb = load_nbin(bin, net);
    if (sb == NET_OK) return NET_OK;
    NetStatus s = load_ascii(weights, norms, net);
    // a .nbin that exists but is damaged, with no ASCII to fall back to: report the damage
    if (s == NET_NOWEIGHTS && sb != NET_NOWEIGHTS) return sb == NET_WRITEERR ? NET_BADWEIGHTS : sb;
    if (s == NET_OK && write_cache) save_nbin(bin, net);   // failure ignored, as nn.cpp:613-618
    return s;
}

bool load_window(const std::string &path, int len, std::vector<float> &win)
{
    File fp(path.c_str(), "r");
    if (!fp.f) return false;
    win.resize(len);
    for (int i = 0; i < len; i++)
        if (fscanf(fp.f, "%f", &win[i]) != 1) return false;
    return true;
}

}  // namespace phnrec
