// nnet_io.h -- host-side readers/writers for PhnRec's model files.
//
// File formats follow the reference (they are the on-disk contract of the
// drop-in): .nbin nn.cpp:464-592, ASCII weights/norms nn.cpp:116-412, the
// .nbin-preferred lookup of NeuralNet::Load nn.cpp:594-621, half-context
// windows traps.cpp:549-570.  In memory the nets are kept UNPADDED; the HIP
// side re-packs them into MFMA fragment order (lcrc_pack.cpp).
#ifndef PHNREC_NNET_IO_H
#define PHNREC_NNET_IO_H

#include <string>
#include <vector>

namespace phnrec {

// Status codes mirror nn.h:35-42 so callers can report what the reference would.
enum NetStatus {
    NET_OK = 0, NET_NOWEIGHTS = 1, NET_BADWEIGHTS = 2, NET_NONORMS = 3, NET_BADNORMS = 4,
    NET_MEMORY = 5, NET_CREATEERR = 6, NET_WRITEERR = 7
};

struct HostNet {
    int n_inp = 0, n_hid = 0, n_out = 0;
    std::vector<float> w1;    // [n_hid][n_inp]
    std::vector<float> w2;    // [n_out][n_hid]
    std::vector<float> b1;    // [n_hid]
    std::vector<float> b2;    // [n_out]
    std::vector<float> mean;  // [n_inp]
    std::vector<float> dev;   // [n_inp]  multiplier (1/sigma)
    bool has_norms = false;
};

inline int pad4(int n) { return (n + 3) & ~3; }

NetStatus load_nbin(const std::string &path, HostNet &net);
NetStatus save_nbin(const std::string &path, const HostNet &net);
NetStatus load_ascii(const std::string &weights, const std::string &norms, HostNet &net);
// NeuralNet::Load: <weights without suffix>.nbin if readable, else ASCII and
// (best effort, like the reference) cache the .nbin next to it.
NetStatus load_net(const std::string &weights, const std::string &norms, HostNet &net,
                   bool write_cache = true);
// 16 whitespace-separated floats.
bool load_window(const std::string &path, int len, std::vector<float> &win);

const char *net_status_str(NetStatus s);

}  // namespace phnrec
#endif
