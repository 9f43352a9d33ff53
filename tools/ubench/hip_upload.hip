// hip_upload.hip -- how a fresh process gets ~6.6 MB of packed weights onto the device fastest (dev aid, see
// lcrc_api.cpp upload_net): pageable hipMemcpy, a second one, registered memory, pinned staging.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv)
{
    using clk = std::chrono::steady_clock;
    auto t = clk::now();
    auto mark = [&](const char *what) {
        const auto now = clk::now();
        printf("%-52s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    };
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t n = 6600000;
    std::vector<char> h(n, 1), h2(n, 2);
    hipFree(nullptr);                                    mark("HIP up");
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);  mark("stream");
    char *d = nullptr, *d2 = nullptr;
    hipMalloc(&d, n); hipMalloc(&d2, n);                 mark("2 x hipMalloc 6.6 MB");
    if (mode == 0) {
        hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);    mark("mode 0: hipMemcpy pageable (first)");
        hipMemcpy(d2, h2.data(), n, hipMemcpyHostToDevice);  mark("        hipMemcpy pageable (second)");
    } else if (mode == 1) {
        hipHostRegister(h.data(), n, hipHostRegisterDefault);  mark("mode 1: hipHostRegister 6.6 MB");
        hipMemcpyAsync(d, h.data(), n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);  mark("        hipMemcpyAsync registered + sync");
        hipHostUnregister(h.data());                          mark("        hipHostUnregister");
    } else if (mode == 2) {
        char *p = nullptr;
        hipHostMalloc(&p, n, hipHostMallocDefault);           mark("mode 2: hipHostMalloc 6.6 MB");
        memcpy(p, h.data(), n);                               mark("        memcpy into pinned");
        hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);  mark("        hipMemcpyAsync pinned + sync");
    } else if (mode == 3) {
        for (int k = 0; k < 3; k++) { hipMemcpy(d + k * 2200000, h.data() + k * 2200000, 2200000, hipMemcpyHostToDevice); }
        mark("mode 3: 3 x hipMemcpy pageable 2.2 MB");
    } else if (mode == 4) {
        hipMemcpyAsync(d, h.data(), n, hipMemcpyHostToDevice, s); mark("mode 4: hipMemcpyAsync pageable (returns)");
        hipStreamSynchronize(s);                              mark("        sync");
    }
    fflush(stdout);
    _Exit(0);
}
