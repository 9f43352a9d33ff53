// hip_startup.hip -- what a fresh process pays before its first kernel result on this box (dev aid for the
// single-file latency of the CLI, bench.py `single_file`): each HIP call of a context's creation, timed.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void touch(float *p) { p[threadIdx.x] = 1.0f; }

int main()
{
    using clk = std::chrono::steady_clock;
    auto t = clk::now();
    const auto t_start = t;
    auto mark = [&](const char *what) {
        const auto now = clk::now();
        printf("%-44s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    };
    int n = 0;
    hipInit(0);                                             mark("hipInit");
    hipGetDeviceCount(&n);                                  mark("hipGetDeviceCount");
    hipSetDevice(0);                                        mark("hipSetDevice(0)");
    hipFree(nullptr);                                       mark("hipFree(0) (primary context)");
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);     mark("hipStreamCreateWithFlags");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);               mark("2 x hipEventCreate");
    std::vector<float> h(1650000, 1.0f);                    // ~6.6 MB: one system's packed weights
    float *d = nullptr;
    hipMalloc(&d, h.size() * 4);                            mark("hipMalloc 6.6 MB");
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);   mark("hipMemcpy H2D 6.6 MB (pageable)");
    for (int i = 0; i < 18; i++) { float *q; hipMalloc(&q, 4096); hipMemcpy(q, h.data(), 4096, hipMemcpyHostToDevice); }
    mark("18 x (hipMalloc + hipMemcpy 4 KB)");
    float *hp = nullptr;
    hipHostMalloc(&hp, 4 << 20, hipHostMallocDefault);      mark("hipHostMalloc 4 MB");
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d);  mark("first kernel launch (code object load)");
    hipStreamSynchronize(s);                                mark("hipStreamSynchronize");
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d);
    hipStreamSynchronize(s);                                mark("second launch + sync");
    printf("%-44s %9.3f ms\n", "total since main()", std::chrono::duration<double, std::milli>(clk::now() - t_start).count());
    if (getenv("FAST_EXIT")) _Exit(0);
    return 0;
}
