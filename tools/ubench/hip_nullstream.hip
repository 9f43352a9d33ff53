// hip_nullstream.hip -- does the default stream come cheaper than a created one?  (hip_startup.hip: hipStreamCreateWithFlags is
// 30 ms of a fresh process.)  Same steps as there, on stream 0 and without creating a stream.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void touch(float *p) { p[threadIdx.x] = 1.0f; }

int main(int argc, char **argv)
{
    using clk = std::chrono::steady_clock;
    auto t = clk::now();
    const auto t_start = t;
    auto mark = [&](const char *what) {
        const auto now = clk::now();
        printf("%-52s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    };
    const bool per_thread = argc > 1;                       // any argument: hipStreamPerThread instead of stream 0
    hipStream_t s = per_thread ? hipStreamPerThread : nullptr;
    hipInit(0);                                             mark("hipInit");
    hipSetDevice(0);
    hipFree(nullptr);                                       mark("hipSetDevice + hipFree(0) (primary context)");
    std::vector<float> h(1650000, 1.0f);
    float *d = nullptr;
    hipMalloc(&d, h.size() * 4);                            mark("hipMalloc 6.6 MB");
    hipMemcpyAsync(d, h.data(), h.size() * 4, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);                                mark(per_thread ? "H2D 6.6 MB on hipStreamPerThread + sync" : "H2D 6.6 MB on stream 0 + sync");
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d);  mark("first kernel launch");
    hipStreamSynchronize(s);                                mark("hipStreamSynchronize");
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, d);
    hipStreamSynchronize(s);                                mark("second launch + sync");
    printf("%-52s %9.3f ms\n", "total since main()", std::chrono::duration<double, std::milli>(clk::now() - t_start).count());
    fflush(stdout);
    _Exit(0);
}
