// pin_probe -- what pinned, device-mapped host memory costs when it is made, and when the process ends.
//   pin_probe MODE MiB N     MODE 0: nothing pinned; 1: N x hipHostMalloc(Mapped|Portable); 2: N x (mmap + MADV_HUGEPAGE +
//                            touch + hipHostRegister(Mapped|Portable)); 3: as 2 without the huge-page advice
// Prints the time of the allocations; the caller times the whole process (exit teardown = process - main).
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void touch(float *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    const size_t bytes = (size_t)(argc > 2 ? atoi(argv[2]) : 40) << 20;
    const int n = argc > 3 ? atoi(argv[3]) : 3;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    hipStream_t s;
    (void)hipStreamCreate(&s);
    const auto t1 = clk::now();
    for (int k = 0; k < n && mode > 0; k++) {
        void *h = nullptr;
        if (mode == 1) {
            if (hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
        } else {
            h = mmap(nullptr, bytes + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (h == MAP_FAILED) { printf("mmap failed\n"); return 1; }
            h = (void *)(((size_t)h + (2 << 20) - 1) & ~(size_t)((2 << 20) - 1));
            if (mode == 2) madvise(h, bytes, MADV_HUGEPAGE);
            memset(h, 0, bytes);
            if (hipHostRegister(h, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { printf("hipHostRegister failed\n"); return 1; }
        }
        float *d = nullptr;
        (void)hipHostGetDevicePointer((void **)&d, h, 0);
        touch<<<(unsigned)((bytes / 4 + 255) / 256), 256, 0, s>>>(d, bytes / 4);      // the device really maps and walks it
        (void)hipStreamSynchronize(s);
    }
    const auto t2 = clk::now();
    printf("mode %d  %d x %zu MiB: runtime up %.1f ms, buffers %.1f ms, main %.1f ms\n", mode, n, bytes >> 20,
           std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
           std::chrono::duration<double, std::milli>(t2 - t0).count());
    fflush(stdout);
    _Exit(0);
}
