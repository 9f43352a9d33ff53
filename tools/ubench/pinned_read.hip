// How fast do host threads read what the GPU has copied into pinned memory?  The host Viterbi of a list run reads
// 4 * nOut bytes per frame (744 B for HU) out of the context's pinned posterior buffer right after the D2H copy.
// Compares, per thread count: pinned (hipHostMalloc portable) filled by a D2H copy, the same after a CPU memcpy into
// pageable memory, and pageable memory written by the CPU.  ./pinned_read [MiB] [threads]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double read_all(const float *p, size_t n, int threads)
{
    std::vector<std::thread> th;
    std::vector<double> sums(threads);
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            const size_t a = n * t / threads, b = n * (t + 1) / threads;
            double s = 0;
            for (size_t i = a; i < b; i += 16) s += p[i];       // one value per cache line
            sums[t] = s;
        });
    for (auto &x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double s = 0;
    for (double v : sums) s += v;
    if (s == 12345.678) printf("!");
    return dt;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? atoi(argv[1]) : 256;
    const int threads = argc > 2 ? atoi(argv[2]) : 16;
    const size_t n = mib * 1024 * 1024 / 4;
    float *d = nullptr, *h = nullptr, *hm = nullptr;
    hipMalloc((void **)&d, n * 4);
    hipMemset(d, 0x3c, n * 4);
    for (unsigned flags : {(unsigned)hipHostMallocPortable, (unsigned)(hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent),
                           (unsigned)hipHostMallocNonCoherent}) {
        if (hipHostMalloc((void **)&h, n * 4, flags) != hipSuccess) { printf("flags %u: alloc failed\n", flags); continue; }
        memset(h, 0, n * 4);
        for (int rep = 0; rep < 3; rep++) {
            hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
            const double t1 = read_all(h, n, threads), t2 = read_all(h, n, threads);
            printf("flags 0x%x: pinned after D2H: %.1f GB/s (first read), %.1f GB/s (again), %d threads\n", flags,
                   n * 4 / t1 / 1e9, n * 4 / t2 / 1e9, threads);
        }
        hipHostFree(h);
    }
    std::vector<float> pg(n, 1.0f);
    hm = pg.data();
    const double t3 = read_all(hm, n, threads), t4 = read_all(hm, n, threads);
    printf("pageable: %.1f GB/s, %.1f GB/s\n", n * 4 / t3 / 1e9, n * 4 / t4 / 1e9);
    hipFree(d);
    return 0;
}
