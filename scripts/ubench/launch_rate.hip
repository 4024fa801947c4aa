// How many DEPENDENT small kernels per millisecond does the chip retire over several streams?  S host threads, one stream
// each, N launches per stream of a kernel with G workgroups that spins for ~D microseconds (D = 0: empty).
// hipcc --offload-arch=gfx950 -O3 -o launch_rate launch_rate.hip -lpthread;  ./launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void spin_kernel(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 1024) *sink = 1;
}

static double run(int S, int N, int G, double us) {
    std::vector<hipStream_t> st(S);
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);          // (numerically lowest = highest priority)
    for (int i = 0; i < S; ++i) {
        if (getenv("PRIO")) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi + i % (lo - hi + 1));
        else hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    }
    const unsigned long long ticks = (unsigned long long)(us * 100.0);      // 100 MHz counter
    auto work = [&](int i) {
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(spin_kernel, dim3(G), dim3(256), 0, st[i], ticks, nullptr);
        hipStreamSynchronize(st[i]);
    };
    for (int i = 0; i < S; ++i) work(i);          // warm-up
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int i = 0; i < S; ++i) th.emplace_back(work, i);
    for (auto& t : th) t.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (auto& s : st) hipStreamDestroy(s);
    return S * (double)N / dt * 1e-3;             // kernels per millisecond
}

int main() {
    const int N = 2000;
    if (getenv("QUICK")) {
        for (int S : {1, 2, 3, 4, 5, 6, 8, 12, 16, 24})
            printf("spin 8 us, 16 workgroups, %2d streams: %7.1f kernels/ms overall\n", S, run(S, N, 16, 8.0));
        return 0;
    }
    for (double us : {0.0, 3.0, 8.0})
        for (int G : {1, 16, 64})
            for (int S : {1, 2, 4, 8, 16}) {
                const double r = run(S, N, G, us);
                printf("spin %.0f us, %2d workgroups, %2d streams: %7.1f kernels/ms overall (%.2f us per kernel and stream)\n", us, G, S, r, S / r * 1e3);
                fflush(stdout);
            }
    return 0;
}
