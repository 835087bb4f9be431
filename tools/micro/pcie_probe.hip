// What the host-pointer entry points can hope for on this box: H2D / D2H rates from pageable, registered and pinned memory, both
// directions at once, the price of hipHostRegister, and the rate at which host threads fill a pinned staging buffer.
// hipcc -O2 -o /tmp/pcie_probe tools/micro/pcie_probe.hip -lpthread && /tmp/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = (size_t)400 << 20;  // 400 MB per array
    char *pg = (char *)malloc(N), *pg2 = (char *)malloc(N);
    memset(pg, 1, N), memset(pg2, 2, N);
    char *pin, *pin2, *dev, *dev2;
    CK(hipHostMalloc((void **)&pin, N, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&pin2, N, hipHostMallocDefault));
    CK(hipMalloc((void **)&dev, N));
    CK(hipMalloc((void **)&dev2, N));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto rate = [&](const char *what, auto fn) {
        fn();
        const double t0 = now();
        for (int i = 0; i < 3; i++) fn();
        const double dt = (now() - t0) / 3;
        printf("%-58s %7.2f ms  %6.1f GB/s\n", what, dt * 1e3, N / dt / 1e9);
    };
    rate("H2D pageable (hipMemcpyAsync + sync)", [&] { CK(hipMemcpyAsync(dev, pg, N, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
    rate("D2H pageable", [&] { CK(hipMemcpyAsync(pg, dev, N, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); });
    rate("H2D pinned", [&] { CK(hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
    rate("D2H pinned", [&] { CK(hipMemcpyAsync(pin, dev, N, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); });
    rate("H2D + D2H pinned at once (rate per direction)", [&] {
        CK(hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s1));
        CK(hipMemcpyAsync(pin2, dev2, N, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    });
    rate("H2D + D2H pageable at once, two host threads", [&] {
        std::thread a([&] { CK(hipMemcpyAsync(dev, pg, N, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
        std::thread b([&] { CK(hipMemcpyAsync(pg2, dev2, N, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); });
        a.join(), b.join();
    });
    {
        const double t0 = now();
        CK(hipHostRegister(pg, N, hipHostRegisterDefault));
        const double t1 = now();
        printf("%-58s %7.2f ms  %6.1f GB/s\n", "hipHostRegister of 400 MB", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
        rate("H2D registered", [&] { CK(hipMemcpyAsync(dev, pg, N, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
        const double t2 = now();
        CK(hipHostUnregister(pg));
        printf("%-58s %7.2f ms\n", "hipHostUnregister", (now() - t2) * 1e3);
    }
    for (int nt : {1, 2, 4, 8}) {
        char what[96];
        snprintf(what, sizeof what, "memcpy pageable -> pinned, %d host thread(s)", nt);
        rate(what, [&] {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++) th.emplace_back([&, t] { memcpy(pin + N / nt * t, pg2 + N / nt * t, N / nt); });
            for (auto &x : th) x.join();
        });
    }
    for (size_t chunk : {(size_t)8 << 20, (size_t)32 << 20}) {
        char what[96];
        snprintf(what, sizeof what, "H2D pageable in chunks of %zu MB, one stream", chunk >> 20);
        rate(what, [&] { for (size_t o = 0; o < N; o += chunk) CK(hipMemcpyAsync(dev + o, pg2 + o, chunk, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
    }
    printf("host threads available: %u\n", std::thread::hardware_concurrency());
    return 0;
}
