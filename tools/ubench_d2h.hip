// ubench_d2h.hip -- how should rtw_render_* bring a 24.9 MB frame (1920x1080 RGB{Float32}) from HBM into the caller's
// PAGEABLE buffer?  (a) one hipMemcpy into the pageable buffer; (b) D2H into a persistent pinned buffer + one memcpy;
// (c) chunked: D2H of chunk k into pinned double buffers overlapped with the memcpy of chunk k - 1; (d) like (c) with the
// memcpy split over T host threads.  Prints ms per frame (median of 9).
// MI355X box, round 3: (a) 0.447 ms  (b) 1.279  (c) 0.93 - 1.04 with one thread, 0.59 - 0.66 with 2 - 4 threads; host memcpy alone 0.86.
// -> rtw_hip.hip copy_out() uses (a).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = (size_t)1920 * 1080 * 3 * 4;
    char *d, *pin, *page = (char *)malloc(bytes);
    memset(page, 1, bytes);
    (void)hipMalloc(&d, bytes); (void)hipMemset(d, 7, bytes);
    (void)hipHostMalloc((void **)&pin, bytes, hipHostMallocDefault);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::vector<double> t;
    for (int i = 0; i < 9; ++i) { double t0 = now(); (void)hipMemcpyAsync(page, d, bytes, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); t.push_back(now() - t0); }
    printf("(a) hipMemcpyAsync into pageable memory          %.3f ms\n", med(t)); t.clear();
    for (int i = 0; i < 9; ++i) { double t0 = now(); (void)hipMemcpyAsync(pin, d, bytes, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); double t1 = now(); memcpy(page, pin, bytes); t.push_back(now() - t0); if (i == 8) printf("    (D2H into pinned alone %.3f ms)\n", t1 - t0); }
    printf("(b) D2H into pinned + one memcpy                 %.3f ms\n", med(t)); t.clear();
    for (size_t chunk : {1u << 20, 2u << 20, 4u << 20, 8u << 20}) {
        for (int nthr : {1, 2, 4}) {
            hipEvent_t ev[64];
            const int nch = (int)((bytes + chunk - 1) / chunk);
            for (int k = 0; k < nch; ++k) (void)hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
            for (int i = 0; i < 9; ++i) {
                double t0 = now();
                for (int k = 0; k < nch; ++k) {
                    const size_t off = (size_t)k * chunk, n = std::min(chunk, bytes - off);
                    (void)hipMemcpyAsync(pin + off, d + off, n, hipMemcpyDeviceToHost, s);
                    (void)hipEventRecord(ev[k], s);
                }
                auto worker = [&](int tid) {
                    for (int k = 0; k < nch; ++k) {
                        const size_t off = (size_t)k * chunk, n = std::min(chunk, bytes - off);
                        if (tid == 0) (void)hipEventSynchronize(ev[k]);
                        else while (hipEventQuery(ev[k]) != hipSuccess) {}
                        const size_t part = (n + nthr - 1) / nthr, a = std::min(n, part * tid), b = std::min(n, part * (tid + 1));
                        memcpy(page + off + a, pin + off + a, b - a);
                    }
                };
                std::vector<std::thread> th;
                for (int q = 1; q < nthr; ++q) th.emplace_back(worker, q);
                worker(0);
                for (auto &x : th) x.join();
                t.push_back(now() - t0);
            }
            printf("(c) chunks of %zu MB, %d memcpy thread(s)          %.3f ms\n", chunk >> 20, nthr, med(t)); t.clear();
        }
    }
    double t0 = now(); memcpy(page, pin, bytes); printf("    (host memcpy alone %.3f ms)\n", now() - t0);
    return page[5] == 7 ? 0 : 1;
}
