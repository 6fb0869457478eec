// bar_probe.hip -- can the CPU write a frame straight into device memory (large BAR), and what does it cost?
// hipcc --offload-arch=gfx950 -O2 bar_probe.hip -o bar_probe
#include <hip/hip_runtime.h>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <vector>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void sum_kernel(const uint4* p, int n16, unsigned* out) {
    unsigned s = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) { uint4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    atomicAdd(out, s);
}
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
int main() {
    const size_t n = 115200;
    std::vector<unsigned char> src(n);
    for (size_t i = 0; i < n; i++) src[i] = (unsigned char)(i * 7 + 3);
    unsigned want = 0;
    for (size_t i = 0; i < n; i += 4) { unsigned w; memcpy(&w, &src[i], 4); want += w; }
    unsigned* d_out; hipMalloc((void**)&d_out, 4);
    struct { const char* name; int kind; } kinds[] = {{"hipMalloc", 0}, {"hipExtMallocWithFlags finegrained", 1}, {"hipExtMallocWithFlags uncached", 2}, {"hipMallocManaged", 3}};
    for (auto& k : kinds) {
        void* d = nullptr;
        hipError_t e = hipSuccess;
        if (k.kind == 0) e = hipMalloc(&d, n);
        else if (k.kind == 1) e = hipExtMallocWithFlags(&d, n, hipDeviceMallocFinegrained);
        else if (k.kind == 2) e = hipExtMallocWithFlags(&d, n, hipDeviceMallocUncached);
        else e = hipMallocManaged(&d, n);
        if (e != hipSuccess) { printf("%s: alloc failed %s\n", k.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        signal(SIGSEGV, on_segv); signal(SIGBUS, on_segv);
        if (sigsetjmp(jb, 1)) { printf("%s: CPU write faults\n", k.name); signal(SIGSEGV, SIG_DFL); continue; }
        double best = 1e9;
        unsigned got = 0;
        for (int rep = 0; rep < 20; rep++) {
            src[0] = (unsigned char)rep;
            const double t0 = now();
            memcpy(d, src.data(), n);
            __builtin_ia32_sfence();
            const double t1 = now();
            if (t1 - t0 < best) best = t1 - t0;
            hipMemset(d_out, 0, 4);
            sum_kernel<<<32, 256>>>((const uint4*)d, (int)(n / 16), d_out);
            hipMemcpy(&got, d_out, 4, hipMemcpyDeviceToHost);
            unsigned w2 = 0;
            for (size_t i = 0; i < n; i += 4) { unsigned w; memcpy(&w, &src[i], 4); w2 += w; }
            if (got != w2) { printf("%s: rep %d kernel saw stale data (%08x vs %08x)\n", k.name, rep, got, w2); break; }
        }
        printf("%s: CPU memcpy of %zu B into it: best %.2f us (%.1f GB/s); kernel sees the data: %s\n", k.name, n, best, n / best * 1e-3, "yes (unless noted)");
        signal(SIGSEGV, SIG_DFL); signal(SIGBUS, SIG_DFL);
    }
    // pinned host memory for comparison
    void* h; hipHostMalloc(&h, n, hipHostMallocMapped);
    double best = 1e9;
    for (int rep = 0; rep < 20; rep++) { const double t0 = now(); memcpy(h, src.data(), n); const double t1 = now(); if (t1 - t0 < best) best = t1 - t0; }
    printf("pinned host: CPU memcpy best %.2f us\n", best);
    (void)want;
    return 0;
}
