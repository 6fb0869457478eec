// launch_cost.hip -- CPU time of one kernel launch by argument size / dynamic LDS / API (hipcc --offload-arch=gfx950 -O2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
struct Big { char b[208]; };
__global__ void k_empty() {}
__global__ void k_big(Big a) { if (a.b[0] == 77 && threadIdx.x == 9999) printf("x"); }
__global__ void k_ptr(const Big* a, int s) { if (a->b[0] == 77 && s == 12345678 && threadIdx.x == 9999) printf("x"); }
__global__ void k_lds(Big a) { extern __shared__ char sm[]; if (a.b[0] == 77 && threadIdx.x == 9999) sm[0] = 1; }
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }
template <typename F> static void run(const char* name, F f, hipStream_t st) {
    for (int i = 0; i < 200; i++) { f(); hipStreamSynchronize(st); }
    double tl = 0, tt = 0;
    const int N = 2000;
    for (int i = 0; i < N; i++) { const double a = now(); f(); const double b = now(); hipStreamSynchronize(st); const double c = now(); tl += b - a; tt += c - a; }
    printf("%-44s launch %.2f us, launch + synchronise %.2f us\n", name, tl / N, tt / N);
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    Big big = {}; Big* d_big; hipMalloc((void**)&d_big, sizeof big); hipMemcpy(d_big, &big, sizeof big, hipMemcpyHostToDevice);
    run("empty kernel, 1 group", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }, st);
    run("empty kernel, 150 groups x 1024", [&] { hipLaunchKernelGGL(k_empty, dim3(150), dim3(1024), 0, st); }, st);
    run("208-byte argument, 150 x 1024", [&] { hipLaunchKernelGGL(k_big, dim3(150), dim3(1024), 0, st, big); }, st);
    run("pointer + int argument, 150 x 1024", [&] { hipLaunchKernelGGL(k_ptr, dim3(150), dim3(1024), 0, st, (const Big*)d_big, 5); }, st);
    run("208-byte argument + 22 KB dynamic LDS", [&] { hipLaunchKernelGGL(k_lds, dim3(150), dim3(1024), 22528, st, big); }, st);
    void* args[] = {&big};
    run("hipLaunchKernel (args array), 208 B + LDS", [&] { hipLaunchKernel((const void*)k_lds, dim3(150), dim3(1024), args, 22528, st); }, st);
    return 0;
}
