// Does this box honour kernel-argument preload into user SGPRs (gfx940+; hipcc -mllvm -amdgpu-kernarg-preload-count=N, leading
// scalar parameters only)?  Build twice:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 kernarg_preload.hip -o kernarg_preload_off
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=14 kernarg_preload.hip -o kernarg_preload_on
// A hipGraph chain alternates an L2-evicting kernel (reads 96 MB) with a one-workgroup test kernel whose only work depends on
// its arguments; the chain time per pair, minus the same chain with an argument-free test kernel, is what the argument fetch
// costs on a cold L2.  If preload works the two builds differ by the cold kernarg fetch (~0.5 us).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_evict(const float4* __restrict__ p, float* sink, long n) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.w; }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_args(int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7, int a8, int a9, int a10, int a11, float* out) {
    const int s = a0 + a1 * 3 + a2 * 5 + a3 * 7 + a4 * 11 + a5 * 13 + a6 * 17 + a7 * 19 + a8 * 23 + a9 * 29 + a10 * 31 + a11 * 37;
    if (threadIdx.x == 0) out[blockIdx.x] = (float)s;
}
__global__ void k_noargs() {}

static double chain(int mode, const float4* buf, float* sink, long n4, float* out) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    const int N = 100;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(k_evict, dim3(2048), dim3(256), 0, s, buf, sink, n4);
        if (mode == 0) hipLaunchKernelGGL(k_noargs, dim3(1), dim3(64), 0, s);
        else hipLaunchKernelGGL(k_args, dim3(1), dim3(64), 0, s, i, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, out);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int r = 0; r < 7; ++r) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best * 1e3 / N;
}

int main() {
    const long bytes = 96l << 20, n4 = bytes / 16;
    float4* buf; float *sink, *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&out, 4096));
    const double t0 = chain(0, buf, sink, n4, out), t1 = chain(1, buf, sink, n4, out);
    printf("pair (evict 96 MB + test kernel): no-argument kernel %.3f us, 13-argument kernel %.3f us -> argument cost %.3f us\n", t0, t1, t1 - t0);
    return 0;
}
