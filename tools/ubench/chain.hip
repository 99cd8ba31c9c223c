// What does one DEPENDENT kernel launch cost inside a long chain (the "launch floor" of DESIGN.md section 3.1)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 chain.hip -o chain
// A chain of N identical kernels is captured into a hipGraph from one stream (every node depends on its
// predecessor, exactly like the denoising step) and replayed; wall time / N is the per-launch cost including the
// kernel boundary.  Variants: grid size, block size, dynamic LDS, a 352-byte by-value argument block (the size of
// ConvArgs), and how many bytes each kernel leaves dirty for the boundary's write-back.  Also eager launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Big { float* p; int n; int pad[85]; };   // 352 bytes like ConvArgs
static_assert(sizeof(Big) == 352, "size");

__global__ void k_empty(float* p, int n) {}
__global__ void k_touch(float* p, int n) {      // every thread writes n/threads floats (16 B each store)
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    for (long i = tid * 4; i + 3 < n; i += nth * 4) *reinterpret_cast<float4*>(p + i) = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_big(const Big a) {            // reads the LAST word of its argument block: one more kernarg line
    if (a.pad[84] == 12345) a.p[0] = 1.f;
}
__global__ void k_lds(float* p, int n) {
    extern __shared__ float sm[];
    if (n == -1) p[0] = sm[threadIdx.x];
}
__global__ void k_rw(float* p, int n) {         // dependent read-modify-write of one line per block (a true data dependency)
    if (threadIdx.x == 0) p[blockIdx.x * 32] += 1.0f;
}

template <class F>
static void bench(const char* name, int N, F launch) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const int reps = 10;
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double graph_us = ms * 1e3 / (reps * N);
    // eager
    for (int i = 0; i < N; ++i) launch(s);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) for (int i = 0; i < N; ++i) launch(s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s graph %6.2f us/launch   eager %6.2f us/launch\n", name, graph_us, ms * 1e3 / (reps * N));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
}

int main() {
    float* buf; const int NB = 64 << 20; CK(hipMalloc(&buf, NB)); CK(hipMemset(buf, 0, NB));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    const int N = 200;
    bench("empty   1 x 64", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, buf, 0); });
    bench("empty 256 x 256", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, buf, 0); });
    bench("empty 256 x 512", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s, buf, 0); });
    bench("empty 2048 x 512", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(512), 0, s, buf, 0); });
    bench("empty 256 x 512, 64 KB dynamic LDS", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 64 * 1024, s, buf, 0); });
    Big b{}; b.p = buf; b.n = 0;
    bench("352-byte argument block, 256 x 512", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_big, dim3(256), dim3(512), 0, s, b); });
    bench("dependent RMW, 256 x 64 (one line per block)", N, [&](hipStream_t s) { hipLaunchKernelGGL(k_rw, dim3(256), dim3(64), 0, s, buf, 0); });
    for (int mb : {0, 1, 4, 16}) {
        char nm[96]; snprintf(nm, sizeof nm, "touch: %2d MB left dirty per kernel, 256 x 512", mb);
        const int n = mb ? mb * (1 << 18) : 4 * 256 * 512;   // floats (0 MB row: 16 B per thread = 2 MB... see name)
        if (mb == 0) snprintf(nm, sizeof nm, "touch: 16 B per thread (2 MB), 256 x 512");
        bench(nm, N, [&](hipStream_t s) { hipLaunchKernelGGL(k_touch, dim3(256), dim3(512), 0, s, buf, n); });
    }
    return 0;
}
