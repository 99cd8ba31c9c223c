// How fast can a kernel pull a COLD 24 MB buffer into the L2s, by access pattern and by how much of the chip takes part?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 cold_stream.hip -o cold_stream
// Between timed launches a 512 MB buffer is streamed (evicts L2 and the 256 MB memory-side cache).  Patterns:
//   f4   : every lane loads 16 contiguous bytes (the copy-kernel pattern)
//   d64  : one dword per 64 bytes (touch-only prefetch), lanes on consecutive 64-byte segments
//   d128 : one dword per 128 bytes
// each with 8 independent loads in flight per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_read(const float* p, size_t bytes, float* sink) {
    const size_t step = MODE == 0 ? 16 : (MODE == 1 ? 64 : 128);
    const size_t n = bytes / step;
    const size_t nth = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (size_t e0 = tid; e0 < n; e0 += 8 * nth) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t e = e0 + u * nth;
            if (MODE == 0) {
                const float4 t = e < n ? *reinterpret_cast<const float4*>((const char*)p + e * 16) : make_float4(0, 0, 0, 0);
                v[u] = t.x + t.y + t.z + t.w;
            } else {
                v[u] = e < n ? *reinterpret_cast<const float*>((const char*)p + e * step) : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_flush(float* p, size_t n) {
    const size_t nth = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = tid * 4; i + 3 < n; i += nth * 4) *reinterpret_cast<float4*>(p + i) = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const size_t bytes = 24u << 20, fl = 512u << 20;
    float *buf, *flush, *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    CK(hipMalloc(&flush, fl)); CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"f4 (16 B per lane, contiguous)", "d64 (one dword per 64 B)", "d128 (one dword per 128 B)"};
    for (int mode = 0; mode < 3; ++mode)
        for (int wgs : {64, 128, 256, 512, 1024}) {
            float best = 1e9f, sum = 0.f;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(k_flush, dim3(1024), dim3(256), 0, 0, flush, fl / 4);
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(k_read<0>, dim3(wgs), dim3(256), 0, 0, buf, bytes, sink);
                else if (mode == 1) hipLaunchKernelGGL(k_read<1>, dim3(wgs), dim3(256), 0, 0, buf, bytes, sink);
                else hipLaunchKernelGGL(k_read<2>, dim3(wgs), dim3(256), 0, 0, buf, bytes, sink);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best; sum += ms;
            }
            printf("%-34s %5d WGs x 256: best %7.1f us  mean %7.1f us  -> %5.2f TB/s (cold, 24 MB)\n", names[mode], wgs, best * 1e3, sum / 5 * 1e3, bytes / (best * 1e-3) / 1e12);
        }
    // warm (no flush): the same 24 MB again
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k_read<0>, dim3(512), dim3(256), 0, 0, buf, bytes, sink);
        CK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(k_read<0>, dim3(512), dim3(256), 0, 0, buf, bytes, sink);
        else hipLaunchKernelGGL(k_read<1>, dim3(512), dim3(256), 0, 0, buf, bytes, sink);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s   512 WGs x 256, warm (L2/MALL): %7.1f us\n", names[mode], ms * 1e3);
    }
    return 0;
}
