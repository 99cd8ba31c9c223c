// Does data one kernel loaded stay in the XCD's L2 for the NEXT kernel of a dependent chain (hipGraph, same stream)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 l2_persist.hip -o l2_persist
// Every launch, thread 0 of workgroup b times (s_memtime) one 64-byte global load from the SAME address p + 1024 b, and
// the first s_load of a kernel-argument word.  Workgroup b of consecutive launches runs on XCD b % 8 (observed), so a
// warm L2 shows as ~200-cycle loads from the second launch on; a cold one as ~900+ every time.  Variants: nothing in
// between, a 16 MB streaming kernel in between, and loads with the sc1 (L1-bypass) policy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args { const float* p; unsigned long long* out; int slot; int pad[61]; };   // 264 bytes: several kernarg lines

__global__ void k_lat(const Args a) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int last = a.pad[60];                                   // a kernel-argument word on another line
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const float v = __builtin_nontemporal_load(a.p + 1024 * blockIdx.x + last);   // (last == 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    unsigned long long* o = a.out + ((size_t)a.slot * gridDim.x + blockIdx.x) * 2;
    o[0] = t1 - t0;
    o[1] = (t2 - t1) + (v == 12345.f ? 1 : 0);
}
__global__ void k_lat_plain(const Args a) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int last = a.pad[60];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const float v = a.p[1024 * blockIdx.x + last];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    unsigned long long* o = a.out + ((size_t)a.slot * gridDim.x + blockIdx.x) * 2;
    o[0] = t1 - t0;
    o[1] = (t2 - t1) + (v == 12345.f ? 1 : 0);
}
__global__ void k_stream(float* p, int n) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
    for (long i = tid * 4; i + 3 < n; i += nth * 4) *reinterpret_cast<float4*>(p + i) = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    const int NB = 64, N = 24;
    float *p, *big;
    unsigned long long* out;
    CK(hipMalloc(&p, NB * 1024 * 4 + 4096)); CK(hipMemset(p, 0, NB * 1024 * 4 + 4096));
    CK(hipMalloc(&big, 64 << 20));
    CK(hipMalloc(&out, (size_t)N * NB * 16));
    for (int variant = 0; variant < 4; ++variant) {
        CK(hipMemset(out, 0, (size_t)N * NB * 16));
        hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            Args a{}; a.p = p; a.out = out; a.slot = i;
            if (variant == 2) hipLaunchKernelGGL(k_lat, dim3(NB), dim3(64), 0, s, a);
            else hipLaunchKernelGGL(k_lat_plain, dim3(NB), dim3(64), 0, s, a);
            if (variant == 1 || variant == 3) hipLaunchKernelGGL(k_stream, dim3(256), dim3(256), 0, s, big, variant == 1 ? (4 << 20) : (1 << 18));
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));      // first replay: everything cold
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)N * NB * 2);
        CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        const char* names[] = {"plain loads, back to back", "plain loads, 16 MB streamed in between", "nontemporal loads, back to back", "plain loads, 1 MB streamed in between"};
        printf("%-44s launch: kernarg-line / data-load latency (median over %d blocks, s_memtime ticks)\n", names[variant], NB);
        for (int i = 0; i < N; i += (i < 4 ? 1 : 5)) {
            std::vector<unsigned long long> ka, da;
            for (int b = 0; b < NB; ++b) { ka.push_back(h[((size_t)i * NB + b) * 2]); da.push_back(h[((size_t)i * NB + b) * 2 + 1]); }
            std::sort(ka.begin(), ka.end()); std::sort(da.begin(), da.end());
            printf("   launch %2d: kernarg %5llu   data %5llu  (min %llu max %llu)\n", i, ka[NB / 2], da[NB / 2], da[0], da[NB - 1]);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
    }
    return 0;
}
