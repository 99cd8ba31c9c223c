// Micro-benchmarks that calibrate the roofline on the box: f32 MFMA rate, HBM copy bandwidth,
// L2-resident streaming, launch gap.  hipcc --offload-arch=gfx950 -O3 mfma_bw.hip -o mfma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_mfma(float* out, int iters, float a0) {
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = a0 * 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
    float s = 0;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_copy(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_read(const f32x4* __restrict__ in, float* out, size_t n) {
    f32x4 s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s[0] == 12345.f) out[0] = s[0] + s[1] + s[2] + s[3];
}
__global__ void k_empty() {}

int main() {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    float* out; CK(hipMalloc(&out, 1 << 24));
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int blocks = 256 * waves_per_simd, iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            double fl = 2.0 * 16 * 16 * 4 * 8.0 * iters * blocks * 4;
            printf("mfma f32 16x16x4: %d blocks x256thr: %.3f ms  %.1f TFLOP/s\n", blocks, ms, fl / ms / 1e9);
        }
    }
    // short mfma kernel (like a 5us conv): 1024 waves x 100 iters x 8 = 800 MFMA per wave ~ 10.7us at peak
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mfma, dim3(256), dim3(256), 0, 0, out, 100, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("short mfma (800 MFMA/wave, ideal 10.7us @2.4GHz): %.2f us\n", ms * 1e3);
    }
    size_t nbytes = (size_t)1 << 30;
    f32x4 *a, *b; CK(hipMalloc(&a, nbytes)); CK(hipMalloc(&b, nbytes));
    CK(hipMemset(a, 1, nbytes));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, a, b, nbytes / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy 1 GiB: %.3f ms  %.1f GB/s (read+write)\n", ms, 2.0 * nbytes / ms / 1e6);
    }
    for (size_t sz : {(size_t)16 << 20, (size_t)64 << 20, (size_t)512 << 20}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, a, out, sz / 16);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("read %zu MiB: %.2f us  %.1f GB/s\n", sz >> 20, ms * 1e3, sz / ms / 1e6);
        }
    }
    // launch gap: 200 empty kernels back to back, and event-bracketed single empty kernel
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("200 empty kernels: %.2f us each\n", ms * 1e3 / 200);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("event-bracketed empty kernel: %.2f us\n", ms * 1e3);
    }
    return 0;
}
