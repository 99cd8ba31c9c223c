// What does a dependent launch pay for the FIRST read of rows the previous launch wrote -- by where the producer ran?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 xcd_affinity.hip -o xcd_affinity
// A chain of N launches in one hipGraph; launch k: workgroup b reads the TILE (TB bytes, every thread 16-byte loads) that workgroup
// (b + shift) % NB of launch k - 1 wrote, adds 1, writes its own tile of the other buffer.  Workgroup b runs on XCD b % 8 (observed: l2_persist.hip), so
//   shift 0 = the producer was this XCD (and, in practice, this CU), shift 8 = this XCD, another CU, shift 1 / 4 = another XCD.
// Store policies: plain | write-through (sc0 sc1: what the product's mtv_store_out4 issues).  Printed: us per launch (boundary included) and, from s_memtime stamps of
// lane 0 of four sampled workgroups, cycles from kernel entry to "all my loads are back".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int WT, int PER, int LD = 0>
__global__ __launch_bounds__(512) void k_hop(const float* __restrict__ in, float* __restrict__ out, int shift, int nb, unsigned long long* stamps, int slot) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int b = blockIdx.x, tid = threadIdx.x;
    int src = b + shift;
    src = src >= nb ? src - nb : src;
    const size_t tile_f = (size_t)512 * 4 * PER;                       // floats per tile
    const float* p = in + (size_t)src * tile_f + 4 * tid;
    f32x4 v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        if (LD == 0) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)u * 2048);
        else {       // LD 1: sc1 (system-coherent: past this XCD's L2?)  LD 2: nt  LD 3: sc0 sc1
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(tile_f * 4), 0x00020000);
            v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rl, u * 8192, 0, LD == 1 ? 16 : (LD == 2 ? 2 : 17)));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float* q = out + (size_t)b * tile_f + 4 * tid;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)(tile_f * 4), 0x00020000);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const f32x4 w = v[u] + 1.0f;
        if (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w), rs, u * 8192, 0, 17);     // sc0 sc1
        else *reinterpret_cast<f32x4*>(q + (size_t)u * 2048) = w;
    }
    if (tid == 0 && (b == 0 || b == 9 || b == nb / 2 + 3 || b == nb - 1)) {
        const int k = b == 0 ? 0 : (b == 9 ? 1 : (b == nb - 1 ? 3 : 2));
        stamps[(size_t)slot * 4 + k] = t1 - t0;
    }
}

template <int WT, int PER, int LD = 0>
static void run(int shift, int nb, int N, float* A, float* B, unsigned long long* st_d) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k_hop<WT, PER, LD>), dim3(nb), dim3(512), 0, s, (i & 1) ? B : A, (i & 1) ? A : B, shift, nb, st_d, i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const int reps = 20;
    CK(hipEventRecord(e0, s));
    for (int w = 0; w < reps; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st((size_t)N * 4);
    CK(hipMemcpy(st.data(), st_d, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> all;
    for (int i = 4; i < N; ++i) for (int k = 0; k < 4; ++k) all.push_back(st[(size_t)i * 4 + k]);
    std::sort(all.begin(), all.end());
    printf("  %-13s ld %d %3d KB/WG  shift %2d   %6.2f us per launch   entry -> loads back: median %5llu  min %5llu  max %5llu ticks\n", WT ? "write-through" : "plain stores", LD, PER * 8, shift,
           1e3 * ms / (reps * N), all[all.size() / 2], all.front(), all.back());
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
}

int main() {
    const int nb = 256, N = 40;
    float *A, *B; unsigned long long* st;
    const size_t bytes = (size_t)nb * 512 * 16 * 8;
    CK(hipMalloc(&A, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMalloc(&st, (size_t)N * 4 * 8));
    CK(hipMemset(A, 0, bytes)); CK(hipMemset(B, 0, bytes)); CK(hipMemset(st, 0, (size_t)N * 4 * 8));
    printf("chain of %d dependent launches, %d workgroups x 512 threads; s_memtime ticks (shader clock)\n", N, nb);
    for (int shift : {0, 8, 16, 1, 4, 3}) {
        run<0, 2>(shift, nb, N, A, B, st);
        run<1, 2>(shift, nb, N, A, B, st);
    }
    for (int shift : {0, 8, 1}) {
        run<0, 8>(shift, nb, N, A, B, st);
        run<1, 8>(shift, nb, N, A, B, st);
    }
    printf("8 KB per workgroup (2 MB per tensor)\n");
    for (int shift : {0, 8, 1, 4}) {
        run<0, 1>(shift, nb, N, A, B, st);
        run<1, 1>(shift, nb, N, A, B, st);
    }
    printf("load policies on the consumer side (16 KB per workgroup): ld 1 = sc1, 2 = nt, 3 = sc0 sc1\n");
    for (int shift : {0, 1}) {
        run<0, 2, 1>(shift, nb, N, A, B, st);
        run<0, 2, 2>(shift, nb, N, A, B, st);
        run<0, 2, 3>(shift, nb, N, A, B, st);
        run<1, 2, 1>(shift, nb, N, A, B, st);
        run<1, 2, 3>(shift, nb, N, A, B, st);
    }
    return 0;
}
