// Can consecutive DEPENDENT kernels of a chain overlap: kernel k+1 runs its data-independent prologue while kernel k
// computes, and picks up k's output through a device flag instead of a kernel boundary?   
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 overlap.hip -o overlap
// Every kernel of the chain = [prologue: P us that depend on nothing] -> [wait for the predecessor] -> [body: read the
// predecessor's 1 MB output, add 1, write the own output, Q us in all] -> [publish].  Variants:
//   chain     one stream, the kernel boundary is the dependency (today's step graph)
//   2branch   even kernels on one capture stream, odd ones on another (in-stream order k-2 -> k); the k -> k+1
//             dependency is a device counter that k's workgroups bump after draining their write-through stores and
//             that one lane per workgroup of k+1 polls (bounded spin) after the prologue
// both as a hipGraph and as eager launches, for several grid shapes.  Two hand-off protocols are priced:
//   sc1       producer: sc0 sc1 (write-through) 16-byte stores, s_waitcnt vmcnt(0), relaxed agent atomic add
//             consumer: relaxed agent poll, barrier, sc0 sc1 16-byte loads            (MI355X_MICROARCH.md, valid forms)
//   fence     producer: plain stores, barrier, lane-0 release fence, asm vmcnt(0), relaxed add
//             consumer: relaxed poll, lane-0 acquire fence, barrier, plain loads
//   go        like sc1, but nobody polls the arrival counter: the workgroup whose (returning) atomic add completes the
//             count writes one GO word per consumer workgroup (256 words, one 4-byte write-through store per thread) and
//             consumer workgroup j polls go[j] only -- 256 pollers on one word were what made `sc1` cost ~10 us per hop
//   go8       the same with 8 GO words (128-byte lines apart), consumer j polls go8[j % 8]
// After R replays of an N-kernel chain every word of the buffer must read R*N: checked, every word.  Workgroups are
// skewed (every 7th one spins 1 us longer in its body) so arrivals are uneven.  A spin that exceeds 20 ms gives up and
// raises an error word, so a scheduling deadlock shows up as a failed check, not as a hung GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
    const float* in;
    float* out;
    int n4;                 // 16-byte units
    int pro_ticks, body_ticks;   // 100 MHz ticks (s_memrealtime)
    const int* wait;        // predecessor's arrival counter (nullptr: the kernel boundary is the dependency)
    int wait_per_epoch;     // arrivals per replay = predecessor's grid size
    int* done;              // own arrival counter
    const int* epoch;       // replay number (1-based), advanced by k_epoch at the end of every replay
    int* err;
    int proto;              // 0 sc1, 1 fence, 2 go, 3 go8
    int* go_out;            // GO words this kernel's last arriver writes (proto 2/3)
    const int* go_in;       // GO words this kernel polls
    int n_go;
    const float* wts;       // optional: prologue streams `wts_bytes` per workgroup from here (cold weights) instead of spinning
    int wts_f4_per_wg;
    float* sink;
};

__device__ __forceinline__ long long now() { return (long long)__builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void spin_until(long long t) { while (now() < t) __builtin_amdgcn_s_sleep(1); }

__device__ __forceinline__ f32x4 load_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_sc1(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

__global__ __launch_bounds__(512) void k_stage(const Args a) {
    const int tid = threadIdx.x;
    const long long t0 = now();
    const int ep = *a.epoch;
    // ---- prologue (independent of the predecessor)
    if (a.wts) {
        const f32x4* w = reinterpret_cast<const f32x4*>(a.wts) + (size_t)blockIdx.x * a.wts_f4_per_wg;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < a.wts_f4_per_wg; i += blockDim.x) acc += w[i];
        if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) a.sink[0] = acc[0];
    }
    spin_until(t0 + a.pro_ticks);
    // ---- wait for the predecessor
    if (a.wait) {
        if (tid == 0) {
            const long long dl = now() + 2000000;     // 20 ms
            if (a.proto >= 2) {
                const int* g = a.proto == 2 ? a.go_in + blockIdx.x : a.go_in + 32 * (blockIdx.x & 7);
                while (__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ep) {
                    __builtin_amdgcn_s_sleep(4);
                    if (now() > dl) { atomicAdd(a.err, 1); break; }
                }
            } else {
                const int want = ep * a.wait_per_epoch;
                while (__hip_atomic_load(a.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (now() > dl) { atomicAdd(a.err, 1); break; }
                }
            }
            if (a.proto == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // ---- body
    const long long t1 = now();
    const int gid = blockIdx.x * blockDim.x + tid, nth = gridDim.x * blockDim.x;
    for (int i = gid; i < a.n4; i += nth) {
        f32x4 v;
        if (a.wait && a.proto != 1) v = load_sc1(a.in + 4 * (size_t)i);
        else v = *reinterpret_cast<const f32x4*>(a.in + 4 * (size_t)i);
        v += 1.0f;
        if (a.done && a.proto != 1) store_sc1(a.out + 4 * (size_t)i, v);
        else *reinterpret_cast<f32x4*>(a.out + 4 * (size_t)i) = v;
    }
    spin_until(t1 + a.body_ticks + (blockIdx.x % 7 == 3 ? 100 : 0));
    // ---- publish
    if (a.done) {
        if (a.proto == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(a.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (a.proto >= 2) {
            __shared__ int s_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) s_last = __hip_atomic_fetch_add(a.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ep * (int)gridDim.x - 1;
            __syncthreads();
            if (s_last) {
                if (a.proto == 2) { for (int j = tid; j < a.n_go; j += blockDim.x) __hip_atomic_store(a.go_out + j, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else if (tid < 8) __hip_atomic_store(a.go_out + 32 * tid, ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(a.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ void k_epoch(int* epoch) { if (threadIdx.x == 0) *epoch += 1; }

struct Cfg { int grid, block, pro_us, body_us, proto; bool wts; };

static float* g_buf[2];
static int *g_flags, *g_epoch, *g_err, *g_go;
static float *g_wts, *g_sink;
static const int N4 = 65536;   // 1 MB

static Args make_args(const Cfg& c, int k, bool flagged) {
    Args a{};
    a.in = g_buf[k & 1];
    a.out = g_buf[(k + 1) & 1];
    a.n4 = N4;
    a.pro_ticks = c.pro_us * 100;
    a.body_ticks = c.body_us * 100;
    a.wait = flagged && k > 0 ? g_flags + 32 * (k - 1) : nullptr;
    a.wait_per_epoch = c.grid;
    a.done = flagged ? g_flags + 32 * k : nullptr;
    a.epoch = g_epoch;
    a.err = g_err;
    a.proto = c.proto;
    a.go_out = flagged ? g_go + 4096 * k : nullptr;                 // (16 KB per kernel: room for 4096 GO words or 8 spaced ones)
    a.go_in = flagged && k > 0 ? g_go + 4096 * (k - 1) : nullptr;
    a.n_go = c.grid;
    a.wts = c.wts ? g_wts + (size_t)(k % 8) * (8u << 20) : nullptr;     // 8 x 32 MB regions, revisited every 8 kernels
    a.wts_f4_per_wg = c.wts ? (int)((32u << 20) / 16 / c.grid) : 0;
    if (c.wts) a.pro_ticks = 0;
    a.sink = g_sink;
    return a;
}

static bool check(const char* what, int expect) {
    std::vector<float> h((size_t)N4 * 4);
    CK(hipMemcpy(h.data(), g_buf[0], h.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (float v : h) bad += v != (float)expect;
    int err = 0;
    CK(hipMemcpy(&err, g_err, 4, hipMemcpyDeviceToHost));
    if (bad || err) printf("   !! %s: %zu of %zu words wrong (expected %d, first %g), %d spin time-outs\n", what, bad, h.size(), expect, h[0], err);
    return !bad && !err;
}

static void reset() {
    CK(hipMemset(g_buf[0], 0, (size_t)N4 * 16));
    CK(hipMemset(g_buf[1], 0, (size_t)N4 * 16));
    CK(hipMemset(g_flags, 0, 32 * 4 * 1024));
    CK(hipMemset(g_err, 0, 4));
    CK(hipMemset(g_go, 0, (size_t)4096 * 4 * 128));
    const int one = 1;
    CK(hipMemcpy(g_epoch, &one, 4, hipMemcpyHostToDevice));
}

static void run(const Cfg& c, int N) {
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef, ej;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    const int reps = 6, warm = 2;
    double us[4] = {0, 0, 0, 0};
    bool ok[4] = {true, true, true, true};
    for (int variant = 0; variant < 4; ++variant) {     // 0 chain graph, 1 2branch graph, 2 chain eager, 3 2branch eager
        const bool two = variant & 1, graph = variant < 2;
        auto issue = [&]() {
            if (two) { CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0)); }
            for (int k = 0; k < N; ++k) {
                const Args a = make_args(c, k, two);
                hipLaunchKernelGGL(k_stage, dim3(c.grid), dim3(c.block), 0, (two && (k & 1)) ? s1 : s0, a);
            }
            if (two) { CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); }
            hipLaunchKernelGGL(k_epoch, dim3(1), dim3(64), 0, s0, g_epoch);
        };
        reset();
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        if (graph) {
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
            issue();
            CK(hipStreamEndCapture(s0, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        auto once = [&]() { if (graph) CK(hipGraphLaunch(ge, s0)); else issue(); };
        for (int w = 0; w < warm; ++w) once();
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(e0, s0));
        for (int r = 0; r < reps; ++r) once();
        CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us[variant] = ms * 1e3 / (reps * N);
        CK(hipDeviceSynchronize());
        ok[variant] = check(variant == 0 ? "chain graph" : variant == 1 ? "2branch graph" : variant == 2 ? "chain eager" : "2branch eager", (reps + warm) * N);
        if (ge) CK(hipGraphExecDestroy(ge));
        if (g) CK(hipGraphDestroy(g));
    }
    printf("%4d x %3d  pro %s body %2d us  %-5s | chain: graph %6.2f eager %6.2f | 2branch: graph %6.2f%s eager %6.2f%s  us/kernel\n", c.grid, c.block,
           c.wts ? "32MB" : (c.pro_us == 4 ? " 4us" : c.pro_us == 2 ? " 2us" : c.pro_us == 0 ? " 0us" : " ?us"), c.body_us, c.proto == 0 ? "sc1" : c.proto == 1 ? "fence" : c.proto == 2 ? "go" : "go8", us[0], us[2], us[1],
           ok[1] ? "" : " (BAD)", us[3], ok[3] ? "" : " (BAD)");
    fflush(stdout);
    CK(hipStreamDestroy(s0)); CK(hipStreamDestroy(s1));
}

int main() {
    CK(hipMalloc(&g_buf[0], (size_t)N4 * 16)); CK(hipMalloc(&g_buf[1], (size_t)N4 * 16));
    CK(hipMalloc(&g_flags, 32 * 4 * 1024)); CK(hipMalloc(&g_epoch, 4)); CK(hipMalloc(&g_err, 4)); CK(hipMalloc(&g_go, (size_t)4096 * 4 * 128));
    CK(hipMalloc(&g_wts, (size_t)256 << 20)); CK(hipMemset(g_wts, 0, (size_t)256 << 20));
    CK(hipMalloc(&g_sink, 64));
    const int N = 100;
    const Cfg cfgs[] = {
        {256, 256, 4, 5, 0, false}, {256, 256, 4, 5, 1, false}, {256, 256, 4, 5, 2, false}, {256, 256, 4, 5, 3, false},
        {128, 512, 4, 5, 0, false}, {128, 512, 4, 5, 2, false}, {128, 512, 4, 5, 3, false},
        {256, 512, 4, 5, 2, false}, {256, 512, 4, 5, 3, false},
        {512, 256, 4, 5, 0, false}, {512, 256, 4, 5, 2, false}, {512, 256, 4, 5, 3, false}, {512, 512, 4, 5, 2, false},
        {1024, 256, 4, 5, 0, false}, {1024, 256, 4, 5, 2, false}, {1024, 256, 4, 5, 3, false},
        {256, 256, 2, 3, 2, false}, {256, 256, 4, 10, 2, false}, {256, 256, 0, 5, 2, false},
        {256, 256, 0, 1, 0, false}, {256, 256, 0, 1, 1, false}, {256, 256, 0, 1, 2, false}, {256, 256, 0, 1, 3, false},
        {64, 256, 0, 1, 2, false}, {64, 256, 0, 1, 3, false}, {64, 256, 4, 5, 2, false},
        {256, 512, 0, 5, 0, true}, {256, 512, 0, 5, 2, true}, {256, 512, 0, 10, 2, true},
    };
    for (const Cfg& c : cfgs) run(c, N);
    return 0;
}
