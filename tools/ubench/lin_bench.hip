// Isolated benchmark of the lean 1x1 kernel k_lin (moditalker_amd/csrc/lin.hip) in the regime of the denoising step:
// a DEPENDENT chain of launches inside a hipGraph, activations hot (written by the previous launch), weights cold
// (each launch has its own weight matrix; a 512 MB sweep before every replay evicts L2 + Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DLIN_STAMP [-DLIN_ABLATE=n] lin_bench.hip -o lin_bench
//   lin_bench L K N kind(0 proj: residual + statistics, 1 qkv: GroupNorm prologue) [MT NT NWV]...
// Prints us per launch ((graph time - sweep time) / launches) and the in-kernel phase stamps of three sampled workgroups.
#include "../../moditalker_amd/csrc/lin.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mtv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_sweep(const float4* p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x;
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
    if (argc < 5) { printf("usage: lin_bench L K N kind [MT NT NWV]...\n"); return 1; }
    const int L = atoi(argv[1]), K = atoi(argv[2]), N = atoi(argv[3]), kind = atoi(argv[4]);
    const int NL = 32;                               // launches per chain
    const SegInfo seg{L / 2, L * 3 / 4, L};
    float *x[2], *W, *bias, *gamma, *beta, *big, *sink;
    double *sums, *st;
    unsigned long long* dbg;
    const size_t wfl = (size_t)N * K;
    CK(hipMalloc(&x[0], (size_t)L * (kind ? 3 * K : N) * 4 + (size_t)L * K * 4)); CK(hipMalloc(&x[1], (size_t)L * (kind ? 3 * K : N) * 4 + (size_t)L * K * 4));
    CK(hipMalloc(&W, wfl * 4 * NL)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&gamma, K * 4)); CK(hipMalloc(&beta, K * 4));
    CK(hipMalloc(&big, (size_t)512 << 20)); CK(hipMalloc(&sink, 64)); CK(hipMemset(big, 0, (size_t)512 << 20));
    CK(hipMalloc(&sums, 8 * 192 * 8)); CK(hipMalloc(&st, (size_t)NL * 8 * 192 * 8)); CK(hipMalloc(&dbg, 4096)); CK(hipMemset(dbg, 0, 4096));
    std::vector<float> h(wfl * NL);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f * 0.005f;
    CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.resize((size_t)L * K); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(x[0], h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x[1], h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(beta, 0, K * 4));
    std::vector<float> ones(K, 1.0f); CK(hipMemcpy(gamma, ones.data(), K * 4, hipMemcpyHostToDevice));
    std::vector<double> hs(8 * 192, 0.0); for (int i = 0; i < 96; ++i) hs[2 * i + 1] = 1000.0;
    CK(hipMemcpy(sums, hs.data(), hs.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(st, 0, (size_t)NL * 8 * 192 * 8));

    auto args_of = [&](int i) {
        ConvArgs a{};
        a.ntaps = 1; a.nmain = 1; a.Cmain = K; a.C[0] = K; a.Lout = a.Lsrc = a.Lskip = L; a.B = 1; a.N = N;
        a.Wnk = W + wfl * i; a.W = a.Wnk; a.bias = bias; a.seg_src = seg; a.seg_out = seg; a.stat_cstride = 192;
        if (kind == 0) {           // proj: out_i = W_i out_{i-1} + out_{i-1}; statistics for one consumer (N == K)
            a.src[0] = x[i & 1]; a.out = x[(i + 1) & 1]; a.res = x[i & 1];
            a.stat[0] = StatOut{st + (size_t)i * 8 * 192, N / 32, 0, 1.0f / (float)(N / 32)}; a.nstat = 1;
        } else {                   // qkv: GroupNorm prologue, [L][K] -> [L][N]; the input stays the same (hot) tensor
            a.src[0] = x[0]; a.out = x[1];
            a.gn = GnIn{sums, gamma, beta, nullptr, 0, K / 32, i & 1, 0, 192u};
            a.gn.inv_gs = 1.0f / (float)(K / 32);
            const double gs = K / 32;
            a.gn.inv_n[0] = 1.0 / (seg.b1 * gs); a.gn.inv_n[1] = 1.0 / ((seg.b2 - seg.b1) * gs); a.gn.inv_n[2] = 1.0 / ((seg.L - seg.b2) * gs); a.gn.inv_n[3] = 1.0 / (seg.L * gs);
        }
        a.dbg = i == NL / 2 ? dbg : nullptr;
        return a;
    };
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto graph_us = [&](bool with_lin, ConvTile t) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_sweep, dim3(2048), dim3(256), 0, s, (const float4*)big, ((size_t)512 << 20) / 16, sink);
        if (with_lin)
            for (int i = 0; i < NL; ++i) CK(launch_lin(args_of(i), t, s));
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 2; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const int reps = 5;
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return (double)ms * 1e3 / reps;
    };
    const double sweep = graph_us(false, ConvTile{});
    const double flops = 2.0 * L * N * (double)K;
    printf("lin L=%d K=%d N=%d %s: %.1f MFLOP, W %.2f MB; sweep %.1f us; ablate %d\n", L, K, N, kind ? "qkv(gn)" : "proj(res+stats)", flops / 1e6, wfl * 4 / 1e6, sweep, LIN_ABLATE);
    auto run = [&](ConvTile t) {
        if (16 * t.NT * t.KS > N && t.KS > 1) return;
        const double us = (graph_us(true, t) - sweep) / NL;
        unsigned long long d[24];
        CK(hipMemcpy(d, dbg, sizeof d, hipMemcpyDeviceToHost));
        printf("  k_lin<%d,%d,%d>  WGs %5d : %6.2f us/launch  %5.1f TF/s |", t.MT, t.NT, t.KS, ((L + 16 * t.MT - 1) / (16 * t.MT)) * ((N + 16 * t.NT * t.KS - 1) / (16 * t.NT * t.KS)), us, flops / us / 1e6);
        for (int b = 0; b < 3; ++b) {
            const unsigned long long* v = d + 8 * b;
            auto us_of = [&](int k1, int k0) { return v[k1] && v[k0] ? (double)(v[k1] - v[k0]) / 2100.0 : 0.0; };
            printf(" [args %.2f issue %.2f gn %.2f k %.2f epi %.2f stats %.2f]", us_of(1, 0), us_of(2, 1), us_of(3, 2), us_of(5, 3), us_of(6, 5), us_of(7, 6));
        }
        printf("\n");
        CK(hipMemset(dbg, 0, 4096));
    };
    if (argc >= 8) {
        for (int i = 5; i + 2 < argc; i += 3) run(ConvTile{atoi(argv[i]), atoi(argv[i + 1]), 64, atoi(argv[i + 2]), 0});
    } else {
        for (int MT = 1; MT <= 2; ++MT)
            for (int NT = 1; NT <= 4; NT *= 2)
                for (int NWV = 1; NWV <= 4; NWV *= 2) run(ConvTile{MT, NT, 64, NWV, 0});
    }
    return 0;
}
