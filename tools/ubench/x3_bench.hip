// Isolated benchmark of k_x3_prep + k_conv_x3 (moditalker_amd/csrc/conv_x3.hip: split-bf16 convolution, both operands by LDS-DMA)
// with ablated variants: a dependent chain of launches in a hipGraph, activations ping-ponging between two buffers; the first
// launch's output is checked against a double-precision reference on sampled outputs (gn = 0 only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics [-DX3_ABLATE=n] x3_bench.hip -o x3_bench_aN
//   x3_bench rows C N taps gn(0|1) [MT NT]...
// rows = B * L tokens (one "clip" of that many tokens), C input channels per tap, N output channels, taps 1 or 9 (a gather
// table of shifted rows stands in for the 3x3 neighbourhood).  Prints us per launch and the f32-equivalent TFLOP/s.
#include "../../moditalker_amd/csrc/conv_x3.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using namespace mtv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { printf("usage: x3_bench rows C N taps gn [MT NT]...\n"); return 1; }
    const int L = atoi(argv[1]), C = atoi(argv[2]), N = atoi(argv[3]), taps = atoi(argv[4]), gn = atoi(argv[5]);
    const int NL = 16, K = taps * C;
    const SegInfo seg{L / 2, L * 3 / 4, L};
    float *x[2], *W, *bias, *gamma, *beta;
    void* W3;
    double *sums, *st;
    int* gather;
    void* X3;
    unsigned long long* dbg;
    const size_t plane = (size_t)K * N * 2;
    CK(hipMalloc(&x[0], (size_t)L * (C > N ? C : N) * 4)); CK(hipMalloc(&x[1], (size_t)L * (C > N ? C : N) * 4));
    CK(hipMalloc(&W, (size_t)K * N * 4)); CK(hipMalloc(&W3, 3 * plane)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4));
    CK(hipMalloc(&sums, 8 * 192 * 8)); CK(hipMalloc(&st, (size_t)8 * 192 * 8)); CK(hipMalloc(&gather, (size_t)taps * L * 4)); CK(hipMalloc(&dbg, 4096)); CK(hipMemset(dbg, 0, 4096)); CK(hipMalloc(&X3, conv_x3_scratch_bytes(1, L, C, L, 0)));
    std::vector<float> h((size_t)K * N);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f * 0.01f;
    CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.resize((size_t)L * (C > N ? C : N)); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(x[0], h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x[1], h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(beta, 0, C * 4));
    std::vector<float> ones(C, 1.0f); CK(hipMemcpy(gamma, ones.data(), C * 4, hipMemcpyHostToDevice));
    std::vector<double> hs(8 * 192, 0.0); for (int i = 0; i < 96; ++i) hs[2 * i + 1] = 1000.0;
    CK(hipMemcpy(sums, hs.data(), hs.size() * 8, hipMemcpyHostToDevice)); CK(hipMemset(st, 0, (size_t)8 * 192 * 8));
    std::vector<int> hg((size_t)taps * L);
    for (int t = 0; t < taps; ++t)
        for (int i = 0; i < L; ++i) {
            const int s = i + (t / 3 - taps / 6) * 64 + (t % 3) - (taps > 1 ? 1 : 0);
            hg[(size_t)t * L + i] = (s < 0 || s >= L) ? -1 : s;
        }
    CK(hipMemcpy(gather, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(launch_split_w3(W, W3, plane, 0, K, N, nullptr)); CK(hipDeviceSynchronize());
    CK(conv_x3_init_attrs());

    auto args_of = [&](int i) {
        ConvArgs a{};
        a.ntaps = taps; a.nmain = 1; a.Cmain = C; a.C[0] = C; a.Lout = a.Lsrc = a.Lskip = L; a.B = 1; a.N = N; a.ldw = N;
        a.W = W; a.x3 = X3; a.W3 = W3; a.w3_plane = plane; a.bias = bias; a.seg_src = seg; a.seg_out = seg; a.stat_cstride = 192;
        a.gather = taps > 1 ? gather : nullptr;
        a.src[0] = x[i & 1]; a.out = x[(i + 1) & 1];
        a.dbg = i == NL / 2 ? dbg : nullptr;
        if (gn) {
            a.gn = GnIn{sums, gamma, beta, nullptr, 0, C / 32, 0, 1, 192u};
            a.gn.inv_gs = 1.0f / (float)(C / 32);
            const double gs = C / 32;
            a.gn.inv_n[0] = 1.0 / (seg.b1 * gs); a.gn.inv_n[1] = 1.0 / ((seg.b2 - seg.b1) * gs); a.gn.inv_n[2] = 1.0 / ((seg.L - seg.b2) * gs); a.gn.inv_n[3] = 1.0 / (seg.L * gs);
            a.stat[0] = StatOut{st, N / 32, 0, 1.0f / (float)(N / 32)}; a.nstat = 1;
        }
        return a;
    };
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto graph_us = [&](ConvTile t) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NL; ++i) CK(launch_conv_x3(args_of(i), t, s));
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
        return (double)ms * 1e3 / reps / NL;
    };
    const double flops = 2.0 * L * N * (double)K;
    printf("x3 rows=%d C=%d N=%d taps=%d gn=%d: %.1f MFLOP, W3 %.2f MB; ablate %d\n", L, C, N, taps, gn, flops / 1e6, 3 * plane / 1e6, X3_ABLATE);
    // prep only, for the record
    {
        ConvArgs a = args_of(0);
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 20; ++r) CK(launch_x3_prep(a, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  k_x3_prep alone: %.2f us (stream launches)\n", ms * 1e3 / 20);
    }
    auto check = [&](ConvTile t) {
        std::vector<float> hx((size_t)L * C), hw((size_t)K * N), ho((size_t)L * N);
        for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
        CK(hipMemcpy(x[0], hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(hw.data(), W, hw.size() * 4, hipMemcpyDeviceToHost));
        std::vector<float> keep((size_t)L * (C > N ? C : N));
        CK(hipMemcpy(keep.data(), x[1], keep.size() * 4, hipMemcpyDeviceToHost));
        CK(launch_conv_x3(args_of(0), t, s)); CK(hipStreamSynchronize(s));
        CK(hipMemcpy(ho.data(), x[1], ho.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(x[1], keep.data(), keep.size() * 4, hipMemcpyHostToDevice));
        double worst = 0.0, scale = 0.0;
        for (int smp = 0; smp < 4096; ++smp) {
            const int r = smp < 64 ? smp : (smp < 128 ? L - 1 - (smp - 64) : rand() % L), n = rand() % N;
            double ref = 0.0;
            for (int tp = 0; tp < taps; ++tp) {
                const int sr = taps > 1 ? hg[(size_t)tp * L + r] : r;
                if (sr < 0) continue;
                for (int c = 0; c < C; ++c) ref += (double)hx[(size_t)sr * C + c] * (double)hw[((size_t)tp * C + c) * N + n];
            }
            const double d = fabs(ref - (double)ho[(size_t)r * N + n]);
            worst = d > worst ? d : worst;
            scale = fabs(ref) > scale ? fabs(ref) : scale;
        }
        printf("  check <%d,%d>: max |err| %.3e (max |ref| %.3f) %s\n", t.MT, t.NT, worst, scale, worst <= 2e-6 * (scale > 1 ? scale : 1) * 4 ? "ok" : "MISMATCH");
    };
    auto run = [&](int MT, int NT) {
        const ConvTile t{MT, NT, 96, 1, 0};
        if (conv_x3_smem_bytes(args_of(0), t) > CONV_X3_MAX_LDS) return;
        if (!gn && !X3_ABLATE) check(t);
        const double us = graph_us(t);
        printf("  k_conv_x3<%d,%d>  WGs %5d  chunks %3d : %7.2f us/launch  %6.1f TF/s  %6.0f ns/chunk\n", MT, NT, ((L + 32 * MT - 1) / (32 * MT)) * ((N + 64 * NT - 1) / (64 * NT)), K / 32, us,
               flops / us / 1e6, us * 1e3 / (K / 32));
#ifdef X3_STAMP
        unsigned long long d[64];
        CK(hipMemcpy(d, dbg, sizeof d, hipMemcpyDeviceToHost));
        for (int w = 0; w < 2; ++w)
            for (int it = 0; it < 4; ++it) {
                const unsigned long long* v = d + w * 32 + it * 8;
                if (!v[0]) continue;
                printf("      wave %d it %d (cycles): frags+issue %llu wait %llu barrier %llu mfma %llu wait+barrier %llu\n", w ? 7 : 0, 8 + it, v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], v[5] - v[4]);
            }
        CK(hipMemset(dbg, 0, 4096));
#endif
    };
    if (argc >= 8) {
        for (int i = 6; i + 1 < argc; i += 2) run(atoi(argv[i]), atoi(argv[i + 1]));
    } else {
        const int tl[7][2] = {{2, 1}, {4, 1}, {2, 2}, {4, 2}, {8, 1}, {8, 2}, {4, 4}};
        for (auto& t : tl) run(t[0], t[1]);
    }
    return 0;
}
