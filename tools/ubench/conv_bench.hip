// Isolated benchmark of k_conv on one shape: sweeps tile shapes and ablations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics conv_bench.hip -o conv_bench
//   conv_bench Lout N Cin ntaps gn(0/1) [MT NT NW KS]...
#include "../../moditalker_amd/csrc/conv.hip"
#include <vector>
#include <cstring>
using namespace mtv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static std::vector<int> gather3(int r, int t, int& L, SegInfo& seg) {
    const int b1 = r * r, b2 = b1 + t * r;
    L = b2 + t * r;
    seg = SegInfo{b1, b2, L};
    std::vector<int> g((size_t)9 * L, -1);
    for (int p = 0; p < 3; ++p) {
        const int h = p == 0 ? r : t, w = r, off = p == 0 ? 0 : (p == 1 ? b1 : b2);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
            g[(size_t)(ky * 3 + kx) * L + off + y * w + x] = off + yy * w + xx;
        }
    }
    return g;
}

int main(int argc, char** argv) {
    if (argc < 6) { printf("usage: conv_bench R T N Cin ntaps gn [MT NT NW KS]...\n"); return 1; }
    const int R = atoi(argv[1]), T = atoi(argv[2]), N = atoi(argv[3]), Cin = atoi(argv[4]), ntaps = atoi(argv[5]), gn = atoi(argv[6]);
    int L; SegInfo seg;
    std::vector<int> g = gather3(R, T, L, seg);
    CK(conv_init_attrs());
    float *x, *W, *bias, *out, *gamma, *beta, *slab; double* sums; int* dg;
    const int ldw = (N + 63) / 64 * 64;
    CK(hipMalloc(&x, (size_t)L * Cin * 4)); CK(hipMalloc(&W, (size_t)ntaps * Cin * ldw * 4)); CK(hipMalloc(&bias, N * 4));
    CK(hipMalloc(&out, (size_t)L * N * 4)); CK(hipMalloc(&gamma, Cin * 4)); CK(hipMalloc(&beta, Cin * 4));
    CK(hipMalloc(&sums, 192 * 8)); CK(hipMalloc(&dg, g.size() * 4)); CK(hipMalloc(&slab, (size_t)16 * L * N * 4));
    std::vector<float> h((size_t)ntaps * Cin * ldw);
    for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.resize((size_t)L * Cin); for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(beta, 0, Cin * 4));
    std::vector<float> ones(Cin, 1.0f); CK(hipMemcpy(gamma, ones.data(), Cin * 4, hipMemcpyHostToDevice));
    std::vector<double> hs(192); for (int i = 0; i < 96; ++i) { hs[2 * i] = 0.0; hs[2 * i + 1] = 1000.0; }
    CK(hipMemcpy(sums, hs.data(), 192 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    ConvArgs a{};
    a.src[0] = x; a.C[0] = Cin; a.nmain = 1; a.Cmain = Cin; a.gather = ntaps == 9 ? dg : nullptr; a.ntaps = ntaps;
    a.Lout = L; a.Lsrc = L; a.Lskip = L; a.B = 1; a.W = W; a.ldw = ldw; a.N = N; a.bias = bias; a.out = out; a.seg_src = seg; a.seg_out = seg;
    a.slab = slab;
    CK(hipMalloc(&a.dbg, 4096)); CK(hipMemset(a.dbg, 0, 4096));
    CK(hipMalloc(&a.tickets, 1 << 20)); CK(hipMemset(a.tickets, 0, 1 << 20));
    if (const char* e = getenv("STATS")) {      // STATS="gs0[,gs1]": fused GroupNorm statistics targets
        int g0 = 0, g1 = 0;
        const int n = sscanf(e, "%d,%d", &g0, &g1);
        double* st; CK(hipMalloc(&st, 2 * 192 * 8)); CK(hipMemset(st, 0, 2 * 192 * 8));
        if (n >= 1 && g0 > 0) a.stat[a.nstat++] = StatOut{st, g0, 0};
        if (n >= 2 && g1 > 0) a.stat[a.nstat++] = StatOut{st + 192, g1, 0};
        printf("stats targets: %d (gs %d %d)\n", a.nstat, g0, g1);
    }
    if (gn) a.gn = GnIn{sums, gamma, beta, nullptr, 0, Cin / 32, 0, 1};
    const double flops = 2.0 * L * N * (double)ntaps * Cin;
    const double wbytes = 4.0 * ntaps * Cin * N;
    printf("conv L=%d N=%d Cin=%d taps=%d gn=%d: %.3f GFLOP, weights %.2f MB: ideal mfma %.2f us, ideal hbm %.2f us\n", L, N, Cin, ntaps, gn,
           flops / 1e9, wbytes / 1e6, flops / 155e6, wbytes / 5e6);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](ConvTile t) {
        const int reps = 20;
        for (int i = 0; i < 3; ++i) CK(launch_conv(a, t, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) CK(launch_conv(a, t, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        const int tiles = ((L + 16 * t.MT - 1) / (16 * t.MT)) * ((N + 16 * t.NT - 1) / (16 * t.NT));
#if MTV_ABLATE & 64
        {
            unsigned long long st[32];
            CK(hipMemcpy(st, a.dbg, sizeof st, hipMemcpyDeviceToHost));
            for (int blk = 0; blk < 4; ++blk) {
                printf("    blk(y%d,z%d) phases(clk): seg %5lld  prologue %5lld  kloop %6lld  wait+tree %6lld  epilogue %5lld | start+%lld", blk >> 1, blk & 1,
                       (long long)(st[blk * 8 + 1] - st[blk * 8]), (long long)(st[blk * 8 + 2] - st[blk * 8 + 1]), (long long)(st[blk * 8 + 3] - st[blk * 8 + 2]),
                       (long long)(st[blk * 8 + 4] - st[blk * 8 + 3]), (long long)(st[blk * 8 + 5] - st[blk * 8 + 4]), (long long)(st[blk * 8] - st[0]));
                printf("\n");
            }
        }
#endif
        printf("  tile %d,%d,%d,%d,%d  WGs %5d waves %6d : %8.2f us  %6.1f TF/s  %7.1f GB/s(w)\n", t.MT, t.NT, t.NW, t.KS, t.XM, tiles * t.KS, tiles * t.KS * t.NW,
               us, flops / us / 1e6, wbytes / us / 1e3);
    };
    if (argc >= 12) {
        for (int i = 7; i + 4 < argc; i += 5) run(ConvTile{atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]), atoi(argv[i + 4])});
    } else {
        const int nchunks = ntaps * Cin / 16;
        ConvTile p = conv_pick_tile(1, L, N, nchunks, Cin, gn);
        printf("  picked:"); run(p);
        static const int cand[][2] = {{4, 4}, {2, 4}, {1, 4}, {2, 2}, {1, 2}};
        for (auto& c : cand) for (int NW : {1, 2, 4, 8}) for (int KS : {1, 2, 4, 8}) {
            if (NW * KS > nchunks) continue;
            const int tiles = ((L + 16 * c[0] - 1) / (16 * c[0])) * ((N + 16 * c[1] - 1) / (16 * c[1]));
            const int waves = tiles * NW * KS;
            if (waves < 256 || waves > 8192) continue;
            run(ConvTile{c[0], c[1], NW, KS, 0});
        }
    }
    return 0;
}
