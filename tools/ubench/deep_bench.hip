// k_deep_conv (moditalker_amd/csrc/deep.hip): correctness against a plain CPU conv on the tri-plane grid, and the chain
// experiment VERDICT r3 item 1 asks for -- a dependent chain of [32 x 512, K = 4608] convs with DISTINCT 9.4 MB weights and a
// GroupNorm between consecutive ops, (a) as today's k_conv launches (statistics by epilogue atomics, in-launch split-K),
// (b) as k_deep_conv launches (consumer-side slab sums + statistics), both replayed as one hipGraph.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics deep_bench.hip -o deep_bench
//   deep_bench check          correctness cases
//   deep_bench chain [nops] [r t]   timing (default 40 ops = 377 MB of weights: larger than the 256 MB Infinity Cache; r t = 4 2)
#include "../../moditalker_amd/csrc/conv.hip"
#include "../../moditalker_amd/csrc/deep.hip"
#include "../../moditalker_amd/csrc/block.hip"
#include <cmath>
#include <functional>
#include <cstring>
#include <vector>
using namespace mtv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static unsigned g_seed = 12345;
static float frand() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
template <class T> static T* dnew(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T) + 16)); CK(hipMemset(p, 0, n * sizeof(T) + 16)); return p; }
static float* dup(const std::vector<float>& h) { float* p = dnew<float>(h.size()); CK(hipMemcpy(p, h.data(), h.size() * 4, hipMemcpyHostToDevice)); return p; }

struct Case {
    const char* name;
    int r, t, up;            // output-level geometry; up: tapped source on the coarser level
    int ntaps, Cm0, Cm1, Cs0, Cs1, N;
    int gn, whole, act, film, res, up_res;
    int ks_in, ks_res, KS, nrg, B;
};

// a slab tensor with a known sum: returns the summed host copy, uploads ks random slabs that add up to it
static DeepSrc make_src(int B, int L, int C, int ks, std::vector<float>& sum) {
    sum.assign((size_t)B * L * C, 0.f);
    std::vector<float> slabs((size_t)ks * B * L * C);
    for (int k = 0; k < ks; ++k)
        for (size_t e = 0; e < sum.size(); ++e) slabs[(size_t)k * sum.size() + e] = frand() * (k ? 0.5f : 1.0f);
    for (size_t e = 0; e < sum.size(); ++e) {          // slab order, as the kernel adds them
        float s = slabs[e];
        for (int k = 1; k < ks; ++k) s += slabs[(size_t)k * sum.size() + e];
        sum[e] = s;
    }
    DeepSrc d{};
    d.p = dup(slabs);
    d.slab_stride = (unsigned)sum.size();
    d.ks = ks;
    d.C = C;
    return d;
}

static int run_case(const Case& c) {
    const int r = c.r, t = c.t, b1 = r * r, b2 = b1 + t * r, L = b2 + t * r;
    // up / up_res: 1 = that source on the next-coarser level (nearest x2), 2 = on the next-FINER level (2x2 mean per plane: ResBlock(down=True))
    const int rs = c.up == 1 ? r / 2 : (c.up == 2 ? 2 * r : r), ts = c.up == 1 ? t / 2 : (c.up == 2 ? 2 * t : t), b1s = rs * rs, b2s = b1s + ts * rs, Ls = b2s + ts * rs;
    const int rr = c.up_res == 1 ? r / 2 : (c.up_res == 2 ? 2 * r : r), tr = c.up_res == 1 ? t / 2 : (c.up_res == 2 ? 2 * t : t), Lr = rr * rr + 2 * tr * rr;
    // the four finer-level tokens under a token of this level (same plane)
    auto finer4 = [&](int tok, int (&o)[4]) {
        const int p = tok >= b2 ? 2 : (tok >= b1 ? 1 : 0), off = p == 0 ? 0 : (p == 1 ? b1 : b2);
        const int r2 = 2 * r, t2 = 2 * t, b1f = r2 * r2, b2f = b1f + t2 * r2, offf = p == 0 ? 0 : (p == 1 ? b1f : b2f);
        const int y = (tok - off) / r, x = (tok - off) % r;
        o[0] = offf + 2 * y * r2 + 2 * x; o[1] = o[0] + 1; o[2] = o[0] + r2; o[3] = o[2] + 1;
    };
    const int Cmain = c.Cm0 + c.Cm1, Cskip = c.Cs0 + c.Cs1, K = c.ntaps * Cmain + Cskip, ldw = (c.N + 63) / 64 * 64;
    DeepArgs a{};
    std::vector<float> xm[2], xs[2], xr;
    a.main[0] = make_src(c.B, Ls, c.Cm0, c.ks_in, xm[0]);
    if (c.Cm1) a.main[1] = make_src(c.B, Ls, c.Cm1, c.ks_in > 1 ? c.ks_in / 2 : 1, xm[1]);
    if (c.Cs0) a.skip[0] = make_src(c.B, L, c.Cs0, c.ks_in, xs[0]);
    if (c.Cs1) a.skip[1] = make_src(c.B, L, c.Cs1, 1, xs[1]);
    if (c.res) a.res = make_src(c.B, Lr, c.N, c.ks_res, xr);
    a.Cmain = Cmain; a.Cskip = Cskip; a.ntaps = c.ntaps; a.up_main = c.up == 1; a.up_res = c.up_res == 1; a.pool_main = c.up == 2; a.pool_res = c.up_res == 2; a.r = r; a.t = t;
    a.B = c.B; a.Lout = L; a.Lsrc = Ls; a.Lres = Lr; a.N = c.N;
    std::vector<float> W((size_t)K * ldw), bias(c.N), bias2(c.N), gamma(Cmain), beta(Cmain), film((size_t)c.B * 2 * Cmain);
    const float wsc = 1.0f / sqrtf((float)K);
    for (auto& v : W) v = frand() * wsc;
    for (auto& v : bias) v = frand() * 0.1f;
    for (auto& v : bias2) v = frand() * 0.1f;
    for (auto& v : gamma) v = 1.0f + 0.3f * frand();
    for (auto& v : beta) v = 0.2f * frand();
    for (auto& v : film) v = 0.3f * frand();
    float* dW = dup(W);
    a.bias = dup(bias);
    if (Cskip) a.bias2 = dup(bias2);
    a.gn = c.gn; a.whole = c.whole; a.act = c.act; a.gs = Cmain / 32;
    if (c.gn) { a.gamma = dup(gamma); a.beta = dup(beta); }
    if (c.film) { a.film = dup(film); a.film_stride = 2 * Cmain; }
    a.KS = c.KS; a.CSm = Cmain / c.KS; a.CSs = Cskip / c.KS; a.nrg = c.nrg;
    a.zeros = dnew<float>(2 * Cmain);
    float* out = dnew<float>((size_t)c.KS * c.B * L * c.N);
    a.out = out; a.out_slab_stride = (unsigned)((size_t)c.B * L * c.N);
    DeepTile tl{};
    if (!deep_tile_for(a, &tl)) { printf("%-28s no tile\n", c.name); return 1; }
    float* dWd = dnew<float>(deep_weight_floats(a, tl.NT));
    a.tiles_n = c.N / (16 * tl.NT);
    CK(launch_deep_repack(dW, ldw, dWd, a, tl.NT, 0));
    a.W = dWd;
    {
        const std::vector<int> tab = deep_rowtab(a, tl);
        int* dt = dnew<int>(tab.size());
        CK(hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        a.rowtab = dt;
    }
    CK(hipMemset(out, 0xFF, (size_t)c.KS * c.B * L * c.N * 4));        // poison: NaN wherever nothing is written
    CK(launch_deep_conv(a, tl, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> got((size_t)c.KS * c.B * L * c.N);
    CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
    // ---- CPU reference (double accumulation)
    std::vector<double> ref((size_t)c.B * L * c.N, 0.0);
    auto xmain = [&](int b, int tok, int ch) -> float { return ch < c.Cm0 ? xm[0][((size_t)b * Ls + tok) * c.Cm0 + ch] : xm[1][((size_t)b * Ls + tok) * c.Cm1 + ch - c.Cm0]; };
    std::vector<float> act((size_t)c.B * Ls * Cmain);
    for (int b = 0; b < c.B; ++b) {
        const int gs = Cmain / 32;
        for (int g = 0; g < 32; ++g)
            for (int p = 0; p < (c.whole ? 1 : 3); ++p) {
                const int t0 = c.whole ? 0 : (p == 0 ? 0 : (p == 1 ? b1s : b2s)), t1 = c.whole ? Ls : (p == 0 ? b1s : (p == 1 ? b2s : Ls));
                double s = 0, ss = 0;
                for (int tok = t0; tok < t1; ++tok)
                    for (int ch = g * gs; ch < (g + 1) * gs; ++ch) { const double v = xmain(b, tok, ch); s += v; ss += v * v; }
                const double n = (double)(t1 - t0) * gs, mean = s / n, var = std::max(0.0, ss / n - mean * mean), rstd = 1.0 / sqrt(var + 1e-5);
                for (int tok = t0; tok < t1; ++tok)
                    for (int ch = g * gs; ch < (g + 1) * gs; ++ch) {
                        double v = xmain(b, tok, ch);
                        if (c.gn) {
                            v = (v - mean) * rstd * gamma[ch] + beta[ch];
                            if (c.film) v = v * (1.0 + film[(size_t)b * 2 * Cmain + ch]) + film[(size_t)b * 2 * Cmain + Cmain + ch];
                            if (c.act) v = v / (1.0 + exp(-v));
                        }
                        act[((size_t)b * Ls + tok) * Cmain + ch] = (float)v;
                    }
            }
    }
    for (int b = 0; b < c.B; ++b)
        for (int tok = 0; tok < L; ++tok) {
            double* o = &ref[((size_t)b * L + tok) * c.N];
            for (int tap = 0; tap < c.ntaps; ++tap) {
                int src;
                if (c.ntaps == 9) {
                    const int g = geo_source(r, t, tok, tap / 3, tap % 3, c.up == 1);
                    if (g < 0) continue;
                    src = g & 0x0FFFFFFF;
                } else src = c.up == 1 ? (geo_source(r, t, tok, 1, 1, true) & 0x0FFFFFFF) : tok;
                int f4[4] = {src, src, src, src};
                if (c.up == 2) finer4(src, f4);                       // the conv reads the 2x2 mean of the transformed finer-level rows
                for (int ch = 0; ch < Cmain; ++ch) {
                    const double av = c.up == 2 ? 0.25 * ((double)act[((size_t)b * Ls + f4[0]) * Cmain + ch] + act[((size_t)b * Ls + f4[1]) * Cmain + ch] +
                                                          act[((size_t)b * Ls + f4[2]) * Cmain + ch] + act[((size_t)b * Ls + f4[3]) * Cmain + ch])
                                                : (double)act[((size_t)b * Ls + src) * Cmain + ch];
                    const float* wr = &W[((size_t)tap * Cmain + ch) * ldw];
                    for (int n = 0; n < c.N; ++n) o[n] += av * wr[n];
                }
            }
            for (int ch = 0; ch < Cskip; ++ch) {
                const double av = ch < c.Cs0 ? xs[0][((size_t)b * L + tok) * c.Cs0 + ch] : xs[1][((size_t)b * L + tok) * c.Cs1 + ch - c.Cs0];
                const float* wr = &W[((size_t)c.ntaps * Cmain + ch) * ldw];
                for (int n = 0; n < c.N; ++n) o[n] += av * wr[n];
            }
            const int rtok = c.up_res == 1 ? (geo_source(r, t, tok, 1, 1, true) & 0x0FFFFFFF) : tok;
            int r4[4] = {rtok, rtok, rtok, rtok};
            if (c.up_res == 2) finer4(tok, r4);
            for (int n = 0; n < c.N; ++n) {
                o[n] += bias[n] + (Cskip ? bias2[n] : 0.f);
                if (c.res) o[n] += c.up_res == 2 ? 0.25 * ((double)xr[((size_t)b * Lr + r4[0]) * c.N + n] + xr[((size_t)b * Lr + r4[1]) * c.N + n] +
                                                          xr[((size_t)b * Lr + r4[2]) * c.N + n] + xr[((size_t)b * Lr + r4[3]) * c.N + n])
                                                : (double)xr[((size_t)b * Lr + rtok) * c.N + n];
            }
        }
    double worst = 0.0, scale = 0.0;
    bool nan = false;
    for (size_t e = 0; e < ref.size(); ++e) {
        double s = 0;
        for (int k = 0; k < c.KS; ++k) { const float v = got[(size_t)k * ref.size() + e]; if (v != v) nan = true; s += v; }
        worst = std::max(worst, fabs(s - ref[e]));
        scale = std::max(scale, fabs(ref[e]));
    }
    const bool ok = !nan && worst <= 2e-5 * std::max(1.0, scale) * 4;
    printf("%-28s tile RT%d NT%d KS%d nrg%d  max|err| %.3e (|ref| <= %.2f)%s  %s\n", c.name, tl.RT, tl.NT, c.KS, c.nrg, worst, scale, nan ? " NaN" : "", ok ? "PASS" : "FAIL");
    return ok ? 0 : 1;
}

static int do_check() {
    const Case cases[] = {
        // name                       r  t up taps Cm0  Cm1  Cs0  Cs1   N  gn wh act film res upr ksi ksr KS nrg B
        {"m32 conv3 gn",              4, 2, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 0, 0, 0, 1, 1, 8, 1, 1},
        {"m32 conv3 gn film res ks",  4, 2, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 1, 1, 0, 8, 4, 8, 1, 1},
        {"m32 conv3 cat+skipconv",    4, 2, 0, 9, 128,   0, 128, 128, 128, 1, 0, 1, 1, 0, 0, 4, 1, 4, 1, 1},
        {"m32 conv3 cat main",        4, 2, 0, 9, 128, 128,   0,   0, 128, 1, 0, 1, 0, 0, 0, 4, 1, 8, 1, 1},
        {"m32 1x1 whole qkv",         4, 2, 0, 1, 128,   0,   0,   0, 384, 1, 1, 0, 0, 0, 0, 8, 1, 8, 1, 1},
        {"m32 1x1 raw + res",         4, 2, 0, 1, 128,   0,   0,   0, 128, 0, 0, 0, 0, 1, 0, 1, 8, 2, 1, 1},
        {"m128 conv3 rg2",            8, 4, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 1, 1, 0, 4, 4, 4, 2, 1},
        {"m128 conv3 rg1",            8, 4, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 0, 0, 0, 2, 1, 8, 1, 1},
        {"m128 conv3 up rg2",         8, 4, 1, 9, 128,   0,   0,   0, 128, 1, 0, 1, 0, 0, 0, 8, 1, 4, 2, 1},
        {"m128 conv3 up_res rg2",     8, 4, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 1, 1, 1, 4, 8, 4, 2, 1},
        {"m128 1x1 whole rg1 NT3",    8, 4, 0, 1, 256,   0,   0,   0, 768, 1, 1, 0, 0, 0, 0, 4, 1, 8, 1, 1},
        {"m128 1x1 2d rg2",           8, 4, 0, 1, 128,   0,   0,   0, 384, 1, 0, 0, 0, 0, 0, 4, 1, 4, 2, 1},
        {"m128 B2 conv3 rg2",         8, 4, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 1, 1, 0, 4, 2, 4, 2, 2},
        {"ragged r6 t3 conv3",        6, 3, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 0, 1, 0, 2, 2, 4, 2, 1},
        {"ragged r3 t1 conv3 rg1",    3, 1, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 0, 0, 0, 2, 1, 8, 1, 1},
        {"full m32 k4608",            4, 2, 0, 9, 512,   0,   0,   0, 512, 1, 0, 1, 1, 1, 0, 8, 8, 8, 1, 1},
        {"full m32 out conv2 k5632",  4, 2, 0, 9, 512,   0, 512, 512, 512, 1, 0, 1, 1, 0, 0, 8, 1, 8, 1, 1},
        {"full m128 k9216 rg2",       8, 4, 0, 9, 512, 512,   0,   0, 512, 1, 0, 1, 0, 0, 0, 4, 1, 4, 2, 1},
        {"full m128 qkv whole",       8, 4, 0, 1, 512,   0,   0,   0, 1536, 1, 1, 0, 0, 0, 0, 4, 1, 8, 1, 1},
        // ResBlock(down=True): AvgPool2d folded in (up = 2: conv1 reads the pooled, transformed finer level; upr = 2: conv2's residual is the pooled raw input)
        {"pool m32 conv1 (in9.0.h1)",  4, 2, 2, 9, 512,   0,   0,   0, 512, 1, 0, 1, 0, 0, 0, 4, 1, 8, 1, 1},
        {"pool m32 conv2 res ks8",     4, 2, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 1, 1, 2, 8, 8, 8, 1, 1},
        {"pool m32 conv2 res ks4 KS8", 4, 2, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 1, 1, 2, 8, 4, 8, 1, 1},
        {"pool m128 conv1 rg2 (in6.0)",8, 4, 2, 9, 256,   0,   0,   0, 256, 1, 0, 1, 0, 0, 0, 1, 1, 8, 2, 1},
        {"pool m128 conv2 res plain",  8, 4, 0, 9, 128,   0,   0,   0, 128, 1, 0, 1, 1, 1, 2, 8, 1, 4, 2, 1},
        {"pool m128 B2 conv1 rg2",     8, 4, 2, 9, 128,   0,   0,   0, 128, 1, 0, 1, 0, 0, 0, 2, 1, 4, 2, 2},
        {"pool ragged r6 t3 conv1",    6, 3, 2, 9, 128,   0,   0,   0, 128, 1, 0, 1, 0, 1, 2, 2, 2, 4, 2, 1},
    };
    int bad = 0;
    for (auto& c : cases) bad += run_case(c);
    printf("%s (%d failing)\n", bad ? "CHECK FAILED" : "CHECK OK", bad);
    return bad;
}

// ---------------------------------------------------------------------------------------------------------------- chain timing
static void do_chain(int nops, int r, int t, int ks_arg, int nrg_arg) {
    const int C = 512, N = 512, K = 9 * C, ldw = 512, b1 = r * r, b2 = b1 + t * r, L = b2 + t * r;
    printf("chain of %d dependent convs [%d x %d, K = %d] (r %d t %d), distinct weights %.1f MB each (%.0f MB in all), GroupNorm + SiLU between ops\n", nops, L, N, K, r, t,
           K * N * 4e-6, nops * K * N * 4e-6);
    std::vector<float> W((size_t)K * ldw), ones(C, 1.0f), zeros(C, 0.f), x((size_t)L * C);
    const float wsc = 1.0f / sqrtf((float)K);
    for (auto& v : x) v = frand();
    std::vector<float*> dW(nops), dWd(nops);
    float *gamma = dup(ones), *beta = dup(zeros), *bias = dup(zeros);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    // ---- legacy chain: k_conv with the tile of the product's plan (t1,2,8,8x at 32 tokens, t1,4,8,4x at 128), statistics by epilogue atomics
    CK(conv_init_attrs());
    CK(deep_init_attrs());
    std::vector<int> g((size_t)9 * L, -1);
    for (int tok = 0; tok < L; ++tok)
        for (int tap = 0; tap < 9; ++tap) { const int gg = geo_source(r, t, tok, tap / 3, tap % 3, false); g[(size_t)tap * L + tok] = gg < 0 ? -1 : (gg & 0x0FFFFFFF); }
    int* dg = dnew<int>(g.size());
    CK(hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    for (int o = 0; o < nops; ++o) {
        for (auto& v : W) v = frand() * wsc;
        dW[o] = dup(W);
    }
    float* actA[2] = {dup(x), dnew<float>((size_t)L * C)};
    double* sites = dnew<double>((size_t)(nops + 1) * STAT_COPIES * 192);
    const unsigned cstride = (unsigned)((nops + 1) * 192);
    float* slab = dnew<float>((size_t)16 * L * N);
    int* tickets = dnew<int>(1 << 16);
    const ConvTile tile = L <= 32 ? ConvTile{1, 2, 8, 8, 1} : ConvTile{1, 4, 8, 4, 1};
    std::vector<ConvArgs> ca(nops);
    for (int o = 0; o < nops; ++o) {
        ConvArgs a{};
        a.src[0] = actA[o & 1]; a.C[0] = C; a.nmain = 1; a.Cmain = C; a.gather = dg; a.ntaps = 9; a.Lout = L; a.Lsrc = L; a.Lskip = L; a.B = 1;
        a.W = dW[o]; a.ldw = ldw; a.N = N; a.bias = bias; a.out = actA[(o + 1) & 1];
        a.seg_src = SegInfo{b1, b2, L}; a.seg_out = a.seg_src; a.slab = slab; a.tickets = tickets;
        a.geo_main = 1; a.geo_r = r; a.geo_t = t;
        a.gn = GnIn{sites + (size_t)o * 192, gamma, beta, nullptr, 0, C / 32, 0, 1, cstride};
        a.gn.inv_gs = 1.0f / (C / 32);
        a.gn.inv_n[0] = 1.0 / ((double)b1 * (C / 32)); a.gn.inv_n[1] = 1.0 / ((double)(b2 - b1) * (C / 32)); a.gn.inv_n[2] = 1.0 / ((double)(L - b2) * (C / 32)); a.gn.inv_n[3] = 1.0 / ((double)L * (C / 32));
        a.stat[0] = StatOut{sites + (size_t)(o + 1) * 192, C / 32, 0, 1.0f / (C / 32)};
        a.nstat = 1; a.stat_cstride = cstride;
        ca[o] = a;
    }
    auto time_graph = [&](const char* name, std::function<void()> body, int reps) {
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        body();
        CK(hipStreamEndCapture(s, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("  %-34s %9.2f us per chain  %7.2f us per op  %6.2f TB/s of weights  (%.2f us per 10 ops)\n", name, us, us / nops, nops * K * N * 4e-6 / us, 10 * us / nops);
        return us;
    };
    // statistics of the chain input (site 0): computed once on the host so that op 0 normalises like the others
    {
        std::vector<double> hs(192, 0.0);
        for (int tok = 0; tok < L; ++tok)
            for (int ch = 0; ch < C; ++ch) { const int p = tok >= b2 ? 2 : (tok >= b1 ? 1 : 0); const double v = x[(size_t)tok * C + ch]; hs[(p * 32 + ch / 16) * 2] += v; hs[(p * 32 + ch / 16) * 2 + 1] += v * v; }
        CK(hipMemcpy(sites, hs.data(), 192 * 8, hipMemcpyHostToDevice));
    }
    const double t_old = time_graph("k_conv chain (product tile)", [&]() {
        CK(hipMemsetAsync(sites + 192, 0, (size_t)((nops + 1) * STAT_COPIES * 192 - 192) * 8, s));     // (one memset per chain: the product zeroes in the head conv)
        for (int o = 0; o < nops; ++o) CK(launch_conv(ca[o], tile, s));
    }, 30);
    // ---- deep chain
    std::vector<DeepArgs> da(nops);
    const int KS = ks_arg ? ks_arg : (L <= 32 ? 8 : 4), nrg = nrg_arg ? nrg_arg : (L <= 32 ? 1 : 2);
    printf("  deep: KS %d nrg %d\n", KS, nrg);
    float* slabs[2] = {dnew<float>((size_t)8 * L * C), dnew<float>((size_t)8 * L * C)};
    float* zeros_d = dnew<float>(2 * C);
    int* rowtab_d = nullptr;
    CK(hipMemcpy(slabs[0], x.data(), x.size() * 4, hipMemcpyHostToDevice));
    DeepTile tl{};
    for (int o = 0; o < nops; ++o) {
        DeepArgs a{};
        a.main[0] = DeepSrc{slabs[o & 1], (unsigned)((size_t)L * C), o == 0 ? 1 : KS, C};
        a.Cmain = C; a.ntaps = 9; a.r = r; a.t = t; a.B = 1; a.Lout = L; a.Lsrc = L; a.Lres = L; a.N = N;
        a.bias = bias; a.gamma = gamma; a.beta = beta; a.gs = C / 32; a.gn = 1; a.act = 1;
        a.out = slabs[(o + 1) & 1]; a.out_slab_stride = (unsigned)((size_t)L * N);
        a.KS = KS; a.CSm = C / KS; a.nrg = nrg;
        a.zeros = zeros_d;
        if (!deep_tile_for(a, &tl)) { printf("no tile\n"); exit(1); }
        a.tiles_n = N / (16 * tl.NT);
        dWd[o] = dnew<float>((size_t)K * N);
        CK(launch_deep_repack(dW[o], ldw, dWd[o], a, tl.NT, 0));
        a.W = dWd[o];
        if (o == 0) {
            const std::vector<int> tab = deep_rowtab(a, tl);
            rowtab_d = dnew<int>(tab.size());
            CK(hipMemcpy(rowtab_d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        }
        a.rowtab = rowtab_d;
#ifdef MTV_DEEP_STAMP
        if (o == nops / 2) a.dbg = dnew<unsigned long long>(64);
#endif
        da[o] = a;
    }
    CK(hipDeviceSynchronize());
    {   // the two chains compute the same function: compare after 2 ops (40 chaotic layers amplify rounding differences to O(1))
        CK(hipMemcpy(slabs[0], x.data(), x.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(actA[0], x.data(), x.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemsetAsync(sites + 192, 0, (size_t)((nops + 1) * STAT_COPIES * 192 - 192) * 8, s));
        for (int o = 0; o < 2; ++o) { CK(launch_conv(ca[o], tile, s)); CK(launch_deep_conv(da[o], tl, s)); }
        CK(hipStreamSynchronize(s));
        std::vector<float> a1((size_t)L * C), a2((size_t)KS * L * C);
        CK(hipMemcpy(a1.data(), actA[0], a1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(a2.data(), slabs[0], a2.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, sc = 0;
        for (size_t e = 0; e < a1.size(); ++e) {
            float v = 0;
            for (int k = 0; k < KS; ++k) v += a2[(size_t)k * a1.size() + e];
            worst = std::max(worst, (double)fabsf(v - a1[e]));
            sc = std::max(sc, (double)fabsf(a1[e]));
        }
        printf("  after 2 ops, k_conv vs k_deep_conv: max|diff| %.3e (|x| <= %.2f)\n", worst, sc);
        CK(hipMemcpy(slabs[0], x.data(), x.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(actA[0], x.data(), x.size() * 4, hipMemcpyHostToDevice));
    }
    const double t_new = time_graph("k_deep_conv chain", [&]() { for (int o = 0; o < nops; ++o) CK(launch_deep_conv(da[o], tl, s)); }, 30);
    printf("  ratio k_conv / k_deep_conv = %.2fx\n", t_old / t_new);
#ifdef MTV_DEEP_STAMP
    {
        unsigned long long h[64];
        CK(hipMemcpy(h, da[nops / 2].dbg, sizeof h, hipMemcpyDeviceToHost));
        static const char* nm[16] = {"entry", "decoded", "stats added", "barrier 1", "transformed+barrier", "mfma done", "barrier", "end", "tables built", "main staged", "skip staged", "-", "slabs issued", "weights issued", "barrier 0", "first group done"};
        static const int order[2][16] = {{1, 12, 13, 8, 14, 9, 10, 2, 3, 4, 5, 6, 7, -1}, {1, 12, 13, 8, 14, 9, 10, 2, 3, 4, 5, 6, 7, -1}};
        for (int blk = 0; blk < 2; ++blk)
            for (int role = 0; role < 2; ++role) {
                printf("  stamps op %d wg %s %s (cycles since entry):", nops / 2, blk ? "mid" : "0", role ? "wave 4" : "wave 0");
                for (int k = 0; order[role][k] >= 0; ++k) printf("  %s %lld", nm[order[role][k]], (long long)(h[blk * 32 + role * 16 + order[role][k]] - h[blk * 32 + role * 16]));
                printf("\n");
            }
    }
#endif
    // the two chains compute the same function of the same input: compare the final activations
    {
        std::vector<float> a1((size_t)L * C), a2((size_t)KS * L * C);
        CK(hipMemcpy(a1.data(), actA[nops & 1], a1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(a2.data(), slabs[nops & 1], a2.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, sc = 0;
        for (size_t e = 0; e < a1.size(); ++e) {
            float v = 0;
            for (int k = 0; k < KS; ++k) v += a2[(size_t)k * a1.size() + e];
            worst = std::max(worst, (double)fabsf(v - a1[e]));
            sc = std::max(sc, (double)fabsf(a1[e]));
        }
        printf("  final activations of the two chains: max|diff| %.3e (|x| <= %.2f)\n", worst, sc);
    }
}


// ---------------------------------------------------------------------------------------------------------------- k_deep_attn
static int run_attn(int r, int t, int C, int H, int whole, int res_ks, bool timing) {
    const int b1 = r * r, b2 = b1 + t * r, L = b2 + t * r, d = C / H, ldw = (C + 63) / 64 * 64, B = 1;
    std::vector<float> qkv((size_t)B * L * 3 * C), W((size_t)C * ldw), bias(C), xr;
    for (auto& v : qkv) v = frand();
    const float wsc = 1.0f / sqrtf((float)C);
    for (auto& v : W) v = frand() * wsc;
    for (auto& v : bias) v = frand() * 0.1f;
    DeepAttnArgs a{};
    a.qkv = dup(qkv); a.B = B; a.L = L; a.C = C; a.H = H; a.r = r; a.t = t; a.whole = whole;
    a.scale = 1.0f / sqrtf(sqrtf((float)d));
    a.Wp = dup(W); a.ldw = ldw; a.bias = dup(bias);
    a.res = make_src(B, L, C, res_ks, xr);
    if (!deep_attn_configure(a)) { printf("attn L%d d%d: not configurable\n", L, d); return 1; }
    float* out = dnew<float>((size_t)8 * B * L * C);
    a.out = out; a.out_slab_stride = (unsigned)((size_t)B * L * C);
#ifdef MTV_DEEP_STAMP
    a.dbg = dnew<unsigned long long>(64);
#endif
    CK(hipMemset(out, 0xFF, (size_t)8 * B * L * C * 4));
    CK(launch_deep_attn(a, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> got((size_t)a.nhg * B * L * C);
    CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
    // CPU reference (double): softmax(q k^T d^-1/2) v per head over the keys the query may see, then proj + bias + residual
    std::vector<double> att((size_t)L * C), ref((size_t)L * C);
    for (int h = 0; h < H; ++h)
        for (int q = 0; q < L; ++q) {
            const int qp = q >= b2 ? 2 : (q >= b1 ? 1 : 0);
            std::vector<double> sc(L, -1e300);
            double mx = -1e300;
            for (int k = 0; k < L; ++k) {
                const int kp = k >= b2 ? 2 : (k >= b1 ? 1 : 0);
                if (!whole && kp != qp) continue;
                double s = 0;
                for (int e = 0; e < d; ++e) s += (double)qkv[(size_t)q * 3 * C + h * 3 * d + e] * qkv[(size_t)k * 3 * C + h * 3 * d + d + e];
                sc[k] = s / sqrt((double)d);
                mx = std::max(mx, sc[k]);
            }
            double den = 0;
            for (int k = 0; k < L; ++k) if (sc[k] > -1e299) { sc[k] = exp(sc[k] - mx); den += sc[k]; } else sc[k] = 0;
            for (int e = 0; e < d; ++e) {
                double o = 0;
                for (int k = 0; k < L; ++k) o += sc[k] * qkv[(size_t)k * 3 * C + h * 3 * d + 2 * d + e];
                att[(size_t)q * C + h * d + e] = o / den;
            }
        }
    for (int q = 0; q < L; ++q)
        for (int n = 0; n < C; ++n) {
            double o = bias[n] + xr[(size_t)q * C + n];
            for (int k = 0; k < C; ++k) o += att[(size_t)q * C + k] * W[(size_t)k * ldw + n];
            ref[(size_t)q * C + n] = o;
        }
    double worst = 0, sc2 = 0;
    bool nan = false;
    for (size_t e = 0; e < ref.size(); ++e) {
        double s2 = 0;
        for (int k = 0; k < a.nhg; ++k) { const float v = got[(size_t)k * ref.size() + e]; if (v != v) nan = true; s2 += v; }
        worst = std::max(worst, fabs(s2 - ref[e]));
        sc2 = std::max(sc2, fabs(ref[e]));
    }
    const bool ok = !nan && worst <= 1e-4 * std::max(1.0, sc2);
    printf("attn+proj L%-3d C%-3d d%-2d %s res_ks%d  HPW%d NC%d -> %d slabs, %d WGs  max|err| %.3e (|ref| <= %.2f)%s  %s\n", L, C, d, whole ? "1d" : "2d", res_ks, a.HPW, a.NC,
           a.nhg, a.ncg * a.B * a.nqg * a.nhg, worst, sc2, nan ? " NaN" : "", ok ? "PASS" : "FAIL");
    if (timing) {
        hipStream_t s; CK(hipStreamCreate(&s));
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 20; ++i) CK(launch_deep_attn(a, s));
        CK(hipStreamEndCapture(s, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("    %.2f us per launch (graph of 20 back-to-back launches, boundary included)\n", ms * 1e3 / 400);
#ifdef MTV_DEEP_STAMP
        unsigned long long h[64];
        CK(hipMemcpy(h, a.dbg, sizeof h, hipMemcpyDeviceToHost));
        for (int blk = 0; blk < 2; ++blk) {
            printf("    stamps wg %s (cycles since entry):", blk ? "mid" : "0");
            for (int k = 1; k < 12; ++k) if (h[blk * 32 + k]) printf(" [%d] %lld", k, (long long)(h[blk * 32 + k] - h[blk * 32]));
            printf("\n");
        }
#endif
    }
    return ok ? 0 : 1;
}
static int do_attn(bool timing) {
    int bad = 0;
    bad += run_attn(4, 2, 512, 8, 1, 1, timing);
    bad += run_attn(4, 2, 512, 8, 0, 4, timing);
    bad += run_attn(8, 4, 512, 8, 1, 1, timing);
    bad += run_attn(8, 4, 512, 8, 0, 8, timing);
    bad += run_attn(8, 4, 128, 8, 1, 2, false);       // the test-size model: d = 16
    bad += run_attn(8, 4, 128, 8, 0, 1, false);
    bad += run_attn(8, 4, 256, 8, 1, 1, false);       // d = 32
    bad += run_attn(6, 3, 256, 8, 0, 1, false);       // ragged planes 36 | 18 | 18
    bad += run_attn(3, 1, 128, 2, 0, 1, false);       // 15 tokens, d = 64
    printf("%s (%d failing)\n", bad ? "ATTN CHECK FAILED" : "ATTN CHECK OK", bad);
    return bad;
}


// ---------------------------------------------------------------------------------------------------------------- k_deep_block
// the whole attention block in one launch (block.hip) against a double-precision CPU restatement of
// GroupNorm -> qkv -> QKVAttentionLegacy -> proj_out + residual; the reduced qkv scratch is checked too (localises a failure)
static int run_block(int r, int t, int C, int H, int whole, int x_ks, int B, int force_cl, bool timing, int force_rq = 0) {
    const int b1 = r * r, b2 = b1 + t * r, L = b2 + t * r, d = C / H, gs = C / 32 > 0 ? C / 32 : 1;
    std::vector<float> Wq((size_t)3 * C * C), bq(3 * C), Wp((size_t)C * C), bp(C), ga(C), be(C), xr;
    const float wsc = 1.0f / sqrtf((float)C);
    for (auto& v : Wq) v = frand() * wsc;
    for (auto& v : Wp) v = frand() * wsc;
    for (auto& v : bq) v = frand() * 0.1f;
    for (auto& v : bp) v = frand() * 0.1f;
    for (auto& v : ga) v = 1.0f + 0.2f * frand();
    for (auto& v : be) v = 0.2f * frand();
    DeepBlockArgs a{};
    a.x = make_src(B, L, C, x_ks, xr);
    a.B = B; a.L = L; a.C = C; a.H = H; a.r = r; a.t = t; a.whole = whole;
    a.scale = 1.0f / sqrtf(sqrtf((float)d));
    a.gamma = dup(ga); a.beta = dup(be); a.gs = gs;
    a.Wq = dup(Wq); a.bq = dup(bq); a.Wp = dup(Wp); a.bp = dup(bp);
    if (!deep_block_configure(a, force_cl, force_rq)) { printf("block L%d C%d d%d cl%d rq%d: not configurable\n", L, C, d, force_cl, force_rq); return (force_cl || force_rq) ? 0 : 1; }
    if (a.RQ > 1) a.stg = dnew<float>(deep_block_stg_floats(a));
    const size_t slab = (size_t)B * L * C;
    float* out = dnew<float>((size_t)8 * slab);
    a.out = out; a.out_slab_stride = (unsigned)slab;
    a.part = dnew<float>(deep_block_part_floats(a));
    a.qkv = dnew<float>(deep_block_qkv_floats(a));
    a.cnt = dnew<unsigned long long>((size_t)B * H * 2);
    a.fault = dnew<int>(16);
#ifdef MTV_DEEP_STAMP
    a.dbg = dnew<unsigned long long>(64);
#endif
    CK(hipMemset(out, 0xFF, (size_t)8 * slab * 4));
    int fault = 0;
    std::vector<float> got((size_t)H * slab), gq(deep_block_qkv_floats(a));
    // twice: the second launch runs on counters the first one left behind (monotonic, never reset)
    for (int rep = 0; rep < 2; ++rep) {
        CK(launch_deep_block(a, 0));
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(&fault, a.fault, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gq.data(), a.qkv, gq.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, sc2 = 0, worst_q = 0;
    bool nan = false;
    for (int b = 0; b < B; ++b) {
        const float* x = xr.data() + (size_t)b * L * C;
        // GroupNorm (biased variance, eps 1e-5) per plane or over all planes
        std::vector<double> xn((size_t)L * C), qkv((size_t)L * 3 * C), att((size_t)L * C);
        for (int g = 0; g < C / gs; ++g)
            for (int p = 0; p < (whole ? 1 : 3); ++p) {
                const int t0 = whole ? 0 : (p == 0 ? 0 : (p == 1 ? b1 : b2)), t1 = whole ? L : (p == 0 ? b1 : (p == 1 ? b2 : L));
                double sm = 0, sq = 0;
                for (int tk = t0; tk < t1; ++tk)
                    for (int c = g * gs; c < (g + 1) * gs; ++c) { sm += x[(size_t)tk * C + c]; sq += (double)x[(size_t)tk * C + c] * x[(size_t)tk * C + c]; }
                const double n = (double)(t1 - t0) * gs, mean = sm / n, var = sq / n - mean * mean, rstd = 1.0 / sqrt(var + 1e-5);
                for (int tk = t0; tk < t1; ++tk)
                    for (int c = g * gs; c < (g + 1) * gs; ++c) xn[(size_t)tk * C + c] = (x[(size_t)tk * C + c] - mean) * rstd * ga[c] + be[c];
            }
        for (int tk = 0; tk < L; ++tk)
            for (int n = 0; n < 3 * C; ++n) {
                double o = bq[n];
                for (int c = 0; c < C; ++c) o += xn[(size_t)tk * C + c] * Wq[(size_t)n * C + c];
                qkv[(size_t)tk * 3 * C + n] = o;
                const int hh = n / (3 * d), col = n - hh * 3 * d;
                const float gv = gq[((((size_t)b * H + hh) * L + tk) * 3 * d + col) * 2];      // (8-byte {value, tag} granules)
                if (gv != gv) nan = true;
                worst_q = std::max(worst_q, fabs((double)gv - o));
            }
        for (int h = 0; h < H; ++h)
            for (int q = 0; q < L; ++q) {
                const int qp = q >= b2 ? 2 : (q >= b1 ? 1 : 0);
                std::vector<double> sc(L, -1e300);
                double mx = -1e300;
                for (int k = 0; k < L; ++k) {
                    const int kp = k >= b2 ? 2 : (k >= b1 ? 1 : 0);
                    if (!whole && kp != qp) continue;
                    double sx = 0;
                    for (int e = 0; e < d; ++e) sx += qkv[(size_t)q * 3 * C + h * 3 * d + e] * qkv[(size_t)k * 3 * C + h * 3 * d + d + e];
                    sc[k] = sx / sqrt((double)d);
                    mx = std::max(mx, sc[k]);
                }
                double den = 0;
                for (int k = 0; k < L; ++k) if (sc[k] > -1e299) { sc[k] = exp(sc[k] - mx); den += sc[k]; } else sc[k] = 0;
                for (int e = 0; e < d; ++e) {
                    double o = 0;
                    for (int k = 0; k < L; ++k) o += sc[k] * qkv[(size_t)k * 3 * C + h * 3 * d + 2 * d + e];
                    att[(size_t)q * C + h * d + e] = o / den;
                }
            }
        for (int q = 0; q < L; ++q)
            for (int n = 0; n < C; ++n) {
                double o = bp[n] + x[(size_t)q * C + n];
                for (int k = 0; k < C; ++k) o += att[(size_t)q * C + k] * Wp[(size_t)n * C + k];
                double s2 = 0;
                for (int k = 0; k < H; ++k) { const float v = got[(size_t)k * slab + ((size_t)b * L + q) * C + n]; if (v != v) nan = true; s2 += v; }
                worst = std::max(worst, fabs(s2 - o));
                sc2 = std::max(sc2, fabs(o));
            }
    }
    const bool ok = !nan && !fault && worst <= 1e-4 * std::max(1.0, sc2) && worst_q <= 1e-4 * 8;
    printf("block L%-3d C%-3d d%-2d H%d %s x_ks%d B%d  cl%d cs%d rq%d -> %d WGs, %zu B LDS  qkv max|err| %.3e  out max|err| %.3e (|ref| <= %.2f)%s%s  %s\n", L, C, d, H,
           whole ? "1d" : "2d", x_ks, B, a.CL, a.CS, a.RQ, B * H * a.CL, deep_block_smem_bytes(a), worst_q, worst, sc2, nan ? " NaN" : "", fault ? " FAULT(hand-off timeout)" : "",
           ok ? "PASS" : "FAIL");
    if (timing && ok) {
        hipStream_t s; CK(hipStreamCreate(&s));
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 20; ++i) CK(launch_deep_block(a, s));
        CK(hipStreamEndCapture(s, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&fault, a.fault, 4, hipMemcpyDeviceToHost));
        printf("    %.2f us per launch (graph of 20 back-to-back launches, boundary included)%s\n", ms * 1e3 / 400, fault ? "  FAULT" : "");
        // bit-equal repeats: the hand-offs must never deliver a stale partial
        std::vector<float> again(got.size());
        CK(hipMemcpy(again.data(), out, again.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t e = 0; e < got.size(); ++e) diff += memcmp(&again[e], &got[e], 4) != 0;
        printf("    %zu of %zu output words differ from the first run after 460 more launches%s\n", diff, got.size(), diff ? "  NOT BIT-EQUAL" : "");
        if (diff || fault) return 1;
#ifdef MTV_DEEP_STAMP
        unsigned long long hst[32];
        CK(hipMemcpy(hst, a.dbg, sizeof hst, hipMemcpyDeviceToHost));
        for (int blk = 0; blk < 2; ++blk) {
            printf("    stamps wg %s (100 MHz ticks since entry):", blk ? "mid" : "0");
            for (int k = 1; k < 11; ++k) if (hst[blk * 16 + k]) printf(" [%d] %lld", k, (long long)(hst[blk * 16 + k] - hst[blk * 16]));
            printf("\n");
        }
#endif
    }
    return ok ? 0 : 1;
}
static int do_block(bool timing) {
    int bad = 0;
    bad += run_block(4, 2, 512, 8, 1, 8, 1, 0, timing);       // the base model's 32-token blocks (in9 .. out1): input = 8 slabs
    bad += run_block(4, 2, 512, 8, 0, 8, 1, 0, timing);       // mid.1 (per plane: 16 | 8 | 8 tokens)
    bad += run_block(8, 4, 512, 8, 1, 4, 1, 0, timing);       // 128-token blocks (in7 / in8 / out2 .. out4), AttentionBlock1D
    bad += run_block(8, 4, 512, 8, 0, 4, 1, 0, timing);       // ... per plane (64 | 32 | 32)
    bad += run_block(8, 4, 256, 8, 1, 8, 1, 0, timing);       // in6.a1: d = 32
    if (timing) {                                             // other cluster shapes on the headline shapes
        bad += run_block(4, 2, 512, 8, 1, 8, 1, 8, true);
        bad += run_block(8, 4, 512, 8, 1, 4, 1, 0, true, 1);     // [128 x 512] with every workgroup staging all tokens (16 slices of 32 channels)
        bad += run_block(8, 4, 512, 8, 1, 4, 1, 0, true, 2);     // two row groups (8 slices of 64 channels)
        bad += run_block(8, 4, 512, 8, 0, 4, 1, 0, true, 2);
    }
    bad += run_block(8, 4, 512, 8, 1, 4, 2, 0, false);        // two clips
    bad += run_block(8, 4, 128, 8, 0, 1, 2, 0, false);        // the test-size model: d = 16, groups of 4 channels
    bad += run_block(8, 4, 128, 8, 1, 2, 1, 0, false);
    bad += run_block(6, 3, 256, 8, 0, 1, 1, 0, false);        // ragged planes 36 | 18 | 18 (72 tokens: 5 query tiles)
    bad += run_block(6, 3, 256, 8, 1, 2, 2, 0, false);
    bad += run_block(8, 2, 256, 8, 1, 2, 1, 0, false);        // 96 tokens (LP / 32 = 3 row groups is not launchable -- two groups of 48 rows)
    bad += run_block(8, 2, 512, 8, 0, 4, 2, 0, false);        // ... per plane (64 | 16 | 16), two clips
    bad += run_block(7, 3, 256, 8, 1, 1, 2, 0, false);        // 91 tokens (padded to 96)
    bad += run_block(3, 1, 128, 2, 0, 1, 1, 0, false);        // 15 tokens, d = 64, two heads
    bad += run_block(8, 4, 64, 2, 0, 2, 1, 0, false);         // groups of 2 channels (narrow test model at a small geometry)
    bad += run_block(4, 2, 32, 2, 1, 1, 2, 0, false);         // groups of 1 channel, d = 16
    printf("%s (%d failing)\n", bad ? "BLOCK CHECK FAILED" : "BLOCK CHECK OK", bad);
    return bad;
}


// ---------------------------------------------------------------------------------------------------------------- k_conv_win
// chain of dependent 3x3 convs of a LARGE level ([L x C] -> [L x C], GroupNorm + SiLU in the prologue, statistics by the epilogue),
// distinct weights per op: k_conv tiles against the k_conv_win tiles, us per op inside one hipGraph; stamps of the win kernel
static void do_win(int nops, int r, int t, int C) {
    const int N = C, K = 9 * C, ldw = (C + 63) / 64 * 64, b1 = r * r, b2 = b1 + t * r, L = b2 + t * r;
    printf("win: chain of %d dependent 3x3 convs [%d x %d, K = %d] (r %d t %d), weights %.2f MB each\n", nops, L, N, K, r, t, K * N * 4e-6);
    std::vector<float> W((size_t)K * ldw), ones(C, 1.0f), zeros(C, 0.f), x((size_t)L * C);
    const float wsc = 1.0f / sqrtf((float)K);
    for (auto& v : x) v = frand();
    float *gamma = dup(ones), *beta = dup(zeros), *bias = dup(zeros);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(conv_init_attrs());
    CK(deep_init_attrs());
    std::vector<int> g((size_t)9 * L, -1);
    for (int tok = 0; tok < L; ++tok)
        for (int tap = 0; tap < 9; ++tap) { const int gg = geo_source(r, t, tok, tap / 3, tap % 3, false); g[(size_t)tap * L + tok] = gg < 0 ? -1 : (gg & 0x0FFFFFFF); }
    int* dg = dnew<int>(g.size());
    CK(hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    std::vector<float*> dW(nops);
    for (int o = 0; o < nops; ++o) {
        for (auto& v : W) v = frand() * wsc;
        dW[o] = dup(W);
    }
    float* actA[2] = {dup(x), dnew<float>((size_t)L * C)};
    double* sites = dnew<double>((size_t)(nops + 1) * STAT_COPIES * 192);
    const unsigned cstride = (unsigned)((nops + 1) * 192);
    float* slab = dnew<float>((size_t)16 * L * N);
    int* tickets = dnew<int>(1 << 16);
    unsigned long long* dbg = dnew<unsigned long long>(64);
    std::vector<ConvArgs> ca(nops);
    for (int o = 0; o < nops; ++o) {
        ConvArgs a{};
        a.src[0] = actA[o & 1]; a.C[0] = C; a.nmain = 1; a.Cmain = C; a.gather = dg; a.ntaps = 9; a.Lout = L; a.Lsrc = L; a.Lskip = L; a.B = 1;
        a.W = dW[o]; a.ldw = ldw; a.N = N; a.bias = bias; a.out = actA[(o + 1) & 1];
        a.seg_src = SegInfo{b1, b2, L}; a.seg_out = a.seg_src; a.slab = slab; a.tickets = tickets;
        a.geo_main = 1; a.geo_r = r; a.geo_t = t;
        a.gn = GnIn{sites + (size_t)o * 192, gamma, beta, nullptr, 0, C / 32, 0, 1, cstride};
        a.gn.inv_gs = 1.0f / (C / 32);
        a.gn.inv_n[0] = 1.0 / ((double)b1 * (C / 32)); a.gn.inv_n[1] = 1.0 / ((double)(b2 - b1) * (C / 32)); a.gn.inv_n[2] = 1.0 / ((double)(L - b2) * (C / 32)); a.gn.inv_n[3] = 1.0 / ((double)L * (C / 32));
        a.stat[0] = StatOut{sites + (size_t)(o + 1) * 192, C / 32, 0, 1.0f / (C / 32)};
        a.nstat = 1; a.stat_cstride = cstride;
        if (o == nops / 2) a.dbg = dbg;
        ca[o] = a;
    }
    {
        std::vector<double> hs(192, 0.0);
        for (int tok = 0; tok < L; ++tok)
            for (int ch = 0; ch < C; ++ch) { const int p = tok >= b2 ? 2 : (tok >= b1 ? 1 : 0); const double v = x[(size_t)tok * C + ch]; hs[(p * 32 + ch / (C / 32)) * 2] += v; hs[(p * 32 + ch / (C / 32)) * 2 + 1] += v * v; }
        CK(hipMemcpy(sites, hs.data(), 192 * 8, hipMemcpyHostToDevice));
    }
    std::vector<float> ref;
    auto run_tile = [&](ConvTile tile) {
        if (tile.NW == 80 && !conv_win_eligible(ca[0], tile.MT, tile.NT)) return;
        if (tile.NW != 80 && conv_smem_bytes(ca[0], tile) > 120 * 1024) return;
        CK(hipMemcpy(actA[0], x.data(), x.size() * 4, hipMemcpyHostToDevice));
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(sites + 192, 0, (size_t)((nops + 1) * STAT_COPIES * 192 - 192) * 8, s));
        for (int o = 0; o < nops; ++o) CK(launch_conv(ca[o], tile, s));
        CK(hipStreamEndCapture(s, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        std::vector<float> got((size_t)L * C);
        CK(hipMemcpy(got.data(), actA[nops & 1], got.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        if (ref.empty()) ref = got;
        else for (size_t e = 0; e < got.size(); ++e) worst = std::max(worst, (double)fabsf(got[e] - ref[e]));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        const int reps = 30;
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  tile %d,%d,%d,%d,%d  %7.2f us per op   (max|diff| to the first tile after %d ops %.2e)\n", tile.MT, tile.NT, tile.NW, tile.KS, tile.XM, ms * 1e3 / reps / nops, nops, worst);
#ifdef MTV_DEEP_STAMP
        if (tile.NW == 80) {
            unsigned long long h[64];
            CK(hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost));
            static const char* nm[10] = {"entry", "decoded", "requests issued", "tables+barrier", "stats barrier", "window parked", "mfma done", "partials parked", "-", "end"};
            static const int order[] = {1, 2, 3, 4, 5, 6, 7, 9, -1};
            for (int blk = 0; blk < 2; ++blk)
                for (int role = 0; role < 2; ++role) {
                    printf("    stamps wg %s %s:", blk ? "mid" : "0", role ? "wave 4" : "wave 0");
                    for (int k = 0; order[k] >= 0; ++k) printf("  %s %lld", nm[order[k]], (long long)(h[blk * 32 + role * 16 + order[k]] - h[blk * 32 + role * 16]));
                    printf("\n");
                }
        }
#endif
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(gr));
    };
    const ConvTile tiles[] = {{1, 4, 8, 1, 0}, {2, 4, 8, 1, 0}, {2, 2, 8, 1, 0}, {2, 4, 4, 1, 0}, {1, 2, 8, 1, 0}, {1, 4, 80, 1, 0}, {1, 2, 80, 1, 0}, {2, 2, 80, 1, 0}, {1, 2, 80, 1, 1}, {2, 2, 80, 1, 1}, {2, 4, 80, 1, 0}};
    for (const ConvTile& tl : tiles) run_tile(tl);
}


// ---------------------------------------------------------------------------------------------------------------- k_conv_pw
// 20 back-to-back launches of one qkv-shaped 1x1 conv ([L x K] -> [L x N], GroupNorm prologue, no statistics epilogue) in one hipGraph:
// k_conv tiles against the k_conv_pw tiles, us per launch; stamps of the pw kernel
static void do_pw(int r, int t, int K, int N) {
    const int nops = 20, b1 = r * r, b2 = b1 + t * r, L = b2 + t * r, ldw = (N + 63) / 64 * 64;
    printf("pw: %d launches of a 1x1 conv [%d x %d] -> [%d x %d] (r %d t %d), weights %.2f MB\n", nops, L, K, L, N, r, t, K * N * 4e-6);
    std::vector<float> W((size_t)K * ldw, 0.f), Wnk((size_t)N * K), ones(K, 1.0f), zeros(N > K ? N : K, 0.f), x((size_t)L * K);
    const float wsc = 1.0f / sqrtf((float)K);
    for (auto& v : x) v = frand();
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) { const float v = frand() * wsc; W[(size_t)k * ldw + n] = v; Wnk[(size_t)n * K + k] = v; }
    float *gamma = dup(ones), *beta = dup(zeros), *bias = dup(zeros), *dW = dup(W), *dWnk = dup(Wnk), *dx = dup(x);
    std::vector<float> Wpk((size_t)N * K);                             // k_conv_pw's lane-linear copy (ConvArgs::Wpk; the product: k_repack_pw)
    for (int bn = 0; bn < N / 16; ++bn)
        for (int c = 0; c < K / 16; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) Wpk[(((size_t)bn * (K / 16) + c) * 64 + lane) * 4 + e] = Wnk[(size_t)(16 * bn + (lane & 15)) * K + 16 * c + 4 * (lane >> 4) + e];
    float* dWpk = dup(Wpk);
    float* dy = dnew<float>((size_t)L * N);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(conv_init_attrs());
    CK(deep_init_attrs());
    double* sites = dnew<double>((size_t)STAT_COPIES * 192);
    float* slab = dnew<float>((size_t)16 * L * N);
    int* tickets = dnew<int>(1 << 16);
    unsigned long long* dbg = dnew<unsigned long long>(64);
    {
        std::vector<double> hs((size_t)STAT_COPIES * 192, 0.0);
        for (int tok = 0; tok < L; ++tok)
            for (int ch = 0; ch < K; ++ch) { const int p = tok >= b2 ? 2 : (tok >= b1 ? 1 : 0); const double v = x[(size_t)tok * K + ch]; hs[(p * 32 + ch / (K / 32)) * 2] += v; hs[(p * 32 + ch / (K / 32)) * 2 + 1] += v * v; }
        CK(hipMemcpy(sites, hs.data(), hs.size() * 8, hipMemcpyHostToDevice));
    }
    ConvArgs a{};
    a.src[0] = dx; a.C[0] = K; a.nmain = 1; a.Cmain = K; a.ntaps = 1; a.Lout = L; a.Lsrc = L; a.Lskip = L; a.B = 1;
    a.W = dW; a.Wnk = dWnk; a.Wpk = dWpk; a.ldw = ldw; a.N = N; a.bias = bias; a.out = dy;
    a.seg_src = SegInfo{b1, b2, L}; a.seg_out = a.seg_src; a.slab = slab; a.tickets = tickets;
    a.gn = GnIn{sites, gamma, beta, nullptr, 0, K / 32, 0, 0, 192};
    a.gn.inv_gs = 1.0f / (K / 32);
    a.gn.inv_n[0] = 1.0 / ((double)b1 * (K / 32)); a.gn.inv_n[1] = 1.0 / ((double)(b2 - b1) * (K / 32)); a.gn.inv_n[2] = 1.0 / ((double)(L - b2) * (K / 32)); a.gn.inv_n[3] = 1.0 / ((double)L * (K / 32));
    a.nstat = 0; a.stat_cstride = 192;
    a.dbg = dbg;
    std::vector<float> ref;
    auto run_tile = [&](ConvTile tile) {
        if (tile.NW == 96 && !conv_pw_eligible(a, tile.MT, tile.NT)) return;
        if (tile.NW != 96 && (conv_smem_bytes(a, tile) > 120 * 1024 || tile.NW * tile.KS > K / 16)) return;
        hipGraph_t gr; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int o = 0; o < nops; ++o) CK(launch_conv(a, tile, s));
        CK(hipStreamEndCapture(s, &gr));
        CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        std::vector<float> got((size_t)L * N);
        CK(hipMemcpy(got.data(), dy, got.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        if (ref.empty()) ref = got;
        else for (size_t e = 0; e < got.size(); ++e) worst = std::max(worst, (double)fabsf(got[e] - ref[e]));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        const int reps = 30;
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  tile %d,%d,%d,%d,%d  %7.2f us per launch   (max|diff| to the first tile %.2e)\n", tile.MT, tile.NT, tile.NW, tile.KS, tile.XM, ms * 1e3 / reps / nops, worst);
#ifdef MTV_DEEP_STAMP
        if (tile.NW == 96) {
            unsigned long long h[64];
            CK(hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost));
            static const char* nm[12] = {"entry", "decoded", "requests issued", "stats barrier", "stats2", "rows parked", "mfma done", "image", "-", "end", "transformed", "epilogue operands requested"};
            static const int order[] = {1, 2, 3, 4, 10, 11, 5, 6, 7, 9, -1};
            for (int blk = 0; blk < 2; ++blk)
                for (int role = 0; role < 2; ++role) {
                    printf("    stamps wg %s %s:", blk ? "mid" : "0", role ? "wave 4" : "wave 0");
                    for (int k = 0; order[k] >= 0; ++k) printf("  %s %lld", nm[order[k]], (long long)(h[blk * 32 + role * 16 + order[k]] - h[blk * 32 + role * 16]));
                    printf("\n");
                }
        }
#endif
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(gr));
    };
    const ConvTile tiles[] = {{2, 4, 4, 1, 0}, {1, 4, 4, 1, 0}, {1, 4, 8, 1, 0}, {1, 2, 4, 1, 0}, {1, 2, 8, 1, 0}, {1, 1, 96, 1, 0}, {2, 1, 96, 1, 0}, {1, 2, 96, 1, 0}, {2, 2, 96, 1, 0}};
    for (const ConvTile& tl : tiles) run_tile(tl);
}

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "check")) { CK(deep_init_attrs()); return do_check(); }
    if (argc >= 2 && !strcmp(argv[1], "chain")) {
        const int nops = argc >= 3 ? atoi(argv[2]) : 40;
        const int r = argc >= 5 ? atoi(argv[3]) : 4, t = argc >= 5 ? atoi(argv[4]) : 2;
        do_chain(nops, r, t, argc >= 6 ? atoi(argv[5]) : 0, argc >= 7 ? atoi(argv[6]) : 0);
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "win")) {
        do_win(argc >= 3 ? atoi(argv[2]) : 20, argc >= 5 ? atoi(argv[3]) : 32, argc >= 5 ? atoi(argv[4]) : 16, argc >= 6 ? atoi(argv[5]) : 128);
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "pw")) {
        do_pw(argc >= 4 ? atoi(argv[2]) : 32, argc >= 4 ? atoi(argv[3]) : 16, argc >= 5 ? atoi(argv[4]) : 128, argc >= 6 ? atoi(argv[5]) : 384);
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "attn")) { CK(deep_init_attrs()); return do_attn(argc >= 3); }
    if (argc >= 2 && !strcmp(argv[1], "block")) { CK(deep_block_init_attrs()); return do_block(argc >= 3); }
    printf("usage: deep_bench check | chain [nops] [r t] | attn [time]\n");
    return 1;
}
