// k_lin<MT, NT, NWV>: the lean kernel for the step's 1x1 convolutions on identity rows -- the attention blocks' qkv and
// proj_out conv1d (MToV/models/ddpm/unet.py:234,242,251,253 and :281,289,298,300), 80 of a denoising step's 181 launches.
//
// Why a second kernel.  k_conv treats a 1x1 conv as a one-tap convolution: chunk records, a row table for an identity
// gather, K = 128..512 split over 8 waves that each multiply ONE or two 16-channel chunks, then an 8-wave LDS
// reduction.  At K <= 512 all of that is fixed cost around ~1.3 us of multiplying (profiles/r02_conv_phase_stamps.txt:
// proj [2048x128 k128] 5.9 us in-kernel, qkv [2048x384 k128] 10.6 us).  Here
//   * a wave owns a (16 MT) x (16 NT) output tile and walks the WHOLE K itself: no cross-wave reduction, no split-K;
//     the NWV waves of a workgroup sit side by side along N and share nothing but the GroupNorm coefficient table;
//   * addresses are arithmetic (row = token, no gather, no tables, no records), so the first loads leave ~100
//     instructions after entry, and for K = 128 every operand of the tile is requested in that first batch;
//   * the weight is read in the checkpoint's own layout [N][K] (K contiguous): the B fragment of an MFMA is then the
//     same 16-byte load as the A fragment (lane (j, q) holds W[n][c0 + 4q .. + 3]), for any NT;
//   * the epilogue runs straight from the accumulators (lane (j, q) holds NT consecutive output channels of rows
//     4q .. 4q + 3: one NT-float store per row), residual rows were requested at entry;
//   * GroupNorm statistics of the output (proj feeds the next block's GroupNorm) are reduced by lane shuffles to one
//     fp64 atomic pair per (wave, plane, group).
// Arithmetic per output element is k_conv's, operation for operation: the same folded affine x * A_c + B_c, the same
// v_mfma_f32_16x16x4_f32 chain over the channels in ascending chunks -- but ONE chain per element instead of a sum of
// per-wave partial chains, so results differ from k_conv's at the 1e-7 level (tolerances of the parity tests unchanged).
#include "mtv_internal.h"

namespace mtv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const char gchar;

struct LinArgs {
    const float* x;        // [B][L][K]
    const float* W;        // [N][K]
    const float* bias;     // [N]
    const float* res;      // [B][L][N] or nullptr
    float* out;            // [B][L][N]
    int B, L, K, N;
    int tiles_m, tiles_n;  // row tiles per batch element, column tiles
    float inv_tiles_n, inv_tiles_m;
    int nch;               // K / 16
    int xcd_rows;          // > 0: workgroup id & 7 owns a contiguous range of `xcd_rows` row tiles (A rows fetched by one L2)
    int nstat;
    unsigned stat_cstride;
    SegInfo seg;
    GnIn gn;
    StatOut stat[2];
    unsigned long long* dbg;   // tools/ubench/lin_bench (-DLIN_STAMP): phase timestamps of three sampled workgroups; unused otherwise
};

#ifdef LIN_STAMP
#define LSTAMP(k) do { if (threadIdx.x == 0 && a.dbg) { const int sb_ = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : (blockIdx.x == gridDim.x - 1 ? 2 : -1)); \
                       if (sb_ >= 0) a.dbg[sb_ * 8 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define LSTAMP(k) do { } while (0)
#endif
#ifndef LIN_ABLATE
#define LIN_ABLATE 0       // lin_bench builds ablated variants: 1 no kernarg touch, 2 no statistics, 4 no stores, 8 no MFMA
#endif

__device__ __forceinline__ float silu_lin(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ int plane_of(const SegInfo& s, int tok) { return tok >= s.b2 ? 2 : (tok >= s.b1 ? 1 : 0); }

template <int MT, int NT, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_lin(const LinArgs a) {
    LSTAMP(0);
    if constexpr (!(LIN_ABLATE & 1)) touch_kernargs<(int)sizeof(LinArgs)>();
    LSTAMP(1);
    extern __shared__ __attribute__((aligned(16))) float lin_smem[];     // GroupNorm coefficients [3][K] float2
    __shared__ float2 s_mr[3][32];
    constexpr int NTH = 64 * NWV, ROWS = 16 * MT, WCOLS = 16 * NT, COLS = WCOLS * NWV;
    constexpr int DEPTH = (MT + NT) <= 3 ? 8 : ((MT + NT) <= 4 ? 6 : 4);     // chunks in flight per wave ((MT + NT) x 4 VGPRs each)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & 15, q = lane >> 4;
    // ---- block -> (batch element, row tile, column tile)
    int bx, by;
    {
        const int blk = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
        if (a.xcd_rows > 0) {
            const int xcd = blk & 7, j = blk >> 3;                       // j = local tile index inside this XCD's row range
            const int lr = FDiv{a.inv_tiles_n}(j, a.tiles_n);
            by = j - lr * a.tiles_n;
            bx = xcd * a.xcd_rows + lr;
        } else {
            bx = FDiv{a.inv_tiles_n}(blk, a.tiles_n);
            by = blk - bx * a.tiles_n;
        }
    }
    const int b = __builtin_amdgcn_readfirstlane(FDiv{a.inv_tiles_m}(bx, a.tiles_m));
    const int tok0 = (bx - b * a.tiles_m) * ROWS;
    const int nw0 = by * COLS + wave * WCOLS;            // this wave's first output channel
    const int K = a.K, L = a.L;
    const bool do_gn = a.gn.sums != nullptr;

    // ---- operand pointers (global address space, 16-byte fragments); rows / columns past the end are clamped, masked later
    gchar* ap[MT];
    gchar* wp[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int tok = min(tok0 + 16 * mt + i, L - 1);
        ap[mt] = (gchar*)(unsigned long long)(a.x + ((size_t)b * L + tok) * K + 4 * q);
    }
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
        const int col = min(nw0 + NT * i + nb, a.N - 1);
        wp[nb] = (gchar*)(unsigned long long)(a.W + (size_t)col * K + 4 * q);
    }
    struct Raw { f32x4 a[MT]; f32x4 b[NT]; };
    auto issue = [&](int ch, Raw& o) {
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) o.b[nb] = *(const __attribute__((address_space(1))) f32x4*)(wp[nb] + 64 * ch);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) o.a[mt] = *(const __attribute__((address_space(1))) f32x4*)(ap[mt] + 64 * ch);
    };
    Raw ring[DEPTH];
    const int nch = a.nch;
    int nx = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nch) { issue(nx, ring[d]); ++nx; }

    // ---- epilogue operands: bias and residual rows of this lane, requested now
    const int col0 = nw0 + NT * i;
    const bool colok = col0 + NT - 1 < a.N;
    float bias[NT];
    float resv[MT][4][NT];
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) bias[nb] = colok ? a.bias[col0 + nb] : 0.f;
    if (a.res) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = min(tok0 + 16 * mt + 4 * q + r, L - 1);
                const float* rp = a.res + ((size_t)b * L + tok) * a.N + (colok ? col0 : 0);
                if constexpr (NT == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(rp); resv[mt][r][0] = t[0]; resv[mt][r][1] = t[1]; resv[mt][r][2] = t[2]; resv[mt][r][3] = t[3]; }
                else if constexpr (NT == 2) { const f32x2 t = *reinterpret_cast<const f32x2*>(rp); resv[mt][r][0] = t[0]; resv[mt][r][1] = t[1]; }
                else resv[mt][r][0] = rp[0];
            }
    }

    LSTAMP(2);
    // ---- GroupNorm coefficients (only qkv has a GroupNorm in front): per (plane, channel) {A, B}, y = x * A + B
    float2* coef = reinterpret_cast<float2*>(lin_smem);
    if (do_gn) {
        const bool whole = a.gn.whole != 0;
        for (int e = tid; e < 96; e += NTH) {
            const int sg = e >> 5, g = e & 31;
            // (same association as k_conv: copies summed per plane, then (p0 + p1) + p2 for a cross-plane site)
            f64x2 vp[3] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                if (!whole && p != 0) continue;
                const int ent = whole ? p * 32 + g : e;
#pragma unroll
                for (int k = 0; k < STAT_COPIES; ++k)
                    vp[p] += *reinterpret_cast<const f64x2*>(a.gn.sums + (size_t)k * a.gn.cstride + (size_t)b * 192 + (size_t)ent * 2);
            }
            const f64x2 v = whole ? (vp[0] + vp[1]) + vp[2] : vp[0];
            const double inv_n = whole ? a.gn.inv_n[3] : a.gn.inv_n[sg];
            const double mean = v[0] * inv_n;
            double var = v[1] * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mr[sg][g] = make_float2((float)mean, 1.0f / sqrtf((float)var + 1e-5f));
        }
        __syncthreads();
        for (int c0 = tid * 4; c0 < K; c0 += NTH * 4) {
            const f32x4 ga = *reinterpret_cast<const f32x4*>(a.gn.gamma + c0), be = *reinterpret_cast<const f32x4*>(a.gn.beta + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int g = FDiv{a.gn.inv_gs}(c0 + k, a.gn.gs);
#pragma unroll
                for (int sg = 0; sg < 3; ++sg) {
                    const float2 mr = s_mr[sg][g];
                    const float sc = mr.y * ga[k];
                    coef[sg * K + c0 + k] = make_float2(sc, be[k] - sc * mr.x);
                }
            }
        }
        __syncthreads();
    }
    LSTAMP(3);
    const bool act = a.gn.act != 0;
    int pl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) pl[mt] = plane_of(a.seg, min(tok0 + 16 * mt + i, L - 1));

    // ---- K loop: the whole K of this wave's tile, DEPTH chunks in flight
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[mt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int ch, const Raw& in) {
        f32x4 av[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v = in.a[mt];
            if (do_gn) {
                const f32x4* cf = reinterpret_cast<const f32x4*>(coef + pl[mt] * K + 16 * ch + 4 * q);
                const f32x4 k0 = cf[0], k1 = cf[1];
                float y0 = fmaf(v[0], k0[0], k0[1]), y1 = fmaf(v[1], k0[2], k0[3]);
                float y2 = fmaf(v[2], k1[0], k1[1]), y3 = fmaf(v[3], k1[2], k1[3]);
                if (act) { y0 = silu_lin(y0); y1 = silu_lin(y1); y2 = silu_lin(y2); y3 = silu_lin(y3); }
                v = f32x4{y0, y1, y2, y3};
            }
            av[mt] = v;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) {
                    if constexpr (LIN_ABLATE & 8) acc[mt][nb][s] += av[mt][s] * in.b[nb][s];
                    else acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][s], in.b[nb][s], acc[mt][nb], 0, 0, 0);
                }
    };
    {
        int done = 0;
        while (nch - done >= 2 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                mma(done + d, ring[d]);
                issue(nx, ring[d]);
                ++nx;
            }
            done += DEPTH;
        }
        const int n = nch - done;            // < 2 * DEPTH chunks left, min(n, DEPTH) of them in the ring
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d < n) {
                mma(done + d, ring[d]);
                if (d + DEPTH < n) { issue(nx, ring[d]); ++nx; }
            }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d + DEPTH < n) mma(done + DEPTH + d, ring[d]);
    }

    LSTAMP(5);
    // ---- epilogue from the accumulators: lane (i, q) holds rows 4q + r (r = 0..3) x channels col0 .. col0 + NT - 1
    const int tile_p0 = plane_of(a.seg, tok0), tile_p1 = plane_of(a.seg, min(tok0 + ROWS - 1, L - 1));
    double ps[3][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};       // (sum, sum of squares) of this lane's elements per plane
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tok = tok0 + 16 * mt + 4 * q + r;
            float o[NT];
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                o[nb] = acc[mt][nb][r] + bias[nb];
                if (a.res) o[nb] += resv[mt][r][nb];
            }
            if (tok < L && colok && !(LIN_ABLATE & 4)) {
                float* op = a.out + ((size_t)b * L + tok) * a.N + col0;
                if constexpr (NT == 4) *reinterpret_cast<f32x4*>(op) = f32x4{o[0], o[1], o[2], o[3]};
                else if constexpr (NT == 2) *reinterpret_cast<f32x2*>(op) = f32x2{o[0], o[1]};
                else op[0] = o[0];
                if (a.nstat > 0) {
                    double s = 0.0, ss = 0.0;
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        s += (double)o[nb];
                        ss += (double)o[nb] * (double)o[nb];
                    }
                    const int p = tile_p0 == tile_p1 ? tile_p0 : plane_of(a.seg, tok);
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp)
                        if (p == pp) { ps[pp][0] += s; ps[pp][1] += ss; }
                }
            }
        }
    LSTAMP(6);
    if (a.nstat == 0 || (LIN_ABLATE & 2)) return;
    // ---- statistics for the GroupNorm sites that consume this tensor: rows live in the four q lane groups, a group's
    // channels in gs / NT neighbouring lanes i.  Shuffle-reduce both, then ONE fp64 atomic pair per (wave, plane, group).
    for (int t = 0; t < a.nstat; ++t) {
        const int gs = a.stat[t].gs;
        const int lpg = gs / NT;                                     // lanes i per group (a power of two, >= 1: checked by the host)
        const int g = FDiv{a.stat[t].inv_gs}(a.stat[t].coff + (colok ? col0 : 0), gs);
        for (int p = tile_p0; p <= tile_p1; ++p) {
            double s = p == 0 ? ps[0][0] : (p == 1 ? ps[1][0] : ps[2][0]);
            double ss = p == 0 ? ps[0][1] : (p == 1 ? ps[1][1] : ps[2][1]);
            s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
            s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
            // power-of-two groups are aligned blocks of lanes i: tree over them; otherwise (gs = 12, 24: a GroupNorm over a
            // 384- / 768-channel concatenation) every lane adds its own NT channels
            const bool tree = (lpg & (lpg - 1)) == 0;
            if (tree)
                for (int o = 1; o < lpg && o < 16; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
            const bool mine = tree ? (i & ((lpg < 16 ? lpg : 16) - 1)) == 0 : true;
            if (q == 0 && mine && colok && ss != 0.0) {
                double* dst = a.stat[t].sums + (size_t)(blockIdx.x & (STAT_COPIES - 1)) * a.stat_cstride + (((size_t)b * 3 + p) * 32 + g) * 2;
                atomicAdd(dst, s);
                atomicAdd(dst + 1, ss);
            }
        }
    }
    LSTAMP(7);
}

// ---------------------------------------------------------------------------------------------------------------
bool conv_lin_eligible(const ConvArgs& a) {
    if (!a.Wnk || a.ntaps != 1 || a.nmain != 1 || a.nskip != 0 || a.Cskip != 0 || a.gather || a.gather_skip || a.geo_main || a.geo_skip) return false;
    if (a.out_cm || a.ddim || a.bias_b || a.bias2 || a.gn.film) return false;
    if ((a.Cmain & 15) || (a.N & 3) || a.Lsrc != a.Lout || (a.res && a.Lskip != a.Lout)) return false;
    for (int t = 0; t < a.nstat; ++t) {
        const int gs = a.stat[t].gs;
        if ((gs & 3) || (a.stat[t].coff % gs)) return false;      // a lane's NT (<= 4) consecutive channels lie in one group
        if ((gs & (gs - 1)) == 0 && gs > 64) return false;
    }
    return true;
}

size_t lin_smem_bytes(const ConvArgs& a) { return a.gn.sums ? (size_t)24 * a.Cmain : 16; }

template <int MT, int NT, int NWV>
static hipError_t launch_lin_t(const ConvArgs& c, hipStream_t s) {
    LinArgs a{};
    a.x = c.src[0];
    a.W = c.Wnk;
    a.bias = c.bias;
    a.res = c.res;
    a.out = c.out;
    a.B = c.B; a.L = c.Lout; a.K = c.Cmain; a.N = c.N;
    constexpr int ROWS = 16 * MT, COLS = 16 * NT * NWV;
    a.tiles_m = (a.L + ROWS - 1) / ROWS;
    a.tiles_n = (a.N + COLS - 1) / COLS;
    a.inv_tiles_m = 1.0f / (float)a.tiles_m;
    a.inv_tiles_n = 1.0f / (float)a.tiles_n;
    a.nch = a.K / 16;
    const long rows = (long)a.B * a.tiles_m, nblk = rows * a.tiles_n;
    if (nblk >= (1L << 21)) return hipErrorInvalidValue;
    a.xcd_rows = rows % 8 == 0 ? (int)(rows / 8) : 0;
    a.nstat = c.nstat;
    a.stat_cstride = c.stat_cstride;
    a.seg = c.seg_out;
    a.gn = c.gn;
    a.stat[0] = c.stat[0];
    a.stat[1] = c.stat[1];
    a.dbg = c.dbg;
    hipLaunchKernelGGL((k_lin<MT, NT, NWV>), dim3((unsigned)nblk), dim3(64 * NWV), lin_smem_bytes(c), s, a);
    return hipGetLastError();
}

template <int MT, int NT>
static hipError_t launch_lin_w(const ConvArgs& c, int nwv, hipStream_t s) {
    switch (nwv) {
        case 1: return launch_lin_t<MT, NT, 1>(c, s);
        case 2: return launch_lin_t<MT, NT, 2>(c, s);
        case 4: return launch_lin_t<MT, NT, 4>(c, s);
    }
    return hipErrorInvalidValue;
}

// tiles are encoded as ConvTile{MT, NT, NW = 64, KS = waves along N, XM = 0}
hipError_t launch_lin(const ConvArgs& c, ConvTile t, hipStream_t s) {
    if (!conv_lin_eligible(c)) return hipErrorInvalidValue;
    if (t.MT == 1 && t.NT == 1) return launch_lin_w<1, 1>(c, t.KS, s);
    if (t.MT == 1 && t.NT == 2) return launch_lin_w<1, 2>(c, t.KS, s);
    if (t.MT == 1 && t.NT == 4) return launch_lin_w<1, 4>(c, t.KS, s);
    if (t.MT == 2 && t.NT == 1) return launch_lin_w<2, 1>(c, t.KS, s);
    if (t.MT == 2 && t.NT == 2) return launch_lin_w<2, 2>(c, t.KS, s);
    if (t.MT == 2 && t.NT == 4) return launch_lin_w<2, 4>(c, t.KS, s);
    return hipErrorInvalidValue;
}

}  // namespace mtv
