// k_conv_b3<MT, NT>: the LDS-staged convolution for LARGE token counts (several clips batched on one GPU, the 512 x 512
// geometry, the autoencoder's 16384-token GEMMs) on the bf16 matrix pipe at f32 accuracy: a three-term bf16 split of
// both operands, six partial products per pair, f32 accumulation.
//
// Why here and not in the one-clip step.  v_mfma_f32_16x16x4_f32 runs at the f32 vector rate and never co-executes with
// VALU work (profiles/r03_mfma_valu_counters.txt); v_mfma_f32_16x16x32_bf16 retires 8x the products per instruction in half
// the cycles: six of them replace sixteen f32 MFMAs (a 5x cut in matrix-pipe time) -- IF the split is cheap.  In the attention
// core it is not (the probabilities must be split per key block and query: attn_b3.hip, slower).  In a convolution at large
// M it is: a weight is split ONCE, at load time (ConvArgs::W3: three bf16 planes, 6 bytes per weight); an activation is
// split once per (tap, column tile) on its way into LDS by the thread that stages it -- 22 VALU operations per 8 channels,
// amortised over the BN = 128 output channels of the tile.  Round 2's two attempts (tools/experiments/r02_*.patch) failed
// on operand delivery (k_conv: every wave fetches its own W planes) and on occupancy (4-wave workgroups, 125-150 KB of LDS:
// one wave per SIMD).  This kernel follows their post-mortem: 8 waves per workgroup (2 x 4), one workgroup per CU.
//
// Tile: (32 MT) x (64 NT) outputs per workgroup, wave (wm, wn) of 2 x 4 owns (16 MT) x (16 NT); K walks in chunks of
// 32 channels (one MFMA step).  Per chunk all 512 threads stage
//   A: rows gathered by tap, GroupNorm / FiLM / SiLU applied, split -> three planes [BM][32 + 8 pad] bf16
//   W: the three pre-split planes of the chunk, [3][4 k-groups][BN] x 16 bytes, column positions permuted so that a fragment
//      read of 16 lanes hits 16 consecutive items while a lane's NT accumulators are NT CONSECUTIVE output channels
// double buffered (next chunk's global loads in flight under this chunk's MFMAs).  Per (row block, column block) the six
// products a2 w0, a1 w1, a0 w2, a1 w0, a0 w1, a0 w0 (small first) carry the f32 product to ~2^-24 relative.
// Same arguments and epilogue semantics as k_conv_lds (bias, residual, fused GroupNorm statistics), no split-K.
#include "mtv_internal.h"

namespace mtv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const char gchar;

__device__ __forceinline__ float b3_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ unsigned b3_pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ __forceinline__ float b3_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float b3_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ int b3_seg(const SegInfo& s, int tok) { return tok >= s.b2 ? 2 : (tok >= s.b1 ? 1 : 0); }
__device__ __forceinline__ int b3_div(int n, int d, float inv) { return FDiv{inv}(n, d); }

// 8 floats -> three bf16 planes of 8 (round to nearest even; the residuals are exact)
__device__ __forceinline__ void b3_split8(const float (&y)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float a = y[2 * h], b = y[2 * h + 1];
        const unsigned t0 = b3_pk(a, b);
        const float ra = a - b3_lo(t0), rb = b - b3_hi(t0);
        const unsigned t1 = b3_pk(ra, rb);
        p0[h] = t0;
        p1[h] = t1;
        p2[h] = b3_pk(ra - b3_lo(t1), rb - b3_hi(t1));
    }
}

#ifndef B3_ABLATE
#define B3_ABLATE 0          // tools/ubench/b3_bench builds ablated variants: 1 no MFMA, 2 no transform / split, 4 no LDS stores, 8 no global loads, 16 no fragment reads
#endif

struct B3Rec {                // one per 32-channel chunk, in LDS (same fields as conv.hip's ChunkRec)
    unsigned a_lo, a_hi;      // source part base + 4 * (first channel of the chunk)
    unsigned wrow;            // W row of the chunk's first channel
    unsigned meta;            // coefficient channel (16 bits) | channels of the part / 16 (8 bits) << 16 | tap (4 bits) << 24 | skip << 28
};

template <int MT, int NT>
struct B3Shape {
    static constexpr int BM = 32 * MT, BN = 64 * NT, NTH = 512;
    static constexpr int ASTR = 80;                                  // bytes per staged A row and plane: 32 bf16 + 8 pad
    static constexpr int A_STAGE = 3 * BM * ASTR, W_STAGE = 12 * BN * 16;
};

// LDS (bytes): chunk records | row table [(ntaps+1)][BM] ints | coefficients [3][Cmain] float2 | A stages | W stages | statistics slots
// 160 KB per workgroup minus the kernel's static LDS (s_mr 768 B + s_dp 1536 B), rounded down
static constexpr size_t B3_MAX_DYN_LDS = CONV_B3_MAX_LDS;
static size_t b3_lds_bytes(int MT, int NT, int ntaps, int Cmain, int Cskip, bool has_gn, int* a_off = nullptr, int* qs_off = nullptr) {
    const int BM = 32 * MT, BN = 64 * NT;
    const int nch = ntaps * (Cmain / 32) + Cskip / 32;
    size_t o = (size_t)((nch + 3) & ~3) * 16 + (size_t)(((ntaps + 1) * BM + 3) & ~3) * 4 + (has_gn ? (size_t)24 * Cmain : 0);
    o = (o + 15) & ~(size_t)15;
    if (a_off) *a_off = (int)o;
    o += 2 * ((size_t)3 * BM * 80 + (size_t)12 * BN * 16);
    if (qs_off) *qs_off = (int)o;
    return o + (size_t)3 * (BN / 4) * 2 * 8;
}

template <int MT, int NT>
__global__ __launch_bounds__(512) void k_conv_b3(const ConvArgs a) {
    touch_kernargs<(int)sizeof(ConvArgs)>();
    typedef B3Shape<MT, NT> SH;
    constexpr int BM = SH::BM, BN = SH::BN, NTH = SH::NTH, ASTR = SH::ASTR;
    constexpr int APT = (BM * 4 + NTH - 1) / NTH;          // A items per thread per chunk: (row, group of 8 channels)
    constexpr int WPT = (12 * BN + NTH - 1) / NTH;         // W items per thread per chunk: (plane, k-group, column) x 16 bytes
    constexpr int D = NT >= 4 ? 2 : 4;                        // chunks in flight in registers
    extern __shared__ __attribute__((aligned(16))) char b3_smem[];
    __shared__ float2 s_mr[3][32];
    __shared__ f64x2 s_dp[96];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & 15, q = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int tiles_per_b = a.tiles_per_b, tiles_n = a.tiles_n;
    const int blk = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    const int bx = __builtin_amdgcn_readfirstlane(b3_div(blk, tiles_n, a.inv_tiles_n));      // column tiles fastest
    const int by = blk - bx * tiles_n;
    const int b = __builtin_amdgcn_readfirstlane(b3_div(bx, tiles_per_b, a.inv_tiles_per_b));
    const int tok0 = (bx - b * tiles_per_b) * BM;
    const int n0 = by * BN;
    const int Cmain = a.Cmain;
    const bool do_gn = a.gn.sums != nullptr;

    // GroupNorm inputs first (needed last), as in k_conv / k_conv_lds
    f32x4 ga, be, s1, sh;
    const float* film = (do_gn && a.gn.film) ? a.gn.film + (size_t)b * a.gn.film_stride : nullptr;
    auto fetch = [&](int c0) {
        ga = *reinterpret_cast<const f32x4*>(a.gn.gamma + c0);
        be = *reinterpret_cast<const f32x4*>(a.gn.beta + c0);
        s1 = f32x4{1.f, 1.f, 1.f, 1.f};
        sh = f32x4{0.f, 0.f, 0.f, 0.f};
        if (film) {
            s1 += *reinterpret_cast<const f32x4*>(film + c0);
            sh = *reinterpret_cast<const f32x4*>(film + Cmain + c0);
        }
    };
    f64x2 v0 = {0.0, 0.0};
    if (do_gn) {
        if (tid * 4 < Cmain) fetch(tid * 4);
        if (tid < 96) {
#pragma unroll
            for (int k = 0; k < STAT_COPIES; ++k)
                v0 += *reinterpret_cast<const f64x2*>(a.gn.sums + (size_t)k * a.gn.cstride + (size_t)b * 192 + (size_t)tid * 2);
        }
    }

    const int nchunks = a.cps_q;                                       // chunks of 32 channels (set by the launcher)
    B3Rec* recs = reinterpret_cast<B3Rec*>(b3_smem);
    int* idx = reinterpret_cast<int*>(b3_smem + (size_t)((nchunks + 3) & ~3) * 16);
    float2* coef = reinterpret_cast<float2*>(reinterpret_cast<char*>(idx) + (size_t)(((a.ntaps + 1) * BM + 3) & ~3) * 4);
    char* const As = b3_smem + a.rec_cap;                              // (rec_cap / qs_off: byte offsets here, from launch_conv_b3)
    char* const Ws = As + 2 * SH::A_STAGE;
    double* qs = reinterpret_cast<double*>(b3_smem + a.qs_off);
    {
        const int cpt = a.cpt;                                         // 32-channel chunks per tap
        const int nmainch = a.ntaps * cpt;
        for (int e = tid; e < nchunks; e += NTH) {
            const bool skip = e >= nmainch;
            const int tap = skip ? a.ntaps : b3_div(e, cpt, a.inv_cpt);
            const int w = skip ? e - nmainch : e - tap * cpt;
            const int c0_32 = (skip ? a.C[2] : a.C[0]) >> 5;
            const bool second = w >= c0_32;
            const int c = (second ? w - c0_32 : w) << 5;
            const float* sp = skip ? (second ? a.src[3] : a.src[2]) : (second ? a.src[1] : a.src[0]);
            const int Cp = skip ? (second ? a.C[3] : a.C[2]) : (second ? a.C[1] : a.C[0]);
            const int coff = second ? (skip ? a.C[2] : a.C[0]) : 0;
            const unsigned long long ab = reinterpret_cast<unsigned long long>(sp) + (unsigned long long)c * 4ull;
            B3Rec r;
            r.a_lo = (unsigned)(ab & 0xFFFFFFFFull);
            r.a_hi = (unsigned)(ab >> 32);
            r.wrow = (unsigned)((skip ? a.ntaps * Cmain : tap * Cmain) + coff + c);
            r.meta = (unsigned)(coff + c) | ((unsigned)(Cp >> 4) << 16) | ((unsigned)tap << 24) | ((unsigned)(skip ? 1 : 0) << 28);
            recs[e] = r;
        }
    }
    for (int e = NTH - 1 - tid; e < (a.ntaps + 1) * BM; e += NTH) {
        const int t = e / BM, r = e - t * BM;
        const int tok = tok0 + r;
        int v = -1;
        if (tok < a.Lout) {
            if (t < a.ntaps) {
                if (a.geo_main) {
                    const int ky = t >= 6 ? 2 : (t >= 3 ? 1 : 0);
                    v = geo_source_t<FDiv>(FDiv{a.geo_inv_r}, a.geo_r, a.geo_t, tok, ky, t - 3 * ky, a.geo_main == 2);
                } else {
                    const int st = a.gather ? a.gather[t * a.Lout + tok] : tok;
                    v = st < 0 ? -1 : (st | (b3_seg(a.seg_src, st) << 28));
                }
            } else if (a.geo_skip) {
                v = geo_source_t<FDiv>(FDiv{a.geo_inv_r}, a.geo_r, a.geo_t, tok, 1, 1, true) & 0x0FFFFFFF;
            } else {
                v = a.gather_skip ? a.gather_skip[tok] : tok;
            }
        }
        idx[e] = v;
    }
    for (int e = tid; e < 3 * (BN / 4) * 2; e += NTH) qs[e] = 0.0;
    if (do_gn) {
        const bool whole = a.gn.whole != 0;
        if (whole && tid < 96) s_dp[tid] = v0;
        __syncthreads();
        if (tid < 96) {
            const int sg = tid >> 5, g = tid & 31;
            f64x2 v;
            double inv_n;
            if (whole) {
                v = (s_dp[g] + s_dp[32 + g]) + s_dp[64 + g];
                inv_n = a.gn.inv_n[3];
            } else {
                v = v0;
                inv_n = a.gn.inv_n[sg];
            }
            const double mean = v[0] * inv_n;
            double var = v[1] * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mr[sg][g] = make_float2((float)mean, 1.0f / sqrtf((float)var + 1e-5f));
        }
        __syncthreads();
        for (int c0 = tid * 4; c0 < Cmain; c0 += NTH * 4) {
            if (c0 != tid * 4) fetch(c0);
            int grp[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) grp[k] = b3_div(c0 + k, a.gn.gs, a.gn.inv_gs);
#pragma unroll
            for (int sg = 0; sg < 3; ++sg)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float2 mr = s_mr[sg][grp[k]];
                    const float sc = mr.y * ga[k];
                    const float bi = be[k] - sc * mr.x;
                    coef[sg * Cmain + c0 + k] = make_float2(sc * s1[k], fmaf(bi, s1[k], sh[k]));
                }
        }
    }
    __syncthreads();

    // ---- staging: global -> registers (raw) -> [transform, split] -> LDS
    const bool act = a.gn.act != 0;
    const unsigned bL[2] = {(unsigned)b * (unsigned)a.Lsrc, (unsigned)b * (unsigned)a.Lskip};
    gchar* const W3g = (gchar*)(unsigned long long)a.W3;
    // register queue of D chunks in flight: one workgroup per CU has nothing else to hide the global-load latency behind (a chunk's
    // MFMA work is ~0.2 us, a load from L2 / HBM 0.5 - 2 us), so chunk ch + D is requested while chunk ch computes
    f32x4 ra[D][APT][2];
    int re[D][APT];
    u32x4 rw[D][WPT];
    int rcc[D], rskip[D];
    auto gload = [&](const int d, int ch) {
        const B3Rec* rp = recs + ch;
        const unsigned a_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)rp->a_lo), a_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)rp->a_hi);
        const unsigned wrow = (unsigned)__builtin_amdgcn_readfirstlane((int)rp->wrow), meta = (unsigned)__builtin_amdgcn_readfirstlane((int)rp->meta);
        const int tap = (int)((meta >> 24) & 15u);
        rskip[d] = (int)(meta >> 28);
        rcc[d] = (int)(meta & 0xFFFFu);
        const unsigned Cp4 = ((meta >> 16) & 0xFFu) << 6;
        gchar* abase = (gchar*)(((unsigned long long)a_hi << 32) | a_lo);
        const unsigned bl = rskip[d] ? bL[1] : bL[0];
#pragma unroll
        for (int k = 0; k < APT; ++k) {
            const int it = tid + NTH * k;                    // item = (row, k-group of 8 channels)
            const int row = it >> 2, kg = it & 3;
            const int e = it < BM * 4 ? idx[tap * BM + row] : -1;
            re[d][k] = e;
            const unsigned st = e < 0 ? 0u : (unsigned)(e & 0x0FFFFFFF);
            gchar* p = abase + ((bl + st) * Cp4 + 32u * kg);
            if constexpr (B3_ABLATE & 8) {
                ra[d][k][0] = f32x4{1.f, 2.f, 3.f, (float)(unsigned long long)p};
                ra[d][k][1] = ra[d][k][0];
            } else {
                ra[d][k][0] = *(const __attribute__((address_space(1))) f32x4*)p;
                ra[d][k][1] = *(const __attribute__((address_space(1))) f32x4*)(p + 16);
            }
        }
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int it = tid + NTH * k;                    // item = (plane, k-group, column): 16 bytes = 8 consecutive K rows of one column
            const int seg = it / BN, col = it - seg * BN;
            const int pl = seg >> 2, kg = seg & 3;
            rw[d][k] = (it < 12 * BN && n0 + col < a.ldw && !(B3_ABLATE & 8))       // (a column tile may reach past the padded row of W: never stored)
                        ? *(const __attribute__((address_space(1))) u32x4*)(W3g + ((size_t)pl * a.w3_plane + ((size_t)((wrow >> 3) + kg) * (size_t)a.ldw + (size_t)(n0 + col)) * 16u))
                        : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto lstore = [&](const int d, int buf) {
        char* Ab = As + buf * SH::A_STAGE;
        char* Wb = Ws + buf * SH::W_STAGE;
#pragma unroll
        for (int k = 0; k < APT; ++k) {
            const int it = tid + NTH * k;
            if (it >= BM * 4) continue;
            const int row = it >> 2, kg = it & 3;
            float y[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) { y[u] = ra[d][k][0][u]; y[4 + u] = ra[d][k][1][u]; }
            const int e = re[d][k];
            if (do_gn && !rskip[d] && !(B3_ABLATE & 2)) {
                const int sg = e < 0 ? 0 : ((e >> 28) & 3);
                const f32x4* cf = reinterpret_cast<const f32x4*>(coef + sg * Cmain + rcc[d] + 8 * kg);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 kk = cf[u];                  // {A, B} of channels 2u, 2u + 1
                    y[2 * u] = fmaf(y[2 * u], kk[0], kk[1]);
                    y[2 * u + 1] = fmaf(y[2 * u + 1], kk[2], kk[3]);
                }
                if (act) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) y[u] = b3_silu(y[u]);
                }
            }
            if (e < 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) y[u] = 0.f;      // conv zero padding applies AFTER norm / activation
            }
            u32x4 p0, p1, p2;
            if constexpr (B3_ABLATE & 2) {
                p0 = u32x4{b3_pk(y[0], y[1]), b3_pk(y[2], y[3]), b3_pk(y[4], y[5]), b3_pk(y[6], y[7])};
                p1 = p0;
                p2 = p0;
            } else {
                b3_split8(y, p0, p1, p2);
            }
            char* dst = Ab + row * ASTR + kg * 16;
            if constexpr (B3_ABLATE & 4) {
                asm volatile("" ::"v"(p0), "v"(p1), "v"(p2));
                continue;
            }
            *reinterpret_cast<u32x4*>(dst) = p0;
            *reinterpret_cast<u32x4*>(dst + BM * ASTR) = p1;
            *reinterpret_cast<u32x4*>(dst + 2 * BM * ASTR) = p2;
        }
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int it = tid + NTH * k;
            if (it >= 12 * BN) continue;
            const int seg = it / BN, col = it - seg * BN;
            // column position inside LDS: wave group (16 NT columns) | nb | lane j, for column = group * 16 NT + NT * j + nb
            const int grp = col / (16 * NT), cw = col - grp * (16 * NT);
            const int j = cw / NT, nb = cw - j * NT;
            if constexpr (B3_ABLATE & 4) {
                asm volatile("" ::"v"(rw[d][k]));
                continue;
            }
            *reinterpret_cast<u32x4*>(Wb + ((size_t)seg * BN + grp * 16 * NT + nb * 16 + j) * 16) = rw[d][k];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[mt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const char* Ab = As + buf * SH::A_STAGE + (wm * 16 * MT + i) * ASTR + q * 16;
        const char* Wb = Ws + buf * SH::W_STAGE + ((size_t)q * BN + wn * 16 * NT + i) * 16;
        bf16x8 af[3][MT], wf[3][NT];
        if constexpr (B3_ABLATE & 16) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) af[p][mt] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)p, (unsigned)mt, (unsigned)buf, 0x3f803f80u});
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) wf[p][nb] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)p, (unsigned)nb, (unsigned)buf, 0x3f803f80u});
            }
        } else
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[p][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ab + p * BM * ASTR + 16 * mt * ASTR));
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) wf[p][nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wb + ((size_t)p * 4 * BN + nb * 16) * 16));
        }
        // six partial products per (row block, column block), small terms first; independent accumulators back to back
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PW[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < NT; ++nb)
                    if constexpr (B3_ABLATE & 1) asm volatile("" ::"v"(af[PA[t]][mt]), "v"(wf[PW[t]][nb]));
                    else acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PA[t]][mt], wf[PW[t]][nb], acc[mt][nb], 0, 0, 0);
    };

#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nchunks) gload(d, d);
    lstore(0, 0);
    if (D < nchunks) gload(0, D);
    __syncthreads();
    for (int ch0 = 0; ch0 < nchunks; ch0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int ch = ch0 + d;
            if (ch >= nchunks) break;
            // chunk ch + 1 (requested D chunks ago) -> the other LDS stage, its slot re-armed with chunk ch + 1 + D; then this chunk's
            // MFMAs with those loads in flight.  (Stage (ch + 1) & 1 was last read before the previous barrier.)
            if (ch + 1 < nchunks) {
                lstore((d + 1) % D, (ch + 1) & 1);
                if (ch + 1 + D < nchunks) gload((d + 1) % D, ch + 1 + D);
            }
            compute(ch & 1);
            __syncthreads();
        }
    }

    // ---- epilogue from the accumulators: lane (i, q) of wave (wm, wn) holds, for row 4q + r of row block mt, the NT
    // consecutive output channels n0 + 16 NT wn + NT i ..
    const bool fast = a.nstat > 0;
    constexpr int QPR = BN / 4;
    const int colw = n0 + wn * 16 * NT + NT * i;
    if (colw < a.N) {
        float bias[NT];
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) {
            bias[nb] = a.bias[colw + nb];
            if (a.bias2) bias[nb] += a.bias2[colw + nb];
            if (a.bias_b) bias[nb] += a.bias_b[(size_t)b * a.bias_b_stride + colw + nb];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 16 * MT + 16 * mt + 4 * q + r;
                const int tok = tok0 + row;
                if (tok >= a.Lout) continue;
                float o[NT];
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) o[nb] = acc[mt][nb][r] + bias[nb];
                if (a.res) {
                    const int rs = idx[a.ntaps * BM + row];
                    const float* rp = a.res + ((size_t)b * a.Lskip + rs) * a.N + colw;
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) o[nb] += rp[nb];
                }
                float* op = a.out + ((size_t)b * a.Lout + tok) * a.N + colw;
                if constexpr (NT == 4) *reinterpret_cast<f32x4*>(op) = f32x4{o[0], o[1], o[2], o[3]};
                else if constexpr (NT == 2) *reinterpret_cast<f32x2*>(op) = f32x2{o[0], o[1]};
                else op[0] = o[0];
                if (fast) {
                    const int sgq = b3_seg(a.seg_out, tok);
                    double sq = 0.0, ssq = 0.0;
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) {
                        sq += (double)o[nb];
                        ssq += (double)o[nb] * o[nb];
                    }
                    const int cq = (colw - n0) >> 2;               // (NT < 4: several lanes share a quad slot)
                    atomicAdd(&qs[(sgq * QPR + cq) * 2], sq);
                    atomicAdd(&qs[(sgq * QPR + cq) * 2 + 1], ssq);
                }
            }
    }
    if (!fast) return;
    __syncthreads();
    for (int e = tid; e < a.nstat * 3 * QPR; e += NTH) {
        const int t = e / (3 * QPR), r2 = e - t * 3 * QPR;
        const int sgi = r2 / QPR, cq = r2 - sgi * QPR;
        const int n = n0 + cq * 4;
        if (n >= a.N) continue;
        const int gs = a.stat[t].gs, coff = a.stat[t].coff;
        const float inv_gs = a.stat[t].inv_gs;
        const int g = b3_div(coff + n, gs, inv_gs);
        if (cq > 0 && b3_div(coff + n - 4, gs, inv_gs) == g) continue;
        const int qend = min(QPR, min((a.N - n0 + 3) >> 2, ((g + 1) * gs - coff - n0 + 3) >> 2));
        double s = 0.0, ss = 0.0;
        for (int c2 = cq; c2 < qend; ++c2) {
            s += qs[(sgi * QPR + c2) * 2];
            ss += qs[(sgi * QPR + c2) * 2 + 1];
        }
        if (ss != 0.0) {
            double* dst = a.stat[t].sums + (size_t)(blockIdx.x & (STAT_COPIES - 1)) * a.stat_cstride + (((size_t)b * 3 + sgi) * 32 + g) * 2;
            atomicAdd(dst, s);
            atomicAdd(dst + 1, ss);
        }
    }
}

// W [K][ldw] f32 rows [row0, row0 + rows) -> the three bf16 planes W3 [3][K/8][ldw][8] (W = W0 + W1 + W2, round to nearest even)
__global__ __launch_bounds__(256) void k_split_w3(const float* __restrict__ W, char* __restrict__ W3, size_t plane_bytes, int row0, int rows8, int ldw) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)rows8 * ldw) return;
    const int g8 = (int)(e / ldw), n = (int)(e - (long)g8 * ldw);
    float y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) y[u] = W[((size_t)row0 + (size_t)g8 * 8 + u) * ldw + n];
    u32x4 p0, p1, p2;
    b3_split8(y, p0, p1, p2);
    char* dst = W3 + (((size_t)(row0 >> 3) + g8) * ldw + n) * 16;
    *reinterpret_cast<u32x4*>(dst) = p0;
    *reinterpret_cast<u32x4*>(dst + plane_bytes) = p1;
    *reinterpret_cast<u32x4*>(dst + 2 * plane_bytes) = p2;
}

hipError_t launch_split_w3(const float* W, void* W3, size_t plane_bytes, int row0, int rows, int ldw, hipStream_t s) {
    if ((row0 & 7) || (rows & 7)) return hipErrorInvalidValue;
    const long n = (long)(rows / 8) * ldw;
    hipLaunchKernelGGL(k_split_w3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, reinterpret_cast<char*>(W3), plane_bytes, row0, rows / 8, ldw);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
bool conv_b3_eligible(const ConvArgs& a) {
    if (!a.W3 || a.out_cm || a.ddim || (a.N & 3) || a.N < 64) return false;
    if ((a.Cmain & 31) || (a.Cskip & 31)) return false;
    for (int k = 0; k < 4; ++k)
        if (a.C[k] & 31) return false;
    for (int t = 0; t < a.nstat; ++t)
        if (a.stat[t].gs & 3) return false;
    return true;
}

size_t conv_b3_smem_bytes(const ConvArgs& a, ConvTile t) { return b3_lds_bytes(t.MT, t.NT, a.ntaps, a.Cmain, a.Cskip, a.gn.sums != nullptr); }

template <int MT, int NT>
static hipError_t launch_b3_t(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    constexpr int BM = 32 * MT, BN = 64 * NT;
    const int tiles = (a.Lout + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const long nblk = (long)a.B * tiles * tiles_n;
    if (nblk >= (1L << 21) || !conv_b3_eligible(a)) return hipErrorInvalidValue;
    a.KS = 1;
    a.xmap = 0;
    a.tiles_per_b = tiles;
    a.tiles_n = tiles_n;
    a.inv_tiles_per_b = 1.0f / (float)tiles;
    a.inv_tiles_n = 1.0f / (float)tiles_n;
    a.cpt = a.Cmain / 32;
    a.inv_cpt = 1.0f / (float)a.cpt;
    a.geo_inv_r = a.geo_r > 0 ? 1.0f / (float)a.geo_r : 0.f;
    a.cps_q = a.ntaps * (a.Cmain / 32) + a.Cskip / 32;
    a.cps_r = 0;
    int a_off = 0, qs_off = 0;
    const size_t smem = b3_lds_bytes(MT, NT, a.ntaps, a.Cmain, a.Cskip, a.gn.sums != nullptr, &a_off, &qs_off);
    if (smem > B3_MAX_DYN_LDS) return hipErrorInvalidValue;
    a.rec_cap = a_off;            // (byte offsets in this kernel)
    a.qs_off = qs_off;
    hipLaunchKernelGGL((k_conv_b3<MT, NT>), dim3((unsigned)nblk), dim3(512), smem, s, a);
    return hipGetLastError();
}

hipError_t conv_b3_init_attrs() {
    const void* fns[] = {reinterpret_cast<const void*>(&k_conv_b3<4, 2>), reinterpret_cast<const void*>(&k_conv_b3<2, 2>),
                         reinterpret_cast<const void*>(&k_conv_b3<4, 1>), reinterpret_cast<const void*>(&k_conv_b3<2, 1>),
                         reinterpret_cast<const void*>(&k_conv_b3<2, 4>), reinterpret_cast<const void*>(&k_conv_b3<4, 4>)};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, B3_MAX_DYN_LDS);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// tiles are encoded as ConvTile{MT, NT, NW = 48, KS = 1, XM = 0}
hipError_t launch_conv_b3(const ConvArgs& a, ConvTile t, hipStream_t s) {
    if (t.MT == 4 && t.NT == 2) return launch_b3_t<4, 2>(a, s);
    if (t.MT == 2 && t.NT == 2) return launch_b3_t<2, 2>(a, s);
    if (t.MT == 4 && t.NT == 1) return launch_b3_t<4, 1>(a, s);
    if (t.MT == 2 && t.NT == 1) return launch_b3_t<2, 1>(a, s);
    if (t.MT == 2 && t.NT == 4) return launch_b3_t<2, 4>(a, s);
    if (t.MT == 4 && t.NT == 4) return launch_b3_t<4, 4>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace mtv
