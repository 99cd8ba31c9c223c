// k_conv_x3<MT, NT>: the convolution for LARGE token counts (several clips batched on one GPU, the 512 x 512 geometry, the
// autoencoder's 16384-token GEMMs) on the bf16 matrix pipe at f32 accuracy, as TWO launches:
//
//   k_x3_prep    one elementwise pass over the conv's input: GroupNorm / FiLM / SiLU applied ONCE per element (not once per
//                tap and column tile), the result split into three bf16 terms x = x0 + x1 + x2 (round to nearest; the
//                residuals are exact) and stored k-group-major -- planes [3][C/8][rows][8 bf16], the same shape the weights
//                are kept in (ConvArgs::W3) -- with one all-zero row for the conv's zero padding.
//   k_conv_x3    a gathering GEMM over those planes: BOTH operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4:
//                no staging registers, no ds_write pass, no VALU work in the K loop), a ring of NS stages, one barrier per
//                32-channel step, counted vmcnt waits; six v_mfma_f32_16x16x32_bf16 per (row block, column block) carry
//                the f32 product to ~2^-24 relative: a2 w0, a1 w1, a0 w2, a1 w0, a0 w1, a0 w0 (small terms first).
//
// Why two launches.  The first version of this kernel (k_conv_b3, commit 1c33703, removed) fused the prologue and the split into
// the register staging of every (tap, column tile): its phases serialised (profiles/r03_conv_b3_ablation.txt: LDS stores ~310 ns,
// split ~180 ns, loads ~240 ns, MFMA ~335 ns of a 1485 ns chunk -- bf16 MFMAs barely co-execute with VALU,
// profiles/r03_mfma_valu_counters.txt mode 12 -- and ds_write_b128 moves 79 B/clk/CU).  Here the K loop is MFMA + ds_read_b128 +
// DMA issue only, and the two halves of the workgroup alternate between them (see the main loop).
//
// Tile: (32 MT) x (64 NT) outputs per workgroup of 8 waves (2 x 4), wave (wm, wn) owns (16 MT) x (16 NT).  A stage holds
// A [3 planes][4 k-groups][BM rows] and W [3][4][BN columns] in 16-byte items (8 consecutive channels of one row / column):
// a fragment read of 16 lanes hits 16 consecutive items, the four k-groups sit a multiple of 256 bytes apart (conflict
// free in ds_read_b128's lane groups).  A DMA piece is one wave instruction = 64 consecutive items of one (plane,
// k-group); LDS destinations are lane-linear, so the column order inside LDS (a lane's NT accumulators are NT CONSECUTIVE
// output channels) and the tap gather are applied on the SOURCE address.  Same epilogue as k_conv_lds.
#include "mtv_internal.h"

namespace mtv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const char gchar;
typedef __attribute__((address_space(3))) char lchar;

#ifndef X3_NSCAP
#define X3_NSCAP 4
#endif
#ifndef X3_PRIO
#define X3_PRIO 1
#endif
#ifndef X3_ABLATE
#define X3_ABLATE 0          // tools/ubench/x3_bench: 1 no MFMA, 2 no DMA, 4 no fragment reads, 8 no A pieces, 16 no W pieces
#endif

__device__ __forceinline__ float x3_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }      // (= ae.hip's gelu_erf)
__device__ __forceinline__ float x3_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ unsigned x3_pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ __forceinline__ float x3_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float x3_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ int x3_seg(const SegInfo& s, int tok) { return tok >= s.b2 ? 2 : (tok >= s.b1 ? 1 : 0); }
__device__ __forceinline__ int x3_div(int n, int d, float inv) { return FDiv{inv}(n, d); }

// 8 floats -> three bf16 planes of 8 (round to nearest even; the residuals are exact)
__device__ __forceinline__ void x3_split8(const float (&y)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float a = y[2 * h], b = y[2 * h + 1];
        const unsigned t0 = x3_pk(a, b);
        const float ra = a - x3_lo(t0), rb = b - x3_hi(t0);
        const unsigned t1 = x3_pk(ra, rb);
        p0[h] = t0;
        p1[h] = t1;
        p2[h] = x3_pk(ra - x3_lo(t1), rb - x3_hi(t1));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the elementwise pass
struct X3Prep {
    const float* src[4];      // as ConvArgs: [0..1] tapped parts, [2..3] parts of the fused 1x1 skip conv
    int C[4];
    int Cmain, Cskip, Lsrc, Lskip, B;
    GnIn gn;
    SegInfo seg_src;
    char* A3;                 // [3][Cmain / 8][rowsM][16 bytes]; row rowsM - 1 = zeros
    char* S3;                 // [3][Cskip / 8][rowsS][16 bytes]
    unsigned rowsM, rowsS;
    int tilesM, tilesS;       // 64-row blocks per clip
    int cbM, cbS;             // 32-channel blocks
    int geglu;                // 1: src[0] is a GEGLU pre-activation [rows][2 Cmain]; the element is h[c] * gelu(h[Cmain + c]) (vit_modules.py:88-91)
};

constexpr int X3P_TS = 66;    // items per (plane, k-group) row of the transpose tile (64 + 2: the four k-groups of a row land in different banks)

// one workgroup = 64 rows x 32 channels of one clip: load (coalesced along channels), transform, split, transpose through LDS,
// store (coalesced along rows)
__global__ __launch_bounds__(256) void k_x3_prep(const X3Prep a) {
    __shared__ u32x4 T[12 * X3P_TS];
    __shared__ float2 s_mr[3][32];
    __shared__ f64x2 s_dp[96];
    __shared__ float2 coef[3][32];
    const int tid = threadIdx.x;
    int blk = (int)blockIdx.x;
    const int nmain = a.B * a.tilesM * a.cbM;
    const bool skip = blk >= nmain;
    if (skip) blk -= nmain;
    const int tiles = skip ? a.tilesS : a.tilesM, ncb = skip ? a.cbS : a.cbM;
    const int cb = blk % ncb, bt = blk / ncb;
    const int b = bt / tiles, row0 = (bt - b * tiles) * 64;
    const int L = skip ? a.Lskip : a.Lsrc, C = skip ? a.Cskip : a.Cmain;
    const float* s0 = skip ? a.src[2] : a.src[0];
    const float* s1 = skip ? a.src[3] : a.src[1];
    const int C0 = skip ? a.C[2] : a.C[0], C1 = skip ? a.C[3] : a.C[1];
    char* const dst = skip ? a.S3 : a.A3;
    const unsigned rows = skip ? a.rowsS : a.rowsM;
    const int KG = C >> 3, Cmain = a.Cmain;
    const bool do_gn = !skip && a.gn.sums != nullptr;

    // issue the operand loads first
    const int row = tid >> 2, kg = tid & 3;
    const int tok = row0 + row;
    const int c = 32 * cb + 8 * kg;
    f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f};
    if (tok < L) {
        if (a.geglu) {        // the GEGLU of the autoencoder's feed-forward, fused: the same expression, in the same order, as ae.hip's k_geglu
            const float* hp = s0 + ((size_t)b * L + tok) * (2 * (size_t)C) + c;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(hp), a1 = *reinterpret_cast<const f32x4*>(hp + 4);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(hp + C), g1 = *reinterpret_cast<const f32x4*>(hp + C + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u0[e] = a0[e] * x3_gelu_erf(g0[e]);
                u1[e] = a1[e] * x3_gelu_erf(g1[e]);
            }
        } else {
            const bool second = c >= C0;
            const float* sp = second ? s1 : s0;
            const int Cp = second ? C1 : C0, cc = second ? c - C0 : c;
            u0 = *reinterpret_cast<const f32x4*>(sp + ((size_t)b * L + tok) * Cp + cc);
            u1 = *reinterpret_cast<const f32x4*>(sp + ((size_t)b * L + tok) * Cp + cc + 4);
        }
    }
    if (do_gn) {      // statistics -> affine coefficients of this block's 32 channels, as the fused prologues do (conv.hip)
        const float* film = a.gn.film ? a.gn.film + (size_t)b * a.gn.film_stride : nullptr;
        f64x2 v0 = {0.0, 0.0};
        float ga = 0.f, be = 0.f, sc1 = 1.f, sh = 0.f;
        if (tid < 96) {
#pragma unroll
            for (int k = 0; k < STAT_COPIES; ++k)
                v0 += *reinterpret_cast<const f64x2*>(a.gn.sums + (size_t)k * a.gn.cstride + (size_t)b * 192 + (size_t)tid * 2);
            const int ch = 32 * cb + (tid & 31);
            ga = a.gn.gamma[ch];
            be = a.gn.beta[ch];
            if (film) {
                sc1 += film[ch];
                sh = film[Cmain + ch];
            }
        }
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
        if (tid < 96) {
            const int sg = tid >> 5, ch = 32 * cb + (tid & 31);
            const float2 mr = s_mr[sg][x3_div(ch, a.gn.gs, a.gn.inv_gs)];
            const float sc = mr.y * ga;
            const float bi = be - sc * mr.x;
            coef[sg][tid & 31] = make_float2(sc * sc1, fmaf(bi, sc1, sh));
        }
        __syncthreads();
    }
    if (bt == 0 && tid < 12) {          // the zero row of this block's (plane, k-group) slabs
        const int p = tid >> 2, k2 = tid & 3;
        *reinterpret_cast<u32x4*>(dst + (((size_t)p * KG + 4 * cb + k2) * rows + (rows - 1)) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    float y[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) { y[u] = u0[u]; y[4 + u] = u1[u]; }
    if (do_gn && tok < L) {
        const int sg = x3_seg(a.seg_src, tok);
        const f32x4* cf = reinterpret_cast<const f32x4*>(&coef[sg][8 * kg]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 kk = cf[u];
            y[2 * u] = fmaf(y[2 * u], kk[0], kk[1]);
            y[2 * u + 1] = fmaf(y[2 * u + 1], kk[2], kk[3]);
        }
        if (a.gn.act) {
#pragma unroll
            for (int u = 0; u < 8; ++u) y[u] = x3_silu(y[u]);
        }
    }
    u32x4 p0, p1, p2;
    x3_split8(y, p0, p1, p2);
    T[(0 * 4 + kg) * X3P_TS + row] = p0;
    T[(1 * 4 + kg) * X3P_TS + row] = p1;
    T[(2 * 4 + kg) * X3P_TS + row] = p2;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int e = tid + 256 * k;
        const int pk = e >> 6, r = e & 63;
        if (row0 + r < L) {
            const int p = pk >> 2, k2 = pk & 3;
            *reinterpret_cast<u32x4*>(dst + (((size_t)p * KG + 4 * cb + k2) * rows + (size_t)b * L + row0 + r) * 16) = T[pk * X3P_TS + r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the gathering GEMM
struct X3Args {
    const char* A3;
    const char* S3;
    const char* W3;
    unsigned long long w3_plane;
    unsigned rowsM, rowsS;
    int KGm, KGs;
    int ldw, N;
    int ntaps, cpt, nmainch, nch;          // taps; 32-channel chunks per tap, of all taps, in total (with the skip part)
    int Lout, Lsrc, Lskip, B;
    const int* gather;
    const int* gather_skip;
    int geo_main, geo_skip, geo_r, geo_t;
    float geo_inv_r;
    const float* bias;
    const float* bias2;
    const float* bias_b;
    int bias_b_stride;
    const float* res;
    float* out;
    SegInfo seg_out;
    StatOut stat[2];
    int nstat;
    unsigned stat_cstride;
    int tiles_per_b, tiles_n;
    float inv_tiles_per_b, inv_tiles_n;
    int nblk, xcd_q, xcd_r;                // workgroups (all K slices); their number / 8 and % 8 (XCD-contiguous block order)
    int ntile;                             // output tiles = nblk / KS
    int KS, cps_q, cps_r;                  // cross-workgroup split-K: slices; chunks per slice nch / KS and nch % KS
    float inv_ntile;
    float* slab;                           // [KS][B][Lout][N] partial tiles
    int* tickets;                          // [ntile] arrival counters (zero between launches)
    unsigned long long* dbg;               // tools/ubench/x3_bench (-DX3_STAMP): phase timestamps; unused otherwise
};

template <int MT, int NT>
struct X3Shape {
    static constexpr int BM = 32 * MT, BN = 64 * NT;
    static constexpr int A_BYTES = 192 * BM, STAGE = 192 * (BM + BN);
    static constexpr int PA = 6 * MT, PW = 12 * NT, PT = PA + PW, P = (PT + 7) / 8;      // DMA pieces per stage: A, W, all, per wave
    static constexpr int NSMAX = ((int)CONV_X3_MAX_LDS - 13 * 1024) / STAGE;            // (13 KB: the largest row table + statistics + dump)
    static constexpr int NS = NSMAX > X3_NSCAP ? X3_NSCAP : NSMAX;
    static_assert(NS >= 2 && (MT % 2) == 0, "tile");
};
// LDS: row table [(ntaps + 1)][BM] ints | statistics slots [3][BN / 4][2] doubles | dump (1 KB, target of the padding pieces) | stages
__host__ __device__ inline int x3_tab_bytes(int BM, int BN, int ntaps) { return ((((ntaps + 1) * BM * 4 + 3 * (BN / 4) * 16 + 1024) + 1023) & ~1023); }

// s_waitcnt immediate (gfx9 encoding): vmcnt = n (bits 3:0 and 15:14), expcnt = 7 (bits 6:4: no wait), lgkmcnt = 0 (bits 11:8)
constexpr int x3_waitcnt_vm_lgkm0(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4); }
constexpr int x3_waitcnt_vm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }        // ... lgkmcnt = 15: no wait

__device__ __forceinline__ void x3_dma16(gchar* g, lchar* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

#ifdef X3_STAMP
#define XSTAMP(it, k) do { if (a.dbg && blockIdx.x == 0 && (wave == 0 || wave == 7) && (it) >= 8 && (it) < 12) { \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) a.dbg[(wave ? 32 : 0) + ((it) - 8) * 8 + (k)] = t_; } } while (0)
#else
#define XSTAMP(it, k) do { } while (0)
#endif

template <int MT, int NT>
__global__ __launch_bounds__(512) void k_conv_x3(const X3Args a) {
    touch_kernargs<(int)sizeof(X3Args)>();
    typedef X3Shape<MT, NT> SH;
    constexpr int BM = SH::BM, BN = SH::BN, NS = SH::NS, P = SH::P, PA = SH::PA, PT = SH::PT;
    extern __shared__ __attribute__((aligned(1024))) char x3_smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & 15, q = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    // XCD-contiguous order: hardware block h runs on XCD h % 8; logical blocks [x * q + min(x, r), ...) go to XCD x, so the column
    // tiles of a row tile (and neighbouring row tiles: shared halo rows) meet in one L2
    int blk = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    {
        const int x = blk & 7, j = blk >> 3;
        blk = (x < a.xcd_r ? x * (a.xcd_q + 1) : a.xcd_r * (a.xcd_q + 1) + (x - a.xcd_r) * a.xcd_q) + j;
    }
    const int ks = __builtin_amdgcn_readfirstlane(a.KS > 1 ? x3_div(blk, a.ntile, a.inv_ntile) : 0);      // K slice (slowest)
    blk -= ks * a.ntile;
    const int tile_id = blk;
    const int bx = __builtin_amdgcn_readfirstlane(x3_div(blk, a.tiles_n, a.inv_tiles_n));      // column tiles fastest
    const int by = blk - bx * a.tiles_n;
    const int c_begin = ks * a.cps_q + min(ks, a.cps_r), c_end = c_begin + a.cps_q + (ks < a.cps_r ? 1 : 0);     // this slice's chunks
    const int b = __builtin_amdgcn_readfirstlane(x3_div(bx, a.tiles_per_b, a.inv_tiles_per_b));
    const int tok0 = (bx - b * a.tiles_per_b) * BM;
    const int n0 = by * BN;
    const int ntaps = a.ntaps;

    int* idx = reinterpret_cast<int*>(x3_smem);
    double* qs = reinterpret_cast<double*>(x3_smem + (ntaps + 1) * BM * 4);
    const int tab = x3_tab_bytes(BM, BN, ntaps);
    lchar* const lds0 = (lchar*)x3_smem;
    const int dump_off = tab - 1024;

    // row table: source ROW (in the planes' row axis: clip offset included) per (tap, tile row), the zero row for padding
    for (int e = tid; e < (ntaps + 1) * BM; e += 512) {
        const int t = e / BM, r = e - t * BM;
        const int tok = tok0 + r;
        int v;
        if (t < ntaps) {
            int st = -1;
            if (tok < a.Lout) {
                if (a.geo_main) {
                    const int ky = t >= 6 ? 2 : (t >= 3 ? 1 : 0);
                    st = geo_source_t<FDiv>(FDiv{a.geo_inv_r}, a.geo_r, a.geo_t, tok, ky, t - 3 * ky, a.geo_main == 2);
                    if (st >= 0) st &= 0x0FFFFFFF;
                } else {
                    st = a.gather ? a.gather[t * a.Lout + tok] : tok;
                }
            }
            v = st < 0 ? (int)a.rowsM - 1 : b * a.Lsrc + st;
        } else {
            int st = 0;
            if (tok < a.Lout) {
                if (a.geo_skip) st = geo_source_t<FDiv>(FDiv{a.geo_inv_r}, a.geo_r, a.geo_t, tok, 1, 1, true) & 0x0FFFFFFF;
                else st = a.gather_skip ? a.gather_skip[tok] : tok;
            }
            v = st;          // (within the clip: the residual read needs it that way; the skip planes add b * Lskip below)
        }
        idx[e] = v;
    }
    for (int e = tid; e < 3 * (BN / 4) * 2; e += 512) qs[e] = 0.0;

    // ---- this wave's DMA pieces: piece id = wave + 8 k; ids [0, PA) are A pieces (plane, k-group, 64-row block), [PA, PT) W
    // pieces (plane, k-group, 64-column block), the rest padding (so that every wave has P operations per stage in flight)
    int pc_lds[P], pc_pk[P], pc_isw[P], pc_slot[P];
    unsigned pc_v[P];            // per lane: W pieces (n0 + column) * 16; A pieces: source row * 16 of the current tap
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int id = wave + 8 * k;
        if (id < PA) {
            const int pk = id / (MT / 2), rb = id - pk * (MT / 2);
            pc_isw[k] = 0;
            pc_pk[k] = pk;
            pc_lds[k] = (pk * BM + rb * 64) * 16;
            pc_slot[k] = rb * 64 + lane;
            pc_v[k] = 0;
        } else {
            const int idw = id < PT ? id - PA : 0;
            const int pk = idw / NT, cb = idw - pk * NT;
            pc_isw[k] = 1;
            pc_pk[k] = pk;
            pc_lds[k] = id < PT ? SH::A_BYTES + (pk * BN + cb * 64) * 16 : -1;
            pc_slot[k] = 0;
            // LDS position -> output column: position = group (16 NT columns of one wn) | nb | lane j  <->  column = group * 16 NT + NT * j + nb
            const int pos = cb * 64 + lane;
            const int grp = pos / (16 * NT), rem = pos - grp * (16 * NT);
            const int nb = rem >> 4, j = rem & 15;
            int col = n0 + grp * 16 * NT + NT * j + nb;
            if (col >= a.ldw) col = 0;          // (a column tile may reach past the padded row of W: never stored)
            pc_v[k] = (unsigned)col * 16u;
        }
    }
    __syncthreads();

    // issue state: the next chunk to request.  A piece's source = piece-constant base (plane, k-group slab) + chunk offset
    // (k-groups advance 4 slabs per chunk; W rows 4 k-group rows per chunk) + the lane's row / column offset.
    const char* pc_base[P];
    auto set_bases = [&](bool skipc) {
        const char* abase = skipc ? a.S3 : a.A3;
        const size_t KG = skipc ? a.KGs : a.KGm, slab = (size_t)(skipc ? a.rowsS : a.rowsM) * 16u;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int p = pc_pk[k] >> 2, kg = pc_pk[k] & 3;
            if (!pc_isw[k]) pc_base[k] = abase + ((size_t)p * KG + kg) * slab;
            else if (!skipc) pc_base[k] = a.W3 + (size_t)p * a.w3_plane + (size_t)kg * (size_t)a.ldw * 16u;
        }
    };
    auto tap_rows = [&](int tapi, bool skipc) {        // A pieces: source row of this lane's item for tap `tapi`
#pragma unroll
        for (int k = 0; k < P; ++k)
            if (!pc_isw[k]) {
                const int v = idx[tapi * BM + pc_slot[k]];
                pc_v[k] = (unsigned)(skipc ? b * a.Lskip + v : v) * 16u;
            }
    };
    const size_t strideW = (size_t)a.ldw * 64u;
    int x_ch = c_begin, x_tap, x_w;
    size_t chA, chW = (size_t)c_begin * strideW, strideA;
    if (c_begin < a.nmainch) {
        x_tap = c_begin / a.cpt;
        x_w = c_begin - x_tap * a.cpt;
        strideA = (size_t)a.rowsM * 64u;
        chA = (size_t)x_w * strideA;
        set_bases(false);
        tap_rows(x_tap, false);
    } else {                                  // (a slice that starts inside the skip part)
        x_tap = ntaps;
        x_w = 0;
        strideA = (size_t)a.rowsS * 64u;
        chA = (size_t)(c_begin - a.nmainch) * strideA;
        set_bases(false);
        set_bases(true);
        tap_rows(ntaps, true);
    }
    auto issue = [&](int st) {
        lchar* const sb = lds0 + tab + st * SH::STAGE;
        if (x_ch < c_end) {
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const char* g = pc_base[k] + (pc_isw[k] ? chW : chA);
                lchar* l = pc_lds[k] < 0 ? lds0 + dump_off : sb + pc_lds[k];
                if constexpr ((X3_ABLATE & 8) != 0) { if (!pc_isw[k]) continue; }
                if constexpr ((X3_ABLATE & 16) != 0) { if (pc_isw[k]) continue; }
                if constexpr (!(X3_ABLATE & 2)) x3_dma16((gchar*)g + pc_v[k], l);
            }
            chW += strideW;
            chA += strideA;
            ++x_ch;
            if (++x_w == a.cpt) {          // next tap (or the skip part, whose chunks simply keep advancing)
                x_w = 0;
                ++x_tap;
                if (x_ch < c_end) {
                    const bool skipc = x_ch >= a.nmainch;
                    if (skipc) {
                        if (x_ch == a.nmainch) {
                            set_bases(true);
                            strideA = (size_t)a.rowsS * 64u;
                            chA = 0;
                            tap_rows(ntaps, true);
                        }
                    } else {
                        chA = 0;
                        tap_rows(x_tap, false);
                    }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < P; ++k)
                if constexpr (!(X3_ABLATE & 2)) x3_dma16((gchar*)a.W3 + (unsigned)lane * 16u, lds0 + dump_off);      // keeps the in-flight count constant at the tail
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[mt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_frags = [&](int st, bf16x8 (&af)[3][MT], bf16x8 (&wf)[3][NT]) {
        const char* Ab = x3_smem + tab + st * SH::STAGE + ((size_t)q * BM + wm * 16 * MT + i) * 16;
        const char* Wb = x3_smem + tab + st * SH::STAGE + SH::A_BYTES + ((size_t)q * BN + wn * 16 * NT + i) * 16;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                if constexpr ((X3_ABLATE & 4) != 0) wf[p][nb] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)p, (unsigned)nb, (unsigned)st, 0x3f803f80u});
                else wf[p][nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Wb + ((size_t)p * 4 * BN + nb * 16) * 16));
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr ((X3_ABLATE & 4) != 0) af[p][mt] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)p, (unsigned)mt, (unsigned)st, 0x3f803f80u});
                else af[p][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ab + ((size_t)p * 4 * BM + 16 * mt) * 16));
            }
        }
    };
    auto mfmas = [&](const bf16x8 (&af)[3][MT], const bf16x8 (&wf)[3][NT]) {
        constexpr int PAi[6] = {2, 1, 0, 1, 0, 0}, PWi[6] = {0, 1, 2, 0, 1, 0};
        if constexpr (X3_PRIO != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) {
                    if constexpr ((X3_ABLATE & 1) != 0) asm volatile("" ::"v"(af[PAi[t]][mt]), "v"(wf[PWi[t]][nb]));
                    else acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[PAi[t]][mt], wf[PWi[t]][nb], acc[mt][nb], 0, 0, 0);
                }
        if constexpr (X3_PRIO != 0) __builtin_amdgcn_s_setprio(0);
    };

    // ---- ring of NS stages: chunk c lives in stage c % NS; D = NS - 1 chunks are requested ahead.
    // The two halves of the workgroup (waves 0-3 | 4-7: one wave of each per SIMD) run HALF AN ITERATION APART, a barrier between
    // half-iterations ("slots"), so that every SIMD always has one wave issuing MFMAs while the other does the memory work
    // (profiles/r03_conv_x3_stamps.txt: in lockstep the 6 DMA issues + 18 fragment reads of a step cost as much as its 48 MFMAs):
    //   slot       2c            2c+1          2c+2
    //   waves 0-3  MEM(c)        MFMA(c)       MEM(c+1)            MEM(c)  = fragment reads of chunk c, request chunk c + D
    //   waves 4-7  MFMA(c-1)     MEM(c)        MFMA(c)                       into the stage of chunk c - 1
    // Chunk c must have landed before slot 2c: every wave waits for ITS pieces of chunk c (all but the (D - 1) P newest) before
    // the barrier that ends slot 2c - 1 -- the first half at the end of MFMA(c-1), the second at the end of MEM(c-1).  The stage
    // of chunk c - 1 is free from slot 2c on (second half read it in slot 2c - 1 and waits lgkmcnt(0) before that slot's barrier).
    {
        constexpr int D = NS - 1;
        bf16x8 af[3][MT], wf[3][NT];
#pragma unroll
        for (int s = 0; s < D; ++s) issue(s);
        __builtin_amdgcn_s_waitcnt(x3_waitcnt_vm_lgkm0((D - 1) * P));
        __builtin_amdgcn_s_barrier();
        int st_c = 0, st_i = D;          // stage of chunk c; stage to refill (= stage of chunk c - 1)
        const int nch = c_end - c_begin;
        if (wave < 4) {
            for (int c = 0; c < nch; ++c) {
                XSTAMP(c, 0);
                load_frags(st_c, af, wf);
                issue(st_i);
                XSTAMP(c, 1);
                __builtin_amdgcn_s_waitcnt(x3_waitcnt_vm_lgkm0(63));          // (lgkmcnt(0): my fragment reads are complete)
                XSTAMP(c, 2);
                __builtin_amdgcn_s_barrier();
                XSTAMP(c, 3);
                mfmas(af, wf);
                XSTAMP(c, 4);
                __builtin_amdgcn_s_waitcnt(x3_waitcnt_vm((D - 1) * P));       // my pieces of chunk c + 1 have landed
                __builtin_amdgcn_s_barrier();
                XSTAMP(c, 5);
                st_c = st_c + 1 == NS ? 0 : st_c + 1;
                st_i = st_i + 1 == NS ? 0 : st_i + 1;
            }
        } else {
            __builtin_amdgcn_s_barrier();
            for (int c = 0; c < nch; ++c) {
                XSTAMP(c, 0);
                load_frags(st_c, af, wf);
                issue(st_i);
                XSTAMP(c, 1);
                __builtin_amdgcn_s_waitcnt(x3_waitcnt_vm_lgkm0((D - 1) * P)); // my pieces of chunk c + 1 have landed; my fragment reads are complete
                XSTAMP(c, 2);
                __builtin_amdgcn_s_barrier();
                XSTAMP(c, 3);
                mfmas(af, wf);
                XSTAMP(c, 4);
                if (c + 1 < nch) __builtin_amdgcn_s_barrier();
                XSTAMP(c, 5);
                st_c = st_c + 1 == NS ? 0 : st_c + 1;
                st_i = st_i + 1 == NS ? 0 : st_i + 1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- lane (i, q) of wave (wm, wn) holds, for row 4q + r of row block mt, the NT consecutive output channels
    // n0 + 16 NT wn + NT i ..
    const bool fast = a.nstat > 0;
    constexpr int QPR = BN / 4;
    const int colw = n0 + wn * 16 * NT + NT * i;
    if (a.KS > 1) {
        // Cross-workgroup split-K completed inside the launch, the protocol of k_conv (conv.hip): every slice parks its partial tile
        // in the slab with write-through (agent-scope) stores and drains them, one lane takes a ticket, the slice that draws the
        // last one re-reads ALL slices in slice order (the result does not depend on who finishes last) and runs the epilogue.
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        typedef __attribute__((address_space(1))) unsigned gu32;
        const size_t sstride = (size_t)a.B * a.Lout * a.N;
        if (colw < a.N) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tok = tok0 + wm * 16 * MT + 16 * mt + 4 * q + r;
                    if (tok >= a.Lout) continue;
                    float* dp = a.slab + (size_t)ks * sstride + ((size_t)b * a.Lout + tok) * a.N + colw;
                    if constexpr (NT == 1) {
                        __hip_atomic_store((gu32*)(unsigned long long)dp, __float_as_uint(acc[mt][0][r]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
#pragma unroll
                        for (int h = 0; h < NT / 2; ++h)
                            __hip_atomic_store((gu64*)(unsigned long long)dp + h, ((unsigned long long)__float_as_uint(acc[mt][2 * h + 1][r]) << 32) | __float_as_uint(acc[mt][2 * h][r]),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* s_last = reinterpret_cast<int*>(x3_smem + dump_off);          // (the padding pieces have landed: the dump is free)
        if (tid == 0) {
            const bool last = __hip_atomic_fetch_add(a.tickets + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.KS - 1;
            if (last) __hip_atomic_store(a.tickets + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // self-cleaning for the next launch
            s_last[0] = last ? 1 : 0;
        }
        __syncthreads();
        if (!s_last[0]) return;
        if (colw < a.N) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tok = tok0 + wm * 16 * MT + 16 * mt + 4 * q + r;
                    if (tok >= a.Lout) continue;
                    const float* sp = a.slab + ((size_t)b * a.Lout + tok) * a.N + colw;
                    float v[NT];
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) v[nb] = 0.f;
                    for (int s2 = 0; s2 < a.KS; ++s2) {
                        if constexpr (NT == 1) {
                            v[0] += __uint_as_float(__hip_atomic_load((gu32*)(unsigned long long)(sp + (size_t)s2 * sstride), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        } else {
#pragma unroll
                            for (int h = 0; h < NT / 2; ++h) {
                                const unsigned long long t = __hip_atomic_load((gu64*)(unsigned long long)(sp + (size_t)s2 * sstride) + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                v[2 * h] += __uint_as_float((unsigned)t);
                                v[2 * h + 1] += __uint_as_float((unsigned)(t >> 32));
                            }
                        }
                    }
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) acc[mt][nb][r] = v[nb];
                }
        }
    }
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
                    const int rs = idx[ntaps * BM + row];
                    const float* rp = a.res + ((size_t)b * a.Lskip + rs) * a.N + colw;
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) o[nb] += rp[nb];
                }
                float* op = a.out + ((size_t)b * a.Lout + tok) * a.N + colw;
                if constexpr (NT == 4) *reinterpret_cast<f32x4*>(op) = f32x4{o[0], o[1], o[2], o[3]};
                else if constexpr (NT == 2) *reinterpret_cast<f32x2*>(op) = f32x2{o[0], o[1]};
                else op[0] = o[0];
                if (fast) {
                    const int sgq = x3_seg(a.seg_out, tok);
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
    for (int e = tid; e < a.nstat * 3 * QPR; e += 512) {
        const int t = e / (3 * QPR), r2 = e - t * 3 * QPR;
        const int sgi = r2 / QPR, cq = r2 - sgi * QPR;
        const int n = n0 + cq * 4;
        if (n >= a.N) continue;
        const int gs = a.stat[t].gs, coff = a.stat[t].coff;
        const float inv_gs = a.stat[t].inv_gs;
        const int g = x3_div(coff + n, gs, inv_gs);
        if (cq > 0 && x3_div(coff + n - 4, gs, inv_gs) == g) continue;
        const int qend = min(QPR, min((a.N - n0 + 3) >> 2, ((g + 1) * gs - coff - n0 + 3) >> 2));
        double s = 0.0, ss = 0.0;
        for (int c2 = cq; c2 < qend; ++c2) {
            s += qs[(sgi * QPR + c2) * 2];
            ss += qs[(sgi * QPR + c2) * 2 + 1];
        }
        if (ss != 0.0) {
            double* dstp = a.stat[t].sums + (size_t)(blockIdx.x & (STAT_COPIES - 1)) * a.stat_cstride + (((size_t)b * 3 + sgi) * 32 + g) * 2;
            atomicAdd(dstp, s);
            atomicAdd(dstp + 1, ss);
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
    x3_split8(y, p0, p1, p2);
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
bool conv_x3_eligible(const ConvArgs& a) {
    if (!a.W3 || a.out_cm || a.ddim || (a.N & 3) || a.N < 64) return false;
    if ((a.Cmain & 31) || (a.Cskip & 31)) return false;
    for (int k = 0; k < 4; ++k)
        if (a.C[k] & 7) return false;
    for (int t = 0; t < a.nstat; ++t)
        if ((a.stat[t].gs & 3) || (a.stat[t].coff & 3)) return false;      // (the statistics epilogue assigns whole 4-column quads to a group)
    return true;
}

// bytes of the split-activation scratch a conv needs (main planes, then the skip planes)
static size_t x3_main_bytes(int B, int Lsrc, int Cmain) { return (((size_t)3 * (Cmain / 8) * ((size_t)B * Lsrc + 1) * 16) + 255) & ~(size_t)255; }
size_t conv_x3_scratch_bytes(int B, int Lsrc, int Cmain, int Lskip, int Cskip) {
    return x3_main_bytes(B, Lsrc, Cmain) + (Cskip ? (size_t)3 * (Cskip / 8) * ((size_t)B * Lskip + 1) * 16 : 0) + 256;
}

template <int MT, int NT>
static size_t x3_lds_t(int ntaps) { return (size_t)x3_tab_bytes(32 * MT, 64 * NT, ntaps) + (size_t)X3Shape<MT, NT>::NS * X3Shape<MT, NT>::STAGE; }

#define X3_TILES(F) F(4, 2) F(8, 2) F(4, 4) F(2, 2) F(4, 1) F(2, 1) F(8, 1)

bool x3_tile_exists(int MT, int NT) {
#define X3_E(M, N_) if (MT == M && NT == N_) return true;
    X3_TILES(X3_E)
#undef X3_E
    return false;
}

size_t conv_x3_smem_bytes(const ConvArgs& a, ConvTile t) {
#define X3_S(M, N_) if (t.MT == M && t.NT == N_) return x3_lds_t<M, N_>(a.ntaps);
    X3_TILES(X3_S)
#undef X3_S
    return (size_t)1 << 30;
}

static hipError_t launch_x3_prep(const ConvArgs& a, hipStream_t s, const float* geglu_h = nullptr) {
    X3Prep p{};
    for (int k = 0; k < 4; ++k) { p.src[k] = a.src[k]; p.C[k] = a.C[k]; }
    if (geglu_h) {
        if (a.nmain != 1 || a.Cskip || a.gn.sums) return hipErrorInvalidValue;
        p.src[0] = geglu_h;
        p.geglu = 1;
    }
    p.Cmain = a.Cmain; p.Cskip = a.Cskip; p.Lsrc = a.Lsrc; p.Lskip = a.Lskip; p.B = a.B;
    p.gn = a.gn;
    p.seg_src = a.seg_src;
    p.A3 = reinterpret_cast<char*>(a.x3);
    p.S3 = p.A3 + x3_main_bytes(a.B, a.Lsrc, a.Cmain);
    p.rowsM = (unsigned)a.B * a.Lsrc + 1;
    p.rowsS = (unsigned)a.B * a.Lskip + 1;
    p.tilesM = (a.Lsrc + 63) / 64;
    p.tilesS = a.Cskip ? (a.Lskip + 63) / 64 : 0;
    p.cbM = a.Cmain / 32;
    p.cbS = a.Cskip / 32;
    hipLaunchKernelGGL(k_x3_prep, dim3((unsigned)(a.B * (p.tilesM * p.cbM + p.tilesS * p.cbS))), dim3(256), 0, s, p);
    return hipGetLastError();
}

template <int MT, int NT>
static hipError_t launch_x3_t(const ConvArgs& a, int KSv, hipStream_t s) {
    constexpr int BM = 32 * MT, BN = 64 * NT;
    X3Args x{};
    x.A3 = reinterpret_cast<const char*>(a.x3);
    x.S3 = x.A3 + x3_main_bytes(a.B, a.Lsrc, a.Cmain);
    x.W3 = reinterpret_cast<const char*>(a.W3);
    x.w3_plane = a.w3_plane;
    x.rowsM = (unsigned)a.B * a.Lsrc + 1;
    x.rowsS = (unsigned)a.B * a.Lskip + 1;
    x.KGm = a.Cmain / 8; x.KGs = a.Cskip / 8;
    x.ldw = a.ldw; x.N = a.N;
    x.ntaps = a.ntaps; x.cpt = a.Cmain / 32; x.nmainch = a.ntaps * x.cpt; x.nch = x.nmainch + a.Cskip / 32;
    x.Lout = a.Lout; x.Lsrc = a.Lsrc; x.Lskip = a.Lskip; x.B = a.B;
    x.gather = a.gather; x.gather_skip = a.gather_skip;
    x.geo_main = a.geo_main; x.geo_skip = a.geo_skip; x.geo_r = a.geo_r; x.geo_t = a.geo_t;
    x.geo_inv_r = a.geo_r > 0 ? 1.0f / (float)a.geo_r : 0.f;
    x.bias = a.bias; x.bias2 = a.bias2; x.bias_b = a.bias_b; x.bias_b_stride = a.bias_b_stride;
    x.res = a.res; x.out = a.out; x.seg_out = a.seg_out;
    x.stat[0] = a.stat[0]; x.stat[1] = a.stat[1]; x.nstat = a.nstat; x.stat_cstride = a.stat_cstride;
    const int tiles = (a.Lout + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int KS = KSv < 1 ? 1 : KSv;
    const long ntile = (long)a.B * tiles * tiles_n, nblk = ntile * KS;
    if (nblk >= (1L << 21) || KS > x.nch || (KS > 1 && (!a.slab || !a.tickets))) return hipErrorInvalidValue;
    x.ntile = (int)ntile; x.inv_ntile = 1.0f / (float)ntile;
    x.KS = KS; x.cps_q = x.nch / KS; x.cps_r = x.nch % KS;
    x.slab = a.slab; x.tickets = a.tickets;
    x.tiles_per_b = tiles; x.tiles_n = tiles_n;
    x.inv_tiles_per_b = 1.0f / (float)tiles; x.inv_tiles_n = 1.0f / (float)tiles_n;
    x.nblk = (int)nblk; x.xcd_q = (int)(nblk / 8); x.xcd_r = (int)(nblk % 8);
    x.dbg = a.dbg;
    const size_t smem = x3_lds_t<MT, NT>(a.ntaps);
    if (smem > CONV_X3_MAX_LDS) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_conv_x3<MT, NT>), dim3((unsigned)nblk), dim3(512), smem, s, x);
    return hipGetLastError();
}

hipError_t conv_x3_init_attrs() {
#define X3_A(M, N_) { const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_x3<M, N_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CONV_X3_MAX_LDS); if (e != hipSuccess) return e; }
    X3_TILES(X3_A)
#undef X3_A
    return hipSuccess;
}

// tiles are encoded as ConvTile{MT, NT, NW = 48, KS, XM = 0}; the launch is the elementwise pass + the GEMM (KS K slices per tile)
hipError_t launch_conv_x3(const ConvArgs& a, ConvTile t, hipStream_t s) { return launch_conv_x3_geglu(a, t, nullptr, s); }

// ... with the input taken as GEGLU(h), h [B * rows][2 Cmain] (the elementwise pass applies it: no separate k_geglu launch, no
// [rows][Cmain] round trip); h = nullptr: the plain form
hipError_t launch_conv_x3_geglu(const ConvArgs& a, ConvTile t, const float* h, hipStream_t s) {
    if (!conv_x3_eligible(a) || !a.x3) return hipErrorInvalidValue;
    hipError_t e = launch_x3_prep(a, s, h);
    if (e != hipSuccess) return e;
#define X3_L(M, N_) if (t.MT == M && t.NT == N_) return launch_x3_t<M, N_>(a, t.KS, s);
    X3_TILES(X3_L)
#undef X3_L
    return hipErrorInvalidValue;
}

}  // namespace mtv
