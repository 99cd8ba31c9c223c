// stage_bench -- VERDICT r5 item 2: ONE real level-0 stage of the base UNet (`in1`: ResBlock(128 -> 128) + AttentionBlock x3 planes +
// AttentionBlock1D, MToV/models/ddpm/unet.py:1031-1053 with :178-207, :210-300, :303-326) at B = 1, (R, T) = (32, 16) -- 2048 tokens
// (xy 32x32 | yt 16x32 | xt 16x32), C = 128, 8 heads of d = 16 -- built twice from the SAME eight phase bodies:
//   chain      : eight dependent launches per stage (what csrc/plan.hip does today: one launch per op), one hipGraph;
//   persistent : ONE resident launch per stage (or per run of stages), 256 workgroups x 512 threads, the eight dependency edges inside
//                the launch -- producers store write-through (sc0 sc1), drain, arrive on XCD-sharded counters; consumers poll the
//                counters and read with L1-bypassing loads; GroupNorm statistics are the only all-to-all payload (fp64 atomics into
//                8 privatised tables, as the product's epilogues do); the weights of phase k + 1 are requested BEFORE the edge.
// Both are checked against a double-precision CPU restatement of the stage, then timed as R stages back to back with distinct
// weights.  The question it answers: what does an in-launch edge cost against a kernel boundary on THIS chain.
//   stage_bench check           correctness of both forms (stage 0 and the last stage of a 3-stage run)
//   stage_bench time [R]        us per stage: chain | persistent per stage | persistent over all R stages | per-phase stamps
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int L = 2048, C = 128, H = 8, DH = 16, NQKV = 384;
constexpr int PB1 = 1024, PB2 = 1536;                 // plane boundaries: xy [0, 1024) 32 rows | yt [1024, 1536) 16 rows | xt [1536, 2048) 16 rows; row width 32
constexpr int NTH = 512, NWG = 256;
constexpr int STAT_COPIES = 8, STAT_DOUBLES = 3 * 32 * 2;           // one site: [copy][plane][group][sum | sumsq]
constexpr int NSITE = 5;                               // x, h1, h2 (per plane) , h3 (whole), h4 = the next stage's x
constexpr int LSTR = 136;                              // LDS row stride (floats) of a 128-channel row: 16 distinct bank quads per ds_read_b128 lane group
constexpr int WIN_ROWS = 3 * 34;                       // conv window: 3 image rows x (32 + 2 zero pads)
constexpr int LDS_FLOATS = WIN_ROWS * LSTR + 8 * 1024 + 256;        // (qkv: rows 32 x 136 + 64 + image 32 x 100 fits)        // window | 8 partial 32x32 tiles | scratch
constexpr int NEDGE = 8;

struct StageW {
    const float *gn1_g, *gn1_b, *w1, *bias1;           // in_layers: GroupNorm, conv3x3 as [tap][c / 4][n][4]
    const float *gn2_g, *gn2_b, *film, *w2, *bias2;    // out_layers: GroupNorm x FiLM (scale [0, C), shift [C, 2C)), conv3x3
    const float *ga_g, *ga_b, *wqa, *bqa, *wpa, *bpa;  // AttentionBlock (per plane): norm, qkv [c / 4][384][4], proj [c / 4][128][4]
    const float *gb_g, *gb_b, *wqb, *bqb, *wpb, *bpb;  // AttentionBlock1D (all tokens)
};
struct StageBufs {
    const float* x;                                    // [L][C] the stage's input (the previous stage's h4)
    float *h1, *h2, *qkv, *att, *h3, *h4;
    double* stats;                                     // [NSITE][STAT_COPIES][STAT_DOUBLES]; site 0 filled by the previous stage (or k_input_stats)
    double* stats_next;                                // site 0 of the next stage
    unsigned long long* bar;                           // [NEDGE][8 shards][8] arrival counters (zero at graph start)
    unsigned long long* dbg;                           // phase stamps of workgroup 0 (or nullptr)
};

// -------------------------------------------------------------------------------------------------------------------- memory policy
// COH = true (persistent form): everything another workgroup of the SAME launch wrote / will read goes write-through and is read past
// the L1 (MI355X_MICROARCH.md, inter-workgroup visibility: "{sc0 sc1 stores and loads both sides}").  COH = false: plain.
template <bool COH> __device__ __forceinline__ f32x4 ld16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, COH ? 17 : 0));
}
template <bool COH> __device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, byte_off, 0, COH ? 17 : 0);
}
template <bool COH> __device__ __forceinline__ void st8(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float a, float b) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(a), __float_as_uint(b)}, rs, byte_off, 0, COH ? 17 : 0);
}
template <bool COH> __device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float a) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a), rs, byte_off, 0, COH ? 17 : 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ int plane_of(int tok) { return tok >= PB2 ? 2 : (tok >= PB1 ? 1 : 0); }

// -------------------------------------------------------------------------------------------------------------------- in-launch edge
// All stores of the phase are write-through; every wave drains its own (vmcnt counts stores and atomics on gfx9), the workgroup
// meets, one lane arrives on its shard, wave 0 polls the eight shards past the L1.  (One counter set per edge, zeroed at graph start.)
__device__ __forceinline__ void grid_edge(unsigned long long* cnt, int wg, int tid, unsigned long long* dbg = nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dbg && wg == 0 && tid == 0) dbg[0] = __builtin_amdgcn_s_memtime();        // this wave's stores and atomics acknowledged
    __syncthreads();
    if (dbg && wg == 0 && tid == 0) dbg[1] = __builtin_amdgcn_s_memtime();        // ... every wave's
    if (tid == 0) __hip_atomic_fetch_add(cnt + (wg & 7) * 8, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        const __amdgpu_buffer_rsrc_t rs = rsrc(cnt, 8 * 8 * 8);
        for (int tries = 0; tries < (1 << 22); ++tries) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)(tid & 7) * 64u, 0, 17);
            if (__all((int)(v[0] >= (unsigned)(NWG / 8)))) break;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

#define STAMP(k) do { if (dbg && wg == 0 && tid == 0) dbg[k] = __builtin_amdgcn_s_memtime(); } while (0)

// -------------------------------------------------------------------------------------------------------------------- GroupNorm inputs
// (mean, rstd) of the 32 groups of plane `p` (or of all planes: whole) from the 8 privatised (sum, sumsq) tables -> LDS ms[32][2]
template <bool COH>
__device__ __forceinline__ void gn_moments(const double* site, int p, bool whole, float* ms, int tid) {
    if (tid < 32) {
        const __amdgpu_buffer_rsrc_t rs = rsrc(site, STAT_COPIES * STAT_DOUBLES * 8);
        double s = 0.0, ss = 0.0;
        const int p0 = whole ? 0 : p, p1 = whole ? 3 : p + 1;
        for (int pp = p0; pp < p1; ++pp)
#pragma unroll
            for (int cpy = 0; cpy < STAT_COPIES; ++cpy) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((cpy * 3 + pp) * 32 + tid) * 16u, 0, COH ? 17 : 0);
                s += __hiloint2double((int)v[1], (int)v[0]);
                ss += __hiloint2double((int)v[3], (int)v[2]);
            }
        const double n = whole ? (double)L * 4.0 : (p == 0 ? 1024.0 : 512.0) * 4.0;
        const double mean = s / n, var = ss / n - mean * mean;
        ms[2 * tid] = (float)mean;
        ms[2 * tid + 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
    }
}

// Epilogue shared by the conv and the proj phases: `nparts` partial 32 x 32 tiles in LDS red[part][32][32] -> + bias (+ residual) -> out rows
// [r0, r0 + 32) x columns [n0, n0 + 32), and the (sum, sumsq) of the 8 GroupNorm groups of those columns into the producer's statistics site.
template <bool COH>
__device__ __forceinline__ void tile_epilogue(const float* red, int nparts, const float* bias, const float* res, float* out, double* site, int r0, int n0, int wg,
                                              int tid, double* dsum /* LDS [8][2] */) {
    const int row = tid >> 4, cp = (tid & 15) * 2;
    float v0 = 0.f, v1 = 0.f;
    for (int w = 0; w < nparts; ++w) {
        v0 += red[(w * 32 + row) * 32 + cp];
        v1 += red[(w * 32 + row) * 32 + cp + 1];
    }
    v0 += bias[n0 + cp];
    v1 += bias[n0 + cp + 1];
    const unsigned off = (unsigned)((r0 + row) * C + n0 + cp) * 4u;
    if (res) {
        const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc(res, L * C * 4), off, 0, COH ? 17 : 0);
        v0 += __uint_as_float(r[0]);
        v1 += __uint_as_float(r[1]);
    }
    // 16-byte write-through stores: pairs of threads meet through a lane swap (the even one stores the quad)
    {
        const float w0 = __shfl_xor(v0, 1), w1 = __shfl_xor(v1, 1);
        if (!(tid & 1)) st16<COH>(rsrc(out, L * C * 4), off, f32x4{v0, v1, w0, w1});
    }
    if (site) {
        if (tid < 16) dsum[tid] = 0.0;
        __syncthreads();
        double s = (double)v0 + (double)v1, ss = (double)v0 * v0 + (double)v1 * v1;
        // a group = 4 channels = 2 adjacent threads; a wave = 4 rows x 16 threads: fold the pair and the 4 rows, then one LDS atomic per (wave, group)
        s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);
        s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
        s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
        if ((tid & 63) < 16 && !(tid & 1)) {
            atomicAdd(&dsum[(tid & 15) >> 1 << 1], s);
            atomicAdd(&dsum[((tid & 15) >> 1 << 1) + 1], ss);
        }
        __syncthreads();
        if (tid < 16) {
            const int g = (n0 >> 2) + (tid >> 1);
            atomicAdd(site + (((wg & 7) * 3 + plane_of(r0)) * 32 + g) * 2 + (tid & 1), dsum[tid]);
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------- conv3x3 phase
// workgroup = (image row y of a plane = 32 tokens, 32 output channels); wave w = input channels [16 w, 16 w + 16) of all nine taps.
#ifndef SB_WRING
#define SB_WRING 9
#endif
struct ConvPre { f32x4 wf[9][2]; const float* Wr; };
__device__ __forceinline__ void conv_wtap(ConvPre& pre, int t, int slot, int wg, int tid) {
    const int ct = wg & 3, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs = rsrc(pre.Wr, 9 * 32 * C * 16);
#ifdef SB_WKN
    // the SAME bytes read as the product reads its [krow][N] matrix: row = t * 128 + 16 wave + 4 g + s, columns 32 ct + 2 j, + 1 -> element [s] of
    // fragment n = the value for column 32 ct + 2 j + n (the column permutation only relabels which weight a lane multiplies: timing experiment)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)((t * 128 + 16 * wave + 4 * g + e) * C + 32 * ct + 2 * j) * 4u, 0, 0);
        pre.wf[slot][0][e] = __uint_as_float(v[0]);
        pre.wf[slot][1][e] = __uint_as_float(v[1]);
    }
#else
#pragma unroll
    for (int n = 0; n < 2; ++n) pre.wf[slot][n] = ld16<false>(rs, (unsigned)(((t * 32 + 4 * wave + g) * C) + 32 * ct + 16 * n + j) * 16u);
#endif
}
__device__ __forceinline__ void conv_pre(ConvPre& pre, const float* Wr, int wg, int tid) {
    const int ct = wg & 3, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs = rsrc(Wr, 9 * 32 * C * 16);
    pre.Wr = Wr;
    (void)ct; (void)wave; (void)j; (void)g; (void)rs;
#pragma unroll
    for (int t = 0; t < SB_WRING; ++t) conv_wtap(pre, t, t, wg, tid);
}
template <bool COH>
__device__ __forceinline__ void conv_main(ConvPre& pre, const float* src, const double* site_in, const float* gamma, const float* beta, const float* film,
                                          const float* bias, const float* res, float* out, double* site_out, int wg, int tid, float* lds, unsigned long long* cdbg = nullptr) {
#define CST(k) do { if (cdbg && wg == 0 && tid == 0) cdbg[k] = __builtin_amdgcn_s_memtime(); } while (0)
    CST(0);
    const int rt = wg >> 2, ct = wg & 3, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int r0 = 32 * rt, p = plane_of(r0), pbase = p == 0 ? 0 : (p == 1 ? PB1 : PB2), hrows = p == 0 ? 32 : 16, y = (r0 - pbase) >> 5;
    float* win = lds;
    float* red = lds + WIN_ROWS * LSTR;
    float* ms = red + 8 * 1024;
    double* dsum = reinterpret_cast<double*>(ms + 64);
    // raw window rows (3 image rows x 32 tokens x 32 channel quads = 6 quads per thread) requested together with the statistics
    const __amdgpu_buffer_rsrc_t srs = rsrc(src, L * C * 4);
    f32x4 raw[6];
    const int c4 = tid & 31;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int row = (tid >> 5) + 16 * u, ky = row >> 5, xx = row & 31, yy = y + ky - 1;
        const bool ok = yy >= 0 && yy < hrows;
        raw[u] = ld16<COH>(srs, ok ? (unsigned)((pbase + yy * 32 + xx) * C + 4 * c4) * 4u : 0xFFFFFFF0u);     // (out of range: zeros)
    }
    CST(1);
    gn_moments<COH>(site_in, p, false, ms, tid);
    // zero pads of the window: columns 0 and 33 of each of the 3 rows
    if (tid < 6 * 32) *reinterpret_cast<f32x4*>(win + ((tid >> 5) / 2 * 34 + ((tid >> 5) & 1) * 33) * LSTR + 4 * (tid & 31)) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    CST(2);
    {
        const float mean = ms[2 * c4], rstd = ms[2 * c4 + 1];
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * c4), bt = *reinterpret_cast<const f32x4*>(beta + 4 * c4);
        f32x4 A = gm * rstd, Bc = bt - A * mean;
        if (film) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(film + 4 * c4) + 1.0f, sh = *reinterpret_cast<const f32x4*>(film + C + 4 * c4);
            A = A * sc;
            Bc = Bc * sc + sh;
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int row = (tid >> 5) + 16 * u, ky = row >> 5, xx = row & 31, yy = y + ky - 1;
            f32x4 v = raw[u] * A + Bc;
            v = f32x4{silu(v[0]), silu(v[1]), silu(v[2]), silu(v[3])};
            if (!(yy >= 0 && yy < hrows)) v = f32x4{0.f, 0.f, 0.f, 0.f};                 // zero padding is applied AFTER the activation (F.conv2d pads the activated tensor)
            *reinterpret_cast<f32x4*>(win + (ky * 34 + xx + 1) * LSTR + 4 * c4) = v;
        }
    }
    __syncthreads();
    CST(3);
    f32x4 acc[2][2] = {};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - 3 * ky;
        f32x4 af[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) af[m] = *reinterpret_cast<const f32x4*>(win + (ky * 34 + 16 * m + j + kx) * LSTR + 16 * wave + 4 * g);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][s], pre.wf[t % SB_WRING][n][s], acc[m][n], 0, 0, 0);
        if (SB_WRING < 9 && t + SB_WRING < 9) conv_wtap(pre, t + SB_WRING, t % SB_WRING, wg, tid);      // (experiment: refill the slot just used)
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 32 + 16 * m + 4 * g + r) * 32 + 16 * n + j] = acc[m][n][r];
    CST(4);
    __syncthreads();
    CST(5);
    tile_epilogue<COH>(red, 8, bias, res, out, site_out, r0, 32 * ct, wg, tid, dsum);
    CST(6);
#undef CST
}

// -------------------------------------------------------------------------------------------------------------------- qkv phase (1x1, GroupNorm prologue)
// workgroup = (32 rows, 96 of the 384 output channels); waves 0 .. 5 own one 16-column tile each, whole K.
struct QkvPre { f32x4 wf[8]; };
__device__ __forceinline__ void qkv_pre(QkvPre& pre, const float* Wr, int wg, int tid) {
    const int ct = wg & 3, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs = rsrc(Wr, 32 * NQKV * 16);
    const int col = 96 * ct + 16 * (wave < 6 ? wave : 0) + j;
#pragma unroll
    for (int c = 0; c < 8; ++c) pre.wf[c] = ld16<false>(rs, (unsigned)((4 * c + g) * NQKV + col) * 16u);
}
template <bool COH>
__device__ __forceinline__ void qkv_main(const QkvPre& pre, const float* src, const double* site_in, bool whole, const float* gamma, const float* beta, const float* bias,
                                         float* out, int wg, int tid, float* lds) {
    const int rt = wg >> 2, ct = wg & 3, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int r0 = 32 * rt, p = plane_of(r0);
    float* rows = lds;
    float* ms = lds + 32 * LSTR;
    const __amdgpu_buffer_rsrc_t srs = rsrc(src, L * C * 4);
    const int c4 = tid & 31;
    f32x4 raw[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) raw[u] = ld16<COH>(srs, (unsigned)((r0 + (tid >> 5) + 16 * u) * C + 4 * c4) * 4u);
    gn_moments<COH>(site_in, p, whole, ms, tid);
    __syncthreads();
    {
        const float mean = ms[2 * c4], rstd = ms[2 * c4 + 1];
        const f32x4 A = *reinterpret_cast<const f32x4*>(gamma + 4 * c4) * rstd, Bc = *reinterpret_cast<const f32x4*>(beta + 4 * c4) - A * mean;
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<f32x4*>(rows + ((tid >> 5) + 16 * u) * LSTR + 4 * c4) = raw[u] * A + Bc;
    }
    __syncthreads();
    if (wave < 6) {
        f32x4 acc[2] = {};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            f32x4 af[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) af[m] = *reinterpret_cast<const f32x4*>(rows + (16 * m + j) * LSTR + 16 * c + 4 * g);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][s], pre.wf[c][s], acc[m], 0, 0, 0);
        }
        const float b = bias[96 * ct + 16 * wave + j];
        float* img = lds + 32 * LSTR + 64;                      // [32 rows][100]: the tile as an image, so that the stores are 16 bytes wide
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * m + 4 * g + r) * 100 + 16 * wave + j] = acc[m][r] + b;
    }
    __syncthreads();
    {
        const float* img = lds + 32 * LSTR + 64;
        const __amdgpu_buffer_rsrc_t ors = rsrc(out, L * NQKV * 4);
        for (int e = tid; e < 32 * 24; e += NTH) {              // 32 rows x 24 column quads
            const int row = e / 24, cq = e - row * 24;
            st16<COH>(ors, (unsigned)((r0 + row) * NQKV + 96 * ct + 4 * cq) * 4u, *reinterpret_cast<const f32x4*>(img + row * 100 + 4 * cq));
        }
    }
}

// -------------------------------------------------------------------------------------------------------------------- proj phase (1x1 + residual + statistics)
// workgroup = (32 rows, 32 columns); wave w = tile (m = w & 1, n = (w >> 1) & 1), K half (w >> 2)
struct ProjPre { f32x4 wf[4]; };
__device__ __forceinline__ void proj_pre(ProjPre& pre, const float* Wr, int wg, int tid) {
    const int ct = wg & 3, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs = rsrc(Wr, 32 * C * 16);
    const int n = (wave >> 1) & 1, kh = wave >> 2;
#pragma unroll
    for (int c = 0; c < 4; ++c) pre.wf[c] = ld16<false>(rs, (unsigned)((4 * (4 * kh + c) + g) * C + 32 * ct + 16 * n + j) * 16u);
}
template <bool COH>
__device__ __forceinline__ void proj_main(const ProjPre& pre, const float* src, const float* bias, const float* res, float* out, double* site_out, int wg, int tid, float* lds) {
    const int rt = wg >> 2, ct = wg & 3, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4;
    const int r0 = 32 * rt, m = wave & 1, n = (wave >> 1) & 1, kh = wave >> 2;
    float* red = lds;
    double* dsum = reinterpret_cast<double*>(lds + 2 * 1024);
    const __amdgpu_buffer_rsrc_t srs = rsrc(src, L * C * 4);
    f32x4 af[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) af[c] = ld16<COH>(srs, (unsigned)((r0 + 16 * m + j) * C + 16 * (4 * kh + c) + 4 * g) * 4u);
    f32x4 acc = {};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][s], pre.wf[c][s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(kh * 32 + 16 * m + 4 * g + r) * 32 + 16 * n + j] = acc[r];
    __syncthreads();
    tile_epilogue<COH>(red, 2, bias, res, out, site_out, r0, 32 * ct, wg, tid, dsum);
}

// -------------------------------------------------------------------------------------------------------------------- attention phase
// QKVAttentionLegacy (unet.py:303-326): channel = head * 48 + {q: 0.., k: 16.., v: 32..}; softmax((q d^-1/4)(k d^-1/4)^T) v.
// One wave: NQ query tiles of 16 (consecutive tokens from q0) against keys [k0, k0 + nk): S^T = K Q^T so that a query is a lane COLUMN
// (row maxima / sums are register values + two lane swaps, and the P^T registers are directly the B operand of O^T += V^T P^T);
// V^T goes through a per-wave LDS tile.  Returns un-normalised O^T, the running maximum (log2 domain) and the row sum.
template <bool COH, int NQ>
__device__ __forceinline__ void attn_wave(const __amdgpu_buffer_rsrc_t qrs, int head, int q0, int k0, int nk, float* vt /* per wave [16][20] */, f32x4 (&O)[NQ], float (&M)[NQ],
                                          float (&Ls)[NQ], int lane) {
    const int j = lane & 15, g = lane >> 4;
    const float sc = 0.25f * 1.4426950408889634f;                    // d^-1/2 (= d^-1/4 on q and on k) and log2(e)
    f32x4 qf[NQ];
#pragma unroll
    for (int a = 0; a < NQ; ++a) {
        qf[a] = ld16<COH>(qrs, (unsigned)((q0 + 16 * a + j) * NQKV + head * 48 + 4 * g) * 4u) * sc;
        O[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        M[a] = -INFINITY;
        Ls[a] = 0.f;
    }
    const int nt = nk >> 4;
    f32x4 kf[2], vf[2];
    auto issue = [&](int t, int slot) {
        const unsigned o = (unsigned)((k0 + 16 * t + j) * NQKV + head * 48 + 16 + 4 * g) * 4u;
        kf[slot] = ld16<COH>(qrs, t < nt ? o : 0xFFFFFFF0u);
        vf[slot] = ld16<COH>(qrs, t < nt ? o + 64u : 0xFFFFFFF0u);
    };
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (cur == 0) issue(t + 1, 1); else issue(t + 1, 0);
        const f32x4 kk = cur ? kf[1] : kf[0], vv = cur ? vf[1] : vf[0];
        // V^T tile: lane (key j, quad g) holds v[key][4g ..]; stored as vt[dim][key]
#pragma unroll
        for (int c = 0; c < 4; ++c) vt[(4 * g + c) * 20 + j] = vv[c];
        const f32x4 va = *reinterpret_cast<const f32x4*>(vt + j * 20 + 4 * g);      // lane (dim j, quad g): V^T[dim][key 4g ..]
#pragma unroll
        for (int a = 0; a < NQ; ++a) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_16x16x4f32(kk[e], qf[a][e], s, 0, 0, 0);       // lane (query j, g): keys 4g .. 4g + 3
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(M[a], mx), alpha = __builtin_amdgcn_exp2f(M[a] - mn);
            M[a] = mn;
            f32x4 pq;
#pragma unroll
            for (int e = 0; e < 4; ++e) pq[e] = __builtin_amdgcn_exp2f(s[e] - mn);
            Ls[a] = Ls[a] * alpha + ((pq[0] + pq[1]) + (pq[2] + pq[3]));
            O[a] = O[a] * alpha;
#pragma unroll
            for (int e = 0; e < 4; ++e) O[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[e], pq[e], O[a], 0, 0, 0);     // O^T[dim 4g + r][query j]
        }
    }
#pragma unroll
    for (int a = 0; a < NQ; ++a) {
        Ls[a] += __shfl_xor(Ls[a], 16);
        Ls[a] += __shfl_xor(Ls[a], 32);
    }
}
// merge scratch: part[wave][tile][16 dims][16 queries] + (m, l)[wave][tile][16]
template <int NQ>
__device__ __forceinline__ void attn_park(float* mrg, int wave, const f32x4 (&O)[NQ], const float (&M)[NQ], const float (&Ls)[NQ], int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int a = 0; a < NQ; ++a) {
        float* o = mrg + (wave * 4 + a) * 288;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(4 * g + r) * 16 + j] = O[a][r];
        if (g == 0) { o[256 + j] = M[a]; o[272 + j] = Ls[a]; }
    }
}
// out[token q0 + 16 a + query][head * 16 + dim] = sum over the 8 key parts, normalised
template <bool COH>
__device__ __forceinline__ void attn_merge(const float* mrg, int ntile, int head, int q0, float* out, int tid) {
    const int a = tid >> 6, q = (tid >> 2) & 15, dq = tid & 3;       // 4 tiles x 16 queries x 4 dim quads = 256 threads
    if (tid < 64 * ntile) {
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < 8; ++w) mx = fmaxf(mx, mrg[(w * 4 + a) * 288 + 256 + q]);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float* p = mrg + (w * 4 + a) * 288;
            const float f = __builtin_amdgcn_exp2f(p[256 + q] - mx);
            l += p[272 + q] * f;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += p[(4 * dq + e) * 16 + q] * f;
        }
        st16<COH>(rsrc(out, L * C * 4), (unsigned)((q0 + 16 * a + q) * C + head * 16 + 4 * dq) * 4u, o * (1.0f / l));
    }
}
// per-plane attention: workgroup = (head, j): xy queries [32 j, 32 j + 32) over 1024 keys, yt / xt queries [16 j, 16 j + 16) over 512 keys each -- the
// same arithmetic per workgroup everywhere; every wave takes an eighth of the keys
template <bool COH>
__device__ __forceinline__ void attn2d_main(const float* qkv, float* out, int wg, int tid, float* lds) {
    const int head = wg & 7, jj = wg >> 3, wave = tid >> 6, lane = tid & 63;
    float* vt = lds + wave * 320;
    float* mrg = lds + 8 * 320;
    const __amdgpu_buffer_rsrc_t qrs = rsrc(qkv, L * NQKV * 4);
    {
        f32x4 O[2]; float M[2], Ls[2];
        attn_wave<COH, 2>(qrs, head, 32 * jj, 128 * wave, 128, vt, O, M, Ls, lane);
        attn_park<2>(mrg, wave, O, M, Ls, lane);
    }
#pragma unroll
    for (int pl = 1; pl < 3; ++pl) {
        const int base = pl == 1 ? PB1 : PB2;
        f32x4 O[1]; float M[1], Ls[1];
        attn_wave<COH, 1>(qrs, head, base + 16 * jj, base + 64 * wave, 64, vt, O, M, Ls, lane);
        // tiles 2 and 3 of the merge scratch
        const int j = lane & 15, g = lane >> 4;
        float* o = mrg + (wave * 4 + 1 + pl) * 288;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(4 * g + r) * 16 + j] = O[0][r];
        if (g == 0) { o[256 + j] = M[0]; o[272 + j] = Ls[0]; }
    }
    __syncthreads();
    // tiles 0, 1: xy; tile 2: yt; tile 3: xt
    {
        const int a = tid >> 6, q = (tid >> 2) & 15, dq = tid & 3;
        if (tid < 256) {
            const int tok = a < 2 ? 32 * jj + 16 * a + q : (a == 2 ? PB1 : PB2) + 16 * jj + q;
            float mx = -INFINITY;
#pragma unroll
            for (int w = 0; w < 8; ++w) mx = fmaxf(mx, mrg[(w * 4 + a) * 288 + 256 + q]);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            float l = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float* p = mrg + (w * 4 + a) * 288;
                const float f = __builtin_amdgcn_exp2f(p[256 + q] - mx);
                l += p[272 + q] * f;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += p[(4 * dq + e) * 16 + q] * f;
            }
            st16<COH>(rsrc(out, L * C * 4), (unsigned)(tok * C + head * 16 + 4 * dq) * 4u, o * (1.0f / l));
        }
    }
}
// cross-plane attention: workgroup = (head, 64 queries), wave = 256 of the 2048 keys
template <bool COH>
__device__ __forceinline__ void attn1d_main(const float* qkv, float* out, int wg, int tid, float* lds) {
    const int head = wg & 7, jj = wg >> 3, wave = tid >> 6, lane = tid & 63;
    float* vt = lds + wave * 320;
    float* mrg = lds + 8 * 320;
    const __amdgpu_buffer_rsrc_t qrs = rsrc(qkv, L * NQKV * 4);
    f32x4 O[4]; float M[4], Ls[4];
    attn_wave<COH, 4>(qrs, head, 64 * jj, 256 * wave, 256, vt, O, M, Ls, lane);
    attn_park<4>(mrg, wave, O, M, Ls, lane);
    __syncthreads();
    attn_merge<COH>(mrg, 4, head, 64 * jj, out, tid);
}

// -------------------------------------------------------------------------------------------------------------------- the two forms
// chain: one kernel per phase
#define CSTAMP(ph, k) do { if (b.dbg && blockIdx.x == 0 && threadIdx.x == 0) b.dbg[32 + 2 * (ph) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
__global__ __launch_bounds__(NTH) void k_p_conv(StageW w, StageBufs b, int which) {
    extern __shared__ float lds[];
    const int wg = blockIdx.x, tid = threadIdx.x;
    CSTAMP(which ? 1 : 0, 0);
    ConvPre pre;
    conv_pre(pre, which ? w.w2 : w.w1, wg, tid);
    if (!which) conv_main<false>(pre, b.x, b.stats + 0 * STAT_COPIES * STAT_DOUBLES, w.gn1_g, w.gn1_b, nullptr, w.bias1, nullptr, b.h1, b.stats + 1 * STAT_COPIES * STAT_DOUBLES, wg, tid, lds, b.dbg ? b.dbg + 64 : nullptr);
    else conv_main<false>(pre, b.h1, b.stats + 1 * STAT_COPIES * STAT_DOUBLES, w.gn2_g, w.gn2_b, w.film, w.bias2, b.x, b.h2, b.stats + 2 * STAT_COPIES * STAT_DOUBLES, wg, tid, lds);
    CSTAMP(which ? 1 : 0, 1);
}
__global__ __launch_bounds__(NTH) void k_p_qkv(StageW w, StageBufs b, int which) {
    extern __shared__ float lds[];
    const int wg = blockIdx.x, tid = threadIdx.x;
    CSTAMP(which ? 5 : 2, 0);
    QkvPre pre;
    qkv_pre(pre, which ? w.wqb : w.wqa, wg, tid);
    if (!which) qkv_main<false>(pre, b.h2, b.stats + 2 * STAT_COPIES * STAT_DOUBLES, false, w.ga_g, w.ga_b, w.bqa, b.qkv, wg, tid, lds);
    else qkv_main<false>(pre, b.h3, b.stats + 3 * STAT_COPIES * STAT_DOUBLES, true, w.gb_g, w.gb_b, w.bqb, b.qkv, wg, tid, lds);
    CSTAMP(which ? 5 : 2, 1);
}
__global__ __launch_bounds__(NTH) void k_p_attn(StageW w, StageBufs b, int which) {
    extern __shared__ float lds[];
    CSTAMP(which ? 6 : 3, 0);
    if (!which) attn2d_main<false>(b.qkv, b.att, blockIdx.x, threadIdx.x, lds);
    else attn1d_main<false>(b.qkv, b.att, blockIdx.x, threadIdx.x, lds);
    CSTAMP(which ? 6 : 3, 1);
}
__global__ __launch_bounds__(NTH) void k_p_proj(StageW w, StageBufs b, int which) {
    extern __shared__ float lds[];
    const int wg = blockIdx.x, tid = threadIdx.x;
    CSTAMP(which ? 7 : 4, 0);
    ProjPre pre;
    proj_pre(pre, which ? w.wpb : w.wpa, wg, tid);
    if (!which) proj_main<false>(pre, b.att, w.bpa, b.h2, b.h3, b.stats + 3 * STAT_COPIES * STAT_DOUBLES, wg, tid, lds);
    else proj_main<false>(pre, b.att, w.bpb, b.h3, b.h4, b.stats_next, wg, tid, lds);
    CSTAMP(which ? 7 : 4, 1);
}

template <class T> __device__ __forceinline__ T load_uniform(const T* p) {       // a POD record through the constant address space: s_load into SGPRs
    static_assert(sizeof(T) % 8 == 0, "8-byte words");
    typedef const __attribute__((address_space(4))) unsigned long long* cq;
    const cq q = (cq)(unsigned long long)p;
    union { T t; unsigned long long u[sizeof(T) / 8]; } x;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 8; ++i) x.u[i] = q[i];
    return x.t;
}
// persistent: `nstage` stages in one launch (descriptors in device memory), 8 edges per stage
struct StageDesc { StageW w; StageBufs b; };
template <int prefetch>
__global__ __launch_bounds__(NTH) void k_stage_persistent(const StageDesc* descs_g, int nstage) {
    extern __shared__ float lds[];
    const int wg = blockIdx.x, tid = threadIdx.x;
    // (descriptors through the constant address space: scalar loads into SGPRs -- as generic loads the 32 pointers of a stage lived in VGPRs
    // and the kernel spilled 298 registers)
#pragma unroll 1
    for (int st = 0; st < nstage; ++st) {
        const StageW w = load_uniform(&descs_g[st].w);
        const StageBufs b = load_uniform(&descs_g[st].b);
        unsigned long long* dbg = b.dbg;
        double* S = b.stats;
        constexpr int SD = STAT_COPIES * STAT_DOUBLES;
        STAMP(0);
        {   // P1 conv1 (its input edge is the previous stage's last edge, or the launch boundary)
            ConvPre pre;
            conv_pre(pre, w.w1, wg, tid);
            conv_main<true>(pre, b.x, S, w.gn1_g, w.gn1_b, nullptr, w.bias1, nullptr, b.h1, S + SD, wg, tid, lds);
        }
        STAMP(1);
        {   // P2 conv2: weights requested before the edge
            ConvPre pre;
            if (prefetch) conv_pre(pre, w.w2, wg, tid);
            grid_edge(b.bar + 0 * 64, wg, tid, dbg ? dbg + 16 + 2 * 0 : nullptr);
            STAMP(2);
            if (!prefetch) conv_pre(pre, w.w2, wg, tid);
            conv_main<true>(pre, b.h1, S + SD, w.gn2_g, w.gn2_b, w.film, w.bias2, b.x, b.h2, S + 2 * SD, wg, tid, lds);
        }
        STAMP(3);
        {   // P3 qkv (per-plane GroupNorm)
            QkvPre pre;
            if (prefetch) qkv_pre(pre, w.wqa, wg, tid);
            grid_edge(b.bar + 1 * 64, wg, tid, dbg ? dbg + 16 + 2 * 1 : nullptr);
            STAMP(4);
            if (!prefetch) qkv_pre(pre, w.wqa, wg, tid);
            qkv_main<true>(pre, b.h2, S + 2 * SD, false, w.ga_g, w.ga_b, w.bqa, b.qkv, wg, tid, lds);
        }
        STAMP(5);
        grid_edge(b.bar + 2 * 64, wg, tid, dbg ? dbg + 16 + 2 * 2 : nullptr);
        STAMP(6);
        attn2d_main<true>(b.qkv, b.att, wg, tid, lds);
        STAMP(7);
        {   // P5 proj + h2 -> h3, statistics over all tokens
            ProjPre pre;
            if (prefetch) proj_pre(pre, w.wpa, wg, tid);
            grid_edge(b.bar + 3 * 64, wg, tid, dbg ? dbg + 16 + 2 * 3 : nullptr);
            STAMP(8);
            if (!prefetch) proj_pre(pre, w.wpa, wg, tid);
            proj_main<true>(pre, b.att, w.bpa, b.h2, b.h3, S + 3 * SD, wg, tid, lds);
        }
        STAMP(9);
        {   // P6 qkv (GroupNorm over all tokens)
            QkvPre pre;
            if (prefetch) qkv_pre(pre, w.wqb, wg, tid);
            grid_edge(b.bar + 4 * 64, wg, tid, dbg ? dbg + 16 + 2 * 4 : nullptr);
            STAMP(10);
            if (!prefetch) qkv_pre(pre, w.wqb, wg, tid);
            qkv_main<true>(pre, b.h3, S + 3 * SD, true, w.gb_g, w.gb_b, w.bqb, b.qkv, wg, tid, lds);
        }
        STAMP(11);
        grid_edge(b.bar + 5 * 64, wg, tid, dbg ? dbg + 16 + 2 * 5 : nullptr);
        STAMP(12);
        attn1d_main<true>(b.qkv, b.att, wg, tid, lds);
        STAMP(13);
        {   // P8 proj + h3 -> h4, statistics for the next stage
            ProjPre pre;
            if (prefetch) proj_pre(pre, w.wpb, wg, tid);
            grid_edge(b.bar + 6 * 64, wg, tid, dbg ? dbg + 16 + 2 * 6 : nullptr);
            STAMP(14);
            if (!prefetch) proj_pre(pre, w.wpb, wg, tid);
            proj_main<true>(pre, b.att, w.bpb, b.h3, b.h4, b.stats_next, wg, tid, lds);
        }
        STAMP(15);
        if (st + 1 < nstage) grid_edge(b.bar + 7 * 64, wg, tid);      // the edge into the next stage's conv1
    }
}

// statistics of the first stage's input (in the product: the epilogue of whatever produced it)
__global__ void k_input_stats(const float* x, double* site) {
    const int tok = blockIdx.x, c = threadIdx.x;                     // 2048 x 128
    const double v = x[tok * C + c];
    double s = v, ss = v * v;
    s += __shfl_xor(s, 1); ss += __shfl_xor(ss, 1);
    s += __shfl_xor(s, 2); ss += __shfl_xor(ss, 2);
    if (!(c & 3)) {
        double* d = site + (((tok & 7) * 3 + plane_of(tok)) * 32 + (c >> 2)) * 2;
        atomicAdd(d, s);
        atomicAdd(d + 1, ss);
    }
}

// ==================================================================================================================== host
static unsigned long long g_rng = 0x9E3779B97F4A7C15ull;
static float urand() {
    g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17;
    return (float)((g_rng >> 11) * (1.0 / 9007199254740992.0)) * 2.0f - 1.0f;
}
struct HostStage {
    std::vector<float> gn1_g, gn1_b, w1, bias1, gn2_g, gn2_b, film, w2, bias2, ga_g, ga_b, wqa, bqa, wpa, bpa, gb_g, gb_b, wqb, bqb, wpb, bpb;
};
static void fill(std::vector<float>& v, size_t n, float scale, float offset = 0.f) {
    v.resize(n);
    for (auto& e : v) e = offset + scale * urand();
}
static HostStage make_stage() {
    HostStage h;
    fill(h.gn1_g, C, 0.2f, 1.f); fill(h.gn1_b, C, 0.2f); fill(h.w1, (size_t)9 * C * C, std::sqrt(3.0f / (9 * C))); fill(h.bias1, C, 0.2f);
    fill(h.gn2_g, C, 0.2f, 1.f); fill(h.gn2_b, C, 0.2f); fill(h.film, 2 * C, 0.3f); fill(h.w2, (size_t)9 * C * C, std::sqrt(3.0f / (9 * C))); fill(h.bias2, C, 0.2f);
    fill(h.ga_g, C, 0.2f, 1.f); fill(h.ga_b, C, 0.2f); fill(h.wqa, (size_t)C * NQKV, std::sqrt(3.0f / C)); fill(h.bqa, NQKV, 0.2f);
    fill(h.wpa, (size_t)C * C, std::sqrt(3.0f / C)); fill(h.bpa, C, 0.2f);
    fill(h.gb_g, C, 0.2f, 1.f); fill(h.gb_b, C, 0.2f); fill(h.wqb, (size_t)C * NQKV, std::sqrt(3.0f / C)); fill(h.bqb, NQKV, 0.2f);
    fill(h.wpb, (size_t)C * C, std::sqrt(3.0f / C)); fill(h.bpb, C, 0.2f);
    return h;
}
template <class T> static T* dnew(size_t n) { T* p = nullptr; CK(hipMalloc((void**)&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
static const float* up(const std::vector<float>& v) { float* p = dnew<float>(v.size()); CK(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return p; }
static StageW upload(const HostStage& h) {
    StageW w;
    w.gn1_g = up(h.gn1_g); w.gn1_b = up(h.gn1_b); w.w1 = up(h.w1); w.bias1 = up(h.bias1);
    w.gn2_g = up(h.gn2_g); w.gn2_b = up(h.gn2_b); w.film = up(h.film); w.w2 = up(h.w2); w.bias2 = up(h.bias2);
    w.ga_g = up(h.ga_g); w.ga_b = up(h.ga_b); w.wqa = up(h.wqa); w.bqa = up(h.bqa); w.wpa = up(h.wpa); w.bpa = up(h.bpa);
    w.gb_g = up(h.gb_g); w.gb_b = up(h.gb_b); w.wqb = up(h.wqb); w.bqb = up(h.bqb); w.wpb = up(h.wpb); w.bpb = up(h.bpb);
    return w;
}

// ---- the stage in double precision, written from the reference's definitions (not from the kernels' decomposition)
typedef std::vector<double> dvec;
static void cpu_gn(const dvec& x, int Cc, bool whole, const std::vector<float>& gamma, const std::vector<float>& beta, dvec& y) {
    y.resize(x.size());
    const int segs[4] = {0, PB1, PB2, L};
    for (int p = 0; p < (whole ? 1 : 3); ++p) {
        const int t0 = whole ? 0 : segs[p], t1 = whole ? L : segs[p + 1];
        for (int g = 0; g < 32; ++g) {
            const int gs = Cc / 32;
            double s = 0, ss = 0;
            for (int t = t0; t < t1; ++t)
                for (int c = g * gs; c < (g + 1) * gs; ++c) { s += x[(size_t)t * Cc + c]; ss += x[(size_t)t * Cc + c] * x[(size_t)t * Cc + c]; }
            const double n = (double)(t1 - t0) * gs, mean = s / n, var = ss / n - mean * mean, rstd = 1.0 / std::sqrt(var + 1e-5);
            for (int t = t0; t < t1; ++t)
                for (int c = g * gs; c < (g + 1) * gs; ++c) y[(size_t)t * Cc + c] = (x[(size_t)t * Cc + c] - mean) * rstd * gamma[c] + beta[c];
        }
    }
}
static void cpu_conv3(const dvec& a, const std::vector<float>& Wr, const std::vector<float>& bias, dvec& out) {   // Wr[tap][c / 4][n][4]
    out.assign((size_t)L * C, 0.0);
    for (int t = 0; t < L; ++t) {
        const int p = t >= PB2 ? 2 : (t >= PB1 ? 1 : 0), base = p == 0 ? 0 : (p == 1 ? PB1 : PB2), hr = p == 0 ? 32 : 16;
        const int y = (t - base) / 32, x = (t - base) % 32;
        double* o = &out[(size_t)t * C];
        for (int n = 0; n < C; ++n) o[n] = bias[n];
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = y + ky - 1, xx = x + kx - 1;
                if (yy < 0 || yy >= hr || xx < 0 || xx >= 32) continue;
                const double* s = &a[(size_t)(base + yy * 32 + xx) * C];
                const float* wt = &Wr[(size_t)(ky * 3 + kx) * 32 * C * 4];
                for (int c = 0; c < C; ++c) {
                    const double v = s[c];
                    const float* wc = wt + ((size_t)(c >> 2) * C) * 4 + (c & 3);
                    for (int n = 0; n < C; ++n) o[n] += v * wc[n * 4];
                }
            }
    }
}
static void cpu_lin(const dvec& a, int K, int N, const std::vector<float>& Wr, const std::vector<float>& bias, dvec& out) {   // Wr[c / 4][n][4]
    out.assign((size_t)L * N, 0.0);
    for (int t = 0; t < L; ++t)
        for (int n = 0; n < N; ++n) {
            double s = bias[n];
            for (int c = 0; c < K; ++c) s += a[(size_t)t * K + c] * Wr[((size_t)(c >> 2) * N + n) * 4 + (c & 3)];
            out[(size_t)t * N + n] = s;
        }
}
static void cpu_attn(const dvec& qkv, bool whole, dvec& out) {
    out.assign((size_t)L * C, 0.0);
    const int segs[4] = {0, PB1, PB2, L};
    const double sc = 1.0 / std::sqrt(std::sqrt((double)DH));
    std::vector<double> w;
    for (int p = 0; p < (whole ? 1 : 3); ++p) {
        const int t0 = whole ? 0 : segs[p], t1 = whole ? L : segs[p + 1];
        w.resize(t1 - t0);
        for (int h = 0; h < H; ++h)
            for (int q = t0; q < t1; ++q) {
                double mx = -1e300;
                for (int k = t0; k < t1; ++k) {
                    double s = 0;
                    for (int d = 0; d < DH; ++d) s += (qkv[(size_t)q * NQKV + h * 48 + d] * sc) * (qkv[(size_t)k * NQKV + h * 48 + 16 + d] * sc);
                    w[k - t0] = s;
                    mx = std::max(mx, s);
                }
                double l = 0;
                for (auto& e : w) { e = std::exp(e - mx); l += e; }
                for (int d = 0; d < DH; ++d) {
                    double o = 0;
                    for (int k = t0; k < t1; ++k) o += w[k - t0] * qkv[(size_t)k * NQKV + h * 48 + 32 + d];
                    out[(size_t)q * C + h * 16 + d] = o / l;
                }
            }
    }
}
static void cpu_stage(const HostStage& h, const dvec& x, dvec& h4) {
    dvec a, h1, h2, qkv, att, pr, h3;
    cpu_gn(x, C, false, h.gn1_g, h.gn1_b, a);
    for (auto& v : a) v = v / (1.0 + std::exp(-v));
    cpu_conv3(a, h.w1, h.bias1, h1);
    cpu_gn(h1, C, false, h.gn2_g, h.gn2_b, a);
    for (int t = 0; t < L; ++t)
        for (int c = 0; c < C; ++c) {
            double v = a[(size_t)t * C + c] * (1.0 + h.film[c]) + h.film[C + c];
            a[(size_t)t * C + c] = v / (1.0 + std::exp(-v));
        }
    cpu_conv3(a, h.w2, h.bias2, h2);
    for (size_t i = 0; i < h2.size(); ++i) h2[i] += x[i];
    cpu_gn(h2, C, false, h.ga_g, h.ga_b, a);
    cpu_lin(a, C, NQKV, h.wqa, h.bqa, qkv);
    cpu_attn(qkv, false, att);
    cpu_lin(att, C, C, h.wpa, h.bpa, pr);
    h3 = h2;
    for (size_t i = 0; i < h3.size(); ++i) h3[i] += pr[i];
    cpu_gn(h3, C, true, h.gb_g, h.gb_b, a);
    cpu_lin(a, C, NQKV, h.wqb, h.bqb, qkv);
    cpu_attn(qkv, true, att);
    cpu_lin(att, C, C, h.wpb, h.bpb, pr);
    h4 = h3;
    for (size_t i = 0; i < h4.size(); ++i) h4[i] += pr[i];
}

struct Run {
    int R;
    std::vector<HostStage> hs;
    std::vector<StageW> ws;
    std::vector<StageBufs> bufs;
    std::vector<float*> act;          // R + 1 activations (x0, h4 of stage 0, ...)
    double* stats;                    // [R + 1][NSITE][copies][..]
    unsigned long long* bar;          // [R][NEDGE][64]
    unsigned long long* dbg;
    StageDesc* descs;
    size_t stats_bytes, bar_bytes;
};
static Run make_run(int R, const std::vector<float>& x0) {
    Run r;
    r.R = R;
    constexpr size_t SD = (size_t)STAT_COPIES * STAT_DOUBLES;
    r.stats_bytes = (size_t)(R + 1) * NSITE * SD * 8;
    r.stats = dnew<double>((size_t)(R + 1) * NSITE * SD);
    r.bar_bytes = (size_t)R * NEDGE * 64 * 8;
    r.bar = dnew<unsigned long long>((size_t)R * NEDGE * 64);
    r.dbg = dnew<unsigned long long>(128);
    r.act.resize(R + 1);
    for (auto& p : r.act) p = dnew<float>((size_t)L * C);
    CK(hipMemcpy(r.act[0], x0.data(), x0.size() * 4, hipMemcpyHostToDevice));
    float* h1 = dnew<float>((size_t)L * C); float* h2 = dnew<float>((size_t)L * C); float* h3 = dnew<float>((size_t)L * C);
    float* qkv = dnew<float>((size_t)L * NQKV); float* att = dnew<float>((size_t)L * C);
    std::vector<StageDesc> hd(R);
    for (int s = 0; s < R; ++s) {
        r.hs.push_back(make_stage());
        r.ws.push_back(upload(r.hs.back()));
        StageBufs b;
        b.x = r.act[s]; b.h1 = h1; b.h2 = h2; b.qkv = qkv; b.att = att; b.h3 = h3; b.h4 = r.act[s + 1];
        b.stats = r.stats + (size_t)s * NSITE * SD;
        b.stats_next = r.stats + (size_t)(s + 1) * NSITE * SD;
        b.bar = r.bar + (size_t)s * NEDGE * 64;
        b.dbg = s == 0 ? r.dbg : nullptr;
        r.bufs.push_back(b);
        hd[s].w = r.ws.back();
        hd[s].b = b;
    }
    CK(hipMalloc((void**)&r.descs, sizeof(StageDesc) * R));
    CK(hipMemcpy(r.descs, hd.data(), sizeof(StageDesc) * R, hipMemcpyHostToDevice));
    return r;
}
static const size_t SMEM = (size_t)LDS_FLOATS * 4;
static void reset(const Run& r, hipStream_t s) {
    CK(hipMemsetAsync(r.stats, 0, r.stats_bytes, s));
    CK(hipMemsetAsync(r.bar, 0, r.bar_bytes, s));
    hipLaunchKernelGGL(k_input_stats, dim3(L), dim3(C), 0, s, r.act[0], r.stats);
}
static void launch_chain(const Run& r, hipStream_t s) {
    for (int st = 0; st < r.R; ++st) {
        const StageW& w = r.ws[st]; const StageBufs& b = r.bufs[st];
        hipLaunchKernelGGL(k_p_conv, dim3(NWG), dim3(NTH), SMEM, s, w, b, 0);
        hipLaunchKernelGGL(k_p_conv, dim3(NWG), dim3(NTH), SMEM, s, w, b, 1);
        hipLaunchKernelGGL(k_p_qkv, dim3(NWG), dim3(NTH), SMEM, s, w, b, 0);
        hipLaunchKernelGGL(k_p_attn, dim3(NWG), dim3(NTH), SMEM, s, w, b, 0);
        hipLaunchKernelGGL(k_p_proj, dim3(NWG), dim3(NTH), SMEM, s, w, b, 0);
        hipLaunchKernelGGL(k_p_qkv, dim3(NWG), dim3(NTH), SMEM, s, w, b, 1);
        hipLaunchKernelGGL(k_p_attn, dim3(NWG), dim3(NTH), SMEM, s, w, b, 1);
        hipLaunchKernelGGL(k_p_proj, dim3(NWG), dim3(NTH), SMEM, s, w, b, 1);
    }
}
static void launch_persistent(const Run& r, hipStream_t s, bool all_in_one, int prefetch) {
    auto go = [&](const StageDesc* d, int n) {
        if (prefetch) hipLaunchKernelGGL(k_stage_persistent<1>, dim3(NWG), dim3(NTH), SMEM, s, d, n);
        else hipLaunchKernelGGL(k_stage_persistent<0>, dim3(NWG), dim3(NTH), SMEM, s, d, n);
    };
    if (all_in_one) go(r.descs, r.R);
    else for (int st = 0; st < r.R; ++st) go(r.descs + st, 1);
}
static double compare(const Run& r, int stage, const dvec& ref, const char* what) {
    std::vector<float> got((size_t)L * C);
    CK(hipMemcpy(got.data(), r.act[stage + 1], got.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    bool nan = false;
    for (size_t i = 0; i < got.size(); ++i) {
        if (!(got[i] == got[i])) nan = true;
        worst = std::max(worst, std::fabs((double)got[i] - ref[i]));
        scale = std::max(scale, std::fabs(ref[i]));
    }
    printf("%-44s stage %d output: max|err| %.3e  (|ref| <= %.2f)%s\n", what, stage, worst, scale, nan ? "  NaN" : "");
    return nan ? 1e9 : worst / std::max(1.0, scale);
}

int main(int argc, char** argv) {
    const bool timing = argc >= 2 && !strcmp(argv[1], "time");
    const int R = timing ? (argc >= 3 ? atoi(argv[2]) : 12) : 3;
    for (const void* f : {(const void*)k_p_conv, (const void*)k_p_qkv, (const void*)k_p_attn, (const void*)k_p_proj, (const void*)k_stage_persistent<0>, (const void*)k_stage_persistent<1>})
        CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
    std::vector<float> x0((size_t)L * C);
    for (auto& v : x0) v = urand();
    Run r = make_run(R, x0);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    int bad = 0;
    if (!timing) {
        std::vector<dvec> ref(R + 1);
        ref[0].assign(x0.begin(), x0.end());
        for (int st = 0; st < R; ++st) cpu_stage(r.hs[st], ref[st], ref[st + 1]);
        struct { const char* name; int form, aio, pf; } forms[] = {{"chain (8 launches per stage)", 0, 0, 0}, {"persistent, one launch per stage", 1, 0, 1},
                                                                    {"persistent, all stages in one launch", 1, 1, 1}, {"persistent, no weight prefetch", 1, 1, 0}};
        for (auto& f : forms)
            for (int rep = 0; rep < 2; ++rep) {          // (twice: the second run starts from the state the first left)
                for (int st = 1; st <= R; ++st) CK(hipMemsetAsync(r.act[st], 0xFF, (size_t)L * C * 4, s));
                reset(r, s);
                if (f.form == 0) launch_chain(r, s); else launch_persistent(r, s, f.aio, f.pf);
                CK(hipStreamSynchronize(s));
                CK(hipGetLastError());
                if (compare(r, 0, ref[1], f.name) > 2e-4) ++bad;
                if (compare(r, R - 1, ref[R], f.name) > 1e-3) ++bad;
            }
        printf(bad ? "STAGE CHECK FAILED (%d)\n" : "STAGE CHECK OK\n", bad);
        return bad ? 1 : 0;
    }
    // ---- timing: R stages back to back (distinct weights), one hipGraph per form, 30 replays
    struct Form { const char* name; int form, aio, pf; } forms[] = {{"chain: 8 launches per stage", 0, 0, 0}, {"persistent: one launch per stage", 1, 0, 1},
                                                                    {"persistent: all stages, one launch", 1, 1, 1}, {"persistent: all stages, no weight prefetch", 1, 1, 0}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass)
        for (auto& f : forms) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            if (f.form == 0) launch_chain(r, s); else launch_persistent(r, s, f.aio, f.pf);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            float best = 1e9f, sum = 0.f;
            const int reps = 30;
            for (int i = 0; i < reps + 3; ++i) {
                reset(r, s);
                CK(hipEventRecord(e0, s));
                CK(hipGraphLaunch(ge, s));
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (i >= 3) { best = std::min(best, ms); sum += ms; }
            }
            printf("%-46s %7.2f us per stage (mean of %d replays of %d stages; best %.2f)\n", f.name, 1e3 * sum / reps / R, reps, R, 1e3 * best / R);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    // phase stamps of workgroup 0, stage 0 of the last persistent run (s_memtime = shader clock)
    unsigned long long st[64];
    CK(hipMemcpy(st, r.dbg, sizeof st, hipMemcpyDeviceToHost));
    const char* nm[16] = {"start", "conv1 done", "edge0 passed", "conv2 done", "edge1 passed", "qkv done", "edge2 passed", "attn2d done", "edge3 passed", "proj done",
                          "edge4 passed", "qkv1d done", "edge5 passed", "attn1d done", "edge6 passed", "proj1d done"};
    printf("stamps (workgroup 0, stage 0, cycles since start):\n");
    for (int k = 1; k < 16; ++k) printf("  %-14s %8llu  (+%llu)\n", nm[k], st[k] - st[0], st[k] - st[k - 1]);
    printf("edges (workgroup 0): phase done -> own wave drained -> workgroup drained -> all 256 arrived and seen\n");
    for (int e = 0; e < 7; ++e) printf("  edge %d: +%llu  +%llu  +%llu\n", e, st[16 + 2 * e] - st[2 * e + 1], st[17 + 2 * e] - st[16 + 2 * e], st[2 * e + 2] - st[17 + 2 * e]);
    const char* pn[8] = {"conv1", "conv2", "qkv", "attn2d", "proj", "qkv1d", "attn1d", "proj1d"};
    printf("chain form, workgroup 0 of each launch, cycles from entry to exit (stage 0 of the last chain replay):\n");
    for (int ph = 0; ph < 8; ++ph) printf("  %-8s %8llu   (persistent body: %llu)\n", pn[ph], st[33 + 2 * ph] - st[32 + 2 * ph], st[2 * ph + 1] - st[2 * ph]);
    {
        unsigned long long cs[8];
        CK(hipMemcpy(cs, r.dbg + 64, sizeof cs, hipMemcpyDeviceToHost));
        printf("conv1 of the chain form, wave 0 of workgroup 0 (cycles): requests issued %llu | statistics in + barrier %llu | window transformed + parked %llu | "
               "144 MFMAs per wave, 2 waves per SIMD (ideal 9216) %llu | partials stored %llu | reduction + epilogue + statistics %llu\n",
               cs[1] - cs[0], cs[2] - cs[1], cs[3] - cs[2], cs[4] - cs[3], cs[5] - cs[4], cs[6] - cs[5]);
    }
    return 0;
}
