// One attention block of the DEEP levels (<= 128 tokens per clip) in ONE launch, for gfx950:
//     GroupNorm -> qkv 1x1 conv -> QKVAttentionLegacy -> proj_out 1x1 conv + residual
// (MToV/models/ddpm/unet.py:210-300 AttentionBlock / AttentionBlock1D, :303-326 QKVAttentionLegacy; GroupNorm32
// diffusionmodules.py:156-173), replacing the three launches k_deep_finalize (slabs -> plain input + statistics), k_conv
// (qkv) and k_deep_attn (core + proj_out) of deep.hip, round 4: 18 x 3 launches of 4.6 + 7.7-11.4 + 8.8-15.3 us per DDIM step.
//
// Every block boundary inside the block is an all-to-all dependency (the qkv conv reads all channels, a head reads all tokens,
// proj_out reads all heads), which is why round 4 paid one kernel boundary + one cold start per dependency.  Here the block is
// cut along HEADS instead: a CLUSTER of CL workgroups owns one (clip, head) and nothing but the block's input and output ever
// crosses clusters --
//   stage 1  workgroup j of the cluster takes K slice j (CS = C / CL channels, whole GroupNorm groups) of the head's qkv GEMM:
//            stages ALL tokens of its channel slice (adding the input's K-slice slabs in slab order, as every consumer of the
//            deep levels does), computes the GroupNorm statistics of its groups itself (per plane, or over all planes for
//            AttentionBlock1D), normalises once, multiplies by W_qkv[head's 3d rows][slice] -> partial [L x 3d]   -> scratch
//   hand-off 1 (data-tagged granules, see below)
//   stage 2  workgroup j adds the CL partials of ITS rows (slice order: run-to-run bit-equal) + bias -> the head's q | k | v rows
//   hand-off 2
//   stage 3  workgroup j = (query tile, column part): K rows, V^T and its 16 queries to LDS, S^T = K Q^T per key tile (a query is
//            a lane column, as k_attention / k_deep_attn), softmax, O^T += V^T P^T, key tiles merged through LDS; then
//            [16 x cols] = attention rows x W_proj[cols][head's d columns]; the HEAD is the K slice of the projection: the
//            partial result goes to output slab `head` (slab h also carries input slab h as the residual, slab 0 the bias).
// The output is a deep tensor of H slabs: its consumers add them up (deep.hip).
//
// Hand-offs inside the launch (MI355X_MICROARCH.md, hand-off price list: "8-byte {data, tag} granules", no drain, no flag, no ordering): every
// workgroup takes ONE entry ticket on its cluster's 64-bit monotonic counter at kernel entry (relaxed agent atomic add; never reset: no ABA) -- ticket n
// belongs to launch n / CL, the EPOCH, which tags every granule the launch writes; its round trip hides under the first loads.  Writers use 16-byte
// sc0 sc1 (write-through) stores of two {value, epoch} granules and never wait; readers poll the data itself with sc0 sc1 (L1-bypassing) loads until
// every tag is the epoch.  Correct under any workgroup -> XCD placement.  Scratch and counters belong to ONE op (two ops count their epochs
// separately: in a shared buffer op B would accept what op A wrote at the same count).  All workgroups of the launch must be resident together
// (a poll waits for workgroups of the same launch): the grid is B x H x CL <= 128 workgroups of 512 threads, one per CU on half the chip.
// A poll that exceeds 2^21 retries (seconds) raises *fault (host-mapped) and falls through -- garbage, never a hang; the host refuses every later call on
// the context (plan.hip check_ready, mtv_last_error: "an in-launch hand-off ... timed out").
#include <cstdio>
#include <cstdlib>

#include "mtv_internal.h"

namespace mtv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BLK_NTH = 512;
constexpr int BLK_MAX_NG = 32;        // GroupNorm groups per channel slice
constexpr int BLK_MAX_RETRY = 1 << 21;                        // granule polls per thread (x s_sleep 1 + a fabric round trip: seconds -- a bound against a hang, never a pace)
constexpr int BLK_SC = 17;                                    // buffer cache policy sc0 | sc1: write-through stores, L1 / L2-bypassing loads
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef __attribute__((address_space(1))) unsigned long long blk_gu64;

__device__ __forceinline__ int blk_usgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- data-tagged granules (MI355X_MICROARCH.md, hand-off price list: 8-byte {data, tag}, no drain, no flag): a float travels with
// the launch's epoch; the reader re-reads a granule until its tag is this launch's.  Two granules per 16-byte store / load.
__device__ __forceinline__ void blk_put_granules(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, const f32x4& v, unsigned tag) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), tag, __float_as_uint(v[1]), tag}, rs, byte_off, 0, BLK_SC);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[2]), tag, __float_as_uint(v[3]), tag}, rs, byte_off + 16, 0, BLK_SC);
}
__device__ __forceinline__ bool blk_get_granules(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, unsigned tag, f32x4* out) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, BLK_SC);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off + 16, 0, BLK_SC);
    *out = f32x4{__uint_as_float(a[0]), __uint_as_float(a[2]), __uint_as_float(b[0]), __uint_as_float(b[2])};
    return a[1] == tag && a[3] == tag && b[1] == tag && b[3] == tag;
}

__device__ __forceinline__ float blk_swap_max16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float blk_swap_max32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

}  // namespace

#ifdef MTV_DEEP_STAMP   // s_memtime of lane 0 of wave 0 of workgroups 0 and gridDim / 2: [wg][16 slots]
#define BLK_STAMP(k) do { if (a.dbg && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) \
                               a.dbg[(blockIdx.x ? 16 : 0) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BLK_STAMP(k) do { } while (0)
#endif

// LDS plan (floats), shared by host sizing and the kernel
struct BlkLds {
    int LP;                 // token rows padded to a multiple of 16
    int XS, WS;             // row strides of the staged input slice / qkv weight slice
    int xs, wt, stat, img;  // stage 1: input slice [rows of a row group][XS] | weights [3D][WS] | statistics (doubles)  (img: unused since the partials leave from registers)
    int ks, vt, qs, att, os, ml, po;   // stage 3: K rows [kcap][D + 4] | V^T [D][kcap + 4] | Q [16][D + 4] | attention rows [16][D + 8] |
                                       // key-tile partials [8][16][D + 4] | (m, l) [8][16][2] | proj image [16][ncols + 4]
    int epoch;
    int total;
};
__host__ __device__ inline BlkLds blk_lds(int L, int D, int CS, int ncols, int RQ) {
    BlkLds p;
    p.LP = (L + 15) / 16 * 16;
    p.XS = CS + 8;
    p.WS = CS + 4;
    const int NQ = 3 * D;
    p.xs = 0;
    p.wt = p.xs + (p.LP / RQ) * p.XS;
    int s1 = p.wt + NQ * p.WS;
    s1 = (s1 + 3) & ~3;
    p.stat = s1;
    s1 += 3 * BLK_MAX_NG * 2 * 2;
    p.img = 0;
    const int st1 = s1;
    const int kcap = p.LP;
    p.ks = 0;
    p.vt = p.ks + kcap * (D + 4);
    p.qs = p.vt + D * (kcap + 4);
    p.att = p.qs + 16 * (D + 4);
    p.os = p.att + 16 * (D + 8);
    p.ml = p.os + 8 * 16 * (D + 4);
    p.po = p.ml + 8 * 16 * 2;
    const int st3 = p.po + 16 * (ncols + 4);
    p.total = ((st1 > st3 ? st1 : st3) + 3) & ~3;
    p.epoch = p.total;                                  // the launch's epoch (tag of the granules), written by thread 0
    p.total += 4;
    return p;
}

template <int D>
__global__ __launch_bounds__(BLK_NTH) void k_deep_block(const DeepBlockArgs a) {
    const int tid = threadIdx.x;
    BLK_STAMP(0);
    touch_kernargs<(int)sizeof(DeepBlockArgs)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int NQ = 3 * D, NDT = D / 16, QPR = D / 4, KSTR = D + 4, AS = D + 8;
    constexpr int PT = 2;                                   // proj column tiles per wave (ncols <= 256)
    const int lane = tid & 63;
    const int wave = blk_usgpr(tid >> 6);
    const int jl = lane & 15, g = lane >> 4;
    const int blk = blk_usgpr((int)blockIdx.x);
    const int CL = a.CL, CS = a.CS, L = a.L, C = a.C, H = a.H;
    const int j = blk & (CL - 1);                           // workgroup of the cluster: stage 1 (K slice ks, row group rq), stage 2 row share, stage-3 item
    const int ks = j & (a.KSN - 1), rq = j >> a.ksn_shift;
    const int RPQ = a.RPQ, row_base = rq * RPQ;             // this workgroup's rows in stage 1
    const int bh = blk >> a.cl_shift;
    const int b = blk_usgpr(bh / H), h = blk_usgpr(bh - (bh / H) * H);
    const int b1 = a.r * a.r, b2 = b1 + a.t * a.r;
    const BlkLds lp = blk_lds(L, D, CS, a.ncols, a.RQ);
    const int LP = lp.LP, XS = lp.XS, WS = lp.WS;
    float* const xs = smem + lp.xs;
    float* const wt = smem + lp.wt;
    double* const sdp = reinterpret_cast<double*>(smem + lp.stat);       // [3 planes][BLK_MAX_NG][2]
    const int cs0 = ks * CS;
    const int qw_shift = a.qw_shift, QW = 1 << qw_shift;                 // quads per row of the slice
    const int qd = tid & (QW - 1), rl = tid >> qw_shift, RP = BLK_NTH >> qw_shift;
    unsigned long long* const cnt = a.cnt + (size_t)bh * 2;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, (int)a.part_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(a.qkv, 0, (int)a.qkv_bytes, 0x00020000);
    // the launch's epoch: every workgroup of a cluster takes one ticket of counter 1 at entry (monotonic, exactly CL per launch):
    // ticket n belongs to launch n / CL.  Requested now, used after stage 1 -- its round trip hides under the loads.
    unsigned long long early = 0;
    if (tid == 0) early = __hip_atomic_fetch_add(cnt + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // =========================================================================================================== stage 1
    // request order: input slice (first pass; L2 / Infinity-Cache warm) -> qkv weight slice -> proj fragments (HBM-cold, needed
    // last): a wave's loads return in order
    const int xks = a.x.ks;
    const float* const xcol = a.x.p + (size_t)b * L * a.x.C + cs0 + 4 * qd;
    f32x4 xv[8];
    // (UNCONDITIONAL loads into their final registers inside a wave-uniform branch: a per-lane conditional load ends in a register copy
    //  at the join, and a copy of a loaded value waits for every older load -- deep.hip found that three times)
    if (blk_usgpr(rl) < RPQ && row_base + blk_usgpr(rl) < L) {      // this wave's first row (rows ascend with the lane): any row to stage?
        const int row = (rl < RPQ && row_base + rl < L) ? row_base + rl : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = *reinterpret_cast<const f32x4*>(xcol + (size_t)row * a.x.C + (size_t)(k < xks ? k : 0) * a.x.slab_stride);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int WU = (NQ * 32 + BLK_NTH - 1) / BLK_NTH;                // weight quads per thread at CS = 128
    f32x4 wreg[WU];
    {
        const float* wb = a.Wq + (size_t)(h * NQ) * C + cs0;
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int e = tid + BLK_NTH * u;
            const int n = e >> qw_shift, wq = e & (QW - 1);
            // (whole-workgroup uniform bound: a request costs issue time on the CU's memory path whether its data is wanted or not)
            if (BLK_NTH * u < (NQ << qw_shift)) wreg[u] = *reinterpret_cast<const f32x4*>(wb + (size_t)(n < NQ ? n : 0) * C + 4 * wq);
            else wreg[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(a.gamma + cs0 + 4 * qd);
    const f32x4 be = *reinterpret_cast<const f32x4*>(a.beta + cs0 + 4 * qd);
    // proj_out fragments of this workgroup's stage-3 item (when it has exactly one: the usual case): B operand straight from the
    // checkpoint layout [n][k]: lane (jl, g) holds Wp[n0 + 16 ct + jl][h D + 16 cc + 4 g .. + 3]
    const int nitems = a.nqt * a.ncp;
    const int nct2 = a.ncols >> 4;
    f32x4 wp[PT][NDT];
    auto load_wp = [&](int cp) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int ct = wave + 8 * pt;                                  // (wave-uniform: no request for a tile this wave does not have)
            const int n = cp * a.ncols + 16 * (ct < nct2 ? ct : 0) + jl;
            if (ct < nct2) {
#pragma unroll
                for (int cc = 0; cc < NDT; ++cc) wp[pt][cc] = *reinterpret_cast<const f32x4*>(a.Wp + (size_t)n * C + h * D + 16 * cc + 4 * g);
            } else {
#pragma unroll
                for (int cc = 0; cc < NDT; ++cc) wp[pt][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    if (j < nitems) load_wp(j / a.nqt);
    BLK_STAMP(1);
    for (int e = tid; e < 3 * BLK_MAX_NG * 2; e += BLK_NTH) sdp[e] = 0.0;
    __syncthreads();
    // input slice -> LDS (slab order), statistics per (plane, group) of the slice
    const int gs = a.gs;
    {
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        const bool wide = (gs & 3) == 0;                                  // a quad lies inside one group
        auto consume = [&](int lrow, const f32x4& x) {            // lrow: row within the row group; token = row_base + lrow
            *reinterpret_cast<f32x4*>(xs + lrow * XS + 4 * qd) = x;
            const int row = row_base + lrow;
            if (row >= L) return;
            const int p = row >= b2 ? 2 : (row >= b1 ? 1 : 0);
            if (wide) {
                const double s = ((double)x[0] + (double)x[1]) + ((double)x[2] + (double)x[3]);
                const double ss = ((double)x[0] * x[0] + (double)x[1] * x[1]) + ((double)x[2] * x[2] + (double)x[3] * x[3]);
                acc[0] += p == 0 ? s : 0.0; acc[1] += p == 0 ? ss : 0.0;
                acc[2] += p == 1 ? s : 0.0; acc[3] += p == 1 ? ss : 0.0;
                acc[4] += p == 2 ? s : 0.0; acc[5] += p == 2 ? ss : 0.0;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {                              // (narrow test models only: groups of 1 or 2 channels)
                    const int gi = (4 * qd + k) / gs;
                    atomicAdd(&sdp[(p * BLK_MAX_NG + gi) * 2], (double)x[k]);
                    atomicAdd(&sdp[(p * BLK_MAX_NG + gi) * 2 + 1], (double)x[k] * x[k]);
                }
            }
        };
        {
            f32x4 x = xv[0];
#pragma unroll
            for (int k = 1; k < 8; ++k)
                if (k < xks) x += xv[k];                                   // slab order
            if (rl < RPQ) consume(rl, row_base + rl < L ? x : f32x4{0.f, 0.f, 0.f, 0.f});
        }
        for (int row = rl + RP; row < RPQ; row += RP) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (row_base + row < L) {
                f32x4 t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(xcol + (size_t)(row_base + row) * a.x.C + (size_t)(k < xks ? k : 0) * a.x.slab_stride);
                x = t[0];
#pragma unroll
                for (int k = 1; k < 8; ++k)
                    if (k < xks) x += t[k];
            }
            consume(row, x);
        }
        if (wide) {
            const int gi = (4 * qd) / gs;
#pragma unroll
            for (int p = 0; p < 3; ++p)
                if (acc[2 * p + 1] != 0.0) {
                    atomicAdd(&sdp[(p * BLK_MAX_NG + gi) * 2], acc[2 * p]);
                    atomicAdd(&sdp[(p * BLK_MAX_NG + gi) * 2 + 1], acc[2 * p + 1]);
                }
        }
    }
    // qkv weight slice -> LDS [n][k] (a straight copy of the checkpoint's rows)
#pragma unroll
    for (int u = 0; u < WU; ++u) {
        const int e = tid + BLK_NTH * u;
        const int n = e >> qw_shift, wq = e & (QW - 1);
        if (n < NQ) *reinterpret_cast<f32x4*>(wt + n * WS + 4 * wq) = wreg[u];
    }
    if (tid == 0) *reinterpret_cast<unsigned*>(smem + lp.epoch) = (unsigned)(early >> a.cl_shift) + 1u;     // (the ticket returned long ago)
    BLK_STAMP(2);
    __syncthreads();
    const unsigned epoch = *reinterpret_cast<const unsigned*>(smem + lp.epoch);
    if (a.RQ > 1) {
        // The statistics above cover this row group's rows only: the RQ row groups of a K slice exchange their partial (sum, sum of squares)
        // -- 3 planes x <= 32 groups x 2 doubles, one 16-byte {lo, tag, hi, tag} granule pair each -- and every one adds them in row-group
        // order (all get the same bits).  The wait hides under the weight slice still on its way from HBM.
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.stg, 0, (int)a.stg_bytes, 0x00020000);
        constexpr int NST = 3 * BLK_MAX_NG * 2;
        const unsigned sbase = (unsigned)(((size_t)bh * a.KSN + ks) * a.RQ) * NST * 16u;
        if (tid < NST) {
            const unsigned long long u = __builtin_bit_cast(unsigned long long, sdp[tid]);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{(unsigned)u, epoch, (unsigned)(u >> 32), epoch}, srs, sbase + (unsigned)(rq * NST + tid) * 16u, 0, BLK_SC);
            double tot = 0.0;
            for (int q2 = 0; q2 < a.RQ; ++q2) {
                double v = sdp[tid];
                if (q2 != rq) {
                    u32x4 g4;
                    int tries = 0;
                    for (;;) {
                        g4 = __builtin_amdgcn_raw_buffer_load_b128(srs, sbase + (unsigned)(q2 * NST + tid) * 16u, 0, BLK_SC);
                        if (g4[1] == epoch && g4[3] == epoch) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++tries > BLK_MAX_RETRY) { __hip_atomic_store(a.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                    }
                    v = __builtin_bit_cast(double, ((unsigned long long)g4[2] << 32) | g4[0]);
                }
                tot += v;                                                  // row-group order
            }
            sdp[tid] = tot;
        }
        __syncthreads();
    }
    // normalise in place: y = (x - mean) rstd gamma + beta, statistics per plane or over all planes (whole)
    {
        int cur_p = -1;
        f32x4 A = {0.f, 0.f, 0.f, 0.f}, Bc = A;
        for (int lrow = rl; lrow < RPQ && row_base + lrow < L; lrow += RP) {
            const int row = row_base + lrow;
            const int p = row >= b2 ? 2 : (row >= b1 ? 1 : 0);
            if (p != cur_p) {
                cur_p = p;
                float mu_w = 0.f, rstd_w = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k > 0 && (gs & 3) == 0) {                          // the four channels of a quad share their group: one (mean, rstd)
                        const float sc = rstd_w * ga[k];
                        A[k] = sc;
                        Bc[k] = be[k] - sc * mu_w;
                        continue;
                    }
                    const int gi = (4 * qd + k) / gs;
                    double sx, sy;
                    if (a.whole) {
                        sx = (sdp[(0 * BLK_MAX_NG + gi) * 2] + sdp[(1 * BLK_MAX_NG + gi) * 2]) + sdp[(2 * BLK_MAX_NG + gi) * 2];
                        sy = (sdp[(0 * BLK_MAX_NG + gi) * 2 + 1] + sdp[(1 * BLK_MAX_NG + gi) * 2 + 1]) + sdp[(2 * BLK_MAX_NG + gi) * 2 + 1];
                    } else {
                        sx = sdp[(p * BLK_MAX_NG + gi) * 2];
                        sy = sdp[(p * BLK_MAX_NG + gi) * 2 + 1];
                    }
                    const double inv_n = a.whole ? a.inv_n[3] : a.inv_n[p];
                    const double mean = sx * inv_n;
                    double var = sy * inv_n - mean * mean;
                    var = var < 0.0 ? 0.0 : var;
                    const float mu = (float)mean, rstd = 1.0f / sqrtf((float)var + 1e-5f);
                    mu_w = mu; rstd_w = rstd;
                    const float sc = rstd * ga[k];
                    A[k] = sc;
                    Bc[k] = be[k] - sc * mu;
                }
            }
            f32x4* cell = reinterpret_cast<f32x4*>(xs + lrow * XS + 4 * qd);
            f32x4 v = *cell;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaf(v[k], A[k], Bc[k]);
            *cell = v;
        }
    }
    __syncthreads();
    BLK_STAMP(3);
    // partial qkv of this K slice: [LP x NQ] = xs [LP x CS] . wt^T; tiles (row tile, column tile) dealt round-robin to the waves
    const int nrt = RPQ >> 4;
    constexpr int NCT = NQ / 16;
    constexpr int MAXT = (8 * NCT + 7) / 8;                              // tiles per wave at 128 tokens
    f32x4 tacc[MAXT];
    {
        const int ntile = nrt * NCT, nchunk = CS >> 4;
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            tacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int tl = wave + 8 * u;
            if (tl < ntile) {
                const int rt = tl / NCT, ct = tl - rt * NCT;
                for (int cc = 0; cc < nchunk; ++cc) {
                    const f32x4 af = *reinterpret_cast<const f32x4*>(xs + (16 * rt + jl) * XS + 16 * cc + 4 * g);
                    const f32x4 bf = *reinterpret_cast<const f32x4*>(wt + (16 * ct + jl) * WS + 16 * cc + 4 * g);
#pragma unroll
                    for (int sI = 0; sI < 4; ++sI) tacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[sI], bf[sI], tacc[u], 0, 0, 0);
                }
            }
        }
        // the partial tiles leave as data-tagged granules straight from the accumulators -- no LDS image, no drain, no flag: stage 2 polls
        // the data.  Layout [cluster][slice][row pair][col][2 rows]: a lane (col jl, group g) holds rows 4 g .. 4 g + 3 of its column = two
        // 16-byte stores, 16 lanes = 256 contiguous bytes
        const unsigned pslice = (unsigned)((((size_t)bh * a.KSN + ks) * (LP >> 1) + (row_base >> 1)) * NQ * 16);
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            const int tl = wave + 8 * u;
            if (tl < ntile) {
                const int rt = tl / NCT, ct = tl - rt * NCT;
                const unsigned o = pslice + (unsigned)((8 * rt + 2 * g) * NQ + 16 * ct + jl) * 16u;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(tacc[u][0]), epoch, __float_as_uint(tacc[u][1]), epoch}, prs, o, 0, BLK_SC);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(tacc[u][2]), epoch, __float_as_uint(tacc[u][3]), epoch}, prs, o + (unsigned)NQ * 16u, 0, BLK_SC);
            }
        }
    }
    BLK_STAMP(4);
    BLK_STAMP(5);
    // =========================================================================================================== stage 2
    // this workgroup's row pairs of the head's q | k | v: the CL partials in slice order + bias -> granules tagged with the launch's epoch.
    // thread -> (row pair, column): one 16-byte load per slice (2 rows of its column), polled until every slice's tag is this launch's
    {
        const int pp = a.rows_per >> 1, pr0 = j * pp;
        const unsigned pb = (unsigned)((size_t)bh * a.KSN * (LP >> 1) * NQ * 16);
        const int KSN = a.KSN;
        const unsigned gb = (unsigned)((size_t)bh * L * NQ * 8);
        for (int e = tid; e < pp * NQ; e += BLK_NTH) {
            const int prl = e / NQ, col = e - prl * NQ;
            const int pr = pr0 + prl;
            if (2 * pr >= L) continue;
            const unsigned o = pb + (unsigned)(pr * NQ + col) * 16u;
            const unsigned sstride = (unsigned)((LP >> 1) * NQ) * 16u;
            u32x4 t[16];
            int tries = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 16; ++k) t[k] = __builtin_amdgcn_raw_buffer_load_b128(prs, o + (unsigned)(k < KSN ? k : 0) * sstride, 0, BLK_SC);
#pragma unroll
                for (int k = 0; k < 16; ++k) ok = ok && t[k][1] == epoch && t[k][3] == epoch;
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if (++tries > BLK_MAX_RETRY) { __hip_atomic_store(a.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
            float v0 = __uint_as_float(t[0][0]), v1 = __uint_as_float(t[0][2]);
#pragma unroll
            for (int k = 1; k < 16; ++k)
                if (k < KSN) { v0 += __uint_as_float(t[k][0]); v1 += __uint_as_float(t[k][2]); }     // slice order
            const float bias = a.bq[h * NQ + col];
            blk_gu64* dst = (blk_gu64*)(unsigned long long)(reinterpret_cast<char*>(a.qkv) + gb + (size_t)((2 * pr) * NQ + col) * 8);
            __hip_atomic_store(dst, ((unsigned long long)epoch << 32) | __float_as_uint(v0 + bias), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (2 * pr + 1 < L)
                __hip_atomic_store(dst + NQ, ((unsigned long long)epoch << 32) | __float_as_uint(v1 + bias), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    BLK_STAMP(6);
    __syncthreads();                                                      // (stage 3 reuses the LDS the GEMM of stage 1 read from)
    BLK_STAMP(7);
    // =========================================================================================================== stage 3
    float* const Ks = smem + lp.ks;
    float* const Vt = smem + lp.vt;
    float* const Qs = smem + lp.qs;
    float* const Att = smem + lp.att;
    float* const Os = smem + lp.os;
    float* const ml = smem + lp.ml;
    float* const po = smem + lp.po;
    const int VSTR = LP + 4, POS = a.ncols + 4;
    const unsigned gbh = (unsigned)((size_t)bh * L * NQ * 8);           // byte offset of this head's granules
    // one quad (4 granules) of the head's q | k | v rows, polled until this launch's stage 2 has written it
    auto fetch = [&](int row, int col) -> f32x4 {
        f32x4 v;
        const unsigned off = gbh + (unsigned)(row * NQ + col) * 8u;
        int tries = 0;
        while (!blk_get_granules(qrs, off, epoch, &v)) {
            __builtin_amdgcn_s_sleep(1);
            if (++tries > BLK_MAX_RETRY) { __hip_atomic_store(a.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        }
        return v;
    };
    for (int it = j; it < nitems; it += CL) {
        const int qt = it % a.nqt, cp = it / a.nqt;
        const int q0 = 16 * qt;
        if (it != j) load_wp(cp);
        // keys this query tile can see: all (whole) or the planes its queries live in
        int k0 = 0, k1 = L;
        if (!a.whole) {
            const int qlast = (q0 + 16 < L ? q0 + 16 : L) - 1;
            k0 = q0 >= b2 ? b2 : (q0 >= b1 ? b1 : 0);
            k1 = qlast >= b2 ? L : (qlast >= b1 ? b2 : b1);
        }
        const int nk = k1 - k0, nkt = (nk + 15) >> 4;
        // residual share of this head: input slabs h, h + H, ... of this tile's rows / columns (requested now, used in the epilogue)
        const int nq4 = a.ncols >> 2;
        constexpr int RU = 2;                                              // 16 x 256 / 4 = 1024 quads / 512 threads
        f32x4 rres[RU], rbias[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            rres[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            rbias[u] = rres[u];
            const int e = tid + BLK_NTH * u;
            const int rr = e / nq4, cq = e - rr * nq4;
            const int tok = q0 + rr;
            if (e < 16 * nq4 && tok < L && h < xks)
                rres[u] = *reinterpret_cast<const f32x4*>(a.x.p + (size_t)h * a.x.slab_stride + ((size_t)b * L + tok) * a.x.C + cp * a.ncols + 4 * cq);
            // (head 0 carries the bias: requested here too -- read in the epilogue it was a first-touch miss on the tail of the launch's slowest workgroups)
            if (e < 16 * nq4 && tok < L && h == 0) rbias[u] = *reinterpret_cast<const f32x4*>(a.bp + cp * a.ncols + 4 * cq);
        }
        // K rows (x d^-1/4), V^T, Q (x d^-1/4 log2 e: scores in the log2 domain) -> LDS
        for (int e = tid; e < nkt * 16 * QPR; e += BLK_NTH) {
            const int key = e / QPR, kq = e - key * QPR;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (key < nk) {
                kv = fetch(k0 + key, D + 4 * kq) * a.scale;
                vv = fetch(k0 + key, 2 * D + 4 * kq);
            }
            *reinterpret_cast<f32x4*>(Ks + key * KSTR + 4 * kq) = kv;
#pragma unroll
            for (int c = 0; c < 4; ++c) Vt[(4 * kq + c) * VSTR + key] = vv[c];
        }
        if (tid < 16 * QPR) {
            const int qr = tid / QPR, kq = tid - qr * QPR;
            const int tok = q0 + qr;
            f32x4 qv = {0.f, 0.f, 0.f, 0.f};
            if (tok < L) qv = fetch(tok, 4 * kq) * (a.scale * LOG2E);
            *reinterpret_cast<f32x4*>(Qs + qr * KSTR + 4 * kq) = qv;
        }
        __syncthreads();
        BLK_STAMP(8);
        // ---- wave = key tile: S^T = K Q^T (a query is a lane column), softmax pieces, O^T = V^T P^T
        {
            const int kt = wave;
            float m = -INFINITY, lsum = 0.f;
            f32x4 oacc[NDT];
#pragma unroll
            for (int o = 0; o < NDT; ++o) oacc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < nkt) {
                f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < NDT; ++u) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + jl) * KSTR + 16 * u + 4 * g);
                    const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + jl * KSTR + 16 * u + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[e], sacc, 0, 0, 0);
                }
                const int qtok = q0 + jl;                                  // this lane's query (column jl)
                const int qpl = qtok >= b2 ? 2 : (qtok >= b1 ? 1 : 0);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int key = k0 + 16 * kt + 4 * g + rr;
                    const int kpl = key >= b2 ? 2 : (key >= b1 ? 1 : 0);
                    const bool dead = key >= k1 || (!a.whole && kpl != qpl);
                    sacc[rr] = dead ? -INFINITY : sacc[rr];
                    m = fmaxf(m, sacc[rr]);
                }
                m = blk_swap_max16(m);
                m = blk_swap_max32(m);
                const bool live = m != -INFINITY;                          // (a key tile may hold only keys of other planes)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float pz = live ? __builtin_amdgcn_exp2f(sacc[rr] - m) : 0.f;
                    sacc[rr] = pz;
                    lsum += pz;
                }
                lsum += __shfl_xor(lsum, 16);
                lsum += __shfl_xor(lsum, 32);
#pragma unroll
                for (int o = 0; o < NDT; ++o) {
                    const f32x4 vf = *reinterpret_cast<const f32x4*>(Vt + (16 * o + jl) * VSTR + 16 * kt + 4 * g);
#pragma unroll
                    for (int sI = 0; sI < 4; ++sI) oacc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[sI], sacc[sI], oacc[o], 0, 0, 0);
                }
            }
            // park (m, l, O^T) of this key tile: O^T lane (query jl, group g) reg r = d index 16 o + 4 g + r
            float* op = Os + ((size_t)wave * 16 + jl) * KSTR + 4 * g;
#pragma unroll
            for (int o = 0; o < NDT; ++o) *reinterpret_cast<f32x4*>(op + 16 * o) = oacc[o];
            if (g == 0) { ml[(wave * 16 + jl) * 2] = m; ml[(wave * 16 + jl) * 2 + 1] = lsum; }
        }
        __syncthreads();
        // ---- merge the key tiles of every query: thread -> (query row, d quad)
        if (tid < 16 * QPR) {
            const int qr = tid / QPR, dq = tid - qr * QPR;
            float M = -INFINITY;
            for (int w = 0; w < nkt; ++w) M = fmaxf(M, ml[(w * 16 + qr) * 2]);
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            float lt = 0.f;
            for (int w = 0; w < nkt; ++w) {
                const float mm = ml[(w * 16 + qr) * 2];
                const float f = mm == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mm - M);
                lt += ml[(w * 16 + qr) * 2 + 1] * f;
                o += *reinterpret_cast<const f32x4*>(Os + ((size_t)w * 16 + qr) * KSTR + 4 * dq) * f;
            }
            const float inv = lt > 0.f ? 1.0f / lt : 0.f;                  // (rows past L: no live key)
            *reinterpret_cast<f32x4*>(Att + qr * AS + 4 * dq) = o * inv;
        }
        __syncthreads();
        BLK_STAMP(9);
        // ---- proj: [16 rows][ncols] = Att [16][D] x Wp[cols][h D ..]^T; wave -> column tiles wave, wave + 8
        {
            f32x4 af[NDT];
#pragma unroll
            for (int cc = 0; cc < NDT; ++cc) af[cc] = *reinterpret_cast<const f32x4*>(Att + jl * AS + 16 * cc + 4 * g);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int ct = wave + 8 * pt;
                if (ct < nct2) {
                    f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int cc = 0; cc < NDT; ++cc)
#pragma unroll
                        for (int sI = 0; sI < 4; ++sI) pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[cc][sI], wp[pt][cc][sI], pacc, 0, 0, 0);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) po[(4 * g + rr) * POS + 16 * ct + jl] = pacc[rr];
                }
            }
        }
        __syncthreads();
        // epilogue: thread -> (row, column quad); slab h = proj partial of head h + input slabs h, h + H, ... (+ bias in slab 0)
        float* const outp = a.out + (size_t)h * a.out_slab_stride;
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int e = tid + BLK_NTH * u;
            const int rr = e / nq4, cq = e - rr * nq4;
            const int tok = q0 + rr;
            if (e >= 16 * nq4 || tok >= L) continue;
            const int n = cp * a.ncols + 4 * cq;
            f32x4 v = *reinterpret_cast<const f32x4*>(po + rr * POS + 4 * cq);
            v += rres[u];
            for (int k = h + H; k < xks; k += H)
                v += *reinterpret_cast<const f32x4*>(a.x.p + (size_t)k * a.x.slab_stride + ((size_t)b * L + tok) * a.x.C + n);
            if (h == 0) v += rbias[u];
            mtv_store_out4(outp, ((size_t)b * L + tok) * C + n, v);
        }
        __syncthreads();                                                    // (LDS is reused by the next item)
    }
    BLK_STAMP(10);
}

// =====================================================================================
// host side
// =====================================================================================
size_t deep_block_smem_bytes(const DeepBlockArgs& a) {
    const int D = a.C / a.H;
    return (size_t)blk_lds(a.L, D, a.CS, a.ncols, a.RQ).total * 4;
}

// Cluster size / slicing of one attention block.  `a` arrives with B, L, C, H, r, t, whole, gs and the input set; on success CL, CS
// and the stage-2 / stage-3 work split are filled in.  false: this block keeps the three-launch path of deep.hip.
bool deep_block_configure(DeepBlockArgs& a, int force_cl, int force_rq, int max_wgs) {
    if (max_wgs > 128) max_wgs = 128;
    if (a.B < 1 || a.H < 1 || a.L < 1 || a.L > 128 || a.C % a.H || (a.C & 15)) return false;
    if (a.H > 8 || (a.H & (a.H - 1))) return false;                      // the heads are the output slabs: 1, 2, 4 or 8
    const int d = a.C / a.H;
    if (d != 16 && d != 32 && d != 64) return false;
    if (a.gs < 1 || a.C % a.gs) return false;
    if (!(a.x.ks == 1 || a.x.ks == 2 || a.x.ks == 4 || a.x.ks == 8)) return false;
    const int LP = (a.L + 15) / 16 * 16, nqt = LP / 16;
    // Row groups: above 32 tokens a workgroup stages 32-row groups of its K slice instead of all tokens (fewer, wider K slices: the partial
    // q | k | v rows that cross the cluster shrink with the slice count -- at [128 x 512] 16 slices were 12.6 MB of hand-off traffic per block)
    int rq_want = force_rq > 0 ? force_rq : (LP <= 32 ? 1 : (LP / 32 >= 4 ? 4 : LP / 32));
    while (rq_want & (rq_want - 1)) rq_want &= rq_want - 1;             // a power of two (the kernel indexes the cluster with shifts and masks): LP = 96 -> 2 groups of 48 rows
    for (int RQ = rq_want; RQ >= 1; RQ /= 2) {
        if (LP % (16 * RQ)) continue;
        int best = 0;
        for (int CL = RQ; CL <= 16; CL *= 2) {
            if (force_cl > 0 && CL != force_cl) continue;
            const int KSN = CL / RQ;
            if (a.C % KSN) continue;
            const int CS = a.C / KSN;
            if (CS < 16 || CS > 128 || (CS & (CS - 1)) || CS % a.gs || CS / a.gs > BLK_MAX_NG) continue;
            if ((long)a.B * a.H * CL > max_wgs) continue;                      // all workgroups resident together, on half of the CUs the launch may use (<= 128)
            int ncp = CL > nqt ? CL / nqt : 1;                               // column parts: one stage-3 item per workgroup where the cluster allows,
            while (a.C / ncp > 256) ncp *= 2;                                 // at most 256 columns per item (two column tiles per wave)
            if (a.C % ncp) continue;
            const int ncols = a.C / ncp;
            if (ncols > 256 || (ncols & 15)) continue;
            DeepBlockArgs probe = a;
            probe.CS = CS; probe.ncols = ncols; probe.RQ = RQ;
            if (deep_block_smem_bytes(probe) > 160 * 1024) continue;
            best = CL;                                                          // the largest admissible cluster
        }
        if (!best) continue;
        a.CL = best;
        a.RQ = RQ;
        a.KSN = best / RQ;
        a.RPQ = LP / RQ;
        a.CS = a.C / a.KSN;
        a.nqt = nqt;
        a.ncp = best > nqt ? best / nqt : 1;
        while (a.C / a.ncp > 256) a.ncp *= 2;
        a.ncols = a.C / a.ncp;
        a.rows_per = 2 * ((LP / 2 + best - 1) / best);                       // row PAIRS are dealt to the workgroups
        if (!deep_block_launchable(a)) continue;                            // (the ONE statement of what the kernel can run: also used by the launcher and the host self-test)
        return true;
    }
    return false;
}

// What k_deep_block can run: power-of-two cluster / row groups / K slices (blk & (CL - 1), cl_shift = ctz(CL) in the kernel), whole
// 16-row tiles per row group, the residency bound of the in-launch hand-offs, column parts of whole 16-column tiles.  Pointers are the launcher's business.
bool deep_block_launchable(const DeepBlockArgs& a) {
    if (a.CL < 1 || (a.CL & (a.CL - 1)) || a.CL > 16 || a.RQ < 1 || (a.RQ & (a.RQ - 1)) || a.KSN < 1 || (a.KSN & (a.KSN - 1)) || a.KSN * a.RQ != a.CL) return false;
    if (a.CS * a.KSN != a.C || a.CS < 16 || a.CS > 128 || (a.CS & (a.CS - 1))) return false;
    if (a.RPQ * a.RQ != (a.L + 15) / 16 * 16 || (a.RPQ & 15)) return false;
    if ((long)a.B * a.H * a.CL > 128 || a.ncols > 256 || (a.ncols & 15) || a.ncols * a.ncp != a.C) return false;
    return deep_block_smem_bytes(a) <= 160 * 1024;
}

size_t deep_block_part_floats(const DeepBlockArgs& a) { return (size_t)a.B * a.KSN * ((a.L + 15) / 16 * 16) * 3 * a.C * 2; }   // 8-byte {data, tag} granules, rows padded to 16
size_t deep_block_qkv_floats(const DeepBlockArgs& a) { return (size_t)a.B * a.L * 3 * a.C * 2; }     // 8-byte {data, tag} granules
size_t deep_block_stg_floats(const DeepBlockArgs& a) { return (size_t)a.B * a.H * a.KSN * a.RQ * (3 * BLK_MAX_NG * 2) * 4; }   // 16-byte granule pairs

hipError_t launch_deep_block(const DeepBlockArgs& a0, hipStream_t s) {
    DeepBlockArgs a = a0;
    const int d = a.C / a.H;
    if (!deep_block_launchable(a)) return hipErrorInvalidValue;
    if (!a.part || !a.qkv || !a.cnt || !a.fault || !a.x.p || !a.out || (a.RQ > 1 && !a.stg)) return hipErrorInvalidValue;
    a.stg_bytes = (unsigned)(deep_block_stg_floats(a) * 4);
    a.ksn_shift = __builtin_ctz(a.KSN);
    a.part_bytes = (unsigned)(deep_block_part_floats(a) * 4);
    a.qkv_bytes = (unsigned)(deep_block_qkv_floats(a) * 4);
    a.cl_shift = __builtin_ctz(a.CL);
    a.qw_shift = __builtin_ctz(a.CS / 4);
    {
        const double gsd = (double)a.gs;
        a.inv_n[0] = 1.0 / ((double)a.r * a.r * gsd);
        a.inv_n[1] = a.inv_n[2] = 1.0 / ((double)a.t * a.r * gsd);
        a.inv_n[3] = 1.0 / ((double)a.L * gsd);
    }
    const size_t smem = deep_block_smem_bytes(a);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(a.B * a.H * a.CL));
    if (d == 16) hipLaunchKernelGGL((k_deep_block<16>), grid, dim3(BLK_NTH), smem, s, a);
    else if (d == 32) hipLaunchKernelGGL((k_deep_block<32>), grid, dim3(BLK_NTH), smem, s, a);
    else if (d == 64) hipLaunchKernelGGL((k_deep_block<64>), grid, dim3(BLK_NTH), smem, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t deep_block_init_attrs() {
    const void* fn[] = {reinterpret_cast<const void*>(&k_deep_block<16>), reinterpret_cast<const void*>(&k_deep_block<32>),
                        reinterpret_cast<const void*>(&k_deep_block<64>)};
    for (const void* f : fn) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace mtv
