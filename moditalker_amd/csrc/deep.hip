// Convolutions of the DEEP levels of the tri-plane UNet (<= 128 tokens per clip: levels 2 and 3 of the base model) for gfx950.
//
// Replaces, at those levels: ResBlock in_layers / out_layers / skip_connection (MToV/models/ddpm/unet.py:131-167, 178-207) and
// the attention blocks' qkv / proj_out conv1d (unet.py:234, 242, 251, 253) with the GroupNorm in front of them
// (diffusionmodules.py:156-173).  Same arithmetic as conv.hip (exact f32 on v_mfma_f32_16x16x4_f32), different dataflow:
//
// At <= 128 tokens a conv is a weight stream (9.4-18.9 MB for 0.15-1.2 GFLOP) and a launch of k_conv spends most of its
// 13-26 us on dependent round trips: GroupNorm statistics from the producers' atomics, gather tables, a ring fill that only
// starts after them, and a three-hop cross-workgroup split-K hand-off (profiles/r03_conv_phase_stamps.txt).  Here
//   * the K dimension is cut into KS channel SLICES (all taps of CSm input channels each) and every slice writes its
//     partial result to its own slab -- nobody finishes the sum inside the launch: no slab round trip, no ticket;
//   * every CONSUMER adds the slabs of the channel slice it reads (slab order: run-to-run bit-equal) and computes the
//     GroupNorm statistics of that slice itself -- a workgroup stages ALL tokens of its channel slice in LDS, so the
//     statistics need no atomics and no table, and GroupNorm / FiLM / SiLU are applied ONCE per element on the way in
//     (k_conv re-applies them per tap and per column tile);
//   * a workgroup's weights are one contiguous run of the repacked matrix (k_deep_repack) that depends on nothing: the four
//     MFMA waves request them at kernel entry, straight into registers (1 KB per wave instruction, non-temporal), while the
//     four STAGER waves fetch, sum, normalise and park the activation slice -- the two streams have separate vmcnt queues
//     (a wave's loads return in order: one wave doing both would wait for its HBM-cold weights before it could touch the
//     L2-warm activations).
//
// Workgroup = (clip, row group, K slice, column tile of 16 NT channels), 512 threads.  Row group: all planes (nrg = 1) or
// {xy} | {yt, xt} (nrg = 2; GroupNorm statistics are per plane, 3x3 taps never cross planes) -- RT row tiles of 16 tokens.
// MFMA operand maps as in conv.hip: A lane (i, q) holds x[row i][c0 + 4q .. + 3] (one ds_read_b128 from the staged slice, row
// chosen by the tap through a per-workgroup row table, zero padding = a zero row), B lane (j, q) holds
// W[c0 + 4q .. + 3][n0 + j] (one 16-byte global load, lane-linear in the repacked matrix), MFMA step s consumes component s.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "mtv_internal.h"

namespace mtv {

typedef float f32x4 __attribute__((ext_vector_type(4)));


namespace {

constexpr int DEEP_PAD = 8;          // LDS row stride = slice channels + 8 floats: the 16 lanes of a ds_read_b128 lane group hit
                                     // 16 distinct bank quads (stride/4 = 2 mod 16 over rows i, + q; MI355X_MICROARCH.md LDS table)
constexpr int DEEP_NTH = 512;
constexpr int DEEP_MAX_NG = 32;      // GroupNorm groups per channel slice

__device__ __forceinline__ float deep_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ int deep_usgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// sum of the KSI slabs of one 16-byte quad, slab order
template <int KSI>
__device__ __forceinline__ void slab_issue(const float* p, unsigned stride, f32x4 (&v)[KSI]) {
#pragma unroll
    for (int k = 0; k < KSI; ++k) v[k] = *reinterpret_cast<const f32x4*>(p + (size_t)k * stride);
}
template <int KSI>
__device__ __forceinline__ f32x4 slab_fold(const f32x4 (&v)[KSI]) {
    f32x4 s = v[0];
#pragma unroll
    for (int k = 1; k < KSI; ++k) s += v[k];
    return s;
}
__device__ __forceinline__ f32x4 slab_sum_rt(const float* p, unsigned stride, int ks) {     // run-time slab count <= 8 (epilogue residual)
    if (ks == 1) return *reinterpret_cast<const f32x4*>(p);
    f32x4 t[8];                                               // all eight requested at once (absent slabs re-read slab 0): a loop that
#pragma unroll                                                // adds as it goes waits a full round trip per slab
    for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(p + (size_t)(k < ks ? k : 0) * stride);
    f32x4 s = t[0];
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < ks) s += t[k];                                // slab order
    return s;
}

// Staging of a channel slice (nrows rows x 4 QW channels) of a slab tensor, all 512 threads: thread (rl, qd) takes the quad
// column qd of rows rl, rl + RP, ... (RP = 512 / QW), NLD = 16 sixteen-byte loads in flight per pass (U rows x KSI slabs).
// The FIRST pass is split into issue / consume so that the caller can request its weights in between: a wave's loads return
// in order, so the (L2 / Infinity-Cache warm) activations must be requested BEFORE the HBM-cold weights.
template <int KSI>
struct StageRegs {
    static constexpr int U = 16 / KSI;
    f32x4 v[U][KSI];
};
template <int KSI>
__device__ __forceinline__ void stage_issue(StageRegs<KSI>& sr, const float* colp, unsigned slab_stride, int C, int nrows, int row0, int RP) {
    const int rbase = row0 - (row0 % RP);                // (row0 = this thread's row lane + a multiple of RP)
#pragma unroll
    for (int u = 0; u < StageRegs<KSI>::U; ++u) {
        if (rbase + u * RP >= nrows) break;              // wave-uniform: nobody has a row in this slot -- no loads at all
        const int row = row0 + u * RP;
        const float* p = colp + (size_t)(row < nrows ? row : 0) * C;
#pragma unroll
        for (int k = 0; k < KSI; ++k) sr.v[u][k] = *reinterpret_cast<const f32x4*>(p + (size_t)k * slab_stride);
    }
}
template <int KSI, bool STATS>
__device__ __forceinline__ void stage_consume(const StageRegs<KSI>& sr, int nrows, int row0, int RP, float* ldsq, int lstride, int tok0, int b1s, int b2s,
                                              double (&acc)[6]) {
#pragma unroll
    for (int u = 0; u < StageRegs<KSI>::U; ++u) {
        const int row = row0 + u * RP;
        if (row < nrows) {
            f32x4 x = sr.v[u][0];
#pragma unroll
            for (int k = 1; k < KSI; ++k) x += sr.v[u][k];           // slab order
            *reinterpret_cast<f32x4*>(ldsq + row * lstride) = x;
            if constexpr (STATS) {
                const int tk = tok0 + row;
                const double s = ((double)x[0] + (double)x[1]) + ((double)x[2] + (double)x[3]);
                const double ss = ((double)x[0] * x[0] + (double)x[1] * x[1]) + ((double)x[2] * x[2] + (double)x[3] * x[3]);
                const bool p2 = tk >= b2s, p1 = tk >= b1s && !p2;
                acc[0] += (!p1 && !p2) ? s : 0.0;
                acc[1] += (!p1 && !p2) ? ss : 0.0;
                acc[2] += p1 ? s : 0.0;
                acc[3] += p1 ? ss : 0.0;
                acc[4] += p2 ? s : 0.0;
                acc[5] += p2 ? ss : 0.0;
            }
        }
    }
}
// one slice: first pass issue -> mid() -> consume -> further passes
template <int KSI, bool STATS, class Mid>
__device__ __forceinline__ void stage_slice(const float* base, unsigned slab_stride, int C, int nrows, int qw_shift, int tid, float* lds, int lstride, int tok0,
                                            int b1s, int b2s, double (&acc)[6], Mid mid) {
    const int QW = 1 << qw_shift, RP = DEEP_NTH >> qw_shift;
    const int qd = tid & (QW - 1), rl = tid >> qw_shift;
    const float* colp = base + 4 * qd;
    float* ldsq = lds + 4 * qd;
    StageRegs<KSI> sr;
    stage_issue<KSI>(sr, colp, slab_stride, C, nrows, rl, RP);
    mid();
    stage_consume<KSI, STATS>(sr, nrows, rl, RP, ldsq, lstride, tok0, b1s, b2s, acc);
    for (int row0 = rl + RP * StageRegs<KSI>::U; row0 < nrows; row0 += RP * StageRegs<KSI>::U) {
        stage_issue<KSI>(sr, colp, slab_stride, C, nrows, row0, RP);
        stage_consume<KSI, STATS>(sr, nrows, row0, RP, ldsq, lstride, tok0, b1s, b2s, acc);
    }
}
template <bool STATS, class Mid>
__device__ __forceinline__ void stage_slice_ks(const DeepSrc& src, const float* base, int nrows, int qw_shift, int tid, float* lds, int lstride, int tok0,
                                               int b1s, int b2s, double (&acc)[6], Mid mid) {
    switch (src.ks) {
        case 1: stage_slice<1, STATS>(base, src.slab_stride, src.C, nrows, qw_shift, tid, lds, lstride, tok0, b1s, b2s, acc, mid); break;
        case 2: stage_slice<2, STATS>(base, src.slab_stride, src.C, nrows, qw_shift, tid, lds, lstride, tok0, b1s, b2s, acc, mid); break;
        case 4: stage_slice<4, STATS>(base, src.slab_stride, src.C, nrows, qw_shift, tid, lds, lstride, tok0, b1s, b2s, acc, mid); break;
        default: stage_slice<8, STATS>(base, src.slab_stride, src.C, nrows, qw_shift, tid, lds, lstride, tok0, b1s, b2s, acc, mid); break;
    }
}


// ---- in-launch completion (DeepFin): shared by k_deep_conv and k_deep_attn --------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned long long deep_gu64;
__device__ __forceinline__ void deep_park_quad(float* dst, const f32x4& v) {          // write-through (sc1) 8-byte stores
    deep_gu64* d = (deep_gu64*)(unsigned long long)dst;
    __hip_atomic_store(d, ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 1, ((unsigned long long)__float_as_uint(v[3]) << 32) | __float_as_uint(v[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// All slabs of one quad, L1-bypassing loads, all in flight, summed in slab order.
__device__ __forceinline__ f32x4 deep_gather_quad(const float* slab0, unsigned stride, int ks) {
    const deep_gu64* s0 = (const deep_gu64*)(unsigned long long)slab0;
    unsigned long long t0[8], t1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t o = (size_t)(k < ks ? k : 0) * (stride / 2);
        t0[k] = __hip_atomic_load(s0 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t1[k] = __hip_atomic_load(s0 + o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    f32x4 v = {__uint_as_float((unsigned)t0[0]), __uint_as_float((unsigned)(t0[0] >> 32)), __uint_as_float((unsigned)t1[0]), __uint_as_float((unsigned)(t1[0] >> 32))};
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < ks) {
            v[0] += __uint_as_float((unsigned)t0[k]);
            v[1] += __uint_as_float((unsigned)(t0[k] >> 32));
            v[2] += __uint_as_float((unsigned)t1[k]);
            v[3] += __uint_as_float((unsigned)(t1[k] >> 32));
        }
    return v;
}
// After every thread of the workgroup has parked its quads: drain, take the tile's ticket; true in the workgroup that arrived last.
// `scratch`: LDS, [0] = flag, statistics slots from +4 (zeroed here for the last arriver).
__device__ __forceinline__ bool deep_fin_arrive(const DeepFin& f, int tile, int arrivals, float* scratch, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    __syncthreads();
    int* flag = reinterpret_cast<int*>(scratch);
    if (tid == 0) {
        const bool last = __hip_atomic_fetch_add(f.tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == arrivals - 1;
        if (last) __hip_atomic_store(f.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning for the next launch
        *flag = last ? 1 : 0;
    }
    double* st = reinterpret_cast<double*>(scratch + 4);
    for (int e = tid; e < f.nstat * 96 * 2; e += DEEP_NTH) st[e] = 0.0;
    __syncthreads();
    return *flag != 0;
}
// (consumer `t` of the tensor: its group size / channel offset BY VALUE -- a run-time index into a stat[] array inside a local struct
// would put the struct in scratch memory)
__device__ __forceinline__ void deep_stat_one(const StatOut so, int t, const SegInfo seg, float* scratch, int tok, int n, const f32x4& v) {
    double* st = reinterpret_cast<double*>(scratch + 4);
    const int sg = tok >= seg.b2 ? 2 : (tok >= seg.b1 ? 1 : 0);
    const int gs = so.gs;
    if ((gs & 3) == 0) {
        const int g = (so.coff + n) / gs;
        atomicAdd(&st[((t * 96) + sg * 32 + g) * 2], ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]));
        atomicAdd(&st[((t * 96) + sg * 32 + g) * 2 + 1], ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]));
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = (so.coff + n + k) / gs;
            atomicAdd(&st[((t * 96) + sg * 32 + g) * 2], (double)v[k]);
            atomicAdd(&st[((t * 96) + sg * 32 + g) * 2 + 1], (double)v[k] * v[k]);
        }
    }
}
__device__ __forceinline__ void deep_stat_flush_one(const StatOut so, int t, unsigned cstride, const float* scratch, int b, int tid) {
    const double* st = reinterpret_cast<const double*>(scratch + 4);
    for (int e = tid; e < 96; e += DEEP_NTH) {
        const double sx = st[(t * 96 + e) * 2], sy = st[(t * 96 + e) * 2 + 1];
        if (sy != 0.0) {
            double* dst = so.sums + (size_t)(blockIdx.x & (STAT_COPIES - 1)) * cstride + ((size_t)b * 96 + e) * 2;
            atomicAdd(dst, sx);
            atomicAdd(dst + 1, sy);
        }
    }
}
__device__ __forceinline__ void deep_fin_stat(const DeepFin& f, float* scratch, int tok, int n, const f32x4& v) {
    if (f.nstat > 0) deep_stat_one(f.stat[0], 0, f.seg, scratch, tok, n, v);
    if (f.nstat > 1) deep_stat_one(f.stat[1], 1, f.seg, scratch, tok, n, v);
}
__device__ __forceinline__ void deep_fin_flush(const DeepFin& f, float* scratch, int b, int tid) {
    if (!f.nstat) return;
    __syncthreads();
    deep_stat_flush_one(f.stat[0], 0, f.stat_cstride, scratch, b, tid);
    if (f.nstat > 1) deep_stat_flush_one(f.stat[1], 1, f.stat_cstride, scratch, b, tid);
}
constexpr int DEEP_FIN_FLOATS = 4 + 2 * 96 * 2 * 2;      // flag + [2 consumers][96][2] doubles

// ---- completion by data-tagged granules (round 5; same transport as block.hip): two 8-byte {value, epoch} granules per 16-byte store / load,
// cache policy sc0 | sc1 (write-through, L1-bypassing).  No drain, no flag, no ticket wait: the reader polls the data.
typedef unsigned deep_u32x4 __attribute__((ext_vector_type(4)));
constexpr int DEEP_SC = 17;
constexpr int DEEP_MAX_RETRY = 1 << 21;      // (seconds: a bound against a hang, never a pace)
__device__ __forceinline__ void deep_put_granules(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, const f32x4& v, unsigned tag) {
    __builtin_amdgcn_raw_buffer_store_b128(deep_u32x4{__float_as_uint(v[0]), tag, __float_as_uint(v[1]), tag}, rs, byte_off, 0, DEEP_SC);
    __builtin_amdgcn_raw_buffer_store_b128(deep_u32x4{__float_as_uint(v[2]), tag, __float_as_uint(v[3]), tag}, rs, byte_off + 16, 0, DEEP_SC);
}
// quad `qoff` (granule index of its first element) of partial slices 1 .. ns - 1, polled until every tag is this launch's, added to v in slice order
__device__ __forceinline__ void deep_add_granules(__amdgpu_buffer_rsrc_t rs, unsigned qoff, unsigned slice_granules, int ns, unsigned tag, int* fault, f32x4& v) {
    deep_u32x4 ta[7], tb[7];
    int tries = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const unsigned o = (qoff + (unsigned)(k < ns - 1 ? k : 0) * slice_granules) * 8u;
            ta[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, DEEP_SC);
            tb[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, o + 16, 0, DEEP_SC);
        }
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (k < ns - 1) ok = ok && ta[k][1] == tag && ta[k][3] == tag && tb[k][1] == tag && tb[k][3] == tag;
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
        if (++tries > DEEP_MAX_RETRY) { __hip_atomic_store(fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (k < ns - 1) {
            v[0] += __uint_as_float(ta[k][0]); v[1] += __uint_as_float(ta[k][2]);
            v[2] += __uint_as_float(tb[k][0]); v[3] += __uint_as_float(tb[k][2]);
        }
}

}  // namespace

#ifdef MTV_DEEP_STAMP   // s_memtime of lane 0 of waves 0 and 4 of workgroups 0 and gridDim / 2: [wg][wave][16 slots]
#define DEEP_STAMP(k) do { if (a.dbg && (tid == 0 || tid == 256) && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) \
                               a.dbg[(blockIdx.x ? 32 : 0) + (tid ? 16 : 0) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DEEP_STAMP(k) do { } while (0)
#endif

template <int RT, int NT>
__global__ __launch_bounds__(DEEP_NTH) void k_deep_conv(const DeepArgs a) {
    const int tid = threadIdx.x;
    DEEP_STAMP(0);
    touch_kernargs<(int)sizeof(DeepArgs)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * RT;
    constexpr int G = RT * NT >= 24 ? 2 : 3;           // chunks per weight group; two groups in flight per wave
    const int lane = tid & 63;
    const int wave = deep_usgpr(tid >> 6);
    const int i = lane & 15, q = lane >> 4;
    // ---- workgroup -> (column tile j, clip b, row group rg, K slice s).  Consecutive workgroup ids (round-robin over the 8
    // XCDs) share the column tile and differ in (b, rg, s): all column tiles of one (rg, s) -- which read the SAME activation
    // slice -- meet on one XCD when nslots is a multiple of 8 (speed only).
    const int blk = deep_usgpr((int)blockIdx.x);
    int j, slot;
    if (a.xm) {
        // (weight-dominated convs at 128 tokens: the row groups {xy} | {yt, xt} of a (K slice, column tile) read the same weight run from two XCDs -- 2 x the weights
        // through the fabric, profiles/r06_per_op_traffic.txt.  Here the XCD = (K slice, low bits of the column tile) and both row groups sit on it; an activation
        // slice is then read by 8 / KS XCDs instead of one.  Speed only.)
        const int xcd = blk & 7, idx = blk >> 3;
        j = ((idx >> 1) << a.xm_jbits) + (xcd >> a.ks_shift);
        slot = ((idx & 1) << a.ks_shift) + (xcd & (a.KS - 1));
    } else {
        j = FDiv{a.inv_nslots}(blk, a.nslots);
        slot = blk - j * a.nslots;
    }
    j = deep_usgpr(j);
    slot = deep_usgpr(slot);
    const int s = slot & (a.KS - 1);
    const int rgb = slot >> a.ks_shift;
    const int rg = rgb & (a.nrg - 1);
    const int b = rgb >> (a.nrg - 1);
    // tagged completion (DeepFin::tagged): this workgroup's entry ticket -> the launch's epoch; requested first of all, read much later
    // (NT == 1 tiles only: the 48-column tiles have no registers to spare -- launch_deep_conv rejects the combination)
    const bool tg = NT == 1 && a.fin.out != nullptr && a.fin.tagged != 0;
    unsigned long long early = 0;
    if (tg && tid == 0) early = __hip_atomic_fetch_add(a.fin.ecnt + ((b * a.nrg + rg) * a.tiles_n + j), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // geometry: output level and the level of the tapped source
    const int r = a.r, t = a.t;
    const int b1 = r * r, b2 = b1 + t * r, L = b2 + t * r;
    const int rs = a.up_main ? r >> 1 : (a.pool_main ? r << 1 : r), ts = a.up_main ? t >> 1 : (a.pool_main ? t << 1 : t);
    const int b1s = rs * rs, b2s = b1s + ts * rs, Ls = b2s + ts * rs;
    const int rg_tok0 = (a.nrg == 2 && rg) ? b1 : 0;
    const int rg_ntok = a.nrg == 2 ? (rg ? L - b1 : b1) : L;
    const int src_tok0 = (a.nrg == 2 && rg) ? b1s : 0;
    const int src_ntok = a.nrg == 2 ? (rg ? Ls - b1s : b1s) : Ls;
    const int SM = a.CSm + DEEP_PAD, SS = a.CSs + DEEP_PAD;
    const int zoff_main = a.src_rows_max * SM;         // float offset of the zero row of the main slice
    float* const lmain = smem;
    float* const lskip = smem + a.lds_skip;
    int* const idx = reinterpret_cast<int*>(smem + a.lds_idx);       // [ntaps][ROWS] float offsets into lmain | [ROWS] into lskip
    double* const sdp = reinterpret_cast<double*>(smem + a.lds_stat);     // [3 planes][DEEP_MAX_NG][2]: (sum, sum of squares) of the slice
    const int nch = a.nch;
    DEEP_STAMP(1);
    // the row table of this row group -- [ntaps][ROWS] float offsets into the staged main slice (zero row = padding), then [ROWS] into
    // the skip slice -- comes from the host (deep_rowtab): requested before anything else, parked in LDS once the weights are on their way
    int treg[3];
    {
        const int* tb = a.rowtab + rg * ((a.ntaps + 1) * ROWS);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = tid + DEEP_NTH * u;
            treg[u] = tb[e < (a.ntaps + 1) * ROWS ? e : 0];
        }
    }

    // =========================================================================================================== phase 0
    // request order per wave: activation slice (first pass) -> GroupNorm vectors -> weights; then the row table while they fly
    // Weights through a buffer descriptor over this workgroup's run of the repacked matrix: chunk indices past the end of the
    // slice read as ZERO without a branch, so every wave executes the same straight-line request sequence (exact vmcnt).
    f32x4 bq[2][G][NT];
    const int n_it = (nch + 7) >> 3;                   // iterations of the chunk loop: wave w takes chunks w, w + 8, ...
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.W + ((size_t)(j * a.KS + s) * nch) * (NT * 256)), 0, nch * NT * 1024, 0x00020000);
    auto wload = [&](f32x4 (&dst)[G][NT], int k0) {     // iterations k0 .. k0 + G - 1
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int soff = (wave + 8 * (k0 + g)) * (NT * 1024);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                dst[g][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16 + nt * 1024, soff, 2));   // aux 2 = nt
        }
    };
    const int qw_shift = a.cpt_shift + 2;              // log2(CSm / 4)
    const int QWm = 1 << qw_shift;
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    // (raw registers: ANY use of a loaded value makes the wave wait for every OLDER load too -- `1 + scale` is formed in phase 1)
    f32x4 ga, be, fsc, sh;
    const int cm0 = s * a.CSm;                         // first channel of this slice on the concatenated channel axis
    auto mid = [&]() {
        DEEP_STAMP(12);
        {   // GroupNorm / FiLM vectors of this thread's channel quad (the same in every row it stages).  UNCONDITIONAL loads
            // into their final registers: the launcher points absent vectors at a zero buffer -- a conditional load here ends
            // in a register copy at the join, and a copy of a loaded value waits for every older load (the whole slice)
            const int cg = cm0 + 4 * (tid & (QWm - 1));
            const float* fm = a.film + (size_t)b * a.film_stride;
            ga = *reinterpret_cast<const f32x4*>(a.gamma + cg);
            be = *reinterpret_cast<const f32x4*>(a.beta + cg);
            fsc = *reinterpret_cast<const f32x4*>(fm + cg);
            sh = *reinterpret_cast<const f32x4*>(fm + a.Cmain + cg);
        }
        wload(bq[0], 0);
        wload(bq[1], G);
        DEEP_STAMP(13);
        // row table (precomputed on the host, requested first of all: deep_rowtab), zero rows, statistics slots: LDS work under the loads
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = tid + DEEP_NTH * u;
            if (e < (a.ntaps + 1) * ROWS) idx[e] = treg[u];
        }
        for (int e = tid; e < a.CSm; e += DEEP_NTH) lmain[zoff_main + e] = 0.f;
        if (a.Cskip)
            for (int e = tid; e < a.CSs; e += DEEP_NTH) lskip[ROWS * SS + e] = 0.f;
        for (int e = tid; e < 3 * DEEP_MAX_NG * 2; e += DEEP_NTH) sdp[e] = 0.0;
        DEEP_STAMP(8);
        __syncthreads();                                              // (statistics slots are zero before anybody adds to them)
        DEEP_STAMP(14);
    };
    {
        const int part = (a.main[1].p && cm0 >= a.main[0].C) ? 1 : 0;
        const DeepSrc& src = a.main[part];
        const float* base = src.p + ((size_t)b * a.Lsrc + src_tok0) * src.C + (cm0 - (part ? a.main[0].C : 0));
        if (a.gn) stage_slice_ks<true>(src, base, src_ntok, qw_shift, tid, lmain, SM, src_tok0, b1s, b2s, acc, mid);
        else stage_slice_ks<false>(src, base, src_ntok, qw_shift, tid, lmain, SM, src_tok0, b1s, b2s, acc, mid);
    }
    DEEP_STAMP(9);
    if (a.Cskip) {
        // (requested behind the weights, which the MFMA loop needs anyway: the slice arrives with them)
        const int cs0 = s * a.CSs;
        const int sp = (a.skip[1].p && cs0 >= a.skip[0].C) ? 1 : 0;
        const DeepSrc& ssrc = a.skip[sp];
        const float* sbase = ssrc.p + ((size_t)b * a.Lout + rg_tok0) * ssrc.C + (cs0 - (sp ? a.skip[0].C : 0));
        int sq_shift = 0;
        while ((4 << sq_shift) < a.CSs) ++sq_shift;                  // log2(CSs / 4)
        double dummy[6];
        stage_slice_ks<false>(ssrc, sbase, rg_ntok, sq_shift, tid, lskip, SS, 0, 0, 0, dummy, []() {});
    }
    DEEP_STAMP(10);
    const int qdm = tid & (QWm - 1);
    const int gi = (4 * qdm) / (a.gn ? a.gs : 4);                      // GroupNorm group of this thread's quad, within the slice
    if (a.gn) {
        // per-thread partial sums -> the slice's (plane, group) slots.  Quads of one group sit in adjacent lanes: fold them first
        // (2 DPP-free shuffles would cost more than the contention they save at gs <= 16: plain LDS fp64 atomics)
        // the gs / 4 quads of a group sit in adjacent lanes: fold pairs / quads of lanes first (DPP quad permutes: VALU, no LDS)
        const int gq = a.gs >> 2;
        auto dpp_add = [](double v, auto ctrl) -> double {
            constexpr int CTRL = decltype(ctrl)::value;
            const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
            const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, false);
            const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, false);
            return v + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        };
        if (gq >= 2) {
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] = dpp_add(acc[k], std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
        }
        if (gq >= 4) {
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] = dpp_add(acc[k], std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
        }
        const int fold = gq >= 4 ? 4 : (gq >= 2 ? 2 : 1);
        if ((qdm & (fold - 1)) == 0) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
                if (acc[2 * p + 1] != 0.0) {
                    atomicAdd(&sdp[(p * DEEP_MAX_NG + gi) * 2], acc[2 * p]);
                    atomicAdd(&sdp[(p * DEEP_MAX_NG + gi) * 2 + 1], acc[2 * p + 1]);
                }
        }
    }
    if (tg && tid == 0) *reinterpret_cast<unsigned*>(smem + a.lds_fin) = (unsigned)(early >> a.ks_shift) + 1u;     // (the ticket returned long ago)
    DEEP_STAMP(2);
    __syncthreads();
    DEEP_STAMP(3);
    // =========================================================================================================== phase 1
    if (a.gn) {
        // y = x * A + B with A = rstd gamma (1 + film_scale), B = (beta - rstd gamma mean)(1 + film_scale) + film_shift, then SiLU:
        // applied ONCE per element, in place in LDS (this thread's quad column: coefficients per plane in registers)
        const int RP = DEEP_NTH >> qw_shift;
        const int rl = tid >> qw_shift;
        const bool act = a.act != 0;
        int cur_p = -1;
        f32x4 A = {0.f, 0.f, 0.f, 0.f}, Bc = A;
        for (int row = rl; row < src_ntok; row += RP) {
            const int tk = src_tok0 + row;
            const int p = tk >= b2s ? 2 : (tk >= b1s ? 1 : 0);
            if (p != cur_p) {                                  // (rows ascend: at most three times, once for most threads)
                cur_p = p;
                double sx, sy;
                if (a.whole) {
                    sx = (sdp[(0 * DEEP_MAX_NG + gi) * 2] + sdp[(1 * DEEP_MAX_NG + gi) * 2]) + sdp[(2 * DEEP_MAX_NG + gi) * 2];
                    sy = (sdp[(0 * DEEP_MAX_NG + gi) * 2 + 1] + sdp[(1 * DEEP_MAX_NG + gi) * 2 + 1]) + sdp[(2 * DEEP_MAX_NG + gi) * 2 + 1];
                } else {
                    sx = sdp[(p * DEEP_MAX_NG + gi) * 2];
                    sy = sdp[(p * DEEP_MAX_NG + gi) * 2 + 1];
                }
                const double inv_n = a.whole ? a.inv_n[3] : a.inv_n[p];
                const double mean = sx * inv_n;
                double var = sy * inv_n - mean * mean;
                var = var < 0.0 ? 0.0 : var;
                const float mu = (float)mean, rstd = 1.0f / sqrtf((float)var + 1e-5f);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float sc = rstd * ga[k];
                    const float bi = be[k] - sc * mu;
                    const float s1 = 1.0f + fsc[k];
                    A[k] = sc * s1;
                    Bc[k] = fmaf(bi, s1, sh[k]);
                }
            }
            f32x4* cell = reinterpret_cast<f32x4*>(lmain + row * SM + 4 * qdm);
            f32x4 v = *cell;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float y = fmaf(v[k], A[k], Bc[k]);
                v[k] = act ? deep_silu(y) : y;
            }
            *cell = v;
        }
    }
    __syncthreads();
    if (a.pool_main) {
        // ResBlock(down=True): the conv reads AvgPool2d(SiLU(GroupNorm(x))) (unet.py:179-184): 2x2 mean, per plane, of the transformed source
        // rows -> this level's rows, parked behind the source slice; the row table points there.  Sum order as F.avg_pool2d: row-major window.
        float* const lpool = smem + a.lds_pool;
        const int RP = DEEP_NTH >> qw_shift;
        const int qdp = tid & (QWm - 1);
        for (int ro = tid >> qw_shift; ro <= rg_ntok; ro += RP) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ro < rg_ntok) {
                const int tok = rg_tok0 + ro;
                const int p = tok >= b2 ? 2 : (tok >= b1 ? 1 : 0);
                const int off = p == 0 ? 0 : (p == 1 ? b1 : b2), offs = p == 0 ? 0 : (p == 1 ? b1s : b2s);
                const int local = tok - off, y = FDiv{a.inv_r}(local, r), x = local - y * r;
                const int s00 = offs + (2 * y) * rs + 2 * x - src_tok0;
                const float* c0 = lmain + s00 * SM + 4 * qdp;
                const f32x4 v00 = *reinterpret_cast<const f32x4*>(c0), v01 = *reinterpret_cast<const f32x4*>(c0 + SM);
                const f32x4 v10 = *reinterpret_cast<const f32x4*>(c0 + rs * SM), v11 = *reinterpret_cast<const f32x4*>(c0 + rs * SM + SM);
                v = (((v00 + v01) + v10) + v11) * 0.25f;
            }
            *reinterpret_cast<f32x4*>(lpool + ro * SM + 4 * qdp) = v;          // (row rg_ntok = the zero row of the padding taps)
        }
        __syncthreads();
    }
    DEEP_STAMP(4);
    // =========================================================================================================== phase 2
    // epilogue operands of K slice 0 (bias, residual: they depend on nothing computed here) are requested now by every
    // thread for its first output quad; they land under the MFMA loop
    constexpr int QPR = 4 * NT, QUADS = ROWS * QPR;
    const int n0 = j * (16 * NT);
    f32x4 pre = {0.f, 0.f, 0.f, 0.f};
    auto epi_operands = [&](int e) -> f32x4 {
        const int rr = e / QPR, cq = e - rr * QPR;
        const int tok = rg_tok0 + rr, n = n0 + 4 * cq;
        f32x4 o = *reinterpret_cast<const f32x4*>(a.bias + n);
        if (a.bias2) o += *reinterpret_cast<const f32x4*>(a.bias2 + n);
        if (a.bias_b) o += *reinterpret_cast<const f32x4*>(a.bias_b + (size_t)b * a.bias_b_stride + n);
        if (a.res.p && !a.pool_res) {
            const int rtok = a.up_res ? (geo_source_t<FDiv>(FDiv{a.inv_r}, r, t, tok, 1, 1, true) & 0x0FFFFFFF) : tok;
            o += slab_sum_rt(a.res.p + ((size_t)b * a.Lres + rtok) * a.res.C + n, a.res.slab_stride, a.res.ks);
        }
        return o;
    };
    // ResBlock(down=True), identity skip: the residual is AvgPool2d of the raw input (unet.py:182-184).  The mean is linear, so K slice s adds
    // the pooled source slabs s, s + KS, ... (the consumers add the slices up anyway): four 16-byte loads per slab instead of 4 x ks in slice 0
    auto pooled_res = [&](int e) -> f32x4 {
        const int rr = e / QPR, cq = e - rr * QPR;
        const int tok = rg_tok0 + rr, n = n0 + 4 * cq;
        const int p = tok >= b2 ? 2 : (tok >= b1 ? 1 : 0);
        const int rr2 = r << 1, b1f = rr2 * rr2, b2f = b1f + (t << 1) * rr2;
        const int off = p == 0 ? 0 : (p == 1 ? b1 : b2), offf = p == 0 ? 0 : (p == 1 ? b1f : b2f);
        const int local = tok - off, y = FDiv{a.inv_r}(local, r), x = local - y * r;
        const float* base = a.res.p + ((size_t)b * a.Lres + offf + (2 * y) * rr2 + 2 * x) * a.res.C + n;
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
        for (int k = s; k < a.res.ks; k += a.KS) {
            const float* pk = base + (size_t)k * a.res.slab_stride;
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(pk), v01 = *reinterpret_cast<const f32x4*>(pk + a.res.C);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(pk + (size_t)rr2 * a.res.C), v11 = *reinterpret_cast<const f32x4*>(pk + (size_t)(rr2 + 1) * a.res.C);
            acc4 += (((v00 + v01) + v10) + v11) * 0.25f;
        }
        return acc4;
    };
    const bool pool_r = a.pool_res != 0 && a.res.p != nullptr && s < a.res.ks;
    const bool pre_ok = s == 0 && tid < QUADS && tid / QPR < rg_ntok;
    if (pre_ok) pre = epi_operands(tid);
    // (the pooled residual of this thread's first quad: requested now as well -- where the tile leaves the registers for it)
    const bool prep_ok = RT * NT <= 8 && pool_r && tid < QUADS && tid / QPR < rg_ntok;
    f32x4 prep = {0.f, 0.f, 0.f, 0.f};
    if constexpr (RT * NT <= 8)
        if (prep_ok) prep = pooled_res(tid);

    f32x4 accm[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accm[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        // Chunk loop, software pipelined: the A fragments of iteration k + 1 (row-table entries, then one ds_read_b128 per row
        // tile) are requested before the MFMAs of iteration k; the weights of iteration k + 2G replace those of k as soon as
        // its group is done.  Iterations past this wave's last chunk run on zero weights (at most one, in the last round).
        const int cpt_mask = (1 << a.cpt_shift) - 1;
        auto loadA = [&](int k, f32x4 (&av)[RT]) {
            int c = wave + 8 * k;
            c = c < nch ? c : nch - 1;                                 // (clamped: a valid address; its weights are zero)
            const bool sk = c >= a.nmain_ch;
            const int tap = sk ? a.ntaps : c >> a.cpt_shift;
            const int cc = sk ? c - a.nmain_ch : c & cpt_mask;
            const float* ab = (sk ? lskip : lmain) + cc * 16 + 4 * q;
            const int* ip = idx + tap * ROWS + i;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) av[rt] = *reinterpret_cast<const f32x4*>(ab + ip[16 * rt]);
        };
        auto mma = [&](const f32x4 (&av)[RT], const f32x4 (&w)[NT]) {
#pragma unroll
            for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        accm[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][sI], w[nt][sI], accm[rt][nt], 0, 0, 0);
        };
        constexpr bool PIPE = RT * NT < 24;                             // (the largest tile has no registers for a second fragment set
                                                                       //  -- and 96 MFMAs per chunk to hide one LDS round trip under)
        f32x4 avA[RT], avB[PIPE ? RT : 1];
        // one iteration: `par` = which fragment set holds iteration kk's rows (PIPE); the other one receives iteration kk + 1's
        auto step = [&](int kk, int par, const f32x4 (&w)[NT]) {
            if constexpr (PIPE) {
                if (par) { loadA(kk + 1, avA); mma(avB, w); }
                else { loadA(kk + 1, avB); mma(avA, w); }
            } else {
                loadA(kk, avA);
                mma(avA, w);
            }
        };
        if constexpr (PIPE) loadA(0, avA);
        int k = 0;
        for (; k + 2 * G <= n_it; k += 2 * G) {                         // (2G = 6 iterations per round: fragment parity = g & 1, then (G + g) & 1)
#pragma unroll
            for (int g = 0; g < G; ++g) step(k + g, g & 1, bq[0][g]);
            wload(bq[0], k + 2 * G);
#pragma unroll
            for (int g = 0; g < G; ++g) step(k + G + g, (G + g) & 1, bq[1][g]);
            wload(bq[1], k + 3 * G);
        }
        // tail: n_it - k in 0 .. 2G - 1 iterations, weights already requested; uniform guards, no memory requests inside
        const int rem = n_it - k;
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (g < rem) step(k + g, g & 1, bq[0][g]);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (G + g < rem) step(k + G + g, (G + g) & 1, bq[1][g]);
    }
    DEEP_STAMP(5);
    __syncthreads();
    DEEP_STAMP(6);
    // =========================================================================================================== phase 3
    // the eight partial tiles (one per wave) -> LDS as (row, col) images, summed in wave order by all threads; tiles too large
    // for eight images at once fold waves 4-7 into waves 0-3 first
    constexpr int LDR = 16 * NT + 4;
    constexpr bool ONE_ROUND = 8 * ROWS * LDR * 4 <= 96 * 1024;
    constexpr int NIMG = ONE_ROUND ? 8 : 4;
    float* const red = smem + a.lds_red;
    auto park = [&](int slot_w) {
        float* my = red + (size_t)slot_w * ROWS * LDR + (4 * q) * LDR + i;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) my[(16 * rt + rr) * LDR + 16 * nt] = accm[rt][nt][rr];
    };
    if constexpr (ONE_ROUND) {
        park(wave);
    } else {
        if (wave >= 4) park(wave - 4);
        __syncthreads();
        if (wave < 4) {
            const float* my = red + (size_t)wave * ROWS * LDR + (4 * q) * LDR + i;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) accm[rt][nt][rr] += my[(16 * rt + rr) * LDR + 16 * nt];
            park(wave);
        }
    }
    __syncthreads();
    float* const outp = a.out + (size_t)s * a.out_slab_stride;
    const bool fin = a.fin.out != nullptr && !tg;
    float* const fscratch = smem + a.lds_fin;
    const unsigned epoch = tg ? *reinterpret_cast<const unsigned*>(fscratch) : 0u;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(a.fin.gran, 0, tg ? (int)a.fin.gran_bytes : 0, 0x00020000);
    const unsigned slice_gran = (unsigned)((size_t)a.B * a.Lout * a.N);        // granules per partial slice
    if (tg && s == 0) {                                   // (uniform) statistics slots of the completing workgroup
        double* st = reinterpret_cast<double*>(fscratch + 4);
        for (int e = tid; e < a.fin.nstat * 96 * 2; e += DEEP_NTH) st[e] = 0.0;
        __syncthreads();
    }
    for (int e = tid; e < QUADS; e += DEEP_NTH) {
        const int rr = e / QPR, cq = e - rr * QPR;
        if (rr >= rg_ntok) continue;
        const float* rp = red + rr * LDR + 4 * cq;
        f32x4 v = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
        for (int w = 1; w < NIMG; ++w) v += *reinterpret_cast<const f32x4*>(rp + (size_t)w * ROWS * LDR);
        if (s == 0) v += (e == tid && pre_ok) ? pre : epi_operands(e);
        if (pool_r) v += (e == tid && prep_ok) ? prep : pooled_res(e);
        const size_t qoff = ((size_t)b * a.Lout + rg_tok0 + rr) * a.N + n0 + 4 * cq;
        float* dst = outp + qoff;
        if (fin) deep_park_quad(dst, v);
        else mtv_store_out4(outp, qoff, v);          // (tagged completion too: consumers emitted before the plain copy was asked for read the slabs)
        if (tg) {
            if (s) {
                deep_put_granules(grs, (unsigned)(((size_t)(s - 1) * slice_gran + qoff) * 8), v, epoch);
            } else {
                deep_add_granules(grs, (unsigned)qoff, slice_gran, a.KS, epoch, a.fin.fault, v);      // slices 1 .. KS - 1, slice order
                mtv_store_out4(a.fin.out, qoff, v);
                deep_fin_stat(a.fin, fscratch, rg_tok0 + rr, n0 + 4 * cq, v);
            }
        }
    }
    if (tg && s == 0) deep_fin_flush(a.fin, fscratch, b, tid);
    if (fin) {
        // ---- in-launch completion: the K slice that arrives last turns the tile's slabs into the plain tensor (+ statistics)
        float* scratch = smem + a.lds_fin;
        if (deep_fin_arrive(a.fin, (b * a.nrg + rg) * a.tiles_n + j, a.KS, scratch, tid)) {
            for (int e = tid; e < QUADS; e += DEEP_NTH) {
                const int rr = e / QPR, cq = e - rr * QPR;
                if (rr >= rg_ntok) continue;
                const int tok = rg_tok0 + rr, n = n0 + 4 * cq;
                const size_t off = ((size_t)b * a.Lout + tok) * a.N + n;
                const f32x4 v = deep_gather_quad(a.out + off, a.out_slab_stride, a.KS);
                mtv_store_out4(a.fin.out, off, v);
                deep_fin_stat(a.fin, scratch, tok, n, v);
            }
            deep_fin_flush(a.fin, scratch, b, tid);
        }
    }
    DEEP_STAMP(7);
}

// legacy conv matrix [krow = tap * Cmain + c | ntaps * Cmain + c_skip][ldw] -> [column tile][K slice][chunk][NT][lane][4]
__global__ void k_deep_repack(const float* W, int ldw, float* dst, DeepArgs a, int NT, long total) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int e = (int)(id & 3), lane = (int)((id >> 2) & 63);
    long rest = id >> 8;
    const int nt = (int)(rest % NT);
    rest /= NT;
    const int c = (int)(rest % a.nch);
    rest /= a.nch;
    const int s = (int)(rest % a.KS), j = (int)(rest / a.KS);
    const int jj = lane & 15, q = lane >> 4;
    int krow;
    if (c < a.nmain_ch) {
        const int tap = c >> a.cpt_shift, cc = c & ((1 << a.cpt_shift) - 1);
        krow = tap * a.Cmain + s * a.CSm + cc * 16 + 4 * q + e;
    } else {
        krow = a.ntaps * a.Cmain + s * a.CSs + (c - a.nmain_ch) * 16 + 4 * q + e;
    }
    dst[id] = W[(size_t)krow * ldw + (j * NT + nt) * 16 + jj];
}

// slabs -> plain tensor (+ GroupNorm statistics for legacy consumers, conv.hip / k_pool_down).  Grid: (channel blocks of 64,
// token blocks of 8, clips) -- wide on purpose: the pass is a latency chain (slab loads -> sum -> store), so it wants every CU
// to hold a few quads, not a few CUs to hold everything.  64 channels x 8 tokens = 128 quads per workgroup of 128 threads.
__global__ __launch_bounds__(128) void k_deep_finalize(const DeepFinArgs a) {
    touch_kernargs<(int)sizeof(DeepFinArgs)>();
    __shared__ double sdp[2][96][2];
    const int tid = threadIdx.x;
    const int b = blockIdx.z, tok = blockIdx.y * 8 + (tid >> 4), ch = blockIdx.x * 64 + 4 * (tid & 15);
    const int C = a.src.C;
    const bool live = tok < a.L && ch < C;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const size_t off = ((size_t)b * a.L + tok) * C + ch;
        const float* p = a.src.p + off;
        f32x4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(p + (size_t)(k < a.src.ks ? k : 0) * a.src.slab_stride);   // all in flight
        v = t[0];
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < a.src.ks) v += t[k];                       // slab order
        mtv_store_out4(a.out, off, v);
    }
    if (!a.nstat) return;
    for (int e = tid; e < a.nstat * 96 * 2; e += 128) (&sdp[0][0][0])[e] = 0.0;
    __syncthreads();
    if (live) {
        const int sg = tok >= a.seg.b2 ? 2 : (tok >= a.seg.b1 ? 1 : 0);
        for (int t = 0; t < a.nstat; ++t) {
            const int gs = a.stat[t].gs;
            if ((gs & 3) == 0) {
                const int g = (a.stat[t].coff + ch) / gs;
                atomicAdd(&sdp[t][sg * 32 + g][0], ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]));
                atomicAdd(&sdp[t][sg * 32 + g][1], ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int g = (a.stat[t].coff + ch + k) / gs;
                    atomicAdd(&sdp[t][sg * 32 + g][0], (double)v[k]);
                    atomicAdd(&sdp[t][sg * 32 + g][1], (double)v[k] * v[k]);
                }
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < a.nstat * 96; e += 128) {
        const int t = e / 96, r2 = e - t * 96;
        const double sx = sdp[t][r2][0], sy = sdp[t][r2][1];
        if (sy != 0.0) {
            double* dst = a.stat[t].sums + (size_t)((blockIdx.x + blockIdx.y) & (STAT_COPIES - 1)) * a.stat_cstride + ((size_t)b * 96 + r2) * 2;
            atomicAdd(dst, sx);
            atomicAdd(dst + 1, sy);
        }
    }
}


// =====================================================================================
// attention core + proj_out at <= 128 tokens
// =====================================================================================
__device__ __forceinline__ float deep_swap_max16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float deep_swap_max32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// 512 threads = 8 waves = QT query tiles (16 queries) x 8 / QT key parts (DeepAttnArgs::QT: 1 since round 6, 2 before).  Per head of the group: K rows, V^T and Q go to LDS (q and k
// scaled by d^-1/4, q also by log2 e: scores in the log2 domain), each wave computes S^T = K Q^T for its key tiles (a query is a
// lane COLUMN: softmax reductions are register values + two lane swaps; the P^T registers are the B operand of
// O^T += V^T P^T, as in k_attention), the four key parts of a query tile are merged through LDS into the normalised output
// rows.  Then D = attention rows x Wp[head group rows][column group] on the same MFMA, K split over the waves, summed in LDS.
template <int D>
__global__ __launch_bounds__(DEEP_NTH) void k_deep_attn(const DeepAttnArgs a) {
    const int tid = threadIdx.x;
    DEEP_STAMP(0);
    touch_kernargs<(int)sizeof(DeepAttnArgs)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int KSTR = D + 4, NDT = D / 16, QPR = D / 4;
    const int lane = tid & 63;
    const int wave = deep_usgpr(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int blk = deep_usgpr((int)blockIdx.x);
    const int cg = deep_usgpr(FDiv{a.inv_nslots}(blk, a.nslots));
    const int slot = blk - cg * a.nslots;
    const int hg = slot % a.nhg;
    const int rest = slot / a.nhg;
    const int qg = rest % a.nqg, b = rest / a.nqg;
    const int QT = deep_usgpr(a.QT), QR = 16 * QT;            // query tiles / query rows of this workgroup
    const int q0 = qg * QR;
    const int L = a.L, C = a.C, HPW = a.HPW, NC = a.NC;
    // tagged completion over the head groups (DeepFin::tagged): entry ticket -> the launch's epoch
    const bool tg = a.fin.out != nullptr && a.fin.tagged != 0;
    unsigned long long early = 0;
    if (tg && tid == 0) early = __hip_atomic_fetch_add(a.fin.ecnt + ((b * a.nqg + qg) * a.ncg + cg), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int b1 = a.r * a.r, b2 = b1 + a.t * a.r;
    // keys this query group can see: all (whole) or the planes its queries live in
    int k0 = 0, k1 = L;
    if (!a.whole) {
        const int qlast = (q0 + QR < L ? q0 + QR : L) - 1;
        k0 = q0 >= b2 ? b2 : (q0 >= b1 ? b1 : 0);
        k1 = qlast >= b2 ? L : (qlast >= b1 ? b2 : b1);
    }
    const int nk = k1 - k0, nkt = (nk + 15) >> 4;
    const int VSTR = a.kcap + 4, AS = HPW * D + 8, WS = HPW * D + 4;
    // V^T bank swizzle (round 6): thread (key, d quad) writes V^T[4 qd + c][key] -- with d / 4 = 8 or 16 quads per key the lanes of a wave that differ only in qd >> 1 hit the
    // same bank (VSTR = 4 mod 32: bank = 16 (qd & 1) + 4 c + key), a 4- / 8-way conflict on every one of the 16 scalar stores per thread and head: 60 % of this kernel's LDS-active
    // cycles were conflict re-issues (profiles/r06_lds_bank_conflicts.txt).  The 4-key block index of a row is XORed with (qd >> 1) & 7 (rows 8 s .. 8 s + 7 of V^T share s): two lanes per
    // bank, the minimum; keys stay contiguous inside a block, so the fragment reads remain 16-byte reads.  Needs whole 32-key groups: an even number of key tiles.
    const int vswz = (nkt & 1) ? 0 : 7;
    const int wswz = ((HPW * D) & 31) ? 0 : 7;                // ... and of the transposed proj slice Wt[column][k] (whole 32-k groups)
    float* const Ks = smem;                                   // [kcap][KSTR]
    float* const Vt = Ks + a.kcap * KSTR;                     // [D][VSTR]
    float* const Qs = Vt + D * VSTR;                          // [QR][KSTR]
    float* const Att = Qs + QR * KSTR;                        // [QR][AS]: normalised attention rows of the head group
    float* const Wt = Att + QR * AS;                          // [NC][WS]: proj slice, transposed (k contiguous)
    float* const Os = Wt + NC * WS;                           // [8 waves][16 queries][D + 4]: key-part partials (then proj partials)
    constexpr int OSF = 8 * 16 * (D + 4) > 8 * 16 * 20 + DEEP_FIN_FLOATS ? 8 * 16 * (D + 4) : 8 * 16 * 20 + DEEP_FIN_FLOATS;
    float* const ml = Os + OSF;                               // [8 waves][16 queries][2]: (m, l) of the key parts
    const int qt = QT == 2 ? (wave & 1) : 0, kp = QT == 2 ? (wave >> 1) : wave;      // this wave: query tile, key part
    const int KPS = 8 / QT;                                   // key parts (= the stride of a wave's key tiles)
    // ---- proj slice requested first: it depends on nothing (held in registers until the first barrier)
    const int kw = HPW * D, wq = NC >> 2;                     // K rows of the slice, column quads
    f32x4 wreg[4];                                            // (NC = 64: up to four quads per thread)
    const int wu_n = (kw * wq + DEEP_NTH - 1) / DEEP_NTH;     // (uniform)
    {
        const float* wb = a.Wp + (size_t)(hg * kw) * a.ldw + cg * NC;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = tid + DEEP_NTH * u;
            const int kr = e / wq, cq = e - kr * wq;
            if (u < 2 || u < wu_n) wreg[u] = *reinterpret_cast<const f32x4*>(wb + (size_t)(kr < kw ? kr : 0) * a.ldw + 4 * cq);
            else wreg[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // epilogue operands of head group 0 (bias + the block's input as residual): requested now, they land under the attention
    constexpr int MAXI = (128 * QPR + DEEP_NTH - 1) / DEEP_NTH;        // K/V items (key, quad) per thread at 128 keys
    const int nq4 = NC >> 2;
    // (RAW registers, added up in the epilogue: using a loaded value here would make the wave wait before it requests K / V)
    f32x4 pre_b = {0.f, 0.f, 0.f, 0.f}, pre_r = {0.f, 0.f, 0.f, 0.f};
    const bool pre_ok = hg == 0 && tid < QR * nq4 && q0 + tid / nq4 < L;
    if (pre_ok) {
        const int rr = tid / nq4, n = cg * NC + 4 * (tid - rr * nq4);
        pre_b = *reinterpret_cast<const f32x4*>(a.bias + n);
        pre_r = *reinterpret_cast<const f32x4*>(a.res.p + ((size_t)b * L + q0 + rr) * a.res.C + n);      // slab 0; further slabs in the epilogue
    }
    // K / V / Q of a head: global -> registers (issue) -> LDS (park); the next head's are requested while this one is computed
    f32x4 kreg[MAXI], vreg[MAXI], qregl;
    const int nitem = nkt * 16 * QPR;
    auto issue = [&](int hh) {
        const float* base = a.qkv + (size_t)b * L * 3 * C + (size_t)(hg * HPW + hh) * 3 * D;
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int e = tid + DEEP_NTH * u;
            if (DEEP_NTH * u >= nitem) break;                            // (uniform)
            const int key = e / QPR, qd = e - key * QPR;
            const float* p = base + (size_t)(k0 + (key < nk ? key : 0)) * 3 * C + D + 4 * qd;
            kreg[u] = *reinterpret_cast<const f32x4*>(p);
            vreg[u] = *reinterpret_cast<const f32x4*>(p + D);
        }
        if (tid < QR * QPR) {
            const int qr = tid / QPR, qd = tid - qr * QPR;
            const int tok = q0 + qr;
            qregl = *reinterpret_cast<const f32x4*>(base + (size_t)(tok < L ? tok : 0) * 3 * C + 4 * qd);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int u = 0; u < MAXI; ++u) {
            const int e = tid + DEEP_NTH * u;
            if (DEEP_NTH * u >= nitem) break;
            if (e < nitem) {
                const int key = e / QPR, qd = e - key * QPR;
                const bool in = key < nk;                                // rows past the last key of a tile are zero
                const f32x4 kv = in ? kreg[u] * a.scale : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(Ks + key * KSTR + 4 * qd) = kv;
                const int keys = key ^ (((qd >> 1) & vswz) << 2);
#pragma unroll
                for (int c = 0; c < 4; ++c) Vt[(4 * qd + c) * VSTR + keys] = in ? vreg[u][c] : 0.f;
            }
        }
        if (tid < QR * QPR) {
            const int qr = tid / QPR, qd = tid - qr * QPR;
            *reinterpret_cast<f32x4*>(Qs + qr * KSTR + 4 * qd) = qregl * (a.scale * LOG2E);
        }
    };
    static_assert(32 * QPR <= DEEP_NTH, "one Q quad per thread");
    issue(0);
    DEEP_STAMP(1);
    for (int hh = 0; hh < HPW; ++hh) {
        park();
        if (hh == 0) DEEP_STAMP(2);
        if (hh == 1) DEEP_STAMP(10);
        if (hh + 1 < HPW) issue(hh + 1);
        if (hh == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = tid + DEEP_NTH * u;
                const int kr = e / wq, cq = e - kr * wq;
                if (kr < kw && (u < 2 || u < wu_n)) {
                    const int krs = kr ^ (((cq >> 1) & wswz) << 2);              // (the V^T swizzle, for the same transposing store: row stride = 4 mod 32)
#pragma unroll
                    for (int c = 0; c < 4; ++c) Wt[(4 * cq + c) * WS + krs] = wreg[u][c];
                }
            }
        }
        __syncthreads();
        if (hh == 0) DEEP_STAMP(3);
        if (hh == 1) DEEP_STAMP(11);
        // ---- this wave: query tile qt, key tiles kp, kp + KPS (nkt <= 8: with one query tile per workgroup the second is always past the end)
        float qreg[NDT][4];
#pragma unroll
        for (int u = 0; u < NDT; ++u) {
            const f32x4 tq = *reinterpret_cast<const f32x4*>(Qs + (16 * qt + j) * KSTR + 16 * u + 4 * g);
            qreg[u][0] = tq[0]; qreg[u][1] = tq[1]; qreg[u][2] = tq[2]; qreg[u][3] = tq[3];
        }
        const int qtok = q0 + 16 * qt + j;                     // this lane's query (column j)
        const int qpl = qtok >= b2 ? 2 : (qtok >= b1 ? 1 : 0);
        f32x4 st[2];
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int kt = kp + KPS * w;
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
            if (kt < nkt) {
#pragma unroll
                for (int u = 0; u < NDT; ++u) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (16 * kt + j) * KSTR + 16 * u + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qreg[u][e], sacc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int key = k0 + 16 * kt + 4 * g + rr;
                const int kpl = key >= b2 ? 2 : (key >= b1 ? 1 : 0);
                const bool dead = kt >= nkt || key >= k1 || (!a.whole && kpl != qpl);
                sacc[rr] = dead ? -INFINITY : sacc[rr];
                m = fmaxf(m, sacc[rr]);
            }
            st[w] = sacc;
        }
        m = deep_swap_max16(m);
        m = deep_swap_max32(m);
        const bool live = m != -INFINITY;                    // (a key part may hold only keys of other planes)
        float lsum = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float pz = live ? __builtin_amdgcn_exp2f(st[w][rr] - m) : 0.f;
                st[w][rr] = pz;
                lsum += pz;
            }
        lsum += __shfl_xor(lsum, 16);
        lsum += __shfl_xor(lsum, 32);
        f32x4 oacc[NDT];
#pragma unroll
        for (int o = 0; o < NDT; ++o) oacc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int kt = kp + KPS * w;
            if (kt < nkt) {
#pragma unroll
                for (int o = 0; o < NDT; ++o) {
                    const f32x4 vf = *reinterpret_cast<const f32x4*>(Vt + (16 * o + j) * VSTR + 4 * ((4 * kt + g) ^ ((2 * o + (j >> 3)) & vswz)));
#pragma unroll
                    for (int sI = 0; sI < 4; ++sI) oacc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[sI], st[w][sI], oacc[o], 0, 0, 0);
                }
            }
        }
        // park (m, l, O^T) of this key part: O^T lane (query j, group g) reg r = d index 16 o + 4 g + r
        {
            float* op = Os + ((size_t)wave * 16 + j) * (D + 4) + 4 * g;
#pragma unroll
            for (int o = 0; o < NDT; ++o) *reinterpret_cast<f32x4*>(op + 16 * o) = oacc[o];
            if (g == 0) { ml[(wave * 16 + j) * 2] = m; ml[(wave * 16 + j) * 2 + 1] = lsum; }
        }
        if (hh == 0) DEEP_STAMP(4);
        if (hh == 1) DEEP_STAMP(12);
        __syncthreads();
        if (hh == 0) DEEP_STAMP(5);
        if (hh == 1) DEEP_STAMP(13);
        // ---- merge the KPS key parts of every query: thread -> (query row, d quad)
        for (int e = tid; e < QR * QPR; e += DEEP_NTH) {
            const int qr = e / QPR, dq = e - qr * QPR;
            const int tq = qr >> 4, jq = qr & 15;
            float mm[8], ll[8], M = -INFINITY;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                const int w = tq + QT * (k2 < KPS ? k2 : 0);
                mm[k2] = k2 < KPS ? ml[(w * 16 + jq) * 2] : -INFINITY;
                ll[k2] = k2 < KPS ? ml[(w * 16 + jq) * 2 + 1] : 0.f;
                M = fmaxf(M, mm[k2]);
            }
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            float lt = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {
                if (k2 >= KPS) break;                                     // (uniform)
                const int w = tq + QT * k2;
                const float f = mm[k2] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mm[k2] - M);
                lt += ll[k2] * f;
                o += *reinterpret_cast<const f32x4*>(Os + ((size_t)w * 16 + jq) * (D + 4) + 4 * dq) * f;
            }
            *reinterpret_cast<f32x4*>(Att + qr * AS + hh * D + 4 * dq) = o * (1.0f / lt);
        }
        __syncthreads();
    }
    DEEP_STAMP(6);
    // ---- proj: [QR rows][NC cols] = Att [QR][kw] x Wt^T; wave -> (tile, K part)
    const int nct = NC >> 4, ntile = QT * nct, kparts = 8 / ntile;      // (ntile 1, 2 or 4)
    const int tile = wave % ntile, kpart = wave / ntile;
    const int rt = QT == 2 ? (tile & 1) : 0, ct = QT == 2 ? (tile >> 1) : tile;
    const int kchunks = kw >> 4, cpp = (kchunks + kparts - 1) / kparts;  // 16-channel chunks per K part
    f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
    for (int cc = kpart * cpp; cc < (kpart + 1) * cpp && cc < kchunks; ++cc) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(Att + (16 * rt + j) * AS + 16 * cc + 4 * g);
        const f32x4 bf = *reinterpret_cast<const f32x4*>(Wt + (16 * ct + j) * WS + 4 * ((4 * cc + g) ^ ((2 * ct + (j >> 3)) & wswz)));
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[sI], bf[sI], pacc, 0, 0, 0);
    }
    {   // D lane (col j, group g) reg r = row 4 g + r -> [wave][row][col]
        float* pp = Os + (size_t)wave * 16 * 20 + (4 * g) * 20 + j;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) pp[rr * 20] = pacc[rr];
    }
    float* const fscratch = Os + 8 * 16 * 20;                // past the proj partials (the key-part partials that lay here are dead)
    if (tg && tid == 0) *reinterpret_cast<unsigned*>(fscratch) = (unsigned)(early >> __builtin_ctz(a.nhg)) + 1u;
    if (tg && hg == 0) {
        double* st = reinterpret_cast<double*>(fscratch + 4);
        for (int e = tid; e < a.fin.nstat * 96 * 2; e += DEEP_NTH) st[e] = 0.0;
    }
    __syncthreads();
    DEEP_STAMP(7);
    const unsigned epoch = tg ? *reinterpret_cast<const unsigned*>(fscratch) : 0u;
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(a.fin.gran, 0, tg ? (int)a.fin.gran_bytes : 0, 0x00020000);
    const unsigned slice_gran = (unsigned)((size_t)a.B * L * C);
    // epilogue: thread -> (row, column quad); K parts summed in order; head group 0 adds bias + residual
    for (int e = tid; e < QR * (NC >> 2); e += DEEP_NTH) {
        const int rr = e / (NC >> 2), cq = e - rr * (NC >> 2);
        const int tok = q0 + rr;
        if (tok >= L) continue;
        const int t2 = (rr >> 4) + QT * ((4 * cq) >> 4);       // tile of this quad
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int k2 = 0; k2 < kparts; ++k2)
            v += *reinterpret_cast<const f32x4*>(Os + (size_t)(t2 + ntile * k2) * 16 * 20 + (rr & 15) * 20 + ((4 * cq) & 15));
        const int n = cg * NC + 4 * cq;
        if (hg == 0) {                                          // (e == tid: 32 NC / 4 <= 512 quads, one per thread)
            v += pre_b + pre_r;
            for (int k2 = 1; k2 < a.res.ks; ++k2) v += *reinterpret_cast<const f32x4*>(a.res.p + (size_t)k2 * a.res.slab_stride + ((size_t)b * L + tok) * a.res.C + n);
        }
        const size_t qoff = ((size_t)b * L + tok) * C + n;
        float* dst = a.out + (size_t)hg * a.out_slab_stride + qoff;
        if (a.fin.out && !tg) deep_park_quad(dst, v);
        else mtv_store_out4(a.out + (size_t)hg * a.out_slab_stride, qoff, v);
        if (tg) {
            if (hg) {
                deep_put_granules(grs, (unsigned)(((size_t)(hg - 1) * slice_gran + qoff) * 8), v, epoch);
            } else {
                deep_add_granules(grs, (unsigned)qoff, slice_gran, a.nhg, epoch, a.fin.fault, v);
                mtv_store_out4(a.fin.out, qoff, v);
                deep_fin_stat(a.fin, fscratch, tok, n, v);
            }
        }
    }
    if (tg && hg == 0) deep_fin_flush(a.fin, fscratch, b, tid);
    if (a.fin.out && !tg) {
        // ---- in-launch completion over the head groups (the K slices of the projection)
        float* scratch = Os + 8 * 16 * 20;                       // past the proj partials
        if (deep_fin_arrive(a.fin, (b * a.nqg + qg) * a.ncg + cg, a.nhg, scratch, tid)) {
            for (int e = tid; e < QR * (NC >> 2); e += DEEP_NTH) {
                const int rr = e / (NC >> 2), cq = e - rr * (NC >> 2);
                const int tok = q0 + rr, n = cg * NC + 4 * cq;
                if (tok >= L) continue;
                const size_t off = ((size_t)b * L + tok) * C + n;
                const f32x4 v = deep_gather_quad(a.out + off, a.out_slab_stride, a.nhg);
                mtv_store_out4(a.fin.out, off, v);
                deep_fin_stat(a.fin, scratch, tok, n, v);
            }
            deep_fin_flush(a.fin, scratch, b, tid);
        }
    }
    DEEP_STAMP(9);
}


// =====================================================================================
// k_conv_win: 3x3 conv of the LARGE levels (512 / 2048 tokens) with the transformed input window of a row tile staged in LDS
// =====================================================================================
// Same arguments, same results layout and same statistics hand-over as k_conv (conv.hip: ConvArgs; GroupNorm statistics of the
// input come from the producers' site table, those of the output go to the consumers'), one difference in the K loop: the
// 16 MT output rows of a workgroup and their 3x3 halo -- a contiguous range of source tokens, at most 16 MT + 2 r + 2 rows --
// are normalised / FiLM-ed / SiLU-ed ONCE and parked in LDS, and all nine taps read their A fragments from there
// (BASELINE.json north_star: "LDS-staged input tiles").  k_conv transforms every element once per tap AND per column tile in
// the K loop, on the VALU that the exact-f32 MFMA blocks (conv.hip section 3.1: K loop 630 us per step for 364 us of MFMA).
// Workgroup = (clip, row tile, column tile of 16 NT channels), 512 threads; the 8 waves split K (chunk c -> wave c mod 8), B
// fragments straight from k_conv's weight layout (lane (j, q) loads W[c0 + 4q + s][n0 + NT j ..+NT-1]: output channel
// n0 + NT j + nb sits at lane j of column block nb), summed in wave order through LDS.
// Source-token window of the row tile that starts at output token tok0 (a superset of what its taps touch, one contiguous range:
// planes are contiguous and a 3x3 tap moves at most one image row + one token).  Host (LDS sizing) and device use the same code.
__host__ __device__ inline void conv_win_window(int r, int t, bool up, int Lout, int tok0, int rows, int* lo, int* n) {
    const int b1 = r * r, b2 = b1 + t * r, L = b2 + t * r;
    const int first = tok0, last = (tok0 + rows < Lout ? tok0 + rows : Lout) - 1;
    const int pf = first >= b2 ? 2 : (first >= b1 ? 1 : 0), pl = last >= b2 ? 2 : (last >= b1 ? 1 : 0);
    const int sf = pf == 0 ? 0 : (pf == 1 ? b1 : b2), el = pl == 0 ? b1 : (pl == 1 ? b2 : L);
    if (!up) {
        const int a0 = first - r - 1, a1 = last + r + 1;
        const int wlo = a0 > sf ? a0 : sf, whi = a1 < el - 1 ? a1 : el - 1;
        *lo = wlo;
        *n = whi - wlo + 1;
        return;
    }
    const int rs = r >> 1, ts = t >> 1, b1s = rs * rs, b2s = b1s + ts * rs;
    const int hl = pl == 0 ? r : t;
    const int sl = pl == 0 ? 0 : (pl == 1 ? b1 : b2);
    int yf = (first - sf) / r - 1, yl = (last - sl) / r + 1;
    yf = yf < 0 ? 0 : yf;
    yl = yl > hl - 1 ? hl - 1 : yl;
    const int of = pf == 0 ? 0 : (pf == 1 ? b1s : b2s), ol = pl == 0 ? 0 : (pl == 1 ? b1s : b2s);
    const int wlo = of + (yf >> 1) * rs, whi = ol + (yl >> 1) * rs + rs - 1;
    *lo = wlo;
    *n = whi - wlo + 1;
}

template <int MT, int NT>
__global__ __launch_bounds__(DEEP_NTH) void k_conv_win(const ConvArgs a) {
    touch_kernargs<(int)sizeof(ConvArgs)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT, COLS = 16 * NT;
    constexpr int G = MT * NT <= 2 ? 5 : (MT * NT == 8 ? 2 : 3);       // weight chunks per register set (two sets: 2 G chunks of look-ahead per wave; 2 x 4: 64 of its registers)
    constexpr int MAXR = 12;                               // window rows in flight per thread (first pass)
    static_assert(MAXR >= 2 * G, "the first 2 G weight chunks are requested between the rows of the first pass");
    typedef float bvec __attribute__((ext_vector_type(NT)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = deep_usgpr(tid >> 6);
    const int i = lane & 15, q = lane >> 4;
    // block -> (column tile, clip, row tile): column tiles of one row tile are 1 / tiles_n of the grid apart, so with a multiple of 8
    // row tiles they share an XCD (and the window they all stage)
    // xmap 1 / 2 (ConvTile::XM = 1; weight-dominated shapes): workgroup id mod 8 = the XCD selects the column tile, so a weight column
    // tile is fetched by ONE L2 (tiles_n a multiple of 8: Sx column tiles per XCD) or by 8 / tiles_n of them (tiles_n = 2^Sx < 8) and
    // the (smaller) activation windows are what the XCDs re-read.  Speed only.
    DEEP_STAMP(0);
    // K slices (ConvTile::KS = 2 / 4, round 6): slice z = the z-th 1 / KS of the input channels, all nine taps, and the z-th 1 / KS of the
    // fused skip conv's channels -- its window and raw rows hold only those channels -- and the slices of a tile are Bt x tiles_n workgroups apart (each slice keeps the
    // block order below).  Partial tiles meet in the slab exactly as k_conv's (write-through stores, one ticket per tile, the last slice
    // sums them in slice order and runs the epilogue).
    const int KS = deep_usgpr(a.KS);
    int blk = deep_usgpr((int)blockIdx.x);
    int z = 0;
    if (KS > 1) {
        z = deep_usgpr(FDiv{a.inv_tiles_n}(blk, a.Bt * a.tiles_n));
        blk -= z * a.Bt * a.tiles_n;
    }
    int ct, brt;
    if (a.xmap == 0) {
        ct = FDiv{a.inv_Bt}(blk, a.Bt);
        brt = blk - ct * a.Bt;
    } else if (a.xmap == 1) {
        const int xcd = blk & 7, j = blk >> 3;
        brt = FDiv{a.inv_Sx}(j, a.Sx);
        ct = xcd + 8 * (j - brt * a.Sx);
    } else if (a.xmap == 2) {
        const int xcd = blk & 7, j = blk >> 3;
        ct = xcd & ((1 << a.Sx) - 1);
        brt = (j << (3 - a.Sx)) + (xcd >> a.Sx);
    } else {                                               // xmap 3: XCD x owns the row tiles [x Sx, (x + 1) Sx) -- a contiguous eighth of the tokens, whatever the tile height
        const int xcd = blk & 7, j = blk >> 3;
        ct = FDiv{a.inv_Sx}(j, a.Sx);
        brt = xcd * a.Sx + (j - ct * a.Sx);
    }
    ct = deep_usgpr(ct);
    brt = deep_usgpr(brt);
    const int b = deep_usgpr(FDiv{a.inv_tiles_per_b}(brt, a.tiles_per_b));
    const int tok0 = (brt - b * a.tiles_per_b) * ROWS, n0 = ct * COLS;
    const int Cmain = a.Cmain, Cskip = a.Cskip;
    const int Cs = a.cps_r, coff = z * Cs;                 // this slice's channels [coff, coff + Cs) (Cs = Cmain without K slices)
    const int Css = a.cps_q, soff = z * Css;               // ... and of the fused skip conv's channels [soff, soff + Css)
    const int SW = Cs + DEEP_PAD, SK = Css + DEEP_PAD;
    DEEP_STAMP(1);
    const int wcap = a.rec_cap;                            // window capacity in rows (host); row wcap of the window = zeros
    float* const lwin = smem;                              // [wcap + 1][SW]
    float* const lraw = lwin + (wcap + 1) * SW;            // [ROWS + 1][SK]: raw rows of the fused 1x1 skip conv
    int* const idx = reinterpret_cast<int*>(lraw + (Cskip ? (ROWS + 1) * SK : 0));       // [9][ROWS] float offsets into lwin | [ROWS] into lraw
    double* const sdp = reinterpret_cast<double*>(idx + 10 * ROWS);                    // [96][2] input statistics, copies added up
    float2* const s_mr = reinterpret_cast<float2*>(sdp + 192);                         // [3][32] (mean, rstd)
    const bool do_gn = a.gn.sums != nullptr;
    int wlo, wn;
    conv_win_window(a.geo_r, a.geo_t, a.geo_main == 2, a.Lout, tok0, ROWS, &wlo, &wn);
    wn = wn < wcap ? wn : wcap;                            // (<= wcap by construction: conv_win_layout takes the maximum over the tiles)
    // ---- every request of the launch, oldest first = needed first: input statistics, GroupNorm vectors of this thread's channel quad,
    // its window rows (RAW registers: nothing loaded is used before the weights are requested), raw skip rows, weights
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 vraw[STAT_COPIES];
    if (do_gn && tid < 96) {
#pragma unroll
        for (int k = 0; k < STAT_COPIES; ++k) vraw[k] = *reinterpret_cast<const f64x2*>(a.gn.sums + (size_t)k * a.gn.cstride + (size_t)b * 192 + (size_t)tid * 2);
    }
    const int QW = Cs >> 2, RP = DEEP_NTH / QW;            // staging map: thread -> quad column qd of rows rl, rl + RP, ...
    const int qd = tid % QW, rl = tid / QW;
    const bool stager = rl < RP;
    const int c4l = 4 * qd, c4 = coff + c4l, C0 = a.C[0];  // c4l: column in the window, c4: channel of the conv's input
    const float* film = (do_gn && a.gn.film) ? a.gn.film + (size_t)b * a.gn.film_stride : nullptr;
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = ga, f1 = ga, f2 = ga;
    if (do_gn && stager) {
        ga = *reinterpret_cast<const f32x4*>(a.gn.gamma + c4);
        be = *reinterpret_cast<const f32x4*>(a.gn.beta + c4);
        f1 = *reinterpret_cast<const f32x4*>((film ? film : a.gn.gamma) + c4);           // (selected at use, not at load: no early wait)
        f2 = *reinterpret_cast<const f32x4*>((film ? film + Cmain : a.gn.beta) + c4);
    }
    const float* xcol = c4 < C0 ? a.src[0] + ((size_t)b * a.Lsrc + wlo) * C0 + c4 : a.src[1] + ((size_t)b * a.Lsrc + wlo) * a.C[1] + (c4 - C0);
    const int xstride = c4 < C0 ? C0 : a.C[1];
    f32x4 xr[MAXR];
    if (stager) {
#pragma unroll
        for (int u = 0; u < MAXR; ++u) {
            if (u * RP >= wn) break;                                  // (uniform)
            const int row = rl + u * RP;
            xr[u] = *reinterpret_cast<const f32x4*>(xcol + (size_t)(row < wn ? row : 0) * xstride);
        }
    }
    constexpr int SRN = 4;                                 // raw skip quads in flight per thread (first pass: 16 MT rows x up to 512 / MT channels)
    f32x4 sraw[SRN];
    const int QS = Css >> 2, C2 = a.C[2];
    auto skip_ptr = [&](int e) {
        const int row = e / QS, c = soff + 4 * (e - row * QS);
        return c < C2 ? a.src[2] + ((size_t)b * a.Lskip + tok0 + row) * C2 + c : a.src[3] + ((size_t)b * a.Lskip + tok0 + row) * a.C[3] + (c - C2);
    };
    if (Cskip) {
#pragma unroll
        for (int u = 0; u < SRN; ++u) {
            const int e = tid + u * DEEP_NTH;
            if (u * DEEP_NTH < ROWS * QS)                          // (uniform)
                sraw[u] = *reinterpret_cast<const f32x4*>(skip_ptr(e < ROWS * QS && tok0 + e / QS < a.Lout ? e : 0));
        }
    }
    const int cpt = a.cpt, nmain_ch = 9 * cpt, nch = nmain_ch + (Css >> 4);
    const int n_it = (nch + 7) >> 3;
    const int wrows = 9 * Cmain + Cskip;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W), 0, wrows * a.ldw * 4, 0x00020000);
    const int wlane = ((4 * q) * a.ldw + n0 + NT * i) * 4;
    bvec bq[2][G][4];
    auto wload1 = [&](bvec (&dstg)[4], int kc) {
        {
            const int c = wave + 8 * kc;
            // W row of the chunk's first channel: tap-major over the concatenated main channels, then the skip channels (k_conv's
            // order of rows); a chunk past the end reads out of range = zeros
            int krow = 9 * Cmain + soff + (c - nmain_ch) * 16;
            if (c < nmain_ch) { const int tap = FDiv{a.inv_cpt}(c, cpt); krow = tap * Cmain + coff + (c - tap * cpt) * 16; }
            const int soff = c < nch ? krow * a.ldw * 4 : 0x7F000000;
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                if constexpr (NT == 4) dstg[sI] = __builtin_bit_cast(bvec, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + sI * a.ldw * 4, soff, 0));
                else dstg[sI] = __builtin_bit_cast(bvec, __builtin_amdgcn_raw_buffer_load_b64(wrsrc, wlane + sI * a.ldw * 4, soff, 0));
            }
        }
    };
    auto wload = [&](bvec (&dst)[G][4], int k0) {
#pragma unroll
        for (int g = 0; g < G; ++g) wload1(dst[g], k0 + g);
    };
#ifndef MTV_WIN_WORDER
#define MTV_WIN_WORDER 0
#endif
    // MTV_WIN_WORDER (experiment): 0 = weights requested between the rows of the transform, 1 = all 2 G chunks here, 2 = G here, G below
    constexpr int W_EARLY = MTV_WIN_WORDER == 1 ? 2 * G : (MTV_WIN_WORDER == 2 ? G : 0);
#pragma unroll
    for (int u = 0; u < W_EARLY; ++u) wload1(bq[u / G][u % G], u);
    DEEP_STAMP(2);
    // ---- while they fly: row table (LDS float offsets: window row of every (tap, output row), the zero row for padding), zero rows
    for (int e = tid; e < 9 * ROWS; e += DEEP_NTH) {
        const int tap = e / ROWS, ri = e - tap * ROWS;
        const int tok = tok0 + ri;
        int src = -1;
        if (tok < a.Lout) {
            const int ky = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0);
            const int g = geo_source_t<FDiv>(FDiv{a.geo_inv_r}, a.geo_r, a.geo_t, tok, ky, tap - 3 * ky, a.geo_main == 2);
            src = g < 0 ? -1 : (g & 0x0FFFFFFF) - wlo;
        }
        idx[e] = (src < 0 || src >= wn) ? wcap * SW : src * SW;
    }
    for (int e = tid; e < ROWS; e += DEEP_NTH) idx[9 * ROWS + e] = tok0 + e < a.Lout ? e * SK : ROWS * SK;
    for (int e = tid; e < Cs; e += DEEP_NTH) lwin[wcap * SW + e] = 0.f;
    if (Cskip)
        for (int e = tid; e < Css; e += DEEP_NTH) lraw[ROWS * SK + e] = 0.f;
    if (do_gn && tid < 96) {                                       // input statistics: the 8 copies added up (first use of a loaded value)
        f64x2 v0 = vraw[0];
#pragma unroll
        for (int k = 1; k < STAT_COPIES; ++k) v0 += vraw[k];
        if (a.gn.whole) {
            sdp[2 * tid] = v0[0];
            sdp[2 * tid + 1] = v0[1];
        } else {                                                   // per-plane statistics: (mean, rstd) of this (plane, group) right here
            const int sg = tid >> 5;
            const double n0 = a.gn.inv_n[0], n1 = a.gn.inv_n[1], n2 = a.gn.inv_n[2];     // (scalar loads; an indexed read becomes a vector load + vmcnt(0))
            const double inv_n = sg == 0 ? n0 : (sg == 1 ? n1 : n2);
            const double mean = v0[0] * inv_n;
            double var = v0[1] * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mr[tid] = make_float2((float)mean, 1.0f / sqrtf((float)var + 1e-5f));
        }
    }
    __syncthreads();
    DEEP_STAMP(3);
    if (do_gn && a.gn.whole) {                                     // whole-L statistics (not a resblock conv): the three planes added up
        if (tid < 96) {
            const int g = tid & 31;
            const double sx = (sdp[2 * g] + sdp[2 * (32 + g)]) + sdp[2 * (64 + g)];
            const double sy = (sdp[2 * g + 1] + sdp[2 * (32 + g) + 1]) + sdp[2 * (64 + g) + 1];
            const double inv_n = a.gn.inv_n[3];
            const double mean = sx * inv_n;
            double var = sy * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mr[tid] = make_float2((float)mean, 1.0f / sqrtf((float)var + 1e-5f));
        }
        __syncthreads();
    }
    DEEP_STAMP(4);
    // ---- transform this thread's window rows ONCE (y = x A + B, SiLU) and park them; then the raw skip rows
    DEEP_STAMP(4);
    {
        const bool act = a.gn.act != 0;
        const SegInfo ss = a.seg_src;
        int cur = -1;
        f32x4 A = {1.f, 1.f, 1.f, 1.f}, Bc = {0.f, 0.f, 0.f, 0.f};
        auto transform = [&](int row, f32x4 y) {
            if (do_gn) {
                const int tk = wlo + row;
                const int sg = tk >= ss.b2 ? 2 : (tk >= ss.b1 ? 1 : 0);
                if (sg != cur) {                                   // (rows ascend: at most three times)
                    cur = sg;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float2 mr = s_mr[sg * 32 + FDiv{a.gn.inv_gs}(c4 + k, a.gn.gs)];
                        const float sc = mr.y * ga[k];
                        const float bi = be[k] - sc * mr.x;
                        const float s1 = film ? 1.0f + f1[k] : 1.0f, sh = film ? f2[k] : 0.f;
                        A[k] = sc * s1;
                        Bc[k] = fmaf(bi, s1, sh);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = fmaf(y[k], A[k], Bc[k]);
                    y[k] = act ? deep_silu(t) : t;
                }
            }
            *reinterpret_cast<f32x4*>(lwin + row * SW + c4l) = y;
        };
#pragma unroll
        for (int u = 0; u < MAXR; ++u) {
            if (u + W_EARLY < 2 * G) wload1(bq[(u + W_EARLY) / G][(u + W_EARLY) % G], u + W_EARLY);       // (every thread; chunk indices past the end read as zeros)
            if (stager && u * RP < wn) {                          // (uniform but for the stager test)
                const int row = rl + u * RP;
                if (row < wn) transform(row, xr[u]);
            }
        }
        if (stager)
            for (int row = rl + MAXR * RP; row < wn; row += RP)   // (windows taller than MAXR RP rows: requested late, behind the weights)
                transform(row, *reinterpret_cast<const f32x4*>(xcol + (size_t)row * xstride));
    }
    if (Cskip) {
#pragma unroll
        for (int u = 0; u < SRN; ++u) {
            const int e = tid + u * DEEP_NTH;
            if (u * DEEP_NTH < ROWS * QS && e < ROWS * QS) {
                const int row = e / QS, c = 4 * (e - row * QS);
                if (tok0 + row < a.Lout) *reinterpret_cast<f32x4*>(lraw + row * SK + c) = sraw[u];
            }
        }
        for (int e = tid + SRN * DEEP_NTH; e < ROWS * QS; e += DEEP_NTH) {       // (wider skips: requested late, behind the weights)
            const int row = e / QS, c = 4 * (e - row * QS);
            if (tok0 + row < a.Lout) *reinterpret_cast<f32x4*>(lraw + row * SK + c) = *reinterpret_cast<const f32x4*>(skip_ptr(e));
        }
    }
    // ---- epilogue operands of this thread's output quad, requested BEFORE the K loop (round 6; they were requested after it, and the
    // residual -- written by an earlier launch, a first-touch miss like every read of a launch -- arrived after the partial tiles had been exchanged:
    // the wait sat on the launch's tail.  16 registers held across the loop.)
    constexpr int QPR = COLS / 4;
    static_assert(ROWS * QPR <= DEEP_NTH, "one output quad per thread");
    const int e_rr = tid / QPR, e_cq = tid - e_rr * QPR;
    const int e_tok = tok0 + e_rr, e_n = n0 + 4 * e_cq;
    const bool e_on = tid < ROWS * QPR && e_tok < a.Lout;
    f32x4 e_add = {0.f, 0.f, 0.f, 0.f}, e_b2 = e_add, e_bb = e_add, e_res = e_add;
    if (e_on) {
        e_add = *reinterpret_cast<const f32x4*>(a.bias + e_n);
        if (a.bias2) e_b2 = *reinterpret_cast<const f32x4*>(a.bias2 + e_n);
        if (a.bias_b) e_bb = *reinterpret_cast<const f32x4*>(a.bias_b + (size_t)b * a.bias_b_stride + e_n);
        if (a.res) {
            const int rs = a.geo_skip ? (geo_source_t<FDiv>(FDiv{a.geo_inv_r}, a.geo_r, a.geo_t, e_tok, 1, 1, true) & 0x0FFFFFFF) : e_tok;
            e_res = *reinterpret_cast<const f32x4*>(a.res + ((size_t)b * a.Lskip + rs) * a.N + e_n);
        }
    }
    __syncthreads();
    // ---- K loop: A fragments from the window (one chunk ahead), weights of iteration k + 2G replace those of k
    DEEP_STAMP(5);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[mt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        auto loadA = [&](int k, f32x4 (&av)[MT]) {
            int c = wave + 8 * k;
            c = c < nch ? c : nch - 1;
            const bool sk = c >= nmain_ch;
            const int tap = sk ? 9 : FDiv{a.inv_cpt}(c, cpt);
            const int cc = sk ? c - nmain_ch : c - tap * cpt;
            const float* ab = (sk ? lraw : lwin) + cc * 16 + 4 * q;
            const int* ip = idx + tap * ROWS + i;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(ab + ip[16 * mt]);
        };
        auto mma = [&](const f32x4 (&av)[MT], const bvec (&w)[4]) {
#pragma unroll
            for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][sI], w[sI][nb], acc[mt][nb], 0, 0, 0);
        };
        // (measured and dropped, profiles/r04_conv_win_stamps.txt: a two-step-deep A pipeline pinned with sched_barrier, 13.2 -> 14.3 us per
        // op on [2048 x 128] -- the barrier also keeps the next chunk's address arithmetic out of the MFMA block; the three planes'
        // coefficient sets hoisted out of the row loop, 13.2 -> 13.6: as an array they go to scratch, as named registers they cost more
        // than the 4 LDS reads per row they save)
        f32x4 avA[MT], avB[MT];
        auto step = [&](int kk, int par, const bvec (&w)[4]) {
            if (par) { loadA(kk + 1, avA); mma(avB, w); }
            else { loadA(kk + 1, avB); mma(avA, w); }
        };
        loadA(0, avA);
        int k = 0;
        for (; k + 2 * G <= n_it; k += 2 * G) {
#pragma unroll
            for (int g = 0; g < G; ++g) step(k + g, g & 1, bq[0][g]);
            wload(bq[0], k + 2 * G);
#pragma unroll
            for (int g = 0; g < G; ++g) step(k + G + g, (G + g) & 1, bq[1][g]);
            wload(bq[1], k + 3 * G);
        }
        const int rem = n_it - k;
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (g < rem) step(k + g, g & 1, bq[0][g]);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (G + g < rem) step(k + G + g, (G + g) & 1, bq[1][g]);
    }
    DEEP_STAMP(6);
    __syncthreads();
    // ---- the eight partial tiles -> (row, col) images (lane (i, q) holds, for row 4q + r, the NT consecutive columns NT i ..)
    constexpr int LDR = COLS + 4;
    float* const red = smem;
    {
        float* my = red + (size_t)wave * ROWS * LDR + (4 * q) * LDR + NT * i;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                bvec t;
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) t[nb] = acc[mt][nb][rr];
                *reinterpret_cast<bvec*>(my + (16 * mt + rr) * LDR) = t;
            }
    }
    __syncthreads();
    DEEP_STAMP(7);
    // ---- epilogue: bias / per-clip bias / residual, coalesced store, statistics of the output for its consumers
    float* scratch = red + 8 * ROWS * LDR;                   // statistics slots (deep_stat_one layout: 4 floats, then [2][96][2] doubles)
    {
        double* st = reinterpret_cast<double*>(scratch + 4);
        for (int e = tid; e < a.nstat * 96 * 2; e += DEEP_NTH) st[e] = 0.0;
        if (a.nstat) __syncthreads();
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (e_on) {                                               // (partials in wave order)
        const float* rp = red + e_rr * LDR + 4 * e_cq;
        v = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
        for (int w = 1; w < 8; ++w) v += *reinterpret_cast<const f32x4*>(rp + (size_t)w * ROWS * LDR);
    }
    if (KS > 1) {
        // cross-workgroup split-K completed inside the launch, k_conv's protocol (conv.hip; cdna_hip_programming.md G16, form R1): park the
        // partial tile with write-through 8-byte stores, drain, ONE ticket per workgroup; the workgroup that draws the last ticket of its
        // tile re-reads the KS partials past its L1 / L2 and sums them in slice order (the result does not depend on who is last)
        typedef __attribute__((address_space(1))) unsigned long long gu64;
        const size_t sstride = (size_t)a.B * a.Lout * a.N;
        const size_t eoff = ((size_t)b * a.Lout + e_tok) * a.N + e_n;
        if (e_on) {
            gu64* dst = (gu64*)(unsigned long long)(a.slab + (size_t)z * sstride + eoff);
            __hip_atomic_store(dst, ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, ((unsigned long long)__float_as_uint(v[3]) << 32) | __float_as_uint(v[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* const s_last = reinterpret_cast<int*>(scratch);     // (the flag word of the statistics scratch: dynamic LDS -- the launch may ask for all 160 KB)
        int* ticket = a.tickets + ((size_t)b * a.tiles_per_b + (tok0 / ROWS)) * a.tiles_n + ct;
        if (tid == 0) *s_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == KS - 1;
        __syncthreads();
        if (!*s_last) return;
        if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // self-cleaning for the next launch
        if (e_on) {
            gu64* src = (gu64*)(unsigned long long)(a.slab + eoff);
            unsigned long long t0[4], t1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = k < KS;
                t0[k] = in ? __hip_atomic_load(src + (size_t)k * (sstride / 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                t1[k] = in ? __hip_atomic_load(src + (size_t)k * (sstride / 2) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            }
            v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[0] += __uint_as_float((unsigned)t0[k]);
                v[1] += __uint_as_float((unsigned)(t0[k] >> 32));
                v[2] += __uint_as_float((unsigned)t1[k]);
                v[3] += __uint_as_float((unsigned)(t1[k] >> 32));
            }
        }
    }
    if (e_on) {                                               // (same order of additions as before: bias, bias2, per-clip bias, residual)
        v += e_add;
        if (a.bias2) v += e_b2;
        if (a.bias_b) v += e_bb;
        if (a.res) v += e_res;
        mtv_store_out4(a.out, ((size_t)b * a.Lout + e_tok) * a.N + e_n, v);
        if (a.nstat > 0) deep_stat_one(a.stat[0], 0, a.seg_out, scratch, e_tok, e_n, v);
        if (a.nstat > 1) deep_stat_one(a.stat[1], 1, a.seg_out, scratch, e_tok, e_n, v);
    }
    if (a.nstat) {
        __syncthreads();
        deep_stat_flush_one(a.stat[0], 0, a.stat_cstride, scratch, b, tid);
        if (a.nstat > 1) deep_stat_flush_one(a.stat[1], 1, a.stat_cstride, scratch, b, tid);
    }
    DEEP_STAMP(9);
}

// =====================================================================================
// k_conv_pw: 1x1 conv on identity rows at the LARGE levels (qkv / proj_out of the attention blocks at 512 / 2048 tokens)
// =====================================================================================
// Same arguments, results layout and statistics hand-over as k_conv (ConvArgs).  k_conv runs a 1x1 conv as a one-tap convolution with
// K = 128 .. 512 split over its 8 waves (one or two 16-channel chunks each, then an 8-wave reduction); k_lin gave every wave the whole K
// but re-derived the GroupNorm coefficients and re-transformed the rows per wave.  Here the 16 MT rows of a workgroup are normalised ONCE
// into LDS (k_conv_win's prologue: statistics -> (mean, rstd) through one barrier, per-thread coefficients of one channel quad), the 8
// waves sit side by side along N (16 NTW columns each, whole K, no reduction), weights come in the checkpoint's own [N][K] layout
// (ConvArgs::Wnk: lane (j, q) loads W[n][c0 + 4q .. + 3], one 16-byte request per 16-channel chunk and column block), requested
// between the rows of the transform, and the accumulators go through LDS once so that stores, residual reads and statistics are
// 16-byte wide.  Workgroup = (clip, row tile, group of 128 NTW columns), 512 threads.
// (round 6: NWA = the waves that multiply -- 8, 6, 4 or 2 -- so that the column tile is 16 NTW NWA wide and the grid can be made to FILL the chip:
// qkv at N = 384 / 768 / 1536 has 3 / 6 / 12 tiles of 128 columns -- 192 workgroups with 32- / 16-row tiles on 256 CUs -- but 4 / 8 / 16 tiles of 96;
// proj_out at N = 128 / 256 one or two tiles of 128 but 256 workgroups with 64- / 32-column tiles.  The idle waves still stage rows and store.
// tools/ubench/stage_bench.hip's lean qkv / proj bodies -- 256 workgroups each -- measured 6.8 / 5.1 us where these launches took 8.8 / 6.6-7.2)
template <int MT, int NTW, int NWA>
__global__ __launch_bounds__(DEEP_NTH) void k_conv_pw(const ConvArgs a) {
    touch_kernargs<(int)sizeof(ConvArgs)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ROWS = 16 * MT, COLS = 16 * NTW * NWA;
    constexpr int G = NTW == 1 ? 8 : 4;                    // weight chunks per register set (two sets: K = 256 is in flight whole at NTW = 1)
    constexpr int MAXR = 8;                                // rows in flight per thread
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = deep_usgpr(tid >> 6);
    const int i = lane & 15, q = lane >> 4;
    DEEP_STAMP(0);
    const int blk = deep_usgpr((int)blockIdx.x);
    // block -> (column group, row tile).  xmap 0: the column groups of one row tile are Bt workgroups apart -- with a multiple of 8 row tiles they share an
    // XCD and the rows they all stage, and every XCD reads the whole weight matrix.  xmap 1 (ConvTile::XM = 1, weight-dominated shapes: the qkv convs at 128
    // tokens re-fetched their 3.1 MB once per XCD, 25.8 MB per launch in profiles/r06_per_op_traffic.txt): workgroup id mod 8 = the XCD selects the column
    // group (k_conv_win's xmap 1), so a column group's weights go through ONE L2 and the (smaller) rows are what the XCDs re-read.  Speed only.
    int cg, brt;
    if (a.xmap == 0) {
        cg = FDiv{a.inv_Bt}(blk, a.Bt);
        brt = blk - cg * a.Bt;
    } else if (a.xmap == 1) {
        const int xcd = blk & 7, j = blk >> 3;
        brt = FDiv{a.inv_Sx}(j, a.Sx);
        cg = xcd + 8 * (j - brt * a.Sx);
    } else {                                               // xmap 3: XCD x owns the row tiles [x Sx, (x + 1) Sx) (k_conv_win's xmap 3)
        const int xcd = blk & 7, j = blk >> 3;
        cg = FDiv{a.inv_Sx}(j, a.Sx);
        brt = xcd * a.Sx + (j - cg * a.Sx);
    }
    cg = deep_usgpr(cg);
    brt = deep_usgpr(brt);
    const int b = deep_usgpr(FDiv{a.inv_tiles_per_b}(brt, a.tiles_per_b));
    const int tok0 = (brt - b * a.tiles_per_b) * ROWS, n0 = cg * COLS;
    const int K = a.Cmain, SA = K + DEEP_PAD;
    float* const lA = smem;                                // [ROWS][SA]
    double* const sdp = reinterpret_cast<double*>(lA + ROWS * SA);      // [96][2] whole-L statistics only
    float2* const s_mr = reinterpret_cast<float2*>(sdp + 192);         // [3][32] (mean, rstd)
    const bool do_gn = a.gn.sums != nullptr;
    DEEP_STAMP(1);
    // ---- requests, oldest first = needed first: input statistics, GroupNorm vectors of this thread's channel quad, its rows (raw)
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 vraw[STAT_COPIES];
    if (do_gn && tid < 96) {
#pragma unroll
        for (int k = 0; k < STAT_COPIES; ++k) vraw[k] = *reinterpret_cast<const f64x2*>(a.gn.sums + (size_t)k * a.gn.cstride + (size_t)b * 192 + (size_t)tid * 2);
    }
    const int QW = K >> 2, RP = DEEP_NTH / QW;             // staging map: thread -> quad column qd of rows rl, rl + RP, ...
    const int qd = tid % QW, rl = tid / QW;
    const bool stager = rl < RP;
    const int c4 = 4 * qd;
    const float* film = (do_gn && a.gn.film) ? a.gn.film + (size_t)b * a.gn.film_stride : nullptr;
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = ga, f1 = ga, f2 = ga;
    if (do_gn && stager) {
        ga = *reinterpret_cast<const f32x4*>(a.gn.gamma + c4);
        be = *reinterpret_cast<const f32x4*>(a.gn.beta + c4);
        f1 = *reinterpret_cast<const f32x4*>((film ? film : a.gn.gamma) + c4);
        f2 = *reinterpret_cast<const f32x4*>((film ? film + K : a.gn.beta) + c4);
    }
    const int nrows = a.Lout - tok0 < ROWS ? a.Lout - tok0 : ROWS;
    const float* xcol = a.src[0] + ((size_t)b * a.Lsrc + tok0) * K + c4;
    f32x4 xr[MAXR];
    if (stager) {
#pragma unroll
        for (int u = 0; u < MAXR; ++u)
            if (u * RP < ROWS) {                                       // (uniform)
                const int row = rl + u * RP;
                xr[u] = *reinterpret_cast<const f32x4*>(xcol + (size_t)(row < nrows ? row : 0) * K);
            }
    }
    // weights: wave w owns the NTW column blocks of 16 that start at n0 + 16 NTW w; lane (j, q) holds, of column block nb, output
    // channel nw0 + 16 nb + j
    const int nw0 = n0 + 16 * NTW * wave;
    const bool wave_on = wave < NWA && nw0 < a.N;          // (N is a multiple of 16 NTW: a wave is all in or all out)
    const int nch = K >> 4;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wpk), 0, a.N * K * 4, 0x00020000);
    // (round 6: the weights come lane-linear -- ConvArgs::Wpk, [N / 16][K / 16][64 lanes][4] -- so a wave's B fragment of a chunk is ONE contiguous KB.  Read from the
    // checkpoint's [N][K] rows, lane (j, q) fetched 16 bytes of row j: the 16 lanes of a quarter-wave hit 16 different lines and every line was touched by four
    // quarter-waves -- 64 tag look-ups per instruction instead of 16, and the 8 ... 16 weight requests of a wave stood in front of the K loop for ~50-100 cycles each:
    // profiles/r06_conv_pw_packed_weights.txt)
    const int wlane = ((nw0 >> 4) * nch * 64 + lane) * 16;
    f32x4 bq[2][G][NTW];
    auto wload1 = [&](f32x4 (&dstg)[NTW], int c) {
        const int soff = (c < nch && wave_on) ? c * 1024 : 0x7F000000;      // (past the end: out of range = zeros)
#pragma unroll
        for (int nb = 0; nb < NTW; ++nb) dstg[nb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + nb * nch * 1024, soff, 0));
    };
    auto wload = [&](f32x4 (&dst)[G][NTW], int c0) {
#pragma unroll
        for (int g = 0; g < G; ++g) wload1(dst[g], c0 + g);
    };
    DEEP_STAMP(2);
    if (do_gn && tid < 96) {
        f64x2 v0 = vraw[0];
#pragma unroll
        for (int k = 1; k < STAT_COPIES; ++k) v0 += vraw[k];
        if (a.gn.whole) {
            sdp[2 * tid] = v0[0];
            sdp[2 * tid + 1] = v0[1];
        } else {
            const int sg = tid >> 5;
            const double i0 = a.gn.inv_n[0], i1 = a.gn.inv_n[1], i2 = a.gn.inv_n[2];
            const double inv_n = sg == 0 ? i0 : (sg == 1 ? i1 : i2);
            const double mean = v0[0] * inv_n;
            double var = v0[1] * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mr[tid] = make_float2((float)mean, 1.0f / sqrtf((float)var + 1e-5f));
        }
    }
    if (do_gn) __syncthreads();
    DEEP_STAMP(3);
    if (do_gn && a.gn.whole) {
        if (tid < 96) {
            const int g = tid & 31;
            const double sx = (sdp[2 * g] + sdp[2 * (32 + g)]) + sdp[2 * (64 + g)];
            const double sy = (sdp[2 * g + 1] + sdp[2 * (32 + g) + 1]) + sdp[2 * (64 + g) + 1];
            const double inv_n = a.gn.inv_n[3];
            const double mean = sx * inv_n;
            double var = sy * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mr[tid] = make_float2((float)mean, 1.0f / sqrtf((float)var + 1e-5f));
        }
        __syncthreads();
    }
    DEEP_STAMP(4);
    // ---- transform this thread's rows ONCE (y = x A + B, SiLU where the norm has one) and park them; weights go out in between
    {
        const bool act = a.gn.act != 0;
        const SegInfo ss = a.seg_src;
        int cur = -1;
        f32x4 A = {1.f, 1.f, 1.f, 1.f}, Bc = {0.f, 0.f, 0.f, 0.f};
        auto transform = [&](int row, f32x4 y) {
            if (do_gn) {
                const int tk = tok0 + row;
                const int sg = tk >= ss.b2 ? 2 : (tk >= ss.b1 ? 1 : 0);
                if (sg != cur) {
                    cur = sg;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float2 mr = s_mr[sg * 32 + FDiv{a.gn.inv_gs}(c4 + k, a.gn.gs)];
                        const float sc = mr.y * ga[k];
                        const float bi = be[k] - sc * mr.x;
                        const float s1 = film ? 1.0f + f1[k] : 1.0f, sh = film ? f2[k] : 0.f;
                        A[k] = sc * s1;
                        Bc[k] = fmaf(bi, s1, sh);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = fmaf(y[k], A[k], Bc[k]);
                    y[k] = act ? deep_silu(t) : t;
                }
            }
            *reinterpret_cast<f32x4*>(lA + row * SA + c4) = y;
        };
#pragma unroll
        for (int u = 0; u < (2 * G > MAXR ? 2 * G : MAXR); ++u) {
            // (every thread; chunks past K read as zeros without traffic.  Measured and dropped, profiles/r04_conv_pw_stamps.txt: all
            // 2 G chunks requested with the rows -- they end up inside the wait of the statistics' first use, 8.75 -> 11.2 us per launch at
            // [2048 x 384] -- or between the statistics barrier and the transform, 11.4 us: the tile then holds too many registers for two
            // workgroups per CU.  The ~5k cycles to "rows parked" are the round trip of the rows themselves: every kernel starts on a cold L2)
            if (u < 2 * G) wload1(bq[u / G][u % G], u);
            if (u < MAXR && stager && u * RP < ROWS) {
                const int row = rl + u * RP;
                if (row < ROWS) {
                    if (row < nrows) transform(row, xr[u]);
                    else *reinterpret_cast<f32x4*>(lA + row * SA + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (stager)
            for (int row = rl + MAXR * RP; row < ROWS; row += RP) {        // (K = 512 with 32 rows: requested late)
                if (row < nrows) transform(row, *reinterpret_cast<const f32x4*>(xcol + (size_t)row * K));
                else *reinterpret_cast<f32x4*>(lA + row * SA + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    DEEP_STAMP(10);
    // epilogue operands of this thread's first output quad: requested before the K loop's barrier (a round trip off the tail)
    constexpr int QPR = COLS / 4, NQ = ROWS * QPR, EPT = (NQ + DEEP_NTH - 1) / DEEP_NTH;
    // (the bias too -- round 6: read inside the epilogue loop it was a first-touch miss on the launch's tail: every kernel starts on invalidated caches)
    f32x4 e_res[EPT], e_bias[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        e_res[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        e_bias[u] = e_res[u];
        const int e = tid + u * DEEP_NTH;
        const int rr = e / QPR, cq = e - rr * QPR;
        if (e < NQ && tok0 + rr < a.Lout && n0 + 4 * cq < a.N) {
            e_bias[u] = *reinterpret_cast<const f32x4*>(a.bias + n0 + 4 * cq);
            if (a.res) e_res[u] = *reinterpret_cast<const f32x4*>(a.res + ((size_t)b * a.Lskip + tok0 + rr) * a.N + n0 + 4 * cq);
        }
    }
    DEEP_STAMP(11);
    __syncthreads();
    DEEP_STAMP(5);
    // ---- K loop: a wave multiplies the whole K for its 16 NTW columns; A fragments from the parked rows
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NTW; ++nb) acc[mt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (wave_on) {
        const float* ab = lA + i * SA + 4 * q;
        auto step = [&](int c, const f32x4 (&w)[NTW]) {
            f32x4 av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const f32x4*>(ab + 16 * mt * SA + c * 16);
#pragma unroll
            for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nb = 0; nb < NTW; ++nb) acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][sI], w[nb][sI], acc[mt][nb], 0, 0, 0);
        };
        int c = 0;
        for (; c + 2 * G <= nch; c += 2 * G) {
#pragma unroll
            for (int g = 0; g < G; ++g) step(c + g, bq[0][g]);
            if (c + 2 * G < nch) wload(bq[0], c + 2 * G);          // (uniform; K = 128 is one pass: nothing more to request)
#pragma unroll
            for (int g = 0; g < G; ++g) step(c + G + g, bq[1][g]);
            if (c + 3 * G < nch) wload(bq[1], c + 3 * G);
        }
        const int rem = nch - c;
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (g < rem) step(c + g, bq[0][g]);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (G + g < rem) step(c + G + g, bq[1][g]);
    }
    DEEP_STAMP(6);
    __syncthreads();                                               // (the parked rows are dead: their LDS becomes the output image)
    // ---- accumulators -> [ROWS][COLS + 4] image: lane (i, q) holds, for row 4q + r of row tile mt, channels NTW i .. NTW i + NTW - 1 of its wave
    constexpr int LDR = COLS + 4;
    float* const red = smem;
    if (wave_on) {
        float* my = red + (4 * q) * LDR + 16 * NTW * wave + i;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int nb = 0; nb < NTW; ++nb) my[(16 * mt + rr) * LDR + 16 * nb] = acc[mt][nb][rr];
    }
    float* scratch = red + ROWS * LDR;
    {
        double* st = reinterpret_cast<double*>(scratch + 4);
        for (int e = tid; e < a.nstat * 96 * 2; e += DEEP_NTH) st[e] = 0.0;
    }
    __syncthreads();
    DEEP_STAMP(7);
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int e = tid + u * DEEP_NTH;
        const int rr = e / QPR, cq = e - rr * QPR;
        const int tok = tok0 + rr, n = n0 + 4 * cq;
        if (e < NQ && tok < a.Lout && n < a.N) {
            f32x4 v = *reinterpret_cast<const f32x4*>(red + rr * LDR + 4 * cq);
            v += e_bias[u];
            if (a.res) v += e_res[u];
            mtv_store_out4(a.out, ((size_t)b * a.Lout + tok) * a.N + n, v);
            if (a.nstat > 0) deep_stat_one(a.stat[0], 0, a.seg_out, scratch, tok, n, v);
            if (a.nstat > 1) deep_stat_one(a.stat[1], 1, a.seg_out, scratch, tok, n, v);
        }
    }
    if (a.nstat) {
        __syncthreads();
        deep_stat_flush_one(a.stat[0], 0, a.stat_cstride, scratch, b, tid);
        if (a.nstat > 1) deep_stat_flush_one(a.stat[1], 1, a.stat_cstride, scratch, b, tid);
    }
    DEEP_STAMP(9);
}

// =====================================================================================
// host side
// =====================================================================================
size_t deep_weight_floats(const DeepArgs& a, int NT) {
    (void)NT;
    return (size_t)a.N * ((size_t)a.ntaps * a.Cmain + a.Cskip);
}

static int deep_rows(const DeepArgs& a, int* src_rows) {     // rows of the largest row group: output, tapped source
    const int b1 = a.r * a.r, L = b1 + 2 * a.t * a.r;
    const int rs = a.up_main ? a.r >> 1 : (a.pool_main ? a.r << 1 : a.r), ts = a.up_main ? a.t >> 1 : (a.pool_main ? a.t << 1 : a.t);
    const int b1s = rs * rs, Ls = b1s + 2 * ts * rs;
    if (a.nrg == 2) {
        *src_rows = b1s > Ls - b1s ? b1s : Ls - b1s;
        return b1 > L - b1 ? b1 : L - b1;
    }
    *src_rows = Ls;
    return L;
}

bool deep_tile_for(const DeepArgs& a, DeepTile* t) {
    int srows = 0;
    const int rows = deep_rows(a, &srows);
    int RT = (rows + 15) / 16;
    RT = RT <= 1 ? 1 : (RT <= 2 ? 2 : (RT <= 4 ? 4 : (RT <= 8 ? 8 : 0)));
    if (!RT) return false;
    const int NT = (a.N % 48 == 0 && a.N >= 768) ? 3 : 1;
    if (a.N % (16 * NT)) return false;
    t->RT = RT;
    t->NT = NT;
    return true;
}

static size_t deep_layout(DeepArgs& a, DeepTile t);
size_t deep_smem_bytes(const DeepArgs& a0, DeepTile t) {
    DeepArgs a = a0;
    return deep_layout(a, t);
}
static size_t deep_layout(DeepArgs& a, DeepTile t) {
    int srows = 0;
    const int orows = deep_rows(a, &srows);
    const int ROWS = 16 * t.RT;
    a.src_rows_max = srows;
    const int SM = a.CSm + DEEP_PAD, SS = a.CSs + DEEP_PAD;
    int off = (srows + 1) * SM;
    a.lds_pool = 0;
    if (a.pool_main) {                                                      // pooled rows of the largest row group + the zero row
        a.lds_pool = off;
        off += (orows + 1) * SM;
    }
    a.lds_skip = off;
    if (a.Cskip) off += (ROWS + 1) * SS;
    a.lds_idx = off;
    off += (a.ntaps + 1) * ROWS;
    off = (off + 3) & ~3;
    a.lds_stat = off;
    off += (3 * DEEP_MAX_NG * 2) * 2;                                  // (sum, sum of squares) per (plane, group): doubles
    a.lds_red = 0;                                                     // the reduction scratch reuses the staged slices
    const int LDR = 16 * t.NT + 4;
    const int red = (8 * ROWS * LDR * 4 <= 96 * 1024 ? 8 : 4) * ROWS * LDR;
    a.lds_fin = ((off > red ? off : red) + 3) & ~3;                     // completion scratch: past everything the epilogue still reads
    return (size_t)(a.lds_fin + DEEP_FIN_FLOATS) * 4;
}

// Slicing of one conv of the deep levels: row groups, column tile, K slices.  `a` arrives with sources, channel counts, geometry and
// GroupNorm flags set; on success nrg / KS / CSm / CSs / tiles_n are filled in.  K slices: the most that keep the grid at <= 256
// workgroups (one per CU: every CU streams its share of the weights), within what the kernel supports -- power-of-two slices of
// 16 ... 256 channels that hold whole GroupNorm groups and do not straddle the parts of a channel concatenation.
bool deep_configure(DeepArgs& a, DeepTile* t, int max_ks) {
    if (a.B < 1 || a.Lout < 1 || a.Lout > 128 || (a.N & 15) || (a.Cmain & 15) || (a.Cskip & 15)) return false;
    if (a.ntaps != 9 && a.ntaps != 1) return false;
    a.nrg = (a.Lout > 64 && !(a.gn && a.whole)) ? 2 : 1;
    if (!deep_tile_for(a, t)) return false;
    if (max_ks == 1) t->NT = 1;                        // un-sliced K (plain output): column tiles are the only parallelism
    a.tiles_n = a.N / (16 * t->NT);
    const int C0m = a.main[1].p ? a.main[0].C : 0, C0s = a.skip[1].p ? a.skip[0].C : 0;
    int best = 0;
    for (int KS = 1; KS <= max_ks; KS *= 2) {
        if (a.Cmain % KS) continue;
        const int CSm = a.Cmain / KS, CSs = a.Cskip / KS;
        if (CSm < 16 || CSm > 512 || (CSm & (CSm - 1)) || (C0m && C0m % CSm)) continue;
        if (a.gn && ((a.gs & 3) || (a.gs & (a.gs - 1)) || CSm % a.gs || CSm / a.gs > DEEP_MAX_NG)) continue;
        if (a.Cskip && (a.Cskip % KS || CSs < 16 || CSs > 256 || (CSs & (CSs - 1)) || (C0s && C0s % CSs))) continue;
        DeepArgs probe = a;
        probe.KS = KS; probe.CSm = CSm; probe.CSs = CSs;
        probe.nmain_ch = 0;
        if (deep_smem_bytes(probe, *t) > 160 * 1024) continue;
        if (!best || (long)a.B * a.nrg * a.tiles_n * KS <= 256) best = KS;
    }
    if (!best) return false;
    a.KS = best;
    a.CSm = a.Cmain / best;
    a.CSs = a.Cskip / best;
    for (const DeepSrc* sp : {&a.main[0], &a.main[1], &a.skip[0], &a.skip[1], &a.res})
        if (sp->p && !(sp->ks == 1 || sp->ks == 2 || sp->ks == 4 || sp->ks == 8)) return false;
    return true;
}

// Row table of a configured conv (host): per row group [ntaps][ROWS] float offsets into the staged main slice -- row of the source
// token each (tap, output row) reads, the zero row for padding / rows past the group -- then [ROWS] offsets into the skip slice.
std::vector<int> deep_rowtab(const DeepArgs& a0, DeepTile t) {
    DeepArgs a = a0;
    (void)deep_layout(a, t);
    const int ROWS = 16 * t.RT, SM = a.CSm + DEEP_PAD, SS = a.CSs + DEEP_PAD;
    const int r = a.r, tt = a.t, b1 = r * r, L = b1 + 2 * tt * r;
    const int rs = a.up_main ? r >> 1 : r, b1s = rs * rs;
    std::vector<int> tab((size_t)a.nrg * (a.ntaps + 1) * ROWS);
    for (int rg = 0; rg < a.nrg; ++rg) {
        const int tok0 = (a.nrg == 2 && rg) ? b1 : 0, ntok = a.nrg == 2 ? (rg ? L - b1 : b1) : L;
        const int stok0 = (a.nrg == 2 && rg) ? b1s : 0;
        if (a.pool_main) {
            // the taps read the POOLED rows (this level's own grid, row i of the group at lds_pool + i SM; the zero row follows the group)
            int* tbp = tab.data() + (size_t)rg * (a.ntaps + 1) * ROWS;
            for (int tap = 0; tap < a.ntaps; ++tap)
                for (int ri = 0; ri < ROWS; ++ri) {
                    int row = -1;
                    if (ri < ntok) {
                        const int tok = tok0 + ri;
                        if (a.ntaps == 9) {
                            const int g = geo_source(r, tt, tok, tap / 3, tap % 3, false);
                            row = g < 0 ? -1 : (g & 0x0FFFFFFF) - tok0;
                        } else row = ri;
                    }
                    tbp[tap * ROWS + ri] = a.lds_pool + (row < 0 ? ntok : row) * SM;
                }
            for (int ri = 0; ri < ROWS; ++ri) tbp[a.ntaps * ROWS + ri] = ri < ntok ? ri * SS : ROWS * SS;
            continue;
        }
        int* tb = tab.data() + (size_t)rg * (a.ntaps + 1) * ROWS;
        for (int tap = 0; tap < a.ntaps; ++tap)
            for (int ri = 0; ri < ROWS; ++ri) {
                int row = -1;
                if (ri < ntok) {
                    const int tok = tok0 + ri;
                    if (a.ntaps == 9) {
                        const int g = geo_source(r, tt, tok, tap / 3, tap % 3, a.up_main != 0);
                        row = g < 0 ? -1 : (g & 0x0FFFFFFF) - stok0;
                    } else if (a.up_main) {
                        row = (geo_source(r, tt, tok, 1, 1, true) & 0x0FFFFFFF) - stok0;
                    } else {
                        row = tok - stok0;
                    }
                }
                tb[tap * ROWS + ri] = row < 0 ? a.src_rows_max * SM : row * SM;
            }
        for (int ri = 0; ri < ROWS; ++ri) tb[a.ntaps * ROWS + ri] = ri < ntok ? ri * SS : ROWS * SS;
    }
    return tab;
}

template <int RT, int NT>
static hipError_t deep_launch_t(const DeepArgs& a, size_t smem, hipStream_t s) {
    hipLaunchKernelGGL((k_deep_conv<RT, NT>), dim3((unsigned)(a.tiles_n * a.nslots)), dim3(DEEP_NTH), smem, s, a);
    return hipGetLastError();
}

hipError_t launch_deep_conv(const DeepArgs& a0, DeepTile t, hipStream_t s) {
    DeepArgs a = a0;
    if (a.KS < 1 || (a.KS & (a.KS - 1)) || a.KS > 8 || (a.nrg != 1 && a.nrg != 2)) return hipErrorInvalidValue;
    if (a.CSm < 16 || (a.CSm & (a.CSm - 1)) || a.CSm > 512 || a.CSm * a.KS != a.Cmain) return hipErrorInvalidValue;
    if (a.Cskip && (a.CSs * a.KS != a.Cskip || (a.CSs & (a.CSs - 1)) || a.CSs < 16 || a.CSs > 256)) return hipErrorInvalidValue;
    if (a.gn && ((a.gs & 3) || (a.gs & (a.gs - 1)) || a.CSm % a.gs || a.CSm / a.gs > DEEP_MAX_NG)) return hipErrorInvalidValue;
    if (a.N % (16 * t.NT) || (a.N & 3)) return hipErrorInvalidValue;
    if (a.gn && a.whole && a.nrg != 1) return hipErrorInvalidValue;      // statistics over all planes need all planes in one workgroup
    if (a.fin.out && a.fin.tagged && (t.NT != 1 || !a.fin.gran || !a.fin.ecnt || !a.fin.fault || a.KS < 2 || a.KS > 8)) return hipErrorInvalidValue;
    if (!a.zeros || !a.rowtab) return hipErrorInvalidValue;
    if ((a.pool_main && (a.up_main || a.main[1].p)) || (a.pool_res && (a.up_res || !a.res.p))) return hipErrorInvalidValue;
    if (!a.gn) { a.gamma = a.beta = a.zeros; }                           // (the kernel loads these vectors unconditionally)
    if (!a.gn || !a.film) { a.film = a.zeros; a.film_stride = 0; }
    a.tiles_n = a.N / (16 * t.NT);
    a.ks_shift = __builtin_ctz(a.KS);
    a.cpt_shift = __builtin_ctz(a.CSm / 16);
    a.nmain_ch = a.ntaps * (a.CSm / 16);
    a.nch = a.nmain_ch + (a.Cskip ? a.CSs / 16 : 0);
    a.nslots = a.B * a.nrg * a.KS;
    a.inv_nslots = 1.0f / (float)a.nslots;
    a.xm = 0;
    a.xm_jbits = 0;
    {
        static const int env_xm = getenv("MTV_DEEP_XM") ? atoi(getenv("MTV_DEEP_XM")) : 1;       // (A/B hook: 0 = the plain order everywhere)
        if (env_xm && a.B == 1 && a.nrg == 2 && a.KS <= 8 && a.tiles_n % (8 >> a.ks_shift) == 0) {
            a.xm = 1;
            a.xm_jbits = 3 - a.ks_shift;
        }
    }
    a.inv_r = 1.0f / (float)a.r;
    a.inv_rs = 0.f;
    {   // reciprocal element counts of the GroupNorm statistics (source level): planes 0, 1, 2 and all planes together
        const int rs = a.up_main ? a.r >> 1 : (a.pool_main ? a.r << 1 : a.r), ts = a.up_main ? a.t >> 1 : (a.pool_main ? a.t << 1 : a.t);
        const double gsd = a.gn ? (double)a.gs : 1.0;
        a.inv_n[0] = 1.0 / ((double)rs * rs * gsd);
        a.inv_n[1] = a.inv_n[2] = 1.0 / ((double)ts * rs * gsd);
        a.inv_n[3] = 1.0 / ((double)(rs * rs + 2 * ts * rs) * gsd);
    }
    const size_t smem = deep_layout(a, t);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
#define MTV_DEEP_GO(R, N) if (t.RT == R && t.NT == N) return deep_launch_t<R, N>(a, smem, s)
    MTV_DEEP_GO(1, 1); MTV_DEEP_GO(2, 1); MTV_DEEP_GO(4, 1); MTV_DEEP_GO(8, 1);
    MTV_DEEP_GO(1, 3); MTV_DEEP_GO(2, 3); MTV_DEEP_GO(4, 3); MTV_DEEP_GO(8, 3);
#undef MTV_DEEP_GO
    return hipErrorInvalidValue;
}

hipError_t deep_init_attrs() {
    const void* fn[] = {reinterpret_cast<const void*>(&k_deep_conv<1, 1>), reinterpret_cast<const void*>(&k_deep_conv<2, 1>),
                        reinterpret_cast<const void*>(&k_deep_conv<4, 1>), reinterpret_cast<const void*>(&k_deep_conv<8, 1>),
                        reinterpret_cast<const void*>(&k_deep_conv<1, 3>), reinterpret_cast<const void*>(&k_deep_conv<2, 3>),
                        reinterpret_cast<const void*>(&k_deep_conv<4, 3>), reinterpret_cast<const void*>(&k_deep_conv<8, 3>)};
    for (const void* f : fn) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    const void* fa[] = {reinterpret_cast<const void*>(&k_deep_attn<16>), reinterpret_cast<const void*>(&k_deep_attn<32>), reinterpret_cast<const void*>(&k_deep_attn<64>),
                        reinterpret_cast<const void*>(&k_conv_win<1, 4>), reinterpret_cast<const void*>(&k_conv_win<1, 2>),
                        reinterpret_cast<const void*>(&k_conv_win<2, 2>), reinterpret_cast<const void*>(&k_conv_win<2, 4>),
                        reinterpret_cast<const void*>(&k_conv_pw<1, 1, 8>), reinterpret_cast<const void*>(&k_conv_pw<1, 2, 8>),
                        reinterpret_cast<const void*>(&k_conv_pw<2, 1, 8>), reinterpret_cast<const void*>(&k_conv_pw<2, 2, 8>),
                        reinterpret_cast<const void*>(&k_conv_pw<1, 1, 6>), reinterpret_cast<const void*>(&k_conv_pw<2, 1, 6>),
                        reinterpret_cast<const void*>(&k_conv_pw<1, 1, 4>), reinterpret_cast<const void*>(&k_conv_pw<2, 1, 4>),
                        reinterpret_cast<const void*>(&k_conv_pw<1, 1, 2>), reinterpret_cast<const void*>(&k_conv_pw<2, 1, 2>)};
    for (const void* f : fa) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_deep_repack(const float* W, int ldw, float* dst, const DeepArgs& a0, int NT, hipStream_t s) {
    DeepArgs a = a0;
    a.cpt_shift = __builtin_ctz(a.CSm / 16);
    a.nmain_ch = a.ntaps * (a.CSm / 16);
    a.nch = a.nmain_ch + (a.Cskip ? a.CSs / 16 : 0);
    const long total = (long)deep_weight_floats(a, NT);
    hipLaunchKernelGGL(k_deep_repack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, ldw, dst, a, NT, total);
    return hipGetLastError();
}


static size_t deep_attn_smem(const DeepAttnArgs& a, int D) {
    const int KSTR = D + 4, VSTR = a.kcap + 4, AS = a.HPW * D + 8, WS = a.HPW * D + 4;
    const int os = 8 * 16 * (D + 4) > 8 * 16 * 20 + DEEP_FIN_FLOATS ? 8 * 16 * (D + 4) : 8 * 16 * 20 + DEEP_FIN_FLOATS;   // key-part partials | proj partials + completion scratch
    const int QR = 16 * (a.QT == 1 ? 1 : 2);
    return (size_t)(a.kcap * KSTR + D * VSTR + QR * KSTR + QR * AS + a.NC * WS + os + 8 * 16 * 2) * 4;
}

bool deep_attn_configure(DeepAttnArgs& a) {
    if (a.H < 1 || a.C % a.H || a.L < 1 || a.L > 128 || (a.C & 15)) return false;
    const int d = a.C / a.H;
    if (d != 16 && d != 32 && d != 64) return false;
    a.kcap = (a.L + 15) / 16 * 16;
    // round 6: ONE query tile per workgroup x 8 key parts, proj slices up to 64 columns -- at [128 x 512] 8 query groups x 4 head groups x 8 column groups = the same
    // 256 workgroups, each recomputing half the attention (1.9482 -> 1.9385 ms per step same-box, profiles/r06_deep_attn_query_tile_ab.txt).  MTV_DEEP_ATTN_QT=2: round 4's
    // 2 query tiles x 4 key parts, 16- / 32-column slices
    static const int env_qt = getenv("MTV_DEEP_ATTN_QT") ? atoi(getenv("MTV_DEEP_ATTN_QT")) : 1;
    a.QT = env_qt == 2 ? 2 : 1;
    a.nqg = (a.L + 16 * a.QT - 1) / (16 * a.QT);
    // head group = K slice of the projection = an output slab: as few slabs as the grid allows (consumers re-read every slab)
    for (int hpw : {8, 4, 2, 1}) {
        if (a.H % hpw || hpw * d > 128) continue;
        for (int nc : {64, 32, 16}) {
            if (a.C % nc || (nc == 64 && a.QT != 1)) continue;
            if ((hpw * d) * (nc / 4) > (nc == 64 ? 4 : 2) * DEEP_NTH) continue;             // proj slice: two (NC = 64: four) 16-byte quads per thread
            a.HPW = hpw; a.NC = nc; a.nhg = a.H / hpw; a.ncg = a.C / nc;
            if (a.nhg > 8 || (a.nhg & (a.nhg - 1))) continue;
            if (deep_attn_smem(a, d) > 160 * 1024) continue;
            const long wgs = (long)a.B * a.nqg * a.nhg * a.ncg;
            if (wgs >= 128) return true;                                     // enough to occupy half the chip: take the fewest slabs
        }
    }
    // small models: anything that fits
    for (int hpw : {1, 2, 4, 8}) {
        if (a.H % hpw || hpw * d > 128) continue;
        a.HPW = hpw; a.NC = 16; a.nhg = a.H / hpw; a.ncg = a.C / 16;
        if (a.nhg > 8 || (a.nhg & (a.nhg - 1)) || (hpw * d) * 4 > 2 * DEEP_NTH) continue;
        if (deep_attn_smem(a, d) <= 160 * 1024) return true;
    }
    return false;
}

hipError_t launch_deep_attn(const DeepAttnArgs& a0, hipStream_t s) {
    DeepAttnArgs a = a0;
    const int d = a.C / a.H;
    a.nslots = a.B * a.nqg * a.nhg;
    a.inv_nslots = 1.0f / (float)a.nslots;
    const size_t smem = deep_attn_smem(a, d);
    if (a.QT != 1 && a.QT != 2) return hipErrorInvalidValue;
    if (smem > 160 * 1024 || (a.NC != 16 && a.NC != 32 && !(a.NC == 64 && a.QT == 1)) || !(a.res.ks >= 1 && a.res.ks <= 8)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(a.ncg * a.nslots));
    if (d == 16) hipLaunchKernelGGL((k_deep_attn<16>), grid, dim3(DEEP_NTH), smem, s, a);
    else if (d == 32) hipLaunchKernelGGL((k_deep_attn<32>), grid, dim3(DEEP_NTH), smem, s, a);
    else if (d == 64) hipLaunchKernelGGL((k_deep_attn<64>), grid, dim3(DEEP_NTH), smem, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- k_conv_win (ConvTile{MT, NT, NW = 80, KS = 1 | 2 | 4, XM}) ----
static size_t conv_win_layout(const ConvArgs& a, int MT, int NT, int KS, int* wcap_out) {
    const int ROWS = 16 * MT, COLS = 16 * NT;
    int wcap = 1;                                           // the tallest window of any row tile (same arithmetic as the kernel)
    for (int tok0 = 0; tok0 < a.Lout; tok0 += ROWS) {
        int lo, n;
        conv_win_window(a.geo_r, a.geo_t, a.geo_main == 2, a.Lout, tok0, ROWS, &lo, &n);
        if (lo < 0 || lo + n > a.Lsrc) return (size_t)1 << 30;          // (never: the window is clamped to its planes)
        wcap = n > wcap ? n : wcap;
    }
    if (wcap_out) *wcap_out = wcap;
    size_t fl = (size_t)(wcap + 1) * (a.Cmain / KS + DEEP_PAD);
    if (a.Cskip) fl += (size_t)(ROWS + 1) * (a.Cskip / KS + DEEP_PAD);
    fl += 10 * ROWS + 384 + 192;                                        // row table | statistics (doubles) | (mean, rstd)
    const size_t red = (size_t)8 * ROWS * (COLS + 4) + DEEP_FIN_FLOATS;
    return (fl > red ? fl : red) * 4 + 64;
}
// Host-only check of conv_win_window (the arithmetic the kernel and the LDS sizing share): for both row-tile heights and every row
// tile of a level, the window lies inside the source tensor, is no taller than 16 MT + 2 r + 2 rows, and contains the source token of
// every tap of every output row of the tile.  0 = ok.
int conv_win_selftest(int r, int t, bool up, int Lout, int Lsrc) {
    for (int rows = 16; rows <= 32; rows += 16)
        for (int tok0 = 0; tok0 < Lout; tok0 += rows) {
            int lo = 0, n = 0;
            conv_win_window(r, t, up, Lout, tok0, rows, &lo, &n);
            if (lo < 0 || n < 1 || lo + n > Lsrc) return 1;
            if (!up && n > rows + 2 * r + 2) return 2;
            for (int ri = 0; ri < rows && tok0 + ri < Lout; ++ri)
                for (int tap = 0; tap < 9; ++tap) {
                    const int g = geo_source(r, t, tok0 + ri, tap / 3, tap % 3, up);
                    if (g < 0) continue;
                    const int src = g & 0x0FFFFFFF;
                    if (src < lo || src >= lo + n) return 3;
                }
        }
    return 0;
}

bool conv_win_eligible(const ConvArgs& a, int MT, int NT, int KS) {
    if (!((MT == 1 && (NT == 2 || NT == 4)) || (MT == 2 && (NT == 2 || NT == 4)))) return false;       // (2 x 4: round 6, with a 2 x 2-chunk weight ring)
    if (KS != 1 && KS != 2 && KS != 4) return false;
    // K slices: whole 16-channel chunks of the tapped and of the skip channels per slice; slab + counters of the plan (attached to every
    // conv by finish_split_k: checked at launch, not here -- tiles are chosen before that)
    if (KS > 1 && ((a.Cskip / 16) % KS || (a.Cmain / 16) % KS || (a.N & 3))) return false;
    if (a.ntaps != 9 || (a.geo_main != 1 && a.geo_main != 2) || a.out_cm || a.ddim || a.N % (16 * NT) || (a.Cmain & 15) || (a.Cskip & 15) || a.Cmain > 2048) return false;
    if (a.nmain == 2 && (a.C[0] & 3)) return false;
    if (a.nskip == 2 && (a.C[2] & 3)) return false;
    if (a.res && a.gather_skip && !a.geo_skip) return false;               // (residual rows by table only: not here)
    for (int t = 0; t < a.nstat; ++t)
        if (a.stat[t].coff & 3) return false;
    if ((long)(9 * a.Cmain + a.Cskip) * a.ldw * 4 >= 0x7F000000L) return false;
    return conv_win_layout(a, MT, NT, KS, nullptr) <= 160 * 1024;
}
size_t conv_win_smem_bytes(const ConvArgs& a, ConvTile t) { return conv_win_layout(a, t.MT, t.NT, t.KS < 1 ? 1 : t.KS, nullptr); }

template <int MT, int NT>
static hipError_t conv_win_launch_t(const ConvArgs& a0, int xm, int KS, hipStream_t s) {
    ConvArgs a = a0;
    if (!conv_win_eligible(a, MT, NT, KS) || (KS > 1 && (!a.slab || !a.tickets))) return hipErrorInvalidValue;
    const int tiles = (a.Lout + 16 * MT - 1) / (16 * MT), tiles_n = a.N / (16 * NT);
    a.KS = KS;
    a.xmap = 0;
    if (xm) {                                               // (a grid the XCD map does not tile keeps the plain order)
        const long nblk = (long)a.B * tiles * tiles_n;
        if (tiles_n % 8 == 0) {
            a.xmap = 1;
            a.Sx = tiles_n / 8;
            a.inv_Sx = 1.0f / (float)a.Sx;
        } else if ((tiles_n == 1 || tiles_n == 2 || tiles_n == 4) && nblk % 8 == 0) {
            a.xmap = 2;
            a.Sx = tiles_n == 1 ? 0 : (tiles_n == 2 ? 1 : 2);
        }
    }
    a.tiles_per_b = tiles;
    a.tiles_n = tiles_n;
    a.Bt = a.B * tiles;
    {
        static const int env_rb = getenv("MTV_ROW_BLOCKED") ? atoi(getenv("MTV_ROW_BLOCKED")) : 0;       // (A/B hook)
        if (env_rb && a.xmap == 0 && a.Bt % 8 == 0) {
            a.xmap = 3;
            a.Sx = a.Bt / 8;
            a.inv_Sx = 1.0f / (float)a.Sx;
        }
    }
    a.inv_tiles_per_b = 1.0f / (float)tiles;
    a.inv_Bt = 1.0f / (float)a.Bt;
    a.cps_q = a.Cskip / KS;                                  // skip / tapped channels of a K slice; the kernel's chunk walk is per slice:
    a.cps_r = a.Cmain / KS;
    a.cpt = a.cps_r / 16;                                    // 16-channel chunks per tap and slice
    a.inv_cpt = 1.0f / (float)a.cpt;
    a.inv_tiles_n = 1.0f / (float)(a.Bt * tiles_n);          // (block -> K slice)
    a.geo_inv_r = 1.0f / (float)a.geo_r;
    int wcap = 0;
    const size_t smem = conv_win_layout(a, MT, NT, KS, &wcap);
    a.rec_cap = wcap;
    if ((long)a.Bt * tiles_n * KS >= (1L << 21)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_conv_win<MT, NT>), dim3((unsigned)(a.Bt * tiles_n * KS)), dim3(DEEP_NTH), smem, s, a);
    return hipGetLastError();
}
hipError_t launch_conv_win(const ConvArgs& a, ConvTile t, hipStream_t s) {
    static const int env_xm = getenv("MTV_WIN_XM") ? atoi(getenv("MTV_WIN_XM")) : -1;       // (A/B hook: block order of every k_conv_win launch)
    if (env_xm >= 0) t.XM = env_xm;
    const int KS = t.KS < 1 ? 1 : t.KS;
    if (t.MT == 1 && t.NT == 4) return conv_win_launch_t<1, 4>(a, t.XM, KS, s);
    if (t.MT == 1 && t.NT == 2) return conv_win_launch_t<1, 2>(a, t.XM, KS, s);
    if (t.MT == 2 && t.NT == 2) return conv_win_launch_t<2, 2>(a, t.XM, KS, s);
    if (t.MT == 2 && t.NT == 4) return conv_win_launch_t<2, 4>(a, t.XM, KS, s);
    return hipErrorInvalidValue;
}

// ---- k_conv_pw (ConvTile{MT, NTW, NW = 96, KS = 1 | 6 | 4 | 2 = multiplying waves, XM}) ----
static int conv_pw_waves(int code) { return code == 1 ? 8 : code; }      // ConvTile::KS of a k_conv_pw tile: 1 = all 8 waves multiply, else 6 / 4 / 2
static size_t conv_pw_layout(const ConvArgs& a, int MT, int NTW, int NWA) {
    const int ROWS = 16 * MT, COLS = 16 * NTW * NWA;
    const size_t stage = (size_t)ROWS * (a.Cmain + DEEP_PAD) + 384 + 192;          // rows | statistics (doubles) | (mean, rstd)
    const size_t image = (size_t)ROWS * (COLS + 4) + DEEP_FIN_FLOATS;
    return (stage > image ? stage : image) * 4 + 64;
}
bool conv_pw_eligible(const ConvArgs& a, int MT, int NTW, int wcode) {
    const int NWA = conv_pw_waves(wcode);
    if (!((MT == 1 || MT == 2) && (NTW == 1 || NTW == 2))) return false;       // (NTW = 3 measured slower than 1 and 2 on every shape: not built)
    if (!(NWA == 8 || (NTW == 1 && (NWA == 6 || NWA == 4 || NWA == 2)))) return false;
    if (!a.Wpk || (a.N & 15) || a.ntaps != 1 || a.nmain != 1 || a.nskip != 0 || a.Cskip != 0 || a.gather || a.gather_skip || a.geo_main || a.geo_skip) return false;
    if (a.out_cm || a.ddim || a.bias_b || a.bias2) return false;
    if ((a.Cmain & 15) || a.Cmain < 64 || a.Cmain > 512 || (512 % (a.Cmain >> 2)) || a.N % (16 * NTW) || a.Lsrc != a.Lout || (a.res && a.Lskip != a.Lout)) return false;
    if (16 * MT > 8 * (512 / (a.Cmain >> 2)) * 2) return false;              // (rows beyond the first pass are fetched late: keep that to one extra pass)
    for (int t = 0; t < a.nstat; ++t)
        if (a.stat[t].coff & 3) return false;
    if ((long)a.N * a.Cmain * 4 >= 0x7F000000L) return false;
    return conv_pw_layout(a, MT, NTW, NWA) <= 160 * 1024;
}
size_t conv_pw_smem_bytes(const ConvArgs& a, ConvTile t) { return conv_pw_layout(a, t.MT, t.NT, conv_pw_waves(t.KS)); }

template <int MT, int NTW, int NWA>
static hipError_t conv_pw_launch_t(const ConvArgs& a0, int xm, hipStream_t s) {
    ConvArgs a = a0;
    if (!conv_pw_eligible(a, MT, NTW, NWA == 8 ? 1 : NWA)) return hipErrorInvalidValue;
    constexpr int COLS = 16 * NTW * NWA;
    const int tiles = (a.Lout + 16 * MT - 1) / (16 * MT), groups = (a.N + COLS - 1) / COLS;
    a.KS = 1;
    a.xmap = 0;
    if (xm && groups % 8 == 0) {                            // (a grid the XCD map does not tile keeps the plain order)
        a.xmap = 1;
        a.Sx = groups / 8;
        a.inv_Sx = 1.0f / (float)a.Sx;
    }
    a.tiles_per_b = tiles;
    a.tiles_n = groups;
    a.Bt = a.B * tiles;
    {
        static const int env_rb = getenv("MTV_ROW_BLOCKED") ? atoi(getenv("MTV_ROW_BLOCKED")) : 0;       // (A/B hook)
        if (env_rb && a.xmap == 0 && a.Bt % 8 == 0) {
            a.xmap = 3;
            a.Sx = a.Bt / 8;
            a.inv_Sx = 1.0f / (float)a.Sx;
        }
    }
    a.inv_tiles_per_b = 1.0f / (float)tiles;
    a.inv_Bt = 1.0f / (float)a.Bt;
    if ((long)a.Bt * groups >= (1L << 21)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_conv_pw<MT, NTW, NWA>), dim3((unsigned)(a.Bt * groups)), dim3(DEEP_NTH), conv_pw_layout(a, MT, NTW, NWA), s, a);
    return hipGetLastError();
}
hipError_t launch_conv_pw(const ConvArgs& a, ConvTile t, hipStream_t s) {
    static const int env_xm = getenv("MTV_PW_XM") ? atoi(getenv("MTV_PW_XM")) : -1;         // (A/B hook: block order of every k_conv_pw launch)
    int xm = env_xm >= 0 ? env_xm : t.XM;
    if (xm == 2) xm = (long)a.N >= 2L * a.B * a.Lout;                                       // (hook value 2: only where the weights are twice the rows)
    const int nwa = conv_pw_waves(t.KS);
    if (t.MT == 1 && t.NT == 1 && nwa == 8) return conv_pw_launch_t<1, 1, 8>(a, xm, s);
    if (t.MT == 1 && t.NT == 2 && nwa == 8) return conv_pw_launch_t<1, 2, 8>(a, xm, s);
    if (t.MT == 2 && t.NT == 1 && nwa == 8) return conv_pw_launch_t<2, 1, 8>(a, xm, s);
    if (t.MT == 2 && t.NT == 2 && nwa == 8) return conv_pw_launch_t<2, 2, 8>(a, xm, s);
    if (t.MT == 1 && t.NT == 1 && nwa == 6) return conv_pw_launch_t<1, 1, 6>(a, xm, s);
    if (t.MT == 2 && t.NT == 1 && nwa == 6) return conv_pw_launch_t<2, 1, 6>(a, xm, s);
    if (t.MT == 1 && t.NT == 1 && nwa == 4) return conv_pw_launch_t<1, 1, 4>(a, xm, s);
    if (t.MT == 2 && t.NT == 1 && nwa == 4) return conv_pw_launch_t<2, 1, 4>(a, xm, s);
    if (t.MT == 1 && t.NT == 1 && nwa == 2) return conv_pw_launch_t<1, 1, 2>(a, xm, s);
    if (t.MT == 2 && t.NT == 1 && nwa == 2) return conv_pw_launch_t<2, 1, 2>(a, xm, s);
    return hipErrorInvalidValue;
}

hipError_t launch_deep_finalize(const DeepFinArgs& a, hipStream_t s) {
    if (a.nstat < 0 || a.nstat > 2 || (a.src.C & 3) || a.src.ks < 1 || a.src.ks > 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_deep_finalize, dim3((unsigned)((a.src.C + 63) / 64), (unsigned)((a.L + 7) / 8), (unsigned)a.B), dim3(128), 0, s, a);
    return hipGetLastError();
}

}  // namespace mtv
