// Implicit-GEMM convolution for gfx950 on the exact-f32 matrix instruction v_mfma_f32_16x16x4_f32.
//
// Replaces: ResBlock in_layers / out_layers / skip_connection (MToV/models/ddpm/unet.py:131-167,
// 178-207), the attention blocks' qkv / proj_out conv1d (unet.py:234,242,251,253), the stem and
// head convs (unet.py:714,971-975), and the GroupNorm that precedes each of them
// (diffusionmodules.py:156-173): normalisation + FiLM + SiLU are applied to the A operand on its
// way to the matrix core, and the statistics of the OUTPUT are accumulated in the epilogue for the
// GroupNorm sites that will consume it, so no separate normalisation pass ever touches HBM.
//
// MFMA 16x16x4 f32 operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&15][k = l>>4]     B: lane l holds B[k = l>>4][j = l&15]
//   D: lane l, reg r holds D[row = 4*(l>>4) + r][col = l&15]
// K is a reduction index, so A and B may use any common permutation of K, and the column index of
// B/D may be permuted too.  Both freedoms are used so that every operand fragment is a plain
// 16-byte global load (no LDS staging of operands is needed at the f32 MFMA rate):
//   A: lane (i,q) loads x[row i][c0 + 4q .. 4q+3]; MFMA step s consumes component s, i.e. K slot q
//      of step s is channel c0 + 4q + s;
//   B: lane (j,q) loads W[c0 + 4q + s][n0 + NT*j .. +NT-1] for s = 0..3; column block nb of the
//      wave tile is output channel n0 + NT*j + nb.
//
// Work split: a 1-D grid of B * ceil(Lout / 16MT) row tiles (a tile never straddles a batch element)
// x N / 16NT column tiles x KS cross-workgroup K slices, decoded from blockIdx.x in an XCD-aware
// order (ConvArgs::xmap); the NW waves of a workgroup split their K range further.  K = (tap,
// 16-channel chunk) pairs; each wave walks its chunk range with a ring of DEPTH (2-4) chunks in
// flight (loads of chunks i+1.. under the MFMAs of chunk i).  Waves are summed in LDS in fixed order
// (deterministic); with KS > 1 the partial tiles go to a slab and the slice that finishes last adds
// them in fixed order inside the same launch.
#include <cstdio>
#include <cstdlib>

#include "mtv_internal.h"

namespace mtv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ int seg_of(const SegInfo& s, int tok) { return tok >= s.b2 ? 2 : (tok >= s.b1 ? 1 : 0); }

// K walk.  K is cut into chunks of 16 channels, ordered tap by tap over the tapped source part(s), then over the
// parts of the fused 1x1 skip conv.  Every workgroup first writes one 16-byte RECORD per chunk of its K range into
// LDS (all threads, one chunk each: source pointer, W row, coefficient channel, row stride, tap), so that a wave's
// walk is branch-free: chunk n -> read record n (one broadcast ds_read_b128), read the tap's row table, issue the
// loads.  (The previous design walked (tap, part) segments with a scalar cursor; its set-up and segment changes
// cost ~1000 straight-line instructions per wave before the first load -- at two waves per SIMD that was 2.5-4.6 us
// of every launch; measured with the phase stamps, tools/stamps.py.)
#ifndef MTV_ABLATE
#define MTV_ABLATE 0          // tools/ubench/conv_bench builds ablated variants: 1 no MFMA, 2 no A loads, 4 no B loads,
                              // 8 no transform, 16 prologue only, 32 no reduction/epilogue
#endif

struct ChunkRec {             // 16 bytes, one per K chunk of the workgroup, in LDS
    unsigned a_lo, a_hi;      // source part base + 4 * (first channel of the chunk)
    unsigned wrow;            // W row of the chunk's first channel
    unsigned meta;            // coefficient channel (16 bits) | channels of the part / 16 (8 bits) << 16 | tap (4 bits) << 24 | skip << 28
};

__device__ __forceinline__ int usgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

// exact n / d for 0 <= n < 2^22, d > 0, given inv = 1.0f / d: a float multiply and a +-1 correction instead of the
// ~40-instruction integer division sequence (these divisions sit on every wave's critical path before its first load)
__device__ __forceinline__ int fdiv(int n, int d, float inv) { return FDiv{inv}(n, d); }

#if MTV_ABLATE & 64   // phase timestamps of thread 0 of four sampled blocks (first two, middle, last) -> a.dbg (conv_bench, MTV_STAMPS)
#define MTV_STAMP(k) do { if (threadIdx.x == 0 && a.dbg) { const int sb_ = blockIdx.x < 2 ? (int)blockIdx.x : (blockIdx.x == gridDim.x / 2 ? 2 : (blockIdx.x == gridDim.x - 1 ? 3 : -1)); \
                          if (sb_ >= 0) a.dbg[sb_ * 16 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define MTV_STAMP(k) do { } while (0)
#endif

typedef __attribute__((address_space(1))) const char gchar;   // global (not flat) loads: vmcnt only, saddr addressing

template <int MT, int NT>
struct Raw {
    f32x4 a[MT];
    int e[MT];
    float b[4][NT];
    int cc, skip;     // uniform: coefficient channel of this chunk, skip flag
};

// what a wave needs to turn a chunk record into loads (all uniform except woff / q16 / i)
struct WalkCtx {
    const ChunkRec* recs;     // LDS, indexed by chunk - wg_ch0
    const int* idx;           // LDS row table [(ntaps + 1)][ROWS]: source token | plane << 28, or -1 (zero pad)
    int wg_ch0;
    unsigned bL[2];           // b * Lsrc, b * Lskip
    gchar* W;
    unsigned ldw4;            // bytes per W row
    unsigned woff;            // per lane: ((4q) * ldw + n0 + NT * i) * 4
    unsigned q16;             // per lane: 16 * q (bytes)
    int i;
};

// issue the loads of one chunk: W fragment rows wrow .. wrow+15, A rows given by the source-token entries e[]
template <int MT, int NT>
__device__ __forceinline__ void issue_chunk(const WalkCtx& w, unsigned a_lo, unsigned a_hi, unsigned wrow, unsigned meta, const int (&e)[MT],
                                            Raw<MT, NT>& o) {
    const int skip = (int)(meta >> 28);
    const unsigned Cp4 = ((meta >> 16) & 0xFFu) << 6;            // bytes per source row of this part
    o.cc = (int)(meta & 0xFFFFu);
    o.skip = skip;
    gchar* wbase = w.W + (size_t)wrow * (size_t)w.ldw4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        gchar* p = wbase + (w.woff + (unsigned)s * w.ldw4);
        if constexpr (MTV_ABLATE & 4) {
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) o.b[s][nb] = 0.5f;
        } else if constexpr (NT == 4) {
            const f32x4 t = *(const __attribute__((address_space(1))) f32x4*)p;
            o.b[s][0] = t[0]; o.b[s][1] = t[1]; o.b[s][2] = t[2]; o.b[s][3] = t[3];
        } else if constexpr (NT == 2) {
            const f32x2 t = *(const __attribute__((address_space(1))) f32x2*)p;
            o.b[s][0] = t[0]; o.b[s][1] = t[1];
        } else {
            o.b[s][0] = *(const __attribute__((address_space(1))) float*)p;
        }
    }
    gchar* abase = (gchar*)(((unsigned long long)a_hi << 32) | a_lo);
    const unsigned bl = skip ? w.bL[1] : w.bL[0];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        o.e[mt] = e[mt];
        const unsigned st = e[mt] < 0 ? 0u : (unsigned)(e[mt] & 0x0FFFFFFF);   // padded rows read a valid address, zeroed later
        const unsigned rowoff = (bl + st) * Cp4 + w.q16;
        if constexpr (MTV_ABLATE & 2) o.a[mt] = f32x4{1.f, 2.f, 3.f, 4.f};
        else o.a[mt] = *(const __attribute__((address_space(1))) f32x4*)(abase + rowoff);
    }
}

// chunk `ch` through the LDS tables (record + row table)
template <int MT, int NT, int ROWS>
__device__ __forceinline__ void load_chunk(const WalkCtx& w, int ch, Raw<MT, NT>& o) {
    const ChunkRec* rp = w.recs + (ch - w.wg_ch0);
    const unsigned a_lo = (unsigned)usgpr((int)rp->a_lo), a_hi = (unsigned)usgpr((int)rp->a_hi);
    const unsigned wrow = (unsigned)usgpr((int)rp->wrow), meta = (unsigned)usgpr((int)rp->meta);
    const int tap = (int)((meta >> 24) & 15u);
    int e[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) e[mt] = w.idx[tap * ROWS + 16 * mt + w.i];
    issue_chunk<MT, NT>(w, a_lo, a_hi, wrow, meta, e, o);
}

// the record of chunk `ch` from the launch arguments alone (what the table builder stores in LDS)
__device__ __forceinline__ ChunkRec chunk_record(const ConvArgs& a, int ch) {
    const int nmainch = a.ntaps * a.cpt, Cmain = a.Cmain;
    const bool skip = ch >= nmainch;
    const int tap = skip ? a.ntaps : fdiv(ch, a.cpt, a.inv_cpt);
    const int w = skip ? ch - nmainch : ch - tap * a.cpt;
    const int c0_16 = (skip ? a.C[2] : a.C[0]) >> 4;
    const bool second = w >= c0_16;                            // second part of a channel concatenation
    const int c = (second ? w - c0_16 : w) << 4;
    const float* sp = skip ? (second ? a.src[3] : a.src[2]) : (second ? a.src[1] : a.src[0]);
    const int Cp = skip ? (second ? a.C[3] : a.C[2]) : (second ? a.C[1] : a.C[0]);
    const int coff = second ? (skip ? a.C[2] : a.C[0]) : 0;
    const unsigned long long ab = reinterpret_cast<unsigned long long>(sp) + (unsigned long long)c * 4ull;
    ChunkRec r;
    r.a_lo = (unsigned)(ab & 0xFFFFFFFFull);
    r.a_hi = (unsigned)(ab >> 32);
    r.wrow = (unsigned)((skip ? a.ntaps * Cmain : tap * Cmain) + coff + c);
    r.meta = (unsigned)(coff + c) | ((unsigned)(Cp >> 4) << 16) | ((unsigned)tap << 24) | ((unsigned)(skip ? 1 : 0) << 28);
    return r;
}

// the row-table entry of (tap t, output token tok): source token | plane << 28, -1 = zero padding / past the end
__device__ __forceinline__ int row_entry(const ConvArgs& a, int t, int tok) {
    if (tok >= a.Lout) return -1;
    if (t < a.ntaps) {
        if (a.geo_main) {
            const int ky = t >= 6 ? 2 : (t >= 3 ? 1 : 0);
            return geo_source_t<FDiv>(FDiv{a.geo_inv_r}, a.geo_r, a.geo_t, tok, ky, t - 3 * ky, a.geo_main == 2);
        }
        const int st = a.gather ? a.gather[t * a.Lout + tok] : tok;
        return st < 0 ? -1 : (st | (seg_of(a.seg_src, st) << 28));
    }
    if (a.geo_skip) return geo_source_t<FDiv>(FDiv{a.geo_inv_r}, a.geo_r, a.geo_t, tok, 1, 1, true) & 0x0FFFFFFF;
    return a.gather_skip ? a.gather_skip[tok] : tok;
}

// coef holds, per (plane, channel), the folded affine {A, B}: y = x*A + B with A = gn_scale*(1+film_scale),
// B = gn_bias*(1+film_scale) + film_shift  (GroupNorm affine and FiLM are both per-channel affine maps).
template <int MT, int NT>
__device__ __forceinline__ void mma_chunk(const float2* coef, int Cmain, bool do_gn, bool act, int q,
                                          const Raw<MT, NT>& in, f32x4 (&acc)[MT][NT]) {
    float av[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 v = in.a[mt];
        const int e = in.e[mt];
        if (do_gn && !in.skip && !(MTV_ABLATE & 8)) {
            const int sg = e < 0 ? 0 : ((e >> 28) & 3);
            const f32x4* cf = reinterpret_cast<const f32x4*>(coef + sg * Cmain + in.cc + 4 * q);
            const f32x4 k0 = cf[0], k1 = cf[1];     // {A0,B0,A1,B1}, {A2,B2,A3,B3}
            float y0 = fmaf(v[0], k0[0], k0[1]), y1 = fmaf(v[1], k0[2], k0[3]);
            float y2 = fmaf(v[2], k1[0], k1[1]), y3 = fmaf(v[3], k1[2], k1[3]);
            if (act) { y0 = silu_fast(y0); y1 = silu_fast(y1); y2 = silu_fast(y2); y3 = silu_fast(y3); }
            v = f32x4{y0, y1, y2, y3};
        }
        if (e < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};              // conv zero padding applies AFTER norm/activation
        av[mt][0] = v[0]; av[mt][1] = v[1]; av[mt][2] = v[2]; av[mt][3] = v[3];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < NT; ++nb) {
                if constexpr (MTV_ABLATE & 1) acc[mt][nb][s] += av[mt][s] * in.b[s][nb];
                else acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][s], in.b[s][nb], acc[mt][nb], 0, 0, 0);
            }
}

// bias / per-batch bias / residual epilogue of one output element
__device__ __forceinline__ float epi(const ConvArgs& a, float v, int b, int tok, int rs, int nn) {
    float o = v + a.bias[nn];
    if (a.bias2) o += a.bias2[nn];
    if (a.bias_b) o += a.bias_b[(size_t)b * a.bias_b_stride + nn];
    if (a.res) o += a.res[((size_t)b * a.Lskip + rs) * a.N + nn];
    return o;
}

__device__ __forceinline__ void stat_add(const ConvArgs& a, int b, int sg, int n, double s, double ss) {
    for (int t = 0; t < a.nstat; ++t) {
        const int g = fdiv(a.stat[t].coff + n, a.stat[t].gs, a.stat[t].inv_gs);
        double* dst = a.stat[t].sums + (size_t)(blockIdx.x & (STAT_COPIES - 1)) * a.stat_cstride + (((size_t)b * 3 + sg) * 32 + g) * 2;
        atomicAdd(dst, s);
        atomicAdd(dst + 1, ss);
    }
}

template <int MT, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void k_conv(const ConvArgs a) {
    MTV_STAMP(7);
    touch_kernargs<(int)sizeof(ConvArgs)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float2 s_mr[3][32];
    __shared__ f64x2 s_dp[96];                 // (sum, sum of squares) per (plane, group), copies added up
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // SGPR: the K walk is scalar
    const int i = lane & 15, q = lane >> 4;
    constexpr int ROWS = 16 * MT, COLS = 16 * NT, NTH = NW * 64;
    const int tiles_per_b = a.tiles_per_b, tiles_n = a.tiles_n;       // (launch_conv precomputes the tiling and the
                                                                       // reciprocals the decode below divides by)
    // ---- sampler-step head (mtv_internal.h DdimFuse): the hand-over record and the step index are requested at
    // entry and used once this workgroup's operand ring is in flight, so their latency is hidden
    const DdimFuse* dd = a.ddim;
    DdimFuse ddv{};
    int step_now = 0;
    if (dd) {
        ddv = *dd;
        step_now = *a.step_counter;
    }
    // block -> (row tile, column tile, K slice).  The dispatcher round-robins consecutive workgroups over
    // the 8 XCDs (private L2s).  xmap 0: row tiles vary fastest (every L2 sees the whole -- small -- weight
    // matrix).  xmap 1, for weight-dominated layers: workgroup id % 8 selects the (column tile, K slice)
    // class, so each weight slice is fetched from HBM by exactly ONE L2 (speed only, never correctness).
    int bx, by, bz;
    {
        const int blk = usgpr((int)blockIdx.x);
        if (a.xmap == 0) {
            const int rest = fdiv(blk, a.Bt, a.inv_Bt);
            bx = blk - rest * a.Bt;
            bz = fdiv(rest, tiles_n, a.inv_tiles_n);
            by = rest - bz * tiles_n;
        } else {
            const int xcd = blk & 7, j = blk >> 3;
            bx = fdiv(j, a.Sx, a.inv_Sx);
            const int sl = xcd + 8 * (j - bx * a.Sx);
            if (sl >= tiles_n * a.KS) return;          // padding block (whole workgroup, before any barrier)
            bz = fdiv(sl, tiles_n, a.inv_tiles_n);
            by = sl - bz * tiles_n;
        }
        bx = usgpr(bx); by = usgpr(by); bz = usgpr(bz);
    }
    const int b = usgpr(fdiv(bx, tiles_per_b, a.inv_tiles_per_b));
    const int tok0 = (bx - b * tiles_per_b) * ROWS;
    const int n0 = by * COLS;
    const int Cmain = a.Cmain;
    const bool do_gn = a.gn.sums != nullptr;

    // GroupNorm inputs are requested FIRST (they are needed last, after the operand ring is issued): the statistics of
    // the producer(s) -- (plane, group) sums over the privatised copies, 8 independent 16-byte loads per thread -- and this
    // thread's first 4 channels of gamma / beta / FiLM.  Their round trip overlaps the table build and the ring issue.
    constexpr int PER = 4;                          // channels per thread per round (one 16-byte load per operand)
    f32x4 ga, be, s1, sh;
    const float* film = (do_gn && a.gn.film) ? a.gn.film + (size_t)b * a.gn.film_stride : nullptr;
    auto fetch = [&](int c0) {                      // (Cmain % 16 == 0 and every table is 16-byte aligned)
        ga = *reinterpret_cast<const f32x4*>(a.gn.gamma + c0);
        be = *reinterpret_cast<const f32x4*>(a.gn.beta + c0);
        s1 = f32x4{1.f, 1.f, 1.f, 1.f};
        sh = f32x4{0.f, 0.f, 0.f, 0.f};
        if (film) {
            s1 += *reinterpret_cast<const f32x4*>(film + c0);
            sh = *reinterpret_cast<const f32x4*>(film + Cmain + c0);
        }
    };
    // (the 8 copies stay in registers un-summed until the ring is issued -- adding them here would wait for them here;
    // the largest tiles have no registers to spare and sum at once)
    constexpr bool HOLD = MT * NT <= 8 && NW < 16;
    f64x2 v0 = {0.0, 0.0};
    f64x2 vraw[HOLD ? STAT_COPIES : 1];
    if (do_gn) {
        if (tid * PER < Cmain) fetch(tid * PER);
        if (tid < 96) {
#pragma unroll
            for (int k = 0; k < STAT_COPIES; ++k) {
                const f64x2 t = *reinterpret_cast<const f64x2*>(a.gn.sums + (size_t)k * a.gn.cstride + (size_t)b * 192 + (size_t)tid * 2);
                if constexpr (HOLD) vraw[k] = t;
                else v0 += t;
            }
        }
    }

    MTV_STAMP(0);
    // ---- LDS: chunk records of this workgroup's K range | source-token table [(ntaps+1)][ROWS] | folded
    // per-(plane, channel) affine {A, B}
    const int slice0 = bz * NW;                                        // this workgroup's K slices: slice0 .. slice0 + NW - 1
    const int wg_ch0 = slice0 * a.cps_q + min(slice0, a.cps_r);        // slice s covers chunks [s*q + min(s, r), +q (+1 if s < r))
    const int wg_ch1 = (slice0 + NW) * a.cps_q + min(slice0 + NW, a.cps_r);
    ChunkRec* recs = reinterpret_cast<ChunkRec*>(smem);
    int* idx = reinterpret_cast<int*>(smem + a.rec_cap * 4);
    const int idx_floats = ((a.ntaps + 1) * ROWS + 3) & ~3;
    float2* coef = reinterpret_cast<float2*>(smem + a.rec_cap * 4 + idx_floats);
    // ---- this wave's chunk range.  Its FIRST chunk is requested right away, straight from the launch arguments
    // (record and row entries computed in registers): the first -- HBM-cold -- round trip starts before the tables
    // are built instead of after them.
    const int slice = slice0 + wave;
    const int ch0 = slice * a.cps_q + min(slice, a.cps_r);
    const int ch1 = ch0 + a.cps_q + (slice < a.cps_r ? 1 : 0);
    WalkCtx wc;
    wc.recs = recs;
    wc.idx = idx;
    wc.wg_ch0 = wg_ch0;
    wc.bL[0] = (unsigned)b * (unsigned)a.Lsrc;
    wc.bL[1] = (unsigned)b * (unsigned)a.Lskip;
    wc.W = (gchar*)(unsigned long long)a.W;
    wc.ldw4 = (unsigned)a.ldw * 4u;
    wc.woff = (4u * q * (unsigned)a.ldw + (unsigned)(n0 + NT * i)) * 4u;
    wc.q16 = 16u * q;
    wc.i = i;
    // chunks in flight per wave, bounded by the register budget (1024-thread blocks get 128 VGPRs)
    constexpr int DEPTH = NW == 16 ? (MT * NT >= 4 ? 2 : (MT * NT >= 2 ? 3 : 4)) : (MT * NT >= 16 ? 2 : (MT * NT >= 8 ? 3 : 4));
    Raw<MT, NT> ring[DEPTH];
    int nx = ch0;                                       // next chunk to request
    if (ch0 < ch1) {
        const ChunkRec r0 = chunk_record(a, ch0);
        const unsigned m0 = (unsigned)usgpr((int)r0.meta);
        const int tap0 = (int)((m0 >> 24) & 15u);
        int e0[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) e0[mt] = row_entry(a, tap0, tok0 + 16 * mt + i);
        issue_chunk<MT, NT>(wc, (unsigned)usgpr((int)r0.a_lo), (unsigned)usgpr((int)r0.a_hi), (unsigned)usgpr((int)r0.wrow), m0, e0, ring[0]);
        ++nx;
    }
    MTV_STAMP(10);
    for (int e = tid; e < wg_ch1 - wg_ch0; e += NTH) recs[e] = chunk_record(a, wg_ch0 + e);
    // source-token table: identity, arithmetic (geo_source) or -- only if the host check of the formula ever failed
    // -- the gather tables in global memory.  (Built by the LAST threads of the workgroup, the chunk records above by
    // the first ones: in an 8-wave workgroup no wave executes both, which halves the instructions on this stretch.)
    for (int e = NTH - 1 - tid; e < (a.ntaps + 1) * ROWS; e += NTH) {
        const int t = e / ROWS, r = e - t * ROWS;
        idx[e] = row_entry(a, t, tok0 + r);
    }
    MTV_STAMP(8);
    __syncthreads();
    MTV_STAMP(9);
    // the rest of the operand ring, through the tables
#pragma unroll
    for (int d = 1; d < DEPTH; ++d)
        if (d < ch1 - ch0) {
            load_chunk<MT, NT, ROWS>(wc, nx, ring[d]);
            ++nx;
        }

    DdimStep stp{};
    if (dd) {
        // every workgroup of the launch zeroes its slice of the OTHER parity's statistics arena and copies its slice
        // of the next step's FiLM row (nobody reads either any more in this step): fire-and-forget stores that
        // drain under the conv itself; the step's scalars are fetched for the epilogue
        for (long long e = (long long)blockIdx.x * NTH + tid; e < ddv.zero_vec4; e += (long long)gridDim.x * NTH)
            mtv_store_out4(ddv.zero_arena, (size_t)(4 * e), f32x4{0.f, 0.f, 0.f, 0.f});
        if (step_now + 1 < ddv.n_steps) {
            const float* src = ddv.film_tab + (size_t)(step_now + 1) * ddv.film_total;
            for (int e = blockIdx.x * NTH + tid; e < ddv.film_total; e += gridDim.x * NTH) ddv.film_out[e] = src[e];
        }
        stp = ddv.steps[step_now];
    }
    MTV_STAMP(1);
    if (do_gn) {
        // A cross-plane site (AttentionBlock1D) adds its three planes up through LDS -- summing 24 entries per thread
        // in registers made the compiler serialise the loads.
        const bool whole = a.gn.whole != 0;
        if constexpr (HOLD) {
            if (tid < 96) {
#pragma unroll
                for (int k = 0; k < STAT_COPIES; ++k) v0 += vraw[k];
            }
        }
        if constexpr (NTH < 96) {                       // one-wave workgroups: entries 64..95 in a second round
            if (tid < 32) {
                f64x2 v1 = {0.0, 0.0};
#pragma unroll
                for (int k = 0; k < STAT_COPIES; ++k)
                    v1 += *reinterpret_cast<const f64x2*>(a.gn.sums + (size_t)k * a.gn.cstride + (size_t)b * 192 + (size_t)(64 + tid) * 2);
                s_dp[64 + tid] = v1;
            }
            if (tid < 64) s_dp[tid] = v0;
            __syncthreads();
        } else if (whole) {
            if (tid < 96) s_dp[tid] = v0;
            __syncthreads();
        }
        for (int e = tid; e < 96; e += NTH) {
            const int sg = e >> 5, g = e & 31;
            f64x2 v;
            double inv_n;                               // 1 / (tokens of the plane (or of all planes) x channels per group), from the host
            if (whole) {
                v = (s_dp[g] + s_dp[32 + g]) + s_dp[64 + g];
                inv_n = a.gn.inv_n[3];
            } else {
                v = NTH < 96 ? s_dp[e] : v0;
                inv_n = a.gn.inv_n[sg];
            }
            const double mean = v[0] * inv_n;
            double var = v[1] * inv_n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mr[sg][g] = make_float2((float)mean, 1.0f / sqrtf((float)var + 1e-5f));   // fp64 only where cancellation bites
        }
        __syncthreads();
        for (int c0 = tid * PER; c0 < Cmain; c0 += NTH * PER) {
            if (c0 != tid * PER) fetch(c0);
            int grp[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) grp[k] = fdiv(c0 + k, a.gn.gs, a.gn.inv_gs);
#pragma unroll
            for (int sg = 0; sg < 3; ++sg)
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const float2 mr = s_mr[sg][grp[k]];
                    const float sc = mr.y * ga[k];
                    const float bi = be[k] - sc * mr.x;
                    coef[sg * Cmain + c0 + k] = make_float2(sc * s1[k], fmaf(bi, s1[k], sh[k]));
                }
        }
    }
    __syncthreads();

    if constexpr (MTV_ABLATE & 16) return;   // prologue only
    MTV_STAMP(2);
    // ---- K loop over this wave's chunk range: a ring of DEPTH chunks in flight
    const bool act = a.gn.act != 0;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[mt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // epilogue operands of this thread's first quad are requested now, so their latency hides under the K loop
    constexpr int QPR0 = COLS / 4;
    f32x4 pre_bias = f32x4{0.f, 0.f, 0.f, 0.f}, pre_res = f32x4{0.f, 0.f, 0.f, 0.f};
    bool pre_ok = false;
    {
        const int rr = tid / QPR0, cq = tid - rr * QPR0;
        const int tok = tok0 + rr, n = n0 + cq * 4;
        if (NW < 16 && a.KS == 1 && rr < ROWS && tok < a.Lout && n + 3 < a.N) {   // (1024-thread blocks have no registers to spare)
            pre_ok = true;
            pre_bias = *reinterpret_cast<const f32x4*>(a.bias + n);
            if (a.bias2) pre_bias += *reinterpret_cast<const f32x4*>(a.bias2 + n);
            if (a.bias_b) pre_bias += *reinterpret_cast<const f32x4*>(a.bias_b + (size_t)b * a.bias_b_stride + n);
            if (a.res) {
                const int rs = idx[a.ntaps * ROWS + rr];            // skip/residual source row of this output row
                pre_res = *reinterpret_cast<const f32x4*>(a.res + ((size_t)b * a.Lskip + rs) * a.N + n);
            }
        }
    }

    if (ch0 < ch1) {
        // ring of DEPTH chunks in flight.  The walk has no conditionals (chunk records), so every ring slot keeps
        // fixed registers and the loads of the next DEPTH-1 chunks stay in flight under the MFMAs.
        int n = ch1 - ch0;                       // chunks not yet multiplied
        while (n >= 2 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                mma_chunk<MT, NT>(coef, Cmain, do_gn, act, q, ring[d], acc);
                load_chunk<MT, NT, ROWS>(wc, nx, ring[d]);
                ++nx;
            }
            n -= DEPTH;
        }
        // drain: n < 2*DEPTH chunks left, min(n, DEPTH) of them already in the ring
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d < n) {
                mma_chunk<MT, NT>(coef, Cmain, do_gn, act, q, ring[d], acc);
                if (d + DEPTH < n) {
                    load_chunk<MT, NT, ROWS>(wc, nx, ring[d]);
                    ++nx;
                }
            }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d + DEPTH < n) mma_chunk<MT, NT>(coef, Cmain, do_gn, act, q, ring[d], acc);
    }

    MTV_STAMP(3);
    // sampler-step head: the sample and the noise of this thread's output quad are requested here, ahead of the cross-wave reduction
    // (round 6: read in the epilogue they were first-touch misses on the launch's tail; the step record `stp` arrived under the K loop)
    float dd_x[4] = {0.f, 0.f, 0.f, 0.f}, dd_nz[4] = {0.f, 0.f, 0.f, 0.f};
    bool dd_pre = false;
    if (dd) {
        constexpr int QPR0 = COLS / 4;
        const int rr = tid / QPR0, cq = tid - rr * QPR0;
        const int tok = tok0 + rr, n = n0 + cq * 4;
        dd_pre = tid < ROWS * QPR0 && tok < a.Lout && n < a.N;
        if (dd_pre) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const size_t ix = ((size_t)b * a.N + n + (n + k < a.N ? k : 0)) * a.Lout + tok;
                dd_x[k] = ddv.x[ix];
                dd_nz[k] = stp.noise_index >= 0 ? ddv.noise[(size_t)stp.noise_index * ddv.n_per_draw + ix] : 0.f;
            }
        }
    }
    // ---- fixed-order tree over the NW waves (lane-linear LDS image: conflict-free), LDS reused
    __syncthreads();
    float* red = smem;
    // per-(plane, channel quad) statistics slots of this tile: their own LDS region past everything else (a.qs_off),
    // zeroed here, filled by LDS fp64 atomics from the epilogue pass below
    double* qs = reinterpret_cast<double*>(smem + a.qs_off);     // [3 planes][QPR quads][2]
    bool fast = a.nstat > 0 && (a.N & 3) == 0;
    for (int t = 0; t < a.nstat; ++t) fast = fast && ((a.stat[t].gs & 3) == 0);
    if (fast)
        for (int e = tid; e < 3 * (COLS / 4) * 2; e += NTH) qs[e] = 0.0;
    constexpr int TILE_REGS = MT * NT * 4;
    constexpr int LDR = COLS + 4;
    constexpr int WTILE = ROWS * LDR;            // floats of one wave's partial tile as a (row, col) image (ONE_STAGE)
    constexpr bool ONE_STAGE = NW > 1 && NW * WTILE * 4 <= 48 * 1024;   // all partials fit in LDS at once
    if constexpr (NW == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) red[(16 * mt + 4 * q + r) * LDR + NT * i + nb] = acc[mt][nb][r];
        __syncthreads();
    } else if constexpr (ONE_STAGE) {
        // every wave parks its partial tile as a (row, col) image: lane (i, q) holds, for row 4q + r, the NT consecutive
        // columns NT i .. -> one 16-byte (8-, 4-byte) store per (row block, r), and pass 1 reads one 16-byte quad per wave
        // (the earlier lane-linear image cost 16 scalar stores and 4 NW scalar loads per thread)
        float* my = red + (size_t)wave * WTILE + (4 * q) * LDR + NT * i;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* dstp = my + (16 * mt + r) * LDR;
                if constexpr (NT == 4) *reinterpret_cast<f32x4*>(dstp) = f32x4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
                else if constexpr (NT == 2) *reinterpret_cast<f32x2*>(dstp) = f32x2{acc[mt][0][r], acc[mt][1][r]};
                else dstp[0] = acc[mt][0][r];
            }
        __syncthreads();
    } else {
#pragma unroll
        for (int s = NW / 2; s >= 1; s >>= 1) {
            if (wave >= s && wave < 2 * s) {
                float* my = red + (size_t)(wave - s) * TILE_REGS * 64 + lane;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) my[((mt * NT + nb) * 4 + r) * 64] = acc[mt][nb][r];
            }
            __syncthreads();
            if (wave < s) {
                const float* my = red + (size_t)wave * TILE_REGS * 64 + lane;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[mt][nb][r] += my[((mt * NT + nb) * 4 + r) * 64];
            }
            __syncthreads();
        }
        if (wave == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nb = 0; nb < NT; ++nb) red[(16 * mt + 4 * q + r) * LDR + NT * i + nb] = acc[mt][nb][r];
        }
        __syncthreads();
    }

    MTV_STAMP(4);
    // ---- pass 1 (row-major over the tile): epilogue + coalesced store
    constexpr int QPR = COLS / 4, QUADS = ROWS * QPR;
    const bool want_stats = a.nstat > 0;
    float* fin = ONE_STAGE ? red + (size_t)NW * WTILE : red;   // (row, col) image of the finished tile for pass 2
    auto tile_quad = [&](int rr, int cq) -> f32x4 {                     // this workgroup's (partial) result for one quad
        if constexpr (ONE_STAGE) {
            // fixed order over the waves (run-to-run bit-equal)
            f32x4 v = *reinterpret_cast<const f32x4*>(red + rr * LDR + cq * 4);
#pragma unroll
            for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(red + (size_t)w * WTILE + rr * LDR + cq * 4);
            return v;
        } else {
            return *reinterpret_cast<const f32x4*>(red + rr * LDR + cq * 4);
        }
    };
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    const size_t sstride = (size_t)a.B * a.Lout * a.N;                  // floats between K slices of the slab
    if (a.KS > 1) {
        // Cross-workgroup split-K, completed INSIDE this launch (cdna_hip_programming.md G16, form R1):
        // every slice parks its partial tile in the slab with WRITE-THROUGH (sc1) 8-byte stores, drains
        // them, then ONE lane takes a ticket; the slice that draws the last ticket re-reads all KS partials
        // (sc1 loads: L2/fabric-served, never a stale L1 line) and runs the epilogue.  Sum order is fixed
        // (slice 0..KS-1), so the result does not depend on which slice finishes last.  (N % 4 == 0 here.)
        for (int e = tid; e < QUADS; e += NTH) {
            const int rr = e / QPR, cq = e - rr * QPR;
            const int tok = tok0 + rr, n = n0 + cq * 4;
            if (tok >= a.Lout || n >= a.N) continue;
            const f32x4 v = tile_quad(rr, cq);
            gu64* dst = (gu64*)(unsigned long long)(a.slab + (size_t)bz * sstride + ((size_t)b * a.Lout + tok) * a.N + n);
            __hip_atomic_store(dst, ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, ((unsigned long long)__float_as_uint(v[3]) << 32) | __float_as_uint(v[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
        __syncthreads();
        __shared__ int s_last[4];
        int* ticket = a.tickets + ((size_t)b * tiles_per_b + (tok0 / ROWS)) * tiles_n + by;
        if (tid == 0) s_last[0] = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.KS - 1;
        __syncthreads();
        if (!s_last[0]) return;
        if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning for the next launch
    }
    for (int e = tid; e < QUADS; e += NTH) {
        const int rr = e / QPR, cq = e - rr * QPR;
        const int tok = tok0 + rr, n = n0 + cq * 4;
        if (tok >= a.Lout || n >= a.N) continue;
        f32x4 v;
        if (a.KS > 1) {
            gu64* src = (gu64*)(unsigned long long)(a.slab + ((size_t)b * a.Lout + tok) * a.N + n);
            v = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < a.KS; k0 += 8) {      // up to 16 loads in flight, then a fixed-order sum
                unsigned long long t0[8], t1[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool in = k0 + k < a.KS;
                    t0[k] = in ? __hip_atomic_load(src + (size_t)(k0 + k) * (sstride / 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    t1[k] = in ? __hip_atomic_load(src + (size_t)(k0 + k) * (sstride / 2) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[0] += __uint_as_float((unsigned)t0[k]);
                    v[1] += __uint_as_float((unsigned)(t0[k] >> 32));
                    v[2] += __uint_as_float((unsigned)t1[k]);
                    v[3] += __uint_as_float((unsigned)(t1[k] >> 32));
                }
            }
        } else {
            v = tile_quad(rr, cq);
        }
        if (pre_ok && e == tid) {
            // same association as epi(): ((v + bias) + bias2 ...) is replaced by v + (bias + bias2 ...) + res;
            // a sub-ulp reassociation of the bias terms only
            v = (v + pre_bias) + pre_res;
        } else {
            const int rs = a.res ? row_entry(a, a.ntaps, tok) : tok;   // (the LDS index table is gone by now)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (n + k < a.N) v[k] = epi(a, v[k], b, tok, rs, n + k);
        }
        if (!a.out_cm && n + 3 < a.N) {
            mtv_store_out4(a.out, ((size_t)b * a.Lout + tok) * a.N + n, v);
        } else if (dd) {
            // sampler-step head (N == 4 == the sample's channels): v is eps; x <- DDIM update, in the external
            // layout and as channels 0..3 of the packed input of the next step's stem conv.  The (up to) four x and four
            // noise values are requested together: one round trip instead of one per channel (the store of channel k
            // may alias the load of channel k + 1 as far as the compiler knows).
            float xo[4], nz[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = n + k < a.N;
                const size_t ix = ((size_t)b * a.N + n + (in ? k : 0)) * a.Lout + tok;
                if (e == tid && dd_pre) {
                    xo[k] = dd_x[k];
                    nz[k] = dd_nz[k];
                } else {
                    xo[k] = ddv.x[ix];
                    nz[k] = stp.noise_index >= 0 ? ddv.noise[(size_t)stp.noise_index * ddv.n_per_draw + ix] : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (n + k >= a.N) break;
                const size_t ix = ((size_t)b * a.N + n + k) * a.Lout + tok;
                const float xn = ddim_update_elem(stp, xo[k], v[k], nz[k]);
                ddv.x[ix] = xn;
                ddv.h0[((size_t)b * a.Lout + tok) * 16 + n + k] = xn;
            }
        } else {
            for (int k = 0; k < 4 && n + k < a.N; ++k) {
                if (a.out_cm) a.out[((size_t)b * a.N + n + k) * a.Lout + tok] = v[k];
                else a.out[((size_t)b * a.Lout + tok) * a.N + n + k] = v[k];
            }
        }
        if (fast) {
            // GroupNorm statistics of the output for its consumers: (sum, sum of squares) of this quad in fp64, added to
            // the tile's (plane, quad) slot in LDS (lanes of a wave-instruction hold different quads of <= 4 rows: at
            // most a 4-way same-address serialisation); combined per group after the pass
            const int sgq = seg_of(a.seg_out, tok);
            const double sq = ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
            const double ssq = ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]);
            atomicAdd(&qs[(sgq * QPR + cq) * 2], sq);
            atomicAdd(&qs[(sgq * QPR + cq) * 2 + 1], ssq);
        } else if (want_stats) {
            *reinterpret_cast<f32x4*>(fin + rr * LDR + cq * 4) = v;
        }
    }
    MTV_STAMP(5);
    if (dd) {
        // the workgroup that arrives last advances the step counter: every workgroup of the launch has read the
        // counter (entry duties, step record) before its own arrival, so none can see the new value
        __syncthreads();
        if (tid == 0) {
            if (__hip_atomic_fetch_add(ddv.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.B * tiles_per_b * tiles_n - 1) {
                __hip_atomic_store(ddv.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ddv.counter, step_now + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (!want_stats) return;

    // ---- statistics hand-over.  A workgroup issues ONE fp64 atomic pair per (plane, group, consumer) -- the
    // per-address atomic rate is what bounds this epilogue (measured: quads of a wide group hitting one address cost
    // 5x the whole kernel) -- into copy blockIdx & 7 of the consumer site's table.
    __syncthreads();
    if (!fast) {
        // groups narrower than a channel quad (test-size models only): per-channel sums, rows reduced by wave shuffles
        constexpr int W = ROWS < 64 ? ROWS : 64;         // lanes that share one channel quad
        for (int base = 0; base < QUADS; base += NTH) {
            const int e = base + tid;
            const int cq = e / ROWS, rr = e - cq * ROWS;
            const int tok = tok0 + rr, n = n0 + cq * 4;
            const bool ok = e < QUADS && tok < a.Lout && n < a.N;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(fin + rr * LDR + cq * 4);
            const int sg = ok ? seg_of(a.seg_out, tok) : -1;
            for (int sgi = 0; sgi < 3; ++sgi) {
                if (!__any(sg == sgi)) continue;
                const bool mine = sg == sgi;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double s = mine ? (double)v[k] : 0.0;
                    double ss = mine ? (double)v[k] * v[k] : 0.0;
#pragma unroll
                    for (int o = W / 2; o >= 1; o >>= 1) {
                        s += __shfl_xor(s, o);
                        ss += __shfl_xor(ss, o);
                    }
                    if ((lane & (W - 1)) == 0 && e < QUADS && n + k < a.N) stat_add(a, b, sgi, n + k, s, ss);
                }
            }
        }
        return;
    }
    // one thread per (consumer, plane, quad): the first quad of each group inside this tile adds up its group
    for (int e = tid; e < a.nstat * 3 * QPR; e += NTH) {
        const int t = e / (3 * QPR), r2 = e - t * 3 * QPR;
        const int sgi = r2 / QPR, cq = r2 - sgi * QPR;
        const int n = n0 + cq * 4;
        if (n >= a.N) continue;
        const int gs = a.stat[t].gs, coff = a.stat[t].coff;
        const float inv_gs = a.stat[t].inv_gs;
        const int g = fdiv(coff + n, gs, inv_gs);
        if (cq > 0 && fdiv(coff + n - 4, gs, inv_gs) == g) continue;           // not the first quad of its group in this tile
        const int qend = min(QPR, min((a.N - n0 + 3) >> 2, ((g + 1) * gs - coff - n0 + 3) >> 2));   // quads of group g inside this tile
        double s = 0.0, ss = 0.0;
        for (int c2 = cq; c2 < qend; ++c2) {
            s += qs[(sgi * QPR + c2) * 2];
            ss += qs[(sgi * QPR + c2) * 2 + 1];
        }
        if (ss != 0.0) {
            double* dst = a.stat[t].sums + (size_t)(blockIdx.x & (STAT_COPIES - 1)) * a.stat_cstride + (((size_t)b * 3 + sgi) * 32 + g) * 2;
            atomicAdd(dst, s);
            atomicAdd(dst + 1, ss);
            if (a.dbg_stat_dup) {
                atomicAdd(dst + a.dbg_stat_dup, s);
                atomicAdd(dst + a.dbg_stat_dup + 1, ss);
            }
        }
    }
    MTV_STAMP(6);
}

// --------------------------------------------------------------------------------------------
// k_conv_lds<WM, WN>: the same convolution for LARGE token counts (several clips batched on one GPU, the
// 512x512 geometry, the autoencoder's 16384-token GEMMs), where k_conv's one-wave-one-tile operand delivery is
// bound by the L2 -> CU path (measured: 6 KB per 32 MFMAs and wave at tile 32x64 = 47 B/clk/CU of the 64 B/clk a CU
// can take).  A workgroup of 2 x 2 waves owns a (32 WM) x (32 WN) output tile and walks the whole K range; each
// 16-channel chunk of A (rows gathered by tap, GroupNorm / FiLM / SiLU applied ONCE on the way) and of W is staged in
// LDS by all 256 threads, double buffered (next chunk's global loads in flight under this chunk's MFMAs), and read
// back as MFMA fragments by the four waves: 16 KB from L2 per 256 MFMAs instead of 48 KB.  Same arguments, same
// epilogue semantics (bias, residual, statistics); no split-K (large M has enough tiles), token-major output only.
// --------------------------------------------------------------------------------------------
template <int WM, int WN>
__global__ __launch_bounds__(256) void k_conv_lds(const ConvArgs a) {
    touch_kernargs<(int)sizeof(ConvArgs)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float2 s_mr[3][32];
    __shared__ f64x2 s_dp[96];
    constexpr int BM = 32 * WM, BN = 32 * WN, NTH = 256;
    constexpr int ASTR = 20;                    // floats per staged A row: 16 + 4 (16-byte fragment reads of 16 rows hit 16 distinct bank quads)
    constexpr int BSTR = BN + 4;
    constexpr int APT = BM / 64;                // A float4 per thread per chunk (row = tid/4 + 64 k, quad = tid & 3)
    constexpr int BPT = BN / 64;                // W float4 per thread per chunk (k row = tid / (BN/4) + (1024/BN) k, column quad = tid % (BN/4))
    constexpr int BROWS = 1024 / BN;            // W rows covered per pass of the 256 threads
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = lane & 15, q = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    const int tiles_per_b = a.tiles_per_b, tiles_n = a.tiles_n;
    const int blk = usgpr((int)blockIdx.x);
    const int bx = usgpr(fdiv(blk, tiles_n, a.inv_tiles_n));           // column tiles fastest: neighbours share the A rows in L2
    const int by = blk - bx * tiles_n;
    const int b = usgpr(fdiv(bx, tiles_per_b, a.inv_tiles_per_b));
    const int tok0 = (bx - b * tiles_per_b) * BM;
    const int n0 = by * BN;
    const int Cmain = a.Cmain;
    const bool do_gn = a.gn.sums != nullptr;

    // GroupNorm inputs first (needed last): see k_conv
    constexpr int PER = 4;
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
        if (tid * PER < Cmain) fetch(tid * PER);
        if (tid < 96) {
#pragma unroll
            for (int k = 0; k < STAT_COPIES; ++k)
                v0 += *reinterpret_cast<const f64x2*>(a.gn.sums + (size_t)k * a.gn.cstride + (size_t)b * 192 + (size_t)tid * 2);
        }
    }

    // ---- LDS: chunk records | row table [(ntaps+1)][BM] | coefficients | A stages | W stages | statistics slots
    const int nchunks = a.cps_q;                                       // (KS * NW == 1 here: one slice = all chunks)
    ChunkRec* recs = reinterpret_cast<ChunkRec*>(smem);
    int* idx = reinterpret_cast<int*>(smem + a.rec_cap * 4);
    const int idx_floats = ((a.ntaps + 1) * BM + 3) & ~3;
    float2* coef = reinterpret_cast<float2*>(smem + a.rec_cap * 4 + idx_floats);
    float* As = smem + a.rec_cap * 4 + idx_floats + (do_gn ? 6 * Cmain : 0);
    float* Bs = As + 2 * BM * ASTR;
    double* qs = reinterpret_cast<double*>(smem + a.qs_off);
    {
        const int nmainch = a.ntaps * a.cpt;
        for (int e = tid; e < nchunks; e += NTH) {
            const bool skip = e >= nmainch;
            const int tap = skip ? a.ntaps : fdiv(e, a.cpt, a.inv_cpt);
            const int w = skip ? e - nmainch : e - tap * a.cpt;
            const int c0_16 = (skip ? a.C[2] : a.C[0]) >> 4;
            const bool second = w >= c0_16;
            const int c = (second ? w - c0_16 : w) << 4;
            const float* sp = skip ? (second ? a.src[3] : a.src[2]) : (second ? a.src[1] : a.src[0]);
            const int Cp = skip ? (second ? a.C[3] : a.C[2]) : (second ? a.C[1] : a.C[0]);
            const int coff = second ? (skip ? a.C[2] : a.C[0]) : 0;
            const unsigned long long ab = reinterpret_cast<unsigned long long>(sp) + (unsigned long long)c * 4ull;
            ChunkRec r;
            r.a_lo = (unsigned)(ab & 0xFFFFFFFFull);
            r.a_hi = (unsigned)(ab >> 32);
            r.wrow = (unsigned)((skip ? a.ntaps * Cmain : tap * Cmain) + coff + c);
            r.meta = (unsigned)(coff + c) | ((unsigned)(Cp >> 4) << 16) | ((unsigned)tap << 24) | ((unsigned)(skip ? 1 : 0) << 28);
            recs[e] = r;
        }
    }
    auto skip_src = [&](int tok) -> int {
        if (a.geo_skip) return geo_source_t<FDiv>(FDiv{a.geo_inv_r}, a.geo_r, a.geo_t, tok, 1, 1, true) & 0x0FFFFFFF;
        return a.gather_skip ? a.gather_skip[tok] : tok;
    };
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
                    v = st < 0 ? -1 : (st | (seg_of(a.seg_src, st) << 28));
                }
            } else {
                v = skip_src(tok);
            }
        }
        idx[e] = v;
    }
    for (int e = tid; e < 3 * (BN / 4) * 2; e += NTH) qs[e] = 0.0;
    if (do_gn) {
        const bool whole = a.gn.whole != 0;
        if (whole) {
            if (tid < 96) s_dp[tid] = v0;
        }
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
        for (int c0 = tid * PER; c0 < Cmain; c0 += NTH * PER) {
            if (c0 != tid * PER) fetch(c0);
            int grp[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) grp[k] = fdiv(c0 + k, a.gn.gs, a.gn.inv_gs);
#pragma unroll
            for (int sg = 0; sg < 3; ++sg)
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const float2 mr = s_mr[sg][grp[k]];
                    const float sc = mr.y * ga[k];
                    const float bi = be[k] - sc * mr.x;
                    coef[sg * Cmain + c0 + k] = make_float2(sc * s1[k], fmaf(bi, s1[k], sh[k]));
                }
        }
    }
    __syncthreads();

    // ---- staging: global -> registers (raw) -> [transform] -> LDS
    const bool act = a.gn.act != 0;
    const unsigned bL[2] = {(unsigned)b * (unsigned)a.Lsrc, (unsigned)b * (unsigned)a.Lskip};
    gchar* const Wg = (gchar*)(unsigned long long)a.W;
    const unsigned ldw4 = (unsigned)a.ldw * 4u;
    const int arow = tid >> 2, aq = tid & 3;               // this thread's A rows: arow + 64 k, channel quad aq
    const int bcol = tid % (BN / 4), brow = tid / (BN / 4);
    f32x4 ra[APT], rb[BPT];
    int re[APT];
    int rcc = 0, rskip = 0;
    auto gload = [&](int ch) {
        const ChunkRec* rp = recs + ch;
        const unsigned a_lo = (unsigned)usgpr((int)rp->a_lo), a_hi = (unsigned)usgpr((int)rp->a_hi);
        const unsigned wrow = (unsigned)usgpr((int)rp->wrow), meta = (unsigned)usgpr((int)rp->meta);
        const int tap = (int)((meta >> 24) & 15u);
        rskip = (int)(meta >> 28);
        rcc = (int)(meta & 0xFFFFu);
        const unsigned Cp4 = ((meta >> 16) & 0xFFu) << 6;
        gchar* abase = (gchar*)(((unsigned long long)a_hi << 32) | a_lo);
        const unsigned bl = rskip ? bL[1] : bL[0];
#pragma unroll
        for (int k = 0; k < APT; ++k) {
            const int e = idx[tap * BM + arow + 64 * k];
            re[k] = e;
            const unsigned st = e < 0 ? 0u : (unsigned)(e & 0x0FFFFFFF);
            ra[k] = *(const __attribute__((address_space(1))) f32x4*)(abase + ((bl + st) * Cp4 + 16u * aq));
        }
        gchar* wbase = Wg + (size_t)wrow * (size_t)ldw4;
#pragma unroll
        for (int k = 0; k < BPT; ++k)
            rb[k] = n0 + 4 * bcol < a.ldw     // (a column tile may reach past the padded row of W: those columns are never stored)
                        ? *(const __attribute__((address_space(1))) f32x4*)(wbase + ((unsigned)(brow + BROWS * k) * ldw4 + (unsigned)(n0 + 4 * bcol) * 4u))
                        : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto lstore = [&](int buf) {
        float* Ab = As + buf * BM * ASTR;
        float* Bb = Bs + buf * 16 * BSTR;
#pragma unroll
        for (int k = 0; k < APT; ++k) {
            f32x4 v = ra[k];
            const int e = re[k];
            if (do_gn && !rskip) {
                const int sg = e < 0 ? 0 : ((e >> 28) & 3);
                const f32x4* cf = reinterpret_cast<const f32x4*>(coef + sg * Cmain + rcc + 4 * aq);
                const f32x4 k0 = cf[0], k1 = cf[1];
                float y0 = fmaf(v[0], k0[0], k0[1]), y1 = fmaf(v[1], k0[2], k0[3]);
                float y2 = fmaf(v[2], k1[0], k1[1]), y3 = fmaf(v[3], k1[2], k1[3]);
                if (act) { y0 = silu_fast(y0); y1 = silu_fast(y1); y2 = silu_fast(y2); y3 = silu_fast(y3); }
                v = f32x4{y0, y1, y2, y3};
            }
            if (e < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(Ab + (arow + 64 * k) * ASTR + 4 * aq) = v;
        }
#pragma unroll
        for (int k = 0; k < BPT; ++k) *reinterpret_cast<f32x4*>(Bb + (brow + BROWS * k) * BSTR + 4 * bcol) = rb[k];
    };

    f32x4 acc[WM][WN];
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nb = 0; nb < WN; ++nb) acc[mt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const float* Ab = As + buf * BM * ASTR + (wm * 16 * WM + i) * ASTR + 4 * q;
        // a lane's WN output columns: WN <= 4: consecutive (wn 16 WN + WN i + nb); WN == 8: two quads 64 columns apart
        // (wn 128 + 64 (nb >> 2) + 4 i + (nb & 3)) so that both are conflict-free 16-byte reads of 16 consecutive quads
        const float* Bb = Bs + buf * 16 * BSTR + (4 * q) * BSTR + wn * 16 * WN + (WN == 8 ? 4 : WN) * i;
        f32x4 af[WM];
#pragma unroll
        for (int mt = 0; mt < WM; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(Ab + 16 * mt * ASTR);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float bv[WN];
            if constexpr (WN == 8) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(Bb + s * BSTR), u = *reinterpret_cast<const f32x4*>(Bb + s * BSTR + 64);
                bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3];
                bv[4] = u[0]; bv[5] = u[1]; bv[6] = u[2]; bv[7] = u[3];
            } else if constexpr (WN == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(Bb + s * BSTR);
                bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3];
            } else {
                const f32x2 t = *reinterpret_cast<const f32x2*>(Bb + s * BSTR);
                bv[0] = t[0]; bv[1] = t[1];
            }
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                for (int nb = 0; nb < WN; ++nb) acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][s], bv[nb], acc[mt][nb], 0, 0, 0);
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more) gload(ch + 1);                 // in flight under this chunk's MFMAs
        compute(ch & 1);
        if (more) lstore((ch + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue straight from the accumulators: lane (i, q) of wave (wm, wn) holds, for row 4q + r of row block mt, the
    // WN consecutive output channels n0 + 16 WN wn + WN i ..
    const bool fast = a.nstat > 0;
    constexpr int QPR = BN / 4;
    constexpr int NG = WN == 8 ? 2 : 1, GW = WN / NG;          // column groups of a lane / their width (see compute)
#pragma unroll
    for (int cg = 0; cg < NG; ++cg) {
        const int colw = n0 + wn * 16 * WN + (WN == 8 ? 64 * cg + 4 * i : WN * i);
        if (colw >= a.N) continue;
        float bias[GW];
#pragma unroll
        for (int nb = 0; nb < GW; ++nb) {
            bias[nb] = a.bias[colw + nb];
            if (a.bias2) bias[nb] += a.bias2[colw + nb];
            if (a.bias_b) bias[nb] += a.bias_b[(size_t)b * a.bias_b_stride + colw + nb];
        }
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 16 * WM + 16 * mt + 4 * q + r;
                const int tok = tok0 + row;
                if (tok >= a.Lout) continue;
                float o[GW];
#pragma unroll
                for (int nb = 0; nb < GW; ++nb) o[nb] = acc[mt][cg * GW + nb][r] + bias[nb];
                if (a.res) {
                    const int rs = idx[a.ntaps * BM + row];
                    const float* rp = a.res + ((size_t)b * a.Lskip + rs) * a.N + colw;
#pragma unroll
                    for (int nb = 0; nb < GW; ++nb) o[nb] += rp[nb];
                }
                float* op = a.out + ((size_t)b * a.Lout + tok) * a.N + colw;
                if constexpr (GW == 4) *reinterpret_cast<f32x4*>(op) = f32x4{o[0], o[1], o[2], o[3]};
                else *reinterpret_cast<f32x2*>(op) = f32x2{o[0], o[1]};
                if (fast) {
                    const int sgq = seg_of(a.seg_out, tok);
                    double sq = 0.0, ssq = 0.0;
#pragma unroll
                    for (int nb = 0; nb < GW; ++nb) {
                        sq += (double)o[nb];
                        ssq += (double)o[nb] * o[nb];
                    }
                    const int cq = (colw - n0) >> 2;               // (GW == 2: two lanes share a quad slot)
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
        const int g = fdiv(coff + n, gs, inv_gs);
        if (cq > 0 && fdiv(coff + n - 4, gs, inv_gs) == g) continue;
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

// --------------------------------------------------------------------------------------------
// tile / split selection: a small analytic cost model (times in microseconds)
// --------------------------------------------------------------------------------------------
// chunk records a workgroup needs room for: its NW slices of ceil-ish nchunks / (KS * NW) chunks each
static int rec_capacity(int nchunks, int NW, int KS) {
    const int ns = NW * KS, q = nchunks / ns, r = nchunks % ns;
    return (NW * q + (NW < r ? NW : r) + 3) & ~3;
}

// LDS layout: [chunk records | row table | coefficients] during the K loop, reused by [wave partials | finished tile]
// afterwards; the per-quad statistics slots sit past both (qs_off, in floats).  Returns the total in bytes.
static size_t lds_bytes(int MT, int NT, int NW, int KS, int ntaps, int Cmain, int Cskip, bool has_gn, int* qs_off = nullptr) {
    const int ROWS = 16 * MT, COLS = 16 * NT;
    const int nchunks = ntaps * (Cmain / 16) + Cskip / 16;
    const size_t idx = (size_t)(((ntaps + 1) * ROWS + 3) & ~3) * 4 + (size_t)rec_capacity(nchunks, NW, KS) * 16;
    const size_t coef = has_gn ? (size_t)24 * Cmain : 0;
    const size_t part = (size_t)MT * NT * 4 * 64 * 4;                 // one wave's partial tile, lane-linear (tree reduction)
    const size_t fin = (size_t)ROWS * (COLS + 4) * 4;                 // a (row, col) tile image: finished tile / ONE_STAGE partial
    const bool one_stage = NW > 1 && NW * fin <= 48 * 1024;
    const size_t redu = one_stage ? NW * fin + fin : ((size_t)(NW / 2) * part > fin ? (size_t)(NW / 2) * part : fin);
    size_t r = idx + coef;
    if (redu > r) r = redu;
    r = (r + 15) & ~(size_t)15;
    if (qs_off) *qs_off = (int)(r / 4);
    return r + (size_t)3 * (COLS / 4) * 2 * 8;                       // + per-(plane, quad) statistics
}

// k_conv_lds: tiles are encoded as ConvTile{WM, WN, NW = 32, KS = 1, XM = 0}
static size_t lds_bytes_tiled(int WM, int WN, int ntaps, int Cmain, int Cskip, bool has_gn, int* qs_off = nullptr) {
    const int BM = 32 * WM, BN = 32 * WN;
    const int nchunks = ntaps * (Cmain / 16) + Cskip / 16;
    size_t fl = (size_t)((nchunks + 3) & ~3) * 4 + (size_t)(((ntaps + 1) * BM + 3) & ~3) + (has_gn ? (size_t)6 * Cmain : 0) +
                (size_t)2 * BM * 20 + (size_t)2 * 16 * (BN + 4);
    fl = (fl + 3) & ~(size_t)3;
    if (qs_off) *qs_off = (int)fl;
    return fl * 4 + (size_t)3 * (BN / 4) * 2 * 8;
}

bool conv_lds_eligible(const ConvArgs& a) {
    if (a.out_cm || a.ddim || (a.N & 3) || a.N < 64) return false;
    for (int t = 0; t < a.nstat; ++t)
        if (a.stat[t].gs & 3) return false;
    return true;
}

template <int WM, int WN>
static hipError_t launch_conv_lds_t(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    constexpr int BM = 32 * WM, BN = 32 * WN;
    const int tiles = (a.Lout + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const long nblk = (long)a.B * tiles * tiles_n;
    if (nblk >= (1L << 21) || !conv_lds_eligible(a)) return hipErrorInvalidValue;
    const int nchunks = a.ntaps * (a.Cmain / 16) + a.Cskip / 16;
    a.KS = 1;
    a.xmap = 0;
    a.tiles_per_b = tiles;
    a.tiles_n = tiles_n;
    a.Bt = a.B * tiles;
    a.inv_tiles_per_b = 1.0f / (float)tiles;
    a.inv_tiles_n = 1.0f / (float)tiles_n;
    a.inv_Bt = 1.0f / (float)a.Bt;
    a.cpt = a.Cmain / 16;
    a.inv_cpt = 1.0f / (float)a.cpt;
    a.geo_inv_r = a.geo_r > 0 ? 1.0f / (float)a.geo_r : 0.f;
    a.cps_q = nchunks;
    a.cps_r = 0;
    a.rec_cap = (nchunks + 3) & ~3;
    const size_t smem = lds_bytes_tiled(WM, WN, a.ntaps, a.Cmain, a.Cskip, a.gn.sums != nullptr, &a.qs_off);
    hipLaunchKernelGGL((k_conv_lds<WM, WN>), dim3((unsigned)nblk), dim3(256), smem, s, a);
    return hipGetLastError();
}

ConvTile conv_pick_tile(int B, int Lout, int N, int nchunks, int Cmain, bool has_gn) {
    static int forced[4] = {-1, 0, 0, 0};   // tuning aid: MTV_FORCE_TILE="MT,NT,NW,KS"
    if (forced[0] == -1) {
        forced[0] = 0;
        if (const char* e = getenv("MTV_FORCE_TILE")) {
            int a = 0, b = 0, c = 0, d = 1;
            if (sscanf(e, "%d,%d,%d,%d", &a, &b, &c, &d) >= 3) { forced[0] = a; forced[1] = b; forced[2] = c; forced[3] = d < 1 ? 1 : d; }
        }
    }
    if (forced[0] > 0) {
        ConvTile t{forced[0], forced[1], forced[2], forced[3], 0};
        while (t.NW * t.KS > nchunks && t.KS > 1) t.KS /= 2;
        while (t.NW * t.KS > nchunks && t.NW > 1) t.NW /= 2;
        return t;
    }
    static double lat_us = -1.0, fin_us = 3.0;
    if (lat_us < 0) {
        lat_us = 0.7;
        if (const char* e = getenv("MTV_LAT_US")) lat_us = atof(e);
        if (const char* e = getenv("MTV_FIN_US")) fin_us = atof(e);
    }
    static const int cand[][2] = {{4, 4}, {2, 4}, {1, 4}, {2, 2}, {1, 2}, {1, 1}};
    const int ntaps_guess = 9;
    ConvTile best{1, 1, 1, 1, 0};
    double best_t = 1e30;
    for (auto& c : cand) {
        const int MT = c[0], NT = c[1];
        if (NT > 1 && NT * 8 >= N) continue;                       // more than half of the tile's columns would be padding
        if (MT > 1 && 16 * (MT / 2) >= Lout) continue;             // tile taller than needed
        const double tiles = (double)B * ((Lout + 16 * MT - 1) / (16 * MT)) * ((N + 16 * NT - 1) / (16 * NT));
        for (int NW = 1; NW <= 16; NW *= 2) {
            if (lds_bytes(MT, NT, NW, 1, ntaps_guess, Cmain, 0, has_gn) > 96 * 1024) continue;
            if (NW == 16 && MT * NT >= 8) continue;                   // 1024-thread blocks cap VGPRs at 128: these spill
            for (int KS = 1; KS <= 16; KS *= 2) {
                const int slices = NW * KS;
                if (slices > nchunks || (KS > 1 && (N & 3))) continue;
                const double waves = tiles * slices;
                const double cps = (double)nchunks / slices;
                const double t_mfma = tiles * nchunks * (4.0 * MT * NT) * 32.0 / (waves < 1024 ? waves : 1024.0) / 2400.0;
                const double t_lat = cps * lat_us;
                const double t_l2 = tiles * nchunks * (MT + NT) * 1024.0 / 12e6;       // bytes / (12 TB/s) in us
                double t = t_mfma > t_lat ? t_mfma : t_lat;
                t = t > t_l2 ? t : t_l2;
                t += 2.0 + 0.15 * (NW > 1 ? __builtin_ctz(NW) : 0);
                if (KS > 1)   // slab round trip (write + KS-fold read by the last slice) + ticket
                    t += fin_us + (double)(KS + 1) * B * Lout * N * 4.0 / 3e6 + 0.05 * KS;
                t += 0.002 * waves / 64;                                              // dispatch cost of very wide grids
                if (t < best_t - 1e-9) {
                    best_t = t;
                    best = ConvTile{MT, NT, NW, KS, 0};
                }
            }
        }
    }
    return best;
}

size_t conv_smem_bytes(const ConvArgs& a, ConvTile t) {
    if (t.NW == 80) return conv_win_smem_bytes(a, t);
    if (t.NW == 96) return conv_pw_smem_bytes(a, t);
    if (t.NW == 64) return lin_smem_bytes(a);
    if (t.NW == 48) return conv_x3_smem_bytes(a, t);
    if (t.NW == 32) return lds_bytes_tiled(t.MT, t.NT, a.ntaps, a.Cmain, a.Cskip, a.gn.sums != nullptr);
    return lds_bytes(t.MT, t.NT, t.NW, t.KS, a.ntaps, a.Cmain, a.Cskip, a.gn.sums != nullptr);
}

template <int MT, int NT, int NW>
static hipError_t launch_conv_t(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const int tiles = (a.Lout + 16 * MT - 1) / (16 * MT);
    const int tiles_n = (a.N + 16 * NT - 1) / (16 * NT);
    const long nblk = a.xmap ? 8L * ((tiles_n * a.KS + 7) / 8) * a.B * tiles : (long)a.B * tiles * tiles_n * a.KS;
    if (nblk >= (1L << 21)) return hipErrorInvalidValue;         // (the kernel's float-reciprocal decode is exact below 2^22)
    // everything the kernel would otherwise divide for
    const int nchunks = a.ntaps * (a.Cmain / 16) + a.Cskip / 16, ns = NW * a.KS;
    a.tiles_per_b = tiles;
    a.tiles_n = tiles_n;
    a.Bt = a.B * tiles;
    a.Sx = (tiles_n * a.KS + 7) / 8;
    a.inv_tiles_per_b = 1.0f / (float)tiles;
    a.inv_tiles_n = 1.0f / (float)tiles_n;
    a.inv_Bt = 1.0f / (float)a.Bt;
    a.inv_Sx = 1.0f / (float)a.Sx;
    a.cpt = a.Cmain / 16;
    a.inv_cpt = 1.0f / (float)a.cpt;
    a.geo_inv_r = a.geo_r > 0 ? 1.0f / (float)a.geo_r : 0.f;
    a.cps_q = nchunks / ns;
    a.cps_r = nchunks % ns;
    a.rec_cap = rec_capacity(nchunks, NW, a.KS);
    dim3 grid((unsigned)nblk);
    const size_t smem = lds_bytes(MT, NT, NW, a.KS, a.ntaps, a.Cmain, a.Cskip, a.gn.sums != nullptr, &a.qs_off);
    hipLaunchKernelGGL((k_conv<MT, NT, NW>), grid, dim3(NW * 64), smem, s, a);
    return hipGetLastError();
}

template <int MT, int NT>
static hipError_t launch_conv_nw(const ConvArgs& a, int NW, hipStream_t s) {
    switch (NW) {
        case 1: return launch_conv_t<MT, NT, 1>(a, s);
        case 2: return launch_conv_t<MT, NT, 2>(a, s);
        case 4: return launch_conv_t<MT, NT, 4>(a, s);
        case 8: return launch_conv_t<MT, NT, 8>(a, s);
        case 16:
            if constexpr (MT * NT >= 8) return hipErrorInvalidValue;   // 1024-thread blocks cap VGPRs at 128: these would spill
            else return launch_conv_t<MT, NT, 16>(a, s);
    }
    return hipErrorInvalidValue;
}

template <int MT, int NT, int NW>
static hipError_t conv_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv<MT, NT, NW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
}
template <int MT, int NT>
static hipError_t conv_attr_nw() {
    hipError_t e;
    if ((e = conv_attr<MT, NT, 1>()) != hipSuccess) return e;
    if ((e = conv_attr<MT, NT, 2>()) != hipSuccess) return e;
    if ((e = conv_attr<MT, NT, 4>()) != hipSuccess) return e;
    if ((e = conv_attr<MT, NT, 8>()) != hipSuccess) return e;
    if constexpr (MT * NT >= 8) return hipSuccess;
    else return conv_attr<MT, NT, 16>();
}
// Dynamic LDS above 64 KB must be opted into once per kernel; done at mtv_create (never under capture).
hipError_t conv_init_attrs() {
    hipError_t e;
    if ((e = conv_attr_nw<4, 4>()) != hipSuccess) return e;
    if ((e = conv_attr_nw<2, 4>()) != hipSuccess) return e;
    if ((e = conv_attr_nw<1, 4>()) != hipSuccess) return e;
    if ((e = conv_attr_nw<2, 2>()) != hipSuccess) return e;
    if ((e = conv_attr_nw<1, 2>()) != hipSuccess) return e;
    if ((e = conv_attr_nw<1, 1>()) != hipSuccess) return e;
    const void* tiled[] = {reinterpret_cast<const void*>(&k_conv_lds<4, 4>), reinterpret_cast<const void*>(&k_conv_lds<2, 4>),
                           reinterpret_cast<const void*>(&k_conv_lds<4, 2>), reinterpret_cast<const void*>(&k_conv_lds<2, 2>),
                           reinterpret_cast<const void*>(&k_conv_lds<2, 8>), reinterpret_cast<const void*>(&k_conv_lds<4, 8>)};
    for (const void* f : tiled)
        if ((e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)) != hipSuccess) return e;
    return conv_x3_init_attrs();
}

hipError_t launch_conv(const ConvArgs& a0, ConvTile t, hipStream_t s) {
    ConvArgs a = a0;
    // timing experiment (results unchanged): MTV_DEBUG_STATDUP=1 makes every workgroup issue its GroupNorm-statistics atomics a
    // SECOND time, into the other step parity's arena (unused during this step, zeroed by this step's head conv): the step
    // time it adds is what the real ones cost
    static const bool stat_dup = getenv("MTV_DEBUG_STATDUP") != nullptr;
    if (!stat_dup) a.dbg_stat_dup = 0;
    a.KS = t.KS;
    a.xmap = t.XM;
    if (a.KS > 1 && (!a.slab || !a.tickets || (a.N & 3))) return hipErrorInvalidValue;
    if (a.ddim) {
        if (!a.out_cm || a.N != 4) return hipErrorInvalidValue;
        a.xmap = 0;            // no padding blocks: every block of the grid takes part in the step hand-over
    }
    hipError_t e = hipErrorInvalidValue;
    if (t.NW == 80) {            // window-staged 3x3 kernel of the large levels (deep.hip: k_conv_win)
        if (conv_win_eligible(a, t.MT, t.NT, t.KS)) return launch_conv_win(a, t, s);
        t = conv_pick_tile(a.B, a.Lout, a.N, a.ntaps * (a.Cmain / 16) + a.Cskip / 16, a.Cmain, a.gn.sums != nullptr);   // (statistics targets are
        t.KS = 1;                                                                                                       // attached after the op is created)
        a.KS = 1;
        a.xmap = t.XM;
    }
    if (t.NW == 96) {            // 1x1 kernel of the large levels (deep.hip: k_conv_pw)
        if (conv_pw_eligible(a, t.MT, t.NT, t.KS)) return launch_conv_pw(a, t, s);
        t = conv_pick_tile(a.B, a.Lout, a.N, a.ntaps * (a.Cmain / 16) + a.Cskip / 16, a.Cmain, a.gn.sums != nullptr);
        t.KS = 1;
        a.KS = 1;
        a.xmap = t.XM;
    }
    if (t.NW == 48) {            // split-bf16 kernels for large token counts (conv_x3.hip: elementwise pass + gathering GEMM)
        if (conv_x3_eligible(a) && a.x3) return launch_conv_x3(a, t, s);
        t = conv_pick_tile(a.B, a.Lout, a.N, a.ntaps * (a.Cmain / 16) + a.Cskip / 16, a.Cmain, a.gn.sums != nullptr);
        t.KS = 1;
        a.KS = 1;
        a.xmap = t.XM;
    }
    if (t.NW == 64) {            // lean 1x1 kernel (lin.hip)
        if (conv_lin_eligible(a)) return launch_lin(a, t, s);
        t = conv_pick_tile(a.B, a.Lout, a.N, a.ntaps * (a.Cmain / 16) + a.Cskip / 16, a.Cmain, a.gn.sums != nullptr);   // (statistics targets are
        t.KS = 1;                                                                                                       // attached after the op is created)
        a.KS = 1;
        a.xmap = t.XM;
    }
    if (t.NW == 32 && !conv_lds_eligible(a)) {      // (statistics targets are attached after a plan's op is created)
        t = conv_pick_tile(a.B, a.Lout, a.N, a.ntaps * (a.Cmain / 16) + a.Cskip / 16, a.Cmain, a.gn.sums != nullptr);
        t.KS = 1;
        a.KS = 1;
        a.xmap = t.XM;
    }
    if (t.NW == 32) {            // LDS-tiled kernel
        if (t.MT == 4 && t.NT == 4) return launch_conv_lds_t<4, 4>(a, s);
        if (t.MT == 2 && t.NT == 4) return launch_conv_lds_t<2, 4>(a, s);
        if (t.MT == 4 && t.NT == 2) return launch_conv_lds_t<4, 2>(a, s);
        if (t.MT == 2 && t.NT == 2) return launch_conv_lds_t<2, 2>(a, s);
        if (t.MT == 2 && t.NT == 8) return launch_conv_lds_t<2, 8>(a, s);
        if (t.MT == 4 && t.NT == 8) return launch_conv_lds_t<4, 8>(a, s);
        return hipErrorInvalidValue;
    }
    if (t.MT == 4 && t.NT == 4) e = launch_conv_nw<4, 4>(a, t.NW, s);
    else if (t.MT == 2 && t.NT == 4) e = launch_conv_nw<2, 4>(a, t.NW, s);
    else if (t.MT == 1 && t.NT == 4) e = launch_conv_nw<1, 4>(a, t.NW, s);
    else if (t.MT == 2 && t.NT == 2) e = launch_conv_nw<2, 2>(a, t.NW, s);
    else if (t.MT == 1 && t.NT == 2) e = launch_conv_nw<1, 2>(a, t.NW, s);
    else if (t.MT == 1 && t.NT == 1) e = launch_conv_nw<1, 1>(a, t.NW, s);
    return e;
}

}  // namespace mtv
