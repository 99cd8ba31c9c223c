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

// K walk.  K is cut into SEGMENTS = (tap, source part) pairs -- the tapped parts for every tap, then
// the parts of the fused 1x1 skip conv -- and each segment into chunks of 16 channels.  The segment
// descriptors are built once per workgroup in LDS; a Cursor keeps the wave-uniform state in SGPRs
// (forced with readfirstlane: the compiler cannot prove uniformity of anything derived from the wave
// id) and the per-lane row offsets in VGPRs.  Advancing inside a segment is two scalar pointer bumps.
#ifndef MTV_ABLATE
#define MTV_ABLATE 0          // tools/ubench/conv_bench builds ablated variants: 1 no MFMA, 2 no A loads, 4 no B loads,
                              // 8 no transform, 16 prologue only, 32 no reduction/epilogue
#endif

struct SegDesc {              // 32 bytes, one per segment, in LDS
    unsigned src_lo, src_hi;  // base pointer of the source part
    int Cp;                   // channels of the part
    int Ls;                   // tokens per batch element of the source
    int crow;                 // W row of channel 0 of this segment
    int coff;                 // concat channel of channel 0 (coefficient index)
    int tap;                  // row of the index table (== ntaps for skip segments)
    int skip;
};

__device__ __forceinline__ int usgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }

#if MTV_ABLATE & 64   // phase timestamps of thread 0 of four sampled blocks (first two, middle, last) -> a.dbg (conv_bench, MTV_STAMPS)
#define MTV_STAMP(k) do { if (threadIdx.x == 0 && a.dbg) { const int sb_ = blockIdx.x < 2 ? (int)blockIdx.x : (blockIdx.x == gridDim.x / 2 ? 2 : (blockIdx.x == gridDim.x - 1 ? 3 : -1)); \
                          if (sb_ >= 0) a.dbg[sb_ * 8 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define MTV_STAMP(k) do { } while (0)
#endif

typedef __attribute__((address_space(1))) const char gchar;   // global (not flat) loads: vmcnt only, saddr addressing

template <int MT>
struct Cursor {
    int seg, c, cend, coff, skip;      // uniform
    gchar* abase;                      // uniform: part base + c * 4
    gchar* wbase;                      // uniform: W + (crow + c) * ldw * 4
    unsigned rowoff[MT];               // per lane: ((b * Ls + st) * Cp + 4q) * 4
    int e[MT];                         // per lane: index-table entries of this lane's rows for this tap
};

// uniform half of entering a segment (needs only the segment table): everything the W loads need
template <int MT>
__device__ __forceinline__ void enter_segment_u(Cursor<MT>& k, const SegDesc* segs, const float* W, int ldw, int c) {
    const SegDesc* d = segs + k.seg;
    const unsigned lo = (unsigned)usgpr((int)d->src_lo), hi = (unsigned)usgpr((int)d->src_hi);
    const int Cp = usgpr(d->Cp), crow = usgpr(d->crow);
    k.coff = usgpr(d->coff);
    k.skip = usgpr(d->skip);
    k.c = c;
    k.cend = Cp;
    gchar* sp = (gchar*)(((unsigned long long)hi << 32) | lo);
    k.abase = sp + (size_t)c * 4;
    k.wbase = (gchar*)(unsigned long long)W + (size_t)(crow + c) * (size_t)ldw * 4;
}

// per-lane half: byte offsets of this lane's A rows for the segment's tap.  Normally read from the LDS
// index table; `ident` (no gather tables at all: 1x1 convs) derives them from the token index, which
// needs no memory at all -- such convs start their A loads before the prologue.
template <int MT>
__device__ __forceinline__ void enter_segment_rows(Cursor<MT>& k, const SegDesc* segs, const int* idx, int rows, int b, int i, int q,
                                                   bool ident, int tok0, int Lout, const SegInfo& sgi) {
    const SegDesc* d = segs + k.seg;
    const int Cp = usgpr(d->Cp), Ls = usgpr(d->Ls), tap = usgpr(d->tap);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int e;
        if (ident) {
            const int tok = tok0 + 16 * mt + i;
            e = tok < Lout ? (k.skip ? tok : (tok | (seg_of(sgi, tok) << 28))) : -1;
        } else {
            e = idx[tap * rows + 16 * mt + i];
        }
        k.e[mt] = e;
        const unsigned st = e < 0 ? 0u : (unsigned)(e & 0x0FFFFFFF);   // padded rows read a valid address, zeroed later
        k.rowoff[mt] = (((unsigned)b * (unsigned)Ls + st) * (unsigned)Cp + 4u * q) * 4u;
    }
}

template <int MT, int NT>
struct Raw {
    f32x4 a[MT];
    int e[MT];
    float b[4][NT];
    int cc, skip;     // uniform: coefficient channel of this chunk, skip flag
};

template <int MT, int NT>
__device__ __forceinline__ void load_b(const Cursor<MT>& k, unsigned woff, unsigned ldw4, Raw<MT, NT>& o) {
    o.cc = k.coff + k.c;
    o.skip = k.skip;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        gchar* p = k.wbase + (woff + (unsigned)s * ldw4);
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
}

template <int MT, int NT>
__device__ __forceinline__ void load_a(const Cursor<MT>& k, Raw<MT, NT>& o) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        o.e[mt] = k.e[mt];
        if constexpr (MTV_ABLATE & 2) o.a[mt] = f32x4{1.f, 2.f, 3.f, 4.f};
        else o.a[mt] = *(const __attribute__((address_space(1))) f32x4*)(k.abase + k.rowoff[mt]);
    }
}

template <int MT, int NT>
__device__ __forceinline__ void load_chunk(const Cursor<MT>& k, unsigned woff, unsigned ldw4, Raw<MT, NT>& o) {
    load_b<MT, NT>(k, woff, ldw4, o);
    load_a<MT, NT>(k, o);
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
        const int g = (a.stat[t].coff + n) / a.stat[t].gs;
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
    const int tiles_per_b = (a.Lout + ROWS - 1) / ROWS;
    // ---- sampler-step head (mtv_internal.h DdimFuse): the hand-over record and the step index are requested at
    // entry and used once this workgroup's operand ring is in flight (head_duties below), so their latency is hidden
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
    const int tiles_n = (a.N + COLS - 1) / COLS;
    const int Bt = a.B * tiles_per_b;
    int bx, by, bz;
    if (a.xmap == 0) {
        bx = blockIdx.x % Bt;
        const int rest = blockIdx.x / Bt;
        by = rest % tiles_n;
        bz = rest / tiles_n;
    } else {
        const int S = tiles_n * a.KS, Sx = (S + 7) >> 3;
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int sl = xcd + 8 * (j % Sx);
        if (sl >= S) return;                       // padding block (whole workgroup, before any barrier)
        bx = j / Sx;
        by = sl % tiles_n;
        bz = sl / tiles_n;
    }
    const int b = bx / tiles_per_b;
    const int tok0 = (bx - b * tiles_per_b) * ROWS;
    const int n0 = by * COLS;
    const int Cmain = a.Cmain;
    const bool do_gn = a.gn.sums != nullptr;

    MTV_STAMP(0);
    // ---- LDS: segment descriptors, source-token table [(ntaps+1)][ROWS], folded per-(plane, channel) affine {A, B}
    constexpr int MAXSEG = 24;
    SegDesc* segs = reinterpret_cast<SegDesc*>(smem);
    int* idx = reinterpret_cast<int*>(smem + MAXSEG * 8);
    const int idx_floats = ((a.ntaps + 1) * ROWS + 3) & ~3;
    float2* coef = reinterpret_cast<float2*>(smem + MAXSEG * 8 + idx_floats);
    const int nseg_main = a.ntaps * a.nmain, nseg = nseg_main + a.nskip;
    if (tid < nseg) {
        const bool skip = tid >= nseg_main;
        const int local = skip ? tid - nseg_main : tid;
        const int nparts = skip ? a.nskip : a.nmain;
        const int tap = skip ? a.ntaps : (nparts > 1 ? local >> 1 : local);
        const bool second = nparts > 1 && (local & 1);
        const float* sp = skip ? (second ? a.src[3] : a.src[2]) : (second ? a.src[1] : a.src[0]);
        SegDesc d;
        d.src_lo = (unsigned)(reinterpret_cast<unsigned long long>(sp) & 0xFFFFFFFFull);
        d.src_hi = (unsigned)(reinterpret_cast<unsigned long long>(sp) >> 32);
        d.Cp = skip ? (second ? a.C[3] : a.C[2]) : (second ? a.C[1] : a.C[0]);
        d.Ls = skip ? a.Lskip : a.Lsrc;
        d.coff = second ? (skip ? a.C[2] : a.C[0]) : 0;
        d.crow = (skip ? a.ntaps * Cmain : tap * Cmain) + d.coff;
        d.tap = tap;
        d.skip = skip ? 1 : 0;
        segs[tid] = d;
    }
    // source-token table.  Identity / arithmetic gathers need no memory, so the table is complete before the
    // first barrier and the operand ring can start right away; table gathers are fetched during the prologue.
    const bool ident = a.gather == nullptr && a.gather_skip == nullptr;
    const bool no_tab = (a.gather == nullptr || a.geo_main != 0) && (a.gather_skip == nullptr || a.geo_skip != 0);
    auto skip_src = [&](int tok) -> int {
        if (a.geo_skip) return geo_source(a.geo_r, a.geo_t, tok, 1, 1, true) & 0x0FFFFFFF;
        return a.gather_skip ? a.gather_skip[tok] : tok;
    };
    auto build_idx = [&]() {
        for (int e = tid; e < (a.ntaps + 1) * ROWS; e += NTH) {
            const int t = e / ROWS, r = e - t * ROWS;
            const int tok = tok0 + r;
            int v = -1;
            if (tok < a.Lout) {
                if (t < a.ntaps) {
                    if (a.geo_main) {
                        const int ky = t / 3;
                        v = geo_source(a.geo_r, a.geo_t, tok, ky, t - 3 * ky, a.geo_main == 2);
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
    };
    if (no_tab && !ident) build_idx();
    __syncthreads();

    // ---- this wave's chunk range; the W fragment of its first chunk is requested NOW, so the (HBM-cold)
    // weight latency overlaps the rest of the prologue
    const int nchunks = a.ntaps * (Cmain >> 4) + (a.Cskip >> 4);
    const int slice = bz * NW + wave, nslices = a.KS * NW;
    const int ch0 = usgpr((int)(((unsigned)nchunks * (unsigned)slice) / (unsigned)nslices));
    const int ch1 = usgpr((int)(((unsigned)nchunks * (unsigned)(slice + 1)) / (unsigned)nslices));
    const int ldw = a.ldw;
    const float* Wp = a.W;
    const unsigned ldw4 = (unsigned)ldw * 4u;
    const unsigned woff = (4u * q * (unsigned)ldw + (unsigned)(n0 + NT * i)) * 4u;
    // chunks in flight per wave, bounded by the register budget (1024-thread blocks get 128 VGPRs)
    constexpr int DEPTH = NW == 16 ? (MT * NT >= 4 ? 2 : (MT * NT >= 2 ? 3 : 4)) : (MT * NT >= 16 ? 2 : (MT * NT >= 8 ? 3 : 4));
    Cursor<MT> cur;
    Raw<MT, NT> ring[DEPTH];
    const int Lout_ = a.Lout;
    const SegInfo sgi = a.seg_src;
    auto advance = [&]() {
        cur.c += 16;
        if (cur.c < cur.cend) {
            cur.abase += 64;
            cur.wbase += (size_t)64 * (size_t)ldw;
        } else {
            cur.seg += 1;
            enter_segment_u<MT>(cur, segs, Wp, ldw, 0);
            enter_segment_rows<MT>(cur, segs, idx, ROWS, b, i, q, ident, tok0, Lout_, sgi);
        }
    };
    auto fill_ring = [&]() {                 // slots 1..DEPTH-1 (slot 0 is loaded separately)
#pragma unroll
        for (int d = 1; d < DEPTH; ++d)
            if (d < ch1 - ch0) {
                advance();
                load_chunk<MT, NT>(cur, woff, ldw4, ring[d]);
            }
    };
    if (ch0 < ch1) {
        int seg = 0, left = ch0;
        while (seg + 1 < nseg) {
            const int cp16 = usgpr(segs[seg].Cp) >> 4;
            if (left < cp16) break;
            left -= cp16;
            ++seg;
        }
        cur.seg = seg;
        enter_segment_u<MT>(cur, segs, Wp, ldw, left << 4);
        load_b<MT, NT>(cur, woff, ldw4, ring[0]);
        if (no_tab) {                        // the index table (if any) is ready: the whole ring starts now
            enter_segment_rows<MT>(cur, segs, idx, ROWS, b, i, q, ident, tok0, Lout_, sgi);
            load_a<MT, NT>(cur, ring[0]);
            fill_ring();
        }
    }

    DdimStep stp{};
    if (dd) {
        // every workgroup of the launch zeroes its slice of the OTHER parity's statistics arena and copies its slice
        // of the next step's FiLM row (nobody reads either any more in this step): fire-and-forget stores that
        // drain under the conv itself; the step's scalars are fetched for the epilogue
        for (long long e = (long long)blockIdx.x * NTH + tid; e < ddv.zero_vec4; e += (long long)gridDim.x * NTH)
            *reinterpret_cast<f32x4*>(ddv.zero_arena + 4 * e) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (step_now + 1 < ddv.n_steps) {
            const float* src = ddv.film_tab + (size_t)(step_now + 1) * ddv.film_total;
            for (int e = blockIdx.x * NTH + tid; e < ddv.film_total; e += gridDim.x * NTH) ddv.film_out[e] = src[e];
        }
        stp = ddv.steps[step_now];
    }
    MTV_STAMP(1);
    if (!no_tab) build_idx();
    if (do_gn) {
        // per-channel inputs of the first round are requested BEFORE the statistics are reduced, so
        // the two global-memory latencies overlap
        constexpr int PER = 4;                          // channels per thread per round
        float ga[PER], be[PER], s1[PER], sh[PER];
        const float* film = a.gn.film ? a.gn.film + (size_t)b * a.gn.film_stride : nullptr;
        auto fetch = [&](int c0) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                ga[k] = a.gn.gamma[c0 + k];
                be[k] = a.gn.beta[c0 + k];
                s1[k] = film ? 1.0f + film[c0 + k] : 1.0f;
                sh[k] = film ? film[Cmain + c0 + k] : 0.0f;
            }
        };
        if (tid * PER < Cmain) fetch(tid * PER);
        // (plane, group) sums over the privatised copies: 8 independent 16-byte loads per thread, all in flight
        // together.  A cross-plane site (AttentionBlock1D) adds its three planes up through LDS afterwards --
        // summing 24 entries per thread in registers made the compiler serialise the loads.
        const bool whole = a.gn.whole != 0;
        f64x2 v0 = {0.0, 0.0};
        if (tid < 96) {
#pragma unroll
            for (int k = 0; k < STAT_COPIES; ++k)
                v0 += *reinterpret_cast<const f64x2*>(a.gn.sums + (size_t)k * a.gn.cstride + (size_t)b * 192 + (size_t)tid * 2);
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
            double n;
            if (whole) {
                v = (s_dp[g] + s_dp[32 + g]) + s_dp[64 + g];
                n = (double)a.seg_src.L * a.gn.gs;
            } else {
                v = NTH < 96 ? s_dp[e] : v0;
                const int len = sg == 0 ? a.seg_src.b1 : (sg == 1 ? a.seg_src.b2 - a.seg_src.b1 : a.seg_src.L - a.seg_src.b2);
                n = (double)len * a.gn.gs;
            }
            const double mean = v[0] / n;
            double var = v[1] / n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mr[sg][g] = make_float2((float)mean, 1.0f / sqrtf((float)var + 1e-5f));   // fp64 only where cancellation bites
        }
        __syncthreads();
        for (int c0 = tid * PER; c0 < Cmain; c0 += NTH * PER) {
            if (c0 != tid * PER) fetch(c0);
#pragma unroll
            for (int sg = 0; sg < 3; ++sg)
#pragma unroll
                for (int k = 0; k < PER; ++k) {
                    const float2 mr = s_mr[sg][(c0 + k) / a.gn.gs];
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
                const int rs = skip_src(tok);
                pre_res = *reinterpret_cast<const f32x4*>(a.res + ((size_t)b * a.Lskip + rs) * a.N + n);
            }
        }
    }

    if (ch0 < ch1) {
        if (!no_tab) {
            enter_segment_rows<MT>(cur, segs, idx, ROWS, b, i, q, false, tok0, Lout_, sgi);
            load_a<MT, NT>(cur, ring[0]);
            fill_ring();
        }
        // ring of DEPTH chunks in flight.  Steady state has no conditionals, so every ring slot keeps
        // fixed registers and the loads of the next DEPTH-1 chunks stay in flight under the MFMAs.
        int n = ch1 - ch0;                       // chunks not yet multiplied (ring slot 0 holds the first)
        while (n >= 2 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                mma_chunk<MT, NT>(coef, Cmain, do_gn, act, q, ring[d], acc);
                advance();
                load_chunk<MT, NT>(cur, woff, ldw4, ring[d]);
            }
            n -= DEPTH;
        }
        // drain: n < 2*DEPTH chunks left, min(n, DEPTH) of them already in the ring
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d < n) {
                mma_chunk<MT, NT>(coef, Cmain, do_gn, act, q, ring[d], acc);
                if (d + DEPTH < n) {
                    advance();
                    load_chunk<MT, NT>(cur, woff, ldw4, ring[d]);
                }
            }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d + DEPTH < n) mma_chunk<MT, NT>(coef, Cmain, do_gn, act, q, ring[d], acc);
    }

    MTV_STAMP(3);
    // ---- fixed-order tree over the NW waves (lane-linear LDS image: conflict-free), LDS reused
    __syncthreads();
    float* red = smem;
    constexpr int TILE_REGS = MT * NT * 4;
    constexpr int LDR = COLS + 4;
    constexpr bool ONE_STAGE = NW > 1 && NW * TILE_REGS * 64 * 4 <= 48 * 1024;   // all partials fit in LDS at once
    if constexpr (NW == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) red[(16 * mt + 4 * q + r) * LDR + NT * i + nb] = acc[mt][nb][r];
        __syncthreads();
    } else if constexpr (ONE_STAGE) {
        // every wave parks its partial tile (lane-linear image: conflict-free); pass 1 sums them in wave order
        float* my = red + (size_t)wave * TILE_REGS * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nb = 0; nb < NT; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) my[((mt * NT + nb) * 4 + r) * 64] = acc[mt][nb][r];
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
    float* fin = ONE_STAGE ? red + (size_t)NW * TILE_REGS * 64 : red;   // (row, col) image of the finished tile for pass 2
    auto tile_quad = [&](int rr, int cq) -> f32x4 {                     // this workgroup's (partial) result for one quad
        if constexpr (ONE_STAGE) {
            // element (rr, 4cq+k) of wave w sits at w*TILE + ((mt*NT+nb)*4 + r)*64 + lane(i, q)
            const int mt = rr >> 4, qq = (rr >> 2) & 3, r = rr & 3;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int col = cq * 4 + k, ii = col / NT, nb = col - ii * NT;
                    v[k] += red[(size_t)w * TILE_REGS * 64 + ((mt * NT + nb) * 4 + r) * 64 + qq * 16 + ii];
                }
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
            const int rs = a.res ? skip_src(tok) : tok;   // (the LDS index table is gone by now)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (n + k < a.N) v[k] = epi(a, v[k], b, tok, rs, n + k);
        }
        if (!a.out_cm && n + 3 < a.N) {
            *reinterpret_cast<f32x4*>(a.out + ((size_t)b * a.Lout + tok) * a.N + n) = v;
        } else {
            for (int k = 0; k < 4 && n + k < a.N; ++k) {
                if (dd) {
                    // sampler-step head (N == 4 == the sample's channels): v is eps; x <- DDIM update, in the external
                    // layout and as channels 0..3 of the packed input of the next step's stem conv
                    const size_t ix = ((size_t)b * a.N + n + k) * a.Lout + tok;
                    const float nz = stp.noise_index >= 0 ? ddv.noise[(size_t)stp.noise_index * ddv.n_per_draw + ix] : 0.f;
                    const float xn = ddim_update_elem(stp, ddv.x[ix], v[k], nz);
                    ddv.x[ix] = xn;
                    ddv.h0[((size_t)b * a.Lout + tok) * 16 + n + k] = xn;
                } else if (a.out_cm) a.out[((size_t)b * a.N + n + k) * a.Lout + tok] = v[k];
                else a.out[((size_t)b * a.Lout + tok) * a.N + n + k] = v[k];
            }
        }
        if (want_stats) *reinterpret_cast<f32x4*>(fin + rr * LDR + cq * 4) = v;
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

    // ---- pass 2 (column-major over the tile): GroupNorm statistics of the output for its consumers.
    // Rows are reduced with wave shuffles, the channel quads of one group are then combined in LDS, so a
    // workgroup issues ONE atomic pair per (plane, group, consumer) -- the per-address atomic rate is what
    // bounds this epilogue (measured: quads of a wide group hitting one address cost 5x the whole kernel).
    __syncthreads();
    constexpr int W = ROWS < 64 ? ROWS : 64;         // lanes that share one channel quad
    double* qs = reinterpret_cast<double*>(fin + ROWS * LDR);   // [3 planes][QPR quads][2]
    bool fast = true;
    for (int t = 0; t < a.nstat; ++t) fast = fast && ((a.stat[t].gs & 3) == 0);
    if (fast) {
        for (int e = tid; e < 3 * QPR * 2; e += NTH) qs[e] = 0.0;
        __syncthreads();
    }
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
            if (fast) {
                double s = mine ? ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]) : 0.0;
                double ss = mine ? ((double)v[0] * v[0] + (double)v[1] * v[1]) + ((double)v[2] * v[2] + (double)v[3] * v[3]) : 0.0;
#pragma unroll
                for (int o = W / 2; o >= 1; o >>= 1) {
                    s += __shfl_xor(s, o);
                    ss += __shfl_xor(ss, o);
                }
                // W == 64: one wave per quad (exclusive slot).  W < 64: several waves may hold rows of the
                // same quad only when ROWS > 64, which never happens (ROWS <= 64) -> exclusive as well.
                if ((lane & (W - 1)) == 0 && e < QUADS && n < a.N) {
                    qs[(sgi * QPR + cq) * 2] = s;
                    qs[(sgi * QPR + cq) * 2 + 1] = ss;
                }
            } else {
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
    }
    if (!fast) return;
    __syncthreads();
    // one thread per (consumer, plane, quad): the first quad of each group inside this tile adds up its group
    for (int e = tid; e < a.nstat * 3 * QPR; e += NTH) {
        const int t = e / (3 * QPR), r2 = e - t * 3 * QPR;
        const int sgi = r2 / QPR, cq = r2 - sgi * QPR;
        const int n = n0 + cq * 4;
        if (n >= a.N) continue;
        const int gs = a.stat[t].gs, coff = a.stat[t].coff;
        const int g = (coff + n) / gs;
        if (cq > 0 && (coff + n - 4) / gs == g) continue;           // not the first quad of its group in this tile
        double s = 0.0, ss = 0.0;
        for (int c2 = cq; c2 < QPR && n0 + c2 * 4 < a.N && (coff + n0 + c2 * 4) / gs == g; ++c2) {
            s += qs[(sgi * QPR + c2) * 2];
            ss += qs[(sgi * QPR + c2) * 2 + 1];
        }
        if (ss != 0.0) {
            double* dst = a.stat[t].sums + (size_t)(blockIdx.x & (STAT_COPIES - 1)) * a.stat_cstride + (((size_t)b * 3 + sgi) * 32 + g) * 2;
            atomicAdd(dst, s);
            atomicAdd(dst + 1, ss);
        }
    }
    MTV_STAMP(6);
}

// --------------------------------------------------------------------------------------------
// tile / split selection: a small analytic cost model (times in microseconds)
// --------------------------------------------------------------------------------------------
static size_t lds_bytes(int MT, int NT, int NW, int ntaps, int Cmain, bool has_gn) {
    const int ROWS = 16 * MT, COLS = 16 * NT;
    const size_t idx = (size_t)(((ntaps + 1) * ROWS + 3) & ~3) * 4 + 24 * 32;   // + segment descriptors
    const size_t coef = has_gn ? (size_t)24 * Cmain : 0;
    const size_t part = (size_t)MT * NT * 4 * 64 * 4;                 // one wave's partial tile
    const size_t fin = (size_t)ROWS * (COLS + 4) * 4 + 3 * (COLS / 4) * 2 * 8;   // finished tile + per-quad statistics
    const bool one_stage = NW > 1 && NW * part <= 48 * 1024;
    const size_t redu = one_stage ? NW * part + fin : ((size_t)(NW / 2) * part > fin ? (size_t)(NW / 2) * part : fin);
    size_t r = idx + coef;
    if (redu > r) r = redu;
    return r;
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
            if (lds_bytes(MT, NT, NW, ntaps_guess, Cmain, has_gn) > 96 * 1024) continue;
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
    return lds_bytes(t.MT, t.NT, t.NW, a.ntaps, a.Cmain, a.gn.sums != nullptr);
}

template <int MT, int NT, int NW>
static hipError_t launch_conv_t(const ConvArgs& a, hipStream_t s) {
    const int tiles = (a.Lout + 16 * MT - 1) / (16 * MT);
    const int tiles_n = (a.N + 16 * NT - 1) / (16 * NT);
    const long nblk = a.xmap ? 8L * ((tiles_n * a.KS + 7) / 8) * a.B * tiles : (long)a.B * tiles * tiles_n * a.KS;
    dim3 grid((unsigned)nblk);
    const size_t smem = conv_smem_bytes(a, ConvTile{MT, NT, NW, a.KS, 0});
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
    return conv_attr_nw<1, 1>();
}

hipError_t launch_conv(const ConvArgs& a0, ConvTile t, hipStream_t s) {
    ConvArgs a = a0;
    a.KS = t.KS;
    a.xmap = t.XM;
    if (a.KS > 1 && (!a.slab || !a.tickets || (a.N & 3))) return hipErrorInvalidValue;
    if (a.ddim) {
        if (!a.out_cm || a.N != 4) return hipErrorInvalidValue;
        a.xmap = 0;            // no padding blocks: every block of the grid takes part in the step hand-over
    }
    hipError_t e = hipErrorInvalidValue;
    if (t.MT == 4 && t.NT == 4) e = launch_conv_nw<4, 4>(a, t.NW, s);
    else if (t.MT == 2 && t.NT == 4) e = launch_conv_nw<2, 4>(a, t.NW, s);
    else if (t.MT == 1 && t.NT == 4) e = launch_conv_nw<1, 4>(a, t.NW, s);
    else if (t.MT == 2 && t.NT == 2) e = launch_conv_nw<2, 2>(a, t.NW, s);
    else if (t.MT == 1 && t.NT == 2) e = launch_conv_nw<1, 2>(a, t.NW, s);
    else if (t.MT == 1 && t.NT == 1) e = launch_conv_nw<1, 1>(a, t.NW, s);
    return e;
}

}  // namespace mtv
