// Hand-written gfx950 (CDNA4, wave64) kernels of the MToV denoising step.
//
// Layout: every activation is token-major channels-last [B][L][C] fp32; the three planes of the
// tri-plane latent are contiguous row ranges (SegInfo).  All matmul-shaped work runs on the exact
// f32 matrix instruction v_mfma_f32_16x16x4_f32 (bit-for-bit an fmaf chain, 157 TF peak), because
// the parity bar is 1e-3 max-abs after 250 stochastic steps in fp32 (BASELINE.json north_star).
//
// MFMA 16x16x4 f32 operand maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&15][k = l>>4]     B: lane l holds B[k = l>>4][j = l&15]
//   D: lane l, reg r holds D[row = 4*(l>>4) + r][col = l&15]
// K is a reduction index, so A and B may use ANY common permutation of K, and the N (column)
// index of B/D may be permuted freely too.  Both freedoms are used so that every operand fragment
// is a plain 16-byte global load: no LDS staging of operands is needed at the f32 MFMA rate.
#include "mtv_internal.h"
#include <type_traits>
#include <cstdio>
#include <cstdlib>

namespace mtv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

// =====================================================================================
// GroupNorm statistics: fp64 (sum, sumsq) per (batch, plane, group)
// =====================================================================================
// Replaces the reduction half of GroupNorm32 (diffusionmodules.py:156-173).  Sources may be a
// channel concatenation of two tensors (the skip cat of unet.py:1080-1087 is never materialised).
__global__ __launch_bounds__(256) void k_gn_stats(const StatsArgs a) {
    __shared__ double s_acc[64];
    const int tid = threadIdx.x;
    if (tid < 64) s_acc[tid] = 0.0;
    __syncthreads();
    const int b = blockIdx.z, sg = blockIdx.y;
    const int t0 = sg == 0 ? 0 : (sg == 1 ? a.seg.b1 : a.seg.b2);
    const int t1 = sg == 0 ? a.seg.b1 : (sg == 1 ? a.seg.b2 : a.seg.L);
    const int nq = a.Ctot >> 2;                       // channel quads
    const int nqc = nq < 256 ? nq : 256;              // quads handled concurrently by the block
    const int tpb = 256 / nqc;                        // tokens handled concurrently
    const int trow = tid / nqc;
    const int chunk = (t1 - t0 + gridDim.x - 1) / gridDim.x;
    const int ta = t0 + blockIdx.x * chunk, tb = min(t1, ta + chunk);
    for (int cq = tid % nqc; cq < nq && trow < tpb; cq += nqc) {
        const int c = cq << 2;
        const int part = (a.nparts > 1 && c >= a.C[0]) ? 1 : 0;
        const int cl = c - (part ? a.C[0] : 0);
        const float* sp = a.src[part];
        const int Cp = a.C[part];
        double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
        for (int t = ta + trow; t < tb; t += tpb) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(sp + ((size_t)b * a.seg.L + t) * Cp + cl);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[e] += (double)v[e];
                ss[e] += (double)v[e] * (double)v[e];
            }
        }
        if ((a.gs & 3) == 0) {   // the whole quad lies in one group
            const int g = c / a.gs;
            atomicAdd(&s_acc[g * 2], (s[0] + s[1]) + (s[2] + s[3]));
            atomicAdd(&s_acc[g * 2 + 1], (ss[0] + ss[1]) + (ss[2] + ss[3]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int g = (c + e) / a.gs;
                atomicAdd(&s_acc[g * 2], s[e]);
                atomicAdd(&s_acc[g * 2 + 1], ss[e]);
            }
        }
    }
    __syncthreads();
    if (tid < 64) atomicAdd(&a.sums[((size_t)b * 3 + sg) * 64 + tid], s_acc[tid]);
}

hipError_t launch_gn_stats(const StatsArgs& a, hipStream_t s) {
    const int maxlen = a.seg.b1 > a.seg.L - a.seg.b2 ? a.seg.b1 : a.seg.L - a.seg.b2;
    int nblk = (int)(((long)maxlen * a.Ctot + 16383) / 16384);   // ~64 KB of activations per block
    nblk = nblk < 1 ? 1 : (nblk > 64 ? 64 : nblk);
    hipLaunchKernelGGL(k_gn_stats, dim3(nblk, 3, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// ResBlock(down=True) input path: avgpool2x2(SiLU(GN(x))) and avgpool2x2(x)
// =====================================================================================
__global__ __launch_bounds__(256) void k_pool_down(const PoolArgs a) {
    touch_kernargs<(int)sizeof(PoolArgs)>();
    // grid (blocks over dst tokens x channel quads, B): one batch element per block row, so the group
    // statistics (summed over the privatised copies) are finalised once per block into LDS
    __shared__ float2 s_mr[3][32];
    const int b = blockIdx.y;
    for (int e = threadIdx.x; e < 96; e += 256) {
        const int sg = e >> 5, g = e & 31;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < STAT_COPIES; ++k) {
            const double* S = a.sums + (size_t)k * a.cstride + ((size_t)b * 3 + sg) * 64;
            s1 += S[g * 2];
            s2 += S[g * 2 + 1];
        }
        const int len = sg == 0 ? a.seg_src.b1 : (sg == 1 ? a.seg_src.b2 - a.seg_src.b1 : a.seg_src.L - a.seg_src.b2);
        const double n = (double)len * a.gs;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        s_mr[sg][g] = make_float2((float)mean, 1.0f / sqrtf((float)var + 1e-5f));
    }
    __syncthreads();
    const int nq = a.C >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.seg_dst.L * nq;
    if (idx >= total) return;
    const int cq = (int)(idx % nq);
    const int tok = (int)(idx / nq);
    int sg, y, x, base_s;
    const int wd = a.r_dst;
    if (tok < a.seg_dst.b1) { sg = 0; y = tok / wd; x = tok - y * wd; base_s = 0; }
    else if (tok < a.seg_dst.b2) { sg = 1; const int tt = tok - a.seg_dst.b1; y = tt / wd; x = tt - y * wd; base_s = a.seg_src.b1; }
    else { sg = 2; const int tt = tok - a.seg_dst.b2; y = tt / wd; x = tt - y * wd; base_s = a.seg_src.b2; }
    const int ws = 2 * wd;
    const int c = cq << 2;
    float sc[4], bi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float2 mr = s_mr[sg][(c + e) / a.gs];
        sc[e] = mr.y * a.gamma[c + e];
        bi[e] = a.beta[c + e] - sc[e] * mr.x;
    }
    f32x4 sa = {0, 0, 0, 0}, sx = {0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int st = base_s + (2 * y + dy) * ws + 2 * x + dx;
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)b * a.seg_src.L + st) * a.C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sx[e] += v[e];
                sa[e] += silu_f(fmaf(v[e], sc[e], bi[e]));
            }
        }
    const size_t o = ((size_t)b * a.seg_dst.L + tok) * a.C + c;
    mtv_store_out4(a.out_act, o, sa * 0.25f);
    mtv_store_out4(a.out_x, o, sx * 0.25f);
}

hipError_t launch_pool_down(const PoolArgs& a, hipStream_t s) {
    const long total = (long)a.seg_dst.L * (a.C >> 2);
    hipLaunchKernelGGL(k_pool_down, dim3((unsigned)((total + 255) / 256), a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// QKVAttentionLegacy core (unet.py:312-326), flash-style, exact-f32 MFMA, online softmax
// =====================================================================================
// A workgroup = 4 waves = 64 consecutive queries of one (batch, segment, head); each wave owns 16
// queries.  The keys of the segment are walked in blocks of KB (64, fewer for wide heads); the block's K rows and V^T (keys
// contiguous) are staged in LDS once per workgroup, double buffered: the global loads of block j+1
// are issued before block j is computed and written to the other buffer after it.
// Per wave: S^T = K Q^T (keys x queries) so a query is a lane COLUMN -- the softmax reductions over
// keys are 16 register values plus two cross-lane steps -- and the P^T registers are directly the
// B operand of O^T += V^T P^T (K slot g of step s is key 4g+s, exactly the D layout of S^T).
// q and k are each pre-multiplied by d^-1/4 like the reference.
// max over the lane pairs (l, l^16) / (l, l^32), result in every lane: v_permlane{16,32}_swap_b32 with both
// operands holding x leaves {x_even_rows, x_odd_rows} / {x_low_half, x_high_half} in the two registers
__device__ __forceinline__ float lane_swap_max16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float lane_swap_max32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

#ifdef MTV_ATT_STAMP   // phase timestamps of thread 0 of four sampled workgroups (first two, middle, last) -> a.dbg (tools/stamps.py)
#define ATT_STAMP(k) do { if (threadIdx.x == 0 && a.dbg) { const int sb_ = blockIdx.x < 2 ? (int)blockIdx.x : (blockIdx.x == gridDim.x / 2 ? 2 : (blockIdx.x == gridDim.x - 1 ? 3 : -1)); \
                          if (sb_ >= 0) a.dbg[sb_ * 16 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define ATT_PUT(k, v) do { if (threadIdx.x == 0 && a.dbg) { const int sb_ = blockIdx.x < 2 ? (int)blockIdx.x : (blockIdx.x == gridDim.x / 2 ? 2 : (blockIdx.x == gridDim.x - 1 ? 3 : -1)); \
                          if (sb_ >= 0) a.dbg[sb_ * 16 + (k)] = (v); } } while (0)
#define ATT_ACC(x) do { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); (x) += t1_ - att_t0; att_t0 = t1_; } while (0)
#else
#define ATT_STAMP(k) do { } while (0)
#define ATT_PUT(k, v) do { } while (0)
#define ATT_ACC(x) do { } while (0)
#endif

typedef __bf16 att_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 att_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned att_u32x4 __attribute__((ext_vector_type(4)));
typedef float att_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned att_pk(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(att_f32x2{a, b}, att_bf16x2)); }
// x = x0 + x1 + x2, three bf16 terms (round to nearest even; the residuals are exact): pairs of floats -> packed pairs per plane
__device__ __forceinline__ void att_split2(float a, float b, unsigned& t0, unsigned& t1, unsigned& t2) {
    t0 = att_pk(a, b);
    const float ra = a - __uint_as_float(t0 << 16), rb = b - __uint_as_float(t0 & 0xffff0000u);
    t1 = att_pk(ra, rb);
    t2 = att_pk(ra - __uint_as_float(t1 << 16), rb - __uint_as_float(t1 & 0xffff0000u));
}
__device__ __forceinline__ void att_split4(const f32x4& v, unsigned (&p0)[2], unsigned (&p1)[2], unsigned (&p2)[2]) {
    att_split2(v[0], v[1], p0[0], p1[0], p2[0]);
    att_split2(v[2], v[3], p0[1], p1[1], p2[1]);
}
__device__ __forceinline__ void att_split8(const float (&y)[8], att_u32x4& p0, att_u32x4& p1, att_u32x4& p2) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        unsigned t0, t1, t2;
        att_split2(y[2 * h], y[2 * h + 1], t0, t1, t2);
        p0[h] = t0;
        p1[h] = t1;
        p2[h] = t2;
    }
}

// LDS of one k_attention workgroup (floats): K stages | V^T stages | key-part merge records.  KBX = 2 doubles the key block
// (half the barriers / tile hand-overs per segment; above the 64 KB static limit, hence dynamic LDS throughout).
// QB = 1: the keys are staged as three bf16 planes (k = k0 + k1 + k2) of KPS bytes per key, for QK^T on the bf16 matrix pipe.
template <int D, int QW, int KSP, int KBX, int QB = 0>
struct AttShape {
    static constexpr int KSTR = D + (D >= 16 ? 4 : 0);
    static constexpr int KB = (D >= 128 ? 16 : (D >= 64 ? 32 : (D >= 32 ? 64 : 128))) * KBX;
    static constexpr int VSTR = KB + 4, VROWS = D < 16 ? 16 : D;
    static constexpr int NOB = D >= 16 ? D / 16 : 1, XW = NOB * 4 + 2;
    // bytes per key and plane: D bf16, padded to an odd multiple of 32 bytes (16 keys x {g, g + 1} of a ds_read_b128 lane group
    // then hit 16 distinct bank quads: 32 / 96 / 160 bytes at d = 16 / 32 / 64)
    static constexpr int KPS = 16 * (D / 8 + (((D / 8) % 4) == 2 ? 0 : 2));
    static constexpr int KS_FLOATS = QB ? 2 * 3 * KB * KPS / 4 : 2 * KB * KSTR, VT_FLOATS = 2 * VROWS * VSTR;
    static constexpr int XO_FLOATS = KSP > 1 ? (KSP - 1) * QW * XW * 64 : 4;
    static constexpr size_t BYTES = (size_t)(KS_FLOATS + VT_FLOATS + XO_FLOATS) * 4;
};

template <int D, int QW, int KSP, int KBX = 1, int QB = 0>
__global__ __launch_bounds__(64 * QW * KSP) void k_attention(const AttnArgs a) {
    typedef AttShape<D, QW, KSP, KBX, QB> ASH;
    static_assert(!QB || D == 16 || D == 32 || D == 64, "bf16 QK^T: d = 16, 32, 64");
    touch_kernargs<(int)sizeof(AttnArgs)>();
    ATT_STAMP(0);
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int VW = D >= 16 ? 4 : D / 4;        // floats per q/k fragment
    constexpr int NV = D >= 16 ? D / 16 : 1;       // fragments per row
    constexpr int NOB = D >= 16 ? D / 16 : 1;      // 16-row output blocks of O^T
    // K row stride in LDS (floats): rows + 4 so that the 16 lanes of a 16-byte fragment read (rows j = 0..15, same quad g)
    // hit 16 distinct bank quads -- (KSTR/4 * j + g) mod 16 must be a permutation of j: KSTR/4 odd.  (D = 16 ran unpadded
    // through round 2's first half: stride 16 floats = a 4-way conflict on every K fragment of the level-0 attentions.)
    constexpr int KSTR = D + (D >= 16 ? 4 : 0);
    constexpr int KB = ASH::KB;   // keys per block (LDS budget)
    static_assert(KSTR == ASH::KSTR, "LDS layout");
    constexpr int NKT = KB / 16;                   // 16-key tiles per block
    constexpr int WKT = NKT / KSP;                 // ... of which each wave takes WKT
    static_assert(WKT >= 1, "key split wider than the key block");
    constexpr int VSTR = KB + 4;                   // V^T row stride: KB keys + 4 (conflict-free b128 reads)
    constexpr int VROWS = D < 16 ? 16 : D;
    constexpr int QPR = D / 4;                     // float4 quads per K/V row
    constexpr int NTH = 64 * QW * KSP;
    constexpr int NLD = (KB * QPR + NTH - 1) / NTH;   // float4 loads per thread per tile
    extern __shared__ __attribute__((aligned(16))) float att_smem[];
    typedef float ks_row_t[KB * KSTR];
    typedef float vt_row_t[VROWS * VSTR];
    ks_row_t* Ks = reinterpret_cast<ks_row_t*>(att_smem);                                                       // [2][KB * KSTR]
    vt_row_t* Vt = reinterpret_cast<vt_row_t*>(att_smem + ASH::KS_FLOATS);                // [2][VROWS * VSTR]
    float* Xo = att_smem + ASH::KS_FLOATS + ASH::VT_FLOATS;
    constexpr int KPS = ASH::KPS, KPL = KB * KPS;      // QB: bytes per key / per plane of a stage
    char* const Kb = reinterpret_cast<char*>(att_smem);                                   // QB: [2 stages][3 planes][KB][KPS]
    constexpr int XW = NOB * 4 + 2;                // merge record per lane: m, l, O^T registers

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, g = lane >> 4;
    const int qw = wave % QW, kh = wave / QW;      // query tile of the workgroup, key part of each block
    // which (segment, 64-query block)
    // 1-D grid, head fastest: consecutive workgroup ids go round-robin over the 8 XCDs, so with H = 8 every
    // workgroup of a head runs on the same XCD and the head's K/V rows are fetched into ONE L2 (speed only).
    // (all divisions by float reciprocals from the host: exact below 2^22, which launch_attention checks; the decode is
    // branch-free so that every argument is fetched at entry -- a loop over blk_prefix made the compiler fetch and wait
    // for one kernel argument at a time)
    const int rest = FDiv{a.inv_H}((int)blockIdx.x, a.H);
    const int h = (int)blockIdx.x - rest * a.H;
    const int nblk = a.blk_prefix[3];               // (host: unused prefix slots hold the total)
    const int b = FDiv{a.inv_nblk}(rest, nblk);
    const int qblk = rest - b * nblk;
    int sg, q0, start, len;
    {
        const int su = FDiv{a.inv_bps}(qblk, a.bps);                                        // uniform: segment = qblk / blocks per segment
        const int sn = (qblk >= a.blk_prefix[1] ? 1 : 0) + (qblk >= a.blk_prefix[2] ? 1 : 0);
        const bool uni = a.seg_uniform != 0;
        sg = uni ? su : sn;
        const int first = uni ? su * a.bps : (sn == 0 ? 0 : (sn == 1 ? a.blk_prefix[1] : a.blk_prefix[2]));
        len = uni ? a.seg_uniform : (sn == 0 ? a.seg_len[0] : (sn == 1 ? a.seg_len[1] : a.seg_len[2]));
        start = uni ? su * a.seg_uniform : (sn == 0 ? a.seg_start[0] : (sn == 1 ? a.seg_start[1] : a.seg_start[2]));
        q0 = (qblk - first) * (16 * QW) + qw * 16;
    }
    // self-attention: q | k | v of a head are adjacent in ONE buffer (row stride 3C).  Cross-attention (a.kv != nullptr,
    // CrossAttention of unet.py:429-467): queries [B][L][C] (head-major), keys/values in a second buffer [B][Lkv][2C] with
    // k | v of a head adjacent; one segment, every query sees all Lkv keys (minus the masked ones).
    const bool cross = a.kv != nullptr;
    const int RS = cross ? a.C : 3 * a.C;                              // query row stride
    const int RSK = cross ? 2 * a.C : 3 * a.C;                         // key/value row stride
    const float* base = a.qkv + (size_t)b * a.L * RS + (size_t)h * (cross ? D : 3 * D);
    const float* kbase = cross ? a.kv + (size_t)b * a.Lkv * RSK + (size_t)h * 2 * D : base + D;   // -> k of key 0 (v follows at + D)
    const int klen = cross ? a.Lkv : len, kstart = cross ? 0 : start;
    const unsigned char* kmask = cross && a.kmask ? a.kmask + (size_t)b * a.Lkv : nullptr;
    const float scale = a.scale;

    if (D < 16) {   // rows D..15 of V^T are never written: keep them zero (they feed masked MFMA rows)
        for (int e = tid; e < 2 * VROWS * VSTR; e += NTH) (&Vt[0][0])[e] = 0.f;
        __syncthreads();
    }

    // q fragments: UNCONDITIONAL vector loads from a clamped row (a conditional scalar load per element compiled to 16
    // branches with a full wait each), requested together with the first K/V tile; masked when they are consumed
    typedef float qvec_t __attribute__((ext_vector_type(VW)));
    qvec_t qraw[NV];
    // QB: the B operand of v_mfma_f32_16x16x32_bf16 holds 8 consecutive k-dim values per lane.  d = 16: the 32-wide k dim carries
    // TWO products of 16 (lanes g < 2 | g >= 2), d index 8 (g & 1) + e; d >= 32: one product per 32-wide step, d index 32 ks + 8 g + e.
    constexpr int NKS = D >= 32 ? D / 32 : 1;       // 32-wide k steps
    f32x4 qb_raw[QB ? NKS : 1][2];
    const bool qok = q0 + j < len;
    {
        const float* qrow = base + (size_t)(start + (qok ? q0 + j : 0)) * RS;
        if constexpr (QB) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const float* qp = qrow + (D == 16 ? 8 * (g & 1) : 32 * ks + 8 * g);
                qb_raw[ks][0] = *reinterpret_cast<const f32x4*>(qp);
                qb_raw[ks][1] = *reinterpret_cast<const f32x4*>(qp + 4);
                for (int sl = 1; sl < a.qkv_ks; ++sl) {           // deep levels (deep.hip): qkv = sum of K-slice slabs, slab order
                    qb_raw[ks][0] += *reinterpret_cast<const f32x4*>(qp + (size_t)sl * a.qkv_slab);
                    qb_raw[ks][1] += *reinterpret_cast<const f32x4*>(qp + (size_t)sl * a.qkv_slab + 4);
                }
            }
        } else {
            const float* qp = qrow + VW * g;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                qraw[u] = *reinterpret_cast<const qvec_t*>(qp + 16 * u);
                for (int sl = 1; sl < a.qkv_ks; ++sl) qraw[u] += *reinterpret_cast<const qvec_t*>(qp + (size_t)sl * a.qkv_slab + 16 * u);
            }
        }
    }
    f32x4 oacc[NOB];
#pragma unroll
    for (int o = 0; o < NOB; ++o) oacc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;

    // staging: thread -> (key, quad) of the KB x D tile
    f32x4 kreg[NLD], vreg[NLD];
    auto gload = [&](int kb) {
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int e = tid + NTH * r;
            const int key = e / QPR, qd = e - key * QPR;
            // (e >= KB * QPR only in the last, partial r: then key >= KB, the row is clamped and qd stays < QPR)
            const bool ok = e < KB * QPR && kb + key < klen;
            const float* p = kbase + (size_t)(kstart + (ok ? kb + key : 0)) * RSK + qd * 4;
            // unconditional loads, RAW registers: rows past the segment are zeroed when the tile is written to LDS -- a
            // select here puts the wait for the loads right behind their issue, i.e. in front of the block's math
            kreg[r] = *reinterpret_cast<const f32x4*>(p);
            vreg[r] = *reinterpret_cast<const f32x4*>(p + D);
            for (int sl = 1; sl < a.qkv_ks; ++sl) {               // deep levels: the K-slice slabs of qkv, slab order
                kreg[r] += *reinterpret_cast<const f32x4*>(p + (size_t)sl * a.qkv_slab);
                vreg[r] += *reinterpret_cast<const f32x4*>(p + (size_t)sl * a.qkv_slab + D);
            }
        }
    };
    // (all_tag: every key of the tile exists -- all tiles but possibly the last: no per-row selects, the softmax's VALU is the
    // kernel's bottleneck (profiles/r04_attention_phase_stamps.txt))
    auto lstore = [&](int buf, int kb, auto all_tag) {
        constexpr bool ALL = decltype(all_tag)::value;
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int e = tid + NTH * r;
            const int key = e / QPR, qd = e - key * QPR;
            if (e < KB * QPR) {
                const bool in = ALL || kb + key < klen;
                const f32x4 kq = in ? kreg[r] * scale : f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (QB) {       // three bf16 terms of the 4 values, 8 bytes per plane
                    unsigned p0[2], p1[2], p2[2];
                    att_split4(kq, p0, p1, p2);
                    char* kd = Kb + (size_t)buf * 3 * KPL + key * KPS + qd * 8;
                    *reinterpret_cast<uint2*>(kd) = make_uint2(p0[0], p0[1]);
                    *reinterpret_cast<uint2*>(kd + KPL) = make_uint2(p1[0], p1[1]);
                    *reinterpret_cast<uint2*>(kd + 2 * KPL) = make_uint2(p2[0], p2[1]);
                } else {
                    *reinterpret_cast<f32x4*>(&Ks[buf][key * KSTR + qd * 4]) = kq;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) Vt[buf][(qd * 4 + c) * VSTR + key] = in ? vreg[r][c] : 0.f;
            }
        }
    };

    ATT_STAMP(1);
    gload(0);
    float qreg[NV][VW];
    att_bf16x8 qf[QB ? NKS : 1][3];          // QB, d >= 32: the terms a0, a1, a2 of the step; d = 16: [a0|a0], [a1|a1], [a0|a2]
    if constexpr (QB) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            float y[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[e] = qok ? qb_raw[ks][0][e] * scale * LOG2E : 0.f;
                y[4 + e] = qok ? qb_raw[ks][1][e] * scale * LOG2E : 0.f;
            }
            att_u32x4 a0, a1, a2;
            att_split8(y, a0, a1, a2);
            qf[ks][0] = __builtin_bit_cast(att_bf16x8, a0);
            qf[ks][1] = __builtin_bit_cast(att_bf16x8, a1);
            qf[ks][2] = __builtin_bit_cast(att_bf16x8, (D == 16 && g < 2) ? a0 : a2);
        }
    } else {
#pragma unroll
        for (int u = 0; u < NV; ++u)
#pragma unroll
            for (int e = 0; e < VW; ++e) qreg[u][e] = qok ? qraw[u][e] * scale * LOG2E : 0.f;   // scores in log2 units
    }
    lstore(0, 0, std::false_type{});
    __syncthreads();
    ATT_STAMP(2);
    int buf = 0;
    [[maybe_unused]] unsigned long long att_tb = 0, att_tw = 0, att_ts = 0, att_t0 = 0;
    ATT_ACC(att_tb);
    att_tb = 0;
    // One key block.  FULL (every key of the block exists -- all blocks but possibly the last) is a separate
    // instantiation: no per-element masking, no all-masked guard.  Scores are in the log2 domain (log2 e is
    // folded into q), so p = exp2(s - m) is one subtract + one v_exp_f32 per element; the cross-lane maxima
    // use the gfx950 lane-swap instructions (VALU, no LDS round trip).
    auto block = [&](int kb, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;       // 0: every key exists; 1: partial block; 2: caller's key mask
        constexpr bool FULL = MODE == 0;
        const float* ks = Ks[buf];
        const float* vt = Vt[buf];
        f32x4 st[WKT];
        unsigned mv[WKT][4];
        if constexpr (MODE == 2) {      // all mask bytes of the wave's tiles requested together, ahead of the QK^T MFMAs
#pragma unroll
            for (int w = 0; w < WKT; ++w)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kb + (kh * WKT + w) * 16 + 4 * g + r;
                    mv[w][r] = kmask[key >= klen ? 0 : key];
                }
        }
        // S^T tiles.  A dependent v_mfma_f32_16x16x4_f32 issues 40 cycles after its producer, an independent one after 32
        // (MI355X_MICROARCH.md): the MFMAs are therefore emitted round-robin over >= 2 independent accumulators -- the
        // wave's WKT tiles, or, with a single tile, the even / odd fragments of the head dimension (summed afterwards).
        constexpr int US = (!QB && WKT == 1 && NV >= 2) ? 2 : 1;
        f32x4 sacc[WKT][US];
        if constexpr (QB) {
            // S^T = K Q^T on v_mfma_f32_16x16x32_bf16 at f32 accuracy: k = c0 + c1 + c2, q = a0 + a1 + a2, products a0c2, a2c0, a1c1,
            // a1c0, a0c1, a0c0 (small first).  d = 16: the k dim holds two 16-wide products per instruction -- [c2|c0].[a0|a2],
            // [c0|c1].[a1|a1], [c0|c1].[a0|a0]: 3 MFMAs of 16 cycles instead of 4 f32 ones of 32.  d >= 32: six per 32-wide step.
            const char* kst = Kb + (size_t)buf * 3 * KPL;
            if constexpr (D == 16) {
                att_bf16x8 f1[WKT], f3[WKT];
#pragma unroll
                for (int w = 0; w < WKT; ++w) {
                    const char* kp = kst + ((kh * WKT + w) * 16 + j) * KPS + (g & 1) * 16;
                    f1[w] = *reinterpret_cast<const att_bf16x8*>(kp + (g < 2 ? 0 : KPL));              // [c0 | c1]
                    f3[w] = *reinterpret_cast<const att_bf16x8*>(kp + (g < 2 ? 2 * KPL : 0));          // [c2 | c0]
                    sacc[w][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int w = 0; w < WKT; ++w) sacc[w][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f3[w], qf[0][2], sacc[w][0], 0, 0, 0);
#pragma unroll
                for (int w = 0; w < WKT; ++w) sacc[w][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1[w], qf[0][1], sacc[w][0], 0, 0, 0);
#pragma unroll
                for (int w = 0; w < WKT; ++w) sacc[w][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1[w], qf[0][0], sacc[w][0], 0, 0, 0);
            } else {
                att_bf16x8 kf[WKT][NKS][3];
#pragma unroll
                for (int w = 0; w < WKT; ++w) {
                    const char* kp = kst + ((kh * WKT + w) * 16 + j) * KPS + g * 16;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) kf[w][ks][pl] = *reinterpret_cast<const att_bf16x8*>(kp + pl * KPL + ks * 64);
                    sacc[w][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                constexpr int PK[6] = {2, 1, 0, 1, 0, 0}, PQ[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                        for (int w = 0; w < WKT; ++w)
                            sacc[w][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[w][ks][PK[t]], qf[ks][PQ[t]], sacc[w][0], 0, 0, 0);
            }
        } else {
        float kv[WKT][NV][VW];
#pragma unroll
        for (int w = 0; w < WKT; ++w) {
            const float* kp = ks + ((kh * WKT + w) * 16 + j) * KSTR + VW * g;
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                if constexpr (VW == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(kp + 16 * u);
                    kv[w][u][0] = t[0]; kv[w][u][1] = t[1]; kv[w][u][2] = t[2]; kv[w][u][3] = t[3];
                } else {
#pragma unroll
                    for (int e = 0; e < VW; ++e) kv[w][u][e] = kp[e];
                }
            }
#pragma unroll
            for (int c = 0; c < US; ++c) sacc[w][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u0 = 0; u0 < NV; u0 += US)
#pragma unroll
            for (int e = 0; e < VW; ++e)
#pragma unroll
                for (int c = 0; c < US; ++c)
#pragma unroll
                    for (int w = 0; w < WKT; ++w)
                        if (u0 + c < NV) sacc[w][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv[w][u0 + c][e], qreg[u0 + c][e], sacc[w][c], 0, 0, 0);
        }
#pragma unroll
        for (int w = 0; w < WKT; ++w) {
            const int kt = kh * WKT + w;
            f32x4 s4 = sacc[w][0];
            if constexpr (US == 2) s4 += sacc[w][1];
            if constexpr (!FULL) {
                // keys past the end of the segment, or masked out by the caller (unet.py:452-456), score -inf.  Branch-free:
                // the mask bytes are fetched with a clamped index and folded into a select (the short-circuit form
                // `key >= klen || (kmask && !kmask[key])` lost the mask term in hipcc 7.2's control-flow lowering)
                const int key0 = kb + kt * 16 + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool oob = key0 + r >= klen;
                    bool dead = oob;
                    if constexpr (MODE == 2) dead = oob | (mv[w][r] == 0u);
                    s4[r] = dead ? -INFINITY : s4[r];
                }
            }
            st[w] = s4;
        }
        float mx = st[0][0];
#pragma unroll
        for (int w = 0; w < WKT; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[w][r]);
        mx = lane_swap_max16(mx);
        mx = lane_swap_max32(mx);
        const float mn = fmaxf(m, mx);
        const bool live = FULL || mn != -INFINITY;      // a key-split wave may see only masked keys so far
        const float alpha = live ? __builtin_amdgcn_exp2f(m - mn) : 1.0f;
        m = mn;
        // (pairs: s - m and the row sum as two-wide packed f32 operations -- v_pk_add_f32 -- around the scalar v_exp_f32)
        f32x2 ps2 = {0.f, 0.f};
        const f32x2 mn2 = {mn, mn};
#pragma unroll
        for (int w = 0; w < WKT; ++w)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2 dlt = f32x2{st[w][r], st[w][r + 1]} - mn2;
                const f32x2 p = {live ? __builtin_amdgcn_exp2f(dlt[0]) : 0.f, live ? __builtin_amdgcn_exp2f(dlt[1]) : 0.f};
                st[w][r] = p[0];
                st[w][r + 1] = p[1];
                ps2 += p;
            }
        const float ps = ps2[0] + ps2[1];
        lsum = lsum * alpha + ps;
        // O^T += V^T P^T : A = V^T rows (d index) x keys, read as 4 consecutive keys per lane.  Independent accumulators
        // round-robin again: the NOB output blocks, or (one block: d <= 16) the even / odd key tiles of the wave.
        constexpr int OS = (NOB == 1 && WKT >= 2) ? 2 : 1;
        f32x4 pacc[NOB][OS];
#pragma unroll
        for (int o = 0; o < NOB; ++o) {
            pacc[o][0] = oacc[o] * alpha;
            if constexpr (OS == 2) pacc[o][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int w0 = 0; w0 < WKT; w0 += OS) {
            f32x4 v[OS][NOB];
#pragma unroll
            for (int c = 0; c < OS; ++c)
#pragma unroll
                for (int o = 0; o < NOB; ++o)
                    v[c][o] = *reinterpret_cast<const f32x4*>(vt + (16 * o + j) * VSTR + (kh * WKT + w0 + c) * 16 + 4 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < OS; ++c)
#pragma unroll
                    for (int o = 0; o < NOB; ++o)
                        pacc[o][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[c][o][s], st[w0 + c][s], pacc[o][c], 0, 0, 0);
        }
#pragma unroll
        for (int o = 0; o < NOB; ++o) {
            oacc[o] = pacc[o][0];
            if constexpr (OS == 2) oacc[o] += pacc[o][1];
        }
    };
    for (int kb = 0; kb < klen; kb += KB) {
        const bool more = kb + KB < klen;
        if (more) gload(kb + KB);                       // in flight under this block's math
        if (kmask) block(kb, std::integral_constant<int, 2>{});
        else if (kb + KB <= klen) block(kb, std::integral_constant<int, 0>{});
        else block(kb, std::integral_constant<int, 1>{});
        ATT_ACC(att_tb);
        if (more) {
            if (kb + 2 * KB <= klen) lstore(buf ^ 1, kb + KB, std::true_type{});
            else lstore(buf ^ 1, kb + KB, std::false_type{});
        }
        ATT_ACC(att_tw);
        __syncthreads();
        ATT_ACC(att_ts);
        buf ^= 1;
    }
    ATT_STAMP(3);
    ATT_PUT(8, att_tb);
    ATT_PUT(9, att_tw);
    ATT_PUT(10, att_ts);
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if constexpr (KSP > 1) {
        // merge the key parts of each query tile: part kh > 0 parks (m, l, O^T) in LDS
        float* xo = Xo;                                 // [KSP-1][QW tiles][XW][64 lanes]
        if (kh > 0) {
            float* p = xo + ((size_t)((kh - 1) * QW + qw) * XW) * 64 + lane;
            p[0] = m;
            p[64] = lsum;
#pragma unroll
            for (int o = 0; o < NOB; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[(2 + o * 4 + r) * 64] = oacc[o][r];
        }
        __syncthreads();
        if (kh > 0) return;
#pragma unroll
        for (int k2 = 1; k2 < KSP; ++k2) {
            const float* p = xo + ((size_t)((k2 - 1) * QW + qw) * XW) * 64 + lane;
            const float m2 = p[0], l2 = p[64];
            const float mt = fmaxf(m, m2);
            const float f1 = __builtin_amdgcn_exp2f(m - mt), f2 = m2 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m2 - mt);
            lsum = lsum * f1 + l2 * f2;
#pragma unroll
            for (int o = 0; o < NOB; ++o)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[o][r] = oacc[o][r] * f1 + p[(2 + o * 4 + r) * 64] * f2;
            m = mt;
        }
    }
    ATT_STAMP(4);
    const float inv = 1.0f / lsum;
    if (q0 + j < len) {
        const size_t opi = ((size_t)b * a.L + start + q0 + j) * a.C + (size_t)h * D;
        float* op = a.out + opi;
        if constexpr (D >= 16) {
#pragma unroll
            for (int o = 0; o < NOB; ++o) mtv_store_out4(a.out, opi + 16 * o + 4 * g, oacc[o] * inv);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < D) op[4 * g + r] = oacc[0][r] * inv;
        }
    }
    ATT_STAMP(5);
}

template <int D, int QW, int KSP, int KBX, int QB = 0>
static hipError_t att_launch(const AttnArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((k_attention<D, QW, KSP, KBX, QB>), grid, dim3(64 * QW * KSP), (AttShape<D, QW, KSP, KBX, QB>::BYTES), s, a);
    return hipGetLastError();
}
template <int D, int QW, int KSP, int KBX, int QB = 0>
static hipError_t att_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention<D, QW, KSP, KBX, QB>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
}
// Dynamic LDS above 64 KB must be opted into once per kernel (the KBX = 2 shapes); done at context creation, never under capture.
hipError_t attn_init_attrs() {
    hipError_t e;
#define MTV_QB_ATTR(D, QW, KSP, KBX) if ((e = att_attr<D, QW, KSP, KBX, 1>()) != hipSuccess) return e
    MTV_QB_ATTR(16, 4, 2, 1); MTV_QB_ATTR(16, 4, 2, 2); MTV_QB_ATTR(16, 2, 4, 1);
    MTV_QB_ATTR(32, 4, 2, 1); MTV_QB_ATTR(32, 4, 2, 2); MTV_QB_ATTR(32, 2, 4, 1);
    MTV_QB_ATTR(64, 4, 2, 1); MTV_QB_ATTR(64, 4, 2, 2);
#undef MTV_QB_ATTR
    if ((e = att_attr<16, 4, 2, 2>()) != hipSuccess) return e;
    if ((e = att_attr<16, 1, 4, 2>()) != hipSuccess) return e;
    if ((e = att_attr<32, 4, 2, 2>()) != hipSuccess) return e;
    if ((e = att_attr<32, 1, 4, 2>()) != hipSuccess) return e;
    if ((e = att_attr<64, 4, 2, 2>()) != hipSuccess) return e;
    if ((e = att_attr<64, 1, 4, 2>()) != hipSuccess) return e;
    return hipSuccess;
}

int g_attn_qb_default = 1;    // QK^T on the bf16 pipe (d = 16 / 32; d = 64 only when asked for) unless MTV_ATT_QB / mtv_debug_attention_qb say otherwise
int g_attn_qb_force = -1;     // mtv_debug_attention_qb: -1 = environment / default, 0 off, 1 on

hipError_t launch_attention(const AttnArgs& a0, hipStream_t s) {
    AttnArgs a = a0;
    const int d = a.C / a.H;
    // (round 3's all-bf16 split core k_attention_b3 -- parity-green, 0-40 % slower at every shape -- left the product in round 6:
    // tools/experiments/r03_attn_b3.hip, profiles/r03_attention_b3.txt; its QK^T half lives on here as QB = 1)
    // Workgroup shape: QW query tiles (16 queries each) x KSP key parts of every key block.
    //   long segments : 4 x 2 (8 waves, 2 per SIMD.  Not for overlap -- f32 MFMA time and softmax VALU time ADD on a SIMD,
    //                   tools/ubench/mfma_valu -- but to share one K/V staging among 64 queries)
    //   short segments: 1 x 4 -- there are too few query tiles to fill 256 CUs, so the keys are split instead
    long blocks64 = 0;
    const int nuni = a.seg_uniform ? a.L / a.seg_uniform : 0;
    if (a.seg_uniform) blocks64 = (long)nuni * ((a.seg_uniform + 63) / 64);
    else
        for (int i = 0; i < a.nseg; ++i) blocks64 += (a.seg_len[i] + 63) / 64;
    // (>= 128 workgroups of 8 waves already beat 4x as many 16-query workgroups, each of which stages every key of its
    // segment: L = 1536, d = 64 at R = 64 ran 768 two-wave workgroups for 125 us)
    const bool wide = (blocks64 * a.H * a.B >= 128 && (!a.seg_uniform || a.seg_uniform >= 64)) || d >= 128;
    // 32-query workgroups (2 query tiles x 4 key parts) where 64-query ones give at most one workgroup per CU AND the
    // segments differ in length (the plane attention of a [xy | yt | xt] clip: the xy workgroups walk twice the keys):
    // 512 half-size workgroups, two per CU, let the dispatcher even that out (measured 19.2 -> 16.3 us at L = 2048, d = 16;
    // with ONE segment the same change costs 2 us: twice the K/V staging for nothing).  MTV_ATT_QW=2|4 forces either.
    static int wide_qw = -1;
    if (wide_qw < 0) {
        wide_qw = 0;
        if (const char* e = getenv("MTV_ATT_QW")) wide_qw = atoi(e) == 2 ? 2 : (atoi(e) == 4 ? 4 : 0);
    }
    bool uneven = false;
    for (int i = 1; i < a.nseg && !a.seg_uniform; ++i) uneven |= a.seg_len[i] != a.seg_len[0];
    const bool half = wide && (d == 16 || d == 32) && (wide_qw == 2 || (wide_qw == 0 && uneven && blocks64 * a.H * a.B <= 256));
    const int qw = wide ? (half ? 2 : 4) : 1;
    a.blk_prefix[0] = 0;
    if (a.seg_uniform) {
        a.nseg = 1;
        a.bps = (a.seg_uniform + 16 * qw - 1) / (16 * qw);
        a.blk_prefix[1] = nuni * a.bps;
    } else {
        for (int i = 0; i < a.nseg; ++i) a.blk_prefix[i + 1] = a.blk_prefix[i] + (a.seg_len[i] + 16 * qw - 1) / (16 * qw);
    }
    const int nblk = a.blk_prefix[a.nseg];
    for (int i = a.nseg + 1; i < 4; ++i) a.blk_prefix[i] = nblk;       // the kernel's branch-free decode reads [1], [2] and [3]
    if (!a.seg_uniform) a.bps = 1;
    if ((long)nblk * a.H * a.B >= (1l << 22)) return hipErrorInvalidValue;   // FDiv's exact range
    a.inv_H = 1.0f / (float)a.H;
    a.inv_nblk = 1.0f / (float)nblk;
    a.inv_bps = 1.0f / (float)a.bps;
    dim3 grid((unsigned)(nblk * a.H * a.B));
    // Double key blocks (KBX = 2: 256 / 128 / 64 keys at d = 16 / 32 / 64, 80-88 KB of LDS) where the launch has at most one
    // workgroup per CU anyway and a segment spans more than one base block: half the per-block barriers and tile
    // hand-overs, a prefetch distance longer than the load latency, and -- d = 64, short segments -- four key parts
    // (waves) per workgroup instead of two.  With more workgroups than CUs the smaller footprint (3-4 resident
    // workgroups per CU) hides latency better and stays.  MTV_ATT_KBX=1 switches it off.
    static int kbx_env = -1;
    if (kbx_env < 0) {
        kbx_env = 2;
        if (const char* e = getenv("MTV_ATT_KBX")) kbx_env = atoi(e) == 1 ? 1 : (atoi(e) == 3 ? 3 : 2);      // (3: double blocks at every grid size -- experiment)
    }
    int maxk = a.kv ? a.Lkv : a.seg_uniform;
    if (!a.kv && !a.seg_uniform)
        for (int i = 0; i < a.nseg; ++i) maxk = a.seg_len[i] > maxk ? a.seg_len[i] : maxk;
    const int base_kb = d >= 64 ? 32 : (d >= 32 ? 64 : 128);
    const bool big = kbx_env >= 2 && !half && (d == 16 || d == 32 || d == 64) && ((long)nblk * a.H * a.B <= 256 || kbx_env == 3) && maxk > base_kb;
    // QK^T on the bf16 matrix pipe (three-term split of q and k, f32 accuracy; k_attention<..., QB = 1>) for the 8-wave shapes of
    // d = 16 / 32 / 64: MTV_ATT_QB=0 switches it off, =1 on everywhere.
    // Default: on for d = 16 / 32 (level-0 / -1 attentions: -12 ... -17 % per launch at 2048+ keys, profiles/r03_attention_qb.txt),
    // off for d = 64 (the autoencoder: bound by its V^T staging, no gain) unless asked for explicitly.
    static int qb_env = -1, qb_explicit = 0;
    if (qb_env < 0) {
        qb_env = g_attn_qb_default;
        if (const char* e = getenv("MTV_ATT_QB")) { qb_env = atoi(e) != 0 ? 1 : 0; qb_explicit = 1; }
    }
    const bool qb_asked = g_attn_qb_force == 1 || (g_attn_qb_force < 0 && qb_explicit && qb_env == 1);
    const bool qb = d == 64 ? qb_asked : (g_attn_qb_force >= 0 ? g_attn_qb_force : qb_env) != 0;
#define MTV_ATT_GO(D, QW, KSP, KBX) return att_launch<D, QW, KSP, KBX>(a, grid, s)
#define MTV_ATT_GOQ(D, QW, KSP, KBX) do { if (qb) return att_launch<D, QW, KSP, KBX, 1>(a, grid, s); return att_launch<D, QW, KSP, KBX>(a, grid, s); } while (0)
    switch (d) {
        case 4: if (wide) MTV_ATT_GO(4, 4, 2, 1); else MTV_ATT_GO(4, 1, 4, 1);
        case 8: if (wide) MTV_ATT_GO(8, 4, 2, 1); else MTV_ATT_GO(8, 1, 4, 1);
        case 16:
            if (half) MTV_ATT_GOQ(16, 2, 4, 1);
            if (wide) { if (big) MTV_ATT_GOQ(16, 4, 2, 2); else MTV_ATT_GOQ(16, 4, 2, 1); }
            if (big) MTV_ATT_GO(16, 1, 4, 2); else MTV_ATT_GO(16, 1, 4, 1);
        case 32:
            if (half) MTV_ATT_GOQ(32, 2, 4, 1);
            if (wide) { if (big) MTV_ATT_GOQ(32, 4, 2, 2); else MTV_ATT_GOQ(32, 4, 2, 1); }
            if (big) MTV_ATT_GO(32, 1, 4, 2); else MTV_ATT_GO(32, 1, 4, 1);
        case 48: if (wide) MTV_ATT_GO(48, 4, 2, 1); else MTV_ATT_GO(48, 1, 4, 1);   // the autoencoder's quant stacks (autoencoder_vit.py:140-142: dim_head = 384/8)
        case 64:
            if (wide) { if (big) MTV_ATT_GOQ(64, 4, 2, 2); else MTV_ATT_GOQ(64, 4, 2, 1); }
            if (big) MTV_ATT_GO(64, 1, 4, 2); else MTV_ATT_GO(64, 1, 2, 1);
        case 128: MTV_ATT_GO(128, 4, 1, 1);
        default: return hipErrorInvalidValue;
    }
#undef MTV_ATT_GO
#undef MTV_ATT_GOQ
}

// =====================================================================================
// Small dense layers: time_embed MLP and all ResBlock emb_layers in one launch
// =====================================================================================
// out[b][n] = sum_k f(x[b][k]) * W[n][k] + bias[n]   (unet.py:700-705,148-154,193); one wave per n.
__global__ __launch_bounds__(256) void k_linear(const LinearArgs a) {
    touch_kernargs<(int)sizeof(LinearArgs)>();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    const int b = blockIdx.y;
    if (n >= a.N) return;
    const float* w = a.W + (size_t)n * a.K;
    const float* x = a.x + (size_t)b * a.K;
    float acc = 0.f;
    for (int k = lane * 4; k < a.K; k += 256) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
        f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
        if (a.act_in) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = silu_f(xv[e]);
        }
        acc = fmaf(xv[0], wv[0], acc);
        acc = fmaf(xv[1], wv[1], acc);
        acc = fmaf(xv[2], wv[2], acc);
        acc = fmaf(xv[3], wv[3], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) a.out[(size_t)b * a.out_stride + n] = acc + a.bias[n];
}

hipError_t launch_linear(const LinearArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_linear, dim3((a.N + 3) / 4, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// timestep_embedding (diffusionmodules.py:108-128): [cos(t*f) | sin(t*f)], freqs precomputed on host
__global__ void k_time_sinusoid(const int64_t* t, const float* freqs, float* out, int B, int half) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * half) return;
    const int b = idx / half, k = idx - b * half;
    const float arg = (float)t[b] * freqs[k];
    out[(size_t)b * 2 * half + k] = cosf(arg);
    out[(size_t)b * 2 * half + half + k] = sinf(arg);
}

hipError_t launch_time_sinusoid(const int64_t* t, const float* freqs, float* out, int B, int half, hipStream_t s) {
    hipLaunchKernelGGL(k_time_sinusoid, dim3((B * half + 255) / 256), dim3(256), 0, s, t, freqs, out, B, half);
    return hipGetLastError();
}

// unet.py:1022-1025: h = cat[x(4), cond(8), image_cond(4, xy plane only, zeros elsewhere)] -> [B][L][16]
__global__ void k_pack_input(const float* x, const float* cond, const float* ic, int ic_len, float* out, int B, int L, int RR) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * L * 16) return;
    const int c = (int)(idx & 15);
    const int tok = (int)((idx >> 4) % L);
    const int b = (int)((idx >> 4) / L);
    float v;
    if (c < 4) v = x[((size_t)b * 4 + c) * L + tok];
    else if (c < 12) v = cond[((size_t)b * 8 + (c - 4)) * L + tok];
    else v = tok < RR ? ic[((size_t)b * 4 + (c - 12)) * ic_len + tok] : 0.f;
    out[idx] = v;
}

hipError_t launch_pack_input(const float* x, const float* cond, const float* image_cond, int ic_len,
                             float* out, int B, int L, int RR, hipStream_t s) {
    const long n = (long)B * L * 16;
    hipLaunchKernelGGL(k_pack_input, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, cond, image_cond, ic_len, out, B, L, RR);
    return hipGetLastError();
}

// =====================================================================================
// Sampler set-up, once per mtv_ddim_sample call.  The DDIM update itself (ddpm.py:278-282,346-351,386-398) runs
// in the head conv's epilogue (conv.hip, ddim_update_elem); nothing but UNet launches remains inside a step.
// =====================================================================================
// timestep_embedding of EVERY step's t at once: row i = [cos(t_i * f) | sin(t_i * f)]
__global__ void k_step_sinusoid(const DdimStep* steps, int n_steps, const float* freqs, float* out, int half) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_steps * half) return;
    const int i = idx / half, k = idx - i * half;
    const float arg = (float)(int64_t)steps[i].t * freqs[k];
    out[(size_t)i * 2 * half + k] = cosf(arg);
    out[(size_t)i * 2 * half + half + k] = sinf(arg);
}

hipError_t launch_step_sinusoid(const DdimStep* steps, int n_steps, const float* freqs, float* out, int half, hipStream_t s) {
    hipLaunchKernelGGL(k_step_sinusoid, dim3((n_steps * half + 255) / 256), dim3(256), 0, s, steps, n_steps, freqs, out, half);
    return hipGetLastError();
}

// k_linear for many input rows (the time-embedding MLP and the FiLM matrix applied to all steps of a sampler call):
// a wave owns one output feature n and RB = 8 rows, so W is streamed once per 8 rows instead of once per row.
// Per (row, n) the arithmetic is k_linear's, operation for operation (same lane partition of K, same shuffle tree),
// so a step's FiLM row is bit-identical to what the per-step launches of mtv_forward compute.
__global__ __launch_bounds__(256) void k_linear_rows(const LinearArgs a) {
    constexpr int RB = 8;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    const int b0 = blockIdx.y * RB;
    if (n >= a.N) return;
    const float* w = a.W + (size_t)n * a.K;
    float acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = 0.f;
    for (int k = lane * 4; k < a.K; k += 256) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = b0 + r < a.B ? b0 + r : a.B - 1;       // (rows past the end recompute the last one; never stored)
            f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + (size_t)row * a.K + k);
            if (a.act_in) {
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = silu_f(xv[e]);
            }
            acc[r] = fmaf(xv[0], wv[0], acc[r]);
            acc[r] = fmaf(xv[1], wv[1], acc[r]);
            acc[r] = fmaf(xv[2], wv[2], acc[r]);
            acc[r] = fmaf(xv[3], wv[3], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        float v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && b0 + r < a.B) a.out[(size_t)(b0 + r) * a.out_stride + n] = v + a.bias[n];
    }
}

hipError_t launch_linear_rows(const LinearArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_linear_rows, dim3((a.N + 3) / 4, (a.B + 7) / 8), dim3(256), 0, s, a);
    return hipGetLastError();
}

// step counter <- 0, arrival counter <- 0, FiLM row of step 0 -> film_out
__global__ void k_ddim_init(const DdimFuse* f) {
    const DdimFuse d = *f;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) {
        *d.counter = 0;
        *d.done = 0;
    }
    for (int e = idx; e < d.film_total; e += gridDim.x * blockDim.x) d.film_out[e] = d.film_tab[e];
}

hipError_t launch_ddim_init(const DdimFuse* f, hipStream_t s) {
    hipLaunchKernelGGL(k_ddim_init, dim3(64), dim3(256), 0, s, f);
    return hipGetLastError();
}

// OIHW (or [O][I][1], or [O][I]) -> [tap][I][ld] with output channels contiguous
__global__ void k_repack_conv(const float* src, float* dst, int N, int C, int ntaps, int ld) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * C * ntaps) return;
    const int n = (int)(idx % N);
    const int c = (int)((idx / N) % C);
    const int tap = (int)(idx / ((long)N * C));
    dst[((size_t)tap * C + c) * ld + n] = src[((size_t)n * C + c) * ntaps + tap];
}

// [N][C] (the checkpoint's layout of a 1x1 conv) -> [N / 16][C / 16][64 lanes][4] (ConvArgs::Wpk): one thread per 16-byte quad of the destination
__global__ void k_repack_pw(const float* src, float* dst, int N, int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = C >> 4;
    if (idx >= (long)(N >> 4) * nch * 64) return;
    const int lane = (int)(idx & 63);
    const long bc = idx >> 6;
    const int c = (int)(bc % nch), bn = (int)(bc / nch);
    const int j = lane & 15, q = lane >> 4;
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)(16 * bn + j) * C + 16 * c + 4 * q);
    *reinterpret_cast<float4*>(dst + idx * 4) = v;
}
hipError_t launch_repack_pw(const float* src, float* dst, int N, int C, hipStream_t s) {
    if ((N & 15) || (C & 15)) return hipErrorInvalidValue;
    const long n = (long)(N >> 4) * (C >> 4) * 64;
    hipLaunchKernelGGL(k_repack_pw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, N, C);
    return hipGetLastError();
}
hipError_t launch_repack_conv(const float* src, float* dst, int N, int C, int ntaps, int ld, hipStream_t s) {
    const long n = (long)N * C * ntaps;
    hipLaunchKernelGGL(k_repack_conv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, N, C, ntaps, ld);
    return hipGetLastError();
}

}  // namespace mtv
