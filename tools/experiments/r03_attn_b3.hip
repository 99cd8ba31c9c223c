// k_attention_b3<D, QW, KSP, KBX>: the QKVAttentionLegacy core (MToV/models/ddpm/unet.py:312-326) with its two matrix
// products on the bf16 matrix pipe, at f32 accuracy, through a three-term bf16 split.
//
// Why.  k_attention (kernels.hip) multiplies with v_mfma_f32_16x16x4_f32, the exact-f32 matrix instruction.  On gfx950
// that instruction runs at the f32 VECTOR rate and never co-executes with VALU work: SQ_VALU_MFMA_COEXEC_CYCLES reads
// zero for every f32-MFMA kernel of tools/ubench/mfma_valu (profiles/r03_mfma_valu_counters.txt), so a key block costs
// MFMA time PLUS softmax time (1093 + 458 = 1544 cycles per 64 keys x 16 queries and wave).  v_mfma_f32_16x16x32_bf16
// retires 8x the products per instruction in half the cycles.
//
// How the f32 accuracy is kept.  x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (round to
// nearest even; the subtractions are exact): 24 mantissa bits.  A product x*y is taken as the six partial products
// x0y0 + x0y1 + x1y0 + x0y2 + x1y1 + x2y0 (every one exact in the f32 accumulator; what is dropped is O(2^-24) relative,
// the class of an f32 rounding).  The K = 32 slots of one MFMA hold TWO terms of 4 consecutive head channels (or keys) per
// lane group -- [a | b] against [c | d] contributes a.c + b.d -- so the six products of a 16-deep contraction are THREE
// instructions:      [k0|k1].[q0|q0]   [k0|k1].[q1|q1]   [k2|k0].[q0|q2]
// The probabilities (0..1, feeding a weighted average) are split into two terms only: p0v0 + p0v1 + p1v0 + p1v1 + p0v2,
// again three instructions, with ONE B fragment [p0|p1] against [v0|v0], [v1|v1], [v2|0]; measured on the CPU emulation of exactly
// this arithmetic the output error is 4.5e-6 against 2.7e-6 for plain f32 (tests/test_oracle_golden.py keeps the emulation).
// K and V are split ONCE per tile when they are staged in LDS (amortised over the workgroup's 32-64 queries), q once per
// kernel, p per key block.
//
// Everything else is k_attention's design: a workgroup = QW query tiles x KSP key parts, K rows and V^T staged in LDS
// double buffered, S^T = K Q^T so that a query is a lane column (softmax = register maxima + two lane swaps) and P^T is
// directly the B operand of O^T += V^T P^T, online softmax in the log2 domain, key parts merged through LDS.
// Self-attention over per-plane / whole-clip segments only (the UNet's 72 calls per step); the autoencoder's uniform
// segments, cross-attention and masks stay on k_attention.
#include <type_traits>

#include "mtv_internal.h"

namespace mtv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // {bf16(a) low, bf16(b) high}, round to nearest even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// x[0..3] -> three bf16 terms, each packed as two dwords (elements 0,1 | 2,3)
struct Terms3 { unsigned t0[2], t1[2], t2[2]; };
__device__ __forceinline__ Terms3 split3(const f32x4 x) {
    Terms3 s;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float a = x[2 * h], b = x[2 * h + 1];
        const unsigned p0 = pk_bf16(a, b);
        const float ra = a - bf_lo(p0), rb = b - bf_hi(p0);
        const unsigned p1 = pk_bf16(ra, rb);
        const unsigned p2 = pk_bf16(ra - bf_lo(p1), rb - bf_hi(p1));
        s.t0[h] = p0; s.t1[h] = p1; s.t2[h] = p2;
    }
    return s;
}
// LDS record of 4 consecutive elements: [t0 | t1 | t2 | t0]  (32 bytes; the two 16-byte halves are the A fragments [t0|t1], [t2|t0])
__device__ __forceinline__ void store_rec(char* dst, const Terms3& s) {
    *reinterpret_cast<u32x4*>(dst) = u32x4{s.t0[0], s.t0[1], s.t1[0], s.t1[1]};
    *reinterpret_cast<u32x4*>(dst + 16) = u32x4{s.t2[0], s.t2[1], s.t0[0], s.t0[1]};
}

// V^T record of 4 consecutive keys of one head channel: [t0|t0] [t1|t1] [t2|0]  (48 bytes = the three A fragments that meet B = [p0|p1])
__device__ __forceinline__ void store_rec_v(char* dst, const Terms3& s) {
    *reinterpret_cast<u32x4*>(dst) = u32x4{s.t0[0], s.t0[1], s.t0[0], s.t0[1]};
    *reinterpret_cast<u32x4*>(dst + 16) = u32x4{s.t1[0], s.t1[1], s.t1[0], s.t1[1]};
    *reinterpret_cast<u32x4*>(dst + 32) = u32x4{s.t2[0], s.t2[1], 0u, 0u};
}

__device__ __forceinline__ float b3_swap_max16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float b3_swap_max32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int D, int QW, int KSP, int KBX>
struct AttB3Shape {
    static constexpr int KB = (D >= 64 ? 32 : (D >= 32 ? 64 : 128)) * KBX;   // keys per block
    static constexpr int KROW = D * 8 + 16;           // bytes per staged K row: D/4 records of 32 B + 16 (bank spread of the 16-byte fragment reads)
    static constexpr int VROW = KB * 12 + 16;         // bytes per staged V^T row (one head channel): KB/4 records of 48 B ([v0|v0], [v1|v1], [v2|0]) + 16
    static constexpr int NOB = D / 16, XW = NOB * 4 + 2;
    static constexpr int KS_BYTES = 2 * KB * KROW, VT_BYTES = 2 * D * VROW;
    static constexpr int XO_BYTES = KSP > 1 ? (KSP - 1) * QW * XW * 64 * 4 : 16;
    static constexpr size_t BYTES = (size_t)KS_BYTES + VT_BYTES + XO_BYTES;
};

template <int D, int QW, int KSP, int KBX>
__global__ __launch_bounds__(64 * QW * KSP) void k_attention_b3(const AttnArgs a) {
    touch_kernargs<(int)sizeof(AttnArgs)>();
    static_assert(D == 16 || D == 32 || D == 64, "head dims of the UNet");
    typedef AttB3Shape<D, QW, KSP, KBX> SH;
    constexpr float LOG2E = 1.4426950408889634f;
    constexpr int KB = SH::KB, KROW = SH::KROW, VROW = SH::VROW, NOB = SH::NOB, XW = SH::XW;
    constexpr int NU = D / 16;                     // 16-channel blocks of the head dimension
    constexpr int NKT = KB / 16, WKT = NKT / KSP;  // 16-key tiles per block / per wave
    static_assert(WKT >= 1, "key split wider than the key block");
    constexpr int QPR = D / 4;                     // 4-channel quads per row
    constexpr int NTH = 64 * QW * KSP;
    constexpr int NK = KB * QPR;                   // K staging items: (key, channel quad)
    constexpr int NV = (KB / 4) * D;               // V staging items: (key quad, head channel): 4 keys of one channel = one V^T record
    constexpr int NLDK = (NK + NTH - 1) / NTH, NLDV = (NV + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) char b3_smem[];
    char* const Ks = b3_smem;                                      // [2][KB][KROW]
    char* const Vt = b3_smem + SH::KS_BYTES;                       // [2][D][VROW]
    float* const Xo = reinterpret_cast<float*>(b3_smem + SH::KS_BYTES + SH::VT_BYTES);

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, g = lane >> 4;
    const int qw = wave % QW, kh = wave / QW;
    // block -> (head, batch element, segment, query block): head fastest (one head's K/V in one XCD's L2), see k_attention
    const int rest = FDiv{a.inv_H}((int)blockIdx.x, a.H);
    const int h = (int)blockIdx.x - rest * a.H;
    const int nblk = a.blk_prefix[3];
    const int b = FDiv{a.inv_nblk}(rest, nblk);
    const int qblk = rest - b * nblk;
    const int sn = (qblk >= a.blk_prefix[1] ? 1 : 0) + (qblk >= a.blk_prefix[2] ? 1 : 0);
    const int first = sn == 0 ? 0 : (sn == 1 ? a.blk_prefix[1] : a.blk_prefix[2]);
    const int len = sn == 0 ? a.seg_len[0] : (sn == 1 ? a.seg_len[1] : a.seg_len[2]);
    const int start = sn == 0 ? a.seg_start[0] : (sn == 1 ? a.seg_start[1] : a.seg_start[2]);
    const int q0 = (qblk - first) * (16 * QW) + qw * 16;
    const int RS = 3 * a.C;
    const float* base = a.qkv + (size_t)b * a.L * RS + (size_t)h * 3 * D;      // q | k | v of this head are adjacent
    const float* kbase = base + D;
    const float scale = a.scale;

    // ---- q: 16 queries x D channels of this wave, split once.  Lane (j, g) holds channels 16u + 4g .. + 3 of query j.
    const bool qok = q0 + j < len;
    f32x4 qraw[NU];
    {
        const float* qp = base + (size_t)(start + (qok ? q0 + j : 0)) * RS + 4 * g;
#pragma unroll
        for (int u = 0; u < NU; ++u) qraw[u] = *reinterpret_cast<const f32x4*>(qp + 16 * u);
    }
    // ---- staging registers: thread -> K items (key, quad) and V items (4 keys x 4 channels)
    f32x4 kreg[NLDK], vreg[NLDV];
    auto gload = [&](int kb) {
#pragma unroll
        for (int r = 0; r < NLDK; ++r) {
            const int e = tid + NTH * r;
            const int key = e / QPR, qd = e - key * QPR;
            const bool ok = e < NK && kb + key < len;
            kreg[r] = *reinterpret_cast<const f32x4*>(kbase + (size_t)(start + (ok ? kb + key : 0)) * RS + qd * 4);
        }
#pragma unroll
        for (int r = 0; r < NLDV; ++r) {
            const int e = tid + NTH * r;
            const int kq = e / D, ch = e - kq * D;          // 16 consecutive lanes = 16 consecutive channels of one key: 64-byte segments
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = e < NV && kb + 4 * kq + c < len;
                vreg[r][c] = kbase[D + (size_t)(start + (ok ? kb + 4 * kq + c : 0)) * RS + ch];
            }
        }
    };
    auto lstore = [&](int buf, int kb) {
        char* const kd = Ks + buf * (KB * KROW);
        char* const vd = Vt + buf * (D * VROW);
#pragma unroll
        for (int r = 0; r < NLDK; ++r) {
            const int e = tid + NTH * r;
            const int key = e / QPR, qd = e - key * QPR;
            if (e < NK) {
                const bool in = kb + key < len;
                store_rec(kd + key * KROW + qd * 32, split3(in ? kreg[r] * scale : f32x4{0.f, 0.f, 0.f, 0.f}));
            }
        }
#pragma unroll
        for (int r = 0; r < NLDV; ++r) {
            const int e = tid + NTH * r;
            const int kq = e / D, ch = e - kq * D;
            if (e < NV) {
                f32x4 t;
#pragma unroll
                for (int c = 0; c < 4; ++c) t[c] = kb + 4 * kq + c < len ? vreg[r][c] : 0.f;
                store_rec_v(vd + ch * VROW + kq * 48, split3(t));
            }
        }
    };

    gload(0);
    // q fragments (B operands): Qa = [q0|q0], Qb = [q1|q1], Qc = [q0|q2]   (log2 e and the d^-1/4 scale folded in)
    bf16x8 Qa[NU], Qb[NU], Qc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const Terms3 s = split3(qok ? qraw[u] * (scale * LOG2E) : f32x4{0.f, 0.f, 0.f, 0.f});
        Qa[u] = __builtin_bit_cast(bf16x8, u32x4{s.t0[0], s.t0[1], s.t0[0], s.t0[1]});
        Qb[u] = __builtin_bit_cast(bf16x8, u32x4{s.t1[0], s.t1[1], s.t1[0], s.t1[1]});
        Qc[u] = __builtin_bit_cast(bf16x8, u32x4{s.t0[0], s.t0[1], s.t2[0], s.t2[1]});
    }
    f32x4 oacc[NOB];
#pragma unroll
    for (int o = 0; o < NOB; ++o) oacc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;
    lstore(0, 0);
    __syncthreads();
    int buf = 0;

    auto block = [&](int kb, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const char* const ks = Ks + buf * (KB * KROW);
        const char* const vt = Vt + buf * (D * VROW);
        // ---- S^T tiles: keys (rows) x queries (columns), log2 domain.  MFMAs round-robin over the wave's WKT tiles
        // (independent accumulators back to back), small terms first
        f32x4 st[WKT];
#pragma unroll
        for (int w = 0; w < WKT; ++w) st[w] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            bf16x8 a1[WKT], a2[WKT];
#pragma unroll
            for (int w = 0; w < WKT; ++w) {
                const char* kp = ks + ((kh * WKT + w) * 16 + j) * KROW + g * 32 + u * 128;
                a1[w] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(kp));          // [k0|k1]
                a2[w] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(kp + 16));     // [k2|k0]
            }
#pragma unroll
            for (int w = 0; w < WKT; ++w) st[w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[w], Qc[u], st[w], 0, 0, 0);
#pragma unroll
            for (int w = 0; w < WKT; ++w) st[w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[w], Qb[u], st[w], 0, 0, 0);
#pragma unroll
            for (int w = 0; w < WKT; ++w) st[w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[w], Qa[u], st[w], 0, 0, 0);
        }
        if constexpr (!FULL) {
#pragma unroll
            for (int w = 0; w < WKT; ++w) {
                const int key0 = kb + (kh * WKT + w) * 16 + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) st[w][r] = key0 + r >= len ? -INFINITY : st[w][r];
            }
        }
        float mx = st[0][0];
#pragma unroll
        for (int w = 0; w < WKT; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[w][r]);
        mx = b3_swap_max16(mx);
        mx = b3_swap_max32(mx);
        const float mn = fmaxf(m, mx);
        const bool live = FULL || mn != -INFINITY;
        const float alpha = live ? __builtin_amdgcn_exp2f(m - mn) : 1.0f;
        m = mn;
        float ps = 0.f;
        // ---- p = exp2(s - m), split into two bf16 terms on the spot: ONE B fragment [p0|p1] per tile (no duplicated registers)
        bf16x8 Pf[WKT];
#pragma unroll
        for (int w = 0; w < WKT; ++w) {
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = live ? __builtin_amdgcn_exp2f(st[w][r] - mn) : 0.f;
                ps += p[r];
            }
            const unsigned h0 = pk_bf16(p[0], p[1]), h1 = pk_bf16(p[2], p[3]);
            const unsigned l0 = pk_bf16(p[0] - bf_lo(h0), p[1] - bf_hi(h0)), l1 = pk_bf16(p[2] - bf_lo(h1), p[3] - bf_hi(h1));
            Pf[w] = __builtin_bit_cast(bf16x8, u32x4{h0, h1, l0, l1});
        }
        lsum = lsum * alpha + ps;
        // ---- O^T += V^T P^T.  Independent accumulators back to back: the NOB output blocks, or (d = 16: one block) the
        // even / odd key tiles of the wave, summed afterwards
        constexpr int OS = (NOB == 1 && WKT >= 2) ? 2 : 1;
        f32x4 pacc[NOB][OS];
#pragma unroll
        for (int o = 0; o < NOB; ++o) {
            pacc[o][0] = oacc[o] * alpha;
            if constexpr (OS == 2) pacc[o][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int w0 = 0; w0 < WKT; w0 += OS) {
            bf16x8 a0[OS][NOB], a1[OS][NOB], a2[OS][NOB];
#pragma unroll
            for (int c = 0; c < OS; ++c)
#pragma unroll
                for (int o = 0; o < NOB; ++o) {
                    const char* vp = vt + (16 * o + j) * VROW + ((kh * WKT + w0 + c) * 4 + g) * 48;      // keys 4g .. 4g+3 of the tile, channel 16 o + j
                    a0[c][o] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(vp));          // [v0|v0]
                    a1[c][o] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(vp + 16));     // [v1|v1]
                    a2[c][o] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(vp + 32));     // [v2| 0]
                }
#pragma unroll
            for (int c = 0; c < OS; ++c)
#pragma unroll
                for (int o = 0; o < NOB; ++o) pacc[o][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[c][o], Pf[w0 + c], pacc[o][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < OS; ++c)
#pragma unroll
                for (int o = 0; o < NOB; ++o) pacc[o][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[c][o], Pf[w0 + c], pacc[o][c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < OS; ++c)
#pragma unroll
                for (int o = 0; o < NOB; ++o) pacc[o][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[c][o], Pf[w0 + c], pacc[o][c], 0, 0, 0);
        }
#pragma unroll
        for (int o = 0; o < NOB; ++o) {
            oacc[o] = pacc[o][0];
            if constexpr (OS == 2) oacc[o] += pacc[o][1];
        }
    };
    for (int kb = 0; kb < len; kb += KB) {
        const bool more = kb + KB < len;
        if (more) gload(kb + KB);
        if (kb + KB <= len) block(kb, std::true_type{});
        else block(kb, std::false_type{});
        if (more) lstore(buf ^ 1, kb + KB);
        __syncthreads();
        buf ^= 1;
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if constexpr (KSP > 1) {
        if (kh > 0) {
            float* p = Xo + ((size_t)((kh - 1) * QW + qw) * XW) * 64 + lane;
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
            const float* p = Xo + ((size_t)((k2 - 1) * QW + qw) * XW) * 64 + lane;
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
    const float inv = 1.0f / lsum;
    if (q0 + j < len) {
        float* op = a.out + ((size_t)b * a.L + start + q0 + j) * a.C + (size_t)h * D;
#pragma unroll
        for (int o = 0; o < NOB; ++o) *reinterpret_cast<f32x4*>(op + 16 * o + 4 * g) = oacc[o] * inv;
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <int D, int QW, int KSP, int KBX>
static hipError_t b3_launch(const AttnArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((k_attention_b3<D, QW, KSP, KBX>), grid, dim3(64 * QW * KSP), (AttB3Shape<D, QW, KSP, KBX>::BYTES), s, a);
    return hipGetLastError();
}
template <int D, int QW, int KSP, int KBX>
static hipError_t b3_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_b3<D, QW, KSP, KBX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
hipError_t attn_b3_init_attrs() {
    hipError_t e;
#define B3_A(D, QW, KSP, KBX) if ((e = b3_attr<D, QW, KSP, KBX>()) != hipSuccess) return e
    B3_A(16, 4, 2, 1); B3_A(16, 2, 4, 1); B3_A(16, 1, 4, 1);
    B3_A(32, 4, 2, 1); B3_A(32, 2, 4, 1); B3_A(32, 1, 4, 1);
    B3_A(64, 4, 2, 1); B3_A(64, 1, 2, 1);
#undef B3_A
    return hipSuccess;
}

// Takes the launch when the split-bf16 kernel covers it: plain self-attention over per-plane / whole-clip segments, head
// dim 16 / 32 / 64.  `a` arrives from launch_attention with blk_prefix / bps / reciprocals NOT yet set: the workgroup
// shape is chosen here (same policy as launch_attention: 64-query workgroups from 128 workgroups up, 32-query ones for
// uneven segments that would leave at most one workgroup per CU, 16-query x 4 key parts for short segments).
bool attn_b3_eligible(const AttnArgs& a) {
    if (a.kv || a.kmask || a.seg_uniform || a.H < 1 || a.C % a.H || a.qkv_ks > 1) return false;     // (slab-summed qkv of the deep levels: k_attention only)
    const int d = a.C / a.H;
    return d == 16 || d == 32 || d == 64;
}

hipError_t launch_attention_b3(const AttnArgs& a0, hipStream_t s) {
    AttnArgs a = a0;
    if (!attn_b3_eligible(a)) return hipErrorInvalidValue;
    const int d = a.C / a.H;
    long blocks64 = 0;
    bool uneven = false;
    for (int i = 0; i < a.nseg; ++i) {
        blocks64 += (a.seg_len[i] + 63) / 64;
        uneven |= a.seg_len[i] != a.seg_len[0];
    }
    const bool wide = blocks64 * a.H * a.B >= 128;
    const bool half = wide && (d == 16 || d == 32) && uneven && blocks64 * a.H * a.B <= 256;
    const int qw = wide ? (half ? 2 : 4) : 1;
    a.blk_prefix[0] = 0;
    for (int i = 0; i < a.nseg; ++i) a.blk_prefix[i + 1] = a.blk_prefix[i] + (a.seg_len[i] + 16 * qw - 1) / (16 * qw);
    const int nblk = a.blk_prefix[a.nseg];
    for (int i = a.nseg + 1; i < 4; ++i) a.blk_prefix[i] = nblk;
    a.bps = 1;
    if ((long)nblk * a.H * a.B >= (1l << 22)) return hipErrorInvalidValue;
    a.inv_H = 1.0f / (float)a.H;
    a.inv_nblk = 1.0f / (float)nblk;
    a.inv_bps = 1.0f;
    dim3 grid((unsigned)(nblk * a.H * a.B));
    // (key blocks: 128 / 64 / 32 keys at d = 16 / 32 / 64, ~80 KB of LDS double buffered: two workgroups per CU.  Double
    // blocks would need 170 KB with the 48-byte V^T records.)
#define B3_GO(D, QW, KSP, KBX) return b3_launch<D, QW, KSP, KBX>(a, grid, s)
    switch (d) {
        case 16:
            if (half) B3_GO(16, 2, 4, 1);
            if (wide) B3_GO(16, 4, 2, 1);
            B3_GO(16, 1, 4, 1);
        case 32:
            if (half) B3_GO(32, 2, 4, 1);
            if (wide) B3_GO(32, 4, 2, 1);
            B3_GO(32, 1, 4, 1);
        case 64:           // (32-key blocks: two 16-key tiles, so at most two key parts)
            if (wide) B3_GO(64, 4, 2, 1);
            B3_GO(64, 1, 2, 1);
    }
#undef B3_GO
    return hipErrorInvalidValue;
}

}  // namespace mtv
