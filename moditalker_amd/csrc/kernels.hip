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
#include <cstdio>
#include <cstdlib>

namespace mtv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

// =====================================================================================
// Implicit-GEMM convolution (3x3 via gather table, 1x1, fused 1x1 skip), GN/FiLM/SiLU prologue
// =====================================================================================
// Replaces: ResBlock in_layers / out_layers / skip_connection (unet.py:131-167,178-207), the
// attention blocks' qkv / proj_out conv1d (unet.py:234,242,251,253), the stem and head convs
// (unet.py:714,971-975).
//
// Work split: grid.x = B * ceil(Lout / (16*MT)) row tiles (a tile never straddles a batch
// element), grid.y = N / (16*NT) column tiles; the NW waves of a workgroup split the K axis
// (taps x input channels, in chunks of 16 channels) and are summed in LDS in fixed order
// (deterministic).  Per chunk a wave issues MT + 4 sixteen-byte loads for 16*MT*NT/4 MFMAs:
//   A: lane (i,q) loads x[row i][c0 + 4q .. 4q+3]; MFMA step s uses component s, i.e. K slot q of
//      step s is channel c0 + 4q + s;
//   B: lane (j,q) loads W[c0 + 4q + s][n0 + NT*j .. +NT-1] for s = 0..3; column block nb of the
//      wave tile is output channel n0 + NT*j + nb.
template <int MT, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void k_conv(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, q = lane >> 4;
    constexpr int ROWS = 16 * MT, COLS = 16 * NT;
    const int tiles_per_b = (a.Lout + ROWS - 1) / ROWS;
    const int b = blockIdx.x / tiles_per_b;
    const int tok0 = (blockIdx.x - b * tiles_per_b) * ROWS;
    const int n0 = blockIdx.y * COLS;
    const int Cmain = a.Cmain;

    // ---- prologue: per-(plane, channel) affine coefficients {gn scale, gn bias, 1+film scale, film shift}
    f32x4* coef = reinterpret_cast<f32x4*>(smem);
    float* red = smem;
    if (a.gn.sums) {
        __shared__ float2 s_mr[3][32];
        for (int e = tid; e < 96; e += NW * 64) {
            const int sg = e >> 5, g = e & 31;
            const double* S = a.gn.sums + (size_t)b * 192;
            double s, ss, n;
            if (a.gn.whole) {
                s = S[g * 2] + S[64 + g * 2] + S[128 + g * 2];
                ss = S[g * 2 + 1] + S[64 + g * 2 + 1] + S[128 + g * 2 + 1];
                n = (double)a.seg_src.L * a.gn.gs;
            } else {
                s = S[sg * 64 + g * 2];
                ss = S[sg * 64 + g * 2 + 1];
                const int len = sg == 0 ? a.seg_src.b1 : (sg == 1 ? a.seg_src.b2 - a.seg_src.b1 : a.seg_src.L - a.seg_src.b2);
                n = (double)len * a.gn.gs;
            }
            const double mean = s / n;
            double var = ss / n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            s_mr[sg][g] = make_float2((float)mean, (float)(1.0 / sqrt(var + 1e-5)));
        }
        __syncthreads();
        for (int idx = tid; idx < 3 * Cmain; idx += NW * 64) {
            const int sg = idx / Cmain, c = idx - sg * Cmain;
            const float2 mr = s_mr[sg][c / a.gn.gs];
            const float sc = mr.y * a.gn.gamma[c];
            const float bi = a.gn.beta[c] - sc * mr.x;
            float s1 = 1.0f, sh = 0.0f;
            if (a.gn.film) {
                const float* f = a.gn.film + (size_t)b * a.gn.film_stride;
                s1 = 1.0f + f[c];
                sh = f[Cmain + c];
            }
            coef[idx] = f32x4{sc, bi, s1, sh};
        }
        red = smem + 12 * Cmain;
        __syncthreads();
    }

    // ---- K loop over this wave's chunk range
    const int cpm = Cmain >> 4;
    const int nmain_chunks = a.ntaps * cpm;
    const int nchunks = nmain_chunks + (a.Cskip >> 4);
    const int ch0 = (int)(((long)nchunks * wave) / NW), ch1 = (int)(((long)nchunks * (wave + 1)) / NW);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[mt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool do_gn = a.gn.sums != nullptr;
    const int nbase = n0 + NT * i;   // this lane's first output column (i doubles as j for B)

    for (int ch = ch0; ch < ch1; ++ch) {
        const bool is_skip = ch >= nmain_chunks;
        int tap = 0, c;
        if (!is_skip) {
            tap = ch / cpm;
            c = (ch - tap * cpm) << 4;
        } else {
            c = (ch - nmain_chunks) << 4;
        }
        // which source part holds channel c
        int p = is_skip ? a.nmain : 0;
        int coff = 0;
        const int pend = is_skip ? a.nmain + a.nskip : a.nmain;
        while (p + 1 < pend && c >= coff + a.C[p]) {
            coff += a.C[p];
            ++p;
        }
        const float* sp = a.src[p];
        const int Cp = a.C[p];
        const int Ls = is_skip ? a.Lskip : a.Lsrc;
        const int* gt = is_skip ? a.gather_skip : a.gather;

        // B fragment: 4 rows of W, NT consecutive columns each
        const size_t wrow = (size_t)(is_skip ? nmain_chunks * 16 + c : tap * Cmain + c) + 4 * q;
        const float* wp = a.W + wrow * a.ldw + nbase;
        float bv[4][NT];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (NT == 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(wp + (size_t)s * a.ldw);
                bv[s][0] = t[0]; bv[s][1] = t[1]; bv[s][2] = t[2]; bv[s][3] = t[3];
            } else if constexpr (NT == 2) {
                const float2 t = *reinterpret_cast<const float2*>(wp + (size_t)s * a.ldw);
                bv[s][0] = t.x; bv[s][1] = t.y;
            } else {
                bv[s][0] = wp[(size_t)s * a.ldw];
            }
        }
        // A fragment(s)
        float av[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int tok = tok0 + 16 * mt + i;
            int st = -1;
            if (tok < a.Lout) st = gt ? gt[(is_skip ? 0 : tap * a.Lout) + tok] : tok;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (st >= 0) {
                v = *reinterpret_cast<const f32x4*>(sp + ((size_t)b * Ls + st) * Cp + (c - coff) + 4 * q);
                if (do_gn && !is_skip) {
                    const int sg = a.gn.whole ? 0 : (st >= a.seg_src.b2 ? 2 : (st >= a.seg_src.b1 ? 1 : 0));
                    const f32x4* cf = coef + sg * Cmain + c + 4 * q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x4 k = cf[e];
                        float y = fmaf(v[e], k[0], k[1]);
                        y = fmaf(y, k[2], k[3]);
                        v[e] = a.gn.act ? silu_f(y) : y;
                    }
                }
            }
            av[mt][0] = v[0]; av[mt][1] = v[1]; av[mt][2] = v[2]; av[mt][3] = v[3];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < NT; ++nb)
                    acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][s], bv[s][nb], acc[mt][nb], 0, 0, 0);
    }

    // ---- cross-wave (split-K) reduction in LDS, fixed order, then epilogue
    constexpr int LDR = COLS + 4;
    if (NW > 1) {
        float* my = red + (size_t)wave * ROWS * LDR;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) my[(16 * mt + 4 * q + r) * LDR + NT * i + nb] = acc[mt][nb][r];
        __syncthreads();
        constexpr int QUADS = ROWS * (COLS / 4);
        for (int e = tid; e < QUADS; e += NW * 64) {
            const int rr = e / (COLS / 4), cq = e - rr * (COLS / 4);
            f32x4 v = *reinterpret_cast<const f32x4*>(red + rr * LDR + cq * 4);
#pragma unroll
            for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(red + (size_t)w * ROWS * LDR + rr * LDR + cq * 4);
            const int tok = tok0 + rr, n = n0 + cq * 4;
            if (tok >= a.Lout || n >= a.N) continue;
            int rs = tok;
            if (a.res && a.gather_skip) rs = a.gather_skip[tok];
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int nn = n + e4;
                if (nn >= a.N) break;
                float o = v[e4] + a.bias[nn];
                if (a.bias2) o += a.bias2[nn];
                if (a.bias_b) o += a.bias_b[(size_t)b * a.bias_b_stride + nn];
                if (a.res) o += a.res[((size_t)b * a.Lskip + rs) * a.N + nn];
                if (a.out_cm) a.out[((size_t)b * a.N + nn) * a.Lout + tok] = o;
                else a.out[((size_t)b * a.Lout + tok) * a.N + nn] = o;
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = tok0 + 16 * mt + 4 * q + r;
                if (tok >= a.Lout) continue;
                int rs = tok;
                if (a.res && a.gather_skip) rs = a.gather_skip[tok];
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) {
                    const int nn = nbase + nb;
                    if (nn >= a.N) break;
                    float o = acc[mt][nb][r] + a.bias[nn];
                    if (a.bias2) o += a.bias2[nn];
                    if (a.bias_b) o += a.bias_b[(size_t)b * a.bias_b_stride + nn];
                    if (a.res) o += a.res[((size_t)b * a.Lskip + rs) * a.N + nn];
                    if (a.out_cm) a.out[((size_t)b * a.N + nn) * a.Lout + tok] = o;
                    else a.out[((size_t)b * a.Lout + tok) * a.N + nn] = o;
                }
            }
    }
}

ConvTile conv_pick_tile(int B, int Lout, int N, int nchunks) {
    // Fill >= ~1024 waves (256 CUs x 4 SIMDs) while keeping the per-wave tile as large as the
    // problem allows (fewer L2 bytes per MFMA) and >= 2 chunks per wave.
    static const int cand[][2] = {{4, 4}, {2, 4}, {1, 4}, {1, 2}, {1, 1}};
    // debugging / tuning aid: MTV_FORCE_TILE="MT,NT,NW" pins one tile shape for every conv
    static int forced[3] = {-1, 0, 0};
    if (forced[0] == -1) {
        forced[0] = 0;
        if (const char* e = getenv("MTV_FORCE_TILE")) {
            int a = 0, b = 0, c = 0;
            if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) { forced[0] = a; forced[1] = b; forced[2] = c; }
        }
    }
    if (forced[0] > 0) {
        ConvTile t{forced[0], forced[1], forced[2]};
        if (t.NW > nchunks) t.NW = 1;
        return t;
    }
    ConvTile best{1, 1, 1};
    double best_score = -1.0;
    for (auto& c : cand) {
        const int MT = c[0], NT = c[1];
        if (NT * 16 > ((N + 15) / 16) * 16 && NT > 1) continue;     // tile wider than N
        if (MT > 1 && 16 * (MT / 2) >= Lout) continue;               // tile taller than needed
        const long tiles = (long)B * ((Lout + 16 * MT - 1) / (16 * MT)) * ((N + 16 * NT - 1) / (16 * NT));
        for (int NW = 1; NW <= 16; NW *= 2) {
            if (nchunks / NW < 1) break;
            if ((size_t)NW * 16 * MT * (16 * NT + 4) * 4 > 64 * 1024) break;
            const double waves = (double)tiles * NW;
            const double fill = waves >= 1024.0 ? 1.0 : waves / 1024.0;
            const double eff = (4.0 * MT * NT) / (MT + 4.0) / 8.0;   // MFMAs per load, normalised to (4,4)
            const double chunks_per_wave = (double)nchunks / NW;
            const double depth = chunks_per_wave >= 4 ? 1.0 : 0.6 + 0.1 * chunks_per_wave;
            const double score = fill * (0.55 + 0.45 * eff) * depth;
            if (score > best_score + 1e-9) {
                best_score = score;
                best = ConvTile{MT, NT, NW};
            }
        }
    }
    return best;
}

size_t conv_smem_bytes(const ConvArgs& a, ConvTile t) {
    size_t coef = a.gn.sums ? (size_t)12 * a.Cmain * sizeof(float) : 0;
    size_t red = t.NW > 1 ? (size_t)t.NW * 16 * t.MT * (16 * t.NT + 4) * sizeof(float) : 0;
    return coef + red;
}

template <int MT, int NT, int NW>
static hipError_t launch_conv_t(const ConvArgs& a, hipStream_t s) {
    const int tiles = (a.Lout + 16 * MT - 1) / (16 * MT);
    dim3 grid(a.B * tiles, (a.N + 16 * NT - 1) / (16 * NT));
    const size_t smem = conv_smem_bytes(a, ConvTile{MT, NT, NW});
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
        case 16: return launch_conv_t<MT, NT, 16>(a, s);
    }
    return hipErrorInvalidValue;
}

template <int MT, int NT, int NW>
static hipError_t conv_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv<MT, NT, NW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
}
template <int MT, int NT>
static hipError_t conv_attr_nw() {
    hipError_t e;
    if ((e = conv_attr<MT, NT, 1>()) != hipSuccess) return e;
    if ((e = conv_attr<MT, NT, 2>()) != hipSuccess) return e;
    if ((e = conv_attr<MT, NT, 4>()) != hipSuccess) return e;
    if ((e = conv_attr<MT, NT, 8>()) != hipSuccess) return e;
    return conv_attr<MT, NT, 16>();
}
// Dynamic LDS above 64 KB must be opted into once per kernel; done at mtv_create (never under capture).
hipError_t conv_init_attrs() {
    hipError_t e;
    if ((e = conv_attr_nw<4, 4>()) != hipSuccess) return e;
    if ((e = conv_attr_nw<2, 4>()) != hipSuccess) return e;
    if ((e = conv_attr_nw<1, 4>()) != hipSuccess) return e;
    if ((e = conv_attr_nw<1, 2>()) != hipSuccess) return e;
    return conv_attr_nw<1, 1>();
}

hipError_t launch_conv(const ConvArgs& a, ConvTile t, hipStream_t s) {
    if (t.MT == 4 && t.NT == 4) return launch_conv_nw<4, 4>(a, t.NW, s);
    if (t.MT == 2 && t.NT == 4) return launch_conv_nw<2, 4>(a, t.NW, s);
    if (t.MT == 1 && t.NT == 4) return launch_conv_nw<1, 4>(a, t.NW, s);
    if (t.MT == 1 && t.NT == 2) return launch_conv_nw<1, 2>(a, t.NW, s);
    if (t.MT == 1 && t.NT == 1) return launch_conv_nw<1, 1>(a, t.NW, s);
    return hipErrorInvalidValue;
}

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
    const int nq = a.C >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.B * a.seg_dst.L * nq;
    if (idx >= total) return;
    const int cq = (int)(idx % nq);
    const int tok = (int)((idx / nq) % a.seg_dst.L);
    const int b = (int)(idx / ((long)nq * a.seg_dst.L));
    int sg, y, x, wd, ws, base_s;
    if (tok < a.seg_dst.b1) { sg = 0; wd = a.r_dst; y = tok / wd; x = tok - y * wd; base_s = 0; }
    else if (tok < a.seg_dst.b2) { sg = 1; wd = a.r_dst; const int tt = tok - a.seg_dst.b1; y = tt / wd; x = tt - y * wd; base_s = a.seg_src.b1; }
    else { sg = 2; wd = a.r_dst; const int tt = tok - a.seg_dst.b2; y = tt / wd; x = tt - y * wd; base_s = a.seg_src.b2; }
    ws = 2 * wd;
    const int c = cq << 2;
    // per-channel GN coefficients from the fp64 sums of this plane
    const double* S = a.sums + ((size_t)b * 3 + sg) * 64;
    const int len = sg == 0 ? a.seg_src.b1 : (sg == 1 ? a.seg_src.b2 - a.seg_src.b1 : a.seg_src.L - a.seg_src.b2);
    const double n = (double)len * a.gs;
    float sc[4], bi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int g = (c + e) / a.gs;
        const double mean = S[g * 2] / n;
        double var = S[g * 2 + 1] / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        sc[e] = rstd * a.gamma[c + e];
        bi[e] = a.beta[c + e] - sc[e] * (float)mean;
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
    *reinterpret_cast<f32x4*>(a.out_act + o) = sa * 0.25f;
    *reinterpret_cast<f32x4*>(a.out_x + o) = sx * 0.25f;
}

hipError_t launch_pool_down(const PoolArgs& a, hipStream_t s) {
    const long total = (long)a.B * a.seg_dst.L * (a.C >> 2);
    hipLaunchKernelGGL(k_pool_down, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =====================================================================================
// QKVAttentionLegacy core (unet.py:312-326), flash-style, exact-f32 MFMA, online softmax
// =====================================================================================
// One wave owns 16 queries of one (batch, segment, head) and walks the keys of its segment in
// blocks of 64.  It computes S^T = K Q^T (keys x queries) so that a query is a lane COLUMN: the
// softmax reductions over keys are 16 register values + two cross-lane steps, and the P^T
// registers are directly the B operand of O^T += V^T P^T (K slot g of step s is key 4g+s, which
// is exactly the D layout of S^T).  q and k are each pre-multiplied by d^-1/4 like the reference.
template <int D>
__global__ __launch_bounds__(256) void k_attention(const AttnArgs a) {
    constexpr int VW = D >= 16 ? 4 : D / 4;        // floats per q/k vector load
    constexpr int NV = D >= 16 ? D / 16 : 1;       // vector loads per row
    constexpr int NOB = D >= 16 ? D / 16 : 1;      // 16-row output blocks of O^T
    constexpr int PVW = D >= 64 ? 4 : (D == 32 ? 2 : 1);   // V floats per load
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    const int qt = blockIdx.x * 4 + wave;
    if (qt >= a.tile_prefix[a.nseg]) return;
    int sg = 0;
    while (sg + 1 < a.nseg && qt >= a.tile_prefix[sg + 1]) ++sg;
    const int q0 = (qt - a.tile_prefix[sg]) * 16;
    const int start = a.seg_start[sg], len = a.seg_len[sg];
    const int h = blockIdx.y, b = blockIdx.z;
    const int RS = 3 * a.C;
    const float* base = a.qkv + (size_t)b * a.L * RS + (size_t)h * 3 * D;
    const float scale = a.scale;

    float qreg[NV][VW];
    {
        const bool ok = q0 + j < len;
        const float* qp = base + (size_t)(start + (ok ? q0 + j : 0)) * RS + VW * g;
#pragma unroll
        for (int u = 0; u < NV; ++u)
#pragma unroll
            for (int e = 0; e < VW; ++e) qreg[u][e] = ok ? qp[16 * u + e] * scale : 0.f;
    }
    f32x4 oacc[NOB];
#pragma unroll
    for (int o = 0; o < NOB; ++o) oacc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;

    for (int kb = 0; kb < len; kb += 64) {
        f32x4 st[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int key = kb + kt * 16 + j;          // A operand row i = j
            const bool ok = key < len;
            const float* kp = base + (size_t)(start + (ok ? key : 0)) * RS + D + VW * g;
            f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                float kv[VW];
                if constexpr (VW == 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(kp + 16 * u);
                    kv[0] = t[0]; kv[1] = t[1]; kv[2] = t[2]; kv[3] = t[3];
                } else {
#pragma unroll
                    for (int e = 0; e < VW; ++e) kv[e] = kp[e];
                }
#pragma unroll
                for (int e = 0; e < VW; ++e)
                    s4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ok ? kv[e] * scale : 0.f, qreg[u][e], s4, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kb + kt * 16 + 4 * g + r >= len) s4[r] = -INFINITY;
            st[kt] = s4;
        }
        float mx = st[0][0];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);
        const float alpha = __expf(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(st[kt][r] - mn);
                st[kt][r] = p;
                ps += p;
            }
        lsum = lsum * alpha + ps;
#pragma unroll
        for (int o = 0; o < NOB; ++o) oacc[o] *= alpha;
        // O^T += V^T P^T
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kb + kt * 16 >= len) break;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int key = kb + kt * 16 + 4 * g + s;
                const bool ok = key < len;
                const float* vp = base + (size_t)(start + (ok ? key : 0)) * RS + 2 * D;
                const float p = st[kt][s];
                if constexpr (D >= 64) {
#pragma unroll
                    for (int u = 0; u < D / 64; ++u) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(vp + 64 * u + 4 * j);
                        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb)
                            oacc[4 * u + nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[nb], p, oacc[4 * u + nb], 0, 0, 0);
                    }
                } else if constexpr (D == 32) {
                    float2 v = *reinterpret_cast<const float2*>(vp + 2 * j);
                    if (!ok) v = make_float2(0.f, 0.f);
                    oacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, p, oacc[0], 0, 0, 0);
                    oacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, p, oacc[1], 0, 0, 0);
                } else {
                    const float v = (ok && j < D) ? vp[j] : 0.f;
                    oacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, p, oacc[0], 0, 0, 0);
                }
            }
        }
    }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    const float inv = 1.0f / lsum;
    if (q0 + j < len) {
        float* op = a.out + ((size_t)b * a.L + start + q0 + j) * a.C + (size_t)h * D;
        if constexpr (D >= 64) {
#pragma unroll
            for (int u = 0; u < D / 64; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f32x4 o = f32x4{oacc[4 * u][r], oacc[4 * u + 1][r], oacc[4 * u + 2][r], oacc[4 * u + 3][r]};
                    *reinterpret_cast<f32x4*>(op + 64 * u + 16 * g + 4 * r) = o * inv;
                }
        } else if constexpr (D == 32) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                op[8 * g + 2 * r] = oacc[0][r] * inv;
                op[8 * g + 2 * r + 1] = oacc[1][r] * inv;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < D) op[4 * g + r] = oacc[0][r] * inv;
        }
    }
    (void)PVW;
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
    const int d = a.C / a.H;
    dim3 grid((a.tile_prefix[a.nseg] + 3) / 4, a.H, a.B), block(256);
    switch (d) {
        case 4: hipLaunchKernelGGL(k_attention<4>, grid, block, 0, s, a); break;
        case 8: hipLaunchKernelGGL(k_attention<8>, grid, block, 0, s, a); break;
        case 16: hipLaunchKernelGGL(k_attention<16>, grid, block, 0, s, a); break;
        case 32: hipLaunchKernelGGL(k_attention<32>, grid, block, 0, s, a); break;
        case 64: hipLaunchKernelGGL(k_attention<64>, grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL(k_attention<128>, grid, block, 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// =====================================================================================
// Small dense layers: time_embed MLP and all ResBlock emb_layers in one launch
// =====================================================================================
// out[b][n] = sum_k f(x[b][k]) * W[n][k] + bias[n]   (unet.py:700-705,148-154,193); one wave per n.
__global__ __launch_bounds__(256) void k_linear(const LinearArgs a) {
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
// DDIM update (ddpm.py:278-282,346-351,386-398), eta-general, separate roundings like the reference
// =====================================================================================
__global__ void k_ddim_update(float* x, const float* eps, const float* noise, const DdimStep* steps,
                              const int* counter, int64_t n_per_draw, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const DdimStep st = steps[*counter];
    const float e = eps[idx];
    float x0 = __fsub_rn(__fmul_rn(st.sqrt_recip_ac, x[idx]), __fmul_rn(st.sqrt_recipm1_ac, e));
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    if (st.last) {
        x[idx] = x0;
        return;
    }
    const float nz = st.noise_index >= 0 ? noise[(int64_t)st.noise_index * n_per_draw + idx] : 0.f;
    x[idx] = __fadd_rn(__fadd_rn(__fmul_rn(x0, st.sqrt_ac_next), __fmul_rn(st.c, e)), __fmul_rn(st.sigma, nz));
}

hipError_t launch_ddim_update(float* x, const float* eps, const float* noise, const DdimStep* steps,
                              const int* counter, int64_t n_per_draw, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_ddim_update, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, eps, noise, steps, counter, n_per_draw, n);
    return hipGetLastError();
}

__global__ void k_ddim_advance(const DdimStep* steps, int* counter, int n_steps, int64_t* tbuf, int B) {
    __shared__ int nxt;
    if (threadIdx.x == 0) {
        nxt = *counter + 1;
        *counter = nxt;
    }
    __syncthreads();
    if (nxt < n_steps)
        for (int b = threadIdx.x; b < B; b += blockDim.x) tbuf[b] = steps[nxt].t;
}

hipError_t launch_ddim_advance(const DdimStep* steps, int* counter, int n_steps, int64_t* tbuf, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_ddim_advance, dim3(1), dim3(64), 0, s, steps, counter, n_steps, tbuf, B);
    return hipGetLastError();
}

__global__ void k_ddim_init(const DdimStep* steps, int* counter, int64_t* tbuf, int B) {
    if (threadIdx.x == 0) *counter = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) tbuf[b] = steps[0].t;
}

hipError_t launch_ddim_init(const DdimStep* steps, int* counter, int64_t* tbuf, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_ddim_init, dim3(1), dim3(64), 0, s, steps, counter, tbuf, B);
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

hipError_t launch_repack_conv(const float* src, float* dst, int N, int C, int ntaps, int ld, hipStream_t s) {
    const long n = (long)N * C * ntaps;
    hipLaunchKernelGGL(k_repack_conv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, N, C, ntaps, ld);
    return hipGetLastError();
}

}  // namespace mtv
