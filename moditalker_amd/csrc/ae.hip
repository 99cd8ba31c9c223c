// The autoencoder steps either side of the denoising loop, MI355X-native:
//   ViTAutoencoder.decode_from_sample  (MToV/models/autoencoder/autoencoder_vit.py:257-275)  latents -> RGB frames
//   ViTAutoencoder.extract             (autoencoder_vit.py:212-255)                          video   -> latents
// over the TimeSformer stacks of MToV/models/autoencoder/vit_modules.py:150-303 and the three small "quant"
// transformers of autoencoder_vit.py:15-84.
//
// Layout: tokens row-major [B][T*hp*wp][C] fp32 in (frame, y, x) order, exactly the reference's 'b (f h w) c'.
// Every Linear is a k_conv launch (ntaps = 1: the exact-f32 MFMA GEMM of conv.hip with bias / residual epilogue);
// attention is k_attention<64> (<48> for the quant stacks) in its uniform-segment mode.  What is specific to this
// model lives here: LayerNorm, rotary embedding, GEGLU, the latent -> token expansion, patch gathering, the
// ConvTranspose pixel scatter, and the token re-orderings expressed as GEMM row gathers:
//   * time attention works on '(b n) f d' sequences (16 frames of one site): its qkv GEMM reads the LayerNorm output
//     through a row gather (output token (n, f) <- source token (f, n)) so the sequences are contiguous, and its
//     to_out GEMM gathers back ((f, n) <- (n, f)) while adding the residual in the natural order;
//   * the quant stacks read per-plane sequences + a learned token through a gather kernel.
#include "plan_internal.h"

namespace mtv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// weight repacking
// ---------------------------------------------------------------------------------------------
// Linear [3*H*d][C], rows (q|k|v, head, dd)  ->  [C][ld], columns (head, q|k|v, dd)
__global__ void k_repack_qkv(const float* src, float* dst, int H, int d, int C, int ld) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)3 * H * d * C;
    if (idx >= total) return;
    const int n = (int)(idx / C), c = (int)(idx - (long)n * C);
    const int j = n / (H * d), h = (n - j * H * d) / d, dd = n % d;
    dst[(size_t)c * ld + h * 3 * d + j * d + dd] = src[idx];
}
hipError_t launch_repack_qkv(const float* src, float* dst, int H, int d, int C, int ld, hipStream_t s) {
    const long n = (long)3 * H * d * C;
    hipLaunchKernelGGL(k_repack_qkv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, H, d, C, ld);
    return hipGetLastError();
}
__global__ void k_repeat(const float* src, float* dst, int n, int rep) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n * rep) dst[idx] = src[idx / rep];
}
hipError_t launch_repeat(const float* src, float* dst, int n, int rep, hipStream_t s) {
    hipLaunchKernelGGL(k_repeat, dim3((n * rep + 255) / 256), dim3(256), 0, s, src, dst, n, rep);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the channel axis (PreNorm, vit_modules.py:70-79 / autoencoder_vit.py:15-23): one wave per token,
// two-pass fp32 (mean, then centred sum of squares) like ATen's layer_norm; eps = 1e-5
// ---------------------------------------------------------------------------------------------
template <int VPL>   // float4 per lane: C = 256 * VPL ... handled generally below
__global__ __launch_bounds__(256) void k_layernorm(const float* x, const float* gamma, const float* beta, float* out, long ntok, int C) {
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tok >= ntok) return;
    const float* xp = x + tok * C;
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int c = (lane + 64 * k) * 4;
        v[k] = c < C ? *reinterpret_cast<const f32x4*>(xp + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int c = (lane + 64 * k) * 4;
        if (c < C) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[k][e] - mean;
                ss = fmaf(d, d, ss);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float rstd = 1.0f / sqrtf(ss / (float)C + 1e-5f);
    float* op = out + tok * C;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
        const int c = (lane + 64 * k) * 4;
        if (c < C) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), bb = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (v[k][e] - mean) * rstd * g[e] + bb[e];
            *reinterpret_cast<f32x4*>(op + c) = o4;
        }
    }
}
static hipError_t launch_layernorm(const float* x, const float* g, const float* b, float* out, long ntok, int C, hipStream_t s) {
    if (C % 4 || C > 512) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_layernorm<2>), dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, s, x, g, b, out, ntok, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// rotary embedding on the q and k parts of a head-major qkv buffer (apply_rot_emb, vit_modules.py:13-19):
// t <- t * cos + rotate_every_two(t) * sin, all d = 64 dims; position = token % period (time: frame index of the
// (site, frame)-ordered rows; space: site index).  tab [period][2][d]: sin row, cos row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rotary(float* qkv, const float* tab, long ntok, int H, int d, int period) {
    // one thread per (token, head, q|k, pair-of-pairs): 4 consecutive dims
    const int q4 = d / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = ntok * H * 2 * q4;
    if (idx >= total) return;
    const int c4 = (int)(idx % q4);
    const int j = (int)((idx / q4) % 2);
    const int h = (int)((idx / (2 * q4)) % H);
    const long tok = idx / ((long)2 * q4 * H);
    const int pos = (int)(tok % period);
    float* p = qkv + tok * (3L * H * d) + (long)h * 3 * d + j * d + c4 * 4;
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    const f32x4 sn = *reinterpret_cast<const f32x4*>(tab + ((size_t)pos * 2) * d + c4 * 4);
    const f32x4 cs = *reinterpret_cast<const f32x4*>(tab + ((size_t)pos * 2 + 1) * d + c4 * 4);
    f32x4 o;
    o[0] = t[0] * cs[0] + (-t[1]) * sn[0];
    o[1] = t[1] * cs[1] + t[0] * sn[1];
    o[2] = t[2] * cs[2] + (-t[3]) * sn[2];
    o[3] = t[3] * cs[3] + t[2] * sn[3];
    *reinterpret_cast<f32x4*>(p) = o;
}
static hipError_t launch_rotary(float* qkv, const float* tab, long ntok, int H, int d, int period, hipStream_t s) {
    const long total = ntok * H * 2 * (d / 4);
    hipLaunchKernelGGL(k_rotary, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qkv, tab, ntok, H, d, period);
    return hipGetLastError();
}

// GEGLU (vit_modules.py:88-91): out[t][c] = h[t][c] * gelu(h[t][half + c]), exact (erf) GELU like F.gelu's default;
// mode 1: plain GELU (autoencoder_vit.py:26-32 FeedForward of the quant stacks): out[t][c] = gelu(h[t][c])
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__global__ __launch_bounds__(256) void k_geglu(const float* h, float* out, long ntok, int half, int plain) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q4 = half / 4;
    if (idx >= ntok * q4) return;
    const long tok = idx / q4;
    const int c = (int)(idx - tok * q4) * 4;
    f32x4 o;
    if (plain) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(h + tok * half + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gelu_erf(a[e]);
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(h + tok * 2 * half + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(h + tok * 2 * half + half + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = a[e] * gelu_erf(g[e]);
    }
    *reinterpret_cast<f32x4*>(out + tok * half + c) = o;
}
static hipError_t launch_geglu(const float* h, float* out, long ntok, int half, int plain, hipStream_t s) {
    const long total = ntok * (half / 4);
    hipLaunchKernelGGL(k_geglu, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h, out, ntok, half, plain);
    return hipGetLastError();
}

// decode_from_sample's head (autoencoder_vit.py:258-270): z[b][(f,y,x)][c] = post_xy(h_xy)[c,y,x] + post_yt(h_yt)[c,f,x]
// + post_xt(h_xt)[c,f,y], each a 1x1 conv E -> C with bias over a latent plane of lat [B][E][r*r + 2*T*r]
struct ExpandArgs {
    const float* lat;
    const float *wxy, *bxy, *wyt, *byt, *wxt, *bxt;   // [C][E], [C]
    float* out;
    int B, E, C, r, T;
};
__global__ __launch_bounds__(256) void k_latent_expand(const ExpandArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = a.r * a.r, L = n + 2 * a.T * a.r;
    const long total = (long)a.B * a.T * n * a.C;
    if (idx >= total) return;
    const int c = (int)(idx % a.C);
    const long tokg = idx / a.C;
    const int b = (int)(tokg / ((long)a.T * n));
    const int tok = (int)(tokg - (long)b * a.T * n);
    const int f = tok / n, y = (tok - f * n) / a.r, x = tok % a.r;
    const float* lp = a.lat + (size_t)b * a.E * L;
    float vxy = a.bxy[c], vyt = a.byt[c], vxt = a.bxt[c];
    for (int e = 0; e < a.E; ++e) {
        vxy = fmaf(a.wxy[c * a.E + e], lp[(size_t)e * L + y * a.r + x], vxy);
        vyt = fmaf(a.wyt[c * a.E + e], lp[(size_t)e * L + n + f * a.r + x], vyt);
        vxt = fmaf(a.wxt[c * a.E + e], lp[(size_t)e * L + n + a.T * a.r + f * a.r + y], vxt);
    }
    a.out[idx] = (vxy + vyt) + vxt;
}

// to_pixel's scatter + output activation (autoencoder_vit.py:118-121,272-275): g [B*T*n][3*p*p] is the
// ConvTranspose2d(k = stride = p) expressed as a GEMM (+ bias); frame[(b t)][o][y*p + p1][x*p + p2] = 2*sigmoid(.) - 1
__global__ __launch_bounds__(256) void k_to_pixel(const float* g, float* out, long nframes, int r, int p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;       // over output pixels, x fastest
    const int R = r * p;
    const long total = nframes * 3 * R * R;
    if (idx >= total) return;
    const int X = (int)(idx % R), Y = (int)((idx / R) % R), o = (int)((idx / ((long)R * R)) % 3);
    const long fr = idx / ((long)3 * R * R);
    const int x = X / p, p2 = X % p, y = Y / p, p1 = Y % p;
    const float v = g[((fr * r + y) * r + x) * (3L * p * p) + (o * p + p1) * p + p2];
    out[idx] = 2.0f / (1.0f + expf(-v)) - 1.0f;
}

// TimeSformerEncoder's patch gather (vit_modules.py:212): video [B][3][T][H][W] ('b c t h w', the layout extract() receives)
// -> rows (b, f, y, x) of (p1 p2 c)
__global__ __launch_bounds__(256) void k_patchify(const float* vid, float* out, int B, int T, int r, int p) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int pd = p * p * 3, R = r * p;
    const long total = (long)B * T * r * r * pd;
    if (idx >= total) return;
    const int e = (int)(idx % pd);
    const long tok = idx / pd;
    const int c = e % 3, p2 = (e / 3) % p, p1 = e / (3 * p);
    const int x = (int)(tok % r), y = (int)((tok / r) % r), f = (int)((tok / ((long)r * r)) % T);
    const int b = (int)(tok / ((long)r * r * T));
    out[idx] = vid[((((size_t)b * 3 + c) * T + f) * R + (y * p + p1)) * R + (x * p + p2)];
}

// quant-stack input (autoencoder_vit.py:218-240): sequences over one axis of h [B][T][r][r][C] with the learned token
// appended and the position embedding added.  plane 0 (xy): sequence over t for every (y, x); 1 (yt): over y for every
// (t, x); 2 (xt): over x for every (t, y).  out [nseq][n + 1][C]
struct SeqArgs {
    const float* h;
    const float* token;   // [C]
    const float* pos;     // [n + 1][C]
    float* out;
    int B, T, r, C, plane;
};
__global__ __launch_bounds__(256) void k_plane_sequences(const SeqArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = a.plane == 0 ? a.T : a.r;
    const long nseq = a.plane == 0 ? (long)a.B * a.r * a.r : (long)a.B * a.T * a.r;
    const long total = nseq * (n + 1) * a.C;
    if (idx >= total) return;
    const int c = (int)(idx % a.C);
    const int i = (int)((idx / a.C) % (n + 1));
    const long sq = idx / ((long)a.C * (n + 1));
    float v;
    if (i == n) {
        v = a.token[c];
    } else {
        int b, t, y, x;
        if (a.plane == 0) { x = (int)(sq % a.r); y = (int)((sq / a.r) % a.r); b = (int)(sq / ((long)a.r * a.r)); t = i; }
        else if (a.plane == 1) { x = (int)(sq % a.r); t = (int)((sq / a.r) % a.T); b = (int)(sq / ((long)a.r * a.T)); y = i; }
        else { y = (int)(sq % a.r); t = (int)((sq / a.r) % a.T); b = (int)(sq / ((long)a.r * a.T)); x = i; }
        v = a.h[((((size_t)b * a.T + t) * a.r + y) * a.r + x) * a.C + c];
    }
    a.out[idx] = v + a.pos[(size_t)i * a.C + c];
}

// latent head of extract (autoencoder_vit.py:242-255): position 0 of every sequence -> pre_* 1x1 conv C -> E, tanh ->
// lat [B][E][L] slice of its plane (sequences are already in the plane's (row, col) order)
struct HeadArgs {
    const float* seq;     // [nseq][n + 1][C]
    const float *w, *b;   // [E][C], [E]
    float* lat;           // [B][E][L]
    int nseq_per_b, n1, C, E, L, off, B;
};
__global__ __launch_bounds__(64) void k_latent_head(const HeadArgs a) {
    // one wave per (sequence, e)
    const long sq = blockIdx.x;
    const int e = blockIdx.y, lane = threadIdx.x;
    const float* x = a.seq + (size_t)sq * a.n1 * a.C;      // position 0
    float acc = 0.f;
    for (int c = lane; c < a.C; c += 64) acc = fmaf(x[c], a.w[(size_t)e * a.C + c], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const long b = sq / a.nseq_per_b, i = sq - b * a.nseq_per_b;
        a.lat[((size_t)b * a.E + e) * a.L + a.off + i] = tanhf(acc + a.b[e]);
    }
}

}  // namespace mtv

// =============================================================================================
// plan builder
// =============================================================================================
namespace {

struct AeDims {
    int C, res, T, p, E, depth, H, d, r, n, ntok, L;
};

struct AeBuilder {
    mtv_ctx* c;
    Plan* plan;
    int B;
    AeDims D;
    float* zero_bias;

    void push(const std::string& name, std::function<hipError_t(hipStream_t)> fn, double flops = 0.0, double bytes = 0.0) {
        Op op;
        op.run = std::move(fn);
        op.name = name;
        op.flops = flops;
        op.bytes = bytes;
        plan->ops.push_back(std::move(op));
    }
    static int pad64(int n) { return (n + 63) / 64 * 64; }

    // out[b][tok][:N] = in[b][gather(tok)][:K] @ W + bias (+ res[b][tok][:N]); rows = tokens per batch element
    // geglu_h != nullptr: `in` is GEGLU(geglu_h) -- when the GEMM runs on the split-bf16 pair its elementwise pass computes that itself
    // (x3_fused_geglu) and the separate k_geglu launch that fills `in` is skipped
    std::shared_ptr<ConvOp> gemm(const std::string& name, const float* in, int K, const float* W, int ld, const float* bias, int N, float* out,
              int rows, const float* res = nullptr, const int* gather = nullptr, int batch = -1, const float* geglu_h = nullptr) {
        ConvArgs a{};
        a.ntaps = 1;
        a.B = batch < 0 ? B : batch;
        a.Lout = a.Lsrc = a.Lskip = rows;
        a.N = N;
        a.W = W;
        a.ldw = ld;
        a.bias = bias ? bias : zero_bias;
        a.out = out;
        a.nmain = 1;
        a.src[0] = in;
        a.C[0] = K;
        a.Cmain = K;
        a.seg_src = SegInfo{rows, rows, rows};
        a.seg_out = a.seg_src;
        a.res = res;
        a.gather = gather;
        a.stat_cstride = 0;
        if (x3_wanted((long)a.B * rows)) a.W3 = c->w3_for(W, K, ld, &a.w3_plane);      // split-bf16 weight copy for k_conv_x3
        auto op = std::make_shared<ConvOp>();
        op->a = a;
        op->t = conv_pick_tile(a.B, rows, N, K / 16, K, false);
        force_lds_tile(a, &op->t);
        op->base_name = "gemm:" + name;
        op->op_index = (int)plan->ops.size();
        plan->convs.push_back(op);
        const double flops = 2.0 * a.B * rows * (double)N * K;
        const double bytes = 4.0 * ((double)K * N + N) + 4.0 * a.B * ((double)rows * K + (double)rows * N * (res ? 2 : 1));
        char tag[64];
        snprintf(tag, sizeof tag, "[%dx%d k%d]", a.B * rows, N, K);
        if (geglu_h) {       // the tuner times each candidate as the plan would run it: fused pair, or k_geglu + the GEMM
            float* gl = const_cast<float*>(in);
            const long ntok_all = (long)a.B * rows;
            op->tune_launch = [geglu_h, gl, ntok_all, K](const ConvArgs& ta, ConvTile tt, hipStream_t s) -> hipError_t {
                ConvOp probe;
                probe.a = ta;
                probe.t = tt;
                if (x3_fused_geglu(probe)) return launch_conv_x3_geglu(ta, tt, geglu_h, s);
                hipError_t e = launch_geglu(geglu_h, gl, ntok_all, K, 0, s);
                return e != hipSuccess ? e : launch_conv(ta, tt, s);
            };
        }
        if (geglu_h)
            push(op->base_name + tag, [op, geglu_h](hipStream_t s) { return x3_fused_geglu(*op) ? launch_conv_x3_geglu(op->a, op->t, geglu_h, s) : launch_conv(op->a, op->t, s); }, flops, bytes);
        else
            push(op->base_name + tag, [op](hipStream_t s) { return launch_conv(op->a, op->t, s); }, flops, bytes);
        return op;
    }
    static bool x3_fused_geglu(const ConvOp& op) {
        static const bool off = getenv("MTV_AE_NO_GEGLU_FUSION") != nullptr;
        return !off && op.t.NW == 48 && !op.a.gather && conv_x3_eligible(op.a) && op.a.x3 != nullptr;
    }

    float* lin_w(const std::string& key, int N, int K, int* ld_out) {        // Linear [N][K] -> [K][ld]
        const int ld = pad64(N);
        float* W = c->buf("w." + key, (size_t)K * ld);
        c->slot(key, {N, K}, ROLE_CONV, W, ld);
        *ld_out = ld;
        return W;
    }
    float* vec(const std::string& key, int n) { return c->wcopy(key, {n}); }

    void layernorm(const std::string& nm, const std::string& key, const float* x, float* out, long ntok) {
        const float* g = vec(key + "norm.weight", D.C);
        const float* b = vec(key + "norm.bias", D.C);
        const int C = D.C;
        push("ln:" + nm, [=](hipStream_t s) { return launch_layernorm(x, g, b, out, ntok, C, s); }, 0.0, 8.0 * ntok * C);
    }

    // one TimeSformer attention (vit_modules.py:119-146) incl. PreNorm and the residual: x <- to_out(attn(LN(x))) + x
    void ts_attention(const std::string& nm, const std::string& key, float* x, bool time, const float* rot, const int* g_fwd, const int* g_bwd) {
        const int inner = D.H * D.d, C = D.C;
        const long ntok = (long)B * D.ntok;
        float* ln = c->buf("ae.ln", (size_t)c->cfg.max_batch * D.ntok * C);
        float* qkv = c->buf("ae.qkv", (size_t)c->cfg.max_batch * D.ntok * 3 * inner);
        float* att = c->buf("ae.att", (size_t)c->cfg.max_batch * D.ntok * inner);
        layernorm(nm, key, x, ln, ntok);
        const int ldq = pad64(3 * inner);
        float* Wq = c->buf("w." + key + "fn.to_qkv.weight", (size_t)C * ldq);
        c->slot(key + "fn.to_qkv.weight", {3 * inner, C}, ROLE_QKV_HEADS, Wq, ldq)->aux = D.d;
        gemm(nm + ".qkv", ln, C, Wq, ldq, nullptr, 3 * inner, qkv, D.ntok, nullptr, time ? g_fwd : nullptr);
        const int H = D.H, d = D.d, period = time ? D.T : D.n;
        push("rotary:" + nm, [=](hipStream_t s) { return launch_rotary(qkv, rot, ntok, H, d, period, s); }, 0.0, 16.0 * ntok * inner);
        AttnArgs t{};
        t.qkv = qkv; t.out = att; t.B = B; t.L = D.ntok; t.C = inner; t.H = H;
        t.scale = 1.0f / std::sqrt(std::sqrt((float)d));       // q * d^-1/2 (vit_modules.py:125) split evenly over q and k
        t.seg_uniform = period;
        const double aflops = 4.0 * B * H * (double)D.ntok * period * d;
        push("attn:" + nm, [t](hipStream_t s) { return launch_attention(t, s); }, aflops, 16.0 * ntok * inner);
        int ldo;
        float* Wo = lin_w(key + "fn.to_out.0.weight", C, inner, &ldo);
        const float* bo = vec(key + "fn.to_out.0.bias", C);
        gemm(nm + ".out", att, inner, Wo, ldo, bo, C, x, D.ntok, x, time ? g_bwd : nullptr);
    }

    // the 8-layer stack shared by TimeSformerEncoder / Decoder (vit_modules.py:225-234, 294-303), in place on x
    void timesformer(const std::string& pre, float* x, const float* rot_t, const float* rot_s, const int* g_fwd, const int* g_bwd) {
        const int C = D.C;
        const long ntok = (long)B * D.ntok;
        for (int i = 0; i < D.depth; ++i) {
            const std::string p = pre + "layers." + std::to_string(i) + ".";
            const std::string nm = pre + std::to_string(i);
            ts_attention(nm + ".time", p + "0.", x, true, rot_t, g_fwd, g_bwd);
            ts_attention(nm + ".space", p + "1.", x, false, rot_s, g_fwd, g_bwd);
            float* ln = c->buf("ae.ln", (size_t)c->cfg.max_batch * D.ntok * C);
            float* hid = c->buf("ae.ffh", (size_t)c->cfg.max_batch * D.ntok * 8 * C);
            float* gl = c->buf("ae.ffg", (size_t)c->cfg.max_batch * D.ntok * 4 * C);
            layernorm(nm + ".ff", p + "2.", x, ln, ntok);
            int ld1, ld2;
            float* W1 = lin_w(p + "2.fn.net.0.weight", 8 * C, C, &ld1);
            const float* b1 = vec(p + "2.fn.net.0.bias", 8 * C);
            gemm(nm + ".ff1", ln, C, W1, ld1, b1, 8 * C, hid, D.ntok);
            const int half = 4 * C;
            auto ff2 = std::make_shared<std::shared_ptr<ConvOp>>();      // (the GEGLU launch comes first in the plan, the GEMM it may fold into second)
            push("geglu:" + nm, [=](hipStream_t s) { return *ff2 && x3_fused_geglu(**ff2) ? hipSuccess : launch_geglu(hid, gl, ntok, half, 0, s); }, 0.0, 4.0 * ntok * 12 * C);
            float* W2 = lin_w(p + "2.fn.net.3.weight", C, 4 * C, &ld2);
            const float* b2 = vec(p + "2.fn.net.3.bias", C);
            *ff2 = gemm(nm + ".ff2", gl, 4 * C, W2, ld2, b2, C, x, D.ntok, x, nullptr, -1, hid);
        }
    }

    // autoencoder_vit.py:66-84 Transformer(dim, depth 4, heads 4, dim_head dim/8, mlp 512) on nseq sequences of n1 tokens
    void quant_stack(const std::string& pre, float* x, long nseq, int n1) {
        const int C = D.C, H = 4, d = C / 8, inner = H * d, mlp = 512;
        const long ntok = nseq * n1;
        float* ln = c->buf("ae.q.ln", (size_t)c->cfg.max_batch * qtok_max * C);
        float* qkv = c->buf("ae.q.qkv", (size_t)c->cfg.max_batch * qtok_max * 3 * inner);
        float* att = c->buf("ae.q.att", (size_t)c->cfg.max_batch * qtok_max * inner);
        float* hid = c->buf("ae.q.h", (size_t)c->cfg.max_batch * qtok_max * mlp);
        float* act = c->buf("ae.q.a", (size_t)c->cfg.max_batch * qtok_max * mlp);
        for (int i = 0; i < 4; ++i) {
            const std::string p = pre + "layers." + std::to_string(i) + ".";
            const std::string nm = pre + std::to_string(i);
            layernorm(nm + ".attn", p + "0.", x, ln, ntok);
            const int ldq = pad64(3 * inner);
            float* Wq = c->buf("w." + p + "0.fn.to_qkv.weight", (size_t)C * ldq);
            c->slot(p + "0.fn.to_qkv.weight", {3 * inner, C}, ROLE_QKV_HEADS, Wq, ldq)->aux = d;
            gemm(nm + ".qkv", ln, C, Wq, ldq, nullptr, 3 * inner, qkv, (int)ntok, nullptr, nullptr, 1);
            AttnArgs t{};
            t.qkv = qkv; t.out = att; t.B = 1; t.L = (int)ntok; t.C = inner; t.H = H;
            t.scale = 1.0f / std::sqrt(std::sqrt((float)d));   // dots * d^-1/2 (autoencoder_vit.py:55)
            t.seg_uniform = n1;
            push("attn:" + nm, [t](hipStream_t s) { return launch_attention(t, s); }, 4.0 * H * (double)ntok * n1 * d, 16.0 * ntok * inner);
            int ldo, ld1, ld2;
            float* Wo = lin_w(p + "0.fn.to_out.0.weight", C, inner, &ldo);
            const float* bo = vec(p + "0.fn.to_out.0.bias", C);
            gemm(nm + ".out", att, inner, Wo, ldo, bo, C, x, (int)ntok, x, nullptr, 1);
            layernorm(nm + ".ff", p + "1.", x, ln, ntok);
            float* W1 = lin_w(p + "1.fn.net.0.weight", mlp, C, &ld1);
            const float* b1 = vec(p + "1.fn.net.0.bias", mlp);
            gemm(nm + ".ff1", ln, C, W1, ld1, b1, mlp, hid, (int)ntok, nullptr, nullptr, 1);
            push("gelu:" + nm, [=](hipStream_t s) { return launch_geglu(hid, act, ntok, mlp, 1, s); }, 0.0, 8.0 * ntok * mlp);
            float* W2 = lin_w(p + "1.fn.net.3.weight", C, mlp, &ld2);
            const float* b2 = vec(p + "1.fn.net.3.bias", C);
            gemm(nm + ".ff2", act, mlp, W2, ld2, b2, C, x, (int)ntok, x, nullptr, 1);
        }
    }
    long qtok_max = 0;

    int build(int mode) {
        const int C = D.C;
        zero_bias = c->buf("ae.zero_bias", 8192);
        const float* rot_t = c->bufs["ae.rot_time"];
        const float* rot_s = c->bufs["ae.rot_space"];
        const int* g_fwd = reinterpret_cast<const int*>(c->bufs["ae.g_fwd"]);
        const int* g_bwd = reinterpret_cast<const int*>(c->bufs["ae.g_bwd"]);
        float* x = c->buf("ae.x", (size_t)c->cfg.max_batch * D.ntok * C);
        const long ntok = (long)B * D.ntok;
        qtok_max = (long)D.n * (D.T + 1) > (long)D.T * D.r * (D.r + 1) ? (long)D.n * (D.T + 1) : (long)D.T * D.r * (D.r + 1);
        if (mode == MODE_AE_DECODE) {
            ExpandArgs e{};
            e.lat = c->bufs["ae.lat_in"];
            e.wxy = c->wcopy("post_xy.weight", {C, D.E, 1, 1}); e.bxy = vec("post_xy.bias", C);
            e.wyt = c->wcopy("post_yt.weight", {C, D.E, 1, 1}); e.byt = vec("post_yt.bias", C);
            e.wxt = c->wcopy("post_xt.weight", {C, D.E, 1, 1}); e.bxt = vec("post_xt.bias", C);
            e.out = x; e.B = B; e.E = D.E; e.C = C; e.r = D.r; e.T = D.T;
            const long total = ntok * C;
            push("latent_expand", [e, total](hipStream_t s) {
                hipLaunchKernelGGL(k_latent_expand, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, e);
                return hipGetLastError();
            });
            timesformer("decoder.", x, rot_t, rot_s, g_fwd, g_bwd);
            // to_pixel: ConvTranspose2d(C -> 3, k = stride = p): its weight [C][3][p][p] IS the [K][N] GEMM operand
            const int N = 3 * D.p * D.p, ld = pad64(N);
            if (ld != N) return fail(MTV_ERR_INVALID, "patch size must make 3*p*p a multiple of 64");
            float* Wp = c->buf("w.to_pixel.1.weight", (size_t)C * ld);
            c->slot("to_pixel.1.weight", {C, 3, D.p, D.p}, ROLE_COPY, Wp, 0);
            float* bp = c->buf("w.to_pixel.1.bias.expanded", (size_t)N);
            c->slot("to_pixel.1.bias", {3}, ROLE_REPEAT, bp, 0)->aux = D.p * D.p;
            float* g = c->buf("ae.pix", (size_t)c->cfg.max_batch * D.ntok * N);
            gemm("to_pixel", x, C, Wp, ld, bp, N, g, D.ntok);
            float* out = c->bufs["ae.frames_out"];
            const long nfr = (long)B * D.T;
            const int r = D.r, p = D.p;
            const long tot = nfr * 3 * D.res * D.res;
            push("pixel_scatter", [=](hipStream_t s) {
                hipLaunchKernelGGL(k_to_pixel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, g, out, nfr, r, p);
                return hipGetLastError();
            });
        } else {
            const int pd = 3 * D.p * D.p;
            float* patches = c->buf("ae.patches", (size_t)c->cfg.max_batch * D.ntok * pd);
            const float* vid = c->bufs["ae.video_in"];
            const int Bn = B, T = D.T, r = D.r, p = D.p;
            const long tot = ntok * pd;
            push("patchify", [=](hipStream_t s) {
                hipLaunchKernelGGL(k_patchify, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, vid, patches, Bn, T, r, p);
                return hipGetLastError();
            });
            int ldp;
            float* Wpe = lin_w("encoder.to_patch_embedding.weight", C, pd, &ldp);
            const float* bpe = vec("encoder.to_patch_embedding.bias", C);
            gemm("patch_embed", patches, pd, Wpe, ldp, bpe, C, x, D.ntok);
            timesformer("encoder.", x, rot_t, rot_s, g_fwd, g_bwd);
            float* lat = c->bufs["ae.lat_out"];
            static const char* PL[3] = {"xy", "yt", "xt"};
            for (int pl = 0; pl < 3; ++pl) {
                const std::string P = PL[pl];
                const int n = pl == 0 ? D.T : D.r;
                const long nseq = pl == 0 ? (long)B * D.n : (long)B * D.T * D.r;
                float* seq = c->buf("ae.q.x", (size_t)c->cfg.max_batch * qtok_max * C);
                SeqArgs sa{};
                sa.h = x; sa.token = c->wcopy(P + "_token", {1, 1, C});
                sa.pos = c->wcopy(P + "_pos_embedding", {1, (pl == 0 ? D.T : D.r) + 1, C});
                sa.out = seq; sa.B = B; sa.T = D.T; sa.r = D.r; sa.C = C; sa.plane = pl;
                const long tq = nseq * (n + 1) * C;
                push("sequences:" + P, [sa, tq](hipStream_t s) {
                    hipLaunchKernelGGL(k_plane_sequences, dim3((unsigned)((tq + 255) / 256)), dim3(256), 0, s, sa);
                    return hipGetLastError();
                });
                quant_stack(P + "_quant_attn.", seq, nseq, n + 1);
                HeadArgs ha{};
                ha.seq = seq; ha.w = c->wcopy("pre_" + P + ".weight", {D.E, C, 1, 1}); ha.b = vec("pre_" + P + ".bias", D.E);
                ha.lat = lat; ha.nseq_per_b = (int)(nseq / B); ha.n1 = n + 1; ha.C = C; ha.E = D.E; ha.L = D.L; ha.B = B;
                ha.off = pl == 0 ? 0 : (pl == 1 ? D.n : D.n + D.T * D.r);
                const int E = D.E;
                push("latent_head:" + P, [ha, nseq, E](hipStream_t s) {
                    hipLaunchKernelGGL(k_latent_head, dim3((unsigned)nseq, E), dim3(64), 0, s, ha);
                    return hipGetLastError();
                });
            }
        }
        return finish_split_k(c, plan);
    }
};

AeDims dims_of(const mtv_ae_config& f) {
    AeDims D{};
    D.C = f.channels; D.res = f.resolution; D.T = f.frames; D.p = f.patch; D.E = f.embed_dim; D.depth = f.depth;
    D.H = f.heads; D.d = f.dim_head; D.r = f.resolution / f.patch; D.n = D.r * D.r; D.ntok = D.T * D.n;
    D.L = D.n + 2 * D.T * D.r;
    return D;
}

int get_ae_plan(mtv_ctx* c, const mtv_ae_config& f, int B, int mode, Plan** out) {
    auto it = c->plans.find({B, mode});
    if (it != c->plans.end()) {
        *out = it->second.get();
        return MTV_OK;
    }
    std::unique_ptr<Plan> p(new Plan());
    p->B = B;
    p->mode = mode;
    AeBuilder b{c, p.get(), B, dims_of(f), nullptr};
    int rc = b.build(mode);
    if (rc != MTV_OK) return rc;
    *out = p.get();
    c->plans[{B, mode}] = std::move(p);
    return MTV_OK;
}

}  // namespace

// the AE configuration of a context: owned by the context itself (mtv_ctx::ext), nullptr for any other kind of context
static const mtv_ae_config* ae_cfg_of(const mtv_ctx* c) {
    return c && c->kind == CTX_AE ? static_cast<const mtv_ae_config*>(c->ext.get()) : nullptr;
}

extern "C" {

int mtv_ae_create(const mtv_ae_config* cfg, mtv_ctx** out) {
    if (!cfg || !out) return fail(MTV_ERR_INVALID, "null argument");
    const mtv_ae_config& f = *cfg;
    if (f.channels % 64 || f.channels > 512 || f.heads * f.dim_head % 64) return fail(MTV_ERR_INVALID, "channels must be a multiple of 64, <= 512");
    if (f.dim_head != 64 || (f.channels / 8) != 48) return fail(MTV_ERR_INVALID, "built for dim_head 64 and channels 384 (quant head dim 48)");
    if (f.resolution % f.patch || f.max_batch < 1 || f.frames < 1 || f.depth < 1) return fail(MTV_ERR_INVALID, "bad geometry");
    if ((3 * f.patch * f.patch) % 64) return fail(MTV_ERR_INVALID, "3*patch*patch must be a multiple of 64");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(MTV_ERR_HIP, "no HIP device available");
    std::unique_ptr<mtv_ctx> c(new mtv_ctx());
    c->cfg.max_batch = f.max_batch;
    int rc = ctx_init_common(c.get());
    if (rc != MTV_OK) return rc;
    const AeDims D = dims_of(f);
    // external staging
    if (!c->buf("ae.lat_in", (size_t)f.max_batch * D.E * D.L) || !c->buf("ae.lat_out", (size_t)f.max_batch * D.E * D.L) ||
        !c->buf("ae.frames_out", (size_t)f.max_batch * D.T * 3 * D.res * D.res) ||
        !c->buf("ae.video_in", (size_t)f.max_batch * 3 * D.T * D.res * D.res))
        return fail(MTV_ERR_HIP, "staging allocation failed");
    // token re-ordering tables of the time attention: forward (n, f) <- (f, n), backward (f, n) <- (n, f)
    {
        std::vector<int> fw(D.ntok), bw(D.ntok);
        for (int fr = 0; fr < D.T; ++fr)
            for (int n = 0; n < D.n; ++n) {
                fw[n * D.T + fr] = fr * D.n + n;
                bw[fr * D.n + n] = n * D.T + fr;
            }
        float* gf = c->buf("ae.g_fwd", D.ntok);
        float* gb = c->buf("ae.g_bwd", D.ntok);
        if (!gf || !gb) return fail(MTV_ERR_HIP, "gather table allocation failed");
        HIPCHK(hipMemcpy(gf, fw.data(), (size_t)D.ntok * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(gb, bw.data(), (size_t)D.ntok * 4, hipMemcpyHostToDevice));
    }
    if (!c->buf("ae.rot_time", (size_t)D.T * 2 * D.d) || !c->buf("ae.rot_space", (size_t)D.n * 2 * D.d))
        return fail(MTV_ERR_HIP, "rotary table allocation failed");
    c->kind = CTX_AE;
    c->ext = std::make_shared<mtv_ae_config>(f);
    // build both batch-1 plans now: registers every weight slot
    Plan* p = nullptr;
    if ((rc = get_ae_plan(c.get(), f, 1, MODE_AE_DECODE, &p)) != MTV_OK) return rc;
    if ((rc = get_ae_plan(c.get(), f, 1, MODE_AE_EXTRACT, &p)) != MTV_OK) return rc;
    *out = c.release();
    return MTV_OK;
}

int mtv_ae_destroy(mtv_ctx* c) {
    if (c && c->kind != CTX_AE) return fail(MTV_ERR_INVALID, "not an autoencoder context");
    delete c;
    return MTV_OK;
}

int mtv_ae_set_rotary(mtv_ctx* c, const float* time_tab, const float* space_tab) {
    const mtv_ae_config* cf = ae_cfg_of(c);
    if (!cf || !time_tab || !space_tab) return fail(MTV_ERR_INVALID, "not an autoencoder context / null table");
    const AeDims D = dims_of(*cf);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(c->bufs["ae.rot_time"], time_tab, (size_t)D.T * 2 * D.d * 4, hipMemcpyDefault));
    HIPCHK(hipMemcpy(c->bufs["ae.rot_space"], space_tab, (size_t)D.n * 2 * D.d * 4, hipMemcpyDefault));
    return MTV_OK;
}

static int ae_run(mtv_ctx* c, int mode, const float* in, size_t in_floats, const char* in_buf, float* outp, size_t out_floats,
                  const char* out_buf, int batch, hipStream_t s) {
    const mtv_ae_config* cf = ae_cfg_of(c);
    if (!cf) return fail(MTV_ERR_INVALID, "not an autoencoder context");
    int rc = check_ready(c, batch);
    if (rc != MTV_OK) return rc;
    if (!in || !outp) return fail(MTV_ERR_INVALID, "null tensor pointer");
    HIPCHK(hipSetDevice(c->device));
    Plan* p = nullptr;
    if ((rc = get_ae_plan(c, *cf, batch, mode, &p)) != MTV_OK) return rc;
    if ((rc = autotune(c, p, s)) != MTV_OK) return rc;
    HIPCHK(hipMemcpyAsync(c->bufs[in_buf], in, in_floats * 4, hipMemcpyDeviceToDevice, s));
    if (c->eager) {
        if ((rc = run_ops(c, p, s)) != MTV_OK) return rc;
    } else {
        if (!p->g_forward && (rc = capture(c, p, &p->g_forward)) != MTV_OK) return rc;
        HIPCHK(hipGraphLaunch(p->g_forward, s));
    }
    HIPCHK(hipMemcpyAsync(outp, c->bufs[out_buf], out_floats * 4, hipMemcpyDeviceToDevice, s));
    return MTV_OK;
}

int mtv_ae_decode(mtv_ctx* c, const float* latents, float* frames_out, int batch, void* stream) {
    const mtv_ae_config* cf = ae_cfg_of(c);
    if (!cf) return fail(MTV_ERR_INVALID, "not an autoencoder context");
    const AeDims D = dims_of(*cf);
    return ae_run(c, MODE_AE_DECODE, latents, (size_t)batch * D.E * D.L, "ae.lat_in", frames_out,
                  (size_t)batch * D.T * 3 * D.res * D.res, "ae.frames_out", batch, (hipStream_t)stream);
}

int mtv_ae_extract(mtv_ctx* c, const float* video, float* latents_out, int batch, void* stream) {
    const mtv_ae_config* cf = ae_cfg_of(c);
    if (!cf) return fail(MTV_ERR_INVALID, "not an autoencoder context");
    const AeDims D = dims_of(*cf);
    return ae_run(c, MODE_AE_EXTRACT, video, (size_t)batch * 3 * D.T * D.res * D.res, "ae.video_in", latents_out,
                  (size_t)batch * D.E * D.L, "ae.lat_out", batch, (hipStream_t)stream);
}

int mtv_ae_profile(mtv_ctx* c, int batch, int extract, int iters, mtv_op_time* out, int cap, int* n_out, void* stream) {
    const mtv_ae_config* cf = ae_cfg_of(c);
    if (!cf || !n_out || iters < 1) return fail(MTV_ERR_INVALID, "not an autoencoder context / bad argument");
    int rc = check_ready(c, batch);
    if (rc != MTV_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    Plan* p = nullptr;
    if ((rc = get_ae_plan(c, *cf, batch, extract ? MODE_AE_EXTRACT : MODE_AE_DECODE, &p)) != MTV_OK) return rc;
    if ((rc = autotune(c, p, s)) != MTV_OK) return rc;
    const int n = (int)p->ops.size();
    *n_out = n;
    if (!out) return MTV_OK;
    if (cap < n) return fail(MTV_ERR_INVALID, "profile table too small");
    std::vector<hipEvent_t> ev((size_t)n + 1);
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    for (int itr = 0; itr < iters; ++itr) {
        HIPCHK(hipEventRecord(ev[0], s));
        for (int i = 0; i < n; ++i) {
            hipError_t e = p->ops[i].run(s);
            if (e != hipSuccess) return fail(MTV_ERR_HIP, "launch " + p->ops[i].name + ": " + hipGetErrorString(e));
            HIPCHK(hipEventRecord(ev[i + 1], s));
        }
        HIPCHK(hipStreamSynchronize(s));
        for (int i = 0; i < n; ++i) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            acc[i] += ms;
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    for (int i = 0; i < n; ++i) {
        std::memset(&out[i], 0, sizeof(mtv_op_time));
        std::strncpy(out[i].name, p->ops[i].name.c_str(), sizeof(out[i].name) - 1);
        out[i].ms = (float)(acc[i] / iters);
        out[i].flops = p->ops[i].flops;
        out[i].bytes = p->ops[i].bytes;
    }
    return MTV_OK;
}

}  // extern "C"
