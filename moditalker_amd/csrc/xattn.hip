// CrossAttention (MToV/models/ddpm/unet.py:429-467) as a stand-alone HIP operator.
//
// In the reference this class is the only usable piece of the landmark cross-attention conditioning path:
// SpatialTransformer (unet.py:492-528) instantiates BasicTransformerBlock, which is commented out (unet.py:470-489), so
// UNetModel(use_spatial_transformer=True) cannot be constructed there either and no shipped configuration asks for it.
// The operator itself is built and pinned so that a checkpoint that does carry such blocks has its arithmetic ready:
//   q = x to_q^T, k = ctx to_k^T, v = ctx to_v^T (no biases); per head softmax(q k^T * d^-1/2 [masked]) v; to_out (+ bias)
// GEMMs are k_conv launches (ntaps = 1), the core is k_attention in its cross mode (separate key/value buffer and length,
// optional key mask).  context == NULL is self-attention exactly as the reference's default(context, x).
#include "plan_internal.h"

namespace mtv {
// Linear [H*d][C] (rows (head, dd)) -> columns (head, part of nparts, dd) of a [C][ld] matrix shared by several Linears
__global__ void k_repack_heads(const float* src, float* dst, int H, int d, int C, int ld, int nparts, int part) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)H * d * C) return;
    const int n = (int)(idx / C), c = (int)(idx - (long)n * C);
    const int h = n / d, dd = n - h * d;
    dst[(size_t)c * ld + (h * nparts + part) * d + dd] = src[idx];
}
hipError_t launch_repack_heads(const float* src, float* dst, int H, int d, int C, int ld, int nparts, int part, hipStream_t s) {
    const long n = (long)H * d * C;
    hipLaunchKernelGGL(k_repack_heads, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, H, d, C, ld, nparts, part);
    return hipGetLastError();
}
}  // namespace mtv

namespace {
struct XCfg {
    mtv_xattn_config f;
    float *Wq, *Wkv, *Wo, *bo, *zero, *q, *kv, *att;
    int ldq, ldkv, ldo;
};
const XCfg* xcfg_of(const mtv_ctx* c) {      // owned by the context (mtv_ctx::ext)
    return c && c->kind == CTX_XATTN ? static_cast<const XCfg*>(c->ext.get()) : nullptr;
}
int pad64(int n) { return (n + 63) / 64 * 64; }

hipError_t gemm(const float* in, int K, const float* W, int ld, const float* bias, int N, float* out, int B, int rows, hipStream_t s) {
    ConvArgs a{};
    a.ntaps = 1;
    a.B = B;
    a.Lout = a.Lsrc = a.Lskip = rows;
    a.N = N;
    a.W = W;
    a.ldw = ld;
    a.bias = bias;
    a.out = out;
    a.nmain = 1;
    a.src[0] = in;
    a.C[0] = K;
    a.Cmain = K;
    a.seg_src = SegInfo{rows, rows, rows};
    a.seg_out = a.seg_src;
    ConvTile t = conv_pick_tile(B, rows, N, K / 16, K, false);
    t.KS = 1;                                  // (no slab: this operator owns no split-K workspace)
    while (t.NW > K / 16) t.NW /= 2;
    return launch_conv(a, t, s);
}
}  // namespace

extern "C" {

int mtv_xattn_create(const mtv_xattn_config* cfg, mtv_ctx** out) {
    if (!cfg || !out) return fail(MTV_ERR_INVALID, "null argument");
    const mtv_xattn_config& f = *cfg;
    const int d = f.dim_head, inner = f.heads * f.dim_head;
    if (!(d == 16 || d == 32 || d == 48 || d == 64 || d == 128)) return fail(MTV_ERR_INVALID, "dim_head must be 16, 32, 48, 64 or 128");
    if (f.query_dim % 16 || f.context_dim % 16 || inner % 16) return fail(MTV_ERR_INVALID, "query_dim, context_dim, heads*dim_head must be multiples of 16");
    if (f.max_batch < 1 || f.max_queries < 1 || f.max_keys < 1) return fail(MTV_ERR_INVALID, "bad sizes");
    std::unique_ptr<mtv_ctx> c(new mtv_ctx());
    c->cfg.max_batch = f.max_batch;
    int rc = ctx_init_common(c.get());
    if (rc != MTV_OK) return rc;
    XCfg x{};
    x.f = f;
    x.ldq = pad64(inner);
    x.ldkv = pad64(2 * inner);
    x.ldo = pad64(f.query_dim);
    x.Wq = c->buf("w.to_q.weight", (size_t)f.query_dim * x.ldq);
    x.Wkv = c->buf("w.to_kv", (size_t)f.context_dim * x.ldkv);
    x.Wo = c->buf("w.to_out.0.weight", (size_t)inner * x.ldo);
    x.zero = c->buf("x.zero", (size_t)x.ldkv + x.ldq);
    x.q = c->buf("x.q", (size_t)f.max_batch * f.max_queries * inner);
    x.kv = c->buf("x.kv", (size_t)f.max_batch * f.max_keys * 2 * inner);
    x.att = c->buf("x.att", (size_t)f.max_batch * f.max_queries * inner);
    if (!x.Wq || !x.Wkv || !x.Wo || !x.zero || !x.q || !x.kv || !x.att) return fail(MTV_ERR_HIP, "allocation failed");
    c->slot("to_q.weight", {inner, f.query_dim}, ROLE_CONV, x.Wq, x.ldq);
    c->slot("to_k.weight", {inner, f.context_dim}, ROLE_KV_HEADS, x.Wkv, x.ldkv)->aux = d * 2 + 0;
    c->slot("to_v.weight", {inner, f.context_dim}, ROLE_KV_HEADS, x.Wkv, x.ldkv)->aux = d * 2 + 1;
    c->slot("to_out.0.weight", {f.query_dim, inner}, ROLE_CONV, x.Wo, x.ldo);
    x.bo = c->wcopy("to_out.0.bias", {f.query_dim});
    c->kind = CTX_XATTN;
    c->ext = std::make_shared<XCfg>(x);
    *out = c.release();
    return MTV_OK;
}

int mtv_xattn_destroy(mtv_ctx* c) {
    if (c && c->kind != CTX_XATTN) return fail(MTV_ERR_INVALID, "not a cross-attention context");
    delete c;
    return MTV_OK;
}

int mtv_xattn_forward(mtv_ctx* c, const float* x, const float* context, const unsigned char* mask, float* out, int batch,
                      int n_queries, int n_keys, void* stream) {
    const XCfg* xp = xcfg_of(c);
    if (!xp) return fail(MTV_ERR_INVALID, "not a cross-attention context");
    const XCfg& X = *xp;
    int rc = check_ready(c, batch);
    if (rc != MTV_OK) return rc;
    if (!x || !out) return fail(MTV_ERR_INVALID, "null tensor pointer");
    const int inner = X.f.heads * X.f.dim_head;
    if (!context) {                       // default(context, x): self-attention over the queries (needs context_dim == query_dim)
        if (X.f.context_dim != X.f.query_dim) return fail(MTV_ERR_INVALID, "context is NULL but context_dim != query_dim");
        context = x;
        n_keys = n_queries;
    }
    if (n_queries < 1 || n_queries > X.f.max_queries || n_keys < 1 || n_keys > X.f.max_keys) return fail(MTV_ERR_STATE, "token count outside the context's capacity");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(gemm(x, X.f.query_dim, X.Wq, X.ldq, X.zero, inner, X.q, batch, n_queries, s));
    HIPCHK(gemm(context, X.f.context_dim, X.Wkv, X.ldkv, X.zero, 2 * inner, X.kv, batch, n_keys, s));
    AttnArgs t{};
    t.qkv = X.q;
    t.kv = X.kv;
    t.Lkv = n_keys;
    t.kmask = mask;
    t.out = X.att;
    t.B = batch;
    t.L = n_queries;
    t.C = inner;
    t.H = X.f.heads;
    t.nseg = 1;
    t.seg_start[0] = 0;
    t.seg_len[0] = n_queries;
    t.scale = 1.0f / std::sqrt(std::sqrt((float)X.f.dim_head));      // sim * d^-1/2 (unet.py:449), split evenly over q and k
    HIPCHK(launch_attention(t, s));
    HIPCHK(gemm(X.att, inner, X.Wo, X.ldo, X.bo, X.f.query_dim, out, batch, n_queries, s));
    return MTV_OK;
}

}  // extern "C"
