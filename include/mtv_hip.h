/*
 * mtv_hip.h -- C ABI of libmtv_hip.so: the MI355X (gfx950) denoising step of MoDiTalker's MToV
 * latent-video diffusion sampler (tri-plane UNet forward + eta=1 DDIM update) as hand-written HIP.
 *
 * The reference has NO FFI / plugin interface for this path: its boundary is three Python classes
 * (SURVEY.md section 8b).  This header is therefore the build's own C boundary *under* the Python
 * look-alikes in moditalker_amd/{unet,ddpm}.py; each entry point cites the reference code whose
 * work it replaces.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C, no torch types: device pointers are raw `float*` / `int64_t*` (Tensor.data_ptr()),
 *     BORROWED for the duration of the call (the caller keeps ownership and keeps them alive until
 *     the stream has drained); sizes are ints.
 *   - every function returns MTV_OK (0) or a negative error code; mtv_last_error() gives the text
 *     of the last failure on the calling thread.  No C++ exception crosses the ABI.
 *   - a context is bound to the HIP device that was current at mtv_create() and is NOT re-entrant:
 *     one caller at a time, launches go to the `stream` argument (pass torch's current stream so
 *     torch ordering holds).  `stream` is a hipStream_t passed as void*.
 *   - all tensors fp32.  External activations use the reference's layout, channel-major
 *     [B, C, L] with L = R*R + 2*T*R (xy plane | yt plane | xt plane, unet.py:1027-1029).
 */
#ifndef MTV_HIP_H
#define MTV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTV_OK 0
#define MTV_ERR_INVALID (-1)     /* bad argument / unsupported configuration */
#define MTV_ERR_HIP (-2)         /* a HIP runtime call failed                 */
#define MTV_ERR_WEIGHT (-3)      /* unknown key, wrong shape, or weights missing at forward time */
#define MTV_ERR_STATE (-4)       /* call sequence error (e.g. batch > max_batch) */
#define MTV_IGNORED 1            /* mtv_load_weight: key accepted but unused (output_bg_*) */

#define MTV_MAX_LEVELS 8

/* UNet hyper-parameters: the subset of UNetModel.__init__ kwargs (unet.py:631-659) the hot path
 * depends on, plus the tri-plane geometry the reference hard-wires to (32,16) (unet.py:1027-1029). */
typedef struct mtv_config {
    int32_t model_channels;                       /* unet_config.model_channels (128)            */
    int32_t num_res_blocks;                       /* (2)                                         */
    int32_t num_heads;                            /* (8) heads in every attention block          */
    int32_t n_levels;                             /* len(channel_mult)                           */
    int32_t channel_mult[MTV_MAX_LEVELS];         /* ([1,2,4,4])                                 */
    int32_t n_attention_resolutions;
    int32_t attention_resolutions[MTV_MAX_LEVELS];/* ([4,2,1]) downsample rates with 2-D attention */
    int32_t use_scale_shift_norm;                 /* (1) FiLM: GN(h)*(1+s)+b ; 0: h+emb then GN  */
    int32_t out_channels;                         /* (4)                                         */
    int32_t res;                                  /* R: xy plane is R x R                        */
    int32_t frames;                               /* T: yt / xt planes are T x R                 */
    int32_t max_batch;                            /* workspace is sized for this many clips      */
} mtv_config;

typedef struct mtv_ctx mtv_ctx;

const char* mtv_last_error(void);
int mtv_version(void);

/* Replaces UNetModel.__init__ (unet.py:631-975): builds the block list, allocates the
 * token-major channels-last workspace [B, L, C] per tensor and the im2col gather tables. */
int mtv_create(const mtv_config* cfg, mtv_ctx** out);
int mtv_destroy(mtv_ctx* ctx);

/* Checkpoint interface -- replaces `load_state_dict` (sample.py:229-230).
 * Keys are the reference's state_dict names WITHOUT the `diffusion_model.` prefix, e.g.
 * "input_blocks.4.0.in_layers.2.weight" [256,128,3,3].  `data` may be a host or a device pointer
 * (fp32, contiguous, PyTorch layout: OIHW conv, [O,I,1] conv1d, [O,I] linear); the library repacks
 * into its device layout ([tap][Cin][Cout] etc.).  Keys under output_bg_blocks./output_bg_attns.
 * (dead in the reference forward, unet.py:859-968) are accepted and return MTV_IGNORED. */
int mtv_num_weights(const mtv_ctx* ctx);
int mtv_weight_info(const mtv_ctx* ctx, int index, char* key_out, int key_cap, int* ndim_out, int64_t shape_out[4]);
int mtv_load_weight(mtv_ctx* ctx, const char* key, const float* data, int ndim, const int64_t* shape);
int mtv_weights_missing(const mtv_ctx* ctx);   /* number of required keys not loaded yet */

/* Replaces UNetModel.forward / DiffusionWrapper.forward (unet.py:995-1117, 41-44):
 *   x [B,4,L], cond [B,8,L], image_cond [B,4,image_cond_len] (only the first R*R tokens are
 *   used, the yt/xt planes of image_cond are zeros: unet.py:1022-1025), timesteps [B] int64 on the
 *   device, eps_out [B,out_channels,L].  All device pointers. */
int mtv_forward(mtv_ctx* ctx, const float* x, const float* cond, const float* image_cond,
                int image_cond_len, const int64_t* timesteps, float* eps_out, int batch, void* stream);

/* One entry per DDIM step, computed by the host exactly as ddpm.py:390-394 does (fp32). */
typedef struct mtv_ddim_step {
    int32_t t;                 /* timestep fed to the UNet (ddpm.py:383)                          */
    int32_t last;              /* 1: time_next < 0 -> result is clamped x0 (ddpm.py:386-388)      */
    float sqrt_recip_ac;       /* sqrt(1/alphas_cumprod[t])        (ddpm.py:278-282)              */
    float sqrt_recipm1_ac;     /* sqrt(1/alphas_cumprod[t] - 1)                                   */
    float sqrt_ac_next;        /* alphas_cumprod[t_next].sqrt()    (ddpm.py:398)                  */
    float c;                   /* (1 - alpha_next - sigma^2).sqrt()                               */
    float sigma;               /* eta * ((1-a/a_next)(1-a_next)/(1-a)).sqrt()                     */
    int32_t noise_index;       /* which [B,4,L] slab of `noise` this step adds (-1: none)         */
} mtv_ddim_step;

/* Replaces the loop body of DDPM.ddim_sample / ddim_sample_noised_start (ddpm.py:382-398,
 * 434-448) for `n_steps` consecutive steps: eps = UNet(x_t, cond, image_cond, t);
 * x0 = clamp(sqrt_recip*x_t - sqrt_recipm1*eps, -1, 1); x <- x0*sqrt_ac_next + c*eps + sigma*noise.
 *   x_io      [B,4,L]  in: x_T (or the q_sample'd start); out: the clamped x0 of the last step
 *   noise     [n_noise,B,4,L] explicit N(0,1) draws (the reference draws them from torch's global
 *             generator inside the loop; the Python wrapper owns that RNG)
 * A step is ONE hipGraph replay and consists of the UNet's launches only: the FiLM rows of all steps are computed
 * once per call, the DDIM update and the hand-over to the next step (step counter, packed input, FiLM row,
 * statistics arena) run inside the head conv; steps are device-resident (no host synchronisation in the loop). */
int mtv_ddim_sample(mtv_ctx* ctx, float* x_io, const float* cond, const float* image_cond,
                    int image_cond_len, const float* noise, int n_noise,
                    const mtv_ddim_step* steps, int n_steps, int batch, void* stream);

/* Fault state of the in-launch hand-offs.  Some launches of a step exchange data between their own workgroups (k_deep_block's clusters, the
 * tagged completion of a deep tensor: csrc/block.hip, csrc/deep.hip); a poll that is not served within 2^21 retries (seconds) raises a
 * host-visible fault word and falls through -- never a hang, but the call's output is then garbage.  The reference raises Python exceptions
 * synchronously (unet.py:995-1117 is plain torch); here launches are asynchronous, so:
 *   - mtv_check_fault(ctx) reads the word: MTV_OK, or MTV_ERR_STATE with the explanation in mtv_last_error().  It is meaningful for a call
 *     once the stream that call ran on has drained -- call it after the synchronisation point you have anyway (it does not synchronise);
 *   - every later entry point of a faulted context refuses to run (MTV_ERR_STATE): a fault is sticky, destroy the context.
 * moditalker_amd.DDPM.sample(..., strict=True) / UNetModel.forward(..., strict=True) synchronise the stream and check before returning. */
int mtv_check_fault(mtv_ctx* ctx);
/* CUs this context counts on being resident together (the device's multiProcessorCount unless overridden): every grid whose workgroups wait
 * for each other is planned within it (k_deep_block: half of it; tagged completion: all of it), larger ones fall back to the forms without
 * in-launch hand-offs (three-launch attention block, k_deep_finalize pass).  A stream whose CU mask (hipExtStreamCreateWithCUMask) leaves
 * fewer CUs is refused by mtv_forward / mtv_ddim_sample with MTV_ERR_STATE. */
int mtv_resident_cus(const mtv_ctx* ctx);
/* Testing aids: contexts created after mtv_debug_resident_cus(n) plan for n co-resident CUs (0 = the device's count / MTV_RESIDENT_CUS);
 * mtv_debug_arm_fault makes the NEXT mtv_forward / mtv_ddim_sample on this context end with a launch that raises the fault word exactly as a
 * timed-out poll does (the caller-side path -- same-call detection with strict=True, stickiness -- is then testable without a real time-out). */
int mtv_debug_resident_cus(int n);
int mtv_debug_arm_fault(mtv_ctx* ctx);

/* Test/debug: copy an internal activation (token-major [B, L_level, C]) to `dst` (device or host).
 * Names: "in<i>", "mid", "out<i>" = result after the cross-plane attention of that stage. */
int mtv_debug_tap(mtv_ctx* ctx, const char* name, float* dst, int64_t dst_cap_floats, int* tokens_out, int* channels_out);

/* Introspection for bench.py / DESIGN.md: algorithmic work of one forward at batch 1. */
typedef struct mtv_work {
    double flops_conv3x3, flops_1x1, flops_attn_core, flops_linear;
    double bytes_weights_conv, bytes_weights_other, bytes_act_conv_path;
    int32_t n_launches;        /* launches of one mtv_forward (time-embedding chain + packing + UNet) */
    int32_t n_launches_step;   /* launches of one sampler step inside mtv_ddim_sample (UNet launches only) */
} mtv_work;
int mtv_get_work(const mtv_ctx* ctx, mtv_work* out);

/* Per-launch timing of one UNet forward, measured with hipEvents on `stream` around every launch
 * of the plan (plain launches, averaged over `iters` passes).  Call with out == NULL to get the
 * number of launches in *n_out.  flops/bytes are the algorithmic figures of that launch. */
typedef struct mtv_op_time {
    char name[96];
    float ms;
    double flops;
    double bytes;
} mtv_op_time;
int mtv_profile_forward(mtv_ctx* ctx, int batch, int iters, mtv_op_time* out, int cap, int* n_out, void* stream);
/* The same for one SAMPLER STEP (the launch sequence mtv_ddim_sample replays per step: UNet launches only, the
 * DDIM update inside the head conv).  Runs on the context's own staging buffers with a one-entry step table. */
int mtv_profile_step(mtv_ctx* ctx, int batch, int iters, mtv_op_time* out, int cap, int* n_out, void* stream);

/* Diagnostic build only (conv.hip compiled with -DMTV_ABLATE=64, MTV_STAMPS=1 in the environment): one sampler step
 * with plain launches; the in-kernel phase timestamps of four sampled workgroups per conv are written to `path`. */
int mtv_debug_stamps(mtv_ctx* ctx, int batch, const char* path, void* stream);

/* (All mtv_debug_force_* knobs: set them BEFORE a context builds its first plan of a batch size -- plans of one batch size share their
 * split-K slab and split-activation scratch, sized for the tiles in force when the first of them is built.)
 * Testing aid: plans built after this call run every eligible convolution on the LDS-tiled kernel k_conv_lds<wm, wn>
 * (wave tile 16 wm x 16 wn, workgroup 2 x 2 waves) instead of the tuned choice; wm = 0 switches it off again. */
int mtv_debug_force_lds(int wm, int wn);
/* Testing aid: plans built after this call run every eligible convolution / GEMM (any row count) on the split-bf16 kernels
 * k_x3_prep + k_conv_x3<mt, nt> (workgroup tile 32 mt x 64 nt, csrc/conv_x3.hip) with `ks` K slices per tile (1, 2, 4, 8;
 * convs with fewer than 6 ks 32-channel chunks keep one slice) instead of the tuned choice; mt = 0: off. */
int mtv_debug_force_b3(int mt, int nt, int ks);
/* Testing aid: plans built after this call run every eligible 1x1 convolution (the attention blocks' qkv / proj_out,
 * unet.py:234,253) on the lean kernel k_lin<mt, nt, nwv> (wave tile 16 mt x 16 nt, nwv waves side by side along the
 * output channels, whole K per wave; csrc/lin.hip) instead of the tuned choice; mt = 0 switches it off again. */
int mtv_debug_force_lin(int mt, int nt, int nwv);
/* Testing aid: plans built after this call run every eligible 3x3 convolution (ResBlock in_layers / out_layers, unet.py:131-167) on
 * the window-staged kernel k_conv_win<mt, nt> (row tile 16 mt x column tile 16 nt; the transformed input rows of the tile and their
 * halo staged in LDS once, csrc/deep.hip) instead of the tuned choice; mt = 0 switches it off again. */
int mtv_debug_force_win(int mt, int nt);
/* ... with ks = 1 | 2 | 4 K slices per tile (round 6: each slice stages its 1 / ks of the input channels; partial tiles meet in the plan's
 * split-K slab inside the launch) on the convs that can take them -- no fused skip conv, whole 16-channel chunks per slice; the others run unsliced. */
int mtv_debug_force_win_ks(int mt, int nt, int ks);
/* Testing aid: plans built after this call run every eligible 1x1 convolution on identity rows (the attention blocks' qkv / proj_out,
 * unet.py:234,253) on k_conv_pw<mt, ntw> (16 mt rows normalised once into LDS, 8 waves side by side along 128 ntw output channels, ntw = 1 | 2,
 * whole K per wave; csrc/deep.hip) instead of the tuned choice; mt = 0 switches it off again. */
int mtv_debug_force_pw(int mt, int ntw);
/* ... with `waves` of the 8 waves multiplying (8, or 6 / 4 / 2 at ntw = 1): the column tile is 16 ntw waves wide, so that e.g. qkv at N = 384 is
 * 4 tiles of 96 columns (256 workgroups with 32-row tiles) instead of 3 of 128 (192), proj_out at N = 128 two tiles of 64. */
int mtv_debug_force_pw_waves(int mt, int ntw, int waves);
/* Testing aid: attention launches issued (or captured) after this call compute QK^T on the bf16 matrix pipe through a three-term
 * split of q and k at f32 accuracy (k_attention<..., QB = 1>: the 8-wave shapes of d = 16 / 32 / 64; PV stays on the f32
 * instruction): 1 on, 0 off, -1 back to the build default / MTV_ATT_QB. */
int mtv_debug_attention_qb(int mode);

/* Which convolution kernels the levels of at most 128 tokens per clip run on in plans built after this call (batch <= 2):
 * 1 = the K-sliced kernels of csrc/deep.hip (k_deep_conv: the consumers add the producers' partial slabs and compute the
 * GroupNorm statistics themselves), 0 = k_conv everywhere, -1 = back to the default (on) / the MTV_DEEP environment variable.
 * Replaces the same reference code either way: ResBlock convs and the attention blocks' qkv / proj_out at those levels
 * (MToV/models/ddpm/unet.py:178-207, 234, 253).  The parity tests run both. */
int mtv_debug_deep(int mode);
/* Variants of the deep levels' dataflow, for plans built after this call (testing aid; every variant meets the parity bars, the
 * default is the fastest measured): a bit mask of
 *   MTV_DEEP_OPT_INLAUNCH       a deep tensor that a k_conv / k_pool_down / qkv consumer needs as ONE plain tensor is completed inside
 *                               the producing launch (slab + ticket, last-arriving K slice) instead of by a k_deep_finalize pass;
 *   MTV_DEEP_OPT_SLICED_QKV     the attention blocks' qkv conv K-sliced on k_deep_conv (+ a finalize pass) instead of k_conv;
 *   MTV_DEEP_OPT_UNSLICED_QKV   ... on k_deep_conv with the whole K per workgroup (plain output) where it fits in LDS;
 *   MTV_DEEP_OPT_NO_FUSED_ATTN  k_attention + a proj_out conv instead of the fused k_deep_attn;
 *   MTV_DEEP_OPT_NO_BLOCK       the three-launch attention block of round 4 (k_deep_finalize + qkv conv + k_deep_attn) instead of the
 *                               one-launch k_deep_block (csrc/block.hip: GroupNorm -> qkv -> attention -> proj_out in one kernel, a
 *                               cluster of workgroups per head, two in-launch hand-offs); the four bits above imply it where they
 *                               change the block's dataflow (they describe the three-launch form);
 *   MTV_DEEP_OPT_BLOCK_ALL      k_deep_block for EVERY attention block of the deep levels it can run, not only where it measures faster
 *                               (by default: 32-token blocks and [128 x 256]; at [128 x 512] the three-launch form is faster);
 *   MTV_DEEP_OPT_FIN_PASS       a separate k_deep_finalize pass (round 4's default) wherever a consumer needs a deep tensor as ONE plain
 *                               tensor, instead of round 5's completion inside the producing kernel by data-tagged granules;
 * -1 = back to the defaults / the MTV_DEEP_INLAUNCH, MTV_DEEP_QKV, MTV_DEEP_QKV1, MTV_DEEP_NO_ATTN, MTV_DEEP_NO_BLOCK, MTV_DEEP_BLOCK_ALL,
 * MTV_DEEP_FIN_PASS environment variables. */
#define MTV_DEEP_OPT_INLAUNCH 1
#define MTV_DEEP_OPT_SLICED_QKV 2
#define MTV_DEEP_OPT_UNSLICED_QKV 4
#define MTV_DEEP_OPT_NO_FUSED_ATTN 8
#define MTV_DEEP_OPT_NO_BLOCK 16
#define MTV_DEEP_OPT_BLOCK_ALL 32
#define MTV_DEEP_OPT_FIN_PASS 64
int mtv_debug_deep_options(int mask);

/* 0: replay the step as a hipGraph (default); 1: plain launches (profiling / debugging). */
int mtv_set_eager(mtv_ctx* ctx, int eager);

/* Host-only self-test (needs no device): the conv kernels compute their 3x3 / nearest-upsample gather
 * indices arithmetically on the tri-plane grid (planes xy R x R | yt T x R | xt T x R per level, i.e.
 * the slicing of unet.py:1039-1053 and the Upsample/conv padding of unet.py:531-598); this checks that
 * arithmetic against explicitly constructed index tables for every level of a (res, frames, n_levels)
 * geometry.  Returns 0 when all agree, else 1 + the first level that does not (<0: bad arguments). */
int mtv_selftest_geometry(int res, int frames, int n_levels);
/* Host-only self-test (needs no device) of the row tables of the deep levels (csrc/deep.hip: levels of at most 128 tokens per clip,
 * whose convs stage a whole row group of a channel slice in LDS and address it through a per-launch table of LDS rows): every entry
 * -- all row groupings, 3x3 and 1x1, same-level and nearest-upsampled source -- is checked against the explicitly constructed
 * im2col tables (unet.py:178-207 ResBlock convs incl. the Upsample of :531-598).  0 = all agree, else 1 + the first level that
 * does not (<0: bad arguments). */
int mtv_selftest_deep(int res, int frames, int n_levels);
/* Host-only self-test of k_deep_block's work split (csrc/block.hip: one attention block of a deep level -- unet.py:210-300, 303-326 -- in one
 * launch): for a level of `tokens` tokens (<= 128), `channels`, `heads`, `batch` clips it configures the cluster (K slices x row groups), and checks
 * that the grid stays within the residency bound, LDS within 160 KB, that stage 2 deals every row pair to exactly one workgroup and stage 3 every
 * (query tile, column part) to exactly one, and that every granule offset of the three scratch buffers lies inside its allocation.
 * Returns 0 when all hold, 1 when the shape is (legitimately) not configurable, > 1 on a violated invariant, < 0 on bad arguments. */
int mtv_selftest_block(int tokens, int channels, int heads, int batch);
/* Host-only self-test of k_conv_win's window arithmetic (csrc/deep.hip: the contiguous range of source tokens a row tile of a 3x3
 * conv stages in LDS): for row tiles of 16 and 32 tokens at every level, same-level and nearest-upsampled source, the window lies inside
 * the source tensor and contains the source token of every tap of every output row (the taps of mtv_debug_gather_index).  0 = ok,
 * else 1 + the first level that fails (<0: bad arguments). */
int mtv_selftest_win(int res, int frames, int n_levels);
/* The arithmetic itself, for tests: source token of tap (ky, kx in 0..2) at output token `tok` of a level with
 * planes res x res | frames x res | frames x res; up != 0: the source is the (res/2, frames/2) level under a
 * nearest x2 upsample.  Returns -1 for zero padding, else source_token | plane << 28. */
int mtv_debug_gather_index(int res, int frames, int tok, int ky, int kx, int up);

/* ------------------------------------------------------------------------------------------------
 * The autoencoder steps either side of the denoising loop.
 * Replaces ViTAutoencoder.decode_from_sample / .extract (MToV/models/autoencoder/autoencoder_vit.py:257-275, 212-255;
 * called at MToV/sample.py:328-332,369,386).  A separate context of the same opaque type: weights go in through
 * mtv_num_weights / mtv_weight_info / mtv_load_weight / mtv_weights_missing with the reference's state_dict keys
 * (e.g. "decoder.layers.0.0.fn.to_qkv.weight" [1536,384], "to_pixel.1.weight" [384,3,8,8]).
 * ------------------------------------------------------------------------------------------------ */
typedef struct mtv_ae_config {
    int32_t channels;      /* ddconfig.channels (384)                                                    */
    int32_t resolution;    /* ddconfig.resolution (256)                                                  */
    int32_t frames;        /* ddconfig.timesteps / splits (16)                                           */
    int32_t patch;         /* 8 (4 at resolution 128: autoencoder_vit.py:105-107)                        */
    int32_t embed_dim;     /* 4                                                                          */
    int32_t depth;         /* 8 TimeSformer layers per coder (autoencoder_vit.py:110-116)                */
    int32_t heads;         /* 8                                                                          */
    int32_t dim_head;      /* 64                                                                         */
    int32_t max_batch;
} mtv_ae_config;

int mtv_ae_create(const mtv_ae_config* cfg, mtv_ctx** out);
int mtv_ae_destroy(mtv_ctx* ctx);
/* Rotary tables, computed by the host exactly as RotaryEmbedding / AxialRotaryEmbedding.forward do
 * (vit_modules.py:29-49,57-62): time_tab [frames][2][dim_head], space_tab [(res/patch)^2][2][dim_head]; row 0 = sin,
 * row 1 = cos.  Host or device pointers. */
int mtv_ae_set_rotary(mtv_ctx* ctx, const float* time_tab, const float* space_tab);
/* latents [B, embed_dim, r*r + 2*frames*r] (r = resolution/patch; the sampler's output) -> frames
 * [B*frames, 3, resolution, resolution] in (-1, 1).  Device pointers. */
int mtv_ae_decode(mtv_ctx* ctx, const float* latents, float* frames_out, int batch, void* stream);
/* video [B, 3, frames, resolution, resolution] in [-1, 1] -> latents [B, embed_dim, r*r + 2*frames*r] (tanh outputs). */
int mtv_ae_extract(mtv_ctx* ctx, const float* video, float* latents_out, int batch, void* stream);
/* Per-launch hipEvent timing of decode (extract != 0: of extract), like mtv_profile_forward. */
int mtv_ae_profile(mtv_ctx* ctx, int batch, int extract, int iters, mtv_op_time* out, int cap, int* n_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * CrossAttention (unet.py:429-467) as a stand-alone operator: the one usable piece of the
 * reference's dormant cross-attention conditioning (SpatialTransformer cannot be constructed there, unet.py:470-489,513).
 * Weights through mtv_load_weight with the module's keys: "to_q.weight" [H*d, query_dim], "to_k.weight" / "to_v.weight"
 * [H*d, context_dim], "to_out.0.weight" [query_dim, H*d], "to_out.0.bias" [query_dim].
 * ------------------------------------------------------------------------------------------------ */
typedef struct mtv_xattn_config {
    int32_t query_dim, context_dim, heads, dim_head;
    int32_t max_batch, max_queries, max_keys;
} mtv_xattn_config;
int mtv_xattn_create(const mtv_xattn_config* cfg, mtv_ctx** out);
int mtv_xattn_destroy(mtv_ctx* ctx);
/* x [B, n_queries, query_dim]; context [B, n_keys, context_dim] or NULL (= x: self-attention); mask [B, n_keys] bytes or
 * NULL (0 = key masked out, unet.py:452-456); out [B, n_queries, query_dim].  Device pointers. */
int mtv_xattn_forward(mtv_ctx* ctx, const float* x, const float* context, const unsigned char* mask, float* out, int batch,
                      int n_queries, int n_keys, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MTV_HIP_H */
