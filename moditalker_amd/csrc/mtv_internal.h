// Internal declarations shared by kernels.hip (device code + launchers) and plan.hip (host plan,
// weights, graph capture, C ABI).  gfx950 only; no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mtv {

// Plane boundaries inside one batch element's token axis: [0,b1) xy (r x r), [b1,b2) yt (t x r),
// [b2,L) xt (t x r)  -- MToV/models/ddpm/unet.py:1027-1029 generalised to (R>>lvl, T>>lvl).
struct SegInfo {
    int b1, b2, L;
};

// Consumer-side GroupNorm description: statistics live in a per-site table of fp64 (sum, sumsq)
// per (batch, plane, group); the consumer turns them into per-channel affine coefficients.
struct GnIn {
    const double* sums;   // [STAT_COPIES][...][B][3][32][2] (copy 0 here, copies `cstride` doubles apart) or nullptr
    const float* gamma;   // [Cmain]
    const float* beta;    // [Cmain]
    const float* film;    // per batch: scale at [c], shift at [Cmain + c]; nullptr = none
    int film_stride;      // floats between batch elements of `film`
    int gs;               // channels per group = Cmain / 32
    int whole;            // 1: statistics over all L tokens (AttentionBlock1D); 0: per plane
    int act;              // 1: SiLU after the affine
    unsigned cstride;     // doubles between the privatised copies of the statistics arena
    // filled by add_conv from the fields above (the kernel multiplies instead of dividing):
    float inv_gs;         // 1 / gs
    double inv_n[4];      // 1 / (tokens x gs) of plane 0, 1, 2 and of all planes together (`whole`)
};

// Producer-side GroupNorm statistics: a conv epilogue adds the (sum, sumsq) of its output to the
// fp64 table of every GN site that will normalise this tensor (possibly as one part of a channel
// concatenation: `coff` = offset of this tensor's channel 0 in the consumer's channel axis).
// Producers spread their atomics over STAT_COPIES privatised copies of the table (copy = workgroup id & 7,
// i.e. normally the XCD it runs on; any choice is correct, consumers add the copies up): 8x fewer atomics
// per address, and the per-address atomic rate is what bounds a stats epilogue.
constexpr int STAT_COPIES = 8;
struct StatOut {
    double* sums;   // copy 0 of [B][3][32][2]
    int gs;         // consumer's channels per group
    int coff;
    float inv_gs;   // 1 / gs
};

struct DdimStep;

// Sampler-step duties of the HEAD conv (the last launch of a denoising step): its epilogue applies the eta-general
// DDIM update to the sample it has just predicted eps for, and the launch also prepares the NEXT step -- so a
// step is exactly the UNet's own launches, replayed as one hipGraph (no k_ddim_update / k_ddim_advance /
// k_pack_input / time-embedding / memset launches).  Lives in device memory (rewritten per mtv_ddim_sample call:
// graph kernel arguments are frozen, this record is not).
struct DdimFuse {
    float* x;                 // [B][4][L] current sample, external layout; updated in place (ddpm.py:386-398)
    float* h0;                // [B][L][16] packed UNet input: channels 0..3 <- the new sample (unet.py:1022-1025)
    const float* noise;       // [n][B][4][L] in-loop N(0,1) draws
    const DdimStep* steps;    // per-step scalars
    int* counter;             // index of the current step; advanced by the head's last workgroup
    int* done;                // arrival counter of the head's workgroups (zero between launches)
    long long n_per_draw;     // B*4*L
    const float* film_tab;    // [n_steps][film_total] FiLM rows of every step (t is the same for all clips of a call)
    float* film_out;          // [film_total] the row the ResBlocks read; this launch copies row counter+1 into it
    int film_total, n_steps;
    float* zero_arena;        // the OTHER step parity's GroupNorm statistics arena: zeroed here for the next step
    long long zero_vec4;      // ... its size in 16-byte units
};

struct ConvArgs {
    const float* src[4];  // [0..1]: tapped sources (<=2, concatenated along channels);
    int C[4];             // [2..3]: raw sources of the fused 1x1 skip conv (<=2)
    int nmain, nskip;
    int Cmain, Cskip;
    const int* gather;       // [ntaps][Lout] source token per (tap, output token), -1 = zero pad; nullptr = identity
    const int* gather_skip;  // [Lout] source token for skip parts / residual; nullptr = identity
    int ntaps;
    int Lout, Lsrc, Lskip;   // tokens per batch element: output, tapped source, skip/residual source
    int B;
    const float* W;          // rows [ntaps*Cmain + Cskip][ldw], columns = output channels
    const float* Wnk;        // 1x1 convs on identity rows only: the same weight as stored in the checkpoint, [N][Cmain] (k_lin, lin.hip)
    const float* Wpk;        // 1x1 convs on identity rows only (N, Cmain multiples of 16): k_conv_pw's operand [N / 16][Cmain / 16][64 lanes][4] -- lane (j, q) of (column
                             // block bn, chunk c) holds W[16 bn + j][16 c + 4 q .. + 3], so a wave's B fragment of a chunk is ONE contiguous KB (k_repack_pw, kernels.hip)
    const void* W3;          // the same matrix as three bf16 planes [3][K/8][ldw][8] with W = W0 + W1 + W2 (k_conv_x3, conv_x3.hip) or nullptr
    unsigned long long w3_plane;   // bytes between the planes
    void* x3;                // scratch for this conv's split activations (k_x3_prep -> k_conv_x3, conv_x3.hip) or nullptr
    int ldw;
    int N;
    const float* bias;       // [N]
    const float* bias2;      // [N] or nullptr (bias of the fused skip conv)
    const float* bias_b;     // per-batch additive [B][bias_b_stride] or nullptr
    int bias_b_stride;
    GnIn gn;
    SegInfo seg_src;         // plane boundaries of the tapped source token axis
    const float* res;        // residual [B][Lskip][N] (identity skip) or nullptr
    float* out;
    int out_cm;              // 1: write channel-major [B][N][Lout] (the external layout)
    SegInfo seg_out;         // plane boundaries of the output token axis (for the statistics)
    StatOut stat[2];
    int nstat;
    unsigned stat_cstride;   // doubles between the copies of the statistics arena
    int KS;                  // >1: cross-workgroup split-K; partial tiles go to `slab`, the last slice completes
    int* tickets;            // [B * row tiles * column tiles] arrival counters (zero between launches)
    float* slab;             // [KS][B][Lout][N]
    int xmap;                // ConvTile::XM
    // Gather by arithmetic instead of by table (same values -- mtv_create() checks the formula against the
    // tables on the host and leaves these 0 if they ever disagreed): the kernel then needs no global load
    // before its operand loads.  geo_main: 0 = `gather` table / identity, 1 = 3x3 taps on the output grid,
    // 2 = 3x3 taps on the nearest-x2-upsampled source.  geo_skip: 0 = `gather_skip` table / identity,
    // 2 = nearest-x2-upsampled source.  (geo_r, geo_t) = output-level plane geometry.
    int geo_main, geo_skip, geo_r, geo_t;
    // ---- derived by launch_conv() from the tile (the kernel never divides by a run-time value on its way to the
    // first load: these are the quotients / reciprocals it needs)
    int tiles_per_b, tiles_n, Bt, Sx;      // row tiles per batch element, column tiles, B * tiles_per_b, ceil(tiles_n * KS / 8)
    float inv_tiles_per_b, inv_tiles_n, inv_Bt, inv_Sx, inv_cpt, geo_inv_r;
    int cpt;                               // K chunks (16 channels) per tap = Cmain / 16
    int cps_q, cps_r;                      // chunks per K slice: slice s of KS * NW covers [s*q + min(s, r), + q + (s < r))
    int rec_cap;                           // chunk records per workgroup (LDS capacity, 16 bytes each)
    int qs_off;                            // LDS offset (floats) of the per-quad statistics slots
    const DdimFuse* ddim;    // sampler-step head only (nullptr otherwise)
    const int* step_counter; // ... the device-side step index (DdimFuse::counter; here too so it is loaded at entry)
    unsigned long long* dbg; // phase timestamps for tools/ubench/conv_bench (nullptr in the product)
    long long dbg_stat_dup;  // 0 in the product: offset (doubles) of the other step parity's statistics arena (MTV_DEBUG_STATDUP timing experiment)
};

// Touch every 64-byte line of the kernel-argument block at kernel entry.  The compiler otherwise loads
// struct fields lazily, each first touch of a line being a scalar-cache miss that goes all the way to
// memory (~1 us) and they serialise (load, wait, compute, load the next line, wait ...): three to four
// dependent misses at the head of every launch.  One combined wait here, then every later s_load hits.
#if defined(__HIPCC__)
// Output tensors leave a kernel through WRITE-THROUGH 16-byte stores (sc0 sc1).  A plain store parks the line dirty in the XCD's L2 until the
// write-back at the end of the launch; the next launch -- on other XCDs -- then reads a tensor that has only just started towards memory.
// Written through, it is on its way while the launch still computes and the end-of-launch write-back finds nothing to do.  Same-box A/B of
// the step (profiles/r06_write_through_stores_ab.txt): 1.990 -> 1.980 ms; non-temporal stores instead: slower.  -DMTV_OUT_PLAIN=1
// restores the plain stores.
// `base` must be wave-uniform (a kernel argument, or one plus a workgroup-uniform slab offset): it becomes a buffer descriptor; `idx` = float
// index of the quad (byte offset < 4 GB).  The store is the compiler's own raw-buffer store with the cache-policy operand -- NOT inline
// assembly: the waitcnt pass does not see a memory operation inside an asm statement, so an asm store between loads makes the compiler's
// vmcnt(N) waits pass early (a first version of this helper did exactly that: parity broke in every kernel whose epilogue loops).
typedef float mtv_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned mtv_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mtv_store_out4(float* base, size_t idx, const mtv_f32x4& v) {
#if defined(MTV_OUT_PLAIN) && MTV_OUT_PLAIN
    *reinterpret_cast<mtv_f32x4*>(base + idx) = v;
#else
    const unsigned long long u = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
    float* const b = reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(b, 0, -1, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mtv_u32x4, v), rs, (unsigned)(idx * 4), 0, 17);      // 17 = sc0 | sc1
#endif
}
#endif
#if defined(__HIPCC__)
template <int BYTES>
__device__ __forceinline__ void touch_kernargs() {
    static_assert(BYTES <= 10 * 64, "argument block larger than 10 lines");
    typedef const __attribute__((address_space(4))) void* kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    // written out in assembly so that the loads stay together, first, with ONE wait (left to the scheduler they
    // get interleaved with the first real argument loads and the misses serialise again)
    unsigned t0, t1, t2, t3, t4, t5, t6, t7;
    constexpr int LAST = BYTES - 4;              // (the block need not start on a line boundary)
    if constexpr (BYTES <= 512) {
        asm volatile(
            "s_load_dword %0, %8, 0x0\n\t"
            "s_load_dword %1, %8, %9\n\t"
            "s_load_dword %2, %8, %10\n\t"
            "s_load_dword %3, %8, %11\n\t"
            "s_load_dword %4, %8, %12\n\t"
            "s_load_dword %5, %8, %13\n\t"
            "s_load_dword %6, %8, %14\n\t"
            "s_load_dword %7, %8, %15\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7)
            : "s"(ka), "n"(BYTES > 64 ? 64 : LAST), "n"(BYTES > 128 ? 128 : LAST), "n"(BYTES > 192 ? 192 : LAST),
              "n"(BYTES > 256 ? 256 : LAST), "n"(BYTES > 320 ? 320 : LAST), "n"(BYTES > 384 ? 384 : LAST), "n"(LAST)
            : "memory");
    } else {                                     // (round 5: DeepArgs outgrew eight lines -- ten loads, still ONE wait)
        unsigned t8, t9;
        asm volatile(
            "s_load_dword %0, %10, 0x0\n\t"
            "s_load_dword %1, %10, 0x40\n\t"
            "s_load_dword %2, %10, 0x80\n\t"
            "s_load_dword %3, %10, 0xc0\n\t"
            "s_load_dword %4, %10, 0x100\n\t"
            "s_load_dword %5, %10, 0x140\n\t"
            "s_load_dword %6, %10, 0x180\n\t"
            "s_load_dword %7, %10, 0x1c0\n\t"
            "s_load_dword %8, %10, 0x200\n\t"
            "s_load_dword %9, %10, %11\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9)
            : "s"(ka), "n"(BYTES > 576 ? 576 : LAST)
            : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);           // nothing (in particular no argument load) is scheduled above this
}
#endif

// Source token of (tap ky,kx in 0..2; output token) on the tri-plane grid of a level with planes
// xy r x r | yt t x r | xt t x r; `up`: the source lives on the (r/2, t/2) level (nearest x2 upsample).
// Returns -1 for zero padding, else token | plane << 28 (the layout of the kernel's index table).
// `Div`: how local / r is evaluated -- plain integer division on the host (the checker of mtv_create and the CPU tests),
// the exact float-reciprocal form inside the kernels (FDiv below); everything else is the same code.
struct IDiv {
    __host__ __device__ int operator()(int n, int d) const { return n / d; }
};
template <class Div>
__host__ __device__ inline int geo_source_t(Div dv, int r, int t, int tok, int ky, int kx, bool up) {
    const int b1 = r * r, b2 = b1 + t * r;
    const int p = tok >= b2 ? 2 : (tok >= b1 ? 1 : 0);
    const int off = p == 0 ? 0 : (p == 1 ? b1 : b2), h = p == 0 ? r : t;
    const int local = tok - off, y = dv(local, r), x = local - y * r;
    const int yy = y + ky - 1, xx = x + kx - 1;
    if (yy < 0 || yy >= h || xx < 0 || xx >= r) return -1;
    if (!up) return (off + yy * r + xx) | (p << 28);
    const int rs = r >> 1, ts = t >> 1, b1s = rs * rs, b2s = b1s + ts * rs;
    const int offs = p == 0 ? 0 : (p == 1 ? b1s : b2s);
    return (offs + (yy >> 1) * rs + (xx >> 1)) | (p << 28);
}
__host__ __device__ inline int geo_source(int r, int t, int tok, int ky, int kx, bool up) { return geo_source_t(IDiv{}, r, t, tok, ky, kx, up); }
#if defined(__HIPCC__)
struct FDiv {      // exact n / d for 0 <= n < 2^22 given inv = 1.0f / d (one multiply and a +-1 correction)
    float inv;
    __device__ __forceinline__ int operator()(int n, int d) const {
        int qt = (int)(((float)n + 0.5f) * inv);
        const int r = n - qt * d;
        return qt + (r >= d) - (r < 0);
    }
};
#endif

struct StatsArgs {   // (fallback pass: writes copy 0 only)
    const float* src[2];
    int C[2];
    int nparts;
    int Ctot, gs;
    int B;
    SegInfo seg;
    double* sums;            // [B][3][32][2], pre-zeroed
};

struct PoolArgs {            // ResBlock(down=True): avgpool2x2 of SiLU(GN(x)) and of x (unet.py:179-184)
    const float* x;          // [B][Lsrc][C]
    float* out_act;          // [B][Ldst][C]
    float* out_x;            // [B][Ldst][C]
    const double* sums;
    unsigned cstride;
    const float* gamma;
    const float* beta;
    int B, C, gs;
    SegInfo seg_src, seg_dst;
    int r_dst, t_dst;        // destination plane geometry: xy r_dst x r_dst, yt/xt t_dst x r_dst
};

struct AttnArgs {
    const float* qkv;        // [B][L][3C], channel = head*3d + {q: 0..d, k: d..2d, v: 2d..3d}
    float* out;              // [B][L][C], channel = head*d + i
    int B, L, C, H;
    int nseg;
    int seg_start[3], seg_len[3];
    int blk_prefix[4];       // prefix sums of ceil(seg_len/64): one workgroup per 64 queries
    float scale;             // d^-1/4, applied to q AND k (unet.py:322-323)
    // uniform segmentation (the autoencoder's time / space attention: thousands of equal segments): when
    // seg_uniform > 0 the token axis is cut into L / seg_uniform segments of that length and seg_start/len are unused
    int seg_uniform;
    int bps;                 // query blocks per uniform segment (set by launch_attention)
    float inv_H, inv_nblk, inv_bps;   // reciprocals for the kernel's block decode (set by launch_attention)
    // cross-attention (CrossAttention, unet.py:429-467): `qkv` then holds the queries only ([B][L][C], head-major) and
    // kv the keys/values [B][Lkv][2C] (k | v of a head adjacent); one segment of all L queries
    const float* kv;
    int Lkv;
    const unsigned char* kmask;   // [B][Lkv] or nullptr: 0 = key masked out
    unsigned long long* dbg;      // diagnostic build (-DMTV_ATT_STAMP): phase timestamps of four sampled workgroups, else unused
    // deep levels (deep.hip): `qkv` is the sum of qkv_ks partial slabs, qkv_slab floats apart (0 / 1: a plain tensor)
    int qkv_ks;
    unsigned qkv_slab;
};

// ---- deep levels (<= 128 tokens per clip): K-sliced convs whose partial results are summed by the CONSUMER (deep.hip) ----
// A tensor of the deep region is a sum of `ks` partial slabs [B][L][C] (`slab_stride` floats apart): the K slices of the conv
// that produced it, slice 0 carrying bias + residual.  Nobody finishes the split-K sum in the producing launch (no slab
// round trip, no ticket, no statistics atomics): every consumer adds the slabs of the channel slice it reads, in slab order
// (run-to-run bit-equal), and computes the GroupNorm statistics of that slice itself -- at <= 128 tokens a workgroup holds
// all tokens of its channel slice.  ks == 1 is a plain tensor.
struct DeepSrc {
    const float* p;          // slab 0 (nullptr: absent)
    unsigned slab_stride;    // floats between slabs
    int ks;                  // slabs: 1, 2, 4 or 8
    int C;                   // channels = row stride
};

// Optional in-launch completion of a deep tensor (a consumer outside the deep region -- k_conv, k_pool_down, k_attention, the
// attention blocks' qkv -- wants ONE plain tensor and, for GroupNorm, its statistics): every K slice parks its partial tile with
// write-through 8-byte stores and takes a ticket; the slice that draws the last one re-reads all slabs (slab order), writes the
// plain tile and adds the statistics -- conv.hip's split-K completion (cdna_hip_programming.md G16: 8-byte agent atomics both sides).
struct DeepFin {
    float* out;              // plain [B][L][N] (nullptr: slabs only)
    int* tickets;            // one per output tile (zero between launches: the last slice resets its own)
    StatOut stat[2];
    int nstat;
    unsigned stat_cstride;
    SegInfo seg;             // plane boundaries of the output level
    // ---- round 5: completion by data-tagged granules instead of drain + ticket + re-read (MI355X_MICROARCH.md hand-off price list):
    // every workgroup of a tile takes an entry ticket (-> the launch's epoch); K slices 1 .. KS-1 also write their partial quads as
    // 8-byte {value, epoch} granules; slice 0 polls them, adds them in slice order to its own tile and writes the plain tensor + statistics
    int tagged;              // 1: this form (tickets / parked slabs unused)
    float* gran;             // [slices - 1][B][L][N] granules (8 bytes each)
    unsigned gran_bytes;
    unsigned long long* ecnt;   // [tiles] monotonic entry-ticket counters (never reset)
    int* fault;              // raised when a poll times out
};

struct DeepArgs {
    DeepSrc main[2];         // tapped source, <= 2 parts concatenated along channels (main[1].p == nullptr: one part)
    DeepSrc skip[2];         // parts of the fused 1x1 skip conv (raw, rows = output tokens); skip[0].p == nullptr: none
    DeepSrc res;             // residual added by K slice 0 (res.p == nullptr: none)
    int Cmain, Cskip;        // channels of the concatenations
    int ntaps;               // 9 (3x3 on the tri-plane grid) or 1
    int up_main, up_res;     // 1: that source lives on the next-coarser level (nearest x2 upsampling folded into the row map)
    int pool_main, pool_res; // 1: that source lives on the next-FINER level (ResBlock(down=True), unet.py:179-184): pool_main -- GroupNorm / SiLU are
                             // applied there (once per element, in LDS), then the 2x2 mean per plane (AvgPool2d, unet.py:594) gives this level's rows;
                             // pool_res -- the residual is the 2x2 mean of the raw source, shared out over the K slices (slice s adds source slabs s, s + KS, ...)
    int lds_pool;            // LDS offset (floats) of the pooled rows (pool_main; set by the layout)
    int r, t;                // plane geometry of the OUTPUT level: xy r x r | yt t x r | xt t x r
    int B, Lout, Lsrc, Lres; // tokens per clip: output, tapped source, residual source
    int N;                   // output channels (multiple of 16 NT)
    const float* W;          // deep layout [column tile][K slice][chunk][NT][64 lanes][4]  (k_deep_repack)
    const float* bias;       // [N]
    const float* bias2;      // [N] or nullptr (bias of the fused skip conv)
    const float* bias_b;     // per clip [B][bias_b_stride] or nullptr
    int bias_b_stride;
    const float* gamma;      // GroupNorm of the tapped source (gn != 0): [Cmain]
    const float* beta;
    const float* film;       // per clip: scale at [c], shift at [Cmain + c]; nullptr = none
    int film_stride;
    int gs;                  // channels per group
    int gn, whole, act;      // gn: normalise; whole: statistics over all planes (AttentionBlock1D); act: SiLU
    const int* rowtab;       // device copy of deep_rowtab(): [nrg][(ntaps + 1)][16 RT] LDS offsets (floats)
    const float* zeros;      // >= 2 Cmain zero floats: stands in for absent GroupNorm / FiLM vectors (launch_deep_conv)
    float* out;              // slab 0 of the output [KS][B][Lout][N]
    unsigned out_slab_stride;
    int KS;                  // K slices = output slabs (power of two)
    int CSm, CSs;            // channels per K slice of the tapped source / of the skip source (CSm: power of two multiple of 16)
    int nrg;                 // row groups per clip: 1 = all planes, 2 = {xy}, {yt, xt}
    int tiles_n;             // column tiles (N / 16 NT)
    // ---- derived by launch_deep_conv
    int ks_shift, cpt_shift; // log2(KS), log2(CSm / 16)
    int nmain_ch, nch;       // 16-channel chunks per slice: ntaps * CSm / 16, + CSs / 16
    int nslots;              // B * nrg * KS
    int xm, xm_jbits;        // xm = 1 (B = 1, nrg = 2): workgroup id mod 8 = the XCD selects (K slice, low xm_jbits bits of the column tile), so the two row groups of a (K slice,
                             // column tile) -- which stream the SAME weights -- meet on one XCD
    float inv_nslots, inv_r, inv_rs;
    int lds_skip, lds_idx, lds_stat, lds_red;   // LDS offsets (floats): skip slice | row table | statistics scratch | reduction scratch
    int src_rows_max;        // rows of the staged main slice (largest row group) -- its zero row follows
    double inv_n[4];         // 1 / (tokens x gs) of source plane 0, 1, 2 and of all planes together
    unsigned long long* dbg; // -DMTV_DEEP_STAMP builds (tools/ubench/deep_bench): phase timestamps of two sampled workgroups, else unused
    DeepFin fin;
    int lds_fin;             // LDS offset (floats) of the completion scratch (flag + statistics slots)
};

// slabs -> one plain tensor (+ GroupNorm statistics into the site tables of legacy consumers): the exits of the deep region
struct DeepFinArgs {
    DeepSrc src;
    float* out;              // [B][L][C]
    int B, L;
    SegInfo seg;
    StatOut stat[2];
    int nstat;
    unsigned stat_cstride;
};

// QKVAttentionLegacy core + proj_out of one attention block at <= 128 tokens (deep.hip, k_deep_attn): a workgroup = (clip, head
// group, 32 queries, column group) computes the attention of its heads for its queries and multiplies it by its slice of the
// proj_out matrix: the head group IS the K slice of the projection, its partial result goes to slab `head group`.
struct DeepAttnArgs {
    const float* qkv;        // plain [B][L][3C], channel = head * 3d + {q: 0.., k: d.., v: 2d..}  (unet.py:312-326)
    int B, L, C, H;
    int r, t;                // plane geometry of the level
    int whole;               // 1: every query sees all L keys (AttentionBlock1D); 0: the keys of its own plane
    float scale;             // d^-1/4, applied to q and to k
    const float* Wp;         // proj_out as k_conv stores it: [C rows = input channel][ldw], columns = output channel
    int ldw;
    const float* bias;       // [C]
    DeepSrc res;             // the block's input x (residual), possibly slabs
    float* out;              // slab 0 of [H / HPW][B][L][C]
    unsigned out_slab_stride;
    int HPW, NC;             // heads per workgroup (K slice = HPW d channels), output columns per workgroup (16 or 32; 64 with QT = 1)
    int nhg, nqg, ncg;       // head groups, query groups (16 QT rows), column groups
    int QT;                  // query tiles (16 rows) per workgroup: 2 = 2 query tiles x 4 key parts (round 4), 1 = 1 query tile x 8 key parts (round 6: half the
                             // recomputed attention per workgroup, 64-column proj slices)
    int kcap;                // key capacity of the LDS tiles: 16 ceil(L / 16)
    float inv_nslots;
    int nslots;              // B * nqg * nhg
    unsigned long long* dbg; // -DMTV_DEEP_STAMP builds: phase timestamps, else unused
    DeepFin fin;
};

// One whole attention block of a deep level in ONE launch (block.hip, k_deep_block): GroupNorm -> qkv -> attention -> proj_out +
// residual, cut along heads -- a cluster of CL workgroups owns one (clip, head); two in-launch hand-offs inside the cluster.
struct DeepBlockArgs {
    DeepSrc x;               // the block's input (also its residual): [ks][clips][L][C]
    int B, L, C, H;          // head dim d = C / H (16, 32 or 64); H = 1, 2, 4 or 8 (the heads are the output slabs)
    int r, t;                // plane geometry of the level: xy r x r | yt t x r | xt t x r
    int whole;               // 1: AttentionBlock1D -- statistics and attention over all L tokens; 0: per plane
    float scale;             // d^-1/4, applied to q and to k (unet.py:322-323)
    const float* gamma;      // GroupNorm affine [C]
    const float* beta;
    int gs;                  // channels per GroupNorm group (C / 32)
    const float* Wq;         // qkv weight as the checkpoint stores it: [3C][C], row = output channel head * 3d + {q: 0.., k: d.., v: 2d..}
    const float* bq;         // [3C]
    const float* Wp;         // proj_out weight as stored: [C][C], row = output channel
    const float* bp;         // [C]
    float* out;              // slab 0 of [H][clips][L][C]: slab h = head h's share of the projection (+ input slabs h, h + H, ...; bias in slab 0)
    unsigned out_slab_stride;
    int CL, CS;              // workgroups per cluster = KSN x RQ; channels per K slice = C / KSN
    int KSN, RQ, RPQ;        // K slices of the qkv GEMM x row groups of a cluster (RQ = 1: every workgroup stages all tokens of its slice; RQ > 1: its
                             // RPQ = padded rows / RQ rows only -- the GroupNorm statistics of the row groups are exchanged in-launch); CL = KSN RQ
    float* stg;              // RQ > 1: scratch [B H][KSN][RQ][192] 16-byte granule pairs: partial statistics (fp64 lo | hi) of a row group
    unsigned stg_bytes;
    float* part;             // scratch [B H][CL][L][3d]: partial qkv of the K slices
    float* qkv;              // scratch [B H][L][3d] 8-byte {value, epoch tag} granules: the head's q | k | v rows (bias added)
    unsigned long long* cnt; // [B H][2] monotonic counters (never reset): [0] arrivals of hand-off 1, [1] entry tickets (-> the launch's epoch)
    int* fault;              // set when a hand-off wait timed out
    int rows_per;            // stage 2: rows a workgroup reduces = ceil(L / CL)
    int nqt, ncp, ncols;     // stage 3: query tiles, column parts, columns per part (C / ncp <= 256)
    // ---- derived by launch_deep_block
    int cl_shift, qw_shift, ksn_shift;  // log2(CL), log2(CS / 4), log2(KSN)
    unsigned part_bytes, qkv_bytes;   // sizes of the two scratch buffers (buffer descriptors)
    double inv_n[4];         // 1 / (tokens x gs) of plane 0, 1, 2 and of all planes together
    unsigned long long* dbg; // -DMTV_DEEP_STAMP builds: phase timestamps, else unused
};

struct LinearArgs {
    const float* x;          // [B][K]
    const float* W;          // [N][K]
    const float* bias;       // [N]
    float* out;              // [B][out_stride]
    int B, K, N, out_stride;
    int act_in;              // 0 none, 1 SiLU on the input
};

struct DdimStep {            // mirror of mtv_ddim_step (include/mtv_hip.h)
    int32_t t, last;
    float sqrt_recip_ac, sqrt_recipm1_ac, sqrt_ac_next, c, sigma;
    int32_t noise_index;
};

#if defined(__HIPCC__)
// One element of the eta-general DDIM update (ddpm.py:278-282 predict_start_from_noise, :346-351 clamp, :386-398):
// x0 = clamp(sqrt(1/ac)*x - sqrt(1/ac - 1)*eps, -1, 1);  last step: x <- x0;  else
// x <- x0*sqrt(ac_next) + c*eps + sigma*noise, with the reference's separate roundings (no contraction).
__device__ __forceinline__ float ddim_update_elem(const DdimStep& st, float x, float e, float nz) {
    float x0 = __fsub_rn(__fmul_rn(st.sqrt_recip_ac, x), __fmul_rn(st.sqrt_recipm1_ac, e));
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    if (st.last) return x0;
    return __fadd_rn(__fadd_rn(__fmul_rn(x0, st.sqrt_ac_next), __fmul_rn(st.c, e)), __fmul_rn(st.sigma, nz));
}
#endif

// ---- launchers (kernels.hip) ----
struct ConvTile { int MT, NT, NW, KS, XM; };   // XM: workgroup->tile mapping (0 rows fastest, 1 weight slice per XCD)
                                               // NW == 32: the LDS-tiled kernel k_conv_lds<WM = MT, WN = NT> (KS = 1, XM = 0)
                                               // NW == 64: the lean 1x1 kernel k_lin<MT, NT, NWV = KS> (lin.hip; XM = 0)
                                               // NW == 48: the split-bf16 kernels k_x3_prep + k_conv_x3<MT, NT> (conv_x3.hip; KS = 1, XM = 0)
constexpr size_t CONV_X3_MAX_LDS = 160 * 1024;          // k_conv_x3 has no static LDS: the whole 160 KB
bool conv_x3_eligible(const ConvArgs& a);
bool x3_tile_exists(int MT, int NT);
size_t conv_x3_scratch_bytes(int B, int Lsrc, int Cmain, int Lskip, int Cskip);
size_t conv_x3_smem_bytes(const ConvArgs& a, ConvTile t);
hipError_t conv_x3_init_attrs();
hipError_t launch_conv_x3(const ConvArgs& a, ConvTile t, hipStream_t s);
hipError_t launch_conv_x3_geglu(const ConvArgs& a, ConvTile t, const float* h, hipStream_t s);
hipError_t launch_split_w3(const float* W, void* W3, size_t plane_bytes, int row0, int rows, int ldw, hipStream_t s);
// k_conv_win (deep.hip): 3x3 convs of the large levels with the transformed input window staged in LDS -- ConvTile NW == 80
bool conv_win_eligible(const ConvArgs& a, int MT, int NT, int KS = 1);     // KS = ConvTile::KS of the tile: K slices (1 | 2 | 4)
int conv_win_selftest(int r, int t, bool up, int Lout, int Lsrc);      // host-only check of the window arithmetic (0 = ok)
size_t conv_win_smem_bytes(const ConvArgs& a, ConvTile t);
hipError_t launch_conv_win(const ConvArgs& a, ConvTile t, hipStream_t s);
// k_conv_pw<MT, NTW> (deep.hip): 1x1 conv on identity rows, rows normalised once into LDS, waves side by side along N (ConvTile NW = 96)
bool conv_pw_eligible(const ConvArgs& a, int MT, int NTW, int wcode = 1);     // wcode = ConvTile::KS of the tile: 1 = all 8 waves multiply, 6 / 4 / 2 = that many (column tile 16 NTW x waves)
size_t conv_pw_smem_bytes(const ConvArgs& a, ConvTile t);
hipError_t launch_conv_pw(const ConvArgs& a, ConvTile t, hipStream_t s);
bool conv_lin_eligible(const ConvArgs& a);
size_t lin_smem_bytes(const ConvArgs& a);
hipError_t launch_lin(const ConvArgs& a, ConvTile t, hipStream_t s);
bool conv_lds_eligible(const ConvArgs& a);
ConvTile conv_pick_tile(int B, int Lout, int N, int nchunks, int Cmain, bool has_gn);
size_t conv_smem_bytes(const ConvArgs& a, ConvTile t);
hipError_t launch_conv(const ConvArgs& a, ConvTile t, hipStream_t s);
hipError_t conv_init_attrs();
hipError_t attn_init_attrs();
hipError_t launch_gn_stats(const StatsArgs& a, hipStream_t s);
hipError_t launch_pool_down(const PoolArgs& a, hipStream_t s);
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);
extern int g_attn_qb_default, g_attn_qb_force;   // k_attention<..., QB>: QK^T on the bf16 matrix pipe (kernels.hip)
// deep levels (deep.hip)
struct DeepTile { int RT, NT; };                 // k_deep_conv<RT, NT>: 16 RT rows x 16 NT columns per workgroup
size_t deep_weight_floats(const DeepArgs& a, int NT);
bool deep_tile_for(const DeepArgs& a, DeepTile* t);      // (fills nothing in `a`; false: no instantiation for this row-group size)
}  // namespace mtv
#include <vector>
namespace mtv {
std::vector<int> deep_rowtab(const DeepArgs& a, DeepTile t);   // host: the row table of a configured conv (upload it, pass it as DeepArgs::rowtab)
bool deep_configure(DeepArgs& a, DeepTile* t, int max_ks = 8);   // picks row groups / K slices (<= max_ks) / tile; false: this conv stays on k_conv
size_t deep_smem_bytes(const DeepArgs& a, DeepTile t);
hipError_t launch_deep_conv(const DeepArgs& a, DeepTile t, hipStream_t s);
hipError_t launch_deep_repack(const float* W, int ldw, float* dst, const DeepArgs& a, int NT, hipStream_t s);   // legacy [krow][ldw] -> deep layout
hipError_t launch_deep_finalize(const DeepFinArgs& a, hipStream_t s);
bool deep_attn_configure(DeepAttnArgs& a);                // fills HPW / NC / group counts; false: this block keeps k_attention + a proj conv
hipError_t launch_deep_attn(const DeepAttnArgs& a, hipStream_t s);
hipError_t deep_init_attrs();
// whole attention block of a deep level in one launch (block.hip)
bool deep_block_configure(DeepBlockArgs& a, int force_cl = 0, int force_rq = 0, int max_wgs = 128);   // picks the cluster size (force_cl > 0: that one or nothing); false: keep the three-launch path
size_t deep_block_smem_bytes(const DeepBlockArgs& a);
bool deep_block_launchable(const DeepBlockArgs& a);       // the configuration checks shared by deep_block_configure, launch_deep_block and mtv_selftest_block
size_t deep_block_part_floats(const DeepBlockArgs& a);    // scratch sizes of a configured block
size_t deep_block_qkv_floats(const DeepBlockArgs& a);
size_t deep_block_stg_floats(const DeepBlockArgs& a);
hipError_t launch_deep_block(const DeepBlockArgs& a, hipStream_t s);
hipError_t deep_block_init_attrs();
hipError_t launch_linear(const LinearArgs& a, hipStream_t s);
hipError_t launch_time_sinusoid(const int64_t* t, const float* freqs, float* out, int B, int half, hipStream_t s);
hipError_t launch_pack_input(const float* x, const float* cond, const float* image_cond, int ic_len,
                             float* out, int B, int L, int RR, hipStream_t s);
// sampler set-up (once per mtv_ddim_sample call; the steps themselves are UNet launches only)
hipError_t launch_step_sinusoid(const DdimStep* steps, int n_steps, const float* freqs, float* out, int half, hipStream_t s);
hipError_t launch_linear_rows(const LinearArgs& a, hipStream_t s);   // k_linear for many rows: W read once per 8 rows
hipError_t launch_ddim_init(const DdimFuse* f, hipStream_t s);
hipError_t launch_repack_conv(const float* src, float* dst, int N, int C, int ntaps, int ld, hipStream_t s);
hipError_t launch_repack_pw(const float* src, float* dst, int N, int C, hipStream_t s);      // [N][C] -> ConvArgs::Wpk layout
// ---- autoencoder kernels (ae.hip) ----
hipError_t launch_repack_qkv(const float* src, float* dst, int H, int d, int C, int ld, hipStream_t s);
hipError_t launch_repeat(const float* src, float* dst, int n, int rep, hipStream_t s);
hipError_t launch_repack_heads(const float* src, float* dst, int H, int d, int C, int ld, int nparts, int part, hipStream_t s);   // xattn.hip

}  // namespace mtv
