// Host-side structures shared by plan.hip (UNet plans, C ABI) and ae.hip (autoencoder plans): the context (device
// buffers, weight slots, tile table, launch plans) and the plan / op records.  Internal; gfx950 only.
#pragma once
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mtv_hip.h"
#include "mtv_internal.h"

using namespace mtv;

// error text of the calling thread (mtv_last_error); defined in plan.hip
int fail(int code, const std::string& msg);
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(MTV_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));              \
    } while (0)


struct Level {
    int r, t, L, b1, b2;
    SegInfo seg() const { return SegInfo{b1, b2, L}; }
};

// how mtv_load_weight stores a tensor: COPY as is; CONV: OIHW / [O][I][1] / [O][I] -> [tap][I][ld] (output channels
// contiguous); QKV_HEADS: a Linear [3*H*d][I] whose rows are (q|k|v, head, d) -> [I][ld] with columns (head, q|k|v, d),
// the layout k_attention reads (aux = d); REPEAT: every element repeated aux times (a per-channel bias expanded over
// the aux = patch*patch pixels of a ConvTranspose output)
enum Role { ROLE_COPY = 0, ROLE_CONV = 1, ROLE_QKV_HEADS = 2, ROLE_REPEAT = 3, ROLE_KV_HEADS = 4 };   // KV_HEADS: a Linear [H*d][I] -> columns
                                                                                               // (head, k|v, d) of a shared [I][ld] matrix (aux = 2 d + part)

struct WSlot {
    std::string key;
    std::vector<int64_t> shape;
    Role role;
    float* dst;
    int ld;         // ROLE_CONV: leading dimension (padded output channels)
    bool loaded;
    int aux = 0;    // ROLE_QKV_HEADS: head dim; ROLE_REPEAT: repeat count
    float* dst2 = nullptr;   // ROLE_CONV, 1x1 only: a second copy in the checkpoint's own layout [N][C] (k_lin, lin.hip)
    float* dst3 = nullptr;   // ROLE_CONV, 1x1 only, N and C multiples of 16: a third copy in k_conv_pw's lane-linear layout (ConvArgs::Wpk)
};

struct Tens {
    float* p = nullptr;
    int C = 0;
    int lvl = 0;
    int ks = 1;                // deep levels (deep.hip): the tensor is the sum of `ks` partial slabs, `slab` floats apart
    unsigned slab = 0;
};

struct ResDesc {
    int cin, cout, updown;   // updown: 0 none, 1 down, 2 up
    int film_off;
};
struct Layer {
    int type;                // 0 stem conv, 1 resblock, 2 attention (2-D, per plane)
    ResDesc res;
    int c;
    std::string pre;
};
struct Stage {
    std::vector<Layer> layers;
    int attn1_c = 0;         // channels of the cross-plane AttentionBlock1D after the stage (0: Identity)
    std::string attn1_pre;
    std::string tap;
};

struct Op {
    std::function<hipError_t(hipStream_t)> run;
    std::string name;
    double flops = 0.0;    // algorithmic FLOPs of this launch (at the plan's batch size)
    double bytes = 0.0;    // algorithmic HBM bytes: weights once + activations in/out once
};

struct ConvOp {                 // one convolution of the plan; args/tile are patched after creation
    ConvArgs a;                 // (statistics targets, slab, auto-tuned tile), so launches read them late
    ConvTile t;
    std::string base_name;
    int op_index = -1;
    // set when the launch the plan really issues for a tile is not launch_conv(a, t) alone (the autoencoder's ff2 GEMM: GEGLU folded into
    // the split pass on the split-bf16 pair, a separate k_geglu launch otherwise): the tuner times THIS
    std::function<hipError_t(const ConvArgs&, ConvTile, hipStream_t)> tune_launch;
};

struct DeepOp {                 // one K-sliced conv of the deep levels (deep.hip)
    DeepArgs a;
    DeepTile t;
};
struct FinOp {                  // slabs -> plain tensor (+ statistics for legacy consumers)
    DeepFinArgs a;
};
struct DeepAttnOp {             // attention core + proj_out of a deep level (k_deep_attn)
    DeepAttnArgs a;
};
struct StatSink {               // where add_stats() attaches the statistics targets of a tensor that left the deep region:
    StatOut* stat;              // the stat[2] / nstat of whichever op writes the plain copy (k_deep_finalize, or the in-launch
    int* nstat;                 // completion of the producing k_deep_conv / k_deep_attn); the op is kept alive by `owner`
    std::shared_ptr<void> owner;
};

// A plan is built per (batch size, mode).  FORWARD: one UNetModel.forward for arbitrary per-clip timesteps
// (time-embedding chain + input packing + UNet -> eps).  STEP0 / STEP1: one denoising step of the sampler, UNet
// launches ONLY -- FiLM rows come from a per-call table, the sample is packed by the previous step's head conv,
// the DDIM update runs in the head conv's epilogue; the two differ in the GroupNorm statistics arena they use
// (a step's head zeroes the other parity's arena, so no memset launch either).
enum Mode { MODE_FORWARD = 0, MODE_STEP0 = 1, MODE_STEP1 = 2, MODE_AE_DECODE = 3, MODE_AE_EXTRACT = 4 };

struct Plan {
    int B = 0;
    int mode = MODE_FORWARD;
    std::vector<std::shared_ptr<ConvOp>> convs;
    std::vector<std::pair<std::string, unsigned long long*>> attn_dbg;   // (diagnostic build: stamp buffers of the attention launches)
    bool tuned = false;
    size_t slab_floats = 0;        // floats of the split-K slab this plan's convs share (finish_split_k)
    std::vector<Op> ops;           // UNet forward
    hipGraphExec_t g_forward = nullptr;
};


enum CtxKind { CTX_UNET = 0, CTX_AE = 1, CTX_XATTN = 2 };

struct mtv_ctx {
    mtv_config cfg{};
    int device = 0;
    int kind = CTX_UNET;                        // which C-ABI family owns this context
    std::shared_ptr<void> ext;                  // CTX_AE: mtv_ae_config; CTX_XATTN: its weight / workspace record -- lives and dies
                                                // with the context (no process-global side tables, nothing to lock)
    std::vector<Level> lv;
    std::vector<Stage> inputs, outputs;
    Stage middle;
    int film_total = 0;
    int n_sites = 0;
    int emb_dim = 0;

    std::vector<WSlot> slots;
    std::map<std::string, int> slot_index;
    std::map<std::string, float*> bufs;         // named activation / weight buffers
    std::map<std::string, size_t> buf_floats;   // ... and the size each was allocated with
    std::map<std::string, std::pair<int, int>> taps;   // tap name -> (level, C) ; buffer = bufs["tap." + name]
    std::vector<void*> allocs;
    std::vector<int*> g3, gup3, gup1;           // gather tables per level
    bool geo_ok = true;                         // geo_source() reproduced every table on the host (mtv_create)
    double* stats = nullptr;                     // GN site arenas: [2 step parities][STAT_COPIES][sites][max_batch][3][32][2]
    size_t stats_bytes = 0;                     // bytes of ONE arena
    size_t stats_copy_doubles = 0;              // doubles per privatised copy of an arena
    int site_cursor = 0;
    int site_parity = 0;                        // arena the plan being built uses
    float* freqs = nullptr;
    // external-layout staging (channel-major) and sampler state
    float *xin = nullptr, *condin = nullptr, *icin = nullptr, *eps = nullptr;
    int64_t* tbuf = nullptr;
    // sampler state (mtv_ddim_sample): step table, per-call FiLM table, the head conv's hand-over records
    DdimStep* d_steps = nullptr;
    float *film_tab = nullptr, *sin_steps = nullptr, *e0_steps = nullptr, *e1_steps = nullptr;
    int d_steps_cap = 0;
    int* d_counter = nullptr;                   // [0] step index, [1] arrival counter of the head's workgroups
    DdimFuse* d_fuse = nullptr;                 // [2]: one per step parity
    char* h_pin[2] = {nullptr, nullptr};        // pinned upload staging (step table + hand-over records), double buffered
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    size_t pin_bytes = 0;
    unsigned pin_turn = 0;
    std::map<std::pair<int, int>, std::unique_ptr<Plan>> plans;   // (batch, mode)
    std::map<std::pair<int, int>, hipGraphExec_t> multi;         // (batch, M): M consecutive sampler steps (even, odd, even, ...) captured as ONE graph
    hipStream_t cap_stream = nullptr;
    bool eager = false;
    mtv_work work{};
    bool accounting = false;
    int* fault_h = nullptr;                      // host-mapped fault word of the in-launch hand-offs (block.hip) and its device address
    int* fault_d = nullptr;
    bool debug_fault_armed = false;              // mtv_debug_arm_fault: the next forward / sampler call ends with a launch that raises the fault word
    int resident_cus = 256;                      // CUs a launch of this context may count on being resident TOGETHER (ctx_init_common: the device's
                                                 // multiProcessorCount, or mtv_debug_resident_cus / MTV_RESIDENT_CUS): bounds every grid whose workgroups
                                                 // wait for each other inside the launch (k_deep_block clusters, tagged completion of deep tensors)
    float* staging = nullptr;
    size_t staging_floats = 0;
    // split-bf16 copies of conv / GEMM weight matrices (k_conv_x3): W [K][ld] f32 -> three bf16 planes, rebuilt after weight loads
    struct W3Info { void* p; size_t plane_bytes; int K, ld; bool dirty; };
    std::map<const float*, W3Info> w3;
    // deep-layout copies of conv matrices (k_deep_conv): one per (matrix, slicing), rebuilt after weight loads
    struct WDeepInfo { float* p; const float* W; int ldw; DeepArgs lay; int NT; bool dirty; };
    std::map<std::string, WDeepInfo> wdeep;
    std::map<std::string, std::pair<int, unsigned>> tap_slabs;   // tap name -> (slabs, floats between them) of a deep tensor
    const float* wdeep_for(const float* W, int ldw, const DeepArgs& lay, int NT) {
        char key[160];
        snprintf(key, sizeof key, "%p %d %d %d %d %d %d %d %d %d", (const void*)W, ldw, lay.KS, lay.CSm, lay.CSs, NT, lay.ntaps, lay.Cmain, lay.Cskip, lay.N);
        auto it = wdeep.find(key);
        if (it == wdeep.end()) {
            void* p = nullptr;
            if (dmalloc(&p, deep_weight_floats(lay, NT) * sizeof(float)) != MTV_OK) return nullptr;
            it = wdeep.emplace(key, WDeepInfo{(float*)p, W, ldw, lay, NT, true}).first;
            // a plan built lazily (first run at a new batch size) comes after check_ready's refresh: repack what the weight holds now
            if (launch_deep_repack(W, ldw, (float*)p, lay, NT, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return nullptr;
        }
        return it->second.p;
    }
    const void* w3_for(const float* W, int K, int ld, unsigned long long* plane_out) {
        if (K & 7) return nullptr;
        auto it = w3.find(W);
        if (it == w3.end()) {
            void* p = nullptr;
            const size_t plane = (size_t)K * ld * 2;
            if (dmalloc(&p, 3 * plane) != MTV_OK) return nullptr;
            it = w3.emplace(W, W3Info{p, plane, K, ld, true}).first;
            // a plan built lazily (first run at a new batch size) comes after check_ready's refresh: split what the weight holds now
            if (launch_split_w3(W, p, plane, 0, K, ld, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return nullptr;
        }
        if (it->second.K != K || it->second.ld != ld) return nullptr;
        *plane_out = it->second.plane_bytes;
        return it->second.p;
    }
    std::map<std::string, ConvTile> tune_cache;           // conv shape -> measured best tile
    void* flush = nullptr;                                // cache-flush scratch for cold auto-tune timing
    size_t flush_bytes = 0;
    bool tune_cache_loaded = false;                       // MTV_TUNE_CACHE=<file>: persisted across processes

    // ---------------------------------------------------------------- memory helpers
    int dmalloc(void** p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes ? bytes : 16);
        if (e != hipSuccess) return fail(MTV_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
        allocs.push_back(*p);
        return MTV_OK;
    }
    // named buffer, allocated (zero-filled) on first use.  Asking again for MORE than the first request is an error
    // (nullptr + mtv_last_error): captured graphs hold the pointer, so it can neither move nor be outgrown silently.
    float* buf(const std::string& name, size_t floats) {
        auto it = bufs.find(name);
        if (it != bufs.end()) {
            auto sz = buf_floats.find(name);
            if (sz != buf_floats.end() && floats > sz->second) {
                fail(MTV_ERR_STATE, "buffer '" + name + "' requested with " + std::to_string(floats) + " floats but allocated with " + std::to_string(sz->second));
                return nullptr;
            }
            return it->second;
        }
        void* p = nullptr;
        if (dmalloc(&p, floats * sizeof(float)) != MTV_OK) return nullptr;
        (void)hipMemset(p, 0, floats * sizeof(float));
        bufs[name] = (float*)p;
        buf_floats[name] = floats;
        return (float*)p;
    }
    float* act(const std::string& name, int lvl, int C) {   // [max_batch][L_lvl][C]
        float* p = buf("act." + name, (size_t)cfg.max_batch * lv[lvl].L * C);
        taps[name] = {lvl, C};            // every activation is retrievable by name (mtv_debug_tap)
        bufs["tap." + name] = p;
        tap_slabs.erase(name);            // (a deep plan built earlier may have registered this name as a slab tensor)
        return p;
    }
    WSlot* slot(const std::string& key, std::vector<int64_t> shape, Role role, float* dst, int ld) {
        auto it = slot_index.find(key);
        if (it != slot_index.end()) return &slots[it->second];
        slots.push_back(WSlot{key, std::move(shape), role, dst, ld, false});
        slot_index[key] = (int)slots.size() - 1;
        return &slots.back();
    }
    // plain-copied vector / matrix parameter
    float* wcopy(const std::string& key, std::vector<int64_t> shape) {
        size_t n = 1;
        for (auto d : shape) n *= (size_t)d;
        float* p = buf("w." + key, n);
        slot(key, shape, ROLE_COPY, p, 0);
        return p;
    }
    double* new_site() {
        double* p = stats + (size_t)site_parity * stats_copy_doubles * STAT_COPIES + (size_t)site_cursor * cfg.max_batch * 192;
        ++site_cursor;
        return p;
    }
    ~mtv_ctx() {     // every exit path of mtv_create / mtv_destroy ends here: nothing device-side outlives the context
        int cur = 0;
        const bool sw = hipGetDevice(&cur) == hipSuccess && cur != device;
        if (sw) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        for (auto& kv : plans)
            if (kv.second->g_forward) (void)hipGraphExecDestroy(kv.second->g_forward);
        plans.clear();
        for (auto& kv : multi)
            if (kv.second) (void)hipGraphExecDestroy(kv.second);
        multi.clear();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        for (void* p : allocs) (void)hipFree(p);
        if (staging) (void)hipFree(staging);
        if (fault_h) (void)hipHostFree(fault_h);
        free_step_tables();
        for (int i = 0; i < 2; ++i) {
            if (h_pin[i]) (void)hipHostFree(h_pin[i]);
            if (pin_ev[i]) (void)hipEventDestroy(pin_ev[i]);
        }
        if (sw) (void)hipSetDevice(cur);
    }
    void free_step_tables() {
        for (void* p : {(void*)d_steps, (void*)film_tab, (void*)sin_steps, (void*)e0_steps, (void*)e1_steps})
            if (p) (void)hipFree(p);
        d_steps = nullptr;
        film_tab = sin_steps = e0_steps = e1_steps = nullptr;
        d_steps_cap = 0;
    }
};

// ---- implemented in plan.hip, used by ae.hip ----
int autotune(mtv_ctx* c, Plan* p, hipStream_t s);          // measured tile per conv shape (committed table first)
int finish_split_k(mtv_ctx* c, Plan* p);                   // slab + arrival counters shared by the plan's split-K convs
int run_ops(mtv_ctx* c, Plan* p, hipStream_t s);
int capture(mtv_ctx* c, Plan* p, hipGraphExec_t* out);
constexpr long X3_MIN_ROWS = 1024;
bool x3_wanted(long rows);                                  // plan.hip: offer this conv to the split-bf16 kernels?
int check_ready(mtv_ctx* c, int batch);                     // (also refreshes the split-bf16 weight copies after a weight load)
int ctx_init_common(mtv_ctx* c);
void force_lds_tile(const ConvArgs& a, ConvTile* t);    // MTV_FORCE_LDS / MTV_FORCE_LIN testing aids
