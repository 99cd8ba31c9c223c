// Do a wave's softmax VALU instructions overlap the f32 MFMAs of the OTHER wave on its SIMD (or its own)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_valu.hip -o mfma_valu
// Models one key block of k_attention<16,4,2> per loop iteration and wave: 16 QK^T MFMAs (4 tiles x 4 steps), a softmax
// pass over the 16 scores of every lane (max, 2 lane swaps, 17 exp2, sums), 16 PV MFMAs that take the probabilities as
// B operand.  256 workgroups x 8 waves (2 waves per SIMD, like the level-0 attentions of the B = 1 step).
//   mode 0: MFMAs only          mode 1: VALU only
//   mode 2: phased (QK, softmax, PV), workgroup barrier every iteration (what the kernel does)
//   mode 3: phased, no barrier
//   mode 4: phased, barrier, the odd waves of a SIMD (wave >= 4) start half an iteration late (one extra softmax first)
//   mode 5: two half-blocks software-pipelined in the wave (QK of half B || softmax of half A, PV of A || softmax of B)
//   modes 8-10: the conv K loop's mix (one 16-channel chunk of a k_conv<1,4,*> wave per iteration): SiLU of 4 elements +
//   16 f32 MFMAs, against the same products as a 3-term bf16 split (6 products per pair, v_mfma_f32_16x16x16_bf16)
//   modes 11-13: a 32-channel step with gfx950's v_mfma_f32_16x16x32_bf16 and the split through v_cvt_pk_bf16_f32
// Prints cycles per iteration of wave 0 (s_memtime) and the wall-clock rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float swapmax16(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float swapmax32(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int NT>
__device__ __forceinline__ void qk(f32x4 (&st)[NT], const float (&kq)[4][4], const float (&q)[4]) {
#pragma unroll
    for (int w = 0; w < NT; ++w) st[w] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int w = 0; w < NT; ++w) st[w] = __builtin_amdgcn_mfma_f32_16x16x4f32(kq[w][e], q[e], st[w], 0, 0, 0);
}
template <int NT>
__device__ __forceinline__ float soft(f32x4 (&st)[NT], float& m, float& l) {
    float mx = st[0][0];
#pragma unroll
    for (int w = 0; w < NT; ++w)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[w][r]);
    mx = swapmax16(mx);
    mx = swapmax32(mx);
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int w = 0; w < NT; ++w)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(st[w][r] - mn);
            st[w][r] = p;
            ps += p;
        }
    l = l * alpha + ps;
    return alpha;
}
template <int NT>
__device__ __forceinline__ void pv(f32x4& o, const f32x4 (&st)[NT], const float (&v)[4][4], float alpha) {
    o *= alpha;
#pragma unroll
    for (int w = 0; w < NT; ++w)
#pragma unroll
        for (int s = 0; s < 4; ++s) o = __builtin_amdgcn_mfma_f32_16x16x4f32(v[w][s], st[w][s], o, 0, 0, 0);
}

// ---- the conv K loop's mix: one 16-channel chunk of a k_conv<1,4,*> wave per iteration = the SiLU transform of 4 A elements
// (fma, exp2, add, rcp, mul each) + 16 f32 MFMAs (mode 8) -- or the same products as a 3-term bf16 split: A split in
// registers (two subtract/convert rounds), W pre-split, 6 x 4 = 24 v_mfma_f32_16x16x16_bf16 (modes 9: MFMAs only, 10: all)
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504f * v)); }
__device__ __forceinline__ short bf16_rn(float x) {   // round-to-nearest-even bf16 of a finite float, as bits
    const unsigned u = __float_as_uint(x);
    return (short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_up(short h) { return __uint_as_float(((unsigned)(unsigned short)h) << 16); }

template <int MODE>
__global__ __launch_bounds__(512) void kc(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float x[4], w[4][4];
    s16x4 wb[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x[e] = 0.01f * (float)((lane * 7 + e) % 13) - 0.05f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) w[e][nb] = 0.02f * (float)((lane * 5 + nb + e * 3) % 11) - 0.1f;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) wb[t][nb] = s16x4{(short)(lane + t), (short)(nb * 3 + 1), (short)(t * 5 + 2), (short)(lane ^ nb)};
    f32x4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x[e] += 1e-6f;
            y[e] = (MODE == 9) ? x[e] : silu_fast(fmaf(x[e], 1.01f, 0.02f));
        }
        if constexpr (MODE == 8) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[s2], w[s2][nb], acc[nb], 0, 0, 0);
        } else {
            s16x4 ab[3];
            if constexpr (MODE == 10) {   // 3-term split of the lane's 4 A values
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const short h0 = bf16_rn(y[e]);
                    const float r1 = y[e] - bf16_up(h0);
                    const short h1 = bf16_rn(r1);
                    const float r2 = r1 - bf16_up(h1);
                    ab[0][e] = h0; ab[1][e] = h1; ab[2][e] = bf16_rn(r2);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 3; ++t) ab[t] = s16x4{(short)__float_as_uint(y[0]), (short)t, (short)(t + 1), (short)lane};
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[0], wb[0][nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[0], wb[1][nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[1], wb[0][nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[0], wb[2][nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[1], wb[1][nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab[2], wb[0][nb], acc[nb], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

// ---- the same for a 32-channel step: 8 SiLU + 32 f32 MFMAs (mode 11) against 8 SiLU + the 3-term split through the
// hardware conversion (v_cvt_pk_bf16_f32) + 24 v_mfma_f32_16x16x32_bf16 (mode 12); mode 13: those 24 MFMAs alone
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(512) void kd(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float x[8], w[8][4];
    bf16x8 wb[3][4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        x[e] = 0.01f * (float)((lane * 7 + e) % 13) - 0.05f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) w[e][nb] = 0.02f * (float)((lane * 5 + nb + e * 3) % 11) - 0.1f;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int e = 0; e < 8; ++e) wb[t][nb][e] = (__bf16)(0.01f * (float)((lane + t * 3 + nb * 5 + e) % 17) - 0.08f);
    f32x4 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[e] += 1e-6f;
            y[e] = (MODE == 13) ? x[e] : silu_fast(fmaf(x[e], 1.01f, 0.02f));
        }
        if constexpr (MODE == 11) {
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[s2], w[s2][nb], acc[nb], 0, 0, 0);
        } else {
            bf16x8 ab[3];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (MODE == 12) {
                    const __bf16 h0 = (__bf16)y[e];
                    const float r1 = y[e] - (float)h0;
                    const __bf16 h1 = (__bf16)r1;
                    const float r2 = r1 - (float)h1;
                    ab[0][e] = h0; ab[1][e] = h1; ab[2][e] = (__bf16)r2;
                } else {
                    ab[0][e] = (__bf16)y[e]; ab[1][e] = ab[0][e]; ab[2][e] = ab[0][e];
                }
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[0], wb[0][nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[0], wb[1][nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[1], wb[0][nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[0], wb[2][nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[1], wb[1][nb], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[2], wb[0][nb], acc[nb], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float kq[4][4], v[4][4], q[4];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            kq[w][e] = 0.01f * (float)((lane * 7 + w * 3 + e) % 13) - 0.05f;
            v[w][e] = 0.02f * (float)((lane * 5 + w + e * 3) % 11) - 0.1f;
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) q[e] = 0.03f * (float)((lane + e) % 9) - 0.1f;
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -1e30f, l = 0.f;
    f32x4 st[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) st[w] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
    if (MODE == 4 && wave >= 4) {   // skew: the second wave of every SIMD does one softmax before entering the loop
        const float al = soft<4>(st, m, l);
        o *= al;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        q[0] += 1e-6f;   // (keeps the iterations distinct)
        if constexpr (MODE == 0) {
            qk<4>(st, kq, q);
            pv<4>(o, st, v, 1.0f);
        } else if constexpr (MODE == 1) {
            const float al = soft<4>(st, m, l);
            o *= al;
#pragma unroll
            for (int w = 0; w < 4; ++w) st[w] = st[w] * 0.5f + o;
        } else if constexpr (MODE == 2 || MODE == 3 || MODE == 4) {
            qk<4>(st, kq, q);
            const float al = soft<4>(st, m, l);
            pv<4>(o, st, v, al);
            if constexpr (MODE != 3) __syncthreads();
        } else {
            f32x4 sa[2], sb[2];
            const float ka[4][4] = {{kq[0][0], kq[0][1], kq[0][2], kq[0][3]}, {kq[1][0], kq[1][1], kq[1][2], kq[1][3]}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const float kb[4][4] = {{kq[2][0], kq[2][1], kq[2][2], kq[2][3]}, {kq[3][0], kq[3][1], kq[3][2], kq[3][3]}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const float va[4][4] = {{v[0][0], v[0][1], v[0][2], v[0][3]}, {v[1][0], v[1][1], v[1][2], v[1][3]}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const float vb[4][4] = {{v[2][0], v[2][1], v[2][2], v[2][3]}, {v[3][0], v[3][1], v[3][2], v[3][3]}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            qk<2>(sa, ka, q);
            qk<2>(sb, kb, q);
            const float a1 = soft<2>(sa, m, l);
            pv<2>(o, sa, va, a1);
            const float a2 = soft<2>(sb, m, l);
            pv<2>(o, sb, vb, a2);
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = o[0] + o[1] + o[2] + o[3] + l + m + st[0][0];
}

template <int MODE>
static void run(const char* name, float* out, unsigned long long* cyc, int waves_per_wg) {
    const int iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves_per_wg), 0, 0, out, cyc, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves_per_wg), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-58s waves/WG %d: %7.1f ticks/iter (s_memtime), %6.3f us/iter wall\n", name, waves_per_wg, (double)h / iters, ms * 1e3 / iters);
}

template <int MODE>
static void runc(const char* name, float* out, unsigned long long* cyc, int waves_per_wg) {
    const int iters = 8000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kc<MODE>, dim3(256), dim3(64 * waves_per_wg), 0, 0, out, cyc, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kc<MODE>, dim3(256), dim3(64 * waves_per_wg), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-58s waves/WG %d: %7.1f ticks/iter (s_memtime), %6.3f us/iter wall\n", name, waves_per_wg, (double)h / iters, ms * 1e3 / iters);
}

template <int MODE>
static void rund(const char* name, float* out, unsigned long long* cyc, int waves_per_wg) {
    const int iters = 8000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kd<MODE>, dim3(256), dim3(64 * waves_per_wg), 0, 0, out, cyc, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kd<MODE>, dim3(256), dim3(64 * waves_per_wg), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-58s waves/WG %d: %7.1f ticks/iter (s_memtime), %6.3f us/iter wall\n", name, waves_per_wg, (double)h / iters, ms * 1e3 / iters);
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 64));
    for (int wv : {4, 8}) {
        run<0>("0 MFMA only (32 per iteration)", out, cyc, wv);
        run<1>("1 VALU only (softmax of 16 scores per lane)", out, cyc, wv);
        run<2>("2 phased QK / softmax / PV, barrier", out, cyc, wv);
        run<3>("3 phased, no barrier", out, cyc, wv);
        run<4>("4 phased, barrier, second wave of a SIMD skewed", out, cyc, wv);
        run<5>("5 two half blocks pipelined in the wave, barrier", out, cyc, wv);
        runc<8>("8 conv chunk: SiLU of 4 elements + 16 f32 MFMAs", out, cyc, wv);
        runc<9>("9 conv chunk: 24 bf16 MFMAs (3-term split, 6 products)", out, cyc, wv);
        runc<10>("10 conv chunk: SiLU + A split + 24 bf16 MFMAs", out, cyc, wv);
        rund<11>("11 32 channels: 8 SiLU + 32 f32 MFMAs", out, cyc, wv);
        rund<13>("13 32 channels: 24 bf16 16x16x32 MFMAs alone", out, cyc, wv);
        rund<12>("12 32 channels: 8 SiLU + cvt split + 24 bf16 16x16x32", out, cyc, wv);
    }
    return 0;
}
