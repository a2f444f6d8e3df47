// hm_common.h -- shared device/host helpers for the homan_amd HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HM_OK 0
#define HM_ERR_BAD_ARG (-1)
#define HM_ERR_LAUNCH (-2)
#define HM_ERR_UNSUPPORTED (-3)

#define HM_WAVE 64

// First statement of the small, latency-bound kernels: raise the wave's issue priority.  They share the SIMDs with the
// persistent edge-sweep waves (older, and issue-bound), and at equal priority a lone young wave got ~1/5 of the issue
// slots: the hand-side kernels ran 2-5x longer whenever they overlapped the sweep.
#ifndef HM_PRIO
#define HM_PRIO 3
#endif
#define HM_LATENCY_KERNEL() __builtin_amdgcn_s_setprio(HM_PRIO)
#ifdef HM_CHAIN_STAMPS      // debug build: first-start / last-end wall clock of a few small kernels (one array per translation unit)
static __device__ unsigned long long g_chain_ts[8];
#define HM_STAMP_START(k) do { if (threadIdx.x == 0) atomicMin(&g_chain_ts[2 * (k)], (unsigned long long)wall_clock64()); } while (0)
#define HM_STAMP_END(k) do { if (threadIdx.x == 0) atomicMax(&g_chain_ts[2 * (k) + 1], (unsigned long long)wall_clock64()); } while (0)
#else
#define HM_STAMP_START(k)
#define HM_STAMP_END(k)
#endif
// the kernels of the hand-side chain that are long enough to matter to whatever they overlap (MANO backward, pair terms),
// and the three heavy kernels of the silhouette chain: separately tunable (A/B builds)
#ifndef HM_HAND_PRIO
#define HM_HAND_PRIO HM_PRIO
#endif
#define HM_HAND_KERNEL() __builtin_amdgcn_s_setprio(HM_HAND_PRIO)
#ifdef HM_CHAIN_PRIO
#define HM_CHAIN_KERNEL() __builtin_amdgcn_s_setprio(HM_CHAIN_PRIO)
#else
#define HM_CHAIN_KERNEL()
#endif

// LDS ballast per kernel family (hm_tune_lds_pad, geometry.hip): unused dynamic LDS that caps how many workgroups of a small,
// latency-bound kernel fit on a CU next to the heavy kernel it overlaps (they would otherwise hold every wave slot while
// they wait on memory).  Scheduling only.
enum { HM_PAD_MANO_FWD = 0, HM_PAD_MANO_BWD = 1, HM_PAD_SMALL_LOSSES = 2, HM_PAD_PAIR_TERMS = 3, HM_PAD_RIGID_BWD = 4, HM_PAD_FAMILIES = 8 };
extern thread_local int g_hm_lds_pad[HM_PAD_FAMILIES];      // (launch hints are per calling thread)

#define HM_CHECK_ARG(cond) \
    do {                   \
        if (!(cond)) return HM_ERR_BAD_ARG; \
    } while (0)

// ---- clip batches (the *_clips entry points) --------------------------------------------------------------------
// Several independent clips of equal length are optimised by ONE launch per kernel: the frame axis of every per-frame
// tensor is the concatenation of the clips (clip c = frames [c*clip_len, (c+1)*clip_len)), every per-clip scalar
// (intrinsic scales, normalisers, loss values, reduction tickets / partial records) becomes an array with one entry per
// clip, and per-clip outputs are written `out_stride` floats apart (row c of the caller's (clips, n) table of values).
// clip_len == 0 means "the N frames are one clip" (the plain entry points).  Each clip's reductions are formed by its
// own workgroups in the order a single-clip launch uses, so a batched launch returns bit-identical per-clip results.
#define HM_CLIP_LEN_OK(N, clip_len) ((clip_len) == 0 || ((clip_len) > 0 && (N) % (clip_len) == 0))
#define HM_RED_WS_FLOATS 576            // per-clip slice of a reduce workspace: 512 partials + ticket word at 512

static inline int hm_launch_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HM_OK : HM_ERR_LAUNCH;
}

static inline int hm_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- wave-level reductions (64 lanes) on DPP: no LDS crossbar round trips (a __shfl_xor butterfly is six dependent
// ds_bpermute, ~100 cycles each; the DPP forms below are plain VALU moves).  Fixed, deterministic combination order:
// quad swap, pair swap, row_shr:4, row_shr:8 (row total in lanes 12-15 of every 16-lane row), row_bcast:15 into rows
// 1 and 3, row_bcast:31 into rows 2 and 3; lane 63 then holds the wave total, which readlane broadcasts.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float hm_dpp(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float hm_wave_sum(float v)
{
    v += hm_dpp<0xb1, 0xf>(0.f, v);      // quad_perm [1,0,3,2]
    v += hm_dpp<0x4e, 0xf>(0.f, v);      // quad_perm [2,3,0,1]
    v += hm_dpp<0x114, 0xf>(0.f, v);     // row_shr:4
    v += hm_dpp<0x118, 0xf>(0.f, v);     // row_shr:8
    v += hm_dpp<0x142, 0xa>(0.f, v);     // row_bcast:15 -> rows 1,3
    v += hm_dpp<0x143, 0xc>(0.f, v);     // row_bcast:31 -> rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float hm_wave_min(float v)
{
    v = fminf(v, hm_dpp<0xb1, 0xf>(v, v));
    v = fminf(v, hm_dpp<0x4e, 0xf>(v, v));
    v = fminf(v, hm_dpp<0x114, 0xf>(v, v));
    v = fminf(v, hm_dpp<0x118, 0xf>(v, v));
    v = fminf(v, hm_dpp<0x142, 0xa>(v, v));
    v = fminf(v, hm_dpp<0x143, 0xc>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float hm_wave_max(float v)
{
    v = fmaxf(v, hm_dpp<0xb1, 0xf>(v, v));
    v = fmaxf(v, hm_dpp<0x4e, 0xf>(v, v));
    v = fmaxf(v, hm_dpp<0x114, 0xf>(v, v));
    v = fmaxf(v, hm_dpp<0x118, 0xf>(v, v));
    v = fmaxf(v, hm_dpp<0x142, 0xa>(v, v));
    v = fmaxf(v, hm_dpp<0x143, 0xc>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in every thread.
// `red` must hold >= 16 floats of LDS.  Deterministic: fixed tree.
__device__ __forceinline__ float hm_block_sum(float v, float* red)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = hm_wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
// N block-wide sums with two barriers in all (hm_block_sum costs two per value).  `red` must hold >= 16 * N floats;
// results valid in every thread, same fixed combination order as hm_block_sum.
template <int N>
__device__ __forceinline__ void hm_block_sum_n(float (&v)[N], float* red)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = hm_wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) red[k * 16 + w] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float t = 0.f;
        for (int i = 0; i < nw; ++i) t += red[k * 16 + i];
        v[k] = t;
    }
}
__device__ __forceinline__ float hm_block_min(float v, float* red)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = hm_wave_min(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fminf(t, red[i]);
    return t;
}
__device__ __forceinline__ float hm_block_max(float v, float* red)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = hm_wave_max(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---- single-launch grid reduction ("last block finishes") -----------------------------------------
// Every block stores its partial record, then takes a ticket; the block that draws the last ticket
// re-reads all records in index order (deterministic) and finishes.  Cross-workgroup visibility: the per-XCD L2s are
// not coherent with each other, but an agent-scope RELEASE FENCE is the wrong tool here -- it writes back every dirty
// line of the L2 (buffer_wbl2), once per workgroup, while the rasteriser next door is filling that L2 with its own
// output (measured: the small reductions and the rasteriser both ran ~1.7x slower when overlapped).  Instead the
// records themselves travel at agent scope: hm_partial_store (write-through, sc1) by the producer, s_waitcnt vmcnt(0)
// before the ticket, hm_partial_load (sc1, never served from a stale L2 line) by the finishing block.  Records MUST be
// written with hm_partial_store by the thread that calls the ticket (thread 0) and read with hm_partial_load.
// `counter` must be zero on entry; the last block resets it, so one zero-initialised word serves every launch on the
// stream.
// Usage:   if (hm_last_block(counter, nblocks, &s_flag)) { ...read partials, write result... }
__device__ __forceinline__ void hm_partial_store(float* p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float hm_partial_load(const float* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool hm_last_block(unsigned int* counter, unsigned int nblocks, int* s_flag)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int ticket = atomicAdd(counter, 1u);
        const int last = (ticket == nblocks - 1u);
        if (last) atomicExch(counter, 0u);
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0;
}


// deterministic parallel sum of `n` partial values with stride `stride` (all threads of the LAST block call it)
__device__ __forceinline__ float hm_last_block_sum(const float* partials, int n, int stride, float* red)
{
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += hm_partial_load(partials + (long)i * stride);
    return hm_block_sum(a, red);
}

// ---- sin / cos of an fp32 angle as a DEFINED function: argument reduction by pi/2 and the fdlibm kernel polynomials in
// double, one IEEE operation at a time, the results rounded to fp32.  libm's / OCML's sinf and cosf agree with it to an ulp, but
// not with each other in the last bit - and the CPU oracle (oracle/csrc/lbs_exact.c, same operations) has to produce the hand's
// Rodrigues rotations bit for bit.  |a| up to ~1e5 (two-term Cody-Waite reduction); joint angles are a few radians.
__host__ __device__ __forceinline__ void hm_sincos(float af, float* sn, float* cs)
{
    const double a = (double)af;
    const double k = __builtin_rint(a * 0.63661977236758138243);                  // 2 / pi
    double r = a - k * 1.57079632673412561417e+00;                                 // pi/2, upper 33 bits
    r = r - k * 6.07710050650619224932e-11;                                        // pi/2 - upper part
    const double z = r * r;
    // sin(r) on [-pi/4, pi/4]
    double ps = 1.58969099521155010221e-10;
    ps = -2.50507602534068634195e-08 + z * ps;
    ps = 2.75573137070700676789e-06 + z * ps;
    ps = -1.98412698298579493134e-04 + z * ps;
    ps = 8.33333333332248946124e-03 + z * ps;
    ps = -1.66666666666666324348e-01 + z * ps;
    const double s = r + (r * z) * ps;
    // cos(r)
    double pc = -1.13596475577881948265e-11;
    pc = 2.08757232129817482790e-09 + z * pc;
    pc = -2.75573143513906633035e-07 + z * pc;
    pc = 2.48015872894767294178e-05 + z * pc;
    pc = -1.38888888888741095749e-03 + z * pc;
    pc = 4.16666666666666019037e-02 + z * pc;
    const double c = (1.0 - 0.5 * z) + (z * z) * pc;
    const int q = (int)((long long)k) & 3;
    const double sv = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    const double cv = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
    *sn = (float)sv;
    *cs = (float)cv;
}

// ---- tanh of an fp32 argument as a DEFINED function, like hm_sincos: e = exp(-2|x|) by a ln2 reduction and the fdlibm exp
// kernel in double, tanh = (1 - e) / (1 + e) in double, rounded to fp32 (OCML's and glibc's tanhf differ in the last bit; the
// oracle's written-out contact term, oracle/csrc/lbs_exact.c oc_tanh, evaluates the same operations).
__host__ __device__ __forceinline__ float hm_tanh(float xf)
{
    const double ax = xf < 0.f ? -(double)xf : (double)xf;
    if (ax > 20.0) return xf < 0.f ? -1.0f : 1.0f;
    if (ax < 0.01) {        // (1 - e loses the small argument: odd series, next term 62/2835 x^9)
        const double q = ax * ax;
        const double sm = ax * (1.0 + q * (-3.33333333333333314830e-01 + q * (1.33333333333333331483e-01 + q * -5.39682539682539708542e-02)));
        return (float)(xf < 0.f ? -sm : sm);
    }
    const double t = -2.0 * ax;
    const double k = __builtin_rint(t * 1.44269504088896338700e+00);
    const double r = (t - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10;
    const double z = r * r;
    double p = 4.13813679705723846039e-08;
    p = -1.65339022054652515390e-06 + z * p;
    p = 6.61375632143793436117e-05 + z * p;
    p = -2.77777777770155933842e-03 + z * p;
    p = 1.66666666666666019037e-01 + z * p;
    const double c = r - z * p;
    const double er = 1.0 - ((r * c) / (c - 2.0) - r);
    const double e = __builtin_ldexp(er, (int)k);
    const double th = (1.0 - e) / (1.0 + e);
    return (float)(xf < 0.f ? -th : th);
}

// ---- logistic function 1 / (1 + exp(-x)) of an fp32 argument as a DEFINED function (same exp kernel as hm_tanh), |x| < ~700:
// the ordinal depth term's gradient (oracle/csrc/lbs_exact.c oc_sigmoid evaluates the same operations).
__host__ __device__ __forceinline__ float hm_sigmoid(float xf)
{
    const double t = -(double)xf;
    const double k = __builtin_rint(t * 1.44269504088896338700e+00);
    const double r = (t - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10;
    const double z = r * r;
    double p = 4.13813679705723846039e-08;
    p = -1.65339022054652515390e-06 + z * p;
    p = 6.61375632143793436117e-05 + z * p;
    p = -2.77777777770155933842e-03 + z * p;
    p = 1.66666666666666019037e-01 + z * p;
    const double c = r - z * p;
    const double er = 1.0 - ((r * c) / (c - 2.0) - r);
    const double e = __builtin_ldexp(er, (int)k);
    return (float)(1.0 / (1.0 + e));
}

// ---- rot6d (3x2 row-major, reference homan/utils/geometry.py:9-27) -> rotation matrix (3x3 row-major)
__device__ __forceinline__ void rot6d_to_mat(const float* r6 /*3x2 row-major*/, float* R /*3x3 row-major*/)
{
    const float a1[3] = {r6[0], r6[2], r6[4]}, a2[3] = {r6[1], r6[3], r6[5]};
    const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float nu = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    const float b2[3] = {u[0] / nu, u[1] / nu, u[2] / nu};
    const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
#pragma unroll
    for (int i = 0; i < 3; ++i) { R[3 * i] = b1[i]; R[3 * i + 1] = b2[i]; R[3 * i + 2] = b3[i]; }
}

// dL/dR (3x3 row-major) -> dL/drot6d (3x2 row-major)
__device__ __forceinline__ void rot6d_backward(const float* r6, const float* dR, float* dr6)
{
    const float a1[3] = {r6[0], r6[2], r6[4]}, a2[3] = {r6[1], r6[3], r6[5]};
    const float n1r = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
    const float n1 = fmaxf(n1r, 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float nur = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const float nu = fmaxf(nur, 1e-12f);
    const float b2[3] = {u[0] / nu, u[1] / nu, u[2] / nu};
    float db1[3] = {dR[0], dR[3], dR[6]}, db2[3] = {dR[1], dR[4], dR[7]};
    const float db3[3] = {dR[2], dR[5], dR[8]};
    // b3 = b1 x b2
    db1[0] += b2[1] * db3[2] - b2[2] * db3[1];
    db1[1] += b2[2] * db3[0] - b2[0] * db3[2];
    db1[2] += b2[0] * db3[1] - b2[1] * db3[0];
    db2[0] += db3[1] * b1[2] - db3[2] * b1[1];
    db2[1] += db3[2] * b1[0] - db3[0] * b1[2];
    db2[2] += db3[0] * b1[1] - db3[1] * b1[0];
    // b2 = u / max(|u|, eps)
    float du[3];
    if (nur > 1e-12f) {
        const float s = b2[0] * db2[0] + b2[1] * db2[1] + b2[2] * db2[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) du[i] = (db2[i] - b2[i] * s) / nu;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) du[i] = db2[i] / nu;
    }
    // u = a2 - (b1.a2) b1
    float da2[3] = {du[0], du[1], du[2]};
    const float dd = -(du[0] * b1[0] + du[1] * b1[1] + du[2] * b1[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) { db1[i] += -d * du[i] + dd * a2[i]; da2[i] += dd * b1[i]; }
    float da1[3];
    if (n1r > 1e-12f) {
        const float s = b1[0] * db1[0] + b1[1] * db1[1] + b1[2] * db1[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) da1[i] = (db1[i] - b1[i] * s) / n1;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) da1[i] = db1[i] / n1;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { dr6[2 * i] = da1[i]; dr6[2 * i + 1] = da2[i]; }
}

// ---- order-independent sums ("exact accumulation").  Every gradient term on the object's chain is rounded to a multiple of
// the quantum q = 2^log2q (a value of `magic` = 1.5 * 2^(52 + log2q): (x + magic) - magic in double is x rounded to the grid,
// round-to-nearest-even) and summed in DOUBLE: sums of multiples of q are exact while they stay below 2^53 q, so they do not
// depend on the order in which lanes, waves, units or workgroups add them - the result is a function of the SET of terms.
// The CPU oracle (oracle/csrc/objchain.c) forms the same set with the same per-term arithmetic, which makes the pseudo-
// gradient - and with it the free-running object trajectory of a fit - bit-equal on both sides.  Default grid 2^-44
// (5.7e-14: below the float ulp of the per-corner sums of a normalised silhouette loss; exact up to |sum| < 512); a
// caller whose gradients are O(1) and larger (pose initialisation) passes a coarser one.
#define HM_SUM_LOG2Q_DEFAULT (-44)
__host__ __device__ __forceinline__ double hm_sum_magic(int log2q)
{
    return 6755399441055744.0 /* 1.5 * 2^52 */ * __builtin_ldexp(1.0, log2q ? log2q : HM_SUM_LOG2Q_DEFAULT);
}
__device__ __forceinline__ double hm_quant(float x, double magic) { return ((double)x + magic) - magic; }

// block-wide sums of N doubles (blockDim.x <= 1024, multiple of 64); results valid in every thread.  `red` must hold >= 16 * N
// doubles.  Meant for EXACT sums (multiples of one quantum, see hm_quant): the combination order is then immaterial.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double hm_dpp_f64(double v)          // lanes the DPP control leaves unwritten read 0.0
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double hm_wave_sum_f64(double v)       // same DPP tree as hm_wave_sum: plain VALU moves, no LDS
{
    v += hm_dpp_f64<0xb1, 0xf>(v);
    v += hm_dpp_f64<0x4e, 0xf>(v);
    v += hm_dpp_f64<0x114, 0xf>(v);
    v += hm_dpp_f64<0x118, 0xf>(v);
    v += hm_dpp_f64<0x142, 0xa>(v);
    v += hm_dpp_f64<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
template <int N>
__device__ __forceinline__ void hm_block_sum_n_f64(double (&v)[N], double* red)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = hm_wave_sum_f64(v[k]);
    if (nw == 1) return;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) red[k * 16 + w] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double t = 0.0;
        for (int i = 0; i < nw; ++i) t += red[k * 16 + i];
        v[k] = t;
    }
}

// weighted per-vertex gradient terms of a rigid backward (see k_rigid_bwd, geometry.hip): up to five, NULL = skipped
struct RigidTerms { const float* p[5]; float w[5]; };
