// raster_common.h -- shared device helpers and records of the silhouette rasteriser's translation units
// (raster_setup / raster_fwd / raster_lines / raster_sweep / raster_depth / raster_api .hip; design notes: raster_api.hip).
#pragma once
#include "hm_common.h"
#include <type_traits>

#define HM_TILE 8          // output pixels per tile side
#define HM_STILE 16        // samples per tile side (2x SSAA)
#define RASTER_WAVES 4     // tiles per workgroup
#define STAGE_DW 20        // dwords per staged face in LDS (9 verts + 9 inverse + id + pad)

// output stores of the rasteriser's epilogue: -DRASTER_NT_STORES=1 makes them non-temporal (streamed past the L2: an A/B knob
// for the few microseconds of cache write-back between this launch and the next one of its chain)
typedef int hm_v2i __attribute__((ext_vector_type(2)));
typedef unsigned hm_v4u __attribute__((ext_vector_type(4)));
#ifndef RASTER_NT_STORES
#define RASTER_NT_STORES 0
#endif
__device__ __forceinline__ void hm_out_store(float* p, float v)
{
#if RASTER_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void hm_out_store2(int* p, int a, int b)
{
#if RASTER_NT_STORES
    hm_v2i v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<hm_v2i*>(p));
#else
    *reinterpret_cast<int2*>(p) = make_int2(a, b);
#endif
}
__device__ __forceinline__ void hm_out_store4(unsigned short* p, unsigned a, unsigned b, unsigned c, unsigned d)
{
#if RASTER_NT_STORES
    hm_v4u v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<hm_v4u*>(p));
#else
    *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
#endif
}

struct FaceBox {           // 8 bytes per face: sample-space box + winding mask in x0[15:14]
    unsigned short x0m, y0, x1, y1;
};

__device__ __forceinline__ float topix(float v, int is)
{
    float a = v * (float)is;
    a = a + (float)is;
    a = a - 1.0f;
    return 0.5f * a;
}
__device__ __forceinline__ bool backside(const float* f)
{
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#define SR_MAX 64          // super-regions per frame (is <= 1024)
// log2 of the super-region side in samples: 64^2 up to 512^2 samples, 128^2 above -> at most 8 x 8 = SR_MAX bins.  A
// region workgroup scans its whole bin, so a bin holds 4 (is <= 512) or 16 regions' worth of faces: the finer bins cut
// the scan of the 512^2 rasters by 4 (k_raster_fwd 63.5 -> 60.1 us).
__host__ __device__ __forceinline__ int hm_sr_shift(int is) { return is <= 512 ? 6 : 7; }

__device__ __forceinline__ int hm_wave_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ float rlane(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Sweep bit planes of one 8x8-pixel tile (wave = tile, lane = output pixel (lane>>3, lane&7), bal[2*dy+dx] = ballot of
// "sample (dy,dx) of my pixel is covered"): plane 0 = samples with alpha==0 and g<0 ("wants to be filled", walked by the
// outward sweeps), plane 1 = samples with alpha==1 and g>0 ("wants to be emptied", the only samples the inward sweeps
// can collect from); each plane as row words (bits along x) and column words (bits along y).
__device__ __forceinline__ unsigned spread8(unsigned x)      // bit k of x -> bit 2k
{
    x = (x | (x << 4)) & 0x0f0fu;
    x = (x | (x << 2)) & 0x3333u;
    x = (x | (x << 1)) & 0x5555u;
    return x;
}
// Layout of the sweep planes: TILE-BLOCKED, (B, T, T, 4, 16) u16 with T = is/16 tiles per side and the four
// (orientation, plane) combinations of a tile side by side: [row words plane 0 | row words plane 1 | column words plane 0 |
// column words plane 1], 16 words each = the tile's 128 bytes, written by its wave as ONE full cache line.  (Line-major
// planes made every 2-byte word of a tile a partial write into a different line, and since neighbouring tiles run on
// different XCDs their dirty fragments never merged in an L2: 64 partial HBM writes per tile, 140 MB per launch.)
// A 64-sample word of a line is four tiles' words (hm_plane_word64).
__device__ __forceinline__ long hm_plane_at(int b, int ty, int tx, int combo, int T)
{
    return ((((long)b * T + ty) * T + tx) * 4 + combo) * 16;
}
// 64 samples [64k, 64k+64) of line d0 of `axis` (1: sample row yi = d0, bits along x ; 0: sample column xi = d0, bits
// along the 16-sample groups ygrp of emit_planes), plane pl
__device__ __forceinline__ unsigned long long hm_plane_word64(const unsigned short* __restrict__ planes, int b, int is,
                                                              int axis, int pl, int d0, int k)
{
    const int T = is / 16;
    unsigned long long w = 0ull;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned v;
        if (axis) {
            const int t = is - 1 - d0;
            v = planes[hm_plane_at(b, t >> 4, 4 * k + j, pl, T) + (t & 15)];
        } else {
            v = planes[hm_plane_at(b, T - 1 - (4 * k + j), d0 >> 4, 2 + pl, T) + (d0 & 15)];
        }
        w |= (unsigned long long)v << (16 * j);
    }
    return w;
}
// nb / pb [2*dy+dx] = ballots of "g < 0" / "g > 0" per sample (all four equal when g lives on the pooled grid).
struct Ballots4 { unsigned long long s0, s1, s2, s3; };      // [2*dy+dx]; plain scalars, picked with selects only:
// a lane-dependent index into an array puts the array in scratch memory (80-112 bytes per thread = 100 MB of HBM writes
// per launch when this was `cov[2 * sub]`)
__device__ __forceinline__ void emit_planes(const Ballots4 cov, const Ballots4 nb, const Ballots4 pb, int b, int B, int is,
                                            int tx, int ty, int lane, unsigned short* __restrict__ planes)
{
    // lanes 0..15: row word of sample row (rr2 = l>>1, dy = l&1), plane 0 ; lanes 16..31: same for plane 1 ;
    // lanes 32..47: column word of sample column (cc2 = l>>1, dx = l&1), plane 0 ; lanes 48..63: plane 1
    const int l = lane & 15, pl = (lane >> 4) & 1, hi = l >> 1, sub = l & 1;
    unsigned long long b0, b1;          // the two ballots this lane interleaves
    unsigned word;
    const Ballots4 g = pl == 0 ? nb : pb;
    if (lane < 32) {                    // (dy = sub): dx = 0 -> even bits, dx = 1 -> odd bits ; sample row
        const unsigned long long c0 = sub ? cov.s2 : cov.s0, c1 = sub ? cov.s3 : cov.s1;      // yi = is-1-2*(ty*8+hi)-sub,
        const unsigned long long g0 = sub ? g.s2 : g.s0, g1 = sub ? g.s3 : g.s1;              // i.e. (is-1-yi) & 15 = l
        b0 = pl == 0 ? (~c0 & g0) : (c0 & g0);
        b1 = pl == 0 ? (~c1 & g1) : (c1 & g1);
        const unsigned a = (unsigned)(b0 >> (8 * hi)) & 0xffu, o = (unsigned)(b1 >> (8 * hi)) & 0xffu;
        word = spread8(a) | (spread8(o) << 1);
    } else {                            // (dx = sub): bit 15 - (2*rr2 + dy) <- sample (rr2, dy) of column cc2 = hi
        const unsigned long long c0 = sub ? cov.s1 : cov.s0, c1 = sub ? cov.s3 : cov.s2;      // dy = 0 / dy = 1 ; sample column
        const unsigned long long g0 = sub ? g.s1 : g.s0, g1 = sub ? g.s3 : g.s2;              // xi = 16*tx + l
        b0 = pl == 0 ? (~c0 & g0) : (c0 & g0);
        b1 = pl == 0 ? (~c1 & g1) : (c1 & g1);
        // bits 8*rr2 + cc2 -> one byte with row rr2 at bit 7 - rr2
        const unsigned k0 = (unsigned)((((b0 >> hi) & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56);
        const unsigned k1 = (unsigned)((((b1 >> hi) & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56);
        word = (spread8(k0) << 1) | spread8(k1);    // the word of the 16-sample group ygrp = T-1-ty along y
    }
    // the 64 words of the tile (lane order = memory order, combo = lane >> 4) leave as eight 16-byte stores: sub-dword
    // stores reach HBM as one partial write each (measured: 54 B of WRITE_SIZE per 2-byte store)
    const unsigned p2 = (word & 0xffffu) | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)word, 0x101, 0xf, 0xf, false) << 16);   // row_shl:1
    const unsigned q2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p2, 0x102, 0xf, 0xf, false);      // row_shl:2 : words 2,3
    const unsigned r2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)p2, 0x104, 0xf, 0xf, false);      // row_shl:4 : words 4,5
    const unsigned s2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)q2, 0x104, 0xf, 0xf, false);      // row_shl:4 : words 6,7
    if ((lane & 7) == 0) hm_out_store4(planes + hm_plane_at(b, ty, tx, 0, is / 16) + lane, p2, q2, r2, s2);
}

// In-graph timing (hm_sil_timestamps): byte 1 of workspace word 24 switches it on; then every workgroup of the three heavy
// kernels (every wave of the persistent sweep kernel) stores the device wall clock (s_memrealtime, constant rate) at its
// entry and at its exit into a slot pair of its own - plain 8-byte stores, no same-address atomics that would stretch the
// kernel being measured; the host takes the earliest start and the latest end.  Off: one scalar load per workgroup.
__device__ __forceinline__ bool hm_ts_enabled(const unsigned int* __restrict__ flag_word)
{
    return flag_word && ((flag_word[0] >> 8) & 1u);
}
__device__ __forceinline__ void hm_ts_store(unsigned long long* __restrict__ slots, long unit, int which, unsigned long long t)
{
    __builtin_nontemporal_store(t, slots + 2 * unit + which);
}

// grid (B): per-frame sums of the tile partials, then the last block of every clip (clip_len consecutive frames) finishes:
// loss = (sum_sq / keep_sum[clip]) / clip_len ; iou = mean_b inter_b / (union_b + eps) over the clip's frames.
// out[clip*out_stride + 0]=loss, [+1]=iou.  The clip's ticket word is slot 3 of the frame record of its first frame.
__device__ __forceinline__ void sil_reduce_frame(int b, const float* __restrict__ partials, int ntiles,
                                                 const float* __restrict__ keep_sum, float* __restrict__ frame_rec,
                                                 float* __restrict__ out, float* __restrict__ frame_out, int clip_len,
                                                 int out_stride)
{
    __shared__ float red[16];
    __shared__ int s_flag;
    float sq = 0.f, in = 0.f, un = 0.f;
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        const float4 p = *reinterpret_cast<const float4*>(partials + ((long)b * ntiles + t) * 4);
        sq += p.x; in += p.y; un += p.z;
    }
    sq = hm_block_sum(sq, red);
    in = hm_block_sum(in, red);
    un = hm_block_sum(un, red);
    if (threadIdx.x == 0) { hm_partial_store(frame_rec + 4 * b, sq); hm_partial_store(frame_rec + 4 * b + 1, in / (un + 1e-6f)); }
    // per-frame values (un-normalised sum of squares, IoU): what a loss that keeps the frames apart needs
    if (frame_out && threadIdx.x == 0) { frame_out[2 * b] = sq; frame_out[2 * b + 1] = in / (un + 1e-6f); }
    if (!out) return;
    const int clip = b / clip_len;
    float* crec = frame_rec + 4L * clip * clip_len;
    if (hm_last_block(reinterpret_cast<unsigned int*>(crec + 3), clip_len, &s_flag)) {
        const float total_sq = hm_last_block_sum(crec, clip_len, 4, red);
        const float iou_sum = hm_last_block_sum(crec + 1, clip_len, 4, red);
        if (threadIdx.x == 0) {
            out[(long)clip * out_stride] = (total_sq / keep_sum[clip]) / (float)clip_len;
            out[(long)clip * out_stride + 1] = iou_sum / (float)clip_len;
        }
    }
}

// ---------------------------------------------------------------- records of the backward's work list (raster_lines.hip builds it, raster_sweep.hip walks it)
struct SweepFace {            // 64 B: one face of the flattened work list
    int bf, b, off, flags;    // face slot b*F+fi, frame, first item, bit 0: accumulate with atomics (capacity overflow)
    float px[3], py[3];       // pixel-space corners (px[0..2], py[0..2] contiguous)
    unsigned short cum[12];   // inclusive item counts of the families, family = winding*6 + edge*2 + axis
};
#define SWEEP_PASS_FACES 16   // faces of a unit staged in LDS at a time (a unit with more takes several passes)
#ifndef SWEEP_USHIFT
#define SWEEP_USHIFT 8         // a unit = 256 consecutive items of the global list (see k_bwd_sweep)
#endif
#define SWEEP_UNIT (1 << SWEEP_USHIFT)
#define SWEEP_TRIPS (SWEEP_UNIT / 64)      // stage-1 trips of a unit
#define SWEEP_TBATCH (SWEEP_TRIPS < 4 ? SWEEP_TRIPS : 4)      // trips whose loads are in flight together
struct SweepItem { float x, c0, c1; int base0, base1, nb0, fn, meta; };   // meta: face | t0<<4 | t1<<7 | use0<<10 | use1<<11
struct SweepList {
    SweepFace* tab; int* offs; unsigned int* ufirst; unsigned int* tickets; float* upart;
    unsigned long long* cnt; unsigned int* done; unsigned long long* total;
    int ucap, slot_cap;
};

struct SweepSrc { int d1; float g; int owner; };
#define SWEEP_CUMW 16           // cumulative-count slots per line (is <= 1024)

// Global addresses of the hot loops: with W32 (every array of the workspace below 4 GB, checked by the host) an element address
// is the array's base - a scalar register pair - plus a 32-bit BYTE offset formed in 32-bit arithmetic, which is the
// addressing mode of the global loads themselves; with 64-bit element indices a third of stage 1's instructions were the
// 64-bit multiply-adds, sign extensions and shifts of its three addresses.
template <bool W32> struct HmOff { typedef long t; };
template <> struct HmOff<true> { typedef unsigned t; };
template <bool W32, class T>
__device__ __forceinline__ T* hm_at(T* __restrict__ base, typename HmOff<W32>::t i)          // (T may be const)
{
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type byte_t;
    if (W32) return reinterpret_cast<T*>(reinterpret_cast<byte_t*>(base) + (unsigned)(i * (unsigned)sizeof(T)));
    return base + i;
}
