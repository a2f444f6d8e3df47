// fetch_calib.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for NARROW and SCATTERED accesses on gfx950?
// (the pose initialisation's line expansion shows 532 MB of PMC traffic against a byte model of 239 MB +
// sources; the guide calibrates FETCH_SIZE only for wide coalesced streams - "exactly half the bytes" - and says that other
// widths and WRITE_SIZE are uncalibrated.)
//
// Kernels with KNOWN requested bytes, each launched once per pattern over buffers far larger than the 256 MB Infinity Cache,
// names that say what they do (the PMC rows are matched by kernel name):
//   k_read_stream16 / k_read_stream4      coalesced reads, 16 B / 4 B per lane
//   k_read_gather4 / k_read_gather12      one 4-byte / 12-byte element per lane at a pseudo-random index (every access its own line)
//   k_read_gather4_local                  4-byte gathers whose 64 lanes fall into a window of 4 KB (the owner / gradient gathers of the
//                                         line expansion: neighbours along a line are neighbours in memory across a stride)
//   k_write_stream16 / k_write_stream4    coalesced stores
//   k_write_scatter12 / k_write_scatter4  one 12-byte / 4-byte element per lane at a pseudo-random index
// run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o f -- tools/fetch_calib      (and again with WRITE_SIZE)
//        python tools/fetch_calib_summary.py out > profiles/r06_fetch_calib.json
// The program prints the requested bytes per kernel as JSON on stdout (the summary joins them with the counters).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}
struct S12 { float a, b, c; };

__global__ void k_read_stream16(const float4* __restrict__ src, long n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345f) sink[0] = acc;
}
__global__ void k_read_stream4(const float* __restrict__ src, long n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 1.2345f) sink[0] = acc;
}
__global__ void k_read_gather4(const float* __restrict__ src, long nelem, long n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += src[mix(i) % nelem];
    if (acc == 1.2345f) sink[0] = acc;
}
__global__ void k_read_gather12(const S12* __restrict__ src, long nelem, long n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const S12 v = src[mix(i) % nelem]; acc += v.a + v.b + v.c; }
    if (acc == 1.2345f) sink[0] = acc;
}
__global__ void k_read_gather4_local(const float* __restrict__ src, long nelem, long n, float* __restrict__ sink)
{
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long wave = i >> 6;                                  // the wave's window: 1024 floats at a random place
        const long base = (mix(wave) % (nelem / 1024)) * 1024;
        acc += src[base + (mix(i) & 1023)];
    }
    if (acc == 1.2345f) sink[0] = acc;
}
__global__ void k_write_stream16(float4* __restrict__ dst, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_write_stream4(float* __restrict__ dst, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = (float)i;
}
__global__ void k_write_scatter12(S12* __restrict__ dst, long nelem, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { S12 v = {1.f, 2.f, (float)i}; dst[mix(i) % nelem] = v; }
}
__global__ void k_write_scatter4(float* __restrict__ dst, long nelem, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[mix(i) % nelem] = (float)i;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;                 // 2 GB buffers: eight times the Infinity Cache
    void *a, *b;
    float* sink;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 0, bytes));
    CK(hipDeviceSynchronize());
    const int blocks = 256 * 16, threads = 256;
    const long n16 = (long)(bytes / 16), n4 = (long)(bytes / 4) / 4, ng = 32L << 20;      // streams: 2 GB / 512 MB; gathers: 32 M accesses
    printf("{\n");
    hipLaunchKernelGGL(k_read_stream16, dim3(blocks), dim3(threads), 0, 0, (const float4*)a, n16, sink);
    printf(" \"k_read_stream16\": {\"requested_bytes\": %ld, \"accesses\": %ld, \"bytes_per_access\": 16},\n", n16 * 16, n16);
    hipLaunchKernelGGL(k_read_stream4, dim3(blocks), dim3(threads), 0, 0, (const float*)a, n4, sink);
    printf(" \"k_read_stream4\": {\"requested_bytes\": %ld, \"accesses\": %ld, \"bytes_per_access\": 4},\n", n4 * 4, n4);
    hipLaunchKernelGGL(k_read_gather4, dim3(blocks), dim3(threads), 0, 0, (const float*)a, (long)(bytes / 4), ng, sink);
    printf(" \"k_read_gather4\": {\"requested_bytes\": %ld, \"accesses\": %ld, \"bytes_per_access\": 4},\n", ng * 4, ng);
    hipLaunchKernelGGL(k_read_gather12, dim3(blocks), dim3(threads), 0, 0, (const S12*)a, (long)(bytes / 12), ng, sink);
    printf(" \"k_read_gather12\": {\"requested_bytes\": %ld, \"accesses\": %ld, \"bytes_per_access\": 12},\n", ng * 12, ng);
    hipLaunchKernelGGL(k_read_gather4_local, dim3(blocks), dim3(threads), 0, 0, (const float*)a, (long)(bytes / 4), ng, sink);
    printf(" \"k_read_gather4_local\": {\"requested_bytes\": %ld, \"accesses\": %ld, \"bytes_per_access\": 4},\n", ng * 4, ng);
    hipLaunchKernelGGL(k_write_stream16, dim3(blocks), dim3(threads), 0, 0, (float4*)b, n16);
    printf(" \"k_write_stream16\": {\"requested_bytes\": %ld, \"accesses\": %ld, \"bytes_per_access\": 16},\n", n16 * 16, n16);
    hipLaunchKernelGGL(k_write_stream4, dim3(blocks), dim3(threads), 0, 0, (float*)b, n4);
    printf(" \"k_write_stream4\": {\"requested_bytes\": %ld, \"accesses\": %ld, \"bytes_per_access\": 4},\n", n4 * 4, n4);
    hipLaunchKernelGGL(k_write_scatter12, dim3(blocks), dim3(threads), 0, 0, (S12*)b, (long)(bytes / 12), ng);
    printf(" \"k_write_scatter12\": {\"requested_bytes\": %ld, \"accesses\": %ld, \"bytes_per_access\": 12},\n", ng * 12, ng);
    hipLaunchKernelGGL(k_write_scatter4, dim3(blocks), dim3(threads), 0, 0, (float*)b, (long)(bytes / 4), ng);
    printf(" \"k_write_scatter4\": {\"requested_bytes\": %ld, \"accesses\": %ld, \"bytes_per_access\": 4}\n}\n", ng * 4, ng);
    CK(hipDeviceSynchronize());
    return 0;
}
