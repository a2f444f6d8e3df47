// valu_ceiling.hip -- what does one wave64 VALU instruction cost on a gfx950 SIMD?  
//
// The sweep / raster kernels issue 18.7 M VALU wave-instructions per launch.  Whether that is 0.78 or 0.39 of the SIMDs' issue
// capacity depends on cycles per wave64 instruction: the PMC ratio SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU reads 1.02 quad-cycles
// (= 4 cycles), the micro-architecture guide's throughput table quotes 2 cycles for v_fma_f32.  This program measures it
// directly, per opcode class, at 1 / 2 / 4 / 8 waves per SIMD:
//   every wave runs `iters` trips of a 128-instruction block of INDEPENDENT instructions (16 accumulators, 8 rounds), brackets the
//   loop with s_memtime (shader clock) and stores the elapsed cycles; per class and occupancy
//       cycles per wave-instruction per SIMD = mean elapsed cycles / (instructions per wave x waves per SIMD)
//   and the same from the HIP-event wall time at the nominal 2.4 GHz (the two differ by the clock the chip really ran at).
// The IEEE fp32 division is timed as the compiler emits it (`x / y` with -ffp-contract=off: v_div_scale x2, v_rcp, 4-5 v_fma,
// v_div_fmas, v_div_fixup) and reported per DIVISION and per instruction of that sequence (count read from the ISA by
// tools/valu_mix.py).
// build:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/valu_ceiling tools/valu_ceiling.hip
// run:    tools/valu_ceiling > gpurun_out/r06_valu_ceiling.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 16 independent instructions, one per accumulator (a..p), operands q / r are loop-invariant registers
#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define R8X(B) B B B B B B B B

#define ACC_DECL(T, init) T a0 = init, a1 = init + 1, a2 = init + 2, a3 = init + 3, a4 = init + 4, a5 = init + 5, a6 = init + 6, a7 = init + 7, \
                            a8 = init + 8, a9 = init + 9, a10 = init + 10, a11 = init + 11, a12 = init + 12, a13 = init + 13, a14 = init + 14, a15 = init + 15
#define ACC_IO "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11), \
               "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15)
#define ACC_SUM (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15)

// one asm statement = 8 rounds x 16 accumulators = 128 instructions (operand numbers: %0..%15 accumulators, %16 / %17 invariants)
#define S_(x) #x
#define I3(op, n) op " %" S_(n) ", %" S_(n) ", %16\n"
#define I4(op, n) op " %" S_(n) ", %" S_(n) ", %16, %17\n"

enum { OP_ADD_U32, OP_ADD_F32, OP_MUL_F32, OP_FMA_F32, OP_MAD_U24, OP_ALIGNBIT, OP_OR3, OP_LSHL_ADD, OP_MUL_LO, OP_CMP_CND, OP_ADD_F64,
       OP_FMA_F64, OP_PK_FMA, OP_RCP, OP_CVT, OP_DPP_ADD, OP_MINMAX, OP_DIV_F32, OP_MOV, OP_AND, OP_OR, OP_LSHLREV, OP_SUB_U32, OP_XOR,
       OP_MAX_I32, OP_BFE, OP_CMP_ONLY, OP_CND_ONLY, OP_SUB_F32, OP_N };
static const char* kNames[OP_N] = {"v_add_u32", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_mad_u32_u24", "v_alignbit_b32", "v_or3_b32",
                                   "v_lshl_add_u32", "v_mul_lo_u32", "v_cmp_lt_f32+v_cndmask_b32", "v_add_f64", "v_fma_f64",
                                   "v_pk_fma_f32", "v_rcp_f32", "v_cvt_f32_i32", "v_add_u32_dpp(row_shr:1)", "v_min_f32/v_max_f32",
                                   "ieee_div_f32(sequence)", "v_mov_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_sub_u32", "v_xor_b32",
                                   "v_max_i32", "v_bfe_u32", "v_cmp_lt_f32(to sgpr pair)", "v_cndmask_b32(sgpr pair)", "v_sub_f32"};
// wave-instructions per trip of the timed loop (the division: DIVISIONS per trip; its instruction count comes from the ISA)
static const int kPerTrip[OP_N] = {128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 16,
                                    128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128};

template <int OP>
__global__ __launch_bounds__(256) void k_valu(unsigned long long* __restrict__ cycles, float* __restrict__ sink, int iters, float fq, float fr)
{
    const unsigned tid = threadIdx.x + blockIdx.x * blockDim.x;
    unsigned long long t0 = 0, t1 = 0;
    float res = 0.f;
    if constexpr (OP == OP_ADD_F64 || OP == OP_FMA_F64) {
        ACC_DECL(double, (double)tid);
        const double q = (double)fq, r = (double)fr;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            if constexpr (OP == OP_ADD_F64) {
#define OPX(n) I3("v_add_f64", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else {
#define OPX(n) I4("v_fma_f64", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            }
        }
        t1 = __builtin_readcyclecounter();
        res = (float)ACC_SUM;
    } else if constexpr (OP == OP_PK_FMA) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 i0 = {(float)tid, 1.f};
        f2 a0 = i0, a1 = i0 + 1.f, a2 = i0 + 2.f, a3 = i0 + 3.f, a4 = i0 + 4.f, a5 = i0 + 5.f, a6 = i0 + 6.f, a7 = i0 + 7.f, a8 = i0 + 8.f,
           a9 = i0 + 9.f, a10 = i0 + 10.f, a11 = i0 + 11.f, a12 = i0 + 12.f, a13 = i0 + 13.f, a14 = i0 + 14.f, a15 = i0 + 15.f;
        const f2 q = {fq, fq}, r = {fr, fr};
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#define OPX(n) I4("v_pk_fma_f32", n)
            asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
        }
        t1 = __builtin_readcyclecounter();
        const f2 s = ACC_SUM;
        res = s.x + s.y;
    } else if constexpr (OP == OP_DIV_F32) {
        // the compiler's IEEE division, 16 independent quotients per trip (a chain per accumulator: x <- q / x keeps it bounded)
        ACC_DECL(float, 1.0f + (float)(tid & 7));
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            a0 = fq / a0; a1 = fq / a1; a2 = fq / a2; a3 = fq / a3; a4 = fq / a4; a5 = fq / a5; a6 = fq / a6; a7 = fq / a7;
            a8 = fq / a8; a9 = fq / a9; a10 = fq / a10; a11 = fq / a11; a12 = fq / a12; a13 = fq / a13; a14 = fq / a14; a15 = fq / a15;
            asm volatile("" : ACC_IO);
        }
        t1 = __builtin_readcyclecounter();
        res = ACC_SUM;
    } else if constexpr (OP == OP_ADD_F32 || OP == OP_MUL_F32 || OP == OP_FMA_F32 || OP == OP_RCP || OP == OP_MINMAX || OP == OP_CMP_CND ||
                         OP == OP_CMP_ONLY || OP == OP_CND_ONLY || OP == OP_SUB_F32) {
        ACC_DECL(float, 1.0f + (float)(tid & 7));
        const float q = fq, r = fr;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            if constexpr (OP == OP_ADD_F32) {
#define OPX(n) I3("v_add_f32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_SUB_F32) {
#define OPX(n) I3("v_sub_f32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_CMP_ONLY) {
                // compares only, results into eight SGPR pairs (never read: the accumulators are the sources)
                asm volatile(
                    ".rept 8\n"
                    "v_cmp_lt_f32_e64 s[36:37], %0, %16\n v_cmp_lt_f32_e64 s[38:39], %1, %16\n v_cmp_lt_f32_e64 s[40:41], %2, %16\n"
                    "v_cmp_lt_f32_e64 s[42:43], %3, %16\n v_cmp_lt_f32_e64 s[44:45], %4, %16\n v_cmp_lt_f32_e64 s[46:47], %5, %16\n"
                    "v_cmp_lt_f32_e64 s[48:49], %6, %16\n v_cmp_lt_f32_e64 s[50:51], %7, %16\n"
                    "v_cmp_lt_f32_e64 s[36:37], %8, %16\n v_cmp_lt_f32_e64 s[38:39], %9, %16\n v_cmp_lt_f32_e64 s[40:41], %10, %16\n"
                    "v_cmp_lt_f32_e64 s[42:43], %11, %16\n v_cmp_lt_f32_e64 s[44:45], %12, %16\n v_cmp_lt_f32_e64 s[46:47], %13, %16\n"
                    "v_cmp_lt_f32_e64 s[48:49], %14, %16\n v_cmp_lt_f32_e64 s[50:51], %15, %16\n"
                    ".endr\n"
                    : ACC_IO : "v"(q), "v"(r)
                    : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51");
            } else if constexpr (OP == OP_CND_ONLY) {
                // selects only, all reading one SGPR pair written once before the loop body's block (exec: every lane)
#define OPX(n) "v_cndmask_b32_e64 %" S_(n) ", %" S_(n) ", %17, s[36:37]\n"
                asm volatile("s_mov_b64 s[36:37], exec\n s_nop 4\n" R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r) : "s36", "s37");
#undef OPX
            } else if constexpr (OP == OP_MUL_F32) {
#define OPX(n) I3("v_mul_f32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_FMA_F32) {
#define OPX(n) I4("v_fma_f32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_RCP) {
#define OPX(n) "v_rcp_f32 %" S_(n) ", %" S_(n) "\n"
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_MINMAX) {
#define OPA(n) I3("v_min_f32", n)
#define OPB(n) I3("v_max_f32", n)
                asm volatile(R16(OPA) R16(OPB) R16(OPA) R16(OPB) R16(OPA) R16(OPB) R16(OPA) R16(OPB) : ACC_IO : "v"(q), "v"(r));
#undef OPA
#undef OPB
            } else {
                // eight compares into eight SGPR pairs, then the eight selects that read them (a compare and its select are 8
                // instructions apart: beyond any VALU-writes-SGPR wait state), twice per 16 accumulators = 64 + 64 per trip
#define CMP(n, s) "v_cmp_lt_f32_e64 s[" S_(s) ":" #s "+1], %" S_(n) ", %16\n"
#define CND(n, s) "v_cndmask_b32_e64 %" S_(n) ", %" S_(n) ", %17, s[" S_(s) ":" #s "+1]\n"
#define HALF(b) "v_cmp_lt_f32_e64 s[36:37], %" #b ", %16\n"
                // (written out: the SGPR pair numbers must be literals)
                asm volatile(
                    ".rept 8\n"
                    "v_cmp_lt_f32_e64 s[36:37], %0, %16\n v_cmp_lt_f32_e64 s[38:39], %1, %16\n v_cmp_lt_f32_e64 s[40:41], %2, %16\n"
                    "v_cmp_lt_f32_e64 s[42:43], %3, %16\n v_cmp_lt_f32_e64 s[44:45], %4, %16\n v_cmp_lt_f32_e64 s[46:47], %5, %16\n"
                    "v_cmp_lt_f32_e64 s[48:49], %6, %16\n v_cmp_lt_f32_e64 s[50:51], %7, %16\n"
                    "v_cndmask_b32_e64 %8, %8, %17, s[36:37]\n v_cndmask_b32_e64 %9, %9, %17, s[38:39]\n"
                    "v_cndmask_b32_e64 %10, %10, %17, s[40:41]\n v_cndmask_b32_e64 %11, %11, %17, s[42:43]\n"
                    "v_cndmask_b32_e64 %12, %12, %17, s[44:45]\n v_cndmask_b32_e64 %13, %13, %17, s[46:47]\n"
                    "v_cndmask_b32_e64 %14, %14, %17, s[48:49]\n v_cndmask_b32_e64 %15, %15, %17, s[50:51]\n"
                    ".endr\n"
                    : ACC_IO : "v"(q), "v"(r)
                    : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51");
#undef CMP
#undef CND
#undef HALF
            }
        }
        t1 = __builtin_readcyclecounter();
        res = ACC_SUM;
    } else {
        ACC_DECL(unsigned, tid);
        const unsigned q = (unsigned)fq | 1u, r = (unsigned)fr | 3u;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            if constexpr (OP == OP_ADD_U32) {
#define OPX(n) I3("v_add_u32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_MAD_U24) {
#define OPX(n) I4("v_mad_u32_u24", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_ALIGNBIT) {
#define OPX(n) I4("v_alignbit_b32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_OR3) {
#define OPX(n) I4("v_or3_b32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_LSHL_ADD) {
#define OPX(n) "v_lshl_add_u32 %" S_(n) ", %" S_(n) ", 1, %16\n"
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_MUL_LO) {
#define OPX(n) I3("v_mul_lo_u32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_CVT) {
#define OPX(n) "v_cvt_f32_i32 %" S_(n) ", %" S_(n) "\n"
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_MOV) {
                // (a move chain a_n <- invariant would be dead: the destination alternates between two invariants instead)
#define OPA(n) "v_mov_b32 %" S_(n) ", %16\n"
#define OPB(n) "v_mov_b32 %" S_(n) ", %17\n"
                asm volatile(R16(OPA) R16(OPB) R16(OPA) R16(OPB) R16(OPA) R16(OPB) R16(OPA) R16(OPB) : ACC_IO : "v"(q), "v"(r));
#undef OPA
#undef OPB
            } else if constexpr (OP == OP_AND) {
#define OPX(n) I3("v_and_b32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_OR) {
#define OPX(n) I3("v_or_b32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_XOR) {
#define OPX(n) I3("v_xor_b32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_LSHLREV) {
#define OPX(n) "v_lshlrev_b32 %" S_(n) ", 1, %" S_(n) "\n"
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_SUB_U32) {
#define OPX(n) I3("v_sub_u32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_MAX_I32) {
#define OPX(n) I3("v_max_i32", n)
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else if constexpr (OP == OP_BFE) {
#define OPX(n) "v_bfe_u32 %" S_(n) ", %" S_(n) ", 1, 31\n"
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            } else {   // OP_DPP_ADD
#define OPX(n) "v_add_u32_dpp %" S_(n) ", %" S_(n) ", %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                asm volatile(R8X(R16(OPX)) : ACC_IO : "v"(q), "v"(r));
#undef OPX
            }
        }
        t1 = __builtin_readcyclecounter();
        res = (float)ACC_SUM;
    }
    if ((threadIdx.x & 63) == 0) cycles[tid >> 6] = t1 - t0;
    if (res == 1234.5678f) sink[0] = res;         // (keeps the accumulators alive)
}

typedef void (*kern_t)(unsigned long long*, float*, int, float, float);
template <int OP> static kern_t pick() { return k_valu<OP>; }
static kern_t kernel_of(int op)
{
    switch (op) {
#define C(o) case o: return pick<o>();
        C(OP_ADD_U32) C(OP_ADD_F32) C(OP_MUL_F32) C(OP_FMA_F32) C(OP_MAD_U24) C(OP_ALIGNBIT) C(OP_OR3) C(OP_LSHL_ADD) C(OP_MUL_LO)
        C(OP_CMP_CND) C(OP_ADD_F64) C(OP_FMA_F64) C(OP_PK_FMA) C(OP_RCP) C(OP_CVT) C(OP_DPP_ADD) C(OP_MINMAX) C(OP_DIV_F32)
        C(OP_MOV) C(OP_AND) C(OP_OR) C(OP_LSHLREV) C(OP_SUB_U32) C(OP_XOR) C(OP_MAX_I32) C(OP_BFE) C(OP_CMP_ONLY) C(OP_CND_ONLY) C(OP_SUB_F32)
#undef C
    }
    return nullptr;
}

int main(int argc, char** argv)
{
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    const double nominal_hz = 2.4e9;
    unsigned long long* d_cyc;
    float* d_sink;
    const int max_waves = simds * 8;
    CK(hipMalloc(&d_cyc, sizeof(unsigned long long) * max_waves));
    CK(hipMalloc(&d_sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<unsigned long long> h(max_waves);
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"simds\": %d, \"clock_khz_reported\": %d, \"iters\": %d,\n",
           prop.name, prop.gcnArchName, cus, simds, prop.clockRate, iters);
    printf(" \"method\": \"independent instructions (16 accumulators), 128 per trip, s_memtime around the loop; cycles per wave64 "
           "instruction per SIMD = mean elapsed shader cycles / (instructions per wave * waves per SIMD); wall_* = the same from "
           "HIP events at a nominal 2.4 GHz\",\n \"ops\": {\n");
    for (int op = 0; op < OP_N; ++op) {
        printf("  \"%s\": {", kNames[op]);
        for (int wi = 0, wps = 1; wps <= 8; wps *= 2, ++wi) {
            // wps waves per SIMD: blocks of 256 threads = 4 waves = one wave per SIMD of a CU; wps blocks per CU
            const int blocks = cus * wps;
            kern_t k = kernel_of(op);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, 8, 3.0f, 5.0f);      // warm-up (code fetch)
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, iters, 3.0f, 5.0f);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const int waves = blocks * 4;
            CK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
            double mean = 0.0, mx = 0.0;
            for (int i = 0; i < waves; ++i) { mean += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
            mean /= waves;
            const double per_wave = (double)iters * kPerTrip[op];
            const double cyc = mean / (per_wave * wps);
            const double wall = (ms * 1e-3) * nominal_hz / (per_wave * wps);       // (all SIMDs run the same: per-SIMD time = wall)
            printf("%s\"w%d\": {\"cycles_per_instr\": %.3f, \"wall_cycles_per_instr_at_2.4GHz\": %.3f, \"launch_us\": %.1f}",
                   wi ? ", " : "", wps, cyc, wall, ms * 1e3);
            if (op == OP_DIV_F32 && wps == 8)
                printf(", \"instructions_per_division\": 11, \"w8_wall_cycles_per_instruction_of_the_sequence\": %.3f", wall / 11.0);
        }
        printf("}%s\n", op + 1 < OP_N ? "," : "");
    }
    printf(" }\n}\n");
    return 0;
}
