// valu_lab.hip — issue rate of the vector instructions the Jaccard edge kernel is made of (tools only).
// Every kernel runs the same loop: 64 independent instructions of one kind per iteration (16 registers, 4 rounds), 8 waves per
// SIMD resident, every CU busy; reported: cycles per wave-instruction per SIMD (shader clock from s_memtime).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 2000;

#define BODY16(INSTR)                                                                                                 \
  INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) INSTR(8) INSTR(9) INSTR(10) INSTR(11) INSTR(12) \
  INSTR(13) INSTR(14) INSTR(15)

#define DEFKERNEL(NAME, ASMTEXT)                                                                              \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned long long* cyc) {                       \
    unsigned r[16];                                                                                           \
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 2654435761u + i * 40503u;                              \
    unsigned a = threadIdx.x | 1u, b = blockIdx.x + 3u;                                                       \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < ITER; ++it) {                                                                       \
      _Pragma("unroll") for (int rep = 0; rep < 4; ++rep) {                                                   \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASMTEXT : "+v"(r[i]) : "v"(a), "v"(b));   \
      }                                                                                                       \
    }                                                                                                         \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
    unsigned x = 0;                                                                                           \
    for (int i = 0; i < 16; ++i) x ^= r[i];                                                                   \
    out[blockIdx.x * 256 + threadIdx.x] = x;                                                                  \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
  }

DEFKERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEFKERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEFKERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
DEFKERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %2")
DEFKERNEL(k_min3, "v_min3_u32 %0, %0, %1, %2")
DEFKERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
DEFKERNEL(k_xor_sdwa, "v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
DEFKERNEL(k_mul24_sdwa, "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD")
DEFKERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0")
DEFKERNEL(k_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf")
DEFKERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x1e")
DEFKERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFKERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
DEFKERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 6, %1")
DEFKERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
DEFKERNEL(k_and, "v_and_b32 %0, %0, %1")
DEFKERNEL(k_or, "v_or_b32 %0, %0, %1")
DEFKERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
DEFKERNEL(k_add, "v_add_u32 %0, %0, %1")
DEFKERNEL(k_sub, "v_sub_u32 %0, %0, %1")
DEFKERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0")
DEFKERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 8")
DEFKERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEFKERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 5")
DEFKERNEL(k_min, "v_min_u32 %0, %0, %1")
DEFKERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
DEFKERNEL(k_sad, "v_sad_u32 %0, %0, %1, %2")
DEFKERNEL(k_mov, "v_mov_b32 %0, %1")
DEFKERNEL(k_subf, "v_sub_f32 %0, %0, %1")
DEFKERNEL(k_min3f, "v_min3_f32 %0, |%0|, |%1|, 1.0")
DEFKERNEL(k_minf, "v_min_f32 %0, %0, %1")
DEFKERNEL(k_mulf, "v_mul_f32 %0, %0, %1")
DEFKERNEL(k_pksub, "v_pk_sub_u16 %0, %0, %1")
DEFKERNEL(k_pkmin, "v_pk_min_u16 %0, %0, %1")
DEFKERNEL(k_pkadd, "v_pk_add_u16 %0, %0, %1")
DEFKERNEL(k_pkmul, "v_pk_mul_lo_u16 %0, %0, %1")
DEFKERNEL(k_cmp_addc, "v_cmp_eq_u32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc")
DEFKERNEL(k_cmp, "v_cmp_eq_u32 vcc, %0, %1")
DEFKERNEL(k_xnor, "v_xnor_b32 %0, %0, %1")
DEFKERNEL(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
DEFKERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 3, %1")
DEFKERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
DEFKERNEL(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2")

// Same loop with the source registers varying from instruction to instruction (register i, i+5, i+10 of the same file):
// shows what operand fetch costs beyond the issue rate (VGPR bank conflicts of three-source forms).
#define DEFKERNEL3(NAME, ASMTEXT)                                                                             \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned long long* cyc) {                       \
    unsigned r[16];                                                                                           \
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 2654435761u + i * 40503u;                              \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
    for (int it = 0; it < ITER; ++it) {                                                                       \
      _Pragma("unroll") for (int rep = 0; rep < 4; ++rep) {                                                   \
        _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                        \
          asm volatile(ASMTEXT : "+v"(r[i]) : "v"(r[(i + 5) & 15]), "v"(r[(i + 10) & 15]));                   \
      }                                                                                                       \
    }                                                                                                         \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
    unsigned x = 0;                                                                                           \
    for (int i = 0; i < 16; ++i) x ^= r[i];                                                                   \
    out[blockIdx.x * 256 + threadIdx.x] = x;                                                                  \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
  }
DEFKERNEL3(k3_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEFKERNEL3(k3_xor, "v_xor_b32 %0, %0, %1")
DEFKERNEL3(k3_min3, "v_min3_u32 %0, %0, %1, %2")
DEFKERNEL3(k3_fma, "v_fma_f32 %0, %0, %1, %2")
DEFKERNEL3(k3_and, "v_and_b32 %0, %1, %2")
DEFKERNEL3(k3_lshr, "v_lshrrev_b32 %0, %1, %2")

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int grid = cus * 8;                                // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  unsigned* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)grid * 256 * 4));
  CK(hipMalloc(&cyc, (size_t)grid * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern) {
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, cyc);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    unsigned long long c0 = 0;
    CK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
    const double instr_per_wave = (double)ITER * 64;
    const double waves_per_simd = 8.0;
    // s_memtime counts at a fixed 100 MHz on this part: wall cycles from the event time and an assumed shader clock are not
    // trustworthy either, so report ns per wave-instruction per SIMD and the same relative to v_xor_b32
    const double ns = best * 1e6 / (instr_per_wave * waves_per_simd);
    printf("  %-14s %8.3f ms   %6.3f ns per wave-instruction per SIMD   (memtime ticks of block 0: %llu)\n", name, best, ns, c0);
    return ns;
  };
  printf("device %s, %d CUs, 8 waves per SIMD, %d x 64 instructions per wave\n", prop.name, cus, ITER);
  run("v_xor_b32", k_xor);
  run("v_fma_f32", k_fma);
  run("v_mul_u32_u24", k_mul24);
  run("v_and_or_b32", k_andor);
  run("v_min3_u32", k_min3);
  run("v_add3_u32", k_add3);
  run("v_xor_sdwa", k_xor_sdwa);
  run("v_mul24_sdwa", k_mul24_sdwa);
  run("v_lshlrev_b32", k_lshl);
  run("v_mov_dpp", k_dpp);
  run("v_bitop3_b32", k_bitop3);
  run("v_cndmask_b32", k_cndmask);
  run("v_mul_lo_u32", k_mullo);
  run("v_lshl_add_u32", k_lshladd);
  run("v_cvt_f32_u32", k_cvt);
  run("v_xor_b32 again", k_xor);
  run("v_and_b32", k_and);
  run("v_or_b32", k_or);
  run("v_or3_b32", k_or3);
  run("v_add_u32", k_add);
  run("v_sub_u32", k_sub);
  run("v_lshrrev_b32", k_lshr);
  run("v_bfe_u32", k_bfe);
  run("v_perm_b32", k_perm);
  run("v_alignbit_b32", k_alignbit);
  run("v_min_u32", k_min);
  run("v_mad_u32_u24", k_mad24);
  run("v_sad_u32", k_sad);
  run("v_mov_b32", k_mov);
  run("v_sub_f32", k_subf);
  run("v_min3_f32 abs", k_min3f);
  run("v_min_f32", k_minf);
  run("v_mul_f32", k_mulf);
  run("v_pk_sub_u16", k_pksub);
  run("v_pk_min_u16", k_pkmin);
  run("v_pk_add_u16", k_pkadd);
  run("v_pk_mul_lo_u16", k_pkmul);
  run("cmp_eq + addc", k_cmp_addc);
  run("v_cmp_eq_u32", k_cmp);
  run("v_xnor_b32", k_xnor);
  run("v_bfi_b32", k_bfi);
  run("v_lshl_or_b32", k_lshlor);
  run("v_xad_u32", k_xad);
  run("v_dot4_u32_u8", k_dot4);
  printf("varying source registers:\n");
  run("v_xor_b32 (2 regs)", k3_xor);
  run("v_and_b32 (2 other regs)", k3_and);
  run("v_lshrrev_b32 (2 regs)", k3_lshr);
  run("v_bitop3_b32 (3 regs)", k3_bitop3);
  run("v_min3_u32 (3 regs)", k3_min3);
  run("v_fma_f32 (3 regs)", k3_fma);
  return 0;
}
