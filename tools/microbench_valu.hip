// microbench_valu.hip -- issue cost of the VALU instructions the fp64 look-up is made of, on gfx950.
//
// Every kernel runs ITERS x 8 independent copies of one instruction per wavefront on all SIMDs (4 waves per SIMD,
// so dependent-issue latency is hidden and the number is throughput); cost is reported relative to v_fma_f64,
// whose rate defines the chip's fp64 vector peak (one wavefront instruction per 4 cycles per SIMD).
//   build:  hipcc --offload-arch=gfx950 -O2 -o gpurun_out/microbench_valu tools/microbench_valu.hip
//   run:    gpurun_out/microbench_valu          (prints one JSON object)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITERS 4096
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL_D(name, INSTR)                                                                         \
  __global__ __launch_bounds__(256) void name(double *out, double b, double c) {                      \
    double a0 = 1.0 + threadIdx.x * 1e-6, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,  \
           a6 = a0 + 6, a7 = a0 + 7;                                                                   \
    for (int i = 0; i < ITERS; i++) {                                                                 \
      asm volatile(INSTR : "+v"(a0) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a1) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a2) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a3) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a4) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a5) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a6) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a7) : "v"(b), "v"(c));                                                \
    }                                                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;              \
  }

#define KERNEL_F(name, INSTR)                                                                         \
  __global__ __launch_bounds__(256) void name(double *out, double bd, double cd) {                    \
    float b = (float)bd, c = (float)cd;                                                               \
    float a0 = 1.0f + threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
          a6 = a0 + 6, a7 = a0 + 7;                                                                    \
    for (int i = 0; i < ITERS; i++) {                                                                 \
      asm volatile(INSTR : "+v"(a0) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a1) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a2) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a3) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a4) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a5) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a6) : "v"(b), "v"(c));                                                \
      asm volatile(INSTR : "+v"(a7) : "v"(b), "v"(c));                                                \
    }                                                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;              \
  }

// double destination from a float source (and back): the source operand is %1 / %2 reinterpreted
#define KERNEL_CVT(name, INSTR)                                                                       \
  __global__ __launch_bounds__(256) void name(double *out, double bd, double cd) {                    \
    float f = (float)bd + threadIdx.x;                                                                \
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;                            \
    for (int i = 0; i < ITERS; i++) {                                                                 \
      asm volatile(INSTR : "=v"(a0) : "v"(f));                                                        \
      asm volatile(INSTR : "=v"(a1) : "v"(f));                                                        \
      asm volatile(INSTR : "=v"(a2) : "v"(f));                                                        \
      asm volatile(INSTR : "=v"(a3) : "v"(f));                                                        \
      asm volatile(INSTR : "=v"(a4) : "v"(f));                                                        \
      asm volatile(INSTR : "=v"(a5) : "v"(f));                                                        \
      asm volatile(INSTR : "=v"(a6) : "v"(f));                                                        \
      asm volatile(INSTR : "=v"(a7) : "v"(f));                                                        \
    }                                                                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + cd;         \
  }

KERNEL_D(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL_D(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_D(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_D(k_min_f64, "v_min_f64 %0, %0, %1")
KERNEL_D(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_D(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_D(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL_D(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
KERNEL_D(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1")
KERNEL_D(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
KERNEL_D(k_div_fmas_f64, "v_div_fmas_f64 %0, %0, %1, %2")
KERNEL_D(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %1")
KERNEL_D(k_mov_b64, "v_mov_b64 %0, %1")
KERNEL_F(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_F(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_F(k_exp_f32, "v_exp_f32 %0, %0")
KERNEL_F(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_F(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_F(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_F(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL_CVT(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1")
KERNEL_CVT(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1")

typedef void (*kern_t)(double *, double, double);
struct Entry { char const *name; kern_t k; };

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  int const cus = prop.multiProcessorCount;
  int const blocks = cus * 4 * 4;      // 4 waves per SIMD resident, 4 rounds
  double *out;
  if (hipMalloc(&out, sizeof(double) * blocks * 256) != hipSuccess) return 1;
  Entry const tab[] = {
      {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_min_f64", k_min_f64},
      {"v_rcp_f64", k_rcp_f64}, {"v_rsq_f64", k_rsq_f64}, {"v_sqrt_f64", k_sqrt_f64}, {"v_cmp_lt_f64", k_cmp_f64},
      {"v_ldexp_f64", k_ldexp_f64}, {"v_div_fixup_f64", k_div_fixup_f64}, {"v_div_fmas_f64", k_div_fmas_f64},
      {"v_lshl_add_u64", k_lshl_add_u64}, {"v_mov_b64", k_mov_b64}, {"v_fma_f32", k_fma_f32},
      {"v_rcp_f32", k_rcp_f32}, {"v_exp_f32", k_exp_f32}, {"v_add_u32", k_add_u32}, {"v_mul_lo_u32", k_mul_lo_u32},
      {"v_cndmask_b32", k_cndmask_b32}, {"v_mov_b32", k_mov_b32}, {"v_cvt_f64_f32", k_cvt_f64_f32},
      {"v_cvt_f64_i32", k_cvt_f64_i32}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double ref_ms = 0;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"wave_instr_per_kernel\": %.0f, \"ops\": {", prop.gcnArchName, cus,
         prop.clockRate / 1000, (double)blocks * 4 * ITERS * 8);
  for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++) {
    hipLaunchKernelGGL(tab[i].k, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);   // warm-up
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(tab[i].k, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) { fprintf(stderr, "kernel %s failed\n", tab[i].name); return 1; }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    if (i == 0) ref_ms = ms;
    // cycles per wavefront instruction per SIMD, from the clock the runtime reports
    double const waves = (double)blocks * 4, simds = cus * 4.0;
    double const cyc = ms * 1e-3 * (prop.clockRate * 1e3) * simds / (waves * ITERS * 8);
    printf("%s\"%s\": {\"ms\": %.3f, \"rel_fma_f64\": %.2f, \"cycles_at_reported_clock\": %.2f}", i ? ", " : "", tab[i].name, ms,
           ms / ref_ms, cyc);
  }
  printf("}}\n");
  return 0;
}
