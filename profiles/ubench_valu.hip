// Issue cost of the vector instructions the exact-arithmetic kernels are made of, on gfx950: a loop of 16 (or 8)
// independent dependency chains of one instruction form per lane, at 1 / 2 / 4 / 8 wavefronts per SIMD on every CU.
// Reported: wave-instructions per second and SIMD, and cycles per instruction at 2.4 GHz.  Inline assembly pins the
// forms (the SLP vectoriser would otherwise pack the scalar ones).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_valu profiles/ubench_valu.hip && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

enum Form { FMA32, PKFMA32, MUL32, ADDU32, CNDMASK, RCP32, SQRT32, FMA64, ADD64, MUL64, RCP64, RSQ64, SQRT64, CVT64_32, CVT32_64,
            CVTI32_64, MULLO, MAD64, LSHL64, CMP64, kForms };
static const char *kNames[kForms] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_add_u32", "v_cndmask_b32", "v_rcp_f32", "v_sqrt_f32",
                                     "v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_cvt_f64_f32",
                                     "v_cvt_f32_f64", "v_cvt_i32_f64", "v_mul_lo_u32", "v_mad_u64_u32", "v_lshlrev_b64", "v_cmp_lt_f64"};

#define OP1(text, c) asm volatile(text : "+v"(c))
template <int kForm>
__global__ __launch_bounds__(256) void k(float *out, float a, float b, int trips) {
  float s[16];
  v2f p[8];
  double d[8];
  unsigned u[16];
  unsigned long long w[8];
  for (int i = 0; i < 16; ++i) {
    s[i] = 1.0f + threadIdx.x * 1e-3f + i;
    u[i] = threadIdx.x + i;
  }
  for (int i = 0; i < 8; ++i) {
    p[i] = v2f{s[2 * i], s[2 * i + 1]};
    d[i] = s[i];
    w[i] = u[i];
  }
  const v2f a2 = {a, a}, b2 = {b, b};
  const double ad = a, bd = b;
  const unsigned au = 3;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int h = i & 7;
      if (kForm == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
      if (kForm == PKFMA32 && i < 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[h]) : "v"(a2), "v"(b2));
      if (kForm == MUL32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
      if (kForm == ADDU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(au));
      if (kForm == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(au));
      if (kForm == RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
      if (kForm == SQRT32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[i]));
      if (kForm == FMA64 && i < 8) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[h]) : "v"(ad), "v"(bd));
      if (kForm == ADD64 && i < 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[h]) : "v"(bd));
      if (kForm == MUL64 && i < 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[h]) : "v"(ad));
      if (kForm == RCP64 && i < 8) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[h]));
      if (kForm == RSQ64 && i < 8) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[h]));
      if (kForm == SQRT64 && i < 8) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[h]));
      if (kForm == CVT64_32 && i < 8) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[h]) : "v"(s[i]));
      if (kForm == CVT32_64 && i < 8) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(s[i]) : "v"(d[h]));
      if (kForm == CVTI32_64 && i < 8) asm volatile("v_cvt_i32_f64 %0, %1" : "+v"(u[i]) : "v"(d[h]));
      if (kForm == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(au));
      if (kForm == MAD64 && i < 8) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[h]) : "v"(u[i]), "v"(au) : "vcc");
      if (kForm == LSHL64 && i < 8) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w[h]));
      if (kForm == CMP64 && i < 8) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[h]), "v"(bd) : "vcc");
    }
  }
  float r = 0;
  for (int i = 0; i < 16; ++i) r += s[i] + u[i];
  for (int i = 0; i < 8; ++i) r += p[i][0] + p[i][1] + static_cast<float>(d[i]) + static_cast<float>(w[i]);
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int kForm>
static double run(float *out, int blocks, int trips) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<kForm>, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 1e-3f, trips);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<kForm>, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 1e-3f, trips);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}

template <int kForm>
static void report(float *out, int waves_per_simd, int trips) {
  const int blocks = 256 * waves_per_simd;  // 4 wavefronts per workgroup: one workgroup per CU and step
  const double ms = run<kForm>(out, blocks, trips);
  const bool wide = kForm == PKFMA32 || (kForm >= FMA64 && kForm != MULLO);
  const double insts = double(blocks) * 4 * trips * (wide ? 8 : 16);  // wave-instructions
  const double per_simd = insts / 1024.0 / (ms * 1e-3);
  std::printf("{\"waves_per_simd\": %d, \"form\": \"%s\", \"ms\": %.4f, \"M_inst_per_s_per_simd\": %.1f, \"cycles_at_2.4GHz\": %.2f}\n",
              waves_per_simd, kNames[kForm], ms, per_simd * 1e-6, 2.4e9 / per_simd);
}

template <int kForm>
static void sweep(float *out, int trips) {
  for (int w : {1, 2, 4, 8}) report<kForm>(out, w, trips);
}

int main() {
  float *out;
  (void)hipMalloc(&out, 256 * 16384 * 4);
  const int trips = 16384;
  sweep<FMA32>(out, trips);
  sweep<PKFMA32>(out, trips);
  sweep<MUL32>(out, trips);
  sweep<ADDU32>(out, trips);
  sweep<CNDMASK>(out, trips);
  sweep<RCP32>(out, trips);
  sweep<SQRT32>(out, trips);
  sweep<FMA64>(out, trips);
  sweep<ADD64>(out, trips);
  sweep<MUL64>(out, trips);
  sweep<RCP64>(out, trips);
  sweep<RSQ64>(out, trips);
  sweep<SQRT64>(out, trips);
  sweep<CVT64_32>(out, trips);
  sweep<CVT32_64>(out, trips);
  sweep<CVTI32_64>(out, trips);
  sweep<MULLO>(out, trips);
  sweep<MAD64>(out, trips);
  sweep<LSHL64>(out, trips);
  sweep<CMP64>(out, trips);
  return 0;
}
