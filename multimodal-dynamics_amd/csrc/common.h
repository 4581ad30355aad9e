// Shared device/host helpers for libmmdyn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "mmdyn_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Activation storage type of the bf16-storage mode (BASELINE configs[2]): bf16 in HBM, fp32 in registers.
// ld4 / st4 move four consecutive elements (16 bytes of fp32 or 8 bytes of bf16); rounding is RNE (v_cvt_pk_bf16_f32).
typedef uint16_t bf16_t;
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  f32x4 v;
  v[0] = __uint_as_float(u.x << 16);
  v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16);
  v[3] = __uint_as_float(u.y & 0xffff0000u);
  return v;
}
template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return __uint_as_float((uint32_t)*p << 16); }
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 r;
  r[0] = (__bf16)lo;
  r[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, r);
}
// Exact three-term split of two fp32 values into bf16 terms (the X3 variants of igemm_nt_kernel / wgrad_tn_kernel):
// hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid, every conversion round-to-nearest-even (v_cvt_pk_bf16_f32).  Both
// subtractions are exact in fp32 (hi is within half a bf16 ulp of x; the second residual has at most 8 significant bits), so
// hi + mid + lo == x bit for bit, with |mid| <= 2^-8 |x| and |lo| <= 2^-16 |x|.  Of the nine cross products of two split
// operands the kernels keep six (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi); the dropped three are together below
// 2^-23 |a||b| (measured on random data: max 2^-24.2, rms 2^-27.4, zero mean -- less than ONE fp32 rounding of the product,
// tests/test_model_emu.py::test_three_term_split_*).  Each result packs the pair (x0 low half, x1 high half).
// Non-finite inputs: inf - inf = NaN in the residuals, so an Inf operand yields NaN where the fp32 matrix cores would give Inf.
__device__ __forceinline__ void split3_bf16(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = pack2_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = pack2_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = pack2_bf16(s0, s1);
}
// Eight consecutive channels of one row of a PLANE tensor (rows of [plane][C] bf16: hi | mid | lo, the operand format of the GEMMs
// that take their fp32x3 operands already split): three 16-byte stores.  `row_c0` points at channel c0 of the row's first plane.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_planes8(bf16_t* row_c0, int C, const f32x4& a, const f32x4& b) {
  uint32_t h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
  split3_bf16(a[0], a[1], h0, m0, l0);
  split3_bf16(a[2], a[3], h1, m1, l1);
  split3_bf16(b[0], b[1], h2, m2, l2);
  split3_bf16(b[2], b[3], h3, m3, l3);
  *reinterpret_cast<u32x4_t*>(row_c0) = u32x4_t{h0, h1, h2, h3};
  *reinterpret_cast<u32x4_t*>(row_c0 + C) = u32x4_t{m0, m1, m2, m3};
  *reinterpret_cast<u32x4_t*>(row_c0 + 2 * C) = u32x4_t{l0, l1, l2, l3};
}
// ... and back: the fp32 values of eight consecutive channels (exact: hi + mid has at most 16 significant bits, + lo is x)
__device__ __forceinline__ void load_planes8(const bf16_t* row_c0, int C, f32x4& a, f32x4& b) {
  const u32x4_t h = *reinterpret_cast<const u32x4_t*>(row_c0);
  const u32x4_t m = *reinterpret_cast<const u32x4_t*>(row_c0 + C);
  const u32x4_t l = *reinterpret_cast<const u32x4_t*>(row_c0 + 2 * C);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    a[2 * k] = (__uint_as_float(h[k] << 16) + __uint_as_float(m[k] << 16)) + __uint_as_float(l[k] << 16);
    a[2 * k + 1] = (__uint_as_float(h[k] & 0xffff0000u) + __uint_as_float(m[k] & 0xffff0000u)) + __uint_as_float(l[k] & 0xffff0000u);
    b[2 * k] = (__uint_as_float(h[2 + k] << 16) + __uint_as_float(m[2 + k] << 16)) + __uint_as_float(l[2 + k] << 16);
    b[2 * k + 1] = (__uint_as_float(h[2 + k] & 0xffff0000u) + __uint_as_float(m[2 + k] & 0xffff0000u)) + __uint_as_float(l[2 + k] & 0xffff0000u);
  }
}
// Second 16-bit storage type: IEEE half (the fp16-storage mode, BASELINE configs[4]).  A distinct C++ type so that the
// kernels templated on the storage type get their own instances; same size and alignment as bf16_t.
struct half_t { uint16_t v; };
__device__ __forceinline__ float h2f(uint32_t bits16) { return (float)__builtin_bit_cast(_Float16, (uint16_t)bits16); }
__device__ __forceinline__ uint32_t pack2_f16(float lo, float hi) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 r;
  r[0] = (_Float16)lo;
  r[1] = (_Float16)hi;
  return __builtin_bit_cast(uint32_t, r);
}
template <> __device__ __forceinline__ f32x4 ld4<half_t>(const half_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  f32x4 v;
  v[0] = h2f(u.x & 0xffffu);
  v[1] = h2f(u.x >> 16);
  v[2] = h2f(u.y & 0xffffu);
  v[3] = h2f(u.y >> 16);
  return v;
}
template <> __device__ __forceinline__ float ld1<half_t>(const half_t* p) { return h2f(p->v); }
// pack two values into one dword of the storage type T (bf16_t / half_t)
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack2_bf16(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<half_t>(float lo, float hi) { return pack2_f16(lo, hi); }
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f32x4 v) {
  uint2 u;
  u.x = pack2_bf16(v[0], v[1]);
  u.y = pack2_bf16(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = u;
}
template <> __device__ __forceinline__ void st4<half_t>(half_t* p, f32x4 v) {
  uint2 u;
  u.x = pack2_f16(v[0], v[1]);
  u.y = pack2_f16(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = u;
}
// 16-byte accesses for all storage types: VECW<T> elements per access (4 floats or 8 16-bit values), as VECW/4 f32x4 groups
template <typename T> struct vecw { static constexpr int n = 4; };
template <> struct vecw<bf16_t> { static constexpr int n = 8; };
template <> struct vecw<half_t> { static constexpr int n = 8; };
template <typename T> __device__ __forceinline__ void ldv(const T* p, f32x4* v);
template <> __device__ __forceinline__ void ldv<float>(const float* p, f32x4* v) { v[0] = *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ void ldv<bf16_t>(const bf16_t* p, f32x4* v) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  v[0][0] = __uint_as_float(u.x << 16);
  v[0][1] = __uint_as_float(u.x & 0xffff0000u);
  v[0][2] = __uint_as_float(u.y << 16);
  v[0][3] = __uint_as_float(u.y & 0xffff0000u);
  v[1][0] = __uint_as_float(u.z << 16);
  v[1][1] = __uint_as_float(u.z & 0xffff0000u);
  v[1][2] = __uint_as_float(u.w << 16);
  v[1][3] = __uint_as_float(u.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void stv(T* p, const f32x4* v);
template <> __device__ __forceinline__ void stv<float>(float* p, const f32x4* v) { *reinterpret_cast<f32x4*>(p) = v[0]; }
template <> __device__ __forceinline__ void stv<bf16_t>(bf16_t* p, const f32x4* v) {
  uint4 u;
  u.x = pack2_bf16(v[0][0], v[0][1]);
  u.y = pack2_bf16(v[0][2], v[0][3]);
  u.z = pack2_bf16(v[1][0], v[1][1]);
  u.w = pack2_bf16(v[1][2], v[1][3]);
  *reinterpret_cast<uint4*>(p) = u;
}
// The same accesses with the non-temporal hint, for tensors that are streamed once per launch and are larger than the
// L2 (activations of a whole batch): measured +15 % on a 64 MB read-modify-write pass (tests/microbench/corun_stream.hip)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ void ldv_nt(const T* p, f32x4* v);
template <> __device__ __forceinline__ void ldv_nt<float>(const float* p, f32x4* v) {
  v[0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
}
template <> __device__ __forceinline__ void ldv_nt<bf16_t>(const bf16_t* p, f32x4* v) {
  const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  v[0][0] = __uint_as_float(u[0] << 16);
  v[0][1] = __uint_as_float(u[0] & 0xffff0000u);
  v[0][2] = __uint_as_float(u[1] << 16);
  v[0][3] = __uint_as_float(u[1] & 0xffff0000u);
  v[1][0] = __uint_as_float(u[2] << 16);
  v[1][1] = __uint_as_float(u[2] & 0xffff0000u);
  v[1][2] = __uint_as_float(u[3] << 16);
  v[1][3] = __uint_as_float(u[3] & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void stv_nt(T* p, const f32x4* v);
template <> __device__ __forceinline__ void stv_nt<float>(float* p, const f32x4* v) {
  __builtin_nontemporal_store(v[0], reinterpret_cast<f32x4*>(p));
}
template <> __device__ __forceinline__ void stv_nt<bf16_t>(bf16_t* p, const f32x4* v) {
  u32x4 u;
  u[0] = pack2_bf16(v[0][0], v[0][1]);
  u[1] = pack2_bf16(v[0][2], v[0][3]);
  u[2] = pack2_bf16(v[1][0], v[1][1]);
  u[3] = pack2_bf16(v[1][2], v[1][3]);
  __builtin_nontemporal_store(u, reinterpret_cast<u32x4*>(p));
}
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = (bf16_t)(pack2_bf16(v, 0.f) & 0xffffu); }
template <> __device__ __forceinline__ void st1<half_t>(half_t* p, float v) { p->v = (uint16_t)(pack2_f16(v, 0.f) & 0xffffu); }
__device__ __forceinline__ void unpack8_f16(const uint32_t* u, f32x4* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i >> 1][(i & 1) * 2 + 0] = h2f(u[i] & 0xffffu);
    v[i >> 1][(i & 1) * 2 + 1] = h2f(u[i] >> 16);
  }
}
template <> __device__ __forceinline__ void ldv<half_t>(const half_t* p, f32x4* v) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
  unpack8_f16(w, v);
}
template <> __device__ __forceinline__ void stv<half_t>(half_t* p, const f32x4* v) {
  uint4 u;
  u.x = pack2_f16(v[0][0], v[0][1]);
  u.y = pack2_f16(v[0][2], v[0][3]);
  u.z = pack2_f16(v[1][0], v[1][1]);
  u.w = pack2_f16(v[1][2], v[1][3]);
  *reinterpret_cast<uint4*>(p) = u;
}
template <> __device__ __forceinline__ void ldv_nt<half_t>(const half_t* p, f32x4* v) {
  const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  const uint32_t w[4] = {u[0], u[1], u[2], u[3]};
  unpack8_f16(w, v);
}
template <> __device__ __forceinline__ void stv_nt<half_t>(half_t* p, const f32x4* v) {
  u32x4 u;
  u[0] = pack2_f16(v[0][0], v[0][1]);
  u[1] = pack2_f16(v[0][2], v[0][3]);
  u[2] = pack2_f16(v[1][0], v[1][1]);
  u[3] = pack2_f16(v[1][2], v[1][3]);
  __builtin_nontemporal_store(u, reinterpret_cast<u32x4*>(p));
}

// Last-arriving block finishes (cdna_hip_programming.md section 5 "In-launch split-K reduction", Guideline 16 R1): every
// block of a launch stores its partial result WRITE-THROUGH (st_wt: sc1 stores, so no release fence -- a fence would write
// back the whole L2 slice, i.e. the dirty output lines of whatever GEMM the other lane is running: measured +0.16 ms per
// step), drains them (s_waitcnt vmcnt(0) in every storing wave, then the block barrier) and draws a ticket; the block
// that draws the last one returns true and may read every block's partial: one agent-scope acquire (one lane, then
// s_waitcnt vmcnt(0) + the block barrier before the other waves load) drops its CU's stale L1 lines.  Placement-
// independent.  `counter` is a zero-initialised word owned by this launch site (mmdyn_hip/ops.py hands out a slot per
// launch); the last block puts it back to zero, so a HIP-graph replay finds it zero again.  The reducer must read the
// partials through per-lane (vector) loads -- the scalar cache is not covered by the acquire.
__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool last_block_arrives(unsigned* counter, unsigned nblocks, int* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: its write-through partial has left the core
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == nblocks - 1u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *lds_flag = last;
  }
  __syncthreads();
  return *lds_flag != 0;
}

#define MMDYN_LAUNCH_CHECK()                      \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
    return MMDYN_OK;                              \
  } while (0)

// Opt-in for more than 64 KiB of dynamic LDS.  The attribute belongs to one kernel instance ON ONE DEVICE, so a launcher
// keeps one of these per instance (a function-local static) and asks before every launch; a racing second thread only
// repeats an idempotent call.
struct LdsOptIn {
  static constexpr int MAX_DEVICES = 64;
  bool done[MAX_DEVICES] = {};
  int ensure(const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MAX_DEVICES) return MMDYN_ERR_RANGE;
    if (done[dev]) return MMDYN_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done[dev] = true;
    return MMDYN_OK;
  }
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 64-lane wavefront sum (CDNA wave = 64; never 32)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the ~10-instruction IEEE division sequence
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
// d/dx [x*sigmoid(x)] = s * (1 + x*(1-s))
__device__ __forceinline__ float swish_gradf_(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == MMDYN_ACT_SWISH) return swishf_(x);
  if (act == MMDYN_ACT_RELU) return x > 0.f ? x : 0.f;
  return x;
}
__device__ __forceinline__ float act_grad(float x, int act) {
  if (act == MMDYN_ACT_SWISH) return swish_gradf_(x);
  if (act == MMDYN_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  return 1.f;
}

// One element of BCE-with-logits (torch.nn.functional.binary_cross_entropy_with_logits, problems.py:421-428) and its
// derivative from ONE exponential: e = exp(-|x|) in (0, 1]; loss = max(x, 0) - x t + log(1 + e); sigmoid(x) = 1 / (1 + e) for
// x >= 0 and e / (1 + e) below.  Hardware exp / log / reciprocal (v_exp_f32, v_log_f32, v_rcp_f32: ~1 ulp on these ranges; log
// of 1 + e loses at most 6e-8 absolute, against terms of 0.3-0.7): the library log1pf + two expf + a division made this
// 113 MB pass VALU-bound at 2.1 TB/s.
__device__ __forceinline__ void bce_elem(float x, float t, float& loss, float& sig) {
  const float e = __expf(-fabsf(x));
  const float inv = __frcp_rn(__fadd_rn(1.f, e));
  // (explicitly rounded products and sums: every kernel that inlines this evaluates the same expression tree, whatever
  //  contraction the compiler would pick around it)
  loss = __fadd_rn(__fsub_rn(fmaxf(x, 0.f), __fmul_rn(x, t)), __logf(__fadd_rn(1.f, e)));
  sig = x >= 0.f ? inv : __fmul_rn(e, inv);
}

// conv3.hip: direct kernels for the 3-channel k4 s2 p1 layers (MMDYN_IM2COL3 geometry).  Return MMDYN_OK / an error
// code, or 1 when the shape is not theirs (the caller then takes the generic tiled kernel).
int mmdyn_conv3_nt_try(const float* A, const float* Bp, const float* bias, void* C, void* C_act, float* stats, int G,
                       int Bg, int Hi, int Wi, int Ho, int Wo, int N, int ldc, int act, int splitk, const void* bn_y,
                       const float* bn_mean, const float* bn_rstd, const float* bn_gamma, const float* bn_beta,
                       int c_b16, int bny_b16, int b_b16, hipStream_t st);
int mmdyn_conv3_wgrad_try(const void* D, const float* Gt, float* partial, int Bt, int Hr, int Wr, int Cd, int Hi,
                          int Wi, int Cg, int chunks, int d_b16, hipStream_t st, const float* bn_mean = nullptr,
                          const float* bn_rstd = nullptr, const float* bn_gamma = nullptr, const float* bn_beta = nullptr,
                          int Bg = 0);

// Kernel-experiment knobs (environment variables read by the library: docs/LAB_NOTES.md C) exist only in the LAB build
// (make lab -> libmmdyn_hip_lab.so, -DMMDYN_LAB).  The product library has one code path per launch: lab_env() is
// a constant nullptr there and every test of it folds away.
#ifdef MMDYN_LAB
static inline const char* lab_env(const char* name) { return getenv(name); }
#else
static inline const char* lab_env(const char*) { return nullptr; }
#endif

// grid size for a grid-stride element-wise launch: enough blocks to fill 256 CUs x 8, no more
static inline int ew_grid_cap() {
  static const int cap = [] {
    const char* e = lab_env("MMDYN_EW_BLOCKS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 2048;
  }();
  return cap;
}
static inline int ew_grid(int64_t work_items, int block = 256) {
  int64_t b = ceil_div64(work_items, block);
  if (b > ew_grid_cap()) b = ew_grid_cap();
  if (b < 1) b = 1;
  return (int)b;
}
