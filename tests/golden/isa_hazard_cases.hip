// Negative / positive controls of tools/check_isa.py::pending_hazards (tests/test_host.py).
// `hazard_good` issues an LDS read through inline asm and waits for it before its first use --
// the idiom of csrc/chain3.hip.  `hazard_broken_*` are what the checker exists to catch: the
// destination of the pending read is COPIED (what a register allocator may do: it believes an
// "=v" output is defined as soon as the asm statement has been issued) or handed to an MFMA
// before the s_waitcnt that covers it, or the counted wait allows one read too many in flight.
// Not part of the library; compiled by the test with hipcc --offload-arch=gfx950 -c.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void hazard_good(const float* in, float* out) {
  __shared__ f32x4 tile[256];
  tile[threadIdx.x] = *reinterpret_cast<const f32x4*>(in + 4 * threadIdx.x);
  __syncthreads();
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4*)&tile[threadIdx.x ^ 1];
  const unsigned b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4*)&tile[threadIdx.x ^ 2];
  f32x4 v, w;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
  asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(b) : "memory");
  asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(v) :: "memory");   // v has landed, w may fly
  f32x4 r = v * 2.f;
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w) :: "memory");
  r += w;
  *reinterpret_cast<f32x4*>(out + 4 * threadIdx.x) = r;
}

// the wait allows ONE read in flight, but the value used is the LAST one issued
__global__ void hazard_broken_count(const float* in, float* out) {
  __shared__ f32x4 tile[256];
  tile[threadIdx.x] = *reinterpret_cast<const f32x4*>(in + 4 * threadIdx.x);
  __syncthreads();
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4*)&tile[threadIdx.x ^ 1];
  const unsigned b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4*)&tile[threadIdx.x ^ 2];
  f32x4 v, w;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
  asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(b) : "memory");
  asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(w) :: "memory");   // wrong: w is the one still in flight
  f32x4 r = w;
  asm volatile("v_add_f32 %0, %1, %1" : "=v"(r.x) : "v"(w.x));  // (pinned in place: reads w.x here)
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) :: "memory");
  r += v;
  *reinterpret_cast<f32x4*>(out + 4 * threadIdx.x) = r;
}

// the destination is copied before the wait (the move an allocator is free to insert)
__global__ void hazard_broken_copy(const float* in, float* out) {
  __shared__ f32x4 tile[256];
  tile[threadIdx.x] = *reinterpret_cast<const f32x4*>(in + 4 * threadIdx.x);
  __syncthreads();
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) f32x4*)&tile[threadIdx.x ^ 1];
  f32x4 v, c;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
  asm volatile("v_mov_b32 %0, %1" : "=v"(c.x) : "v"(v.x));      // reads v.x while it is pending
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) :: "memory");
  c.y = v.y; c.z = v.z; c.w = v.w;
  *reinterpret_cast<f32x4*>(out + 4 * threadIdx.x) = c;
}

// a VMEM load issued through inline asm whose destination is stored before vmcnt covers it
__global__ void hazard_broken_vmem(const float* in, float* out) {
  f32x4 v;
  const float* p = in + 4 * threadIdx.x;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  *reinterpret_cast<f32x4*>(out + 4 * threadIdx.x) = v;          // the compiler does not know it is pending
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
