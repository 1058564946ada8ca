#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(64) void k0(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_and_b32 %0, -4, %0\n v_and_b32 %1, -4, %1\n v_and_b32 %2, -4, %2\n v_and_b32 %3, -4, %3\n v_and_b32 %4, -4, %4\n v_and_b32 %5, -4, %5\n v_and_b32 %6, -4, %6\n v_and_b32 %7, -4, %7\n" "v_and_b32 %0, -4, %0\n v_and_b32 %1, -4, %1\n v_and_b32 %2, -4, %2\n v_and_b32 %3, -4, %3\n v_and_b32 %4, -4, %4\n v_and_b32 %5, -4, %5\n v_and_b32 %6, -4, %6\n v_and_b32 %7, -4, %7\n" "v_and_b32 %0, -4, %0\n v_and_b32 %1, -4, %1\n v_and_b32 %2, -4, %2\n v_and_b32 %3, -4, %3\n v_and_b32 %4, -4, %4\n v_and_b32 %5, -4, %5\n v_and_b32 %6, -4, %6\n v_and_b32 %7, -4, %7\n" "v_and_b32 %0, -4, %0\n v_and_b32 %1, -4, %1\n v_and_b32 %2, -4, %2\n v_and_b32 %3, -4, %3\n v_and_b32 %4, -4, %4\n v_and_b32 %5, -4, %5\n v_and_b32 %6, -4, %6\n v_and_b32 %7, -4, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k1(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %9, %4\n v_and_b32 %5, %9, %5\n v_and_b32 %6, %9, %6\n v_and_b32 %7, %9, %7\n" "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %9, %4\n v_and_b32 %5, %9, %5\n v_and_b32 %6, %9, %6\n v_and_b32 %7, %9, %7\n" "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %9, %4\n v_and_b32 %5, %9, %5\n v_and_b32 %6, %9, %6\n v_and_b32 %7, %9, %7\n" "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %9, %4\n v_and_b32 %5, %9, %5\n v_and_b32 %6, %9, %6\n v_and_b32 %7, %9, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k2(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %9, %4\n v_xor_b32 %5, %9, %5\n v_xor_b32 %6, %9, %6\n v_xor_b32 %7, %9, %7\n" "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %9, %4\n v_xor_b32 %5, %9, %5\n v_xor_b32 %6, %9, %6\n v_xor_b32 %7, %9, %7\n" "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %9, %4\n v_xor_b32 %5, %9, %5\n v_xor_b32 %6, %9, %6\n v_xor_b32 %7, %9, %7\n" "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n v_xor_b32 %4, %9, %4\n v_xor_b32 %5, %9, %5\n v_xor_b32 %6, %9, %6\n v_xor_b32 %7, %9, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k3(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %9\n v_sub_u32 %5, %5, %9\n v_sub_u32 %6, %6, %9\n v_sub_u32 %7, %7, %9\n" "v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %9\n v_sub_u32 %5, %5, %9\n v_sub_u32 %6, %6, %9\n v_sub_u32 %7, %7, %9\n" "v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %9\n v_sub_u32 %5, %5, %9\n v_sub_u32 %6, %6, %9\n v_sub_u32 %7, %7, %9\n" "v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %9\n v_sub_u32 %5, %5, %9\n v_sub_u32 %6, %6, %9\n v_sub_u32 %7, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k4(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add_u32 %0, 17, %0\n v_add_u32 %1, 17, %1\n v_add_u32 %2, 17, %2\n v_add_u32 %3, 17, %3\n v_add_u32 %4, 17, %4\n v_add_u32 %5, 17, %5\n v_add_u32 %6, 17, %6\n v_add_u32 %7, 17, %7\n" "v_add_u32 %0, 17, %0\n v_add_u32 %1, 17, %1\n v_add_u32 %2, 17, %2\n v_add_u32 %3, 17, %3\n v_add_u32 %4, 17, %4\n v_add_u32 %5, 17, %5\n v_add_u32 %6, 17, %6\n v_add_u32 %7, 17, %7\n" "v_add_u32 %0, 17, %0\n v_add_u32 %1, 17, %1\n v_add_u32 %2, 17, %2\n v_add_u32 %3, 17, %3\n v_add_u32 %4, 17, %4\n v_add_u32 %5, 17, %5\n v_add_u32 %6, 17, %6\n v_add_u32 %7, 17, %7\n" "v_add_u32 %0, 17, %0\n v_add_u32 %1, 17, %1\n v_add_u32 %2, 17, %2\n v_add_u32 %3, 17, %3\n v_add_u32 %4, 17, %4\n v_add_u32 %5, 17, %5\n v_add_u32 %6, 17, %6\n v_add_u32 %7, 17, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k5(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3\n v_add_u32 %4, 0x12345, %4\n v_add_u32 %5, 0x12345, %5\n v_add_u32 %6, 0x12345, %6\n v_add_u32 %7, 0x12345, %7\n" "v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3\n v_add_u32 %4, 0x12345, %4\n v_add_u32 %5, 0x12345, %5\n v_add_u32 %6, 0x12345, %6\n v_add_u32 %7, 0x12345, %7\n" "v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3\n v_add_u32 %4, 0x12345, %4\n v_add_u32 %5, 0x12345, %5\n v_add_u32 %6, 0x12345, %6\n v_add_u32 %7, 0x12345, %7\n" "v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3\n v_add_u32 %4, 0x12345, %4\n v_add_u32 %5, 0x12345, %5\n v_add_u32 %6, 0x12345, %6\n v_add_u32 %7, 0x12345, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k6(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 2, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3\n v_lshlrev_b32 %4, 2, %4\n v_lshlrev_b32 %5, 2, %5\n v_lshlrev_b32 %6, 2, %6\n v_lshlrev_b32 %7, 2, %7\n" "v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 2, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3\n v_lshlrev_b32 %4, 2, %4\n v_lshlrev_b32 %5, 2, %5\n v_lshlrev_b32 %6, 2, %6\n v_lshlrev_b32 %7, 2, %7\n" "v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 2, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3\n v_lshlrev_b32 %4, 2, %4\n v_lshlrev_b32 %5, 2, %5\n v_lshlrev_b32 %6, 2, %6\n v_lshlrev_b32 %7, 2, %7\n" "v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 2, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3\n v_lshlrev_b32 %4, 2, %4\n v_lshlrev_b32 %5, 2, %5\n v_lshlrev_b32 %6, 2, %6\n v_lshlrev_b32 %7, 2, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k7(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_lshrrev_b32 %0, 2, %0\n v_lshrrev_b32 %1, 2, %1\n v_lshrrev_b32 %2, 2, %2\n v_lshrrev_b32 %3, 2, %3\n v_lshrrev_b32 %4, 2, %4\n v_lshrrev_b32 %5, 2, %5\n v_lshrrev_b32 %6, 2, %6\n v_lshrrev_b32 %7, 2, %7\n" "v_lshrrev_b32 %0, 2, %0\n v_lshrrev_b32 %1, 2, %1\n v_lshrrev_b32 %2, 2, %2\n v_lshrrev_b32 %3, 2, %3\n v_lshrrev_b32 %4, 2, %4\n v_lshrrev_b32 %5, 2, %5\n v_lshrrev_b32 %6, 2, %6\n v_lshrrev_b32 %7, 2, %7\n" "v_lshrrev_b32 %0, 2, %0\n v_lshrrev_b32 %1, 2, %1\n v_lshrrev_b32 %2, 2, %2\n v_lshrrev_b32 %3, 2, %3\n v_lshrrev_b32 %4, 2, %4\n v_lshrrev_b32 %5, 2, %5\n v_lshrrev_b32 %6, 2, %6\n v_lshrrev_b32 %7, 2, %7\n" "v_lshrrev_b32 %0, 2, %0\n v_lshrrev_b32 %1, 2, %1\n v_lshrrev_b32 %2, 2, %2\n v_lshrrev_b32 %3, 2, %3\n v_lshrrev_b32 %4, 2, %4\n v_lshrrev_b32 %5, 2, %5\n v_lshrrev_b32 %6, 2, %6\n v_lshrrev_b32 %7, 2, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k8(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_ashrrev_i32 %0, 2, %0\n v_ashrrev_i32 %1, 2, %1\n v_ashrrev_i32 %2, 2, %2\n v_ashrrev_i32 %3, 2, %3\n v_ashrrev_i32 %4, 2, %4\n v_ashrrev_i32 %5, 2, %5\n v_ashrrev_i32 %6, 2, %6\n v_ashrrev_i32 %7, 2, %7\n" "v_ashrrev_i32 %0, 2, %0\n v_ashrrev_i32 %1, 2, %1\n v_ashrrev_i32 %2, 2, %2\n v_ashrrev_i32 %3, 2, %3\n v_ashrrev_i32 %4, 2, %4\n v_ashrrev_i32 %5, 2, %5\n v_ashrrev_i32 %6, 2, %6\n v_ashrrev_i32 %7, 2, %7\n" "v_ashrrev_i32 %0, 2, %0\n v_ashrrev_i32 %1, 2, %1\n v_ashrrev_i32 %2, 2, %2\n v_ashrrev_i32 %3, 2, %3\n v_ashrrev_i32 %4, 2, %4\n v_ashrrev_i32 %5, 2, %5\n v_ashrrev_i32 %6, 2, %6\n v_ashrrev_i32 %7, 2, %7\n" "v_ashrrev_i32 %0, 2, %0\n v_ashrrev_i32 %1, 2, %1\n v_ashrrev_i32 %2, 2, %2\n v_ashrrev_i32 %3, 2, %3\n v_ashrrev_i32 %4, 2, %4\n v_ashrrev_i32 %5, 2, %5\n v_ashrrev_i32 %6, 2, %6\n v_ashrrev_i32 %7, 2, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k9(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9\n" "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9\n" "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9\n" "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %9\n v_mov_b32 %5, %9\n v_mov_b32 %6, %9\n v_mov_b32 %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k10(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n" "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n" "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n" "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k11(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n v_min_i32 %4, %4, %9\n v_min_i32 %5, %5, %9\n v_min_i32 %6, %6, %9\n v_min_i32 %7, %7, %9\n" "v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n v_min_i32 %4, %4, %9\n v_min_i32 %5, %5, %9\n v_min_i32 %6, %6, %9\n v_min_i32 %7, %7, %9\n" "v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n v_min_i32 %4, %4, %9\n v_min_i32 %5, %5, %9\n v_min_i32 %6, %6, %9\n v_min_i32 %7, %7, %9\n" "v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n v_min_i32 %4, %4, %9\n v_min_i32 %5, %5, %9\n v_min_i32 %6, %6, %9\n v_min_i32 %7, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k12(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n v_max_u32 %4, %4, %9\n v_max_u32 %5, %5, %9\n v_max_u32 %6, %6, %9\n v_max_u32 %7, %7, %9\n" "v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n v_max_u32 %4, %4, %9\n v_max_u32 %5, %5, %9\n v_max_u32 %6, %6, %9\n v_max_u32 %7, %7, %9\n" "v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n v_max_u32 %4, %4, %9\n v_max_u32 %5, %5, %9\n v_max_u32 %6, %6, %9\n v_max_u32 %7, %7, %9\n" "v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n v_max_u32 %4, %4, %9\n v_max_u32 %5, %5, %9\n v_max_u32 %6, %6, %9\n v_max_u32 %7, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k13(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %9\n v_max_f32 %5, %5, %9\n v_max_f32 %6, %6, %9\n v_max_f32 %7, %7, %9\n" "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %9\n v_max_f32 %5, %5, %9\n v_max_f32 %6, %6, %9\n v_max_f32 %7, %7, %9\n" "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %9\n v_max_f32 %5, %5, %9\n v_max_f32 %6, %6, %9\n v_max_f32 %7, %7, %9\n" "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %9\n v_max_f32 %5, %5, %9\n v_max_f32 %6, %6, %9\n v_max_f32 %7, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k14(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %9\n v_min_f32 %5, %5, %9\n v_min_f32 %6, %6, %9\n v_min_f32 %7, %7, %9\n" "v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %9\n v_min_f32 %5, %5, %9\n v_min_f32 %6, %6, %9\n v_min_f32 %7, %7, %9\n" "v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %9\n v_min_f32 %5, %5, %9\n v_min_f32 %6, %6, %9\n v_min_f32 %7, %7, %9\n" "v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %9\n v_min_f32 %5, %5, %9\n v_min_f32 %6, %6, %9\n v_min_f32 %7, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k15(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n" "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n" "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n" "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k16(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %9, %8\n v_max3_f32 %5, %5, %9, %8\n v_max3_f32 %6, %6, %9, %8\n v_max3_f32 %7, %7, %9, %8\n" "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %9, %8\n v_max3_f32 %5, %5, %9, %8\n v_max3_f32 %6, %6, %9, %8\n v_max3_f32 %7, %7, %9, %8\n" "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %9, %8\n v_max3_f32 %5, %5, %9, %8\n v_max3_f32 %6, %6, %9, %8\n v_max3_f32 %7, %7, %9, %8\n" "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %9, %8\n v_max3_f32 %5, %5, %9, %8\n v_max3_f32 %6, %6, %9, %8\n v_max3_f32 %7, %7, %9, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k17(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_med3_i32 %0, %0, %8, %9\n v_med3_i32 %1, %1, %8, %9\n v_med3_i32 %2, %2, %8, %9\n v_med3_i32 %3, %3, %8, %9\n v_med3_i32 %4, %4, %9, %8\n v_med3_i32 %5, %5, %9, %8\n v_med3_i32 %6, %6, %9, %8\n v_med3_i32 %7, %7, %9, %8\n" "v_med3_i32 %0, %0, %8, %9\n v_med3_i32 %1, %1, %8, %9\n v_med3_i32 %2, %2, %8, %9\n v_med3_i32 %3, %3, %8, %9\n v_med3_i32 %4, %4, %9, %8\n v_med3_i32 %5, %5, %9, %8\n v_med3_i32 %6, %6, %9, %8\n v_med3_i32 %7, %7, %9, %8\n" "v_med3_i32 %0, %0, %8, %9\n v_med3_i32 %1, %1, %8, %9\n v_med3_i32 %2, %2, %8, %9\n v_med3_i32 %3, %3, %8, %9\n v_med3_i32 %4, %4, %9, %8\n v_med3_i32 %5, %5, %9, %8\n v_med3_i32 %6, %6, %9, %8\n v_med3_i32 %7, %7, %9, %8\n" "v_med3_i32 %0, %0, %8, %9\n v_med3_i32 %1, %1, %8, %9\n v_med3_i32 %2, %2, %8, %9\n v_med3_i32 %3, %3, %8, %9\n v_med3_i32 %4, %4, %9, %8\n v_med3_i32 %5, %5, %9, %8\n v_med3_i32 %6, %6, %9, %8\n v_med3_i32 %7, %7, %9, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k18(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %9\n v_lshl_add_u32 %5, %5, 2, %9\n v_lshl_add_u32 %6, %6, 2, %9\n v_lshl_add_u32 %7, %7, 2, %9\n" "v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %9\n v_lshl_add_u32 %5, %5, 2, %9\n v_lshl_add_u32 %6, %6, 2, %9\n v_lshl_add_u32 %7, %7, 2, %9\n" "v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %9\n v_lshl_add_u32 %5, %5, 2, %9\n v_lshl_add_u32 %6, %6, 2, %9\n v_lshl_add_u32 %7, %7, 2, %9\n" "v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %9\n v_lshl_add_u32 %5, %5, 2, %9\n v_lshl_add_u32 %6, %6, 2, %9\n v_lshl_add_u32 %7, %7, 2, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k19(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_lshl_or_b32 %0, %0, 2, %8\n v_lshl_or_b32 %1, %1, 2, %8\n v_lshl_or_b32 %2, %2, 2, %8\n v_lshl_or_b32 %3, %3, 2, %8\n v_lshl_or_b32 %4, %4, 2, %9\n v_lshl_or_b32 %5, %5, 2, %9\n v_lshl_or_b32 %6, %6, 2, %9\n v_lshl_or_b32 %7, %7, 2, %9\n" "v_lshl_or_b32 %0, %0, 2, %8\n v_lshl_or_b32 %1, %1, 2, %8\n v_lshl_or_b32 %2, %2, 2, %8\n v_lshl_or_b32 %3, %3, 2, %8\n v_lshl_or_b32 %4, %4, 2, %9\n v_lshl_or_b32 %5, %5, 2, %9\n v_lshl_or_b32 %6, %6, 2, %9\n v_lshl_or_b32 %7, %7, 2, %9\n" "v_lshl_or_b32 %0, %0, 2, %8\n v_lshl_or_b32 %1, %1, 2, %8\n v_lshl_or_b32 %2, %2, 2, %8\n v_lshl_or_b32 %3, %3, 2, %8\n v_lshl_or_b32 %4, %4, 2, %9\n v_lshl_or_b32 %5, %5, 2, %9\n v_lshl_or_b32 %6, %6, 2, %9\n v_lshl_or_b32 %7, %7, 2, %9\n" "v_lshl_or_b32 %0, %0, 2, %8\n v_lshl_or_b32 %1, %1, 2, %8\n v_lshl_or_b32 %2, %2, 2, %8\n v_lshl_or_b32 %3, %3, 2, %8\n v_lshl_or_b32 %4, %4, 2, %9\n v_lshl_or_b32 %5, %5, 2, %9\n v_lshl_or_b32 %6, %6, 2, %9\n v_lshl_or_b32 %7, %7, 2, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k20(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n v_or3_b32 %4, %4, %9, %8\n v_or3_b32 %5, %5, %9, %8\n v_or3_b32 %6, %6, %9, %8\n v_or3_b32 %7, %7, %9, %8\n" "v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n v_or3_b32 %4, %4, %9, %8\n v_or3_b32 %5, %5, %9, %8\n v_or3_b32 %6, %6, %9, %8\n v_or3_b32 %7, %7, %9, %8\n" "v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n v_or3_b32 %4, %4, %9, %8\n v_or3_b32 %5, %5, %9, %8\n v_or3_b32 %6, %6, %9, %8\n v_or3_b32 %7, %7, %9, %8\n" "v_or3_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_or3_b32 %2, %2, %8, %9\n v_or3_b32 %3, %3, %8, %9\n v_or3_b32 %4, %4, %9, %8\n v_or3_b32 %5, %5, %9, %8\n v_or3_b32 %6, %6, %9, %8\n v_or3_b32 %7, %7, %9, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k21(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_xad_u32 %0, %0, %8, %9\n v_xad_u32 %1, %1, %8, %9\n v_xad_u32 %2, %2, %8, %9\n v_xad_u32 %3, %3, %8, %9\n v_xad_u32 %4, %4, %9, %8\n v_xad_u32 %5, %5, %9, %8\n v_xad_u32 %6, %6, %9, %8\n v_xad_u32 %7, %7, %9, %8\n" "v_xad_u32 %0, %0, %8, %9\n v_xad_u32 %1, %1, %8, %9\n v_xad_u32 %2, %2, %8, %9\n v_xad_u32 %3, %3, %8, %9\n v_xad_u32 %4, %4, %9, %8\n v_xad_u32 %5, %5, %9, %8\n v_xad_u32 %6, %6, %9, %8\n v_xad_u32 %7, %7, %9, %8\n" "v_xad_u32 %0, %0, %8, %9\n v_xad_u32 %1, %1, %8, %9\n v_xad_u32 %2, %2, %8, %9\n v_xad_u32 %3, %3, %8, %9\n v_xad_u32 %4, %4, %9, %8\n v_xad_u32 %5, %5, %9, %8\n v_xad_u32 %6, %6, %9, %8\n v_xad_u32 %7, %7, %9, %8\n" "v_xad_u32 %0, %0, %8, %9\n v_xad_u32 %1, %1, %8, %9\n v_xad_u32 %2, %2, %8, %9\n v_xad_u32 %3, %3, %8, %9\n v_xad_u32 %4, %4, %9, %8\n v_xad_u32 %5, %5, %9, %8\n v_xad_u32 %6, %6, %9, %8\n v_xad_u32 %7, %7, %9, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k22(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %9, %8\n v_perm_b32 %5, %5, %9, %8\n v_perm_b32 %6, %6, %9, %8\n v_perm_b32 %7, %7, %9, %8\n" "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %9, %8\n v_perm_b32 %5, %5, %9, %8\n v_perm_b32 %6, %6, %9, %8\n v_perm_b32 %7, %7, %9, %8\n" "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %9, %8\n v_perm_b32 %5, %5, %9, %8\n v_perm_b32 %6, %6, %9, %8\n v_perm_b32 %7, %7, %9, %8\n" "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %9, %8\n v_perm_b32 %5, %5, %9, %8\n v_perm_b32 %6, %6, %9, %8\n v_perm_b32 %7, %7, %9, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k23(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n v_mad_i32_i24 %4, %4, %9, %8\n v_mad_i32_i24 %5, %5, %9, %8\n v_mad_i32_i24 %6, %6, %9, %8\n v_mad_i32_i24 %7, %7, %9, %8\n" "v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n v_mad_i32_i24 %4, %4, %9, %8\n v_mad_i32_i24 %5, %5, %9, %8\n v_mad_i32_i24 %6, %6, %9, %8\n v_mad_i32_i24 %7, %7, %9, %8\n" "v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n v_mad_i32_i24 %4, %4, %9, %8\n v_mad_i32_i24 %5, %5, %9, %8\n v_mad_i32_i24 %6, %6, %9, %8\n v_mad_i32_i24 %7, %7, %9, %8\n" "v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n v_mad_i32_i24 %4, %4, %9, %8\n v_mad_i32_i24 %5, %5, %9, %8\n v_mad_i32_i24 %6, %6, %9, %8\n v_mad_i32_i24 %7, %7, %9, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k24(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %9\n v_mul_u32_u24 %5, %5, %9\n v_mul_u32_u24 %6, %6, %9\n v_mul_u32_u24 %7, %7, %9\n" "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %9\n v_mul_u32_u24 %5, %5, %9\n v_mul_u32_u24 %6, %6, %9\n v_mul_u32_u24 %7, %7, %9\n" "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %9\n v_mul_u32_u24 %5, %5, %9\n v_mul_u32_u24 %6, %6, %9\n v_mul_u32_u24 %7, %7, %9\n" "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %9\n v_mul_u32_u24 %5, %5, %9\n v_mul_u32_u24 %6, %6, %9\n v_mul_u32_u24 %7, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k25(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_bfe_u32 %0, %0, 2, 8\n v_bfe_u32 %1, %1, 2, 8\n v_bfe_u32 %2, %2, 2, 8\n v_bfe_u32 %3, %3, 2, 8\n v_bfe_u32 %4, %4, 2, 8\n v_bfe_u32 %5, %5, 2, 8\n v_bfe_u32 %6, %6, 2, 8\n v_bfe_u32 %7, %7, 2, 8\n" "v_bfe_u32 %0, %0, 2, 8\n v_bfe_u32 %1, %1, 2, 8\n v_bfe_u32 %2, %2, 2, 8\n v_bfe_u32 %3, %3, 2, 8\n v_bfe_u32 %4, %4, 2, 8\n v_bfe_u32 %5, %5, 2, 8\n v_bfe_u32 %6, %6, 2, 8\n v_bfe_u32 %7, %7, 2, 8\n" "v_bfe_u32 %0, %0, 2, 8\n v_bfe_u32 %1, %1, 2, 8\n v_bfe_u32 %2, %2, 2, 8\n v_bfe_u32 %3, %3, 2, 8\n v_bfe_u32 %4, %4, 2, 8\n v_bfe_u32 %5, %5, 2, 8\n v_bfe_u32 %6, %6, 2, 8\n v_bfe_u32 %7, %7, 2, 8\n" "v_bfe_u32 %0, %0, 2, 8\n v_bfe_u32 %1, %1, 2, 8\n v_bfe_u32 %2, %2, 2, 8\n v_bfe_u32 %3, %3, 2, 8\n v_bfe_u32 %4, %4, 2, 8\n v_bfe_u32 %5, %5, 2, 8\n v_bfe_u32 %6, %6, 2, 8\n v_bfe_u32 %7, %7, 2, 8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k26(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_sad_u32 %0, %0, %8, %9\n v_sad_u32 %1, %1, %8, %9\n v_sad_u32 %2, %2, %8, %9\n v_sad_u32 %3, %3, %8, %9\n v_sad_u32 %4, %4, %9, %8\n v_sad_u32 %5, %5, %9, %8\n v_sad_u32 %6, %6, %9, %8\n v_sad_u32 %7, %7, %9, %8\n" "v_sad_u32 %0, %0, %8, %9\n v_sad_u32 %1, %1, %8, %9\n v_sad_u32 %2, %2, %8, %9\n v_sad_u32 %3, %3, %8, %9\n v_sad_u32 %4, %4, %9, %8\n v_sad_u32 %5, %5, %9, %8\n v_sad_u32 %6, %6, %9, %8\n v_sad_u32 %7, %7, %9, %8\n" "v_sad_u32 %0, %0, %8, %9\n v_sad_u32 %1, %1, %8, %9\n v_sad_u32 %2, %2, %8, %9\n v_sad_u32 %3, %3, %8, %9\n v_sad_u32 %4, %4, %9, %8\n v_sad_u32 %5, %5, %9, %8\n v_sad_u32 %6, %6, %9, %8\n v_sad_u32 %7, %7, %9, %8\n" "v_sad_u32 %0, %0, %8, %9\n v_sad_u32 %1, %1, %8, %9\n v_sad_u32 %2, %2, %8, %9\n v_sad_u32 %3, %3, %8, %9\n v_sad_u32 %4, %4, %9, %8\n v_sad_u32 %5, %5, %9, %8\n v_sad_u32 %6, %6, %9, %8\n v_sad_u32 %7, %7, %9, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k27(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_max_i16 %0, %0, %8\n v_max_i16 %1, %1, %8\n v_max_i16 %2, %2, %8\n v_max_i16 %3, %3, %8\n v_max_i16 %4, %4, %9\n v_max_i16 %5, %5, %9\n v_max_i16 %6, %6, %9\n v_max_i16 %7, %7, %9\n" "v_max_i16 %0, %0, %8\n v_max_i16 %1, %1, %8\n v_max_i16 %2, %2, %8\n v_max_i16 %3, %3, %8\n v_max_i16 %4, %4, %9\n v_max_i16 %5, %5, %9\n v_max_i16 %6, %6, %9\n v_max_i16 %7, %7, %9\n" "v_max_i16 %0, %0, %8\n v_max_i16 %1, %1, %8\n v_max_i16 %2, %2, %8\n v_max_i16 %3, %3, %8\n v_max_i16 %4, %4, %9\n v_max_i16 %5, %5, %9\n v_max_i16 %6, %6, %9\n v_max_i16 %7, %7, %9\n" "v_max_i16 %0, %0, %8\n v_max_i16 %1, %1, %8\n v_max_i16 %2, %2, %8\n v_max_i16 %3, %3, %8\n v_max_i16 %4, %4, %9\n v_max_i16 %5, %5, %9\n v_max_i16 %6, %6, %9\n v_max_i16 %7, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k28(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add_u16 %0, %0, %8\n v_add_u16 %1, %1, %8\n v_add_u16 %2, %2, %8\n v_add_u16 %3, %3, %8\n v_add_u16 %4, %4, %9\n v_add_u16 %5, %5, %9\n v_add_u16 %6, %6, %9\n v_add_u16 %7, %7, %9\n" "v_add_u16 %0, %0, %8\n v_add_u16 %1, %1, %8\n v_add_u16 %2, %2, %8\n v_add_u16 %3, %3, %8\n v_add_u16 %4, %4, %9\n v_add_u16 %5, %5, %9\n v_add_u16 %6, %6, %9\n v_add_u16 %7, %7, %9\n" "v_add_u16 %0, %0, %8\n v_add_u16 %1, %1, %8\n v_add_u16 %2, %2, %8\n v_add_u16 %3, %3, %8\n v_add_u16 %4, %4, %9\n v_add_u16 %5, %5, %9\n v_add_u16 %6, %6, %9\n v_add_u16 %7, %7, %9\n" "v_add_u16 %0, %0, %8\n v_add_u16 %1, %1, %8\n v_add_u16 %2, %2, %8\n v_add_u16 %3, %3, %8\n v_add_u16 %4, %4, %9\n v_add_u16 %5, %5, %9\n v_add_u16 %6, %6, %9\n v_add_u16 %7, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k29(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_pk_max_u16 %0, %0, %8\n v_pk_max_u16 %1, %1, %8\n v_pk_max_u16 %2, %2, %8\n v_pk_max_u16 %3, %3, %8\n v_pk_max_u16 %4, %4, %9\n v_pk_max_u16 %5, %5, %9\n v_pk_max_u16 %6, %6, %9\n v_pk_max_u16 %7, %7, %9\n" "v_pk_max_u16 %0, %0, %8\n v_pk_max_u16 %1, %1, %8\n v_pk_max_u16 %2, %2, %8\n v_pk_max_u16 %3, %3, %8\n v_pk_max_u16 %4, %4, %9\n v_pk_max_u16 %5, %5, %9\n v_pk_max_u16 %6, %6, %9\n v_pk_max_u16 %7, %7, %9\n" "v_pk_max_u16 %0, %0, %8\n v_pk_max_u16 %1, %1, %8\n v_pk_max_u16 %2, %2, %8\n v_pk_max_u16 %3, %3, %8\n v_pk_max_u16 %4, %4, %9\n v_pk_max_u16 %5, %5, %9\n v_pk_max_u16 %6, %6, %9\n v_pk_max_u16 %7, %7, %9\n" "v_pk_max_u16 %0, %0, %8\n v_pk_max_u16 %1, %1, %8\n v_pk_max_u16 %2, %2, %8\n v_pk_max_u16 %3, %3, %8\n v_pk_max_u16 %4, %4, %9\n v_pk_max_u16 %5, %5, %9\n v_pk_max_u16 %6, %6, %9\n v_pk_max_u16 %7, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k30(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_pk_lshrrev_b16 %0, 2, %0\n v_pk_lshrrev_b16 %1, 2, %1\n v_pk_lshrrev_b16 %2, 2, %2\n v_pk_lshrrev_b16 %3, 2, %3\n v_pk_lshrrev_b16 %4, 2, %4\n v_pk_lshrrev_b16 %5, 2, %5\n v_pk_lshrrev_b16 %6, 2, %6\n v_pk_lshrrev_b16 %7, 2, %7\n" "v_pk_lshrrev_b16 %0, 2, %0\n v_pk_lshrrev_b16 %1, 2, %1\n v_pk_lshrrev_b16 %2, 2, %2\n v_pk_lshrrev_b16 %3, 2, %3\n v_pk_lshrrev_b16 %4, 2, %4\n v_pk_lshrrev_b16 %5, 2, %5\n v_pk_lshrrev_b16 %6, 2, %6\n v_pk_lshrrev_b16 %7, 2, %7\n" "v_pk_lshrrev_b16 %0, 2, %0\n v_pk_lshrrev_b16 %1, 2, %1\n v_pk_lshrrev_b16 %2, 2, %2\n v_pk_lshrrev_b16 %3, 2, %3\n v_pk_lshrrev_b16 %4, 2, %4\n v_pk_lshrrev_b16 %5, 2, %5\n v_pk_lshrrev_b16 %6, 2, %6\n v_pk_lshrrev_b16 %7, 2, %7\n" "v_pk_lshrrev_b16 %0, 2, %0\n v_pk_lshrrev_b16 %1, 2, %1\n v_pk_lshrrev_b16 %2, 2, %2\n v_pk_lshrrev_b16 %3, 2, %3\n v_pk_lshrrev_b16 %4, 2, %4\n v_pk_lshrrev_b16 %5, 2, %5\n v_pk_lshrrev_b16 %6, 2, %6\n v_pk_lshrrev_b16 %7, 2, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k31(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_cmp_gt_i32 vcc, %0, %8\n v_cmp_gt_i32 vcc, %1, %8\n v_cmp_gt_i32 vcc, %2, %8\n v_cmp_gt_i32 vcc, %3, %8\n v_cmp_gt_i32 vcc, %4, %9\n v_cmp_gt_i32 vcc, %5, %9\n v_cmp_gt_i32 vcc, %6, %9\n v_cmp_gt_i32 vcc, %7, %9\n" "v_cmp_gt_i32 vcc, %0, %8\n v_cmp_gt_i32 vcc, %1, %8\n v_cmp_gt_i32 vcc, %2, %8\n v_cmp_gt_i32 vcc, %3, %8\n v_cmp_gt_i32 vcc, %4, %9\n v_cmp_gt_i32 vcc, %5, %9\n v_cmp_gt_i32 vcc, %6, %9\n v_cmp_gt_i32 vcc, %7, %9\n" "v_cmp_gt_i32 vcc, %0, %8\n v_cmp_gt_i32 vcc, %1, %8\n v_cmp_gt_i32 vcc, %2, %8\n v_cmp_gt_i32 vcc, %3, %8\n v_cmp_gt_i32 vcc, %4, %9\n v_cmp_gt_i32 vcc, %5, %9\n v_cmp_gt_i32 vcc, %6, %9\n v_cmp_gt_i32 vcc, %7, %9\n" "v_cmp_gt_i32 vcc, %0, %8\n v_cmp_gt_i32 vcc, %1, %8\n v_cmp_gt_i32 vcc, %2, %8\n v_cmp_gt_i32 vcc, %3, %8\n v_cmp_gt_i32 vcc, %4, %9\n v_cmp_gt_i32 vcc, %5, %9\n v_cmp_gt_i32 vcc, %6, %9\n v_cmp_gt_i32 vcc, %7, %9\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k32(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %9, vcc\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_addc_co_u32 %6, vcc, %6, %9, vcc\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n" "v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %9, vcc\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_addc_co_u32 %6, vcc, %6, %9, vcc\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n" "v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %9, vcc\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_addc_co_u32 %6, vcc, %6, %9, vcc\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n" "v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %9, vcc\n v_addc_co_u32 %5, vcc, %5, %9, vcc\n v_addc_co_u32 %6, vcc, %6, %9, vcc\n v_addc_co_u32 %7, vcc, %7, %9, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k33(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_subrev_u32 %0, %8, %0\n v_subrev_u32 %1, %8, %1\n v_subrev_u32 %2, %8, %2\n v_subrev_u32 %3, %8, %3\n v_subrev_u32 %4, %9, %4\n v_subrev_u32 %5, %9, %5\n v_subrev_u32 %6, %9, %6\n v_subrev_u32 %7, %9, %7\n" "v_subrev_u32 %0, %8, %0\n v_subrev_u32 %1, %8, %1\n v_subrev_u32 %2, %8, %2\n v_subrev_u32 %3, %8, %3\n v_subrev_u32 %4, %9, %4\n v_subrev_u32 %5, %9, %5\n v_subrev_u32 %6, %9, %6\n v_subrev_u32 %7, %9, %7\n" "v_subrev_u32 %0, %8, %0\n v_subrev_u32 %1, %8, %1\n v_subrev_u32 %2, %8, %2\n v_subrev_u32 %3, %8, %3\n v_subrev_u32 %4, %9, %4\n v_subrev_u32 %5, %9, %5\n v_subrev_u32 %6, %9, %6\n v_subrev_u32 %7, %9, %7\n" "v_subrev_u32 %0, %8, %0\n v_subrev_u32 %1, %8, %1\n v_subrev_u32 %2, %8, %2\n v_subrev_u32 %3, %8, %3\n v_subrev_u32 %4, %9, %4\n v_subrev_u32 %5, %9, %5\n v_subrev_u32 %6, %9, %6\n v_subrev_u32 %7, %9, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k34(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7\n" "v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7\n" "v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7\n" "v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k35(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_or_b32 %0, %8, %0\n v_or_b32 %1, %8, %1\n v_or_b32 %2, %8, %2\n v_or_b32 %3, %8, %3\n v_or_b32 %4, %9, %4\n v_or_b32 %5, %9, %5\n v_or_b32 %6, %9, %6\n v_or_b32 %7, %9, %7\n" "v_or_b32 %0, %8, %0\n v_or_b32 %1, %8, %1\n v_or_b32 %2, %8, %2\n v_or_b32 %3, %8, %3\n v_or_b32 %4, %9, %4\n v_or_b32 %5, %9, %5\n v_or_b32 %6, %9, %6\n v_or_b32 %7, %9, %7\n" "v_or_b32 %0, %8, %0\n v_or_b32 %1, %8, %1\n v_or_b32 %2, %8, %2\n v_or_b32 %3, %8, %3\n v_or_b32 %4, %9, %4\n v_or_b32 %5, %9, %5\n v_or_b32 %6, %9, %6\n v_or_b32 %7, %9, %7\n" "v_or_b32 %0, %8, %0\n v_or_b32 %1, %8, %1\n v_or_b32 %2, %8, %2\n v_or_b32 %3, %8, %3\n v_or_b32 %4, %9, %4\n v_or_b32 %5, %9, %5\n v_or_b32 %6, %9, %6\n v_or_b32 %7, %9, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k36(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_max_i32 %0, 5, %0\n v_max_i32 %1, 5, %1\n v_max_i32 %2, 5, %2\n v_max_i32 %3, 5, %3\n v_max_i32 %4, 5, %4\n v_max_i32 %5, 5, %5\n v_max_i32 %6, 5, %6\n v_max_i32 %7, 5, %7\n" "v_max_i32 %0, 5, %0\n v_max_i32 %1, 5, %1\n v_max_i32 %2, 5, %2\n v_max_i32 %3, 5, %3\n v_max_i32 %4, 5, %4\n v_max_i32 %5, 5, %5\n v_max_i32 %6, 5, %6\n v_max_i32 %7, 5, %7\n" "v_max_i32 %0, 5, %0\n v_max_i32 %1, 5, %1\n v_max_i32 %2, 5, %2\n v_max_i32 %3, 5, %3\n v_max_i32 %4, 5, %4\n v_max_i32 %5, 5, %5\n v_max_i32 %6, 5, %6\n v_max_i32 %7, 5, %7\n" "v_max_i32 %0, 5, %0\n v_max_i32 %1, 5, %1\n v_max_i32 %2, 5, %2\n v_max_i32 %3, 5, %3\n v_max_i32 %4, 5, %4\n v_max_i32 %5, 5, %5\n v_max_i32 %6, 5, %6\n v_max_i32 %7, 5, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k37(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_alignbit_b32 %0, %8, %0, 6\n v_alignbit_b32 %1, %8, %1, 6\n v_alignbit_b32 %2, %8, %2, 6\n v_alignbit_b32 %3, %8, %3, 6\n v_alignbit_b32 %4, %9, %4, 6\n v_alignbit_b32 %5, %9, %5, 6\n v_alignbit_b32 %6, %9, %6, 6\n v_alignbit_b32 %7, %9, %7, 6\n" "v_alignbit_b32 %0, %8, %0, 6\n v_alignbit_b32 %1, %8, %1, 6\n v_alignbit_b32 %2, %8, %2, 6\n v_alignbit_b32 %3, %8, %3, 6\n v_alignbit_b32 %4, %9, %4, 6\n v_alignbit_b32 %5, %9, %5, 6\n v_alignbit_b32 %6, %9, %6, 6\n v_alignbit_b32 %7, %9, %7, 6\n" "v_alignbit_b32 %0, %8, %0, 6\n v_alignbit_b32 %1, %8, %1, 6\n v_alignbit_b32 %2, %8, %2, 6\n v_alignbit_b32 %3, %8, %3, 6\n v_alignbit_b32 %4, %9, %4, 6\n v_alignbit_b32 %5, %9, %5, 6\n v_alignbit_b32 %6, %9, %6, 6\n v_alignbit_b32 %7, %9, %7, 6\n" "v_alignbit_b32 %0, %8, %0, 6\n v_alignbit_b32 %1, %8, %1, 6\n v_alignbit_b32 %2, %8, %2, 6\n v_alignbit_b32 %3, %8, %3, 6\n v_alignbit_b32 %4, %9, %4, 6\n v_alignbit_b32 %5, %9, %5, 6\n v_alignbit_b32 %6, %9, %6, 6\n v_alignbit_b32 %7, %9, %7, 6\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k38(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add3_u32 %0, %0, %8, 3\n v_add3_u32 %1, %1, %8, 3\n v_add3_u32 %2, %2, %8, 3\n v_add3_u32 %3, %3, %8, 3\n v_add3_u32 %4, %4, %9, 3\n v_add3_u32 %5, %5, %9, 3\n v_add3_u32 %6, %6, %9, 3\n v_add3_u32 %7, %7, %9, 3\n" "v_add3_u32 %0, %0, %8, 3\n v_add3_u32 %1, %1, %8, 3\n v_add3_u32 %2, %2, %8, 3\n v_add3_u32 %3, %3, %8, 3\n v_add3_u32 %4, %4, %9, 3\n v_add3_u32 %5, %5, %9, 3\n v_add3_u32 %6, %6, %9, 3\n v_add3_u32 %7, %7, %9, 3\n" "v_add3_u32 %0, %0, %8, 3\n v_add3_u32 %1, %1, %8, 3\n v_add3_u32 %2, %2, %8, 3\n v_add3_u32 %3, %3, %8, 3\n v_add3_u32 %4, %4, %9, 3\n v_add3_u32 %5, %5, %9, 3\n v_add3_u32 %6, %6, %9, 3\n v_add3_u32 %7, %7, %9, 3\n" "v_add3_u32 %0, %0, %8, 3\n v_add3_u32 %1, %1, %8, 3\n v_add3_u32 %2, %2, %8, 3\n v_add3_u32 %3, %3, %8, 3\n v_add3_u32 %4, %4, %9, 3\n v_add3_u32 %5, %5, %9, 3\n v_add3_u32 %6, %6, %9, 3\n v_add3_u32 %7, %7, %9, 3\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k39(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %9, %8\n v_and_or_b32 %5, %5, %9, %8\n v_and_or_b32 %6, %6, %9, %8\n v_and_or_b32 %7, %7, %9, %8\n" "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %9, %8\n v_and_or_b32 %5, %5, %9, %8\n v_and_or_b32 %6, %6, %9, %8\n v_and_or_b32 %7, %7, %9, %8\n" "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %9, %8\n v_and_or_b32 %5, %5, %9, %8\n v_and_or_b32 %6, %6, %9, %8\n v_and_or_b32 %7, %7, %9, %8\n" "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %9, %8\n v_and_or_b32 %5, %5, %9, %8\n v_and_or_b32 %6, %6, %9, %8\n v_and_or_b32 %7, %7, %9, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k40(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_max3_i32 %0, %0, %8, %10\n v_max3_i32 %1, %1, %8, %10\n v_max3_i32 %2, %2, %8, %10\n v_max3_i32 %3, %3, %8, %10\n v_max3_i32 %4, %4, %9, %10\n v_max3_i32 %5, %5, %9, %10\n v_max3_i32 %6, %6, %9, %10\n v_max3_i32 %7, %7, %9, %10\n" "v_max3_i32 %0, %0, %8, %10\n v_max3_i32 %1, %1, %8, %10\n v_max3_i32 %2, %2, %8, %10\n v_max3_i32 %3, %3, %8, %10\n v_max3_i32 %4, %4, %9, %10\n v_max3_i32 %5, %5, %9, %10\n v_max3_i32 %6, %6, %9, %10\n v_max3_i32 %7, %7, %9, %10\n" "v_max3_i32 %0, %0, %8, %10\n v_max3_i32 %1, %1, %8, %10\n v_max3_i32 %2, %2, %8, %10\n v_max3_i32 %3, %3, %8, %10\n v_max3_i32 %4, %4, %9, %10\n v_max3_i32 %5, %5, %9, %10\n v_max3_i32 %6, %6, %9, %10\n v_max3_i32 %7, %7, %9, %10\n" "v_max3_i32 %0, %0, %8, %10\n v_max3_i32 %1, %1, %8, %10\n v_max3_i32 %2, %2, %8, %10\n v_max3_i32 %3, %3, %8, %10\n v_max3_i32 %4, %4, %9, %10\n v_max3_i32 %5, %5, %9, %10\n v_max3_i32 %6, %6, %9, %10\n v_max3_i32 %7, %7, %9, %10\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k41(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
__global__ __launch_bounds__(64) void k42(int iters, int *out, int c0, int c1) {
  int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int b0 = c0 + threadIdx.x, b1 = c1 - threadIdx.x;
  for (int i = 0; i < iters; i++) {
    asm volatile("v_add_u32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %9, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %9, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %9, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %9, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" "v_add_u32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %9, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %9, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %9, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %9, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" "v_add_u32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %9, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %9, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %9, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %9, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" "v_add_u32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %9, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %9, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %9, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %9, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1), "s"(c0), "s"(c1) : "vcc");
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = a0;
}
typedef void (*kfn)(int, int *, int, int);
int main() { int *out; (void)hipMalloc(&out, 4); hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); const int iters = 20000; const int W = 4;
const char* names[] = {"v_and_b32 imm","v_and_b32 vv","v_xor_b32 vv","v_sub_u32 vv","v_add_u32 imm","v_add_u32 literal","v_lshlrev_b32 imm","v_lshrrev_b32 imm","v_ashrrev_i32 imm","v_mov_b32","v_cndmask_b32 vcc","v_min_i32","v_max_u32","v_max_f32","v_min_f32","v_add_f32","v_max3_f32","v_med3_i32","v_lshl_add_u32","v_lshl_or_b32","v_or3_b32","v_xad_u32","v_perm_b32","v_mad_i32_i24","v_mul_u32_u24","v_bfe_u32","v_sad_u32","v_max_i16","v_add_u16","v_pk_max_u16","v_pk_lshrrev_b16","v_cmp_gt_i32 (to vcc)","v_addc_co_u32","v_subrev_u32 vv","v_add_u32 (a=a+a)","v_or_b32 vv","v_max_i32 imm","v_alignbit_b32 6","v_add3_u32 + imm","v_and_or_b32 vvv","v_max3_i32 w/ sgpr","v_mov_b32_dpp quad_perm","v_add_u32_sdwa no-sext word"};
kfn fs[] = {k0,k1,k2,k3,k4,k5,k6,k7,k8,k9,k10,k11,k12,k13,k14,k15,k16,k17,k18,k19,k20,k21,k22,k23,k24,k25,k26,k27,k28,k29,k30,k31,k32,k33,k34,k35,k36,k37,k38,k39,k40,k41,k42};
for (int k = 0; k < 43; k++) { const int blocks = 256 * 4 * W; hipLaunchKernelGGL(fs[k], dim3(blocks), dim3(64), 0, 0, 100, out, 1, 2); (void)hipDeviceSynchronize();
 (void)hipEventRecord(e0); hipLaunchKernelGGL(fs[k], dim3(blocks), dim3(64), 0, 0, iters, out, 1, 2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
 double ns = ms * 1e6 / ((double)iters * 32 * W); printf("%-32s %.3f ns/inst/SIMD\n", names[k], ns); } return 0; }