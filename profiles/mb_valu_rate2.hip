#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k0(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k1(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n v_min_i32 %4, %4, %8\n v_min_i32 %5, %5, %8\n v_min_i32 %6, %6, %8\n v_min_i32 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k2(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k3(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k4(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k5(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k6(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_cmp_lt_u32_e64 s[12:13], %0, %8\n v_cmp_lt_u32_e64 s[12:13], %1, %8\n v_cmp_lt_u32_e64 s[12:13], %2, %8\n v_cmp_lt_u32_e64 s[12:13], %3, %8\n v_cmp_lt_u32_e64 s[12:13], %4, %8\n v_cmp_lt_u32_e64 s[12:13], %5, %8\n v_cmp_lt_u32_e64 s[12:13], %6, %8\n v_cmp_lt_u32_e64 s[12:13], %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k7(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_pk_add_i16 %0, %0, %8\n v_pk_add_i16 %1, %1, %8\n v_pk_add_i16 %2, %2, %8\n v_pk_add_i16 %3, %3, %8\n v_pk_add_i16 %4, %4, %8\n v_pk_add_i16 %5, %5, %8\n v_pk_add_i16 %6, %6, %8\n v_pk_add_i16 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k8(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k9(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_pk_mad_i16 %0, %0, %8, %8\n v_pk_mad_i16 %1, %1, %8, %8\n v_pk_mad_i16 %2, %2, %8, %8\n v_pk_mad_i16 %3, %3, %8, %8\n v_pk_mad_i16 %4, %4, %8, %8\n v_pk_mad_i16 %5, %5, %8, %8\n v_pk_mad_i16 %6, %6, %8, %8\n v_pk_mad_i16 %7, %7, %8, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k10(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_pk_ashrrev_i16 %0, 15, %0\n v_pk_ashrrev_i16 %1, 15, %1\n v_pk_ashrrev_i16 %2, 15, %2\n v_pk_ashrrev_i16 %3, 15, %3\n v_pk_ashrrev_i16 %4, 15, %4\n v_pk_ashrrev_i16 %5, 15, %5\n v_pk_ashrrev_i16 %6, 15, %6\n v_pk_ashrrev_i16 %7, 15, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k11(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_perm_b32 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_perm_b32 %3, %3, %8, %8\n v_perm_b32 %4, %4, %8, %8\n v_perm_b32 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_perm_b32 %7, %7, %8, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k12(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_dot2_u32_u16 %0, %0, %8, %0\n v_dot2_u32_u16 %1, %1, %8, %1\n v_dot2_u32_u16 %2, %2, %8, %2\n v_dot2_u32_u16 %3, %3, %8, %3\n v_dot2_u32_u16 %4, %4, %8, %4\n v_dot2_u32_u16 %5, %5, %8, %5\n v_dot2_u32_u16 %6, %6, %8, %6\n v_dot2_u32_u16 %7, %7, %8, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k13(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_sad_u8 %0, %0, %8, %0\n v_sad_u8 %1, %1, %8, %1\n v_sad_u8 %2, %2, %8, %2\n v_sad_u8 %3, %3, %8, %3\n v_sad_u8 %4, %4, %8, %4\n v_sad_u8 %5, %5, %8, %5\n v_sad_u8 %6, %6, %8, %6\n v_sad_u8 %7, %7, %8, %7" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k14(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k15(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_add_u16 %0, %0, %8\n v_add_u16 %1, %1, %8\n v_add_u16 %2, %2, %8\n v_add_u16 %3, %3, %8\n v_add_u16 %4, %4, %8\n v_add_u16 %5, %5, %8\n v_add_u16 %6, %6, %8\n v_add_u16 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k16(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_max_i16 %0, %0, %8\n v_max_i16 %1, %1, %8\n v_max_i16 %2, %2, %8\n v_max_i16 %3, %3, %8\n v_max_i16 %4, %4, %8\n v_max_i16 %5, %5, %8\n v_max_i16 %6, %6, %8\n v_max_i16 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k17(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k18(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k19(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k20(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_sub_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_sub_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k21(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
__global__ void __launch_bounds__(256) k22(uint32_t *out, int iters, uint32_t seed) {
 uint32_t a0=threadIdx.x+seed,a1=a0*3,a2=a0*5,a3=a0*7,a4=a0*11,a5=a0*13,a6=a0*17,a7=a0*19; uint32_t s=seed|1;
 for (int i=0;i<iters;++i) asm volatile("v_bfi_b32 %0, %8, %0, %8\n v_bfi_b32 %1, %8, %1, %8\n v_bfi_b32 %2, %8, %2, %8\n v_bfi_b32 %3, %8, %3, %8\n v_bfi_b32 %4, %8, %4, %8\n v_bfi_b32 %5, %8, %5, %8\n v_bfi_b32 %6, %8, %6, %8\n v_bfi_b32 %7, %8, %7, %8" : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(s) : "vcc","s10","s11","s12","s13");
 out[blockIdx.x*256+threadIdx.x]=a0^a1^a2^a3^a4^a5^a6^a7; }
typedef void (*K)(uint32_t*,int,uint32_t);
int main(){ uint32_t *d; hipMalloc(&d, 256*8*256*4); hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
 struct {const char*n; K f;} T[] = {{"v_add_u32", k0},{"v_min_i32", k1},{"v_lshrrev_b32", k2},{"v_cndmask_e32", k3},{"v_cndmask_e64", k4},{"v_cmp_lt_e32", k5},{"v_cmp_lt_e64", k6},{"v_pk_add_i16", k7},{"v_pk_max_i16", k8},{"v_pk_mad_i16", k9},{"v_pk_ashrrev_i16", k10},{"v_perm_b32", k11},{"v_dot2_u32_u16", k12},{"v_sad_u8", k13},{"v_mul_u32_u24", k14},{"v_add_u16", k15},{"v_max_i16", k16},{"v_mov_dpp", k17},{"v_add_dpp", k18},{"v_xor_b32", k19},{"v_sub_u32_sdwa_b", k20},{"v_lshl_add_u32", k21},{"v_bfi_b32", k22},};
 for (int w : {4, 8}) for (auto &t : T) { int blocks=256*w, iters=20000; hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, d, 100, 1u); hipDeviceSynchronize(); hipEventRecord(e0); hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, d, iters, 1u); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); printf("%-20s waves/SIMD %d: %.2f cyc/instr (2.4GHz)\n", t.n, w, ms*1e-3*2.4e9/((double)iters*8*w)); }
 return 0; }
