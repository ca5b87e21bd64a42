// Probe: v_cndmask_b32 with a DPP wave shift on gfx950 -- lane l takes lane l - 1's value, lanes 0 and 32 (VCC) take the second operand.
// The building block of round 3's "chunk rows moved down one row per step in registers" (a new row ENTERS at lanes 0 and 32 each step:
// measured slower than the LDS reads).  Round 5 walks the distances so that nothing enters (bench_micro/peritem_step.hip): that form is the product's.
// hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o dpp_shift_probe dpp_shift_probe.hip
#include <hip/hip_runtime.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(int* out, const int* in) {
  int a = in[threadIdx.x], nr = in[threadIdx.x + 64], d;
  asm volatile("s_mov_b32 vcc_lo, 1\n\ts_mov_b32 vcc_hi, 1\n\ts_nop 1\n\tv_cndmask_b32_dpp %0, %1, %2, vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1"
               : "=&v"(d) : "v"(a), "v"(nr) : "vcc");
  out[threadIdx.x] = d;
}
int main() {
  int h[128], o[64]; for (int i = 0; i < 128; i++) h[i] = i < 64 ? 100 + i : 1000 + i;
  int *di, *dout; hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o); hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dout, di); hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i++) printf("%d ", o[i]); printf("\n");
  int ok = 1; for (int i = 0; i < 64; i++) { int want = (i == 0 || i == 32) ? 1000 + 64 + i : 100 + i - 1; if (o[i] != want) ok = 0; }
  printf(ok ? "OK\n" : "MISMATCH\n");
  return 0;
}
