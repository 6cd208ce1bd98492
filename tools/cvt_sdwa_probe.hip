// v_cvt_f32_f16_e32 against v_cvt_f32_f16_sdwa src0_sel:WORD_1 on every f16 bit pattern (gfx950), MODE.FP16_OVFL clear and set
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* a, unsigned* b, int ovfl) {
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned lo = e, hi = e << 16, x, y;
  asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(x) : "v"(lo));
  asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(y) : "v"(hi));
  a[e] = x; b[e] = y;
}
int main() {
  unsigned *da, *db; static unsigned a[65536], b[65536];
  hipMalloc(&da, 65536 * 4); hipMalloc(&db, 65536 * 4);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    k<<<256, 256>>>(da, db, ovfl);
    hipMemcpy(a, da, sizeof(a), hipMemcpyDeviceToHost); hipMemcpy(b, db, sizeof(b), hipMemcpyDeviceToHost);
    int diff = 0;
    for (int i = 0; i < 65536; ++i) if (a[i] != b[i]) { if (diff < 8) printf("  f16 0x%04x: e32 0x%08x sdwa 0x%08x\n", i, a[i], b[i]); ++diff; }
    printf("FP16_OVFL=%d: %d of 65536 f16 patterns differ\n", ovfl, diff);
  }
  return 0;
}
