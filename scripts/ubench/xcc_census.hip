#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o){ unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); if(threadIdx.x==0) o[blockIdx.x]=x; }
int main(){ unsigned* d; hipMalloc(&d, 4*64); k<<<64,64>>>(d); unsigned h[64]; hipMemcpy(h,d,256,hipMemcpyDeviceToHost); for(int i=0;i<64;i++) printf("%u ", h[i]&0xf); printf("\n"); }
