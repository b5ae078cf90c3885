#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
__global__ void k(const float* in, float* out) {
  float nx = in[0], ny = in[1], xy = in[2];
  float sx = __fsqrt_rn(nx), sy = __fsqrt_rn(ny);
  float den = sx * sy;
  float v = __fdiv_rn(xy, den);
  out[0]=sx; out[1]=sy; out[2]=den; out[3]=v; out[4]=1.0f - v; out[5] = xy/den; out[6]=sqrtf(nx);
}
int main(){
  float h[3] = {2.3456789f, 1.9876543f, 0.4675309f};
  float *d,*o; hipMalloc(&d,12); hipMalloc(&o,28); hipMemcpy(d,h,12,hipMemcpyHostToDevice);
  k<<<1,1>>>(d,o); float r[7]; hipMemcpy(r,o,28,hipMemcpyDeviceToHost);
  float sx=sqrtf(h[0]), sy=sqrtf(h[1]); float den=sx*sy; float v=h[2]/den;
  float c[7]={sx,sy,den,v,1.0f-v,v,sx};
  for(int i=0;i<7;i++){unsigned a,b; memcpy(&a,&r[i],4); memcpy(&b,&c[i],4); printf("%d gpu %08x cpu %08x %s\n",i,a,b,a==b?"":"DIFF");}
}
