#include "../diskann_amd/csrc/dann_device.h"
#include <cstdio>
#include <cstring>
#include <cmath>
using namespace dann;
__global__ void k(const float* x, const float* y, int dim, float* out) {
  int v = threadIdx.x % 4;
  float d = group_distance_raw<2, OP_COS, 0>(x, y, dim, v);
  // components
  FAcc<OP_COS> acc; acc.init();
  float s[4]={0,0,0,0}, nx[4]={0,0,0,0}, ny[4]={0,0,0,0};
  int rem = dim & 7;
  auto part=[&](float(&a)[4], int which){
    for (int i=0;i<4;++i){ int l=4*(v&1)+i; float xx=0,yy=0; if(l<rem){xx=x[l];yy=y[l];}
      if(which==0)a[i]=__builtin_fmaf(xx,yy,a[i]); else if(which==1)a[i]=__builtin_fmaf(xx,xx,a[i]); else a[i]=__builtin_fmaf(yy,yy,a[i]);}};
  float S=finish_vec<2>(s,[&](float(&a)[4]){part(a,0);});
  float NX=finish_vec<2>(nx,[&](float(&a)[4]){part(a,1);});
  float NY=finish_vec<2>(ny,[&](float(&a)[4]){part(a,2);});
  if (threadIdx.x==0){ out[0]=d; out[1]=S; out[2]=NX; out[3]=NY; out[4]=cosine_finish(NX,NY,S);
     float den=__fsqrt_rn(NX)*__fsqrt_rn(NY); out[5]=den; out[6]=__fdiv_rn(S,den); }
}
int main(){
  unsigned xb[7], yb[7];
  float hx[8], hy[8];
  FILE* f=fopen("xy.bin","rb"); fread(hx,4,7,f); fread(hy,4,7,f); fclose(f);
  float *dx,*dy,*o; hipMalloc(&dx,32); hipMalloc(&dy,32); hipMalloc(&o,32);
  hipMemcpy(dx,hx,28,hipMemcpyHostToDevice); hipMemcpy(dy,hy,28,hipMemcpyHostToDevice);
  k<<<1,64>>>(dx,dy,7,o); float r[7]; hipMemcpy(r,o,28,hipMemcpyDeviceToHost);
  for(int i=0;i<7;i++){unsigned a; memcpy(&a,&r[i],4); printf("%d %08x %.9g\n",i,a,r[i]);}
}
