#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
__global__ void k(const float* a, const float* b, float* s1, float* s2, float* d1, float* d2, int n) {
  int i = blockIdx.x*blockDim.x+threadIdx.x; if (i>=n) return;
  s1[i] = __fsqrt_rn(a[i]); s2[i] = __builtin_sqrtf(a[i]); d1[i] = __fdiv_rn(a[i], b[i]); d2[i] = a[i]/b[i];
}
int main(){
  int n=1<<22; std::vector<float> a(n),b(n); std::mt19937 g(1); std::uniform_real_distribution<float> u(1e-3f, 100.f);
  for(int i=0;i<n;i++){a[i]=u(g); b[i]=u(g);} 
  float *da,*db,*o[4]; hipMalloc(&da,n*4); hipMalloc(&db,n*4); for(auto&p:o) hipMalloc(&p,n*4);
  hipMemcpy(da,a.data(),n*4,hipMemcpyHostToDevice); hipMemcpy(db,b.data(),n*4,hipMemcpyHostToDevice);
  k<<<n/256,256>>>(da,db,o[0],o[1],o[2],o[3],n);
  std::vector<float> r(n); const char* names[4]={"__fsqrt_rn","builtin_sqrtf","__fdiv_rn","operator/"};
  for(int j=0;j<4;j++){ hipMemcpy(r.data(),o[j],n*4,hipMemcpyDeviceToHost); long bad=0; for(int i=0;i<n;i++){ float w = j<2? sqrtf(a[i]) : a[i]/b[i]; if(memcmp(&w,&r[i],4)) bad++; } printf("%s mismatches %ld / %d\n",names[j],bad,n);} 
}
