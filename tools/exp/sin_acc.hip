// accuracy probe of the hardware sin/cos (v_sin_f32 / v_cos_f32 via __sinf/__cosf) against double precision
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n) { int i = blockIdx.x*blockDim.x+threadIdx.x; if (i<n) { s[i]=__sinf(x[i]); c[i]=__cosf(x[i]); } }
int main() {
    const int n = 1<<22; std::vector<float> x(n), s(n), c(n);
    for (int i=0;i<n;i++) x[i] = -6.4f + 12.8f*i/n;
    float *dx,*ds,*dc; hipMalloc(&dx,n*4); hipMalloc(&ds,n*4); hipMalloc(&dc,n*4);
    hipMemcpy(dx,x.data(),n*4,hipMemcpyHostToDevice);
    k<<<n/256,256>>>(dx,ds,dc,n); hipMemcpy(s.data(),ds,n*4,hipMemcpyDeviceToHost); hipMemcpy(c.data(),dc,n*4,hipMemcpyDeviceToHost);
    double es=0, ec=0, en=0; for (int i=0;i<n;i++){ es=fmax(es,fabs(s[i]-sin((double)x[i]))); ec=fmax(ec,fabs(c[i]-cos((double)x[i]))); en=fmax(en,fabs((double)s[i]*s[i]+(double)c[i]*c[i]-1)); }
    printf("hardware sin/cos on [-6.4,6.4]: max abs err sin %.3e cos %.3e, |s^2+c^2-1| %.3e\n", es, ec, en);
    return 0;
}
