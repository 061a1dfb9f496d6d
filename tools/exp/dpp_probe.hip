// dpp_probe.hip - what the DPP controls an 8-lane (two-quad) layout would use really do on gfx950, lane by lane:
//   row_ror:8                      lane i of a 16-lane row reads lane (i + 8) % 16  -> quads 0 <-> 2, 1 <-> 3 swap in ONE instruction
//   row_ror:8 with bank_mask 0xC   only quads 2, 3 of every row are written (the others keep the old value)
//   row_half_mirror                lane i of an 8-lane half row reads lane 7 - i
//   quad_perm [1,2,0,1]            the quad layout's rot1
// build: hipcc --offload-arch=gfx950 -O2 -o dpp_probe tools/exp/dpp_probe.hip ; run on the GPU box, prints the lane maps
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK>
__device__ int dpp(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xf, BANK, false); }
__global__ void probe(int *out) {
    const int l = threadIdx.x;
    out[0 * 64 + l] = dpp<0x128, 0xf>(-1, l);   // row_ror:8
    out[1 * 64 + l] = dpp<0x128, 0xc>(-1, l);   // row_ror:8, banks 2,3 only
    out[2 * 64 + l] = dpp<0x141, 0xf>(-1, l);   // row_half_mirror
    out[3 * 64 + l] = dpp<0x49, 0xf>(-1, l);    // quad_perm [1,2,0,1]
    out[4 * 64 + l] = dpp<0x124, 0xf>(-1, l);   // row_ror:4
    float x = (float)l, y;
    asm volatile("v_mov_b32 %0, 0\n\ts_nop 1\n\tv_fmac_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(y) : "v"(x));
    out[5 * 64 + l] = (int)y;                    // x[(l+8)%16 + row] * x[l]
}
int main() {
    int *d, h[6 * 64];
    hipMalloc(&d, sizeof h);
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"row_ror:8", "row_ror:8 bank 0xC", "row_half_mirror", "quad_perm[1,2,0,1]", "row_ror:4", "fmac_dpp ror8 (x[src]*x[l])"};
    for (int k = 0; k < 6; k++) {
        printf("%-28s", names[k]);
        for (int l = 0; l < 32; l++) printf(" %3d", h[k * 64 + l]);
        printf("\n");
    }
    return 0;
}
