// semantic check of the swizzled packed-f32 forms used by conv_wino4.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const f32x2* in, f32x2* out) {
    f32x2 ba = in[0], ec = in[1];
    const f32x2 k2 = {2.f, -2.f};
    f32x2 t12, t34;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(t12) : "v"(ba));
    asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(t34) : "v"(ec), "v"(k2));
    out[0] = t12; out[1] = t34;
}
int main() {
    f32x2 h[2] = {{3.f, 10.f}, {5.f, 100.f}}, *d, *o, r[2];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    printf("ba=(b=3,a=10): t12 = (%g, %g) want (13, 7)\n", r[0][0], r[0][1]);
    printf("ec=(e=5,c=100): t34 = (%g, %g) want (110, 90)\n", r[1][0], r[1][1]);
    return 0;
}
