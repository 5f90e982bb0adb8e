// Micro-test that pinned the semantics of ds_read_b64_tr_b16 on MI355X (see ds_read_tr16_b64 in csrc/ds_device.h):
//   hipcc --offload-arch=gfx950 -O2 -o tr_test tools/tr16_semantics.hip && ./tr_test
// prints, for row strides of 32 / 48 / 64 bytes, the four 16-bit values each lane receives when lane i of a
// 16-lane group supplies the 8-byte piece (row i>>2, quad i&3) of a 4 x 16 block: lane i gets column i.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short *in, unsigned short *out, int stride_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x;
    // 16-lane group g = l>>4 ; lane i = l&15 supplies row (i>>2), column quad (i&3)
    const int i = l & 15, g = l >> 4;
    const char *base = (const char *)lds + g * 1024 + (i >> 2) * stride_bytes + (i & 3) * 8;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)base);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short h[4096], o[256], *di, *dout;
    for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMalloc(&di, 8192); hipMalloc(&dout, 512);
    hipMemcpy(di, h, 8192, hipMemcpyHostToDevice);
    for (int stride : {32, 48, 64}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, di, dout, stride);
        hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
        printf("stride %d B (elements: row r col c = r*%d + c)\n", stride, stride / 2);
        for (int l = 0; l < 64; l += 1) { if (l < 20 || l >= 60) printf(" lane %2d: %4d %4d %4d %4d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]); }
    }
    return 0;
}
