// ds_read_b64_tr_b16 semantics probe: every lane passes its own 8-byte-aligned LDS address; what does each lane receive?
// LDS holds 16-bit words whose value = their own index.  build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_probe.hip -o tools/ubench/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t* lds_bf16x4_ptr;
__global__ void probe(unsigned short* out, int pitch_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short img[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) img[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, a = l & 15, g = (l >> 4) & 1, kb = l >> 5;
    // lane a of a 16-lane group points at the piece (row a / 4, column quad a % 4) of a [4 rows][16 columns] block; rows = pixels, pitch given
    const unsigned short* p = img + (8 * kb + a / 4) * pitch_elems + 16 * g + 4 * (a % 4);
    bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(const_cast<unsigned short*>(p)));
    unsigned short r[4];
    __builtin_memcpy(r, &v, 8);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    const int pitch = 96;
    probe<<<1, 64>>>(d, pitch);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int c = l & 15, g = (l >> 4) & 1, kb = l >> 5;
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int expect = (8 * kb + j) * pitch + 16 * g + c;       // hypothesis: lane c of the group gets column c of rows j = 0..3
            printf(" %5d%s", h[l * 4 + j], h[l * 4 + j] == expect ? "" : "*");
            bad += h[l * 4 + j] != expect;
        }
        printf("\n");
    }
    printf("mismatches against the hypothesis (elem j of lane c = block[row j][col c]): %d\n", bad);
    return 0;
}
