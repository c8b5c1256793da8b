// csa_kernels.hip -- adler32 of file fragments that are already resident in HBM.
//
// The reference folds adler32 over every <= 2 MiB piece its reader thread pulls from a file
// (csa_io.h:250, csa_adler32.cpp:63-129).  Here the chunk is in device memory anyway, for the
// encoder, so the sums are a reduction kernel over it: one workgroup per <= 16 KiB piece computes
//     A = sum(byte_i)            B = sum((len - i) * byte_i)        (plain integers, no modulo)
// and the host folds the pieces of one fragment in order:
//     b' = b + len * a + B,  a' = a + A     (mod 65521)
// which is adler32's definition applied piecewise.  HBM-bound: 1 byte read per input byte,
// 16-byte aligned coalesced loads (a piece that starts unaligned reads its aligned window and
// masks the bytes outside).
#include <hip/hip_runtime.h>
#include <stdint.h>

struct AdlerPiece {
    const uint8_t *ptr;
    uint32_t len;           // <= kAdlerPiece
    uint32_t pad;
};
struct AdlerSums {
    uint64_t a;
    uint64_t b;
};

static constexpr uint32_t kAdlerThreads = 256;

__device__ __forceinline__ void adler_word(uint32_t w, int64_t i0, int64_t len, uint64_t &a, uint64_t &b)
{
    // the four bytes of w sit at piece positions i0 .. i0+3 (little endian); all inside [0, len)
    uint32_t s = __builtin_amdgcn_sad_u8(w, 0u, 0u);                       // b0+b1+b2+b3
    uint32_t t = __builtin_amdgcn_udot4(w, 0x03020100u, 0u, false);        // 0*b0+1*b1+2*b2+3*b3
    a += s;
    b += (uint64_t)(len - i0) * s - t;
}

__global__ __launch_bounds__(kAdlerThreads) void k_adler_pieces(const AdlerPiece *pieces, AdlerSums *out)
{
    const AdlerPiece pc = pieces[blockIdx.x];
    const uint32_t head = (uint32_t)((uintptr_t)pc.ptr & 15u);
    const uint4 *base = (const uint4 *)(pc.ptr - head);
    const int64_t len = pc.len;
    const uint32_t nvec = (head + pc.len + 15u) >> 4;
    uint64_t a = 0, b = 0;
    for (uint32_t v = threadIdx.x; v < nvec; v += kAdlerThreads) {
        uint4 q = base[v];
        int64_t i0 = (int64_t)v * 16 - head;
        if (i0 >= 0 && i0 + 16 <= len) {
            adler_word(q.x, i0, len, a, b);
            adler_word(q.y, i0 + 4, len, a, b);
            adler_word(q.z, i0 + 8, len, a, b);
            adler_word(q.w, i0 + 12, len, a, b);
        } else {
            uint32_t w[4] = {q.x, q.y, q.z, q.w};
            for (int j = 0; j < 16; j++) {
                int64_t i = i0 + j;
                uint32_t byte = (w[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
                if (i >= 0 && i < len) { a += byte; b += (uint64_t)(len - i) * byte; }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    __shared__ uint64_t red[2 * (kAdlerThreads / 64)];
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[2 * wave] = a; red[2 * wave + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t ta = 0, tb = 0;
        for (uint32_t w = 0; w < kAdlerThreads / 64; w++) { ta += red[2 * w]; tb += red[2 * w + 1]; }
        out[blockIdx.x].a = ta;
        out[blockIdx.x].b = tb;
    }
}

void launch_adler_pieces(const void *pieces, void *out, uint32_t npieces, hipStream_t st)
{
    if (!npieces) return;
    hipLaunchKernelGGL(k_adler_pieces, dim3(npieces), dim3(kAdlerThreads), 0, st, (const AdlerPiece *)pieces, (AdlerSums *)out);
}
