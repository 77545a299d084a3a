// What can ANY kernel reach on the traffic of the level-0 320 -> 320 projection with a residual (attention.py:247,258,267: attn.to_out /
// ff.net.2 + hidden_states)?  The layer moves three [M][320] 16-bit tensors -- X in, residual in, Y out (M = 884 736 at B = 32: 1.70 GB) --
// for 2 M 320^2 flops (0.18 PFLOP: 72 us of a 2.5 PFLOP/s matrix pipe), so its roofline is HBM: bgemm_t256p_kernel runs it at 4.9 TB/s of
// those three tensors (the VERDICT of round 4 asks for a two-workgroup 128 x 320 tile kernel "or the micro-benchmark that shows
// it cannot beat 4.9 TB/s").  This is the micro-benchmark: the same three streams with NO matrix work at all -- y = x + r on 16-byte
// pieces -- in the launch shapes a GEMM could take: (a) a plain grid-stride stream (what LayerNorm-like kernels do), (b) persistent
// workgroups that own 256-row x 320-column tiles like bgemm_t256p (one per CU), (c) 128-row tiles, two workgroups per CU (the asked-for
// shape), each tile read whole before it is written (a GEMM cannot store a row before its K loop is done).
//
//   hipcc -O3 --offload-arch=gfx950 tools/micro/stream3.hip -o tools/micro/stream3 && tools/micro/stream3 [rows]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ u4 add8(const u4 a, const u4 b) {
    const b8 x = __builtin_bit_cast(b8, a), y = __builtin_bit_cast(b8, b);
    b8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)((float)x[e] + (float)y[e]);
    return __builtin_bit_cast(u4, o);
}

// (a) grid-stride over 16-byte pieces
__global__ __launch_bounds__(256) void stream_flat(const u4* __restrict__ x, const u4* __restrict__ r, u4* __restrict__ y, size_t pieces) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < pieces; i += (size_t)gridDim.x * 256) y[i] = add8(x[i], r[i]);
}

// (b) / (c): persistent workgroups walking ROWS-row tiles of 320 columns (40 pieces per row); a tile's X and residual are read whole
// (into registers: ROWS * 40 / THREADS pieces per thread and tensor) before its first store, as behind a K loop
template <int ROWS, int THREADS>
__global__ __launch_bounds__(THREADS) void stream_tiles(const u4* __restrict__ x, const u4* __restrict__ r, u4* __restrict__ y, int tiles) {
    constexpr int PPT = ROWS * 40 / THREADS;                  // pieces per thread and tile
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t base = (size_t)t * ROWS * 40;
        u4 a[PPT], b[PPT];
#pragma unroll
        for (int i = 0; i < PPT; ++i) { a[i] = x[base + threadIdx.x + i * THREADS]; b[i] = r[base + threadIdx.x + i * THREADS]; }
#pragma unroll
        for (int i = 0; i < PPT; ++i) y[base + threadIdx.x + i * THREADS] = add8(a[i], b[i]);
    }
}

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 884736;
    const size_t pieces = rows * 40, bytes = pieces * 16;
    u4 *x, *r, *y;
    hipMalloc(&x, bytes); hipMalloc(&r, bytes); hipMalloc(&y, bytes);
    hipMemset(x, 0x3c, bytes); hipMemset(r, 0x3d, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        const int reps = 20;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        printf("%-58s %8.3f ms  %6.2f TB/s of the three tensors\n", name, ms, 3.0 * bytes / ms / 1e9);
    };
    printf("rows %zu x 320 x 2 bytes: %.2f GB per tensor pass of three\n", rows, 3.0 * bytes / 1e9);
    for (int g : {2048, 8192, 32768})
        time(("(a) flat grid-stride, " + std::to_string(g) + " workgroups of 256").c_str(), [&] { stream_flat<<<g, 256>>>(x, r, y, pieces); });
    const int t256 = (int)(rows / 256), t128 = (int)(rows / 128);
    time("(b) 256 x 320 tiles, 512 threads, 256 persistent workgroups", [&] { stream_tiles<256, 512><<<256, 512>>>(x, r, y, t256); });
    time("(b') 256 x 320 tiles, 512 threads, one workgroup per tile", [&] { stream_tiles<256, 512><<<t256, 512>>>(x, r, y, t256); });
    time("(c) 128 x 320 tiles, 256 threads, 512 persistent workgroups", [&] { stream_tiles<128, 256><<<512, 256>>>(x, r, y, t128); });
    time("(c') 128 x 320 tiles, 256 threads, 1024 persistent workgroups", [&] { stream_tiles<128, 256><<<1024, 256>>>(x, r, y, t128); });
    time("(c'') 128 x 320 tiles, 256 threads, one workgroup per tile", [&] { stream_tiles<128, 256><<<t128, 256>>>(x, r, y, t128); });
    time("(d) 64 x 320 tiles, 256 threads, 2048 persistent workgroups", [&] { stream_tiles<64, 256><<<2048, 256>>>(x, r, y, (int)(rows / 64)); });
    return 0;
}
