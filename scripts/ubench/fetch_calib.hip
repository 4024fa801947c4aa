// FETCH_SIZE / WRITE_SIZE calibration: kernels that move a KNOWN number of bytes with one load width each, to be run
// under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE in a pass of its own).  The buffer (1 GiB) is far beyond the caches and
// every byte is read exactly once per launch.
//   read_b32 / read_b64 / read_b128     contiguous, 4 / 8 / 16 bytes per lane
//   read_rows<W>                        the grouped convs' slab fill: W-byte pieces (72 = 18 channels, 56 = 14, 40 = 10; 8-byte loads, the
//                                       pieces start at 8-byte-aligned offsets of 5760 / 4480 / 3200-byte rows), two neighbouring pieces per
//                                       workgroup row -- the rest of a row is read by OTHER workgroups, so a row is still read once in total
//   write_b128                          contiguous 16-byte stores
// hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/fetch_calib scripts/ubench/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ __launch_bounds__(256) void read_kernel(const T* __restrict__ p, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = p[i];
        const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(T) / 4); ++k) acc += f[k];
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// rows of ROWB bytes; workgroup (x = row block of 256 rows, y = piece pair) reads 2 * W contiguous bytes of each of its rows with 8-byte loads
template <int W, int ROWB>
__global__ __launch_bounds__(256) void read_rows(const char* __restrict__ p, size_t rows, float* __restrict__ sink) {
    constexpr int PW = 2 * W, L8 = PW / 8;            // 8-byte loads per row piece pair
    float acc = 0.f;
    const size_t r0 = (size_t)blockIdx.x * 256;
    for (int i = threadIdx.x; i < 256 * L8; i += 256) {
        const size_t r = r0 + i / L8;
        if (r < rows) {
            const float2 v = *reinterpret_cast<const float2*>(p + r * ROWB + (size_t)blockIdx.y * PW + (i % L8) * 8);
            acc += v.x + v.y;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void write_b128(float4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = float4{1.f, 2.f, 3.f, 4.f};
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    char* buf;
    float* sink;
    hipMalloc(&buf, bytes + 8192);
    hipMalloc(&sink, 64);
    hipMemset(buf, 0, bytes + 8192);
    for (int rep = 0; rep < 3; ++rep) {
        read_kernel<float><<<4096, 256>>>(reinterpret_cast<const float*>(buf), bytes / 4, sink);
        read_kernel<float2><<<4096, 256>>>(reinterpret_cast<const float2*>(buf), bytes / 8, sink);
        read_kernel<float4><<<4096, 256>>>(reinterpret_cast<const float4*>(buf), bytes / 16, sink);
        {   // 18 channels per group: rows of 5760 bytes = 40 piece pairs of 144 bytes
            const size_t rows = bytes / 5760;
            read_rows<72, 5760><<<dim3((unsigned)((rows + 255) / 256), 40), 256>>>(buf, rows, sink);
        }
        {   // 14 channels: rows of 4480 bytes = 40 piece pairs of 112 bytes
            const size_t rows = bytes / 4480;
            read_rows<56, 4480><<<dim3((unsigned)((rows + 255) / 256), 40), 256>>>(buf, rows, sink);
        }
        {   // 10 channels: rows of 3200 bytes = 40 piece pairs of 80 bytes
            const size_t rows = bytes / 3200;
            read_rows<40, 3200><<<dim3((unsigned)((rows + 255) / 256), 40), 256>>>(buf, rows, sink);
        }
        write_b128<<<4096, 256>>>(reinterpret_cast<float4*>(buf), bytes / 16);
    }
    hipDeviceSynchronize();
    printf("bytes per launch: contiguous kernels %zu; rows<72> %zu, rows<56> %zu, rows<40> %zu\n", bytes, bytes / 5760 * 5760, bytes / 4480 * 4480,
           bytes / 3200 * 3200);
    return 0;
}
