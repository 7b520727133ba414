// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters for the access widths the
// particle kernels use (MI355X_MICROARCH.md "HBM": only 16-B-per-lane streams are calibrated).
// Streams a buffer far larger than the 256 MiB Infinity Cache with 8-B-per-lane (double) and
// 16-B-per-lane (double2) coalesced loads, and stores 8 B per lane; the known byte counts are
// printed so that counter values can be divided by them.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.cpp
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- ./fetch_calib
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read8(const double *__restrict__ p, size_t n, double *out)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += stride) s += p[i];
    if (s == 12345.678) out[0] = s;
}

__global__ void read16(const double2 *__restrict__ p, size_t n, double *out)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += stride) { double2 v = p[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}

__global__ void write8(double *__restrict__ p, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 1.0;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;   // 2 GiB
    double *p, *out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(p, 0, bytes);
    (void)hipDeviceSynchronize();
    const size_t n8 = bytes / 8, n16 = bytes / 16;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, p, n8, out);
        hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const double2 *)p, n16, out);
        hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, p, n8);
    }
    (void)hipDeviceSynchronize();
    printf("bytes_per_launch %zu\n", bytes);
    return 0;
}
