// fs_image.hip -- prepare_image on the device.
//
// Reference: learning/nets.py:155-193.  For every (rotation, scale) of the policy's action space the observation
// (C x S x S float32) is rotated about its centre with scipy.ndimage.rotate (cubic B-spline, reshape=False,
// mode='nearest'), centre-cropped (scale < 1) or replicate-padded (scale > 1) to int(scale * S) and resized to
// dim x dim with nearest-neighbour sampling (cv2.INTER_NEAREST).  The reference runs 96 full-size spline resamplings on
// the CPU per observation; only dim*dim pixels of each survive the final nearest resize, so here
//   1. the cubic-spline COEFFICIENTS of the observation are computed once (scipy's recursive prefilter, float64, on the
//      12-pixel edge-padded image exactly like scipy.ndimage._interpolation._prepad_for_spline_filter), and
//   2. one gather kernel evaluates, for each of the T*C*dim*dim output pixels, the rotated image at the single source
//      pixel the crop / pad / nearest-resize chain selects for it (4x4 coefficient taps, scipy's weight formulas and
//      summation order, float64, result rounded to float32 like scipy's output array).
// Index conventions follow the reference's permute(2,1,0) / swapaxes(-1,0) dance: axis 0 of the rotated plane is the
// tensor's LAST (x) axis.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/flingsim.h"
#include "fs_context.h"

#define FS_SPLINE_PAD 12  // scipy pads by 12 samples for mode='nearest' before the recursive filter

// scipy ni_splines.c: one cubic-spline line in place.  pole z = sqrt(3) - 2, gain (1 - z)(1 - 1/z) = 6, mirror
// initialisation (what scipy applies for mode 'nearest' after padding).
__device__ __forceinline__ void fs_spline_line(double *c, int n, int stride) {
    const double z = -0.26794919243112270647;  // sqrt(3) - 2
    // The recursions are sequential, the memory accesses are not: every pass moves FS_SPLINE_CHUNK samples between
    // memory and registers at a time, so a thread has that many loads in flight instead of one dependent load per step.
    constexpr int CH = 8;
    double c0;
    {   // gain + _init_causal_mirror: c0 = sum_i z^i c[i] (mirror tail weighted by z^(n-1))
        const double z_n_1 = pow(z, (double)(n - 1));
        double z_i = z;
        c0 = 6.0 * c[0] + z_n_1 * (6.0 * c[(size_t)(n - 1) * stride]);
        for (int i0 = 1; i0 < n - 1; i0 += CH) {
            double a[CH], b[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int i = i0 + k;
                a[k] = i < n - 1 ? c[(size_t)i * stride] : 0.0;
                b[k] = i < n - 1 ? c[(size_t)(n - 1 - i) * stride] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (i0 + k < n - 1) {
                    c0 += z_i * (6.0 * a[k] + z_n_1 * (6.0 * b[k]));
                    z_i *= z;
                }
        }
        c0 /= 1.0 - z_n_1 * z_n_1;
    }
    // causal pass (with the gain folded in): c[i] = 6 c[i] + z c[i-1]
    double prev = c0;
    c[0] = c0;
    for (int i0 = 1; i0 < n; i0 += CH) {
        double a[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) a[k] = i0 + k < n ? c[(size_t)(i0 + k) * stride] : 0.0;
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (i0 + k < n) {
                prev = 6.0 * a[k] + z * prev;
                a[k] = prev;
            }
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (i0 + k < n) c[(size_t)(i0 + k) * stride] = a[k];
    }
    // _init_anticausal_mirror, then the anticausal pass c[i] = z (c[i+1] - c[i])
    double nxt = (z * c[(size_t)(n - 2) * stride] + prev) * z / (z * z - 1.0);
    c[(size_t)(n - 1) * stride] = nxt;
    for (int i0 = n - 2; i0 >= 0; i0 -= CH) {
        double a[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) a[k] = i0 - k >= 0 ? c[(size_t)(i0 - k) * stride] : 0.0;
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (i0 - k >= 0) {
                nxt = z * (nxt - a[k]);
                a[k] = nxt;
            }
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (i0 - k >= 0) c[(size_t)(i0 - k) * stride] = a[k];
    }
}

// coef[c][a0][a1], a0 = x (+pad), a1 = y (+pad): edge-padded copy of A[x][y][c] = img[c][y][x]
__global__ void fs_k_spline_pad(const float *img, int C, int S, double *coef) {
    const int P = S + 2 * FS_SPLINE_PAD;
    const size_t total = (size_t)C * P * P;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
        const int a1 = (int)(q % P), a0 = (int)((q / P) % P), c = (int)(q / ((size_t)P * P));
        int x = a0 - FS_SPLINE_PAD, y = a1 - FS_SPLINE_PAD;
        x = x < 0 ? 0 : (x > S - 1 ? S - 1 : x);
        y = y < 0 ? 0 : (y > S - 1 ? S - 1 : y);
        coef[q] = (double)img[((size_t)c * S + y) * S + x];
    }
}

// axis 0 lines (stride P) then axis 1 lines (stride 1): one thread per line
__global__ void fs_k_spline_filter(double *coef, int C, int P, int axis) {
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= C * P) return;
    const int c = line / P, k = line % P;
    double *plane = coef + (size_t)c * P * P;
    if (axis == 0) fs_spline_line(plane + k, P, P);           // fixed a1 = k, walks a0
    else fs_spline_line(plane + (size_t)k * P, P, 1);         // fixed a0 = k, walks a1
}

struct FsImageXform {  // one (rotation, scale) of the action space
    double m00, m01, m10, m11, off0, off1;
    int scaled;   // side of the cropped / padded image that is nearest-resized to dim
    int shift;    // rotated index = clamp(scaled index + shift, 0, S - 1): +start for a crop, -n for a pad, 0 otherwise
};

__device__ __forceinline__ double fs_spline_point(const double *plane, int P, double cc0, double cc1) {
    // scipy NI_GeometricTransform, mode 'nearest', order 3
    cc0 += FS_SPLINE_PAD; cc1 += FS_SPLINE_PAD;
    const double hi = (double)(P - 1);
    cc0 = cc0 < 0.0 ? 0.0 : (cc0 > hi ? hi : cc0);
    cc1 = cc1 < 0.0 ? 0.0 : (cc1 > hi ? hi : cc1);
    const double f0 = floor(cc0), f1 = floor(cc1);
    double w0[4], w1[4];
    {
        const double y = cc0 - f0, z = 1.0 - y;
        w0[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
        w0[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
        w0[0] = z * z * z / 6.0;
        w0[3] = 1.0 - w0[0] - w0[1] - w0[2];
    }
    {
        const double y = cc1 - f1, z = 1.0 - y;
        w1[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
        w1[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
        w1[0] = z * z * z / 6.0;
        w1[3] = 1.0 - w1[0] - w1[1] - w1[2];
    }
    const int s0 = (int)f0 - 1, s1 = (int)f1 - 1;
    double t = 0.0;
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0) {
        int i0 = s0 + k0;
        i0 = i0 < 0 ? 0 : (i0 > P - 1 ? P - 1 : i0);
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            int i1 = s1 + k1;
            i1 = i1 < 0 ? 0 : (i1 > P - 1 ? P - 1 : i1);
            double v = plane[(size_t)i0 * P + i1];
            v *= w0[k0];
            v *= w1[k1];
            t += v;
        }
    }
    return t;
}

// out[t][c][i][j]: i = y, j = x of the network input
__global__ void fs_k_transform_gather(const double *coef, int C, int S, const FsImageXform *xf, int T, int dim, float *out) {
    const int P = S + 2 * FS_SPLINE_PAD;
    const size_t total = (size_t)T * C * dim * dim;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(q % dim), i = (int)((q / dim) % dim);
        const int c = (int)((q / ((size_t)dim * dim)) % C), t = (int)(q / ((size_t)dim * dim * C));
        const FsImageXform X = xf[t];
        // cv2.resize INTER_NEAREST: source index = min(floor(dst * src / dst_size), src - 1)
        const double ratio = (double)X.scaled / (double)dim;
        int u0 = (int)floor((double)j * ratio), u1 = (int)floor((double)i * ratio);
        u0 = u0 > X.scaled - 1 ? X.scaled - 1 : u0;
        u1 = u1 > X.scaled - 1 ? X.scaled - 1 : u1;
        int r0 = u0 + X.shift, r1 = u1 + X.shift;
        r0 = r0 < 0 ? 0 : (r0 > S - 1 ? S - 1 : r0);
        r1 = r1 < 0 ? 0 : (r1 > S - 1 ? S - 1 : r1);
        const double cc0 = X.m00 * (double)r0 + X.m01 * (double)r1 + X.off0;
        const double cc1 = X.m10 * (double)r0 + X.m11 * (double)r1 + X.off1;
        out[q] = (float)fs_spline_point(coef + (size_t)c * P * P, P, cc0, cc1);
    }
}

extern "C" size_t fs_prepare_image_work_bytes(int channels, int size, int n_transforms) {
    if (channels <= 0 || size <= 0 || n_transforms < 0) return 0;
    const size_t P = (size_t)size + 2 * FS_SPLINE_PAD;
    return sizeof(double) * (size_t)channels * P * P + sizeof(FsImageXform) * (size_t)n_transforms + 256;
}

extern "C" int fs_prepare_image(const float *d_img, int channels, int size, int n_transforms, const double *matrix,
                                const double *offset, const double *scale, int dim, float *d_out, void *d_work,
                                void *stream) {
    if (!d_img || !d_out || !d_work || !matrix || !offset || !scale || channels <= 0 || size < 4 || n_transforms <= 0 ||
        dim <= 0) {
        fs_set_error("fs_prepare_image: bad arguments");
        return FS_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int P = size + 2 * FS_SPLINE_PAD;
    double *coef = (double *)d_work;
    const size_t coef_bytes = sizeof(double) * (size_t)channels * P * P;
    FsImageXform *d_xf = (FsImageXform *)((char *)d_work + ((coef_bytes + 255) & ~(size_t)255));
    FsImageXform *h_xf = new FsImageXform[n_transforms];
    for (int t = 0; t < n_transforms; ++t) {
        FsImageXform &X = h_xf[t];
        X.m00 = matrix[4 * t]; X.m01 = matrix[4 * t + 1]; X.m10 = matrix[4 * t + 2]; X.m11 = matrix[4 * t + 3];
        X.off0 = offset[2 * t]; X.off1 = offset[2 * t + 1];
        const int new_dim = (int)(scale[t] * (double)size);  // int(scale * img.shape[0])
        if (scale[t] < 1.0) {         // crop_center: start = S//2 - new_dim//2
            X.scaled = new_dim;
            X.shift = size / 2 - new_dim / 2;
        } else if (scale[t] > 1.0) {  // pad: n = (new_dim - S)//2 on every side
            const int npad = (new_dim - size) / 2;
            X.scaled = size + 2 * npad;
            X.shift = -npad;
        } else {
            X.scaled = size;
            X.shift = 0;
        }
        if (X.scaled <= 0) {
            delete[] h_xf;
            fs_set_error("fs_prepare_image: scale too small");
            return FS_ERR_ARG;
        }
    }
    hipError_t err = hipMemcpyAsync(d_xf, h_xf, sizeof(FsImageXform) * n_transforms, hipMemcpyHostToDevice, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);  // h_xf is pageable and freed below
    delete[] h_xf;
    if (!fs_hip_ok(err, "fs_prepare_image upload")) return FS_ERR_HIP;
    hipLaunchKernelGGL(fs_k_spline_pad, dim3(512), dim3(256), 0, st, d_img, channels, size, coef);
    const int lines = channels * P;
    hipLaunchKernelGGL(fs_k_spline_filter, dim3((lines + 63) / 64), dim3(64), 0, st, coef, channels, P, 0);
    hipLaunchKernelGGL(fs_k_spline_filter, dim3((lines + 63) / 64), dim3(64), 0, st, coef, channels, P, 1);
    hipLaunchKernelGGL(fs_k_transform_gather, dim3(2048), dim3(256), 0, st, coef, channels, size, d_xf, n_transforms, dim, d_out);
    return fs_hip_ok(hipGetLastError(), "fs_prepare_image launch") ? FS_OK : FS_ERR_HIP;
}
