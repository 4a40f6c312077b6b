// fs_valuenet.hip -- SpatialValueNet forward as three hand-written gfx950 kernels.
//
// Reference: learning/nets.py:81-141 -- normalise, Conv3x3(C->16)+BN+LeakyReLU, 8 x Residual[Conv3x3(16->16)+BN+ReLU,
// Conv3x3(16->16)+BN, +identity, ReLU], Conv3x3(16->1); stride 1, zero padding 1, eval-mode BatchNorm.  The caller folds
// every BatchNorm into the preceding convolution (w' = w g / sqrt(var + eps), b' = beta - mean g / sqrt(var + eps)) and
// hands the folded weights to fs_value_net_pack().
//
// The network is 16 channels wide: as GEMMs its convolutions are [pixels x 144] x [144 x 16], too thin for a library
// GEMM tiling (MIOpen's implicit-GEMM kernel pads N to 32 and is followed by separate bias / add / ReLU launches).
// Here one residual block is ONE kernel:
//   * a workgroup (8 wavefronts) owns an 8-row x 64-column strip of one image; the strip's input (12 rows, all 16
//     channels, channel-planar) is loaded once into LDS, conv1's output (10 rows) is written to LDS, conv2 reads it from
//     there and takes the identity from the input tile -- per block the activations cross HBM / L2 once in, once out;
//   * both convolutions run on the matrix cores in exact fp32 (v_mfma_f32_16x16x4_f32: M = 16 pixels of a row, N = the
//     16 output channels -- no padding --, K = 4 input channels of one filter tap; 36 instructions per 16-pixel tile).
//     The B operands (the folded weights, 36 registers per convolution) come from LDS when a convolution starts; the A
//     operand is one ds_read_b32 per MFMA with a compile-time offset.  Channel planes are 16 (mod 32) dwords apart, so
//     the four 16-lane channel groups of a wavefront read disjoint banks;
//   * bias, ReLU, the residual add and the zero padding are applied in the accumulator registers;
//   * a wavefront works on 5 (conv1) / 4 (conv2) row tiles at once: independent accumulators cover the 40-cycle
//     dependent-MFMA latency and share the B operand, and the A operands are read one filter tap ahead;
//   * the kernel is persistent (one workgroup per CU -- the 101 KB tile allows one): the B operands are staged in LDS
//     once, and the next strip's rows are requested from global memory before the current strip's convolutions start;
//   * strip ids are mapped so that the 8 strips of an image run on ONE XCD (the dispatcher places workgroup n on XCD
//     n % 8): neighbouring strips' halo rows are served by that XCD's L2.
// Measured (one MI355X, scripts/cnn_timing.py, profiles/r01_cnn_*): 0.36 ms per observation of 96 x 64 x 64 against 1.49 ms
// for the folded network on MIOpen; 2.32 ms for 8 observations = 101 TFLOP/s of useful fp32 arithmetic (64 % of the MFMA
// fp32 peak; 12.5 % more is spent recomputing conv1 on the halo rows instead of a round trip through HBM).
// The first (C -> 16, LeakyReLU) and last (16 -> 1) layers are 9 % and 6 % of a 16 -> 16 layer's arithmetic and run as
// plain VALU kernels with the weights in scalar registers.
//
// Numerics: fp32 throughout, MFMA accumulation is an ordered fmaf chain; results differ from MIOpen's by summation order
// only (tests/test_valuenet_gpu.py: <= 2e-5 absolute on O(1) outputs against the PyTorch fp32 module and against the
// reference's own outputs in tests/golden/nets_golden.npz).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "../../include/flingsim.h"
#include "fs_context.h"

typedef float vn_f32x4 __attribute__((ext_vector_type(4)));

#define VN_W 64              // image width the kernels are built for (the reference's obs_dim)
#define VN_HT_ROWS 8         // output rows per workgroup of the first / last layer kernels
#ifndef VN_ROWS
#define VN_ROWS 8            // output rows per strip of the residual-block kernel (8: one workgroup per CU; 4: two)
#endif
#define VN_THREADS 512
#if VN_ROWS == 8
#define VN_RS 72             // LDS row stride in floats: image column x lives at x + 4 (16-byte aligned), halos at 3 and 68
#define VN_CS_IN 880         // LDS channel stride of the 12-row input tile  (12 * 72 = 864, rounded up to 16 mod 32)
#define VN_CS_MID 720        // LDS channel stride of the 10-row conv1 output (10 * 72 = 720 = 16 mod 32)
#define VN_NT1 5             // row tiles per wavefront: conv1 (10 rows over two wavefront rows)
#define VN_NT2 4             //                          conv2 (8 rows)
#define VN_PERSISTENT_WGS 256  // one workgroup per CU (its LDS tile allows one), multiple of 8
#define VN_WAVES_PER_EU 2
#elif VN_ROWS == 4
#define VN_RS 68             // right halo (column 68) of a row aliases the unused column 0 of the next row
#define VN_CS_IN 560         // 8 rows  * 68 = 544 -> 16 mod 32
#define VN_CS_MID 432        // 6 rows  * 68 = 408 -> 16 mod 32
#define VN_NT1 3
#define VN_NT2 2
#define VN_PERSISTENT_WGS 512  // two workgroups per CU: 2 x 81920 B of LDS = all 160 KiB
#define VN_WAVES_PER_EU 4
#else
#error "VN_ROWS must be 8 or 4"
#endif
#define VN_IN_ROWS (VN_ROWS + 4)
#define VN_MID_ROWS (VN_ROWS + 2)
#define VN_STRIPS (VN_W / VN_ROWS)
#define VN_BLOCK_LDS_BYTES ((16 * (VN_CS_IN + VN_CS_MID) + 2 * 36 * 64) * 4)

// packed parameter block (floats)
#define VN_OFF_MEAN 0
#define VN_OFF_STD 4
#define VN_OFF_HEADW 8                  // [4 ic][9 taps][16 oc]
#define VN_OFF_HEADB (VN_OFF_HEADW + 576)
#define VN_OFF_CONV (VN_OFF_HEADB + 16)  // 16 x { [36 k-steps][64 lanes] B operands, [16] bias }
#define VN_CONV_STRIDE (36 * 64 + 16)
#define VN_OFF_TAILW (VN_OFF_CONV + 16 * VN_CONV_STRIDE)  // [16 ic][9 taps]
#define VN_PARAM_FLOATS (VN_OFF_TAILW + 144)

// workgroup id -> (image, strip) with all strips of an image on one XCD
template <int STRIPS = 8>
__device__ __forceinline__ bool vn_tile_of(int n, int batch, int &image, int &strip) {
    const int xcd = n & 7, k = n >> 3;
    image = (k / STRIPS) * 8 + xcd;
    strip = k % STRIPS;
    return image < batch;
}

// ---- first layer: normalise + Conv3x3(C -> 16) + bias + LeakyReLU(0.01) --------------------------------------------
template <int C>
__global__ __launch_bounds__(VN_THREADS) void fs_k_vn_head(const float *__restrict__ P, const float *__restrict__ obs,
                                                           int obs_channels, int c_off, int batch,
                                                           float *__restrict__ out) {
    __shared__ float tile[C][10][72];
    int b, strip;
    if (!vn_tile_of(blockIdx.x, batch, b, strip)) return;
    const int t = threadIdx.x, y0 = strip * VN_HT_ROWS;
    for (int idx = t; idx < C * 160; idx += VN_THREADS) {
        const int c = idx / 160, rem = idx % 160, r = rem >> 4, q = rem & 15, y = y0 - 1 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)y < (unsigned)VN_W) {
            v = *(const float4 *)(obs + (((size_t)b * obs_channels + c_off + c) * VN_W + y) * VN_W + 4 * q);
            const float m = P[VN_OFF_MEAN + c], s = P[VN_OFF_STD + c];
            v.x = (v.x - m) / s; v.y = (v.y - m) / s; v.z = (v.z - m) / s; v.w = (v.w - m) / s;
        }
        *(float4 *)&tile[c][r][4 + 4 * q] = v;
    }
    if (t < C * 20) {
        const int c = t / 20, rem = t % 20;
        tile[c][rem >> 1][(rem & 1) ? 68 : 3] = 0.f;
    }
    __syncthreads();
    const int x = t & 63, r = t >> 6;
    float acc[16];
#pragma unroll
    for (int oc = 0; oc < 16; ++oc) acc[oc] = P[VN_OFF_HEADB + oc];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float v = tile[c][r + tap / 3][x + 3 + tap % 3];
#pragma unroll
            for (int oc = 0; oc < 16; ++oc) acc[oc] = __builtin_fmaf(v, P[VN_OFF_HEADW + (c * 9 + tap) * 16 + oc], acc[oc]);
        }
    float *dst = out + (((size_t)b * 16) * VN_W + y0 + r) * VN_W + x;
#pragma unroll
    for (int oc = 0; oc < 16; ++oc) {
        const float v = acc[oc];
        dst[(size_t)oc * VN_W * VN_W] = v > 0.f ? v : v * 0.01f;
    }
}

// ---- one residual block --------------------------------------------------------------------------------------------
// 36 k-steps (9 taps x 4 channel groups) over NT row tiles two rows apart; `a` is the lane's LDS base address.
template <int CS, int NT>
__device__ __forceinline__ void vn_mma(const float *a, const float *w, vn_f32x4 (&acc)[NT]) {
    // A operands one filter tap (4 channel groups x NT tiles) ahead of the MFMAs that consume them; the B operands of a
    // tap (w = this lane's column of the staged weights, 64 floats per k-step) are read from LDS with that tap
    float cur[4][NT], nxt[4][NT];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
        for (int j = 0; j < NT; ++j) cur[cg][j] = a[cg * 4 * CS + j * 2 * VN_RS];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float wt[4];
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) wt[cg] = w[(tap * 4 + cg) * 64];
        if (tap < 8) {
            const int off = ((tap + 1) / 3) * VN_RS + ((tap + 1) % 3);
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int j = 0; j < NT; ++j) nxt[cg][j] = a[off + cg * 4 * CS + j * 2 * VN_RS];
        }
#pragma unroll
        for (int cg = 0; cg < 4; ++cg)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[cg][j], wt[cg], acc[j], 0, 0, 0);
#pragma unroll
        for (int cg = 0; cg < 4; ++cg)
#pragma unroll
            for (int j = 0; j < NT; ++j) cur[cg][j] = nxt[cg][j];
    }
}

// global -> register half of the input-tile load: image rows y0-2 .. y0+9 of all 16 channels, zero outside the image
#define VN_FETCH (16 * VN_IN_ROWS * 16 / VN_THREADS)  // float4 per thread of one input tile
__device__ __forceinline__ void vn_fetch_tile(const float *__restrict__ in, int b, int y0, int t, float4 (&r)[VN_FETCH]) {
    const float *src = in + (size_t)b * 16 * VN_W * VN_W;
#pragma unroll
    for (int k = 0; k < VN_FETCH; ++k) {
        const int idx = t + VN_THREADS * k, ch = idx / (VN_IN_ROWS * 16), rem = idx % (VN_IN_ROWS * 16), row = rem >> 4, q = rem & 15;
        const int y = y0 - 2 + row;
        r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)y < (unsigned)VN_W) r[k] = *(const float4 *)(src + ((size_t)ch * VN_W + y) * VN_W + 4 * q);
    }
}

// Persistent: workgroup n handles tiles n, n + gridDim.x, ... (gridDim.x is a multiple of 8, so a tile keeps its XCD).
// The next tile's input is requested from global memory before the current tile's convolutions start and is written to
// LDS after they end; the B operands are loaded once per workgroup.
__global__ __launch_bounds__(VN_THREADS, VN_WAVES_PER_EU) void fs_k_vn_block(const float *__restrict__ P, const float *__restrict__ in,
                                                            int batch, int n_tiles, float *__restrict__ out) {
    extern __shared__ float vn_lds[];
    float *s_in = vn_lds, *s_mid = vn_lds + 16 * VN_CS_IN;
    const int t = threadIdx.x, l = t & 63, wv = t >> 6;
    const int oc = l & 15, kg = l >> 4;
    int tile = blockIdx.x, b, strip;
    if (tile >= n_tiles) return;
    bool live = vn_tile_of<VN_STRIPS>(tile, batch, b, strip);
    float4 pre[VN_FETCH];
    if (live) vn_fetch_tile(in, b, strip * VN_ROWS, t, pre);

    // halo columns are written once; the tile loads never touch them
    if (t < 16 * 2 * VN_IN_ROWS) {
        const int ch = t / (2 * VN_IN_ROWS), rem = t % (2 * VN_IN_ROWS);
        s_in[ch * VN_CS_IN + (rem >> 1) * VN_RS + ((rem & 1) ? 68 : 3)] = 0.f;
    }
    if (t < 16 * 2 * VN_MID_ROWS) {
        const int ch = t / (2 * VN_MID_ROWS), rem = t % (2 * VN_MID_ROWS);
        s_mid[ch * VN_CS_MID + (rem >> 1) * VN_RS + ((rem & 1) ? 68 : 3)] = 0.f;
    }
    // B operands of both convolutions, staged once per workgroup: k-step s = tap * 4 + cg of lane l holds
    // W[oc][4 cg + kg][tap]; each convolution pulls its 36 registers from here when it starts
    float *s_w = s_mid + 16 * VN_CS_MID;
    for (int k = t; k < 36 * 64; k += VN_THREADS) {
        s_w[k] = P[k];
        s_w[36 * 64 + k] = P[VN_CONV_STRIDE + k];
    }
    const float b1 = P[36 * 64 + oc], b2 = P[VN_CONV_STRIDE + 36 * 64 + oc];
    const int xt = wv & 3, rpar = wv >> 2;
    const int a_lane = rpar * VN_RS + xt * 16 + (l & 15) + 3;   // A[m = l & 15][k = l >> 4]
    const int c_lane = xt * 16 + 4 * kg + 4;                    // D[m = 4 (l >> 4) + i][n = l & 15]

    for (;;) {
        const int y0 = strip * VN_ROWS, bcur = b;
        if (live) {
#pragma unroll
            for (int k = 0; k < VN_FETCH; ++k) {
                const int idx = t + VN_THREADS * k, ch = idx / (VN_IN_ROWS * 16), rem = idx % (VN_IN_ROWS * 16), row = rem >> 4, q = rem & 15;
                *(float4 *)(s_in + ch * VN_CS_IN + row * VN_RS + 4 + 4 * q) = pre[k];
            }
        }
        __syncthreads();
        const bool cur_live = live;
        tile += gridDim.x;
        const bool more = tile < n_tiles;
        if (more) {
            live = vn_tile_of<VN_STRIPS>(tile, batch, b, strip);
            if (live) vn_fetch_tile(in, b, strip * VN_ROWS, t, pre);
        }
        if (cur_live) {
            {   // conv1 + bias + ReLU -> s_mid rows rpar, rpar+2, ..., rpar+8  (image rows y0-1+m)
                vn_f32x4 acc[VN_NT1];
#pragma unroll
                for (int j = 0; j < VN_NT1; ++j) acc[j] = (vn_f32x4){b1, b1, b1, b1};
                vn_mma<VN_CS_IN, VN_NT1>(s_in + kg * VN_CS_IN + a_lane, s_w + l, acc);
#pragma unroll
                for (int j = 0; j < VN_NT1; ++j) {
                    const int m = rpar + 2 * j, ym = y0 - 1 + m;
                    vn_f32x4 v = acc[j];
                    const bool inside = (unsigned)ym < (unsigned)VN_W;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (inside && v[i] > 0.f) ? v[i] : 0.f;
                    *(vn_f32x4 *)(s_mid + oc * VN_CS_MID + m * VN_RS + c_lane) = v;
                }
            }
            __syncthreads();
            {   // conv2 + bias + identity + ReLU -> global rows y0 + rpar, +2, +4, +6
                vn_f32x4 acc[VN_NT2];
#pragma unroll
                for (int j = 0; j < VN_NT2; ++j) acc[j] = (vn_f32x4){b2, b2, b2, b2};
                vn_mma<VN_CS_MID, VN_NT2>(s_mid + kg * VN_CS_MID + a_lane, s_w + 36 * 64 + l, acc);
                float *dst = out + ((size_t)bcur * 16 + oc) * VN_W * VN_W;
#pragma unroll
                for (int j = 0; j < VN_NT2; ++j) {
                    const int o = rpar + 2 * j;
                    const vn_f32x4 id = *(const vn_f32x4 *)(s_in + oc * VN_CS_IN + (o + 2) * VN_RS + c_lane);
                    vn_f32x4 v = acc[j] + id;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
                    *(vn_f32x4 *)(dst + (size_t)(y0 + o) * VN_W + xt * 16 + 4 * kg) = v;
                }
            }
        }
        if (!more) break;
        __syncthreads();  // every wave is done with s_in / s_mid before the next tile overwrites them
    }
}

// ---- last layer: Conv3x3(16 -> 1), no bias, no activation -----------------------------------------------------------
__global__ __launch_bounds__(VN_THREADS) void fs_k_vn_tail(const float *__restrict__ P, const float *__restrict__ in,
                                                           int batch, float *__restrict__ out) {
    __shared__ float tile[16][10][72];
    int b, strip;
    if (!vn_tile_of(blockIdx.x, batch, b, strip)) return;
    const int t = threadIdx.x, y0 = strip * VN_HT_ROWS;
    const float *src = in + (size_t)b * 16 * VN_W * VN_W;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int idx = t + VN_THREADS * k, ch = idx / 160, rem = idx % 160, r = rem >> 4, q = rem & 15, y = y0 - 1 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)y < (unsigned)VN_W) v = *(const float4 *)(src + ((size_t)ch * VN_W + y) * VN_W + 4 * q);
        *(float4 *)&tile[ch][r][4 + 4 * q] = v;
    }
    if (t < 16 * 20) {
        const int ch = t / 20, rem = t % 20;
        tile[ch][rem >> 1][(rem & 1) ? 68 : 3] = 0.f;
    }
    __syncthreads();
    const int x = t & 63, r = t >> 6;
    float acc = 0.f;
#pragma unroll
    for (int ic = 0; ic < 16; ++ic)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            acc = __builtin_fmaf(tile[ic][r + tap / 3][x + 3 + tap % 3], P[VN_OFF_TAILW + ic * 9 + tap], acc);
    out[((size_t)b * VN_W + y0 + r) * VN_W + x] = acc;
}

// ---- C-ABI -----------------------------------------------------------------------------------------------------------
extern "C" {

size_t fs_value_net_param_floats(void) { return VN_PARAM_FLOATS; }

size_t fs_value_net_work_bytes(int batch, int size) {
    if (batch <= 0 || size != VN_W) return 0;
    return (size_t)2 * batch * 16 * size * size * sizeof(float);
}

int fs_value_net_pack(int in_channels, const float *mean, const float *std, const float *w_first, const float *b_first,
                      const float *w_blocks, const float *b_blocks, const float *w_last, float *packed) {
    if ((in_channels != 1 && in_channels != 3 && in_channels != 4) || !mean || !std || !w_first || !b_first ||
        !w_blocks || !b_blocks || !w_last || !packed) {
        fs_set_error("fs_value_net_pack: bad arguments");
        return FS_ERR_ARG;
    }
    memset(packed, 0, sizeof(float) * VN_PARAM_FLOATS);
    for (int c = 0; c < 4; ++c) {
        packed[VN_OFF_MEAN + c] = c < in_channels ? mean[c] : 0.f;
        packed[VN_OFF_STD + c] = c < in_channels ? std[c] : 1.f;
    }
    for (int oc = 0; oc < 16; ++oc) {
        packed[VN_OFF_HEADB + oc] = b_first[oc];
        for (int ic = 0; ic < in_channels; ++ic)
            for (int tap = 0; tap < 9; ++tap)
                packed[VN_OFF_HEADW + (ic * 9 + tap) * 16 + oc] = w_first[(oc * in_channels + ic) * 9 + tap];
    }
    for (int conv = 0; conv < 16; ++conv) {
        float *dst = packed + VN_OFF_CONV + conv * VN_CONV_STRIDE;
        const float *w = w_blocks + (size_t)conv * 16 * 16 * 9;
        for (int tap = 0; tap < 9; ++tap)
            for (int cg = 0; cg < 4; ++cg)
                for (int l = 0; l < 64; ++l)
                    dst[(tap * 4 + cg) * 64 + l] = w[((l & 15) * 16 + 4 * cg + (l >> 4)) * 9 + tap];
        for (int oc = 0; oc < 16; ++oc) dst[36 * 64 + oc] = b_blocks[conv * 16 + oc];
    }
    for (int k = 0; k < 144; ++k) packed[VN_OFF_TAILW + k] = w_last[k];
    return FS_OK;
}

int fs_value_net_forward(const float *d_params, const float *d_obs, int obs_channels, int channel_offset,
                         int in_channels, int batch, int size, float *d_out, void *d_work, void *stream) {
    if (!d_params || !d_obs || !d_out || !d_work || batch <= 0 || size != VN_W ||
        (in_channels != 1 && in_channels != 3 && in_channels != 4) || channel_offset < 0 ||
        channel_offset + in_channels > obs_channels) {
        fs_set_error("fs_value_net_forward: bad arguments (the kernels are built for 64 x 64 observations)");
        return FS_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set[64] = {};  // per device: the attribute belongs to the device's copy of the kernel
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void *)fs_k_vn_block, hipFuncAttributeMaxDynamicSharedMemorySize,
                                VN_BLOCK_LDS_BYTES) != hipSuccess) {
            fs_set_error("fs_value_net_forward: cannot reserve LDS for fs_k_vn_block");
            return FS_ERR_HIP;
        }
        attr_set[dev] = true;
    }
    float *act_a = (float *)d_work, *act_b = act_a + (size_t)batch * 16 * size * size;
    const int grid = ((batch + 7) / 8) * 64;                 // first / last layer: 8 strips of 8 rows per image
    const int tiles = ((batch + 7) / 8) * 8 * VN_STRIPS;     // residual blocks: VN_STRIPS strips per image
    if (in_channels == 1)
        hipLaunchKernelGGL(fs_k_vn_head<1>, dim3(grid), dim3(VN_THREADS), 0, st, d_params, d_obs, obs_channels,
                           channel_offset, batch, act_a);
    else if (in_channels == 3)
        hipLaunchKernelGGL(fs_k_vn_head<3>, dim3(grid), dim3(VN_THREADS), 0, st, d_params, d_obs, obs_channels,
                           channel_offset, batch, act_a);
    else
        hipLaunchKernelGGL(fs_k_vn_head<4>, dim3(grid), dim3(VN_THREADS), 0, st, d_params, d_obs, obs_channels,
                           channel_offset, batch, act_a);
    for (int blk = 0; blk < 8; ++blk) {
        hipLaunchKernelGGL(fs_k_vn_block, dim3(tiles < VN_PERSISTENT_WGS ? tiles : VN_PERSISTENT_WGS), dim3(VN_THREADS),
                           VN_BLOCK_LDS_BYTES, st, d_params + VN_OFF_CONV + 2 * blk * VN_CONV_STRIDE, act_a, batch, tiles,
                           act_b);
        float *tmp = act_a; act_a = act_b; act_b = tmp;
    }
    hipLaunchKernelGGL(fs_k_vn_tail, dim3(grid), dim3(VN_THREADS), 0, st, d_params, act_a, batch, d_out);
    if (hipGetLastError() != hipSuccess) {
        fs_set_error("fs_value_net_forward: kernel launch failed");
        return FS_ERR_HIP;
    }
    return FS_OK;
}

}  // extern "C"
