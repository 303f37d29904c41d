// ssim.hip -- the image term of the training loss on the other side of the rasterizer: the base model's
//   main_loss = (1 - lambda) * mean |gt - rgb| + lambda * (1 - SSIM(gt, rgb))
// that rade_gs_model.py:289 inherits (`super().get_loss_dict`: nerfstudio Splatfacto, third-party and absent from the
// reference tree; SSIM = pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3)) [UNVERIFIED-UPSTREAM]:
// per channel, the five 11 x 11 gaussian (sigma 1.5) window means E[x], E[y], E[xx], E[yy], E[xy] over the VALID region
// ((H - 10) x (W - 10) window positions, no padding), ssim = (2 mx my + C1) / (mx^2 + my^2 + C1) * (2 sxy + C2) /
// (sxx + syy + C2) with C1 = 0.01^2, C2 = 0.03^2, averaged over positions and channels.
//
// Forward: one launch over 16 x 16 tiles of window positions -- the 26 x 26 x 3 input patch of both images goes
// through LDS once, the separable filter runs from LDS (rows, then columns), and besides the tile's SSIM sum the kernel
// leaves the three per-position derivative maps the backward needs (d ssim / d E[x], d E[xx], d E[xy]); a second, tiny
// launch adds the tile sums in a fixed order in fp64 and forms the loss value.  Backward: one launch over 16 x 16 tiles
// of PIXELS: the transposed filter of the three maps (the same separable passes, from LDS) combined with the pixel's own
// x and y, plus the L1 term's sign gradient -- the whole d main_loss / d rgb in one pass, nothing accumulated atomically
// (reproducible bit for bit).  HBM traffic at 1080p: forward 2 x 25 MB in + 74 MB of maps out, backward the reverse
// + 25 MB of gradient: ~0.25 GB per step (~60 us at the copy roof) -- but the kernels are bound by vector issue, not by
// memory: 71 / 80 us at 1080p, 22.5 M / 24.2 M wave instructions of 4 cycles each on 1 024 SIMDs (37 / 39 us at full
// issue; SQ counters of this round), and with every global access removed the forward still takes 66 us.  What cut
// instructions: four outputs per thread and pass from 14 values read once, the window folded around its centre
// (5 adds + 6 multiply-adds per plane and output instead of 11 multiply-adds), two reciprocals instead of six IEEE
// divisions, row / column of the loads computed once.  What did not matter (each measured): one workgroup per channel
// or per tile, all loads in flight at once, an XCD-contiguous tile order, leaving out the stores (-20 / -7 us).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

#include "misplat.h"

namespace {

constexpr int kTile = 16;
constexpr int kWin = 11;
constexpr int kPatch = kTile + kWin - 1;        // 26

struct Window { float w[kWin]; };

// pytorch_msssim._fspecial_gauss_1d(11, 1.5): float32 exp of -(c^2) / (2 sigma^2), normalised by the float32 sum
Window make_window() {
    Window W;
    float sum = 0.f;
    for (int i = 0; i < kWin; i++) {
        const float c = (float)(i - kWin / 2);
        W.w[i] = expf(-(c * c) / (2.0f * 1.5f * 1.5f));
        sum += W.w[i];
    }
    for (int i = 0; i < kWin; i++) W.w[i] /= sum;
    return W;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// maps: [3][3 channels][OH][OW] (d ssim / d E[x], / d E[xx], / d E[xy]); partials: one float per workgroup.
// One workgroup = one 16 x 16 tile of window positions, ALL THREE channels: the 26 x 26 x 3 patch of each image is read
// once, as 26 contiguous runs of 312 bytes (the images are channel-interleaved).  The kernel is bound by LDS bandwidth,
// not by memory (one output per thread and pass: 77 LDS reads per output, 84 us at 1080p), so both passes are register
// tiled: a thread filters FOUR adjacent outputs from 14 values it reads once (row pass: 312 (channel, row, column-quad)
// items; column pass: 192 (channel, row-quad, column) items) -- 2.5 x fewer LDS operations, three barriers per workgroup.
constexpr int kRow3 = 3 * kPatch;               // 78 floats of a patch row
constexpr int kRowS = kRow3 + 1;                // LDS row stride (odd: the stride-3 channel walk stays conflict-free)
constexpr int kHzS = 20;                        // row stride of the row-pass results: row quads 16 banks apart
constexpr int kQuads = kTile / 4;
__global__ __launch_bounds__(256) void ssim_fwd_kernel(int H, int Wd, const float* __restrict__ rgb, const float* __restrict__ gt,
                                                       Window win, float C1, float C2, float* __restrict__ maps,
                                                       float* __restrict__ partials) {
    __shared__ float sx[kPatch][kRowS], sy[kPatch][kRowS];
    __shared__ float hz[3][5][kPatch][kHzS];
    __shared__ float red[2][4];
    const int ox = blockIdx.x * kTile, oy = blockIdx.y * kTile;
    const int OW = Wd - (kWin - 1), OH = H - (kWin - 1);
    const int row_floats = 3 * Wd;
    {
        // 234 threads = 3 patch rows of 78 floats per round (row / column computed once, 9 rounds); all of a thread's loads
        // are issued before the first LDS store
        constexpr int kRowsPer = 256 / kRow3, kRounds = (kPatch + kRowsPer - 1) / kRowsPer;
        const int r0 = threadIdx.x / kRow3, q = threadIdx.x - r0 * kRow3;
        const int xf = 3 * ox + q;
        const bool lane_ok = r0 < kRowsPer && xf < row_floats;
        float va[kRounds], vb[kRounds];
#pragma unroll
        for (int i = 0; i < kRounds; i++) {
            const int r = r0 + kRowsPer * i, y = oy + r;
            va[i] = 0.f; vb[i] = 0.f;
            if (lane_ok && r < kPatch && y < H) {
                const size_t o = (size_t)y * row_floats + xf;
                va[i] = rgb[o]; vb[i] = gt[o];
            }
        }
#pragma unroll
        for (int i = 0; i < kRounds; i++) {
            const int r = r0 + kRowsPer * i;
            if (r0 < kRowsPer && r < kPatch) { sx[r][q] = va[i]; sy[r][q] = vb[i]; }
        }
    }
    __syncthreads();
    // ---- the L1 term's sum over the tile's own 16 x 16 pixels (the patch's first 16 rows and columns), all channels
    float l1 = 0.f;
    {
        const int lx = threadIdx.x & (kTile - 1), ly = threadIdx.x >> 4;
        if (ox + lx < Wd && oy + ly < H)
            l1 = (fabsf(sx[ly][3 * lx] - sy[ly][3 * lx]) + fabsf(sx[ly][3 * lx + 1] - sy[ly][3 * lx + 1])) +
                 fabsf(sx[ly][3 * lx + 2] - sy[ly][3 * lx + 2]);
    }
    // ---- row pass: item = (channel, patch row, quad of output columns)
    for (int t = threadIdx.x; t < 3 * kPatch * kQuads; t += 256) {
        const int quad = t % kQuads, cr = t / kQuads;
        const int r = cr % kPatch, c = cr / kPatch;
        // the five planes x, y, xx, yy, xy of the 14 values once, then each of the four outputs folds its 11 taps around
        // the centre (the window is symmetric: w[k] = w[10 - k]): 5 adds + 6 multiply-adds per plane instead of 11
        float v[5][kWin + 3];
#pragma unroll
        for (int j = 0; j < kWin + 3; j++) {
            const float a = sx[r][3 * (4 * quad + j) + c], b = sy[r][3 * (4 * quad + j) + c];
            v[0][j] = a; v[1][j] = b; v[2][j] = a * a; v[3][j] = b * b; v[4][j] = a * b;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int col = 4 * quad + u;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                float acc = win.w[5] * v[q][u + 5];
#pragma unroll
                for (int k = 0; k < 5; k++) acc = fmaf(win.w[k], v[q][u + k] + v[q][u + 10 - k], acc);
                hz[c][q][r][col] = acc;
            }
        }
    }
    __syncthreads();
    // ---- column pass: item = (channel, quad of output rows, column); four outputs per thread
    float s = 0.f;
    if (threadIdx.x < 3 * kQuads * kTile) {
        const int col = threadIdx.x % kTile, crq = threadIdx.x / kTile;
        const int rq = crq % kQuads, c = crq / kQuads;
        float acc[4][5];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int q = 0; q < 5; q++) acc[u][q] = 0.f;
#pragma unroll
        for (int j = 0; j < kWin + 3; j++) {
            float v[5];
#pragma unroll
            for (int q = 0; q < 5; q++) v[q] = hz[c][q][4 * rq + j][col];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (j - u >= 0 && j - u < kWin) {
#pragma unroll
                    for (int q = 0; q < 5; q++) acc[u][q] = fmaf(win.w[j - u], v[q], acc[u][q]);
                }
        }
        const size_t plane = (size_t)OH * OW;
        const int px = ox + col;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int py = oy + 4 * rq + u;
            if (px < OW && py < OH) {
                const float m1 = acc[u][0], m2 = acc[u][1], xx = acc[u][2], yy = acc[u][3], xy = acc[u][4];
                const float s11 = xx - m1 * m1, s22 = yy - m2 * m2, s12 = xy - m1 * m2;
                const float A1 = 2.0f * m1 * m2 + C1, B1 = m1 * m1 + m2 * m2 + C1;
                const float A2 = 2.0f * s12 + C2, B2 = s11 + s22 + C2;
                // (two reciprocals instead of six IEEE divisions -- ~10 instructions each; B1, B2 >= C > 0)
                const float r1 = __builtin_amdgcn_rcpf(B1), r2 = __builtin_amdgcn_rcpf(B2);
                const float lum = A1 * r1, cs = A2 * r2;
                s += lum * cs;
                const float dlum_dm = 2.0f * (m2 - m1 * lum) * r1;
                const float dcs_dm = 2.0f * (m1 * cs - m2) * r2;
                const size_t o = (size_t)c * plane + (size_t)py * OW + px;
                maps[o] = cs * dlum_dm + lum * dcs_dm;
                maps[3 * plane + o] = -lum * cs * r2;
                maps[6 * plane + o] = 2.0f * lum * r2;
            }
        }
    }
    s = wave_sum(s);
    l1 = wave_sum(l1);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = l1; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x, nb = (size_t)gridDim.x * gridDim.y;
        partials[threadIdx.x * nb + b] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

// partials: n SSIM tile sums, then n L1 tile sums.  l1_loss (or NULL): the L1 mean from elsewhere (misplat_loss_fwd);
// NULL = the kernel's own sum.
__global__ __launch_bounds__(1024) void ssim_final_kernel(int n, const float* __restrict__ partials, double inv_count,
                                                          double inv_n3, const float* __restrict__ l1_loss, float lambda,
                                                          float* __restrict__ ssim_out, float* __restrict__ main_out) {
    __shared__ double sm[2][1024];
    double s = 0.0, a = 0.0;
    for (int b = threadIdx.x; b < n; b += 1024) { s += (double)partials[b]; a += (double)partials[n + b]; }
    sm[0][threadIdx.x] = s; sm[1][threadIdx.x] = a;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) { sm[0][threadIdx.x] += sm[0][threadIdx.x + w]; sm[1][threadIdx.x] += sm[1][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float ssim = (float)(sm[0][0] * inv_count);
        const float l1 = l1_loss ? *l1_loss : (float)(sm[1][0] * inv_n3);
        if (ssim_out) *ssim_out = ssim;
        if (main_out) *main_out = (1.0f - lambda) * l1 + lambda * (1.0f - ssim);
    }
}

// v_rgb[q, c] = g * ((1 - lambda) * sign(rgb - gt) / (3 H W) - lambda / (3 OH OW) * d(sum ssim) / d rgb[q, c]).
// One workgroup = one 16 x 16 tile of pixels, all three channels, register tiled like the forward (four outputs per
// thread and pass); the tile of both images comes in and the gradient tile goes out as contiguous 192-byte runs through LDS.
__global__ __launch_bounds__(256) void ssim_bwd_kernel(int H, int Wd, const float* __restrict__ rgb, const float* __restrict__ gt,
                                                       const float* __restrict__ maps, Window win, const float* __restrict__ g_main,
                                                       float w_l1, float w_ssim, float* __restrict__ v_rgb) {
    __shared__ float sg[3][3][kPatch][kPatch + 1];             // [channel][map][row][column]
    __shared__ float hz[3][3][kPatch][kHzS];
    __shared__ float xs[kTile][3 * kTile + 1], ys[kTile][3 * kTile + 1], outs[kTile][3 * kTile + 1];
    const int ix0 = blockIdx.x * kTile, iy0 = blockIdx.y * kTile;
    const int OW = Wd - (kWin - 1), OH = H - (kWin - 1);
    const size_t plane = (size_t)OH * OW;
    const int row_floats = 3 * Wd;
    {
        // window position p covers pixels p .. p + 10: pixel q takes from positions q - 10 .. q (patch column j <-> p = ix0 - 10 + j).
        // 234 threads = 9 (channel, row) lines of 26 positions per round, 9 rounds; indices computed once.
        constexpr int kLinesPer = 256 / kPatch, kLines = 3 * kPatch, kRounds = (kLines + kLinesPer - 1) / kLinesPer;
        const int l0 = threadIdx.x / kPatch, q = threadIdx.x - l0 * kPatch;
        const int px = ix0 - (kWin - 1) + q;
        const bool lane_ok = l0 < kLinesPer && px >= 0 && px < OW;
        float pre[kRounds][3];
#pragma unroll
        for (int i = 0; i < kRounds; i++) {
            const int line = l0 + kLinesPer * i;               // = channel * 26 + row
            const int c = line / kPatch, r = line - c * kPatch;
            const int py = iy0 - (kWin - 1) + r;
            pre[i][0] = 0.f; pre[i][1] = 0.f; pre[i][2] = 0.f;
            if (lane_ok && line < kLines && py >= 0 && py < OH) {
                const size_t o = (size_t)c * plane + (size_t)py * OW + px;
                pre[i][0] = maps[o]; pre[i][1] = maps[3 * plane + o]; pre[i][2] = maps[6 * plane + o];
            }
        }
        for (int t = threadIdx.x; t < kTile * 3 * kTile; t += 256) {
            const int r = t / (3 * kTile), qq = t - r * (3 * kTile);
            const int y = iy0 + r, xf = 3 * ix0 + qq;
            float a = 0.f, b = 0.f;
            if (y < H && xf < row_floats) {
                const size_t o = (size_t)y * row_floats + xf;
                a = rgb[o]; b = gt[o];
            }
            xs[r][qq] = a; ys[r][qq] = b;
        }
#pragma unroll
        for (int i = 0; i < kRounds; i++) {
            const int line = l0 + kLinesPer * i;
            const int c = line / kPatch, r = line - c * kPatch;
            if (l0 < kLinesPer && line < kLines) { sg[c][0][r][q] = pre[i][0]; sg[c][1][r][q] = pre[i][1]; sg[c][2][r][q] = pre[i][2]; }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * kPatch * kQuads; t += 256) {
        const int quad = t % kQuads, cr = t / kQuads;
        const int r = cr % kPatch, c = cr / kPatch;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            float a[kWin + 3];
#pragma unroll
            for (int j = 0; j < kWin + 3; j++) a[j] = sg[c][m][r][4 * quad + j];
#pragma unroll
            for (int u = 0; u < 4; u++) {                 // (weight w[10 - k] = w[k]: folded around the centre)
                float acc = win.w[5] * a[u + 5];
#pragma unroll
                for (int k = 0; k < 5; k++) acc = fmaf(win.w[k], a[u + k] + a[u + 10 - k], acc);
                hz[c][m][r][4 * quad + u] = acc;
            }
        }
    }
    __syncthreads();
    const float g = g_main ? *g_main : 0.f;
    if (threadIdx.x < 3 * kQuads * kTile) {
        const int col = threadIdx.x % kTile, crq = threadIdx.x / kTile;
        const int rq = crq % kQuads, c = crq / kQuads;
        float acc[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) { acc[u][0] = 0.f; acc[u][1] = 0.f; acc[u][2] = 0.f; }
#pragma unroll
        for (int j = 0; j < kWin + 3; j++) {
            const float v0 = hz[c][0][4 * rq + j][col], v1 = hz[c][1][4 * rq + j][col], v2 = hz[c][2][4 * rq + j][col];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (j - u >= 0 && j - u < kWin) {
                    const float w = win.w[j - u];
                    acc[u][0] = fmaf(w, v0, acc[u][0]); acc[u][1] = fmaf(w, v1, acc[u][1]); acc[u][2] = fmaf(w, v2, acc[u][2]);
                }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int ly = 4 * rq + u;
            const float xv = xs[ly][3 * col + c], yv = ys[ly][3 * col + c];
            const float diff = xv - yv;
            const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            outs[ly][3 * col + c] = g * (w_l1 * sgn - w_ssim * (acc[u][0] + 2.0f * xv * acc[u][1] + yv * acc[u][2]));
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kTile * 3 * kTile; t += 256) {
        const int r = t / (3 * kTile), q = t - r * (3 * kTile);
        const int y = iy0 + r, xf = 3 * ix0 + q;
        if (y < H && xf < row_floats) v_rgb[(size_t)y * row_floats + xf] = outs[r][q];
    }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH; }
inline bool shape_ok(int32_t h, int32_t w) { return h >= kWin && w >= kWin && (int64_t)h * w <= (1 << 28); }

}  // namespace

extern "C" int64_t misplat_ssim_scratch_floats(int32_t height, int32_t width) {
    if (!shape_ok(height, width)) return -1;
    const int64_t gx = (width + kTile - 1) / kTile, gy = (height + kTile - 1) / kTile;
    return 9 * (int64_t)(height - (kWin - 1)) * (width - (kWin - 1)) + 2 * gx * gy;
}

extern "C" int misplat_ssim_fwd(int32_t height, int32_t width, const float* rgb, const float* gt, float* scratch,
                                const float* l1_loss, float ssim_lambda, float* ssim, float* main_loss,
                                misplat_stream_t stream) {
    if (!shape_ok(height, width) || !rgb || !gt || !scratch) return MISPLAT_EINVAL;
    const int OW = width - (kWin - 1), OH = height - (kWin - 1);
    // tiles of window positions (the last ones of a row / column may lie wholly outside the valid region: they add 0)
    const dim3 grid((width + kTile - 1) / kTile, (height + kTile - 1) / kTile, 1);
    float* maps = scratch;
    float* partials = scratch + 9 * (size_t)OH * OW;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, s, (int)height, (int)width, rgb, gt, make_window(), 0.01f * 0.01f,
                       0.03f * 0.03f, maps, partials);
    hipLaunchKernelGGL(ssim_final_kernel, dim3(1), dim3(1024), 0, s, (int)(grid.x * grid.y), partials,
                       1.0 / (3.0 * (double)OH * (double)OW), 1.0 / (3.0 * (double)height * (double)width), l1_loss,
                       ssim_lambda, ssim, main_loss);
    return check_launch();
}

extern "C" int misplat_ssim_bwd(int32_t height, int32_t width, const float* rgb, const float* gt, const float* scratch,
                                const float* g_main, float ssim_lambda, float* v_rgb, misplat_stream_t stream) {
    if (!shape_ok(height, width) || !rgb || !gt || !scratch || !v_rgb) return MISPLAT_EINVAL;
    const int OW = width - (kWin - 1), OH = height - (kWin - 1);
    const dim3 grid((width + kTile - 1) / kTile, (height + kTile - 1) / kTile, 1);
    const float w_l1 = (float)((1.0 - (double)ssim_lambda) / (3.0 * (double)height * (double)width));
    const float w_ssim = (float)((double)ssim_lambda / (3.0 * (double)OH * (double)OW));
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (int)height, (int)width, rgb, gt, scratch,
                       make_window(), g_main, w_l1, w_ssim, v_rgb);
    return check_launch();
}
