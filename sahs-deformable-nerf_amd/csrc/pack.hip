// pack.hip -- weight packing and per-frame conditioning for the field kernels.
//
//  * sahs_pack_weights: canonical flat fp32 state_dict buffer -> kernel-friendly stream
//    (MFMA A-fragment order, zero padded; feature grid re-laid channel-last).  Runs once per
//    parameter update (eval: once per checkpoint).
//  * sahs_fold_conditioning: per frame.  Restates AudioNet (modules.py:43-73), the pose
//    encoding (models.py:482-504 + encode_pose_fn :203-207) and folds the per-frame constant
//    inputs of the six layers that see them into those layers' biases.
#include <hip/hip_runtime.h>
#include "sahs_common.hpp"
#include "sahs_layout.hpp"

namespace SAHS_NS {

__device__ const Program dProg = make_program();
__device__ const FlatOffsets dFlat = make_flat_offsets();

// ---- f32 stream: [layer][tile][kblock][lane 64][r 4]; lane = 16*q + i holds W[16t+i][16b+4q+r] ----
__global__ void pack_stream_f32_kernel(const float *__restrict__ flat, float *__restrict__ packed)
{
    const long total = 2 * STREAM_FLOATS;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int level = (int)(e / STREAM_FLOATS);
        const long s = e - (long)level * STREAM_FLOATS;
        int li = 0;
        while (li + 1 < NUM_LAYERS && dProg.layer[li + 1].stream_off <= s) ++li;
        const Layer &L = dProg.layer[li];
        const long w = s - L.stream_off;
        const int per_tile = L.KB * 256;
        const int t = (int)(w / per_tile);
        const int rem = (int)(w - (long)t * per_tile);
        const int b = rem >> 8, lane = (rem & 255) >> 2, r = rem & 3;
        const int i = lane & 15, q = lane >> 4;
        const int row = 16 * t + i - L.row_shift;
        int bb = b, col = -1;
        for (int sg = 0; sg < L.nseg; ++sg) {
            if (bb < L.seg[sg].blocks) {
                const int c = 16 * bb + 4 * q + r;
                if (c < L.seg[sg].valid) col = L.seg[sg].src_col + c;
                break;
            }
            bb -= L.seg[sg].blocks;
        }
        float v = 0.0f;
        if (col >= 0 && row >= 0 && row < L.src_rows) v = flat[L.w_off[level] + (long)row * L.src_ld + col];
        packed[PACK_STREAM_OFF + e] = v;
    }
}

// grid (1,32,D,H,W) channel-first -> [D][H][W][32] so one corner's 32 channels are 128 contiguous bytes
__global__ void pack_grid_f32_kernel(const float *__restrict__ flat, float *__restrict__ packed)
{
    const long vox = (long)G_RES * G_RES * G_RES;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < GRID_FLOATS; e += (long)gridDim.x * blockDim.x) {
        const long v = e / D_GRID; const int c = (int)(e % D_GRID);
        packed[PACK_GRID_OFF + e] = flat[dFlat.grid + (long)c * vox + v];
    }
}

__global__ void pack_table_kernel(float *__restrict__ packed)
{
    uint32_t *tab = reinterpret_cast<uint32_t *>(packed + PACK_TABLE_OFF);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int c = 0;
        for (int li = 0; li < NUM_LAYERS; ++li) {
            const Layer &L = dProg.layer[li];
            const int chunk = L.G * L.KB * 256;
            for (int k = 0; k < L.NT / L.G; ++k) tab[c++] = (uint32_t)(L.stream_off + (long)k * chunk);
        }
        tab[c] = (uint32_t)STREAM_FLOATS;
    }
}

// ---- bf16 stream (field_bf16.hip): one thread per lane-fragment of 8 halfwords ----
__device__ const hb::ProgramH dProgH = hb::make_program_h();

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f)
{
    return __builtin_bit_cast(unsigned short, (__bf16)f);   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
}

__global__ void pack_stream_bf16_kernel(const float *__restrict__ flat, float *__restrict__ packed)
{
    using namespace hb;
    unsigned short *out = reinterpret_cast<unsigned short *>(packed + PACKH_STREAM_OFF);
    const long total = 2 * STREAM_HW / 8;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long hw = e * 8;
        const int level = (int)(hw / STREAM_HW);
        const long sidx = hw - (long)level * STREAM_HW;
        int li = 0;
        while (li + 1 < NUM_LAYERS_H && dProgH.layer[li + 1].stream_off <= sidx) ++li;
        const LayerH &L = dProgH.layer[li];
        const long w = sidx - L.stream_off;
        const int per_tile = L.KB32 * 1024;
        const int t = (int)(w / per_tile);
        const int rem = (int)(w - (long)t * per_tile);
        const int b = rem >> 10, st = (rem >> 9) & 1, lane = (rem & 511) >> 3;
        const int i = lane & 31, h = lane >> 5;
        const int row = 32 * t + i - L.row_shift;
        int bb = b, seg = -1;
        for (int sg = 0; sg < L.nseg; ++sg) {
            if (bb < L.seg[sg].blocks) { seg = sg; break; }
            bb -= L.seg[sg].blocks;
        }
        unsigned short v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 32 * bb + 16 * st + 8 * (j >> 2) + 4 * h + (j & 3);
            float x = 0.0f;
            if (seg >= 0 && c < L.seg[seg].valid && row >= 0 && row < L.src_rows)
                x = flat[L.w_off[level] + (long)row * L.src_ld + L.seg[seg].src_col + c];
            v[j] = f32_to_bf16_rne(x);
        }
        uint4 q;
        q.x = v[0] | ((unsigned)v[1] << 16); q.y = v[2] | ((unsigned)v[3] << 16);
        q.z = v[4] | ((unsigned)v[5] << 16); q.w = v[6] | ((unsigned)v[7] << 16);
        *reinterpret_cast<uint4 *>(out + hw) = q;
    }
}


// ---- bf16x3 stream (field_bf16x3.hip): the bf16 stream with every fragment followed by the fragment of the remainders w - bf16(w) ----
__global__ void pack_stream_bf16x3_kernel(const float *__restrict__ flat, float *__restrict__ packed)
{
    using namespace hb;
    unsigned short *out = reinterpret_cast<unsigned short *>(packed + PACKX_STREAM_OFF);
    const long total = 2 * STREAM_HW / 8;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long hw = e * 8;                                  // position in the plain bf16 stream
        const int level = (int)(hw / STREAM_HW);
        const long sidx = hw - (long)level * STREAM_HW;
        int li = 0;
        while (li + 1 < NUM_LAYERS_H && dProgH.layer[li + 1].stream_off <= sidx) ++li;
        const LayerH &L = dProgH.layer[li];
        const long w = sidx - L.stream_off;
        const int per_tile = L.KB32 * 1024;
        const int t = (int)(w / per_tile);
        const int rem = (int)(w - (long)t * per_tile);
        const int b = rem >> 10, st = (rem >> 9) & 1, lane = (rem & 511) >> 3;
        const int i = lane & 31, h = lane >> 5;
        const int row = 32 * t + i - L.row_shift;
        int bb = b, seg = -1;
        for (int sg = 0; sg < L.nseg; ++sg) {
            if (bb < L.seg[sg].blocks) { seg = sg; break; }
            bb -= L.seg[sg].blocks;
        }
        unsigned short vh[8], vl[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 32 * bb + 16 * st + 8 * (j >> 2) + 4 * h + (j & 3);
            float x = 0.0f;
            if (seg >= 0 && c < L.seg[seg].valid && row >= 0 && row < L.src_rows)
                x = flat[L.w_off[level] + (long)row * L.src_ld + L.seg[seg].src_col + c];
            vh[j] = f32_to_bf16_rne(x);
            vl[j] = f32_to_bf16_rne(x - __builtin_bit_cast(float, (unsigned)vh[j] << 16));
        }
        // fragment f = 2b + st of tile t: hi at 2f, lo at 2f + 1 (512 halfwords each) of the doubled tile
        unsigned short *dst = out + (long)level * 2 * STREAM_HW + 2 * L.stream_off + (long)t * 2 * per_tile + (long)(2 * b + st) * 1024 + lane * 8;
        uint4 q;
        q.x = vh[0] | ((unsigned)vh[1] << 16); q.y = vh[2] | ((unsigned)vh[3] << 16);
        q.z = vh[4] | ((unsigned)vh[5] << 16); q.w = vh[6] | ((unsigned)vh[7] << 16);
        *reinterpret_cast<uint4 *>(dst) = q;
        q.x = vl[0] | ((unsigned)vl[1] << 16); q.y = vl[2] | ((unsigned)vl[3] << 16);
        q.z = vl[4] | ((unsigned)vl[5] << 16); q.w = vl[6] | ((unsigned)vl[7] << 16);
        *reinterpret_cast<uint4 *>(dst + 512) = q;
    }
}

__global__ void pack_table_bf16_kernel(float *__restrict__ packed)
{
    using namespace hb;
    uint32_t *tab = reinterpret_cast<uint32_t *>(packed + PACKH_TABLE_OFF);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int c = 0;
        for (int li = 0; li < NUM_LAYERS_H; ++li) {
            const LayerH &L = dProgH.layer[li];
            const int chunk = L.G32 * L.KB32 * 1024;
            for (int k = 0; k < L.NT32 / L.G32; ++k) tab[c++] = (uint32_t)(L.stream_off + (long)k * chunk);
        }
        tab[c] = (uint32_t)STREAM_HW;
    }
}

// ---- per-frame conditioning ---------------------------------------------------------------
__device__ __forceinline__ float lrelu02(float x) { return x > 0.0f ? x : x * 0.02f; }

__global__ void __launch_bounds__(256) fold_conditioning_kernel(const float *__restrict__ flat, const float *__restrict__ audio,
                                                                const float *__restrict__ pose, int pose_ld,
                                                                float *__restrict__ frame)
{
    __shared__ float a[64 * 16], b[64 * 16], drv[D_DRV], p36[D_POSE];
    const int tid = threadIdx.x;
#if SAHS_MODEL != 0
    // NeRFaceModel: the driving vector is the 76-d expression itself (models.py:368); `audio` points at it
    (void)a; (void)b;
    if (tid < D_DRV) { drv[tid] = audio[tid]; frame[FRAME_DRV_OFF + tid] = audio[tid]; }
#else
    // AudioNet: rows 0:16 of the window, permuted to (29,16) (modules.py:69-70)
    for (int e = tid; e < 29 * 16; e += 256) { int c = e / 16, t = e % 16; a[c * 16 + t] = audio[t * 29 + c]; }
    __syncthreads();
    const int cin[4] = {29, 32, 32, 64}, cout[4] = {32, 32, 64, 64};
    int L = 16;
    float *src = a, *dst = b;
    for (int l = 0; l < 4; ++l) {
        const int Lo = L / 2;
        const float *w = flat + dFlat.conv_w[l], *bs = flat + dFlat.conv_b[l];
        for (int e = tid; e < cout[l] * Lo; e += 256) {
            const int o = e / Lo, t = e % Lo;
            float s = bs[o];
            for (int c = 0; c < cin[l]; ++c)
                for (int k = 0; k < 3; ++k) {
                    const int ti = 2 * t + k - 1;
                    if (ti >= 0 && ti < L) s = fmaf(w[(o * cin[l] + c) * 3 + k], src[c * L + ti], s);
                }
            dst[o * Lo + t] = lrelu02(s);
        }
        __syncthreads();
        float *tmp = src; src = dst; dst = tmp;
        L = Lo;
    }
    // src: (64). fc1: 64->64 lrelu(0.02), 64->76 (modules.py:62-66)
    if (tid < 64) {
        float s = flat[dFlat.fc_b[0] + tid];
        for (int k = 0; k < 64; ++k) s = fmaf(flat[dFlat.fc_w[0] + tid * 64 + k], src[k], s);
        dst[tid] = lrelu02(s);
    }
    __syncthreads();
    if (tid < D_DRV) {
        float s = flat[dFlat.fc_b[1] + tid];
        for (int k = 0; k < 64; ++k) s = fmaf(flat[dFlat.fc_w[1] + tid * 64 + k], dst[k], s);
        drv[tid] = s;
        frame[FRAME_DRV_OFF + tid] = s;
    }
#endif
    // pose -> euler + translation -> PE(L=3, no input) (models.py:482-504, 203-207)
    if (tid < D_POSE) {
        const int kf = tid / 12, fn = (tid % 12) / 6, i = tid % 6;
        float v;
        if (i == 0) v = atan2f(pose[2 * pose_ld + 2], pose[1 * pose_ld + 2]);
        else if (i == 1) v = asinf(-pose[0 * pose_ld + 2]);
        else if (i == 2) v = atan2f(pose[0 * pose_ld + 0], -pose[0 * pose_ld + 1]);
        else v = pose[(i - 3) * pose_ld + 3];
        const float arg = v * (float)(1 << kf);
        const float e = fn ? cosf(arg) : sinf(arg);
        p36[tid] = e;
        frame[FRAME_POSE_OFF + tid] = e;
    }
    __syncthreads();
    // biases (static + folded constants), both levels: one workgroup per (level, layer) -- every workgroup recomputes the tiny
    // conditioning above (so there is no second launch), then folds its own layer; a single workgroup took 0.3 ms per frame
    for (int level = 0; level < 2; ++level) {
        float *bias = frame + FRAME_BIAS_OFF + level * BIAS_FLOATS;
        for (int li = 0; li < NUM_LAYERS; ++li) {
            if ((int)blockIdx.x != level * NUM_LAYERS + li) continue;
            const Layer &Ly = dProg.layer[li];
            if (!Ly.has_bias) continue;   // "A" halves start from the "B" half's pre-activations
            const int rows = Ly.bias_shared ? Ly.src_rows : Ly.NT * 16;
            for (int r = tid; r < rows; r += 256) {
                float s = 0.0f;
                if (r < Ly.src_rows) {
                    s = flat[Ly.b_off[level] + r];
                    const float *wrow = flat + Ly.w_off[level] + (long)r * Ly.src_ld;
                    for (int f = 0; f < Ly.nfold; ++f) {
                        const float *cv = Ly.fold[f].which ? p36 : drv;
                        for (int k = 0; k < Ly.fold[f].count; ++k) s = fmaf(wrow[Ly.fold[f].src_col + k], cv[k], s);
                    }
                }
                bias[Ly.bias_off + (Ly.bias_shared ? Ly.row_shift : 0) + r] = s;
            }
        }
    }
}

}  // namespace SAHS_NS

using namespace SAHS_NS;

extern "C" int SAHS_SYM(sahs_pack_weights_f32_launch)(const float *flat, float *packed, hipStream_t stream)
{
    pack_stream_f32_kernel<<<2048, 256, 0, stream>>>(flat, packed);
    pack_grid_f32_kernel<<<1024, 256, 0, stream>>>(flat, packed);
    pack_table_kernel<<<1, 64, 0, stream>>>(packed);
    return (int)hipGetLastError();
}

extern "C" int SAHS_SYM(sahs_pack_weights_bf16_launch)(const float *flat, float *packed, hipStream_t stream)
{
    pack_stream_bf16_kernel<<<1024, 256, 0, stream>>>(flat, packed);
    pack_grid_f32_kernel<<<1024, 256, 0, stream>>>(flat, packed);    // grid stays fp32, channel-last, at the same offset
    pack_table_bf16_kernel<<<1, 64, 0, stream>>>(packed);
    return (int)hipGetLastError();
}


// [grid fp32 channel-last][hi/lo streams of both levels] for field_bf16x3.hip
extern "C" int SAHS_SYM(sahs_pack_weights_bf16x3_launch)(const float *flat, float *packed, hipStream_t stream)
{
    pack_stream_bf16x3_kernel<<<1024, 256, 0, stream>>>(flat, packed);
    pack_grid_f32_kernel<<<1024, 256, 0, stream>>>(flat, packed);    // same offset as in the other packs (PACKX_GRID_OFF == 0)
    return (int)hipGetLastError();
}
extern "C" long SAHS_SYM(sahs_layout_packed_words_bf16x3)(void) { return hb::PACKX_WORDS; }

extern "C" int SAHS_SYM(sahs_fold_conditioning_launch)(const float *flat, const float *audio, const float *pose, int pose_ld, float *frame,
                                             hipStream_t stream)
{
    fold_conditioning_kernel<<<2 * NUM_LAYERS, 256, 0, stream>>>(flat, audio, pose, pose_ld, frame);
    return (int)hipGetLastError();
}

// sizes of this model's buffers, for the C ABI (capi.hip is built once and cannot see both models' constants)
extern "C" long SAHS_SYM(sahs_layout_param_count)(void) { return kFlat.total; }
extern "C" long SAHS_SYM(sahs_layout_packed_words_f32)(void) { return PACK_FLOATS; }
extern "C" long SAHS_SYM(sahs_layout_packed_words_bf16)(void) { return hb::PACKH_WORDS; }
extern "C" long SAHS_SYM(sahs_layout_frame_words)(void) { return FRAME_FLOATS; }
extern "C" long SAHS_SYM(sahs_layout_act_words)(void) { return act::STRIDE; }
// multiply-accumulates per sample evaluation that the field kernel ISSUES (padded tiles and k-blocks of the layer program; the
// per-frame constant columns are folded into biases and not multiplied): the denominator of an executed-MFMA utilisation
// part: 0 whole network, 1 deformation nets (the layers in front of the radiance trunk), 2 radiance net
extern "C" long SAHS_SYM(sahs_layout_executed_macs)(int precision, int part)
{
    long m = 0;
    if (precision == 0) {
        for (int i = 0; i < NUM_LAYERS; ++i)
            if (part == 0 || (part == 1) == (i < L_T0)) m += (long)kProg.layer[i].NT * 16 * kProg.layer[i].KB * 16;
    } else {
        for (int i = 0; i < hb::NUM_LAYERS_H; ++i)
            if (part == 0 || (part == 1) == (i < hb::H_T0)) m += (long)hb::kProgH.layer[i].NT32 * 32 * hb::kProgH.layer[i].KB32 * 32;
    }
    return m;
}
