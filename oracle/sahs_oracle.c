/*
 * sahs_oracle.c -- CPU restatement of the SAHS deformable-NeRF volume-rendering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  The shipped path (sahs-deformable-nerf_amd/) never
 * links, imports or falls back to it.
 *
 * Parity status: PINNED.  tests/golden/make_golden.py imports the real reference
 * (/root/reference/nerf-pytorch/nerf, unmodified) in the build container and commits its
 * inputs/outputs at every seam as .npz files under tests/golden; tests/test_oracle_vs_golden.py checks
 * every function below against those vectors.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/nerf-pytorch/nerf/).  Arithmetic is plain fp32 in a *defined* order
 * (sequential over the reduction index; dense layers are a k-ordered fmaf chain starting from
 * the bias, which is also what a gfx950 f32 MFMA computes).  Build with -ffp-contract=off so
 * that no product/sum outside the explicit fmaf() calls is fused.
 *
 * Architectures covered (compile-time switch SAHS_MODEL, one shared object each):
 *   0  AudioFaceModel built from config/audio/person_2_auto.yml (all three config/audio yml files share it);
 *   2  NeRFaceModel built from config/expression/person_1.yml: no deformation nets, 10 octaves, trunk fed [PE63 | expression];
 *   1  NeRFaceModel built from config/expression/person_2.yml / person_3.yml (models.py:189-370): the same graph with
 *      15-octave position encodings, a 1-D ambient coordinate encoded without its input, a 4-layer trunk fed with the
 *      76-d expression vector instead of the pose encoding, and no AudioNet (the expression IS the driving vector).
 * See the L_x / D_x constants.  The flat parameter buffer is the model's state_dict, tensors concatenated in state_dict
 * order (the "canonical order" of sahs-deformable-nerf_amd/weights.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef SAHS_MODEL
#define SAHS_MODEL 0
#endif
/* ---- architecture constants (person_2_auto.yml + the code defaults that override it) ---- */
#if SAHS_MODEL == 0
#define L_XYZ 10            /* person_2_auto.yml:93  num_encoding_fn_xyz                     */
#define L_AMB 4             /* person_2_auto.yml:62  num_encoding_fn_ambient                 */
#define AMB 2               /* person_2_auto.yml:64  ambient_coord_dim                       */
#define AMB_INC 1           /* person_2_auto.yml:60  include_input_ambient                   */
#define TR_LAYERS 8         /* person_2_auto.yml:78                                          */
#define TRUNK_SEES_POSE 1   /* models.py:430-470: use_pose=True, include_driving=False       */
#define HAS_AUDIONET 1
#define USE_DEFORM 1
#elif SAHS_MODEL == 1
#define L_XYZ 15            /* expression/person_2.yml:70,85,111 num_encoding_fn_xyz          */
#define L_AMB 15            /* expression/person_2.yml:79 num_encoding_fn_ambient             */
#define AMB 1               /* expression/person_2.yml:81 ambient_coord_dim                   */
#define AMB_INC 0           /* expression/person_2.yml:78 include_input_ambient: False        */
#define TR_LAYERS 4         /* expression/person_2.yml:98 num_layers                          */
#define TRUNK_SEES_POSE 0   /* expression/person_2.yml:124-127: include_driving True, use_pose False */
#define HAS_AUDIONET 0
#define USE_DEFORM 1
#else                       /* 2: NeRFaceModel, config/expression/person_1.yml: use_warp False (:67), use_ambient False (:78) */
#define L_XYZ 10
#define L_AMB 0
#define AMB 0
#define AMB_INC 0
#define TR_LAYERS 4
#define TRUNK_SEES_POSE 0
#define HAS_AUDIONET 0
#define USE_DEFORM 0        /* models.py:231,244: no warp_field_mlp / hyper_sheep_mlp modules; map_points returns the points */
#endif
#define L_DIR 4             /* person_2_auto.yml:100 num_encoding_fn_dir                     */
#define L_POSE 3            /* models.py:203-207 encode_pose_fn, include_input=False         */
#define D_XYZ (3 + 6 * L_XYZ)          /* 63 | 93 */
#define D_DIR (3 + 6 * L_DIR)          /* 27 */
#define D_AMB (AMB_INC * AMB + 2 * AMB * L_AMB)  /* 18 | 30 */
#define D_POSE (2 * 6 * L_POSE)        /* 36 */
#define D_DRV 76                       /* modules.py:44 dim_aud                               */
#define D_GRID 32                      /* models.py:201 channels of spatial_embeddings        */
#define G_RES 32                       /* models.py:201 grid resolution                       */
#define D_DEF_IN (D_XYZ + D_DRV + D_POSE)       /* 175: modules.py:352-357                   */
#define WARP_H 128                     /* person_2_auto.yml:51                                */
#define HYP_H 64                       /* person_2_auto.yml:67                                */
#define DEF_LAYERS 6                   /* person_2_auto.yml:50,66                             */
#define DEF_SKIP 4                     /* person_2_auto.yml:52,68                             */
#define TR_H 256                       /* person_2_auto.yml:81                                */
#define TR_SKIP 3                      /* modules.py:176 default; cfg value never forwarded   */
#define D_TR_CONST (TRUNK_SEES_POSE ? D_POSE : D_DRV)   /* per-frame constant part of the trunk input */
#define D_TR_IN (D_XYZ + D_AMB + D_TR_CONST)    /* 117 | 199: modules.py:203-228             */
#define BR_H 128                       /* modules.py:232,239 hidden_size // 2                 */
#define D_DIR_IN (TR_H + D_DIR + D_GRID)        /* 315: modules.py:234                       */
#define N_SEG 12                       /* modules.py:244                                      */
#define D_RAW 16                       /* [rgb3, seg12, sigma]: modules.py:295                */

#define PB 16 /* points processed together (vector lanes) in the field restatement */

/* ---- canonical flat-buffer offsets (state_dict order) ---- */
typedef struct {
    const float *grid;
    const float *warp_w[DEF_LAYERS], *warp_b[DEF_LAYERS], *warp_fw, *warp_fb;
    const float *hyp_w[DEF_LAYERS], *hyp_b[DEF_LAYERS], *hyp_fw, *hyp_fb;
    struct {
        const float *xyz_w[TR_LAYERS], *xyz_b[TR_LAYERS];
        const float *feat_w, *feat_b, *alpha_w, *alpha_b;
        const float *dir_w[4], *dir_b[4], *rgb_w, *rgb_b;
        const float *seg_w[4], *seg_b[4], *segout_w, *segout_b;
    } lvl[2];
    const float *conv_w[4], *conv_b[4], *fc_w[2], *fc_b[2];
    long total;
} model_t;

static const float *take(const float **p, long n) { const float *r = *p; *p += n; return r; }

static void model_bind(model_t *m, const float *flat)
{
    const float *p = flat;
    m->grid = take(&p, (long)D_GRID * G_RES * G_RES * G_RES);
#if USE_DEFORM
    for (int i = 0; i < DEF_LAYERS; ++i) {
        int in = (i == 0) ? D_DEF_IN : (i == DEF_SKIP ? WARP_H + D_DEF_IN : WARP_H);
        m->warp_w[i] = take(&p, (long)WARP_H * in); m->warp_b[i] = take(&p, WARP_H);
    }
    m->warp_fw = take(&p, 3 * WARP_H); m->warp_fb = take(&p, 3);
    for (int i = 0; i < DEF_LAYERS; ++i) {
        int in = (i == 0) ? D_DEF_IN : (i == DEF_SKIP ? HYP_H + D_DEF_IN : HYP_H);
        m->hyp_w[i] = take(&p, (long)HYP_H * in); m->hyp_b[i] = take(&p, HYP_H);
    }
    m->hyp_fw = take(&p, AMB * HYP_H); m->hyp_fb = take(&p, AMB);
#endif
    for (int l = 0; l < 2; ++l) {
        for (int i = 0; i < TR_LAYERS; ++i) {
            int in = (i == 0) ? D_TR_IN : (i == TR_SKIP ? TR_H + D_TR_IN : TR_H);
            m->lvl[l].xyz_w[i] = take(&p, (long)TR_H * in); m->lvl[l].xyz_b[i] = take(&p, TR_H);
        }
        m->lvl[l].feat_w = take(&p, TR_H * TR_H); m->lvl[l].feat_b = take(&p, TR_H);
        m->lvl[l].alpha_w = take(&p, TR_H); m->lvl[l].alpha_b = take(&p, 1);
        for (int i = 0; i < 4; ++i) {
            int in = (i == 0) ? D_DIR_IN : BR_H;
            m->lvl[l].dir_w[i] = take(&p, (long)BR_H * in); m->lvl[l].dir_b[i] = take(&p, BR_H);
        }
        m->lvl[l].rgb_w = take(&p, 3 * BR_H); m->lvl[l].rgb_b = take(&p, 3);
        for (int i = 0; i < 4; ++i) {
            int in = (i == 0) ? TR_H : BR_H;
            m->lvl[l].seg_w[i] = take(&p, (long)BR_H * in); m->lvl[l].seg_b[i] = take(&p, BR_H);
        }
        m->lvl[l].segout_w = take(&p, N_SEG * BR_H); m->lvl[l].segout_b = take(&p, N_SEG);
    }
#if HAS_AUDIONET
    static const int cin[4] = {29, 32, 32, 64}, cout[4] = {32, 32, 64, 64};
    for (int i = 0; i < 4; ++i) {
        m->conv_w[i] = take(&p, (long)cout[i] * cin[i] * 3); m->conv_b[i] = take(&p, cout[i]);
    }
    m->fc_w[0] = take(&p, 64 * 64); m->fc_b[0] = take(&p, 64);
    m->fc_w[1] = take(&p, D_DRV * 64); m->fc_b[1] = take(&p, D_DRV);
#endif
    m->total = (long)(p - flat);
}

int oracle_model(void) { return SAHS_MODEL; }

/* Number of floats in the flat parameter buffer (2,775,633 for person_2_auto.yml). */
long oracle_param_count(void)
{
    model_t m; model_bind(&m, (const float *)0);
    return m.total;
}

/* ------------------------------------------------------------------------------------------
 * get_ray_bundle: nerf_helpers.py:178-233 (+ meshgrid_xy :84-96).
 * ii[h,w]=w, jj[h,w]=h; d=((ii-W*cx)/fx, -(jj-H*cy)/fy, -1); rd_i = sum_j d_j*c2w[i][j];
 * ro = c2w[:3,3].  c2w is row-major with row stride `ld` (4 for a 3x4 or 4x4 pose).
 * ---------------------------------------------------------------------------------------- */
void oracle_get_ray_bundle(int H, int W, const float *intr, const float *c2w, int ld,
                           float *ro, float *rd)
{
    for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
            float d[3];
            d[0] = ((float)w - (float)W * intr[2]) / intr[0];
            d[1] = -((float)h - (float)H * intr[3]) / intr[1];
            d[2] = -1.0f;
            float *o = ro + ((long)h * W + w) * 3, *r = rd + ((long)h * W + w) * 3;
            for (int i = 0; i < 3; ++i) {
                /* torch.sum(directions[..., None, :] * c2w[:3,:3], dim=-1): 3 products, summed in order */
                float s = d[0] * c2w[i * ld + 0];
                s = s + d[1] * c2w[i * ld + 1];
                s = s + d[2] * c2w[i * ld + 2];
                r[i] = s;
                o[i] = c2w[i * ld + 3];
            }
        }
}

/* ------------------------------------------------------------------------------------------
 * AudioNet: modules.py:43-73.  audio (16,29) -> rows 0:16 -> permute to (29,16) ->
 * 4 x [Conv1d(k3,s2,p1) + LeakyReLU(0.02)] -> (64,1) -> Linear 64->64, LeakyReLU(0.02),
 * Linear 64->76.
 * ---------------------------------------------------------------------------------------- */
#if HAS_AUDIONET
static float lrelu(float x, float slope) { return x > 0.0f ? x : x * slope; }

static void audionet(const model_t *m, const float *audio, float *driving)
{
    static const int cin[4] = {29, 32, 32, 64}, cout[4] = {32, 32, 64, 64};
    float a[64 * 16], b[64 * 16];
    int L = 16;
    for (int c = 0; c < 29; ++c)
        for (int t = 0; t < 16; ++t) a[c * 16 + t] = audio[t * 29 + c]; /* permute(0,2,1) :70 */
    for (int l = 0; l < 4; ++l) {
        int Lo = L / 2;
        for (int o = 0; o < cout[l]; ++o)
            for (int t = 0; t < Lo; ++t) {
                float s = m->conv_b[l][o];
                for (int c = 0; c < cin[l]; ++c)
                    for (int k = 0; k < 3; ++k) {
                        int ti = 2 * t + k - 1;
                        if (ti >= 0 && ti < L) s = fmaf(m->conv_w[l][(o * cin[l] + c) * 3 + k], a[c * L + ti], s);
                    }
                b[o * Lo + t] = lrelu(s, 0.02f);
            }
        L = Lo;
        memcpy(a, b, sizeof(float) * cout[l] * L);
    }
    float h[64];
    for (int o = 0; o < 64; ++o) {
        float s = m->fc_b[0][o];
        for (int k = 0; k < 64; ++k) s = fmaf(m->fc_w[0][o * 64 + k], a[k], s);
        h[o] = lrelu(s, 0.02f);
    }
    for (int o = 0; o < D_DRV; ++o) {
        float s = m->fc_b[1][o];
        for (int k = 0; k < 64; ++k) s = fmaf(m->fc_w[1][o * 64 + k], h[k], s);
        driving[o] = s;
    }
}
#endif

/* NeRFaceModel has no AudioNet: its driving vector is the 76-d expression itself (models.py:368). */
void oracle_audionet(const float *flat, const float *audio, float *driving)
{
#if HAS_AUDIONET
    model_t m; model_bind(&m, flat);
    audionet(&m, audio, driving);
#else
    (void)flat;
    memcpy(driving, audio, sizeof(float) * D_DRV);
#endif
}

/* ------------------------------------------------------------------------------------------
 * Pose conditioning: models.py:482-504 (rot_to_euler, pose_to_euler_trans) followed by
 * encode_pose_fn (models.py:203-207; positional_encoding nerf_helpers.py:305-349 with L=3,
 * include_input=False).  pose is 3x4 (or 4x4) row-major with row stride ld.
 * ---------------------------------------------------------------------------------------- */
void oracle_pose_encoding(const float *pose, int ld, float *pose36)
{
    float p6[6];
    p6[0] = atan2f(pose[2 * ld + 2], pose[1 * ld + 2]); /* e0 = atan2(R22, R12) */
    p6[1] = asinf(-pose[0 * ld + 2]);                   /* e1 = asin(-R02)      */
    p6[2] = atan2f(pose[0 * ld + 0], -pose[0 * ld + 1]);/* e2 = atan2(R00,-R01) */
    p6[3] = pose[0 * ld + 3]; p6[4] = pose[1 * ld + 3]; p6[5] = pose[2 * ld + 3];
    int o = 0;
    for (int k = 0; k < L_POSE; ++k) {
        float f = (float)(1 << k);
        for (int i = 0; i < 6; ++i) pose36[o++] = sinf(p6[i] * f);
        for (int i = 0; i < 6; ++i) pose36[o++] = cosf(p6[i] * f);
    }
}

/* positional_encoding, nerf_helpers.py:305-349: [x] ++ for k: [sin(2^k x), cos(2^k x)] */
void oracle_positional_encoding(const float *x, int d, int L, int include_input, float *out)
{
    int o = 0;
    if (include_input) for (int i = 0; i < d; ++i) out[o++] = x[i];
    for (int k = 0; k < L; ++k) {
        float f = (float)(1 << k);
        for (int i = 0; i < d; ++i) out[o++] = sinf(x[i] * f);
        for (int i = 0; i < d; ++i) out[o++] = cosf(x[i] * f);
    }
}

/* ---- dense layer on a block of PB points; activations stored transposed [feature][PB] ---- */
/* y[o][p] = b[o] + sum_k W[o][k] * x[k][p], k ascending, one fmaf per term. */
static void dense(const float *W, const float *b, int out, int in, const float *x, float *y)
{
    for (int o = 0; o < out; ++o) {
        float acc[PB];
        for (int p = 0; p < PB; ++p) acc[p] = b[o];
        const float *w = W + (long)o * in;
        for (int k = 0; k < in; ++k) {
            const float wk = w[k];
            const float *xk = x + (long)k * PB;
#pragma omp simd
            for (int p = 0; p < PB; ++p) acc[p] = fmaf(wk, xk[p], acc[p]);
        }
        for (int p = 0; p < PB; ++p) y[(long)o * PB + p] = acc[p];
    }
}

#if USE_DEFORM
static void act_relu(float *x, int n) { for (int i = 0; i < n * PB; ++i) x[i] = x[i] > 0.0f ? x[i] : 0.0f; }
#endif
static void act_lrelu(float *x, int n, float s) { for (int i = 0; i < n * PB; ++i) x[i] = x[i] > 0.0f ? x[i] : x[i] * s; }

/* PE of a d-vector per point, transposed layout: out[(feature)][p] */
static void pe_block(const float *x /*[d][PB]*/, int d, int L, int include_input, float *out)
{
    int o = 0;
    if (include_input) for (int i = 0; i < d; ++i, ++o) for (int p = 0; p < PB; ++p) out[o * PB + p] = x[i * PB + p];
    for (int k = 0; k < L; ++k) {
        float f = (float)(1 << k);
        for (int i = 0; i < d; ++i, ++o) for (int p = 0; p < PB; ++p) out[o * PB + p] = sinf(x[i * PB + p] * f);
        for (int i = 0; i < d; ++i, ++o) for (int p = 0; p < PB; ++p) out[o * PB + p] = cosf(x[i * PB + p] * f);
    }
}

/* WarpFieldMLP / HyperSheetMLP trunk: modules.py:371-388 / 444-460. relu MLP with the
 * skip layer consuming cat(h, initial). */
#if USE_DEFORM
static void deform_mlp(const float *const *W, const float *const *B, int hid, const float *in175, float *h /*[hid+175][PB]*/, float *tmp)
{
    /* layer 0 */
    dense(W[0], B[0], hid, D_DEF_IN, in175, tmp);
    act_relu(tmp, hid);
    for (int i = 1; i < DEF_LAYERS; ++i) {
        if (i == DEF_SKIP) {
            memcpy(h, tmp, sizeof(float) * hid * PB);
            memcpy(h + (long)hid * PB, in175, sizeof(float) * D_DEF_IN * PB); /* cat((x, initial)) */
            dense(W[i], B[i], hid, hid + D_DEF_IN, h, tmp);
        } else {
            memcpy(h, tmp, sizeof(float) * hid * PB);
            dense(W[i], B[i], hid, hid, h, tmp);
        }
        act_relu(tmp, hid);
    }
    memcpy(h, tmp, sizeof(float) * hid * PB);
}
#endif

/* 5-D grid_sample, bilinear, zeros padding, align_corners=True: models.py:346-365 calling
 * torch.nn.functional.grid_sample (ATen GridSampler.cpp grid_sampler_3d_cpu_impl):
 * x indexes W (last dim), y -> H, z -> D; corner weights and accumulation order as ATen. */
static void grid_sample_pt(const float *grid, float x, float y, float z, float *out /*[32]*/)
{
    const int R = G_RES;
    float ix = ((x + 1.0f) / 2.0f) * (float)(R - 1);
    float iy = ((y + 1.0f) / 2.0f) * (float)(R - 1);
    float iz = ((z + 1.0f) / 2.0f) * (float)(R - 1);
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    float x0 = fx, x1 = fx + 1.0f, y0 = fy, y1 = fy + 1.0f, z0 = fz, z1 = fz + 1.0f;
    float wt[8];
    wt[0] = (x1 - ix) * (y1 - iy) * (z1 - iz); /* tnw: (x0,y0,z0) */
    wt[1] = (ix - x0) * (y1 - iy) * (z1 - iz); /* tne: (x1,y0,z0) */
    wt[2] = (x1 - ix) * (iy - y0) * (z1 - iz); /* tsw: (x0,y1,z0) */
    wt[3] = (ix - x0) * (iy - y0) * (z1 - iz); /* tse: (x1,y1,z0) */
    wt[4] = (x1 - ix) * (y1 - iy) * (iz - z0); /* bnw: (x0,y0,z1) */
    wt[5] = (ix - x0) * (y1 - iy) * (iz - z0); /* bne */
    wt[6] = (x1 - ix) * (iy - y0) * (iz - z0); /* bsw */
    wt[7] = (ix - x0) * (iy - y0) * (iz - z0); /* bse */
    /* Range test in float first: a NaN/huge coordinate must not reach the int conversion. */
    int ok = (fx >= -1.0f && fx <= (float)R && fy >= -1.0f && fy <= (float)R && fz >= -1.0f && fz <= (float)R);
    int xi = ok ? (int)fx : -2, yi = ok ? (int)fy : -2, zi = ok ? (int)fz : -2;
    for (int c = 0; c < D_GRID; ++c) out[c] = 0.0f;
    for (int n = 0; n < 8; ++n) {
        int cx = xi + (n & 1), cy = yi + ((n >> 1) & 1), cz = zi + (n >> 2);
        if (cx < 0 || cx >= R || cy < 0 || cy >= R || cz < 0 || cz >= R) continue;
        const float *g = grid + ((long)cz * R + cy) * R + cx;
        for (int c = 0; c < D_GRID; ++c) out[c] = out[c] + g[(long)c * R * R * R] * wt[n];
    }
}

/* ------------------------------------------------------------------------------------------
 * Field forward for one block of PB points: AudioFaceModel.forward, models.py:514-528 ->
 * map_points :301-329 (warp :301-306, hyper :308-316) -> sample_from_3dgrid :346-365 ->
 * query_template :331-344 -> NeRFMLP.forward modules.py:254-295.
 * x6: [PB][6] = xyz, raw ray direction (train_utils.py:15-21; un-normalised).
 * ---------------------------------------------------------------------------------------- */
static void field_block(const model_t *m, int level, const float *x6, int xs, const float *driving,
                        const float *pose36, float *raw /*[PB][16]*/, float *dbg_dx, float *dbg_w,
                        float *dbg_grid, float *scr)
{
    float *xyz = scr;                 scr += 3 * PB;
    float *in175 = scr;               scr += D_DEF_IN * PB;
    float *h = scr;                   scr += (TR_H + D_DIR_IN) * PB;
    float *tmp = scr;                 scr += TR_H * PB;
    float *in117 = scr;               scr += D_TR_IN * PB;
    float *feat = scr;                scr += TR_H * PB;
    float *gridf = scr;               scr += D_GRID * PB;
    float *mapped = scr;              scr += 5 * PB;
    float *dirpe = scr;               scr += D_DIR * PB;
    float *rd = scr;                  scr += 3 * PB;

    for (int p = 0; p < PB; ++p)
        for (int i = 0; i < 3; ++i) { xyz[i * PB + p] = x6[p * xs + i]; rd[i * PB + p] = x6[p * xs + 3 + i]; }

#if USE_DEFORM
    /* initial = cat(PE(xyz), driving, pose): models.py:303, modules.py:372-381 */
    pe_block(xyz, 3, L_XYZ, 1, in175);
    for (int k = 0; k < D_DRV; ++k) for (int p = 0; p < PB; ++p) in175[(D_XYZ + k) * PB + p] = driving[k];
    for (int k = 0; k < D_POSE; ++k) for (int p = 0; p < PB; ++p) in175[(D_XYZ + D_DRV + k) * PB + p] = pose36[k];

    /* warp: dx = tanh(fc_final(h)); warped = xyz + dx: modules.py:388, models.py:304-305 */
    deform_mlp(m->warp_w, m->warp_b, WARP_H, in175, h, tmp);
    dense(m->warp_fw, m->warp_fb, 3, WARP_H, h, tmp);
    for (int i = 0; i < 3; ++i) for (int p = 0; p < PB; ++p) {
        float dx = tanhf(tmp[i * PB + p]);
        if (dbg_dx) dbg_dx[p * 3 + i] = dx;
        mapped[i * PB + p] = xyz[i * PB + p] + dx;
    }
    /* hyper: w = fc_ambient(h) (linear): modules.py:461 */
    deform_mlp(m->hyp_w, m->hyp_b, HYP_H, in175, h, tmp);
    dense(m->hyp_fw, m->hyp_fb, AMB, HYP_H, h, tmp);
    for (int i = 0; i < AMB; ++i) for (int p = 0; p < PB; ++p) {
        mapped[(3 + i) * PB + p] = tmp[i * PB + p];
        if (dbg_w) dbg_w[p * AMB + i] = tmp[i * PB + p];
    }
#else
    /* use_warp False, use_ambient False: map_points returns the points unchanged (models.py:316-327) */
    (void)in175; (void)pose36; (void)dbg_w;
    for (int i = 0; i < 3; ++i) for (int p = 0; p < PB; ++p) {
        mapped[i * PB + p] = xyz[i * PB + p];
        if (dbg_dx) dbg_dx[p * 3 + i] = 0.0f;
    }
#endif
    /* grid features at the warped point: models.py:525 */
    for (int p = 0; p < PB; ++p) {
        float g[D_GRID];
        grid_sample_pt(m->grid, mapped[0 * PB + p], mapped[1 * PB + p], mapped[2 * PB + p], g);
        for (int c = 0; c < D_GRID; ++c) { gridf[c * PB + p] = g[c]; if (dbg_grid) dbg_grid[p * D_GRID + c] = g[c]; }
    }
    /* template input: cat(PE10(xyz'), PE4(w), pose36): models.py:332-336, modules.py:255-266 */
    pe_block(mapped, 3, L_XYZ, 1, in117);
    if (AMB > 0) pe_block(mapped + 3 * PB, AMB, L_AMB, AMB_INC, in117 + D_XYZ * PB);
    {   /* AudioFaceModel: pose36 (use_pose); NeRFaceModel: the driving vector (include_driving), modules.py:260-267 */
        const float *cv = TRUNK_SEES_POSE ? pose36 : driving;
        for (int k = 0; k < D_TR_CONST; ++k) for (int p = 0; p < PB; ++p) in117[(D_XYZ + D_AMB + k) * PB + p] = cv[k];
    }
    pe_block(rd, 3, L_DIR, 1, dirpe); /* models.py:340 */

    /* trunk: modules.py:267-273 */
    dense(m->lvl[level].xyz_w[0], m->lvl[level].xyz_b[0], TR_H, D_TR_IN, in117, tmp);
    act_lrelu(tmp, TR_H, 0.01f);
    for (int i = 1; i < TR_LAYERS; ++i) {
        memcpy(h, tmp, sizeof(float) * TR_H * PB);
        if (i == TR_SKIP) {
            memcpy(h + (long)TR_H * PB, in117, sizeof(float) * D_TR_IN * PB);
            dense(m->lvl[level].xyz_w[i], m->lvl[level].xyz_b[i], TR_H, TR_H + D_TR_IN, h, tmp);
        } else {
            dense(m->lvl[level].xyz_w[i], m->lvl[level].xyz_b[i], TR_H, TR_H, h, tmp);
        }
        act_lrelu(tmp, TR_H, 0.01f);
    }
    /* feat = fc_feat(x) (no activation); alpha = fc_alpha(feat): modules.py:274-275 */
    dense(m->lvl[level].feat_w, m->lvl[level].feat_b, TR_H, TR_H, tmp, feat);
    float sigma[PB];
    dense(m->lvl[level].alpha_w, m->lvl[level].alpha_b, 1, TR_H, feat, sigma);
    /* colour branch: cat(feat, dirs, spatial_embedding): modules.py:276-287 */
    memcpy(h, feat, sizeof(float) * TR_H * PB);
    memcpy(h + (long)TR_H * PB, dirpe, sizeof(float) * D_DIR * PB);
    memcpy(h + (long)(TR_H + D_DIR) * PB, gridf, sizeof(float) * D_GRID * PB);
    dense(m->lvl[level].dir_w[0], m->lvl[level].dir_b[0], BR_H, D_DIR_IN, h, tmp);
    act_lrelu(tmp, BR_H, 0.01f);
    for (int i = 1; i < 4; ++i) {
        memcpy(h, tmp, sizeof(float) * BR_H * PB);
        dense(m->lvl[level].dir_w[i], m->lvl[level].dir_b[i], BR_H, BR_H, h, tmp);
        act_lrelu(tmp, BR_H, 0.01f);
    }
    float rgb[3 * PB];
    dense(m->lvl[level].rgb_w, m->lvl[level].rgb_b, 3, BR_H, tmp, rgb);
    /* seg branch: modules.py:289-294 */
    dense(m->lvl[level].seg_w[0], m->lvl[level].seg_b[0], BR_H, TR_H, feat, tmp);
    act_lrelu(tmp, BR_H, 0.01f);
    for (int i = 1; i < 4; ++i) {
        memcpy(h, tmp, sizeof(float) * BR_H * PB);
        dense(m->lvl[level].seg_w[i], m->lvl[level].seg_b[i], BR_H, BR_H, h, tmp);
        act_lrelu(tmp, BR_H, 0.01f);
    }
    float seg[N_SEG * PB];
    dense(m->lvl[level].segout_w, m->lvl[level].segout_b, N_SEG, BR_H, tmp, seg);
    for (int p = 0; p < PB; ++p) { /* cat((rgb, seg, alpha)): modules.py:295 */
        for (int i = 0; i < 3; ++i) raw[p * D_RAW + i] = rgb[i * PB + p];
        for (int i = 0; i < N_SEG; ++i) raw[p * D_RAW + 3 + i] = seg[i * PB + p];
        raw[p * D_RAW + 15] = sigma[p];
    }
}

#define FIELD_SCRATCH ((3 + D_DEF_IN + TR_H + D_DIR_IN + TR_H + D_TR_IN + TR_H + D_GRID + 5 + D_DIR + 3) * PB)

/* run_network + model forward (train_utils.py:9-50): x is (P, xs>=6) rows [xyz, rd, ...];
 * driving76 / pose36 are the per-frame conditioning vectors (every point-chunk of the
 * reference recomputes the same values, models.py:517-521).  Optional debug outputs:
 * dx (P,3), w (P,2), grid features (P,32). */
void oracle_field_forward(const float *flat, int level, long P, const float *x, int xs,
                          const float *driving76, const float *pose36, float *raw,
                          float *dbg_dx, float *dbg_w, float *dbg_grid)
{
    model_t m; model_bind(&m, flat);
    long nblk = (P + PB - 1) / PB;
#pragma omp parallel
    {
        float *scr = (float *)malloc(sizeof(float) * FIELD_SCRATCH);
#pragma omp for schedule(dynamic, 4)
        for (long b = 0; b < nblk; ++b) {
            float xb[PB * 6], rawb[PB * D_RAW], dxb[PB * 3], wb[PB * AMB], gb[PB * D_GRID];
            long p0 = b * PB; int n = (int)((P - p0) < PB ? (P - p0) : PB);
            for (int p = 0; p < PB; ++p) {
                long src = p0 + (p < n ? p : 0);
                for (int i = 0; i < 6; ++i) xb[p * 6 + i] = x[src * xs + i];
            }
            field_block(&m, level, xb, 6, driving76, pose36, rawb, dxb, wb, gb, scr);
            for (int p = 0; p < n; ++p) {
                memcpy(raw + (p0 + p) * D_RAW, rawb + p * D_RAW, sizeof(float) * D_RAW);
                if (dbg_dx) memcpy(dbg_dx + (p0 + p) * 3, dxb + p * 3, sizeof(float) * 3);
                if (dbg_w) memcpy(dbg_w + (p0 + p) * AMB, wb + p * AMB, sizeof(float) * AMB);
                if (dbg_grid) memcpy(dbg_grid + (p0 + p) * D_GRID, gb + p * D_GRID, sizeof(float) * D_GRID);
            }
        }
        free(scr);
    }
}

/* ------------------------------------------------------------------------------------------
 * Partition-invariant uniforms (sahs_ray_uniforms): Philox4x32-10 (Salmon et al. 2011; the published round function and
 * constants), key = seed, counter = (global ray lo, hi, sample / 4, stream id); word s % 4, top 24 bits -> [0,1).
 * ---------------------------------------------------------------------------------------- */
void oracle_ray_uniforms(uint64_t seed, int stream_id, long ray0, long N, int S, float *out)
{
    for (long r = 0; r < N; ++r)
        for (int b = 0; b < (S + 3) / 4; ++b) {
            uint64_t gr = (uint64_t)(ray0 + r);
            uint32_t c[4] = {(uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)b, (uint32_t)stream_id};
            uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
            for (int round = 0; round < 10; ++round) {
                uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
                uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
                c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
                k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
            }
            for (int i = 0; i < 4; ++i)
                if (4 * b + i < S) out[r * S + 4 * b + i] = (float)(c[i] >> 8) * 5.9604644775390625e-08f;
        }
}

/* ------------------------------------------------------------------------------------------
 * Coarse depths: train_utils.py:93-113.  t=linspace(0,1,S); z=near*(1-t)+far*t (or lindisp);
 * perturb: mids, upper/lower, z = lower + (upper-lower)*t_rand.
 * torch.linspace(0,1,S): step=(1-0)/(S-1); i<S/2 ? start+step*i : end-step*(S-1-i), the second half as ONE fused multiply-add (ATen's
 * CPU kernel is compiled with contraction on; checked against torch.linspace for every S <= 300 in tests/test_oracle_vs_golden.py).
 * ---------------------------------------------------------------------------------------- */
static float aten_linspace01(int i, int n, float step)
{
    if (n == 1) return 0.0f;   /* torch.linspace(0, 1, 1) = [start] */
    return (i < n / 2) ? step * (float)i : fmaf(-step, (float)(n - 1 - i), 1.0f);
}
void oracle_linspace01(int n, float *out)
{
    const float step = (n > 1) ? 1.0f / (float)(n - 1) : 0.0f;
    for (int i = 0; i < n; ++i) out[i] = aten_linspace01(i, n, step);
}

void oracle_stratified_depths(long N, int S, const float *near_, const float *far_, int lindisp,
                              const float *t_rand, float *z)
{
    float *t = (float *)malloc(sizeof(float) * S);
    float step = (1.0f - 0.0f) / (float)(S - 1);
    for (int i = 0; i < S; ++i) t[i] = aten_linspace01(i, S, step);
#pragma omp parallel for
    for (long r = 0; r < N; ++r) {
        float *zr = z + r * S;
        float n = near_[r], f = far_[r];
        for (int i = 0; i < S; ++i) {
            if (!lindisp) zr[i] = n * (1.0f - t[i]) + f * t[i];
            else zr[i] = 1.0f / (1.0f / n * (1.0f - t[i]) + 1.0f / f * t[i]);
        }
        if (t_rand) {
            float z0[1024];
            for (int i = 0; i < S; ++i) z0[i] = zr[i];
            for (int i = 0; i < S; ++i) {
                float upper = (i + 1 < S) ? 0.5f * (z0[i + 1] + z0[i]) : z0[S - 1]; /* cat(mids, z[-1]) */
                float lower = (i > 0) ? 0.5f * (z0[i] + z0[i - 1]) : z0[0];         /* cat(z[0], mids)  */
                zr[i] = lower + (upper - lower) * t_rand[r * S + i];
            }
        }
    }
    free(t);
}

/* pts = ro + rd * z (train_utils.py:115,168) packed as rows [xyz, rd] for run_network. */
void oracle_make_points(long N, int S, const float *ro, const float *rd, const float *z, float *x6)
{
#pragma omp parallel for
    for (long r = 0; r < N; ++r)
        for (int s = 0; s < S; ++s) {
            float *o = x6 + (r * S + s) * 6;
            for (int i = 0; i < 3; ++i) { o[i] = ro[r * 3 + i] + rd[r * 3 + i] * z[r * S + s]; o[3 + i] = rd[r * 3 + i]; }
        }
}

/* ------------------------------------------------------------------------------------------
 * volume_render_radiance_field: volume_rendering_utils.py:7-78 (cumprod_exclusive
 * nerf_helpers.py:99-120), with the caller's background overwrite train_utils.py:135-136
 * applied first when bg != NULL (raw is modified in place, as in the reference).
 * noise: (N,S) standard-normal draws already multiplied by radiance_field_noise_std, or NULL.
 * Outputs: rgb (N,15), disp, acc, weights (N,S), depth.
 * ---------------------------------------------------------------------------------------- */
void oracle_composite(long N, int S, float *raw, const float *z, const float *rd, const float *noise,
                      const float *bg, int white_bkgd, float *rgb_map, float *disp, float *acc,
                      float *weights, float *depth)
{
#pragma omp parallel for
    for (long r = 0; r < N; ++r) {
        float *rw = raw + r * S * D_RAW;
        const float *zr = z + r * S;
        if (bg) for (int c = 0; c < 15; ++c) rw[(S - 1) * D_RAW + c] = bg[r * 15 + c];
        float nrm = sqrtf(rd[r * 3] * rd[r * 3] + rd[r * 3 + 1] * rd[r * 3 + 1] + rd[r * 3 + 2] * rd[r * 3 + 2]);
        float T = 1.0f, out[15], dsum = 0.0f, asum = 0.0f;
        for (int c = 0; c < 15; ++c) out[c] = 0.0f;
        for (int s = 0; s < S; ++s) {
            const float *q = rw + s * D_RAW;
            float col[15];
            if (bg && s == S - 1) {
                for (int c = 0; c < 15; ++c) col[c] = q[c];        /* the prior, verbatim :33 */
            } else if (bg) {
                for (int c = 0; c < 3; ++c) col[c] = 1.0f / (1.0f + expf(-q[c]));   /* :29 */
                float mx = q[3];
                for (int c = 4; c < 15; ++c) mx = q[c] > mx ? q[c] : mx;
                float e[12], es = 0.0f;
                for (int c = 0; c < 12; ++c) { e[c] = expf(q[3 + c] - mx); es += e[c]; }
                for (int c = 0; c < 12; ++c) col[3 + c] = e[c] / es;               /* :31 */
            } else {
                for (int c = 0; c < 15; ++c) col[c] = 1.0f / (1.0f + expf(-q[c]));  /* :35 */
            }
            float dist = (s + 1 < S) ? (zr[s + 1] - zr[s]) : 1e10f;
            dist = dist * nrm;
            float sg = q[15] + (noise ? noise[r * S + s] : 0.0f);
            sg = sg > 0.0f ? sg : 0.0f;
            if (s == S - 1) sg += 1e-6f;                                            /* :57 */
            float alpha = 1.0f - expf(-sg * dist);
            float w = alpha * T;                                                    /* :59 */
            T = T * ((1.0f - alpha) + 1e-10f);
            weights[r * S + s] = w;
            for (int c = 0; c < 15; ++c) out[c] += w * col[c];
            dsum += w * zr[s];
            asum += w;
        }
        if (white_bkgd) for (int c = 0; c < 15; ++c) out[c] = out[c] + (1.0f - asum);
        for (int c = 0; c < 15; ++c) rgb_map[r * 15 + c] = out[c];
        depth[r] = dsum; acc[r] = asum;
        float dd = dsum / asum;
        disp[r] = 1.0f / (dd > 1e-10f ? dd : 1e-10f);  /* torch.max(1e-10, x): NaN propagates */
        if (dd != dd) disp[r] = dd;
    }
}

/* ------------------------------------------------------------------------------------------
 * sample_pdf_2: nerf_helpers.py:454-497.  bins (N,nb), weights (N,nb-1) -> samples (N,ns).
 * u == NULL means det=True (u = linspace(0,1,ns)).  inds (N,ns) int64 optional output
 * (searchsorted(cdf, u, right=True)).
 * ---------------------------------------------------------------------------------------- */
/* torch.sum of a contiguous float row as ATen's CPU kernel forms it (aten/src/ATen/native/cpu/SumKernel.cpp: vectorized_inner_sum ->
 * row_sum -> multi_row_sum; the kernel is built for 8-lane vectors also where ATen reports AVX512): the row is read as n/8 vectors of 8
 * lanes, accumulated into FOUR interleaved vector accumulators (vector 4i+k -> accumulator k; the vectors beyond the last full group of
 * four -> accumulator 0), accumulators 1..3 are added to 0 in order, then one scalar takes the n%8 trailing elements in order followed by
 * the eight lanes in order.  All in fp32.  (The cascade levels of multi_row_sum only engage from 16 groups of four = 512 elements; n < 512
 * here.)  Rows shorter than one vector take the scalar variant of the same scheme (four scalar accumulators).  Restated here because a cdf
 * knot that moves by an ulp moves a searchsorted index; checked against torch.sum itself for every n in tests/test_oracle_vs_golden.py. */
static float aten_sum_f32(const float *x, int n)
{
    if (n < 8) {
        float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const int g = n / 4;
        for (int i = 0; i < g; ++i) for (int k = 0; k < 4; ++k) p[k] += x[4 * i + k];
        for (int i = 4 * g; i < n; ++i) p[0] += x[i];
        p[0] += p[1]; p[0] += p[2]; p[0] += p[3];
        return p[0];
    }
    const int vs = n / 8, g = vs / 4;
    float acc = 0.0f;
    for (int k = vs * 8; k < n; ++k) acc += x[k];
    for (int l = 0; l < 8; ++l) {
        float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = 0; i < g; ++i) for (int k = 0; k < 4; ++k) p[k] += x[(4 * i + k) * 8 + l];
        for (int i = 4 * g; i < vs; ++i) p[0] += x[i * 8 + l];
        p[0] += p[1]; p[0] += p[2]; p[0] += p[3];
        acc += p[0];
    }
    return acc;
}
float oracle_aten_sum_f32(const float *x, int n) { return aten_sum_f32(x, n); }

void oracle_sample_pdf_2(long N, int nb, int ns, const float *bins, const float *weights,
                         const float *u_in, float *samples, int64_t *inds_out)
{
    float *ulin = (float *)malloc(sizeof(float) * ns);
    float step = (ns > 1) ? (1.0f / (float)(ns - 1)) : 0.0f;
    for (int i = 0; i < ns; ++i) ulin[i] = aten_linspace01(i, ns, step);
#pragma omp parallel for
    for (long r = 0; r < N; ++r) {
        float cdf[512];
        const float *w = weights + r * (nb - 1);
        float wp[512];
        for (int i = 0; i < nb - 1; ++i) wp[i] = w[i] + 1e-5f;                      /* :459 */
        const float sum = aten_sum_f32(wp, nb - 1);                                  /* torch.sum(weights, dim=-1), :460 */
        cdf[0] = 0.0f;
        double c = 0.0;                                                              /* torch.cumsum(pdf, dim=-1), :461: ATen CPU accumulates */
        for (int i = 0; i < nb - 1; ++i) { c += (double)(wp[i] / sum); cdf[i + 1] = (float)c; }   /* float in double and rounds every prefix */
        const float *b = bins + r * nb;
        for (int j = 0; j < ns; ++j) {
            float u = u_in ? u_in[r * ns + j] : ulin[j];
            int lo = 0, hi = nb;           /* first index with cdf[idx] > u */
            while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
            int ind = lo;
            int below = ind - 1 > 0 ? ind - 1 : 0;
            int above = ind < nb - 1 ? ind : nb - 1;
            float denom = cdf[above] - cdf[below];
            if (denom < 1e-5f) denom = 1.0f;
            float t = (u - cdf[below]) / denom;
            samples[r * ns + j] = b[below] + t * (b[above] - b[below]);
            if (inds_out) inds_out[r * ns + j] = ind;
        }
    }
    free(ulin);
}

static int cmp_float(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* z_vals_mid, sample_pdf, cat, sort: train_utils.py:157-166. */
void oracle_resample(long N, int S, int nf, const float *z, const float *weights, const float *u,
                     float *z_samples, float *z_out, int64_t *inds_out)
{
    float *mids = (float *)malloc(sizeof(float) * N * (S - 1));
    float *wmid = (float *)malloc(sizeof(float) * N * (S - 2));
    for (long r = 0; r < N; ++r) {
        for (int i = 0; i < S - 1; ++i) mids[r * (S - 1) + i] = 0.5f * (z[r * S + i + 1] + z[r * S + i]);
        for (int i = 0; i < S - 2; ++i) wmid[r * (S - 2) + i] = weights[r * S + 1 + i];
    }
    oracle_sample_pdf_2(N, S - 1, nf, mids, wmid, u, z_samples, inds_out);
#pragma omp parallel for
    for (long r = 0; r < N; ++r) {
        float *o = z_out + r * (S + nf);
        memcpy(o, z + r * S, sizeof(float) * S);
        memcpy(o + S, z_samples + r * nf, sizeof(float) * nf);
        qsort(o, S + nf, sizeof(float), cmp_float);
    }
    free(mids); free(wmid);
}

/* ------------------------------------------------------------------------------------------
 * predict_and_render_radiance for one ray chunk: train_utils.py:72-206.
 * rays (N,8+): [ro3, rd3, near, far, ...] with row stride rs.  Explicit random tensors
 * (NULL = that draw is skipped: perturb off / noise std 0), drawn in the reference's order
 * (SURVEY appendix A.9): t_rand (N,Sc), noise_c (N,Sc), u (N,nf), noise_f (N,Sc+nf); noise
 * tensors are already scaled by radiance_field_noise_std.
 * Outputs: rgb_c (N,15), disp_c, acc_c, rgb_f (N,15), disp_f, acc_f, w_bg (N)=weights_f[:,-1],
 * depth_f; optional z_fine (N,Sc+nf), weights_c (N,Sc).
 * ---------------------------------------------------------------------------------------- */
void oracle_render_rays(const float *flat, long N, const float *rays, int rs, int Sc, int nf,
                        int lindisp, int white_bkgd, const float *driving76, const float *pose36,
                        const float *bg, const float *t_rand, const float *noise_c, const float *u,
                        const float *noise_f, float *rgb_c, float *disp_c, float *acc_c,
                        float *rgb_f, float *disp_f, float *acc_f, float *w_bg, float *depth_f,
                        float *z_fine_out, float *weights_c_out)
{
    int Sf = Sc + nf;
    float *ro = (float *)malloc(sizeof(float) * N * 3), *rd = (float *)malloc(sizeof(float) * N * 3);
    float *nr = (float *)malloc(sizeof(float) * N), *fr = (float *)malloc(sizeof(float) * N);
    for (long r = 0; r < N; ++r) {
        for (int i = 0; i < 3; ++i) { ro[r * 3 + i] = rays[r * rs + i]; rd[r * 3 + i] = rays[r * rs + 3 + i]; }
        nr[r] = rays[r * rs + 6]; fr[r] = rays[r * rs + 7];
    }
    float *z = (float *)malloc(sizeof(float) * N * Sf);
    float *x6 = (float *)malloc(sizeof(float) * N * Sf * 6);
    float *raw = (float *)malloc(sizeof(float) * N * Sf * D_RAW);
    float *wts = (float *)malloc(sizeof(float) * N * Sf);
    float *dep = (float *)malloc(sizeof(float) * N);
    oracle_stratified_depths(N, Sc, nr, fr, lindisp, t_rand, z);
    oracle_make_points(N, Sc, ro, rd, z, x6);
    oracle_field_forward(flat, 0, N * Sc, x6, 6, driving76, pose36, raw, 0, 0, 0);
    oracle_composite(N, Sc, raw, z, rd, noise_c, bg, white_bkgd, rgb_c, disp_c, acc_c, wts, dep);
    if (weights_c_out) memcpy(weights_c_out, wts, sizeof(float) * N * Sc);
    if (nf > 0) {
        float *zs = (float *)malloc(sizeof(float) * N * nf);
        float *zf = (float *)malloc(sizeof(float) * N * Sf);
        oracle_resample(N, Sc, nf, z, wts, u, zs, zf, 0);
        oracle_make_points(N, Sf, ro, rd, zf, x6);
        oracle_field_forward(flat, 1, N * Sf, x6, 6, driving76, pose36, raw, 0, 0, 0);
        oracle_composite(N, Sf, raw, zf, rd, noise_f, bg, white_bkgd, rgb_f, disp_f, acc_f, wts, depth_f);
        for (long r = 0; r < N; ++r) w_bg[r] = wts[r * Sf + Sf - 1];
        if (z_fine_out) memcpy(z_fine_out, zf, sizeof(float) * N * Sf);
        free(zs); free(zf);
    } else {
        for (long r = 0; r < N; ++r) w_bg[r] = wts[r * Sc + Sc - 1];
    }
    free(ro); free(rd); free(nr); free(fr); free(z); free(x6); free(raw); free(wts); free(dep);
}
