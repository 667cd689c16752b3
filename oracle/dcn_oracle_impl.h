/*
 * oracle/dcn_oracle_impl.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the arithmetic of the reference's CUDA-only extension
 * `dcn_deform_conv_cuda` (the reference has no CPU path at all: SURVEY.md section 0,
 * fact 4).  Included twice by dcn_oracle.c with
 *     T   = float  / double      (tensor scalar type)
 *     FN(x) = x##_f32 / x##_f64  (symbol suffix)
 * Every function cites the reference file:line (relative to /root/reference/) whose
 * behaviour it restates.  Written from the arithmetic, in the reference's algorithmic
 * form (im2col column buffer + per-group contraction), not copied from it.
 *
 * Positions / bilinear weights are computed in T exactly in the reference's operation
 * order; contractions (the cuBLAS addmm of the reference, whose summation order is
 * unspecified) accumulate in double.
 *
 * Layouts (all contiguous):
 *   x      [N, C, H, W]
 *   offset [N, DG*2*kH*kW, Ho, Wo]       channel 2k = dy, 2k+1 = dx, k = i*kW + j
 *   mask   [N, DG*kH*kW,  Ho, Wo]        (modulated only)
 *   weight [Co, C/G, kH, kW]
 *   cols   [C*kH*kW, N, Ho, Wo]          (row r = c*kH*kW + k)
 */

/* lib/models/external/src/dcn_deform_conv_cuda_kernel.cu:83-114 (deformable_im2col_bilinear)
 * and :467-497 (dmcn_im2col_bilinear; identical arithmetic). */
static T FN(bilinear)(const T *plane, int H, int W, T h, T w)
{
    int h_low = (int)floor((double)h);
    int w_low = (int)floor((double)w);
    int h_high = h_low + 1;
    int w_high = w_low + 1;
    T lh = h - (T)h_low;
    T lw = w - (T)w_low;
    T hh = (T)1 - lh, hw = (T)1 - lw;
    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0)           v1 = plane[h_low * W + w_low];
    if (h_low >= 0 && w_high <= W - 1)      v2 = plane[h_low * W + w_high];
    if (h_high <= H - 1 && w_low >= 0)      v3 = plane[h_high * W + w_low];
    if (h_high <= H - 1 && w_high <= W - 1) v4 = plane[h_high * W + w_high];
    T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    T a = w1 * v1;
    T b = w2 * v2;
    T c = w3 * v3;
    T d = w4 * v4;
    return ((a + b) + c) + d;
}

/* _kernel.cu:116-142 (get_gradient_weight) / :499-523 (dmcn_get_gradient_weight). */
static T FN(grad_weight_corner)(T ah, T aw, int h, int w, int H, int W)
{
    if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;
    int hl = (int)floor((double)ah), wl = (int)floor((double)aw);
    int hh = hl + 1, wh = wl + 1;
    T wt = 0;
    if (h == hl && w == wl) wt = ((T)(h + 1) - ah) * ((T)(w + 1) - aw);
    if (h == hl && w == wh) wt = ((T)(h + 1) - ah) * (aw + (T)1 - (T)w);
    if (h == hh && w == wl) wt = (ah + (T)1 - (T)h) * ((T)(w + 1) - aw);
    if (h == hh && w == wh) wt = (ah + (T)1 - (T)h) * (aw + (T)1 - (T)w);
    return wt;
}

/* _kernel.cu:144-187 (get_coordinate_weight) / :525-567 (dmcn_get_coordinate_weight). */
static T FN(coord_weight)(T ah, T aw, int H, int W, const T *plane, int dir)
{
    if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;
    int hl = (int)floor((double)ah), wl = (int)floor((double)aw);
    int hh = hl + 1, wh = wl + 1;
    T wt = 0;
    if (dir == 0) {
        if (hl >= 0 && wl >= 0)         wt += (T)-1 * ((T)(wl + 1) - aw) * plane[hl * W + wl];
        if (hl >= 0 && wh <= W - 1)     wt += (T)-1 * (aw - (T)wl) * plane[hl * W + wh];
        if (hh <= H - 1 && wl >= 0)     wt += ((T)(wl + 1) - aw) * plane[hh * W + wl];
        if (hh <= H - 1 && wh <= W - 1) wt += (aw - (T)wl) * plane[hh * W + wh];
    } else {
        if (hl >= 0 && wl >= 0)         wt += (T)-1 * ((T)(hl + 1) - ah) * plane[hl * W + wl];
        if (hl >= 0 && wh <= W - 1)     wt += ((T)(hl + 1) - ah) * plane[hl * W + wh];
        if (hh <= H - 1 && wl >= 0)     wt += (T)-1 * (ah - (T)hl) * plane[hh * W + wl];
        if (hh <= H - 1 && wh <= W - 1) wt += (ah - (T)hl) * plane[hh * W + wh];
    }
    return wt;
}

/* _kernel.cu:189-242 (deformable_im2col_gpu_kernel) and, with mask != NULL,
 * :569-632 (modulated_deformable_im2col_gpu_kernel): cols[(c*K+k), n, ho, wo]. */
static void FN(im2col)(const T *x, const T *offset, const T *mask, T *cols,
                       int N, int C, int H, int W, int kH, int kW,
                       int sH, int sW, int pH, int pW, int dH, int dW,
                       int DG, int Ho, int Wo)
{
    const int K = kH * kW;
    const int cpdg = C / DG;
    const long plane_o = (long)Ho * Wo;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int n = 0; n < N; ++n) {
            const int dg = c / cpdg;
            const T *xp = x + ((long)n * C + c) * H * W;
            const T *op = offset + ((long)n * DG + dg) * 2 * K * plane_o;
            const T *mp = mask ? mask + ((long)n * DG + dg) * K * plane_o : 0;
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    const int h_in = ho * sH - pH;
                    const int w_in = wo * sW - pW;
                    for (int i = 0; i < kH; ++i)
                        for (int j = 0; j < kW; ++j) {
                            const int k = i * kW + j;
                            const T off_h = op[(long)(2 * k) * plane_o + ho * Wo + wo];
                            const T off_w = op[(long)(2 * k + 1) * plane_o + ho * Wo + wo];
                            const T h_im = (T)(h_in + i * dH) + off_h;
                            const T w_im = (T)(w_in + j * dW) + off_w;
                            T val = 0;
                            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                                val = FN(bilinear)(xp, H, W, h_im, w_im);
                            if (mp) val = val * mp[(long)k * plane_o + ho * Wo + wo];
                            cols[(((long)c * K + k) * N + n) * plane_o + ho * Wo + wo] = val;
                        }
                }
        }
}

/* src/dcn_deform_conv_cuda.cpp:151-258 (deform_conv_forward_cuda): im2col, then for every
 * group g  out[g] += W[g] (Cog x Cg*K) . cols[g] (Cg*K x N*Ho*Wo)  (:229-234).
 * With mask/bias: cpp:486-564 (modulated_deform_conv_cuda_forward, bias add :561-563).
 * The result does not depend on im2col_step, so the whole batch is one step here. */
int FN(dcn_oracle_forward)(const T *x, const T *offset, const T *mask, const T *weight,
                           const T *bias, T *out, int N, int C, int H, int W, int Co,
                           int kH, int kW, int sH, int sW, int pH, int pW, int dH, int dW,
                           int G, int DG)
{
    const int Ho = (H + 2 * pH - (dH * (kH - 1) + 1)) / sH + 1; /* cpp:187-190 */
    const int Wo = (W + 2 * pW - (dW * (kW - 1) + 1)) / sW + 1;
    if (Ho < 1 || Wo < 1 || C % G || Co % G || C % DG) return -1;
    const int K = kH * kW, Cg = C / G, Cog = Co / G;
    const long P = (long)Ho * Wo;
    T *cols = (T *)malloc(sizeof(T) * (size_t)C * K * N * P);
    if (!cols) return -2;
    FN(im2col)(x, offset, mask, cols, N, C, H, W, kH, kW, sH, sW, pH, pW, dH, dW, DG, Ho, Wo);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co) {
            const int g = co / Cog;
            for (long p = 0; p < P; ++p) {
                double acc = 0.0;
                for (int r = 0; r < Cg * K; ++r)
                    acc += (double)weight[(long)co * Cg * K + r] *
                           (double)cols[(((long)g * Cg * K + r) * N + n) * P + p];
                if (bias) acc += (double)bias[co];
                out[((long)n * Co + co) * P + p] = (T)acc;
            }
        }
    free(cols);
    return 0;
}

/* cpp:329-332 (and :617-620): gcols[g] = W[g]^T . gO[g]   ->  gcols[(c*K+k), n, p]. */
static void FN(grad_cols)(const T *weight, const T *gout, T *gcols, int N, int C, int Co,
                          int K, int G, long P)
{
    const int Cg = C / G, Cog = Co / G;
#pragma omp parallel for collapse(2) schedule(static)
    for (int c = 0; c < C; ++c)
        for (int k = 0; k < K; ++k) {
            const int g = c / Cg, cl = c % Cg;
            for (int n = 0; n < N; ++n)
                for (long p = 0; p < P; ++p) {
                    double acc = 0.0;
                    for (int m = 0; m < Cog; ++m)
                        acc += (double)weight[((long)(g * Cog + m) * Cg + cl) * K + k] *
                               (double)gout[((long)n * Co + g * Cog + m) * P + p];
                    gcols[(((long)c * K + k) * N + n) * P + p] = (T)acc;
                }
        }
}

/* cpp:260-371 (deform_conv_backward_input_cuda) -> grad_input, grad_offset; with mask:
 * cpp:566-679 (modulated backward: grad_input, grad_offset, grad_mask).
 *   grad_offset: _kernel.cu:372-435 (deformable_col2im_coord_gpu_kernel), :694-766 (modulated)
 *   grad_input : _kernel.cu:278-334 (deformable_col2im_gpu_kernel),       :634-692 (modulated)
 * The reference's float atomicAdd order is unspecified; here the scatter is sequential
 * per (n,c) plane and accumulated in double.
 * NOTE the modulated col2im launcher passes pad_h twice (_kernel.cu:821); unreachable from
 * the Python API (one int padding for both axes), so pad_w is honoured here. */
int FN(dcn_oracle_backward_input)(const T *x, const T *offset, const T *mask, const T *weight,
                                  const T *gout, T *gx, T *goffset, T *gmask,
                                  int N, int C, int H, int W, int Co, int kH, int kW,
                                  int sH, int sW, int pH, int pW, int dH, int dW, int G, int DG)
{
    const int Ho = (H + 2 * pH - (dH * (kH - 1) + 1)) / sH + 1;
    const int Wo = (W + 2 * pW - (dW * (kW - 1) + 1)) / sW + 1;
    if (Ho < 1 || Wo < 1 || C % G || Co % G || C % DG) return -1;
    const int K = kH * kW, cpdg = C / DG;
    const long P = (long)Ho * Wo;
    T *gcols = (T *)malloc(sizeof(T) * (size_t)C * K * N * P);
    if (!gcols) return -2;
    FN(grad_cols)(weight, gout, gcols, N, C, Co, K, G, P);

    /* grad wrt offset (and mask): one (n, dg, k) triple per iteration, loop over the
     * channels of the deformable group (_kernel.cu:405-431). */
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int dgk = 0; dgk < DG * K; ++dgk) {
            const int dg = dgk / K, k = dgk % K, i = k / kW, j = k % kW;
            const T *op = offset + ((long)n * DG + dg) * 2 * K * P;
            const T *mp = mask ? mask + ((long)n * DG + dg) * K * P : 0;
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    const long p = (long)ho * Wo + wo;
                    const T off_h = op[(long)(2 * k) * P + p];
                    const T off_w = op[(long)(2 * k + 1) * P + p];
                    T inv_h = (T)(ho * sH - pH + i * dH) + off_h;
                    T inv_w = (T)(wo * sW - pW + j * dW) + off_w;
                    int inside = 1;
                    if (inv_h <= -1 || inv_w <= -1 || inv_h >= H || inv_w >= W) {
                        inv_h = inv_w = -2;
                        inside = 0;
                    }
                    const T m = mp ? mp[(long)k * P + p] : (T)1;
                    double vh = 0.0, vw = 0.0, mv = 0.0;
                    for (int cl = 0; cl < cpdg; ++cl) {
                        const int c = dg * cpdg + cl;
                        const T *xp = x + ((long)n * C + c) * H * W;
                        const T gc = gcols[(((long)c * K + k) * N + n) * P + p];
                        if (inside && mp) mv += (double)gc * (double)FN(bilinear)(xp, H, W, inv_h, inv_w);
                        const T wh = FN(coord_weight)(inv_h, inv_w, H, W, xp, 0);
                        const T ww = FN(coord_weight)(inv_h, inv_w, H, W, xp, 1);
                        vh += (double)(wh * gc * m);
                        vw += (double)(ww * gc * m);
                    }
                    goffset[(((long)n * DG + dg) * 2 * K + 2 * k) * P + p] = (T)vh;
                    goffset[(((long)n * DG + dg) * 2 * K + 2 * k + 1) * P + p] = (T)vw;
                    if (gmask) gmask[(((long)n * DG + dg) * K + k) * P + p] = (T)mv;
                }
        }

    /* grad wrt input: scatter every column gradient to the <=4 bilinear corners, found by
     * the reference's 5x5 window scan around (int)position (_kernel.cu:315-332). */
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const int dg = c / cpdg;
            const T *op = offset + ((long)n * DG + dg) * 2 * K * P;
            const T *mp = mask ? mask + ((long)n * DG + dg) * K * P : 0;
            double *acc = (double *)calloc((size_t)H * W, sizeof(double));
            for (int k = 0; k < K; ++k) {
                const int i = k / kW, j = k % kW;
                for (int ho = 0; ho < Ho; ++ho)
                    for (int wo = 0; wo < Wo; ++wo) {
                        const long p = (long)ho * Wo + wo;
                        const T off_h = op[(long)(2 * k) * P + p];
                        const T off_w = op[(long)(2 * k + 1) * P + p];
                        const T ih = (T)(ho * sH - pH + i * dH) + off_h;
                        const T iw = (T)(wo * sW - pW + j * dW) + off_w;
                        T top = gcols[(((long)c * K + k) * N + n) * P + p];
                        if (mp) top = top * mp[(long)k * P + p];
                        const int ch = (int)ih, cw = (int)iw;
                        for (int dy = -2; dy <= 2; ++dy)
                            for (int dx = -2; dx <= 2; ++dx) {
                                const int yy = ch + dy, xx = cw + dx;
                                if (yy >= 0 && yy < H && xx >= 0 && xx < W &&
                                    fabs((double)(ih - (T)yy)) < 1 && fabs((double)(iw - (T)xx)) < 1) {
                                    const T wt = FN(grad_weight_corner)(ih, iw, yy, xx, H, W);
                                    acc[yy * W + xx] += (double)(wt * top);
                                }
                            }
                    }
            }
            T *gp = gx + ((long)n * C + c) * H * W;
            for (int q = 0; q < H * W; ++q) gp[q] = (T)acc[q];
            free(acc);
        }
    free(gcols);
    return 0;
}

/* cpp:373-484 (deform_conv_backward_parameters_cuda): gradW[g] += scale * gO[g] . cols[g]^T
 * (:456-462); modulated: cpp:639-668 incl. grad_bias = sum gO (:657-663).
 * grad_weight / grad_bias are ACCUMULATED INTO (caller zero-fills), as in the reference. */
int FN(dcn_oracle_backward_params)(const T *x, const T *offset, const T *mask, const T *gout,
                                   T *gweight, T *gbias, int N, int C, int H, int W, int Co,
                                   int kH, int kW, int sH, int sW, int pH, int pW, int dH, int dW,
                                   int G, int DG, double scale)
{
    const int Ho = (H + 2 * pH - (dH * (kH - 1) + 1)) / sH + 1;
    const int Wo = (W + 2 * pW - (dW * (kW - 1) + 1)) / sW + 1;
    if (Ho < 1 || Wo < 1 || C % G || Co % G || C % DG) return -1;
    const int K = kH * kW, Cg = C / G, Cog = Co / G;
    const long P = (long)Ho * Wo;
    T *cols = (T *)malloc(sizeof(T) * (size_t)C * K * N * P);
    if (!cols) return -2;
    FN(im2col)(x, offset, mask, cols, N, C, H, W, kH, kW, sH, sW, pH, pW, dH, dW, DG, Ho, Wo);
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Co; ++co) {
        const int g = co / Cog;
        for (int r = 0; r < Cg * K; ++r) {
            double acc = 0.0;
            for (int n = 0; n < N; ++n)
                for (long p = 0; p < P; ++p)
                    acc += (double)gout[((long)n * Co + co) * P + p] *
                           (double)cols[(((long)g * Cg * K + r) * N + n) * P + p];
            gweight[(long)co * Cg * K + r] = (T)((double)gweight[(long)co * Cg * K + r] + scale * acc);
        }
        if (gbias) {
            double acc = 0.0;
            for (int n = 0; n < N; ++n)
                for (long p = 0; p < P; ++p) acc += (double)gout[((long)n * Co + co) * P + p];
            gbias[co] = (T)((double)gbias[co] + acc);
        }
    }
    free(cols);
    return 0;
}

/* Exposed for tests / the CPU baseline: the column buffer itself. */
int FN(dcn_oracle_im2col)(const T *x, const T *offset, const T *mask, T *cols,
                          int N, int C, int H, int W, int kH, int kW,
                          int sH, int sW, int pH, int pW, int dH, int dW, int DG)
{
    const int Ho = (H + 2 * pH - (dH * (kH - 1) + 1)) / sH + 1;
    const int Wo = (W + 2 * pW - (dW * (kW - 1) + 1)) / sW + 1;
    if (Ho < 1 || Wo < 1 || C % DG) return -1;
    FN(im2col)(x, offset, mask, cols, N, C, H, W, kH, kW, sH, sW, pH, pW, dH, dW, DG, Ho, Wo);
    return 0;
}
