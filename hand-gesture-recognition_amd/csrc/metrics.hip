// metrics.hip -- on-device step metrics and input-side rendering.
//
// The reference computes these on the host every step, forcing a device->host sync
// (train.py:66-73): heat-map argmax decode (libs/utils.py:4-32), PCK-style pose accuracy
// (libs/metrics.py:6-62), argmax + macro-F1 of the class logits (train.py:66-69, sklearn) -- and
// renders the Gaussian heat-map targets / normalises the image in dataloader workers
// (libs/load.py:46-50,148-206).  Here they are small HIP reductions so a training loop never
// leaves the device.  Integer counters use integer atomics (exact, order independent).
#include "mtn_types.h"
#include "mtn_kernels.h"

namespace {

// one wave per (b, j) map: first index of the maximum (numpy argmax), its value; preds masked by max > 0
__global__ __launch_bounds__(256) void hm_argmax_kernel(const float* __restrict__ hm, float* __restrict__ preds,
                                                        float* __restrict__ maxvals, int maps, int H, int W) {
    const int lane = threadIdx.x & 63;
    const int map = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (map >= maps) return;                       // whole wave leaves together
    const float* p = hm + (long)map * H * W;
    float best = -3.4e38f;
    int bi = 0x7fffffff;
    for (int i = lane; i < H * W; i += 64) {
        const float v = p[i];
        if (v > best) { best = v; bi = i; }        // strict >: the lane keeps its first maximum
    }
    MTN_UNROLL for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        const float keep = best > 0.f ? 1.f : 0.f;
        preds[2 * map] = (float)(bi % W) * keep;
        preds[2 * map + 1] = (float)(bi / W) * keep;
        if (maxvals) maxvals[map] = best;
    }
}

// single block: acc[0] = mean of valid per-joint accuracies, acc[1..J] per joint (-1: no valid
// sample), acc[J+1] = cnt (number of valid joints)      (libs/metrics.py:31-62)
__global__ __launch_bounds__(256) void pose_acc_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       float* __restrict__ acc, int B, int J, int H, int W, float thr) {
    __shared__ float s_acc[256];
    const int j = threadIdx.x;
    float a = -1.f;
    if (j < J) {
        int valid = 0, hit = 0;
        const float nx = (float)H / 10.f, ny = (float)W / 10.f;     // norm = [h, w] / 10 applied to (x, y)
        for (int b = 0; b < B; ++b) {
            const float tx = tgt[((long)b * J + j) * 2], ty = tgt[((long)b * J + j) * 2 + 1];
            if (tx > 1.f && ty > 1.f) {
                const float dx = pred[((long)b * J + j) * 2] / nx - tx / nx;
                const float dy = pred[((long)b * J + j) * 2 + 1] / ny - ty / ny;
                ++valid;
                if (sqrtf(dx * dx + dy * dy) < thr) ++hit;
            }
        }
        a = valid > 0 ? (float)hit / (float)valid : -1.f;
        acc[1 + j] = a;
    }
    s_acc[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float sum = 0.f;
        int cnt = 0;
        for (int i = 0; i < J; ++i)
            if (s_acc[i] >= 0.f) { sum += s_acc[i]; ++cnt; }
        acc[0] = cnt ? sum / (float)cnt : 0.f;
        acc[J + 1] = (float)cnt;
    }
}

// pred_label[b] = argmax_c logits[b][c] (first maximum); cm[pred][label] += 1 (a label outside
// [0, C) is not counted: nothing is written at an index derived from it)
__global__ __launch_bounds__(256) void cls_confusion_kernel(const float* __restrict__ logits, const long long* __restrict__ label,
                                                            long long* __restrict__ pred_label, int* __restrict__ cm, int B,
                                                            int C) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* l = logits + (long)b * C;
    int best = 0;
    for (int c = 1; c < C; ++c)
        if (l[c] > l[best]) best = c;
    if (pred_label) pred_label[b] = best;
    if (cm) {
        const long long y = label[b];
        if (y >= 0 && y < (long long)C) atomicAdd(&cm[best * C + (int)y], 1);
    }
}

// macro F1 over the labels present in predictions or targets (sklearn semantics, train.py:67-69)
__global__ __launch_bounds__(256) void macro_f1_kernel(const int* __restrict__ cm, float* __restrict__ f1, int C) {
    __shared__ float s_f[256];
    __shared__ int s_p[256];
    const int c = threadIdx.x;
    float f = 0.f;
    int present = 0;
    if (c < C) {
        long tp = cm[c * C + c], row = 0, col = 0;
        for (int k = 0; k < C; ++k) { row += cm[c * C + k]; col += cm[k * C + c]; }
        const long fp = row - tp, fn = col - tp, den = 2 * tp + fp + fn;
        present = (row + col) > 0;
        f = den > 0 ? (float)(2.0 * (double)tp / (double)den) : 0.f;
    }
    s_f[c] = present ? f : 0.f;
    s_p[c] = present;
    __syncthreads();
    if (c == 0) {
        float sum = 0.f;
        int n = 0;
        for (int k = 0; k < C; ++k) { sum += s_f[k]; n += s_p[k]; }
        f1[0] = n ? sum / (float)n : 0.f;
    }
}

// Gaussian heat-map targets (libs/load.py:148-206): one block per (b, j) map.
// joints [B][J][2] in image pixels (f32), vis [B][J]; target [B][J][HH][HW], weight [B][J].
__global__ __launch_bounds__(256) void render_targets_kernel(const float* __restrict__ joints, const float* __restrict__ vis,
                                                             float* __restrict__ target, float* __restrict__ weight, int HW_,
                                                             int HH_, double stride_x, double stride_y, int sigma) {
    const int map = blockIdx.x;
    const int tmp = sigma * 3;
    const int mu_x = (int)((double)joints[2 * map] / stride_x + 0.5);        // Python int(): toward zero
    const int mu_y = (int)((double)joints[2 * map + 1] / stride_y + 0.5);
    const int ulx = mu_x - tmp, uly = mu_y - tmp, brx = mu_x + tmp + 1, bry = mu_y + tmp + 1;
    float wgt = vis[map];
    const bool outside = ulx >= HW_ || uly >= HH_ || brx < 0 || bry < 0;
    if (outside) wgt = 0.f;
    float* t = target + (long)map * HH_ * HW_;
    const bool draw = !outside && wgt > 0.5f;
    const float inv = 1.0f / (2.0f * (float)(sigma * sigma));
    for (int i = threadIdx.x; i < HH_ * HW_; i += blockDim.x) {
        const int y = i / HW_, x = i - y * HW_;
        float v = 0.f;
        if (draw && x >= ulx && x < brx && y >= uly && y < bry) {
            const float dx = (float)(x - mu_x), dy = (float)(y - mu_y);
            v = expf(-(dx * dx + dy * dy) * inv);
        }
        t[i] = v;
    }
    if (threadIdx.x == 0) weight[map] = wgt;
}

// uint8 HWC (BGR as cv2.imread delivers it) -> fp32 NCHW, /255 then (x - mean[c]) / std[c]
// (transforms.ToTensor + Normalize with the ImageNet triple, libs/load.py:46-50)
__global__ __launch_bounds__(256) void normalize_image_kernel(const unsigned char* __restrict__ img, float* __restrict__ out,
                                                              long pixels_per_image, long total) {
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / pixels_per_image, p = i - b * pixels_per_image;
        MTN_UNROLL for (int c = 0; c < 3; ++c)
            out[(b * 3 + c) * pixels_per_image + p] = ((float)img[i * 3 + c] * (1.0f / 255.0f) - mean[c]) / stdv[c];
    }
}

}  // namespace

int mtn_hm_argmax_run(const float* hm, float* preds, float* maxvals, int maps, int H, int W, hipStream_t stream) {
    MTN_REQUIRE(maps > 0 && H > 0 && W > 0, "heatmap_argmax: bad sizes");
    MTN_LAUNCH(hm_argmax_kernel, dim3(mtn_cdiv(maps, 4)), dim3(256), 0, stream, hm, preds, maxvals, maps, H, W);
    return mtn_check_launch("heatmap_argmax");
}

int mtn_pose_acc_run(const float* pred, const float* tgt, float* acc, int B, int J, int H, int W, float thr, hipStream_t stream) {
    MTN_REQUIRE(B > 0 && J > 0 && J <= 256, "pose_accuracy: bad sizes");
    MTN_LAUNCH(pose_acc_kernel, dim3(1), dim3(256), 0, stream, pred, tgt, acc, B, J, H, W, thr);
    return mtn_check_launch("pose_accuracy");
}

int mtn_cls_confusion_run(const float* logits, const long long* label, long long* pred_label, int* cm, int B, int C,
                          hipStream_t stream) {
    MTN_REQUIRE(B > 0 && C > 0, "cls_confusion: bad sizes");
    MTN_LAUNCH(cls_confusion_kernel, dim3(mtn_cdiv(B, 256)), dim3(256), 0, stream, logits, label, pred_label, cm, B, C);
    return mtn_check_launch("cls_confusion");
}

int mtn_macro_f1_run(const int* cm, float* f1, int C, hipStream_t stream) {
    MTN_REQUIRE(C > 0 && C <= 256, "macro_f1: C=%d", C);
    MTN_LAUNCH(macro_f1_kernel, dim3(1), dim3(256), 0, stream, cm, f1, C);
    return mtn_check_launch("macro_f1");
}

int mtn_render_targets_run(const float* joints, const float* vis, float* target, float* weight, int maps, int HH, int HW,
                           int img_h, int img_w, int sigma, hipStream_t stream) {
    MTN_REQUIRE(maps > 0 && HH > 0 && HW > 0 && sigma > 0, "render_targets: bad sizes");
    MTN_LAUNCH(render_targets_kernel, dim3(maps), dim3(256), 0, stream, joints, vis, target, weight, HW, HH,
               (double)img_w / (double)HW, (double)img_h / (double)HH, sigma);
    return mtn_check_launch("render_targets");
}

int mtn_normalize_image_run(const unsigned char* img, float* out, int B, int H, int W, hipStream_t stream) {
    MTN_REQUIRE(B > 0 && H > 0 && W > 0, "normalize_image: bad sizes");
    const long ppi = (long)H * W, total = ppi * B;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    MTN_LAUNCH(normalize_image_kernel, dim3((int)blocks), dim3(256), 0, stream, img, out, ppi, total);
    return mtn_check_launch("normalize_image");
}
