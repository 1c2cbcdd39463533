/*
 * maskrcnn_predict_stream.c — the evaluate LOOP (Sources/maskrcnn/EvaluateCommand.swift:165-179: one prediction after the
 * other over a list of images, the hand-over of each image inside its time) as a plain-C99 host over the PIPELINED host entry
 * of include/maskrcnn_hip.h: while batch i computes, the images of batch i + 1 cross PCIe on the handle's copy stream.
 *
 *     submit(batch 0)
 *     for i = 0 .. n-1:   if (i + 1 < n) submit(batch i + 1);   collect(results of batch i)
 *
 *   cc -std=c99 -Iinclude examples/maskrcnn_predict_stream.c -Lmask-rcnn-coreml_amd -lmaskrcnn_hip \
 *      -Wl,-rpath,$PWD/mask-rcnn-coreml_amd -Wl,-rpath-link,/opt/rocm/lib -o maskrcnn_predict_stream
 *   ./maskrcnn_predict_stream <artefact dir> <images.rgb> <n batches> <batch> [f32|f16|f32s|f32x3] [calibrate]
 *
 * <images.rgb>: n*batch raw RGB8 images of the model's input size, back to back.  With "calibrate" the split modes get their
 * per-tensor power-of-two exponents from one calibration predict on the first batch (mrcnn_model_calibrate_split) before the
 * loop.  Prints, per image, its global index, the number of detections with score > 0.7 and a checksum of its records; the
 * last line is the loop's wall clock.  Results are bit-identical to mrcnn_maskrcnn_predict's (tests/test_c_host.py).
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "maskrcnn_hip.h"

#define CHECK(call)                                                                          \
    do {                                                                                     \
        int st_ = (call);                                                                    \
        if (st_ != MRCNN_OK) {                                                               \
            fprintf(stderr, "%s failed (%d): %s\n", #call, st_, mrcnn_last_error());         \
            return st_;                                                                      \
        }                                                                                    \
    } while (0)

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s <artefact dir> <images.rgb> <n batches> <batch> [f32|f16|f32s|f32x3] [calibrate]\n", argv[0]);
        return 64;
    }
    const char* dir = argv[1];
    const int n_batches = atoi(argv[3]), B = atoi(argv[4]);
    const int dtype = argc <= 5 ? MRCNN_F32 : strcmp(argv[5], "f16") == 0 ? MRCNN_F16 : strcmp(argv[5], "f32s") == 0 ? MRCNN_F32S : strcmp(argv[5], "f32x3") == 0 ? MRCNN_F32X3 : MRCNN_F32;
    const int calibrate = argc > 6 && strcmp(argv[6], "calibrate") == 0;
    if (n_batches < 1 || B < 1) { fprintf(stderr, "bad batch arguments\n"); return 64; }
    char path[4][4096];
    snprintf(path[0], sizeof path[0], "%s/anchors.bin", dir);
    snprintf(path[1], sizeof path[1], "%s/Classifier.mrcw", dir);
    snprintf(path[2], sizeof path[2], "%s/Mask.mrcw", dir);
    snprintf(path[3], sizeof path[3], "%s/MaskRCNN.mrcw", dir);
    CHECK(mrcnn_config_set_anchors_path(path[0]));
    CHECK(mrcnn_config_set_classifier_path(path[1]));
    CHECK(mrcnn_config_set_mask_path(path[2]));
    mrcnn_model* model = NULL;
    CHECK(mrcnn_model_load(MRCNN_MODEL_MASKRCNN, path[3], B, dtype, &model));
    int64_t H = 0, W = 0, D = 0, S = 0;
    CHECK(mrcnn_model_get_int(model, "image_height", &H));
    CHECK(mrcnn_model_get_int(model, "image_width", &W));
    CHECK(mrcnn_model_get_int(model, "max_detections", &D));
    CHECK(mrcnn_model_get_int(model, "mask_size", &S));

    const size_t img = (size_t)H * (size_t)W * 3u, n_img = (size_t)n_batches * (size_t)B;
    uint8_t* src = (uint8_t*)malloc(img * n_img);
    float* det = (float*)malloc(sizeof(float) * (size_t)B * (size_t)D * 6u);
    float* masks = (float*)malloc(sizeof(float) * (size_t)B * (size_t)D * (size_t)S * (size_t)S);
    if (!src || !det || !masks) { fprintf(stderr, "out of memory\n"); return 70; }
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(src, 1, img * n_img, f) != img * n_img) { fprintf(stderr, "%s: cannot read %zu bytes\n", argv[2], img * n_img); return 66; }
    fclose(f);

    if (calibrate && (dtype == MRCNN_F32S || dtype == MRCNN_F32X3)) {
        int64_t lo = 0, hi = 0;
        CHECK(mrcnn_model_calibrate_split(model, src, B, (int)H, (int)W, MRCNN_HOST, 1));
        CHECK(mrcnn_model_get_int(model, "split_min_exponent", &lo));
        CHECK(mrcnn_model_get_int(model, "split_max_exponent", &hi));
        printf("split exponents %lld .. %lld\n", (long long)lo, (long long)hi);
    }

    const double t0 = now_s();
    CHECK(mrcnn_maskrcnn_submit(model, src, B, (int)H, (int)W));
    for (int i = 0; i < n_batches; ++i) {
        if (i + 1 < n_batches) CHECK(mrcnn_maskrcnn_submit(model, src + (size_t)(i + 1) * (size_t)B * img, B, (int)H, (int)W));
        int got = 0;
        CHECK(mrcnn_maskrcnn_collect(model, det, masks, &got));
        for (int b = 0; b < got; ++b) {
            const float* d = det + (size_t)b * (size_t)D * 6u;
            const float* m = masks + (size_t)b * (size_t)D * (size_t)S * (size_t)S;
            int64_t n = 0;
            double sum = 0.0;
            for (int64_t r = 0; r < D; ++r) {
                if ((double)d[r * 6 + 5] > 0.7) ++n;                                   /* Detection.swift:38 */
                for (int k = 0; k < 6; ++k) sum += (double)d[r * 6 + k];
            }
            for (int64_t k = 0; k < D * S * S; ++k) sum += (double)m[k];
            printf("image %d detections %lld checksum %.17g\n", i * B + b, (long long)n, sum);
        }
    }
    const double t1 = now_s();
    printf("batches %d batch %d seconds %.6f\n", n_batches, B, t1 - t0);
    mrcnn_model_destroy(model);
    free(src); free(det); free(masks);
    return 0;
}
