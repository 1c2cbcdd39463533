/*
 * maskrcnn_predict.c — a host in plain C99 over include/maskrcnn_hip.h, the way the Swift host would
 * sit on the same C ABI: no Python, no torch, no HIP headers.
 *
 * Mirrors `maskrcnn evaluate` (Sources/maskrcnn/EvaluateCommand.swift:134-200) for one raw image:
 *   config singleton (:144-150) → model load outside the timed region (:146-156) → `.scaleFit`
 *   letterbox (:157) → prediction, wall-clock around exactly that (:167,179) → detections with
 *   score > 0.7 (Detection.swift:38) printed one per line.
 *
 *   cc -std=c99 -Iinclude examples/maskrcnn_predict.c -Lmask-rcnn-coreml_amd -lmaskrcnn_hip \
 *      -Wl,-rpath,$PWD/mask-rcnn-coreml_amd -Wl,-rpath-link,/opt/rocm/lib -o maskrcnn_predict
 *   ./maskrcnn_predict <artefact dir> <image.rgb> <height> <width> [default|f32|f16|f32s|f32x3]
 *
 * <artefact dir> holds MaskRCNN.mrcw, Classifier.mrcw, Mask.mrcw, anchors.bin; <image.rgb> is raw
 * interleaved RGB8 of height×width.  Exit status 0 on success; on failure the mrcnn_last_error()
 * text goes to stderr and the status code is the exit status.
 */
#define _POSIX_C_SOURCE 200809L   /* clock_gettime under -std=c99 */
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
        fprintf(stderr, "usage: %s <artefact dir> <image.rgb> <height> <width> [default|f32|f16|f32s|f32x3]\n", argv[0]);
        return 64;
    }
    const char* dir = argv[1];
    const int h = atoi(argv[3]), w = atoi(argv[4]);
    /* no precision named (as `MaskRCNN()` in ViewController.swift:37): MRCNN_DEFAULT = the mode the artefact is prepared for */
    const int dtype = argc <= 5 || strcmp(argv[5], "default") == 0 ? MRCNN_DEFAULT : strcmp(argv[5], "f16") == 0 ? MRCNN_F16 : strcmp(argv[5], "f32s") == 0 ? MRCNN_F32S
                    : strcmp(argv[5], "f32x3") == 0 ? MRCNN_F32X3 : MRCNN_F32;
    char path[4][4096];
    snprintf(path[0], sizeof path[0], "%s/anchors.bin", dir);
    snprintf(path[1], sizeof path[1], "%s/Classifier.mrcw", dir);
    snprintf(path[2], sizeof path[2], "%s/Mask.mrcw", dir);
    snprintf(path[3], sizeof path[3], "%s/MaskRCNN.mrcw", dir);

    /* MaskRCNNConfig.defaultConfig must be set before the model is created (AppDelegate.swift:18-20) */
    CHECK(mrcnn_config_set_anchors_path(path[0]));
    CHECK(mrcnn_config_set_classifier_path(path[1]));
    CHECK(mrcnn_config_set_mask_path(path[2]));
    mrcnn_model* model = NULL;
    CHECK(mrcnn_model_load(MRCNN_MODEL_MASKRCNN, path[3], 1, dtype, &model));

    int64_t H = 0, W = 0, max_det = 0;
    CHECK(mrcnn_model_get_int(model, "image_height", &H));
    CHECK(mrcnn_model_get_int(model, "image_width", &W));
    CHECK(mrcnn_model_get_int(model, "max_detections", &max_det));
    const int mask_size = 28;

    const size_t n_src = (size_t)h * (size_t)w * 3u;
    uint8_t* src = (uint8_t*)malloc(n_src);
    uint8_t* canvas = (uint8_t*)malloc((size_t)H * (size_t)W * 3u);
    float* det = (float*)malloc(sizeof(float) * (size_t)max_det * 6u);
    float* masks = (float*)malloc(sizeof(float) * (size_t)max_det * mask_size * mask_size);
    mrcnn_detection* recs = (mrcnn_detection*)malloc(sizeof(mrcnn_detection) * (size_t)max_det);
    if (!src || !canvas || !det || !masks || !recs) { fprintf(stderr, "out of memory\n"); return 70; }
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(src, 1, n_src, f) != n_src) { fprintf(stderr, "%s: cannot read %zu bytes\n", argv[2], n_src); return 66; }
    fclose(f);

    const double t0 = now_s();
    CHECK(mrcnn_letterbox_rgb(src, h, w, MRCNN_HOST, canvas, (int)H, (int)W));
    CHECK(mrcnn_maskrcnn_predict(model, canvas, 1, (int)H, (int)W, MRCNN_HOST, det, masks));
    const double t1 = now_s();

    int64_t n = 0;
    CHECK(mrcnn_detections_decode(det, max_det, 6, recs, max_det, &n));
    printf("seconds %.6f\n", t1 - t0);                     /* EvaluateCommand.swift:193 */
    printf("detections %lld\n", (long long)n);
    for (int64_t i = 0; i < n; ++i) {
        /* row index, class, score, CGRect(x, y, width, height) normalized in the letterboxed frame, mask checksum */
        const float* m = masks + (size_t)recs[i].index * mask_size * mask_size;
        double sum = 0.0;
        for (int k = 0; k < mask_size * mask_size; ++k) sum += (double)m[k];
        printf("%lld %lld %.17g %.17g %.17g %.17g %.17g %.17g\n", (long long)recs[i].index, (long long)recs[i].class_id,
               recs[i].score, recs[i].x, recs[i].y, recs[i].w, recs[i].h, sum);
    }
    mrcnn_model_destroy(model);
    free(src); free(canvas); free(det); free(masks); free(recs);
    return 0;
}
