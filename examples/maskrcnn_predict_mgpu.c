/*
 * maskrcnn_predict_mgpu.c — the multi-GPU host in plain C99 over include/maskrcnn_hip.h: one PROCESS PER GPU of a node,
 * a batch of images sharded over the ranks, one RCCL all-gather of the per-image records (mrcnn_dist_*, no Python, no
 * torch, no HIP / RCCL headers).  It is what the evaluate loop of Sources/maskrcnn/EvaluateCommand.swift:146-179 becomes
 * when the images of a dataset are predicted by 8 GPUs instead of one.
 *
 *   cc -std=c99 -Iinclude examples/maskrcnn_predict_mgpu.c -Lmask-rcnn-coreml_amd -lmaskrcnn_hip \
 *      -Wl,-rpath,$PWD/mask-rcnn-coreml_amd -Wl,-rpath-link,/opt/rocm/lib -o maskrcnn_predict_mgpu
 *   for r in 0 1 2 3 4 5 6 7; do
 *     RANK=$r WORLD_SIZE=8 HIP_VISIBLE_DEVICES=$r MRCNN_DIST_ID_FILE=/tmp/mrcnn.id \
 *       ./maskrcnn_predict_mgpu <artefact dir> <images.rgb> <batch> [f32|f16|f32s|f32x3] &
 *   done; wait
 *
 * <images.rgb>: `batch` raw interleaved RGB8 images of the model's input size (already letterboxed), identical on every
 * rank.  Each process sees ONE device (HIP_VISIBLE_DEVICES).  The 128-byte rendezvous id travels through a file: rank 0
 * writes <MRCNN_DIST_ID_FILE>.tmp and renames it, the others poll.  Every rank ends up with the whole batch's results;
 * rank 0 prints them (score > 0.7, Detection.swift:38) in the format of examples/maskrcnn_predict.c.
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
            fprintf(stderr, "[rank %d] %s failed (%d): %s\n", rank, #call, st_, mrcnn_last_error()); \
            return st_;                                                                      \
        }                                                                                    \
    } while (0)

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

int main(int argc, char** argv)
{
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1);
    if (argc < 4) {
        fprintf(stderr, "usage: RANK=r WORLD_SIZE=n MRCNN_DIST_ID_FILE=path %s <artefact dir> <images.rgb> <batch> [f32|f16|f32s|f32x3]\n", argv[0]);
        return 64;
    }
    const char* dir = argv[1];
    const int batch = atoi(argv[3]);
    const int dtype = argc <= 4 ? MRCNN_F32X3 : strcmp(argv[4], "f16") == 0 ? MRCNN_F16 : strcmp(argv[4], "f32s") == 0 ? MRCNN_F32S : strcmp(argv[4], "f32") == 0 ? MRCNN_F32 : MRCNN_F32X3;
    const char* id_file = getenv("MRCNN_DIST_ID_FILE");
    if (world > 1 && !id_file) { fprintf(stderr, "MRCNN_DIST_ID_FILE must name a path all ranks can reach\n"); return 64; }

    /* ---- rendezvous: 128 bytes from rank 0 to everyone, through a file --------------------------------------------
     * A file left behind by an earlier run must never be taken for this run's id (ncclCommInitRank with mismatched ids
     * hangs): rank 0 removes the path before it creates the id and again once every rank has joined, and the file carries
     * a launch nonce (MRCNN_DIST_NONCE, set by the launcher to the same fresh value for all ranks) that readers verify. */
    const char* nonce = getenv("MRCNN_DIST_NONCE");
    char want[64];
    memset(want, 0, sizeof want);
    if (nonce) strncpy(want, nonce, sizeof want - 1);
    uint8_t id[128];
    if (rank == 0) {
        if (world > 1) (void)remove(id_file);
        CHECK(mrcnn_dist_unique_id(id));
        if (world > 1) {
            char tmp[4096];
            snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
            FILE* f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fwrite(want, 1, sizeof want, f) != sizeof want) { fprintf(stderr, "cannot write %s\n", tmp); return 73; }
            fclose(f);
            if (rename(tmp, id_file) != 0) { fprintf(stderr, "cannot publish %s\n", id_file); return 73; }
        }
    } else {
        const double t_end = now_s() + 120.0;
        for (;;) {
            FILE* f = fopen(id_file, "rb");
            if (f) {
                char got[64];
                const size_t n = fread(id, 1, sizeof id, f), k = fread(got, 1, sizeof got, f);
                fclose(f);
                if (n == sizeof id && k == sizeof got && memcmp(got, want, sizeof want) == 0) break;      /* else: stale or half-written */
            }
            if (now_s() > t_end) { fprintf(stderr, "[rank %d] no rendezvous id for this launch in %s after 120 s\n", rank, id_file); return 75; }
            struct timespec nap = {0, 20 * 1000 * 1000};
            nanosleep(&nap, NULL);
        }
    }
    mrcnn_dist* dist = NULL;
    CHECK(mrcnn_dist_init(rank, world, id, &dist));
    if (rank == 0 && world > 1) (void)remove(id_file);        /* every rank has joined: the id is spent */

    /* ---- every rank loads the same artefacts (replicated weights, ~250 MB) ---- */
    char path[4][4096];
    snprintf(path[0], sizeof path[0], "%s/anchors.bin", dir);
    snprintf(path[1], sizeof path[1], "%s/Classifier.mrcw", dir);
    snprintf(path[2], sizeof path[2], "%s/Mask.mrcw", dir);
    snprintf(path[3], sizeof path[3], "%s/MaskRCNN.mrcw", dir);
    CHECK(mrcnn_config_set_anchors_path(path[0]));
    CHECK(mrcnn_config_set_classifier_path(path[1]));
    CHECK(mrcnn_config_set_mask_path(path[2]));
    int lo = 0, hi = 0;
    CHECK(mrcnn_dist_shard(batch, world, rank, &lo, &hi));
    mrcnn_model* model = NULL;
    CHECK(mrcnn_model_load(MRCNN_MODEL_MASKRCNN, path[3], hi - lo > 0 ? hi - lo : 1, dtype, &model));
    int64_t H = 0, W = 0, max_det = 0, mask_size = 0;
    CHECK(mrcnn_model_get_int(model, "image_height", &H));
    CHECK(mrcnn_model_get_int(model, "image_width", &W));
    CHECK(mrcnn_model_get_int(model, "max_detections", &max_det));
    CHECK(mrcnn_model_get_int(model, "mask_size", &mask_size));

    const size_t n_img = (size_t)batch * (size_t)H * (size_t)W * 3u;
    const size_t n_det = (size_t)batch * (size_t)max_det * 6u, n_mask = (size_t)batch * (size_t)max_det * (size_t)(mask_size * mask_size);
    uint8_t* images = (uint8_t*)malloc(n_img);
    float* det = (float*)malloc(sizeof(float) * n_det);
    float* masks = (float*)malloc(sizeof(float) * n_mask);
    mrcnn_detection* recs = (mrcnn_detection*)malloc(sizeof(mrcnn_detection) * (size_t)max_det);
    if (!images || !det || !masks || !recs) { fprintf(stderr, "out of memory\n"); return 70; }
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(images, 1, n_img, f) != n_img) { fprintf(stderr, "%s: cannot read %zu bytes\n", argv[2], n_img); return 66; }
    fclose(f);

    /* ---- the step: this rank predicts images [lo, hi), one all-gather hands everyone the whole batch ---- */
    const double t0 = now_s();
    CHECK(mrcnn_maskrcnn_predict_sharded(dist, model, images, batch, (int)H, (int)W, MRCNN_HOST, det, masks));
    const double t1 = now_s();

    if (rank == 0) {
        printf("world %d batch %d seconds %.6f\n", world, batch, t1 - t0);
        for (int b = 0; b < batch; ++b) {
            int64_t n = 0;
            CHECK(mrcnn_detections_decode(det + (size_t)b * (size_t)max_det * 6u, max_det, 6, recs, max_det, &n));
            printf("image %d detections %lld\n", b, (long long)n);
            for (int64_t i = 0; i < n; ++i) {
                const float* m = masks + ((size_t)b * (size_t)max_det + (size_t)recs[i].index) * (size_t)(mask_size * mask_size);
                double sum = 0.0;
                for (int k = 0; k < (int)(mask_size * mask_size); ++k) sum += (double)m[k];
                printf("%lld %lld %.17g %.17g %.17g %.17g %.17g %.17g\n", (long long)recs[i].index, (long long)recs[i].class_id,
                       recs[i].score, recs[i].x, recs[i].y, recs[i].w, recs[i].h, sum);
            }
        }
    }
    mrcnn_model_destroy(model);
    mrcnn_dist_destroy(dist);
    free(images); free(det); free(masks); free(recs);
    return 0;
}
