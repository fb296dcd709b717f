/* CPU restatement of the reference's native box NMS (oracle; TEST INFRASTRUCTURE ONLY).
 *
 * The reference's own native code cannot be built here (SURVEY.md §8 c3:
 * cpu_nms.c is Cython-0.29 output that fails against numpy 2.2 headers,
 * nms_kernel.cu needs nvcc), so these functions restate it and are pinned
 * against the importable numpy `nms` (lib/nms/nms.py:35-72) via
 * tests/golden/nms.json on inputs where the three semantics coincide, plus
 * hand-built IoU == thresh cases that pin the > / >= split.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float fmaxf_(float a, float b) { return a >= b ? a : b; }
static inline float fminf_(float a, float b) { return a <= b ? a : b; }

/* lib/nms/nms_kernel.cu:23-31 (devIoU), all fp32, "+1" pixel convention */
static float dev_iou(const float *a, const float *b) {
    float left = fmaxf_(a[0], b[0]), right = fminf_(a[2], b[2]);
    float top = fmaxf_(a[1], b[1]), bottom = fminf_(a[3], b[3]);
    float width = fmaxf_(right - left + 1, 0.f), height = fmaxf_(bottom - top + 1, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
    float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
    return interS / (Sa + Sb - interS);
}

/* lib/nms/nms_kernel.cu:33-77 + :90-143.  boxes: [n,5] fp32 sorted by score desc.
 * mask_out (optional, n*col_blocks uint64) receives the 64-wide bitmask exactly
 * as the device kernel writes it (strict > vs fp32 thresh; diagonal tiles start at j>i). */
int oracle_gpu_nms(int *keep_out, int *num_out, const float *boxes, int n, float thresh,
                   uint64_t *mask_out) {
    int col_blocks = (n + 63) / 64;
    uint64_t *mask = (uint64_t *)calloc((size_t)n * col_blocks + 1, sizeof(uint64_t));
    uint64_t *remv = (uint64_t *)calloc(col_blocks + 1, sizeof(uint64_t));
    if (!mask || !remv) return -1;
    for (int i = 0; i < n; i++)
        for (int cb = 0; cb < col_blocks; cb++) {
            int csize = n - cb * 64 < 64 ? n - cb * 64 : 64;
            int start = (i / 64 == cb) ? (i % 64) + 1 : 0;
            uint64_t t = 0;
            for (int j = start; j < csize; j++)
                if (dev_iou(boxes + 5 * i, boxes + 5 * (cb * 64 + j)) > thresh) t |= 1ULL << j;
            mask[(size_t)i * col_blocks + cb] = t;
        }
    int k = 0;
    for (int i = 0; i < n; i++) {                       /* :126-138 greedy OR-reduce */
        int nb = i / 64, ib = i % 64;
        if (!(remv[nb] & (1ULL << ib))) {
            keep_out[k++] = i;
            for (int j = nb; j < col_blocks; j++) remv[j] |= mask[(size_t)i * col_blocks + j];
        }
    }
    *num_out = k;
    if (mask_out) memcpy(mask_out, mask, (size_t)n * col_blocks * sizeof(uint64_t));
    free(mask); free(remv);
    return 0;
}

/* lib/nms/cpu_nms.pyx:20-71.  dets [n,5] fp32 (unsorted), order = argsort(scores)[::-1]
 * computed by the caller with numpy (as the .pyx does).  IoU in fp32, the test is
 * `ovr >= thresh` with thresh a Python float (double) -> compare in double. */
int oracle_cpu_nms(int *keep_out, int *num_out, const float *dets, const int *order, int n,
                   double thresh) {
    char *sup = (char *)calloc(n + 1, 1);
    float *areas = (float *)malloc(sizeof(float) * (n + 1));
    if (!sup || !areas) return -1;
    for (int i = 0; i < n; i++)
        areas[i] = (dets[5 * i + 2] - dets[5 * i] + 1) * (dets[5 * i + 3] - dets[5 * i + 1] + 1);
    int k = 0;
    for (int _i = 0; _i < n; _i++) {
        int i = order[_i];
        if (sup[i]) continue;
        keep_out[k++] = i;
        float ix1 = dets[5 * i], iy1 = dets[5 * i + 1], ix2 = dets[5 * i + 2], iy2 = dets[5 * i + 3];
        float iarea = areas[i];
        for (int _j = _i + 1; _j < n; _j++) {
            int j = order[_j];
            if (sup[j]) continue;
            float xx1 = fmaxf_(ix1, dets[5 * j]), yy1 = fmaxf_(iy1, dets[5 * j + 1]);
            float xx2 = fminf_(ix2, dets[5 * j + 2]), yy2 = fminf_(iy2, dets[5 * j + 3]);
            float w = fmaxf_(0.0f, xx2 - xx1 + 1), h = fmaxf_(0.0f, yy2 - yy1 + 1);
            float inter = w * h;
            float ovr = inter / (iarea + areas[j] - inter);
            if ((double)ovr >= thresh) sup[j] = 1;
        }
    }
    *num_out = k;
    free(sup); free(areas);
    return 0;
}
