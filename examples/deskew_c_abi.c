/*
 * deskew_c_abi.c -- the drop-in boundary used from plain C: no torch, no Python, only
 * include/omrdeskew.h and libomrdeskew.so.  Builds a synthetic skewed sheet, asks for its angle through
 * the reference's two drivers and deskews it with correct_default (omr.rs:339-448 minus the codecs).
 *
 *   gcc -std=c11 -O2 -Iinclude examples/deskew_c_abi.c -o /tmp/deskew_c_abi \
 *       -Lomr-img-corrector_amd/lib -lomrdeskew -Wl,-rpath,$PWD/omr-img-corrector_amd/lib -lm
 *   /tmp/deskew_c_abi            # needs a HIP device: without one every call returns -217
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "omrdeskew.h"

static void draw_sheet(uint8_t *bgr, int rows, int cols, double skew_deg)
{
    /* white page with black bars, rotated by -skew about the centre (nearest neighbour is enough here) */
    const double a = skew_deg * 3.14159265358979323846 / 180.0, c = cos(a), s = sin(a);
    memset(bgr, 255, (size_t)rows * cols * 3);
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const double u = (x - cols / 2.0) * c - (y - rows / 2.0) * s + cols / 2.0;
            const double v = (x - cols / 2.0) * s + (y - rows / 2.0) * c + rows / 2.0;
            const int bar = ((int)floor(v) / 24) % 3 == 0 && u > cols * 0.1 && u < cols * 0.9 && v > rows * 0.1 && v < rows * 0.9;
            if (bar) memset(bgr + ((size_t)y * cols + x) * 3, 0, 3);
        }
}

int main(void)
{
    printf("libomrdeskew version %d, %d HIP device(s)\n", omr_version(), omr_device_count());
    const int rows = 1150, cols = 1240;
    uint8_t *bgr = (uint8_t *)malloc((size_t)rows * cols * 3);
    if (!bgr) return 2;
    draw_sheet(bgr, rows, cols, 3.4);
    omr_image img = {bgr, rows, cols, 3, (int64_t)cols * 3};

    double angle = 0;
    int rc = omr_get_angle_with_projections(&img, 10, 0.2, 0.2, 1, &angle); /* projection.rs:17-23 */
    if (rc) {
        printf("omr_get_angle_with_projections: %d (%s)\n", rc, omr_last_error());
        free(bgr);
        return rc == OMR_ERR_GPU ? 0 : 1; /* no GPU here: that is the documented behaviour */
    }
    printf("get_angle_with_projections        : %.2f deg\n", angle);

    int32_t status = 0, n_cand = 0;
    double cand[1024];
    rc = omr_get_result_from_projection(&img, 45, 0.2, 248, 230, &angle, &status, cand, 1024, &n_cand); /* omr.rs:52-229 */
    printf("get_result_from_projection        : rc %d, %.2f deg, status %d, %d candidate(s)\n", rc, angle, status, n_cand);

    int32_t need_check = 0;
    omr_image_owned out = {0};
    rc = omr_correct_default(&img, 45, 0.2, 248, 230, 150.0, 50.0, &angle, &need_check, &out); /* omr.rs:339-448 */
    printf("correct_default                   : rc %d, rotate by %.2f deg, need_check %d, canvas %dx%d\n", rc, angle,
           need_check, out.cols, out.rows);
    omr_image_free(&out);
    free(bgr);
    return rc ? 1 : 0;
}
