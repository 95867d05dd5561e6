/*
 * oracle_hough.h -- CPU restatement of the reference's Hough-line deskew path.
 * TEST INFRASTRUCTURE ONLY and "parity unpinned" exactly as oracle.h states; see oracle_hough.c for
 * the upstream files restated and the reference call sites.
 */
#ifndef ORC_ORACLE_HOUGH_H
#define ORC_ORACLE_HOUGH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Sobel(src, CV_16S, 1/0 and 0/1, ksize 3, BORDER_REPLICATE); dx, dy: rows x cols x cn, packed. */
void orc_sobel3_16s(const uint8_t *src, int rows, int cols, int cn, int64_t sstep, int16_t *dx, int16_t *dy);

/* Canny(src, dst, low, high, 3, false); cn 1, 3 or 4 (per pixel the channel with the largest
 * |dx| + |dy|).  dst: 0 / 255.  hough.rs:27, omr.rs:239, omr.rs:323-330. */
int orc_canny(const uint8_t *src, int rows, int cols, int cn, int64_t sstep, double low_thresh, double high_thresh,
              uint8_t *dst, int64_t dstep);

/* numangle = cvRound(pi / (float)theta) and trigtab[2n] = (float)(cos(n theta) / rho), [2n+1] = sin. */
void orc_hough_trigtab(double theta, double rho, int *numangle, float *trigtab);

/* HoughLinesP (progressive probabilistic Hough, RNG seed (uint64)-1).  lines: cap x (x0,y0,x1,y1);
 * *n_lines receives the number found (may exceed cap: call again with a larger buffer). */
int orc_hough_lines_p(const uint8_t *image, int height, int width, int64_t step, double rho, double theta,
                      int threshold, double min_line_length, double max_line_gap, int32_t *lines, int cap,
                      int *n_lines);

/* counters of this thread's last orc_hough_lines_p call: points, walks, points cleared, un-voted */
void orc_hough_last_stats(int64_t out[4]);

/* hough.rs:50-68: atan2 in f32, degrees, "% 45.0". */
float orc_line_angle_f32(const int32_t l[4]);
/* hough.rs:70-92 */
int orc_vote_hough_rs(const float *angles, int n, double *angle_out);
/* omr.rs:268-301; status 0 Believed, 1 NeedCheck, 2 NotAResult */
int orc_vote_omr_rs(const float *angles32, int n, double *angle_out, int *status, double *candidates, int cand_cap,
                    int *cand_len);

/* hough.rs:17-100 without the debug image.  -215 when no line is found (the reference panics). */
int orc_get_angle_with_hough(const uint8_t *gray, int rows, int cols, int cn, int64_t step, double min_line_length,
                             double max_line_gap, double *angle_out, int *n_lines_out);
/* omr.rs:231-302 */
int orc_get_result_from_edges_detection(const uint8_t *src, int rows, int cols, int cn, int64_t step,
                                        double min_line_length, double max_line_gap, double *angle, int *status,
                                        double *candidates, int cand_cap, int *cand_len, int *n_lines_out);
/* omr.rs:351-399 */
void orc_correct_default_decision(double proj_angle, int proj_status, const double *proj_candidates, int n_cand,
                                  double edges_angle, double *rotate_angle, int *need_check);

#ifdef __cplusplus
}
#endif
#endif
