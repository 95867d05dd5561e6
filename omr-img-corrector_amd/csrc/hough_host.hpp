// hough_host.hpp -- host-side building blocks of the Hough-line path (oics_hough.cpp), shared with
// the FFT path (oics_fft.cpp), which ends in the same Canny -> HoughLinesP -> vote chain.
#pragma once
#include <stdint.h>

#include <memory>
#include <vector>

#include "../../include/omrdeskew.h"
#include "engine.hpp"

namespace omr {
namespace hh {

struct HStream {
    hipStream_t s = nullptr;
    CallSlot *slot = nullptr;         // leased for the call: stream + pinned staging (engine.cpp)
    std::unique_ptr<PoolScope> pool;  // the call's device buffers come from / return to the block cache
    ~HStream();
    int create();
};

struct HoughParams {
    double low = 50.0, high = 150.0;  // hough.rs:27, omr.rs:239
    double rho = 1.0, theta = 3.14159265358979323846 / 180.0;
    int threshold = 0;
    double min_line_length = 0, max_line_gap = 0;
};

int have_device();
int check_img(const omr_image *im);
int upload(const omr_image *im, DevBuf *buf, hipStream_t s);
// Canny on n device-resident scans of one shape -> d_map holds the edges (0 / 255), packed;
// d_rowcnt (n x rows) receives the edge pixels per row (what ppht_device starts from)
int canny_device(const uint8_t *d_src, int64_t scan_stride, int64_t step, int rows, int cols, int cn, int n, double low_t,
                 double high_t, uint8_t *d_map, int *d_flag, hipStream_t s, int32_t *d_rowcnt);
// HoughLinesP on n device-resident edge images (packed)
int ppht_device(uint8_t *d_edges, int32_t *d_rowcnt, int rows, int cols, int n, const HoughParams &hp, hipStream_t s,
                std::vector<std::vector<int32_t>> *lines_out);
int edges_lines_device(const uint8_t *d_src, int64_t scan_stride, int64_t step, int rows, int cols, int cn, int n,
                       const HoughParams &hp, hipStream_t s, std::vector<std::vector<int32_t>> *lines);
void line_angles(const std::vector<int32_t> &l, std::vector<float> *ang);
int vote_counts(const std::vector<float> &ang, bool as_f64, hipStream_t s, std::vector<int32_t> *counts);
int select_omr_rs(const std::vector<float> &ang, const std::vector<int32_t> &cnt, double *angle, int32_t *status,
                  double *candidates, int32_t cand_cap, int32_t *cand_len);

}  // namespace hh
// defined in oics_host.cpp: path 2's projection result (omr.rs:52-229) and rotate_mat (transfer.rs:459-523) on a
// packed image that is already on the device -- correct_default uploads its sheet once
int result_from_projection_device(const uint8_t *d_src, int rows, int cols, int cn, uint16_t max_angle, double step,
                                  int32_t max_w, int32_t max_h, hipStream_t s, double *angle, int32_t *status,
                                  double *candidates, int32_t cand_cap, int32_t *cand_len);
int rotate_device_to_host(const uint8_t *d_src, int rows, int cols, int cn, double angle_deg, double scale, int interp,
                          const uint8_t border_value[4], int clip, hipStream_t s, omr_image_owned *dst);
}  // namespace omr
