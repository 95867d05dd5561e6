//! `oics::fft` (reference: packages/lib/src/fft.rs) -> omr_get_fft_image / omr_get_angle_with_fft.
use crate::bridge::{check, into_mat, view};
use crate::ffi;
use crate::transfer::TransformableMatrix;
use opencv::core::{Mat, Vector};
use opencv::imgcodecs;
use opencv::prelude::*;
use std::path::Path;

/// fft.rs:32: 1 - filter, element-wise (float Mat helper of the reference's filter experiments; host OpenCV).
pub fn rev(filter: &Mat) -> opencv::Result<Mat> {
    let mut neg = Mat::default();
    filter.convert_to(&mut neg, -1, -1.0, 1.0)?; // -1 * x + 1
    Ok(neg)
}

/// fft.rs:124-141 -> (magnitude_image, magnitude_log_image), both 8-bit single channel.
pub fn get_fft_image(gray_tm: &TransformableMatrix) -> opencv::Result<(Mat, Mat)> {
    let (mut mag, mut mag_log) = (ffi::OmrImageOwned::empty(), ffi::OmrImageOwned::empty());
    check(unsafe { ffi::omr_get_fft_image(&view(gray_tm.get_mat())?, &mut mag, &mut mag_log) })?;
    let m = into_mat(mag);
    let l = into_mat(mag_log);
    Ok((m?, l?))
}

/// fft.rs:145-256: spectrum picture -> Canny(t1, t2) -> HoughLinesP(threshold 100) -> the reference's vote
/// (including its fft.rs:231 quirk), all on the GPU.  The debug picture (the log spectrum) is written when
/// `edge_image_output_dir` is not empty, as in the reference.
pub fn get_angle_with_fft(
    gray_tm: &TransformableMatrix,
    canny_threshold_1: f64,
    canny_threshold_2: f64,
    min_line_length: f64,
    max_line_gap: f64,
    file_name: &str,
    edge_image_output_dir: &str,
) -> Result<f64, opencv::Error> {
    let v = view(gray_tm.get_mat())?;
    let mut angle = 0.0f64;
    check(unsafe { ffi::omr_get_angle_with_fft(&v, canny_threshold_1, canny_threshold_2, min_line_length, max_line_gap, &mut angle) })?;
    if !edge_image_output_dir.is_empty() {
        let (_, log_pic) = get_fft_image(gray_tm)?;
        let path = Path::new(edge_image_output_dir).join(file_name);
        let params: Vector<i32> = Vector::from_slice(&[imgcodecs::IMWRITE_JPEG_QUALITY, 100]);
        imgcodecs::imwrite(path.to_str().unwrap_or(file_name), &log_pic, &params)?;
    }
    Ok(angle)
}
